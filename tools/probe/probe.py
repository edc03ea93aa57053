import ctypes, torch, os
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libtrread.so'))
def run(addr):
    a = torch.tensor(addr, dtype=torch.int32, device='cuda')
    out = torch.zeros(256, dtype=torch.int16, device='cuda')
    e = lib.run_probe(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    return out.view(64, 4).cpu().tolist()
# pattern 1: every lane its own 8-byte slot: addr = lane*8
r = run([l * 8 for l in range(64)])
print('addr=lane*8:'); [print(l, r[l]) for l in (0, 1, 2, 3, 4, 5, 15, 16, 17, 31, 32, 48, 63)]
# pattern 2: row-major [k][n] tile with pitch 64 elements (128 B): lane i of a 16-group -> row (i>>2), col chunk (i&3)*4 ; group g -> rows 4g..
P = 64
r = run([(((l >> 4) * 4 + ((l & 15) >> 2)) * P + (l & 3) * 4) * 2 for l in range(64)])
print('row-major pitch 64, group g reads rows 4g..4g+3, cols 0..15:'); [print(l, r[l]) for l in (0, 1, 2, 3, 4, 15, 16, 17, 32, 63)]
