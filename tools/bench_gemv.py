"""Few-row GEMM kernel (csrc/gemv.hip) against the tile kernels on the shapes of the B = 1 inference (decode step: M = 1; BERT with six
query tokens: M = 6): us per launch, 200 back-to-back launches between HIP events."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import gpv1_amd.hip as h
h.lib()
dev = 'cuda'
def t(f, n=200):
    for _ in range(10): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
print('%4s %6s %6s   %8s %8s' % ('M', 'N', 'K', 'few-row', 'tiles'))
for M in (1, 2, 4, 6, 8):
    for N, K in ((768, 768), (2304, 768), (3072, 768), (768, 3072), (2048, 768), (768, 2048), (10000, 768)):
        A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(N, K, device=dev).bfloat16()
        Cm = torch.empty(M, N, device=dev, dtype=torch.bfloat16); bias = torch.randn(N, device=dev)
        f = lambda: h.gemm(A, B, Cm, M, N, K, K, K, N, bias=bias)
        h.set_option(h.OPT_GEMV, 1); a = t(f)
        h.set_option(h.OPT_GEMV, 0); b = t(f)
        h.set_option(h.OPT_GEMV, 1)
        print('%4d %6d %6d   %8.2f %8.2f' % (M, N, K, a, b))
