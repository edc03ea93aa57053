"""GPU micro-benchmark of the implicit-GEMM conv / GEMM kernel on the ResNet-50 shapes of the bench workload.
usage: [GPV_HIP_LIB=path/to/lib.so] python tools/bench_conv.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpv1_amd.hip as hip
B = 32
SHAPES = [  # name, Cin, Cout, k, s, p, H, W, with_res
    ('l1.c1', 256, 64, 1, 1, 0, 120, 160, 0), ('l1.c2', 64, 64, 3, 1, 1, 120, 160, 0), ('l1.c3', 64, 256, 1, 1, 0, 120, 160, 1),
    ('l2.c1', 512, 128, 1, 1, 0, 60, 80, 0), ('l2.c2', 128, 128, 3, 1, 1, 60, 80, 0), ('l2.c3', 128, 512, 1, 1, 0, 60, 80, 1),
    ('l3.c1', 1024, 256, 1, 1, 0, 30, 40, 0), ('l3.c2', 256, 256, 3, 1, 1, 30, 40, 0), ('l3.c3', 256, 1024, 1, 1, 0, 30, 40, 1),
    ('l4.c1', 2048, 512, 1, 1, 0, 15, 20, 0), ('l4.c2', 512, 512, 3, 1, 1, 15, 20, 0), ('l4.c3', 512, 2048, 1, 1, 0, 15, 20, 1),
    ('l3.c2s2', 256, 256, 3, 2, 1, 60, 80, 0)]
dev = 'cuda'
tot = 0
for name, ci, co, k, s, p, H, W, wr in SHAPES:
    OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    x = torch.randn(B, H, W, ci, device=dev).to(torch.bfloat16)
    w = (torch.randn(co, k * k, ci, device=dev) / (ci * k * k) ** 0.5).to(torch.bfloat16)
    y = torch.empty(B, OH, OW, co, device=dev, dtype=torch.bfloat16)
    bias = torch.randn(co, device=dev)
    res = torch.randn(B, OH, OW, co, device=dev).to(torch.bfloat16) if wr else None
    def run():
        hip.conv2d(0, x, w, y, B, H, W, ci, ci, OH, OW, co, k, k, s, s, p, p, bias=bias, res=res, act=1)
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    fl = 2.0 * B * OH * OW * co * k * k * ci
    by = (x.numel() + y.numel() * (2 if wr else 1) + w.numel()) * 2
    tot += us
    print('%-8s M=%7d N=%4d K=%4d  %7.1f us  %6.1f TF/s  %6.0f GB/s' % (name, B * OH * OW, co, k * k * ci, us, fl / us / 1e6, by / us / 1e3))
print('sum us', tot)
print('--- wgrad (layer2-4 shapes) / linear wgrad')
tot = 0
for name, ci, co, k, s, p, H, W, wr in SHAPES[3:]:
    OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    x = torch.randn(B, H, W, ci, device=dev).to(torch.bfloat16)
    dy = torch.randn(B, OH, OW, co, device=dev).to(torch.bfloat16)
    dw = torch.zeros(co, k * k, ci, device=dev)
    sc = torch.ones(co, device=dev)
    def run():
        hip.conv2d(2, x, dy, dw, B, H, W, ci, ci, OH, OW, co, k, k, s, s, p, p, rowscale=sc)
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    fl = 2.0 * B * OH * OW * co * k * k * ci
    tot += us
    print('%-8s wgrad out %4dx%5d red %7d  %7.1f us  %6.1f TF/s' % (name, co, k * k * ci, B * OH * OW, us, fl / us / 1e6))
print('sum wgrad us', tot)
for (M, N, K) in [(9600, 512, 256), (9600, 256, 256), (9600, 2048, 256), (9600, 256, 2048), (3200, 768, 768), (3200, 3072, 768), (608, 10000, 768)]:
    dy = torch.randn(M, N, device=dev).to(torch.bfloat16); x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    dw = torch.zeros(N, K, device=dev)
    tiles = ((N + 63) // 64) * ((K + 63) // 64); kt = (M + 31) // 32
    split = max(1, min(kt // 8, 1024 // max(tiles, 1)))
    def run():
        hip.gemm(dy, x, dw, N, K, M, N, K, K, layoutA=hip.TRANS, layoutB=hip.TRANS, accumulate=True, split_k=split)
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print('linear wgrad out %5dx%4d red %5d split %3d  %7.1f us  %6.1f TF/s' % (N, K, M, split, us, 2.0 * M * N * K / us / 1e6))
