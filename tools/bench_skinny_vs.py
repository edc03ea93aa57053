"""plain-GEMM equivalents of the layer3/4 conv shapes: 4-wave kernel vs skinny (in-block k-split) vs 8-wave glds"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpv1_amd.hip as hip
dev = 'cuda'
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for M, N, K in [(9600, 512, 4608), (9600, 512, 2048), (9600, 2048, 512), (9600, 2048, 1024), (38400, 256, 2304), (38400, 256, 1024), (38400, 1024, 256), (38400, 1024, 512), (38400, 512, 1024)]:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16); B = torch.randn(N, K, device=dev).to(torch.bfloat16); C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    run = lambda: hip.gemm(A, B, C, M, N, K, K, K, N)
    res = []
    for sk, gl in ((0, 0), (2, 0), (0, 2)):
        hip.set_option(hip.OPT_SKINNY, sk); hip.set_option(hip.OPT_GLDS, gl)
        res.append(t(run))
    fl = 2.0 * M * N * K
    print('M=%6d N=%5d K=%5d  4-wave %6.1f us (%4.0f TF/s)  skinny %6.1f us (%4.0f)  glds %6.1f us (%4.0f)' % (M, N, K, res[0], fl / res[0] / 1e6, res[1], fl / res[1] / 1e6, res[2], fl / res[2] / 1e6))
