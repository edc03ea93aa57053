"""dx = dz W : W^T copy as K-major B operand vs W itself as reduction-major (GPV_TRANS) B operand"""
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
for M, N, K in [(9600, 2048, 256), (9600, 256, 2048), (9600, 256, 256), (9600, 512, 256), (3200, 768, 768), (3200, 3072, 768), (3200, 768, 3072), (640, 768, 768), (640, 2048, 768)]:
    dz = torch.randn(M, N, device=dev).to(torch.bfloat16); W = torch.randn(N, K, device=dev).to(torch.bfloat16); Wt = W.t().contiguous()
    dx1 = torch.empty(M, K, device=dev, dtype=torch.bfloat16); dx2 = torch.empty_like(dx1)
    a = t(lambda: hip.gemm(dz, Wt, dx1, M, K, N, N, N, K))
    b = t(lambda: hip.gemm(dz, W, dx2, M, K, N, N, K, K, layoutB=hip.TRANS))
    print('dz[%d,%d] W[%d,%d]:  W^T K-major %6.1f us   W reduction-major %6.1f us   maxdiff %g' % (M, N, N, K, a, b, (dx1.float() - dx2.float()).abs().max().item()))
