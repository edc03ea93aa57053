"""<= 1024-row GEMMs: the 96-row two-per-CU rule of gemm_common.h (GPV_BM96 in the tuning build) on / off over M"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gpv1_amd.hip as hip
from bench_attn import timeit
dev = 'cuda'
print('GPV_BM96', os.environ.get('GPV_BM96'))
for (N, K) in [(768, 768), (2304, 768), (2048, 768), (768, 2048), (256, 256), (256, 2048), (768, 3072), (3072, 768)]:
    row = 'N=%4d K=%4d ' % (N, K)
    for M in (100, 192, 300, 320, 384, 512, 640, 800, 1024):
        A = torch.randn(M, K, device=dev).to(torch.bfloat16); B = torch.randn(N, K, device=dev).to(torch.bfloat16)
        C = torch.empty(M, N, device=dev, dtype=torch.bfloat16); b = torch.randn(N, device=dev)
        row += ' M%4d %5.1f' % (M, timeit(lambda: hip.gemm(A, B, C, M, N, K, K, K, N, bias=b)))
    print(row, flush=True)
