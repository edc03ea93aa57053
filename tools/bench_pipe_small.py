import os, sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import gpv1_amd.hip as hip
from bench_attn import timeit
dev = 'cuda'
print('GPV_PIPE_SMALL', os.environ.get('GPV_PIPE_SMALL'))
for (N, K) in [(768, 768), (2304, 768), (2048, 768), (768, 2048), (10000, 768), (256, 256), (256, 2048), (768, 3072), (3072, 768)]:
    row = 'N=%5d K=%4d ' % (N, K)
    for M in (16, 32, 64, 100, 128, 192, 320, 640):
        A = torch.randn(M, K, device=dev).to(torch.bfloat16); B = torch.randn(N, K, device=dev).to(torch.bfloat16)
        C = torch.empty(M, N, device=dev, dtype=torch.bfloat16); b = torch.randn(N, device=dev)
        row += ' M%4d %5.1f' % (M, timeit(lambda: hip.gemm(A, B, C, M, N, K, K, K, N, bias=b)))
    print(row, flush=True)
