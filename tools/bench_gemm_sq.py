"""plain bf16 GEMM throughput of gpv_gemm at square and layer-like shapes (C = A[M,K] . B[N,K]^T)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpv1_amd.hip as hip
dev = 'cuda'
for M, N, K in [(4096, 4096, 4096), (8192, 8192, 8192), (9600, 2048, 256), (9600, 256, 2048), (9600, 256, 256), (3200, 768, 768), (3200, 3072, 768), (192, 768, 768), (192, 768, 3072), (640, 768, 768), (38400, 256, 2304), (153600, 128, 1152)]:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16); B = torch.randn(N, K, device=dev).to(torch.bfloat16)
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    def run(): hip.gemm(A, B, C, M, N, K, K, K, N)
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20 if M * N * K < 1e11 else 5
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    ref = None
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    for _ in range(2): torch.matmul(A, B.t())
    torch.cuda.synchronize(); t0.record()
    for _ in range(n): torch.matmul(A, B.t())
    t1.record(); torch.cuda.synchronize()
    us_t = t0.elapsed_time(t1) * 1e3 / n
    print('M=%6d N=%5d K=%5d  ours %8.1f us %6.1f TF/s | hipBLASLt (torch.matmul) %8.1f us %6.1f TF/s' % (M, N, K, us, 2.0 * M * N * K / us / 1e6, us_t, 2.0 * M * N * K / us_t / 1e6))
