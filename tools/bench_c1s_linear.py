"""K = 256 -> 2048 / 1536 linear GEMMs on the streaming kernel (conv1x1_stream.hip as a gpv_gemm path) against the tile kernels, one process:
usage: python tools/bench_c1s_linear.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpv1_amd.hip as hip
from bench_attn import timeit
dev = 'cuda'
for M, F in ((9600, 2048), (3200, 2048), (9600, 1536), (9600, 1024)):
    D = 256
    x = torch.randn(M, D, device=dev).to(torch.bfloat16); w1 = (torch.randn(F, D, device=dev) / 16).to(torch.bfloat16)
    w2t = (torch.randn(F, D, device=dev) / 45).to(torch.bfloat16); b1 = torch.randn(F, device=dev)
    h = torch.empty(M, F, device=dev, dtype=torch.bfloat16); dy = torch.randn(M, D, device=dev).to(torch.bfloat16); dz = torch.empty_like(h)
    hm = torch.relu(torch.randn(M, F, device=dev)).to(torch.bfloat16)
    runs = {'bias': lambda: hip.gemm(x, w1, h, M, F, D, D, D, F, bias=b1),
            'bias+relu+dropout': lambda: hip.gemm(x, w1, h, M, F, D, D, D, F, bias=b1, act=hip.ACT_RELU, drop_p=0.1, seed=3),
            'dz = dy W2^T(mirror) * mask * alpha': lambda: hip.gemm(dy, w2t, dz, M, F, D, D, D, F, relu_mask=hm, ldm=F, alpha=1.0 / 0.9)}
    for name, run in runs.items():
        row = '%5d x %4d  %-36s' % (M, F, name)
        for mode in (0, 1):
            hip.set_option(hip.OPT_C1S, mode)
            hip.set_option(hip.OPT_C1S_LAUNCHES, 0)
            run()
            used = hip.set_option(hip.OPT_C1S_LAUNCHES, 0)
            row += '  %s %6.1f us' % ('streaming' if used else 'tiles    ', timeit(run))
        hip.set_option(hip.OPT_C1S, 1)
        print(row, flush=True)
