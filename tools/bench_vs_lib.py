"""The step's GEMM shapes: this library's kernels against torch.matmul / F.linear (hipBLASLt) on the same operands -- what a plain
library GEMM would buy where the epilogue is plain (bias only).  usage: python tools/bench_vs_lib.py        (GPU box)"""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpv1_amd.hip as hip
from bench_attn import timeit
dev = 'cuda'
hip.lib()
shapes = [('enc ffn1 9600x2048x256', 9600, 2048, 256), ('enc ffn2 9600x256x2048', 9600, 256, 2048), ('enc qk 9600x512x256', 9600, 512, 256),
          ('enc v/out 9600x256x256', 9600, 256, 256), ('dec ffn1 3200x2048x256', 3200, 2048, 256), ('dec ffn2 3200x256x2048', 3200, 256, 2048),
          ('dec kv 9600x256x256', 9600, 256, 256), ('coatt 3200x768x768', 3200, 768, 768), ('coatt ffn 3200x3072x768', 3200, 3072, 768),
          ('coatt ffn2 3200x768x3072', 3200, 768, 3072), ('text 640x768x768', 640, 768, 768), ('vocab 640x10000x768', 640, 10000, 768),
          ('roi 3200x2048x2304', 3200, 2048, 2304), ('big 8192^3', 8192, 8192, 8192)]
print('%-28s %9s %9s %9s   (us; TF/s of ours / lib)' % ('M x N x K', 'ours', 'F.linear', 'ratio'))
for name, M, N, K in shapes:
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev)
    bb = b.to(torch.bfloat16)
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t0 = timeit(lambda: hip.gemm(x, w, y, M, N, K, K, K, N, bias=b))
    t1 = timeit(lambda: F.linear(x, w, bb))
    fl = 2.0 * M * N * K
    print('%-28s %9.1f %9.1f %9.2f   %6.0f / %6.0f' % (name, t0, t1, t0 / t1, fl / t0 / 1e6, fl / t1 / 1e6), flush=True)
