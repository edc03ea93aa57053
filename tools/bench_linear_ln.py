"""out-projection + LayerNorm: GEMM then gpv_layernorm_pos_fwd (two launches) against gpv_linear_layernorm_fwd (one launch)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gpv1_amd.hip as hip
from bench_attn import timeit
dev, dt, D = 'cuda', torch.bfloat16, 256
for rows in (9600, 3200):
    a = torch.randn(rows, D, device=dev).to(dt); x = torch.randn(rows, D, device=dev).to(dt)
    w = (torch.randn(D, D, device=dev) / 16).to(dt); bias = torch.randn(D, device=dev); gamma = torch.rand(D, device=dev) + 0.5; beta = torch.randn(D, device=dev)
    pos = torch.randn(rows // 32, D, device=dev).to(dt)
    s = torch.empty_like(a); y = torch.empty_like(a); y2 = torch.empty_like(a); m = torch.empty(rows, device=dev); r = torch.empty(rows, device=dev)
    def two():
        hip.gemm(a, w, s, rows, D, D, D, D, D, bias=bias)
        hip.layernorm_fwd(x, s, gamma, beta, y, m, r, rows, D, 1e-5, drop_p=0.1, seed=3, pos=pos, y2=y2)
    def g(): hip.gemm(a, w, s, rows, D, D, D, D, D, bias=bias)
    def one(): hip.linear_layernorm_fwd(a, w, bias, x, gamma, beta, s, y, m, r, rows, 1e-5, drop_p=0.1, seed=3, pos=pos, y2=y2)
    for name, f in (('gemm alone', g), ('two launches', two), ('one launch', one), ('two launches', two), ('one launch', one)):
        print('rows %5d  %-13s %6.1f us' % (rows, name, timeit(f)), flush=True)
