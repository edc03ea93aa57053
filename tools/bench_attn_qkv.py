"""encoder / decoder self-attention: q|k GEMM + v GEMM + core (three launches) against gpv_attention_qkv_fwd (one launch)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gpv1_amd.hip as hip
from bench_attn import timeit
dev, dt = 'cuda', torch.bfloat16
for (B, S) in [(32, 300), (32, 100)]:
    H, dh, D = 8, 32, 256
    M = B * S
    x = torch.randn(M, D, device=dev).to(dt); xp = (x.float() + torch.randn(M, D, device=dev)).to(dt)
    w = (torch.randn(3 * D, D, device=dev) / 16).to(dt); bias = torch.randn(3 * D, device=dev)
    b_qk, b_v = bias[:2 * D].contiguous(), bias[2 * D:].contiguous()
    qk = torch.empty(M, 2 * D, device=dev, dtype=dt); v = torch.empty(M, D, device=dev, dtype=dt); o = torch.empty(M, D, device=dev, dtype=dt)
    lse = torch.empty(B, H, S, device=dev)
    st = ((S * 2 * D, 2 * D), (S * 2 * D, 2 * D), (S * D, D), (S * D, D))
    sc = dh ** -0.5
    def three():
        hip.gemm(xp, w[:2 * D], qk, M, 2 * D, D, D, D, 2 * D, bias=b_qk)
        hip.gemm(x, w[2 * D:], v, M, D, D, D, D, D, bias=b_v)
        hip.attention_fwd(qk[:, :D], qk[:, D:], v, o, st, B, H, S, S, dh, sc, drop_p=0.1, seed=5, lse=lse)
    def core():
        hip.attention_fwd(qk[:, :D], qk[:, D:], v, o, st, B, H, S, S, dh, sc, drop_p=0.1, seed=5, lse=lse)
    def one():
        hip.attention_qkv_fwd(xp, x, w, bias, qk[:, :D], qk[:, D:], v, o, st, B, H, S, sc, drop_p=0.1, seed=5, lse=lse)
    flops = 2.0 * M * D * 3 * D + 4.0 * B * H * S * S * dh
    for name, f in (('three launches', three), ('core alone', core), ('one launch', one), ('three launches', three), ('one launch', one)):
        t = timeit(f)
        print('B=%d S=%d  %-15s %6.1f us   (projections + core = %.2f GFLOP -> %.0f TFLOP/s)' % (B, S, name, t, flops / 1e9, flops / t / 1e6), flush=True)
