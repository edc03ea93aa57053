"""the DETR feed-forward GEMMs (M = 9600 encoder / 3200 decoder rows, 256 <-> 2048) with their real epilogues, every kernel choice:
forward  h = dropout(relu(x W1^T + b1)),  y = h W2^T + b2;  backward  dz = (dy W2) * (h > 0) [* 1/(1-p)],  dx = dz W1 + res.
usage: python tools/bench_ffn.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpv1_amd.hip as hip
from bench_attn import timeit
dev = 'cuda'
CFG = ['256x128', '192x128', '128x128', '160x256', '128x256', '96x256', '64x64s6', '32x64s8']


def sweep(name, run, flops):
    hip.set_option(hip.OPT_PIPE, 0)
    t0 = timeit(run)
    row = '%-34s old %6.1f us |' % (name, t0)
    for i in range(len(CFG)):
        hip.set_option(hip.OPT_PIPE_LAUNCHES, 0)
        hip.set_option(hip.OPT_PIPE, 100 + i)
        run()
        used = hip.set_option(hip.OPT_PIPE_LAUNCHES, 0)
        row += ' %6.1f' % timeit(run) if used else '   --  '
    hip.set_option(hip.OPT_PIPE, 1)
    row += ' | auto %6.1f us' % timeit(run)
    print(row, flush=True)


for M in (9600, 3200):
    D, F = 256, 2048
    x = torch.randn(M, D, device=dev).to(torch.bfloat16); w1 = (torch.randn(F, D, device=dev) / 16).to(torch.bfloat16)
    w2 = (torch.randn(D, F, device=dev) / 45).to(torch.bfloat16); b1 = torch.randn(F, device=dev); b2 = torch.randn(D, device=dev)
    h = torch.empty(M, F, device=dev, dtype=torch.bfloat16); y = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
    dy = torch.randn(M, D, device=dev).to(torch.bfloat16); dz = torch.empty_like(h); dx = torch.empty_like(x); res = torch.randn_like(x)
    print('M = %d   configs: %s' % (M, CFG))
    sweep('ffn1 plain (bias)', lambda: hip.gemm(x, w1, h, M, F, D, D, D, F, bias=b1), 2.0 * M * F * D)
    sweep('ffn1 bias+relu', lambda: hip.gemm(x, w1, h, M, F, D, D, D, F, bias=b1, act=hip.ACT_RELU), 2.0 * M * F * D)
    sweep('ffn1 bias+relu+dropout', lambda: hip.gemm(x, w1, h, M, F, D, D, D, F, bias=b1, act=hip.ACT_RELU, drop_p=0.1, seed=3), 2.0 * M * F * D)
    sweep('ffn2 (bias)', lambda: hip.gemm(h, w2, y, M, D, F, F, F, D, bias=b2), 2.0 * M * F * D)
    sweep('dz = dy W2 * mask (TRANS B)', lambda: hip.gemm(dy, w2, dz, M, F, D, D, F, F, layoutB=hip.TRANS, relu_mask=h, ldm=F, alpha=1.0 / 0.9), 2.0 * M * F * D)
    sweep('dx = dz W1 + res (TRANS B)', lambda: hip.gemm(dz, w1, dx, M, D, F, F, D, D, layoutB=hip.TRANS, res=res, ldr=D), 2.0 * M * F * D)
