"""the K <= 512 1x1 convolutions of layer1 / layer2 at B=32 (HBM-streaming launches), forward (bias [+ residual] + ReLU) and
backward-data (ReLU mask [+ residual]); algorithmic TB/s.  usage: [GPV_EPF=0] python tools/bench_c1.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpv1_amd.hip as hip
from bench_attn import timeit
B, dev = 32, 'cuda'
SH = [('l1.0c1', 64, 64, 120, 160, 0), ('l1.c1', 256, 64, 120, 160, 0), ('l1.c3', 64, 256, 120, 160, 1), ('l1.0ds', 64, 256, 120, 160, 0),
      ('l2.0c1', 256, 128, 120, 160, 0), ('l2.c1', 512, 128, 60, 80, 0), ('l2.c3', 128, 512, 60, 80, 1), ('l3.c3', 256, 1024, 30, 40, 1)]
tot = [0.0, 0.0]
for name, ci, co, H, W, wr in SH:
    M = B * H * W
    x = torch.randn(M, ci, device=dev).to(torch.bfloat16); w = (torch.randn(co, 1, ci, device=dev) / ci ** 0.5).to(torch.bfloat16)
    y = torch.empty(M, co, device=dev, dtype=torch.bfloat16); bias = torch.randn(co, device=dev)
    res = torch.randn(M, co, device=dev).to(torch.bfloat16) if wr else None
    f = lambda: hip.conv2d(0, x, w, y, B, H, W, ci, ci, H, W, co, 1, 1, 1, 1, 0, 0, bias=bias, res=res, act=1)
    tf = timeit(f, 30)
    bf = (M * ci + M * co * (2 if wr else 1)) * 2
    # dgrad of this conv: dy [M, co] -> dx [M, ci], masked by the ReLU of its input (+ the identity gradient for a block input)
    dy = torch.randn(M, co, device=dev).to(torch.bfloat16); wd = (torch.randn(ci, 1, co, device=dev) / co ** 0.5).to(torch.bfloat16)
    dx = torch.empty(M, ci, device=dev, dtype=torch.bfloat16); msk = torch.randn(M, ci, device=dev).to(torch.bfloat16)
    rs = torch.randn(M, ci, device=dev).to(torch.bfloat16) if not wr else None          # conv1 of a block feeds the residual sum
    d = lambda: hip.conv2d(1, dy, wd, dx, B, H, W, co, co, H, W, ci, 1, 1, 1, 1, 0, 0, res=rs, relu_mask=msk)
    td = timeit(d, 30)
    bd = (M * co + M * ci * (3 if rs is not None else 2)) * 2
    tot[0] += tf; tot[1] += td
    print('%-7s %4d->%4d  fwd %6.1f us %5.2f TB/s   dgrad %6.1f us %5.2f TB/s' % (name, ci, co, tf, bf / tf / 1e6, td, bd / td / 1e6), flush=True)
print('sums: fwd %.1f us, dgrad %.1f us' % tuple(tot))
