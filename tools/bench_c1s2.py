"""stride-2 downsample projections and the other sliced shapes, streaming kernel on / off.  usage: python tools/bench_c1s2.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpv1_amd.hip as hip
from bench_attn import timeit
B, dev = 32, 'cuda'
for name, ci, co, H, W, s in (('l2.0ds', 256, 512, 120, 160, 2), ('l3.0ds', 512, 1024, 60, 80, 2), ('l4.0ds', 1024, 2048, 30, 40, 2), ('l3.0c1', 512, 256, 60, 80, 1),
                              ('l3.c3', 256, 1024, 30, 40, 1), ('l4.c3', 512, 2048, 15, 20, 1)):
    OH, OW = (H - 1) // s + 1, (W - 1) // s + 1
    x = torch.randn(B, H, W, ci, device=dev).to(torch.bfloat16); w = (torch.randn(co, 1, ci, device=dev) / ci ** 0.5).to(torch.bfloat16)
    y = torch.empty(B, OH, OW, co, device=dev, dtype=torch.bfloat16); bias = torch.randn(co, device=dev)
    f = lambda: hip.conv2d(0, x, w, y, B, H, W, ci, ci, OH, OW, co, 1, 1, s, s, 0, 0, bias=bias)
    row = '%-7s %4d->%4d s%d ' % (name, ci, co, s)
    for mode in (0, 1):
        prev = hip.set_option(hip.OPT_C1S, mode)
        t = timeit(f, 30)
        hip.set_option(hip.OPT_C1S, prev)
        row += ' c1s=%d %6.1f us' % (mode, t)
    print(row, flush=True)
