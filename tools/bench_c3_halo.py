"""Stride-1 3x3 convolutions of layer3 / layer4 at the bench batch: the halo-image tile kernel (gemm_glds.hip glds_halo_kernel) against the nine-tap-tile
kernel it replaces, forward (bias + ReLU) and backward-data (ReLU mask), same operands, alternating in one process.
usage: python tools/bench_c3_halo.py      (GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpv1_amd.hip as hip                      # noqa: E402

dev, dt = 'cuda', torch.bfloat16
hip.lib()
torch.manual_seed(0)


def timeit(run, n=30):
    for _ in range(5):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000.0 / n


for C, H, W, B in ((256, 30, 40, 32), (512, 15, 20, 32), (256, 30, 40, 64)):
    x = torch.randn(B, H, W, C, device=dev).to(dt)
    w = (torch.randn(C, 3, 3, C, device=dev) / (3 * C ** 0.5)).to(dt)
    bias = torch.randn(C, device=dev)
    saved = torch.randn(B, H, W, C, device=dev).to(dt)
    y = torch.empty(B, H, W, C, device=dev, dtype=dt)
    fwd = lambda: hip.conv2d(0, x, w, y, B, H, W, C, C, H, W, C, 3, 3, 1, 1, 1, 1, bias=bias, act=hip.ACT_RELU)
    bwd = lambda: hip.conv2d(1, x, w, y, B, H, W, C, C, H, W, C, 3, 3, 1, 1, 1, 1, relu_mask=saved)
    fl = 2.0 * B * H * W * C * C * 9
    out = []
    for rep in range(2):
        for mode in (1, 0):
            prev = hip.set_option(hip.OPT_C3_HALO, mode)
            out.append((mode, timeit(fwd), timeit(bwd)))
            hip.set_option(hip.OPT_C3_HALO, prev)
    for mode in (1, 0):
        tf = min(o[1] for o in out if o[0] == mode); tb = min(o[2] for o in out if o[0] == mode)
        print('%4d ch %2d x %2d x %2d images  %s: forward %5.1f us (%4.0f TFLOP/s)  backward-data %5.1f us (%4.0f TFLOP/s)' %
              (C, H, W, B, 'halo image  ' if mode else 'nine tap tiles', tf, fl / tf * 1e-6, tb, fl / tb * 1e-6), flush=True)
