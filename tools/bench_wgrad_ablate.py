"""The backbone's grouped weight-gradient calls (gpv_conv_wgrad_group, one per stage) of the bench workload, replayed alone:
time per stage, TFLOP/s; with GPV_WG_ABL=1|2|3 the timing-only ablations of the k-loop (no MFMA / loads + barriers only / no
global loads) that tell which part of the loop its interval is made of.
usage: [GPV_WG_ABL=n] python tools/bench_wgrad_ablate.py      (GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpv1_amd.hip as hip                      # noqa: E402
import gpv1_amd.backbone as bbm                 # noqa: E402
from gpv1_amd.ops import RT                     # noqa: E402

dev = 'cuda'
hip.lib()
torch.manual_seed(0)
body = bbm.ResNetBody().to(dev)
for n, p in body.named_parameters():
    if 'layer2' not in n and 'layer3' not in n and 'layer4' not in n:
        p.requires_grad_(False)
    else:
        p.grad = torch.zeros_like(p)
RT.set_precise(False)
images = torch.randn(32, 3, 480, 640, device=dev)
calls = []
orig = hip.conv_wgrad_group


def rec(problems):
    calls.append(list(problems))
    return orig(problems)


hip.conv_wgrad_group = rec
with torch.no_grad():
    keep = []
    c5 = body.forward_nhwc(images, keep)
    body.backward_nhwc(keep, torch.randn(c5.shape, device=dev).to(c5.dtype))
hip.conv_wgrad_group = orig
torch.cuda.synchronize()
tot = 0.0
if os.environ.get('WG8') is not None:            # A/B of the eight-phase 256 x 256 kernel in one process: WG8=0 | 1
    hip.set_option(hip.OPT_WG8, int(os.environ['WG8']))
if os.environ.get('WG8H') is not None:           # the half-width tiles (layer2's call): WG8H=0 | 1
    hip.set_option(hip.OPT_WG8H, int(os.environ['WG8H']))
for ci, probs in enumerate(calls):
    fl = sum(2.0 * q[4] * q[10] * q[11] * q[12] * q[13] * q[8] for q in probs)          # B*OH*OW*Cout*KH*KW*Cin
    for _ in range(3):
        orig(probs)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(10):
        orig(probs)
    ev[1].record()
    torch.cuda.synchronize()
    us = ev[0].elapsed_time(ev[1]) * 100.0
    tot += us
    print('ABL=%s call %d: %2d problems  %8.1f us  %6.1f TFLOP/s' % (os.environ.get('GPV_WG_ABL', '0'), ci, len(probs), us, fl / us * 1e-6), flush=True)
print('ABL=%s WG8=%s WG8H=%s KT=%s total %.1f us' % (os.environ.get('GPV_WG_ABL', '0'), os.environ.get('WG8', 'default'), os.environ.get('WG8H', 'default'), os.environ.get('GPV_WG8H_KT', '-'), tot))
