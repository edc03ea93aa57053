"""Per-launch table of the ResNet-50 body at the bench workload (B=32, 480x640, bf16): every C-ABI call that
`ResNetBody.forward_nhwc` / `backward_nhwc` issue is recorded once, then replayed alone on the stream (20 times, HIP events)
and printed with its own roofline bound  max(bytes / 8 TB/s, flops / 2.5 PFLOP/s)  -- the work list of the conv rounds.
usage: python tools/bench_body.py [--batch 32] [--filter l1]      (GPU box)"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpv1_amd.hip as hip                      # noqa: E402
import gpv1_amd.backbone as bbm                 # noqa: E402
from gpv1_amd.ops import RT                     # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--iters', type=int, default=20)
ap.add_argument('--filter', default='')
ap.add_argument('--nomask', action='store_true', help='backward-data launches also WITHOUT their ReLU-mask operand (wrong numbers, right bytes: what a 1-bit mask could save at most)')
args = ap.parse_args()
dev = 'cuda'
hip.lib()
torch.manual_seed(0)
body = bbm.ResNetBody().to(dev)
for n, p in body.named_parameters():
    if 'layer2' not in n and 'layer3' not in n and 'layer4' not in n:
        p.requires_grad_(False)
for n, b in body.named_buffers():
    if n.endswith('running_var'):
        b.uniform_(0.5, 1.5)
RT.set_precise(False)
images = torch.randn(args.batch, 3, 480, 640, device=dev)

calls = []
names = {}
for n, m in body.named_modules():
    if isinstance(m, bbm.ConvW):
        names[id(m)] = n
ENTRY = ('conv2d', 'maxpool3x3s2', 'image_to_nhwc4', 'act_bwd', 'stem_pool', 'conv1x1_dual', 'conv1x1_chain', 'conv_wgrad_group')
orig = {k: getattr(hip, k) for k in ENTRY}
phase = ['fwd']


def rec(k):
    def f(*a, **kw):
        calls.append((phase[0], k, a, kw))
        return orig[k](*a, **kw)
    return f


for k in ENTRY:
    setattr(hip, k, rec(k))
with torch.no_grad():
    for g in body.parameters():
        if g.requires_grad:
            g.grad = torch.zeros_like(g)
    keep = []
    c5 = body.forward_nhwc(images, keep)
    dc5 = torch.randn(c5.shape, device=dev).to(c5.dtype)
    phase[0] = 'bwd'
    body.backward_nhwc(keep, dc5)
for k in ENTRY:
    setattr(hip, k, orig[k])
torch.cuda.synchronize()


def describe(k, a, kw):
    if k == 'stem_pool':
        x, w, shift, y, B, Hp, Wp, CH, CW, PH, PW = a[:11]
        # algorithmic bytes as bench.py counts the stem: padded input + the conv map it no longer writes; the kernel's own traffic is input + pooled map
        return 'stem conv7x7/2 + pool (fused)', B * Hp * Wp * 8 + B * PH * PW * 128, 2.0 * B * CH * CW * 64 * 224
    if k == 'conv1x1_dual':
        a1, w1, a2, w2, bias, y, B, OH, OW, K1, IH2, IW2, K2, s2, N = a[:15]
        by = B * OH * OW * K1 * 2 + B * IH2 * IW2 * K2 * 2 / (s2 * s2) + B * OH * OW * N * 2 + N * (K1 + K2) * 2
        return 'fwd  %d+%d->%4d conv3+downsample/%d %3dx%3d' % (K1, K2, N, s2, OH, OW), by, 2.0 * B * OH * OW * N * (K1 + K2)
    if k == 'conv1x1_chain':
        a1, w1, a2, w2, s2, res, bias, y, wn, bias_n, z, B, OH, OW = a[:14]
        K1, N, N2 = a1.shape[-1], y.shape[-1], z.shape[-1]
        K2 = 0 if a2 is None else a2.shape[-1]
        px = B * OH * OW
        # algorithmic bytes of the TWO convolutions it replaces (bench.py's yardstick): conv3 (+ downsample | + residual) and the next conv1, which re-reads y
        by = px * (K1 + K2) * 2 + px * N * 2 * (2 if res is not None else 1) + N * (K1 + K2) * 2 + px * N * 2 + px * N2 * 2 + N2 * N * 2
        return 'fwd  %d%s->%d->%d tail + next conv1 %3dx%3d' % (K1, ('+%d' % K2) if K2 else ('+id' if res is not None else ''), N, N2, OH, OW), by, 2.0 * px * (N * (K1 + K2) + N2 * N)
    if k == 'conv_wgrad_group':
        by = fl = 0.0
        for x, dy, dw, scale, B, IH, IW, Cs, Cin, OH, OW, Cout, KH, KW, SH, SW, PH, PW in a[0]:
            by += B * IH * IW * Cs * 2 / (SH * SW if KH == 1 else 1) + B * OH * OW * Cout * 2 + Cout * KH * KW * Cin * 4 * 2
            fl += 2.0 * B * OH * OW * Cout * KH * KW * Cin
        return 'wgrd group of %d problems' % len(a[0]), by, fl
    if k != 'conv2d':
        return k, 0.0, 0.0
    mode, x, w, y, B, IH, IW, Cs, Cin, OH, OW, Cout, KH, KW, SH, SW, PH, PW = a[:18]
    T = KH * KW
    if mode == 0:
        by = B * IH * IW * Cs * 2 / (SH * SW if KH == 1 else 1) + B * OH * OW * Cout * 2 * (2 if kw.get('res') is not None else 1) + Cout * T * Cin * 2
        fl = 2.0 * B * OH * OW * Cout * T * Cin
        nm = 'fwd  %4d->%4d %dx%d/%d %3dx%3d' % (Cin, Cout, KH, KW, SH, OH, OW)
    elif mode == 1:
        # dgrad: x = dy [B,IH,IW,Cin=Cout_fwd], y = dx [B,OH,OW,Cout=Cin_fwd]
        ex = (1 if kw.get('res') is not None else 0) + (1 if kw.get('relu_mask') is not None else 0)
        by = B * IH * IW * Cin * 2 + B * OH * OW * Cout * 2 * (1 + ex) + Cout * T * Cin * 2
        fl = 2.0 * B * IH * IW * Cin * T * Cout
        nm = 'dgrd %4d<-%4d %dx%d/%d %3dx%3d' % (Cout, Cin, KH, KW, SH, OH, OW)
    else:
        by = B * IH * IW * Cs * 2 / (SH * SW if KH == 1 else 1) + B * OH * OW * Cout * 2 + Cout * T * Cin * 4 * 2
        fl = 2.0 * B * OH * OW * Cout * T * Cin
        nm = 'wgrd %4dx%4d %dx%d/%d %3dx%3d' % (Cout, Cin, KH, KW, SH, OH, OW)
    return nm, by, fl


tot = {'fwd': [0.0, 0.0], 'bwd': [0.0, 0.0]}
nomask_saved = [0.0]
print('%-4s %-34s %9s %9s %8s %8s %7s' % ('', 'launch', 'us', 'bound us', 'TF/s', 'GB/s', 'excess'))
rows = []
for ph, k, a, kw in calls:
    nm, by, fl = describe(k, a, kw)
    if args.filter and args.filter not in nm:
        continue
    fn = orig[k]
    if k in ('conv1x1_dual', 'conv1x1_chain') and not fn(*a, **kw):
        continue                                     # (declined: the two convolutions that follow in the list did the work)
    for _ in range(3):
        fn(*a, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(args.iters):
        fn(*a, **kw)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / args.iters
    bound = max(by / 8e12, fl / 2.5e15) * 1e6
    tot[ph][0] += us
    tot[ph][1] += bound
    rows.append((us - bound, nm, ph))
    extra = ''
    if args.nomask and k == 'conv2d' and a[0] == 1 and kw.get('relu_mask') is not None:
        kw2 = dict(kw); kw2['relu_mask'] = None
        for _ in range(3):
            fn(*a, **kw2)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(args.iters):
            fn(*a, **kw2)
        e1.record()
        torch.cuda.synchronize()
        us2 = e0.elapsed_time(e1) * 1e3 / args.iters
        nomask_saved[0] += us - us2
        extra = '   without the mask operand %6.1f us (%+.1f)' % (us2, us2 - us)
    print('%-4s %-34s %9.1f %9.1f %8.1f %8.0f %7.1f%s' % (ph, nm, us, bound, fl / us / 1e6 if us else 0, by / us / 1e3 if us else 0, us - bound, extra))
for ph in ('fwd', 'bwd'):
    print('%s total %.1f us, sum of per-launch bounds %.1f us (%.2f)' % (ph, tot[ph][0], tot[ph][1], tot[ph][1] / max(tot[ph][0], 1e-9)))
if args.nomask:
    print('backward-data without the bf16 ReLU-mask operands: %.1f us less per pass' % nomask_saved[0])
