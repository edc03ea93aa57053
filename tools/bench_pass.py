"""Whole-pass timing of the ResNet-50 body at the bench workload (B=32, 480x640, bf16): forward_nhwc + backward_nhwc alone on the
stream, HIP events around the two passes (what bench.py's `roofline.achieved` is made of), for the scheduling variants of the
backward pass -- GPV_WGRAD_STAGES 0 (one grouped weight-gradient call at the end) / 1 (one per stage) / side (on a branch).
usage: python tools/bench_pass.py [--batch 32] [--iters 20] [--variants 0,1,side] [--graph]      (GPU box)"""
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
ap.add_argument('--variants', default='0,1,side')
ap.add_argument('--graph', action='store_true', help='time hipGraph replays of the two passes instead of eager launches')
ap.add_argument('--split', type=int, default=1, help='(with --graph) the batch as this many independent sub-batches on parallel graph branches: do the ramps / tails of one chain hide under the other? (timing only: the weight gradients of the branches race)')
args = ap.parse_args()
dev = 'cuda'
hip.lib()
torch.manual_seed(0)
body = bbm.ResNetBody().to(dev)
for n, p in body.named_parameters():
    if 'layer2' not in n and 'layer3' not in n and 'layer4' not in n:
        p.requires_grad_(False)
    else:
        p.grad = torch.zeros_like(p)
for n, b in body.named_buffers():
    if n.endswith('running_var'):
        b.uniform_(0.5, 1.5)
RT.set_precise(False)
images = torch.randn(args.batch, 3, 480, 640, device=dev)


def run(variant):
    bbm.WGRAD_STAGES = variant
    fwd, bwd = [], []
    dc5 = None
    st = torch.cuda.Stream()
    with torch.no_grad(), torch.cuda.stream(st):
        if args.graph:
            keep = []
            c5 = body.forward_nhwc(images, keep)           # warm-up (weight copies, attributes)
            dc5 = torch.randn(c5.shape, device=dev).to(c5.dtype)
            body.backward_nhwc(keep, dc5)
            torch.cuda.synchronize()
            gf, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            pool = torch.cuda.graph_pool_handle()
            if args.split <= 1:
                keep = []
                gf.capture_begin(pool=pool)
                c5 = body.forward_nhwc(images, keep)
                gf.capture_end()
                gb.capture_begin(pool=pool)
                body.backward_nhwc(keep, dc5)
                gb.capture_end()
            else:
                n = args.split
                parts = [images[i * args.batch // n:(i + 1) * args.batch // n].contiguous() for i in range(n)]
                dparts = [dc5[i * args.batch // n:(i + 1) * args.batch // n].contiguous() for i in range(n)]
                sts = [torch.cuda.Stream() for _ in range(n)]
                keeps = [[] for _ in range(n)]
                for cap, fn in ((gf, lambda i: body.forward_nhwc(parts[i], keeps[i])), (gb, lambda i: body.backward_nhwc(keeps[i], dparts[i]))):
                    cap.capture_begin(pool=pool)
                    for i in range(n):
                        sts[i].wait_stream(st)
                        with torch.cuda.stream(sts[i]):
                            fn(i)
                    for i in range(n):
                        st.wait_stream(sts[i])
                    cap.capture_end()
        for it in range(args.iters + 3):
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record()
            if args.graph:
                gf.replay()
            else:
                keep = []
                c5 = body.forward_nhwc(images, keep)
            e1.record()
            if dc5 is None:
                dc5 = torch.randn(c5.shape, device=dev).to(c5.dtype)
            if args.graph:
                gb.replay()
            else:
                body.backward_nhwc(keep, dc5)
            e2.record()
            if it >= 3:
                fwd.append((e0, e1))
                bwd.append((e1, e2))
        torch.cuda.synchronize()
    f = sorted(a.elapsed_time(b) for a, b in fwd)
    b = sorted(a.elapsed_time(b) for a, b in bwd)
    return f[len(f) // 2], b[len(b) // 2]


for v in args.variants.split(','):
    f, b = run(v)
    print('split %d ' % args.split + 'GPV_WGRAD_STAGES=%-5s %s  forward %.3f ms  backward %.3f ms  body %.3f ms' % (v, 'graph' if args.graph else 'eager', f, b, f + b), flush=True)
