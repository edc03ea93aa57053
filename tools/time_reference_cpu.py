"""Time the REAL reference (imported from /root/reference through tools/ref_harness.py) on this container's CPU cores:
train forward+loss+backward at B=4 and greedy decode at B=1, full size (480x640, 6+6 layers, 100 queries, V=10000),
median of 5 after 2 warm-ups (SURVEY 8(d) protocol).  Build-container only: the reference's Python cannot travel to the
GPU box; the numbers are committed in BASELINE.md / tests/golden/reference_cpu_timing.json next to the goldens, and the
GPU box times the parity-checked oracle instead (bench.py cpu_baseline).

usage: python tools/time_reference_cpu.py [threads]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import ref_harness as RH                                     # noqa: E402
from gen_golden import build_reference, nested               # noqa: E402
from tests import synth                                       # noqa: E402


def roi_align_vectorised(features, boxes, output_size=7, spatial_scale=1.0, sampling_ratio=-1, aligned=False):
    """torchvision.ops.roi_align stand-in for TIMING: the harness' own stand-in (tools/ref_harness.py) delegates to the oracle's
    literal per-sample python loops (B*100 boxes x 49 bins x samples: ~50 s of a step), which says nothing about the reference.
    Same arithmetic as separable per-bin weights: out[n,c,ph,pw] = sum_yx Ay[n,ph,y] Ax[n,pw,x] feat[c,y,x]; checked against the
    literal form below before timing."""
    assert aligned and spatial_scale == 1.0 and sampling_ratio == -1
    outs = []
    for f, bx in zip(features, boxes):
        C, H, W = f.shape

        def axis(start, length, size):
            N = start.shape[0]
            grid = torch.ceil(length / output_size).clamp(min=0)
            gmax = max(int(grid.max().item()), 1)
            p_ = torch.arange(output_size, dtype=f.dtype).view(1, output_size, 1)
            i_ = torch.arange(gmax, dtype=f.dtype).view(1, 1, gmax)
            b_ = (length / output_size).view(N, 1, 1)
            g_ = grid.view(N, 1, 1)
            c = start.view(N, 1, 1) + p_ * b_ + (i_ + 0.5) * b_ / g_.clamp(min=1)
            live = (i_ < g_) & ~((c < -1.0) | (c > size))
            c = c.clamp(min=0)
            lo = c.floor().long()
            edge = lo >= size - 1
            lo = torch.where(edge, torch.full_like(lo, size - 1), lo)
            hi = torch.where(edge, lo, lo + 1)
            c = torch.where(edge, lo.to(c.dtype), c)
            l_ = c - lo.to(c.dtype)
            wl = torch.where(live, 1.0 - l_, torch.zeros_like(l_))
            wh = torch.where(live, l_, torch.zeros_like(l_))
            A = torch.zeros(N, output_size, size, dtype=f.dtype)
            A.scatter_add_(2, lo, wl).scatter_add_(2, hi, wh)
            return A / grid.clamp(min=1).view(N, 1, 1)
        x1, y1, x2, y2 = (bx[:, k] - 0.5 for k in range(4))
        Ay, Ax = axis(y1, y2 - y1, H), axis(x1, x2 - x1, W)
        outs.append(torch.einsum('nph,nqw,chw->ncpq', Ay.detach(), Ax.detach(), f))
    return torch.cat(outs)


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else min(8, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    V, B, H, W, Tl, S = 10000, 4, 480, 640, 6, 18
    G, model, manifest, vocab = build_reference(synth.model_cfg(), V, bert_layers=12)
    from oracle import gpv_oracle as O
    gchk = torch.Generator().manual_seed(3)
    fchk = torch.randn(1, 3, 15, 20, generator=gchk)
    bchk = [torch.tensor([[2.3, 1.1, 11.9, 9.7], [-1.0, -2.0, 6.0, 5.5], [14.2, 9.9, 21.5, 16.0], [3.0, 3.0, 3.0, 3.0]])]
    assert torch.allclose(roi_align_vectorised(fchk, bchk, 7, aligned=True), O.roi_align_direct(fchk[0], bchk[0], 7), atol=1e-5)
    import torchvision
    torchvision.ops.roi_align = roi_align_vectorised           # (the reference looks it up at call time: detr_roi_head.py:45)
    g = torch.Generator().manual_seed(1234)
    images = torch.randn(B, 3, H, W, generator=g)
    mask = torch.zeros(B, H, W, dtype=torch.bool)
    ids = torch.randint(1000, 30000, (B, Tl), generator=g)
    attn = torch.ones(B, Tl, dtype=torch.long)
    words = torch.randint(0, V - 4, (B, S), generator=g)
    targets = [{'task': 'CocoCaptioning', 'answer': ' '.join(f'w{int(w)}' for w in row)} for row in words]
    model.train()
    _, tok = model.encode_answers(targets)
    for i, t in enumerate(targets):
        t['answer_token_ids'] = tok[i, 1:]
    tt = []
    for it in range(7):
        model.zero_grad()
        t0 = time.time()
        loss = model(nested(images, mask), (ids, attn), tok, targets)
        loss.backward()
        if it >= 2:
            tt.append(time.time() - t0)
        print('train iter', it, '%.2f s' % (time.time() - t0), flush=True)
    model.eval()
    gt = []
    with torch.no_grad():
        for it in range(4):
            t0 = time.time()
            model(nested(images[:1], mask[:1]), (ids[:1], attn[:1]), None, None)
            if it >= 1:
                gt.append(time.time() - t0)
            print('greedy iter', it, '%.2f s' % (time.time() - t0), flush=True)
    res = {'what': 'allenai/gpv-1 reference (exp.gpv.models.gpv.GPV, imported through tools/ref_harness.py stubs) on the build '
                   'container CPU; random-init weights, synthetic 480x640 batch, CocoCaptioning-only targets, fp32',
           'threads': threads, 'cpu_count': os.cpu_count(), 'torch': torch.__version__,
           'train_fwd_bwd_B4_s': sorted(tt)[len(tt) // 2], 'train_images_per_s': B / sorted(tt)[len(tt) // 2],
           'greedy_B1_s_per_image': sorted(gt)[len(gt) // 2], 'runs': {'train': tt, 'greedy': gt}}
    out = os.path.join(ROOT, 'tests', 'golden', 'reference_cpu_timing.json')
    json.dump(res, open(out, 'w'), indent=1)
    print(json.dumps(res))


if __name__ == '__main__':
    main()
