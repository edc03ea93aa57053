"""The 3x3 forward convolutions of batch-1 inference as hipGraph nodes (chains of 100): the default dispatch (split reduction + slab pass
for the long-K ones) against the other kernel families.  usage: GPV_TUNING_LIB=1 python tools/bench_c3_bs1.py   (GPV_CONV_SPLIT=0 for the unsplit paths)"""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpv1_amd.hip as hip
dev, bf = 'cuda', torch.bfloat16


def chain(f, n=100):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        f(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            for _ in range(n):
                f()
        best = 1e9
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st); gr.replay(); e1.record(st); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
    return best * 1000.0 / n


SHAPES = [(64, 64, 1, 120, 160), (128, 128, 2, 120, 160), (128, 128, 1, 60, 80), (256, 256, 2, 60, 80), (256, 256, 1, 30, 40), (512, 512, 2, 30, 40), (512, 512, 1, 15, 20)]
FAM = [('default', None, None), ('pipe off', hip.OPT_PIPE, 0), ('glds off', hip.OPT_GLDS, 0), ('glds 8-wave', hip.OPT_GLDS, 2), ('glds 4-wave', hip.OPT_GLDS, 3), ('c3s all', hip.OPT_C3S, 2)] + \
      [('pipe cfg %d' % i, hip.OPT_PIPE, 100 + i) for i in range(6)]
print('GPV_CONV_SPLIT =', os.environ.get('GPV_CONV_SPLIT', '(default 1)'))
for Cin, Cout, s, H, W in SHAPES:
    OH, OW = (H + 2 - 3) // s + 1, (W + 2 - 3) // s + 1
    x = torch.randn(1, H, W, Cin, device=dev).to(bf); w = (torch.randn(Cout, 3, 3, Cin, device=dev) / math.sqrt(9 * Cin)).to(bf)
    y = torch.empty(1, OH, OW, Cout, device=dev, dtype=bf); bias = torch.zeros(Cout, device=dev)
    run = lambda: hip.conv2d(0, x, w, y, 1, H, W, Cin, Cin, OH, OW, Cout, 3, 3, s, s, 1, 1, bias=bias, act=hip.ACT_RELU)
    out = []
    for name, opt, val in FAM:
        prev = hip.set_option(opt, val) if opt is not None else None
        try:
            t = chain(run)
        except RuntimeError:
            t = float('nan')
        if opt is not None:
            hip.set_option(opt, prev)
        out.append('%s %.1f' % (name, t))
    gf = 2.0 * OH * OW * Cout * 9 * Cin / 1e9
    print('%3d -> %3d /%d %3dx%3d (%.2f GFLOP): %s' % (Cin, Cout, s, H, W, gf, ' | '.join(out)), flush=True)
