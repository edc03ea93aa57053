"""each distinct forward / backward-data gpv_gemm shape of the step (tools/gemm_shapes_step.json) under every kernel family: where the default
dispatch is not the fastest.   usage (GPU box): python tools/tune_gemms.py [shapes.json]"""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gpv1_amd.hip as hip
from bench_attn import timeit as _timeit
dev = 'cuda'
CHAIN = '--chain' in sys.argv          # round 6: time every candidate as a chain of dependent hipGraph nodes (what a launch costs INSIDE the step's graphs:
if CHAIN:                              # ramp + tail + node boundary) instead of back-to-back stream launches (throughput: launches overlap their ramps)
    sys.argv.remove('--chain')


def _chain(f, n=50):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        f(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            for _ in range(n):
                f()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st); gr.replay(); e1.record(st); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
    return best * 1000.0 / n


timeit = _chain if CHAIN else _timeit
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), 'gemm_shapes_step.json')
rows = json.load(open(path))
FAM = [('pipe off', hip.OPT_PIPE, 0)] + [('pipe cfg %d' % i, hip.OPT_PIPE, 100 + i) for i in range(8)] + \
      [('glds off', hip.OPT_GLDS, 0), ('glds 8-wave', hip.OPT_GLDS, 2), ('glds 4-wave', hip.OPT_GLDS, 3), ('skinny off', hip.OPT_SKINNY, 0), ('skinny all', hip.OPT_SKINNY, 2),
       ('c1s off', hip.OPT_C1S, 0), ('c1s all', hip.OPT_C1S, 2)]
tot_def = tot_best = 0.0
for (M, N, K, la, lb, batch, acc, has_res, has_mask, act, drop, f32out, ldA, ldB, ldC), count in rows:
    if batch != 1 or acc or ldA or ldB or ldC or f32out:
        continue
    A = torch.randn((K, M) if la else (M, K), device=dev).to(torch.bfloat16)
    B = (torch.randn((K, N) if lb else (N, K), device=dev) / K ** 0.5).to(torch.bfloat16)
    Cm = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    kw = dict(layoutA=la, layoutB=lb, bias=torch.randn(N, device=dev), act=act)
    if has_res: kw.update(res=torch.randn(M, N, device=dev).to(torch.bfloat16), ldr=N)
    if has_mask: kw.update(relu_mask=torch.relu(torch.randn(M, N, device=dev)).to(torch.bfloat16), ldm=N)
    if drop: kw.update(drop_p=0.1, seed=5)
    run = lambda: hip.gemm(A, B, Cm, M, N, K, M if la else K, N if lb else K, N, **kw)
    t0 = timeit(run)
    best, bname = t0, 'default'
    for name, opt, val in FAM:
        prev = hip.set_option(opt, val)
        try:
            run(); t = timeit(run)
        except RuntimeError:
            t = float('inf')
        hip.set_option(opt, prev)
        if t < best * 0.97:
            best, bname = t, name
    tot_def += t0 * count; tot_best += best * count
    flag = '' if bname == 'default' else '   <-- %s %.1f us (x%d: %.0f us per step)' % (bname, best, count, (t0 - best) * count)
    print('%3d x M=%5d N=%5d K=%5d %s%s res%d mask%d act%d drop%d  default %6.1f us%s' % (count, M, N, K, 'T' if la else 'K', 'T' if lb else 'K', has_res, has_mask, act, drop, t0, flag), flush=True)
print('total default %.0f us, best-of-families %.0f us' % (tot_def, tot_best))
