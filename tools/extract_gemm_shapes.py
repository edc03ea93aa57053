"""tools/gemm_shapes_step.json from a GPV_DEBUG_SYNC log of bench.py:
   GPV_DEBUG_SYNC=1 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-decode 2> log; python tools/extract_gemm_shapes.py log 2
every distinct gpv_gemm call shape (M, N, K, layoutA, layoutB, batch, accumulate) with its count per step"""
import collections, json, os, re, sys
log, nsteps = sys.argv[1], int(sys.argv[2])
tens = re.compile(r'\w+\[[^\]]*\]s\[[^\]]*\]')
c = collections.Counter()
for line in open(log):
    if '[gpv-hip]' not in line or ' gemm ' not in line: continue
    t = tens.sub('T', line.split(' gemm ', 1)[1]).split()
    pos = [x for x in t if '=' not in x]
    kw = dict(x.split('=', 1) for x in t if '=' in x)
    M, N, K = (int(x) for x in pos[3:6])
    c[(M, N, K, int(kw.get('layoutA', 0)), int(kw.get('layoutB', 0)), int(kw.get('batch', 1)), kw.get('accumulate', 'False') == 'True')] += 1
out = sorted(([list(k), v // nsteps] for k, v in c.items() if v >= nsteps), key=lambda r: r[0])
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'gemm_shapes_step.json'), 'w'))
print(len(out), 'shapes,', sum(v for _, v in out), 'calls per step')
