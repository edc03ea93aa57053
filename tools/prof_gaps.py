"""GPU idle gaps inside one training step, from a rocprofv3 --kernel-trace csv (run on the box; prints a summary)."""
import csv, glob, os, sys
d = sys.argv[1]
trace = list(csv.DictReader(open(glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)[0])))
trace.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(trace) if 'image_to_nhwc4' in r['Kernel_Name']]
a, b = idx[-2], idx[-1]
seq = trace[a:b]
t0, t1 = int(seq[0]['Start_Timestamp']), int(trace[b]['Start_Timestamp'])
busy = 0; gaps = []; end = t0
for i, r in enumerate(seq):
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if s > end:
        gaps.append((s - end, i))
    busy += max(0, e - max(s, end)); end = max(end, e)
gaps.append((t1 - end, len(seq)))
print('step wall %.2f ms, GPU busy %.2f ms, idle %.2f ms in %d gaps, %d launches' % ((t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, len(gaps), len(seq)))
nm = lambda r: r['Kernel_Name'].replace('(anonymous namespace)::', '')[:60]
hist = {}
for g, i in gaps:
    k = '<2us' if g < 2000 else '<5us' if g < 5000 else '<20us' if g < 20000 else '<100us' if g < 100000 else '>=100us'
    hist[k] = hist.get(k, [0, 0]); hist[k][0] += 1; hist[k][1] += g
print({k: (v[0], '%.2f ms' % (v[1] / 1e6)) for k, v in hist.items()})
for g, i in sorted(gaps, reverse=True)[:14]:
    print('%8.1f us  at launch %4d  after %-60s before %s' % (g / 1e3, i, nm(seq[i - 1]) if i > 0 else '-', nm(seq[i]) if i < len(seq) else 'next step'))
