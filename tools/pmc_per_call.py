"""per-CALL counter values of one kernel from a rocprofv3 --pmc run of tools/bench_body.py: the dispatches of `kernel` in dispatch
order, grouped by `reps` consecutive launches (bench_body replays every recorded call 3 + iters times), next to the 'wgrd' /
'dgrd' / 'fwd' rows bench_body printed.   usage: python tools/pmc_per_call.py <pmc dir> <kernel substring> <reps> <body table> <row filter>"""
import csv, glob, re, sys
d, kern, reps, table, filt = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5]
rows = []
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
rows = [r for r in rows if kern in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Dispatch_Id']))
vals = [float(r['Counter_Value']) for r in rows]
names = [l.rstrip() for l in open(table) if re.match(r'(fwd|bwd)\s', l) and filt in l]
calls = [sum(vals[i:i + reps]) / reps for i in range(0, len(vals) - reps + 1, reps)]
print('# %d dispatches of %s, %d calls, %d table rows' % (len(vals), kern, len(calls), len(names)))
for n, v in zip(names, calls[-len(names):] if len(calls) >= len(names) else calls):
    f = n.split()                                  # ph, name ..., us, bound us, TF/s, GB/s, excess
    us, gbs, name = float(f[-5]), float(f[-2]), ' '.join(f[1:-5])
    alg_mb = gbs * us / 1e3                        # bench_body's algorithmic bytes of the call
    print('%-40s %7.1f us  algorithmic %7.1f MB  counter (KB) x2 -> %8.1f MB  ratio %.2f' % (name, us, alg_mb, v * 2 * 1024 / 1e6, v * 2 * 1024 / 1e6 / max(alg_mb, 1e-9)))
