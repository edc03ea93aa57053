"""One training step as a list: every kernel of the last whole step of a rocprofv3 --kernel-trace csv in start order, with its
start offset, duration and the idle time before it (what a wave of kernels overlaps is visible as negative gaps).
usage: python tools/prof_chain.py <trace dir> [steps_back=1] > chain.txt     (steps are delimited by image_to_nhwc4 launches;
bench.py's isolated conv loop at the end has none, so steps_back=1 is the last TIMED step)"""
import csv, glob, os, sys
d = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
trace = list(csv.DictReader(open(glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)[0])))
trace.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(trace) if 'image_to_nhwc4' in r['Kernel_Name']]
steps = [(a, b) for a, b in zip(idx[:-1], idx[1:]) if any('adamw' in r['Kernel_Name'] for r in trace[a:b])]      # (the isolated conv loop prepares images too)
a, b = steps[-back]
seq = trace[a:b]
t0 = int(seq[0]['Start_Timestamp'])
end = t0
nm = lambda r: r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('_ZN12_GLOBAL__N_1', '').replace('_ZN4gpvk12_GLOBAL__N_1', '')[:70]
print('# %d launches, %.2f ms' % (len(seq), (int(trace[b]['Start_Timestamp']) - t0) / 1e6))
for i, r in enumerate(seq):
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%4d %9.1f us  dur %7.1f  gap %6.1f  grid %8s  %s' % (i, (s - t0) / 1e3, (e - s) / 1e3, (s - end) / 1e3, r.get('Grid_Size', r.get('Grid_Size_X', '?')), nm(r)))
    end = max(end, e)
