"""torch-native (at::native / rocclr) kernels of the last whole training step of a rocprofv3 --kernel-trace csv, by functor and grid:
what is left outside the library's own kernels.   usage: python tools/prof_native.py <trace dir>"""
import csv, glob, os, re, sys, collections
d = sys.argv[1]
trace = list(csv.DictReader(open(glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)[0])))
trace.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(trace) if 'image_to_nhwc4' in r['Kernel_Name']]
steps = [(a, b) for a, b in zip(idx[:-1], idx[1:]) if any('adamw' in r['Kernel_Name'] for r in trace[a:b])]
a, b = steps[-1]
agg = collections.OrderedDict()
for i, r in enumerate(trace[a:b]):
    n = r['Kernel_Name']
    if 'at::native' not in n and 'rocclr' not in n:
        continue
    f = re.findall(r'(\w*Functor\w*|\w+_kernel_cuda\w*|bfloat16_copy\w*|direct_copy\w*|sigmoid\w*|CatArray\w+|index\w+|Reduce\w+<\w+|softmax\w+|arange\w*|scan\w+|MeanOps|NormTwo\w*|sum_functor|rocclr\w+|fill\w+|copy\w+)', n)
    key = (' '.join(dict.fromkeys(f))[:80] or n[:80], r.get('Grid_Size', r.get('Grid_Size_X', '?')))
    v = agg.setdefault(key, [0, 0.0, i])
    v[0] += 1
    v[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
print('# %d launches in the step, torch-native: %d, %.0f us' % (b - a, sum(v[0] for v in agg.values()), sum(v[1] for v in agg.values())))
for (k, g), v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('%4d x %8.1f us  grid %9s  first at launch %4d  %s' % (v[0], v[1], g, v[2], k))
