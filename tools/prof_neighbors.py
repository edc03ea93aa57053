"""which kernels run right before / after a given kernel (substring argv[2]) in the last step of a rocprofv3 --kernel-trace csv:
finds where copies / small torch kernels sit in the step"""
import csv, glob, os, sys, collections
d, pat = sys.argv[1], sys.argv[2]
trace = list(csv.DictReader(open(glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)[0])))
trace.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(trace) if 'image_to_nhwc4' in r['Kernel_Name']]
seq = trace[idx[-2]:idx[-1]]
nm = lambda r: r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('_ZN12_GLOBAL__N_1', '').replace('_ZN4gpvk12_GLOBAL__N_1', '')[:48]
cnt = collections.Counter()
for i, r in enumerate(seq):
    if pat in r['Kernel_Name']:
        cnt[(nm(seq[i - 1]) if i else '-', r.get('Grid_Size', ''), nm(seq[i + 1]) if i + 1 < len(seq) else '-')] += 1
for (a, g, b), c in cnt.most_common(40):
    print('%3d  grid %-9s after %-50s before %s' % (c, g, a, b))
