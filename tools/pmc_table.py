"""pivot a rocprofv3 --pmc counter_collection.csv: mean counter value per (kernel, grid) over the last dispatches"""
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = (r['Kernel_Name'].replace('(anonymous namespace)::', '')[:70], r.get('Grid_Size', ''), r.get('LDS_Block_Size', ''))
    agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
names = sorted({c for v in agg.values() for c in v})
print('kernel | grid | ' + ' | '.join(names))
for k, v in agg.items():
    if not any(t in k[0] for t in ('gemm_kernel', 'glds_', 'skinny', 'pipe_', 'attn_', 'c3r_', 'c3s_', 'c1s_', 'c1c_', 'c1d_', 'stem_', 'wgrad', 'wg8', 'tt8')): continue
    print(k[0][-40:], '|', k[1], '| ' + ' | '.join('%.4g' % (sum(v[c]) / len(v[c])) if c in v else '-' for c in names))
