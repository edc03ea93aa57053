"""One training step of a rocprofv3 --kernel-trace csv cut into its phases (F1 backbone forward | F2 rest of the forward | criterion |
B1 backward down to the backbone | B2 backbone backward | optimizer) with, per phase, the time each kernel family takes ON THE
CRITICAL CHAIN (wall time attributed to the kernel that is running; overlapped branch kernels are listed separately).
usage: python tools/prof_phases.py <trace dir> [steps_back=1]"""
import csv, glob, os, re, sys, collections
d = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
trace = list(csv.DictReader(open(glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)[0])))
trace.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(trace) if 'image_to_nhwc4' in r['Kernel_Name']]
# a training step = an interval between two image_to_nhwc4 launches that contains the optimizer (bench.py's isolated conv loop
# at the end of the run also prepares images); steps_back counts such intervals from the end
steps = [(a, b) for a, b in zip(idx[:-1], idx[1:]) if any('adamw' in r['Kernel_Name'] for r in trace[a:b])]
seq = trace[steps[-back][0]:steps[-back][1]]
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('gpvk::', '')
    n = re.sub(r'^void ', '', n)
    m = re.match(r'_ZN(?:4gpvk)?12_GLOBAL__N_1(\d+)([a-z0-9_]+)I(.*)', n)
    if m:
        n = m.group(2)[:int(m.group(1))] + '<' + m.group(3)[:28] + '>'
    return n[:64]
def is_conv(n):
    return any(t in n for t in ('c1s_kernel', 'c1c_kernel', 'c3r_kernel', 'stem_pool', 'conv1x1', 'c3d2_kernel', 'c1d_kernel', 'glds_wgrad', 'wgrad_group', 'Li2ELi', 'ILi2E', 's2_dgrad', 'maxpool', 'splitk_reduce'))
# phases by landmarks: F1 = [image_to_nhwc4 .. last forward conv before the first LayerNorm]; optimizer = adamw
names = [r['Kernel_Name'] for r in seq]
first_ln = next(i for i, n in enumerate(names) if 'ln_fwd' in n and i > 60)
f1_end = max(i for i in range(first_ln) if is_conv(short(names[i])) or 'c1s' in names[i] or 'pipe_' in names[i] or 'glds_' in names[i])
ce = [i for i, n in enumerate(names) if 'ce_kernelI' in n and 'reduce' not in n]
ce_f, ce_b = ce[0], ce[-1]
adam = next(i for i, n in enumerate(names) if 'adamw' in n)
# B2 starts at the act_bwd that precedes the backbone's first backward convolution
b2_start = next(i for i in range(ce_b, len(names)) if any(t in names[i] for t in ('glds_wgrad', 'pipe_conv1x1', 'pipe_kernelILi2E', 'glds_kernelILi2E', 'glds_conv1x1'))) - 1
phases = [('F1 backbone fwd', 0, f1_end + 1), ('F2 forward rest', f1_end + 1, ce_f), ('criterion', ce_f, ce_b + 1), ('B1 backward body', ce_b + 1, b2_start),
          ('B2 backbone bwd', b2_start, adam), ('optimizer', adam, len(seq))]
for pname, a, b in phases:
    if b <= a:
        continue
    t0 = int(seq[a]['Start_Timestamp'])
    t1 = max(int(r['End_Timestamp']) for r in seq[a:b])
    agg = collections.defaultdict(lambda: [0.0, 0, 0.0])
    end = t0
    for r in seq[a:b]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        k = short(r['Kernel_Name'])
        agg[k][1] += 1
        agg[k][2] += (e - s) / 1e3
        agg[k][0] += max(0, e - max(s, end)) / 1e3         # wall time newly covered by this kernel
        end = max(end, e)
    busy = sum(v[0] for v in agg.values())
    print('\n## %s: %d launches, wall %.2f ms, GPU busy %.2f ms' % (pname, b - a, (t1 - t0) / 1e6, busy / 1e3))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:22]:
        print('  %8.1f us wall  %8.1f us sum  x%-4d avg %6.1f  %s' % (v[0], v[2], v[1], v[2] / v[1], k))
