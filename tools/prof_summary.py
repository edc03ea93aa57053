"""Summarise a rocprofv3 --kernel-trace --stats run of bench.py: top kernels per step + per-conv forward table.
usage: python tools/prof_summary.py <dir with *_kernel_stats.csv / *_kernel_trace.csv> <steps+warmup> > profiles/xxx.md"""
import csv, glob, os, sys
d, nsteps = sys.argv[1], int(sys.argv[2])
stats = list(csv.DictReader(open(glob.glob(os.path.join(d, '**', '*kernel_stats.csv'), recursive=True)[0])))
trace = list(csv.DictReader(open(glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)[0])))
tot = sum(float(r['TotalDurationNs']) for r in stats)
print('# rocprofv3 --kernel-trace --stats summary of `python bench.py --steps 4 --warmup 3` (%d passes of the conv body: 7 training steps + 6 isolated forward/backward passes; kernels outside the body run 7 times)\n' % nsteps)
print('total GPU kernel time per step: %.2f ms over %.0f launches\n' % (tot / nsteps / 1e6, sum(int(r['Calls']) for r in stats) / nsteps))
print('| % | ms/step | calls/step | avg us | kernel |\n|---|---|---|---|---|')
for r in stats[:30]:
    n = r['Name'].replace('(anonymous namespace)::', '')[:110]
    print('| %.2f | %.3f | %.1f | %.1f | `%s` |' % (float(r['Percentage']), float(r['TotalDurationNs']) / nsteps / 1e6, int(r['Calls']) / nsteps, float(r['AverageNs']) / 1e3, n))
idx = [i for i, r in enumerate(trace) if 'image_to_nhwc4' in r['Kernel_Name']]
seq = trace[idx[-1]:]
def out(n, k, s, p): return (n + 2 * p - k) // s + 1
B, H, W = 32, 480, 640
oh, ow = out(H, 7, 2, 3), out(W, 7, 2, 3)
convs = [('stem', B * oh * ow, 64, 7 * 32, B * (H + 6) * (W + 8) * 4 * 2 + B * oh * ow * 64 * 2)]
h, w, inpl = out(oh, 3, 2, 1), out(ow, 3, 2, 1), 64
for li, (pl, n, st) in enumerate(((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)), 1):
    for b in range(n):
        s = st if b == 0 else 1
        h2, w2 = out(h, 3, s, 1), out(w, 3, s, 1)
        convs.append((f'layer{li}.{b}.conv1', B * h * w, pl, inpl, (B * h * w * inpl + B * h * w * pl) * 2 + pl * inpl * 2))
        convs.append((f'layer{li}.{b}.conv2', B * h2 * w2, pl, 9 * pl, (B * h * w * pl + B * h2 * w2 * pl) * 2 + 9 * pl * pl * 2))
        if b == 0:
            convs.append((f'layer{li}.{b}.downsample', B * h2 * w2, pl * 4, inpl, (B * h * w * inpl + B * h2 * w2 * pl * 4) * 2 + inpl * pl * 8))
        convs.append((f'layer{li}.{b}.conv3', B * h2 * w2, pl * 4, pl, (B * h2 * w2 * pl + 2 * B * h2 * w2 * pl * 4) * 2 + pl * pl * 8))
        inpl, h, w = pl * 4, h2, w2
def is_conv_fwd(n):
    return (('gemm_kernel' in n and 'Li2ELi0E' in n) or 'glds_kernelILi2E' in n or 'pipe_kernelILi2E' in n or 'conv1x1_kernel' in n
            or 'conv1x1_nt_kernel' in n or 'c1s_kernel' in n or 'c3r_kernel' in n or 'stem_pool_kernel' in n)


ck = [r for r in seq if is_conv_fwd(r['Kernel_Name'])]
print('\n## backbone forward, one launch per conv (last step): `stem_pool_kernel` (conv1 + pool) / `c1s_kernel` (streaming 1x1) / `c3r_kernel` (streaming 3x3) / `conv1x1_kernel` / `pipe_kernel<OP_CONV>` / `pipe_conv1x1_kernel` / `glds_kernel<OP_CONV>` / `gemm_kernel<OP_CONV>`\n')
print('| conv | M | N | K | us | TFLOP/s | algorithmic GB/s |\n|---|---|---|---|---|---|---|')
tt = tf = tb = 0
for (name, M, N, K, byts), r in zip(convs, ck[:len(convs)]):
    dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tt += dur; tf += 2.0 * M * N * K; tb += byts
    print('| %s | %d | %d | %d | %.1f | %.1f | %.0f |' % (name, M, N, K, dur, 2.0 * M * N * K / dur / 1e6, byts / dur / 1e3))
print('| **total fwd** | | | | %.1f | %.1f | %.0f |' % (tt, tf / tt / 1e6, tb / tt / 1e3))
