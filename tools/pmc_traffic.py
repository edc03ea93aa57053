"""HBM traffic of the implicit-GEMM conv kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py.
usage (on the GPU box): python tools/pmc_traffic.py <dir pass FETCH_SIZE> <dir pass WRITE_SIZE> > profiles/xxx.json
(passes profiled = launches of the stem kernel, one per forward of the body; the per-launch figure divides a pass's bytes by the 135
convolutions of the reference's algorithm -- bench.py conv_algorithmic's yardstick -- not by the number of kernels this build
issues for them: fused stem / block tails and the grouped weight-gradient call make that number smaller)
Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): rocprofv3 reports both counters in KB;
on gfx950 FETCH_SIZE counts 128-byte requests at 64 B -> doubled; WRITE_SIZE is taken as reported (uncalibrated)."""
import csv, glob, json, os, re, sys

def load(d, counter):
    out = {}
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != counter:
                continue
            k = r['Kernel_Name']
            v = out.setdefault(k, [0.0, 0])
            v[0] += float(r['Counter_Value']); v[1] += 1
    return out

def is_conv(name):
    n = name.replace('(anonymous namespace)::', '')
    m = re.search(r'c1s_kernel<([^>]*)>', n)
    if m:                               # template arguments <K, NH, RES, MASK, NT, LIN, NP, BITS>: the LIN instances are the K = 256 -> 2048 linear GEMMs of the transformer (gpv_gemm), not convolutions
        targs = [t.strip() for t in m.group(1).split(',')]
        if len(targs) > 5 and targs[5] == 'true':
            return False
    return (('gemm_kernel' in n and 'Li2ELi0E' in n) or ('gemm_kernel<' in n and ', 1, 2, ' in n) or 'glds_kernelILi2E' in n or 'conv1x1_kernel' in n
            or 'glds_wgrad_kernel' in n or 'pipe_kernelILi2E' in n or 'c1s_kernel' in n or 'conv1x1_nt_kernel' in n
            or 'c3r_kernel' in n or 'c3d2_kernel' in n or 'c1d_kernel' in n or 'c1c_kernel' in n or 'stem_pool_kernel' in n or 'glds_wgrad_group_kernel' in n or 'wg8_group_kernel' in n or 'wg8h_group_kernel' in n or 'glds_halo_kernel' in n)


def is_conv_helper(name):          # second kernel of a conv launch (1x1 stride-2 backward-data: the element-wise three quarters): bytes count, launches do not
    return 's2_dgrad_fill_kernel' in name or 'wgrad_group_reduce_kernel' in name      # (+ the slab pass of the grouped weight gradients)

LAUNCHES_PER_PASS = 135          # conv launches of one forward + backward of the body (bench.py conv_algorithmic)


def compute(fd, wd):
    """-> the result dict from the two pass directories (bench.py runs the passes itself and calls this: roofline.traffic is live)"""
    F, Wr = load(fd, 'FETCH_SIZE'), load(wd, 'WRITE_SIZE')
    fetch_kb = sum(v[0] for k, v in F.items() if is_conv(k) or is_conv_helper(k)); nf = sum(v[1] for k, v in F.items() if is_conv(k))
    write_kb = sum(v[0] for k, v in Wr.items() if is_conv(k) or is_conv_helper(k)); nw = sum(v[1] for k, v in Wr.items() if is_conv(k))
    steps = sum(v[1] for k, v in F.items() if 'stem_pool_kernel' in k) or nf / LAUNCHES_PER_PASS
    kernels_per_pass = nf / steps
    res = {
        'what': 'c1s_kernel / c1c_kernel / c3r_kernel / stem_pool_kernel / gemm_kernel<OP_CONV,...> / conv1x1_kernel / pipe_kernel<OP_CONV> / pipe_conv1x1_kernel / glds_kernel<OP_CONV> / glds_halo_kernel / glds_conv1x1_kernel / c3d2_kernel / c1d_kernel / glds_wgrad_group_kernel / wg8_group_kernel / wg8h_group_kernel (+ s2_dgrad_fill_kernel, wgrad_group_reduce_kernel bytes) launches of `python bench.py` (B=32 train step)',
        'passes_profiled': steps, 'conv_kernel_launches_per_pass': kernels_per_pass, 'conv_launches_fetch_pass': nf, 'conv_launches_write_pass': nw,
        'FETCH_SIZE_KB_raw': fetch_kb, 'WRITE_SIZE_KB_raw': write_kb,
        'fetch_bytes_corrected_x2': fetch_kb * 1024 * 2, 'write_bytes': write_kb * 1024,
        'traffic_bytes_per_step': (fetch_kb * 2 + write_kb) * 1024 / steps,
        'traffic_bytes_per_launch': (fetch_kb * 2 + write_kb) * 1024 / steps / LAUNCHES_PER_PASS,
        'by_kernel_KB': {k.replace('(anonymous namespace)::', '')[:90]: {'fetch_raw': F.get(k, [0, 0])[0], 'write': Wr.get(k, [0, 0])[0], 'launches': F.get(k, [0, 0])[1]}
                         for k in F if is_conv(k) or is_conv_helper(k)},
    }
    return res


if __name__ == '__main__':
    print(json.dumps(compute(sys.argv[1], sys.argv[2]), indent=1))
