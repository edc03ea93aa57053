"""A/B of the pipelined direct-to-LDS kernel (gemm_pipe.hip) against the round-1 kernels on the B=32 ResNet-50 shapes
(forward, backward-data) and the wide transformer GEMMs: every tile configuration forced in turn, results checked
against the round-1 kernel's output.  usage: python tools/bench_pipe.py [quick]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpv1_amd.hip as hip

B = 32
dev = 'cuda'
NCFG = 8
CFG = ['256x128', '192x128', '128x128', '160x256', '128x256', '96x256', '64x64s6', '32x64s8']
FWD = [  # name, Cin, Cout, k, s, p, H, W, with_res
    ('l2.c1', 512, 128, 1, 1, 0, 60, 80, 0), ('l2.c2', 128, 128, 3, 1, 1, 60, 80, 0), ('l2.c3', 128, 512, 1, 1, 0, 60, 80, 1),
    ('l2.0c1', 256, 128, 1, 1, 0, 120, 160, 0), ('l2.0c2s2', 128, 128, 3, 2, 1, 120, 160, 0),
    ('l3.c1', 1024, 256, 1, 1, 0, 30, 40, 0), ('l3.c2', 256, 256, 3, 1, 1, 30, 40, 0), ('l3.c3', 256, 1024, 1, 1, 0, 30, 40, 1),
    ('l3.0c1', 512, 256, 1, 1, 0, 60, 80, 0), ('l3.0c2s2', 256, 256, 3, 2, 1, 60, 80, 0), ('l3.0ds', 512, 1024, 1, 2, 0, 60, 80, 0),
    ('l4.c1', 2048, 512, 1, 1, 0, 15, 20, 0), ('l4.c2', 512, 512, 3, 1, 1, 15, 20, 0), ('l4.c3', 512, 2048, 1, 1, 0, 15, 20, 1),
    ('l4.0c1', 1024, 512, 1, 1, 0, 30, 40, 0), ('l4.0c2s2', 512, 512, 3, 2, 1, 30, 40, 0), ('l4.0ds', 1024, 2048, 1, 2, 0, 30, 40, 0)]
DGRAD = [('l2.c3', 128, 512, 1, 1, 0, 60, 80), ('l2.c2', 128, 128, 3, 1, 1, 60, 80), ('l2.c1', 512, 128, 1, 1, 0, 60, 80),
         ('l2.0c2s2', 128, 128, 3, 2, 1, 120, 160),
         ('l3.c3', 256, 1024, 1, 1, 0, 30, 40), ('l3.c2', 256, 256, 3, 1, 1, 30, 40), ('l3.c1', 1024, 256, 1, 1, 0, 30, 40),
         ('l3.0c2s2', 256, 256, 3, 2, 1, 60, 80), ('l3.0ds', 512, 1024, 1, 2, 0, 60, 80),
         ('l4.c3', 512, 2048, 1, 1, 0, 15, 20), ('l4.c2', 512, 512, 3, 1, 1, 15, 20), ('l4.c1', 2048, 512, 1, 1, 0, 15, 20),
         ('l4.0c2s2', 512, 512, 3, 2, 1, 30, 40), ('l4.0ds', 1024, 2048, 1, 2, 0, 30, 40)]
SMALL = [(192, 768, 768), (192, 2304, 768), (192, 3072, 768), (192, 768, 3072), (640, 768, 768), (640, 2304, 768), (640, 2048, 768),
         (640, 768, 2048), (3392, 1536, 768), (3200, 768, 768), (3200, 768, 2304), (3200, 256, 256), (3200, 512, 256), (3200, 256, 2048),
         (3200, 2048, 256), (9600, 256, 256), (9600, 512, 256)]
GEMMS = [(9600, 2048, 256), (9600, 256, 2048), (9600, 256, 256), (9600, 512, 256), (3200, 768, 2304), (3200, 3072, 768), (3200, 768, 3072),
         (3200, 768, 768), (640, 10000, 768), (4096, 4096, 4096), (8192, 8192, 8192)]


def timeit(run, n=10):
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000.0 / n


def sweep(name, M, N, K, run, out, flops):
    hip.set_option(hip.OPT_PIPE, 0)
    run()
    ref = out.float().clone()
    t0 = timeit(run)
    row = '%-14s M=%7d N=%5d K=%5d  old %7.1f us %6.1f TF |' % (name, M, N, K, t0, flops / t0 / 1e6)
    best = (t0, 'old')
    for i in range(NCFG):
        hip.set_option(hip.OPT_PIPE_LAUNCHES, 0)
        hip.set_option(hip.OPT_PIPE, 100 + i)
        out.zero_()
        run()
        used = hip.set_option(hip.OPT_PIPE_LAUNCHES, 0)
        if not used:
            row += '   --   '
            continue
        err = float((out.float() - ref).abs().max())
        t = timeit(run)
        row += ' %6.1f%s' % (t, '!' if err > 1e-2 * float(ref.abs().max()) else ' ')
        if t < best[0]:
            best = (t, CFG[i])
    hip.set_option(hip.OPT_PIPE, 1)
    hip.set_option(hip.OPT_PIPE_LAUNCHES, 0)
    out.zero_()
    run()
    used = hip.set_option(hip.OPT_PIPE_LAUNCHES, 0)
    th = timeit(run)
    row += ' | auto %6.1f%s best %s %.1f TF' % (th, '*' if used else ' ', best[1], flops / best[0] / 1e6)
    print(row, flush=True)
    return t0, best[0], th


def main():
    print('configs:', CFG)
    tot = [0.0, 0.0, 0.0]
    print('--- forward')
    for name, ci, co, k, s, p, H, W, wr in FWD:
        OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        x = torch.randn(B, H, W, ci, device=dev).to(torch.bfloat16)
        w = (torch.randn(co, k * k, ci, device=dev) / (ci * k * k) ** 0.5).to(torch.bfloat16)
        y = torch.empty(B, OH, OW, co, device=dev, dtype=torch.bfloat16)
        bias = torch.randn(co, device=dev)
        res = torch.randn(B, OH, OW, co, device=dev).to(torch.bfloat16) if wr else None

        def run():
            hip.conv2d(0, x, w, y, B, H, W, ci, ci, OH, OW, co, k, k, s, s, p, p, bias=bias, res=res, act=1)
        r = sweep('f ' + name, B * OH * OW, co, k * k * ci, run, y, 2.0 * B * OH * OW * co * k * k * ci)
        for i in range(3):
            tot[i] += r[i]
    print('fwd sums: old %.1f best %.1f auto %.1f' % tuple(tot))
    tot = [0.0, 0.0, 0.0]
    print('--- dgrad')
    for name, ci, co, k, s, p, H, W in DGRAD:
        OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        dy = torch.randn(B, OH, OW, co, device=dev).to(torch.bfloat16)
        wd = (torch.randn(ci, k * k, co, device=dev) / (co * k * k) ** 0.5).to(torch.bfloat16)
        dx = torch.empty(B, H, W, ci, device=dev, dtype=torch.bfloat16)
        res = torch.randn(B, H, W, ci, device=dev).to(torch.bfloat16)
        msk = torch.randn(B, H, W, ci, device=dev).to(torch.bfloat16)

        def run():
            hip.conv2d(1, dy, wd, dx, B, OH, OW, co, co, H, W, ci, k, k, s, s, p, p, res=res, relu_mask=msk)
        r = sweep('d ' + name, B * H * W, ci, k * k * co, run, dx, 2.0 * B * OH * OW * co * k * k * ci)
        for i in range(3):
            tot[i] += r[i]
    print('dgrad sums: old %.1f best %.1f auto %.1f' % tuple(tot))
    print('--- plain GEMMs (NT)')
    for (M, N, K) in GEMMS:
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        b = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
        c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        bias = torch.randn(N, device=dev)

        def run():
            hip.gemm(a, b, c, M, N, K, K, K, N, bias=bias)
        sweep('g', M, N, K, run, c, 2.0 * M * N * K)


def small():
    print('configs:', CFG)
    print('--- small-M GEMMs (NT): old = skinny / 4-wave kernels')
    tot = [0.0, 0.0, 0.0]
    for (M, N, K) in SMALL:
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        b = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
        c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        bias = torch.randn(N, device=dev)

        def run():
            hip.gemm(a, b, c, M, N, K, K, K, N, bias=bias)
        r = sweep('s', M, N, K, run, c, 2.0 * M * N * K)
        for i in range(3):
            tot[i] += r[i]
    print('small sums: old %.1f best %.1f auto %.1f' % tuple(tot))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'small':
        small()
    else:
        main()
