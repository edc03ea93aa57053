"""Fixed cost vs per-k-tile cost of the pipelined GEMM / conv kernel at one tile per CU (M = 38400 = 240 tiles of 160 x 256, N = 256):
50 launches captured into a hipGraph (no host launch cost in the figure), K = 64 .. 4608:  time(K) = fixed + k-tiles * slope.
usage: python tools/bench_ktile.py        (GPU box)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpv1_amd.hip as hip
dev = 'cuda'
hip.lib()


def graph_time(fn, n=50, reps=5):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (n * reps)


M, N = 38400, 256
rows = []
for K in (64, 128, 256, 512, 1024, 2304):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    wt = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    bias = torch.randn(N, device=dev)
    hip.set_option(hip.OPT_PIPE, 103)                     # 160 x 256 tiles wherever legal
    t = graph_time(lambda: hip.gemm(a, wt, y, M, N, K, K, K, N, bias=bias))
    hip.set_option(hip.OPT_PIPE, 1)
    rows.append(('gemm 160x256 K=%d' % K, K // 64, t))
B, H, W = 32, 30, 40
for Cin in (256, 512):
    x = torch.randn(B, H, W, Cin, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, 3, 3, Cin, device=dev) / (9 * Cin) ** 0.5).to(torch.bfloat16)
    y = torch.empty(B, H, W, N, device=dev, dtype=torch.bfloat16)
    bias = torch.randn(N, device=dev)
    t = graph_time(lambda: hip.conv2d(0, x, w, y, B, H, W, Cin, Cin, H, W, N, 3, 3, 1, 1, 1, 1, bias=bias, act=hip.ACT_RELU))
    rows.append(('conv3x3 Cin=%d' % Cin, 9 * Cin // 64, t))
prev = None
for name, kt, t in rows:
    s = ''
    if prev and prev[0].split()[0] == name.split()[0]:
        slope = (t - prev[2]) / (kt - prev[1])
        s = 'slope %.2f us per k-tile, fixed %.1f us' % (slope, t - kt * slope)
    print('%-22s %4d k-tiles %8.1f us   %s' % (name, kt, t, s))
    prev = (name, kt, t)
