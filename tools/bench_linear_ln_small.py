"""out-projection + LayerNorm at the FEW-ROW shapes of batch-1 inference (DETR encoder: 300 rows, decoder: 100), as nodes of a
hipGraph chain (what they cost inside the inference graph -- stream launches from Python would measure the host): one launch
(gpv_linear_layernorm_fwd) against GEMM + LayerNorm.  usage: python tools/bench_linear_ln_small.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpv1_amd.hip as hip
dev, dt, D = 'cuda', torch.bfloat16, 256


def chain(fns, n=200):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for f in fns:
            f()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            for _ in range(n):
                for f in fns:
                    f()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st); gr.replay(); e1.record(st)
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
    return best * 1000.0 / n


for rows in (100, 300, 600, 1200, 3200):
    a = torch.randn(rows, D, device=dev).to(dt); x = torch.randn(rows, D, device=dev).to(dt)
    w = (torch.randn(D, D, device=dev) / 16).to(dt); bias = torch.randn(D, device=dev); gamma = torch.rand(D, device=dev) + 0.5; beta = torch.randn(D, device=dev)
    pos = torch.randn(rows, D, device=dev).to(dt)
    s = torch.empty_like(a); y = torch.empty_like(a); y2 = torch.empty_like(a); m = torch.empty(rows, device=dev); r = torch.empty(rows, device=dev)
    g = lambda: hip.gemm(a, w, s, rows, D, D, D, D, D, bias=bias)
    ln = lambda: hip.layernorm_fwd(x, s, gamma, beta, y, m, r, rows, D, 1e-5, pos=pos, y2=y2)
    one = lambda: hip.linear_layernorm_fwd(a, w, bias, x, gamma, beta, s, y, m, r, rows, 1e-5, pos=pos, y2=y2)
    with hip.gemm_flags(hip.GEMM_NO_PIPE_SMALL):
        t2, t1, tg = chain([g, ln]), chain([one]), chain([g])
    print('rows %5d   gemm + LayerNorm %6.2f us (gemm alone %5.2f)   one launch %6.2f us' % (rows, t2, tg, t1), flush=True)
