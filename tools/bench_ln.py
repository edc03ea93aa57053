"""LayerNorm forward / backward timings on the model's shapes (rows x cols, bf16, residual + dropout 0.1).
usage: python tools/bench_ln.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpv1_amd.hip as hip
dev = 'cuda'


def timeit(run, n=50):
    for _ in range(5):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000.0 / n


for rows, cols in ((9600, 256), (3200, 256), (3200, 768), (3968, 768), (640, 768), (192, 768)):
    x = torch.randn(rows, cols, device=dev).to(torch.bfloat16); s = torch.randn_like(x); dy = torch.randn_like(x)
    y = torch.empty_like(x); dx = torch.empty_like(x); ds = torch.empty_like(x)
    gamma = torch.ones(cols, device=dev); beta = torch.zeros(cols, device=dev)
    mean = torch.empty(rows, device=dev); rstd = torch.empty(rows, device=dev)
    dg = torch.zeros(cols, device=dev); db = torch.zeros(cols, device=dev)
    tf = timeit(lambda: hip.layernorm_fwd(x, s, gamma, beta, y, mean, rstd, rows, cols, 1e-5, 0.1, 7))
    tb = timeit(lambda: hip.layernorm_bwd(dy, x, s, gamma, mean, rstd, dx, ds, dg, db, rows, cols, 0.1, 7))
    tn = timeit(lambda: hip.layernorm_bwd(dy, x, s, gamma, mean, rstd, dx, ds, None, None, rows, cols, 0.1, 7))    # no dgamma / dbeta: the column-sum tail alone
    nblk = hip.layernorm_bwd_blocks(rows, cols)
    part = torch.empty(nblk, 2 * cols, device=dev)
    tp = timeit(lambda: hip.layernorm_bwd(dy, x, s, gamma, mean, rstd, dx, ds, None, None, rows, cols, 0.1, 7, partials=part))   # partial rows, folded later
    tfo = timeit(lambda: hip.colsum_fold_group([(part, dg, db, nblk, cols)] * 8))
    mb = rows * cols * 2 / 1e6
    print('%5d x %4d  fwd %5.1f us (%4.0f GB/s)  bwd %5.1f us (%4.0f GB/s)  bwd without dgamma / dbeta %5.1f us  bwd with partial rows %5.1f us (%d blocks; fold of 8 such: %5.1f us)' %
          (rows, cols, tf, 3 * mb / tf * 1e3, tb, 5 * mb / tb * 1e3, tn, tp, nblk, tfo), flush=True)
