"""The feed-forward's second GEMM (2048 -> 256 over M rows: 150 tiles of 128 x 128 on 256 CUs) as ONE launch vs the same product split over K as a batched
launch (batch b = columns b K/s .. of A and W, fp32 partial slabs) -- what a split-K variant whose slabs the following LayerNorm sums would buy.
usage: python tools/bench_ffn2_split.py      (GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpv1_amd.hip as hip                      # noqa: E402

dev, dt = 'cuda', torch.bfloat16
hip.lib()
torch.manual_seed(0)


def timeit(run, n=50):
    for _ in range(5):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000.0 / n


for M, N, K in ((9600, 256, 2048), (3200, 256, 2048), (9600, 256, 1024), (640, 768, 3072), (3968, 768, 3072)):
    A = torch.randn(M, K, device=dev).to(dt)
    W = (torch.randn(N, K, device=dev) * 0.02).to(dt)
    out = torch.empty(M, N, device=dev, dtype=dt)
    t1 = timeit(lambda: hip.gemm(A, W, out, M, N, K, K, K, N))
    line = '%5d x %4d x %4d: one launch %5.1f us' % (M, N, K, t1)
    ref = A.float() @ W.float().t()
    for s in (2, 4, 8):
        part = torch.empty(s, M, N, device=dev, dtype=torch.float32)
        ts = timeit(lambda: hip.gemm(A, W, part, M, N, K // s, K, K, N, batch=s, sA=K // s, sB=K // s, sC=M * N))
        err = ((part.sum(0) - ref).abs().max() / ref.abs().max()).item()
        line += '   split %d (fp32 slabs) %5.1f us (err %.1e)' % (s, ts, err)
    print(line, flush=True)
