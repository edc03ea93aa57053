"""conv3 + downsample + add + ReLU of layer2's first bottleneck as one launch (gpv_conv1x1_dual) at the bench shape (B = 32):
time, TFLOP/s, algorithmic GB/s.   usage: python tools/bench_c1d.py      (GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpv1_amd.hip as hip                      # noqa: E402

dev, dt = 'cuda', torch.bfloat16
hip.lib()
torch.manual_seed(0)
for (B, OH, OW, K1, IH2, IW2, K2, s2, N) in ((32, 60, 80, 128, 120, 160, 256, 2, 512), (32, 120, 160, 64, 120, 160, 64, 1, 256)):
    a1 = torch.randn(B, OH, OW, K1, device=dev).to(dt)
    a2 = torch.randn(B, IH2, IW2, K2, device=dev).to(dt)
    w1 = (torch.randn(N, K1, device=dev) * 0.05).to(dt)
    w2 = (torch.randn(N, K2, device=dev) * 0.05).to(dt)
    bias = torch.randn(N, device=dev)
    y = torch.empty(B, OH, OW, N, device=dev, dtype=dt)
    run = lambda: hip.conv1x1_dual(a1, w1, a2, w2, bias, y, B, OH, OW, K1, IH2, IW2, K2, s2, N)
    assert run()
    for _ in range(5):
        run()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(30):
        run()
    ev[1].record()
    torch.cuda.synchronize()
    us = ev[0].elapsed_time(ev[1]) * 1000.0 / 30
    M = B * OH * OW
    fl = 2.0 * M * N * (K1 + K2)
    by = 2.0 * M * (K1 + K2 + N)
    ref = torch.relu(a1.view(M, K1).float() @ w1.float().t() + a2[:, ::s2, ::s2].reshape(M, K2).float() @ w2.float().t() + bias)
    err = ((y.view(M, N).float() - ref).abs().max() / ref.abs().max()).item()
    print('K1 %d K2 %d/%d -> %d over %d pixels: %.1f us  %.0f TFLOP/s  %.0f GB/s   max err %.2e of max' % (K1, K2, s2, N, M, us, fl / us * 1e-6, by / us * 1e-3, err), flush=True)
