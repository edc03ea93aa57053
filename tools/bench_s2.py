"""stride-2 dgrad shapes of the body (first block of layer2/3/4) under the kernel-selection knobs: GPV_GLDS 1 (default) / 0"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpv1_amd.hip as hip
dev = 'cuda'; B = 32
SH = [('l2.0c2s2', 128, 128, 3, 2, 1, 120, 160), ('l2.0ds', 256, 512, 1, 2, 0, 120, 160), ('l3.0c2s2', 256, 256, 3, 2, 1, 60, 80), ('l3.0ds', 512, 1024, 1, 2, 0, 60, 80),
      ('l4.0c2s2', 512, 512, 3, 2, 1, 30, 40), ('l4.0ds', 1024, 2048, 1, 2, 0, 30, 40)]
for name, ci, co, k, s, p, H, W in SH:
    OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    dy = torch.randn(B, OH, OW, co, device=dev).to(torch.bfloat16)
    wd = (torch.randn(ci, k * k, co, device=dev) / (co * k * k) ** 0.5).to(torch.bfloat16)
    dx = torch.empty(B, H, W, ci, device=dev, dtype=torch.bfloat16)
    res = torch.randn(B, H, W, ci, device=dev).to(torch.bfloat16); msk = torch.randn(B, H, W, ci, device=dev).to(torch.bfloat16)
    def run(): hip.conv2d(1, dy, wd, dx, B, OH, OW, co, co, H, W, ci, k, k, s, s, p, p, res=res, relu_mask=msk)
    row = '%-9s M=%7d N=%4d K=%5d ' % (name, B * H * W, ci, k * k * co)
    for g in (1, 0):
        hip.set_option(hip.OPT_GLDS, g)
        hip.set_option(hip.OPT_PIPE, 0)
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        by = (dy.numel() + dx.numel() * 3 + wd.numel()) * 2
        row += ' | glds=%d %7.1f us %5.0f GB/s' % (g, us, by / us / 1e3)
    print(row)
