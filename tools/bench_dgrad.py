import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpv1_amd.hip as hip
dev = 'cuda'; B = 32
SH = [('l2.c3 dgrad', 128, 512, 1, 1, 0, 60, 80), ('l2.c2 dgrad', 128, 128, 3, 1, 1, 60, 80), ('l2.c1 dgrad', 512, 128, 1, 1, 0, 60, 80),
      ('l3.c3 dgrad', 256, 1024, 1, 1, 0, 30, 40), ('l3.c2 dgrad', 256, 256, 3, 1, 1, 30, 40), ('l3.c1 dgrad', 1024, 256, 1, 1, 0, 30, 40),
      ('l3.0c2 s2', 256, 256, 3, 2, 1, 60, 80), ('l3.0ds s2', 512, 1024, 1, 2, 0, 60, 80),
      ('l4.c3 dgrad', 512, 2048, 1, 1, 0, 15, 20), ('l4.c2 dgrad', 512, 512, 3, 1, 1, 15, 20), ('l4.c1 dgrad', 2048, 512, 1, 1, 0, 15, 20),
      ('l4.0c2 s2', 512, 512, 3, 2, 1, 30, 40), ('l4.0ds s2', 1024, 2048, 1, 2, 0, 30, 40)]
tot = 0
for name, ci, co, k, s, p, H, W in SH:
    OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    dy = torch.randn(B, OH, OW, co, device=dev).to(torch.bfloat16)
    wd = (torch.randn(ci, k * k, co, device=dev) / (co * k * k) ** 0.5).to(torch.bfloat16)
    dx = torch.empty(B, H, W, ci, device=dev, dtype=torch.bfloat16)
    res = torch.randn(B, H, W, ci, device=dev).to(torch.bfloat16); msk = torch.randn(B, H, W, ci, device=dev).to(torch.bfloat16)
    def run(): hip.conv2d(1, dy, wd, dx, B, OH, OW, co, co, H, W, ci, k, k, s, s, p, p, res=res, relu_mask=msk)
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    fl = 2.0 * B * OH * OW * co * k * k * ci          # useful flops (same as forward)
    by = (dy.numel() + dx.numel() * 3 + wd.numel()) * 2
    tot += us
    print('%-12s M=%7d N=%4d K=%5d  %7.1f us  %6.1f useful TF/s  %6.0f GB/s' % (name, B * H * W, ci, k * k * co, us, fl / us / 1e6, by / us / 1e3))
print('sum', tot)
