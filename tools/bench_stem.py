"""fused stem (gpv_stem_pool) at the bench shape, B=32: time per launch; env GPV_STEM_ROWS / GPV_STEM_BLOCKS are read by the library"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpv1_amd.hip as h
B, H, W = 32, 480, 640
img = torch.randn(B, 3, H, W, device='cuda')
Hp, Wp = 486, 648
xin = torch.empty(B, Hp, Wp, 4, device='cuda', dtype=torch.bfloat16)
h.image_to_nhwc4(img, xin, B, H, W, 3, Hp, Wp)
ws = (torch.randn(64, 7, 32, device='cuda') * 0.05).to(torch.bfloat16)
sh = torch.randn(64, device='cuda')
z = torch.empty(B, 120, 160, 64, device='cuda', dtype=torch.bfloat16)
for _ in range(3):
    h.stem_pool(xin, ws, sh, z, B, Hp, Wp, 240, 320, 120, 160)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(20):
    h.stem_pool(xin, ws, sh, z, B, Hp, Wp, 240, 320, 120, 160)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 20
print('stem_pool %.1f us  (%.0f GB/s of 80 MB in + 79 MB out, %.0f TFLOP/s of K=224)' % (us, (xin.numel() * 2 + z.numel() * 2) / us / 1e3, 2.0 * B * 240 * 320 * 64 * 224 / us / 1e6))
