"""a few conv launches for PMC collection: fwd / dgrad / wgrad of l3.c2 (3x3) and l2.c1 (1x1)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpv1_amd.hip as hip
dev = 'cuda'; B = 32
for name, ci, co, k, s, p, H, W in [('l3.c2', 256, 256, 3, 1, 1, 30, 40), ('l2.c1', 512, 128, 1, 1, 0, 60, 80), ('l4.c2', 512, 512, 3, 1, 1, 15, 20)]:
    OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    x = torch.randn(B, H, W, ci, device=dev).to(torch.bfloat16); dy = torch.randn(B, OH, OW, co, device=dev).to(torch.bfloat16)
    w = (torch.randn(co, k * k, ci, device=dev) / (ci * k * k) ** 0.5).to(torch.bfloat16)
    wd = w.permute(2, 1, 0).contiguous()
    y = torch.empty(B, OH, OW, co, device=dev, dtype=torch.bfloat16); dx = torch.empty_like(x)
    dw = torch.zeros(co, k * k, ci, device=dev); sc = torch.ones(co, device=dev)
    for _ in range(3):
        hip.conv2d(0, x, w, y, B, H, W, ci, ci, OH, OW, co, k, k, s, s, p, p)
        hip.conv2d(1, dy, wd, dx, B, OH, OW, co, co, H, W, ci, k, k, s, s, p, p)
        hip.conv2d(2, x, dy, dw, B, H, W, ci, ci, OH, OW, co, k, k, s, s, p, p, rowscale=sc, split_k=0)
    torch.cuda.synchronize()
