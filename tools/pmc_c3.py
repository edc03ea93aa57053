"""launches for PMC collection on the streaming 3x3 kernels (conv3x3_stream.hip): layer1 / layer2 conv2 forward, layer2 conv2 backward-data, B=32"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpv1_amd.hip as hip
dev = 'cuda'; B = 32
for name, ci, co, H, W, mode in [('l1.c2', 64, 64, 120, 160, 0), ('l2.c2', 128, 128, 60, 80, 0), ('l2.c2 dgrad', 128, 128, 60, 80, 1)]:
    x = torch.randn(B, H, W, ci, device=dev).to(torch.bfloat16)
    w = (torch.randn(co, 9, ci, device=dev) / (ci * 9) ** 0.5).to(torch.bfloat16)
    y = torch.empty(B, H, W, co, device=dev, dtype=torch.bfloat16)
    bias = torch.randn(co, device=dev)
    msk = torch.randn(B, H, W, co, device=dev).to(torch.bfloat16)
    for _ in range(4):
        if mode == 0:
            hip.conv2d(0, x, w, y, B, H, W, ci, ci, H, W, co, 3, 3, 1, 1, 1, 1, bias=bias, act=1)
        else:
            hip.conv2d(1, x, w, y, B, H, W, ci, ci, H, W, co, 3, 3, 1, 1, 1, 1, relu_mask=msk)
    torch.cuda.synchronize()
