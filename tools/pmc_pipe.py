"""launches for PMC collection on the pipelined conv kernel: l3.c2 / l4.c2 forward, forced tile configuration (argv[1], default auto)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpv1_amd.hip as hip
dev = 'cuda'; B = 32
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 1
hip.set_option(hip.OPT_PIPE, mode)
for name, ci, co, k, s, p, H, W in [('l3.c2', 256, 256, 3, 1, 1, 30, 40), ('l4.c2', 512, 512, 3, 1, 1, 15, 20), ('l2.c2', 128, 128, 3, 1, 1, 60, 80)]:
    OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    x = torch.randn(B, H, W, ci, device=dev).to(torch.bfloat16)
    w = (torch.randn(co, k * k, ci, device=dev) / (ci * k * k) ** 0.5).to(torch.bfloat16)
    y = torch.empty(B, OH, OW, co, device=dev, dtype=torch.bfloat16)
    for _ in range(4):
        hip.conv2d(0, x, w, y, B, H, W, ci, ci, OH, OW, co, k, k, s, s, p, p)
    torch.cuda.synchronize()
