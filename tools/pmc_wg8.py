"""launches for PMC collection on the conv weight-gradient kernels: the layer3 / layer4 3x3 and 1x1 gradients of the bench workload
through gpv_conv_wgrad_group, eight-phase 256 x 256 kernel (argv[1] = 1, default) or the 128 x 128 grouped kernel (0)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpv1_amd.hip as hip
dev = 'cuda'; B = 32
hip.set_option(hip.OPT_WG8, int(sys.argv[1]) if len(sys.argv) > 1 else 1)
probs = []
keep = []
for ci, co, k, s, p, H, W in [(256, 256, 3, 1, 1, 30, 40), (1024, 256, 1, 1, 0, 30, 40), (256, 1024, 1, 1, 0, 30, 40)] * 3:
    OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    x = torch.randn(B, H, W, ci, device=dev).to(torch.bfloat16)
    dy = torch.randn(B, OH, OW, co, device=dev).to(torch.bfloat16)
    dw = torch.zeros(co, k, k, ci, device=dev)
    probs.append((x, dy, dw, None, B, H, W, ci, ci, OH, OW, co, k, k, s, s, p, p))
for _ in range(3):
    hip.conv_wgrad_group(probs)
torch.cuda.synchronize()
