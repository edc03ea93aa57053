import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpv1_amd.hip as hip
dev = 'cuda'; B = 32
SH = [('l2.c1', 512, 128, 1, 1, 0, 60, 80), ('l2.c2', 128, 128, 3, 1, 1, 60, 80), ('l2.c3', 128, 512, 1, 1, 0, 60, 80),
      ('l3.c1', 1024, 256, 1, 1, 0, 30, 40), ('l3.c2', 256, 256, 3, 1, 1, 30, 40), ('l3.c3', 256, 1024, 1, 1, 0, 30, 40),
      ('l4.c1', 2048, 512, 1, 1, 0, 15, 20), ('l4.c2', 512, 512, 3, 1, 1, 15, 20), ('l4.c3', 512, 2048, 1, 1, 0, 15, 20),
      ('l2.0c1', 256, 128, 1, 1, 0, 120, 160)]
for name, ci, co, k, s, p, H, W in SH:
    OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    x = torch.randn(B, H, W, ci, device=dev).to(torch.bfloat16); dy = torch.randn(B, OH, OW, co, device=dev).to(torch.bfloat16)
    dw = torch.zeros(co, k * k, ci, device=dev); sc = torch.ones(co, device=dev)
    res = []
    for split in (0, 4, 8, 16, 32, 64, 96, 128, 192, 256):
        def run():
            hip.conv2d(2, x, dy, dw, B, H, W, ci, ci, OH, OW, co, k, k, s, s, p, p, rowscale=sc, split_k=split)
        for _ in range(2): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        res.append((split, e0.elapsed_time(e1) * 100))
    print('%-7s out %4dx%5d red %7d: ' % (name, co, k * k * ci, B * OH * OW) + '  '.join('s%d:%.0f' % r for r in res))
