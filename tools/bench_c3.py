import os, sys, subprocess
if len(sys.argv) == 1:
    for t in (1, 2, 3):
        env = dict(os.environ, GPV_FORCE_TILE=str(t))
        print('== tile cfg', ['128x128', '128x64', '64x64'][t - 1]); sys.stdout.flush()
        subprocess.run([sys.executable, __file__, 'run'], env=env)
    sys.exit(0)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpv1_amd.hip as hip
dev = 'cuda'; B = 32
for name, ci, co, H, W in [('l1.c3', 64, 256, 120, 160), ('l2.c3', 128, 512, 60, 80), ('l1.c1', 256, 64, 120, 160)]:
    x = torch.randn(B, H, W, ci, device=dev).to(torch.bfloat16); w = torch.randn(co, 1, ci, device=dev).to(torch.bfloat16)
    y = torch.empty(B, H, W, co, device=dev, dtype=torch.bfloat16); bias = torch.randn(co, device=dev)
    res = torch.randn(B, H, W, co, device=dev).to(torch.bfloat16)
    for label, kw in (('bias+relu', dict(bias=bias, act=1)), ('bias+res+relu', dict(bias=bias, res=res, act=1)), ('plain', dict())):
        def run(): hip.conv2d(0, x, w, y, B, H, W, ci, ci, H, W, co, 1, 1, 1, 1, 0, 0, **kw)
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        by = (x.numel() + y.numel() * (2 if 'res' in label else 1)) * 2
        print('%-6s %-14s %7.1f us  %6.0f GB/s' % (name, label, us, by / us / 1e3))
    # same as a plain GEMM through gpv_gemm (no conv gather)
    M = B * H * W
    def run2(): hip.gemm(x, w, y, M, co, ci, ci, ci, co, bias=bias, act=1)
    for _ in range(3): run2()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): run2()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print('%-6s %-14s %7.1f us  %6.0f GB/s' % (name, 'as gemm', us, (x.numel() + y.numel()) * 2 / us / 1e3))
# pure copy reference: y = x (elementwise add kernel) to see achievable HBM GB/s
a = torch.randn(614400 * 256, device=dev).to(torch.bfloat16); b = torch.zeros_like(a); c = torch.empty_like(a)
def run3(): hip.add(a, b, c, a.numel())
for _ in range(3): run3()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(10): run3()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 100
print('add kernel 157M elems: %.1f us  %.0f GB/s' % (us, 3 * a.numel() * 2 / us / 1e3))
