"""LayerNorm fwd/bwd timings on the step's shapes"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpv1_amd.hip as hip
dev = 'cuda'
def t(fn, n=50):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for rows, cols, dt in [(9600, 256, torch.bfloat16), (3200, 256, torch.bfloat16), (3200, 768, torch.bfloat16), (640, 768, torch.bfloat16), (192, 768, torch.bfloat16), (3200, 2304, torch.bfloat16), (3200, 256, torch.float32)]:
    x = torch.randn(rows, cols, device=dev).to(dt); s = torch.randn(rows, cols, device=dev).to(dt); dy = torch.randn(rows, cols, device=dev).to(dt)
    g = torch.ones(cols, device=dev); b = torch.zeros(cols, device=dev)
    y = torch.empty_like(x); mean = torch.empty(rows, device=dev); rstd = torch.empty(rows, device=dev)
    dx = torch.empty_like(x); ds = torch.empty_like(x); dg = torch.zeros(cols, device=dev); db = torch.zeros(cols, device=dev)
    f = t(lambda: hip.layernorm_fwd(x, s, g, b, y, mean, rstd, rows, cols, 1e-5, 0.1, 7))
    bw = t(lambda: hip.layernorm_bwd(dy, x, s, g, mean, rstd, dx, ds, dg, db, rows, cols, 0.1, 7))
    bw0 = t(lambda: hip.layernorm_bwd(dy, x, s, g, mean, rstd, dx, ds, None, None, rows, cols, 0.1, 7))
    bw1 = t(lambda: hip.layernorm_bwd(dy, x, s, g, mean, rstd, dx, ds, dg, db, rows, cols, 0.0, 7))
    byt = rows * cols * x.element_size()
    print('rows %5d cols %4d %-8s fwd %6.1f us (%4.0f GB/s)  bwd %6.1f us (%4.0f GB/s)  bwd no-dgamma %6.1f  bwd no-dropout %6.1f' % (rows, cols, str(dt)[6:], f, 3 * byt / f / 1e3, bw, 5 * byt / bw / 1e3, bw0, bw1))
