"""every distinct gpv_gemm shape of one B=32 training step (tools/gemm_shapes_step.json, extracted from a GPV_DEBUG_SYNC log),
timed back-to-back: where the transformer-side GEMM time goes"""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpv1_amd.hip as hip
import gpv1_amd.ops as ops
dev = 'cuda'
if 'SK' in os.environ: hip.set_option(hip.OPT_SKINNY, int(os.environ['SK']))
shapes = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'gemm_shapes_step.json')))
rows = []
for (M, N, K, la, lb, batch, acc), count in shapes:
    if batch != 1: continue
    A = torch.randn((K, M) if la else (M, K), device=dev).to(torch.bfloat16)
    B = torch.randn((K, N) if lb else (N, K), device=dev).to(torch.bfloat16)
    Cm = torch.zeros(M, N, device=dev, dtype=torch.float32 if acc else torch.bfloat16)
    kw = dict(layoutA=la, layoutB=lb)
    if acc: kw.update(accumulate=True, split_k=ops._split_k(M, N, K))
    def run(): hip.gemm(A, B, Cm, M, N, K, M if la else K, N if lb else K, N, **kw)
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    rows.append((us * count, us, count, M, N, K, la, lb, acc))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print('total %.2f ms over %d calls' % (tot / 1e3, sum(r[2] for r in rows)))
for r in sorted(rows, key=lambda r: (r[3], r[4], r[5], r[6], r[7]))  if os.environ.get('SORT') else rows[:40]:
    t, us, c, M, N, K, la, lb, acc = r
    print('%7.0f us = %3d x %6.1f us  M=%5d N=%5d K=%5d  %s%s %s  %5.0f TF/s' % (t, c, us, M, N, K, 'T' if la else 'K', 'T' if lb else 'K', 'wgrad' if acc else '', 2.0 * M * N * K / us / 1e6))
