"""the step's ~140 linear weight-gradient problems as grouped launches (gpv_gemm_tt_group), ALONE on the chip: what the kernel does without the
backward chain beside it (inside the step the three launches take 0.93 ms of B1 = 433 TFLOP/s).  usage: python tools/bench_tt_group.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpv1_amd.hip as hip
from bench_attn import timeit
dev = 'cuda'
# (N_out, K_in, M_tokens, calls per step): tools/bench_lin_wgrad.py's list of the B = 32 captioning step
SHAPES = [(256, 256, 9600, 24), (768, 768, 3200, 12), (256, 256, 3200, 24), (256, 2048, 9600, 7), (2048, 256, 9600, 6), (512, 256, 9600, 6),
          (1536, 768, 3392, 3), (768, 3072, 3200, 3), (3072, 768, 3200, 3), (768, 768, 640, 10), (256, 2048, 3200, 6), (2048, 256, 3200, 6), (768, 768, 192, 13)]
bufs = {}
probs, flops = [], 0.0
for (N, K, M, calls) in SHAPES:
    Mp = (M + 63) // 64 * 64
    for c in range(calls):
        dy = torch.randn(M, N, device=dev).to(torch.bfloat16); x = torch.randn(M, K, device=dev).to(torch.bfloat16)
        dw = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
        if hip.tt_group_ok(dy, x, dw, N, K, M, N, K, K):
            probs.append((dy, x, dw, db, N, K, M, N, K, K)); flops += 2.0 * N * K * M
print('%d problems, %.0f GFLOP' % (len(probs), flops / 1e9))
if os.environ.get('WG8') is not None:
    hip.set_option(hip.OPT_W8L, int(os.environ['WG8']))
t = timeit(lambda: hip.gemm_tt_group(probs), n=10)
print('all in gpv_gemm_tt_group launches of <= 48: %.0f us = %.0f TFLOP/s' % (t, flops / t / 1e6))
for name, sel in (('reduction 9600 only', lambda q: q[6] == 9600), ('reduction <= 3392 only', lambda q: q[6] <= 3392)):
    sub = [q for q in probs if sel(q)]
    f = sum(2.0 * q[4] * q[5] * q[6] for q in sub)
    t = timeit(lambda: hip.gemm_tt_group(sub), n=10)
    print('%-24s %3d problems %.0f us = %.0f TFLOP/s' % (name, len(sub), t, f / t / 1e6))
