"""What does a node of the batch-1 decode graph cost, and why?  (round 6: tools/probe/grid_sync.hip measured 2.7 us per DEPENDENT
trivial kernel node of a hipGraph -- the decode step's nodes take 4.7 - 7 us.)  Chains of 400 dependent launches in ONE hipGraph:
every decode kernel alone (the same launch repeated: instruction cache hot), the per-token sequence interleaved, torch's fill.
    python tools/probe_decode_nodes.py          (GPU box)"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpv1_amd import hip, ops           # noqa: E402

dev = 'cuda'
dt = torch.bfloat16
D, H, F, V, T, Tm = 768, 8, 2048, 10000, 20, 108
dh = D // H
g = torch.Generator(device='cpu').manual_seed(0)
rn = lambda *s: (0.05 * torch.randn(*s, generator=g)).to(dev)
x, s = rn(1, D).to(dt), rn(1, D).to(dt)
gamma, beta = torch.ones(D, device=dev), torch.zeros(D, device=dev)
Wqkv, bqkv = rn(3 * D, D).to(dt), rn(3 * D)
Wq, bq = rn(D, D).to(dt), rn(D)
W1, b1 = rn(F, D).to(dt), rn(F)
W2, b2 = rn(D, F).to(dt), rn(D)
Wo, bo = rn(D, D).to(dt), rn(D)
Wc = rn(V, D).to(dt)
cache = rn(1, T, 3 * D).to(dt)
kvm = rn(Tm, 2 * D).to(dt)
part = torch.zeros(1, H, D, device=dev, dtype=torch.float32)
xn = torch.empty(1, D, device=dev, dtype=dt)
q = rn(1, D).to(dt)
h = torch.empty(1, F, device=dev, dtype=dt)
s2 = torch.empty(1, D, device=dev, dtype=dt)
lg = torch.empty(1, V, device=dev, dtype=dt)
tok = torch.zeros(1, dtype=torch.long, device=dev)
vm = torch.zeros(V, device=dev)
fill = torch.zeros(256, device=dev)
t = 9


def k_qkv(): hip.ln_linear_rows(x, s, gamma, beta, 1e-5, xn, Wqkv, bqkv, cache[:, t], T * 3 * D, 1, 3 * D, D)
def k_self(): hip.attention_row_proj(cache[:, t], T * 3 * D, cache[:, :, D:], T * 3 * D, 3 * D, cache[:, :, 2 * D:], T * 3 * D, 3 * D, Wo, part, 1, H, t + 1, dh, dh ** -0.5)
def k_q(): hip.ln_linear_rows(x, None, gamma, beta, 1e-5, xn, Wq, bq, q, D, 1, D, D, s_partial=part, s_bias=bo)
def k_cross(): hip.attention_row_proj(q, D, kvm, Tm * 2 * D, 2 * D, kvm[:, D:], Tm * 2 * D, 2 * D, Wo, part, 1, H, Tm, dh, dh ** -0.5)
def k_ff1(): hip.ln_linear_rows(x, None, gamma, beta, 1e-5, xn, W1, b1, h, F, 1, F, D, ops.ACT_RELU, s_partial=part, s_bias=bo)
def k_ff2(): hip.gemm(h, W2, s2, 1, D, F, F, F, D, bias=b2)
def k_logits(): hip.ln_linear_rows(x, s2, gamma, beta, 1e-5, xn, Wc, None, lg, V, 1, V, D)
def k_pick(): hip.argmax_rows(lg, vm, tok, None)
def k_fill(): fill.fill_(1.0)
def k_gemv_small(): hip.gemm(x, Wq, s2, 1, D, D, D, D, D, bias=bq)


TOKEN = [k_qkv, k_self, k_q, k_cross, k_ff1, k_ff2] * 3 + [k_logits, k_pick]


def chain(fns, n):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for f in fns:
            f()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            i = 0
            while i < n:
                for f in fns:
                    f()
                    i += 1
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            gr.replay()
            e1.record(st)
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
    return best * 1000.0 / i, i


print('# chains of dependent launches in one hipGraph, us per node (best of 5 replays)')
for name, fns in [('torch fill (256 floats)', [k_fill]), ('gemv 768x768 (gemv_kernel)', [k_gemv_small]),
                  ('ln + qkv 768 -> 2304 (ln_gemv<2>)', [k_qkv]), ('self-attention row + out-proj (attn1_proj, 10 keys)', [k_self]),
                  ('ln(partials) + q 768 -> 768 (ln_gemv<1>)', [k_q]), ('cross-attention row + out-proj (attn1_proj, 108 keys)', [k_cross]),
                  ('ln(partials) + linear1 768 -> 2048 (ln_gemv<2>)', [k_ff1]), ('linear2 2048 -> 768 (gemv_kernel)', [k_ff2]),
                  ('ln + logits 768 -> 10000 (ln_gemv<4>)', [k_logits]), ('pick (argmax_rows, 1024 threads)', [k_pick]),
                  ('one token: the 20-launch sequence', TOKEN)]:
    us, n = chain(fns, 400)
    print('%-58s %6.2f us per node  (%d nodes)' % (name, us, n))
