"""encoder-size attention fwd/bwd launches for timing / PMC collection (B=32, h=8, S=300, dh=32, dropout 0.1)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpv1_amd.hip as hip
dev = 'cuda'; B, H, S, dh = 32, 8, 300, 32; D = H * dh
qk = torch.randn(B * S, 2 * D, device=dev).to(torch.bfloat16); v = torch.randn(B * S, D, device=dev).to(torch.bfloat16)
o = torch.empty(B * S, D, device=dev, dtype=torch.bfloat16); lse = torch.empty(B, H, S, device=dev)
do = torch.randn(B * S, D, device=dev).to(torch.bfloat16)
dqk = torch.empty_like(qk); dv = torch.empty_like(v)
st = ((S * 2 * D, 2 * D), (S * 2 * D, 2 * D), (S * D, D), (S * D, D))
def fwd(): hip.attention_fwd(qk, qk[:, D:], v, o, st, B, H, S, S, dh, dh ** -0.5, drop_p=0.1, seed=5, lse=lse)
def bwd(): hip.attention_bwd(qk, qk[:, D:], v, o, do, dqk, dqk[:, D:], dv, st, (S * D, D), B, H, S, S, dh, dh ** -0.5, drop_p=0.1, seed=5, lse=lse)
for f, name in ((fwd, 'fwd'), (bwd, 'bwd')):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print(name, '%.1f us' % (e0.elapsed_time(e1) * 50))
