"""encoder attention core (B=32, 8 heads, 300x300, dh 32, dropout 0.1, no key-padding mask) forward + backward launches for PMC collection"""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpv1_amd.hip as hip
dev = 'cuda'; B, H, S, dh = 32, 8, 300, 32; D = H * dh
g = torch.Generator().manual_seed(5)
qk = torch.randn(B * S, 2 * D, generator=g).to(dev).to(torch.bfloat16)
v = torch.randn(B * S, D, generator=g).to(dev).to(torch.bfloat16)
do = torch.randn(B * S, D, generator=g).to(dev).to(torch.bfloat16)
o = torch.empty(B * S, D, device=dev, dtype=torch.bfloat16)
lse = torch.empty(B, H, S, device=dev, dtype=torch.float32)
dqk, dv = torch.empty_like(qk), torch.empty_like(v)
st = ((S * 2 * D, 2 * D), (S * 2 * D, 2 * D), (S * D, D), (S * D, D))
for _ in range(4):
    hip.attention_fwd(qk[:, :D], qk[:, D:], v, o, st, B, H, S, S, dh, 1 / math.sqrt(dh), drop_p=0.1, seed=11, lse=lse)
    hip.attention_bwd(qk[:, :D], qk[:, D:], v, o, do, dqk[:, :D], dqk[:, D:], dv, st, (S * D, D), B, H, S, S, dh, 1 / math.sqrt(dh), drop_p=0.1, seed=11, lse=lse)
torch.cuda.synchronize()
