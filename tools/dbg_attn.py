"""GPU debug helper: replay the first attention call of the small model (encoder layer 0) through the HIP
kernels and the torch emulation, forward and backward, and compare."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import synth, cpu_shim
from tests.test_model_cpu import build_small, nested, GOLD, V, B, H, W, Tl, PAD
import gpv1_amd.ops as ops
import gpv1_amd.hip as hip
DEV = 'cuda'
precise = (sys.argv[1] == 'precise') if len(sys.argv) > 1 else True
ops.RT.set_precise(precise)
model, _ = build_small(); model.to(DEV).eval()
images, mask, ids, attn = synth.synth_batch(B, H, W, Tl, V, pad_to=PAD)
calls = []
orig = hip.attention_fwd
def cap(q, k, v, o, strides, *a, **kw):
    calls.append((q, k, v, strides, a, dict(kw)))
    return orig(q, k, v, o, strides, *a, **kw)
hip.attention_fwd = cap
with torch.no_grad():
    model(nested(images.to(DEV), mask.to(DEV)), (ids.to(DEV), attn.to(DEV)), torch.full((B, 3), V - 3, device=DEV), None)
hip.attention_fwd = orig
q, k, v, st, a, kw = calls[0]
Bn, Hh, Sq, Sk, dh, scale = a
print('call0 B,H,Sq,Sk,dh', Bn, Hh, Sq, Sk, dh, 'strides', st, 'kpm', None if kw.get('kpm') is None else kw['kpm'].sum().item())
D = Hh * dh
def heads(t, bs, rs, S): return torch.as_strided(t, (Bn, Hh, S, dh), (bs, dh, rs, 1), t.storage_offset()).float()
qh, kh = heads(q, *st[0], Sq), heads(k, *st[1], Sk)
s = (qh @ kh.transpose(-1, -2)) * scale
print('max |score|', s.abs().max().item(), 'q absmax', qh.abs().max().item())
def fwd(fn):
    o = torch.empty(Bn * Sq, D, device=DEV, dtype=q.dtype); lse = torch.empty(Bn, Hh, Sq, device=DEV)
    kw2 = dict(kw); kw2['lse'] = lse
    fn(q, k, v, o, st, *a, **kw2)
    return o, lse
o1, l1 = fwd(hip.attention_fwd); o2, l2 = fwd(cpu_shim.attention_fwd)
print('fwd o diff', (o1.float() - o2.float()).abs().max().item(), 'o max', o2.float().abs().max().item(), 'lse diff', (l1 - l2).abs().max().item(), 'lse max', l2.abs().max().item())
do = torch.randn(Bn * Sq, D, device=DEV).to(q.dtype)
def bwd(fn, o, lse):
    dq, dk, dv = torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)
    kw2 = {kk: vv for kk, vv in kw.items() if kk != 'lse'}
    fn(q, k, v, o, do, dq, dk, dv, st, (Sq * D, D), *a, lse=lse, **kw2)
    return dq, dk, dv
ref = bwd(cpu_shim.attention_bwd, o2, l2)
for name, (o, l) in (('hip fwd outputs', (o1, l1)), ('torch fwd outputs', (o2, l2))):
    got = bwd(hip.attention_bwd, o, l)
    print('HIP bwd with', name, [((g.float() - r.float()).abs().max() / r.float().abs().max()).item() for g, r in zip(got, ref)])
