"""GPU debug helper: determinism + per-parameter gradient norms vs reference goldens."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import synth
from tests.test_model_cpu import build_small, nested, GOLD, V, B, H, W, Tl, PAD
import gpv1_amd.ops as ops
import gpv1_amd.backbone as bb
DEV = 'cuda'
gn = json.load(open(os.path.join(GOLD, 'small_gradnorms.json')))
orig_bwd = bb.ResNetBody.backward_nhwc
def patched(self, keep, dc5):
    print('   dc5 norm', float(dc5.float().norm()), 'finite', bool(torch.isfinite(dc5.float()).all()))
    return orig_bwd(self, keep, dc5)
bb.ResNetBody.backward_nhwc = patched
def run(precise):
    ops.RT.set_precise(precise)
    model, _ = build_small(); model.to(DEV).train(); model.bert.model.p = 0.0
    images, mask, ids, attn = synth.synth_batch(B, H, W, Tl, V, pad_to=PAD)
    targets = synth.synth_targets(B, V, S=6)
    for d in targets:
        for k, v in d.items():
            if torch.is_tensor(v): d[k] = v.to(DEV)
    _, tok = model.encode_answers(targets)
    for i, t in enumerate(targets): t['answer_token_ids'] = tok[i, 1:]
    out = []
    for rep in range(2):
        for p in model.parameters(): p.grad = None
        loss = model(nested(images.to(DEV), mask.to(DEV)), (ids.to(DEV), attn.to(DEV)), tok, targets)
        loss.backward()
        torch.cuda.synchronize()
        out.append({n: float(p.grad.norm()) for n, p in model.named_parameters() if p.grad is not None})
    return out
for precise in (True, False, True, False):
    r = run(precise)
    bad = [(n, gn[n], r[0][n], r[1][n]) for n in gn if abs(r[0][n] - gn[n]) > 0.15 * gn[n] + 2e-3 * max(gn.values()) or abs(r[1][n] - gn[n]) > 0.15 * gn[n] + 2e-3 * max(gn.values())]
    print('precise' if precise else 'bf16', 'n_bad', len(bad))
    for b_ in bad[:6] + bad[-3:]:
        print('    %-60s ref %.4g rep0 %.4g rep1 %.4g' % b_)
