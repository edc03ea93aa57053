"""GPU debug helper: per-parameter gradient norms, bf16 mode vs precise mode vs reference goldens."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import synth
from tests.test_model_cpu import build_small, nested, GOLD, V, B, H, W, Tl, PAD
import gpv1_amd.ops as ops
DEV = 'cuda'
gn = json.load(open(os.path.join(GOLD, 'small_gradnorms.json')))
res = {}
for precise in (True, False):
    ops.RT.set_precise(precise)
    model, _ = build_small(); model.to(DEV).train(); model.bert.model.p = 0.0
    images, mask, ids, attn = synth.synth_batch(B, H, W, Tl, V, pad_to=PAD)
    targets = synth.synth_targets(B, V, S=6)
    for d in targets:
        for k, v in d.items():
            if torch.is_tensor(v): d[k] = v.to(DEV)
    _, tok = model.encode_answers(targets)
    for i, t in enumerate(targets): t['answer_token_ids'] = tok[i, 1:]
    loss = model(nested(images.to(DEV), mask.to(DEV)), (ids.to(DEV), attn.to(DEV)), tok, targets)
    loss.backward()
    res[precise] = {n: float(p.grad.norm()) for n, p in model.named_parameters() if p.grad is not None}
    print('loss', precise, float(loss))
for n, ref in gn.items():
    a, b = res[True].get(n, -1), res[False].get(n, -1)
    flag = '' if abs(b - ref) <= 0.15 * ref + 2e-3 * max(gn.values()) else '  <<<<'
    if 'backbone' in n or flag:
        print('%-70s ref %.4g precise %.4g bf16 %.4g%s' % (n, ref, a, b, flag))
