"""GPU debug helper: swap ONE HIP entry point at a time for its torch emulation (tests/cpu_shim.py works on
CUDA tensors too) and report how many parameter-gradient norms disagree with the reference goldens."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import synth, cpu_shim
from tests.test_model_cpu import build_small, nested, GOLD, V, B, H, W, Tl, PAD
import gpv1_amd.ops as ops
DEV = 'cuda'
gn = json.load(open(os.path.join(GOLD, 'small_gradnorms.json')))
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
    loss = model(nested(images.to(DEV), mask.to(DEV)), (ids.to(DEV), attn.to(DEV)), tok, targets)
    loss.backward(); torch.cuda.synchronize()
    r = {n: float(p.grad.norm()) for n, p in model.named_parameters() if p.grad is not None}
    bad = [n for n in gn if abs(r[n] - gn[n]) > 0.15 * gn[n] + 2e-3 * max(gn.values())]
    return float(loss), bad
precise = sys.argv[1] == 'precise' if len(sys.argv) > 1 else True
print('baseline', run(precise)[0], len(run(precise)[1]))
for name in ['gemm', 'conv2d', 'attention_fwd', 'attention_bwd', 'layernorm_fwd', 'layernorm_bwd', 'softmax_ce', 'roi_weights',
             'add', 'add_rowbcast', 'colsum', 'cast_rowscale_t', 'prep_conv_weight', 'embedding', 'relevance_condition', 'act_fwd', 'act_bwd']:
    undo = cpu_shim.install([name])
    try:
        l, bad = run(precise)
        print('%-20s loss %.5f n_bad %d %s' % (name, l, len(bad), bad[:2]))
    except Exception as e:
        print(name, 'EXC', repr(e)[:200])
    undo()
