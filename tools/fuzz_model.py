"""Model-level random sweep: the small GPV (tests/test_model_cpu.build_small) on random batch sizes, image sizes, ragged padding,
query lengths and task mixes -- precise-mode forward + loss + backward on the GPU against the CPU oracle (what __graft_entry__.smoke()
does for one fixed shape), then the same batch through the bf16 trainer eagerly and on the hipGraph path (losses must agree).
usage: python tools/fuzz_model.py [seed] [n]        (GPU box; the oracle is the checker only; FUZZ_ONLY=<case index>, FUZZ_VERBOSE=1: per-step losses)"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpv1_amd.hip as hip
import gpv1_amd.ops as ops
from gpv1_amd.misc import NestedTensor
from gpv1_amd.train import FlatTrainer
from oracle import gpv_oracle as O
from tests import synth
from tests.test_model_cpu import build_small, V

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rng = random.Random(seed)
hip.lib()
dev = 'cuda:0'
if os.environ.get('FUZZ_T_EXACT'):             # queries up to this many tokens enter the graph path unpadded (train.FlatTrainer._classed)
    FlatTrainer.T_EXACT = int(os.environ['FUZZ_T_EXACT'])
bad = 0
for it in range(n):
    B, H, W, Tl = rng.randint(1, 5), 32 * rng.randint(2, 5) + rng.choice([0, 0, 7, 16]), 32 * rng.randint(2, 6) + rng.choice([0, 0, 5, 24]), rng.randint(3, 9)
    pad = [(rng.randint(H // 2, H), rng.randint(W // 2, W)) for _ in range(B)] if rng.random() < 0.6 else None
    tasks = rng.choice([('CocoCaptioning',), ('CocoDetection',), ('CocoCaptioning', 'CocoVqa', 'CocoClassification', 'CocoDetection'), ('CocoVqa', 'CocoDetection')])
    images, mask, ids, attn = synth.synth_batch(B, H, W, Tl, V, seed=100 + it, pad_to=pad)
    targets = synth.synth_targets(B, V, S=rng.randint(2, 8), seed=7 + it, tasks=tasks)
    if os.environ.get('FUZZ_ONLY') not in (None, str(it)):          # one case of the sweep (the random draws above stay in step)
        continue
    ops.RT.set_precise(True)
    model, man = build_small()
    model.to(dev).train()
    model.bert.model.p = 0.0
    gt = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in t.items()} for t in targets]
    _, tok = model.encode_answers(gt)
    for i, t in enumerate(gt):
        t['answer_token_ids'] = tok[i, 1:]
    samples = NestedTensor(images.to(dev), mask.to(dev), None if pad is not None else True)
    loss = model(samples, (ids.to(dev), attn.to(dev)), tok, gt)
    tag = (B, H, W, Tl, 'ragged' if pad else 'full', tasks)
    if loss is None:
        print('no applicable loss', tag)
        continue
    loss.backward()
    gnorm = float(model.detr_joiner.weight.grad.norm())
    Pm = synth.synth_state(man['manifest'])
    Pm['pos_enc'] = torch.zeros(1, 30, 768)
    cfg = synth.small_cfg(0.0)
    cfg['_cls_id'] = V - 3
    for i, t in enumerate(targets):
        t['answer_token_ids'] = tok[i, 1:].cpu()
    w = Pm['detr_joiner.weight'].clone().requires_grad_(True)
    Pm['detr_joiner.weight'] = w
    out = O.gpv_forward(Pm, cfg, images, mask, ids, attn, tok.cpu(), training=True)
    ref, _ = O.gpv_criterion(out, targets, cfg['losses'])
    ref.backward()
    e_loss = abs(float(loss) - float(ref)) / abs(float(ref))
    e_grad = abs(gnorm - float(w.grad.norm())) / max(float(w.grad.norm()), 1e-12)
    ok = e_loss < 1e-3 and e_grad < 5e-3
    # bf16 trainer: eager vs graphed on the same batch (dropout off), 4 steps each
    ops.RT.set_precise(False)
    ls = {}
    for graphs in (False, True):
        m2, _ = build_small()
        m2.to(dev).train()
        m2.bert.model.p = 0.0
        for mod in m2.modules():
            if hasattr(mod, 'p') and isinstance(getattr(mod, 'p'), float):
                mod.p = 0.0
        tr = FlatTrainer(m2, lr=1e-4, lr_backbone=1e-5, graphs=graphs)
        ls[graphs], snaps = [], []
        for _ in range(4):
            ls[graphs].append(float(tr.train_step(samples, (ids.to(dev), attn.to(dev)), [{k: v for k, v in t.items() if k != 'answer_token_ids'} for t in gt])))
            if os.environ.get('FUZZ_VERBOSE'):
                torch.cuda.synchronize()
                snaps.append(tr.P.clone())
        if os.environ.get('FUZZ_VERBOSE'):
            if not graphs:
                snaps_eager = snaps
            else:                       # first step / parameters where the graphed arm leaves the eager one
                for st, (pe, pg) in enumerate(zip(snaps_eager, snaps)):
                    rows = []
                    for e in tr.entries:
                        nme, o, k = e[0], e[3], e[4]
                        d = float((pe[o:o + k] - pg[o:o + k]).abs().max())
                        if d > 0:
                            rows.append((d / max(float(pe[o:o + k].abs().max()), 1e-30), nme))
                    rows.sort(reverse=True)
                    print('   after step %d: %d of %d parameters differ%s' % (st + 1, len(rows), len(tr.entries), ''.join('\n      %.2e %s' % r for r in rows[:14])))
        gs = tr.graph_steps
    dev_ = max(abs(a - b) / max(abs(a), 1e-6) for a, b in zip(ls[False], ls[True]))
    if os.environ.get('FUZZ_VERBOSE'):
        print('   eager  ', ' '.join('%.9g' % v for v in ls[False]))
        print('   graphed', ' '.join('%.9g' % v for v in ls[True]))
    ok2 = dev_ < 2e-2 and gs >= 2
    print('%s  precise vs oracle: loss %.1e grad %.1e %s | bf16 eager vs graphs: %.1e (graph steps %d) %s' % (tag, e_loss, e_grad, 'ok' if ok else 'FAIL', dev_, gs, 'ok' if ok2 else 'FAIL'), flush=True)
    bad += (not ok) + (not ok2)
print('model fuzz done: %d failures' % bad)
