"""Golden vectors for the optimizer part of the training step, produced by the REAL reference model
(imported through tools/ref_harness.py) driven exactly like exp/gpv/train_distr.py:228-253,399-428,468-469:
four AdamW parameter groups in named_parameters() order, clip_grad_norm_(backbone + head parameters, 0.1),
WarmupLinearSchedule stepped per iteration (restated as the LambdaLR it is: pytorch_transformers is not installed).
torch-1.6 `zero_grad` semantics (gradients zeroed, not released) are requested explicitly.

Four steps on the small fixture with different task mixes (caption-only, mixed, detection-only, caption-only), so that
parameters are first touched at different steps (per-parameter Adam step counts) and untouched ones stay put.
Writes tests/golden/optimizer_steps.npz: per step the loss, the learning rates, the clip norm, the set of parameters that hold
optimizer state, and sampled entries of every parameter after the step.   Build container only."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import ref_harness as RH                                     # noqa: E402,F401
from gen_golden import build_reference, nested, GOLD          # noqa: E402
from tests import synth                                       # noqa: E402

LR, LR_BACKBONE, WD, CLIP, WARMUP, T_TOTAL = 1e-3, 1e-4, 1e-4, 0.1, 2, 20
NSAMPLE = 16


def warmup_linear(step):
    if step < WARMUP:
        return float(step) / float(max(1, WARMUP))
    return max(0.0, float(T_TOTAL - step) / float(max(1.0, T_TOTAL - WARMUP)))


def schedule_targets(V, B):
    cap = [{'task': 'CocoCaptioning', 'answer': ' '.join(f'w{(3 * i + j) % (V - 4)}' for j in range(4))} for i in range(B)]
    det = [{'task': 'CocoDetection', 'boxes': torch.tensor([[0.5, 0.5, 0.2, 0.3], [0.3, 0.6, 0.1, 0.1]])[: 1 + i % 2],
            'labels': torch.zeros(1 + i % 2, dtype=torch.long)} for i in range(B)]
    mixed = [cap[i] if i % 2 == 0 else det[i] for i in range(B)]
    return [cap, mixed, det, cap]


def sample_idx(numel):
    return np.unique(np.linspace(0, numel - 1, NSAMPLE).astype(np.int64))


def main():
    torch.set_num_threads(8)
    V, B, H, W, Tl = 40, 4, 96, 128, 5
    G, model, manifest, vocab = build_reference(synth.small_cfg(dropout=0.0), V, bert_layers=2)
    images, mask, ids, attn = synth.synth_batch(B, H, W, Tl, V, pad_to=[(96, 128), (96, 128), (64, 96), (96, 100)])
    params = {'detr_backbone': [], 'detr_head': [], 'bert': [], 'others': []}
    names = {'detr_backbone': [], 'detr_head': [], 'bert': [], 'others': []}
    for n, p in model.named_parameters():                    # train_distr.py:234-242
        g = 'detr_backbone' if 'detr.backbone' in n else 'detr_head' if 'detr' in n else 'bert' if 'bert.' in n else 'others'
        params[g].append(p)
        names[g].append(n)
    optimizer = torch.optim.AdamW([{'params': params['detr_backbone'], 'lr': LR_BACKBONE}, {'params': params['detr_head']},
                                   {'params': params['bert']}, {'params': params['others']}], lr=LR, weight_decay=WD)
    sched = torch.optim.lr_scheduler.LambdaLR(optimizer, warmup_linear)
    flat_names = names['detr_backbone'] + names['detr_head'] + names['bert'] + names['others']
    by_name = dict(model.named_parameters())
    out = {'names': np.array(flat_names), 'hyper': np.array([LR, LR_BACKBONE, WD, CLIP, WARMUP, T_TOTAL])}
    for n in flat_names:
        out['idx:' + n] = sample_idx(by_name[n].numel())
        out['p0:' + n] = by_name[n].detach().flatten()[out['idx:' + n]].numpy().copy()
    for step, tg in enumerate(schedule_targets(V, B)):
        model.train()
        tg = [dict(t) for t in tg]
        _, tok = model.encode_answers(tg)
        for i, t in enumerate(tg):
            t['answer_token_ids'] = tok[i, 1:]
        loss = model(nested(images, mask), (ids, attn), tok, tg)
        optimizer.zero_grad(set_to_none=False)               # torch 1.6: gradients are zeroed, never released
        loss.backward()
        norm = torch.nn.utils.clip_grad_norm_(params['detr_backbone'] + params['detr_head'], CLIP)
        out[f's{step}:lrs'] = np.array([g['lr'] for g in optimizer.param_groups])
        optimizer.step()
        sched.step()
        st = optimizer.state_dict()['state']
        out[f's{step}:loss'] = np.array(float(loss))
        out[f's{step}:clip_norm'] = np.array(float(norm))
        out[f's{step}:has_state'] = np.array(sorted(st.keys()), dtype=np.int64)
        out[f's{step}:pstep'] = np.array([float(st[k]['step']) for k in sorted(st.keys())])
        for n in flat_names:
            out[f's{step}:p:' + n] = by_name[n].detach().flatten()[out['idx:' + n]].numpy().copy()
        print('step', step, 'loss', float(loss), 'norm', float(norm), 'lrs', out[f's{step}:lrs'], 'state', len(st))
    sd = optimizer.state_dict()
    out['param_groups'] = np.array(json.dumps([{k: (v if k != 'params' else v) for k, v in g.items() if k in ('params', 'lr', 'weight_decay', 'betas', 'eps')}
                                               for g in sd['param_groups']]))
    np.savez_compressed(os.path.join(GOLD, 'optimizer_steps.npz'), **out)
    print('wrote', os.path.join(GOLD, 'optimizer_steps.npz'), os.path.getsize(os.path.join(GOLD, 'optimizer_steps.npz')))


if __name__ == '__main__':
    main()
