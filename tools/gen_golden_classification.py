"""Golden vectors for `answering_type: classification` (exp/gpv/models/gpv.py:384-399: the answer is one vocabulary entry behind
__cls__; everything after encode_answers is the generation path on two tokens): the REAL reference through tools/ref_harness.py on
the small synthetic problem.  Build container only:  python tools/gen_golden_classification.py
Writes tests/golden/classification.json (token strings, ids) and classification.npz (teacher-forced logits, loss terms).  Data only."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import gen_golden as GG                                       # noqa: E402
from tests import synth                                       # noqa: E402


def targets_for(vocab, B, V):
    """the mixed-task targets of the other goldens, with every second answer replaced by a single vocabulary entry (a hit) -- the
    rest are multi-word strings (a miss: __unk__)"""
    targets = synth.synth_targets(B, V, S=6)
    for i, t in enumerate(targets):
        if 'answer' in t and i % 2 == 0:
            t['answer'] = vocab[5 + i]
    return targets


def main():
    torch.set_num_threads(8)
    V, B, H, W, Tl = 40, 4, 96, 128, 5
    cfg = synth.small_cfg(dropout=0.0, answering_type='classification')
    G, model, manifest, vocab = GG.build_reference(cfg, V, bert_layers=2)
    images, mask, ids, attn = synth.synth_batch(B, H, W, Tl, V, pad_to=[(96, 128), (96, 128), (64, 96), (96, 100)])
    targets = targets_for(vocab, B, V)
    toks, tok_ids = model.encode_answers(targets)
    for i, t in enumerate(targets):
        t['answer_token_ids'] = tok_ids[i, 1:]
    model.train()
    model.zero_grad()
    outputs = model(GG.nested(images, mask), (ids, attn), tok_ids, None)
    total, ld = model.criterion(outputs, targets)
    total.backward()
    out = {'token_ids': tok_ids, 'answer_logits': outputs['answer_logits'], 'loss_total': total}
    for k, v in ld.items():
        if v is not None:
            out['loss_' + k] = v if torch.is_tensor(v) else torch.tensor(float(v))
    gn = {n: float(p.grad.norm()) for n, p in model.named_parameters() if p.grad is not None and n.startswith(('text_decoder', 'answer_'))}
    json.dump({'tokens': toks, 'answers': [t.get('answer', '') for t in targets], 'gradnorms': gn},
              open(os.path.join(GG.GOLD, 'classification.json'), 'w'))
    np.savez_compressed(os.path.join(GG.GOLD, 'classification.npz'), **GG.to_np(out))
    print('classification goldens:', toks, tok_ids.tolist(), float(total.detach()))


if __name__ == '__main__':
    main()
