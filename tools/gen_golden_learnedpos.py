"""Golden vectors for `detr.position_embedding: learned` (exp/gpv/models/position_encoding.py:50-75), alone and together with
`pre_norm: true`: the REAL reference through tools/ref_harness.py on the small synthetic problem.  Build container only:
    python tools/gen_golden_learnedpos.py
Writes tests/golden/learnedpos_manifest.json / learnedpos_forward.npz / learnedpos_gradnorms.json and the same three with the
prefix learnedpos_prenorm_.  Data only."""
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


def run(tag, **detr_over):
    V, B, H, W, Tl = 40, 4, 96, 128, 5
    cfg = synth.small_cfg(dropout=0.0)
    cfg['detr'] = dict(cfg['detr'], **detr_over)
    G, model, manifest, vocab = GG.build_reference(cfg, V, bert_layers=2)
    assert 'detr.backbone.1.row_embed.weight' in manifest
    json.dump({'manifest': manifest, 'V': V, 'bert_layers': 2}, open(os.path.join(GG.GOLD, tag + '_manifest.json'), 'w'))
    images, mask, ids, attn = synth.synth_batch(B, H, W, Tl, V, pad_to=[(96, 128), (96, 128), (64, 96), (96, 100)])
    out = {}
    model.eval()
    with torch.no_grad():
        ans_ids = torch.randint(0, V, (B, 5), generator=torch.Generator().manual_seed(5))
        ans_ids[:, 0] = V - 3
        o = model(GG.nested(images, mask), (ids, attn), ans_ids, None)
        out.update({'tf_ans_ids': ans_ids, 'tf_pred_relevance_logits': o['pred_relevance_logits'], 'tf_pred_boxes': o['pred_boxes'],
                    'tf_detr_hs': o['detr_hs'], 'tf_answer_logits': o['answer_logits']})
    model.train()
    targets = synth.synth_targets(B, V, S=6)
    toks, tok_ids = model.encode_answers(targets)
    for i, t in enumerate(targets):
        t['answer_token_ids'] = tok_ids[i, 1:]
    model.zero_grad()
    total, ld = model.criterion(model(GG.nested(images, mask), (ids, attn), tok_ids, None), targets)
    total.backward()
    out['loss_total'] = total
    gn = {n: float(p.grad.norm()) for n, p in model.named_parameters() if p.grad is not None}
    json.dump(gn, open(os.path.join(GG.GOLD, tag + '_gradnorms.json'), 'w'))
    for n in ['detr.backbone.1.row_embed.weight', 'detr.backbone.1.col_embed.weight', 'detr.query_embed.weight',
              'detr.transformer.encoder.layers.0.self_attn.in_proj_weight', 'detr.transformer.decoder.layers.1.multihead_attn.in_proj_weight']:
        g = dict(model.named_parameters())[n].grad
        out['grad:' + n] = g.flatten()[:: max(1, g.numel() // 512)][:512].clone()
    np.savez_compressed(os.path.join(GG.GOLD, tag + '_forward.npz'), **GG.to_np(out))
    print(tag, len(manifest), 'keys, loss', float(total.detach()), 'row/col grad norms', gn['detr.backbone.1.row_embed.weight'], gn['detr.backbone.1.col_embed.weight'])


if __name__ == '__main__':
    torch.set_num_threads(8)
    run('learnedpos', position_embedding='learned')
    run('learnedpos_prenorm', position_embedding='learned', pre_norm=True)
