"""Golden vectors for the `pre_norm: true` branch of the DETR transformer (exp/gpv/models/transformer.py:163-175 forward_pre of the
encoder layer, :234-255 of the decoder layer, :37 the encoder's final LayerNorm): the REAL reference, imported through
tools/ref_harness.py, on the small synthetic problem of tools/gen_golden.py with detr.pre_norm = True.  Build container only:

    python tools/gen_golden_prenorm.py

Writes tests/golden/prenorm_manifest.json (state-dict keys: + detr.transformer.encoder.norm.*), prenorm_forward.npz (teacher-forced
outputs, greedy logits, loss terms, matching, sampled gradients) and prenorm_gradnorms.json.  Data only.
"""
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

GOLD = GG.GOLD


def prenorm_cfg():
    cfg = synth.small_cfg(dropout=0.0)
    cfg['detr'] = dict(cfg['detr'], pre_norm=True)
    return cfg


def main():
    torch.set_num_threads(8)
    V, B, H, W, Tl = 40, 4, 96, 128, 5
    G, model, manifest, vocab = GG.build_reference(prenorm_cfg(), V, bert_layers=2)
    assert 'detr.transformer.encoder.norm.weight' in manifest
    json.dump({'manifest': manifest, 'V': V, 'bert_layers': 2}, open(os.path.join(GOLD, 'prenorm_manifest.json'), 'w'))
    images, mask, ids, attn = synth.synth_batch(B, H, W, Tl, V, pad_to=[(96, 128), (96, 128), (64, 96), (96, 100)])
    out = {}
    model.eval()
    with torch.no_grad():
        ans_ids = torch.randint(0, V, (B, 5), generator=torch.Generator().manual_seed(5))
        ans_ids[:, 0] = V - 3
        o = model(GG.nested(images, mask), (ids, attn), ans_ids, None)
        out.update({'tf_ans_ids': ans_ids, 'tf_pred_relevance_logits': o['pred_relevance_logits'], 'tf_pred_boxes': o['pred_boxes'],
                    'tf_detr_hs': o['detr_hs'], 'tf_answer_logits': o['answer_logits']})
        o = model(GG.nested(images, mask), (ids, attn), None, None)
        out.update({'greedy_answer_logits': o['answer_logits'], 'greedy_top1': o['answer_logits'][-1].topk(1, -1).indices[..., 0]})
    model.train()
    targets = synth.synth_targets(B, V, S=6)
    toks, tok_ids = model.encode_answers(targets)
    for i, t in enumerate(targets):
        t['answer_token_ids'] = tok_ids[i, 1:]
    model.zero_grad()
    outputs = model(GG.nested(images, mask), (ids, attn), tok_ids, None)
    total, ld = model.criterion(outputs, targets)
    total.backward()
    out['loss_total'] = total
    for k, v in ld.items():
        if v is not None:
            out['loss_' + k] = v if torch.is_tensor(v) else torch.tensor(float(v))
    idxs = [i for i, t in enumerate(targets) if 'boxes' in t]
    ind = model.criterion.localization_criterion.matcher(
        {'pred_relevance_logits': outputs['pred_relevance_logits'][idxs], 'pred_boxes': outputs['pred_boxes'][idxs]}, [targets[i] for i in idxs])
    out['match_pred'] = torch.cat([a for a, _ in ind])
    out['match_tgt'] = torch.cat([b for _, b in ind])
    gn = {n: float(p.grad.norm()) for n, p in model.named_parameters() if p.grad is not None}
    json.dump(gn, open(os.path.join(GOLD, 'prenorm_gradnorms.json'), 'w'))
    for n in ['detr.transformer.encoder.norm.weight', 'detr.transformer.encoder.layers.0.self_attn.in_proj_weight',
              'detr.transformer.encoder.layers.1.linear1.weight', 'detr.transformer.decoder.layers.1.multihead_attn.out_proj.weight',
              'detr.transformer.decoder.layers.0.norm2.weight', 'detr.transformer.decoder.norm.bias', 'detr.query_embed.weight',
              'detr.input_proj.weight', 'detr.backbone.0.body.layer4.2.conv3.weight']:
        g = dict(model.named_parameters())[n].grad
        out['grad:' + n] = g.flatten()[:: max(1, g.numel() // 512)][:512].clone()
    np.savez_compressed(os.path.join(GOLD, 'prenorm_forward.npz'), **GG.to_np(out))
    print('pre_norm goldens written:', len(manifest), 'keys,', len(gn), 'gradients, loss', float(total))


if __name__ == '__main__':
    main()
