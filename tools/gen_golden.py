"""Golden-vector generator: runs the REAL reference (/root/reference, imported through
tools/ref_harness.py) on deterministic synthetic weights/inputs (tests/synth.py) and stores its
outputs under tests/golden/.  Run in the build container only:

    python tools/gen_golden.py

Fixtures hold data only (name->shape manifests, outputs); no reference source is copied.
"""
import json
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import ref_harness as RH                                     # noqa: E402
from tests import synth                                       # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')


def build_reference(cfg_dict, V, bert_layers):
    G = RH.install()
    tmp = tempfile.mkdtemp()
    vocab = synth.make_vocab(V)
    with open(os.path.join(tmp, 'vocab.json'), 'w') as f:
        json.dump(vocab, f)
    np.save(os.path.join(tmp, 'vocab_embed.npy'),
            synth.synth_tensor('answer_head.vocab_embed', (V, cfg_dict['bert_joiner']['bert_dim'])).numpy())
    cfg_dict = dict(cfg_dict, vocab=os.path.join(tmp, 'vocab.json'),
                    vocab_embed=os.path.join(tmp, 'vocab_embed.npy'))
    G.Bert = lambda cfg=None: RH.FakeBert(cfg, bert_layers, dropout=0.0)
    torch.manual_seed(0)
    model = G.GPV(RH.AttrDict.wrap(cfg_dict))
    manifest = {k: [list(v.shape), str(v.dtype)] for k, v in model.state_dict().items()}
    st = synth.synth_state(manifest)
    missing = model.load_state_dict(st, strict=False)
    assert set(missing.missing_keys) <= {'pos_enc', 'criterion.localization_criterion.set_criterion.empty_weight'}, missing
    return G, model, manifest, vocab


def nested(images, mask):
    from utils.detr_misc import NestedTensor
    return NestedTensor(images, mask)


def to_np(d):
    return {k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


def main():
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(8)
    V, B, H, W, Tl = 40, 4, 96, 128, 5

    # ---------------- small model: forward / greedy / beam / loss+grads -------------
    cfg = synth.small_cfg(dropout=0.0)
    G, model, manifest, vocab = build_reference(cfg, V, bert_layers=2)
    json.dump({'manifest': manifest, 'V': V, 'bert_layers': 2},
              open(os.path.join(GOLD, 'small_manifest.json'), 'w'))
    images, mask, ids, attn = synth.synth_batch(B, H, W, Tl, V, pad_to=[(96, 128), (96, 128), (64, 96), (96, 100)])
    out = {}
    model.eval()
    with torch.no_grad():
        # teacher-forced (targets=None -> outputs dict)
        ans_ids = torch.randint(0, V, (B, 5), generator=torch.Generator().manual_seed(5))
        ans_ids[:, 0] = V - 3
        o = model(nested(images, mask), (ids, attn), ans_ids, None)
        out.update({'tf_ans_ids': ans_ids, 'tf_pred_relevance_logits': o['pred_relevance_logits'],
                    'tf_pred_boxes': o['pred_boxes'], 'tf_detr_hs': o['detr_hs'],
                    'tf_answer_logits': o['answer_logits']})
        # greedy with and without vocab mask
        o = model(nested(images, mask), (ids, attn), None, None)
        out.update({'greedy_answer_logits': o['answer_logits'],
                    'greedy_top1': o['answer_logits'][-1].topk(1, -1).indices[..., 0]})
        vm = torch.zeros(V)
        vm[::3] = -10000.0
        o = model(nested(images, mask), (ids, attn), None, None, vocab_mask=vm)
        out.update({'vocab_mask': vm, 'greedy_vm_answer_logits': o['answer_logits']})
        # beam
        o = model.forward_beam_search(nested(images, mask), (ids, attn), beam_size=3)
        json.dump({'answers': o['answers'], 'answer_probs': o['answer_probs']},
                  open(os.path.join(GOLD, 'small_beam.json'), 'w'))
        # pieces: sine position embedding on the ragged mask, at c5 resolution
        from exp.gpv.models.position_encoding import PositionEmbeddingSine
        import torch.nn.functional as F
        m = F.interpolate(mask[None].float(), size=(3, 4)).to(torch.bool)[0]
        out['pos_mask'] = m
        out['pos_sine'] = PositionEmbeddingSine(128, normalize=True)(nested(torch.zeros(B, 1, 3, 4), m))

    # loss + gradients, train mode with all dropout = 0 (config fields), mixed tasks
    model.train()
    targets = synth.synth_targets(B, V, S=6)
    toks, tok_ids = model.encode_answers(targets)
    for i, t in enumerate(targets):
        t['answer_token_ids'] = tok_ids[i, 1:]
    out['enc_token_ids'] = tok_ids
    json.dump({'targets': [{k: (v.tolist() if torch.is_tensor(v) else v) for k, v in t.items()} for t in targets],
               'tokens': toks}, open(os.path.join(GOLD, 'small_targets.json'), 'w'))
    model.zero_grad()
    outputs = model(nested(images, mask), (ids, attn), tok_ids, None)
    total, ld = model.criterion(outputs, targets)
    total.backward()
    out['loss_total'] = total
    for k, v in ld.items():
        if v is not None:
            out['loss_' + k] = v if torch.is_tensor(v) else torch.tensor(float(v))
    idxs = [i for i, t in enumerate(targets) if 'boxes' in t]
    ind = model.criterion.localization_criterion.matcher(
        {'pred_relevance_logits': outputs['pred_relevance_logits'][idxs], 'pred_boxes': outputs['pred_boxes'][idxs]},
        [targets[i] for i in idxs])
    out['match_pred'] = torch.cat([a for a, _ in ind])
    out['match_tgt'] = torch.cat([b for _, b in ind])
    # same forward through model.forward(targets) must give the same scalar (gpv.py:203-207)
    out['loss_total_via_forward'] = model(nested(images, mask), (ids, attn), tok_ids, targets)
    gn = {}
    for n, p in model.named_parameters():
        if p.grad is not None:
            gn[n] = float(p.grad.norm())
    json.dump(gn, open(os.path.join(GOLD, 'small_gradnorms.json'), 'w'))
    for n in ['detr.backbone.0.body.layer4.2.conv3.weight', 'detr.backbone.0.body.layer2.0.conv1.weight',
              'detr.transformer.encoder.layers.0.self_attn.in_proj_weight',
              'detr.transformer.decoder.layers.1.multihead_attn.out_proj.weight',
              'detr.class_embed.weight', 'detr.bbox_embed.layers.2.weight', 'detr.input_proj.weight',
              'detr_joiner.weight', 'bert_joiner.weight', 'co_att_transformer.0.biattention.query1.weight',
              'co_att_transformer.1.v_output.dense.weight', 'co_att_transformer.1.biOutput.LayerNorm2.weight',
              'relevance_tokens', 'relevance_predictor.weight', 'text_decoder.layers.0.self_attn.in_proj_weight',
              'text_decoder.layers.1.linear2.bias', 'answer_head.classifier_transform.weight',
              'answer_input_embedings.transform.weight', 'detr.query_embed.weight']:
        g = dict(model.named_parameters())[n].grad
        out['grad:' + n] = g.flatten()[:: max(1, g.numel() // 512)][:512].clone()
    np.savez_compressed(os.path.join(GOLD, 'small_forward.npz'), **to_np(out))

    # ---------------- matcher: separated + near-tie costs ---------------------------
    from utils.matcher import HungarianMatcher
    g = torch.Generator().manual_seed(7)
    mo = {}
    hm = HungarianMatcher(1, 5, 2)
    Bm, Q = 3, 12
    logits = torch.randn(Bm, Q, 2, generator=g)
    boxes = torch.cat((0.2 + 0.6 * torch.rand(Bm, Q, 2, generator=g), 0.05 + 0.3 * torch.rand(Bm, Q, 2, generator=g)), -1)
    boxes[1, 3] = boxes[1, 7]                       # exact duplicate predictions -> tied costs
    logits[1, 3] = logits[1, 7]
    tg = []
    for n in (4, 1, 12):
        tb = torch.cat((0.2 + 0.6 * torch.rand(n, 2, generator=g), 0.05 + 0.3 * torch.rand(n, 2, generator=g)), -1)
        tg.append({'labels': torch.zeros(n, dtype=torch.long), 'boxes': tb})
    tg[2]['boxes'][5] = tg[2]['boxes'][6]            # duplicate targets
    ind = hm({'pred_relevance_logits': logits, 'pred_boxes': boxes}, tg)
    mo.update({'logits': logits, 'boxes': boxes, 'sizes': torch.tensor([4, 1, 12]),
               'tgt_boxes': torch.cat([t['boxes'] for t in tg]),
               'pred_idx': torch.cat([a for a, _ in ind]), 'tgt_idx': torch.cat([b for _, b in ind])})
    from utils.box_ops import generalized_box_iou, box_cxcywh_to_xyxy
    mo['giou'] = generalized_box_iou(box_cxcywh_to_xyxy(boxes[0]), box_cxcywh_to_xyxy(tg[2]['boxes']))
    from utils.set_criterion import SetCriterion
    sc = SetCriterion(1, hm, None, 0.1, ['labels', 'boxes'])
    l = sc({'pred_relevance_logits': logits, 'pred_boxes': boxes}, tg)
    mo.update({'sc_loss_ce': l['loss_ce'], 'sc_loss_bbox': l['loss_bbox'], 'sc_loss_giou': l['loss_giou']})
    np.savez_compressed(os.path.join(GOLD, 'matcher.npz'), **to_np(mo))

    # ---------------- full-config state-dict manifest (836 keys) ---------------------
    Vf = 64
    G, fmodel, fman, _ = build_reference(synth.model_cfg(), Vf, bert_layers=12)
    json.dump({'manifest': fman, 'V': Vf, 'n_keys': len(fman),
               'n_params': sum(p.numel() for p in fmodel.parameters()),
               'n_trainable': sum(p.numel() for p in fmodel.parameters() if p.requires_grad),
               'trainable': [n for n, p in fmodel.named_parameters() if p.requires_grad]},
              open(os.path.join(GOLD, 'full_manifest.json'), 'w'))
    print('full model keys', len(fman))
    print('done')


if __name__ == '__main__':
    main()
