"""The CPU oracle (oracle/gpv_oracle.py) against golden vectors produced by running the REAL
reference (tools/gen_golden.py, build container).  fp32, tolerance 1e-4 abs/rel on activations
(same math, different op order), bit-exact on every integer output."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import gpv_oracle as O
from tests import synth

GOLD = os.path.join(os.path.dirname(__file__), 'golden')
V, B, H, W, Tl = 40, 4, 96, 128, 5


@pytest.fixture(scope='module')
def small():
    man = json.load(open(os.path.join(GOLD, 'small_manifest.json')))
    Pm = synth.synth_state(man['manifest'])
    Pm['pos_enc'] = torch.zeros(1, 30, 768)
    cfg = synth.small_cfg(dropout=0.0)
    cfg['_cls_id'] = V - 3
    gold = dict(np.load(os.path.join(GOLD, 'small_forward.npz')))
    batch = synth.synth_batch(B, H, W, Tl, V, pad_to=[(96, 128), (96, 128), (64, 96), (96, 100)])
    return Pm, cfg, gold, batch


def close(a, b, tol=1e-4):
    a = torch.as_tensor(a, dtype=torch.float32)
    b = torch.as_tensor(b, dtype=torch.float32)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs().max().item()
    scale = max(b.abs().max().item(), 1.0)
    assert err <= tol * scale, f'max err {err} (scale {scale})'


def test_teacher_forced_forward(small):
    Pm, cfg, gold, (images, mask, ids, attn) = small
    with torch.no_grad():
        o = O.gpv_forward(Pm, cfg, images, mask, ids, attn, torch.as_tensor(gold['tf_ans_ids']))
    close(o['pred_boxes'], gold['tf_pred_boxes'])
    close(o['pred_relevance_logits'], gold['tf_pred_relevance_logits'])
    close(o['detr_hs'], gold['tf_detr_hs'])
    close(o['answer_logits'], gold['tf_answer_logits'])


def test_greedy(small):
    Pm, cfg, gold, (images, mask, ids, attn) = small
    with torch.no_grad():
        o = O.gpv_forward(Pm, cfg, images, mask, ids, attn, None)
        close(o['answer_logits'], gold['greedy_answer_logits'])
        assert np.array_equal(o['answer_logits'][-1].topk(1, -1).indices[..., 0].numpy(), gold['greedy_top1'])
        o = O.gpv_forward(Pm, cfg, images, mask, ids, attn, None, vocab_mask=torch.as_tensor(gold['vocab_mask']))
        close(o['answer_logits'], gold['greedy_vm_answer_logits'])


def test_beam(small):
    Pm, cfg, gold, (images, mask, ids, attn) = small
    ref = json.load(open(os.path.join(GOLD, 'small_beam.json')))
    with torch.no_grad():
        _, memory = O.gpv_encode(Pm, cfg, images, mask, ids, attn)
        answers, probs, _ = O.beam_search(Pm, cfg, memory, 3, synth.make_vocab(V))
    assert answers == ref['answers']
    close(torch.tensor(probs), torch.tensor(ref['answer_probs']), 1e-4)


def test_sine_position(small):
    _, _, gold, _ = small
    close(O.sine_position(torch.as_tensor(gold['pos_mask'])), gold['pos_sine'], 1e-5)
    _, mask, _, _ = synth.synth_batch(B, H, W, Tl, V, pad_to=[(96, 128), (96, 128), (64, 96), (96, 100)])
    assert np.array_equal(O.downsample_mask(mask, 3, 4).numpy(), gold['pos_mask'])


def test_loss_grads_and_matching(small):
    Pm, cfg, gold, (images, mask, ids, attn) = small
    tj = json.load(open(os.path.join(GOLD, 'small_targets.json')))
    targets = synth.synth_targets(B, V, S=6)
    word_to_idx = {w: i for i, w in enumerate(synth.make_vocab(V))}
    toks, tok_ids = O.encode_answers(targets, word_to_idx, cfg['max_text_len'])
    assert toks == tj['tokens']
    assert np.array_equal(tok_ids.numpy(), gold['enc_token_ids'])
    for i, t in enumerate(targets):
        t['answer_token_ids'] = tok_ids[i, 1:]
    man = json.load(open(os.path.join(GOLD, 'full_manifest.json')))
    gn = json.load(open(os.path.join(GOLD, 'small_gradnorms.json')))
    leaves = {k: v.clone().requires_grad_(True) for k, v in Pm.items() if k in gn}
    Pg = dict(Pm)
    Pg.update(leaves)
    out = O.gpv_forward(Pg, cfg, images, mask, ids, attn, tok_ids, training=True)
    total, ld = O.gpv_criterion(out, targets, cfg['losses'])
    close(total.detach(), gold['loss_total'], 1e-5)
    close(total.detach(), gold['loss_total_via_forward'], 1e-5)
    for k in ('loss_caption', 'loss_vqa', 'loss_cls', 'loss_ce', 'loss_bbox', 'loss_giou'):
        close(ld[k].detach(), gold['loss_' + k], 1e-5)
    ind = ld['_indices']
    assert np.array_equal(torch.cat([a for a, _ in ind]).numpy(), gold['match_pred'])
    assert np.array_equal(torch.cat([b for _, b in ind]).numpy(), gold['match_tgt'])
    total.backward()
    # every parameter the reference gives a gradient to gets the same gradient norm ...
    for n, ref in gn.items():
        g = leaves[n].grad
        assert g is not None, n
        if n == 'answer_head.classifier_transform.bias':
            continue      # shifts every vocab logit equally -> exact gradient is 0, value is roundoff
        assert abs(float(g.norm()) - ref) <= 2e-3 * ref + 1e-6, (n, float(g.norm()), ref)
    # ... and sampled gradient entries agree
    for k in gold:
        if k.startswith('grad:'):
            g = leaves[k[5:]].grad
            close(g.flatten()[:: max(1, g.numel() // 512)][:512], gold[k], 2e-4)


def test_matcher_and_set_criterion_golden():
    g = dict(np.load(os.path.join(GOLD, 'matcher.npz')))
    logits, boxes = torch.as_tensor(g['logits']), torch.as_tensor(g['boxes'])
    sizes = g['sizes'].tolist()
    tb = torch.as_tensor(g['tgt_boxes']).split(sizes)
    tg = [{'labels': torch.zeros(n, dtype=torch.long), 'boxes': b} for n, b in zip(sizes, tb)]
    ind, _ = O.hungarian_match(logits, boxes, tg)
    assert np.array_equal(torch.cat([a for a, _ in ind]).numpy(), g['pred_idx'])   # bit-exact, incl. ties
    assert np.array_equal(torch.cat([b for _, b in ind]).numpy(), g['tgt_idx'])
    close(O.generalized_box_iou(O.box_cxcywh_to_xyxy(boxes[0]), O.box_cxcywh_to_xyxy(tg[2]['boxes'])), g['giou'], 1e-6)
    r = O.set_criterion(logits, boxes, tg)
    for k in ('loss_ce', 'loss_bbox', 'loss_giou'):
        close(r[k], g['sc_' + k], 1e-6)


def test_roi_align_separable_equals_direct():
    torch.manual_seed(0)
    feat = torch.randn(2, 8, 15, 20)
    boxes = torch.rand(2, 12, 4) * torch.tensor([0.6, 0.6, 0.5, 0.5]) + torch.tensor([0.2, 0.2, 0.02, 0.02])
    boxes[0, 0] = torch.tensor([0.02, 0.03, 0.3, 0.2])      # hangs over the top-left border
    boxes[0, 1] = torch.tensor([0.98, 0.97, 0.4, 0.3])      # bottom-right border
    boxes[1, 2] = torch.tensor([0.5, 0.5, 1.0, 1.0])        # whole map
    boxes[1, 3] = torch.tensor([0.5, 0.5, 1e-4, 1e-4])      # tiny
    xy = O.box_cxcywh_to_xyxy(boxes) * torch.tensor([20., 15., 20., 15.])
    a = O.extract_roi(feat, boxes)
    b = torch.stack([O.roi_align_mean_direct(feat[i], xy[i]) for i in range(2)])
    close(a, b, 1e-5)


def test_state_dict_manifest_836_keys():
    m = json.load(open(os.path.join(GOLD, 'full_manifest.json')))
    assert m['n_keys'] == 836 and m['n_trainable'] == 224012552
    pre = {}
    for k in m['manifest']:
        pre[k.split('.')[0]] = pre.get(k.split('.')[0], 0) + 1
    assert pre['detr'] == 458 and pre['bert'] == 199 and pre['co_att_transformer'] == 108 and pre['text_decoder'] == 54


def test_bert_against_hf():
    tr = pytest.importorskip('transformers')
    torch.manual_seed(0)
    hf = tr.BertModel(tr.BertConfig(num_hidden_layers=2)).eval()
    Pm = {'bert.model.' + k: v for k, v in hf.state_dict().items()}
    _, _, ids, attn = synth.synth_batch(3, 8, 8, 7, V)
    with torch.no_grad():
        ref = hf(input_ids=ids, attention_mask=attn)[0]
        got = O.bert_forward(Pm, ids, attn)
    valid = attn.bool()
    close(got[valid], ref[valid], 1e-4)
    close(got, ref, 1e-4)     # padded query positions too: the reference attends them downstream
