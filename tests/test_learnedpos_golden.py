"""`detr.position_embedding: learned` (exp/gpv/models/position_encoding.py:50-75), alone and with `pre_norm: true`, against goldens
of the REAL reference (tools/gen_golden_learnedpos.py).  The point of the branch is the GRADIENT of the two tables: the DETR
transformer takes the sine encoding as a constant of its LayerNorm kernels, a learned one must be differentiated through every
`x + pos` of every layer."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import gpv_oracle as O
from tests import synth

GOLD = os.path.join(os.path.dirname(__file__), 'golden')
V, B, H, W, Tl = 40, 4, 96, 128, 5
PADS = [(96, 128), (96, 128), (64, 96), (96, 100)]
VARIANTS = {'learnedpos': {'position_embedding': 'learned'}, 'learnedpos_prenorm': {'position_embedding': 'learned', 'pre_norm': True}}


def cfg_of(tag):
    cfg = synth.small_cfg(dropout=0.0)
    cfg['detr'] = dict(cfg['detr'], **VARIANTS[tag])
    return cfg


def close(a, b, tol=1e-4):
    a = torch.as_tensor(np.asarray(a.detach().float().cpu() if torch.is_tensor(a) else a), dtype=torch.float32)
    b = torch.as_tensor(b, dtype=torch.float32)
    assert a.shape == b.shape, (a.shape, b.shape)
    err, scale = (a - b).abs().max().item(), max(b.abs().max().item(), 1.0)
    assert err <= tol * scale, f'max err {err} (scale {scale})'


def load(tag):
    man = json.load(open(os.path.join(GOLD, tag + '_manifest.json')))
    gold = dict(np.load(os.path.join(GOLD, tag + '_forward.npz')))
    gn = json.load(open(os.path.join(GOLD, tag + '_gradnorms.json')))
    return man, gold, gn, synth.synth_batch(B, H, W, Tl, V, pad_to=PADS)


def _targets(tok_fn):
    targets = synth.synth_targets(B, V, S=6)
    toks, tok_ids = tok_fn(targets)
    for i, t in enumerate(targets):
        t['answer_token_ids'] = tok_ids[i, 1:]
    return targets, tok_ids


@pytest.mark.parametrize('tag', list(VARIANTS))
def test_oracle_learned_position_vs_reference(tag):
    man, gold, gn, (images, mask, ids, attn) = load(tag)
    Pm = synth.synth_state(man['manifest'])
    Pm['pos_enc'] = torch.zeros(1, 30, 768)
    cfg = cfg_of(tag)
    cfg['_cls_id'] = V - 3
    with torch.no_grad():
        o = O.gpv_forward(Pm, cfg, images, mask, ids, attn, torch.as_tensor(gold['tf_ans_ids']))
        for k in ('pred_boxes', 'pred_relevance_logits', 'detr_hs', 'answer_logits'):
            close(o[k], gold['tf_' + k])
    word_to_idx = {w: i for i, w in enumerate(synth.make_vocab(V))}
    targets, tok_ids = _targets(lambda t: O.encode_answers(t, word_to_idx, cfg['max_text_len']))
    leaves = {k: v.clone().requires_grad_(True) for k, v in Pm.items() if k in gn}
    Pg = dict(Pm)
    Pg.update(leaves)
    total, ld = O.gpv_criterion(O.gpv_forward(Pg, cfg, images, mask, ids, attn, tok_ids, training=True), targets, cfg['losses'])
    close(total.detach(), gold['loss_total'], 1e-5)
    total.backward()
    for n, ref in gn.items():
        if n == 'answer_head.classifier_transform.bias':
            continue
        assert abs(float(leaves[n].grad.norm()) - ref) <= 2e-3 * ref + 1e-6, (n, float(leaves[n].grad.norm()), ref)
    for k in gold:
        if k.startswith('grad:'):
            g = leaves[k[5:]].grad
            close(g.flatten()[:: max(1, g.numel() // 512)][:512], gold[k], 2e-4)


def _build(tag, man):
    from gpv1_amd.gpv import GPV
    cfg = cfg_of(tag)
    cfg['vocab'] = synth.make_vocab(V)
    cfg['vocab_embed'] = synth.synth_tensor('answer_head.vocab_embed', (V, 768))
    cfg['bert_layers'] = 2
    model = GPV(cfg)
    missing = model.load_state_dict(synth.synth_state(man['manifest']), strict=False)
    assert set(missing.missing_keys) <= {'pos_enc', 'criterion.localization_criterion.set_criterion.empty_weight'}, missing
    assert not missing.unexpected_keys, missing.unexpected_keys
    model.bert.model.p = 0.0
    return model


def _run_product(tag, device, precise):
    from gpv1_amd.misc import NestedTensor
    man, gold, gn, (images, mask, ids, attn) = load(tag)
    model = _build(tag, man).to(device)
    images, mask, ids, attn = images.to(device), mask.to(device), ids.to(device), attn.to(device)
    tol = 1e-4 if precise else 5e-2
    model.eval()
    with torch.no_grad():
        o = model(NestedTensor(images, mask), (ids, attn), torch.as_tensor(gold['tf_ans_ids']).to(device), None)
        for k in ('pred_boxes', 'pred_relevance_logits', 'detr_hs', 'answer_logits'):
            close(o[k], gold['tf_' + k], tol)
    model.train()
    targets, tok_ids = _targets(model.encode_answers)
    for t in targets:
        for k, v in t.items():
            if torch.is_tensor(v):
                t[k] = v.to(device)
    model.zero_grad()
    total, ld = model.criterion(model(NestedTensor(images, mask), (ids, attn), tok_ids.to(device), None), targets)
    total.backward()
    close(total, gold['loss_total'], 1e-4 if precise else 3e-2)
    params = dict(model.named_parameters())
    gmax = max(gn.values())
    rtol, floor = (5e-3, 1e-5 * gmax) if precise else (0.2, 2e-3 * gmax)
    bad = []
    for n, ref in gn.items():
        g = params[n].grad
        assert g is not None, n                       # (the two position tables included)
        if abs(float(g.float().norm()) - ref) > rtol * ref + floor:
            bad.append((n, float(g.float().norm()), ref))
    assert not bad, bad[:10]
    if precise:
        for k in gold:
            if k.startswith('grad:'):
                g = params[k[5:]].grad
                close(g.flatten()[:: max(1, g.numel() // 512)][:512], gold[k], 2e-3)


@pytest.mark.parametrize('tag', list(VARIANTS))
def test_product_learned_position_on_the_cpu_shim_vs_reference(tag):
    from tests import cpu_shim
    import gpv1_amd.ops as ops
    undo = cpu_shim.install()
    ops.RT.set_precise(True)
    try:
        _run_product(tag, 'cpu', True)
    finally:
        ops.RT.set_precise(False)
        undo()


@pytest.mark.gpu
@pytest.mark.parametrize('precise', [True, False])
@pytest.mark.parametrize('tag', list(VARIANTS))
def test_hip_learned_position_vs_reference(tag, precise):
    import gpv1_amd.ops as ops
    ops.RT.set_precise(precise)
    try:
        _run_product(tag, 'cuda', precise)
    finally:
        ops.RT.set_precise(False)
