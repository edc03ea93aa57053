"""`answering_type: classification` (exp/gpv/models/gpv.py:384-399): [__cls__, answer-as-one-vocabulary-entry] instead of the
tokenised sentence; the rest of the path is the generation path on two tokens.  Goldens: the REAL reference
(tools/gen_golden_classification.py).  Oracle and the product (cpu shim here, HIP on the GPU) against them."""
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


def _targets(vocab):
    targets = synth.synth_targets(B, V, S=6)
    for i, t in enumerate(targets):
        if 'answer' in t and i % 2 == 0:
            t['answer'] = vocab[5 + i]
    return targets


def close(a, b, tol=1e-4):
    a = torch.as_tensor(np.asarray(a.detach().float().cpu() if torch.is_tensor(a) else a), dtype=torch.float32)
    b = torch.as_tensor(b, dtype=torch.float32)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert (a - b).abs().max().item() <= tol * max(b.abs().max().item(), 1.0)


@pytest.fixture(scope='module')
def fx():
    meta = json.load(open(os.path.join(GOLD, 'classification.json')))
    gold = dict(np.load(os.path.join(GOLD, 'classification.npz')))
    man = json.load(open(os.path.join(GOLD, 'small_manifest.json')))
    return meta, gold, man, synth.synth_batch(B, H, W, Tl, V, pad_to=PADS)


def test_oracle_classification_vs_reference(fx):
    meta, gold, man, (images, mask, ids, attn) = fx
    vocab = synth.make_vocab(V)
    targets = _targets(vocab)
    assert [t.get('answer', '') for t in targets] == meta['answers']
    cfg = synth.small_cfg(dropout=0.0, answering_type='classification')
    cfg['_cls_id'] = V - 3
    toks, tok_ids = O.encode_answers(targets, {w: i for i, w in enumerate(vocab)}, cfg['max_text_len'], answering_type='classification')
    assert toks == meta['tokens'] and np.array_equal(tok_ids.numpy(), gold['token_ids'])
    for i, t in enumerate(targets):
        t['answer_token_ids'] = tok_ids[i, 1:]
    Pm = synth.synth_state(man['manifest'])
    Pm['pos_enc'] = torch.zeros(1, 30, 768)
    with torch.no_grad():
        out = O.gpv_forward(Pm, cfg, images, mask, ids, attn, tok_ids, training=True)
        total, ld = O.gpv_criterion(out, targets, cfg['losses'])
    close(out['answer_logits'], gold['answer_logits'])
    close(total, gold['loss_total'], 1e-5)


def _product(man, device):
    from gpv1_amd.gpv import GPV
    cfg = synth.small_cfg(dropout=0.0, answering_type='classification')
    cfg['vocab'] = synth.make_vocab(V)
    cfg['vocab_embed'] = synth.synth_tensor('answer_head.vocab_embed', (V, 768))
    cfg['bert_layers'] = 2
    model = GPV(cfg)
    missing = model.load_state_dict(synth.synth_state(man['manifest']), strict=False)
    assert not missing.unexpected_keys
    model.bert.model.p = 0.0
    return model.to(device).train()


def _run_product(fx, device, tol):
    from gpv1_amd.misc import NestedTensor
    meta, gold, man, (images, mask, ids, attn) = fx
    model = _product(man, device)
    targets = _targets(model.vocab)
    toks, tok_ids = model.encode_answers(targets)
    assert toks == meta['tokens'] and np.array_equal(tok_ids.cpu().numpy(), gold['token_ids'])
    for i, t in enumerate(targets):
        t['answer_token_ids'] = tok_ids[i, 1:]
        for k, v in t.items():
            if torch.is_tensor(v):
                t[k] = v.to(device)
    images, mask, ids, attn = images.to(device), mask.to(device), ids.to(device), attn.to(device)
    model.zero_grad()
    outputs = model(NestedTensor(images, mask), (ids, attn), tok_ids, None)
    total, ld = model.criterion(outputs, targets)
    total.backward()
    close(outputs['answer_logits'], gold['answer_logits'], tol)
    close(total, gold['loss_total'], tol)
    params = dict(model.named_parameters())
    for n, ref in meta['gradnorms'].items():
        if n == 'answer_head.classifier_transform.bias':
            continue
        assert abs(float(params[n].grad.float().norm()) - ref) <= 50 * tol * ref + 1e-6, n
    # the graphed trainer's padded encoding: two tokens padded to the size class, the pads ignored by the loss
    tok_c, tgt_c, S = model.encode_answers_classed(targets)
    assert S == 2 and tok_c.shape[1] == 4 and (tgt_c[:, 1:] == -100).all() and torch.equal(tgt_c[:, 0], tok_ids[:, 1].to(tgt_c.device))


def test_product_classification_on_the_cpu_shim_vs_reference(fx):
    from tests import cpu_shim
    import gpv1_amd.ops as ops
    undo = cpu_shim.install()
    ops.RT.set_precise(True)
    try:
        _run_product(fx, 'cpu', 1e-4)
    finally:
        ops.RT.set_precise(False)
        undo()


@pytest.mark.gpu
def test_hip_classification_vs_reference(fx):
    import gpv1_amd.ops as ops
    ops.RT.set_precise(True)
    try:
        _run_product(fx, 'cuda', 1e-4)
    finally:
        ops.RT.set_precise(False)


def test_linear_answer_head_is_refused_with_the_reason():
    from gpv1_amd.gpv import GPV
    cfg = synth.small_cfg(dropout=0.0, answer_head='linear')
    cfg['vocab'] = synth.make_vocab(V)
    cfg['vocab_embed'] = synth.synth_tensor('answer_head.vocab_embed', (V, 768))
    cfg['bert_layers'] = 2
    with pytest.raises(NotImplementedError, match='shape-inconsistent'):
        GPV(cfg)
