"""SURVEY 8(f)-2 against fixtures produced by the REFERENCE's own harness functions (tools/gen_golden_harness.py runs
inference.decode_outputs, compute_predictions.create_coco_vocab_mask and compute_predictions.make_predictions from
/root/reference on the small fixture): box order by softmax(relevance)[..., 0] incl. a tie, top-1 answer cut at __stop__ / __pad__,
the classification vocabulary mask with the real class table (and a subset of its synonyms), predictions JSON + boxes file layout.
Stand-ins on the reference side are listed in tests/golden/harness.json['stand_ins'] (detokenisation stays unpinned)."""
import json
import os
import types

import numpy as np
import pytest
import torch

from tests import synth, cpu_shim
from tests.test_model_cpu import build_small, nested, GOLD, V, B, H, W, Tl, PAD

META = json.load(open(os.path.join(GOLD, 'harness.json')))
ARR = dict(np.load(os.path.join(GOLD, 'harness.npz')))


class _Vocab:
    def __init__(self, vocab):
        self.vocab = vocab
        self.word_to_idx = {w: i for i, w in enumerate(vocab)}

    def token_ids_to_words(self, ids):
        return [[self.vocab[int(j)] for j in row] for row in ids]


def _outputs(rel_logits, boxes, top1, nvocab):
    onehot = torch.zeros(1, *top1.shape, nvocab)
    onehot.scatter_(-1, torch.as_tensor(top1)[None, ..., None], 1.0)
    return {'pred_relevance_logits': torch.as_tensor(rel_logits), 'pred_boxes': torch.as_tensor(boxes), 'answer_logits': onehot}


def test_decode_outputs_matches_reference_function():
    from gpv1_amd.inference import decode_outputs
    m = _Vocab(synth.make_vocab(V))
    dec = decode_outputs(_outputs(ARR['dec_in_relevance_logits'], ARR['dec_in_boxes'], ARR['dec_in_top1'], V), m)
    assert [d['answer'] for d in dec] == META['dec_answers']
    assert np.array_equal(np.stack([d['boxes'] for d in dec]), ARR['dec_boxes'])
    assert np.allclose(np.stack([d['relevance'] for d in dec]), ARR['dec_relevance'], rtol=0, atol=1e-7)
    assert all(d['boxes'].dtype == np.float32 and d['relevance'].dtype == np.float32 for d in dec)
    # equal relevance: the reference's sort is stable on the score
    tie = decode_outputs(_outputs(ARR['tie_in_relevance_logits'], ARR['dec_in_boxes'], ARR['dec_in_top1'], V), m)
    assert np.array_equal(tie[0]['boxes'], ARR['tie_boxes0'])


def test_vocab_mask_matches_reference_function():
    from gpv1_amd.compute_predictions import create_vocab_mask, COCO_CLASSES
    assert list(COCO_CLASSES) == META['mask_classes']                   # same class names, same order as the reference's table
    m = _Vocab(META['mask_vocab'])
    toks, mask = create_vocab_mask(m)
    assert toks == META['mask_tokens'] and np.array_equal(mask, ARR['mask']) and mask.dtype == np.float32
    toks, mask = create_vocab_mask(m, synonyms=META['mask_synonyms_subset'], use_syns=True)
    assert toks == META['mask_tokens_syn_subset'] and np.array_equal(mask, ARR['mask_syn_subset'])


def _run_make_predictions(model, task, tmp, dev):
    from gpv1_amd import compute_predictions as cp
    images, mask, ids, attn = synth.synth_batch(B, H, W, Tl, V, pad_to=PAD)
    ref = META['pred_' + task]
    sids = ref['sample_ids']
    batches = [(nested(images[i:i + 2].to(dev), mask[i:i + 2].to(dev)), (ids[i:i + 2].to(dev), attn[i:i + 2].to(dev)), sids[i:i + 2]) for i in (0, 2)]
    preds, jpath, bpath = cp.make_predictions(model, batches, str(tmp), task)
    assert os.path.basename(jpath) == ref['json_name']
    assert os.path.splitext(os.path.basename(bpath))[0] == os.path.splitext(ref['h5_name'])[0]
    assert json.load(open(jpath)) == {str(k): v for k, v in preds.items()}
    assert {str(k): v for k, v in preds.items()} == ref['predictions']
    if bpath.endswith('.npz'):
        z = np.load(bpath)
        got = {k: z[k] for k in z.files}
    else:
        import h5py
        with h5py.File(bpath, 'r') as f:
            got = {f'{g}/{k}': np.asarray(f[g][k]) for g in f for k in f[g]}
    assert sorted(got) == sorted(f'{g}/{k}' for g, d in ref['groups'].items() for k in d)
    for key, a in got.items():
        want = ARR[f'pred_{task}/{key}']
        assert a.dtype == want.dtype and a.shape == want.shape, key
        assert np.abs(a - want).max() <= 2e-3 * max(1.0, np.abs(want).max()), (key, np.abs(a - want).max())


@pytest.fixture()
def shim():
    import gpv1_amd.ops as ops
    undo = cpu_shim.install()
    ops.RT.set_precise(True)
    yield
    ops.RT.set_precise(False)
    undo()


@pytest.mark.parametrize('task', ['CocoClassification', 'CocoVqa'])
def test_make_predictions_files_match_reference_cpu(shim, tmp_path, task):
    model, _ = build_small()
    model.eval()
    _run_make_predictions(model, task, tmp_path, 'cpu')


@pytest.mark.gpu
@pytest.mark.parametrize('task', ['CocoClassification', 'CocoVqa'])
def test_make_predictions_files_match_reference_gpu(tmp_path, task):
    import gpv1_amd.ops as ops
    import gpv1_amd.hip as hip
    hip.lib()
    ops.RT.set_precise(True)
    try:
        model, _ = build_small()
        model.to('cuda').eval()
        _run_make_predictions(model, task, tmp_path, 'cuda')
    finally:
        ops.RT.set_precise(False)
