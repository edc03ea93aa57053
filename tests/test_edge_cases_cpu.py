"""Edge cases of the host logic (CPU, kernels emulated by tests/cpu_shim.py), following what the reference does:
empty / missing targets, ragged image lists, zero ground-truth boxes, answer truncation and out-of-vocabulary words,
matcher tie cases."""
import numpy as np
import pytest
import torch

from tests import synth, cpu_shim
from tests.test_model_cpu import build_small, nested, V, H, W, Tl


@pytest.fixture(scope='module')
def shim():
    import gpv1_amd.ops as ops
    undo = cpu_shim.install()
    ops.RT.set_precise(True)
    yield
    ops.RT.set_precise(False)
    undo()


def _inputs(Bn, sizes=None, seed=0):
    g = torch.Generator().manual_seed(seed)
    sizes = sizes or [(H, W)] * Bn
    imgs = [torch.randn(3, h, w, generator=g) for h, w in sizes]
    ids = torch.randint(1000, 30000, (Bn, Tl), generator=g)
    attn = torch.ones(Bn, Tl, dtype=torch.long)
    return imgs, ids, attn


def test_ragged_image_list_equals_explicit_padding(shim):
    """gpv.py / detr_roi_head.py:72-73: a list of differently sized images is zero-padded to the batch maximum with a mask"""
    from gpv1_amd.misc import NestedTensor, nested_tensor_from_tensor_list
    model, _ = build_small()
    model.eval()
    imgs, ids, attn = _inputs(3, [(96, 128), (64, 96), (80, 100)])
    nt = nested_tensor_from_tensor_list(imgs)
    assert nt.tensors.shape == (3, 3, 96, 128) and nt.all_valid is False
    assert bool(nt.mask[1, 64:, :].all()) and bool(nt.mask[1, :, 96:].all()) and not bool(nt.mask[1, :64, :96].any())
    with torch.no_grad():
        a = model(imgs, (ids, attn), None, None)                       # raw list: the model builds the NestedTensor
        b = model(NestedTensor(nt.tensors, nt.mask), (ids, attn), None, None)
    for k in ('pred_boxes', 'pred_relevance_logits', 'answer_logits'):
        assert torch.equal(a[k], b[k]), k
    same = nested_tensor_from_tensor_list([imgs[0], imgs[0].clone()])
    assert same.all_valid is True and not bool(same.mask.any())


def test_no_applicable_target_and_empty_boxes(shim):
    """losses.py:155-176: a loss whose task has no sample contributes nothing; a batch nobody can score returns None
    (train_distr.py:414 skips the step); a detection sample with zero boxes matches nothing and only pays the
    no-object term"""
    from gpv1_amd.train import FlatTrainer
    model, _ = build_small()
    model.train()
    model.bert.model.p = 0.0
    imgs, ids, attn = _inputs(2)
    tr = FlatTrainer(model, lr=1e-4, lr_backbone=1e-5)
    none = tr.train_step(nested(torch.stack(imgs), torch.zeros(2, H, W, dtype=torch.bool)), (ids, attn),
                         [{'task': 'SomethingElse'}, {'task': 'SomethingElse'}])
    assert none is None and tr.step_count == 0
    tg = [{'task': 'CocoDetection', 'boxes': torch.zeros(0, 4), 'labels': torch.zeros(0, dtype=torch.long)},
          {'task': 'CocoDetection', 'boxes': torch.tensor([[0.5, 0.5, 0.2, 0.2]]), 'labels': torch.zeros(1, dtype=torch.long)}]
    loss = tr.train_step(nested(torch.stack(imgs), torch.zeros(2, H, W, dtype=torch.bool)), (ids, attn), tg)
    assert torch.isfinite(loss) and tr.step_count == 1
    ind = model.criterion.localization_criterion.set_criterion.last_indices
    assert len(ind) == 2 and len(ind[0][0]) == 0 and len(ind[1][0]) == 1


def test_encode_answers_truncation_oov_and_padding(shim):
    """gpv.py:377-430: '__cls__ answer __stop__', lower-cased, OOV -> __unk__, padded with __pad__ to the batch maximum,
    cut at max_text_len (6 in the small configuration)"""
    model, _ = build_small()
    w2i = model.word_to_idx
    toks, ids = model.encode_answers([{'answer': 'W3 w5'}, {'answer': ''}, {'answer': 'w1 zzz w2 w3 w4 w5 w6 w7'}, {}])
    assert ids.shape == (4, 6)
    assert ids[0].tolist() == [w2i['__cls__'], w2i['w3'], w2i['w5'], w2i['__stop__'], w2i['__pad__'], w2i['__pad__']]
    assert ids[1].tolist()[:2] == [w2i['__cls__'], w2i['__stop__']] and ids[3].tolist() == ids[1].tolist()
    assert ids[2].tolist() == [w2i['__cls__'], w2i['w1'], w2i['__unk__'], w2i['w2'], w2i['w3'], w2i['w4']]   # truncated, no __stop__
    assert model.token_ids_to_words(ids[:1])[0][1:3] == ['w3', 'w5']


def test_matcher_ties_and_more_targets_than_queries():
    """matcher.py:32-77 via scipy LSAP: identical predictions tie -> lowest index wins (scipy's rule, pinned by the golden
    matcher fixture); more ground-truth boxes than queries -> min(Q, n) pairs"""
    from gpv1_amd.criterion import HungarianMatcher
    m = HungarianMatcher(cost_class=1, cost_bbox=5, cost_giou=2)
    logits = torch.zeros(1, 3, 2)
    boxes = torch.tensor([[[0.5, 0.5, 0.2, 0.2]] * 3])
    tgt = [{'boxes': torch.tensor([[0.5, 0.5, 0.2, 0.2], [0.3, 0.3, 0.1, 0.1]]), 'labels': torch.zeros(2, dtype=torch.long)}]
    (pi, ti), = m({'pred_relevance_logits': logits, 'pred_boxes': boxes}, tgt)
    assert sorted(ti.tolist()) == [0, 1] and len(set(pi.tolist())) == 2
    many = [{'boxes': torch.rand(5, 4) * 0.3 + 0.3, 'labels': torch.zeros(5, dtype=torch.long)}]
    (pi, ti), = m({'pred_relevance_logits': logits, 'pred_boxes': boxes}, many)
    assert len(pi) == 3 and len(set(ti.tolist())) == 3


def test_load_pretr_detr_and_phase1_freeze(shim, tmp_path):
    """gpv.py:122-135 + train_distr.py:136-140: a DETR checkpoint ({'model': keys without the 'detr.' prefix}) initialises
    the matching tensors only -- same name AND same size --, remembers them in init_detr_params, leaves everything else
    alone; phase 1 of the reference's schedule (training.freeze) then freezes exactly that list."""
    from gpv1_amd.train_distr import freeze_detr_params
    model, _ = build_small()
    before = {k: v.clone() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(7)
    ckpt = {}
    take = ['backbone.0.body.layer2.0.conv1.weight', 'backbone.0.body.layer2.0.bn1.running_var', 'transformer.encoder.layers.0.linear1.weight',
            'transformer.decoder.layers.1.norm3.bias', 'input_proj.bias', 'bbox_embed.layers.2.weight']
    for k in take:
        ckpt[k] = torch.randn(before['detr.' + k].shape, generator=g)
    ckpt['class_embed.weight'] = torch.randn(92, 256, generator=g)          # COCO DETR head: 92 classes -> size mismatch, skipped
    ckpt['class_embed.bias'] = torch.randn(92, generator=g)
    ckpt['query_embed.weight'] = torch.randn(100, 256, generator=g)         # 100 queries vs the fixture's 10 -> skipped
    ckpt['not_in_the_model.weight'] = torch.randn(3, 3, generator=g)        # unknown name -> ignored
    path = tmp_path / 'detr.pth'
    torch.save({'model': ckpt}, path)
    model.cfg['pretr_detr'] = str(path)
    model.load_pretr_detr()
    assert sorted(model.init_detr_params) == sorted('detr.' + k for k in take)
    after = model.state_dict()
    for k, v in after.items():
        lk = k[len('detr.'):] if k.startswith('detr.') else None
        if lk in take:
            assert torch.equal(v, ckpt[lk]), k
        else:
            assert torch.equal(v, before[k]), k                              # incl. the size-mismatched class_embed / query_embed
    # phase 1: exactly the loaded tensors that are parameters stop training (buffers such as running_var are not parameters)
    rg_before = {n: p.requires_grad for n, p in model.named_parameters()}
    freeze_detr_params(model)
    for n, p in model.named_parameters():
        if n in model.init_detr_params:
            assert p.requires_grad is False, n
        else:
            assert p.requires_grad == rg_before[n], n
    freeze_detr_params(model, requires_grad=True)                           # phase 2 (finetune): released again
    assert all(p.requires_grad for n, p in model.named_parameters() if n in model.init_detr_params)


def test_capture_quiesce_waits_only_for_synchronous_collectives(monkeypatch):
    """round 4 slept 0.35 s in front of every stream capture with a process group alive (two per new batch signature, every
    rank in lock-step).  misc.CollectiveClock: the wait is owed only when a SYNCHRONOUS collective ran on the compute stream since the
    device was last seen idle; the trainer's per-step collectives are asynchronous (RCCL's own stream) and never arm it.  A ragged
    stream of 12 new signatures (24 captures) after the construction-time broadcast: one wait."""
    import time
    import torch
    import torch.distributed as dist
    import gpv1_amd.misc as misc
    cc = misc.CollectiveClock
    slept = []
    monkeypatch.setattr(time, 'sleep', lambda s: slept.append(s))
    monkeypatch.setattr(dist, 'is_initialized', lambda: True)
    monkeypatch.setattr(torch.cuda, 'is_available', lambda: True)
    monkeypatch.setattr(torch.cuda, 'synchronize', lambda *a, **k: None)
    monkeypatch.setattr(cc, 'MODE', 'auto')
    monkeypatch.setattr(cc, 'sleeps', 0)
    monkeypatch.setattr(cc, 'calls', 0)
    monkeypatch.setattr(cc, 'idle_since', None)
    monkeypatch.setattr(cc, 'pending', True)
    misc.note_sync_collective()                       # FlatTrainer's parameter broadcast
    for _ in range(12):                               # 12 signatures x (forward capture, backward capture); asynchronous all-reduces between
        misc.quiesce_collectives()
        misc.quiesce_collectives()
    assert cc.calls == 24 and cc.sleeps == 1 and len(slept) == 1 and 0 < slept[0] <= cc.QUIET
    misc.note_sync_collective()                       # a driver's barrier: the next capture waits again, once
    misc.quiesce_collectives()
    misc.quiesce_collectives()
    assert cc.sleeps == 2
    monkeypatch.setattr(cc, 'MODE', 'always')         # round 4's behaviour on request
    misc.quiesce_collectives()
    assert cc.sleeps == 3
