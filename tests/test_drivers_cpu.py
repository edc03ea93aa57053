"""SURVEY §8(f) rows on CPU: Hydra-compatible config loading, the training driver (checkpoint layout, resume,
phase-1 freeze, sharding) and the inference harness (checkpoint prefix, box ordering, answer cut).
The kernels are emulated by tests/cpu_shim.py; the drivers are the product code."""
import os

import numpy as np
import pytest
import torch

from tests import synth, cpu_shim
from tests.test_model_cpu import build_small, V


@pytest.fixture(scope='module')
def shim():
    import gpv1_amd.ops as ops
    undo = cpu_shim.install()
    ops.RT.set_precise(True)
    yield
    ops.RT.set_precise(False)
    undo()


def test_config_interpolation_overrides_and_float_parsing(tmp_path):
    from gpv1_amd.config import load_config, from_dict
    from gpv1_amd.default_config import default_tree
    cfg = from_dict(default_tree(),
                    ['exp_name=run7', 'output_dir=/tmp/out', 'training.freeze=True', 'training.lr=2e-4', 'model.detr.num_queries=50',
                     'training.lr_milestones=[3,4]', 'training.ckpt=null', '+inputs.query=what is this?'])
    assert cfg.ckpt_dir == '/tmp/out/run7/ckpts' and cfg.eval.ckpt == '/tmp/out/run7/ckpts/model.pth'
    assert cfg.training.freeze is True and cfg.training.lr == 2e-4 and cfg.training.ckpt is None
    assert cfg.model.detr.num_queries == 50 and cfg.training.lr_milestones == [3, 4]
    assert cfg.batch_size == 120 and cfg.inputs.query == 'what is this?'
    assert isinstance(cfg.model.losses.CaptionLoss.loss_wts.loss_caption, float)
    assert [k for k, _ in cfg.model.losses.items()] == ['CaptionLoss', 'VqaLoss', 'ClsLoss', 'Localization']
    with pytest.raises(KeyError):
        from_dict(default_tree(), ['training.no_such_key=1'])
    # the shapes the reference's own YAML uses: top-level group referenced from inside `model`, PyYAML-hostile floats
    y = tmp_path / 'ref_style.yaml'
    y.write_text('data_dir: /d\nmodel:\n  vocab: ${data_dir}/vocab.json\n  hidden_dim: 768\n  losses: ${losses}\n  detr:\n    dropout: 0.1\n'
                 '  text_decoder:\n    dropout: ${model.detr.dropout}\n    hidden_dim: ${model.hidden_dim}\nlosses:\n  CaptionLoss:\n    loss_wts:\n      loss_caption: 5e-2\n'
                 'training:\n  lr: 1e-4\n  freeze: False\n')
    c = load_config(str(y), ['model.detr.dropout=0.2'])
    assert c.model.vocab == '/d/vocab.json' and c.model.text_decoder.dropout == 0.2 and c.model.text_decoder.hidden_dim == 768
    assert c.model.losses.CaptionLoss.loss_wts.loss_caption == 0.05 and c.training.lr == 1e-4 and c.training.freeze is False


def _driver_cfg(tmp_path, **training):
    from gpv1_amd.config import from_dict
    m = synth.small_cfg(dropout=0.0)
    m['vocab'] = synth.make_vocab(V)
    m['vocab_embed'] = synth.synth_tensor('answer_head.vocab_embed', (V, 768))
    m['bert_layers'] = 2
    m['bert_dropout'] = 0.0                                   # the CPU shim has no dropout
    tr = {'ckpt': None, 'freeze': False, 'frozen_epochs': 1, 'frozen_batch_size': 2, 'num_epochs': 2, 'batch_size': 2, 'log_step': 1,
          'ckpt_step': 1000, 'lr': 1e-3, 'lr_backbone': 1e-4, 'weight_decay': 1e-4, 'lr_warmup': True, 'lr_linear_decay': True,
          'lr_warmup_fraction': 0.25, 'clip_max_norm': 0.1}
    tr.update(training)
    return from_dict({'ckpt_dir': str(tmp_path / 'ckpts'), 'model': m, 'training': tr, 'synthetic_samples': 4})


def _dataset(model_vocab):
    from gpv1_amd.train_distr import SyntheticCocoDataset
    return SyntheticCocoDataset(4, model_vocab, image_size=(64, 96), query_len=5)


def test_train_driver_checkpoint_layout_and_resume(shim, tmp_path):
    from gpv1_amd import train_distr as td
    vocab = synth.make_vocab(V)
    logs = []
    torch.manual_seed(0)
    cfg = _driver_cfg(tmp_path / 'a')
    model, tr, step = td.train_worker(cfg, dataset=_dataset(vocab), device='cpu', log=logs.append)
    assert any('BERT has random weights' in l for l in logs)            # no model.bert_weights in the fixture: the driver says so
    steps = [l for l in logs if l.startswith('epoch')]
    assert step == 4 and len(steps) == 4 and 'loss' in steps[0]
    ck = torch.load(os.path.join(cfg.ckpt_dir, 'model.pth'), weights_only=False)
    assert set(ck) == {'model', 'optimizer', 'epoch', 'step', 'lr', 'model_selection_metric', 'warmup_scheduler'}
    assert ck['epoch'] == 1 and ck['step'] == 4
    assert all(k.startswith('module.') for k in ck['model']) and len(ck['model']) == len(model.state_dict())
    # interrupted after epoch 0, resumed from its checkpoint: same weights as the uninterrupted run
    torch.manual_seed(0)
    cfg1 = _driver_cfg(tmp_path / 'b')
    cfg1['max_steps'] = 2
    td.train_worker(cfg1, dataset=_dataset(vocab), device='cpu', log=lambda s: None)
    torch.manual_seed(0)
    cfg2 = _driver_cfg(tmp_path / 'b', ckpt=os.path.join(cfg1.ckpt_dir, 'model.pth'))
    m2, tr2, step2 = td.train_worker(cfg2, dataset=_dataset(vocab), device='cpu', log=logs.append)
    assert step2 == 4 and 'resumed' in logs[-3]
    a, b = model.state_dict(), m2.state_dict()
    worst = max(float((a[k].float() - b[k].float()).abs().max()) for k in a if a[k].is_floating_point())
    assert worst < 1e-6, worst
    assert tr2.step_count == tr.step_count == 4


def test_phase1_freeze_and_sharding(shim, tmp_path):
    from gpv1_amd import train_distr as td
    from gpv1_amd.gpv import GPV
    cfg = _driver_cfg(tmp_path, freeze=True)
    model = GPV(cfg.model)
    model.init_detr_params = [n for n, _ in model.named_parameters() if n.startswith('detr.transformer.encoder')]
    td.freeze_detr_params(model)
    assert all(not p.requires_grad for n, p in model.named_parameters() if n in model.init_detr_params)
    assert any(p.requires_grad for n, p in model.named_parameters() if n.startswith('detr.transformer.decoder'))
    # DistributedSampler semantics: disjoint cover (with wrap-around padding), reshuffled per epoch
    s0, s1 = td.shard_indices(10, 0, 0, 4), td.shard_indices(10, 0, 1, 4)
    allr = sum((td.shard_indices(10, 0, r, 4) for r in range(4)), [])
    assert len(s0) == len(s1) == 3 and set(allr) == set(range(10)) and len(allr) == 12
    assert td.shard_indices(10, 1, 0, 4) != s0


def test_inference_harness(shim, tmp_path):
    from gpv1_amd import inference as inf
    from gpv1_amd import train_distr as td
    from gpv1_amd.train import FlatTrainer
    model, _ = build_small()
    model.eval()
    path = str(tmp_path / 'model.pth')
    td.save_checkpoint(path, model, FlatTrainer(model), epoch=0, step=1)
    fresh, _ = build_small()
    with torch.no_grad():
        for p in fresh.parameters():
            p.add_(1.0)
    inf.load_model_state(fresh, path)                                   # 'module.' prefixed checkpoint
    assert all(torch.equal(a, b) for a, b in zip(fresh.state_dict().values(), model.state_dict().values()))
    # decode: boxes by relevance descending, cut; answer up to __stop__
    out = {'pred_relevance_logits': torch.tensor([[[0.0, 1.0], [3.0, 0.0], [1.0, 0.0]]]),
           'pred_boxes': torch.tensor([[[0.1] * 4, [0.2] * 4, [0.3] * 4]]),
           'answer_logits': torch.zeros(1, 1, 4, V)}
    ids = [model.word_to_idx['w3'], model.word_to_idx['w7'], model.word_to_idx['__stop__'], model.word_to_idx['w1']]
    for t, i in enumerate(ids):
        out['answer_logits'][0, 0, t, i] = 5.0
    d = inf.decode_outputs(out, model, num_output_boxes=2)[0]
    assert d['answer'] == 'w3 w7' and d['boxes'].shape == (2, 4)
    assert np.allclose(d['boxes'][:, 0], [0.2, 0.3]) and d['relevance'][0] > d['relevance'][1]
    assert inf.detokenize(['it', 'is', "n't", 'a', 'cat', ',', 'is', 'it', '?']) == "it isn't a cat, is it?"
    img = (np.random.RandomState(0).rand(64, 96, 3) * 255).astype(np.uint8)
    x = inf.preprocess_image(img)
    assert x.shape == (3, 64, 96) and abs(float(x[0].mean()) - (img[..., 0].mean() / 255 - 0.485) / 0.229) < 1e-4
    g = torch.Generator().manual_seed(0)
    q = (torch.randint(1000, 30000, (1, 5), generator=g), torch.ones(1, 5, dtype=torch.long))
    p = inf.predict(model, [img], q, num_output_boxes=3)[0]
    assert p['boxes'].shape == (3, 4) and isinstance(p['answer'], str)
    pb = inf.predict(model, [img], q, beam_size=2, num_output_boxes=3)[0]
    assert 'answer_prob' in pb and 0.0 <= pb['answer_prob'] <= 1.0


def test_compute_predictions_files_and_vocab_mask(shim, tmp_path):
    """compute_predictions.py:30-109 -- vocabulary mask, prediction JSON, boxes file layout"""
    import json
    from gpv1_amd import compute_predictions as cp
    from gpv1_amd import inference as inf
    model, _ = build_small()
    model.eval()
    # mask: 0 on class-name tokens present in the vocabulary and on __stop__/__pad__, -10000 elsewhere
    tokens, mask = cp.create_vocab_mask(model, classes=('w3', 'w5 w9', 'not-a-word'), synonyms={'w3': ['w3', 'w4'], 'w5 w9': ['w5 w9'], 'not-a-word': ['w1']}, use_syns=True)
    on = sorted(int(i) for i in np.nonzero(mask == 0)[0])
    assert on == sorted(model.word_to_idx[t] for t in ('w3', 'w4', 'w5', 'w9', 'w1', '__stop__', '__pad__'))
    assert set(tokens) == {'w3', 'w4', 'w5', 'w9', 'w1', '__stop__', '__pad__'} and float(mask.min()) == -10000.0 and mask.dtype == np.float32
    assert cp.word_tokenize('hot dog') == ['hot', 'dog'] and len(cp.COCO_CLASSES) == 80
    # three batches of two; num_eval_batches = 1 keeps batches 0 and 1 (the reference's `i > num_eval_batches`)
    rs = np.random.RandomState(1)
    g = torch.Generator().manual_seed(3)

    def batches():
        for b in range(3):
            imgs = [inf.preprocess_image(inf.resize_image((rs.rand(50, 70, 3) * 255).astype(np.uint8), (32, 48))) for _ in range(2)]
            q = (torch.randint(1000, 30000, (2, 5), generator=g), torch.ones(2, 5, dtype=torch.long))
            yield imgs, q, [f'{b}_{k}' for k in range(2)]
    _, vm = cp.create_vocab_mask(model, classes=('w2', 'w6'))
    preds, jpath, bpath = cp.make_predictions(model, batches(), str(tmp_path / 'eval'), 'CocoClassification', subset='val',
                                              data_split='gpv_split', num_eval_batches=1, vocab_mask=vm)
    assert os.path.basename(jpath) == 'CocoClassification_gpv_split_val_predictions.json'
    assert sorted(preds) == ['0_0', '0_1', '1_0', '1_1'] and json.load(open(jpath)) == preds
    assert all(set(p['answer'].split()) <= {'w2', 'w6'} for p in preds.values())          # the mask confines the answers
    if bpath.endswith('.npz'):
        z = np.load(bpath)
        boxes, rel = z['1_0/boxes'], z['1_0/relevance']
    else:
        import h5py
        with h5py.File(bpath, 'r') as f:
            boxes, rel = f['1_0']['boxes'][()], f['1_0']['relevance'][()]
    Q = model.cfg.detr.num_queries if hasattr(model, 'cfg') else boxes.shape[0]
    assert boxes.shape == (Q, 4) and rel.shape == (Q,) and boxes.dtype == np.float32 and rel.dtype == np.float32
    assert np.all(np.diff(rel) <= 0)                                                      # sorted by relevance, all boxes kept
    # anti-aliased resize: constant images stay constant, means are preserved, sizes are exact
    c = np.full((50, 70, 3), 0.25, np.float32)
    assert np.abs(inf.resize_image(c, (20, 30)) - 0.25).max() < 1e-6
    img = (rs.rand(97, 131, 3) * 255).astype(np.uint8)
    r = inf.resize_image(img, (48, 64))
    assert r.shape == (48, 64, 3) and r.dtype == np.float32 and abs(float(r.mean()) - img.mean() / 255) < 2e-3
    assert r.std() < (img / 255.0).std()                                                  # low-pass before subsampling


def test_hydra_defaults_groups_and_learning_datasets_override(tmp_path):
    """configs/exp/gpv.yaml:23-25 `defaults:` + scripts/train.sh:14-34 `learning_datasets=all|cap|det`: group files are merged at
    the package their first line declares, an override on a group name selects the file; the built-in tree's options equal the
    reference's eleven configs/learning_datasets/*.yaml (fixture dumped by tools/gen_golden_harness.py)."""
    import json
    from gpv1_amd.config import load_config, from_dict
    from gpv1_amd.default_config import default_tree, GROUP_OPTIONS
    root = tmp_path / 'configs'
    for d in ('exp', 'task', 'learning_datasets'):
        (root / d).mkdir(parents=True)
    (root / 'exp' / 'gpv.yaml').write_text('exp_name: x\ndata_dir: /d\ndefaults:\n  - task: coco_learning_tasks\n  - learning_datasets: vqa\n'
                                           'training:\n  lr: 1e-4\n')
    (root / 'task' / 'coco_learning_tasks.yaml').write_text('# @package task_configs\nimage_dir: ${data_dir}/images\nimage_size:\n  H: 480\n  W: 640\n')
    (root / 'learning_datasets' / 'vqa.yaml').write_text('# @package _group_\nCocoVqa:\n  task_config: coco_vqa\n  name: coco_vqa\n')
    (root / 'learning_datasets' / 'det_cap.yaml').write_text('# @package _group_\nCocoCaptioning:\n  task_config: coco_captioning\n  name: coco_cap\n'
                                                             'CocoDetection:\n  task_config: coco_detection\n  name: coco_det\n')
    cfg = load_config(str(root / 'exp' / 'gpv.yaml'))
    assert list(cfg.learning_datasets) == ['CocoVqa'] and cfg.task_configs.image_dir == '/d/images' and cfg.task_configs.image_size.W == 640
    cfg = load_config(str(root / 'exp' / 'gpv.yaml'), ['learning_datasets=det_cap', 'training.lr=2e-4'])
    assert list(cfg.learning_datasets) == ['CocoCaptioning', 'CocoDetection'] and cfg.training.lr == 2e-4
    with pytest.raises(FileNotFoundError):
        load_config(str(root / 'exp' / 'gpv.yaml'), ['learning_datasets=nope'])
    # built-in tree: same options as the reference's files, same key order (= the order tasks are concatenated in)
    ref = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'harness.json')))['learning_datasets']
    assert sorted(ref) == sorted(GROUP_OPTIONS['learning_datasets'])
    for name, items in ref.items():
        assert [[k, dict(v)] for k, v in GROUP_OPTIONS['learning_datasets'][name].items()] == items, name
    assert list(from_dict(default_tree(), [], group_options=GROUP_OPTIONS).learning_datasets) == ['CocoVqa']      # gpv.yaml:25
    cfg = from_dict(default_tree(), ['learning_datasets=cap'], group_options=GROUP_OPTIONS)
    assert list(cfg.learning_datasets) == ['CocoCaptioning']
    with pytest.raises(KeyError):
        from_dict(default_tree(), ['learning_datasets=nope'], group_options=GROUP_OPTIONS)
