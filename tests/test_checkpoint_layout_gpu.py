"""SURVEY 8(f)-4, the part that needs no network: checkpoints WRITTEN IN THE RELEASED LAYOUTS go through the product's loaders on the
GPU.  The released files themselves (README.md:57-61, setup_data.sh:15-17) are not reachable from this image; what is pinned here
is that their layout loads and that the loaded model computes what the model that wrote the checkpoint computes:
  * GPV checkpoint (exp/gpv/train_distr.py:381-389, read by inference.py:56-62): {'model': DDP state dict -- every key prefixed
    `module.`, conv weights in torch's default (contiguous NCHW) memory format, torchvision BatchNorm `num_batches_tracked` entries
    present --, 'optimizer', 'epoch', 'step', ...} -> `python -m gpv1_amd.inference ckpt=...` (its main(), in process);
  * DETR checkpoint `detr_coco_sce.pth` (exp/gpv/models/gpv.py:122-135): {'model': keys WITHOUT the `detr.` prefix, a class head
    of another size (skipped with "size does not match"), keys the model does not have (ignored)} -> GPV.load_pretr_detr()."""
import contextlib
import io
import json
import os

import numpy as np
import pytest
import torch
import yaml

from tests import synth
from tests.test_model_cpu import build_small, V

pytestmark = pytest.mark.gpu


def _released_state(model):
    sd = {}
    for k, v in model.state_dict().items():
        v = v.detach().cpu().clone()
        sd['module.' + k] = v.contiguous() if v.dim() == 4 else v             # (no channels_last in a torch-1.6 checkpoint)
        if k.endswith('running_var'):
            sd['module.' + k[:-len('running_var')] + 'num_batches_tracked'] = torch.tensor(0)
    return sd


def test_released_layout_checkpoints_load_and_reproduce(tmp_path):
    import gpv1_amd.ops as ops
    from gpv1_amd import inference as inf
    from gpv1_amd.synthetic import write_wordpiece_vocab
    ops.RT.set_precise(False)
    torch.manual_seed(0)
    src, _ = build_small()
    with torch.no_grad():
        for p in src.parameters():                       # not the synthetic fixture state any more: something only the file carries
            p.add_(0.01 * torch.randn_like(p))
    ck = {'model': _released_state(src), 'optimizer': {'state': {}, 'param_groups': []}, 'epoch': 39, 'step': 123456, 'lr': 1e-5,
          'model_selection_metric': 0.5, 'warmup_scheduler': None}
    ckpt = str(tmp_path / 'gpv_coco.pth')
    torch.save(ck, ckpt)
    # the configuration file the CLI reads (Hydra-style tree; vocab as a JSON list, vocabulary embedding as .npy: answer_head.py:61-66)
    m = synth.small_cfg(dropout=0.0)
    vocab_json, emb_npy = str(tmp_path / 'vocab.json'), str(tmp_path / 'vocab_embed.npy')
    json.dump(synth.make_vocab(V), open(vocab_json, 'w'))
    np.save(emb_npy, synth.synth_tensor('answer_head.vocab_embed', (V, 768)).numpy())
    m['vocab'], m['vocab_embed'], m['bert_layers'] = vocab_json, emb_npy, 2
    cfg_yaml = str(tmp_path / 'gpv.yaml')
    yaml.safe_dump({'model': m, 'eval': {'ckpt': None}}, open(cfg_yaml, 'w'))
    words = write_wordpiece_vocab(str(tmp_path / 'vocab.txt'))
    img = (np.random.RandomState(3).rand(64, 96, 3) * 255).astype(np.uint8)
    np.save(str(tmp_path / 'img.npy'), img)
    query = ' '.join(words[i] for i in (3, 17, 5, 200))
    os.environ['GPV_BERT_VOCAB'] = str(tmp_path / 'vocab.txt')
    try:
        out = io.StringIO()
        with contextlib.redirect_stdout(out):
            inf.main(['--config', cfg_yaml, f'ckpt={ckpt}', f'inputs.img={tmp_path / "img.npy"}', f'inputs.query={query}', 'num_output_boxes=4'])
        text = out.getvalue()
        assert 'answer' in text and 'boxes' in text and 'relevance' in text
        # the same prediction from the model that wrote the checkpoint (string query -> the same tokenizer)
        from gpv1_amd.bert import WordPieceTokenizer
        src.bert.tokenizer = WordPieceTokenizer(str(tmp_path / 'vocab.txt'))
        src.cuda().eval()
        want = inf.predict(src, [img], [query], num_output_boxes=4)[0]
        # and through the loader into a fresh model: every tensor identical to the source's
        from gpv1_amd.gpv import GPV
        from gpv1_amd.config import load_config
        fresh = GPV(load_config(cfg_yaml, [], strict=False).model).cuda().eval()
        inf.load_model_state(fresh, ckpt, map_location='cuda:0')
        a, b = src.state_dict(), fresh.state_dict()
        assert set(a) == set(b) and all(torch.equal(a[k].cpu(), b[k].cpu()) for k in a)
        got = inf.predict(fresh, [img], [query], num_output_boxes=4)[0]
        assert got['answer'] == want['answer'] and np.array_equal(got['boxes'], want['boxes']) and np.array_equal(got['relevance'], want['relevance'])
        assert repr(want['answer']) in text or want['answer'] in text
    finally:
        os.environ.pop('GPV_BERT_VOCAB', None)
    # ---- detr_coco_sce.pth -> load_pretr_detr (gpv.py:122-135) ----
    det = {}
    for k, v in src.state_dict().items():
        if k.startswith('detr.'):
            det[k[len('detr.'):]] = (v.detach().cpu().contiguous() if v.dim() == 4 else v.detach().cpu()).clone() + (0.5 if v.is_floating_point() else 0)
    det['class_embed.weight'] = torch.zeros(92, det['class_embed.weight'].shape[1])          # COCO's 91 + 1 classes: another size
    det['class_embed.bias'] = torch.zeros(92)
    det['backbone.0.body.layer1.0.bn1.num_batches_tracked'] = torch.tensor(7)
    det['some.key.the.model.does.not.have'] = torch.zeros(3)
    dpath = str(tmp_path / 'detr_coco_sce.pth')
    torch.save({'model': det, 'args': None}, dpath)
    m2 = dict(synth.small_cfg(dropout=0.0), vocab=synth.make_vocab(V), vocab_embed=synth.synth_tensor('answer_head.vocab_embed', (V, 768)),
              bert_layers=2, pretr_detr=dpath)
    tgt = GPV(m2)
    before = {k: v.clone() for k, v in tgt.state_dict().items()}
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        tgt.load_pretr_detr()
    after = tgt.state_dict()
    assert out.getvalue().count('size does not match') == 2
    loaded = set(tgt.init_detr_params)
    assert 'detr.class_embed.weight' not in loaded and 'detr.backbone.0.body.conv1.weight' in loaded and len(loaded) > 300
    for k in after:
        if k in loaded:
            assert torch.equal(after[k].cpu(), det[k[len('detr.'):]]), k
        else:
            assert torch.equal(after[k], before[k]), k
    tgt.cuda().eval()
    with torch.no_grad():
        o = inf.predict(tgt, [img], (torch.randint(1000, 30000, (1, 5)).cuda(), torch.ones(1, 5, dtype=torch.long).cuda()), num_output_boxes=2)[0]
    assert o['boxes'].shape == (2, 4) and np.isfinite(o['boxes']).all()
