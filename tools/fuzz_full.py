"""Full-size model (ResNet-50, 6+6 DETR layers, 100 queries, 12-layer BERT, 3 + 3 layers, V = 10 000) on random batch sizes, image
sizes (not only 480 x 640), ragged padding, query lengths and task mixes: forward + loss on the GPU against the CPU oracle on the same
weights -- precise mode within 1e-3, bf16 within 5e-2 / 3e-2 -- i.e. tests/test_model_gpu.py::test_full_size_forward_loss_and_matching_
vs_oracle on geometry it does not list.  The oracle is the checker only.   usage: python tools/fuzz_full.py [seed] [n]   (GPU box)"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpv1_amd.hip as hip
import gpv1_amd.ops as ops
from gpv1_amd.misc import NestedTensor
from oracle import gpv_oracle as O
from tests import synth
from tests.test_model_gpu import full_model, rel

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rng = random.Random(seed)
hip.lib()
DEV = 'cuda'
Vf = 10000
torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
model = full_model(Vf, dropout=0.0)
model.bert.model.p = 0.0
model.train()
cfg = synth.model_cfg(vocab=synth.make_vocab(Vf))
cfg['detr']['dropout'] = 0.0
cfg['_cls_id'] = Vf - 3
Pm = {k: v.detach().float().cpu().contiguous() for k, v in model.state_dict().items()}
bad = 0
for it in range(n):
    B = rng.randint(1, 4)
    H, W = rng.choice([(480, 640), (384, 512), (333, 500), (512, 384), (427, 640), (224, 320)])
    tl = rng.randint(3, 12)
    g = torch.Generator().manual_seed(50 + it)
    images = torch.randn(B, 3, H, W, generator=g)
    mask = torch.zeros(B, H, W, dtype=torch.bool)
    ragged = rng.random() < 0.5
    if ragged:
        for i in range(B):
            h, w = rng.randint(H // 2, H), rng.randint(W // 2, W)
            mask[i, h:, :] = True; mask[i, :, w:] = True
            images[i, :, h:, :] = 0; images[i, :, :, w:] = 0
    ids = torch.randint(1000, 30000, (B, tl), generator=g)
    attn = torch.ones(B, tl, dtype=torch.long)
    for i in range(B):
        k = rng.randint(2, tl)
        attn[i, k:] = 0; ids[i, k:] = 0
    tg = []
    for i in range(B):
        if rng.random() < 0.5:
            nw = rng.randint(1, 18)
            tg.append({'task': rng.choice(['CocoCaptioning', 'CocoVqa', 'CocoClassification']), 'answer': ' '.join(f'w{(37 * j + it) % (Vf - 4)}' for j in range(nw))})
        else:
            nb = rng.randint(1, 5)
            cxcy = 0.25 + 0.5 * torch.rand(nb, 2, generator=g)
            wh = 0.05 + 0.3 * torch.rand(nb, 2, generator=g)
            tg.append({'task': 'CocoDetection', 'boxes': torch.cat([cxcy, wh], 1).to(DEV), 'labels': torch.zeros(nb, dtype=torch.long, device=DEV)})
    _, tok = model.encode_answers(tg)
    for i, t in enumerate(tg):
        t['answer_token_ids'] = tok[i, 1:]
    tg_cpu = [{k: (v.cpu() if torch.is_tensor(v) else v) for k, v in t.items()} for t in tg]
    with torch.no_grad():
        ref = O.gpv_forward(Pm, cfg, images, mask, ids, attn, tok.cpu(), training=True)
        ref_loss, _ = O.gpv_criterion(ref, tg_cpu, cfg['losses'])
    res = []
    ok = True
    samples = NestedTensor(images.to(DEV), mask.to(DEV), None if ragged else True)
    for precise, otol, ltol in ((True, 1e-3, 1e-3), (False, 5e-2, 3e-2)):
        ops.RT.set_precise(precise)
        with torch.no_grad():
            out = model._forward_impl(samples, (ids.to(DEV), attn.to(DEV)), tok, None)
            loss = model.criterion(out, tg)[0]
        errs = {k: rel(out[k], ref[k]) for k in ('pred_boxes', 'pred_relevance_logits', 'detr_hs', 'answer_logits')}
        le = abs(float(loss) - float(ref_loss)) / max(abs(float(ref_loss)), 1e-9) if loss is not None and ref_loss is not None else (0.0 if (loss is None) == (ref_loss is None) else 1.0)
        ok = ok and max(errs.values()) < otol and le < ltol
        res.append('%s: out %.1e loss %.1e' % ('precise' if precise else 'bf16', max(errs.values()), le))
    ops.RT.set_precise(False)
    print((B, H, W, tl, 'ragged' if ragged else 'full', [t['task'][4:7] for t in tg]), ' | '.join(res), 'ok' if ok else 'FAIL', flush=True)
    bad += not ok
print('full-size fuzz done: %d failures' % bad)
