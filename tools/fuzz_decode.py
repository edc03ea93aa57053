"""Inference on random geometry: the small GPV, random batch / image size / ragged padding / query length; greedy decoding as one
hipGraph with the KV cache against the reference's full-prefix schedule (precise mode: same token ids), graph replay against first
run, and beam search (graphed) against its eager form.   usage: python tools/fuzz_decode.py [seed] [n]      (GPU box)"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpv1_amd.hip as hip
import gpv1_amd.ops as ops
from gpv1_amd.misc import NestedTensor
from tests import synth
from tests.test_model_cpu import build_small, V

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rng = random.Random(seed)
hip.lib()
dev = 'cuda:0'


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


bad = 0
for it in range(n):
    B, H, W, Tl = rng.randint(1, 6), 32 * rng.randint(2, 5) + rng.choice([0, 9, 16]), 32 * rng.randint(2, 6) + rng.choice([0, 5, 24]), rng.randint(3, 10)
    pad = [(rng.randint(H // 2, H), rng.randint(W // 2, W)) for _ in range(B)] if rng.random() < 0.5 else None
    precise = rng.random() < 0.5
    images, mask, ids, attn = synth.synth_batch(B, H, W, Tl, V, seed=300 + it, pad_to=pad)
    ops.RT.set_precise(precise)
    model, _ = build_small()
    model.to(dev).eval()
    s = NestedTensor(images.to(dev), mask.to(dev), None if pad else True)
    q = (ids.to(dev), attn.to(dev))
    vm = None
    if rng.random() < 0.5:
        vm = torch.zeros(V, device=dev)
        vm[::3] = -10000.0
    with torch.no_grad():
        model.cfg['graph_inference'], model.cfg['kv_decode'] = True, True
        o1 = model(s, q, None, None, vocab_mask=vm)
        o2 = model(s, q, None, None, vocab_mask=vm)
        model.cfg['kv_decode'] = False
        ob = model(s, q, None, None, vocab_mask=vm)
        model.cfg['kv_decode'] = True
        e = rel(o1['answer_logits'], ob['answer_logits'])
        same_ids = torch.equal(o1['answer_logits'][-1].topk(1, -1).indices.cpu(), ob['answer_logits'][-1].topk(1, -1).indices.cpu())
        ok = torch.equal(o1['answer_logits'], o2['answer_logits']) and e < (1e-4 if precise else 3e-2) and (same_ids or not precise)
        bs = rng.choice([1, 2, 3])
        b1 = model.forward_beam_search(s, q, beam_size=bs)
        model.cfg['graph_inference'] = False
        b2 = model.forward_beam_search(s, q, beam_size=bs)
        okb = b1['answers'] == b2['answers'] if precise else True
    print((B, H, W, Tl, 'ragged' if pad else 'full', 'precise' if precise else 'bf16', 'vocab mask' if vm is not None else ''),
          'greedy kv-graph vs full prefix %.1e ids %s replay %s | beam %d graph vs eager %s' % (e, same_ids, torch.equal(o1['answer_logits'], o2['answer_logits']), bs, okb),
          'ok' if ok and okb else 'FAIL', flush=True)
    bad += (not ok) + (not okb)
ops.RT.set_precise(False)
print('decode fuzz done: %d failures' % bad)
