"""Soak of the capture -> evict -> destroy -> recapture cycle: the ragged string-query stream of bench.py (six size-class signatures) against a
trainer with TWO graph slots and an eviction allowed on every miss, so that nearly every step destroys one captured body (four hipGraphs, its
weight / BERT / branch side streams, its activation pool) and captures another; every 25 steps an inference of another batch size with
2 inference-graph slots churns GPV's inference graphs as well; every 60 steps an in-process resume (model + optimizer state through their state dicts) makes every graph stale.  This is the sequence that ended in a segfault inside a later capture_end
before the side stream of a capture became the capture owner's (round 6, ops.Branch: a process-wide side stream outlived the graphs it had been
captured into).  Losses must stay finite, the last loss of every signature must be below its first.
usage (GPU box): python tools/soak_evict.py [steps=240]"""
import os, sys, time, tempfile, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench
import gpv1_amd.hip as hip
from gpv1_amd import synthetic
from gpv1_amd.bert import WordPieceTokenizer
from gpv1_amd.gpv import GPV
from gpv1_amd.misc import nested_tensor_from_tensor_list
from gpv1_amd.ops import RT
from gpv1_amd.train import FlatTrainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 240
hip.lib(); dev = torch.device('cuda:0'); torch.manual_seed(0)
cfg = bench.make_cfg()
cfg['inference_graph_slots'] = 2
model = GPV(cfg)
for n, buf in model.named_buffers():
    if n.endswith('running_var'):
        buf.uniform_(0.5, 1.5)
model.to(dev)
RT.manual_seed(1000)
tr = FlatTrainer(model, lr=1e-4, lr_backbone=1e-5, weight_decay=1e-4, clip_max_norm=0.1, warmup_steps=10, t_total=10000)
tr.graph_slots, tr.evict_interval = int(os.environ.get('SOAK_SLOTS', 2)), int(os.environ.get('SOAK_EVICT_INTERVAL', 1))
with tempfile.TemporaryDirectory() as td:
    words = synthetic.write_wordpiece_vocab(os.path.join(td, 'vocab.txt'))
    model.bert.tokenizer = WordPieceTokenizer(os.path.join(td, 'vocab.txt'))
B, V = 16, bench.V
g = torch.Generator().manual_seed(4321)
# three padded image shapes (a body's signature holds the image tensor's shape: another activation pool per capture), cycled with a period
# coprime with the 18 query / answer combinations.  SOAK_ONE_SHAPE=1: the bench's 480 x 640 only
SHAPES = [bench.IMG] if os.environ.get('SOAK_ONE_SHAPE') else [bench.IMG, (448, 608), (512, 544)]
samples_by_shape = [nested_tensor_from_tensor_list(torch.randn(B, 3, *hw, generator=g).to(dev)) for hw in SHAPES]


def make(it):
    nq, na = 4 + (5 * it) % 11, 1 + (7 * it) % 18
    qs = [' '.join(words[int(j)] for j in torch.randint(0, len(words), (max(2, nq - i % 3),), generator=g)) for i in range(B)]
    tg = [{'task': 'CocoCaptioning', 'answer': ' '.join(f'w{int(j)}' for j in torch.randint(0, V - 4, (max(1, na - i % 4),), generator=g))} for i in range(B)]
    return qs, tg


batches = [make(it) for it in range(18)]
inf = {b: bench.make_batch(7 + b, b, dev) for b in (1, 2, 3, 4)}
first, last, counts = {}, {}, collections.Counter()
captures = evictions = resumes = 0
seen_bodies = set()
t0 = time.perf_counter()
for it in range(steps):
    qs, tg = batches[it % len(batches)]
    samples = samples_by_shape[(it // 7) % len(samples_by_shape)]
    before = set(map(id, tr._bodies.values()))
    loss = tr.train_step(samples, list(qs), [dict(t) for t in tg])
    after = set(map(id, tr._bodies.values()))
    captures += len(after - before)
    evictions += len(before - after)
    lv = float(loss)
    if not (lv == lv and abs(lv) < 1e6):
        torch.cuda.synchronize()
        print('NON-FINITE loss %r at step %d (sig %d, shape %s, graph steps %d, eager %d, captures %d, evictions %d): P finite %s, M finite %s, V finite %s, G finite %s' %
              (lv, it, it % len(batches), SHAPES[(it // 7) % len(SHAPES)], tr.graph_steps, tr.eager_steps, captures, evictions,
               bool(torch.isfinite(tr.P).all()), bool(torch.isfinite(tr.M).all()), bool(torch.isfinite(tr.V).all()), bool(torch.isfinite(tr.G).all())), flush=True)
        raise SystemExit(3)
    if os.environ.get('SOAK_TRACE'):
        print('  step %d sig %d loss %.4f captures %d evictions %d bodies %d' % (it, it % len(batches), lv, captures, evictions, len(tr._bodies)), flush=True)
    key = (it % len(batches), (it // 7) % len(samples_by_shape))
    first.setdefault(key, lv); last[key] = lv; counts[key] += 1
    if it % 25 == 24 and not os.environ.get('SOAK_NO_INFER'):
        model.eval()
        with torch.no_grad():
            for b in ((it // 25) % 4 + 1, ((it // 25) + 2) % 4 + 1):
                images, mask, ids, attn, _ = inf[b]
                o = model(nested_tensor_from_tensor_list(images), (ids, attn), None, None)
                assert torch.isfinite(o['answer_logits'].float()).all()
        model.train()
    if it % 60 == 59 and not os.environ.get('SOAK_NO_RESUME'):
        # an in-process resume: model and optimizer state through their state dicts (train_distr.py:381-389 / 262-275).  load_state_dict
        # bumps the static epoch: every captured body and inference graph is stale, destroyed and recaptured on its next use
        torch.cuda.synchronize()
        msd = {k: v.detach().clone() for k, v in model.state_dict().items()}
        osd = tr.state_dict()
        P_before = tr.P.clone()
        model.load_state_dict(msd)
        tr.load_state_dict(osd)
        assert torch.equal(tr.P, P_before) and tr.step_count == it + 1
        resumes += 1
    if it % 40 == 39:
        torch.cuda.synchronize()
        print('step %4d  loss %.4f  graph steps %d eager %d  captures %d evictions %d  bodies %d  inference graphs %d  %.1f s' %
              (it + 1, lv, tr.graph_steps, tr.eager_steps, captures, evictions, len(tr._bodies), len(model._igraphs), time.perf_counter() - t0), flush=True)
torch.cuda.synchronize()
down = sum(last[k] < first[k] for k in first)
print('soak_evict done: %d steps, %d captures, %d evictions, %d resumes, %d graph steps, %d eager steps, %d / %d batches ended below their first loss, %.1f s' %
      (steps, captures, evictions, resumes, tr.graph_steps, tr.eager_steps, down, len(first), time.perf_counter() - t0))
if not os.environ.get('SOAK_SLOTS') and not os.environ.get('SOAK_EVICT_INTERVAL'):
    assert captures >= steps // 6 and evictions >= steps // 8, (captures, evictions)
revisited = [k for k in first if first[k] is not last[k] and counts[k] > 1]
down = sum(last[k] < first[k] for k in revisited)
assert down >= len(revisited) * 0.8, (down, len(revisited))
