"""The multi-rank training path on a real GPU: two processes share the one MI355X of the test box and exchange
gradients over gloo (RCCL refuses two ranks on one device; the collectives' call sites, the overlap milestone fired
from the autograd thread, the device-side liveness flags and the pinned staging are the same code as over RCCL).
Ranks must stay bit-identical, and the exchanged gradient must be the average of the local ones."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out, comm='fp32', precise=False):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from tests import synth
    from tests.test_model_cpu import build_small, nested, V, B, H, W, Tl, PAD
    import gpv1_amd.ops as ops
    from gpv1_amd.train import FlatTrainer
    ops.RT.set_precise(bool(precise))
    torch.manual_seed(100 + rank)                          # different init per rank: the broadcast must fix it
    model, _ = build_small()
    model.to('cuda').train()
    model.bert.model.p = 0.0
    tr = FlatTrainer(model, lr=1e-4, lr_backbone=1e-5, bucket_mb=8, grad_comm_dtype=comm)
    assert tr.world == 2 and tr.overlap == (os.environ.get('GPV_OVERLAP', '1') != '0')
    images, mask, ids, attn = synth.synth_batch(B, H, W, Tl, V, seed=1234 + rank, pad_to=PAD)
    images, mask, ids, attn = images.cuda(), mask.cuda(), ids.cuda(), attn.cuda()
    tasks = None if rank == 0 else ('CocoCaptioning',)   # the box head is only touched on rank 0
    losses = []
    for step in range(5):
        tg = synth.synth_targets(B, V, S=6) if tasks is None else synth.synth_targets(B, V, S=6, seed=7, tasks=tasks)
        for d in tg:
            for k, v in d.items():
                if torch.is_tensor(v):
                    d[k] = v.cuda()
        model.bert.model.p = 0.0
        loss = tr.train_step(nested(images, mask), (ids, attn), tg)
        # eager step: 'backbone' (everything behind the backbone segment is complete when autograd reaches the backbone); graphed
        # steps: 'head' -- the DETR weight-gradient group runs on a branch of B2's first stage graph, so only the buckets BEHIND the
        # DETR-head segment go between B1 and B2 and the head's follow with layer4's
        names = [m for m, _ in tr.milestone_log]
        if os.environ.get('GPV_OVERLAP', '1') != '0':
            eager = step == 0 or os.environ.get('GPV_TRAIN_GRAPHS', '1') == '0'
            assert names == [('backbone' if eager else 'head'), 'layer4', 'layer3', 'layer2'] and tr.late_touch is None, (tr.milestone_log, tr.late_touch)
            assert tr.left_after_backward == 0
        losses.append(float(loss.detach()))
    torch.cuda.synchronize()
    # steps 2.. replay the captured hipGraphs
    assert os.environ.get('GPV_TRAIN_GRAPHS', '1') == '0' or tr.graphs and len(tr._bodies) >= 1 and tr.graph_steps >= 3, (tr.graphs, len(tr._bodies), tr.graph_steps, tr.eager_steps)
    torch.save({'G': tr.G.cpu(), 'head': (tr.backbone_end, tr.head_end), 'P': tr.P.cpu(), 'live': tr.live_host(), 'touched_local': tr.touched.clone(), 'losses': losses,
                'names': [e[0] for e in tr.entries]}, os.path.join(out, f'rank{rank}.pt'))
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize('comm', ['fp32', 'bf16'])
def test_two_ranks_on_one_gpu_stay_identical(tmp_path, comm):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), comm), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp_path, 'rank0.pt'))
    r1 = torch.load(os.path.join(tmp_path, 'rank1.pt'))
    assert torch.equal(r0['P'], r1['P'])                                    # replicas bit-identical after 3 steps
    assert torch.equal(r0['live'], r1['live'])
    ib = [i for i, n in enumerate(r0['names']) if 'bbox_embed' in n]
    assert r0['touched_local'][ib].all() and not r1['touched_local'][ib].any() and r1['live'][ib].all()
    assert all(map(lambda x: x == x, r0['losses'] + r1['losses']))            # finite


@pytest.mark.timeout(900)
def test_overlapped_exchange_hands_over_complete_gradients(tmp_path):
    """The graphed step hands buckets to the all-reduce between the backward graphs; a bucket handed over before the last kernel
    writing into it has run would exchange a stale gradient on BOTH ranks alike (replicas stay identical -- the test above cannot
    see it).  Reference: the same run with the exchange after the whole pass (GPV_OVERLAP=0).  Not bit-identical run to run
    (float atomics in the weight-gradient kernels), so: the summed gradient of the last step, per segment, within 2 %."""
    runs = {}
    for overlap in ('1', '0'):
        os.environ['GPV_OVERLAP'] = overlap
        out = tmp_path / overlap
        out.mkdir()
        try:
            mp.spawn(_worker, args=(2, _free_port(), str(out), 'fp32'), nprocs=2, join=True)
        finally:
            os.environ.pop('GPV_OVERLAP')
        runs[overlap] = torch.load(os.path.join(out, 'rank0.pt'))
    a, b = runs['1'], runs['0']
    lo, hi = a['head']
    total = a['G'].numel()
    for name, (s, e) in {'backbone': (0, lo), 'detr head': (lo, hi), 'behind the head': (hi, total)}.items():
        ga, gb = a['G'][s:e].double(), b['G'][s:e].double()
        assert gb.norm() > 0
        rel = float((ga - gb).norm() / gb.norm())
        assert rel <= 2e-2, (name, rel)


@pytest.mark.timeout(900)
@pytest.mark.parametrize('mode', ['precise', 'bf16_single_launches'])
def test_overlapped_exchange_with_side_stream_weight_gradients(tmp_path, mode):
    """with one launch per conv weight gradient on the SIDE stream (fp32 "precise" mode, or GPV_WGRAD_GROUP=0) the stage
    milestones used to hand a stage's buckets to the exchange before the side stream had been joined.  Eager steps (the milestones
    fire from the autograd thread), overlap on against overlap off: same summed gradient per segment."""
    runs = {}
    env = {'GPV_TRAIN_GRAPHS': '0'}
    if mode != 'precise':
        env['GPV_WGRAD_GROUP'] = '0'
    for overlap in ('1', '0'):
        os.environ.update(env, GPV_OVERLAP=overlap)
        out = tmp_path / overlap
        out.mkdir()
        try:
            mp.spawn(_worker, args=(2, _free_port(), str(out), 'fp32', mode == 'precise'), nprocs=2, join=True)
        finally:
            for k in list(env) + ['GPV_OVERLAP']:
                os.environ.pop(k)
        runs[overlap] = torch.load(os.path.join(out, 'rank0.pt'))
    a, b = runs['1'], runs['0']
    lo, hi = a['head']
    # run-to-run noise of this 5-step run, measured with tools/dbg_overlap.py (overlap 0 against overlap 0, precise): backbone segment 0.3 - 0.8 %,
    # 1.7 % seen once (fp32 atomics reorder sums, five optimizer steps and the assignment amplify it); a bucket exchanged stale is O(1)
    tol = 4e-2
    for name, (s, e) in {'backbone': (0, lo), 'detr head': (lo, hi), 'behind the head': (hi, a['G'].numel())}.items():
        ga, gb = a['G'][s:e].double(), b['G'][s:e].double()
        assert gb.norm() > 0
        rel = float((ga - gb).norm() / gb.norm())
        assert rel <= tol, (mode, name, rel)
