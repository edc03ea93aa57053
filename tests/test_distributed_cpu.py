"""Multi-process data-parallel path on CPU: world_size 2 over gloo (the GPU run uses the same code over
RCCL).  Kernels are the torch emulation (tests/cpu_shim.py); what is under test is train.FlatTrainer's
gradient exchange: flat-buffer bucketed all-reduce (average), the cross-rank agreement on the
"touched" parameter set, parameter broadcast at start, and that ranks stay bit-identical after steps.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _targets(rank, V):
    from tests import synth
    # rank 0: caption + vqa + cls + detection ; rank 1: captions only -> the box head is touched on ONE rank only
    if rank == 0:
        return synth.synth_targets(4, V, S=6)
    return synth.synth_targets(4, V, S=6, seed=7, tasks=('CocoCaptioning',))


def _worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from tests import synth, cpu_shim
    from tests.test_model_cpu import build_small, nested, V, B, H, W, Tl, PAD
    import gpv1_amd.ops as ops
    from gpv1_amd.train import FlatTrainer
    cpu_shim.install()
    ops.RT.set_precise(True)
    torch.manual_seed(100 + rank)                         # different init per rank: broadcast must fix it
    model, _ = build_small()
    with torch.no_grad():
        if rank == 1:
            model.detr_joiner.weight.add_(1.0)
    model.train()
    model.bert.model.p = 0.0
    tr = FlatTrainer(model, lr=1e-4, lr_backbone=1e-5, bucket_mb=8)
    images, mask, ids, attn = synth.synth_batch(B, H, W, Tl, V, seed=1234 + rank, pad_to=PAD)
    # one step by hand to look at the exchanged gradient
    tg = _targets(rank, V)
    _, tok = model.encode_answers(tg)
    for i, t in enumerate(tg):
        t['answer_token_ids'] = tok[i, 1:]
    loss = model(nested(images, mask), (ids, attn), tok, tg)
    tr.zero_grad()
    loss.backward()
    local = tr.G.clone()
    touched_local = tr.touched.clone()
    tr.allreduce_grads()
    res = {'local': local, 'avg': tr.G.clone() * tr.avg,           # (G = the SUM over ranks; 1 / world rides on the AdamW factor)
            'touched_local': touched_local, 'touched': tr.live_host()}
    res['grad_scale'], res['grad_norm'] = tr.grad_scale, float(tr.grad_norm())
    tr.step()
    for _ in range(1):
        tr.model.bert.model.p = 0.0
        tr.train_step(nested(images, mask), (ids, attn), _targets(rank, V))
    # train_step overlaps the exchange of everything behind the backbone segment with the backbone's backward
    res['milestones'], res['late_touch'], res['overlap'] = tr.milestones, tr.late_touch, tr.overlap
    res['milestone_log'], res['bucket_order'], res['stage_range'] = list(tr.milestone_log), list(tr.bucket_order), dict(tr.stage_range)
    res['left_after_backward'], res['backbone_end'], res['total'] = getattr(tr, 'left_after_backward', None), tr.backbone_end, tr.total
    res['P'] = tr.P.clone()
    res['names'] = [e[0] for e in tr.entries]
    # an UNANNOUNCED synchronous collective (an eval loop's all_gather, a checkpoint barrier: nobody calls
    # note_sync_collective) must arm the capture quiesce by itself -- the trainer installed the hooks; the per-step asynchronous
    # collectives must not (a ragged stream would wait 0.35 s per capture again)
    import gpv1_amd.misc as misc
    cc = misc.CollectiveClock
    arm = {}
    misc.ARM_BACKENDS = ('nccl', 'gloo')          # (this test's "device" backend is gloo)
    cc.pending = False
    dist.barrier()
    arm['barrier'], cc.pending = cc.pending, False
    t = torch.ones(4)
    dist.all_reduce(t)
    arm['all_reduce_sync'], cc.pending = cc.pending, False
    dist.all_reduce(t, async_op=True).wait()
    arm['all_reduce_async'], cc.pending = cc.pending, False
    objs = [{'rank': rank}]
    dist.broadcast_object_list(objs, src=0)
    arm['broadcast_object_list'], cc.pending = cc.pending, False
    misc.ARM_BACKENDS = ('nccl',)
    dist.barrier()                                # a host-side (gloo) collective: nothing on a device stream
    arm['gloo_barrier_as_host_channel'] = cc.pending
    # a whole step, with the step's own process group treated as the device backend: every device collective of train_step must
    # be asynchronous (RCCL's own stream); the one synchronous call is the host agreement channel (FlatTrainer.host_pg)
    sync_calls = []
    real = misc.note_sync_collective
    misc.ARM_BACKENDS = ('nccl', 'gloo')
    misc.note_sync_collective = lambda: (sync_calls.append(''.join(__import__('traceback').format_stack(limit=4))), real())[-1]
    tr.train_step(nested(images, mask), (ids, attn), _targets(rank, V))
    misc.note_sync_collective = real
    arm['train_step_sync_calls'] = len(sync_calls)
    arm['train_step_sync_is_host_channel'] = all('_any_rank_has_loss' in c for c in sync_calls)
    res['arm'] = arm
    torch.save(res, os.path.join(out, f'rank{rank}.pt'))
    dist.destroy_process_group()


@pytest.mark.timeout(1800)
def test_two_rank_gradient_exchange(tmp_path):
    runs = {}
    for overlap in ('1', '0'):                      # exchange overlapped with the backbone backward / after the pass
        os.environ['GPV_OVERLAP'] = overlap
        out = tmp_path / f'overlap{overlap}'
        out.mkdir()
        mp.spawn(_worker, args=(2, _free_port(), str(out)), nprocs=2, join=True)
        runs[overlap] = [torch.load(os.path.join(out, f'rank{r}.pt')) for r in range(2)]
    os.environ.pop('GPV_OVERLAP')
    r0, r1 = runs['1']
    assert r0['overlap'] and r0['late_touch'] is None and r1['late_touch'] is None
    for r in (r0, r1):                              # synchronous collectives arm the capture quiesce by themselves, asynchronous ones do not
        assert r['arm'] == {'barrier': True, 'all_reduce_sync': True, 'all_reduce_async': False, 'broadcast_object_list': True,
                            'gloo_barrier_as_host_channel': False, 'train_step_sync_calls': 1,
                            'train_step_sync_is_host_channel': True}, r['arm']
    # hand-over order: everything behind the backbone segment when the backward pass reaches the backbone, then the backbone's
    # stages as each one's weight gradients have been issued -- layer4 (the largest) while layer3 / layer2 still compute
    for r in (r0, r1):
        assert [m for m, _ in r['milestone_log']] == ['backbone', 'layer4', 'layer3', 'layer2']
        starts = [st for _, st in r['milestone_log']]
        assert starts == [r['backbone_end'], r['stage_range']['layer4'][0], r['stage_range']['layer3'][0], r['stage_range']['layer2'][0]]
        assert starts == sorted(starts, reverse=True) and starts[-1] == 0
        order = r['bucket_order']
        assert order == r0['bucket_order'] and sum(e - s for s, e in order) == r['total']
        firsts = [s for s, _ in order]
        nb = sum(1 for s in firsts if s >= r['backbone_end'])
        assert all(s >= r['backbone_end'] for s in firsts[:nb]) and firsts[nb:] == sorted(firsts[nb:], reverse=True)
        for lo, hi in r['stage_range'].values():                       # no bucket straddles a stage boundary
            assert all(e <= lo or s >= hi or (s >= lo and e <= hi) for s, e in order)
        assert r['left_after_backward'] == 0                           # every bucket was handed over during the backward pass
    assert not runs['0'][0]['overlap'] and runs['0'][0]['milestones'] == 0
    # same parameters after the steps whichever way the gradients were exchanged (same sums, same order per bucket)
    assert torch.equal(r0['P'], runs['0'][0]['P'])
    # exchanged gradient = average of the two local gradients, identical on both ranks
    avg = (r0['local'] + r1['local']) / 2
    # (G holds the SUM over ranks: logged norms go through grad_scale = 1 / world)
    assert r0['grad_scale'] == 0.5 and abs(r0['grad_norm'] - float(avg.norm())) <= 1e-5 * float(avg.norm())
    assert torch.equal(r0['avg'], r1['avg'])
    assert (r0['avg'] - avg).abs().max().item() <= 1e-6 * max(avg.abs().max().item(), 1.0)
    # the box head only received gradients on rank 0 (rank 1 had captions only) ...
    names = r0['names']
    ib = [i for i, n in enumerate(names) if 'bbox_embed' in n]
    assert r0['touched_local'][ib].all() and not r1['touched_local'][ib].any()
    # ... but after the MAX exchange both ranks agree it is touched, so AdamW updates it on both
    assert torch.equal(r0['touched'], r1['touched']) and r0['touched'][ib].all()
    # parameters were broadcast from rank 0 at construction and stay bit-identical after two steps
    assert torch.equal(r0['P'], r1['P'])
    # bf16 gradient exchange (GPV_GRAD_COMM=bf16: half the bytes on the ring): both ranks receive the same bf16 sums -> still
    # bit-identical replicas; the exchanged gradient is the fp32 average to bf16 precision
    os.environ['GPV_GRAD_COMM'] = 'bf16'
    try:
        out = tmp_path / 'bf16'
        out.mkdir()
        mp.spawn(_worker, args=(2, _free_port(), str(out)), nprocs=2, join=True)
    finally:
        os.environ.pop('GPV_GRAD_COMM')
    b0, b1 = [torch.load(os.path.join(out, f'rank{r}.pt')) for r in range(2)]
    assert torch.equal(b0['avg'], b1['avg']) and torch.equal(b0['P'], b1['P'])
    avg = (b0['local'] + b1['local']) / 2
    assert (b0['avg'] - avg).abs().max().item() <= 8e-3 * max(avg.abs().max().item(), 1e-6)
    assert not torch.equal(b0['avg'], r0['avg'])                   # (it really went through bf16)


def _worker_noloss(rank, world, port, out):
    """rank 1 never has an applicable target (losses.py:163-169 returns None): it must still enter every collective."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from tests import synth, cpu_shim
    from tests.test_model_cpu import build_small, nested, V, B, H, W, Tl, PAD
    import gpv1_amd.ops as ops
    from gpv1_amd.train import FlatTrainer
    cpu_shim.install()
    ops.RT.set_precise(True)
    torch.manual_seed(5 + rank)
    model, _ = build_small()
    model.train()
    model.bert.model.p = 0.0
    tr = FlatTrainer(model, lr=1e-4, lr_backbone=1e-5, bucket_mb=8)
    images, mask, ids, attn = synth.synth_batch(B, H, W, Tl, V, seed=1234 + rank, pad_to=PAD)
    none_t = [{'task': 'SomethingElse'} for _ in range(B)]
    res = {'ret': []}
    # step 1: nobody has a loss -> every rank skips (the reference's behaviour), no optimizer step
    res['ret'].append(tr.train_step(nested(images, mask), (ids, attn), [dict(t) for t in none_t]) is None)
    res['steps_after_all_none'] = tr.step_count
    P0 = tr.P.clone()
    # steps 2-3: rank 0 has targets, rank 1 has none: rank 1 contributes zeros and steps with the others
    for _ in range(2):
        tg = synth.synth_targets(B, V, S=6) if rank == 0 else [dict(t) for t in none_t]
        res['ret'].append(tr.train_step(nested(images, mask), (ids, attn), tg) is None)
    res['steps'] = tr.step_count
    res['P'] = tr.P.clone()
    res['moved'] = bool((tr.P != P0).any())
    res['dropout_seed'] = (ops.RT.seed, ops.RT.seed_explicit)
    torch.save(res, os.path.join(out, f'rank{rank}.pt'))
    dist.destroy_process_group()


@pytest.mark.timeout(1800)
def test_rank_without_applicable_target_stays_in_step(tmp_path):
    mp.spawn(_worker_noloss, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = [torch.load(os.path.join(tmp_path, f'rank{r}.pt')) for r in range(2)]
    assert r0['ret'] == [True, False, False] and r1['ret'] == [True, True, True]
    assert r0['steps_after_all_none'] == 0 and r1['steps_after_all_none'] == 0
    assert r0['steps'] == 2 and r1['steps'] == 2
    assert r0['moved'] and torch.equal(r0['P'], r1['P'])
    # nobody seeded the dropout stream: the trainer folded the rank into the default seed (ops.Runtime.per_rank_seed) -- other masks per rank,
    # as the reference's unseeded processes have (exp/gpv/train_distr.py never seeds)
    assert r0['dropout_seed'] == (0x5EED, False) and r1['dropout_seed'][0] != 0x5EED and r1['dropout_seed'][1] is False


def test_per_rank_dropout_seed_rule():
    """ops.Runtime.per_rank_seed: rank 0 keeps the default, other ranks get their own, an explicit manual_seed is left alone, the draws of
    two ranks differ from the first one on."""
    from gpv1_amd.ops import Runtime
    a, b, c = Runtime(), Runtime(), Runtime()
    a.per_rank_seed(0); b.per_rank_seed(1); b.per_rank_seed(1)
    assert a.seed == 0x5EED and b.seed != a.seed and 0 <= b.seed < (1 << 48)
    assert [a.next_seed() for _ in range(4)] != [b.next_seed() for _ in range(4)]
    c.manual_seed(7)
    c.per_rank_seed(3)
    assert c.seed == 7 and c.seed_explicit
    seeds = set()
    for r in range(64):
        t = Runtime(); t.per_rank_seed(r); seeds.add(t.seed)
    assert len(seeds) == 64
