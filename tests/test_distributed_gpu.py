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


def _worker(rank, world, port, out, comm='fp32'):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from tests import synth
    from tests.test_model_cpu import build_small, nested, V, B, H, W, Tl, PAD
    import gpv1_amd.ops as ops
    from gpv1_amd.train import FlatTrainer
    ops.RT.set_precise(False)
    torch.manual_seed(100 + rank)                          # different init per rank: the broadcast must fix it
    model, _ = build_small()
    model.to('cuda').train()
    model.bert.model.p = 0.0
    tr = FlatTrainer(model, lr=1e-4, lr_backbone=1e-5, bucket_mb=8, grad_comm_dtype=comm)
    assert tr.world == 2 and tr.overlap
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
        assert [m for m, _ in tr.milestone_log] == ['backbone', 'layer4', 'layer3', 'layer2'] and tr.late_touch is None, (tr.milestone_log, tr.late_touch)
        losses.append(float(loss.detach()))
    torch.cuda.synchronize()
    # steps 2.. replay the captured hipGraphs; with two ranks the DETR weight-gradient group is flushed at the END of B1 (every
    # gradient behind the backbone segment is complete when its buckets go to the all-reduce between B1 and B2)
    assert tr.graphs and len(tr._bodies) >= 1 and tr.graph_steps >= 3, (tr.graphs, len(tr._bodies), tr.graph_steps, tr.eager_steps)
    torch.save({'P': tr.P.cpu(), 'live': tr.live_host(), 'touched_local': tr.touched.clone(), 'losses': losses,
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
