"""The N > 1 training path under the REAL backend on a one-GPU box: `bench.py --gpus 1` with GPV_FORCE_COMM=1 builds an RCCL
process group of one rank, and FlatTrainer then runs everything it runs on a node -- parameter broadcast, the backward pass
cut into stage graphs with the bucketed asynchronous all-reduces issued between them (train.py FlatTrainer._on_milestone), the
bf16 staging option, the gloo agreement channel -- with collectives that move nothing.  What this pins: RCCL initialises in
this image, its streams / events coexist with the hipGraph replays (strict mode: a failed capture fails the run), the
milestone order on the device path, and that the exchange leaves the numbers alone (same final loss as the run without it).
Reference: the DistributedDataParallel wrap of /root/reference/exp/gpv/train_distr.py:176-179."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra_env, port):
    env = dict(os.environ)
    env.update({'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port), 'HSA_ENABLE_IPC_MODE_LEGACY': '0', 'GPV_GRAPHS_STRICT': '1'})
    env.update(extra_env)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '4', '--warmup', '3', '--no-cpu-baseline',
           '--no-decode', '--no-ragged', '--no-extra']
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{')][-1]
    return json.loads(line)


def test_forced_single_rank_rccl_exchange_runs_the_multi_gpu_path_and_changes_nothing():
    plain = _bench({}, 29541)
    assert 'comm' not in plain
    forced = _bench({'GPV_FORCE_COMM': '1'}, 29542)
    c = forced['comm']
    assert c['backend'] == 'nccl' and c['rccl_ranks'] == 1
    assert forced['graphs']['enabled'] and 'error' not in forced['graphs'] and c['eager_steps'] <= 2
    # hand-over order of the last (graphed) step: everything behind the DETR head, then the head with layer4, layer3, layer2
    # captures wait for RCCL's watchdog only behind SYNCHRONOUS collectives (parameter broadcast, the bench's barriers), not per capture
    assert c['capture_quiesce']['mode'] == 'auto' and c['capture_quiesce']['calls'] >= 2 and c['capture_quiesce']['sleeps'] <= 2, c['capture_quiesce']
    assert c['milestones_last_step'] == ['head', 'layer4', 'layer3', 'layer2']
    assert c['left_after_backward_bytes'] is not None and c['left_after_backward_bytes'] <= 30 << 20
    assert c['bytes_per_rank_per_step'] > 400 << 20
    # one rank: SUM over ranks / world is the identity -- the step must be the step without the exchange
    a, b = plain['config']['final_loss'], forced['config']['final_loss']
    assert abs(a - b) <= 2e-3 * abs(a), (a, b)                         # (atomics order: not bit-reproducible run to run)
    half = _bench({'GPV_FORCE_COMM': '1', 'GPV_GRAD_COMM': 'bf16'}, 29543)
    assert half['comm']['grad_comm_dtype'] == 'bfloat16' and half['graphs']['enabled']
    h = half['config']['final_loss']
    assert h == h and abs(h - a) <= 5e-2 * abs(a), (a, h)                  # gradients rounded to bf16 once: close, not equal


_UNANNOUNCED = r'''
import os, sys, json
sys.path.insert(0, os.environ['GPV_ROOT'])
import torch
import torch.distributed as dist
from tests import synth
from tests.test_model_cpu import build_small, nested, V, B, H, W, Tl, PAD
import gpv1_amd.misc as misc
from gpv1_amd.train import FlatTrainer, init_process_group
torch.cuda.set_device(0)
init_process_group(0, 1, 'cuda:0')                      # RCCL, one rank; installs the collective hooks
model, _ = build_small()
model.to('cuda').train()
model.bert.model.p = 0.0
tr = FlatTrainer(model, lr=1e-4, lr_backbone=1e-5, bucket_mb=8)
assert tr.comm and dist.get_backend() == 'nccl'
cc = misc.CollectiveClock
losses, sleeps = [], []
tg = synth.synth_targets(B, V, S=6, tasks=('CocoCaptioning',))
for T_q in (5, 6, 7, 8):                                   # four query lengths (exact classes up to 8 tokens) = four signatures = eight captures
    images, mask, ids, attn = [t.cuda() for t in synth.synth_batch(B, H, W, T_q, V, seed=1234, pad_to=PAD)]
    loss = tr.train_step(nested(images, mask), (ids, attn), tg)          # first sight of the signature: eager
    before = cc.sleeps
    # what an eval loop / checkpoint writer does between steps, WITHOUT telling anybody (ADVICE r5): synchronous collectives on the
    # compute stream, then immediately a step whose signature is captured now -- inside the watchdog's 100 ms polling period
    dist.barrier()
    t = torch.ones(8, device='cuda')
    dist.all_reduce(t)
    loss = tr.train_step(nested(images, mask), (ids, attn), tg)          # second sight: GraphedBody captures F1 | F2 | B1 | B2
    loss = tr.train_step(nested(images, mask), (ids, attn), tg)          # replay
    torch.cuda.synchronize()
    losses.append(float(loss))
    sleeps.append(cc.sleeps - before)
print(json.dumps({'losses': losses, 'sleeps': sleeps, 'graph_steps': tr.graph_steps, 'eager_steps': tr.eager_steps, 'graphs': bool(tr.graphs),
                  'bodies': len(tr._bodies), 'mode': cc.MODE}))
dist.destroy_process_group()
'''


@pytest.mark.timeout(900)
def test_unannounced_synchronous_collective_before_a_capture_is_quiesced():
    """in GPV_QUIESCE=auto a capture skipped the watchdog wait unless a caller had announced its synchronous
    collective (note_sync_collective); a checkpoint barrier or an eval all-reduce nobody announced left the next capture to die with
    hipErrorCapturedEvent (round 4: 3 of 20 runs).  Round 6: torch.distributed's entry points arm the clock themselves
    (misc.install_collective_hooks).  Real backend, one rank: unannounced barrier + all_reduce right in front of four new-signature
    captures -- every one of them waits once, none fails (strict mode: a failed capture raises)."""
    env = dict(os.environ)
    env.update({'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': '29547', 'HSA_ENABLE_IPC_MODE_LEGACY': '0', 'GPV_GRAPHS_STRICT': '1',
                'GPV_FORCE_COMM': '1', 'GPV_ROOT': ROOT, 'RANK': '0', 'WORLD_SIZE': '1'})
    r = subprocess.run([sys.executable, '-c', _UNANNOUNCED], cwd=ROOT, env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert out['mode'] == 'auto' and out['graphs'] and out['bodies'] == 4 and out['graph_steps'] >= 8, out
    assert all(s >= 1 for s in out['sleeps']), out                  # the unannounced collectives armed the clock every time
    assert all(l == l for l in out['losses']), out
