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
