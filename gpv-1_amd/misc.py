"""Batch container + collate helpers (reference: utils/detr_misc.py:267-322, 394-410)."""
import contextlib
import os
import gc
from typing import List, Optional

import torch


class NestedTensor(object):
    """detr_misc.py:302-322"""

    def __init__(self, tensors, mask: Optional[torch.Tensor], all_valid: Optional[bool] = None):
        self.tensors = tensors
        self.mask = mask
        # host-side knowledge that no pixel is padding (all images the same size): lets the position encoding use its
        # cached constant without asking the device (`mask.any()` is a host<->device sync per step).  None = unknown.
        self.all_valid = all_valid

    def to(self, device):
        mask = self.mask.to(device) if self.mask is not None else None
        return NestedTensor(self.tensors.to(device), mask, self.all_valid)

    def decompose(self):
        return self.tensors, self.mask

    def __repr__(self):
        return str(self.tensors)


def nested_tensor_from_tensor_list(tensor_list: List[torch.Tensor]):
    """zero-pad (3,H,W) images to the batch max and build the bool padding mask (detr_misc.py:282-299)."""
    if isinstance(tensor_list, torch.Tensor):
        tensor_list = list(tensor_list) if tensor_list.ndim == 4 else [tensor_list]
    if tensor_list[0].ndim != 3:
        raise ValueError('not supported')
    c = tensor_list[0].shape[0]
    h = max(img.shape[1] for img in tensor_list)
    w = max(img.shape[2] for img in tensor_list)
    b = len(tensor_list)
    tensor = torch.zeros((b, c, h, w), dtype=tensor_list[0].dtype, device=tensor_list[0].device)
    mask = torch.ones((b, h, w), dtype=torch.bool, device=tensor_list[0].device)
    for img, pad_img, m in zip(tensor_list, tensor, mask):
        pad_img[:img.shape[0], :img.shape[1], :img.shape[2]].copy_(img)
        m[:img.shape[1], :img.shape[2]] = False
    return NestedTensor(tensor, mask, all(img.shape[1] == h and img.shape[2] == w for img in tensor_list))


class PinnedStager:
    """small host->device transfers without a host<->device synchronisation: a pageable-memory copy
    (`torch.tensor(list, device='cuda')`) waits for everything queued on the stream, i.e. the previous training step.
    A ring of pinned staging buffers + non_blocking copies keeps the host free to run ahead; the ring is deep enough
    (32 slots -- a training step stages up to five small tensors --, each protected by the event of its last copy) that a slot is never rewritten while its copy is pending."""

    def __init__(self, slots=32, nbytes=1 << 16):
        self.slots, self.nbytes, self.bufs, self.events, self.i = slots, nbytes, None, None, 0

    def to_device(self, data, dtype, device):
        t = torch.as_tensor(data, dtype=dtype)
        dev = torch.device(device)
        if dev.type != 'cuda' or t.numel() * t.element_size() > self.nbytes:
            return t.to(dev)
        if self.bufs is None:
            self.bufs = [torch.empty(self.nbytes, dtype=torch.uint8).pin_memory() for _ in range(self.slots)]
            self.events = [None] * self.slots
        k = self.i
        self.i = (k + 1) % self.slots
        if self.events[k] is not None:
            self.events[k].synchronize()                      # only ever waits when the host is 32 transfers ahead
        n = t.numel() * t.element_size()
        host = self.bufs[k][:n].view(dtype).view(t.shape)
        host.copy_(t)
        out = host.to(dev, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        self.events[k] = ev
        return out


STAGER = PinnedStager()
_DESC_STAGER = PinnedStager(slots=16, nbytes=1 << 18)


def upload_bytes(raw, device):
    """a packed descriptor array (bytes) -> uint8 device tensor through the pinned ring (no per-call pinned allocation, no sync)"""
    return _DESC_STAGER.to_device(torch.frombuffer(bytearray(raw), dtype=torch.uint8), torch.uint8, device)


def collate_fn(batch):
    """detr_misc.py:267-270"""
    batch = list(zip(*batch))
    batch[0] = nested_tensor_from_tensor_list(batch[0])
    return tuple(batch)


@torch.no_grad()
def accuracy(output, target, topk=(1,)):
    """detr_misc.py:394-410"""
    if target.numel() == 0:
        return [torch.zeros([], device=output.device)]
    maxk = max(topk)
    _, pred = output.topk(maxk, 1, True, True)
    correct = pred.t().eq(target.view(1, -1).expand_as(pred.t()))
    return [correct[:k].reshape(-1).float().sum(0) * (100.0 / target.size(0)) for k in topk]


class AttrDict(dict):
    """Minimal OmegaConf stand-in: attribute + item access, real bools, `.items()`."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    @staticmethod
    def wrap(d):
        if isinstance(d, dict):
            return AttrDict({k: AttrDict.wrap(v) for k, v in d.items()})
        if isinstance(d, (list, tuple)):
            return [AttrDict.wrap(v) for v in d]
        return d


class CollectiveClock:
    """What a stream capture has to know about this process's collectives (quiesce_collectives).
    pending: a SYNCHRONOUS collective (one that ran on the caller's stream: dist.broadcast / barrier / all_reduce without async_op
    under the NCCL = RCCL backend) has been issued since the device was last observed idle for QUIET seconds.  Asynchronous
    collectives -- everything train.FlatTrainer issues per step -- run on RCCL's own stream, which never captures: the watchdog's
    event queries on them are legal whatever the compute stream does, so they do not arm the clock."""
    QUIET = 0.35                   # three polling periods of ProcessGroupNCCL's watchdog (100 ms)
    MODE = os.environ.get('GPV_QUIESCE', 'auto')      # auto | always (round 4: sleep before every capture) | never
    pending = True                 # (process-group construction and whatever ran before the trainer existed)
    idle_since = None
    sleeps = 0                     # how often a capture actually had to wait (tests: a ragged stream in steady state: 0)
    calls = 0


def note_sync_collective():
    """call after issuing a synchronous collective on a stream that may capture later (parameter broadcast, barriers)"""
    CollectiveClock.pending = True
    CollectiveClock.idle_since = None


# Round 6: the clock is armed by the collectives THEMSELVES, not by the callers' good manners.  Every
# torch.distributed entry point that runs on the caller's stream when called without async_op=True is wrapped once per process
# (install_collective_hooks: idempotent, installed by train.init_process_group, FlatTrainer and the first quiesce_collectives with a
# process group alive): an eval loop's all_gather of metrics, a checkpoint barrier or a driver's broadcast_object_list now make
# the next capture wait for the watchdog exactly like the trainer's own announced ones.  A collective reached through a name bound
# BEFORE the hooks went in (`from torch.distributed import barrier` at import time) is the documented residue: such drivers call
# note_sync_collective() or run with GPV_QUIESCE=always.
_SYNC_COLLECTIVES = ('broadcast', 'all_reduce', 'all_reduce_coalesced', 'reduce', 'all_gather', 'all_gather_into_tensor',
                     'all_gather_coalesced', 'gather', 'scatter', 'reduce_scatter', 'reduce_scatter_tensor', 'all_to_all',
                     'all_to_all_single', 'barrier', 'monitored_barrier', 'send', 'recv', 'broadcast_object_list',
                     'all_gather_object', 'gather_object', 'scatter_object_list', '_all_gather_base', '_reduce_scatter_base')
_HOOKED = [False]
ARM_BACKENDS = ('nccl',)            # backends whose synchronous collectives run on the caller's DEVICE stream (tests add 'gloo')


def _runs_on_device_stream(group):
    """does a synchronous collective of this process group run on the calling thread's device stream?  (gloo -- the trainer's
    host-side agreement channel -- runs on the host: nothing for a capture to trip over)"""
    import torch.distributed as dist
    try:
        return any(b in str(dist.get_backend(group)) for b in ARM_BACKENDS)
    except Exception:
        return True                 # (unknown group: the safe side)


def install_collective_hooks():
    """wrap torch.distributed's collectives so that a synchronous call arms CollectiveClock (asynchronous ones -- async_op=True: on
    RCCL's own stream, which never captures -- do not).  Returns the number of entry points wrapped by THIS call."""
    if _HOOKED[0]:
        return 0
    import functools
    import torch.distributed as dist
    if not dist.is_available():
        return 0
    mods = [dist]
    c10d = getattr(dist, 'distributed_c10d', None)
    if c10d is not None:
        mods.append(c10d)
    n = 0
    for name in _SYNC_COLLECTIVES:
        fn = getattr(mods[-1], name, None) or getattr(dist, name, None)
        if fn is None or getattr(fn, '_gpv_sync_hook', False):
            continue

        def make(fn):
            @functools.wraps(fn)
            def hooked(*a, **k):
                try:
                    return fn(*a, **k)
                finally:
                    if not k.get('async_op', False) and _runs_on_device_stream(k.get('group')):
                        note_sync_collective()
            hooked._gpv_sync_hook = True
            hooked._gpv_wrapped = fn
            return hooked
        h = make(fn)
        for m in mods:
            if getattr(m, name, None) is fn:
                setattr(m, name, h)
        n += 1
    _HOOKED[0] = True
    note_sync_collective()             # whatever ran before the hooks existed is unknown: the next capture waits once
    return n


def quiesce_collectives():
    """With a process group alive, a stream capture must not begin while ProcessGroupNCCL's watchdog thread still polls the
    completion event of a collective that ran ON THE STREAM ABOUT TO CAPTURE (HIP: hipErrorCapturedEvent on the query, the capture
    invalidated, the watchdog's exception ends the process -- found with GPV_FORCE_COMM=1 on one GPU: 3 of 20 runs): idle device,
    then three polling periods for the watchdog to retire what it was watching.
    Round 5: that wait is paid only when such a collective is actually outstanding (CollectiveClock.pending).  Round 4 slept 0.35 s in
    front of EVERY capture -- two per new batch signature, on every rank in lock-step: seconds of a ragged stream's first epoch.  The
    per-step collectives of the trainer are asynchronous (RCCL's stream, never captured) and do not arm the clock; the synchronous
    ones (parameter broadcast at construction, barriers of the drivers, any eval / checkpoint collective) do: torch.distributed's
    entry points are wrapped (install_collective_hooks, round 6) and arm the clock themselves; note_sync_collective() remains for
    collectives issued through other bindings.  GPV_QUIESCE=always restores the unconditional wait."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or not torch.cuda.is_available():
        return
    import time
    cc = CollectiveClock
    cc.calls += 1
    install_collective_hooks()          # (no-op after the first call; a first call arms the clock: history unknown)
    torch.cuda.synchronize()
    if cc.MODE == 'never' or (cc.MODE != 'always' and not cc.pending):
        return
    now = time.monotonic()
    if cc.idle_since is None or cc.MODE == 'always':
        cc.idle_since = now
    wait = cc.QUIET - (now - cc.idle_since)
    if wait > 0:
        time.sleep(wait)
        cc.sleeps += 1
    cc.pending = False


@contextlib.contextmanager
def capture_guard():
    """No cyclic garbage collection while a stream is capturing: a CUDAGraph that becomes collectable in there (the captured graphs
    of a model that was dropped earlier -- GPV and its decoders reference each other, so only the cycle collector frees them) would
    be destroyed inside the capture, which HIP refuses and PyTorch's destructor turns into std::terminate (tools/fuzz_decode.py
    found it: a new model per configuration)."""
    quiesce_collectives()
    was = gc.isenabled()
    gc.collect()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()

