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


def quiesce_collectives():
    """With a process group alive, a stream capture must not begin while ProcessGroupNCCL's watchdog thread still polls the
    completion event of a collective that ran ON THE STREAM ABOUT TO CAPTURE (HIP: hipErrorCapturedEvent on the query, the capture
    invalidated, the watchdog's exception ends the process -- found with GPV_FORCE_COMM=1 on one GPU: 3 of 20 runs): idle device,
    then three polling periods for the watchdog to retire what it was watching.
    Round 5: that wait is paid only when such a collective is actually outstanding (CollectiveClock.pending).  Round 4 slept 0.35 s in
    front of EVERY capture -- two per new batch signature, on every rank in lock-step: seconds of a ragged stream's first epoch.  The
    per-step collectives of the trainer are asynchronous (RCCL's stream, never captured) and do not arm the clock; the synchronous
    ones (parameter broadcast at construction, barriers of the drivers) do, through note_sync_collective().  GPV_QUIESCE=always
    restores the unconditional wait."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or not torch.cuda.is_available():
        return
    import time
    cc = CollectiveClock
    cc.calls += 1
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

