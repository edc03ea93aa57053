"""Autograd layer over the C-ABI kernels (gpv1_amd.hip).

Design
  * Activations are 2-D row-major [rows, channels] tensors in the COMPUTE dtype
    (``RT.dtype``: bf16 for production, fp32 "precise" split-bf16 MFMA mode for parity runs).
  * Parameters stay fp32 nn.Parameters with the reference's names/shapes (state-dict compatible).
    Kernels read low-precision compute copies (W and W^T) that are rebuilt whenever
    ``RT.weights_epoch`` changes (after an optimizer step / load_state_dict).
  * Parameter gradients never travel through torch.autograd: every backward accumulates straight
    into ``param.grad`` (fp32, pre-allocated, e.g. a view of one flat buffer) with the GEMM/conv
    wgrad kernels (fp32 atomics, split-K) -- so the data-parallel all-reduce can run over one flat
    buffer (train.py).  torch.autograd only carries activation gradients between our Functions.
  * No CPU / eager fallback: everything below calls the HIP library and raises if it is missing.
"""
import math

import os
import weakref

import torch
from torch.autograd import Function

from . import hip

ACT_NONE, ACT_RELU, ACT_GELU = hip.ACT_NONE, hip.ACT_RELU, hip.ACT_GELU


class Runtime:
    def __init__(self):
        self.dtype = torch.bfloat16
        self.weights_epoch = 0       # bumped after every optimizer step (trainer-managed parameters changed)
        self.static_epoch = 0        # bumped only when ANY tensor may have changed (load_state_dict, .to(), manual edits)
        self.seed = 0x5EED
        self.seed_explicit = False       # manual_seed() was called: per_rank_seed() leaves the stream alone
        self._ctr = 0
        self.cache = {}
        self.backward_milestone = None   # set by train.FlatTrainer for the duration of a backward pass (gradient-exchange overlap)
        self.split = None                # set by train.GraphedBody while it captures / replays: the backbone runs outside autograd
        self.seed_dev = None             # device int64 word lent to the library as the dropout seed epoch (hipGraph replays)
        self.defer_list = None           # set by train.GraphedBody while it captures a backward: deferred weight-gradient launches
        self.backward_boundary = None    # callback(tag) from ops.BoundaryFn.backward (same capture)
        self.branch_stream = None        # the side stream a capture owner lends to ops.Branch for the duration of its capture
        self.multi_wait = None           # set by train.GraphedBody while it captures F2: joins the branch that ran refresh_multi

    def set_precise(self, on=True):
        self.dtype = torch.float32 if on else torch.bfloat16
        self.cache.clear()
        # the compute copies just freed are baked into every captured graph (GPV._igraphs, GreedyKVDecoder.graphs,
        # train.GraphedBody): their keys carry the epochs, so bumping them retires those graphs
        self.bump_weights()

    def enable_seed_epoch(self, device):
        """install the device-resident seed epoch (include/gpv_hip.h: gpv_set_seed_device); idempotent"""
        if self.seed_dev is None or self.seed_dev.device != torch.device(device):
            self.seed_dev = torch.zeros(1, dtype=torch.int64, device=device)
            hip.set_seed_device(self.seed_dev)
        return self.seed_dev

    def bump_weights(self, everything=True):
        """call after parameters changed.  everything=False: only the parameters a trainer manages changed
        (optimizer step) -> frozen weights (BERT, conv1/layer1, FrozenBN folds) keep their compute copies."""
        self.weights_epoch += 1
        if everything:
            self.static_epoch += 1

    def epoch_of(self, p):
        return self.weights_epoch if getattr(p, '_gpv_managed', False) else self.static_epoch

    def manual_seed(self, s):
        """restart the dropout stream: the host counter behind next_seed() AND the device-resident epoch the replayed graphs add to their
        frozen seeds (train.GraphedBody bumps it once per replayed step; it is process-wide and outlives trainers) -- without the second,
        a graphed run from the same seed drew other masks than the run before it.  Not inside a capture (the reset would become a node)."""
        self.seed, self._ctr, self.seed_explicit = int(s), 0, True
        if self.seed_dev is not None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError('Runtime.manual_seed inside a stream capture')
            self.seed_dev.zero_()

    def per_rank_seed(self, rank):
        """A trainer on a process group calls this once: unless the caller seeded the stream itself (manual_seed), fold the rank into the
        default seed so that the data-parallel ranks draw DIFFERENT dropout masks.  The reference never seeds (exp/gpv/train_distr.py):
        each of its processes starts from torch's own per-process default seed, i.e. independent masks per rank; with one default seed
        here every rank applied the same masks to its shard of the global batch.  Idempotent per rank; the host counter is kept."""
        if not self.seed_explicit and rank:
            self.seed = (0x5EED + int(rank) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFF

    def next_seed(self):
        self._ctr += 1
        return (self.seed * 0x9E3779B1 + self._ctr * 0x85EBCA77) & 0xFFFFFFFFFFFF


RT = Runtime()


def off_critical_path(fn, *tensors):
    """Weight gradients are needed by nobody until the optimizer.  While train.GraphedBody captures the backward of the model
    body (single GPU) the weight-gradient GEMMs of the transformer / co-attention / text-decoder layers -- ~140 launches of
    36-600 workgroups, latency-bound -- are not launched in place: they are collected and captured as ONE parallel branch
    of the backbone's backward graph (fork at its start, join at its end), where they fill the tails of the convolution
    launches.  A fork per gradient was measured and lost: every cross-branch edge costs more than the overlap wins.
    Outside such a capture fn() simply runs in place.  `tensors` keep the operands fn reads alive with the closure."""
    if RT.defer_list is None:
        fn()
        return
    RT.defer_list.append((fn, tensors, None))


def wgrad_linear(dz, x2, wg, bg, N, K, M, split, lda=None):
    """dW[N,K] += dz[M,N]^T x2[M,K] (+ bias gradient = column sums of dz): in place, or -- while a backward is being captured --
    handed to the deferred list as a PROBLEM (train.GraphedBody groups the eligible ones into gpv_gemm_tt_group launches).
    lda: row pitch of dz when it is a column slice of a wider buffer (multi_linear)"""
    lda = N if lda is None else lda

    def run():
        hip.gemm(dz, x2, wg, N, K, M, lda, K, K, layoutA=hip.TRANS, layoutB=hip.TRANS, accumulate=True, split_k=split, a_rowsum=bg)
    if RT.defer_list is None:
        run()
        return
    prob = None
    if wg.stride(1) == 1 and hip.tt_group_ok(dz, x2, wg, N, K, M, lda, K, wg.stride(0)):
        prob = (dz, x2, wg, bg, N, K, M, lda, K, wg.stride(0))
    RT.defer_list.append((run, (dz, x2, wg), prob))            # (tensors[2]: the gradient written -- train.GraphedBody checks where it lives)


class BoundaryFn(Function):
    """identity in the forward; its backward tells the runtime that every node created after this point has run
    (autograd orders ready nodes by creation sequence) -- a place where train.GraphedBody forks work onto a side branch"""

    @staticmethod
    def forward(ctx, x, tag):
        ctx.tag = tag
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        if RT.backward_boundary is not None:
            RT.backward_boundary(ctx.tag)
        return g, None


def boundary(x, tag):
    return BoundaryFn.apply(x, tag) if (torch.is_grad_enabled() and x.requires_grad) else x


def ensure_grad(p):
    """fp32 gradient buffer with the SAME physical layout as the parameter."""
    if p.grad is None:
        p.grad = torch.zeros_like(p, memory_format=torch.preserve_format)
    touch = getattr(p, '_gpv_touch', None)          # train.FlatTrainer keeps the torch-1.6 "touched" set
    if touch is not None:
        touch()
    return p.grad


# --------------------------------------------------------------------------------------------
# weight references
# --------------------------------------------------------------------------------------------
def _lp(p):
    """W [N,K] of a 2-D (or [N,K,1,1]) parameter in the compute dtype.

    precise mode: the fp32 parameter itself.  bf16 mode: parameters a FlatTrainer manages carry ``_gpv_lp`` -- a view
    into the trainer's flat bf16 mirror that the AdamW kernel rewrites in the same pass as the fp32 master weight (no
    cast launches per step); everything else (frozen weights, inference without a trainer) is cast once per epoch.
    No transposed copy exists: backward-data GEMMs read W as a reduction-major operand (GPV_TRANS), which the kernel
    stages with the LDS transpose read at the same speed (tools/bench_dx_layout.py)."""
    N = p.shape[0]
    K = p.numel() // N
    if RT.dtype == torch.float32:
        src = p.detach().reshape(N, K)
        return src if src.is_contiguous() else src.contiguous()
    lp = getattr(p, '_gpv_lp', None)
    if lp is not None:
        if p._gpv_lp_static != RT.static_epoch:          # load_state_dict / manual edit since the mirror was written
            hip.cast(p._gpv_flat, lp, lp.numel())
            p._gpv_lp_static = RT.static_epoch
        return lp.view(N, K)
    key = ('lin', id(p), RT.dtype)
    hit = RT.cache.get(key)
    ep = RT.epoch_of(p)
    if hit is not None and hit[0] == ep:
        return hit[1]
    src = p.detach().reshape(N, K)
    if not src.is_contiguous():
        src = src.contiguous()
    w = torch.empty(N, K, device=p.device, dtype=RT.dtype)
    hip.cast(src, w, N * K)
    RT.cache[key] = (ep, w)
    return w


_T_PARAMS = {}            # id -> weakref of every parameter a transposed mirror has been asked for (refresh_transposed walks it)
DX_MIRROR = os.environ.get('GPV_DX_MIRROR', '1') != '0'


def _lpT(p):
    """W^T [K, N] of a 2-D parameter in the compute dtype, rebuilt (one cast-transpose launch from the fp32 master) when the
    parameter's epoch changes.  Backward-data GEMMs dX = dY W then run as K-major x K-major GEMMs on the pipelined
    direct-to-LDS kernels: 9600 x 256 x 2048 (dz W1 of the DETR feed-forward) 54 -> 24 us against the reduction-major-B form
    of the register-staged kernel (tools/bench_ffn.py).  Under train.GraphedBody the rebuilds run on a branch of F1."""
    N = p.shape[0]
    K = p.numel() // N
    key = ('linT', id(p), RT.dtype)
    hit = RT.cache.get(key)
    ep = RT.epoch_of(p)
    if hit is not None and hit[2]() is not p:            # id() recycled by a parameter of another model
        hit = None
    if hit is not None and hit[0] == ep:
        return hit[1]
    src = getattr(p, '_gpv_flat', None)
    src = p.detach().reshape(N, K) if src is None else src.view(N, K)
    if not src.is_contiguous():
        src = src.contiguous()
    wt = hit[1] if hit is not None else torch.empty(K, N, device=p.device, dtype=RT.dtype)      # same buffer every epoch
    hip.cast_rowscale_t(src.float() if src.dtype != torch.float32 else src, None, None, wt, N, K)
    ref = weakref.ref(p)
    RT.cache[key] = (ep, wt, ref)
    _T_PARAMS[id(p)] = ref
    return wt


def refresh_transposed():
    """rebuild every stale transposed mirror now (train.GraphedBody: on a graph branch beside the backbone forward)"""
    items = []
    for k, r in list(_T_PARAMS.items()):
        p = r()
        hit = RT.cache.get(('linT', k, RT.dtype))
        if p is None or hit is None or hit[2]() is not p:
            _T_PARAMS.pop(k, None)
            continue
        ep = RT.epoch_of(p)
        if hit[0] == ep:
            continue
        N = p.shape[0]
        src = getattr(p, '_gpv_flat', None)
        src = p.detach().reshape(N, -1) if src is None else src.view(N, -1)
        if src.dtype != torch.float32 or not src.is_contiguous():
            _lpT(p)                                   # (not the trainer's fp32 master layout: one by one)
            continue
        items.append((src, hit[1]))
        RT.cache[('linT', k, RT.dtype)] = (ep, hit[1], hit[2])
    hip.cast_transpose_group(items)                  # one launch for all of them (gpv_cast_transpose_group)


class W:
    """rows [r0:r1) of a Linear-style weight parameter [N_total, K] (+ matching bias slice)."""

    def __init__(self, weight, bias=None, r0=0, r1=None):
        self.weight, self.bias = weight, bias
        self.r0 = r0
        self.r1 = weight.shape[0] if r1 is None else r1
        self.N = self.r1 - self.r0
        self.Ntot = weight.shape[0]
        self.K = weight.numel() // weight.shape[0]

    def lp(self):
        return _lp(self.weight)[self.r0:self.r1]      # [N, K] contiguous rows, ld = K

    def mirror_ok(self):
        return DX_MIRROR and self.weight.dim() == 2 and self.K % 8 == 0 and self.Ntot % 8 == 0 and self.r0 % 8 == 0 and self.N % 8 == 0

    def lpT(self):
        return _lpT(self.weight)[:, self.r0:self.r1]  # [K, N] view of W^T, ld = N_total

    def dx_gemm(self, dz, dx, M, **epi):
        """dx[M, K] = epilogue(dz[M, N] W): through the transposed mirror as a K-major x K-major GEMM, or on W itself read
        reduction-major"""
        if self.mirror_ok():
            hip.gemm(dz, self.lpT(), dx, M, self.K, self.N, self.N, self.Ntot, self.K, **epi)
        else:
            hip.gemm(dz, self.lp(), dx, M, self.K, self.N, self.N, self.K, self.K, layoutB=hip.TRANS, **epi)

    def bias_f32(self):
        return None if self.bias is None else self.bias.detach()[self.r0:self.r1]

    def wgrad(self):
        g = ensure_grad(self.weight)
        return g.reshape(self.Ntot, self.K)[self.r0:self.r1]

    def bgrad(self):
        return ensure_grad(self.bias)[self.r0:self.r1]


def _split_k(out_rows, out_cols, red):
    """split of the reduction for wgrad GEMMs.  The splits are summed through a workspace (two-pass, no atomics on the
    gradient), so the split only trades parallelism against 8 bytes of scratch traffic per output and split:
    ~768 blocks of 64x64, never more than 8 splits, never fewer than 16 k-tiles per split
    (tools/bench_lin_wgrad.py sweep); split 1 is a read-modify-write epilogue."""
    tiles = ((out_rows + 63) // 64) * ((out_cols + 63) // 64)
    kt = (red + 31) // 32
    return max(1, min(-(-768 // max(tiles, 1)), 8, kt // 16))


def _c(x):
    return x if x.is_contiguous() else x.contiguous()


def _as_compute(x):
    return x if x.dtype == RT.dtype else x.to(RT.dtype)


# --------------------------------------------------------------------------------------------
# Gradient chains: the gradient of an activation with several consumers, summed inside the consumers' own kernels
# --------------------------------------------------------------------------------------------
class GradChain:
    """A layer input feeds the residual of its LayerNorm AND the projection GEMMs (encoder: src -> norm1, Wqk (through src+pos),
    Wv); autograd sums the three gradients with two extra element-wise passes per layer (45 such launches per step, ~5 us each,
    all on the serial chain).  Consumers that share a GradChain hand the running sum along instead: the first one to run in the
    backward deposits its gradient, every later one adds it in the `res` epilogue of its backward-data GEMM, and only the last one
    returns a gradient to autograd (the others return None).  Consumers register while the forward is recorded, so a consumer
    that will never run a backward (no grad path) is not waited for."""
    ENABLED = os.environ.get('GPV_GRAD_CHAIN', '1') != '0'

    _live = []                    # chains created since the last check_chains() (weak references)

    def __init__(self):
        self.total = 0            # consumers registered in the forward
        self.left = 0             # ... that have not run yet in the current backward pass
        self.acc = None
        GradChain._live.append(weakref.ref(self))

    def join(self):
        self.total += 1
        self.left += 1
        return self

    def done(self, g):
        """g = this consumer's gradient with self.acc already added; returns the total if this was the last consumer.
        Re-arms itself: the same autograd graph may be walked again (retain_graph -- train.GraphedBody captures one backward per
        set of criterion outputs from a single recorded forward)."""
        self.left -= 1
        if self.left == 0:
            self.left = self.total
            self.acc = None
            return g
        self.acc = g
        return None

    def rearm(self):
        self.left, self.acc = self.total, None


def grad_chain(x):
    return GradChain() if (GradChain.ENABLED and torch.is_grad_enabled() and torch.is_tensor(x) and x.requires_grad) else None


def check_chains(clear=True, chains=None):
    """after a backward pass: every chain must be complete (all of its consumers ran, or none).  A chain whose consumers can
    run independently of each other (two outputs of a layer, one without a loss) silently loses gradient -- the members of a
    chain must sit behind ONE output (see vilbert.BertConnectionLayer).  Raises instead of training on wrong gradients."""
    bad = 0
    for r in (GradChain._live if chains is None else chains):      # chains: the list a train.GraphedBody owns for its forward
        c = r()
        if c is None:
            continue
        # a GradSlots buffer still held after the pass: its consumer (MultiLinearFn.backward) never ran, the columns written
        # into it reached nobody
        if c.left != c.total or (isinstance(c, GradSlots) and c.acc is not None):
            bad += 1
            c.rearm()
    if clear and chains is None:
        del GradChain._live[:]
    if bad:
        raise RuntimeError('%d GradChain(s) ended a backward pass half walked: a chained consumer did not run' % bad)


def reset_chains(chains=None):
    """before a backward pass over a recorded forward that may have been walked before (retain_graph: train.GraphedBody captures
    one backward per criterion variant, and retries a capture that failed): a pass that ABORTED half way leaves `acc` / `left` /
    `fresh` of its chains mid-walk, and the next pass over the same graph would add a stale partial sum, hand out a half-filled
    buffer or return None for a gradient.  Chains carry no state between complete passes, so re-arming is free."""
    for r in (GradChain._live if chains is None else chains):
        c = r()
        if c is not None:
            c.rearm()


class GradSink(GradChain):
    """The gradient of ONE wide buffer whose column slices are consumed by several attention calls (the DETR decoder's cross-
    attention keys / values of all layers, projected by one GEMM: multi_linear): every consumer's backward writes its columns
    straight into the shared gradient buffer (`acc`), the last one hands the complete buffer to autograd, the others return None --
    instead of one zero-filled full-width gradient per slice summed by autograd.  Registered with the chains: check_chains()
    raises if a backward pass left it half written."""

    def __init__(self, n):
        super().__init__()
        self.total = self.left = n
        self.events = []

    def slot(self, like):
        if self.acc is None:
            self.acc = torch.empty_like(like)
        return self.acc

    def note_writer(self):
        """a backward has just launched its writes into the buffer on the current stream: the buffer's consumer may run on another
        stream (ops.Branch) and autograd orders it only behind the writer that HANDED the buffer over"""
        if Branch.ENABLED and self.acc is not None and self.acc.is_cuda:
            ev = torch.cuda.Event()
            ev.record()
            self.events.append(ev)

    def sync_writers(self):
        """the consumer's stream waits for every writer (no-ops for writers of the same stream)"""
        evs, self.events = self.events, []
        if evs:
            cur = torch.cuda.current_stream()
            for ev in evs:
                cur.wait_event(ev)

    def wrote(self):
        self.left -= 1
        if self.left == 0:
            g, self.acc, self.left = self.acc, None, self.total
            return g
        return None


class GradSlots(GradSink):
    """GradSink for consumers that may NOT all run in a backward pass (the co-attention layer's two attention calls: a
    detection-only batch gives the language stream of the last layer no gradient): the first writer allocates the buffer
    zero-filled and hands it to autograd at once, later writers fill their columns in place (same stream, before the buffer's
    consumer runs) and return None; the consumer (MultiLinearFn.backward) releases it."""

    def __init__(self):
        super().__init__(0)
        self.fresh = False

    def slot(self, like):
        if self.acc is None:
            self.acc = torch.zeros_like(like)
            self.fresh = True
        return self.acc

    def wrote(self):
        if self.fresh:
            self.fresh = False
            return self.acc
        return None

    def release(self):
        self.sync_writers()
        self.acc, self.fresh = None, False

    def rearm(self):
        self.left, self.acc, self.fresh, self.events = self.total, None, False, []


class Branch:
    """Round 6: a side stream for work that is independent of the main chain between a fork and a join -- the co-attention layer's
    LANGUAGE stream (192 rows at B = 32: eight launches of 5 - 13 us per layer that used to sit in line with the vision stream's,
    vilbert.BertConnectionLayer).  Inside a capture the side stream's launches become a parallel branch of the graph; autograd runs the
    backward of what was recorded on the side stream there too and orders every gradient edge between the streams itself.  What does
    NOT travel on an autograd edge is ordered here: a GradSlots buffer that a main-stream attention backward fills in place after it
    was handed over (GradSink.note_writer / sync_writers).  The deferred weight gradients are launched behind a boundary node whose
    own inputs come after the last side-stream backward (the language stream's gradient returns to the main stream through
    bert_joiner's backward), so their operands are complete.  GPV_COATT_BRANCH=0: in line."""
    ENABLED = os.environ.get('GPV_COATT_BRANCH', '1') != '0'
    _streams = {}                 # device -> the side stream of EAGER launches (never part of a capture)

    # Inside a capture the side stream is the capture owner's (RT.branch_stream: train.GraphedBody / GPV._graphed make one per body /
    # per inference graph -- owned_stream(): a hipStream of the owner's own -- and destroy it after the owner's graphs).  What went wrong
    # with anything less (a process-wide stream; then torch's pooled streams per owner): a segmentation fault in a later replay /
    # capture_end after 8 - 16 evictions, from three causes -- owned_stream(), _dummy() and train.FlatTrainer._grad_accs say which
    # (tools/soak_evict.py is the reproducer and the test).
    def __init__(self, device, stream):
        self.dev = device
        self.side = stream

    def fork(self):
        self.side.wait_stream(torch.cuda.current_stream(self.dev))

    def join(self):
        torch.cuda.current_stream(self.dev).wait_stream(self.side)

    def on(self):
        return torch.cuda.stream(self.side)


    @staticmethod
    def join_captured(device):
        """before a capture ends: if the side stream is part of it, the capturing stream waits for it once more (every fork above is
        joined where its results are needed, and autograd joins what it moves between the streams -- this closes whatever a backward
        pass may have left on the side stream behind its last gradient edge: a capture must not end with unjoined work on a forked
        stream)"""
        st = RT.branch_stream
        if st is None:
            return
        with torch.cuda.stream(st):
            cap = torch.cuda.is_current_stream_capturing()
        if cap:
            torch.cuda.current_stream(device).wait_stream(st)


def report_unpinned_leaves(roots, pinned, model, where):
    """debugging aid (GPV_DEBUG_STREAMS=1): the AccumulateGrad nodes reachable from `roots` that are not among `pinned` (the trainer's, made
    on its own stream) -- each is created lazily on the stream of its first use and shared by every graph recorded while it lives; this
    walk found the process-wide dummy leaf of linear() (see _dummy)"""
    pinned_ids = {id(a) for a in pinned}
    names = {id(p): n for n, p in model.named_parameters()}
    seen, todo, odd = {}, [t.grad_fn for t in roots if t.grad_fn is not None], []
    while todo:
        nd = todo.pop()
        if nd is None or id(nd) in seen:
            continue
        seen[id(nd)] = nd                      # (kept alive: the wrappers' ids are recycled otherwise)
        if nd.name().endswith('AccumulateGrad') and id(nd) not in pinned_ids:
            odd.append((names.get(id(nd.variable), 'not a parameter'), tuple(nd.variable.shape)))
        todo.extend(f for f, _ in nd.next_functions)
    print('[gpv debug] %s: %d nodes, AccumulateGrad nodes that are not pinned: %s' % (where, len(seen), odd[:16]), flush=True)


def still_capturing(own, where):
    """debugging aid (GPV_DEBUG_STREAMS=1): right AFTER a capture ended, none of the owner's side streams may still be in capture mode"""
    if not DEBUG_STREAMS:
        return
    for name, st in own:
        if st is None:
            continue
        with torch.cuda.stream(st):
            cap = torch.cuda.is_current_stream_capturing()
        if cap:
            print('[gpv debug] %s: own stream %s is STILL capturing after capture_end' % (where, name), flush=True)


FRESH_STREAMS = os.environ.get('GPV_FRESH_STREAMS', '1') != '0'
DEBUG_STREAMS = os.environ.get('GPV_DEBUG_STREAMS', '0') == '1'      # debugging: owned streams are never destroyed, foreign_capturing() reports
_ALL_OWNED = []
_LEAKED = []                    # (handle, hipError) of side streams the runtime refused to destroy (release_stream)


def foreign_capturing(own, where):
    """debugging aid (GPV_DEBUG_STREAMS=1): which owned streams that do NOT belong to the capturing owner are in capture mode right now --
    something forked them into this capture (an autograd node that remembers a stream of another body, a stale event wait)"""
    if not DEBUG_STREAMS:
        return
    mine = {getattr(st, '_gpv_handle', None) for st in own if st is not None}
    for i, st in enumerate(_ALL_OWNED):
        if st._gpv_handle in mine:
            continue
        with torch.cuda.stream(st):
            cap = torch.cuda.is_current_stream_capturing()
        if cap:
            print('[gpv debug] %s: owned stream #%d (role %s of an OTHER owner) is capturing' % (where, i, ('side', 'wside', 'bside')[i % 3]), flush=True)
_hiprt = None


def owned_stream(device):
    """A side stream that is its owner's alone: a hipStream created here (hipStreamCreateWithFlags, non-blocking) and wrapped as
    torch.cuda.ExternalStream; release_stream() destroys it once the graphs captured on it are gone.
    torch.cuda.Stream() does NOT create a stream: it hands out one of a fixed pool of 32 hipStreams per device, round-robin.  A capture owner
    that takes three of them (BERT, weight and ops.Branch side streams) recycles the pool after ten owners -- and a hipStream that had been
    part of the capture of a graph destroyed since is what made later capture_end / replay calls segfault on ROCm 7.2 (round 6: first with one
    process-wide branch stream, then -- tools/soak_evict.py: 2 graph slots, an eviction on every miss -- with "per-owner" streams from the
    pool after 12 - 16 evictions).  GPV_FRESH_STREAMS=0: streams from torch's pool, as before."""
    global _hiprt
    device = torch.device(device)
    if not FRESH_STREAMS:
        return torch.cuda.Stream(device=device)
    import ctypes
    if _hiprt is None:
        _hiprt = ctypes.CDLL('libamdhip64.so')
        _hiprt.hipStreamCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
        _hiprt.hipStreamDestroy.argtypes = [ctypes.c_void_p]
    h = ctypes.c_void_p()
    with torch.cuda.device(device):
        err = _hiprt.hipStreamCreateWithFlags(ctypes.byref(h), 1)          # hipStreamNonBlocking
    if err != 0 or not h.value:
        raise RuntimeError('hipStreamCreateWithFlags failed: hipError %d' % err)
    st = torch.cuda.ExternalStream(h.value, device=device)
    st._gpv_handle = h.value
    if DEBUG_STREAMS:
        _ALL_OWNED.append(st)
    return st


def release_stream(st):
    """destroy a stream made by owned_stream (after its owner's graphs were destroyed and the device is idle); pool streams: nothing to do"""
    h = getattr(st, '_gpv_handle', None) if st is not None else None
    if DEBUG_STREAMS:
        return                              # (kept: foreign_capturing() asks them)
    if h and _hiprt is not None:
        st._gpv_handle = None
        err = _hiprt.hipStreamDestroy(ctypes_void_p(h))
        if err != 0:
            # The runtime refused (seen once in ~300 destroyed bodies: hipErrorStreamCaptureUnsupported -- it still counted the stream
            # as part of a capture).  Its last-error word is sticky per thread and torch reads it behind its NEXT launch: unchecked,
            # the refusal surfaced as 'operation not permitted when stream is capturing' from an innocent tensor.clone().  Clear it and
            # keep the stream (a leaked handle, never used again) instead of breaking the step.
            _hiprt.hipGetLastError()
            _LEAKED.append((h, err))
            if len(_LEAKED) in (1, 10, 100):
                import sys
                print('[gpv1_amd] hipStreamDestroy refused a side stream (hipError %d); kept alive, %d so far' % (err, len(_LEAKED)), file=sys.stderr, flush=True)


_PENDING = []                   # (graphs, streams) of capture owners finalised while a capture was in progress: release_pending()


def retire(graphs, streams):
    """a capture owner is done: destroy its graphs, then -- the device idle -- the streams of its own.  If a capture is in progress on this
    thread right now (the cyclic collector may finalise an owner at any allocation), neither a synchronize nor a stream destroy is legal:
    the resources are parked and release_pending() retires them at the next point where the caller has just synchronised anyway"""
    if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        _PENDING.append((graphs, streams))
        return
    del graphs[:]
    if any(getattr(st, '_gpv_handle', None) for st in streams if st is not None):
        torch.cuda.synchronize()
        for st in streams:
            release_stream(st)


def release_pending():
    """call where no capture is open and the device has just been synchronised (train.GraphedBody before its captures)"""
    while _PENDING:
        graphs, streams = _PENDING.pop()
        retire(graphs, streams)


def ctypes_void_p(v):
    import ctypes
    return ctypes.c_void_p(v)


def branch_wait(waiter):
    """`waiter` (a stream about to launch work whose operands a backward on the branch stream may have written -- the deferred weight
    gradients on train.GraphedBody's weight branch) waits for the capture owner's branch stream, if that stream is part of the capture:
    a side-stream backward that ends in frozen inputs (the target-embedding transform) hands no gradient back to the main stream, so no
    autograd edge orders it before the main stream's next launches"""
    st = RT.branch_stream
    if st is None or st is waiter:
        return
    with torch.cuda.stream(st):
        cap = torch.cuda.is_current_stream_capturing()
    if cap:
        waiter.wait_stream(st)


def branch_for(x):
    if not (Branch.ENABLED and torch.is_tensor(x) and x.is_cuda):
        return None
    if torch.cuda.is_current_stream_capturing():
        st = RT.branch_stream                  # (a capture whose owner lends no side stream runs in line)
        return Branch(x.device, st) if st is not None else None
    st = Branch._streams.get(x.device)
    if st is None:
        st = Branch._streams[x.device] = torch.cuda.Stream(device=x.device)
    return Branch(x.device, st)


# concatenated compute copies of the weights behind a multi_linear site: [n*N, K] (forward), [K, n*N] (backward-data), fp32 biases.
# They change with every optimizer step; built once per weights epoch -- inline on first use, or, under train.GraphedBody, for all
# registered sites on the weight branch of F2 (refresh_multi) beside the transformer's chain, where three concatenation launches per
# site cost nothing (in line they were 29 launches / 190 us per step)
_MULTI = {}


def _multi_key(ws):
    return ('multi', tuple((id(w.weight), w.r0, w.r1) for w in ws), RT.dtype)


def _multi_build(ws, old=None):
    """old: the site's previous (wcat, bcat, wT) -- rewritten in place (stable addresses: captured graphs of several bodies and
    backward variants read them, like the transposed mirrors of _lpT)"""
    o = old if old is not None else (None, None, None)
    wcat = torch.cat([w.lp() for w in ws], 0, out=o[0])
    bcat = torch.cat([w.bias_f32() for w in ws], 0, out=o[1]) if ws[0].bias is not None else None
    wT = torch.cat([w.lpT() for w in ws], 1, out=o[2]) if all(w.mirror_ok() for w in ws) else None
    return wcat, bcat, wT


def _multi_get(ws):
    key = _multi_key(ws)
    ep = max(RT.epoch_of(w.weight) for w in ws)
    hit = RT.cache.get(key)
    if hit is not None and hit[0] == ep and all(r() is w.weight for r, w in zip(hit[4], ws)):
        return hit[1:4]
    same = hit is not None and all(r() is w.weight for r, w in zip(hit[4], ws))
    built = _multi_build(ws, hit[1:4] if same else None)
    RT.cache[key] = (ep,) + built + (tuple(weakref.ref(w.weight) for w in ws),)
    _MULTI[key] = tuple(ws)
    return built


def refresh_multi():
    """rebuild the concatenated copies of every known multi_linear site now (train.GraphedBody: captured on the weight branch of F2,
    after refresh_transposed on the same stream -- the [K, n*N] copies are cut from the transposed mirrors)"""
    for key, ws in list(_MULTI.items()):
        hit = RT.cache.get(key)
        if hit is None or key[2] != RT.dtype or any(r() is not w.weight for r, w in zip(hit[4], ws)):
            _MULTI.pop(key, None)
            continue
        ep = max(RT.epoch_of(w.weight) for w in ws)
        RT.cache[key] = (ep,) + _multi_build(ws, hit[1:4]) + (hit[4],)


class MultiLinearFn(Function):
    """y[M, n*N] = x [M, K] . [W_0; W_1; ...]^T + [b_0; b_1; ...]: n Linear layers of equal width on the SAME input as one GEMM over
    the concatenated weights (the six DETR decoder layers' cross-attention key -- or value -- projections of the encoder memory,
    transformer.py:221-225: memory is layer-invariant).  Backward: one GEMM dx = dy . Wcat (K = n*N) instead of n chained ones,
    weight gradients per layer from the column slices of dy (row pitch n*N)."""

    @staticmethod
    def forward(ctx, x, ws, sink=None, chain=None):
        ctx.sink = sink
        ctx.chain = chain.join() if (chain is not None and ctx.needs_input_grad[0]) else None
        K, N, n = ws[0].K, ws[0].N, len(ws)
        x2 = _c(_as_compute(x)).reshape(-1, K)
        M = x2.shape[0]
        if RT.multi_wait is not None:
            RT.multi_wait()                                                    # (the branch that rebuilt the copies: joined at first use)
        wcat, bcat, _ = _multi_get(ws)                                         # [n*N, K] compute dtype, fp32 biases
        y = torch.empty(M, n * N, device=x.device, dtype=RT.dtype)
        hip.gemm(x2, wcat, y, M, n * N, K, K, K, n * N, bias=bcat)
        ctx.ws, ctx.xshape = ws, x.shape
        ctx.save_for_backward(x2)
        return y

    @staticmethod
    def backward(ctx, dy):
        ws = ctx.ws
        (x2,) = ctx.saved_tensors
        K, N, n = ws[0].K, ws[0].N, len(ws)
        M = x2.shape[0]
        dz = _c(_as_compute(dy)).reshape(M, n * N)
        if ctx.sink is not None:
            ctx.sink.release()
        for i, w in enumerate(ws):
            need_b = w.bias is not None and w.bias.requires_grad
            if w.weight.requires_grad:
                wgrad_linear(dz[:, i * N:(i + 1) * N], x2, w.wgrad(), w.bgrad() if need_b else None, N, K, M, _split_k(N, K, M), lda=n * N)
            elif need_b:
                hip.colsum(dz[:, i * N:(i + 1) * N], w.bgrad(), M, N, n * N)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(M, K, device=dz.device, dtype=RT.dtype)
            wcat, _, wT = _multi_get(ws)
            ch = ctx.chain
            epi = {'res': ch.acc.reshape(M, K), 'ldr': K} if (ch is not None and ch.acc is not None) else {}
            if wT is not None:                                                 # [K, n*N]: K-major x K-major on the pipelined kernels
                hip.gemm(dz, wT, dx, M, K, n * N, n * N, n * N, K, **epi)
            else:
                hip.gemm(dz, wcat, dx, M, K, n * N, n * N, K, K, layoutB=hip.TRANS, **epi)
            dx = dx.reshape(ctx.xshape)
            if ch is not None:
                dx = ch.done(dx)
        return dx, None, None, None


def multi_linear(x, ws, sink=None, chain=None):
    """sink: the ops.GradSlots through which the consumers deliver this output's gradient (released by the backward);
    chain: the ops.GradChain of x (LinearFn's: the running gradient of x rides in the backward-data GEMM's `res`)"""
    dummy_needed = torch.is_grad_enabled() and not x.requires_grad and any(w.weight.requires_grad for w in ws)
    if dummy_needed:                     # (no caller needs it: these inputs always carry a gradient when the weights train)
        raise RuntimeError('multi_linear: input without gradient but trainable weights')
    return MultiLinearFn.apply(x, tuple(ws), sink, chain)


# --------------------------------------------------------------------------------------------
# Linear:  y = dropout(act(x W^T + b))      (x: [..., K])
# --------------------------------------------------------------------------------------------
def _linear_backward(w, x2, dz, chain, need_dx, xshape):
    """backward of y = x2 w^T + b given dz = dL/dy [M, N] (contiguous, compute dtype): weight / bias gradients (deferred while a
    backward is captured), dx = dz W with the chain's running sum in the epilogue; -> the gradient to hand to autograd (None while
    other consumers of the chain are still to run)"""
    K, N = w.K, w.N
    M = x2.shape[0]
    need_b = w.bias is not None and w.bias.requires_grad
    if w.weight.requires_grad:
        # dW += dz^T x ; the bias gradient (column sums of dz) rides along in the same launch (a_rowsum)
        wgrad_linear(dz, x2, w.wgrad(), w.bgrad() if need_b else None, N, K, M, _split_k(N, K, M))
    elif need_b:
        hip.colsum(dz, w.bgrad(), M, N, N)
    dx = None
    if need_dx:
        dx = torch.empty(M, K, device=dz.device, dtype=RT.dtype)
        ch = chain
        if ch is not None and ch.acc is not None:
            w.dx_gemm(dz, dx, M, res=ch.acc.reshape(M, K), ldr=K)         # dx = dz W + (the other consumers' gradients so far)
        else:
            w.dx_gemm(dz, dx, M)                                           # dx = dz W
        dx = dx.reshape(xshape)
        if ch is not None:
            dx = ch.done(dx)
    return dx


class LinearFn(Function):
    @staticmethod
    def forward(ctx, x, w, act, drop_p, out_f32, dummy=None, chain=None):
        ctx.chain = chain.join() if (chain is not None and ctx.needs_input_grad[0]) else None
        K, N = w.K, w.N
        x2 = _c(_as_compute(x)).reshape(-1, K)
        M = x2.shape[0]
        out_dtype = torch.float32 if out_f32 else RT.dtype
        y = torch.empty(M, N, device=x.device, dtype=out_dtype)
        seed = RT.next_seed() if drop_p > 0 else 0
        z = None
        if act == ACT_GELU and any(ctx.needs_input_grad):   # keep the pre-activation for backward (inference / frozen BERT: GELU in the epilogue)
            z = torch.empty_like(y)
            hip.gemm(x2, w.lp(), z, M, N, K, K, K, N, bias=w.bias_f32())
            hip.act_fwd(z, y, M * N, ACT_GELU)
        else:
            hip.gemm(x2, w.lp(), y, M, N, K, K, K, N, bias=w.bias_f32(), act=act, drop_p=drop_p, seed=seed)
        ctx.w, ctx.act, ctx.drop_p, ctx.xshape, ctx.in_dtype = w, act, drop_p, x.shape, x.dtype
        ctx.save_for_backward(x2, y if act == ACT_RELU else None, z)
        return y.reshape(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        w, act = ctx.w, ctx.act
        x2, y, z = ctx.saved_tensors
        K, N = w.K, w.N
        M = x2.shape[0]
        dz = _c(_as_compute(dy)).reshape(M, N)
        if act == ACT_RELU:
            t = torch.empty_like(dz)
            hip.act_bwd(dz, _as_compute(y), t, M * N, ACT_RELU, 1.0 / (1.0 - ctx.drop_p) if ctx.drop_p > 0 else 1.0)
            dz = t
        elif act == ACT_GELU:
            t = torch.empty_like(dz)
            hip.act_bwd(dz, _as_compute(z), t, M * N, ACT_GELU, 1.0)
            dz = t
        dx = _linear_backward(w, x2, dz, ctx.chain, ctx.needs_input_grad[0], ctx.xshape)
        return dx, None, None, None, None, None, None


_DUMMY = {}


def _dummy(device):
    """1-element leaf that requires grad: forces autograd to call our backward (parameter gradients are
    accumulated by the kernels, not by autograd) when the activation input itself needs no gradient.
    ONE PER STREAM.  autograd visits the leaf's AccumulateGrad node in every backward that reaches it (with an undefined gradient) and
    ends the pass by recording an event on that node's stream -- the stream that was current when the node was CREATED, which happens
    lazily and lasts for as long as some recorded graph references the leaf.  One process-wide leaf, first used on a captured body's
    branch stream (round 6: the teacher-forcing prologue -- the target-embedding transform and the vocabulary classifiers read frozen
    inputs), tied every later body's backward to THAT body's stream: after its eviction an event record on a stream that belonged to
    somebody else or no longer existed (GraphTask::exec_post_processing -> hipErrorInvalidHandle with streams of their own, a
    segmentation fault in a later replay with torch's recycled pool streams; tools/soak_evict.py).  Keyed by the stream in use, the
    node always lives on the stream that is current -- inside a capture: part of it."""
    device = torch.device(device)
    key = (device, torch.cuda.current_stream(device).cuda_stream) if device.type == 'cuda' else (device, 0)
    d = _DUMMY.get(key)
    if d is None:
        # (no allocation per key: a new key's first use may be inside a capture, whose allocations belong to the capture owner's private
        #  pool -- every leaf aliases ONE element allocated on first use, which the eager warm-up step in front of every capture makes)
        base = _DUMMY.get(device)
        if base is None:
            base = _DUMMY[device] = torch.zeros(1, device=device)
        d = _DUMMY[key] = base.detach().requires_grad_(True)
    return d


def linear(x, w, act=ACT_NONE, drop_p=0.0, out_f32=False, chain=None):
    assert not (drop_p > 0 and act != ACT_RELU), 'epilogue dropout is only differentiated through the ReLU form'
    dummy = None
    if torch.is_grad_enabled() and not x.requires_grad and (w.weight.requires_grad or (w.bias is not None and w.bias.requires_grad)):
        dummy = _dummy(x.device)
    return LinearFn.apply(x, w, act, drop_p, out_f32, dummy, chain)


# y = a @ b^T with both operands activations (answer head: h x Wc^T)
class MatmulNTFn(Function):
    @staticmethod
    def forward(ctx, a, b):
        M, K = a.shape
        N = b.shape[0]
        y = torch.empty(M, N, device=a.device, dtype=RT.dtype)
        hip.gemm(a, b, y, M, N, K, K, K, N)
        ctx.save_for_backward(a, b)
        return y

    @staticmethod
    def backward(ctx, dy):
        a, b = ctx.saved_tensors
        M, K = a.shape
        N = b.shape[0]
        dy = _c(_as_compute(dy))
        da = db = None
        if ctx.needs_input_grad[0]:
            da = torch.empty(M, K, device=a.device, dtype=RT.dtype)
            hip.gemm(dy, b, da, M, K, N, N, K, K, layoutB=hip.TRANS)
        if ctx.needs_input_grad[1]:
            db = torch.empty(N, K, device=a.device, dtype=RT.dtype)
            hip.gemm(dy, a, db, N, K, M, N, K, K, layoutA=hip.TRANS, layoutB=hip.TRANS)
        return da, db


def matmul_nt(a, b):
    return MatmulNTFn.apply(_c(_as_compute(a)), _c(_as_compute(b)))


# --------------------------------------------------------------------------------------------
# y = LayerNorm(x + dropout(s)) * gamma + beta
# --------------------------------------------------------------------------------------------
def _pos_rows(pos, cols):
    """pos as [pos_rows, cols] rows in the compute dtype (a constant of autograd: its gradient, if any, goes to `pos_param`)"""
    p2 = _c(_as_compute(pos.detach())).reshape(-1, cols)
    return p2


def _pos_sink(pos_param, dy2, rows, cols):
    """gradient of a learned, row-broadcast position term (DETR's query_embed, [Q, cols]): column sums of dy2 viewed as
    [rows / Q, Q * cols], accumulated straight into the parameter's fp32 gradient (parameter gradients never travel through
    autograd here) -- instead of one [rows, cols] gradient per use summed by autograd with an element-wise launch each"""
    if pos_param is None or not pos_param.requires_grad or dy2 is None:
        return
    n = pos_param.numel()
    g = ensure_grad(pos_param).reshape(-1)
    d = dy2.reshape(-1, n)
    # (nobody reads this gradient before the optimizer: with the deferred weight gradients, off the backward chain -- 12 launches a step)
    off_critical_path(lambda: hip.colsum(d, g, rows * cols // n, n, n), d, d, g)


def _two_grads(dy, dy2, rows, cols):
    """(dy, dy2) as the kernels take them: the first one must exist"""
    a = None if dy is None else _c(_as_compute(dy)).reshape(rows, cols)
    b = None if dy2 is None else _c(_as_compute(dy2)).reshape(rows, cols)
    return (b, None) if a is None else (a, b)


LN_FOLD = os.environ.get('GPV_LN_FOLD', '1') != '0'     # 0: dgamma / dbeta by fp32 atomics inside every LayerNorm backward (A/B)


def _ln_backward(d1, d2, x2, s2, gamma, beta, mean, rstd, dx, ds, rows, cols, drop_p, seed):
    """gpv_layernorm_bwd3.  Nobody needs dgamma / dbeta before the optimizer: while train.GraphedBody captures a backward the
    launch leaves its per-workgroup partial column sums in a buffer (plain stores) and the sum into the parameters' gradients
    joins the deferred weight-gradient work (ONE grouped gpv_colsum_fold_group per flush, on the side branch) -- the same-address
    atomics of 160 workgroups were a third of the launch on the DETR shapes, on the backward's critical chain 52 times a step"""
    need_g = gamma is not None and gamma.requires_grad
    gd = None if gamma is None else gamma.detach()
    if need_g and LN_FOLD and RT.defer_list is not None:
        nblk = hip.layernorm_bwd_blocks(rows, cols)
        part = torch.empty(nblk, 2 * cols, device=dx.device, dtype=torch.float32)
        hip.layernorm_bwd(d1, x2, s2, gd, mean, rstd, dx, ds, None, None, rows, cols, drop_p, seed, dy2=d2, partials=part)
        dg, db = ensure_grad(gamma), ensure_grad(beta)
        prob = (part, dg, db, nblk, cols)
        RT.defer_list.append((lambda: hip.colsum_fold_group([prob]), (part, db, dg), ('fold',) + prob))
        return
    hip.layernorm_bwd(d1, x2, s2, gd, mean, rstd, dx, ds, ensure_grad(gamma) if need_g else None, ensure_grad(beta) if need_g else None,
                      rows, cols, drop_p, seed, dy2=d2)


class AddLayerNormFn(Function):
    """y = LayerNorm(x + dropout(s)); with `pos` a second output y2 = y + pos (rows broadcast): gpv_layernorm_pos_fwd"""

    @staticmethod
    def forward(ctx, x, s, gamma, beta, eps, drop_p, chain=None, pos=None, pos_param=None):
        ctx.chain = chain.join() if (chain is not None and ctx.needs_input_grad[0]) else None
        ctx.set_materialize_grads(False)
        cols = x.shape[-1]
        x2 = _c(_as_compute(x)).reshape(-1, cols)
        s2 = None if s is None else _c(_as_compute(s)).reshape(-1, cols)
        rows = x2.shape[0]
        y = torch.empty_like(x2)
        mean = torch.empty(rows, device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        seed = RT.next_seed() if (drop_p > 0 and s is not None) else 0
        if s is None:
            drop_p = 0.0
        p2 = None if pos is None else _pos_rows(pos, cols)
        y2 = None if pos is None else torch.empty_like(x2)
        hip.layernorm_fwd(x2, s2, None if gamma is None else gamma.detach(), None if beta is None else beta.detach(),
                          y, mean, rstd, rows, cols, eps, drop_p, seed, pos=p2, y2=y2)
        ctx.gamma, ctx.beta, ctx.drop_p, ctx.seed, ctx.shape = gamma, beta, drop_p, seed, x.shape
        ctx.has_s, ctx.pos_param = s is not None, pos_param
        ctx.save_for_backward(x2, s2, mean, rstd)
        if pos is None:
            return y.reshape(x.shape)
        return y.reshape(x.shape), y2.reshape(x.shape)

    @staticmethod
    def backward(ctx, dy, dy2=None):
        nin = 9
        if dy is None and dy2 is None:
            return (None,) * nin
        x2, s2, mean, rstd = ctx.saved_tensors
        rows, cols = x2.shape
        if dy2 is not None:
            _pos_sink(ctx.pos_param, _c(_as_compute(dy2)).reshape(rows, cols), rows, cols)
        d1, d2 = _two_grads(dy, dy2, rows, cols)
        dx = torch.empty_like(x2)
        ds = torch.empty_like(x2) if (ctx.has_s and ctx.drop_p > 0) else None
        _ln_backward(d1, d2, x2, s2, ctx.gamma, ctx.beta, mean, rstd, dx, ds, rows, cols, ctx.drop_p, ctx.seed)
        dxr = dx.reshape(ctx.shape)
        dsr = None
        if ctx.has_s:
            dsr = (ds if ds is not None else dx).reshape(ctx.shape)
        if ctx.chain is not None and ctx.needs_input_grad[0]:
            ch = ctx.chain
            if ch.acc is not None:                      # (a LayerNorm is normally the first consumer to run; if not: one add)
                t = torch.empty_like(dx)
                hip.add(dx, _c(ch.acc).reshape(rows, cols), t, t.numel())
                dxr = t.reshape(ctx.shape)
            elif dsr is not None and ds is None:
                dxr = dxr.clone()                       # dx doubles as ds here: the chain's later `res` reads must not alias a live output
            dxr = ch.done(dxr)
        return ((dxr if ctx.needs_input_grad[0] else None), (dsr if ctx.needs_input_grad[1] else None)) + (None,) * (nin - 2)


PROJ_LN = os.environ.get('GPV_PROJ_LN', '1') != '0'
PROJ_LN_MIN_ROWS = int(os.environ.get('GPV_PROJ_LN_MIN_ROWS', '2048'))


def proj_layernorm_ok(w, x):
    """gpv_linear_layernorm_fwd's range: a 256 -> 256 projection in front of a width-256 LayerNorm, bf16 (the DETR attention sublayers)"""
    # rows: the one launch stages the whole 128 KB weight per workgroup (8.7 us whatever the row count); as nodes of a hipGraph chain
    # GEMM + LayerNorm take 6.6 us up to ~1200 rows and 9.6 at 3200 against 9.3 (tools/bench_linear_ln_small.py): batch-1 inference
    # (300 / 100 rows) keeps the two launches, the training shapes (9600 / 3200 rows) the one
    return (PROJ_LN and RT.dtype == torch.bfloat16 and w.K == 256 and w.N == 256 and x.shape[-1] == 256 and w.bias is not None
            and x.numel() // 256 >= PROJ_LN_MIN_ROWS)


class ProjAddLayerNormFn(Function):
    """y = LayerNorm(x + dropout(a w^T + b)) (+ y2 = y + pos) with the projection INSIDE the LayerNorm launch
    (gpv_linear_layernorm_fwd; transformer.py:153-157, 218-226: out_proj of nn.MultiheadAttention, then norm).  The launch also
    writes s = a w^T + b where the GEMM would have: the backward is the one of the two launches it replaces -- AddLayerNormFn's
    (gpv_layernorm_bwd3, the residual's GradChain) followed by LinearFn's (weight / bias gradient deferred, backward-data GEMM)."""

    @staticmethod
    def forward(ctx, x, a, w, gamma, beta, eps, drop_p, chain=None, pos=None, pos_param=None):
        ctx.chain = chain.join() if (chain is not None and ctx.needs_input_grad[0]) else None
        ctx.set_materialize_grads(False)
        cols = x.shape[-1]
        x2 = _c(_as_compute(x)).reshape(-1, cols)
        a2 = _c(_as_compute(a)).reshape(-1, w.K)
        rows = x2.shape[0]
        s2 = torch.empty_like(x2)
        y = torch.empty_like(x2)
        mean = torch.empty(rows, device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        seed = RT.next_seed() if drop_p > 0 else 0
        p2 = None if pos is None else _pos_rows(pos, cols)
        y2 = None if pos is None else torch.empty_like(x2)
        hip.linear_layernorm_fwd(a2, w.lp(), w.bias_f32(), x2, None if gamma is None else gamma.detach(), None if beta is None else beta.detach(),
                                 s2, y, mean, rstd, rows, eps, drop_p, seed, pos=p2, y2=y2)
        ctx.gamma, ctx.beta, ctx.drop_p, ctx.seed, ctx.shape, ctx.ashape = gamma, beta, drop_p, seed, x.shape, a.shape
        ctx.w, ctx.pos_param = w, pos_param
        ctx.save_for_backward(x2, s2, mean, rstd, a2)
        if pos is None:
            return y.reshape(x.shape)
        return y.reshape(x.shape), y2.reshape(x.shape)

    @staticmethod
    def backward(ctx, dy, dy2=None):
        nin = 10
        if dy is None and dy2 is None:
            return (None,) * nin
        x2, s2, mean, rstd, a2 = ctx.saved_tensors
        rows, cols = x2.shape
        if dy2 is not None:
            _pos_sink(ctx.pos_param, _c(_as_compute(dy2)).reshape(rows, cols), rows, cols)
        d1, d2 = _two_grads(dy, dy2, rows, cols)
        dx = torch.empty_like(x2)
        ds = torch.empty_like(x2) if ctx.drop_p > 0 else None
        _ln_backward(d1, d2, x2, s2, ctx.gamma, ctx.beta, mean, rstd, dx, ds, rows, cols, ctx.drop_p, ctx.seed)
        dxr = dx.reshape(ctx.shape)
        if ctx.chain is not None and ctx.needs_input_grad[0]:
            ch = ctx.chain
            if ch.acc is not None:                      # (a LayerNorm is normally the first consumer to run; if not: one add)
                t = torch.empty_like(dx)
                hip.add(dx, _c(ch.acc).reshape(rows, cols), t, t.numel())
                dxr = t.reshape(ctx.shape)
            elif ds is None:
                dxr = dxr.clone()                       # dx doubles as ds here: the chain's later `res` reads must not alias the projection's dz
            dxr = ch.done(dxr)
        da = _linear_backward(ctx.w, a2, ds if ds is not None else dx, None, ctx.needs_input_grad[1], ctx.ashape)
        return ((dxr if ctx.needs_input_grad[0] else None), da) + (None,) * (nin - 2)


def proj_add_layernorm(x, a, w, gamma, beta, eps, drop_p=0.0, chain=None, pos=None, pos_param=None):
    return ProjAddLayerNormFn.apply(x, a, w, gamma, beta, eps, drop_p, chain, pos, pos_param)


# gpv_ffn_fused_fwd (csrc/ffn_fused.hip): the sub-layer as one launch.  Correct (tests/test_kernels_gpu.py) and NOT faster as built --
# 71 us against 71 us at M = 9600, 66 against 42 at M = 3200, step 17.6 - 17.8 against 17.4 - 17.6 ms -- so it is opt-in
FFN_FUSED = os.environ.get('GPV_FFN_FUSED', '0') == '1'


class FFNBlockFn(Function):
    """out = LayerNorm(x + dropout(W2 . dropout(relu(W1 x + b1)) + b2)): the whole post-norm feed-forward sub-layer
    (transformer.py:156-160, 226-231) as ONE autograd node.  x has two consumers (FFN input, residual); as separate nodes
    autograd sums their gradients with an extra elementwise pass -- here the residual gradient from the LayerNorm backward is
    the `res` operand of the last backward-data GEMM (dx = dz W1 + dx_res), and the ReLU/dropout derivative sits in the epilogue
    of dz = dy W2: alpha = 1/(1-p), ReLU mask = the saved hidden activation, which is zero exactly where the unit was
    dropped or inactive (no separate pass over the [M, 2048] gradient).  With `pos`: second output out + pos (AddLayerNormFn)."""

    @staticmethod
    def forward(ctx, x, w1, w2, gamma, beta, eps, drop_p, pos=None, pos_param=None):
        ctx.set_materialize_grads(False)
        K, Fh = w1.K, w1.N
        x2 = _c(_as_compute(x)).reshape(-1, K)
        M = x2.shape[0]
        h = torch.empty(M, Fh, device=x.device, dtype=RT.dtype)
        seed1 = RT.next_seed() if drop_p > 0 else 0
        y = torch.empty(M, K, device=x.device, dtype=RT.dtype)
        out = torch.empty_like(x2)
        mean = torch.empty(M, device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        seed2 = RT.next_seed() if drop_p > 0 else 0
        p2 = None if pos is None else _pos_rows(pos, K)
        out2 = None if pos is None else torch.empty_like(x2)
        # (opt-in: one launch when the kernel takes the shape -- the DETR layers: 256 -> 2048 -> 256, bf16; the same seeds either way)
        if not (FFN_FUSED and RT.dtype == torch.bfloat16 and w1.bias is not None and w2.bias is not None and
                hip.ffn_fused_fwd(x2, w1.lp(), w1.bias_f32(), w2.lp(), w2.bias_f32(), gamma.detach(), beta.detach(), h, y, out, mean, rstd,
                                  M, K, Fh, eps, drop_p, seed1, seed2, pos=p2, out2=out2)):
            hip.gemm(x2, w1.lp(), h, M, Fh, K, K, K, Fh, bias=w1.bias_f32(), act=ACT_RELU, drop_p=drop_p, seed=seed1)
            hip.gemm(h, w2.lp(), y, M, K, Fh, Fh, Fh, K, bias=w2.bias_f32())
            hip.layernorm_fwd(x2, y, gamma.detach(), beta.detach(), out, mean, rstd, M, K, eps, drop_p, seed2, pos=p2, y2=out2)
        ctx.w1, ctx.w2, ctx.gamma, ctx.beta, ctx.drop_p, ctx.seed2, ctx.xshape = w1, w2, gamma, beta, drop_p, seed2, x.shape
        ctx.pos_param = pos_param
        ctx.save_for_backward(x2, h, y, mean, rstd)
        if pos is None:
            return out.reshape(x.shape)
        return out.reshape(x.shape), out2.reshape(x.shape)

    @staticmethod
    def backward(ctx, dout, dout2=None):
        nin = 9
        if dout is None and dout2 is None:
            return (None,) * nin
        w1, w2, gamma, beta = ctx.w1, ctx.w2, ctx.gamma, ctx.beta
        x2, h, y, mean, rstd = ctx.saved_tensors
        M, K = x2.shape
        Fh = w1.N
        if dout2 is not None:
            _pos_sink(ctx.pos_param, _c(_as_compute(dout2)).reshape(M, K), M, K)
        d2, d2b = _two_grads(dout, dout2, M, K)
        dx_res = torch.empty_like(x2)
        ds = torch.empty_like(x2) if ctx.drop_p > 0 else None
        _ln_backward(d2, d2b, x2, y, gamma, beta, mean, rstd, dx_res, ds, M, K, ctx.drop_p, ctx.seed2)
        dy2 = ds if ds is not None else dx_res
        nb2 = w2.bias is not None and w2.bias.requires_grad
        if w2.weight.requires_grad:
            wgrad_linear(dy2, h, w2.wgrad(), w2.bgrad() if nb2 else None, K, Fh, M, _split_k(K, Fh, M))
        elif nb2:
            hip.colsum(dy2, w2.bgrad(), M, K, K)
        dz = torch.empty(M, Fh, device=d2.device, dtype=RT.dtype)
        w2.dx_gemm(dy2, dz, M, relu_mask=h, ldm=Fh, alpha=1.0 / (1.0 - ctx.drop_p) if ctx.drop_p > 0 else 1.0)
        nb1 = w1.bias is not None and w1.bias.requires_grad
        if w1.weight.requires_grad:
            wgrad_linear(dz, x2, w1.wgrad(), w1.bgrad() if nb1 else None, Fh, K, M, _split_k(Fh, K, M))
        elif nb1:
            hip.colsum(dz, w1.bgrad(), M, Fh, Fh)
        dx = torch.empty(M, K, device=dz.device, dtype=RT.dtype)
        w1.dx_gemm(dz, dx, M, res=dx_res, ldr=K)
        return (dx.reshape(ctx.xshape),) + (None,) * (nin - 1)


LN_POS = os.environ.get('GPV_LN_POS', '1') != '0'       # 0: TIMING A/B ONLY -- the sums as separate launches, query_embed loses the sink's gradient


def _unfused_pos(y, pos, pos_param=None):
    if pos_param is not None and pos_param.requires_grad and torch.is_grad_enabled():
        # (the separate add takes pos as a constant -- query_embed would silently train without this share of its gradient)
        raise RuntimeError('GPV_LN_POS=0 is a timing switch: it drops the gradient of the learned position term (query_embed); '
                           'not available while that parameter trains')
    cols = y.shape[-1]
    return y, add(y, _pos_rows(pos, cols))


def ffn_block(x, w1, w2, gamma, beta, eps, drop_p=0.0, pos=None, pos_param=None):
    """LayerNorm(x + dropout(ffn(x))) -- one node when gradients flow, the plain composition otherwise; with `pos` -> (out, out + pos)"""
    if pos is not None and not LN_POS:
        return _unfused_pos(ffn_block(x, w1, w2, gamma, beta, eps, drop_p), pos, pos_param)
    if not (torch.is_grad_enabled() and x.requires_grad):
        return add_layernorm(x, linear(linear(x, w1, ACT_RELU, drop_p), w2), gamma, beta, eps, drop_p, pos=pos, pos_param=pos_param)
    return FFNBlockFn.apply(x, w1, w2, gamma, beta, eps, drop_p, pos, pos_param)


def add_layernorm(x, s, gamma, beta, eps, drop_p=0.0, chain=None, pos=None, pos_param=None):
    """pos ([rows_p, cols], rows a multiple of rows_p; a constant for autograd): -> (y, y + pos); pos_param: the learned parameter
    behind a row-broadcast pos, whose gradient is accumulated by the backward (ops._pos_sink)"""
    if pos is not None and not LN_POS:
        return _unfused_pos(AddLayerNormFn.apply(x, s, gamma, beta, eps, drop_p, chain, None, None), pos, pos_param)
    return AddLayerNormFn.apply(x, s, gamma, beta, eps, drop_p, chain, pos, pos_param)


# --------------------------------------------------------------------------------------------
# attention core over (possibly fused) projection buffers
# --------------------------------------------------------------------------------------------
class AttentionFn(Function):
    """bufs: distinct [B*S, width] tensors; roles: ((buf_idx, col_off) for q, k, v).  Every column of
    every buf must be covered by exactly one role (true for qk|v, qkv, q|kv fusions)."""

    @staticmethod
    def forward(ctx, meta, *bufs):
        roles, B, H, Sq, Sk, dh, kpm, causal, drop_p = meta[:9]
        D = H * dh
        bufs = tuple(_c(_as_compute(b)) for b in bufs)
        (qi, qo), (ki, ko), (vi, vo) = roles
        qb, kb, vb = bufs[qi], bufs[ki], bufs[vi]
        o = torch.empty(B * Sq, D, device=qb.device, dtype=RT.dtype)
        lse = torch.empty(B, H, Sq, device=qb.device, dtype=torch.float32)
        st = ((Sq * qb.shape[1], qb.shape[1]), (Sk * kb.shape[1], kb.shape[1]), (Sk * vb.shape[1], vb.shape[1]), (Sq * D, D))
        seed = RT.next_seed() if drop_p > 0 else 0
        scale = 1.0 / math.sqrt(dh)
        hip.attention_fwd(qb[:, qo:], kb[:, ko:], vb[:, vo:], o, st, B, H, Sq, Sk, dh, scale, kpm=kpm, causal=causal,
                          drop_p=drop_p, seed=seed, lse=lse)
        ctx.meta, ctx.seed, ctx.st, ctx.scale = meta, seed, st, scale
        ctx.save_for_backward(o, lse, *bufs)
        return o

    @staticmethod
    def backward(ctx, do):
        roles, B, H, Sq, Sk, dh, kpm, causal, drop_p = ctx.meta[:9]
        sinks = ctx.meta[9] if len(ctx.meta) > 9 and ctx.meta[9] is not None else (None,) * (len(ctx.saved_tensors) - 2)
        D = H * dh
        o, lse, *bufs = ctx.saved_tensors
        (qi, qo), (ki, ko), (vi, vo) = roles
        do = _c(_as_compute(do))
        # a buffer with a GradSink is shared with other attention calls (column slices of one wide projection): its gradient
        # columns are written into the sink's buffer, which the last writer hands to autograd
        grads = [torch.empty_like(b) if s is None else s.slot(b) for b, s in zip(bufs, sinks)]
        hip.attention_bwd(bufs[qi][:, qo:], bufs[ki][:, ko:], bufs[vi][:, vo:], o, do,
                          grads[qi][:, qo:], grads[ki][:, ko:], grads[vi][:, vo:], ctx.st, (Sq * D, D),
                          B, H, Sq, Sk, dh, ctx.scale, kpm=kpm, causal=causal, drop_p=drop_p, seed=ctx.seed, lse=lse)
        for s_ in sinks:
            if s_ is not None:
                s_.note_writer()
        return (None, *[g if s is None else s.wrote() for g, s in zip(grads, sinks)])


def attention(bufs, roles, B, H, Sq, Sk, dh, kpm=None, causal=False, drop_p=0.0, sinks=None):
    """sinks: per buffer None or the ops.GradSink of a buffer shared with other attention calls (see AttentionFn.backward)"""
    return AttentionFn.apply((roles, B, H, Sq, Sk, dh, kpm, causal, drop_p, sinks), *bufs)


ATTN_QKV = os.environ.get('GPV_ATTN_QKV', '1') != '0'


def attention_qkv_ok(E, H, S, causal, B=None):
    """gpv_attention_qkv_fwd's range: the DETR encoder / decoder self-attention (width 256, 8 heads of 32, <= 320 tokens, bf16); one
    workgroup per (batch, head) multiplies all of its rows: below 64 of them (batch-1 inference: 8) the projection GEMMs, which
    spread over the chip, are faster (23 us against 21 at batch 1)"""
    return ATTN_QKV and RT.dtype == torch.bfloat16 and E == 256 and H == 8 and 0 < S <= 320 and not causal and (B is None or B * H >= 64)


class SelfAttnQKVFn(Function):
    """o = attention(xp Wq^T + bq, xp Wk^T + bk, x Wv^T + bv) with the projections INSIDE the attention launch
    (gpv_attention_qkv_fwd; transformer.py:148-155, 216-219).  The launch also writes q | k and v where the projection GEMMs
    would have: the backward is the one of the three launches it replaces -- gpv_attention_bwd, then the two projections'
    weight / bias gradients and backward-data GEMMs (value first, as autograd orders them) with their GradChains."""

    @staticmethod
    def forward(ctx, meta, xp, x, dummy=None):
        w, B, H, S, kpm, drop_p, chain_qk, chain_v = meta
        E = w.K
        xp2 = _c(_as_compute(xp)).reshape(-1, E)
        x2 = _c(_as_compute(x)).reshape(-1, E)
        M = xp2.shape[0]
        qk = torch.empty(M, 2 * E, device=x2.device, dtype=RT.dtype)
        v = torch.empty(M, E, device=x2.device, dtype=RT.dtype)
        o = torch.empty(M, E, device=x2.device, dtype=RT.dtype)
        lse = torch.empty(B, H, S, device=x2.device, dtype=torch.float32)
        st = ((S * 2 * E, 2 * E), (S * 2 * E, 2 * E), (S * E, E), (S * E, E))
        seed = RT.next_seed() if drop_p > 0 else 0
        scale = 1.0 / math.sqrt(E // H)
        hip.attention_qkv_fwd(xp2, x2, w.lp(), w.bias_f32(), qk[:, :E], qk[:, E:], v, o, st, B, H, S, scale, kpm=kpm, drop_p=drop_p,
                              seed=seed, lse=lse)
        ctx.cq = chain_qk.join() if (chain_qk is not None and ctx.needs_input_grad[1]) else None
        ctx.cv = chain_v.join() if (chain_v is not None and ctx.needs_input_grad[2]) else None
        ctx.meta, ctx.seed, ctx.st, ctx.scale = meta, seed, st, scale
        ctx.xp_shape, ctx.x_shape = xp.shape, x.shape
        ctx.save_for_backward(xp2, x2, qk, v, o, lse)
        return o

    @staticmethod
    def backward(ctx, do):
        w, B, H, S, kpm, drop_p = ctx.meta[:6]
        E = w.K
        xp2, x2, qk, v, o, lse = ctx.saved_tensors
        do = _c(_as_compute(do))
        dqk, dv = torch.empty_like(qk), torch.empty_like(v)
        hip.attention_bwd(qk[:, :E], qk[:, E:], v, o, do, dqk[:, :E], dqk[:, E:], dv, ctx.st, (S * E, E), B, H, S, S, E // H, ctx.scale,
                          kpm=kpm, drop_p=drop_p, seed=ctx.seed, lse=lse)
        dx = _linear_backward(W(w.weight, w.bias, 2 * E, 3 * E), x2, dv, ctx.cv, ctx.needs_input_grad[2], ctx.x_shape)
        dxp = _linear_backward(W(w.weight, w.bias, 0, 2 * E), xp2, dqk, ctx.cq, ctx.needs_input_grad[1], ctx.xp_shape)
        return None, dxp, dx, None


def attention_qkv(xp, x, w, B, H, S, kpm=None, drop_p=0.0, chains=(None, None)):
    """w: ops.W over ALL 3 E rows of in_proj_weight (+ bias); chains: the GradChains of xp (q | k projection) and x (value projection)"""
    dummy = None
    if torch.is_grad_enabled() and not (xp.requires_grad or x.requires_grad) and (w.weight.requires_grad or (w.bias is not None and w.bias.requires_grad)):
        dummy = _dummy(x.device)
    return SelfAttnQKVFn.apply((w, B, H, S, kpm, drop_p, chains[0], chains[1]), xp, x, dummy)


# --------------------------------------------------------------------------------------------
# small element-wise autograd ops
# --------------------------------------------------------------------------------------------
class AddFn(Function):
    """y = a + b (same shape) or a + b broadcast over leading rows (b: [rows_b, cols])"""

    @staticmethod
    def forward(ctx, a, b):
        ctx.set_materialize_grads(False)                # a chained consumer may hand back None: nothing to pass on
        a2, b2 = _c(_as_compute(a)), _c(_as_compute(b))
        y = torch.empty_like(a2)
        cols = a2.shape[-1]
        ctx.bshape, ctx.ashape = b.shape, a.shape
        if a2.numel() == b2.numel():
            hip.add(a2, b2, y, a2.numel())
        else:
            hip.add_rowbcast(a2, b2, y, a2.numel() // cols, b2.numel() // cols, cols)
        return y

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return None, None
        db = None
        if ctx.needs_input_grad[1]:
            if math.prod(ctx.bshape) == dy.numel():
                db = dy.reshape(ctx.bshape)
            else:                                           # broadcast rows: sum over the leading repeats
                cols = ctx.bshape[-1]
                rows_b = math.prod(ctx.bshape) // cols
                acc = torch.zeros(rows_b * cols, device=dy.device, dtype=torch.float32)
                d2 = _c(dy).reshape(-1, rows_b * cols)
                hip.colsum(d2, acc, d2.shape[0], rows_b * cols, rows_b * cols)
                db = acc.to(dy.dtype).reshape(ctx.bshape)
        return (dy if ctx.needs_input_grad[0] else None), db


def add(a, b):
    return AddFn.apply(a, b)


class EmbeddingFn(Function):
    """frozen-table gather (nn.Embedding.from_pretrained(freeze=True), gpv.py:50-51; BERT embeddings)"""

    @staticmethod
    def forward(ctx, table, ids):
        ids = _c(ids)
        dim = table.shape[1]
        out = torch.empty(ids.numel(), dim, device=ids.device, dtype=RT.dtype)
        hip.embedding(table.detach(), ids.reshape(-1), out, ids.numel(), dim)
        return out.reshape(*ids.shape, dim)

    @staticmethod
    def backward(ctx, dy):
        return None, None


def embedding(table, ids):
    return EmbeddingFn.apply(table, ids)


class RelevanceConditionFn(Function):
    """gpv.py:364-375: y = x + softmax(logits) @ tokens   x:[rows,D] logits:[rows,2] fp32 tokens:[2,D] fp32 param"""

    @staticmethod
    def forward(ctx, x, logits, tokens):
        x2 = _c(_as_compute(x))
        lg = _c(logits.float())
        y = torch.empty_like(x2)
        hip.relevance_condition(x2, lg, tokens.detach(), y, x2.shape[0], x2.shape[1])
        ctx.tokens = tokens
        ctx.save_for_backward(lg)
        return y

    @staticmethod
    def backward(ctx, dy):
        (lg,) = ctx.saved_tensors
        tokens = ctx.tokens
        rows, D = dy.shape
        dyc = _c(_as_compute(dy))
        p = lg.softmax(-1)                                             # [rows,2] fp32 (tiny)
        # dp[r,j] = dy[r] . tok[j]   ;  dtok[j] += sum_r p[r,j] dy[r]
        tok_lp = _as_compute(tokens.detach())
        dp = torch.empty(rows, 2, device=dy.device, dtype=torch.float32)
        hip.gemm(dyc, tok_lp, dp, rows, 2, D, D, D, 2)
        if tokens.requires_grad:
            hip.gemm(_as_compute(p), dyc, ensure_grad(tokens), 2, D, rows, 2, D, D, layoutA=hip.TRANS, layoutB=hip.TRANS,
                     accumulate=True, split_k=_split_k(2, D, rows))
        dlg = p * (dp - (p * dp).sum(-1, keepdim=True))
        return (dy if ctx.needs_input_grad[0] else None), (dlg if ctx.needs_input_grad[1] else None), None


def relevance_condition(x, logits, tokens):
    return RelevanceConditionFn.apply(x, logits, tokens)


class SoftmaxCEFn(Function):
    """per-row cross entropy (reduction none), logits [rows, V]"""

    @staticmethod
    def forward(ctx, logits, target):
        lg = _c(logits)
        rows, V = lg.shape
        loss = torch.empty(rows, device=lg.device, dtype=torch.float32)
        hip.softmax_ce(lg, V, _c(target), loss, None, None, rows, V)
        ctx.save_for_backward(lg, target)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        lg, target = ctx.saved_tensors
        rows, V = lg.shape
        dl = torch.empty_like(lg)
        scratch = torch.empty(rows, device=lg.device, dtype=torch.float32)
        hip.softmax_ce(lg, V, _c(target), scratch, dl, _c(dloss.float()), rows, V)
        return dl, None


def softmax_ce(logits, target):
    return SoftmaxCEFn.apply(logits, target)


# --------------------------------------------------------------------------------------------
# RoIAlign(7x7, aligned) + mean over bins, separable form (detr_roi_head.py:44-56)
# --------------------------------------------------------------------------------------------
class RoiPoolFn(Function):
    """feat [B, H*W, C] (NHWC rows), boxes [B, Q, 4] fp32 n-cxcywh (no gradient to boxes: torchvision
    roi_align has none) -> pooled [B, Q, C]"""

    @staticmethod
    def forward(ctx, feat, boxes, H, Wd):
        B, Pn, Cc = feat.shape
        Q = boxes.shape[1]
        ldw = ((Pn + 31) // 32) * 32
        wgt = torch.empty(B * Q, ldw, device=feat.device, dtype=RT.dtype)
        hip.roi_weights(_c(boxes.detach().float().reshape(-1, 4)), wgt, B * Q, H, Wd, ldw)
        f = _c(_as_compute(feat))
        out = torch.empty(B, Q, Cc, device=feat.device, dtype=RT.dtype)
        hip.gemm(wgt, f, out, Q, Cc, Pn, ldw, Cc, Cc, layoutB=hip.TRANS, batch=B, sA=Q * ldw, sB=Pn * Cc, sC=Q * Cc,
                 kpad_finite=True)               # roi_weights zero-fills [Pn, ldw)
        ctx.dims = (B, Pn, Cc, Q, ldw)
        ctx.save_for_backward(wgt)
        return out

    @staticmethod
    def backward(ctx, dout):
        (wgt,) = ctx.saved_tensors
        B, Pn, Cc, Q, ldw = ctx.dims
        d = _c(_as_compute(dout))
        df = torch.empty(B, Pn, Cc, device=d.device, dtype=RT.dtype)
        hip.gemm(wgt, d, df, Pn, Cc, Q, ldw, Cc, Cc, layoutA=hip.TRANS, layoutB=hip.TRANS, batch=B, sA=Q * ldw,
                 sB=Q * Cc, sC=Pn * Cc)
        return df, None, None, None


def roi_pool(feat, boxes, H, Wd):
    return RoiPoolFn.apply(feat, boxes, H, Wd)
