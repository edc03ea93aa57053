"""Training step driver: the hot loop of exp/gpv/train_distr.py:399-428 rebuilt for one process per GPU.

  encode_answers -> model(imgs, queries, answer_token_ids, targets) -> backward -> gradient all-reduce
  (RCCL over xGMI, flat fp32 buffer, bucketed + asynchronous) -> clip_grad_norm_(DETR params, 0.1)
  -> AdamW (4 parameter groups, train_distr.py:228-253) -> WarmupLinearSchedule step.

MI355X-first choices
  * every parameter that can receive a gradient lives in ONE flat fp32 buffer (and so do its gradient
    and Adam moments): zero_grad is one memset, the all-reduce runs over a few large contiguous
    buckets (ring all-reduce over xGMI is per-link bound, so fewer/larger messages), the clip norm is
    one reduction kernel and AdamW is one fused kernel per contiguous touched range;
  * parameters that can never receive a gradient (BERT under no_grad, vision_token, lang_token,
    q_dense1/2 -- 113 M of the reference's 224 M "trainable" parameters) are left out of the
    gradient exchange instead of relying on DDP find_unused_parameters=True;
  * torch-1.6 optimizer semantics kept: a parameter is only updated (incl. weight decay) once it has
    received a gradient at least once; the touched set is agreed across ranks (MAX all-reduce).
"""
import collections
import gc
import os
import time

import torch
import torch.distributed as dist

from . import hip
from .ops import RT

GROUPS = ('detr_backbone', 'detr_head', 'bert', 'others')


def param_group_of(name):
    """train_distr.py:234-242"""
    if 'detr.backbone' in name:
        return 'detr_backbone'
    if 'detr' in name:
        return 'detr_head'
    if 'bert.' in name:
        return 'bert'
    return 'others'


NEVER_GRAD = ('bert.', 'vision_token', 'lang_token', 'q_dense1', 'q_dense2', 'pos_enc', 'vocab_embed', 'embedding_layer',
              'pooler')


def warmup_linear(step, warmup_steps, t_total):
    """pytorch_transformers.WarmupLinearSchedule (train_distr.py:298-302)"""
    if step < warmup_steps:
        return float(step) / float(max(1, warmup_steps))
    return max(0.0, float(t_total - step) / float(max(1.0, t_total - warmup_steps)))


# Stream captures prohibit "unsafe" runtime calls (event queries, allocations).  'thread_local' restricts that to the capturing
# thread: RCCL's watchdog thread polls the events of the in-flight bucket all-reduces while B2 is being captured.
CAPTURE_MODE = os.environ.get('GPV_CAPTURE_MODE', 'thread_local')
_TRACE = os.environ.get('GPV_CAPTURE_TRACE', '0') == '1'


def _trace(what):
    if _TRACE:
        import sys
        import datetime
        print('[gpv1_amd trace %s] %s' % (datetime.datetime.now().strftime('%H:%M:%S.%f'), what), file=sys.stderr, flush=True)


def init_process_group(rank, world, device, backend=None):
    """torch.distributed over RCCL for one rank per GPU (train_distr.py:176-179 wraps in DDP over NCCL; here FlatTrainer exchanges
    the flat gradient itself).  RCCL's kernels go on a HIGH-priority stream: the bucket all-reduces are issued between the backward
    graphs and must get their few workgroups onto a chip the convolutions fill (256 CUs, two blocks each), or the overlap turns
    into a tail.  device_id: eager communicator init bound to this rank's GPU (no lazy init inside the first step)."""
    cuda = str(device).startswith('cuda')
    backend = backend or ('nccl' if cuda else 'gloo')
    from .misc import install_collective_hooks
    install_collective_hooks()           # synchronous collectives arm the capture quiesce themselves (misc.CollectiveClock)
    if backend != 'nccl':
        dist.init_process_group(backend, rank=rank, world_size=world)
        return
    opts = None
    if os.environ.get('GPV_RCCL_HIGH_PRIO', '1') != '0':
        opts = dist.ProcessGroupNCCL.Options()
        opts.is_high_priority_stream = True
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device(device), pg_options=opts)


def ops_check_chains(clear=True):
    from . import ops
    ops.check_chains(clear)


FLUSH_TAGS = tuple(t for t in os.environ.get('GPV_WGRAD_FLUSH', 'detr').split(',') if t)
HOST_PROF = None          # bench.py: a dict -> host seconds per phase of the graphed step (time.perf_counter, no syncs)


class GraphedBody:
    """Forward and backward of the model body (everything of train_distr.py:413-421 between the host-side tokenisation and
    the criterion) as hipGraphs, for one static input signature (image / query / answer-token shapes).

    Why: a training step is ~1500 kernel launches issued from Python at ~12 us each -- 18 ms of host work per step
    (tools/host_floor.py) against < 26 ms of GPU work; below that the host is the bound.  Replayed graphs cost ~15 us each.

    Shape of the capture (torch.cuda.CUDAGraph; our kernels are launched on the capturing stream through the C ABI):
      F1  backbone forward            images -> c5                      (called outside autograd: ops.RT.split)
      F2  rest of the forward         c5 (autograd leaf) -> outputs dict
      --  criterion: EAGER (Hungarian matcher + set criterion stay on the host by north_star; the text cross-entropy
          is one kernel); its tiny autograd graph ends at detached copies of the outputs
      B1  backward of F2              d(outputs) -> parameter gradients (accumulated by the kernels) + d(c5)
      B2  backbone backward           d(c5) -> backbone parameter gradients
    B1 exists once per set of outputs that carry a gradient (answer logits / box + relevance outputs): a caption-only batch
    must leave the box head untouched (torch-1.6 optimizer semantics), so its B1 never visits it.
    F1 | F2 and B1 | B2 are separate graphs because the trainer hands the gradient buckets behind the backbone segment to
    RCCL between B1 and B2 (the overlap of train_distr.py's DDP), and bench.py brackets F1 / B2 with HIP events.
    Dropout: the seed argument of a captured launch is frozen, the device-resident seed epoch (gpv_set_seed_device) is
    bumped once per replayed step; forward and backward of a step read the same value.
    Optimizer, gradient exchange and clip stay eager (a dozen launches; their scalars change every step)."""

    GRAD_KEYS = ('answer_logits', 'pred_relevance_logits', 'pred_boxes')

    def __init__(self, trainer, images, queries, tok, lang_extra=None):
        from .misc import NestedTensor
        self.tr = trainer
        model = trainer.model
        dev = images.tensors.device
        self.s_img, self.s_mask = images.tensors.clone(), images.mask.clone()
        self.all_valid = getattr(images, 'all_valid', None)
        self.s_ids, self.s_attn = queries[0].clone(), queries[1].clone()
        self.s_tok = tok.clone()
        self.s_extra = None if lang_extra is None else lang_extra.clone()      # size-class padding of the queries (FlatTrainer._classed)
        self.epochs = (RT.static_epoch, RT.dtype)
        self.keep, self.c5, self.c5_leaf = None, None, None
        self.variants = {}
        RT.enable_seed_epoch(dev)
        _trace('capture forward begin')
        trainer.quiesce_collectives()
        torch.cuda.synchronize()
        gc.collect()
        from .ops import release_pending
        release_pending()                  # (capture owners finalised while another capture was open: ops.retire)
        self.pool = torch.cuda.graph_pool_handle()
        self.f1, self.f2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        saved = trainer.touched.clone()
        assert torch.cuda.current_stream(dev) == trainer.stream            # captures run on the trainer's side stream
        # the frozen BERT (no_grad, ~110 launches of 36-144 workgroups) only needs the query ids: it is captured as a parallel
        # branch (fork / join on a second stream) -- beside the DETR transformer's latency-bound chain in F2, where its small
        # kernels find idle CUs (under the backbone's full-chip convolutions in F1 they cost the convolutions more)
        self.bert_mode = int(os.environ.get('GPV_BERT_BRANCH', '2'))      # 2: branch of F2 beside the DETR transformer (F1 3.88 -> 3.71 ms, F2 3.68 -> 3.81), 1: branch of F1, 0: in line
        from .ops import owned_stream
        self.side = owned_stream(dev) if self.bert_mode else None                  # (streams of this body's own: ops.owned_stream)
        self.wside = owned_stream(dev)
        self.bside = owned_stream(dev)          # ops.Branch's side stream inside this body's captures (forward and backward variants)
        self.zero_in_graph = os.environ.get('GPV_ZERO_IN_GRAPH', '1') != '0'
        # the gradient chains of THIS recorded forward belong to the body: an eager step's check_chains(clear=True) must not
        # drop them from under the backward variants captured later (ops.GradChain._live is the eager steps' list)
        from . import ops as _ops
        live_before, _ops.GradChain._live = _ops.GradChain._live, []
        RT.split = self
        RT.branch_stream = self.bside
        self._open = None                      # the graph whose capture is open (ended / aborted on any failure below)
        try:
            self.f1.capture_begin(pool=self.pool, capture_error_mode=CAPTURE_MODE)
            self._open = self.f1
            q_enc = None
            if self.bert_mode == 1:
                self.side.wait_stream(trainer.stream)
                with torch.cuda.stream(self.side), torch.no_grad():
                    q_enc, _ = model.bert((self.s_ids, self.s_attn))
            elif self.bert_mode == 2:
                q_enc = self._bert_join                # forked when F2 opens (backbone_forward), joined where the model needs it
            self.q_enc = q_enc
            self.outs = model._forward_impl(NestedTensor(self.s_img, self.s_mask, self.all_valid), (self.s_ids, self.s_attn),
                                            self.s_tok, None, query_encodings=q_enc, lang_extra=self.s_extra)
            if self._open is not self.f2:
                raise RuntimeError('GraphedBody: the model never reached backbone_forward (F1 was not closed)')
            torch.cuda.current_stream(dev).wait_stream(self.wside)          # join the weight-mirror branch
            _ops.Branch.join_captured(dev)
            _ops.foreign_capturing((self.side, self.wside, self.bside), 'F2 end')
            self.f2.capture_end()
            self._open = None
            _ops.still_capturing((('side', self.side), ('wside', self.wside), ('bside', self.bside)), 'after F2')
        except BaseException:
            self._abort_open()
            raise
        finally:
            RT.split = None
            RT.branch_stream = None
            RT.multi_wait = None
            self.chains, _ops.GradChain._live = _ops.GradChain._live, live_before
        self.fwd_touched = trainer.touched.clone()            # (forward kernels never write gradients: stays empty)
        trainer.touched |= saved

    def _abort_open(self):
        """a failure inside a capture must not leave the trainer's stream capturing (the eager fallback would run on it)"""
        g, self._open = self._open, None
        if g is not None:
            try:
                g.capture_end()
            except Exception:
                pass
        for st in (self.side, self.wside, self.bside):
            if st is not None:
                try:
                    st.synchronize()
                except Exception:
                    pass

    # called by backbone.BackboneBase.forward while this body is being captured (ops.RT.split)
    def backbone_forward(self, body, x, train=True, hw=None):
        """train=False: no backbone block is trainable (phase-1 `training.freeze`, lr_backbone = 0): F1 keeps nothing for a
        backward, c5 is a constant of F2 and B2 has no backbone part"""
        self.keep = [] if train else None
        c5 = body.forward_nhwc(x, self.keep, hw)
        if self.bert_mode == 1:
            torch.cuda.current_stream(x.device).wait_stream(self.side)         # join the BERT branch before F1 ends
        self._prep_forked = False
        self.f1.capture_end()
        self._open = None
        from . import ops as _ops2
        _ops2.still_capturing((('side', self.side), ('wside', self.wside), ('bside', self.bside)), 'after F1')
        self.f2.capture_begin(pool=self.pool, capture_error_mode=CAPTURE_MODE)
        self._open = self.f2
        if self.bert_mode == 2:
            cur = torch.cuda.current_stream(x.device)
            self.side.wait_stream(cur)
            with torch.cuda.stream(self.side), torch.no_grad():
                self._q_enc, _ = self.tr.model.bert((self.s_ids, self.s_attn))
        # the W^T mirrors of the Linear weights (ops._lpT; needed by B1's backward-data GEMMs) are refreshed by one grouped
        # cast-transpose launch on a branch of F2: 360 MB of streaming under a chain of latency-bound kernels.  (As a branch of
        # F1 it ran beside the stem convolution and cost it 0.2 ms.)
        from . import ops
        cur = torch.cuda.current_stream(x.device)
        self.wside.wait_stream(cur)
        with torch.cuda.stream(self.wside):
            ops.refresh_transposed()
            ops.refresh_multi()            # concatenated weights of the multi_linear sites (cross-attention K / V, co-attention q|k|v)
            multi_ready = torch.cuda.Event()
            multi_ready.record(self.wside)
            if self.zero_in_graph:
                # the flat gradient buffer is cleared here, on the branch beside the transformer's latency-bound chain (444 MB of
                # streaming stores), instead of between the criterion and B1 on the critical path: nobody reads G between the
                # previous step's AdamW and this step's first weight-gradient kernel (B1)
                self.tr.G.zero_()
        waited = set()

        def multi_wait():                  # first multi_linear of F2 ON EACH STREAM: the copies are ready (long before: they follow the six encoder layers)
            cur = torch.cuda.current_stream(x.device)        # (one-shot was only right while the first call ran on the main stream;
            if cur.cuda_stream not in waited:                #  a first call on the BERT / side branch left later main-stream calls unordered)
                cur.wait_event(multi_ready)
                waited.add(cur.cuda_stream)
        RT.multi_wait = multi_wait
        self.c5 = c5
        self.c5_leaf = c5.detach().requires_grad_(bool(train))
        self.body = body
        return self.c5_leaf

    def _bert_join(self):
        torch.cuda.current_stream(self.s_img.device).wait_stream(self.side)
        return self._q_enc

    # the backbone's per-step weight copies (42 launches of 5-9 us) as a branch of F1 beside the stem and layer1
    def prep_fork(self, body):
        if os.environ.get('GPV_PREP_BRANCH', '1') == '0':
            return
        cur = torch.cuda.current_stream(self.s_img.device)
        self.wside.wait_stream(cur)
        with torch.cuda.stream(self.wside):
            body.prep_weights()
            self._prep_done = torch.cuda.Event()
            self._prep_done.record(self.wside)
        self._prep_forked = True

    def prep_join(self):
        if getattr(self, '_prep_forked', False):
            torch.cuda.current_stream(self.s_img.device).wait_event(self._prep_done)

    def stale(self):
        return self.epochs != (RT.static_epoch, RT.dtype)

    def __del__(self):
        """graphs first, then -- with the device idle -- the streams they were captured on (ops.retire; parked if a capture is open)"""
        try:
            from .ops import retire
            streams = [getattr(self, n, None) for n in ('side', 'wside', 'bside')]
            graphs = [getattr(self, n, None) for n in ('f1', 'f2')] + list(getattr(self, 'variants', {}).values())
            self.f1 = self.f2 = None
            if hasattr(self, 'variants'):
                self.variants.clear()
            self.keep = None
            retire(graphs, streams)
        except Exception:                      # (interpreter shutdown: modules may be gone)
            pass

    def forward(self, images, queries, tok, lang_extra=None, leaves=True):
        """replay F1 + F2 on the current stream; returns the outputs dict as fresh autograd leaves (leaves=False: nothing -- the
        criterion runs inside the backward graph, backward_fused)"""
        from . import backbone as bbm
        self.s_img.copy_(images.tensors, non_blocking=True)
        self.s_mask.copy_(images.mask, non_blocking=True)
        self.s_ids.copy_(queries[0], non_blocking=True)
        self.s_attn.copy_(queries[1], non_blocking=True)
        self.s_tok.copy_(tok, non_blocking=True)
        if self.s_extra is not None:
            self.s_extra.copy_(lang_extra, non_blocking=True)
        RT.seed_dev.add_(1)
        ev = bbm._prof('conv_fwd')
        self.f1.replay()
        if ev is not None:
            ev.record()
        ev = bbm._prof('graph_f2')
        self.f2.replay()
        if ev is not None:
            ev.record()
        if not leaves:
            return None
        leaves = {}
        for k, v in self.outs.items():
            if torch.is_tensor(v):
                leaves[k] = v.detach().requires_grad_(v.requires_grad and k in self.GRAD_KEYS)
            elif k == 'aux_outputs':
                leaves[k] = [{kk: vv.detach().requires_grad_(vv.requires_grad) for kk, vv in a.items()} for a in v]
            else:
                leaves[k] = v
        return leaves

    def _roots(self, leaves):
        """(static output, gradient of its leaf) pairs for every output the criterion reached"""
        pairs = []
        for k in self.GRAD_KEYS:
            if k in leaves and leaves[k].grad is not None:
                pairs.append((k, self.outs[k], leaves[k].grad))
        for i, a in enumerate(leaves.get('aux_outputs', ())):
            for kk, vv in a.items():
                if vv.grad is not None:
                    pairs.append((('aux', i, kk), self.outs['aux_outputs'][i][kk], vv.grad))
        return pairs

    def backward(self, leaves):
        """d(outputs) = the .grad of the leaves forward() returned -> replay B1 (+ the trainer's milestone) + B2"""
        tr = self.tr
        pairs = self._roots(leaves)
        if not pairs:
            return
        vkey = (tuple(k for k, _, _ in pairs), bool(tr.defer_wgrad))
        var = self.variants.get(vkey)
        if var is None:
            var = self.variants[vkey] = self._capture_backward(pairs)
        for (k, _, g), sg in zip(pairs, var['grads']):
            sg.copy_(g, non_blocking=True)
        self._replay_backward(var)

    def _replay_backward(self, var):
        from . import backbone as bbm
        tr = self.tr
        tr.touched |= var['touched']
        ev = bbm._prof('graph_b1')
        var['b1'].replay()
        if ev is not None:
            ev.record()
        if RT.backward_milestone is not None:
            RT.backward_milestone('head' if var['head_late'] else 'backbone')
        ev = bbm._prof('conv_bwd')
        for g, tag in var['b2']:                 # one graph (single GPU) or one per backbone stage (several ranks: stage milestones)
            g.replay()
            if tag is not None and RT.backward_milestone is not None:
                RT.backward_milestone(tag)
        if ev is not None:
            ev.record()

    # ---- criterion inside the graph: batches whose every sample carries the SAME text task and no boxes (caption-only = BASELINE
    # configs[1], VQA-only, classification-only).  Their criterion is one cross-entropy kernel plus a handful of tiny reductions,
    # all on the device; eager, those ~10 launches and the copies of d(outputs) into B1's static inputs sit between F2 and B1 with
    # host-paced gaps (0.47 ms wall for 0.13 ms of kernels, profiles/r03_step_phases.txt).  Here criterion forward, its backward
    # and B1 are ONE graph whose only input is the CE target ids.  Batches with boxes keep the eager criterion (Hungarian matching
    # on the host, north_star).
    @staticmethod
    def fused_task(criterion, targets):
        """the single text task of a batch the fused variant serves, or None"""
        if os.environ.get('GPV_FUSED_CRITERION', '1') == '0' or not targets:
            return None
        task = targets[0].get('task')
        for t in targets:
            if t.get('task') != task or 'answer' not in t or 'boxes' in t or 'answer_token_ids' not in t:
                return None
        names = [n for n in criterion.criterion_names if getattr(getattr(criterion, n), 'task', '-') == task]
        return task if len(names) == 1 else None

    def backward_fused(self, task, ce_targets):
        """ce_targets: [B, S_c - 1] int64 device tensor (the rows the targets' 'answer_token_ids' are views of) -> the loss"""
        tr = self.tr
        vkey = ('fused', task, bool(tr.defer_wgrad))
        var = self.variants.get(vkey)
        if var is None:
            var = self.variants[vkey] = self._capture_backward([], fused=(task, ce_targets))
        var['s_ce'].copy_(ce_targets, non_blocking=True)
        self._replay_backward(var)
        return var['loss']

    @staticmethod
    def _flush(deferred):
        """launch the collected weight gradients: every problem the grouped kernel takes (128-multiples, bf16) in
        gpv_gemm_tt_group launches of up to 48 problems, the rest (2- and 4-wide heads, ragged shapes) one by one"""
        is_fold = lambda prob: prob is not None and isinstance(prob[0], str)      # ('fold', partials, dgamma, dbeta, nblk, cols): ops._ln_backward
        group = [prob for _, _, prob in deferred if prob is not None and not is_fold(prob)]
        folds = [prob[1:] for _, _, prob in deferred if is_fold(prob)]
        if os.environ.get('GPV_WGRAD_GROUP', '1') == '0':
            group, folds = [], []
        if group:
            hip.gemm_tt_group(group)
        if folds:
            hip.colsum_fold_group(folds)           # the LayerNorm backwards' partial dgamma | dbeta rows of this flush, one launch
        for fn, _, prob in deferred:
            if prob is None or not (folds if is_fold(prob) else group):
                fn()

    def _capture_backward(self, pairs, fused=None):
        from . import ops as _ops
        tr = self.tr
        _trace('capture backward begin (fused=%r)' % (fused is not None,))
        tr.quiesce_collectives()
        torch.cuda.synchronize()
        gc.collect()
        _ops.release_pending()
        grads = [torch.zeros_like(g) for _, _, g in pairs]
        s_ce, s_targets, loss_static = None, None, None
        if fused is not None:
            task, ce = fused
            s_ce = ce.clone()
            s_targets = [{'task': task, 'answer': '', 'answer_token_ids': s_ce[i]} for i in range(s_ce.shape[0])]
        b1, b2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        b2_list = []
        saved = tr.touched.clone()
        tr.touched.zero_()
        milestone, RT.backward_milestone = RT.backward_milestone, None
        self.c5_leaf.grad = None
        from .ops import _DUMMY
        for d in _DUMMY.values():
            d.grad = None
        # The weight-gradient GEMMs of the model body are collected during B1 (ops.wgrad_linear) and launched as grouped kernels:
        # the text / co-attention group on a branch beside the DETR backward, the DETR group on a branch of B2 under the
        # backbone's convolutions (single GPU) or at the end of B1 (several ranks: the buckets behind the backbone segment
        # must be complete when the trainer hands them to RCCL between B1 and B2).
        dev = self.s_img.device
        defer = bool(tr.defer_wgrad)
        deferred = []
        side_a = []

        def at_boundary(tag):
            # The model body's backward is a latency-bound chain of small kernels.  When it reaches the DETR stream, the weight
            # gradients collected so far (text decoder, answer head, co-attention: the big 768-wide ones) start on a side branch
            # and run beside the DETR decoder / encoder backward; those of the DETR layers follow under the backbone (B2).
            # (A second boundary, 'mem' = the encoder output, can do the same for the DETR decoder's weight gradients beside the
            #  encoder backward -- GPV_WGRAD_FLUSH=detr,mem -- measured zero-sum: B2 6.54 -> 6.24 ms, B1 4.98 -> 5.31 ms.)
            if tag not in FLUSH_TAGS or not deferred or not defer:
                return
            cur = torch.cuda.current_stream(dev)
            self.wside.wait_stream(cur)
            _ops.branch_wait(self.wside)            # (operands written by backward nodes on the branch stream, ops.Branch)
            with torch.cuda.stream(self.wside):
                self._flush(deferred)
            side_a.extend(deferred)
            del deferred[:]
        RT.branch_stream = self.bside
        try:
            b1.capture_begin(pool=self.pool, capture_error_mode=CAPTURE_MODE)
            self._open = b1
            RT.defer_list = deferred if defer else None
            RT.backward_boundary = at_boundary if os.environ.get('GPV_WGRAD_SPLIT', '1') != '0' else None
            from . import ops as _ops
            _ops.reset_chains(self.chains)          # (an earlier capture over this recorded forward may have aborted mid-walk)
            if fused is None:
                torch.autograd.backward([o for _, o, _ in pairs], grads, retain_graph=True)
            else:
                loss = tr.model.criterion(self.outs, s_targets)[0]
                if loss is None:
                    raise RuntimeError('GraphedBody: the fused criterion found no applicable loss for task %r' % (fused[0],))
                loss_static = loss.detach()
                if _ops.DEBUG_STREAMS:
                    _ops.report_unpinned_leaves([loss] + [v for v in self.outs.values() if torch.is_tensor(v)], tr._grad_accs, tr.model, 'B1 capture')
                torch.autograd.backward([loss], retain_graph=True)
                del loss
            _ops.check_chains(clear=False, chains=self.chains)     # (the recorded forward -- and its chains -- serve further backward variants)
            RT.defer_list = None
            RT.backward_boundary = None
            head_late = False
            if deferred and tr.comm:
                # more than one rank.  The buckets behind the DETR-head segment go to RCCL between B1 and B2; the DETR layers' group
                # may stay where it is on one GPU (a branch of B2's first stage graph, under layer4's convolutions) iff every
                # gradient it writes lies in the head segment, whose buckets are then handed over with layer4's ('head' milestone
                # instead of 'backbone').  Otherwise (or GPV_HEAD_LATE=0) it runs here, at the end of B1.
                lo, hi = tr.backbone_end, tr.head_end
                g0 = tr.G.data_ptr()
                head_late = os.environ.get('GPV_HEAD_LATE', '1') != '0' and bool(self.keep) and self.c5_leaf.grad is not None and \
                    all(len(t) > 2 and lo <= (t[2].data_ptr() - g0) // 4 < hi for _, t, _ in deferred)
                if not head_late:
                    _ops.branch_wait(torch.cuda.current_stream(dev))
                    self._flush(deferred)
                    side_a.extend(deferred)
                    del deferred[:]
            if side_a:
                torch.cuda.current_stream(dev).wait_stream(self.wside)
            _ops.Branch.join_captured(dev)
            _ops.foreign_capturing((self.side, self.wside, self.bside), 'B1 end')
            b1.capture_end()
            self._open = None
            _ops.still_capturing((('side', self.side), ('wside', self.wside), ('bside', self.bside)), 'after B1')
            dc5 = self.c5_leaf.grad
            bb_bwd = bool(self.keep) and dc5 is not None        # (frozen backbone: c5 is a constant, nothing behind it)
            if deferred or bb_bwd:
                b2.capture_begin(pool=self.pool, capture_error_mode=CAPTURE_MODE)
                self._open = b2
                if deferred and bb_bwd:
                    self.wside.wait_stream(torch.cuda.current_stream(dev))
                    with torch.cuda.stream(self.wside):
                        self._flush(deferred)
                elif deferred:
                    self._flush(deferred)
                if bb_bwd:
                    # several ranks: B2 is cut into one graph per backbone stage (layer4 | layer3 | layer2) so that the trainer
                    # can hand a finished stage's gradient buckets to the all-reduce between the replays (the exchange is not
                    # captured); tensors crossing a cut live in the shared graph pool, like c5 between F1 and F2
                    cut = tr.comm or tr.dry_overlap or os.environ.get('GPV_B2_STAGES', '0') == '1'
                    last_li = self.body.stage_of(self.keep[0][0])
                    cur = [b2]

                    def stage_done(li):
                        if li == last_li:
                            return
                        if deferred and len(b2_list) == 0:          # (single rank + cut: the DETR weight-gradient branch forked at the
                            torch.cuda.current_stream(dev).wait_stream(self.wside)      # top of B2 joins before the first cut)
                        cur[0].capture_end()
                        b2_list.append((cur[0], 'layer%d' % li))
                        g = torch.cuda.CUDAGraph()
                        g.capture_begin(pool=self.pool, capture_error_mode=CAPTURE_MODE)
                        self._open = cur[0] = g
                    self.body.backward_nhwc(self.keep, dc5.to(RT.dtype), stage_done if cut else None)
                    if deferred:
                        torch.cuda.current_stream(dev).wait_stream(self.wside)
                    cur[0].capture_end()
                    b2_list.append((cur[0], 'layer%d' % last_li))
                else:
                    b2.capture_end()
                    _ops.still_capturing((('side', self.side), ('wside', self.wside), ('bside', self.bside)), 'after B2')
                    b2_list.append((b2, None))
                self._open = None
        except BaseException:
            self._abort_open()
            raise
        finally:
            RT.defer_list = None
            RT.backward_boundary = None
            RT.branch_stream = None
        RT.backward_milestone = milestone
        _trace('capture backward end: %d stage graphs' % len(b2_list))
        var = {'head_late': head_late, 'b1': b1, 'b2': b2_list, 'grads': grads, 's_ce': s_ce, 'loss': loss_static, 'touched': tr.touched.clone(), 'dc5': dc5, 'deferred': deferred + side_a}
        tr.touched |= saved
        return var


class FlatTrainer:
    """AdamW over ONE flat fp32 buffer (P / G / M / V) + the data-parallel gradient exchange (train_distr.py:228-253, 399-428).

    `G` AND EVERY `p.grad` VIEW HOLD THE SUM OVER RANKS after allreduce_grads(), not the average the reference's DDP leaves in
    `p.grad`: the 1 / world rides on the factor the AdamW kernel multiplies every gradient with (and the clip compares |sum| with
    world x max_norm), which saves a pass over the 444 MB buffer per step.  Anything that reads gradients between the exchange and
    the step -- norm logging, external clipping, debugging hooks -- must multiply by `grad_scale` (= 1 / world; `grad_norm()` does).
    One consequence in the last bits: the clip factor is max_norm x world / (|sum| + 1e-6) where the reference computes
    max_norm / (|average| + 1e-6) -- the epsilon weighs 1 / world as much (1e-6 against norms of 1e-1 .. 1e2: below fp32 resolution
    of the factor for world <= 8)."""

    @property
    def grad_scale(self):
        """what turns G / p.grad (sums over ranks after the exchange) into the reference's averaged gradients"""
        return self.avg

    def grad_norm(self, start=0, end=None):
        """L2 norm of the AVERAGED gradient over the flat range [start, end) -- what the reference would log from p.grad"""
        return self.G[start:self.total if end is None else end].norm() * self.grad_scale

    def __init__(self, model, lr=1e-4, lr_backbone=1e-5, weight_decay=1e-4, clip_max_norm=0.1, warmup_steps=0,
                 t_total=0, betas=(0.9, 0.999), eps=1e-8, bucket_mb=128, process_group=None, manual_gc=True, gc_interval=200,
                 graphs=None, lr_milestones=None, lr_drop=0.1, warmup_iters=0, grad_comm_dtype=None):
        self.model = model
        # gradient exchange precision: fp32 (the reference's DDP) or bf16 (half the bytes on the xGMI ring: 222 instead of
        # 444 MB per step; the local gradients stay fp32, every rank receives the same bf16 sums -> replicas stay bit-identical)
        gcd = grad_comm_dtype if grad_comm_dtype is not None else os.environ.get('GPV_GRAD_COMM', 'fp32')
        self.grad_comm_dtype = {'fp32': torch.float32, 'bf16': torch.bfloat16}.get(gcd, gcd)
        self.comm_prof = None                    # bench.py: a list -> (first wait, all buckets back) event pairs of every exchange
        self.lr_milestones, self.lr_drop, self.warmup_iters = (list(lr_milestones) if lr_milestones is not None else None), lr_drop, warmup_iters
        self.epoch, self.it_in_epoch = 0, 0
        # hipGraph replay of the model body (GraphedBody): default on for bf16 on the GPU; GPV_TRAIN_GRAPHS=0 turns it off
        self.graphs = (os.environ.get('GPV_TRAIN_GRAPHS', '1') != '0') if graphs is None else bool(graphs)
        self._bodies, self._seen, self.stream = collections.OrderedDict(), {}, None      # captured bodies, least recently used first
        self.graph_slots = int(os.environ.get('GPV_TRAIN_GRAPH_SLOTS', '8'))             # each body pins its activation pool (a few GB at B = 32)
        self.graph_steps, self.eager_steps = 0, 0
        self._extra_cache = {}
        self.defer_wgrad = None          # decided below (single rank only): model-body weight gradients as a branch of the backbone backward graph
        # Python's cyclic collector costs 1-3 ms per step once it has a few hundred thousand module / tensor objects to
        # walk (measured: 970-1020 vs 1062-1070 images/s over 20 steps), while a step leaves ~17 small cycles and no
        # device memory behind (tools/gc_growth.py).  With manual_gc the trainer freezes the long-lived objects, turns
        # the automatic collector off and collects every gc_interval steps itself (Megatron-style).
        self.manual_gc, self.gc_interval, self._gc_armed = manual_gc, gc_interval, False
        self.lr = {'detr_backbone': lr_backbone, 'detr_head': lr, 'bert': lr, 'others': lr}
        self.wd, self.clip, self.betas, self.eps = weight_decay, clip_max_norm, betas, eps
        self.warmup_steps, self.t_total = warmup_steps, t_total
        self.step_count = 0
        self.evict_interval = int(os.environ.get('GPV_GRAPH_EVICT_INTERVAL', '50'))       # steps between two evictions of a captured body
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if self.world > 1 else 0
        # comm: the gradient exchange and everything that exists for it (bucket milestones, the stage cut of B2, the host agreement
        # channel) is on.  GPV_FORCE_COMM=1 turns it on for a process group of ONE rank: the whole N > 1 code path under the real
        # backend (RCCL) on a single-GPU box -- collectives that move nothing, same streams / events / graph cuts as on a node
        self.comm = self.world > 1 or (dist.is_available() and dist.is_initialized() and os.environ.get('GPV_FORCE_COMM', '0') == '1')
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad and not any(s in n for s in NEVER_GRAD)]
        named.sort(key=lambda np_: GROUPS.index(param_group_of(np_[0])))           # stable: module order inside a group
        self.entries = []                                                           # (name, param, group, offset, numel)
        off = 0
        for n, p in named:
            self.entries.append((n, p, param_group_of(n), off, p.numel()))
            off += (p.numel() + 7) // 8 * 8                                          # keep every segment 32-B aligned
        self.total = off
        dev = named[0][1].device
        if self.graphs and dev.type == 'cuda':
            self.stream = torch.cuda.Stream(device=dev)
        self.P = torch.zeros(off, device=dev, dtype=torch.float32)
        self.G = torch.zeros_like(self.P)
        self.M = torch.zeros_like(self.P)
        self.V = torch.zeros_like(self.P)
        self.Pb = torch.zeros(off, device=dev, dtype=torch.bfloat16)      # bf16 mirror of P, rewritten by the AdamW kernel
        self.touched = torch.zeros(len(self.entries), dtype=torch.bool)           # host: which parameters THIS rank's kernels wrote
        # device: which parameters any rank has ever written (torch-1.6 optimizers skip the rest).  The AdamW kernel reads
        # it through seg_id (parameter index of every 8-element chunk), so no host round trip is needed to pick ranges.
        self.live = torch.zeros(len(self.entries), device=dev, dtype=torch.int32)
        # per-parameter Adam step counts (torch keeps `step` per parameter, starting at its first gradient): += live per step
        self.pstep = torch.zeros(len(self.entries), device=dev, dtype=torch.int32)
        sid = torch.zeros(off // 8, dtype=torch.int16)
        for i, (n, p, g, o, k) in enumerate(self.entries):
            sid[o // 8:(o + k + 7) // 8] = i
        self.seg_id = sid.to(dev)
        self.group_range = {}
        for (n, p, g, o, k) in self.entries:
            lo, hi = self.group_range.get(g, (o, o))
            self.group_range[g] = (min(lo, o), max(hi, o + (k + 7) // 8 * 8))
        for i, (n, p, g, o, k) in enumerate(self.entries):
            pv, gv = self._view(self.P, p, o, k), self._view(self.G, p, o, k)
            pv.copy_(p.data)
            p.data = pv
            p.grad = gv
            p._gpv_managed = True                                                  # compute copies follow weights_epoch
            p._gpv_flat, p._gpv_lp = self.P[o:o + k], self.Pb[o:o + k]            # fp32 master / bf16 mirror segments
            p._gpv_lp_static = -1                                                  # mirror not yet written (ops._lp casts on first use)
            p._gpv_touch = (lambda i=i: self._mark(i))                            # kernel-accumulated gradients
            p.register_hook(lambda grad, i=i: self._mark(i))                       # autograd-delivered gradients
        # Every parameter's AccumulateGrad node is made HERE, on the trainer's stream, and kept for the trainer's life.  autograd runs such
        # a node on the stream that was current when the node was CREATED, and creates it lazily where the parameter is first used: a
        # parameter first used on a captured body's branch stream (ops.Branch: the co-attention language weights, the teacher-forcing
        # prologue) had its accumulation pinned to THAT body's stream -- a later body's backward capture then ran it on a stream outside
        # its capture (forked in by the engine's event wait, joined by nobody: an invalid capture that ROCm 7.2 answers with a
        # segmentation fault in some later replay / capture_end), or, once the old body was evicted, on a stream that no longer existed
        # (round 6, tools/soak_evict.py: 2 graph slots, an eviction on every miss -- 12 evictions to the crash; with the branches off,
        # or without evictions, never).  On the trainer's stream the engine's wait is the join of the branch.
        self._grad_accs = []
        if getattr(self, 'stream', None) is not None:
            with torch.cuda.stream(self.stream), torch.enable_grad():
                for (n, p, g, o, k) in self.entries:
                    if p.requires_grad:
                        self._grad_accs.append(p.view_as(p).grad_fn.next_functions[0][0])
        self.gscale = torch.ones(1, device=dev, dtype=torch.float32)
        # several ranks: G holds the SUM over ranks after the exchange; the 1 / world of the average rides on the factor the AdamW
        # kernel multiplies every gradient with anyway (clip factor x avg for the DETR groups, avg alone for the others) instead
        # of a pass over the 444 MB buffer (0.2 ms per step)
        self.avg = 1.0 / self.world
        self.avg_t = torch.full((1,), self.avg, device=dev, dtype=torch.float32) if self.world > 1 else None
        self._clip_ws = torch.zeros(hip.CLIP_PARTIALS, device=dev, dtype=torch.float32)
        self._published = None                   # the host 'touched' marks as last sent to the device
        self._step_state = None
        b = bucket_mb * 1024 * 1024 // 4
        self.backbone_end = max([o + (k + 7) // 8 * 8 for (n, p, g, o, k) in self.entries if g == 'detr_backbone'], default=0)
        # flat ranges of the backbone's stages (module order = flat order: layer2 | layer3 | layer4): each is a milestone of the
        # backward pass -- its buckets go to the all-reduce as soon as the stage's weight gradients have been issued
        self.stage_range = {}
        for (n, p, g, o, k) in self.entries:
            if g != 'detr_backbone':
                continue
            st = next((t for t in ('layer1', 'layer2', 'layer3', 'layer4') if '.%s.' % t in n), 'stem')
            lo, hi = self.stage_range.get(st, (o, o))
            self.stage_range[st] = (min(lo, o), max(hi, o + (k + 7) // 8 * 8))
        self.head_end = max(self.backbone_end, self.group_range.get('detr_head', (0, 0))[1])
        cuts = sorted({0, self.backbone_end, self.head_end, off} | {v for r in self.stage_range.values() for v in r})
        self.buckets = [(s, min(hi, s + b)) for lo, hi in zip(cuts[:-1], cuts[1:]) for s in range(lo, hi, b)]   # no bucket straddles a stage / the backbone boundary
        # the order in which EVERY rank issues them: behind the backbone first, then the backbone stages last-to-first
        # (behind the DETR head | the DETR head | the backbone's stages)
        self.bucket_order = [q for q in self.buckets if q[0] >= self.head_end] + \
            [q for q in self.buckets if self.backbone_end <= q[0] < self.head_end] + \
            sorted((q for q in self.buckets if q[0] < self.backbone_end), key=lambda q: -q[0])
        self.Gc = torch.zeros(off, device=dev, dtype=torch.bfloat16) if (self.comm and self.grad_comm_dtype == torch.bfloat16) else None
        self.overlap = self.comm and os.environ.get('GPV_OVERLAP', '1') != '0'
        self.dry_overlap = False             # tests: run the milestone / guard logic without communicating
        self._closed_from, self._works, self.late_touch, self.milestones = None, [], None, 0
        self._next_bucket, self.milestone_log = 0, []
        self.defer_wgrad = os.environ.get('GPV_DEFER_WGRAD', '1') != '0'       # (with several ranks the groups stay inside B1)
        self.host_pg = None
        if self.comm:
            from .misc import install_collective_hooks
            install_collective_hooks()       # (a process group the caller built itself: the hooks go in here at the latest)
            RT.per_rank_seed(self.rank)      # independent dropout masks per rank unless the caller seeded (ops.Runtime.per_rank_seed)
            dist.broadcast(self.P, src=0, group=self.pg)
            # DDP broadcasts every parameter AND buffer from rank 0 at construction; the tensors this trainer does not manage
            # (frozen BERT, vocabulary embedding, FrozenBN statistics, the frozen stem / layer1) must not depend on each rank's seed
            for t in list(model.buffers()) + [p.data for n, p in model.named_parameters() if not getattr(p, '_gpv_managed', False)]:
                if t.numel() == 0:
                    continue
                if t.is_contiguous():
                    dist.broadcast(t, src=0, group=self.pg)
                else:                                                                 # channels_last conv weights
                    c = t.contiguous()
                    dist.broadcast(c, src=0, group=self.pg)
                    t.copy_(c)
            from .misc import note_sync_collective
            note_sync_collective()              # (synchronous broadcasts on this stream: the first capture waits for the watchdog once)
            # host-side agreement channel (train_step): gloo, so that no device synchronisation is involved
            self.host_pg = self.pg if dist.get_backend(self.pg) == 'gloo' else dist.new_group(backend='gloo')
        RT.bump_weights()

    @staticmethod
    def _view(flat, p, off, numel):
        if p.dim() == 4:                                                            # conv weight: channels_last memory
            co, ci, kh, kw = p.shape
            return flat[off:off + numel].view(co, kh, kw, ci).permute(0, 3, 1, 2)
        return flat[off:off + numel].view(p.shape)

    def _mark(self, i):
        self.touched[i] = True
        if self._closed_from is not None and self.entries[i][3] >= self._closed_from:
            # a gradient kernel was issued for a parameter whose bucket is already being all-reduced: its contribution
            # would stay rank-local.  Never observed (see _on_milestone); if the autograd order ever changes, fail loudly.
            self.late_touch = self.entries[i][0]
            if not self.dry_overlap:
                raise RuntimeError(f'gpv1_amd.FlatTrainer: gradient of {self.entries[i][0]} written after its bucket was '
                                   f'handed to the all-reduce (set GPV_OVERLAP=0 to disable the overlap)')

    def zero_grad(self):
        self.G.zero_()

    # ---- gradient exchange: average over ranks, a few large buckets, async, overlapped with the backbone backward ----
    # Flat order = [detr_backbone | detr_head | others]; the backward pass runs the other way round, and when autograd
    # reaches ResNetFn.backward every node created after the backbone in the forward (transformer, heads, co-attention,
    # text decoder: 350 MB of the 444 MB) has already run -- the engine orders ready nodes by creation sequence.  At
    # that milestone the buckets behind the backbone segment are all-reduced on RCCL's stream while the backbone
    # backward (8 of the ~22 ms) computes; the backbone segment follows after the pass.  _mark() guards the assumption.
    def _on_milestone(self, what):
        """'backbone': the backward pass has reached the backbone (every gradient behind its flat segment is complete);
        'layerN': every weight gradient of backbone stage N has been issued (backbone.backward_nhwc stage_done).  The flat
        offsets from the milestone's start on are closed (_mark guards it) and their buckets handed to the all-reduce."""
        if not (self.overlap or self.dry_overlap):
            return
        if what == 'backbone':
            start = self.backbone_end
        elif what == 'head':                 # graphed step whose DETR weight-gradient group runs inside B2's first stage graph
            start = self.head_end
        elif what in self.stage_range:
            start = self.stage_range[what][0]
        else:
            return
        if self._closed_from is not None and start >= self._closed_from:
            return
        self._closed_from = start
        self.milestones += 1
        self.milestone_log.append((what, start))
        if self.overlap:
            self._issue_upto(start)

    def _issue_upto(self, start):
        """issue, in the canonical order, every bucket not yet issued that lies at or behind flat offset `start`"""
        while self._next_bucket < len(self.bucket_order) and self.bucket_order[self._next_bucket][0] >= start:
            s, e = self.bucket_order[self._next_bucket]
            self._works.append(self._reduce_bucket(s, e))
            self._next_bucket += 1

    def _reduce_bucket(self, s, e):
        """asynchronous SUM all-reduce of G[s:e] (through the bf16 staging buffer when grad_comm_dtype is bf16)"""
        _trace('all-reduce bucket [%d, %d) step %d' % (s, e, self.step_count))
        if self.Gc is None:
            return (dist.all_reduce(self.G[s:e], op=dist.ReduceOp.SUM, group=self.pg, async_op=True), s, e)
        self.Gc[s:e].copy_(self.G[s:e])
        return (dist.all_reduce(self.Gc[s:e], op=dist.ReduceOp.SUM, group=self.pg, async_op=True), s, e)

    def quiesce_collectives(self):
        """Before a stream capture with a process group alive.  ProcessGroupNCCL's watchdog thread polls the completion event of every
        collective it has not yet seen finish (every 100 ms).  HIP refuses a query of an event whose last record was on a stream that
        is capturing NOW (hipErrorCapturedEvent) and invalidates that capture: a collective issued on the trainer's stream a few
        ms before a capture begins on the same stream -- the synchronous ones run on the caller's stream -- killed 2 of 12 runs
        (capture error 901 + the watchdog's exception ends the process; found with GPV_FORCE_COMM=1 on one GPU).  Two measures:
        every collective of the trainer is asynchronous (RCCL's own stream never captures), and a capture starts only after the
        device is idle and the watchdog has had three polling periods to retire what it was watching."""
        if self.comm:
            from .misc import quiesce_collectives
            quiesce_collectives()

    def begin_backward(self):
        self._closed_from, self._works, self.late_touch, self.milestones = None, [], None, 0
        self._next_bucket, self.milestone_log = 0, []
        RT.backward_milestone = self._on_milestone

    def allreduce_grads(self):
        RT.backward_milestone = None
        self._closed_from = None
        if not self.comm:
            return
        # Every rank issues the buckets in the SAME order (bucket_order: behind the backbone segment, then layer4, layer3, layer2)
        # whether or not its backward reached the milestones (a rank without an applicable target runs no backward at all, see
        # train_step): whatever has not been handed over yet goes now.
        first_late = self._next_bucket              # buckets [0, first_late) of bucket_order were handed over during the backward pass
        self._issue_upto(0)
        works = list(self._works)
        self._works = []
        self.left_after_backward = sum(e - s for s, e in self.bucket_order[first_late:]) * (2 if self.Gc is not None else 4)   # bytes
        self._publish_touched()
        # (stays on the device: no host sync.  Asynchronous + wait: on RCCL's stream, see quiesce_collectives)
        dist.all_reduce(self.live, op=dist.ReduceOp.MAX, group=self.pg, async_op=True).wait()
        prof = self.comm_prof is not None and self.G.is_cuda
        if prof:                                   # exposed communication: what the compute stream waits for from here on
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for w, s, e in works:
            w.wait()
            if self.Gc is not None:
                self.G[s:e].copy_(self.Gc[s:e])
        if prof:
            e1.record()
            self.comm_prof.append((e0, e1))

    def comm_bytes_per_step(self):
        """bytes every rank hands to the all-reduce per step"""
        return self.total * (2 if self.Gc is not None else 4)

    def _publish_touched(self):
        """host-side 'touched' marks of this step -> device flags (idempotent; pinned staging, asynchronous)"""
        from .misc import STAGER
        if self._published is not None and torch.equal(self._published, self.touched) and not self.comm:
            return                               # (steady state: the same parameters every step -- nothing new to tell the device)
        loc = STAGER.to_device(self.touched.to(torch.int32), torch.int32, self.live.device)
        torch.maximum(self.live, loc, out=self.live)
        self._published = self.touched.clone()

    def live_host(self):
        return self.live.cpu().bool()

    def _ranges(self, pred):
        """contiguous flat ranges of touched entries satisfying pred(group), merged per group"""
        out = []
        for i, (n, p, g, o, k) in enumerate(self.entries):
            if not self.touched[i] or not pred(g):
                continue
            end = o + (k + 7) // 8 * 8
            if out and out[-1][0] == g and out[-1][2] == o:
                out[-1][2] = end
            else:
                out.append([g, o, end])
        return out

    def _begin_step(self):
        """first half of the optimizer step, legal as soon as this step's `touched` set is final: liveness flags to the device,
        per-parameter Adam step counts += live, schedule position.  Idempotent within a step."""
        if self._step_state is not None:
            return self._step_state
        sched = self.lr_factor()
        self._publish_touched()
        hip.clip_scale(None, 0.0, self._clip_ws, self.gscale, self.pstep, self.live)          # pstep += live (one tiny launch)
        self.step_count += 1
        t = self.step_count
        b1, b2 = self.betas
        self._step_state = {'sched': sched, 'bc': (1 - b1 ** t, 1 - b2 ** t), 'done': set()}
        return self._step_state

    def _adamw_group(self, g, st, clip_here):
        s, e = self.group_range[g]
        b1, b2 = self.betas
        hip.adamw(self.P[s:e], self.G[s:e], self.M[s:e], self.V[s:e], self.Pb[s:e], e - s, self.lr[g] * st['sched'], b1, b2,
                  self.eps, self.wd, st['bc'][0], st['bc'][1], self.gscale if clip_here else self.avg_t,
                  seg_id=self.seg_id[s // 8:e // 8], seg_live=self.pstep)
        st['done'].add(g)

    def step(self):
        """clip_grad_norm_(detr params) + AdamW + schedule (train_distr.py:423-428,468-469)"""
        hp = HOST_PROF
        t0 = time.perf_counter() if hp is not None else 0.0
        use_clip = self.clip is not None and self.clip > 0
        st = self._begin_step()
        if hp is not None:
            t1 = time.perf_counter(); hp['opt_publish'] = hp.get('opt_publish', 0.0) + t1 - t0; t0 = t1
        # clip factor over the DETR groups (backbone | head: adjacent in the flat buffer; untouched gradients are zero), two launches
        # (gpv_clip_scale).  Every rank must get the SAME bits from the same all-reduced gradient, or the replicas drift apart one
        # ulp of the clip factor per step: the kernel sums in a fixed order (gpv_sumsq's float atomics do not -- found by
        # tests/test_distributed_gpu.py); no host sync.
        rng = [self.group_range[g] for g in ('detr_backbone', 'detr_head') if g in self.group_range] if use_clip else []
        if rng:
            s0, e1 = min(r[0] for r in rng), max(r[1] for r in rng)
            if sum(r[1] - r[0] for r in rng) != e1 - s0:
                raise RuntimeError('FlatTrainer: the DETR groups are not contiguous in the flat buffer')
            # (G = sum over ranks: |sum| against world x max_norm is |average| against max_norm; then the average's factor on top)
            hip.clip_scale(self.G[s0:e1], self.clip * self.world, self._clip_ws, self.gscale)
            if self.world > 1:
                self.gscale.mul_(self.avg)
        if hp is not None:
            t1 = time.perf_counter(); hp['opt_clip'] = hp.get('opt_clip', 0.0) + t1 - t0; t0 = t1
        for g in self.group_range:                                           # one launch per group; the kernel skips dead parameters
            if g not in st['done']:
                self._adamw_group(g, st, bool(rng) and g in ('detr_backbone', 'detr_head'))
        self._step_state = None
        RT.bump_weights(everything=False)
        if hp is not None:
            hp['opt_adamw'] = hp.get('opt_adamw', 0.0) + time.perf_counter() - t0

    # ---- checkpointing (train_distr.py:381-389 saves optimizer.state_dict() + the warm-up scheduler's) ----
    def _torch_param_order(self):
        """the reference's optimizer parameter numbering: four groups (train_distr.py:228-253) filled in
        model.named_parameters() order -- EVERY parameter, trainable or not -- numbered consecutively group after group"""
        groups = {g: [] for g in GROUPS}
        for n, p in self.model.named_parameters():
            groups[param_group_of(n)].append(n)
        order, idx = [], 0
        for g in GROUPS:
            order.append((g, list(range(idx, idx + len(groups[g]))), groups[g]))
            idx += len(groups[g])
        return order

    def state_dict(self):
        """optimizer state in torch.optim.AdamW.state_dict() layout (what the reference's checkpoints hold,
        train_distr.py:381-389): {'state': {param index: {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups': [4 groups]}
        with the reference's parameter numbering; only parameters that have taken a step have state, like torch's.
        Extra keys ('gpv1_amd': names, schedule position) make it self-describing; torch ignores them."""
        by_name = {n: (p, o, k, i) for i, (n, p, g, o, k) in enumerate(self.entries)}
        pstep = self.pstep.cpu()
        state, groups, names = {}, [], {}
        b1, b2 = self.betas
        sched = self.lr_factor()
        for g, ids, ns in self._torch_param_order():
            for pid, n in zip(ids, ns):
                names[pid] = n
                e = by_name.get(n)
                if e is None or int(pstep[e[3]]) == 0:
                    continue
                p, o, k, i = e
                shape = p.shape
                state[pid] = {'step': torch.tensor(float(pstep[i])),
                              'exp_avg': self._view(self.M, p, o, k).detach().cpu().clone().contiguous().view(shape),
                              'exp_avg_sq': self._view(self.V, p, o, k).detach().cpu().clone().contiguous().view(shape)}
            groups.append({'params': ids, 'lr': self.lr[g] * sched, 'initial_lr': self.lr[g], 'betas': (b1, b2), 'eps': self.eps,
                           'weight_decay': self.wd, 'amsgrad': False, 'maximize': False, 'foreach': None, 'capturable': False,
                           'differentiable': False, 'fused': None})
        return {'state': state, 'param_groups': groups,
                'gpv1_amd': {'names': names, 'step': self.step_count, 'warmup_steps': self.warmup_steps, 't_total': self.t_total,
                             'epoch': self.epoch}}

    def load_state_dict(self, sd, step=None):
        """accepts a torch.optim.AdamW state dict with the reference's parameter numbering (a reference checkpoint's
        'optimizer', or state_dict() above) and the name-keyed format this trainer wrote in round 1.  `step`: the global
        step of the checkpoint (ckpt['step'], train_distr.py:275) when the dict itself does not carry it."""
        if 'param_groups' not in sd:                                          # round-1 format: {'state': {name: ...}, 'step': ...}
            self.step_count = int(sd['step'])
            pstep = torch.zeros(len(self.entries), dtype=torch.int32)
            for i, (n, p, g, o, k) in enumerate(self.entries):
                e = sd['state'].get(n)
                if e is None or e['exp_avg'].numel() != k:
                    continue
                self.M[o:o + k].copy_(e['exp_avg'].reshape(-1))
                self.V[o:o + k].copy_(e['exp_avg_sq'].reshape(-1))
                if e.get('touched'):
                    self.touched[i] = True
                    pstep[i] = self.step_count
            self.pstep.copy_(pstep)
            self._publish_touched()
            return
        extra = sd.get('gpv1_amd', {})
        self.step_count = int(extra.get('step', step if step is not None else 0))
        if 'epoch' in extra:
            self.epoch = int(extra['epoch'])
        by_name = {n: (p, o, k, i) for i, (n, p, g, o, k) in enumerate(self.entries)}
        pstep = torch.zeros(len(self.entries), dtype=torch.int32)
        taken = 0
        for g, ids, ns in self._torch_param_order():
            for pid, n in zip(ids, ns):
                st = sd['state'].get(pid, sd['state'].get(str(pid)))
                e = by_name.get(n)
                if st is None or e is None:
                    continue
                p, o, k, i = e
                if st['exp_avg'].numel() != k:
                    continue
                self._view(self.M, p, o, k).copy_(st['exp_avg'].view(p.shape))
                self._view(self.V, p, o, k).copy_(st['exp_avg_sq'].view(p.shape))
                pstep[i] = int(float(st['step']))
                self.touched[i] = True
                taken += 1
        self.pstep.copy_(pstep)
        self._publish_touched()
        return taken

    def lr_factor(self):
        """multiplier on every group's base learning rate for the NEXT optimizer step.
        lr_linear_decay (configs/exp/gpv.yaml:142, the shipped setting): WarmupLinearSchedule stepped per iteration
        (train_distr.py:298-302,468-469).  Otherwise (train_distr.py:288-292,303-308,470-474): MultiStepLR(lr_milestones,
        lr_drop) stepped per EPOCH times GradualWarmupScheduler(multiplier=1, total_epoch=iterations of epoch 0) during epoch 0."""
        if self.t_total > 0:
            return warmup_linear(self.step_count, self.warmup_steps, self.t_total)
        f = 1.0
        if self.lr_milestones is not None:
            f = self.lr_drop ** sum(1 for m in self.lr_milestones if self.epoch >= m)
        if self.warmup_iters > 0 and self.epoch == 0 and self.it_in_epoch < self.warmup_iters:
            f *= float(self.it_in_epoch) / float(self.warmup_iters)
        return f

    def set_epoch(self, epoch, it_in_epoch=0):
        """position inside the run for the MultiStepLR / GradualWarmup schedule (train_distr.train_worker calls it per iteration)"""
        self.epoch, self.it_in_epoch = int(epoch), int(it_in_epoch)

    def current_lrs(self):
        sched = self.lr_factor()
        return {g: lr * sched for g, lr in self.lr.items()}

    def train_step(self, images, queries, targets):
        """one iteration of train_distr.py:399-428; returns the (detached) loss tensor, or None: no applicable target.

        With graphs on, the whole step -- eager warm-up steps, captures, replays, optimizer -- runs on ONE dedicated side
        stream: hipGraph capture needs a non-default stream, and autograd's AccumulateGrad nodes remember the stream they
        were created on (a node made on the default stream by an earlier eager step and kept alive by a stashed loss would
        make the captured backward synchronise with the legacy stream, which is illegal during capture)."""
        if self.stream is None:
            return self._train_step_impl(images, queries, targets)
        cur = torch.cuda.current_stream(self.stream.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            loss = self._train_step_impl(images, queries, targets)
        cur.wait_stream(self.stream)
        return loss

    def _train_step_impl(self, images, queries, targets):
        model = self.model
        if self.manual_gc:
            if not self._gc_armed:
                gc.collect(); gc.freeze(); gc.disable()
                self._gc_armed = True
            elif self.step_count % self.gc_interval == self.gc_interval - 1:
                gc.collect()
        if not model.training:                      # nn.Module.train() walks ~600 modules (0.66 ms): only when needed
            model.train()
        lang_extra, orig_queries = None, queries
        cls = self._classed(images, queries, targets)
        if cls is not None:
            # hipGraph path: string queries are tokenised here on the host; query and answer token axes are padded to a few size
            # classes with the padding masked out exactly (GPV._forward_impl `lang_extra`, CE target -100), so one captured body
            # serves every batch of its class with the unpadded batch's result (up to kernel summation order: _classed)
            queries, lang_extra, answer_token_ids, ce_targets = cls
            for i, t in enumerate(targets):
                t['answer_token_ids'] = ce_targets[i]
            body = self._graphed_body(images, queries, answer_token_ids, lang_extra)
            if body is not None:
                self.graph_steps += 1
                return self._train_step_graphed(body, images, queries, answer_token_ids, targets, lang_extra, orig_queries, ce_targets)
        else:
            _, answer_token_ids = model.encode_answers(targets)
            for i, t in enumerate(targets):
                t['answer_token_ids'] = answer_token_ids[i, 1:]
        self.eager_steps += 1
        loss = model._forward_impl(images, queries, answer_token_ids, targets, lang_extra=lang_extra)
        # The reference skips the update when no criterion applies to the batch (losses.py:163-169, train_distr.py:420); under
        # its DDP a rank-local skip leaves the other ranks waiting in the gradient all-reduce forever.  Here the ranks agree on
        # the host (one int through gloo, issued while the GPU still runs the forward): nobody has a loss -> everyone skips
        # like the reference; somebody has -> the ranks without one contribute zero gradients, enter every collective and step.
        if not self._any_rank_has_loss(loss is not None):
            return None
        self.zero_grad()
        self.begin_backward()
        if loss is not None:
            loss.backward()
        ops_check_chains()
        self.allreduce_grads()
        self.step()
        return None if loss is None else loss.detach()     # (a stashed loss would keep the step's autograd graph alive)

    # ---- hipGraph path ----
    # size classes of the token axes: query tokens in multiples of 8 (device-tensor queries of <= 8 tokens are taken as they are),
    # answer tokens in multiples of 8 up to max_text_len -- six signatures cover T_l <= 16 x S <= 20; every signature costs one
    # eager warm-up step and one capture, every captured body pins its activation pool
    T_EXACT, T_STEP, S_STEP = 8, 8, 8

    def _graphs_apply(self, images):
        from .misc import NestedTensor
        return bool(self.graphs) and isinstance(images, NestedTensor) and torch.is_tensor(images.tensors) and images.tensors.is_cuda \
            and images.mask is not None and RT.dtype == torch.bfloat16 and not torch.cuda.is_current_stream_capturing()

    def _classed(self, images, queries, targets):
        """-> ((ids, attn) [B, T_c], lang_extra uint8 [B, T_c], answer ids [B, S_c], CE targets [B, S_c - 1]) on the device, or None
        when the graph path does not apply (then the step runs as the reference does, on the batch's own lengths).
        The reference pads a batch to ITS longest query / answer (bert.py:12-15 padding=True, gpv.py:377-430) and both lengths
        enter the numerics (padded BERT tokens are attended by the co-attention, pad answer tokens are CE targets), so a captured
        graph is only valid for one (T, S).  Real batches vary in both: the token axes are therefore padded on to a few size
        classes and the EXTRA positions are masked out exactly -- query tokens beyond the batch's longest as attention keys,
        answer positions beyond the batch's longest as CE rows -- which leaves loss and gradients those of the unpadded batch in exact
        arithmetic: the extra positions contribute exact zeros.  Bit-for-bit it is the unpadded batch only where both row counts take the
        same kernels: B x T_b and B x T_c rows can fall on different sides of a row-count dispatch rule (small-M GEMM families, the
        projection + LayerNorm launch), whose k-summation orders differ -- measured with tools/fuzz_model.py (FUZZ_ONLY=1: 9 tokens -> 16,
        27 -> 48 language rows): parameters 1e-8 .. 4e-6 apart after one step against the unpadded eager step, loss 5.6e-4 apart after four
        (Adam turns noise-level gradient differences into lr-sized steps); exactly 0 with the queries taken unpadded (FUZZ_T_EXACT=16) and
        for every case of <= 8 tokens."""
        if not self._graphs_apply(images):
            return None
        from .misc import STAGER
        model, dev = self.model, images.tensors.device
        if isinstance(queries, (tuple, list)) and len(queries) == 2 and all(torch.is_tensor(q) for q in queries):
            ids, attn = queries
            if ids.is_cuda and ids.shape[1] <= self.T_EXACT:
                ids_c, attn_c = ids, attn                                           # device tensors of an exact class: as they are
                Tb = Tc = ids.shape[1]
            else:
                ids, attn = ids.cpu(), attn.cpu()
                ids_c = None
        else:
            tok = getattr(model.bert, 'tokenizer', None)
            if tok is None:
                return None                                                         # (Bert.forward raises with the reason)
            ids, attn = tok(list(queries))
            ids_c = None
        if ids_c is None:
            Tb = ids.shape[1]
            Tc = -(-Tb // self.T_STEP) * self.T_STEP
            pi = torch.zeros(ids.shape[0], Tc, dtype=torch.long)
            pa = torch.zeros(ids.shape[0], Tc, dtype=torch.long)
            pi[:, :Tb], pa[:, :Tb] = ids, attn
            ids_c, attn_c = STAGER.to_device(pi, torch.long, dev), STAGER.to_device(pa, torch.long, dev)
        key = (dev, ids_c.shape[0], Tc, Tb)
        extra = self._extra_cache.get(key)
        if extra is None:
            e = torch.zeros(ids_c.shape[0], Tc, dtype=torch.uint8)
            e[:, Tb:] = 1
            extra = self._extra_cache[key] = e.to(dev)
        tok_c, tgt_c, _ = model.encode_answers_classed(targets, self.S_STEP)
        return (ids_c, attn_c), extra, tok_c, tgt_c

    def _graphed_body(self, images, queries, tok, lang_extra=None):
        """the GraphedBody for this batch signature, captured the second time the signature is seen (the first, eager,
        step is the warm-up: weight copies, kernel attributes, workspaces); None -> eager step.  At most `graph_slots` bodies
        are kept (each pins its activation pool): the least recently used one makes room."""
        if not self._graphs_apply(images) or not isinstance(queries, (tuple, list)) or len(queries) != 2 \
                or not all(torch.is_tensor(q) for q in queries):
            return None
        key = (tuple(images.tensors.shape), images.tensors.dtype, getattr(images, 'all_valid', None), tuple(queries[0].shape),
               tuple(tok.shape), lang_extra is not None)
        body = self._bodies.get(key)
        if body is not None and body.stale():
            del self._bodies[key]
            body = None
        if body is None:
            n = self._seen.get(key, 0)
            self._seen[key] = n + 1
            if n < 1:
                return None
            if len(self._bodies) >= self.graph_slots:
                # More live signatures than slots: an eviction costs a four-graph capture (+ a cache flush; round 4: + 0.7 s of collective
                # quiescing with several ranks -- round 5: none in steady state, misc.CollectiveClock).  At most one per `evict_interval` steps -- a stream that cycles through more
                # signatures than there are slots runs its misses eagerly instead of recapturing on every step.
                last = getattr(self, '_last_evict', None)
                if last is not None and self.step_count - last < self.evict_interval:
                    return None
                self._last_evict = self.step_count
                self._bodies.popitem(last=False)                                   # least recently used
                gc.collect()
                if self.comm:
                    from .misc import quiesce_collectives
                    quiesce_collectives()                                          # (no cache flush under collectives in flight)
                torch.cuda.empty_cache()
            try:
                body = self._bodies[key] = GraphedBody(self, images, queries, tok, lang_extra)
            except RuntimeError as err:
                self._graphs_failed(err)
                return None
        else:
            self._bodies.move_to_end(key)
        return body

    def _graphs_failed(self, err):
        """A capture or replay failed.  On one GPU that is a bug and is raised; with several ranks (a path no single-GPU box
        can exercise under RCCL) the trainer drops to eager steps for the rest of the run instead of losing the job."""
        if self.world == 1 or os.environ.get('GPV_GRAPHS_STRICT', '0') == '1':
            raise err
        self.disable_graphs('%s: %s' % (type(err).__name__, str(err).splitlines()[0] if str(err) else ''))

    def disable_graphs(self, why):
        """eager steps from here on (every rank must do this at the same step: the eager and the graphed step enter the same
        collectives, so a mixed job stays correct, but its timing is the eager ranks')"""
        import sys
        print('[gpv1_amd] rank %d: hipGraph path switched off (%s) -- continuing with eager steps' % (self.rank, why), file=sys.stderr, flush=True)
        self.graphs = False
        self.graphs_off_reason = why
        self._bodies.clear()
        RT.split = None
        RT.defer_list = None
        RT.backward_boundary = None

    def _train_step_graphed(self, body, images, queries, tok, targets, lang_extra=None, orig_queries=None, ce_targets=None):
        hp = HOST_PROF
        t0 = time.perf_counter() if hp is not None else 0.0
        fused = GraphedBody.fused_task(self.model.criterion, targets) if ce_targets is not None else None
        try:
            outs = body.forward(images, queries, tok, lang_extra, leaves=fused is None)
            if hp is not None:
                t1 = time.perf_counter(); hp['replay_f1_f2'] = hp.get('replay_f1_f2', 0.0) + t1 - t0; t0 = t1
            loss = self.model.criterion(outs, targets)[0] if fused is None else True
        except RuntimeError as err:                       # nothing collective has been entered yet: redo the step eagerly
            self._graphs_failed(err)
            return self._train_step_impl(images, queries if orig_queries is None else orig_queries, targets)
        if not self._any_rank_has_loss(loss is not None):
            return None
        if not body.zero_in_graph:
            self.zero_grad()
        self.begin_backward()
        if hp is not None:
            t1 = time.perf_counter(); hp['criterion_zero'] = hp.get('criterion_zero', 0.0) + t1 - t0; t0 = t1
        if loss is not None:
            try:
                if fused is not None:
                    loss = body.backward_fused(fused, ce_targets).clone()        # (the static scalar is rewritten by the next replay)
                else:
                    loss.backward()
                    if hp is not None:
                        t1 = time.perf_counter(); hp['criterion_backward'] = hp.get('criterion_backward', 0.0) + t1 - t0; t0 = t1
                    body.backward(outs)
                if hp is not None:
                    t1 = time.perf_counter(); hp['replay_b1_b2'] = hp.get('replay_b1_b2', 0.0) + t1 - t0; t0 = t1
            except RuntimeError as err:                   # the ranks already agreed to step: redo this rank's part eagerly
                self._graphs_failed(err)
                torch.cuda.synchronize()
                self.zero_grad()
                self.begin_backward()
                loss = self.model._forward_impl(images, queries, tok, targets, lang_extra=lang_extra)
                if loss is not None:
                    loss.backward()
        self.allreduce_grads()
        self.step()
        if hp is not None:
            hp['optimizer'] = hp.get('optimizer', 0.0) + time.perf_counter() - t0
            hp['steps'] = hp.get('steps', 0) + 1
        return None if loss is None else loss.detach()

    def _any_rank_has_loss(self, has):
        if not self.comm:
            return has
        t = torch.tensor([1 if has else 0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.host_pg)
        return bool(int(t))
