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
import gc
import os

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


class FlatTrainer:
    def __init__(self, model, lr=1e-4, lr_backbone=1e-5, weight_decay=1e-4, clip_max_norm=0.1, warmup_steps=0,
                 t_total=0, betas=(0.9, 0.999), eps=1e-8, bucket_mb=128, process_group=None, manual_gc=True, gc_interval=200):
        self.model = model
        # Python's cyclic collector costs 1-3 ms per step once it has a few hundred thousand module / tensor objects to
        # walk (measured: 970-1020 vs 1062-1070 images/s over 20 steps), while a step leaves ~17 small cycles and no
        # device memory behind (tools/gc_growth.py).  With manual_gc the trainer freezes the long-lived objects, turns
        # the automatic collector off and collects every gc_interval steps itself (Megatron-style).
        self.manual_gc, self.gc_interval, self._gc_armed = manual_gc, gc_interval, False
        self.lr = {'detr_backbone': lr_backbone, 'detr_head': lr, 'bert': lr, 'others': lr}
        self.wd, self.clip, self.betas, self.eps = weight_decay, clip_max_norm, betas, eps
        self.warmup_steps, self.t_total = warmup_steps, t_total
        self.step_count = 0
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad and not any(s in n for s in NEVER_GRAD)]
        named.sort(key=lambda np_: GROUPS.index(param_group_of(np_[0])))           # stable: module order inside a group
        self.entries = []                                                           # (name, param, group, offset, numel)
        off = 0
        for n, p in named:
            self.entries.append((n, p, param_group_of(n), off, p.numel()))
            off += (p.numel() + 7) // 8 * 8                                          # keep every segment 32-B aligned
        self.total = off
        dev = named[0][1].device
        self.P = torch.zeros(off, device=dev, dtype=torch.float32)
        self.G = torch.zeros_like(self.P)
        self.M = torch.zeros_like(self.P)
        self.V = torch.zeros_like(self.P)
        self.Pb = torch.zeros(off, device=dev, dtype=torch.bfloat16)      # bf16 mirror of P, rewritten by the AdamW kernel
        self.touched = torch.zeros(len(self.entries), dtype=torch.bool)           # host: which parameters THIS rank's kernels wrote
        # device: which parameters any rank has ever written (torch-1.6 optimizers skip the rest).  The AdamW kernel reads
        # it through seg_id (parameter index of every 8-element chunk), so no host round trip is needed to pick ranges.
        self.live = torch.zeros(len(self.entries), device=dev, dtype=torch.int32)
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
        self.gsq = torch.zeros(1, device=dev, dtype=torch.float32)
        self.gscale = torch.ones(1, device=dev, dtype=torch.float32)
        b = bucket_mb * 1024 * 1024 // 4
        self.backbone_end = max([o + (k + 7) // 8 * 8 for (n, p, g, o, k) in self.entries if g == 'detr_backbone'], default=0)
        self.buckets = [(s, min(self.backbone_end, s + b)) for s in range(0, self.backbone_end, b)] + \
                       [(s, min(off, s + b)) for s in range(self.backbone_end, off, b)]      # no bucket straddles the backbone boundary
        self.overlap = self.world > 1 and os.environ.get('GPV_OVERLAP', '1') != '0'
        self.dry_overlap = False             # tests: run the milestone / guard logic without communicating
        self._closed_from, self._works, self.late_touch, self.milestones = None, [], None, 0
        self.host_pg = None
        if self.world > 1:
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
        if what != 'backbone' or self._closed_from is not None or not (self.overlap or self.dry_overlap):
            return
        self._closed_from = self.backbone_end
        self.milestones += 1
        if self.overlap:
            self._works = [dist.all_reduce(self.G[s:e], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
                           for s, e in self.buckets if s >= self.backbone_end]

    def begin_backward(self):
        self._closed_from, self._works, self.late_touch, self.milestones = None, [], None, 0
        RT.backward_milestone = self._on_milestone

    def allreduce_grads(self):
        RT.backward_milestone = None
        closed = self._closed_from
        self._closed_from = None
        if self.world == 1:
            return
        # Every rank issues the buckets in the SAME order -- [behind the backbone segment] then [backbone segment] -- whether or
        # not its backward reached the milestone (a rank without an applicable target runs no backward at all, see train_step).
        works = list(self._works)
        if closed is None:
            works += [dist.all_reduce(self.G[s:e], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
                      for s, e in self.buckets if s >= self.backbone_end]
        works += [dist.all_reduce(self.G[s:e], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
                  for s, e in self.buckets if s < self.backbone_end]
        self._works = []
        self._publish_touched()
        dist.all_reduce(self.live, op=dist.ReduceOp.MAX, group=self.pg)          # stays on the device: no host sync
        for w in works:
            w.wait()
        self.G.mul_(1.0 / self.world)

    def _publish_touched(self):
        """host-side 'touched' marks of this step -> device flags (idempotent; pinned staging, asynchronous)"""
        from .misc import STAGER
        loc = STAGER.to_device(self.touched.to(torch.int32), torch.int32, self.live.device)
        torch.maximum(self.live, loc, out=self.live)

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

    def step(self):
        """clip_grad_norm_(detr params) + AdamW + schedule (train_distr.py:423-428,468-469)"""
        sched = warmup_linear(self.step_count, self.warmup_steps, self.t_total) if self.t_total > 0 else 1.0
        use_clip = self.clip is not None and self.clip > 0
        self._publish_touched()
        if use_clip:
            # ||g||^2 over the DETR groups (untouched gradients are zero: whole groups).  Every rank must get the SAME bits
            # from the same all-reduced gradient, or the replicas drift apart one ulp of the clip factor per step:
            # torch's norm is a fixed-order tree reduction; gpv_sumsq accumulates its blocks with float atomics, whose
            # order differs from run to run (found by tests/test_distributed_gpu.py).
            self.gsq.zero_()
            for g in ('detr_backbone', 'detr_head'):
                if g in self.group_range:
                    s, e = self.group_range[g]
                    n = torch.linalg.vector_norm(self.G[s:e])
                    self.gsq.addcmul_(n, n)
            # scale = min(1, max_norm / (norm + 1e-6)) on device, no host sync
            torch.clamp(self.clip / (self.gsq.sqrt() + 1e-6), max=1.0, out=self.gscale)
        self.step_count += 1
        t = self.step_count
        b1, b2 = self.betas
        bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
        for g, (s, e) in self.group_range.items():                           # one launch per group; the kernel skips dead parameters
            clip_here = use_clip and g in ('detr_backbone', 'detr_head')
            hip.adamw(self.P[s:e], self.G[s:e], self.M[s:e], self.V[s:e], self.Pb[s:e], e - s, self.lr[g] * sched, b1, b2,
                      self.eps, self.wd, bc1, bc2, self.gscale if clip_here else None,
                      seg_id=self.seg_id[s // 8:e // 8], seg_live=self.live)
        RT.bump_weights(everything=False)

    # ---- checkpointing (train_distr.py:381-389 saves optimizer.state_dict() + the warm-up scheduler's) ----
    def state_dict(self):
        """optimizer + schedule state, keyed by parameter NAME so that it survives a different flattening order"""
        st = {}
        live = self.live_host()
        for i, (n, p, g, o, k) in enumerate(self.entries):
            st[n] = {'exp_avg': self.M[o:o + k].detach().cpu().clone(), 'exp_avg_sq': self.V[o:o + k].detach().cpu().clone(),
                     'touched': bool(live[i])}
        return {'state': st, 'step': self.step_count, 'warmup_steps': self.warmup_steps, 't_total': self.t_total,
                'lr': dict(self.lr), 'weight_decay': self.wd, 'betas': tuple(self.betas), 'eps': self.eps}

    def load_state_dict(self, sd):
        self.step_count = int(sd['step'])
        for i, (n, p, g, o, k) in enumerate(self.entries):
            e = sd['state'].get(n)
            if e is None or e['exp_avg'].numel() != k:
                continue
            self.M[o:o + k].copy_(e['exp_avg'])
            self.V[o:o + k].copy_(e['exp_avg_sq'])
            self.touched[i] = bool(e['touched'])
        self._publish_touched()

    def current_lrs(self):
        sched = warmup_linear(self.step_count, self.warmup_steps, self.t_total) if self.t_total > 0 else 1.0
        return {g: lr * sched for g, lr in self.lr.items()}

    def train_step(self, images, queries, targets):
        """one iteration of train_distr.py:399-428; returns the loss tensor (or None: no applicable target)"""
        model = self.model
        if self.manual_gc:
            if not self._gc_armed:
                gc.collect(); gc.freeze(); gc.disable()
                self._gc_armed = True
            elif self.step_count % self.gc_interval == self.gc_interval - 1:
                gc.collect()
        if not model.training:                      # nn.Module.train() walks ~600 modules (0.66 ms): only when needed
            model.train()
        _, answer_token_ids = model.encode_answers(targets)
        for i, t in enumerate(targets):
            t['answer_token_ids'] = answer_token_ids[i, 1:]
        loss = model(images, queries, answer_token_ids, targets)
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
        self.allreduce_grads()
        self.step()
        return loss

    def _any_rank_has_loss(self, has):
        if self.world == 1:
            return has
        t = torch.tensor([1 if has else 0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.host_pg)
        return bool(int(t))
