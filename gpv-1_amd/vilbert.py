"""Co-attention layer (the live part of the reference's ViLBERT copy: exp/gpv/models/vilbert.py
BertConnectionLayer :859-900, BertBiAttention :696-824, BertBiOutput :827-856,
Bert(Image)Intermediate/Output :488-516,649-677, erf-GELU :111-117, BertLayerNorm eps 1e-12).

Stream 1 = language tokens, stream 2 = vision tokens (gpv.py:149-154).  The six Q/K/V projections are
GEMMs (per-stream fusion is a follow-up); both attention directions run on the same MFMA attention
kernel (dh = 48, K-dim padded to 64); residual+dropout+LayerNorm is one kernel, GELU a second pass
that keeps the pre-activation for backward.  Parameter names match the reference so checkpoints load,
including the declared-but-unused q_dense1/q_dense2 (vilbert.py:835-843).
"""
import os

import torch
import torch.nn as nn

from . import ops
from .ops import W
from .transformer import LinearP, LayerNormP

FUSE_QKV = os.environ.get('GPV_COATT_QKV', '1') != '0'


class BertBiAttention(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        if cfg.bi_hidden_size % cfg.bi_num_attention_heads != 0:
            raise ValueError('bi_hidden_size must be a multiple of bi_num_attention_heads')
        self.num_attention_heads = cfg.bi_num_attention_heads
        self.attention_head_size = cfg.bi_hidden_size // cfg.bi_num_attention_heads
        self.all_head_size = cfg.bi_hidden_size
        self.query1 = LinearP(cfg.v_hidden_size, self.all_head_size)
        self.key1 = LinearP(cfg.v_hidden_size, self.all_head_size)
        self.value1 = LinearP(cfg.v_hidden_size, self.all_head_size)
        self.query2 = LinearP(cfg.hidden_size, self.all_head_size)
        self.key2 = LinearP(cfg.hidden_size, self.all_head_size)
        self.value2 = LinearP(cfg.hidden_size, self.all_head_size)
        self.p1 = cfg.v_attention_probs_dropout_prob
        self.p2 = cfg.attention_probs_dropout_prob

    def forward(self, t1, t2, B, T1, T2, own1=None, own2=None, kpm1=None, branch=None):
        """t1 [B*T1, D] (language), t2 [B*T2, D] (vision) -> ctx1 [B*T2, D], ctx2 [B*T1, D].
        ops.GradChain: a chain's members must all sit behind the same layer output (a detection-only batch gives the language
        output of the last layer no gradient).  ctx2 -> stream 1's output uses q1, k2, v2; ctx1 -> stream 2's output uses q2, k1, v1:
        own1 / own2 = the chain query_i shares with stream i's residual LayerNorm; k_i, v_i form a chain of their own."""
        H, dh, D = self.num_attention_heads, self.attention_head_size, self.all_head_size
        p1 = self.p1 if self.training else 0.0
        p2 = self.p2 if self.training else 0.0
        if FUSE_QKV and t1.shape[-1] == t2.shape[-1]:
            # q | k | v of a stream as ONE GEMM over the three concatenated weights (ops.multi_linear: 6 projection launches -> 2, and
            # 2 backward-data GEMMs with K = 3 D instead of 6); both attention calls read column slices of the two buffers and
            # write their gradient columns into shared buffers (ops.GradSlots: either call may be absent from a backward pass)
            grad = torch.is_grad_enabled() and (t1.requires_grad or t2.requires_grad)
            s1, s2 = (ops.GradSlots(), ops.GradSlots()) if grad else (None, None)
            if branch is not None:                      # the language stream's projection on the side stream, beside the vision stream's
                branch.fork()
                with branch.on():
                    qkv1 = ops.multi_linear(t1, [W(m.weight, m.bias) for m in (self.query1, self.key1, self.value1)], s1)
            else:
                qkv1 = ops.multi_linear(t1, [W(m.weight, m.bias) for m in (self.query1, self.key1, self.value1)], s1)
            # (the vision stream's output always carries a gradient: its projections share the chain of its residual LayerNorm;
            #  the language stream's may not -- a detection-only batch in the last layer -- so its sum is left to autograd)
            qkv2 = ops.multi_linear(t2, [W(m.weight, m.bias) for m in (self.query2, self.key2, self.value2)], s2, chain=own2)
            if branch is not None:
                branch.join()                           # both attention calls read both buffers: they run on the main stream
            roles = ((0, 0), (1, D), (1, 2 * D))
            ctx1 = ops.attention([qkv2, qkv1], roles, B, H, T2, T1, dh, kpm=kpm1, drop_p=p1, sinks=(s2, s1) if grad else None)
            ctx2 = ops.attention([qkv1, qkv2], roles, B, H, T1, T2, dh, drop_p=p2, sinks=(s1, s2) if grad else None)
            return ctx1, ctx2
        x1, x2 = ops.grad_chain(t1), ops.grad_chain(t2)
        q1, k1, v1 = self.query1(t1, chain=own1), self.key1(t1, chain=x1), self.value1(t1, chain=x1)
        q2, k2, v2 = self.query2(t2, chain=own2), self.key2(t2, chain=x2), self.value2(t2, chain=x2)
        # scores1 = q2 k1^T -> probs (dropout1) @ v1 : vision queries over language keys (vilbert.py:770-787)
        ctx1 = ops.attention([q2, k1, v1], ((0, 0), (1, 0), (2, 0)), B, H, T2, T1, dh, kpm=kpm1, drop_p=p1)
        # scores2 = q1 k2^T -> probs (dropout2) @ v2 : language queries over vision keys (:790-810)
        ctx2 = ops.attention([q1, k2, v2], ((0, 0), (1, 0), (2, 0)), B, H, T1, T2, dh, drop_p=p2)
        return ctx1, ctx2


class BertBiOutput(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.dense1 = LinearP(cfg.bi_hidden_size, cfg.v_hidden_size)
        self.LayerNorm1 = LayerNormP(cfg.v_hidden_size, eps=1e-12)
        self.q_dense1 = LinearP(cfg.bi_hidden_size, cfg.v_hidden_size)     # unused in forward (as in the reference)
        self.dense2 = LinearP(cfg.bi_hidden_size, cfg.hidden_size)
        self.LayerNorm2 = LayerNormP(cfg.hidden_size, eps=1e-12)
        self.q_dense2 = LinearP(cfg.bi_hidden_size, cfg.hidden_size)       # unused in forward
        self.p1, self.p2 = cfg.v_hidden_dropout_prob, cfg.hidden_dropout_prob

    def forward(self, hidden1, input1, hidden2, input2, c1=None, c2=None):
        p1 = self.p1 if self.training else 0.0
        p2 = self.p2 if self.training else 0.0
        return (self.LayerNorm1(input1, self.dense1(hidden1), p1, chain=c1),
                self.LayerNorm2(input2, self.dense2(hidden2), p2, chain=c2))


class _Intermediate(nn.Module):
    def __init__(self, hidden, inter, act):
        super().__init__()
        if act != 'gelu':
            raise NotImplementedError('GPV-1 configs use hidden_act: gelu')
        self.dense = LinearP(hidden, inter)

    def forward(self, x, chain=None):
        return self.dense(x, ops.ACT_GELU, chain=chain)


class _Output(nn.Module):
    def __init__(self, inter, hidden, p):
        super().__init__()
        self.dense = LinearP(inter, hidden)
        self.LayerNorm = LayerNormP(hidden, eps=1e-12)
        self.p = p

    def forward(self, h, inp, chain=None):
        return self.LayerNorm(inp, self.dense(h), self.p if self.training else 0.0, chain=chain)


class BertConnectionLayer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.biattention = BertBiAttention(cfg)
        self.biOutput = BertBiOutput(cfg)
        self.v_intermediate = _Intermediate(cfg.v_hidden_size, cfg.v_intermediate_size, cfg.v_hidden_act)
        self.v_output = _Output(cfg.v_intermediate_size, cfg.v_hidden_size, cfg.v_hidden_dropout_prob)
        self.t_intermediate = _Intermediate(cfg.hidden_size, cfg.intermediate_size, cfg.hidden_act)
        self.t_output = _Output(cfg.intermediate_size, cfg.hidden_size, cfg.hidden_dropout_prob)

    def forward(self, t1, t2, B, T1, T2, kpm1=None):
        """vilbert.py:872-900.  No attention masks: GPV passes None, so padded BERT tokens are attended.
        kpm1 (uint8 [B, T1], 1 = ignore): NOT a reference argument -- the trainer's size-classed batches (train.FlatTrainer) carry
        language tokens beyond the batch's own longest query; masking exactly those keys reproduces the unpadded batch (up to the
        summation order of the kernels the two row counts dispatch to: train.FlatTrainer._classed)."""
        c1, c2 = ops.grad_chain(t1), ops.grad_chain(t2)          # each stream input: three projections + a residual
        # round 6: stream 1 (language: B * T1 = 192 rows at B = 32) runs on a side stream / graph branch beside stream 2 (vision: 3200 rows)
        # wherever the two are independent -- its q | k | v projection, and everything behind the two attention calls (ops.Branch)
        br = ops.branch_for(t1) if (FUSE_QKV and t1.shape[-1] == t2.shape[-1]) else None
        bi1, bi2 = self.biattention(t1, t2, B, T1, T2, c1, c2, kpm1, branch=br)
        bo = self.biOutput
        if br is None:
            a1, a2 = bo(bi2, t1, bi1, t2, c1, c2)
            ca1, ca2 = ops.grad_chain(a1), ops.grad_chain(a2)    # feed-forward input + residual
            o1 = self.v_output(self.v_intermediate(a1, ca1), a1, ca1)
            o2 = self.t_output(self.t_intermediate(a2, ca2), a2, ca2)
            return o1, o2
        # (the launches are ISSUED in the in-line order -- stream 1's, stream 2's, stream 1's, stream 2's -- so that the dropout seeds are
        #  drawn in the same order with and without the branch: same masks; what remains between the two runs is the run-to-run spread of the
        #  grouped weight gradients' atomics, tests/test_model_gpu.py::test_coattention_language_branch_changes_nothing)
        br.fork()
        with br.on():
            a1 = bo.LayerNorm1(t1, bo.dense1(bi2), bo.p1 if bo.training else 0.0, chain=c1)
        a2 = bo.LayerNorm2(t2, bo.dense2(bi1), bo.p2 if bo.training else 0.0, chain=c2)
        ca1, ca2 = ops.grad_chain(a1), ops.grad_chain(a2)
        with br.on():
            o1 = self.v_output(self.v_intermediate(a1, ca1), a1, ca1)
        o2 = self.t_output(self.t_intermediate(a2, ca2), a2, ca2)
        br.join()
        return o1, o2
