"""DETR transformer on the HIP kernels (reference: exp/gpv/models/transformer.py).

Same parameter names as the reference (``self_attn.in_proj_weight``, ``out_proj``, ``linear1`` ...), post-norm
layers only (configs/exp/gpv.yaml: pre_norm False).  Internally batch-first: activations are
[B*S, C] row matrices in the compute dtype; every block is
    projection GEMM(s) -> attention kernel -> out-proj GEMM -> fused (residual + dropout + LayerNorm)
    -> FFN GEMM (bias+ReLU+dropout epilogue) -> GEMM -> fused (residual + dropout + LayerNorm).
"""
import os

import torch
import torch.nn as nn

from . import ops
from .ops import W

HOIST_KV = os.environ.get('GPV_HOIST_KV', '1') != '0'       # decoder cross-attention K / V of all layers as two GEMMs (ops.multi_linear)


class LinearP(nn.Module):
    """nn.Linear parameters (same init) evaluated by the GEMM kernel."""

    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        ref = nn.Linear(in_features, out_features, bias=bias)
        self.weight = nn.Parameter(ref.weight.detach().clone())
        self.bias = nn.Parameter(ref.bias.detach().clone()) if bias else None
        self.in_features, self.out_features = in_features, out_features

    def forward(self, x, act=ops.ACT_NONE, drop_p=0.0, out_f32=False, chain=None):
        return ops.linear(x, W(self.weight, self.bias), act, drop_p, out_f32, chain)


def ffn_block(x, l1, l2, norm, drop_p, pos=None, pos_param=None):
    """norm(x + dropout(linear2(dropout(relu(linear1(x)))))) as one autograd node (ops.FFNBlockFn); pos: -> (out, out + pos)"""
    return ops.ffn_block(x, W(l1.weight, l1.bias), W(l2.weight, l2.bias), norm.weight, norm.bias, norm.eps, drop_p, pos, pos_param)


class LayerNormP(nn.Module):
    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))
        self.eps = eps

    def forward(self, x, s=None, drop_p=0.0, chain=None, pos=None, pos_param=None):
        """LayerNorm(x + dropout(s)); chain: the ops.GradChain of x (its gradient is handed to x's other consumers);
        pos (rows broadcast): -> (y, y + pos) from one launch, pos_param: the learned parameter behind pos (gradient sink)"""
        return ops.add_layernorm(x, s, self.weight, self.bias, self.eps, drop_p, chain, pos, pos_param)

    def proj(self, x, a, lin, drop_p=0.0, chain=None, pos=None, pos_param=None):
        """LayerNorm(x + dropout(lin(a))): the projection inside the LayerNorm launch where the kernel takes it (width 256, bf16)"""
        w = W(lin.weight, lin.bias)
        # (GPV_LN_POS=0 -- the timing A/B of the fused position sum -- must switch THIS site too: the two-launch path goes through
        #  add_layernorm's handling of the switch and its guard)
        if ops.proj_layernorm_ok(w, x) and (pos is None or ops.LN_POS):
            return ops.proj_add_layernorm(x, a, w, self.weight, self.bias, self.eps, drop_p, chain, pos, pos_param)
        return self.forward(x, lin(a), drop_p, chain, pos, pos_param)


class MultiheadAttention(nn.Module):
    """torch.nn.MultiheadAttention parameters (packed in_proj), HIP attention core."""

    def __init__(self, embed_dim, num_heads, dropout=0.0):
        super().__init__()
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        self.head_dim = embed_dim // num_heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = LinearP(embed_dim, embed_dim)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.out_proj.bias, 0.)

    def forward(self, q_in, k_in, v_in, B, Sq, Sk, key_padding_mask=None, causal=False, chains=(None, None, None), kv=None, out_drop=0.0,
                no_proj=False):
        """chains: ops.GradChain (or None) of the tensor behind q_in / k_in / v_in -- only where the projection's input gradient
        IS that tensor's gradient (the input itself, or input + a constant position term).
        kv = (K buffer, key column, V buffer, value column, sink_K, sink_V): keys / values already projected, as column slices of
        buffers shared with other layers (ops.multi_linear over all decoder layers; K and V may be the SAME buffer, then one
        sink); k_in / v_in are then unused.  no_proj: return the heads' output BEFORE out_proj (the caller folds the projection into
        the LayerNorm launch: LayerNormP.proj)"""
        E = self.embed_dim
        w, b = self.in_proj_weight, self.in_proj_bias
        cq, ck, cv = chains
        if kv is not None:
            k_all, kcol, v_all, vcol, sk, sv = kv
            q = ops.linear(q_in, W(w, b, 0, E), chain=cq)
            if k_all is v_all:
                bufs, roles, sinks = [q, k_all], ((0, 0), (1, kcol), (1, vcol)), (None, sk)
            else:
                bufs, roles, sinks = [q, k_all, v_all], ((0, 0), (1, kcol), (2, vcol)), (None, sk, sv)
            o = ops.attention(bufs, roles, B, self.num_heads, Sq, Sk, self.head_dim, kpm=key_padding_mask,
                              causal=causal, drop_p=self.dropout if self.training else 0.0,
                              sinks=sinks if (sk is not None or sv is not None) else None)
            return o if no_proj else self.out_proj(o, drop_p=out_drop)
        if q_in is k_in and k_in is v_in:
            bufs = [ops.linear(q_in, W(w, b, 0, 3 * E), chain=cq)]
            roles = ((0, 0), (0, E), (0, 2 * E))
        elif q_in is k_in:
            if Sq == Sk and ops.attention_qkv_ok(E, self.num_heads, Sq, causal, B):
                # q = k = x + pos, value = x: the three projections inside the attention launch (gpv_attention_qkv_fwd)
                o = ops.attention_qkv(q_in, v_in, W(w, b, 0, 3 * E), B, self.num_heads, Sq, kpm=key_padding_mask,
                                      drop_p=self.dropout if self.training else 0.0, chains=(cq, cv))
                return o if no_proj else self.out_proj(o, drop_p=out_drop)
            bufs = [ops.linear(q_in, W(w, b, 0, 2 * E), chain=cq), ops.linear(v_in, W(w, b, 2 * E, 3 * E), chain=cv)]
            roles = ((0, 0), (0, E), (1, 0))
        elif k_in is v_in:
            bufs = [ops.linear(q_in, W(w, b, 0, E), chain=cq), ops.linear(k_in, W(w, b, E, 3 * E), chain=ck)]
            roles = ((0, 0), (1, 0), (1, E))
        else:
            bufs = [ops.linear(q_in, W(w, b, 0, E), chain=cq), ops.linear(k_in, W(w, b, E, 2 * E), chain=ck),
                    ops.linear(v_in, W(w, b, 2 * E, 3 * E), chain=cv)]
            roles = ((0, 0), (1, 0), (2, 0))
        o = ops.attention(bufs, roles, B, self.num_heads, Sq, Sk, self.head_dim, kpm=key_padding_mask, causal=causal,
                          drop_p=self.dropout if self.training else 0.0)
        return o if no_proj else self.out_proj(o, drop_p=out_drop)


class TransformerEncoderLayer(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1):
        super().__init__()
        self.self_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = LinearP(d_model, dim_feedforward)
        self.linear2 = LinearP(dim_feedforward, d_model)
        self.norm1 = LayerNormP(d_model)
        self.norm2 = LayerNormP(d_model)
        self.p = dropout

    def forward(self, src, qk, pos, B, S, kpm):
        """transformer.py:148-161 (forward_post).  qk = src + pos arrives with src: the second output of the LayerNorm that
        produced src (the previous layer's norm2; layer 0: one add) -- and this layer's norm2 emits the next layer's.
        -> (out, out + pos)"""
        p = self.p if self.training else 0.0
        ch = ops.grad_chain(src)                 # src feeds norm1's residual, Wqk (through src + pos, pos constant) and Wv
        # (out_proj rides in norm1's launch: LayerNormP.proj)
        o = self.self_attn(qk, qk, src, B, S, S, kpm, chains=(ch, ch, ch), no_proj=True)
        src = self.norm1.proj(src, o, self.self_attn.out_proj, p, chain=ch)
        return ffn_block(src, self.linear1, self.linear2, self.norm2, p, pos=pos)

    def forward_pre(self, src, pos, B, S, kpm, pos_grad=False):
        """transformer.py:163-175 (`pre_norm: true`; no shipped config): normalise, attend / feed forward, add to the stream.
        Plain composition of the same kernels -- LayerNorm with its `+ pos` second output, attention, GEMMs with the dropout in
        their epilogues, element-wise add; autograd sums the stream's two consumers.  Not tuned: correctness branch."""
        p = self.p if self.training else 0.0
        if pos_grad:                              # a LEARNED position term: the sum is a launch autograd can differentiate
            s2 = self.norm1(src)
            qk = ops.add(s2, pos)
        else:
            s2, qk = self.norm1(src, pos=pos)
        src = ops.add(src, self.self_attn(qk, qk, s2, B, S, S, kpm, out_drop=p))
        s2 = self.norm2(src)
        return ops.add(src, self.linear2(self.linear1(s2, ops.ACT_RELU, p), drop_p=p))


class TransformerDecoderLayer(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1):
        super().__init__()
        self.self_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.multihead_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = LinearP(d_model, dim_feedforward)
        self.linear2 = LinearP(dim_feedforward, d_model)
        self.norm1 = LayerNormP(d_model)
        self.norm2 = LayerNormP(d_model)
        self.norm3 = LayerNormP(d_model)
        self.p = dropout

    def forward(self, tgt, tgt_qp, memory, mem_pos, qpos, qpos_param, B, Q, S, kpm, mem_chain=None, emit=True, kv=None):
        """transformer.py:211-232 (forward_post); mem_pos = memory + pos is layer-invariant.  mem_chain: the GradChain of the
        encoder memory, shared by the K (through memory + pos) and V projections of all six layers.
        tgt_qp = tgt + query_pos arrives with tgt (second output of the LayerNorm that produced tgt); norm1 emits the
        cross-attention's query sum, norm3 the next layer's tgt_qp (emit).  qpos: query_embed as [Q, C] rows in the compute dtype
        (a constant for autograd), qpos_param: the parameter itself -- the gradients of the sums w.r.t. query_pos are column sums
        accumulated into its gradient by the LayerNorm backward (ops._pos_sink); w.r.t. tgt they are summed inside that kernel.
        -> (out, out + query_pos | None)"""
        p = self.p if self.training else 0.0
        ch = ops.grad_chain(tgt)                 # tgt feeds norm1's residual and Wv (and, through tgt + query_pos, Wqk)
        o = self.self_attn(tgt_qp, tgt_qp, tgt, B, Q, Q, chains=(None, None, ch), no_proj=True)
        tgt, tq = self.norm1.proj(tgt, o, self.self_attn.out_proj, p, chain=ch, pos=qpos, pos_param=qpos_param)
        a = self.multihead_attn(tq, mem_pos, memory, B, Q, S, kpm, chains=(None, mem_chain, mem_chain), kv=kv, no_proj=True)
        tgt = self.norm2.proj(tgt, a, self.multihead_attn.out_proj, p)
        if not emit:
            return ffn_block(tgt, self.linear1, self.linear2, self.norm3, p), None
        return ffn_block(tgt, self.linear1, self.linear2, self.norm3, p, pos=qpos, pos_param=qpos_param)

    def forward_pre(self, tgt, memory, mem_pos, qpos, qpos_param, B, Q, S, kpm):
        """transformer.py:234-255 (`pre_norm: true`): see TransformerEncoderLayer.forward_pre.  query_pos enters through the
        LayerNorms' second output, its gradient through ops._pos_sink as in forward()."""
        p = self.p if self.training else 0.0
        t2, tq = self.norm1(tgt, pos=qpos, pos_param=qpos_param)
        tgt = ops.add(tgt, self.self_attn(tq, tq, t2, B, Q, Q, out_drop=p))
        t2, tq = self.norm2(tgt, pos=qpos, pos_param=qpos_param)
        tgt = ops.add(tgt, self.multihead_attn(tq, mem_pos, memory, B, Q, S, kpm, out_drop=p))
        t2 = self.norm3(tgt)
        return ops.add(tgt, self.linear2(self.linear1(t2, ops.ACT_RELU, p), drop_p=p))


class TransformerEncoder(nn.Module):
    def __init__(self, make_layer, num_layers, norm=None):
        super().__init__()
        self.layers = nn.ModuleList([make_layer() for _ in range(num_layers)])
        self.norm = norm                          # LayerNorm only with pre_norm (transformer.py:37)


class TransformerDecoder(nn.Module):
    def __init__(self, make_layer, num_layers, d_model):
        super().__init__()
        self.layers = nn.ModuleList([make_layer() for _ in range(num_layers)])
        self.norm = LayerNormP(d_model)


class Transformer(nn.Module):
    def __init__(self, d_model=512, nhead=8, num_encoder_layers=6, num_decoder_layers=6, dim_feedforward=2048,
                 dropout=0.1, normalize_before=False, return_intermediate_dec=False):
        super().__init__()
        self.normalize_before = bool(normalize_before)
        self.encoder = TransformerEncoder(lambda: TransformerEncoderLayer(d_model, nhead, dim_feedforward, dropout),
                                          num_encoder_layers, LayerNormP(d_model) if normalize_before else None)
        self.decoder = TransformerDecoder(lambda: TransformerDecoderLayer(d_model, nhead, dim_feedforward, dropout),
                                          num_decoder_layers, d_model)
        self.d_model, self.nhead = d_model, nhead
        for p in self.parameters():                       # transformer.py:41-44
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def forward(self, src, mask, query_embed, pos, need_all_layers):
        """src [B,S,C] rows (input_proj output), mask [B,S] bool, query_embed [Q,C] param, pos [B,S,C].
        Returns list of decoder-normed outputs [B,Q,C] (all layers, or only the last) -- transformer.py:46-58,94-123."""
        B, S, C = src.shape
        Q = query_embed.shape[0]
        kpm = mask.to(torch.uint8).contiguous() if mask is not None else None
        x = src.reshape(B * S, C)
        pe = pos.reshape(B * S, C)
        if self.normalize_before:
            return self._forward_pre(x, pe, kpm, query_embed, B, S, C, Q, need_all_layers)
        if torch.is_grad_enabled() and pe.requires_grad:
            return self._forward_plain(x, pe, kpm, query_embed, B, S, C, Q, need_all_layers)
        if len(self.encoder.layers) == 0:
            xq = ops.add(x, pe)
        else:
            xq = ops.add(x, pe)                   # layer 0's q / k input; every later `x + pos` leaves the LayerNorm that makes x
            for layer in self.encoder.layers:
                x, xq = layer(x, xq, pe, B, S, kpm)
        memory = ops.boundary(x, 'mem')           # (backward: every decoder layer has run when the gradient arrives here)
        mem_pos = xq                              # memory + pos: the last encoder LayerNorm's second output
        n = len(self.decoder.layers)
        k_all = v_all = sk = sv = mem_chain = None
        if HOIST_KV and n > 1:
            # the cross-attention keys / values of ALL decoder layers: memory is layer-invariant, so the 2 n projections of it are
            # two GEMMs over the concatenated weights (9600 x 256 -> 1536 at B = 32 instead of twelve 9600 x 256 -> 256), and
            # their backward-data one GEMM each with K = 1536 instead of twelve chained ones
            ca = [l.multihead_attn for l in self.decoder.layers]
            k_all = ops.multi_linear(mem_pos, [W(a.in_proj_weight, a.in_proj_bias, C, 2 * C) for a in ca])
            v_all = ops.multi_linear(memory, [W(a.in_proj_weight, a.in_proj_bias, 2 * C, 3 * C) for a in ca])
            if torch.is_grad_enabled() and k_all.requires_grad:
                sk, sv = ops.GradSink(n), ops.GradSink(n)
        else:
            mem_chain = ops.grad_chain(memory)
        qp = query_embed.detach().to(ops.RT.dtype).contiguous()                     # [Q, C] rows for the LayerNorm kernels
        tgt = torch.zeros(B * Q, C, device=src.device, dtype=ops.RT.dtype)
        # layer 0: tgt = 0, so tgt + query_pos IS query_pos (bit-identical to the add); its gradient reaches query_embed through
        # autograd (expand), the later layers' through ops._pos_sink
        tq = query_embed.to(ops.RT.dtype).unsqueeze(0).expand(B, Q, C).reshape(B * Q, C)
        outs = []
        for i, layer in enumerate(self.decoder.layers):
            tgt, tq = layer(tgt, tq, memory, mem_pos, qp, query_embed, B, Q, S, kpm, mem_chain, emit=i + 1 < n,
                            kv=None if k_all is None else (k_all, i * C, v_all, i * C, sk, sv))
            if need_all_layers or i == n - 1:
                outs.append(self.decoder.norm(tgt).reshape(B, Q, C))
        return outs, memory.reshape(B, S, C)


def _transformer_forward_pre(self, x, pe, kpm, query_embed, B, S, C, Q, need_all_layers):
    """`pre_norm: true` (transformer.py:46-58 with forward_pre layers and the encoder's final LayerNorm): the plain schedule -- every
    layer projects its own keys / values, no gradient chains or sinks"""
    pos_grad = torch.is_grad_enabled() and pe.requires_grad
    pe = ops._as_compute(pe) if pos_grad else ops._pos_rows(pe, C)
    for layer in self.encoder.layers:
        x = layer.forward_pre(x, pe, B, S, kpm, pos_grad)
    if pos_grad:
        memory = self.encoder.norm(x)
        mem_pos = ops.add(memory, pe)
    else:
        memory, mem_pos = self.encoder.norm(x, pos=pe)
    qp = query_embed.detach().to(ops.RT.dtype).contiguous()
    tgt = torch.zeros(B * Q, C, device=x.device, dtype=ops.RT.dtype)
    outs = []
    n = len(self.decoder.layers)
    for i, layer in enumerate(self.decoder.layers):
        tgt = layer.forward_pre(tgt, memory, mem_pos, qp, query_embed, B, Q, S, kpm)
        if need_all_layers or i == n - 1:
            outs.append(self.decoder.norm(tgt).reshape(B, Q, C))
    return outs, memory.reshape(B, S, C)


Transformer._forward_pre = _transformer_forward_pre


def _transformer_forward_plain(self, x, pe, kpm, query_embed, B, S, C, Q, need_all_layers):
    """Post-norm layers (transformer.py:148-161, 211-232) in the reference's own order with every position sum an element-wise
    launch autograd differentiates -- the schedule for a position term that is LEARNED (`position_embedding: learned`): forward()
    treats pos as a constant (second output of the LayerNorm kernels), which would drop its gradient.  No gradient chains, no
    hoisted keys / values: a correctness branch, not a tuned one."""
    pe = ops._as_compute(pe)
    for layer in self.encoder.layers:
        p = layer.p if layer.training else 0.0
        qk = ops.add(x, pe)
        x = layer.norm1(x, layer.self_attn(qk, qk, x, B, S, S, kpm), p)
        x = ffn_block(x, layer.linear1, layer.linear2, layer.norm2, p)
    memory = x
    mem_pos = ops.add(memory, pe)
    qe = query_embed.to(ops.RT.dtype)                                  # [Q, C]: ops.add broadcasts it over the batch and sums its gradient
    tgt = torch.zeros(B * Q, C, device=x.device, dtype=ops.RT.dtype)
    outs = []
    n = len(self.decoder.layers)
    for i, layer in enumerate(self.decoder.layers):
        p = layer.p if layer.training else 0.0
        tq = ops.add(tgt, qe)
        tgt = layer.norm1(tgt, layer.self_attn(tq, tq, tgt, B, Q, Q), p)
        tgt = layer.norm2(tgt, layer.multihead_attn(ops.add(tgt, qe), mem_pos, memory, B, Q, S, kpm), p)
        tgt = ffn_block(tgt, layer.linear1, layer.linear2, layer.norm3, p)
        if need_all_layers or i == n - 1:
            outs.append(self.decoder.norm(tgt).reshape(B, Q, C))
    return outs, memory.reshape(B, S, C)


Transformer._forward_plain = _transformer_forward_plain


def build_transformer(args):
    return Transformer(d_model=args.hidden_dim, dropout=args.dropout, nhead=args.nheads,
                       dim_feedforward=args.dim_feedforward, num_encoder_layers=args.num_encoder_layers,
                       num_decoder_layers=args.num_decoder_layers, normalize_before=args.pre_norm,
                       return_intermediate_dec=True)
