"""KV-cached greedy text decoding, one hipGraph per step (reference: the greedy branch of GPV.forward,
exp/gpv/models/gpv.py:178-196, which re-runs the whole prefix through the decoder 20 times and re-projects
the vocabulary embedding on every call).

Per image batch:
  * once: cross-attention keys/values of the memory for every decoder layer, the vocabulary classifiers W_c
    (answer_head.py:31-33, batch independent) and the transformed input-embedding table row of ``__cls__``;
  * per step t: ONE new token per sequence goes through the 3 decoder layers: the q | k | v projection of the new row as one
    GEMM that writes row t of the layer's [B, T, 3D] cache (ldc = T*3D; the attention kernel reads q, k and v out of it by
    stride), single-query attention over the t+1 cached keys and over the memory, FFN, the logits of that position written
    into row t of the [B, T, V] result, and one launch that picks the next token (logit + vocabulary mask, arg-max) and
    stores it as the next input and as ids[:, t+1].  At B <= 8 every projection runs on the few-row kernel (csrc/gemv.hip),
    at B <= 4 every LayerNorm is computed inside the projection that consumes it (gpv_ln_linear_rows): 27 launches per token;
    at B = 1 the out-projections ride in the single-query attention launches (gpv_attention_row_proj): 21.
Because the decoder is causal, the hidden state of position t only depends on tokens <= t: the per-step logits
are exactly the rows of the reference's final full pass, so ``answer_logits`` (1,B,T,V) is assembled from them.

Every step has static shapes, so it is captured once into a HIP graph (torch.cuda.CUDAGraph == hipGraph on
ROCm; our kernels are launched on the capturing stream through the C ABI) and replayed: 19+1 graph launches per
decode instead of ~20 x 60 kernel launches from Python.
"""
import os

import torch

from . import hip, ops
from .ops import RT, W


class GreedyKVDecoder:
    def __init__(self, model, B, Tm, use_graphs=True):
        self.m = model
        cfg = model.cfg
        self.B, self.Tm, self.T = B, Tm, cfg.max_text_len
        self.D = cfg.text_decoder.hidden_dim
        self.H = cfg.text_decoder.nheads
        self.V = len(model.vocab)
        self.L = len(model.text_decoder.layers)
        dev = model.vision_token.device
        dt = RT.dtype
        D, T = self.D, self.T
        self.memory = torch.empty(B * Tm, D, device=dev, dtype=dt)
        self.qkvc = [torch.zeros(B, T, 3 * D, device=dev, dtype=dt) for _ in range(self.L)]     # q | k | v of every decoded position
        self.kvm = [torch.empty(B * Tm, 2 * D, device=dev, dtype=dt) for _ in range(self.L)]
        self.tok = torch.zeros(B, dtype=torch.long, device=dev)
        self.logits = torch.empty(B, T, self.V, device=dev, dtype=dt)
        self.ids = torch.zeros(B, T, dtype=torch.long, device=dev)
        self.vocab_mask = torch.zeros(self.V, device=dev, dtype=torch.float32)
        self.wc = torch.empty(self.V, D, device=dev, dtype=dt)          # static: the captured graphs read these addresses
        self.use_graphs = use_graphs and dev.type == 'cuda'
        # B <= 4: LayerNorm inside the consuming projection, out-projection inside the single-query attention (csrc/gemv.hip)
        self.fused_rows = B <= hip.LN_LINEAR_MAX_ROWS and D <= hip.LN_LINEAR_MAX_COLS and D % 8 == 0
        self.part = torch.empty(B, self.H, D, device=dev, dtype=torch.float32) if self.fused_rows else None
        # out-projection inside the attention launch: 5.31-5.38 -> 5.24 ms at B = 1, 7.80 -> 8.18 ms at B = 4 (same box) -- B = 1 only
        self.row_proj = os.environ.get('GPV_DECODE_ROW_PROJ', '1' if B == 1 else '0') != '0'
        # greedy steps: the next token's input row (embedding -> transform -> + position) is written by the launch that picks the
        # token, from the input transform applied to the WHOLE vocabulary once per weights (gpv_argmax_rows_embed): 3 of a step's
        # dependent launches fewer.  GPV_DECODE_EMBED=0: the three launches (A/B)
        self.embed_in_pick = os.environ.get('GPV_DECODE_EMBED', '1') != '0'
        self.xin = torch.zeros(B, D, device=dev, dtype=dt)
        self.table = None                      # [V, D] transformed input embeddings, built with the first decode after a weights change
        self.pos = None                        # [T, D] position rows in the compute dtype
        self.graphs = [None] * T
        self.key = (RT.weights_epoch, RT.static_epoch, dt)

    # ---- once per batch -------------------------------------------------------------------
    def _prepare(self):
        m = self.m
        D = self.D
        for l, layer in enumerate(m.text_decoder.layers):
            a = layer.multihead_attn
            kv = ops.linear(self.memory, W(a.in_proj_weight, a.in_proj_bias, D, 3 * D))     # [B*Tm, 2D] = k | v
            self.kvm[l].copy_(kv)
        self.wc.copy_(m.answer_head.classifiers())                                         # [V, D]

    # ---- one decoding step (static shapes for a fixed t) ------------------------------------
    def _ln_lin(self, norm, x, s, w, bias, y, ldy, N, act=ops.ACT_NONE):
        """x' = norm(x + s); y[:, :N] (row pitch ldy) = act(x' w^T + bias); returns x'.  One launch at B <= 4 (gpv_ln_linear_rows);
        s: a tensor, or (partial rows, bias) from _attend"""
        B, D = x.shape
        if self.fused_rows:
            xn = torch.empty_like(x)
            sp, sb = s if isinstance(s, tuple) else (None, None)
            hip.ln_linear_rows(x, None if sp is not None else s, norm.weight.detach(), norm.bias.detach(), norm.eps, xn, w, bias, y, ldy,
                               B, N, D, act, s_partial=sp, s_bias=sb)
            return xn
        xn = norm(x, s)
        hip.gemm(xn, w, y, B, N, D, D, D, ldy, bias=bias, act=act)
        return xn

    def _attend(self, att, q, q_bs, k, v, kv_bs, kv_rs, Sk):
        """the sublayer output of one query row per sequence: attention over Sk keys + out_proj.  At B <= 4 one launch that leaves
        per-head partial rows for the LayerNorm + Linear kernel to sum (gpv_attention_row_proj), else attention + GEMM"""
        B, D, H = self.B, self.D, self.H
        dh = D // H
        if self.fused_rows and self.row_proj and Sk <= hip.ROW_PROJ_MAX_KEYS:
            wo = W(att.out_proj.weight, att.out_proj.bias)
            hip.attention_row_proj(q, q_bs, k, kv_bs, kv_rs, v, kv_bs, kv_rs, wo.lp(), self.part, B, H, Sk, dh, 1.0 / dh ** 0.5)
            return self.part, wo.bias_f32()
        o = torch.empty(B, D, device=self.tok.device, dtype=RT.dtype)
        hip.attention_fwd(q, k, v, o, ((q_bs, D), (kv_bs, kv_rs), (kv_bs, kv_rs), (D, D)), B, H, 1, Sk, dh, 1.0 / dh ** 0.5)
        return att.out_proj(o)

    def _input_rows(self, t):
        """the decoder input of position t for the tokens in self.tok: embedding -> transform (+ position), gpv.py:178-183"""
        m = self.m
        x = m.answer_input_embedings(self.tok)                                              # [B, D]
        if m.cfg.text_decoder.pos_enc is True:
            x = ops.add(x, m.pos_enc[0, t:t + 1].to(RT.dtype).contiguous())
        return x

    def _build_table(self):
        m = self.m
        ids = torch.arange(self.V, device=self.tok.device)
        tab = m.answer_input_embedings(ids).to(RT.dtype).contiguous()                       # [V, D]
        if self.table is None or self.table.dtype != tab.dtype:
            self.table = tab
        else:
            self.table.copy_(tab)                                                           # (captured graphs read this address)
        self.pos = m.pos_enc[0, :self.T].to(RT.dtype).contiguous() if m.cfg.text_decoder.pos_enc is True else None

    def _step_core(self, t, x=None):
        m, B, D, H, T, Tm = self.m, self.B, self.D, self.H, self.T, self.Tm
        dh = D // H
        dev = self.tok.device
        if x is None:
            x = self._input_rows(t)
        s = prev = None
        for l, layer in enumerate(m.text_decoder.layers):
            sa, ca = layer.self_attn, layer.multihead_attn
            wqkv = W(sa.in_proj_weight, sa.in_proj_bias, 0, 3 * D)
            c = self.qkvc[l]
            # q_t | k_t | v_t written in place: row b of the GEMM output lands at cache[b, t, :]
            if prev is None:
                hip.gemm(x, wqkv.lp(), c[:, t], B, 3 * D, D, D, D, T * 3 * D, bias=wqkv.bias_f32())
            else:
                x = self._ln_lin(prev.norm3, x, s, wqkv.lp(), wqkv.bias_f32(), c[:, t], T * 3 * D, 3 * D)
            s = self._attend(sa, c[:, t], T * 3 * D, c[:, :, D:], c[:, :, 2 * D:], T * 3 * D, 3 * D, t + 1)
            wq = W(ca.in_proj_weight, ca.in_proj_bias, 0, D)
            q = torch.empty(B, D, device=dev, dtype=RT.dtype)
            x = self._ln_lin(layer.norm1, x, s, wq.lp(), wq.bias_f32(), q, D, D)
            kvm = self.kvm[l]
            s = self._attend(ca, q, D, kvm, kvm[:, D:], Tm * 2 * D, 2 * D, Tm)
            w1 = W(layer.linear1.weight, layer.linear1.bias)
            h = torch.empty(B, w1.N, device=dev, dtype=RT.dtype)
            x = self._ln_lin(layer.norm2, x, s, w1.lp(), w1.bias_f32(), h, w1.N, w1.N, ops.ACT_RELU)
            s = layer.linear2(h)
            prev = layer
        lg = self.logits[:, t]                                                              # [B, V] logits of position t
        self._ln_lin(prev.norm3, x, s, self.wc, None, lg, T * self.V, self.V)
        return lg

    def _step(self, t):
        fused = self.embed_in_pick and self.table is not None
        lg = self._step_core(t, self.xin if fused else None)
        # next input token = arg-max of logit + vocabulary mask (gpv.py:185-188), also stored as ids[:, t+1]
        if fused and t + 1 < self.T:
            hip.argmax_rows(lg, self.vocab_mask, self.tok, self.ids[:, t + 1], table=self.table,
                            pos_row=None if self.pos is None else self.pos[t + 1], xnext=self.xin)
        else:
            hip.argmax_rows(lg, self.vocab_mask, self.tok, self.ids[:, t + 1] if t + 1 < self.T else None)

    def reorder(self, perm, upto):
        """beam search: sequence i continues the hypothesis that lived in slot perm[i]; positions < upto are valid"""
        for l in range(self.L):
            self.qkvc[l][:, :upto].copy_(self.qkvc[l][:, :upto].index_select(0, perm))

    @torch.no_grad()
    def decode(self, memory, vocab_mask=None):
        """memory [B, Tm, D] -> (answer_logits [1,B,T,V] incl. the vocab mask like the reference, ids [B,T])"""
        m = self.m
        if self.key != (RT.weights_epoch, RT.static_epoch, RT.dtype):       # weights changed: graphs hold stale copies
            self.graphs = [None] * self.T
            self.key = (RT.weights_epoch, RT.static_epoch, RT.dtype)
            self.table = None
        if self.embed_in_pick and self.table is None and not (self.tok.is_cuda and torch.cuda.is_current_stream_capturing()):
            self._build_table()              # (a decode first met inside a capture runs the three-launch input path)
        self.memory.copy_(memory.reshape(self.B * self.Tm, self.D))
        self.vocab_mask.zero_()
        if vocab_mask is not None:
            self.vocab_mask.copy_(vocab_mask.float())
        self._prepare()
        cls = m.word_to_idx['__cls__']
        self.tok.fill_(cls)
        self.ids.zero_()
        self.ids[:, 0] = cls
        if self.embed_in_pick and self.table is not None:
            self.xin.copy_(self._input_rows(0))
        for t in range(self.T):
            if not self.use_graphs:
                self._step(t)
                continue
            if self.graphs[t] is None:
                tok_in, ids_in, xin_in = self.tok.clone(), self.ids.clone(), self.xin.clone()
                self._step(t)                                   # warm-up (kernel attributes, caches) outside capture
                torch.cuda.synchronize()
                self.tok.copy_(tok_in)
                self.ids.copy_(ids_in)
                self.xin.copy_(xin_in)
                g = torch.cuda.CUDAGraph()
                from .misc import capture_guard
                with capture_guard(), torch.cuda.graph(g, capture_error_mode='thread_local'):
                    self._step(t)
                self.graphs[t] = g
                self.tok.copy_(tok_in)
                self.ids.copy_(ids_in)
                self.xin.copy_(xin_in)
            self.graphs[t].replay()
        if self.use_graphs:
            # Drain the stream after the 20 graph launches.  Without a STREAM synchronisation every few decodes the
            # HIP runtime (ROCm 7.2) faults with HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION once ~140 graph launches
            # have been issued while the GPU is behind the host (B=64; tools/repro_graph_launch_fault.py reproduces it: eager
            # launches never fault, event waits do not help, a stream sync every <=4 decodes does).  The sync costs
            # no GPU time: the next call's encoder is host-issued in ~20 ms either way.
            torch.cuda.current_stream().synchronize()
        # fresh tensors like the reference returns: self.logits is this decoder's persistent buffer, the next decode of
        # the same (batch, memory length) overwrites it
        out = self.logits.float() + self.vocab_mask if vocab_mask is not None else self.logits.clone()
        return out.unsqueeze(0), self.ids.clone()
