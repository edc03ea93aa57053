"""GPV: the drop-in for ``exp.gpv.models.gpv.GPV`` (reference: exp/gpv/models/gpv.py:58-466).

Same constructor config schema (configs/exp/gpv.yaml ``model:`` block), same 836 state-dict keys,
same methods/attributes the reference's drivers use: ``forward``, ``forward_beam_search``,
``encode_answers``, ``token_ids_to_words``, ``vocab``, ``word_to_idx``, ``init_detr_params``,
``load_pretr_detr``, ``criterion``.  All tensor math runs on the HIP kernels (gpv1_amd.hip) -- there is
no CPU path; the module raises if it is called with CPU tensors or without the library.
"""
import copy
import json
import math
import contextlib
import os
import re

import numpy as np
import torch
import torch.nn as nn

from . import ops
from . import hip
from . import detr as detr_mod
from .ops import W, RT
from .detr import create_detr, create_detr_roi_head
from .bert import Bert
from .vilbert import BertConnectionLayer
from . import transformer as transformer_mod
from .transformer import LinearP, LayerNormP, MultiheadAttention, ffn_block
from .criterion import GPVCriterion
from .misc import AttrDict, NestedTensor

try:                                              # the reference uses nltk's word_tokenize (gpv.py:4,409)
    from nltk.tokenize import word_tokenize as _word_tokenize
except Exception:                                 # nltk is not in the image: Treebank-like regex split
    def _word_tokenize(s):
        return re.findall(r"__\w+__|\w+(?:'\w+)?|[^\w\s]", s)


def positionalencoding1d(d_model, length):
    """gpv.py:18-34"""
    if d_model % 2 != 0:
        raise ValueError("Cannot use sin/cos positional encoding with odd dim (got dim={:d})".format(d_model))
    pe = torch.zeros(length, d_model)
    position = torch.arange(0, length).unsqueeze(1)
    div_term = torch.exp((torch.arange(0, d_model, 2, dtype=torch.float) * -(math.log(10000.0) / d_model)))
    pe[:, 0::2] = torch.sin(position.float() * div_term)
    pe[:, 1::2] = torch.cos(position.float() * div_term)
    return pe


TEXT_EARLY = os.environ.get('GPV_TEXT_EARLY', '1') != '0'      # teacher forcing: target embedding, vocabulary classifiers and the first text-decoder layer's self-attention on a branch beside the co-attention stage


class TextDecoderLayer(nn.Module):
    """torch.nn.TransformerDecoderLayer(d_model, nhead, dim_feedforward=2048, relu, post-norm) parameters."""

    def __init__(self, d_model, nhead, dropout, dim_feedforward=2048):
        super().__init__()
        self.self_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.multihead_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = LinearP(d_model, dim_feedforward)
        self.linear2 = LinearP(dim_feedforward, d_model)
        self.norm1, self.norm2, self.norm3 = LayerNormP(d_model), LayerNormP(d_model), LayerNormP(d_model)
        self.p = dropout

    def self_part(self, tgt, B, Tt):
        """the causal self-attention sublayer alone: it depends on the target tokens only -- teacher forcing runs the FIRST layer's on a
        branch beside the co-attention stage (GPV._forward_impl)"""
        p = self.p if self.training else 0.0
        c0 = ops.grad_chain(tgt)                       # (ops.GradChain: tgt feeds the projection and the residual)
        return self.norm1(tgt, self.self_attn(tgt, tgt, tgt, B, Tt, Tt, causal=True, chains=(c0, c0, c0)), p, chain=c0)

    def forward(self, tgt, memory, B, Tt, Tm, mem_chain=None, mem_kpm=None, kv=None, self_done=False):
        """kv: the cross-attention keys | values of this layer as columns of the buffer GPV.decode_text projected for all layers;
        self_done: tgt is already self_part's output"""
        p = self.p if self.training else 0.0
        if not self_done:
            tgt = self.self_part(tgt, B, Tt)
        c1 = ops.grad_chain(tgt)
        tgt = self.norm2(tgt, self.multihead_attn(tgt, memory, memory, B, Tt, Tm, key_padding_mask=mem_kpm,
                                                  chains=(c1, mem_chain, mem_chain), kv=kv), p,
                         chain=c1)                     # no memory padding mask in the reference (mem_kpm: size-class padding only)
        return ffn_block(tgt, self.linear1, self.linear2, self.norm3, p)


class TextDecoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.layers = nn.ModuleList([TextDecoderLayer(cfg.hidden_dim, cfg.nheads, cfg.dropout) for _ in range(cfg.num_layers)])


class AnswerHead(nn.Module):
    """answer_head.py:8-33: logits = h @ Linear(vocab_embed)^T (frozen vocabulary embedding)."""

    def __init__(self, vocab, classifier_transform, vocab_embed):
        super().__init__()
        self.vocab = vocab
        self.vocab_embed = nn.Parameter(vocab_embed, requires_grad=False)
        self.classifier_transform = classifier_transform
        self._wc = None

    def classifiers(self):
        # W_c is batch independent: in eval it is computed once per weights version
        if not torch.is_grad_enabled() and self._wc is not None and self._wc[0] == (RT.weights_epoch, RT.static_epoch, RT.dtype):
            return self._wc[1]
        e = ops._as_compute(self.vocab_embed.detach())
        wc = self.classifier_transform(e)                       # [V, D]
        if not torch.is_grad_enabled():
            self._wc = ((RT.weights_epoch, RT.static_epoch, RT.dtype), wc)
        return wc

    def forward(self, h, wc=None):
        """h [..., D] -> [..., V]"""
        D = h.shape[-1]
        return ops.matmul_nt(h.reshape(-1, D), self.classifiers() if wc is None else wc).reshape(*h.shape[:-1], -1)


class AnswerInputEmbedding(nn.Module):
    """gpv.py:46-55"""

    def __init__(self, weight, transform, freeze_embeddings=True):
        super().__init__()
        self.transform = transform
        self.embedding_layer = nn.Embedding.from_pretrained(weight.clone(), freeze=freeze_embeddings)

    def forward(self, token_ids):
        return self.transform(ops.embedding(self.embedding_layer.weight, token_ids))


def images_on_gpu(images):
    t = images.tensors if isinstance(images, NestedTensor) else (images[0] if isinstance(images, (list, tuple)) and len(images) else images)
    return torch.is_tensor(t) and t.is_cuda


class GPV(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        if isinstance(cfg, dict) and not isinstance(cfg, AttrDict):
            cfg = AttrDict.wrap(cfg)
        self.cfg = cfg
        if cfg.answering_type not in ('generation', 'classification'):
            raise ValueError(f'answering_type {cfg.answering_type!r}: generation | classification (gpv.py:384-401)')
        if cfg.answer_head is not None:
            # answer_head: linear builds nn.Linear(detr.hidden_dim = 256, V) (answer_head.py:36-59, build_answer_head passes
            # cfg.detr.hidden_dim) and applies it to the text decoder's 768-wide output (gpv.py:466): the reference itself cannot run
            # this branch with GPV-1's widths, so there is nothing to be faithful to
            raise NotImplementedError('answer_head: linear is shape-inconsistent in the reference itself (Linear(256, V) on 768-wide '
                                      'decoder outputs); gpv1_amd builds answer_head: null')
        self.detr = create_detr_roi_head(cfg.detr) if cfg.roi_head is True else create_detr(cfg.detr)
        self.detr_joiner = LinearP(cfg.detr_joiner.detr_dim, cfg.detr_joiner.out_dim)
        self.init_detr_params = []
        self.bert = Bert(num_layers=cfg.get('bert_layers', 12) if isinstance(cfg, dict) else 12,
                         weights=cfg.get('bert_weights') if isinstance(cfg, dict) else None)    # model.bert_weights: local bert-base-uncased
        if isinstance(cfg, dict) and cfg.get('bert_dropout') is not None:          # test knob; the reference keeps HF's 0.1
            self.bert.model.p = float(cfg.get('bert_dropout'))
        self.bert_joiner = LinearP(cfg.bert_joiner.bert_dim, cfg.bert_joiner.out_dim)
        layer = BertConnectionLayer(cfg.co_att)
        self.co_att_transformer = nn.ModuleList([copy.deepcopy(layer) for _ in range(cfg.co_att.num_layers)])
        self.relevance_predictor = LinearP(cfg.hidden_dim, cfg.detr.num_classes + 1)
        self.text_decoder = TextDecoder(cfg.text_decoder)
        if isinstance(cfg.vocab, (list, tuple)):
            vocab = list(cfg.vocab)
        else:
            with open(cfg.vocab) as f:
                vocab = json.load(f)
        if cfg.vocab_embed is None:
            vocab_embed = 0.1 * torch.randn(len(vocab), cfg.bert_joiner.bert_dim)
        elif torch.is_tensor(cfg.vocab_embed):
            vocab_embed = cfg.vocab_embed.float()
        else:
            vocab_embed = torch.from_numpy(np.load(cfg.vocab_embed)).float()
        self.answer_head = AnswerHead(vocab, LinearP(cfg.bert_joiner.bert_dim, cfg.bert_joiner.out_dim), vocab_embed)
        self.vocab = self.answer_head.vocab
        self.word_to_idx = {w: i for i, w in enumerate(self.vocab)}
        self.answer_input_embedings = AnswerInputEmbedding(
            self.answer_head.vocab_embed.data, LinearP(cfg.bert_joiner.bert_dim, cfg.bert_joiner.out_dim))
        self.vision_token = nn.Parameter(0.1 * torch.randn([cfg.hidden_dim]))        # declared, unused (gpv.py:107-110)
        self.lang_token = nn.Parameter(0.1 * torch.randn([cfg.hidden_dim]))
        self.relevance_tokens = nn.Parameter(0.1 * torch.randn([2, cfg.hidden_dim]))
        self.criterion = GPVCriterion(cfg.losses)
        self._kvdec = {}
        self._igraphs = {}
        import collections
        self._qcache, self.qcache_hits, self._last_q_enc, self._host_ids = collections.OrderedDict(), 0, None, None
        self._qcache_epoch, self._graph_q_enc = None, None
        self.pos_enc = nn.Parameter(positionalencoding1d(cfg.text_decoder.hidden_dim, cfg.max_pos_enc_len)
                                    .view(1, cfg.max_pos_enc_len, -1), requires_grad=False)

    # ------------------------------------------------------------------ plumbing
    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        RT.bump_weights()
        return r

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        RT.bump_weights()
        return r

    def load_pretr_detr(self):
        """gpv.py:122-135"""
        loaded = torch.load(self.cfg.pretr_detr, map_location='cpu')['model']
        cur = self.state_dict()
        for lk in loaded.keys():
            k = 'detr.' + lk
            if k in cur:
                if cur[k].size() == loaded[lk].size():
                    self.init_detr_params.append(k)
                    cur[k] = loaded[lk]
                else:
                    print(f'    {lk} size does not match')
        self.load_state_dict(cur)

    # ------------------------------------------------------------------ encoder shared by all branches
    def _encode(self, images, queries, query_encodings=None, lang_extra=None, after_detr=None):
        """query_encodings: BERT features computed by the caller (train.GraphedBody runs the frozen, no_grad BERT as a
        parallel branch of the backbone's hipGraph: 110 launches of <= 144 workgroups hide under the convolutions)"""
        outputs = self.detr(images)
        # (backward: everything downstream of the DETR stream -- text decoder, answer head, co-attention -- is done when this fires)
        if not detr_mod.BOUNDARY_BELOW_ROI:
            outputs['detr_hs'] = ops.boundary(outputs['detr_hs'], 'detr')
        if after_detr is not None:
            # work that does not depend on the encoder: forked here, beside the co-attention stage.  AFTER the boundary node: autograd runs
            # ready nodes latest-created first, so these nodes' backward -- and the weight gradients it defers -- come before the boundary
            # fires and launches the deferred group (created before it they ran after it: the group missed them)
            after_detr(outputs)
        outputs['detr_hs'] = self.detr_joiner(outputs['detr_hs'])     # [L,B,Q,768]
        forked = callable(query_encodings)
        if forked:                                     # (a branch forked earlier: joined here, where the features are first needed)
            query_encodings = query_encodings()
        self._last_q_enc = None
        if query_encodings is None:
            with torch.no_grad():
                query_encodings, _ = self.bert(queries)
            self._last_q_enc = query_encodings
        elif forked:
            self._last_q_enc = query_encodings        # (inside an inference graph: the static tensor every replay rewrites)
        lv = self.bert_joiner(query_encodings.detach())                            # [B,Tl,768]
        B, Tl, D = lv.shape
        vl = outputs['detr_hs'][-1]
        Tv = vl.shape[1]
        lv2, vl2 = lv.reshape(B * Tl, D), vl.reshape(B * Tv, D)
        for layer in self.co_att_transformer:
            lv2, vl2 = layer(lv2, vl2, B, Tl, Tv, kpm1=lang_extra)
        rel = self.relevance_predictor(vl2, out_f32=True).reshape(B, Tv, -1)       # fp32
        outputs['pred_relevance_logits'] = outputs['pred_relevance_logits'] + rel
        if self.cfg.detr.aux_loss:
            for aux in outputs['aux_outputs']:
                aux['pred_relevance_logits'] = aux['pred_relevance_logits'] + rel
        if self.cfg.relevance_conditioning is not False:                            # gpv.py:364-375
            vl2 = ops.relevance_condition(vl2, outputs['pred_relevance_logits'].reshape(B * Tv, 2), self.relevance_tokens)
        memory = torch.cat((vl2.reshape(B, Tv, D), lv2.reshape(B, Tl, D)), 1)      # [B, Tv+Tl, D]
        return outputs, memory

    def _text_input(self, target):
        """target embeddings [B,Tt,D] (+ position encoding) -> the first decoder layer's input rows [B*Tt, D]"""
        B, Tt, D = target.shape
        if self.cfg.text_decoder.pos_enc is True:
            target = ops.add(target.reshape(B * Tt, D), self.pos_enc[0, :Tt].to(RT.dtype)).reshape(B, Tt, D)
        return target.reshape(B * Tt, D)

    def decode_text(self, target, memory, mem_kpm=None, wc=None, x0=None):
        """target [B,Tt,D], memory [B,Tm,D] -> logits [B,Tt,V]   (gpv.py:449-466); wc: the vocabulary classifiers, x0: the first layer's
        self-attention sublayer output, when the caller has already computed them (on a branch)"""
        B, Tt, D = target.shape
        Tm = memory.shape[1]
        x = self._text_input(target) if x0 is None else x0
        mem = memory.reshape(B * Tm, D)
        layers = self.text_decoder.layers
        if transformer_mod.HOIST_KV and len(layers) > 1:
            # the co-attention output feeds the k | v projection of every layer: one GEMM over the concatenated weights
            # (ops.multi_linear), one gradient buffer the layers' attention backwards fill (ops.GradSink)
            ws = [W(l.multihead_attn.in_proj_weight, l.multihead_attn.in_proj_bias, D, 3 * D) for l in layers]
            kv_all = ops.multi_linear(mem, ws)
            sink = ops.GradSink(len(layers)) if (torch.is_grad_enabled() and kv_all.requires_grad) else None
            for i, layer in enumerate(layers):
                x = layer(x, mem, B, Tt, Tm, None, mem_kpm, kv=(kv_all, 2 * D * i, kv_all, 2 * D * i + D, sink, sink), self_done=(i == 0 and x0 is not None))
            return self.answer_head(x, wc).reshape(B, Tt, -1)
        mem_chain = ops.grad_chain(mem)                # the co-attention output feeds the K|V projection of every layer
        for i, layer in enumerate(layers):
            x = layer(x, mem, B, Tt, Tm, mem_chain, mem_kpm, self_done=(i == 0 and x0 is not None))
        return self.answer_head(x, wc).reshape(B, Tt, -1)

    # ------------------------------------------------------------------ reference API
    def _host_tokenize(self, images, queries):
        """list[str] queries -> (ids, mask) on the images' device, padded to the batch's own longest query exactly like the
        reference's tokenizer call (bert.py:12-15 padding=True); staged through pinned memory (no host<->device sync).  Lets the
        string-query inference of inference.py / compute_predictions.py replay the captured graphs instead of launching eagerly."""
        if not (isinstance(queries, (list, tuple)) and queries and all(isinstance(q, str) for q in queries)):
            return queries
        tok = getattr(self.bert, 'tokenizer', None)
        dev = images.tensors.device if hasattr(images, 'tensors') else None
        if tok is None or dev is None or dev.type != 'cuda':
            return queries
        from .misc import STAGER
        ids, attn = tok(list(queries))
        out = STAGER.to_device(ids, torch.long, dev), STAGER.to_device(attn, torch.long, dev)
        self._host_ids = (out[0], [(tuple(r), tuple(a)) for r, a in zip(ids.tolist(), attn.tolist())])   # keys of the feature cache
        return out

    def forward(self, images, queries, answer_token_ids, targets=None, vocab_mask=None):
        if not self.training and not torch.is_grad_enabled() and images_on_gpu(images):
            # inference: the small GEMMs of these batch sizes run faster without gemm_pipe.hip's small-M configurations (greedy batch 64
            # 15.0 -> 14.05 ms, batch 1 -0.08 ms; same box) -- the option is read when the launches are issued / captured
            with hip.gemm_flags(hip.GEMM_NO_PIPE_SMALL):          # per call, this thread only (not the process-wide GPV_OPT_PIPE_SMALL)
                return self._forward_entry(images, queries, answer_token_ids, targets, vocab_mask)
        return self._forward_entry(images, queries, answer_token_ids, targets, vocab_mask)

    def _forward_entry(self, images, queries, answer_token_ids, targets=None, vocab_mask=None):
        if (answer_token_ids is None and targets is None and not self.training and torch.is_grad_enabled() is False
                and self.cfg.get('graph_inference', True) and self.cfg.get('kv_decode', True)):
            queries = self._host_tokenize(images, queries)
            g = self._graphed_greedy(images, queries, vocab_mask)
            if g is not None:
                return g
        return self._forward_impl(images, queries, answer_token_ids, targets, vocab_mask)

    # ---- whole greedy inference as ONE hipGraph ---------------------------------------------------------------
    def _graphed_greedy(self, images, queries, vocab_mask):
        # Query-string -> BERT-feature cache (the reference's detection / classification queries are 18 fixed
        # templates, data/coco/preprocess_coco_detection.py:14-33): in eval() BERT is deterministic (no dropout), so the features
        # of a tokenised query row -- padding included: the reference attends padded tokens -- are a function of that row alone.
        # When every row of a host-tokenised batch has been seen, the inference graph WITHOUT the BERT branch is replayed on the
        # cached features (exact: they are the bits an earlier replay produced).  Device-tensor queries are not cached (reading
        # them back would synchronise).
        keys = self._query_keys(queries)
        if keys is not None and all(k in self._qcache for k in keys):
            for k in keys:
                self._qcache.move_to_end(k)
            q_enc = torch.stack([self._qcache[k] for k in keys])
            self.qcache_hits += 1
            return self._graphed(('greedy_cached',), lambda im, q, vm, qe: self._forward_impl(im, q, None, None, vm, kv_graphs=False,
                                                                                             query_encodings=qe),
                                 images, queries, vocab_mask, extra=q_enc)
        out = self._graphed(('greedy',), lambda im, q, vm, qe: self._forward_impl(im, q, None, None, vm, kv_graphs=False,
                                                                               query_encodings=self._bert_fork(im, q)),
                            images, queries, vocab_mask)
        if out is not None and keys is not None and self._graph_q_enc is not None:
            feats = self._graph_q_enc.detach().clone()             # [B, T, 768]: what this replay's BERT branch computed
            for i, k in enumerate(keys):
                self._qcache[k] = feats[i]
            while len(self._qcache) > int(self.cfg.get('query_cache_rows', 4096)):
                self._qcache.popitem(last=False)
        return out

    def _query_keys(self, queries):
        """cache keys of a batch that _host_tokenize produced from strings (None: anything else)"""
        if self.cfg.get('query_cache', True) is False or self.training:
            return None
        if self._qcache_epoch != (RT.static_epoch, RT.dtype):          # BERT's weights / the compute dtype may have changed
            self._qcache.clear()
            self._qcache_epoch = (RT.static_epoch, RT.dtype)
        h = getattr(self, '_host_ids', None)
        if h is None or not isinstance(queries, (tuple, list)) or len(queries) != 2 or queries[0] is not h[0]:
            return None
        return h[1]

    def _bert_fork(self, images, queries):
        """the frozen BERT on a side stream -- inside a capture: a parallel branch of the inference graph beside the backbone and the
        DETR transformer (~110 launches of <= 8 token rows at batch 1); returns the join, called where the features are first
        needed (_encode).  GPV_INFER_BERT_BRANCH=0: in line."""
        dev = images.tensors.device
        if dev.type != 'cuda' or os.environ.get('GPV_INFER_BERT_BRANCH', '1') == '0':
            return None
        side = getattr(self, '_iside', None)
        if side is None:
            side = self._iside = torch.cuda.Stream(device=dev)
        cur = torch.cuda.current_stream(dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side), torch.no_grad():
            enc, _ = self.bert(queries)

        def join():
            torch.cuda.current_stream(dev).wait_stream(side)
            return enc
        return join

    def _graphed(self, kind, fn, images, queries, vocab_mask, extra=None):
        """Greedy inference (gpv.py:178-196) has static shapes for a fixed batch: ~1000 encoder launches + 20 decode
        steps are captured once into a single HIP graph (torch.cuda.CUDAGraph; our kernels are launched on the
        capturing stream through the C ABI) and replayed.  At batch 1 the eager path is bound by ~20 us of Python per
        launch (20 ms per image for ~6 ms of GPU work).  Returns None when the call does not qualify (host-side
        tokenisation, CPU tensors, nested capture): the caller then runs the eager path."""
        from .misc import NestedTensor
        if not isinstance(images, NestedTensor) or not isinstance(queries, (tuple, list)) or len(queries) != 2 \
                or not all(torch.is_tensor(q) for q in queries):
            return None
        x, m = images.tensors, images.mask
        ids, attn = queries
        if not x.is_cuda or torch.cuda.is_current_stream_capturing():
            return None
        key = kind + (tuple(x.shape), tuple(ids.shape), x.dtype, RT.dtype, vocab_mask is not None, getattr(images, 'all_valid', None),
                      None if extra is None else tuple(extra.shape), RT.weights_epoch, RT.static_epoch)
        ent = self._igraphs.get(key)
        if ent is None:
            for k in [k for k in self._igraphs if k[-2:] != key[-2:]]:          # weights changed: those graphs hold stale copies
                self._drop_igraph(k)
            while len(self._igraphs) >= int(self.cfg.get('inference_graph_slots', 8)):      # string queries: one graph per (batch, query length)
                self._drop_igraph(next(iter(self._igraphs)))                      # least recently used first (re-inserted on every hit below)
            sx, sm, sids, sattn = x.clone(), m.clone(), ids.clone(), attn.clone()
            svm = vocab_mask.clone().float() if vocab_mask is not None else None
            sex = extra.clone() if extra is not None else None                    # (a further static input: cached BERT features)
            run = lambda: fn(NestedTensor(sx, sm, getattr(images, 'all_valid', None)), (sids, sattn), svm, sex)
            for _ in range(2):                                                    # warm-up: weight copies, kernel attributes, decoder buffers
                run()
            torch.cuda.synchronize()
            ops.release_pending()
            graph = torch.cuda.CUDAGraph()
            from .misc import capture_guard
            bside = ops.owned_stream(x.device)              # ops.Branch's side stream inside THIS graph: a hipStream of the graph's own, destroyed with it (_drop_igraph)
            with capture_guard(), torch.cuda.graph(graph, capture_error_mode='thread_local'):
                RT.branch_stream = bside
                try:
                    out = run()
                    ops.Branch.join_captured(x.device)
                except BaseException:
                    import traceback
                    traceback.print_exc()          # (the capture's teardown can abort the process before the exception surfaces)
                    raise
                finally:
                    RT.branch_stream = None
            ent = self._igraphs[key] = (graph, (sx, sm, sids, sattn, svm, sex, bside), out, self._last_q_enc)
        self._igraphs[key] = self._igraphs.pop(key)                               # most recently used last
        graph, (sx, sm, sids, sattn, svm, sex, _bside), out, self._graph_q_enc = ent        # (_graph_q_enc: the static BERT output this graph rewrites)
        sx.copy_(x); sm.copy_(m); sids.copy_(ids); sattn.copy_(attn)
        if svm is not None:
            svm.copy_(vocab_mask)
        if sex is not None:
            sex.copy_(extra)
        graph.replay()
        res = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in out.items()}     # the graph's outputs are overwritten by the next replay
        torch.cuda.current_stream().synchronize()       # see decode.py: graph launches are not left queued behind a busy GPU
        return res

    def _drop_igraph(self, key):
        """destroy one captured inference graph: the graph first, then -- the device idle -- the stream of its own it was captured on
        (ops.owned_stream: a stream that took part in the capture of a destroyed graph is never used again)"""
        ent = self._igraphs.pop(key, None)
        if ent is None:
            return
        graphs, bside = [ent[0]], ent[1][6]
        del ent
        ops.retire(graphs, [bside])

    def _forward_impl(self, images, queries, answer_token_ids, targets=None, vocab_mask=None, kv_graphs=None, query_encodings=None,
                      lang_extra=None):
        """lang_extra (uint8 [B, T_l], 1 = a query token beyond the batch's own longest query): the trainer pads queries to a few
        size classes so that one captured hipGraph serves many batches (train.FlatTrainer); those tokens are masked as keys in the
        co-attention and in the text decoder's memory, which reproduces the unpadded batch: the masked positions contribute exact zeros, so the
        result is the unpadded batch's up to the summation order of the kernels the two row counts dispatch to (bit-for-bit when they are the
        same kernels; see train.FlatTrainer._classed).  Teacher forcing only."""
        # teacher forcing: the target embedding (gather + input transform) and the vocabulary classifiers (10000 x 768 x 768) depend on the
        # answer ids / the weights only -- forked beside the co-attention stage (ops.Branch), joined where decode_text needs them
        pre = {}

        def early(outs):
            # ISSUED here whether or not there is a side stream to put it on (ops.branch_for -> None: in line, at this position): layer 0's
            # self-attention sublayer draws two dropout seeds, and drawn behind the co-attention stage in one configuration and ahead of it
            # in the other they shifted every seed in between -- other masks with GPV_COATT_BRANCH=0 than with 1 (found by the dropout-on
            # arm of test_coattention_language_branch_changes_nothing).  GPV_TEXT_EARLY=0 is the knob that moves the issue position.
            if not TEXT_EARLY:
                return
            ref = outs['pred_boxes']
            br = ops.branch_for(ref)
            if br is not None:
                br.fork()
            with (br.on() if br is not None else contextlib.nullcontext()):
                pre['target'] = self.answer_input_embedings(answer_token_ids.to(ref.device))
                pre['wc'] = self.answer_head.classifiers()
                tshape = pre['target'].shape
                pre['x0'] = self.text_decoder.layers[0].self_part(self._text_input(pre['target']), tshape[0], tshape[1])
            pre['br'] = br
        outputs, memory = self._encode(images, queries, query_encodings, lang_extra, after_detr=early if answer_token_ids is not None else None)
        B = memory.shape[0]
        dev = memory.device
        if answer_token_ids is None:                                               # greedy, gpv.py:178-196
            if not self.training and self.cfg.get('kv_decode', True):
                # KV-cached, hipGraph-captured decode step (decode.py): same outputs, 1/20th of the decoder work
                from .decode import GreedyKVDecoder
                use_graphs = self.cfg.get('kv_graphs', True) if kv_graphs is None else kv_graphs
                key = (B, memory.shape[1], str(dev), RT.dtype, use_graphs)
                dec = self._kvdec.get(key)
                if dec is None:
                    dec = self._kvdec[key] = GreedyKVDecoder(self, B, memory.shape[1], use_graphs=use_graphs)
                outputs['answer_logits'], _ = dec.decode(memory, vocab_mask)
            else:
                outputs['answer_logits'] = self.greedy_full_prefix(memory, vocab_mask)
        else:                                                                      # teacher forcing, :197-201
            if 'target' in pre:
                if pre['br'] is not None:
                    pre['br'].join()
                target = pre['target']
            else:
                target = self.answer_input_embedings(answer_token_ids.to(dev))
            mem_kpm = None
            if lang_extra is not None:                                             # memory = [vision tokens | language tokens]
                Tv = memory.shape[1] - lang_extra.shape[1]
                mem_kpm = torch.cat((torch.zeros(B, Tv, dtype=torch.uint8, device=dev), lang_extra), 1).contiguous()
            outputs['answer_logits'] = self.decode_text(target, memory, mem_kpm, wc=pre.get('wc'), x0=pre.get('x0'))[:, :-1].unsqueeze(0)
        if targets is None:
            return outputs
        return self.criterion(outputs, targets)[0]

    def greedy_full_prefix(self, memory, vocab_mask=None):
        """the reference's own greedy schedule (gpv.py:178-196): full-prefix decoder pass per generated token"""
        B, dev = memory.shape[0], memory.device
        ids = torch.full((B, 1), self.word_to_idx['__cls__'], dtype=torch.long, device=dev)
        for _ in range(self.cfg.max_text_len - 1):
            logits = self.decode_text(self.answer_input_embedings(ids), memory)[:, -1].float()
            if vocab_mask is not None:
                logits = logits + vocab_mask
            ids = torch.cat((ids, torch.topk(logits, k=1, dim=-1).indices), -1)
        logits = self.decode_text(self.answer_input_embedings(ids), memory)
        if vocab_mask is not None:
            logits = logits.float() + vocab_mask
        return logits.unsqueeze(0)

    @torch.no_grad()
    def forward_beam_search(self, images, queries, beam_size=1):
        # (no GPV_OPT_PIPE_SMALL override here: the 320-row GEMMs of beam 5 x batch 64 want the small-M configurations -- 26.0 against 36.8 ms per batch)
        """gpv.py:209-362, quirks preserved (no length normalisation, finished beams keep extending --
        the reference's `is True` test never fires --, last seqs slot never written, stable tie order)."""
        outputs = None
        if (not self.training and not torch.is_grad_enabled() and self.cfg.get('graph_inference', True)
                and self.cfg.get('kv_decode', True)):
            # the whole beam search (encoder + 19 KV-cached steps with their top-k / sort / cache reordering) has static
            # shapes: one hipGraph, like the greedy path
            queries = self._host_tokenize(images, queries)
            outputs = self._graphed(('beam', beam_size), lambda im, q, vm, qe: self._beam_device(im, q, beam_size), images, queries, None)
        if outputs is None:
            outputs = self._beam_device(images, queries, beam_size)
        seqs, seq_lp = outputs.pop('_beam_seqs'), outputs.pop('_beam_lp')
        K, B, T = seqs.shape
        seqs_c, lp = seqs.cpu(), seq_lp.exp().cpu()
        answers, probs = [], []
        for b in range(B):
            answers.append([])
            probs.append([])
            for k in range(K):
                words = []
                for t in range(T):
                    word = self.vocab[int(seqs_c[k, b, t])]
                    if word in ('__stop__', '__pad__'):
                        break
                    words.append(word)
                answers[b].append(words)
                probs[b].append(float(lp[b, k]))
        outputs['answers'], outputs['answer_probs'] = answers, probs
        return outputs

    def _beam_device(self, images, queries, beam_size):
        """device part of the beam search: outputs dict + '_beam_seqs' [K,B,T] + '_beam_lp' [B,K] (no host round trip)"""
        graphed = isinstance(queries, (tuple, list)) and len(queries) == 2 and all(torch.is_tensor(q) for q in queries) and \
            hasattr(images, 'tensors') and images.tensors.is_cuda and torch.cuda.is_current_stream_capturing()
        outputs, memory = self._encode(images, queries, self._bert_fork(images, queries) if graphed else None)
        B, K, T = memory.shape[0], beam_size, self.cfg.max_text_len
        dev = memory.device
        tok = torch.full((K, B, 1), self.word_to_idx['__cls__'], dtype=torch.long, device=dev)
        seq_lp = torch.zeros(B, K, device=dev)
        seqs = torch.zeros(K, B, T, dtype=torch.long, device=dev)
        memK = memory.repeat(K, 1, 1)                                              # all K beams in ONE decoder pass
        kv = None
        if not self.training and not torch.is_grad_enabled() and self.cfg.get('kv_decode', True):
            # KV-cached decode step (decode.py) instead of the reference's full-prefix pass per token: the decoder is
            # causal, so the logits of the newest position are the same; the caches follow the beams (reorder below)
            from .decode import GreedyKVDecoder
            key = (K * B, memK.shape[1], str(dev), RT.dtype, 'beam')
            kv = self._kvdec.get(key)
            if kv is None:
                kv = self._kvdec[key] = GreedyKVDecoder(self, K * B, memK.shape[1], use_graphs=False)
            kv.memory.copy_(memK.reshape(K * B * memK.shape[1], -1))
            kv._prepare()
        for t in range(T - 1):
            if kv is not None:
                kv.tok.copy_(tok[:, :, -1].reshape(K * B))
                logits = kv._step_core(t).float()
            else:
                logits = self.decode_text(self.answer_input_embedings(tok.reshape(K * B, -1)), memK)[:, -1].float()
            top = torch.log_softmax(logits, -1).topk(K, -1)                        # [K*B, K]
            vals, last = top.values.view(K, B, K), top.indices.view(K, B, K)
            scores = (seq_lp.t().unsqueeze(-1) + vals).permute(1, 0, 2).contiguous()   # [B, K1, K2]
            if t == 0:
                scores[:, 1:] = scores[:, 1:] * 0 - 1e9
            flat = scores.view(B, K * K)
            order = torch.sort(flat, dim=1, descending=True, stable=True).indices[:, :K]   # [B,K]
            k1, k2 = order // K, order % K
            bi = torch.arange(B, device=dev).unsqueeze(1).expand(B, K)
            w = last[k1, bi, k2]                                                   # [B,K]
            seq_lp = flat.gather(1, order)
            new_tok = torch.cat((tok[k1, bi], w.unsqueeze(-1)), -1).permute(1, 0, 2).contiguous()
            new_seqs = seqs[k1, bi].permute(1, 0, 2).contiguous()
            new_seqs[:, :, t] = w.t()
            tok, seqs = new_tok, new_seqs
            if kv is not None:                                                     # slot (k, b) continues parent (k1[b,k], b)
                kv.reorder((k1.t() * B + bi.t()).reshape(K * B), t + 1)
        outputs['_beam_seqs'], outputs['_beam_lp'] = seqs, seq_lp
        return outputs

    def encode_answers(self, targets):
        """gpv.py:377-430 (generation: tokenised sentence with __cls__ / __stop__ / __pad__; classification: [__cls__, answer])"""
        padded_inputs, ids = self._encode_answers_host(targets)
        dev = self.vision_token.device
        from .misc import STAGER
        return padded_inputs, STAGER.to_device(ids, torch.long, dev)          # no host<->device sync (misc.PinnedStager)

    def _encode_answers_host(self, targets):
        """-> (token strings per sample padded to the batch's longest, their vocabulary ids as host lists)"""
        answers = [t.get('answer', '') for t in targets]
        if self.cfg.answering_type == 'classification':
            # gpv.py:384-399: the answer is ONE vocabulary entry looked up as it is (no tokenisation, no lower-casing, no __stop__)
            padded_inputs = [['__cls__', a] for a in answers]
            return padded_inputs, [[self.word_to_idx.get(w, self.word_to_idx['__unk__']) for w in toks] for toks in padded_inputs]
        padded_inputs, S = [], 0
        for a in answers:
            sent = '__cls__ __stop__' if a == '' else f'__cls__ {a} __stop__'
            padded_inputs.append([w.lower() for w in _word_tokenize(sent)])
            S = max(S, len(padded_inputs[-1]))
        ids = []
        for toks in padded_inputs:
            toks.extend(['__pad__'] * (S - len(toks)))
            ids.append([self.word_to_idx.get(w, self.word_to_idx['__unk__']) for w in toks][:self.cfg.max_text_len])
        return padded_inputs, ids

    def encode_answers_classed(self, targets, multiple=4):
        """encode_answers (gpv.py:377-430) with the token axis padded up to a multiple of `multiple` (<= max_text_len):
        -> (input token ids [B, S_c] padded with __pad__, CE targets [B, S_c - 1] with -100 beyond the batch's own length S).
        The decoder is causal and the cross-entropy ignores -100 rows, so logits, loss and gradients of the first S - 1 positions
        are those of the unpadded batch (pad positions INSIDE the batch's own length stay targets, as in the reference)."""
        _, padded = self._encode_answers_host(targets)
        S = len(padded[0])
        Sc = min(self.cfg.max_text_len, -(-S // multiple) * multiple)
        pad = self.word_to_idx['__pad__']
        ids = [row + [pad] * (Sc - len(row)) for row in padded]
        tgt = [row[1:] + [-100] * (Sc - len(row)) for row in padded]
        dev = self.vision_token.device
        from .misc import STAGER
        return STAGER.to_device(ids, torch.long, dev), STAGER.to_device(tgt, torch.long, dev), S

    def token_ids_to_words(self, token_ids):
        B, S = token_ids.shape
        return [[self.vocab[int(token_ids[i, j])] for j in range(S)] for i in range(B)]

    @property
    def cls_token(self):
        dev = self.vision_token.device
        return self.answer_input_embedings(torch.tensor([self.word_to_idx['__cls__']], device=dev))[0]
