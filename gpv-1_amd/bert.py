"""Frozen BERT-base query encoder (reference: exp/gpv/models/bert.py:5-22, HF transformers 3.0.2
BertTokenizer + BertModel('bert-base-uncased'), always run under no_grad: gpv.py:142-145).

Own implementation on the HIP kernels with the HF parameter names (``bert.model.embeddings...``,
``bert.model.encoder.layer.N...``, ``bert.model.pooler.dense``) so a reference checkpoint loads.
Reference quirk kept: the module lives inside GPV, so ``model.train()`` turns its 0.1 dropouts ON
even though it is only a feature extractor.

Tokenisation: a WordPiece tokenizer is included (needs the ``vocab.txt`` of bert-base-uncased, which
is not redistributable here / not available offline).  ``forward`` therefore also accepts
pre-tokenised input: a ``(input_ids, attention_mask)`` pair of int64 tensors.
"""
import os
import unicodedata

import torch
import torch.nn as nn

from . import hip, ops
from .transformer import LinearP, LayerNormP

BERT_NO_PIPE_SMALL = os.environ.get('GPV_BERT_NO_PIPE_SMALL', '1') != '0'


class WordPieceTokenizer:
    """bert-base-uncased BasicTokenizer (lower-case, strip accents, split punctuation) + greedy
    longest-match WordPiece; padding=True semantics of the HF call in bert.py:12-15."""

    def __init__(self, vocab_file):
        with open(vocab_file, encoding='utf-8') as f:
            self.vocab = {w.rstrip('\n'): i for i, w in enumerate(f)}
        self.unk, self.cls, self.sep, self.pad = (self.vocab[t] for t in ('[UNK]', '[CLS]', '[SEP]', '[PAD]'))
        self.SPECIALS = tuple(t for t in self.SPECIALS if t in self.vocab)

    @staticmethod
    def _is_punct(ch):
        cp = ord(ch)
        if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126:
            return True
        return unicodedata.category(ch).startswith('P')

    SPECIALS = ('[UNK]', '[SEP]', '[PAD]', '[CLS]', '[MASK]')

    @staticmethod
    def _is_cjk(cp):
        return (0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0x20000 <= cp <= 0x2A6DF or 0x2A700 <= cp <= 0x2B73F or
                0x2B740 <= cp <= 0x2B81F or 0x2B820 <= cp <= 0x2CEAF or 0xF900 <= cp <= 0xFAFF or 0x2F800 <= cp <= 0x2FA1F)

    def basic(self, text):
        """HF BasicTokenizer(do_lower_case=True): clean (drop NUL / U+FFFD / control characters, any whitespace -> space), put
        spaces around CJK characters, lower-case, NFD + strip combining marks, split on whitespace and punctuation"""
        cleaned = []
        for ch in text:
            cp = ord(ch)
            if cp == 0 or cp == 0xFFFD:
                continue
            if ch in '\t\n\r':
                cleaned.append(' ')
                continue
            cat = unicodedata.category(ch)
            if cat.startswith('C'):
                continue
            if cat == 'Zs':
                cleaned.append(' ')
            elif self._is_cjk(cp):
                cleaned.append(' ' + ch + ' ')
            else:
                cleaned.append(ch)
        text = unicodedata.normalize('NFD', ''.join(cleaned).lower())
        out, cur = [], ''
        for ch in text:
            if unicodedata.category(ch) == 'Mn':
                continue
            if ch.isspace():
                if cur:
                    out.append(cur)
                    cur = ''
            elif self._is_punct(ch):
                if cur:
                    out.append(cur)
                    cur = ''
                out.append(ch)
            else:
                cur += ch
        if cur:
            out.append(cur)
        return out

    def tokenize_ids(self, text):
        """special tokens written in the text are kept whole (HF splits the text on them before anything else)"""
        import re
        ids = []
        for part in re.split('(' + '|'.join(re.escape(t) for t in self.SPECIALS) + ')', text):
            if part in self.SPECIALS:
                ids.append(self.vocab[part])
            elif part:
                ids.extend(i for w in self.basic(part) for i in self.wordpiece(w))
        return ids

    def wordpiece(self, word):
        if len(word) > 100:
            return [self.unk]
        ids, start = [], 0
        while start < len(word):
            end, cur = len(word), None
            while start < end:
                sub = ('##' if start > 0 else '') + word[start:end]
                if sub in self.vocab:
                    cur = self.vocab[sub]
                    break
                end -= 1
            if cur is None:
                return [self.unk]
            ids.append(cur)
            start = end
        return ids

    def __call__(self, sentences):
        seqs = [[self.cls] + self.tokenize_ids(s) + [self.sep] for s in sentences]
        T = max(len(s) for s in seqs)
        ids = torch.full((len(seqs), T), self.pad, dtype=torch.long)
        attn = torch.zeros(len(seqs), T, dtype=torch.long)
        for i, s in enumerate(seqs):
            ids[i, :len(s)] = torch.tensor(s)
            attn[i, :len(s)] = 1
        return ids, attn


class _SelfAttention(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.query, self.key, self.value = LinearP(d, d), LinearP(d, d), LinearP(d, d)


class _SelfOutput(nn.Module):
    def __init__(self, din, d):
        super().__init__()
        self.dense = LinearP(din, d)
        self.LayerNorm = LayerNormP(d, eps=1e-12)


class _Attention(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.self = _SelfAttention(d)
        self.output = _SelfOutput(d, d)


class _Dense(nn.Module):
    def __init__(self, din, dout):
        super().__init__()
        self.dense = LinearP(din, dout)


class _Layer(nn.Module):
    def __init__(self, d, inter):
        super().__init__()
        self.attention = _Attention(d)
        self.intermediate = _Dense(d, inter)
        self.output = _SelfOutput(inter, d)


class _Embeddings(nn.Module):
    def __init__(self, vocab, d, max_pos, types):
        super().__init__()
        self.word_embeddings = nn.Embedding(vocab, d)
        self.position_embeddings = nn.Embedding(max_pos, d)
        self.token_type_embeddings = nn.Embedding(types, d)
        self.LayerNorm = LayerNormP(d, eps=1e-12)


class _Encoder(nn.Module):
    def __init__(self, n, d, inter):
        super().__init__()
        self.layer = nn.ModuleList([_Layer(d, inter) for _ in range(n)])


class BertModel(nn.Module):
    def __init__(self, vocab_size=30522, hidden=768, layers=12, heads=12, intermediate=3072, max_pos=512, dropout=0.1):
        super().__init__()
        self.embeddings = _Embeddings(vocab_size, hidden, max_pos, 2)
        self.encoder = _Encoder(layers, hidden, intermediate)
        self.pooler = _Dense(hidden, hidden)               # present in the checkpoint, unused by GPV (outputs[0])
        self.heads, self.hidden, self.p = heads, hidden, dropout
        for m in self.modules():                           # HF init: N(0, 0.02)
            if isinstance(m, (LinearP, nn.Embedding)):
                nn.init.normal_(m.weight, std=0.02)
            if isinstance(m, LinearP) and m.bias is not None:
                nn.init.zeros_(m.bias)

    @staticmethod
    def _qkv(sa):
        """packed query/key/value projection of one layer in the compute dtype (BERT is frozen: cached per static epoch)"""
        RT = ops.RT
        key = ('bert_qkv', id(sa), RT.dtype)
        hit = RT.cache.get(key)
        if hit is not None and hit[0] == RT.static_epoch:
            return hit[1], hit[2]
        w = torch.cat([sa.query.weight, sa.key.weight, sa.value.weight]).detach().to(RT.dtype).contiguous()
        b = torch.cat([sa.query.bias, sa.key.bias, sa.value.bias]).detach().float().contiguous()
        RT.cache[key] = (RT.static_epoch, w, b)
        return w, b

    @torch.no_grad()
    def forward(self, input_ids, attention_mask):
        # The frozen BERT's GEMMs (B x T = 192 rows in the training step) are nodes of a dependent chain on their graph branch: timed that way
        # (tools/tune_gemms.py --chain) they are 1.4 - 2.7 us shorter WITHOUT gemm_pipe.hip's small-M configurations (7.4 -> 4.7 us for
        # 192 x 768 x 768; back-to-back stream launches, which overlap their ramps, had preferred them): 96 us less on the branch per step
        if BERT_NO_PIPE_SMALL and input_ids.is_cuda:
            with hip.gemm_flags(hip.GEMM_NO_PIPE_SMALL):
                return self._forward(input_ids, attention_mask)
        return self._forward(input_ids, attention_mask)

    def _forward(self, input_ids, attention_mask):
        B, T = input_ids.shape
        D, H = self.hidden, self.heads
        p = self.p if self.training else 0.0
        e = self.embeddings
        x = ops.embedding(e.word_embeddings.weight, input_ids).reshape(B * T, D)
        pt = (e.position_embeddings.weight[:T] + e.token_type_embeddings.weight[0]).to(ops.RT.dtype).contiguous()
        x = ops.add(x, pt)                                                       # rows broadcast over the batch
        x = e.LayerNorm(x)
        if p > 0:
            y = torch.empty_like(x)
            hip.dropout(x, y, x.numel(), p, ops.RT.next_seed())
            x = y
        kpm = (attention_mask == 0).to(torch.uint8).contiguous()
        for l in self.encoder.layer:
            sa = l.attention.self
            wqkv, bqkv = self._qkv(sa)                     # one [3D, D] GEMM instead of three (frozen weights: cached)
            qkv = torch.empty(B * T, 3 * D, device=x.device, dtype=ops.RT.dtype)
            hip.gemm(x, wqkv, qkv, B * T, 3 * D, D, D, D, 3 * D, bias=bqkv)
            a = ops.attention([qkv], ((0, 0), (0, D), (0, 2 * D)), B, H, T, T, D // H, kpm=kpm, drop_p=p)
            x = l.attention.output.LayerNorm(x, l.attention.output.dense(a), p)
            h = l.intermediate.dense(x, ops.ACT_GELU)
            x = l.output.LayerNorm(x, l.output.dense(h), p)
        return x.reshape(B, T, D)


class Bert(nn.Module):
    """bert.py:5-22.  ``forward(sentences, device)`` -> (last hidden state B x T x 768, token_inputs)."""

    def __init__(self, cfg=None, vocab_file=None, num_layers=12, weights=None):
        super().__init__()
        vocab_file = vocab_file or os.environ.get('GPV_BERT_VOCAB')
        self.tokenizer = WordPieceTokenizer(vocab_file) if vocab_file and os.path.exists(vocab_file) else None
        self.model = BertModel(layers=num_layers)
        weights = weights or os.environ.get('GPV_BERT_WEIGHTS')
        self.pretrained = False
        if weights:
            self.load_pretrained(weights)

    def load_pretrained(self, path):
        """bert.py:8-9 `BertModel.from_pretrained('bert-base-uncased')`: the reference downloads the weights; here they come
        from a local file -- a torch-saved HF state dict (pytorch_model.bin), a directory holding one, or a .safetensors file.
        Key names are HF's (with or without the 'bert.' prefix); LayerNorm gamma/beta of old checkpoints are renamed."""
        if os.path.isdir(path):
            cands = [os.path.join(path, f) for f in ('model.safetensors', 'pytorch_model.bin')]
            path = next((c for c in cands if os.path.exists(c)), None)
            if path is None:
                raise FileNotFoundError(f'Bert.load_pretrained: none of {cands} exists')
        if path.endswith('.safetensors'):
            from safetensors.torch import load_file
            sd = load_file(path)
        else:
            sd = torch.load(path, map_location='cpu', weights_only=True)
        ren = {}
        for k, v in sd.items():
            k = k[len('bert.'):] if k.startswith('bert.') else k
            k = k.replace('LayerNorm.gamma', 'LayerNorm.weight').replace('LayerNorm.beta', 'LayerNorm.bias')
            ren[k] = v
        own = self.model.state_dict()
        missing = [k for k in own if k not in ren and 'position_ids' not in k]
        if missing:
            raise RuntimeError(f'Bert.load_pretrained: {len(missing)} tensors missing from {path}, e.g. {missing[:3]}')
        self.model.load_state_dict({k: ren[k] for k in own if k in ren}, strict=False)
        self.pretrained = True
        ops.RT.bump_weights()

    def forward(self, sentences, device=None):
        if isinstance(sentences, (tuple, list)) and len(sentences) == 2 and torch.is_tensor(sentences[0]):
            ids, attn = sentences
        else:
            if self.tokenizer is None:
                raise RuntimeError('Bert: no WordPiece vocabulary (set GPV_BERT_VOCAB=/path/to/bert-base-uncased/vocab.txt) '
                                   '-- or pass pre-tokenised (input_ids, attention_mask) tensors as `queries`.')
            ids, attn = self.tokenizer(list(sentences))
        dev = self.model.embeddings.word_embeddings.weight.device
        ids, attn = ids.to(dev), attn.to(dev)
        return self.model(ids, attn), {'input_ids': ids, 'attention_mask': attn}
