"""2-D sine position embedding (reference: exp/gpv/models/position_encoding.py:12-48).

The embedding depends only on the padding mask; at GPV's fixed 480x640 input the mask is all-False
and the result is a constant, so it is computed once per (B,h,w) and cached.  Returned in
the row layout the transformer kernels use: [B, h*w, 2*num_pos_feats] in the compute dtype.
"""
import math

import torch
import torch.nn as nn

from .ops import RT


class PositionEmbeddingSine(nn.Module):
    def __init__(self, num_pos_feats=64, temperature=10000, normalize=False, scale=None):
        super().__init__()
        if scale is not None and normalize is False:
            raise ValueError("normalize should be True if scale is passed")
        self.num_pos_feats, self.temperature, self.normalize = num_pos_feats, temperature, normalize
        self.scale = 2 * math.pi if scale is None else scale
        self._cache = {}

    def compute(self, mask):
        not_mask = (~mask).float()
        y_embed = not_mask.cumsum(1)
        x_embed = not_mask.cumsum(2)
        if self.normalize:
            eps = 1e-6
            y_embed = y_embed / (y_embed[:, -1:, :] + eps) * self.scale
            x_embed = x_embed / (x_embed[:, :, -1:] + eps) * self.scale
        i = torch.arange(self.num_pos_feats, dtype=torch.float32, device=mask.device)
        dim_t = self.temperature ** (2 * torch.div(i, 2, rounding_mode='floor') / self.num_pos_feats)
        px = x_embed[..., None] / dim_t
        py = y_embed[..., None] / dim_t
        px = torch.stack((px[..., 0::2].sin(), px[..., 1::2].cos()), dim=4).flatten(3)
        py = torch.stack((py[..., 0::2].sin(), py[..., 1::2].cos()), dim=4).flatten(3)
        return torch.cat((py, px), dim=3)                     # [B,h,w,2*npf]  (channels last)

    def forward(self, tensor_list):
        mask = tensor_list.mask
        assert mask is not None
        B, h, w = mask.shape
        hint = getattr(tensor_list, 'all_valid', None)
        key = (B, h, w, str(mask.device), RT.dtype)
        if mask.is_cuda and torch.cuda.is_current_stream_capturing():        # no host round trip inside a graph capture
            if hint is True and key in self._cache:                          # host-known all-valid batch: the constant of the eager steps
                return self._cache[key]                                      # (the hint is part of every graph's signature) -- 40 launches less per replay
            return self.compute(mask).reshape(B, h * w, -1).to(RT.dtype).contiguous()
        plain = hint if hint is not None else not bool(mask.any())      # the device round trip only when nothing is known on the host
        if plain and key in self._cache:
            return self._cache[key]
        pos = self.compute(mask).reshape(B, h * w, -1).to(RT.dtype).contiguous()
        if plain:
            self._cache[key] = pos
        return pos


class PositionEmbeddingLearned(nn.Module):
    """position_encoding.py:50-75 (`position_embedding: learned | v3`; no shipped GPV-1 config): a learned row and a learned column
    table of 50 entries each, the same grid for every image (the padding mask is not looked at).  Returned like the sine one --
    [B, h*w, 2*num_pos_feats] rows in the compute dtype -- but WITH its autograd history: while it requires a gradient the DETR
    transformer adds it with element-wise launches autograd can see (transformer.Transformer._forward_plain) instead of taking it
    as a constant of the LayerNorm kernels' second output."""

    def __init__(self, num_pos_feats=256):
        super().__init__()
        self.row_embed = nn.Embedding(50, num_pos_feats)
        self.col_embed = nn.Embedding(50, num_pos_feats)
        nn.init.uniform_(self.row_embed.weight)
        nn.init.uniform_(self.col_embed.weight)

    def forward(self, tensor_list):
        B, h, w = tensor_list.mask.shape
        if h > 50 or w > 50:
            raise ValueError(f'learned position embedding: {h} x {w} feature map, the tables hold 50 rows / columns')
        x_emb, y_emb = self.col_embed.weight[:w], self.row_embed.weight[:h]
        pos = torch.cat((x_emb.unsqueeze(0).expand(h, w, -1), y_emb.unsqueeze(1).expand(h, w, -1)), -1)      # [h, w, C]: (column | row) halves
        return pos.reshape(1, h * w, -1).expand(B, h * w, -1).to(RT.dtype)


def build_position_encoding(args):
    n_steps = args.hidden_dim // 2
    if args.position_embedding in ('v2', 'sine'):
        return PositionEmbeddingSine(n_steps, normalize=True)
    if args.position_embedding in ('v3', 'learned'):
        return PositionEmbeddingLearned(n_steps)
    raise ValueError(f"not supported {args.position_embedding}")
