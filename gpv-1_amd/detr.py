"""DETR visual stream with the RoI head (reference: exp/gpv/models/detr_roi_head.py:20-133; the
``roi_head: False`` variant exp/gpv/models/detr.py:19-73 is the same module without RoI pooling).

backbone (NHWC c5) -> input_proj (1x1 conv == GEMM over the c5 rows) -> transformer -> class / box heads
(fp32 outputs, they feed the matcher) -> RoIAlign(7x7)+mean as a batched GEMM with separable bilinear
weights -> LayerNorm (no affine) -> concat with the decoder states (2048 + 256 = 2304).
"""
import os

import torch
import torch.nn as nn

from . import ops
from .ops import W
from .backbone import build_backbone
from .transformer import build_transformer, LinearP
from .misc import NestedTensor, nested_tensor_from_tensor_list


class MLP(nn.Module):
    """detr_roi_head.py:105-117"""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(LinearP(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))

    def forward(self, x):
        for i, layer in enumerate(self.layers):
            last = i == self.num_layers - 1
            x = layer(x, ops.ACT_NONE if last else ops.ACT_RELU, out_f32=last)
        return x


BOUNDARY_BELOW_ROI = os.environ.get('GPV_BOUNDARY', 'roi') == 'hs'


class Conv1x1P(nn.Module):
    """nn.Conv2d(cin, cout, kernel_size=1) parameters; evaluated as a GEMM over NHWC rows."""

    def __init__(self, cin, cout):
        super().__init__()
        ref = nn.Conv2d(cin, cout, kernel_size=1)
        self.weight = nn.Parameter(ref.weight.detach().clone())
        self.bias = nn.Parameter(ref.bias.detach().clone())

    def forward(self, rows):
        return ops.linear(rows, W(self.weight, self.bias))


class DETR(nn.Module):
    def __init__(self, backbone, transformer, num_classes, num_queries, last_layer_only, aux_loss=False, roi_head=True):
        super().__init__()
        self.num_queries = num_queries
        self.transformer = transformer
        hidden_dim = transformer.d_model
        self.class_embed = LinearP(hidden_dim, num_classes + 1)
        self.bbox_embed = MLP(hidden_dim, hidden_dim, 4, 3)
        self.query_embed = nn.Embedding(num_queries, hidden_dim)
        self.input_proj = Conv1x1P(backbone.num_channels, hidden_dim)
        self.backbone = backbone
        self.last_layer_only = last_layer_only
        self.aux_loss = aux_loss
        self.roi_head = roi_head

    def forward(self, samples):
        if isinstance(samples, (list, torch.Tensor)):
            samples = nested_tensor_from_tensor_list(samples)
        features, pos = self.backbone(samples)
        c5, mask = features[-1].decompose()                   # c5 [B,h,w,C] NHWC
        B, h, w, C = c5.shape
        rows = c5.reshape(B, h * w, C)
        src = self.input_proj(rows)                            # [B,S,256]
        need_all = not (self.last_layer_only is True or self.training is not True)
        outs, _ = self.transformer(src, mask.flatten(1), self.query_embed.weight, pos[-1], need_all or self.aux_loss)
        # (backward: when the gradient arrives here, the text decoder, the co-attention, the heads and the RoI head have run --
        #  train.GraphedBody launches their grouped weight gradients on a side branch from this point, beside the DETR layers.
        #  Placed after the RoI head it made that head's LDS-heavy backward kernels wait for the grouped GEMM's blocks:
        #  LayerNorm backward 20 -> 160 us, pooling backward 60 -> 155 us.)
        hs = torch.stack(outs)                                 # [L,B,Q,D]
        if BOUNDARY_BELOW_ROI:
            hs = ops.boundary(hs, 'detr')
        if not need_all:
            hs = hs[-1:]
        outputs_class = self.class_embed(hs, out_f32=True)     # fp32 logits
        outputs_coord = self.bbox_embed(hs).sigmoid()          # fp32 boxes
        out = {'pred_relevance_logits': outputs_class[-1], 'pred_boxes': outputs_coord[-1], 'detr_hs': hs}
        if self.aux_loss:
            out['aux_outputs'] = [{'pred_relevance_logits': a, 'pred_boxes': b}
                                  for a, b in zip(outputs_class[:-1], outputs_coord[:-1])]
        if self.roi_head:
            # boxes are detached: torchvision's roi_align has no gradient w.r.t. the boxes, so in a batch without
            # box targets the bbox MLP receives no gradient at all (and is not touched by the optimizer)
            roi = ops.roi_pool(rows, out['pred_boxes'].detach(), h, w)     # [B,Q,2048]
            roi = ops.add_layernorm(roi, None, None, None, 1e-5)           # F.layer_norm, no affine (:91)
            out['detr_hs'] = torch.cat((roi.unsqueeze(0).to(hs.dtype), hs), -1)
        return out


def create_detr_roi_head(cfg):
    return DETR(build_backbone(cfg), build_transformer(cfg), num_classes=cfg.num_classes, num_queries=cfg.num_queries,
                last_layer_only=cfg.last_layer_only, aux_loss=cfg.aux_loss, roi_head=True)


def create_detr(cfg):
    return DETR(build_backbone(cfg), build_transformer(cfg), num_classes=cfg.num_classes, num_queries=cfg.num_queries,
                last_layer_only=cfg.last_layer_only, aux_loss=cfg.aux_loss, roi_head=False)
