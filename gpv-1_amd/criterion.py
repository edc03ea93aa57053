"""Host-side criterion: box algebra, Hungarian matcher, DETR set criterion and the task-filtered GPV
criterion.  north_star keeps these on the host ("Hungarian matching and set_criterion kept on
host"): they are small fp32 torch ops on (B,100,{2,4}) tensors plus scipy's LSAP; the only heavy
piece, the vocabulary cross-entropy, is the HIP kernel (ops.softmax_ce).

Reference: utils/box_ops.py, utils/matcher.py, utils/set_criterion.py, exp/gpv/models/losses.py.
"""
import torch
import torch.nn as nn
from scipy.optimize import linear_sum_assignment

from . import ops


# ---------------------------------------------------------------- box_ops.py:9-59
def box_cxcywh_to_xyxy(x):
    cx, cy, w, h = x.unbind(-1)
    return torch.stack((cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h), dim=-1)


def box_xyxy_to_cxcywh(x):
    x0, y0, x1, y1 = x.unbind(-1)
    return torch.stack(((x0 + x1) / 2, (y0 + y1) / 2, x1 - x0, y1 - y0), dim=-1)


def box_area(b):
    return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])


def box_iou(b1, b2):
    a1, a2 = box_area(b1), box_area(b2)
    lt = torch.max(b1[:, None, :2], b2[:, :2])
    rb = torch.min(b1[:, None, 2:], b2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    union = a1[:, None] + a2 - inter
    return inter / union, union


def generalized_box_iou(b1, b2):
    """pairwise (N,M) GIoU of xyxy boxes; degenerate boxes assert like box_ops.py:49-50."""
    assert (b1[:, 2:] >= b1[:, :2]).all()
    assert (b2[:, 2:] >= b2[:, :2]).all()
    iou, union = box_iou(b1, b2)
    lt = torch.min(b1[:, None, :2], b2[:, :2])
    rb = torch.max(b1[:, None, 2:], b2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    area = wh[:, :, 0] * wh[:, :, 1]
    return iou - (area - union) / area


# ---------------------------------------------------------------- matcher.py:32-77
class HungarianMatcher(nn.Module):
    def __init__(self, cost_class: float = 1, cost_bbox: float = 1, cost_giou: float = 1):
        super().__init__()
        assert cost_class != 0 or cost_bbox != 0 or cost_giou != 0, "all costs cant be 0"
        self.cost_class, self.cost_bbox, self.cost_giou = cost_class, cost_bbox, cost_giou

    @torch.no_grad()
    def forward(self, outputs, targets):
        logits = outputs['pred_relevance_logits'].float()
        bs, nq = logits.shape[:2]
        prob = logits.flatten(0, 1).softmax(-1)
        boxes = outputs['pred_boxes'].float().flatten(0, 1)
        tgt_ids = torch.cat([t['labels'] for t in targets])
        tgt_boxes = torch.cat([t['boxes'] for t in targets]).float()
        c_class = -prob[:, tgt_ids]
        c_bbox = torch.cdist(boxes, tgt_boxes, p=1)
        c_giou = -generalized_box_iou(box_cxcywh_to_xyxy(boxes), box_cxcywh_to_xyxy(tgt_boxes))
        cost = (self.cost_bbox * c_bbox + self.cost_class * c_class + self.cost_giou * c_giou).view(bs, nq, -1).cpu()
        sizes = [len(t['boxes']) for t in targets]
        out = []
        for i, c in enumerate(cost.split(sizes, -1)):
            r, cidx = linear_sum_assignment(c[i])
            out.append((torch.as_tensor(r, dtype=torch.int64), torch.as_tensor(cidx, dtype=torch.int64)))
        return out


# ---------------------------------------------------------------- set_criterion.py:44-97,150-191
class SetCriterion(nn.Module):
    def __init__(self, num_classes, matcher, weight_dict, eos_coef, losses):
        super().__init__()
        self.num_classes, self.matcher, self.weight_dict = num_classes, matcher, weight_dict
        self.eos_coef, self.losses = eos_coef, losses
        empty_weight = torch.ones(num_classes + 1)
        empty_weight[-1] = eos_coef
        self.register_buffer('empty_weight', empty_weight)

    @staticmethod
    def _src_idx(indices):
        batch_idx = torch.cat([torch.full_like(src, i) for i, (src, _) in enumerate(indices)])
        return batch_idx, torch.cat([src for src, _ in indices])

    def loss_labels(self, outputs, targets, indices, num_boxes):
        logits = outputs['pred_relevance_logits'].float()
        idx = self._src_idx(indices)
        dev = logits.device
        tgt_o = torch.cat([t['labels'][j.to(t['labels'].device)] for t, (_, j) in zip(targets, indices)]).to(dev)
        tgt = torch.full(logits.shape[:2], self.num_classes, dtype=torch.int64, device=dev)
        tgt[(idx[0].to(dev), idx[1].to(dev))] = tgt_o
        ew = self.empty_weight.to(dev)
        return {'loss_ce': nn.functional.cross_entropy(logits.transpose(1, 2), tgt, ew)}

    def loss_boxes(self, outputs, targets, indices, num_boxes):
        dev = outputs['pred_boxes'].device
        idx = self._src_idx(indices)
        src = outputs['pred_boxes'].float()[(idx[0].to(dev), idx[1].to(dev))]
        tgt = torch.cat([t['boxes'][i.to(t['boxes'].device)] for t, (_, i) in zip(targets, indices)], dim=0).float().to(dev)
        l1 = nn.functional.l1_loss(src, tgt, reduction='none').sum() / num_boxes
        giou = torch.diag(generalized_box_iou(box_cxcywh_to_xyxy(src), box_cxcywh_to_xyxy(tgt)))
        return {'loss_bbox': l1, 'loss_giou': (1 - giou).sum() / num_boxes}

    def forward(self, outputs, targets):
        base = {k: v for k, v in outputs.items() if k != 'aux_outputs'}
        indices = self.matcher(base, targets)
        # NOTE: local normalisation -- the cross-rank all_reduce of num_boxes is commented out in the
        # reference (set_criterion.py:165-168); reproduced as is.
        num_boxes = max(float(sum(len(t['labels']) for t in targets)), 1.0)
        fns = {'labels': self.loss_labels, 'boxes': self.loss_boxes}
        losses = {}
        for name in self.losses:
            losses.update(fns[name](outputs, targets, indices, num_boxes))
        if 'aux_outputs' in outputs:
            for i, aux in enumerate(outputs['aux_outputs']):
                ind = self.matcher(aux, targets)
                for name in self.losses:
                    losses.update({f'{k}_{i}': v for k, v in fns[name](aux, targets, ind, num_boxes).items()})
        self.last_indices = indices
        return losses


# ---------------------------------------------------------------- losses.py
class AnswerClassification(nn.Module):
    task = None
    key = 'loss_answer'

    def __init__(self, cfg):
        super().__init__()
        self.ignore_index = -100 if cfg.pad_idx is None else cfg.pad_idx

    def compute_ce_loss(self, logits, tgts):
        """logits (L,B',S,V) compute dtype; tgts list of (S,) int64.  losses.py:20-26:
        CE(reduction none) -> mean over batch, sum over positions and layers."""
        L, Bn, S, V = logits.shape
        t = torch.stack(tgts).to(logits.device)                           # (B',S)
        t = t.view(1, Bn, S).expand(L, Bn, S).reshape(-1)
        if self.ignore_index != -100:
            t = torch.where(t == self.ignore_index, torch.full_like(t, -100), t)
        rows = ops.softmax_ce(logits.reshape(L * Bn * S, V), t)          # HIP kernel, fp32 per-row loss
        return rows.view(L, Bn, S).mean(1).sum()

    def forward(self, outputs, targets):
        sel = [i for i, t in enumerate(targets) if 'answer' in t and (self.task is None or t['task'] == self.task)]
        if not sel:
            return {self.key: None}
        logits = outputs['answer_logits']
        if len(sel) != logits.shape[1]:                         # indexing with a host list = H2D copy + sync: only when a subset
            logits = logits[:, sel]
        return {self.key: self.compute_ce_loss(logits, [targets[i]['answer_token_ids'] for i in sel])}


class CaptionLoss(AnswerClassification):
    task, key = 'CocoCaptioning', 'loss_caption'


class VqaLoss(AnswerClassification):
    task, key = 'CocoVqa', 'loss_vqa'


class ClsLoss(AnswerClassification):
    task, key = 'CocoClassification', 'loss_cls'


class Localization(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.matcher = HungarianMatcher(cost_class=cfg.cost_wts.ce, cost_bbox=cfg.cost_wts.bbox,
                                        cost_giou=cfg.cost_wts.giou)
        self.set_criterion = SetCriterion(num_classes=cfg.num_classes, matcher=self.matcher, weight_dict=None,
                                          eos_coef=cfg.eos_coef, losses=['labels', 'boxes'])

    def forward(self, outputs, targets):
        sel = [i for i, t in enumerate(targets) if 'boxes' in t]
        if not sel:
            return {'loss_ce': None, 'loss_bbox': None, 'loss_giou': None}
        fo = {'pred_relevance_logits': outputs['pred_relevance_logits'][sel], 'pred_boxes': outputs['pred_boxes'][sel]}
        if 'aux_outputs' in outputs:
            fo['aux_outputs'] = [{'pred_relevance_logits': a['pred_relevance_logits'][sel], 'pred_boxes': a['pred_boxes'][sel]}
                                 for a in outputs['aux_outputs']]
        losses = self.set_criterion(fo, [targets[i] for i in sel])
        ret = {'loss_ce': 0, 'loss_bbox': 0, 'loss_giou': 0}
        for name in ret:
            for k, v in losses.items():
                if name in k:
                    ret[name] = ret[name] + v
        return ret


_LOSS_MODULES = {'CaptionLoss': CaptionLoss, 'VqaLoss': VqaLoss, 'ClsLoss': ClsLoss, 'Localization': Localization,
                 'AnswerClassification': AnswerClassification}


class GPVCriterion(nn.Module):
    """losses.py:141-176"""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.criterion_names = []
        self.loss_wts = {}
        for module_name, loss_cfg in cfg.items():
            setattr(self, loss_cfg.name, _LOSS_MODULES[module_name](loss_cfg))
            self.criterion_names.append(loss_cfg.name)
            self.loss_wts.update(loss_cfg.loss_wts)

    def forward(self, outputs, targets):
        loss_dict = {}
        for name in self.criterion_names:
            loss_dict.update(getattr(self, name)(outputs, targets))
        if all(v is None for v in loss_dict.values()):
            return None, loss_dict
        total = 0
        for k, wt in self.loss_wts.items():
            if loss_dict[k] is not None:
                total = total + wt * loss_dict[k]
        return total, loss_dict
