"""Build-container-only harness that makes /root/reference importable on CPU.

NOT shipped to the GPU box as a dependency of anything: it is used by
tools/gen_golden.py (golden vector generator) and tools/time_reference_cpu.py only.
It installs import stubs for the third-party packages the image lacks
(torchvision, hydra, nltk, boto3, tensorboard, torch._six) and patches
``Tensor.cuda`` / ``Module.cuda`` to identity (the reference hard-codes
``.cuda(device)``).  torchvision's ResNet-50 topology and ``roi_align`` are
provided by small stand-ins written here from torchvision's documented
semantics (the latter delegates to oracle.roi_align_direct); see DESIGN.md
"parity unpinned" for those two.
"""
import importlib.machinery
import os
import sys
import types

import torch
import torch.nn as nn

REF = '/root/reference'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Bottleneck(nn.Module):
    def __init__(self, inplanes, planes, stride, downsample, norm_layer):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = norm_layer(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = norm_layer(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = norm_layer(planes * 4)
        self.relu = nn.ReLU()
        self.downsample = downsample

    def forward(self, x):
        idt = x
        o = self.relu(self.bn1(self.conv1(x)))
        o = self.relu(self.bn2(self.conv2(o)))
        o = self.bn3(self.conv3(o))
        if self.downsample is not None:
            idt = self.downsample(x)
        return self.relu(o + idt)


class _ResNet50(nn.Module):
    """torchvision.models.resnet50 topology / parameter names (v1.5)."""

    def __init__(self, norm_layer):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = norm_layer(64)
        self.relu = nn.ReLU()
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        inpl = 64
        for li, (planes, n, stride) in enumerate(((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)), 1):
            blocks = []
            for b in range(n):
                ds = None
                if b == 0:
                    ds = nn.Sequential(nn.Conv2d(inpl, planes * 4, 1, stride if b == 0 else 1, bias=False),
                                       norm_layer(planes * 4))
                blocks.append(_Bottleneck(inpl, planes, stride if b == 0 else 1, ds, norm_layer))
                inpl = planes * 4
            setattr(self, f'layer{li}', nn.Sequential(*blocks))
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(2048, 1000)


class _IntermediateLayerGetter(nn.ModuleDict):
    def __init__(self, model, return_layers):
        layers = {}
        rl = dict(return_layers)
        for name, module in model.named_children():
            layers[name] = module
            if name in rl:
                del rl[name]
            if not rl:
                break
        super().__init__(layers)
        self.return_layers = dict(return_layers)

    def forward(self, x):
        out = {}
        for name, module in self.items():
            x = module(x)
            if name in self.return_layers:
                out[self.return_layers[name]] = x
        return out


def install():
    """Idempotent. Returns the imported reference `exp.gpv.models.gpv` module."""
    if 'exp.gpv.models.gpv' in sys.modules:
        return sys.modules['exp.gpv.models.gpv']
    from transformers import BertModel, BertConfig  # noqa: F401  (must precede the torchvision stub)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, REF)
    from oracle import gpv_oracle as O

    def roi_align(features, boxes, output_size=7, spatial_scale=1.0, sampling_ratio=-1, aligned=False):
        assert aligned and spatial_scale == 1.0 and sampling_ratio == -1
        return torch.cat([O.roi_align_direct(features[i], b, output_size) for i, b in enumerate(boxes)])

    def resnet50(replace_stride_with_dilation=None, pretrained=False, norm_layer=nn.BatchNorm2d):
        return _ResNet50(norm_layer)

    tv = _mod('torchvision', __version__='0.9.0')
    tv.ops = _mod('torchvision.ops', roi_align=roi_align)
    tv.ops.boxes = _mod('torchvision.ops.boxes',
                        box_area=lambda b: (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]))
    tv.ops.misc = _mod('torchvision.ops.misc', interpolate=torch.nn.functional.interpolate)
    tv.models = _mod('torchvision.models', resnet50=resnet50)
    tv.models._utils = _mod('torchvision.models._utils', IntermediateLayerGetter=_IntermediateLayerGetter)
    tv.transforms = _mod('torchvision.transforms')
    _mod('boto3')
    _mod('botocore')
    _mod('botocore.exceptions', ClientError=Exception)
    if not hasattr(torch, '_six'):
        torch._six = _mod('torch._six', inf=float('inf'))
    hy = _mod('hydra', main=lambda *a, **k: (lambda f: f))
    from oracle.gpv_oracle import simple_word_tokenize
    nl = _mod('nltk')
    nl.tokenize = _mod('nltk.tokenize', word_tokenize=simple_word_tokenize)
    _mod('torch.utils.tensorboard', SummaryWriter=object)
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    import exp.gpv.models.gpv as G
    return G


class AttrDict(dict):
    """attribute + item access, real bools, `.items()` -- what the model code needs of OmegaConf."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError:
            raise AttributeError(k)
        return v

    @staticmethod
    def wrap(d):
        if isinstance(d, dict):
            return AttrDict({k: AttrDict.wrap(v) for k, v in d.items()})
        if isinstance(d, list):
            return [AttrDict.wrap(v) for v in d]
        return d


class FakeBert(nn.Module):
    """Stand-in for exp/gpv/models/bert.py:Bert -- HF BertModel with random init, fed token ids
    directly (the bert-base-uncased tokenizer/weights are not available offline).
    `sentences` is a (input_ids, attention_mask) pair instead of list[str]."""

    def __init__(self, cfg=None, num_layers=12, dropout=0.1):
        super().__init__()
        from transformers import BertModel, BertConfig
        # NOTE (reference quirk): the real Bert() sits inside GPV, so model.train() switches its
        # 0.1 dropouts ON even though it only ever runs under no_grad (gpv.py:142-143).  Goldens
        # are generated with dropout=0.0 to be deterministic.
        self.model = BertModel(BertConfig(num_hidden_layers=num_layers, hidden_dropout_prob=dropout,
                                          attention_probs_dropout_prob=dropout))

    def forward(self, sentences, device=None):
        ids, attn = sentences
        out = self.model(input_ids=ids, attention_mask=attn)
        return out[0], {'input_ids': ids, 'attention_mask': attn}
