"""ResNet-50 (v1.5) backbone with frozen BatchNorm on the HIP implicit-GEMM convolution kernel, NHWC.

Reference: exp/gpv/models/backbone.py (FrozenBatchNorm2d :19-54, BackboneBase :57-79, Backbone :82-97,
Joiner :100-113) + torchvision 0.7 resnet50 topology (keys conv1, bn1, layerN.M.{conv1..3,bn1..3,
downsample.0/1}).  Differences by design (MI355X-first):
  * activations are NHWC bf16 (fp32 in precise mode); FrozenBN is folded: scale into the compute
    copy of the weights, shift into the conv epilogue together with ReLU and the residual add;
  * the 7x7/2 stem reads a zero-padded NHWC4 image: one (row r) tap = 8 pixels x 4 channels = a
    contiguous 64-byte run, so the stem is the same implicit-GEMM kernel with KH=7, KW=1, Cin=32;
  * the whole body is ONE autograd node: forward keeps the block activations, backward runs
    dgrad (fused with the ReLU mask and the identity-branch gradient) and wgrad (fp32 atomics into
    param.grad, BN scale applied per output channel) for the trainable layers only
    (conv1/layer1 are always frozen: backbone.py:61-63).
"""
import math
import os

import torch
import torch.nn as nn
from torch.autograd import Function

from . import hip
from .ops import RT, ensure_grad
from .misc import NestedTensor
from .position_encoding import build_position_encoding

RELU = hip.ACT_RELU
FUSED_STEM = os.environ.get('GPV_FUSED_STEM', '1') != '0'
FUSED_TAIL = os.environ.get('GPV_FUSED_TAIL', '1') != '0'
FUSED_CHAIN = os.environ.get('GPV_FUSED_CHAIN', '1') != '0'      # a layer1 block's tail + the next block's conv1 in one launch (gpv_conv1x1_chain)
WGRAD_STREAM = os.environ.get('GPV_WGRAD_STREAM', '1') != '0'
WGRAD_GROUP = os.environ.get('GPV_WGRAD_GROUP', '1') != '0'      # all conv weight gradients of a backward pass as one grouped call (bf16)
# ... issued per STAGE (layer4, then layer3, then layer2) right behind that stage's backward-data chain, while the stage's dy / x
# are still in the 256 MB MALL -- and so that a data-parallel trainer can hand layer4's gradients (60 of the backbone's 94 MB) to
# the all-reduce while layer3 / layer2 still compute (train.FlatTrainer stage milestones).  0: one call at the end of the pass
# (round 3); 'side': the stage groups on a parallel branch
WGRAD_STAGES = os.environ.get('GPV_WGRAD_STAGES', '1')
_WSTREAMS = {}


def _wgrad_stream(dev):
    st = _WSTREAMS.get(dev)
    if st is None:
        st = _WSTREAMS[dev] = torch.cuda.Stream(device=dev)
    return st
PROF = None     # bench.py sets this to a list: (tag, start_event, end_event) per backbone forward / backward


def _prof(tag):
    if PROF is None:
        return None
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    PROF.append((tag, a, b))
    return b


class FrozenBatchNorm2d(nn.Module):
    """backbone.py:19-54: fixed statistics + affine; eps 1e-5 inside the rsqrt."""

    def __init__(self, n):
        super().__init__()
        self.register_buffer('weight', torch.ones(n))
        self.register_buffer('bias', torch.zeros(n))
        self.register_buffer('running_mean', torch.zeros(n))
        self.register_buffer('running_var', torch.ones(n))

    def _load_from_state_dict(self, state_dict, prefix, *args):
        state_dict.pop(prefix + 'num_batches_tracked', None)
        super()._load_from_state_dict(state_dict, prefix, *args)

    def scale_shift(self):
        scale = self.weight * (self.running_var + 1e-5).rsqrt()
        return scale.float().contiguous(), (self.bias - self.running_mean * scale).float().contiguous()


class ConvW(nn.Module):
    """holder of a conv weight [Cout,Cin,k,k] stored channels_last (= [Cout][kh][kw][Cin] in memory)."""

    def __init__(self, cin, cout, k, stride, pad):
        super().__init__()
        w = torch.empty(cout, cin, k, k)
        nn.init.kaiming_normal_(w, mode='fan_out', nonlinearity='relu')
        self.weight = nn.Parameter(w.contiguous(memory_format=torch.channels_last))
        self.cin, self.cout, self.k, self.stride, self.pad = cin, cout, k, stride, pad

    def phys(self, t=None):
        """[Cout, k*k, Cin] view of the parameter (or its grad) in memory order."""
        t = self.weight if t is None else t
        v = t.detach().permute(0, 2, 3, 1)
        if not v.is_contiguous():
            raise RuntimeError('conv weight / grad must be channels_last (call GPV.ensure_layout())')
        return v.reshape(self.cout, self.k * self.k, self.cin)


def _bn_fold(bn):
    key = ('bn', id(bn))
    hit = RT.cache.get(key)
    if hit is not None and hit[0] == RT.static_epoch:
        return hit[1], hit[2]
    scale, shift = bn.scale_shift()
    RT.cache[key] = (RT.static_epoch, scale, shift)
    return scale, shift


def _conv_copies(conv, bn, need_wd):
    key = ('conv', id(conv), RT.dtype)
    hit = RT.cache.get(key)
    ep = RT.epoch_of(conv.weight)
    if hit is not None and hit[0] == ep and (hit[2] is not None or not need_wd):
        return hit[1:]
    scale, shift = _bn_fold(bn)
    T = conv.k * conv.k
    wf = torch.empty(conv.cout, T, conv.cin, device=scale.device, dtype=RT.dtype)
    wd = torch.empty(conv.cin, T, conv.cout, device=scale.device, dtype=RT.dtype) if need_wd else None
    hip.prep_conv_weight(conv.phys(), scale, wf, wd, conv.cout, T, conv.cin)
    RT.cache[key] = (ep, wf, wd, scale, shift)
    return wf, wd, scale, shift


def _out(n, k, s, p):
    return (n + 2 * p - k) // s + 1


class Bottleneck(nn.Module):
    def __init__(self, inplanes, planes, stride, downsample):
        super().__init__()
        self.conv1 = ConvW(inplanes, planes, 1, 1, 0)
        self.bn1 = FrozenBatchNorm2d(planes)
        self.conv2 = ConvW(planes, planes, 3, stride, 1)
        self.bn2 = FrozenBatchNorm2d(planes)
        self.conv3 = ConvW(planes, planes * 4, 1, 1, 0)
        self.bn3 = FrozenBatchNorm2d(planes * 4)
        self.downsample = None
        if downsample:
            self.downsample = nn.Sequential(ConvW(inplanes, planes * 4, 1, stride, 0), FrozenBatchNorm2d(planes * 4))

    def convs(self):
        c = [(self.conv1, self.bn1), (self.conv2, self.bn2), (self.conv3, self.bn3)]
        if self.downsample is not None:
            c.append((self.downsample[0], self.downsample[1]))
        return c

    def trainable(self):
        return any(c.weight.requires_grad for c, _ in self.convs())


# Round 6: ReLU masks as one bit per element (include/gpv_hip.h gpv_conv_args.y_mask_bits / relu_mask_bits).  A block's output serves the
# backward pass twice: as the operand of the next block's conv1 weight gradient, and as "was it > 0" in that conv1's backward-data
# epilogue -- read there as bf16 it is a third of the launch's bytes (157 MB per layer2 block at B = 32).  Where the streaming 1x1 kernel
# writes the block output (conv3 + identity + ReLU of layer2 / layer3) it also writes the bits; the tensor carries them (`_gpv_bits`)
# and _conv_dgrad hands them over instead of the bf16 mask where the kernel reads them.  Bit-identical results.  GPV_MASK_BITS=0: off.
MASK_BITS = os.environ.get('GPV_MASK_BITS', '1') != '0'
_BITS_OK = {}


def _bits_ok(key, args, kw):
    """cached gpv_conv2d_mask_bits_ok per call signature (shapes + which operands exist; the allocator's 512-byte alignment makes the
    pointer checks shape-independent)"""
    hit = _BITS_OK.get(key)
    if hit is None:
        hit = _BITS_OK[key] = hip.conv2d_mask_bits_ok(*args, **kw)
    return hit


def _conv_fwd(x, conv, bn, need_wd, act, res=None, bits=False):
    """bits: also produce the one-bit ReLU mask of the output (kept on the tensor as `_gpv_bits`) when the launch can"""
    B, H, Wd, Cin = x.shape
    OH, OW = _out(H, conv.k, conv.stride, conv.pad), _out(Wd, conv.k, conv.stride, conv.pad)
    wf, _, _, shift = _conv_copies(conv, bn, need_wd)
    y = torch.empty(B, OH, OW, conv.cout, device=x.device, dtype=RT.dtype)
    args = (0, x, wf, y, B, H, Wd, Cin, Cin, OH, OW, conv.cout, conv.k, conv.k, conv.stride, conv.stride, conv.pad, conv.pad)
    if bits and MASK_BITS and x.is_cuda and RT.dtype == torch.bfloat16 and conv.k == 1 and conv.stride == 1 and res is not None and act == RELU \
            and conv.cout % 256 == 0:
        mb = torch.empty(B * OH * OW, conv.cout // 32, device=x.device, dtype=torch.int32)
        kw = dict(bias=shift, res=res, act=act, y_mask_bits=mb)
        if _bits_ok(('f', B, H, Wd, Cin, conv.cout, hip.get_option_cached(hip.OPT_C1S)), args, kw):
            hip.conv2d(*args, **kw)
            y._gpv_bits = mb
            return y
        del mb
    hip.conv2d(*args, bias=shift, res=res, act=act)
    return y


def _block_tail_fused(x, a2, blk, tr, need_wd):
    """ReLU(conv3(a2) + downsample(x)) of a stage's first block as ONE launch (gpv_conv1x1_dual: layer1 / layer2); None = not a
    shape the kernel takes"""
    c3, cd = blk.conv3, blk.downsample[0]
    B, OH, OW, K1 = a2.shape
    _, IH, IW, K2 = x.shape
    if (K1, K2, c3.cout) not in ((64, 64, 256), (128, 256, 512)):
        return None
    w3, _, _, s3 = _conv_copies(c3, blk.bn3, tr)
    wd, _, _, sd = _conv_copies(cd, blk.downsample[1], need_wd)
    key = ('tail_shift', id(blk))
    hit = RT.cache.get(key)
    if hit is None or hit[0] != RT.static_epoch:
        hit = RT.cache[key] = (RT.static_epoch, (s3 + sd).contiguous())          # FrozenBN shifts: constants of the static epoch
    y = torch.empty(B, OH, OW, c3.cout, device=x.device, dtype=RT.dtype)
    if tr and MASK_BITS and (K1, K2, c3.cout) == (128, 256, 512) and x.is_cuda:
        # layer2's first block output is the ReLU mask of the second block's conv1 backward-data: its bits ride in this launch
        mb = torch.empty(B * OH * OW, c3.cout // 32, device=x.device, dtype=torch.int32)
        if hip.conv1x1_dual(a2, w3, x, wd, hit[1], y, B, OH, OW, K1, IH, IW, K2, cd.stride, c3.cout, RELU, y_mask_bits=mb):
            y._gpv_bits = mb
            return y
        del mb
    if not hip.conv1x1_dual(a2, w3, x, wd, hit[1], y, B, OH, OW, K1, IH, IW, K2, cd.stride, c3.cout, RELU):
        return None
    return y


def _block_tail_chain(x, a2, blk, nxt, tr_next):
    """(yb, a1_next): this block's tail -- conv3 (+ downsample | + identity) + ReLU -- and the NEXT block's conv1 + ReLU as ONE launch
    (gpv_conv1x1_chain: the 256-channel map is written once and not read back); None = not a shape the kernel takes.  Frozen
    layer1 blocks only (64 planes): nothing between the two convolutions is needed by a backward pass."""
    c3, c1n = blk.conv3, nxt.conv1
    if c3.cin != 64 or c3.cout != 256 or c1n.cin != 256 or c1n.cout not in (64, 128) or c1n.stride != 1 or blk.trainable():
        return None
    B, OH, OW, _ = a2.shape
    w3, _, _, s3 = _conv_copies(c3, blk.bn3, False)
    wn, _, _, sn = _conv_copies(c1n, nxt.bn1, tr_next)
    yb = torch.empty(B, OH, OW, 256, device=x.device, dtype=RT.dtype)
    an = torch.empty(B, OH, OW, c1n.cout, device=x.device, dtype=RT.dtype)
    if blk.downsample is not None:
        cd = blk.downsample[0]
        if cd.cin != 64 or cd.stride != 1:
            return None
        wd, _, _, sd = _conv_copies(cd, blk.downsample[1], False)
        key = ('tail_shift', id(blk))
        hit = RT.cache.get(key)
        if hit is None or hit[0] != RT.static_epoch:
            hit = RT.cache[key] = (RT.static_epoch, (s3 + sd).contiguous())
        ok = hip.conv1x1_chain(a2, w3, x, wd, 1, None, hit[1], yb, wn, sn, an, B, OH, OW)
    else:
        if tr_next and MASK_BITS and c1n.cout == 128 and x.is_cuda:
            # layer2.0's conv1 output is the ReLU mask of that block's stride-2 3x3 backward-data (157 MB as bf16 at B = 32): its bits ride here
            mb = torch.empty(B * OH * OW, c1n.cout // 32, device=x.device, dtype=torch.int32)
            if hip.conv1x1_chain(a2, w3, None, None, 1, x, s3, yb, wn, sn, an, B, OH, OW, z_mask_bits=mb):
                an._gpv_bits = mb
                return yb, an
            del mb
        ok = hip.conv1x1_chain(a2, w3, None, None, 1, x, s3, yb, wn, sn, an, B, OH, OW)
    return (yb, an) if ok else None


def _conv_dgrad(dy, conv, bn, xshape, res=None, relu_mask=None):
    """dx[B,H,W,Cin] = convT(dy) (+res) * (relu_mask > 0)"""
    B, H, Wd, Cin = xshape
    _, OH, OW, Cout = dy.shape
    _, wd, _, _ = _conv_copies(conv, bn, True)
    dx = torch.empty(B, H, Wd, Cin, device=dy.device, dtype=RT.dtype)
    args = (1, dy, wd, dx, B, OH, OW, Cout, Cout, H, Wd, Cin, conv.k, conv.k, conv.stride, conv.stride, conv.pad, conv.pad)
    mb = getattr(relu_mask, '_gpv_bits', None) if relu_mask is not None else None
    if mb is not None and MASK_BITS:
        kw = dict(res=res, relu_mask_bits=mb)
        if _bits_ok(('d', B, H, Wd, Cin, Cout, conv.k, conv.stride, res is not None, hip.get_option_cached(hip.OPT_C1S), hip.get_option_cached(hip.OPT_C3S)), args, kw):
            hip.conv2d(*args, **kw)          # the mask as one bit per element, written by the forward launch that produced relu_mask
            return dx
    hip.conv2d(*args, res=res, relu_mask=relu_mask)
    return dx


def _conv_wgrad(x, dy, conv, bn, group=None):
    if not conv.weight.requires_grad:
        return
    B, H, Wd, Cin = x.shape
    _, OH, OW, Cout = dy.shape
    _, _, scale, _ = _conv_copies(conv, bn, False)
    g = conv.phys(ensure_grad(conv.weight))
    if group is not None:
        group.append((x, dy, g, scale, B, H, Wd, Cin, Cin, OH, OW, Cout, conv.k, conv.k, conv.stride, conv.stride, conv.pad, conv.pad))
        return
    hip.conv2d(2, x, dy, g, B, H, Wd, Cin, Cin, OH, OW, Cout, conv.k, conv.k, conv.stride, conv.stride, conv.pad, conv.pad,
               rowscale=scale)


class ResNetBody(nn.Module):
    """torchvision.models.resnet50 without avgpool/fc (IntermediateLayerGetter(return layer4))."""

    def __init__(self):
        super().__init__()
        self.conv1 = ConvW(3, 64, 7, 2, 3)
        self.bn1 = FrozenBatchNorm2d(64)
        inpl = 64
        for li, (planes, n, stride) in enumerate(((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)), 1):
            blocks = []
            for b in range(n):
                blocks.append(Bottleneck(inpl, planes, stride if b == 0 else 1, b == 0))
                inpl = planes * 4
            setattr(self, f'layer{li}', nn.Sequential(*blocks))

    def blocks(self):
        return [b for li in range(1, 5) for b in getattr(self, f'layer{li}')]

    def stage_of(self, blk):
        m = getattr(self, '_stage_map', None)
        if m is None:
            m = self._stage_map = {id(b): li for li in range(1, 5) for b in getattr(self, f'layer{li}')}
        return m[id(blk)]

    def _stem_weight(self):
        key = ('stem', id(self), RT.dtype)
        hit = RT.cache.get(key)
        if hit is not None and hit[0] == RT.static_epoch:
            return hit[1], hit[2]
        scale, shift = _bn_fold(self.bn1)
        w = self.conv1.weight.detach().float() * scale.view(-1, 1, 1, 1)          # [64,3,7,7]
        ws = torch.zeros(64, 7, 8, 4, device=w.device, dtype=torch.float32)        # [co][r][8 px][4 ch]
        ws[:, :, :7, :3] = w.permute(0, 2, 3, 1)
        ws = ws.reshape(64, 7, 32).to(RT.dtype).contiguous()
        RT.cache[key] = (RT.static_epoch, ws, shift)
        return ws, shift

    def forward_nhwc(self, images, keep, hw=None):
        """images: NCHW fp32 (ImageNet-normalised) -- or, with hw = (H, W), the stem's zero-padded NHWC4 input [B, H+6, Wp, 4] as
        input_pipeline.DeviceImagePipeline prepares it.  Returns c5 [B,h,w,2048]; `keep` collects
        (block, x, a1, a2, y) tuples of the trainable blocks for backward."""
        if hw is None:
            B, _, H, Wd = images.shape
        else:
            B, (H, Wd) = images.shape[0], hw
        OH, OW = _out(H, 7, 2, 3), _out(Wd, 7, 2, 3)
        Hp = H + 6
        Wp = ((max(Wd + 6, 2 * (OW - 1) + 8) + 7) // 8) * 8
        if RT.split is not None and keep is not None:
            RT.split.prep_fork(self)
        if hw is None:
            xin = torch.empty(B, Hp, Wp, 4, device=images.device, dtype=RT.dtype)
            hip.image_to_nhwc4(images.contiguous(), xin, B, H, Wd, 3, Hp, Wp)
        else:
            if tuple(images.shape) != (B, Hp, Wp, 4) or images.dtype != RT.dtype:
                raise ValueError(f'prepared stem input: expected {(B, Hp, Wp, 4)} {RT.dtype}, got {tuple(images.shape)} {images.dtype}')
            xin = images.contiguous()
        ws, shift = self._stem_weight()
        PH, PW = _out(OH, 3, 2, 1), _out(OW, 3, 2, 1)
        x = torch.empty(B, PH, PW, 64, device=images.device, dtype=RT.dtype)
        if RT.dtype == torch.bfloat16 and FUSED_STEM:
            # conv1 + bn1 + relu + maxpool in one launch: the 240x320x64 conv map (314 MB at B = 32) is never written
            hip.stem_pool(xin, ws, shift, x, B, Hp, Wp, OH, OW, PH, PW)
        else:                                      # fp32 "precise" mode: the generic conv kernel + the pooling kernel
            y = torch.empty(B, OH, OW, 64, device=images.device, dtype=RT.dtype)
            hip.conv2d(0, xin, ws, y, B, Hp, Wp, 4, 32, OH, OW, 64, 7, 1, 2, 2, 0, 0, bias=shift, act=RELU)
            hip.maxpool3x3s2(y, x, B, OH, OW, 64, PH, PW)
            del y
        del xin
        if RT.split is not None and keep is not None:
            RT.split.prep_join()                   # the weight copies were prepared on a branch beside the stem (prep_weights)
        seen_trainable = False
        blocks = self.blocks()
        a1_next = None                              # the next block's conv1 output when the previous tail already computed it
        for bi, blk in enumerate(blocks):
            tr = blk.trainable() and keep is not None
            need_wd = tr and seen_trainable         # dgrad into this block's input only if something upstream trains
            a1 = a1_next if a1_next is not None else _conv_fwd(x, blk.conv1, blk.bn1, tr, RELU)
            a1_next = None
            a2 = _conv_fwd(a1, blk.conv2, blk.bn2, tr, RELU)
            yb = None
            if FUSED_CHAIN and RT.dtype == torch.bfloat16 and bi + 1 < len(blocks):
                nxt = blocks[bi + 1]
                ch = _block_tail_chain(x, a2, blk, nxt, nxt.trainable() and keep is not None)
                if ch is not None:
                    yb, a1_next = ch
            if yb is None:
                yb = _block_tail_fused(x, a2, blk, tr, need_wd) if (blk.downsample is not None and FUSED_TAIL and RT.dtype == torch.bfloat16) else None
            if yb is None:
                idt = x if blk.downsample is None else _conv_fwd(x, blk.downsample[0], blk.downsample[1], need_wd, hip.ACT_NONE)
                yb = _conv_fwd(a2, blk.conv3, blk.bn3, tr, RELU, res=idt, bits=keep is not None)
            if tr:
                keep.append((blk, x, a1, a2, yb, seen_trainable))
                seen_trainable = True
            x = yb
        return x

    def prep_weights(self):
        """the bf16 copies (BN scale folded in; forward and backward-data layouts) of every trainable block's weights: 42 small
        launches a training forward would otherwise issue one by one in front of its convolutions"""
        seen_trainable = False
        for blk in self.blocks():
            if not blk.trainable():
                continue
            _conv_copies(blk.conv1, blk.bn1, True)
            _conv_copies(blk.conv2, blk.bn2, True)
            if blk.downsample is not None:
                _conv_copies(blk.downsample[0], blk.downsample[1], seen_trainable)
            _conv_copies(blk.conv3, blk.bn3, True)
            seen_trainable = True

    def backward_nhwc(self, keep, dc5, stage_done=None):
        """dc5: gradient w.r.t. the (post-ReLU) c5 output, [B,h,w,2048] compute dtype.

        The backward-data convolutions form a serial chain; the 42 weight gradients hang off it and nobody needs them before the
        optimizer (or the gradient exchange).  bf16: they are collected and handed to gpv_conv_wgrad_group (equal work units over
        all problems, layer4 without a split reduction) -- one call per STAGE, issued right behind the stage's last backward-data
        launch (GPV_WGRAD_STAGES=0: one call at the end of the pass; 'side': on a parallel branch).  `stage_done(li)` is called
        when every gradient of layer li has been issued (train.py: bucket milestones / graph cuts).
        GPV_WGRAD_GROUP=0 / fp32 "precise" mode: one launch each on a SIDE stream (a parallel branch when the pass is captured
        into a hipGraph), ordered behind the backward-data launch that produced its dy, one join at the end
        (GPV_WGRAD_STREAM=0 puts them back in line)."""
        if not keep:
            return
        dev = dc5.device
        main = torch.cuda.current_stream(dev) if dev.type == 'cuda' else None
        group = [] if (WGRAD_GROUP and dc5.dtype == torch.bfloat16) else None
        side = _wgrad_stream(dev) if (main is not None and WGRAD_STREAM and group is None) else None
        gside = _wgrad_stream(dev) if (main is not None and group is not None and WGRAD_STAGES == 'side') else None
        per_stage = WGRAD_STAGES != '0' or stage_done is not None
        held = []                                        # operands of the side-stream launches stay referenced until the join

        def wgrad(xa, dy, conv, bn):
            if not conv.weight.requires_grad:
                return
            if group is not None:
                _conv_wgrad(xa, dy, conv, bn, group)
                return
            if side is None:
                _conv_wgrad(xa, dy, conv, bn)
                return
            side.wait_stream(main)
            with torch.cuda.stream(side):
                _conv_wgrad(xa, dy, conv, bn)
            held.append((xa, dy))

        def flush(li):
            if group:
                if gside is not None:
                    gside.wait_stream(main)
                    with torch.cuda.stream(gside):
                        hip.conv_wgrad_group(group)
                    held.extend(group)
                else:
                    hip.conv_wgrad_group(group)
                del group[:]
            if stage_done is not None:
                # a stage milestone hands the stage's gradient buckets to the exchange (train.FlatTrainer._on_milestone), ordered
                # behind `main` only: every weight-gradient launch of the stage must have been joined first -- the grouped call on
                # its branch AND the one-launch-each side stream of fp32 "precise" mode / GPV_WGRAD_GROUP=0 (that one
                # was joined only at the end of the pass, so a bucket could leave while its gradients were still accumulating)
                if gside is not None:
                    main.wait_stream(gside)
                if side is not None:
                    main.wait_stream(side)
                stage_done(li)
        y_last = keep[-1][4]
        gz = torch.empty_like(y_last)
        hip.act_bwd(dc5.contiguous(), y_last, gz, gz.numel(), RELU, 1.0)          # through the final ReLU
        order = list(reversed(keep))
        for i, (blk, x, a1, a2, yb, need_dx) in enumerate(order):
            # gz = gradient w.r.t. (conv3 + shift + identity), i.e. already masked by (yb > 0)
            li = self.stage_of(blk)
            wgrad(a2, gz, blk.conv3, blk.bn3)
            g2 = _conv_dgrad(gz, blk.conv3, blk.bn3, a2.shape, relu_mask=a2)
            wgrad(a1, g2, blk.conv2, blk.bn2)
            g1 = _conv_dgrad(g2, blk.conv2, blk.bn2, a1.shape, relu_mask=a1)
            wgrad(x, g1, blk.conv1, blk.bn1)
            if blk.downsample is not None:
                wgrad(x, gz, blk.downsample[0], blk.downsample[1])
            if not need_dx:
                flush(li)
                break
            if blk.downsample is not None:
                side_g = _conv_dgrad(gz, blk.downsample[0], blk.downsample[1], x.shape)
            else:
                side_g = gz
            # input gradient, masked by the previous block's ReLU (x is that block's output)
            gz = _conv_dgrad(g1, blk.conv1, blk.bn1, x.shape, res=side_g, relu_mask=x)
            del g1, g2, side_g
            last = i + 1 == len(order)
            if last or (per_stage and self.stage_of(order[i + 1][0]) != li):
                flush(li)
        if side is not None:
            main.wait_stream(side)
        if gside is not None:
            main.wait_stream(gside)
        del held


class ResNetFn(Function):
    @staticmethod
    def forward(ctx, images, body, dummy, need_bwd, hw=None):
        keep = [] if need_bwd else None          # (grad mode is always off inside Function.forward)
        ev = _prof('conv_fwd')
        c5 = body.forward_nhwc(images, keep, hw)
        if ev is not None:
            ev.record()
        ctx.body, ctx.keep = body, keep
        return c5

    @staticmethod
    def backward(ctx, dc5):
        if RT.backward_milestone is not None:          # everything downstream of the backbone has finished its backward
            RT.backward_milestone('backbone')
        ev = _prof('conv_bwd')
        ms = RT.backward_milestone
        ctx.body.backward_nhwc(ctx.keep, dc5.to(RT.dtype), (lambda li: ms('layer%d' % li)) if ms is not None else None)
        if ev is not None:
            ev.record()
        ctx.keep = None
        return None, None, None, None, None


class BackboneBase(nn.Module):
    def __init__(self, body, train_backbone, num_channels):
        super().__init__()
        for name, p in body.named_parameters():
            if not train_backbone or ('layer2' not in name and 'layer3' not in name and 'layer4' not in name):
                p.requires_grad_(False)
        self.body = body
        self.num_channels = num_channels
        self._dummy = None

    def forward(self, tensor_list: NestedTensor):
        """-> {'0': NestedTensor(c5 as [B,h,w,C] NHWC, mask [B,h,w])}   (backbone.py:71-79)"""
        x, m = tensor_list.tensors, tensor_list.mask
        assert m is not None
        if self._dummy is None or self._dummy.device != x.device:
            self._dummy = torch.zeros(1, device=x.device, requires_grad=True)       # makes autograd call backward
        need_bwd = torch.is_grad_enabled() and any(b.trainable() for b in self.body.blocks())
        # a prepared stem input (input_pipeline.DeviceImagePipeline): [B, H+6, Wp, 4] in the compute dtype, H x W = the mask's
        hw = tuple(m.shape[-2:]) if (x.dim() == 4 and x.shape[1] != 3 and x.shape[-1] == 4) else None
        if hw is None:
            x = x.float()
        if RT.split is not None and torch.is_grad_enabled():
            # train.GraphedBody: the backbone is its own pair of hipGraphs (forward here, backward called by the trainer
            # between the two halves of the gradient exchange); the rest of the model sees c5 as an autograd leaf.  With a
            # frozen backbone (phase-1 training.freeze, lr_backbone = 0) the forward graph still ends here, without a backward.
            c5 = RT.split.backbone_forward(self.body, x, train=need_bwd, hw=hw)
        else:
            c5 = ResNetFn.apply(x, self.body, self._dummy, need_bwd, hw)
        h, w = c5.shape[1:3]
        H, Wd = m.shape[-2:]
        hint = getattr(tensor_list, 'all_valid', None)
        if hint is True:
            # the host knows there is no padding: the down-sampled mask is a constant (8 index / arange launches less per step)
            key = (m.shape[0], h, w, m.device)
            mask = self._zero_masks.get(key) if hasattr(self, '_zero_masks') else None
            if mask is None:
                if not hasattr(self, '_zero_masks'):
                    self._zero_masks = {}
                if not (m.is_cuda and torch.cuda.is_current_stream_capturing()):
                    mask = self._zero_masks[key] = torch.zeros(m.shape[0], h, w, dtype=torch.bool, device=m.device)
        else:
            mask = None
        if mask is None:
            iy = torch.div(torch.arange(h, device=m.device) * H, h, rounding_mode='floor')   # F.interpolate nearest
            ix = torch.div(torch.arange(w, device=m.device) * Wd, w, rounding_mode='floor')
            mask = m[:, iy][:, :, ix]
        return {'0': NestedTensor(c5, mask, hint)}


class Backbone(BackboneBase):
    def __init__(self, name, train_backbone, return_interm_layers, dilation, frozenbatchnorm=True):
        if name != 'resnet50' or return_interm_layers or dilation or not frozenbatchnorm:
            raise NotImplementedError('gpv1_amd builds the configuration GPV-1 ships: resnet50, C5 only, no dilation, '
                                      'FrozenBatchNorm (configs/exp/gpv.yaml:41-52)')
        super().__init__(ResNetBody(), train_backbone, 2048)


class Joiner(nn.Sequential):
    """backbone.py:100-113"""

    def __init__(self, backbone, position_embedding):
        super().__init__(backbone, position_embedding)

    def forward(self, tensor_list):
        xs = self[0](tensor_list)
        out, pos = [], []
        for _, x in xs.items():
            out.append(x)
            pos.append(self[1](x))
        return out, pos


def build_backbone(args):
    position_embedding = build_position_encoding(args)
    backbone = Backbone(args.backbone, args.lr_backbone > 0, args.masks, args.dilation, args.frozenbatchnorm)
    model = Joiner(backbone, position_embedding)
    model.num_channels = backbone.num_channels
    return model
