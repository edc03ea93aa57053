"""Device-side input pipeline: decoded uint8 images -> the stem's NHWC4 bf16 batch on the GPU.

Reference: datasets/coco_generic_dataset.py:49-62 (``resize(img, (imh, imw), anti_aliasing=True)``), datasets/coco_datasets.py
(:26-38 detection, :137-150 classification: ``ToPILImage -> RandomApply([ColorJitter(0.4, 0.4, 0.4, 0.1)], p=0.8) ->
[RandomHorizontalFlip] -> RandomGrayscale(p=0.2) -> ToTensor -> Normalize``; the other tasks: ToTensor -> Normalize only),
30 CPU worker processes (configs/exp/gpv.yaml:126).

Here the host only decodes (out of scope: JPEG) and draws the random parameters in torchvision's order; everything per pixel runs
in two HIP launches (csrc/image_pipeline.hip) and lands in the layout the fused stem kernel reads, so neither an fp32 NCHW batch
(118 MB per 32 images over PCIe) nor gpv_image_to_nhwc4 exists on this path:

    pipe = DeviceImagePipeline(size=(480, 640), train=True)
    samples = pipe(list_of_uint8_HWC_arrays, tasks)          # NestedTensor: .tensors = [B, H+6, Wp, 4] bf16, .mask [B, H, W] (all False)
    loss = trainer.train_step(samples, queries, targets)       # the backbone recognises the prepared stem input

Uploads go through pinned staging and are asynchronous; call it one batch ahead to hide the PCIe copy under the previous step.
"""
import ctypes as C
import math
import random

import numpy as np
import torch

from . import hip
from .misc import NestedTensor, upload_bytes
from .ops import RT

JITTER = (0.4, 0.4, 0.4, 0.1)            # brightness, contrast, saturation, hue  (coco_datasets.py:30,141)
AUGMENT = {'CocoDetection': ('jitter', 'gray'), 'CocoClassification': ('jitter', 'flip', 'gray')}


def draw_params(task, rng, train=True):
    """the random decisions of the reference's transform for one sample, in torchvision's order of draws:
    RandomApply(p=0.8) -> ColorJitter.get_params (factors, then a random order of the four steps) -> RandomHorizontalFlip(0.5)
    -> RandomGrayscale(0.2).  -> dict(jitter, order, brightness, contrast, saturation, hue, flip, gray)"""
    p = {'jitter': 0, 'order': (0, 1, 2, 3), 'brightness': 1.0, 'contrast': 1.0, 'saturation': 1.0, 'hue': 0.0, 'flip': 0, 'gray': 0}
    aug = AUGMENT.get(task, ()) if train else ()
    if 'jitter' in aug and rng.random() < 0.8:
        b, c, s, h = JITTER
        p.update(jitter=1, brightness=rng.uniform(max(0.0, 1 - b), 1 + b), contrast=rng.uniform(max(0.0, 1 - c), 1 + c),
                 saturation=rng.uniform(max(0.0, 1 - s), 1 + s), hue=rng.uniform(-h, h))
        order = [0, 1, 2, 3]
        rng.shuffle(order)
        p['order'] = tuple(order)
    if 'flip' in aug and rng.random() < 0.5:
        p['flip'] = 1
    if 'gray' in aug and rng.random() < 0.2:
        p['gray'] = 1
    return p


def stem_geometry(H, W):
    """padded NHWC4 extent the stem reads (backbone.ResNetBody.forward_nhwc)"""
    OW = (W + 6 - 7) // 2 + 1
    return H + 6, ((max(W + 6, 2 * (OW - 1) + 8) + 7) // 8) * 8


class DeviceImagePipeline:
    def __init__(self, size=(480, 640), train=True, seed=0, device='cuda'):
        self.size, self.train, self.rng, self.device = tuple(size), train, random.Random(seed), torch.device(device)
        self._scratch = {}

    def __call__(self, images, tasks=None, params=None):
        """images: list of HxWx3 uint8 arrays / tensors (host or device); tasks: list[str] (which augmentation applies);
        params: optional list of draw_params() dicts (tests) -- drawn here otherwise"""
        B = len(images)
        H, W = self.size
        Hp, Wp = stem_geometry(H, W)
        dev = self.device
        if params is None:
            params = [draw_params(tasks[i] if tasks is not None else None, self.rng, self.train) for i in range(B)]
        srcs = []
        for im in images:
            t = torch.as_tensor(im)
            if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
                raise ValueError('DeviceImagePipeline: images must be HxWx3 uint8 (decode / grey->RGB on the host, as the reference does)')
            if max(t.shape[0] / H, t.shape[1] / W) > 9:
                raise ValueError('DeviceImagePipeline: more than 9x down-scaling is not supported')
            if not t.is_cuda:
                t = t.contiguous().pin_memory().to(dev, non_blocking=True)
            srcs.append(t.contiguous())
        descs = (hip.ImageDesc * B)()
        for d, t, p in zip(descs, srcs, params):
            d.src, d.H, d.W = t.data_ptr(), t.shape[0], t.shape[1]
            d.flip, d.gray, d.jitter = int(p['flip']), int(p['gray']), int(p['jitter'])
            d.order[:] = list(p['order'])
            d.brightness, d.contrast, d.saturation, d.hue = p['brightness'], p['contrast'], p['saturation'], p['hue']
        raw = upload_bytes(bytes(descs), dev)
        key = (B, H, W)
        sc = self._scratch.get(key)
        if sc is None:
            sc = self._scratch[key] = (torch.empty(B * H * W * 3, dtype=torch.uint8, device=dev), torch.zeros(B, device=dev))
        out = torch.empty(B, Hp, Wp, 4, device=dev, dtype=RT.dtype)
        hip.image_pipeline(raw, B, sc[0], sc[1], out, H, W, 3, Hp, Wp)
        self._keep = (srcs, raw)                       # the sources must outlive the asynchronous launches
        mask = torch.zeros(B, H, W, dtype=torch.bool, device=dev)
        return NestedTensor(out, mask, True)
