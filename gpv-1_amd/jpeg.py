"""JPEG files -> decoded uint8 RGB images on the GPU (the first stage of the device-side input pipeline).

Reference: `img = skio.imread(img_path)` + grey -> 3 channels in the loader workers (datasets/coco_generic_dataset.py:48-58,
datasets/coco_datasets.py:151-163, inference_util.py:9-16): Pillow / libjpeg-turbo on 30 CPU processes.  Here the host only walks the
entropy-coded bit stream (gpv_jpeg_parse, C++, one image per pool thread, the GIL released), the quantised coefficients (2 bytes per
sample position, mostly zero -- but a fixed-size upload) go to the GPU through pinned memory and everything per block / per pixel --
dequantisation, the 8x8 inverse DCT, chroma upsampling, YCbCr -> RGB -- runs there for the whole batch in two launches
(gpv_jpeg_decode, csrc/jpeg.hip), bit-exact against Pillow's decoder.

    dec = DeviceJpegDecoder()
    images = dec([open(p, 'rb').read() for p in paths])        # list of [H, W, 3] uint8 CUDA tensors
    samples = DeviceImagePipeline(train=True)(images, tasks)   # resize + augmentation + normalisation -> the stem's input

Scope: baseline sequential Huffman JPEG, 8 bit, grey or YCbCr with 4:4:4 / 4:2:2 / 4:2:0 sampling, restart markers -- what COCO's
files are.  Anything else raises hip.JpegUnsupported (no silent CPU fallback).  EXIF orientation is not applied."""
import ctypes as C
from concurrent.futures import ThreadPoolExecutor

import torch

from . import hip
from .misc import upload_bytes


class DeviceJpegDecoder:
    def __init__(self, device='cuda', threads=8, slots=3, max_pixels=64 << 20):
        """max_pixels: refuse files whose header declares more than this many pixels (default 64 MPix; COCO's largest is 0.4) BEFORE
        anything is sized from it -- a 300-byte file may declare 65535 x 65535 -- and keep every offset of gpv_jpeg_desc inside its
        32-bit fields (3 components x 64 MPix x 1 byte < 2^31)"""
        self.device = torch.device(device)
        self.max_pixels = int(min(max_pixels, 512 << 20))
        self.pool = ThreadPoolExecutor(max_workers=threads) if threads > 1 else None
        # pinned staging for the coefficients: a ring of buffers, each guarded by the event of its last upload (allocating
        # pinned memory per batch costs more than decoding the batch)
        self._stage, self._events, self._turn = [None] * slots, [None] * slots, 0
        self._keep = None

    def _staging(self, n):
        k = self._turn
        self._turn = (k + 1) % len(self._stage)
        if self._events[k] is not None:
            self._events[k].synchronize()
        if self._stage[k] is None or self._stage[k].numel() < n:
            self._stage[k] = torch.empty(max(n, 1 << 22), dtype=torch.int16).pin_memory()
        return k, self._stage[k][:n]

    def __call__(self, files):
        """files: list of bytes objects (whole .jpg files) -> list of [H, W, 3] uint8 tensors on the device (asynchronous: ordered on
        the current stream)"""
        B = len(files)
        if B == 0:
            return []
        run = (lambda f, it: list(self.pool.map(f, it))) if self.pool is not None else (lambda f, it: [f(x) for x in it])
        infos = run(hip.jpeg_parse, files)                                            # header pass: sizes
        starts, total = [], 0
        for i, inf in enumerate(infos):
            if inf.width * inf.height > self.max_pixels or int(inf.coef_count) > 4 * self.max_pixels:
                raise ValueError(f'DeviceJpegDecoder: file {i} declares {inf.width} x {inf.height} pixels, more than max_pixels = {self.max_pixels}')
            starts.append(total)
            total += int(inf.coef_count)
        slot, stage = self._staging(total)                                            # one pinned buffer, one upload
        run(lambda i: hip.jpeg_parse(files[i], stage[starts[i]:starts[i] + int(infos[i].coef_count)]), range(B))
        dev = self.device
        coefs = stage.to(dev, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        self._events[slot] = ev
        pl_sizes = [sum(inf.bh[c] * inf.bw[c] * 64 for c in range(inf.ncomp)) for inf in infos]
        planes = torch.empty(sum(pl_sizes), dtype=torch.uint8, device=dev)
        outs = [torch.empty(inf.height, inf.width, 3, dtype=torch.uint8, device=dev) for inf in infos]
        descs = (hip.JpegDesc * B)()
        pl0 = 0
        for d, inf, st, out, psz in zip(descs, infos, starts, outs, pl_sizes):
            d.coefs = coefs.data_ptr() + 2 * st
            d.planes = planes.data_ptr() + pl0
            d.out = out.data_ptr()
            d.width, d.height, d.ncomp, d.hmax, d.vmax = inf.width, inf.height, inf.ncomp, inf.hmax, inf.vmax
            off = 0
            for c in range(inf.ncomp):
                d.bh[c], d.bw[c] = inf.bh[c], inf.bw[c]
                if int(inf.coef_offset[c]) >= 1 << 31 or off >= 1 << 31:        # (cannot happen under max_pixels; never truncate silently)
                    raise OverflowError('DeviceJpegDecoder: component offset does not fit gpv_jpeg_desc')
                d.coef_off[c] = int(inf.coef_offset[c])
                d.plane_off[c] = off
                off += inf.bh[c] * inf.bw[c] * 64
            C.memmove(d.quant, inf.quant, C.sizeof(inf.quant))
            pl0 += psz
        raw = upload_bytes(bytes(descs), dev)
        hip.jpeg_decode(raw, B, max(p // 64 for p in pl_sizes), max(inf.width * inf.height for inf in infos))
        self._keep = (coefs, planes, raw)                     # operands of the asynchronous launches
        return outs
