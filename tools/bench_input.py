"""Device-side input pipeline at the training batch: 32 COCO-sized JPEG files -> decoded RGB on the GPU -> resized /
augmented / normalised stem input.  Host stage (entropy decoding, thread pool), upload, and the four device launches are timed apart.
usage: python tools/bench_input.py [--batch 32] [--threads 8]          (GPU box; needs Pillow to WRITE the test files)"""
import argparse
import io
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpv1_amd.jpeg import DeviceJpegDecoder                    # noqa: E402
from gpv1_amd.input_pipeline import DeviceImagePipeline        # noqa: E402
import gpv1_amd.hip as hip                                      # noqa: E402


def make_files(n, seed=0):
    from PIL import Image
    r = np.random.RandomState(seed)
    files = []
    for i in range(n):
        h, w = ((480, 640), (427, 640), (640, 480), (500, 375))[i % 4]
        yy, xx = np.mgrid[0:h, 0:w]
        img = np.stack([128 + 100 * np.sin(xx / (11.0 + i)) * np.cos(yy / 23.0), xx * 255.0 / w, yy * 255.0 / h], -1) + r.randn(h, w, 3) * 14
        buf = io.BytesIO()
        Image.fromarray(np.clip(img, 0, 255).astype(np.uint8)).save(buf, 'JPEG', quality=90, subsampling=2)
        files.append(buf.getvalue())
    return files


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--threads', type=int, default=8)
    ap.add_argument('--iters', type=int, default=10)
    a = ap.parse_args()
    hip.lib()
    files = make_files(a.batch)
    dec = DeviceJpegDecoder(threads=a.threads)
    pipe = DeviceImagePipeline(size=(480, 640), train=True)
    tasks = ['CocoClassification'] * a.batch
    for _ in range(2):
        pipe(dec(files), tasks)
    torch.cuda.synchronize()
    # host stage alone
    t0 = time.perf_counter()
    for _ in range(a.iters):
        infos = list(dec.pool.map(hip.jpeg_parse, files)) if dec.pool else [hip.jpeg_parse(f) for f in files]
        bufs = [np.empty(int(i.coef_count), np.int16) for i in infos]
        if dec.pool:
            list(dec.pool.map(lambda k: hip.jpeg_parse(files[k], bufs[k]), range(a.batch)))
        else:
            [hip.jpeg_parse(files[k], bufs[k]) for k in range(a.batch)]
    host_ms = (time.perf_counter() - t0) * 1e3 / a.iters
    # whole path, wall clock and GPU time
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    for _ in range(a.iters):
        out = pipe(dec(files), tasks)
    e1.record()
    torch.cuda.synchronize()
    wall_ms = (time.perf_counter() - t0) * 1e3 / a.iters
    # device launches alone: decode + pipeline on resident inputs
    imgs = dec(files)
    torch.cuda.synchronize()
    g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g0.record()
    for _ in range(a.iters):
        pipe(imgs, tasks)
    g1.record()
    torch.cuda.synchronize()
    mb = sum(len(f) for f in files) / 1e6
    coef_mb = sum(int(hip.jpeg_parse(f).coef_count) for f in files) * 2 / 1e6
    print('batch %d, %.1f MB of JPEG files, %.1f MB of coefficients uploaded' % (a.batch, mb, coef_mb))
    print('host entropy decoding (%d threads): %.2f ms per batch = %.0f images/s' % (a.threads, host_ms, a.batch / host_ms * 1e3))
    print('files -> stem input, wall clock: %.2f ms per batch = %.0f images/s' % (wall_ms, a.batch / wall_ms * 1e3))
    print('resize + augment + normalise launches alone: %.3f ms per batch' % (g0.elapsed_time(g1) / a.iters))
    print('output', tuple(out.tensors.shape), out.tensors.dtype)


if __name__ == '__main__':
    main()
