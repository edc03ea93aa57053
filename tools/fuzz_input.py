"""Random-size sweep of the device input path against its oracles: JPEG files of random geometry / coding options (vs Pillow), and
the resize + augmentation pipeline on random source sizes and drawn parameters (vs oracle/image_oracle.py, same tolerance as
tests/test_kernels_gpu.py::test_device_input_pipeline_vs_oracle).   usage: python tools/fuzz_input.py [seed] [n]     (GPU box)"""
import io, os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PIL import Image
from oracle import image_oracle as IO
from gpv1_amd.jpeg import DeviceJpegDecoder
from gpv1_amd.input_pipeline import DeviceImagePipeline, draw_params

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
r = np.random.RandomState(seed)
rng = random.Random(seed)
bad = 0
# ---- JPEG
files, exp = [], []
for i in range(n * 3):
    h, w = int(r.randint(1, 400)), int(r.randint(1, 400))
    grey = r.rand() < 0.15
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([127 + 90 * np.sin(xx / (3.0 + c + i % 5) + yy / 11.0) for c in range(1 if grey else 3)], -1) + r.randn(h, w, 1 if grey else 3) * r.choice([2, 10, 40])
    img = np.clip(img, 0, 255).astype(np.uint8)
    kw = dict(quality=int(r.randint(3, 101)), optimize=bool(r.rand() < 0.5))
    if not grey:
        kw['subsampling'] = int(r.randint(0, 3))
    if r.rand() < 0.3:
        kw['restart_marker_blocks'] = int(r.randint(1, 20))
    buf = io.BytesIO()
    try:
        Image.fromarray(img[..., 0] if grey else img, 'L' if grey else 'RGB').save(buf, 'JPEG', **kw)
    except OSError:                                      # (Pillow's encoder buffer on some option mixes: not our subject)
        continue
    files.append(buf.getvalue())
    e = np.asarray(Image.open(io.BytesIO(buf.getvalue())))
    exp.append(e if e.ndim == 3 else np.repeat(e[..., None], 3, 2))
outs = DeviceJpegDecoder(threads=4)(files)
torch.cuda.synchronize()
for i, (o, e) in enumerate(zip(outs, exp)):
    if tuple(o.shape) != e.shape or not np.array_equal(o.cpu().numpy(), e):
        bad += 1
        print('JPEG MISMATCH', i, e.shape)
print('jpeg: %d files, %d mismatches' % (len(files), bad))
# ---- resize + augmentation
H, W = 96, 128
pipe = DeviceImagePipeline(size=(H, W), train=True)
worst = 0.0
for i in range(n):
    ih, iw = int(r.randint(4, 9 * H)), int(r.randint(4, 9 * W))
    if i % 5 == 0:
        ih, iw = int(r.randint(4, 40)), int(r.randint(4, 40))            # strong up-scaling
    yy, xx = np.mgrid[0:ih, 0:iw]
    img = np.clip(np.stack([127 + 100 * np.sin(yy / 9.0 + c) * np.cos(xx / 13.0 - c) for c in range(3)], -1) + r.randn(ih, iw, 3) * 12, 0, 255).astype(np.uint8)
    p = draw_params(rng.choice(['CocoClassification', 'CocoDetection', 'CocoVqa']), rng, True)
    nt = pipe([torch.from_numpy(img)], params=[p])
    got = nt.tensors.float().cpu().numpy()[0, 3:3 + H, 3:3 + W, :3]
    ref = IO.pipeline(img, (H, W), p).transpose(1, 2, 0)
    steps = np.abs(got - ref) * (255.0 * IO.STD)
    ok = (steps <= 3.0).mean() >= 0.995 and steps.max() <= 5.0
    worst = max(worst, float(steps.max()))
    if not ok:
        bad += 1
        print('PIPELINE MISMATCH', (ih, iw), p, float(steps.max()), float((steps <= 3).mean()))
print('pipeline: %d images, worst difference %.2f uint8 steps; total failures %d' % (n, worst, bad))
