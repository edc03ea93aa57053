"""JPEG decoding on the device (csrc/jpeg.hip through gpv1_amd.jpeg.DeviceJpegDecoder) against the Pillow goldens, bit-exact, and into the
device input pipeline."""
import glob
import io
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'jpeg')
FILES = sorted(glob.glob(os.path.join(GOLD, '*.jpg')))


def as_rgb(a):
    return a if a.ndim == 3 else np.repeat(a[..., None], 3, 2)          # coco_generic_dataset.py:55-56


def test_device_decoder_is_bit_exact_on_the_golden_files_as_one_batch():
    from gpv1_amd.jpeg import DeviceJpegDecoder
    exp = np.load(os.path.join(GOLD, 'expected.npz'))
    dec = DeviceJpegDecoder()
    outs = dec([open(f, 'rb').read() for f in FILES])                   # mixed sizes / sampling modes / grey in ONE batch
    torch.cuda.synchronize()
    for f, o in zip(FILES, outs):
        e = as_rgb(exp[os.path.basename(f)[:-4]])
        assert o.dtype == torch.uint8 and tuple(o.shape) == e.shape
        assert np.array_equal(o.cpu().numpy(), e), os.path.basename(f)


def test_device_decoder_against_pillow_at_dataset_size():
    """COCO-sized files (480x640 and a portrait one with partial MCUs), encoded here, decoded by Pillow and by the device path"""
    from PIL import Image
    from gpv1_amd.jpeg import DeviceJpegDecoder
    r = np.random.RandomState(5)
    files, exp = [], []
    for h, w, sub, q in ((480, 640, 2, 90), (640, 427, 2, 75), (333, 500, 1, 85), (375, 500, 0, 95)):
        yy, xx = np.mgrid[0:h, 0:w]
        img = np.stack([128 + 100 * np.sin(xx / 17.0) * np.cos(yy / 23.0), (xx * 255.0 / w), (yy * 255.0 / h)], -1) + r.randn(h, w, 3) * 12
        img = np.clip(img, 0, 255).astype(np.uint8)
        buf = io.BytesIO()
        Image.fromarray(img).save(buf, 'JPEG', quality=q, subsampling=sub)
        files.append(buf.getvalue())
        exp.append(np.asarray(Image.open(io.BytesIO(buf.getvalue()))))
    outs = DeviceJpegDecoder(threads=1)(files)
    torch.cuda.synchronize()
    for o, e in zip(outs, exp):
        assert np.array_equal(o.cpu().numpy(), e)


def test_jpeg_files_to_stem_input_equals_pipeline_on_decoded_arrays():
    """files -> DeviceJpegDecoder -> DeviceImagePipeline == the pipeline fed with the reference decoder's arrays"""
    from gpv1_amd.jpeg import DeviceJpegDecoder
    from gpv1_amd.input_pipeline import DeviceImagePipeline, draw_params
    import random
    exp = np.load(os.path.join(GOLD, 'expected.npz'))
    names = ['c420_big', 'c444_q90', 'gray_q80', 'c422_q75']
    files = [open(os.path.join(GOLD, n + '.jpg'), 'rb').read() for n in names]
    rng = random.Random(1)
    params = [draw_params('CocoClassification', rng, True) for _ in names]
    pipe = DeviceImagePipeline(size=(96, 128), train=True)
    a = pipe(DeviceJpegDecoder()(files), params=params).tensors.float().cpu()
    b = pipe([as_rgb(exp[n]) for n in names], params=params).tensors.float().cpu()
    assert torch.equal(a, b)


def test_device_decoder_random_files_property():
    """40 freshly encoded files of random size (1..200 pixels a side), sampling mode, quality, Huffman optimisation and restart
    interval, colour and grey, decoded as two batches: every one bit-exact against Pillow"""
    from PIL import Image
    from gpv1_amd.jpeg import DeviceJpegDecoder
    r = np.random.RandomState(11)
    files, exp = [], []
    for i in range(40):
        h, w = int(r.randint(1, 201)), int(r.randint(1, 201))
        grey = i % 7 == 3
        base = r.rand(h, w, 1 if grey else 3) * 255
        if i % 3 == 0:                                                   # smooth content: long zero runs, EOB-heavy blocks
            yy, xx = np.mgrid[0:h, 0:w]
            base = np.stack([(xx * 3 + yy * 2 + 40 * c) % 256 for c in range(base.shape[2])], -1).astype(np.float64)
        img = np.clip(base, 0, 255).astype(np.uint8)
        kw = dict(quality=int(r.randint(5, 100)), optimize=bool(i % 2))
        if not grey:
            kw['subsampling'] = int(r.randint(0, 3))
        if i % 4 == 1:
            kw['restart_marker_blocks'] = int(r.randint(1, 9))
        buf = io.BytesIO()
        Image.fromarray(img[..., 0] if grey else img, 'L' if grey else 'RGB').save(buf, 'JPEG', **kw)
        files.append(buf.getvalue())
        exp.append(as_rgb(np.asarray(Image.open(io.BytesIO(buf.getvalue())))))
    dec = DeviceJpegDecoder(threads=4)
    outs = dec(files[:25]) + dec(files[25:])
    torch.cuda.synchronize()
    for i, (o, e) in enumerate(zip(outs, exp)):
        assert tuple(o.shape) == e.shape and np.array_equal(o.cpu().numpy(), e), (i, e.shape)
