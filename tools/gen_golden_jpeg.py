"""Golden fixtures for the JPEG decoder: small JPEG files written AND decoded by the Pillow / libjpeg-turbo of this
image -- the decoder the reference's `skimage.io.imread` ends in (datasets/coco_generic_dataset.py:54).

    python tools/gen_golden_jpeg.py        ->  tests/golden/jpeg/*.jpg, tests/golden/jpeg/expected.npz

Covers 4:4:4 / 4:2:2 / 4:2:0, odd sizes (partial MCUs, odd chroma widths), qualities 30..97, optimised Huffman tables, restart
intervals, grayscale, a one-MCU image and a flat image (DC only)."""
import io
import os

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'tests', 'golden', 'jpeg')


def picture(h, w, seed):
    r = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.stack([127 + 120 * np.sin(xx / (3.0 + seed) + yy / 7.0), 255 * xx / max(w - 1, 1), 255 * yy / max(h - 1, 1)], -1)
    img += r.randn(h, w, 3) * (4 + 6 * (seed % 3))
    cy, cx = h // 2, w // 3
    img[max(cy - 5, 0):cy + 6, max(cx - 7, 0):cx + 8] = (250, 20, 30)                     # a saturated block: sharp chroma edges
    img[:: max(h // 6, 1), :, :] = 255 - img[:: max(h // 6, 1), :, :]
    return np.clip(img, 0, 255).astype(np.uint8)


CASES = [  # name, H, W, mode, save kwargs
    ('c444_q90', 48, 64, 'RGB', dict(quality=90, subsampling=0)),
    ('c422_q75', 48, 64, 'RGB', dict(quality=75, subsampling=1)),
    ('c420_q75', 48, 64, 'RGB', dict(quality=75, subsampling=2)),
    ('c420_odd_q85', 37, 53, 'RGB', dict(quality=85, subsampling=2)),
    ('c422_odd_q60', 29, 43, 'RGB', dict(quality=60, subsampling=1)),
    ('c444_odd_q97', 21, 19, 'RGB', dict(quality=97, subsampling=0)),
    ('c420_opt_q30', 64, 80, 'RGB', dict(quality=30, subsampling=2, optimize=True)),
    ('c420_rst_q80', 50, 70, 'RGB', dict(quality=80, subsampling=2, restart_marker_blocks=3)),
    ('c444_rst_rows', 40, 56, 'RGB', dict(quality=70, subsampling=0, restart_marker_rows=1)),
    ('gray_q80', 45, 61, 'L', dict(quality=80)),
    ('gray_rst', 33, 40, 'L', dict(quality=55, restart_marker_blocks=2, optimize=True)),
    ('c420_one_mcu', 16, 16, 'RGB', dict(quality=75, subsampling=2)),
    ('c420_tiny', 3, 5, 'RGB', dict(quality=75, subsampling=2)),
    ('c420_big', 120, 160, 'RGB', dict(quality=88, subsampling=2)),
]


def main():
    os.makedirs(OUT, exist_ok=True)
    exp = {}
    for i, (name, h, w, mode, kw) in enumerate(CASES):
        img = picture(h, w, i)
        if name == 'c420_one_mcu':
            img[:] = (90, 140, 200)                                                       # flat: DC coefficients only
        im = Image.fromarray(img if mode == 'RGB' else img[..., 0], mode)
        buf = io.BytesIO()
        im.save(buf, 'JPEG', **kw)
        data = buf.getvalue()
        open(os.path.join(OUT, name + '.jpg'), 'wb').write(data)
        exp[name] = np.asarray(Image.open(io.BytesIO(data)))
        print(name, len(data), 'bytes ->', exp[name].shape)
    np.savez_compressed(os.path.join(OUT, 'expected.npz'), **exp)


if __name__ == '__main__':
    main()
