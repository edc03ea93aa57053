"""CPU oracle of the input pipeline (TEST INFRASTRUCTURE, like oracle/gpv_oracle.py: only tests/ may import it).

Restates, with numpy / scipy in float64, what the reference's data loader does to one image (datasets/coco_generic_dataset.py:49-62,
datasets/coco_datasets.py:26-38,137-150): skimage.transform.resize(anti_aliasing=True) -> (255 x).astype(uint8) -> PIL /
torchvision-0.7 ColorJitter steps in a given order -> horizontal flip -> grayscale -> ToTensor -> Normalize.

**parity unpinned**: scikit-image, PIL and torchvision are not in the image, so none of this can be checked against the real
libraries here; the resize follows scikit-image's published algorithm (Gaussian pre-filter sigma = (s - 1) / 2, order-1 zoom on the
pixel-centre grid, mirrored borders -- the same restatement as gpv1_amd.inference.resize_image), the colour steps follow PIL's
ImageEnhance (blend with black / mean grey / grey image, uint8 rounding after every step), `L = (19595 R + 38470 G + 7471 B +
32768) >> 16`, and torchvision's tensor formula for the hue rotation."""
import numpy as np
from scipy import ndimage as ndi

MEAN = np.array([0.485, 0.456, 0.406])
STD = np.array([0.229, 0.224, 0.225])


def resize_u8(img, size):
    a = np.asarray(img).astype(np.float64)
    factors = np.array([a.shape[0] / size[0], a.shape[1] / size[1], 1.0])
    sigma = np.maximum(0.0, (factors - 1.0) / 2.0)
    if sigma.max() > 0:
        a = ndi.gaussian_filter(a, sigma, mode='mirror')
    out = ndi.zoom(a, 1.0 / factors, order=1, mode='mirror', grid_mode=True)
    return np.floor(np.clip(out, 0.0, 255.0)).astype(np.uint8)        # resize -> [0,1] floats -> (255 x).astype(uint8) truncates


def grey(x):
    x = x.astype(np.uint32)
    return ((19595 * x[..., 0] + 38470 * x[..., 1] + 7471 * x[..., 2] + 32768) >> 16).astype(np.float64)


def _round_u8(x):
    return np.rint(np.clip(x, 0.0, 255.0))


def color_jitter(img_u8, p):
    x = img_u8.astype(np.float64)
    for op in p['order']:
        if op == 0:
            x = _round_u8(x * p['brightness'])
        elif op == 1:
            m = np.floor(grey(x.astype(np.uint8)).mean() + 0.5)
            x = _round_u8(m + p['contrast'] * (x - m))
        elif op == 2:
            l = grey(x.astype(np.uint8))[..., None]
            x = _round_u8(l + p['saturation'] * (x - l))
        else:
            r, g, b = (x[..., i] / 255.0 for i in range(3))
            mx, mn = np.maximum(r, np.maximum(g, b)), np.minimum(r, np.minimum(g, b))
            df = mx - mn
            safe = np.where(df > 0, df, 1.0)
            h = np.where(mx == r, (g - b) / safe, np.where(mx == g, 2.0 + (b - r) / safe, 4.0 + (r - g) / safe)) / 6.0
            h = np.where(df > 0, h - np.floor(h), 0.0)
            s = np.where(mx > 0, df / np.where(mx > 0, mx, 1.0), 0.0)
            v = mx
            h = h + p['hue']
            h = h - np.floor(h)
            h6 = h * 6.0
            i = np.floor(h6).astype(np.int64) % 6
            f = h6 - np.floor(h6)
            pp, q, t = v * (1 - s), v * (1 - s * f), v * (1 - s * (1 - f))
            r2 = np.choose(i, [v, q, pp, pp, t, v])
            g2 = np.choose(i, [t, v, v, q, pp, pp])
            b2 = np.choose(i, [pp, pp, t, v, v, q])
            x = _round_u8(np.stack([r2, g2, b2], -1) * 255.0)
    return x.astype(np.uint8)


def pipeline(img, size, p):
    """-> float64 [3, H, W] normalised image, what the model's NCHW input holds"""
    x = resize_u8(img, size)
    if p['jitter']:
        x = color_jitter(x, p)
    if p['flip']:
        x = x[:, ::-1]
    x = x.astype(np.float64)
    if p['gray']:
        l = grey(x.astype(np.uint8))
        x = np.stack([l, l, l], -1)
    return ((x / 255.0 - MEAN) / STD).transpose(2, 0, 1)
