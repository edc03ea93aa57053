"""JPEG decoding, CPU side: the oracle against the Pillow / libjpeg-turbo goldens (tools/gen_golden_jpeg.py), and the
product's HOST stage (gpv_jpeg_parse: header + Huffman decoding, no GPU work) against the oracle's coefficients."""
import glob
import io
import os

import numpy as np
import pytest

from oracle import jpeg_oracle as J

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'jpeg')
FILES = sorted(glob.glob(os.path.join(GOLD, '*.jpg')))
NAMES = [os.path.basename(f)[:-4] for f in FILES]


def test_fixture_set_is_complete():
    exp = np.load(os.path.join(GOLD, 'expected.npz'))
    assert len(FILES) == 14 and sorted(exp.files) == NAMES


@pytest.mark.parametrize('name', NAMES)
def test_oracle_is_bit_exact_against_the_reference_decoder(name):
    exp = np.load(os.path.join(GOLD, 'expected.npz'))[name]
    out = J.decode(open(os.path.join(GOLD, name + '.jpg'), 'rb').read())
    assert out.shape == exp.shape and out.dtype == np.uint8
    assert np.array_equal(out, exp)


@pytest.mark.parametrize('name', NAMES)
def test_host_entropy_decoder_matches_the_oracle(name):
    import gpv1_amd.hip as hip
    data = open(os.path.join(GOLD, name + '.jpg'), 'rb').read()
    info = hip.jpeg_parse(data)                                    # header pass
    fr = J.parse(data)
    assert (info.width, info.height, info.ncomp, info.hmax, info.vmax) == (fr['width'], fr['height'], len(fr['comps']), fr['hmax'], fr['vmax'])
    buf = np.full(info.coef_count, 7, np.int16)
    hip.jpeg_parse(data, buf)
    for c, comp in enumerate(fr['comps']):
        assert (info.bh[c], info.bw[c]) == comp['blocks'].shape[:2]
        mine = buf[info.coef_offset[c]:info.coef_offset[c] + comp['blocks'].size].reshape(comp['blocks'].shape)
        assert np.array_equal(mine, comp['blocks'])
        assert np.array_equal(np.array(list(info.quant[c])), fr['qt'][comp['tq']])


def test_host_decoder_rejects_what_it_does_not_decode():
    import gpv1_amd.hip as hip
    from PIL import Image
    img = (np.random.RandomState(0).rand(40, 56, 3) * 255).astype(np.uint8)
    buf = io.BytesIO()
    Image.fromarray(img).save(buf, 'JPEG', progressive=True)
    with pytest.raises(hip.JpegUnsupported):
        hip.jpeg_parse(buf.getvalue())
    with pytest.raises(J.Unsupported):
        J.parse(buf.getvalue())
    buf = io.BytesIO()
    Image.fromarray(np.dstack([img, img[..., :1]]), 'CMYK').save(buf, 'JPEG')
    with pytest.raises(hip.JpegUnsupported):
        hip.jpeg_parse(buf.getvalue())
    good = open(FILES[0], 'rb').read()
    # an Adobe APP14 segment with transform = 0 (RGB-coded components) spliced in front of the frame header
    adobe = good[:2] + b'\xff\xee\x00\x0eAdobe\x00\x64\x00\x00\x00\x00\x00' + good[2:]
    with pytest.raises(hip.JpegUnsupported):
        hip.jpeg_parse(adobe)
    with pytest.raises(J.Unsupported):
        J.parse(adobe)
    with pytest.raises(ValueError):
        hip.jpeg_parse(b'not a jpeg at all')
    info = hip.jpeg_parse(good)
    with pytest.raises(ValueError):                                 # capacity too small
        hip.jpeg_parse(good, np.zeros(info.coef_count - 1, np.int16))
    for cut in (len(good) // 3, len(good) // 2, len(good) - 40):    # truncated entropy data must not read out of bounds: it
        out = np.zeros(info.coef_count, np.int16)                   # decodes zeros past the end or reports a bad code
        try:
            hip.jpeg_parse(good[:cut], out)
        except ValueError:
            pass


def test_random_images_round_trip_through_both_decoders():
    """fresh encodes (not fixtures): oracle == Pillow on random content, every sampling mode, odd sizes"""
    from PIL import Image
    r = np.random.RandomState(3)
    for i, (h, w, sub, q) in enumerate([(24, 40, 0, 92), (31, 17, 1, 70), (45, 38, 2, 50), (9, 9, 2, 95)]):
        img = (r.rand(h, w, 3) * 255).astype(np.uint8)
        buf = io.BytesIO()
        Image.fromarray(img).save(buf, 'JPEG', quality=q, subsampling=sub)
        exp = np.asarray(Image.open(io.BytesIO(buf.getvalue())))
        assert np.array_equal(J.decode(buf.getvalue()), exp), (h, w, sub, q)


def test_narrow_images_use_replication_not_the_triangle_filter():
    """jdsample.c picks the fancy up-sampler only for components wider than two samples: images up to four pixels wide (and any
    such height / sampling mix) are replicated -- found by the random-file test on the device decoder"""
    from PIL import Image
    r = np.random.RandomState(5)
    for i in range(120):
        h, w = int(r.randint(1, 40)), int(r.randint(1, 8))
        if i % 2:
            h, w = w, h
        img = (r.rand(h, w, 3) * 255).astype(np.uint8)
        buf = io.BytesIO()
        Image.fromarray(img).save(buf, 'JPEG', quality=int(r.randint(20, 100)), subsampling=int(r.randint(0, 3)))
        exp = np.asarray(Image.open(io.BytesIO(buf.getvalue())))
        assert np.array_equal(J.decode(buf.getvalue()), exp), (h, w)


def test_decoder_refuses_a_header_that_declares_a_huge_image_before_sizing_anything():
    """a ~300-byte file may declare 65535 x 65535 pixels: DeviceJpegDecoder checks the header against max_pixels before any buffer
    is sized from it (26 GB pinned + offsets wrapping gpv_jpeg_desc's 32-bit fields)"""
    from gpv1_amd.jpeg import DeviceJpegDecoder
    data = bytearray(open(os.path.join(GOLD, 'c444_odd_q97.jpg'), 'rb').read())
    i = data.index(b'\xff\xc0')                       # SOF0: marker, length(2), precision(1), height(2), width(2)
    data[i + 5:i + 9] = b'\xff\xff\xff\xff'
    dec = DeviceJpegDecoder(threads=1)
    with pytest.raises(ValueError, match='max_pixels'):
        dec([bytes(data)])
    small = DeviceJpegDecoder(threads=1, max_pixels=100)
    with pytest.raises(ValueError, match='max_pixels'):
        small([open(os.path.join(GOLD, 'c444_odd_q97.jpg'), 'rb').read()])
