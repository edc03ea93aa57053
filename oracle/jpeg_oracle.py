"""CPU restatement of baseline JPEG decoding as the reference's data loader performs it -- TEST INFRASTRUCTURE ONLY (the checker of
gpv1_amd.jpeg / csrc/jpeg.hip; nothing in the product imports this file).

Reference path: `skio.imread(img_path)` (datasets/coco_generic_dataset.py:54, datasets/coco_datasets.py:157, inference_util.py:10) ->
imageio / Pillow -> libjpeg(-turbo) with its defaults: `dct_method = JDCT_ISLOW`, `do_fancy_upsampling = TRUE`, YCbCr -> RGB.  The
algorithm lives in a third-party dependency that is not part of /root/reference (libjpeg-turbo, bundled with the Pillow wheel the
reference's environment installs); it is restated here from the published algorithm (ITU-T T.81 for the entropy coding, the IJG
`jidctint.c` / `jdsample.c` / `jdcolor.c` integer arithmetic for everything after it) and PINNED bit-exactly against the decoder
itself: tools/gen_golden_jpeg.py encodes and decodes the fixtures of tests/golden/jpeg/ with the Pillow of this image
(tests/test_jpeg_cpu.py).

Scope: baseline sequential DCT (SOF0), 8-bit, one interleaved scan, 1 or 3 components, luma sampling 1x1 / 2x1 / 2x2 with 1x1 chroma,
restart intervals.  Progressive / arithmetic / 12-bit / CMYK files raise `Unsupported`."""
import numpy as np

ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63])


class Unsupported(ValueError):
    pass


class _Huff:
    """T.81 Annex C / F.2.2.3: canonical code tables -> (mincode, maxcode, valptr) per length"""

    def __init__(self, counts, symbols):
        self.symbols = list(symbols)
        self.mincode, self.maxcode, self.valptr = [0] * 17, [-1] * 17, [0] * 17
        code, k = 0, 0
        for ln in range(1, 17):
            self.valptr[ln] = k
            self.mincode[ln] = code
            code += counts[ln - 1]
            k += counts[ln - 1]
            self.maxcode[ln] = code - 1 if counts[ln - 1] else -1
            code <<= 1


class _Bits:
    def __init__(self, data, pos):
        self.d, self.p, self.acc, self.n = data, pos, 0, 0

    def bit(self):
        if self.n == 0:
            b = self.d[self.p]
            self.p += 1
            if b == 0xFF:
                nxt = self.d[self.p]
                if nxt == 0:
                    self.p += 1                       # stuffed zero
                else:                                 # a marker inside entropy data: feed zeros (T.81 F.2.2.5), do not consume it
                    self.p -= 1
                    b = 0
            self.acc, self.n = b, 8
        self.n -= 1
        return (self.acc >> self.n) & 1

    def bits(self, k):
        v = 0
        for _ in range(k):
            v = (v << 1) | self.bit()
        return v

    def decode(self, h):
        code = 0
        for ln in range(1, 17):
            code = (code << 1) | self.bit()
            if h.maxcode[ln] >= 0 and code <= h.maxcode[ln] and code >= h.mincode[ln]:
                return h.symbols[h.valptr[ln] + code - h.mincode[ln]]
        raise ValueError('bad Huffman code')

    def restart(self):
        """byte-align and step over the RSTn marker"""
        self.n = 0
        while not (self.d[self.p] == 0xFF and 0xD0 <= self.d[self.p + 1] <= 0xD7):
            self.p += 1
        self.p += 2


def _extend(v, t):
    return v if t == 0 or v >= (1 << (t - 1)) else v - (1 << t) + 1


def parse(data):
    """JPEG bytes -> dict(width, height, comps=[dict(h, v, tq, blocks [bh, bw, 64] int16 quantised coefficients in natural order)],
    qt = {id: [64] in natural order}, hmax, vmax)"""
    d = bytes(data)
    if d[:2] != b'\xff\xd8':
        raise ValueError('not a JPEG')
    p, qt, hd, ha = 2, {}, {}, {}
    frame, ri, adobe_rgb = None, 0, False
    while True:
        while d[p] != 0xFF:
            p += 1
        while d[p] == 0xFF:
            p += 1
        m = d[p]
        p += 1
        if m in (0xD8, 0x01) or 0xD0 <= m <= 0xD7:
            continue
        if m == 0xD9:
            raise ValueError('no scan')
        ln = (d[p] << 8) | d[p + 1]
        seg = d[p + 2:p + ln]
        if m == 0xDB:
            q = 0
            while q < len(seg):
                pq, tq = seg[q] >> 4, seg[q] & 15
                q += 1
                if pq:
                    raise Unsupported('16-bit quantisation table')
                t = np.zeros(64, np.int32)
                t[ZIGZAG] = np.frombuffer(seg[q:q + 64], np.uint8)
                qt[tq] = t
                q += 64
        elif m == 0xC4:
            q = 0
            while q < len(seg):
                tc, th = seg[q] >> 4, seg[q] & 15
                counts = list(seg[q + 1:q + 17])
                n = sum(counts)
                (ha if tc else hd)[th] = _Huff(counts, seg[q + 17:q + 17 + n])
                q += 17 + n
        elif m == 0xC0 or m == 0xC1:
            if seg[0] != 8:
                raise Unsupported('sample precision %d' % seg[0])
            H, W, nf = (seg[1] << 8) | seg[2], (seg[3] << 8) | seg[4], seg[5]
            comps = [dict(id=seg[6 + 3 * i], h=seg[7 + 3 * i] >> 4, v=seg[7 + 3 * i] & 15, tq=seg[8 + 3 * i]) for i in range(nf)]
            frame = dict(width=W, height=H, comps=comps)
        elif m in (0xC2, 0xC3, 0xC5, 0xC6, 0xC7, 0xC9, 0xCA, 0xCB, 0xCD, 0xCE, 0xCF):
            raise Unsupported('SOF%d (progressive / lossless / arithmetic)' % (m - 0xC0))
        elif m == 0xEE:
            if len(seg) >= 12 and seg[:5] == b'Adobe' and seg[11] == 0:
                adobe_rgb = True                      # components are RGB / CMYK, no YCbCr transform
        elif m == 0xDD:
            ri = (seg[0] << 8) | seg[1]
        elif m == 0xDA:
            if frame is None:
                raise ValueError('SOS before SOF')
            ns = seg[0]
            if ns != len(frame['comps']):
                raise Unsupported('non-interleaved scans')
            for i in range(ns):
                c = frame['comps'][i]
                if seg[1 + 2 * i] != c['id']:
                    raise Unsupported('scan component order')
                c['td'], c['ta'] = seg[2 + 2 * i] >> 4, seg[2 + 2 * i] & 15
            p += ln
            break
        p += ln
    comps = frame['comps']
    if len(comps) not in (1, 3):
        raise Unsupported('%d components' % len(comps))
    if len(comps) == 3 and adobe_rgb:
        raise Unsupported('Adobe RGB-coded file (no YCbCr transform)')
    hmax, vmax = max(c['h'] for c in comps), max(c['v'] for c in comps)
    if len(comps) == 3 and ((comps[0]['h'], comps[0]['v']) not in ((1, 1), (2, 1), (2, 2)) or any((c['h'], c['v']) != (1, 1) for c in comps[1:])):
        raise Unsupported('sampling factors')
    if len(comps) == 1:
        comps[0]['h'] = comps[0]['v'] = hmax = vmax = 1         # a single-component scan is never interleaved (T.81 A.2.2)
    W, H = frame['width'], frame['height']
    mx, my = -(-W // (8 * hmax)), -(-H // (8 * vmax))
    for c in comps:
        c['blocks'] = np.zeros((my * c['v'], mx * c['h'], 64), np.int16)
    br = _Bits(d, p)
    pred = [0] * len(comps)
    for mcu in range(mx * my):
        if ri and mcu and mcu % ri == 0:
            br.restart()
            pred = [0] * len(comps)
        y0, x0 = divmod(mcu, mx)
        for ci, c in enumerate(comps):
            for v in range(c['v']):
                for h in range(c['h']):
                    blk = c['blocks'][y0 * c['v'] + v, x0 * c['h'] + h]
                    t = br.decode(hd[c['td']])
                    pred[ci] += _extend(br.bits(t), t) if t else 0
                    blk[0] = pred[ci]
                    k = 1
                    while k < 64:
                        rs = br.decode(ha[c['ta']])
                        r, s = rs >> 4, rs & 15
                        if s == 0:
                            if r != 15:
                                break
                            k += 16
                            continue
                        k += r
                        blk[ZIGZAG[k]] = _extend(br.bits(s), s)
                        k += 1
    frame.update(qt=qt, hmax=hmax, vmax=vmax, mcus_x=mx, mcus_y=my, restart_interval=ri)
    return frame


# ---- IJG jidctint.c (jpeg_idct_islow): the accurate integer inverse DCT, exact ------------------------------------------------------
CB, P1 = 13, 2
F_0_298, F_0_390, F_0_541, F_0_765, F_0_899, F_1_175 = 2446, 3196, 4433, 6270, 7373, 9633
F_1_501, F_1_847, F_1_961, F_2_053, F_2_562, F_3_072 = 12299, 15137, 16069, 16819, 20995, 25172


def _pass(x, shift):
    """one 1-D pass over axis -1 of int64 data [..., 8]"""
    i0, i1, i2, i3, i4, i5, i6, i7 = (x[..., k] for k in range(8))
    z1 = (i2 + i6) * F_0_541
    t2 = z1 - i6 * F_1_847
    t3 = z1 + i2 * F_0_765
    t0 = (i0 + i4) << CB
    t1 = (i0 - i4) << CB
    t10, t13, t11, t12 = t0 + t3, t0 - t3, t1 + t2, t1 - t2
    o0, o1, o2, o3 = i7, i5, i3, i1
    z1, z2, z3, z4 = o0 + o3, o1 + o2, o0 + o2, o1 + o3
    z5 = (z3 + z4) * F_1_175
    o0, o1, o2, o3 = o0 * F_0_298, o1 * F_2_053, o2 * F_3_072, o3 * F_1_501
    z1, z2, z3, z4 = -z1 * F_0_899, -z2 * F_2_562, -z3 * F_1_961 + z5, -z4 * F_0_390 + z5
    o0, o1, o2, o3 = o0 + z1 + z3, o1 + z2 + z4, o2 + z2 + z3, o3 + z1 + z4
    r = np.stack([t10 + o3, t11 + o2, t12 + o1, t13 + o0, t13 - o0, t12 - o1, t11 - o2, t10 - o3], -1)
    return (r + (1 << (shift - 1))) >> shift


def idct_islow(blocks, q):
    """blocks [..., 64] quantised coefficients (natural order), q [64] -> samples [..., 8, 8] uint8"""
    x = (blocks.astype(np.int64) * q.astype(np.int64)).reshape(blocks.shape[:-1] + (8, 8))
    ws = _pass(np.swapaxes(x, -1, -2), CB - P1)              # pass 1: columns (axis -2 of x)
    ws = np.swapaxes(ws, -1, -2)
    out = _pass(ws, CB + P1 + 3)                              # pass 2: rows
    return np.clip(out + 128, 0, 255).astype(np.uint8)


def planes(frame):
    """component sample planes [bh*8, bw*8] uint8 (whole MCUs, as libjpeg's coefficient controller produces them)"""
    out = []
    for c in frame['comps']:
        s = idct_islow(c['blocks'], frame['qt'][c['tq']])                # [bh, bw, 8, 8]
        bh, bw = s.shape[:2]
        out.append(s.transpose(0, 2, 1, 3).reshape(bh * 8, bw * 8))
    return out


# ---- IJG jdsample.c: "fancy" (triangle filter) chroma upsampling, exact ------------------------------------------------------------
def _h2v1_fancy(p, w):
    """p [rows, cw] -> [rows, 2*cw]; cw = downsampled width (only real columns are used)"""
    a = p[:, :w].astype(np.int32)
    prev = np.concatenate([a[:, :1], a[:, :-1]], 1)
    nxt = np.concatenate([a[:, 1:], a[:, -1:]], 1)
    even = (a * 3 + prev + 1) >> 2
    odd = (a * 3 + nxt + 2) >> 2
    even[:, 0] = a[:, 0]
    odd[:, -1] = a[:, -1]
    if w == 1:
        even[:, 0] = odd[:, 0] = a[:, 0]
    out = np.empty((a.shape[0], 2 * w), np.int32)
    out[:, 0::2], out[:, 1::2] = even, odd
    return out.astype(np.uint8)


def _h2v2_fancy(p, w, h):
    """p [>= h rows, >= w cols] -> [2*h, 2*w]: rows above the first / below the last real row are copies of it (the context rows of
    jdmainct.c at the image edges)"""
    a = p[:h, :w].astype(np.int32)
    up = np.concatenate([a[:1], a[:-1]], 0)
    dn = np.concatenate([a[1:], a[-1:]], 0)
    out = np.empty((2 * h, 2 * w), np.int32)
    for v, nb in ((0, up), (1, dn)):
        cs = a * 3 + nb                                       # column sums: 3/4 nearer row + 1/4 further row
        last = np.concatenate([cs[:, :1], cs[:, :-1]], 1)
        nxt = np.concatenate([cs[:, 1:], cs[:, -1:]], 1)
        even = (cs * 3 + last + 8) >> 4
        odd = (cs * 3 + nxt + 7) >> 4
        even[:, 0] = (cs[:, 0] * 4 + 8) >> 4
        odd[:, -1] = (cs[:, -1] * 4 + 7) >> 4
        out[v::2, 0::2], out[v::2, 1::2] = even, odd
    return out.astype(np.uint8)


# ---- IJG jdcolor.c: YCbCr -> RGB with its 16-bit fixed-point tables, exact ---------------------------------------------------------
def _fix(x):
    return int(x * 65536 + 0.5)


_X = np.arange(256, dtype=np.int64) - 128
CR_R = (_fix(1.40200) * _X + 32768) >> 16
CB_B = (_fix(1.77200) * _X + 32768) >> 16
CR_G = -_fix(0.71414) * _X
CB_G = -_fix(0.34414) * _X + 32768


def ycc_to_rgb(y, cb, cr):
    y = y.astype(np.int64)
    r = y + CR_R[cr]
    g = y + ((CB_G[cb] + CR_G[cr]) >> 16)
    b = y + CB_B[cb]
    return np.clip(np.stack([r, g, b], -1), 0, 255).astype(np.uint8)


def decode(data):
    """JPEG bytes -> [H, W, 3] uint8 RGB, or [H, W] uint8 for a single-component file (what skimage.io.imread returns)"""
    f = parse(data)
    W, H = f['width'], f['height']
    pl = planes(f)
    if len(pl) == 1:
        return pl[0][:H, :W].copy()
    hs, vs = f['hmax'], f['vmax']
    cw, ch = -(-W // hs), -(-H // vs)                          # downsampled_width / height of the chroma components
    ups = []
    for p in pl[1:]:
        if (hs, vs) == (1, 1):
            u = p
        elif cw <= 2:                                         # jdsample.c: the triangle filter needs downsampled_width > 2,
            u = np.repeat(np.repeat(p[:ch, :cw], vs, 0), hs, 1)   # narrower components are replicated (h2v1_upsample / h2v2_upsample)
        elif (hs, vs) == (2, 1):
            u = _h2v1_fancy(p, cw)
        else:
            u = _h2v2_fancy(p, cw, ch)
        ups.append(u[:H, :W])
    return ycc_to_rgb(pl[0][:H, :W], ups[0], ups[1])
