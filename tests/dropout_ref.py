"""Host restatement (numpy) of the dropout keep decisions the HIP kernels draw -- TEST INFRASTRUCTURE, the checker only.

Before round 6 every dropout mask in the suite was validated against another kernel's mask or by its keep fraction.  hipcc 7.2 then
miscompiled a packed 16-bit min / max chain in a way that keeps the fraction and replicates one word's flags into its neighbours
(DESIGN.md section 0): a correlated mask passes a fraction test.  The kernels' packed compares (attention.hip attn_drop_bits,
conv1x1_stream.hip: __builtin_elementwise_sub_sat on short2) are of that family, so the masks are pinned here to the SPECIFICATION,
bit for bit.  Each function cites what it restates.
"""
import numpy as np

M32 = np.uint64(0xFFFFFFFF)
GOLD64 = 0x9E3779B97F4A7C15


def _u64(a):
    return np.asarray(a, dtype=np.uint64)


def _umul24(x, c):
    """__umul24: product of the low 24 bits of both operands, low 32 bits of the result"""
    return ((x & np.uint64(0xFFFFFF)) * np.uint64(c & 0xFFFFFF)) & M32


def eff_seed(seed, epoch=None):
    """csrc/common.h eff_seed: seed ^ (*epoch * 0x9E3779B97F4A7C15) when a device seed epoch is installed (gpv_set_seed_device)"""
    if epoch is None:
        return int(seed) & 0xFFFFFFFFFFFFFFFF
    return (int(seed) ^ ((int(epoch) * GOLD64) & 0xFFFFFFFFFFFFFFFF)) & 0xFFFFFFFFFFFFFFFF


def drop_thresh(p):
    """csrc/common.h drop_thresh: (uint32) ((double) (float) p * 2^32), clamped"""
    t = float(np.float32(p)) * 4294967296.0
    return int(min(max(t, 0.0), 4294967295.0))


def hash_u32(seed, idx):
    """csrc/common.h hash_u32 (lowbias32 finalizer on index ^ seed, high words folded in)"""
    seed = int(seed)
    idx = _u64(idx)
    x = (idx & M32) ^ np.uint64(seed & 0xFFFFFFFF)
    x = x ^ ((((idx >> np.uint64(32)) & M32) ^ np.uint64((seed >> 32) & 0xFFFFFFFF)) * np.uint64(0x9E3779B9) & M32)
    x = x ^ (x >> np.uint64(16)); x = (x * np.uint64(0x7FEB352D)) & M32
    x = x ^ (x >> np.uint64(15)); x = (x * np.uint64(0x846CA68B)) & M32
    x = x ^ (x >> np.uint64(16))
    return x


def _mix24(x):
    """the shared tail of drop_pair_bits / attn_pair_bits: shifts, xors, two 24-bit multiplies"""
    x = x ^ (x >> np.uint64(16)); x = _umul24(x, 0x85EBCB)
    x = x ^ (x >> np.uint64(13)); x = _umul24(x, 0xC2B2AF)
    x = x ^ (x >> np.uint64(16))
    return x


def drop_pair_bits(seed, pair):
    """csrc/common.h drop_pair_bits: one 32-bit word per PAIR of consecutive flat element indices"""
    seed = int(seed)
    pair = _u64(pair)
    x = ((pair & M32) * np.uint64(0x9E3779B9) + np.uint64(seed & 0xFFFFFFFF)) & M32
    x = (x + ((((pair >> np.uint64(32)) & M32) ^ np.uint64((seed >> 32) & 0xFFFFFFFF)) * np.uint64(0x85EBCA6B) & M32)) & M32
    return _mix24(x)


def keep_flat(seed, idx, p):
    """csrc/common.h drop_keep: element idx is kept iff its 16 bits (low half: even index, high half: odd) >= thresh >> 16.
    The consumers' flat indices: gpv_dropout i; GEMM epilogues (batch * M + m) * N + n; LayerNorm row * cols + c;
    the streaming 1x1 kernel pixel * N + column."""
    idx = _u64(idx)
    w = drop_pair_bits(seed, idx >> np.uint64(1))
    half = np.where((idx & np.uint64(1)) == 1, w >> np.uint64(16), w & np.uint64(0xFFFF))
    return half >= np.uint64(drop_thresh(p) >> 16)


def keep_attention(seed, Bn, H, Sq, Sk, p):
    """csrc/attention.hip: row = (b * H + h) * Sq + q, row seed = hash_u32(seed, row), key pair j = k >> 1 draws
    attn_pair_bits(row seed + j * 0x9E3779B9); key k is kept iff its half (low: even k, high: odd k) read as a SIGNED halfword
    is >= floor(p * 65536) - 32768.  -> bool [Bn, H, Sq, Sk]"""
    rows = np.arange(Bn * H * Sq, dtype=np.uint64)
    rs = hash_u32(seed, rows)[:, None]
    j = (np.arange(Sk, dtype=np.uint64) >> np.uint64(1))[None, :]
    w = _mix24((rs + j * np.uint64(0x9E3779B9)) & M32)
    k_odd = (np.arange(Sk, dtype=np.uint64) & np.uint64(1))[None, :] == 1
    half = np.where(k_odd, w >> np.uint64(16), w & np.uint64(0xFFFF)).astype(np.int64)
    half = np.where(half >= 32768, half - 65536, half)
    ts = (drop_thresh(p) >> 16) - 32768
    return (half >= ts).reshape(Bn, H, Sq, Sk)
