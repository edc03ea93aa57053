// Multi-head attention core for short sequences (Sk <= 320) on gfx950 MFMA.
//
// Forward / dQ kernels: one workgroup (4 waves) per (batch, head, 64-query tile); each wave owns
// 16 queries and holds the WHOLE score row block in registers (no online softmax needed: Sk <= 320).
//   S^T tile  = mfma(A = K rows (keys), B = Q cols (queries))  -> lane holds S^T[key=4g+i][q=lane&15]
//   so a lane owns ONE query: row max / sum are a register reduction + two xor-shuffles (16, 32),
//   and the exponentiated scores are directly the B-operand fragments of  O^T = V^T P^T
//   (k-slot order {4g+i} U {16+4g+i} is used consistently for both operands).
// dK/dV kernel: one workgroup per (batch, head, 64-key tile); each wave owns 16 keys and streams
// query chunks through LDS (Q, dO and their transposes).
// K is staged [key][d] (ds_read_b128 fragments), V / K / Q / dO transposes are staged [d][t] through
// a register transpose (pairs of rows -> ds_write_b32) and read with ds_read_b64.
// T = bf16: bf16 I/O.  T = float: fp32 I/O, every operand split hi+lo bf16 (3 MFMAs per product).
#include "common.h"
#include "../../include/gpv_hip.h"

namespace {

int g_bwd1_mode = tune_env("GPV_ATTN_BWD1", 1);   // gpv_set_option(GPV_OPT_ATTN_BWD1, .)
long g_bwd1_launches = 0;      // gpv_set_option(GPV_OPT_ATTN_BWD1_LAUNCHES, .)

struct AttnK {
  const void* q; const void* k; const void* v; void* o;
  int64_t q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs;
  int B, H, Sq, Sk, dh, skp, nsplit, sqp;
  float scale;
  const uint8_t* kpm; int causal;
  uint32_t dthresh; float dscale; uint64_t seed; const uint64_t* seed_dev;
  float* lse;
  const void* dout; int64_t do_bs, do_rs;
  void* dq; void* dk; void* dv;
};

// Attention dropout: the keep decisions of a (batch, head, query) row come from one 32-bit word per PAIR of keys, 16 bits per
// element (drop probability floor(p * 65536) / 65536), mixed from the row's seed with full-rate VALU only (shifts, xors, 24-bit
// multiplies).  The 64-bit counter hash of common.h per SCORE (three quarter-rate 32-bit multiplies) made the dh = 32 kernels
// spend more on the mask than on the softmax; forward, dQ and dK/dV kernels must agree on this function.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t ATTN_PAIR_STEP = 0x9E3779B9u;
__device__ __forceinline__ uint32_t attn_row_seed(uint64_t seed, uint64_t row) { return hash_u32(seed, row); }
__device__ __forceinline__ uint32_t attn_pair_bits(uint32_t x) {     // x = row seed + pair index * ATTN_PAIR_STEP
  x ^= x >> 16; x = __umul24(x, 0x85EBCBu);
  x ^= x >> 13; x = __umul24(x, 0xC2B2AFu);
  x ^= x >> 16;
  return x;
}
// An element is kept when its 16 bits, read as a SIGNED halfword, are >= ts = floor(p * 65536) - 32768 (same rate as an unsigned
// test against floor(p * 65536); the signed form lets the forward mask a packed pair of bf16 probabilities with three packed-math
// instructions -- saturating subtract, arithmetic shift, and-not -- instead of two compares and two selects on the floats).
__device__ __forceinline__ int attn_ts(uint32_t dthresh) { return (int)(dthresh >> 16) - 32768; }
__device__ __forceinline__ bool attn_keep_lo(uint32_t w, int ts) { return (int)(short)(w & 0xffffu) >= ts; }
__device__ __forceinline__ bool attn_keep_hi(uint32_t w, int ts) { return ((int)w >> 16) >= ts; }
__device__ __forceinline__ uint32_t attn_drop_bits(uint32_t w, uint32_t ts2) {       // 0xffff in every DROPPED half; ts2 = ts | ts << 16
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  const s16x2 d = __builtin_elementwise_sub_sat(__builtin_bit_cast(s16x2, w), __builtin_bit_cast(s16x2, ts2));
  return __builtin_bit_cast(uint32_t, (s16x2)(d >> (s16x2){15, 15}));
}

template <typename T> struct R8 {  // 8 staged values as floats
  float v[8];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = 0.f;
  }
  __device__ __forceinline__ void load(const T* p) { Ld8<T>::ld(p, v); }
  __device__ __forceinline__ bf16x8 hi() const {
    bf16x8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = (bf16)v[i];
    return r;
  }
  __device__ __forceinline__ bf16x8 lo() const {
    bf16x8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = (bf16)(v[i] - (float)(bf16)v[i]);
    return r;
  }
};

// 8 consecutive elements as loaded (16 bytes of bf16, 32 of fp32): what a thread holds while a batch of staging loads is in flight
template <typename T> struct Raw8;
template <> struct Raw8<bf16> {
  bf16x8 r;
  __device__ __forceinline__ void zero() { r = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; }
  __device__ __forceinline__ void load(const bf16* p) { r = *reinterpret_cast<const bf16x8*>(p); }
  __device__ __forceinline__ R8<bf16> get() const {
    R8<bf16> x;
#pragma unroll
    for (int i = 0; i < 8; ++i) x.v[i] = (float)r[i];
    return x;
  }
};
template <> struct Raw8<float> {
  float v[8];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = 0.f;
  }
  __device__ __forceinline__ void load(const float* p) { Ld8<float>::ld(p, v); }
  __device__ __forceinline__ R8<float> get() const {
    R8<float> x;
#pragma unroll
    for (int i = 0; i < 8; ++i) x.v[i] = v[i];
    return x;
  }
};

// The K / V staging of the per-(batch, head) kernels, all loads first: with run-time loop bounds hipcc kept one load in flight
// per thread and iteration (8 dependent round trips to L2/HBM before the first MFMA: the waves spent 57 % of their life in
// s_waitcnt, PMC in profiles/r02_pmc_attention.txt).  MAXR bounds the rows at compile time, so the loops unroll and every
// load of the block's staging is issued before the first LDS store.
// Round 5: the loads are UNCONDITIONAL (a row / slice beyond the tile re-reads the last valid one) and the zero padding is applied in
// store(): with `if (in range) load; else zero` per element hipcc branched around every load and drained the queue (vmcnt(0)) at joins
// (seen in the ISA) -- "all loads first" was several dependent round trips.
template <typename T, int DH, int MAXR, int NTHR> struct RowBatch {      // [rows][DH] row-major -> LDS [rows_pad][DH + 8]
  static constexpr int SL = DH / 8, IT = (MAXR * SL + NTHR - 1) / NTHR;
  Raw8<T> x[IT];
  int rv_, dh_;
  __device__ __forceinline__ void load(const T* g, int64_t rs, int rows_valid, int rows_pad, int dh) {
    rv_ = rows_valid; dh_ = dh;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int idx = threadIdx.x + it * NTHR, r = idx / SL, sl = idx - r * SL;
      x[it].load(g + (int64_t)min(r, rows_valid - 1) * rs + min(sl * 8, dh - 8));
    }
  }
  template <bool PRECISE> __device__ __forceinline__ void store(int rows_pad, bf16* hi, bf16* lo) const {
    constexpr int PITCH = DH + 8;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int idx = threadIdx.x + it * NTHR, r = idx / SL, sl = idx - r * SL;
      if (idx < rows_pad * SL) {
        R8<T> v = x[it].get();
        if (r >= rv_ || sl * 8 >= dh_) v.zero();
        *reinterpret_cast<bf16x8*>(hi + r * PITCH + sl * 8) = v.hi();
        if (PRECISE) *reinterpret_cast<bf16x8*>(lo + r * PITCH + sl * 8) = v.lo();
      }
    }
  }
};
template <typename T, int DH, int MAXR, int NTHR> struct ColBatch {      // transposed: LDS [DH][pitch], element (d, r) = g[r][d]
  static constexpr int DG = DH / 8, IT = ((MAXR / 2) * DG + NTHR - 1) / NTHR;
  Raw8<T> a[IT], b[IT];
  int rv_;
  __device__ __forceinline__ void load(const T* g, int64_t rs, int rows_valid, int rows_pad) {
    const int np = rows_pad / 2;
    rv_ = rows_valid;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int idx = threadIdx.x + it * NTHR, dg = min(idx / np, DG - 1), rp = idx - (idx / np) * np;
      a[it].load(g + (int64_t)min(2 * rp, rows_valid - 1) * rs + dg * 8);
      b[it].load(g + (int64_t)min(2 * rp + 1, rows_valid - 1) * rs + dg * 8);
    }
  }
  template <bool PRECISE> __device__ __forceinline__ void store(int rows_pad, int pitch, bf16* hi, bf16* lo) const {
    const int np = rows_pad / 2;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int idx = threadIdx.x + it * NTHR, dg = idx / np, rp = idx - dg * np;
      if (idx < np * DG) {
        R8<T> u = a[it].get(), w = b[it].get();
        if (2 * rp >= rv_) u.zero();
        if (2 * rp + 1 >= rv_) w.zero();
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          bf16x2 h; h[0] = (bf16)u.v[c]; h[1] = (bf16)w.v[c];
          *reinterpret_cast<bf16x2*>(hi + (dg * 8 + c) * pitch + 2 * rp) = h;
          if (PRECISE) {
            bf16x2 l; l[0] = (bf16)(u.v[c] - (float)h[0]); l[1] = (bf16)(w.v[c] - (float)h[1]);
            *reinterpret_cast<bf16x2*>(lo + (dg * 8 + c) * pitch + 2 * rp) = l;
          }
        }
      }
    }
  }
};

// stage a [rows_valid x dh] row-major global tile into LDS [rows_pad][PITCH] (k-contiguous), zero padded
template <typename T, bool PRECISE, int DHK, int NTHR = 256>
__device__ __forceinline__ void stage_rows(const T* g, int64_t rs, int rows_valid, int rows_pad, int dh, bf16* hi, bf16* lo) {
  constexpr int PITCH = DHK + 8;
  constexpr int SL = DHK / 8;
  for (int idx = threadIdx.x; idx < rows_pad * SL; idx += NTHR) {
    int r = idx / SL, sl = idx - r * SL;
    R8<T> x;
    if (r < rows_valid && sl * 8 < dh) x.load(g + (int64_t)r * rs + sl * 8); else x.zero();
    *reinterpret_cast<bf16x8*>(hi + r * PITCH + sl * 8) = x.hi();
    if (PRECISE) *reinterpret_cast<bf16x8*>(lo + r * PITCH + sl * 8) = x.lo();
  }
}
// stage transposed: LDS [DHV][pitch] with element (d, r) = g[r][d]; rows_pad even
template <typename T, bool PRECISE, int DHV, int NTHR = 256>
__device__ __forceinline__ void stage_cols(const T* g, int64_t rs, int rows_valid, int rows_pad, int pitch, bf16* hi, bf16* lo) {
  constexpr int DG = DHV / 8;
  const int np = rows_pad / 2;
  for (int idx = threadIdx.x; idx < np * DG; idx += NTHR) {
    int dg = idx / np, rp = idx - dg * np;
    R8<T> a, b;
    if (2 * rp < rows_valid) a.load(g + (int64_t)(2 * rp) * rs + dg * 8); else a.zero();
    if (2 * rp + 1 < rows_valid) b.load(g + (int64_t)(2 * rp + 1) * rs + dg * 8); else b.zero();
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      bf16x2 h; h[0] = (bf16)a.v[c]; h[1] = (bf16)b.v[c];
      *reinterpret_cast<bf16x2*>(hi + (dg * 8 + c) * pitch + 2 * rp) = h;
      if (PRECISE) {
        bf16x2 l; l[0] = (bf16)(a.v[c] - (float)h[0]); l[1] = (bf16)(b.v[c] - (float)h[1]);
        *reinterpret_cast<bf16x2*>(lo + (dg * 8 + c) * pitch + 2 * rp) = l;
      }
    }
  }
}

__device__ __forceinline__ bf16x8 ld_pair64(const bf16* p0, const bf16* p1) {
  bf16x4 a = *reinterpret_cast<const bf16x4*>(p0);
  bf16x4 b = *reinterpret_cast<const bf16x4*>(p1);
  bf16x8 r;
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3]; r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
  return r;
}

// MODE 0: forward (writes o, lse).  MODE 1: dQ (reads dout, o, lse; writes dq)
// Workgroup = 8 waves on one (batch, head) [x one of gridDim-derived query splits]; FULL: every one of the NT key tiles is live
// (skp == 16 NT: the encoder's 300 keys, the decoder's 100), which makes the tile loops branch-free -- hipcc then overlaps the LDS
// reads of a tile group with the MFMAs of the one before.
constexpr int QW = 8, QTHR = QW * 64;
template <typename T, int DHK, int DHV, int NT, int MODE, bool MASKED, bool FULL>
__global__ __launch_bounds__(QTHR) void attn_q_kernel(AttnK p) {
  if (p.dthresh) p.seed = eff_seed(p.seed, p.seed_dev);
  constexpr bool PRECISE = sizeof(T) == 4;
  constexpr int KP = DHK + 8;
  constexpr int KC = DHK / 32;
  constexpr int DT = DHV / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16* sm = reinterpret_cast<bf16*>(smem_raw);
  const int skp = FULL ? NT * 16 : p.skp, ntr = FULL ? NT : skp / 16, vtp = skp + 8;
  // LDS carve: K[skp][KP] (hi,lo) | X^T[DHV][vtp] (hi,lo)  (X = V fwd, K for dQ) | (dQ only) V[skp][KP] (hi,lo)
  bf16* Kh = sm;
  bf16* Kl = Kh + (PRECISE ? skp * KP : 0);
  bf16* Th = Kl + skp * KP;
  bf16* Tl = Th + (PRECISE ? DHV * vtp : 0);
  bf16* Vh = Tl + DHV * vtp;
  bf16* Vl = Vh + (PRECISE ? skp * KP : 0);

  // XCD-aware block order (workgroup i runs on XCD i % 8, used for speed only): every XCD gets a CONTIGUOUS range of
  // (batch, head, split) triples, so the eight heads of one batch element -- interleaved 64-byte slices of the same K / V / Q
  // rows -- and the query splits of one head share one L2 instead of pulling every 128-byte line into several.
  int b, h, xs;
  {
    const int nsp = p.nsplit, total = (int)gridDim.x;
    const int qd = total >> 3, r = total & 7, xcd = (int)blockIdx.x & 7, loc = (int)blockIdx.x >> 3;
    const int v = (xcd < r ? xcd * (qd + 1) : r * (qd + 1) + (xcd - r) * qd) + loc;
    xs = v % nsp;
    const int bh = v / nsp;
    h = bh % p.H; b = bh / p.H;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
  const T* kg = reinterpret_cast<const T*>(p.k) + b * p.k_bs + h * p.dh;
  const T* vg = reinterpret_cast<const T*>(p.v) + b * p.v_bs + h * p.dh;
  // One block serves `per` consecutive 16-query tiles of its (batch, head): K and V are staged ONCE and every wave walks its
  // share of the tiles.  The per-tile operands (Q rows; dO, O, lse for dQ) are fetched one tile ahead -- the first before the
  // staging -- so that their L2 / HBM round trip (1-2 us, once per tile) overlaps the previous tile's softmax.
  const int nqt = (p.Sq + 15) >> 4;
  const int per = (nqt + p.nsplit - 1) / p.nsplit;
  const int qt_end = min(nqt, (xs + 1) * per);
  const int qt_first = xs * per + wave;
  Raw8<T> nq[KC], nd[MODE ? KC : 1], no[MODE ? KC : 1];
  float nlse = 0.f;
  // Unconditional loads (round 5): a tile / row beyond the range re-reads the LAST query row -- such rows are never stored -- and the
  // zero padding of the head dimension (dh = 48 in a 64-wide fragment) is applied where the fragment is USED.  With `if (ok) load; else
  // zero` hipcc branched around the load and waited for it -- s_waitcnt vmcnt(0) -- at the join: the "prefetch" of the next tile's Q
  // was an exposed L2 round trip per tile (seen in the ISA).
  auto fetch = [&](int qt_) {
    const int q_ = min(qt_ * 16 + (lane & 15), p.Sq - 1);
    const T* qp = reinterpret_cast<const T*>(p.q) + b * p.q_bs + (int64_t)q_ * p.q_rs + h * p.dh;
  #pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      const int d0 = min(kc * 32 + g * 8, p.dh - 8);
      nq[kc].load(qp + d0);
      if constexpr (MODE == 1) {
        const T* dop = reinterpret_cast<const T*>(p.dout) + b * p.do_bs + (int64_t)q_ * p.do_rs + h * p.dh;
        const T* op = reinterpret_cast<const T*>(p.o) + b * p.o_bs + (int64_t)q_ * p.o_rs + h * p.dh;
        nd[kc].load(dop + d0); no[kc].load(op + d0);
      }
    }
    if constexpr (MODE == 1) nlse = p.lse[((int64_t)b * p.H + h) * p.Sq + q_];
  };
  fetch(qt_first);
  if constexpr (PRECISE) {            // (fp32 test path: the plain loops)
    stage_rows<T, PRECISE, DHK, QTHR>(kg, p.k_rs, p.Sk, skp, p.dh, Kh, Kl);
    if (MODE == 0) stage_cols<T, PRECISE, DHV, QTHR>(vg, p.v_rs, p.Sk, skp, vtp, Th, Tl);
    else {
      stage_cols<T, PRECISE, DHV, QTHR>(kg, p.k_rs, p.Sk, skp, vtp, Th, Tl);
      stage_rows<T, PRECISE, DHK, QTHR>(vg, p.v_rs, p.Sk, skp, p.dh, Vh, Vl);
    }
  } else {
    RowBatch<T, DHK, NT * 16, QTHR> kr;
    ColBatch<T, DHV, NT * 16, QTHR> xc;
    kr.load(kg, p.k_rs, p.Sk, skp, p.dh);
    xc.load(MODE == 0 ? vg : kg, MODE == 0 ? p.v_rs : p.k_rs, p.Sk, skp);
    if constexpr (MODE == 0) {
      kr.template store<PRECISE>(skp, Kh, Kl);
      xc.template store<PRECISE>(skp, vtp, Th, Tl);
    } else {
      RowBatch<T, DHK, NT * 16, QTHR> vr;
      vr.load(vg, p.v_rs, p.Sk, skp, p.dh);
      kr.template store<PRECISE>(skp, Kh, Kl);
      xc.template store<PRECISE>(skp, vtp, Th, Tl);
      vr.template store<PRECISE>(skp, Vh, Vl);
    }
  }
  // Dead keys (index >= Sk, key-padding mask set) are an additive -inf per key, staged once per block: it rides in as the initial
  // value of the score accumulators (an LDS read in place of the zeroing moves), so neither the forward nor dQ spends a
  // per-score instruction on padding.  Only the causal mask, which depends on the query, is tested per score.
  float* kbias = reinterpret_cast<float*>(MODE ? Vl + skp * KP : Vh);
  for (int idx = threadIdx.x; idx < skp; idx += QTHR)
    kbias[idx] = (idx >= p.Sk || (p.kpm && p.kpm[(int64_t)b * p.Sk + idx])) ? -INFINITY : 0.f;
  __syncthreads();

  for (int qt = qt_first; qt < qt_end; qt += QW) {
    const int q = qt * 16 + (lane & 15);
    const bool qok = q < p.Sq;
    // Q (and dO, O) fragments come straight from global, lane (q, g) holds d = kc*32 + g*8 .. +7 -- fetched one tile ahead
    bf16x8 qh[KC], ql[KC], doh[KC], dol[KC];
    float delta = 0.f;
    const float lse0 = nlse;
  #pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      const bool dpad = kc * 32 + g * 8 >= p.dh;          // (head dimension padded to the fragment width: zeros)
      R8<T> x = nq[kc].get();
      if (dpad) x.zero();
      qh[kc] = x.hi(); ql[kc] = x.lo();
      if (MODE) {
        R8<T> y = nd[kc].get(), z = no[kc].get();
        if (dpad) { y.zero(); z.zero(); }
        doh[kc] = y.hi(); dol[kc] = y.lo();
  #pragma unroll
        for (int e = 0; e < 8; ++e) delta += y.v[e] * z.v[e];
      }
    }
    if (MODE) { delta += __shfl_xor(delta, 16); delta += __shfl_xor(delta, 32); }
    fetch(qt + QW);

    if constexpr (MODE == 1) {
      // ---- dQ, streamed over pairs of 16-key tiles: the probabilities are recomputed from the saved log-sum-exp, so
      // nothing needs all Sk scores at once.  (Holding them -- as the forward must for its max/sum -- cost 256+ VGPRs:
      // one wave per SIMD, 159 us for the encoder shape; two live tiles fit 3-4 waves per SIMD.) ----
      const float c2q = p.scale * 1.4426950408889634f;
      const float lse1 = lse0 > -INFINITY ? -lse0 * 1.4426950408889634f : 0.f;   // exp(s*scale - lse) = exp2(s*c2 + lse1); a row with no live key: P = 0
      const uint32_t rs1 = p.dthresh ? attn_row_seed(p.seed, ((uint64_t)b * p.H + h) * p.Sq + q) : 0u;
      const int ts = attn_ts(p.dthresh);
      f32x4 dq[DT];
  #pragma unroll
      for (int dt = 0; dt < DT; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  #pragma unroll 2
      for (int kb = 0; kb < NT / 2; ++kb) {
        if (kb * 2 >= ntr) break;
        f32x4 sj[2];
  #pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int j = 2 * kb + t;
          f32x4 sc = *reinterpret_cast<const f32x4*>(kbias + j * 16 + g * 4), dp = f32x4{0.f, 0.f, 0.f, 0.f};
          {                                            // (ntr is a multiple of 4: a pair is never half empty)
  #pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
              const int off = (j * 16 + (lane & 15)) * KP + kc * 32 + g * 8;
              const bf16x8 kh = *reinterpret_cast<const bf16x8*>(Kh + off);
              const bf16x8 vh = *reinterpret_cast<const bf16x8*>(Vh + off);
              sc = mfma16(kh, qh[kc], sc);
              dp = mfma16(vh, doh[kc], dp);
              if (PRECISE) {
                const bf16x8 kl = *reinterpret_cast<const bf16x8*>(Kl + off);
                const bf16x8 vl = *reinterpret_cast<const bf16x8*>(Vl + off);
                sc = mfma16(kl, qh[kc], sc);
                sc = mfma16(kh, ql[kc], sc);
                dp = mfma16(vl, doh[kc], dp);
                dp = mfma16(vh, dol[kc], dp);
              }
            }
          }
          uint32_t w01 = 0u, w23 = 0u;
          if (p.dthresh) {
            const uint32_t pb = rs1 + (uint32_t)(j * 8 + g * 2) * ATTN_PAIR_STEP;
            w01 = attn_pair_bits(pb); w23 = attn_pair_bits(pb + ATTN_PAIR_STEP);
          }
  #pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int key = j * 16 + g * 4 + i;
            float pr = __builtin_amdgcn_exp2f(fmaf(sc[i], c2q, lse1));         // normalised P (0 for dead keys: sc = -inf)
            if (MASKED && p.causal) pr = key > q ? 0.f : pr;
            float d = dp[i];
            if (p.dthresh) {
              const uint32_t w = i < 2 ? w01 : w23;
              d = ((i & 1) ? attn_keep_hi(w, ts) : attn_keep_lo(w, ts)) ? d * p.dscale : 0.f;
            }
            sj[t][i] = pr * (d - delta) * p.scale;
          }
        }
        bf16x8 ph, pl;
  #pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float a = sj[0][i], c = sj[1][i];
          ph[i] = (bf16)a; ph[4 + i] = (bf16)c;
          pl[i] = (bf16)(a - (float)ph[i]); pl[4 + i] = (bf16)(c - (float)ph[4 + i]);
        }
  #pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const int off = (dt * 16 + (lane & 15)) * vtp + kb * 32 + g * 4;
          const bf16x8 xh = ld_pair64(Th + off, Th + off + 16);
          dq[dt] = mfma16(xh, ph, dq[dt]);
          if (PRECISE) {
            const bf16x8 xl = ld_pair64(Tl + off, Tl + off + 16);
            dq[dt] = mfma16(xl, ph, dq[dt]);
            dq[dt] = mfma16(xh, pl, dq[dt]);
          }
        }
      }
      if (!qok) continue;
      T* outp1 = reinterpret_cast<T*>(p.dq) + b * p.q_bs + (int64_t)q * p.q_rs + h * p.dh;
  #pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const int d = dt * 16 + g * 4;
        if (sizeof(T) == 4) {
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(outp1) + d) = make_float4(dq[dt][0], dq[dt][1], dq[dt][2], dq[dt][3]);
        } else {
          bf16x4 o4;
  #pragma unroll
          for (int i = 0; i < 4; ++i) o4[i] = (bf16)dq[dt][i];
          *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(outp1) + d) = o4;
        }
      }
      continue;
    }

    // ---- forward: scores S^T[key][q] for all key tiles, softmax over the registers, out^T = V^T P^T ----
    // The core is VALU-bound at dh = 32 (2 MFMAs per 256 scores against every per-score VALU instruction): what is spent per
    // score is one max, one fma + v_exp (scale and max folded into the fma, base-2 exponent), one add, the dropout select and
    // the bf16 pack.  Masks cost nothing when there are none (no key-padding mask, not causal): only the tail tiles test keys.
    // Key tiles are handled in groups of four (skp is a multiple of 64, zero padded): ONE uniform test per group.  (A test per
    // tile -- round 2's first build -- made every QK MFMA its own basic block behind a full s_waitcnt, and the twenty 64-bit
    // conditions were spilled to VGPR lanes: 242 v_readlane per query tile.)
    f32x4 s[NT];
  #pragma unroll
    for (int j0 = 0; j0 < NT; j0 += 4) {
      if (j0 < ntr) {
  #pragma unroll
        for (int j = j0; j < j0 + 4; ++j) {
          s[j] = *reinterpret_cast<const f32x4*>(kbias + j * 16 + g * 4);
  #pragma unroll
          for (int kc = 0; kc < KC; ++kc) {
            const int off = (j * 16 + (lane & 15)) * KP + kc * 32 + g * 8;
            bf16x8 kh = *reinterpret_cast<const bf16x8*>(Kh + off);
            s[j] = mfma16(kh, qh[kc], s[j]);
            if (PRECISE) {
              bf16x8 kl = *reinterpret_cast<const bf16x8*>(Kl + off);
              s[j] = mfma16(kl, qh[kc], s[j]);
              s[j] = mfma16(kh, ql[kc], s[j]);
            }
          }
        }
      } else {
  #pragma unroll
        for (int j = j0; j < j0 + 4; ++j) s[j] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      }
    }
    if constexpr (MASKED) {
      if (p.causal) {
  #pragma unroll
        for (int j = 0; j < NT; ++j)
  #pragma unroll
          for (int i = 0; i < 4; ++i) s[j][i] = (j * 16 + g * 4 + i > q) ? -INFINITY : s[j][i];
      }
    }
    float mx = -INFINITY;
  #pragma unroll
    for (int j = 0; j < NT; ++j)
  #pragma unroll
      for (int i = 0; i < 4; ++i) mx = fmaxf(mx, s[j][i]);
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float c2 = p.scale * 1.4426950408889634f;           // exp(scale * (s - mx)) = exp2(s * c2 - mx * c2)
    const float nm = mx == -INFINITY ? 0.f : -mx * c2;
    float lsum = 0.f;
  #pragma unroll
    for (int j0 = 0; j0 < NT; j0 += 4) {
      if (j0 < ntr) {
  #pragma unroll
        for (int j = j0; j < j0 + 4; ++j)
  #pragma unroll
          for (int i = 0; i < 4; ++i) { const float e = __builtin_amdgcn_exp2f(fmaf(s[j][i], c2, nm)); s[j][i] = e; lsum += e; }
      }
    }
    lsum += __shfl_xor(lsum, 16);
    lsum += __shfl_xor(lsum, 32);
    if (g == 0 && qok && p.lse) p.lse[((int64_t)b * p.H + h) * p.Sq + q] = (mx == -INFINITY ? 0.f : mx * p.scale) + logf(lsum);

    // dropout: the 1/(1-p) of the kept probabilities is applied to the output row; the drops themselves are and-ed out of the
    // packed bf16 pairs below (fp32 path: selected on the floats here, its low-order split needs the masked value)
    const uint32_t gb = p.dthresh ? attn_row_seed(p.seed, ((uint64_t)b * p.H + h) * p.Sq + q) + (uint32_t)(g * 2) * ATTN_PAIR_STEP : 0u;
    const int ts = attn_ts(p.dthresh);
    const uint32_t ts2 = ((uint32_t)ts & 0xffffu) * 0x10001u;
    if (PRECISE && p.dthresh) {
  #pragma unroll
      for (int j0 = 0; j0 < NT; j0 += 4) {
        if (j0 < ntr) {
  #pragma unroll
          for (int j = j0; j < j0 + 4; ++j) {
            const uint32_t w0 = attn_pair_bits(gb + (uint32_t)(j * 8) * ATTN_PAIR_STEP);
            const uint32_t w1 = attn_pair_bits(gb + (uint32_t)(j * 8 + 1) * ATTN_PAIR_STEP);
            s[j][0] = attn_keep_lo(w0, ts) ? s[j][0] : 0.f;
            s[j][1] = attn_keep_hi(w0, ts) ? s[j][1] : 0.f;
            s[j][2] = attn_keep_lo(w1, ts) ? s[j][2] : 0.f;
            s[j][3] = attn_keep_hi(w1, ts) ? s[j][3] : 0.f;
          }
        }
      }
    }

    // ---- out^T[d][q] = sum_key V^T[d][key] * P^T[key][q] ----
    // (scheduling fences: left alone, hipcc hoists the V^T fragment reads of all ten key blocks above the softmax -- 80 more live
    //  registers on top of the 80 scores: 256 VGPRs and 106 spilled to scratch in round 1's build of this kernel)
    f32x4 oacc[DT];
  #pragma unroll
    for (int dt = 0; dt < DT; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_sched_barrier(0);
  #pragma unroll
    for (int j0 = 0; j0 < NT; j0 += 4) {
      __builtin_amdgcn_sched_barrier(0);
      if (j0 < ntr) {
  #pragma unroll
        for (int kb = j0 / 2; kb < j0 / 2 + 2; ++kb) {
          bf16x8 ph, pl;
  #pragma unroll
          for (int i = 0; i < 4; ++i) {
            float a = s[2 * kb][i], c = s[2 * kb + 1][i];
            ph[i] = (bf16)a; ph[4 + i] = (bf16)c;
            if (PRECISE) { pl[i] = (bf16)(a - (float)ph[i]); pl[4 + i] = (bf16)(c - (float)ph[4 + i]); }
          }
          if (!PRECISE && p.dthresh) {
            u32x4 pw = __builtin_bit_cast(u32x4, ph);       // words: keys (0,1), (2,3) of tile 2kb, then of tile 2kb+1
  #pragma unroll
            for (int t = 0; t < 2; ++t) {
              const uint32_t pb = gb + (uint32_t)((2 * kb + t) * 8) * ATTN_PAIR_STEP;
              pw[2 * t] &= ~attn_drop_bits(attn_pair_bits(pb), ts2);
              pw[2 * t + 1] &= ~attn_drop_bits(attn_pair_bits(pb + ATTN_PAIR_STEP), ts2);
            }
            ph = __builtin_bit_cast(bf16x8, pw);
          }
  #pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            const int off = (dt * 16 + (lane & 15)) * vtp + kb * 32 + g * 4;
            bf16x8 xh = ld_pair64(Th + off, Th + off + 16);
            oacc[dt] = mfma16(xh, ph, oacc[dt]);
            if (PRECISE) {
              bf16x8 xl = ld_pair64(Tl + off, Tl + off + 16);
              oacc[dt] = mfma16(xl, ph, oacc[dt]);
              oacc[dt] = mfma16(xh, pl, oacc[dt]);
            }
          }
        }
      }
    }
    if (!qok) continue;
    T* outp = reinterpret_cast<T*>(p.o) + b * p.o_bs + (int64_t)q * p.o_rs + h * p.dh;
    const float inv = lsum > 0.f ? (p.dthresh ? p.dscale : 1.f) / lsum : 0.f;
  #pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const int d = dt * 16 + g * 4;
      if (sizeof(T) == 4) {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(outp) + d) =
            make_float4(oacc[dt][0] * inv, oacc[dt][1] * inv, oacc[dt][2] * inv, oacc[dt][3] * inv);
      } else {
        bf16x4 o4;
  #pragma unroll
        for (int i = 0; i < 4; ++i) o4[i] = (bf16)(oacc[dt][i] * inv);
        *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(outp) + d) = o4;
      }
    }
  }
}

// ================================================================================================================================
// Round 5: self-attention with the q | k | v PROJECTIONS inside the launch (ref: exp/gpv/models/transformer.py:148-155 -- q = k =
// src + pos, value = src, nn.MultiheadAttention's in_proj; transformer.py:216-219 for the decoder's self-attention).
//
//   q = xp Wq^T + bq, k = xp Wk^T + bk, v = x Wv^T + bv      (xp = x + pos: the second output of the LayerNorm that produced x)
//   o = softmax(q k^T scale + key bias) v                      per (batch, head), dh = 32, model width 256, bf16
//
// The three launches it replaces (q | k GEMM 9600 x 512 x 256: 12.5 us, v GEMM 9600 x 256 x 256: 9.8 us, core 17 us inside the step's
// graph) are latency-shaped: 1.9 GFLOP of projections under 22 us of launch ramps and a 15 MB round trip through HBM.  Here the
// workgroup of one (batch, head) first multiplies its rows by ITS 96 rows of in_proj_weight (staged in LDS once, 50 KB):
//   * Q^T / K^T tiles = mfma(A = W rows (d), B = xp rows (token)): a lane ends up with d = 4 g + i and 16 + 4 g + i of ONE token -- packed,
//     that IS the lane's operand fragment of S^T = K Q^T in the k-slot order {4 g + i} U {16 + 4 g + i} (any order works as long as both
//     operands use it): Q never leaves the registers of the wave that owns the query tile, K goes to LDS as one 16-byte store;
//   * V tiles = mfma(A = x rows (token), B = W rows (d)): a lane holds 4 consecutive tokens of one d = 8 bytes of the V^T image;
//   * q, k, v are also written to HBM in the layout of the projection GEMMs' outputs (the backward -- attn_bwd1_kernel, the
//     backward-data and weight-gradient GEMMs -- is unchanged and needs them);
// then the core of attn_q_kernel runs on the staged K / V^T (same softmax, same dropout words, same lse).
// MFMA work per launch: 6.7 GFLOP (projections 3.8 + core 2.9) instead of the core's 2.9 on the same VALU work.
struct QkvK { const bf16* xp; const bf16* x; int64_t x_bs, x_rs; const bf16* w; const float* bias; };

template <int NT, bool FULL>
__global__ __launch_bounds__(QTHR) void attn_qkv_kernel(AttnK p, QkvK xk) {
  if (p.dthresh) p.seed = eff_seed(p.seed, p.seed_dev);
  constexpr int DH = 32, KP = DH + 8, KD = 256, WROW = KD * 2, KCX = KD / 32, DT = 2, NR = (NT + QW - 1) / QW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16* sm = reinterpret_cast<bf16*>(smem_raw);
  const int skp = FULL ? NT * 16 : p.skp, ntr = FULL ? NT : skp / 16, vtp = skp + 8;
  bf16* Kh = sm;                                          // K[skp][KP], d in k-slot order
  bf16* Th = Kh + skp * KP;                               // V^T[DH][vtp]
  float* kbias = reinterpret_cast<float*>(Th + DH * vtp);
  bf16* Wl = reinterpret_cast<bf16*>(kbias + skp);        // rows 0..31 Wq_h, 32..63 Wk_h, 64..95 Wv_h: [96][256] bf16, 16-byte slots XOR-swizzled by (row & 15)
  float* bl = reinterpret_cast<float*>(Wl + 96 * KD);     // the 96 biases
  int b, h;
  {
    const int total = (int)gridDim.x;
    const int qd = total >> 3, r = total & 7, xcd = (int)blockIdx.x & 7, loc = (int)blockIdx.x >> 3;
    const int v = (xcd < r ? xcd * (qd + 1) : r * (qd + 1) + (xcd - r) * qd) + loc;     // contiguous (batch, head) range per XCD: the heads of an image share x
    h = v % p.H; b = v / p.H;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int nqt = (p.Sq + 15) >> 4;
  const bf16* xpg = xk.xp + b * xk.x_bs;
  const bf16* xg = xk.x + b * xk.x_bs;

  bf16x8 nxp[KCX], nx[KCX];
  // (predicated loads, zero rows beyond the sequence: the unconditional clamped form of attn_q_kernel's fetch measured 1.5 us SLOWER here,
  //  28.0 -> 29.5 us, three boxes each)
  auto fetch = [&](int t) {
    const int tok = t * 16 + li;
    const bool ok = t < nqt && tok < p.Sq;
#pragma unroll
    for (int kc = 0; kc < KCX; ++kc) {
      if (ok) {
        nxp[kc] = *reinterpret_cast<const bf16x8*>(xpg + (int64_t)tok * xk.x_rs + kc * 32 + g * 8);
        nx[kc] = *reinterpret_cast<const bf16x8*>(xg + (int64_t)tok * xk.x_rs + kc * 32 + g * 8);
      } else {
        nxp[kc] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        nx[kc] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      }
    }
  };
  fetch(wave);
  // this head's 96 weight rows go L2 -> LDS by DMA, 512-byte rows without padding, the bank swizzle made on the source side (slot s of row L
  // <- chunk s ^ (L & 15): linear_ln.hip / conv1x1_stream.hip); 48 instructions of 1 KB, six per wave, all in flight with the first rows of x
  {
    typedef __attribute__((address_space(3))) void lds_void_t;
    constexpr int OOB = 0x7ffffff0;
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(xk.w), (short)0, OOB, 0x00020000);
    const int wv = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int inst = wv * 6 + i;
      const int L = inst * 2 + (lane >> 5), sl = lane & 31;
      const int voff = (((L >> 5) * KD + h * DH + (L & 31)) * KD + ((sl ^ (L & 15)) * 8)) * 2;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_void_t*)(reinterpret_cast<unsigned char*>(Wl) + inst * 1024), 16, voff, 0, 0, 0);
    }
  }
  if (tid < 96) bl[tid] = xk.bias ? xk.bias[(tid >> 5) * KD + h * DH + (tid & 31)] : 0.f;
  for (int idx = tid; idx < skp; idx += QTHR)
    kbias[idx] = (idx >= p.Sk || (p.kpm && p.kpm[(int64_t)b * p.Sk + idx])) ? -INFINITY : 0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (this wave's DMA pieces have landed; the barrier orders everyone's for the fragment reads)
  __syncthreads();

  // ---- projections: this wave's token tiles (tiles beyond the sequence are multiplied too -- zero rows + bias: the padded K rows and
  // V^T columns must hold finite values, their probabilities are exactly 0) ----
  const int wtx = (g ^ li) * 16;                          // chunk 4 kc + g of weight row (16 n + li) sits in slot (4 kc + g) ^ li = 4 kc ^ (g ^ li)
  bf16x8 qreg[NR];
  bf16* qg = reinterpret_cast<bf16*>(const_cast<void*>(p.q)) + b * p.q_bs + h * DH;
  bf16* kg = reinterpret_cast<bf16*>(const_cast<void*>(p.k)) + b * p.k_bs + h * DH;
  bf16* vg = reinterpret_cast<bf16*>(const_cast<void*>(p.v)) + b * p.v_bs + h * DH;
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int t = wave + r * QW;
    qreg[r] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    if (t < ntr) {
      bf16x8 xpf[KCX], xf[KCX];
#pragma unroll
      for (int kc = 0; kc < KCX; ++kc) { xpf[kc] = nxp[kc]; xf[kc] = nx[kc]; }
      fetch(t + QW);
      f32x4 aq[DT], ak[DT], av[DT];
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const f32x4 bq = *reinterpret_cast<const f32x4*>(bl + dt * 16 + g * 4);
        const f32x4 bk = *reinterpret_cast<const f32x4*>(bl + 32 + dt * 16 + g * 4);
        const float bv = bl[64 + dt * 16 + li];
        aq[dt] = bq; ak[dt] = bk; av[dt] = f32x4{bv, bv, bv, bv};
      }
#pragma unroll
      for (int kc = 0; kc < KCX; ++kc) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const unsigned char* wr = reinterpret_cast<const unsigned char*>(Wl) + (dt * 16 + li) * WROW + ((kc * 64) ^ wtx);
          const bf16x8 wq = *reinterpret_cast<const bf16x8*>(wr);
          const bf16x8 wk = *reinterpret_cast<const bf16x8*>(wr + 32 * WROW);
          const bf16x8 wv = *reinterpret_cast<const bf16x8*>(wr + 64 * WROW);
          aq[dt] = mfma16(wq, xpf[kc], aq[dt]);
          ak[dt] = mfma16(wk, xpf[kc], ak[dt]);
          av[dt] = mfma16(xf[kc], wv, av[dt]);
        }
      }
      bf16x8 qf, kf;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        qf[i] = (bf16)aq[0][i]; qf[4 + i] = (bf16)aq[1][i];
        kf[i] = (bf16)ak[0][i]; kf[4 + i] = (bf16)ak[1][i];
      }
      qreg[r] = qf;
      const int tok = t * 16 + li;
      *reinterpret_cast<bf16x8*>(Kh + tok * KP + g * 8) = kf;
      bf16x4 v4[DT];
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v4[dt][i] = (bf16)av[dt][i];
        *reinterpret_cast<bf16x4*>(Th + (dt * 16 + li) * vtp + t * 16 + g * 4) = v4[dt];
      }
      if (tok < p.Sq) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          bf16x4 q4, k4;
#pragma unroll
          for (int i = 0; i < 4; ++i) { q4[i] = qf[dt * 4 + i]; k4[i] = kf[dt * 4 + i]; }
          *reinterpret_cast<bf16x4*>(qg + (int64_t)tok * p.q_rs + dt * 16 + g * 4) = q4;
          *reinterpret_cast<bf16x4*>(kg + (int64_t)tok * p.k_rs + dt * 16 + g * 4) = k4;
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int tv = t * 16 + g * 4 + i;
        if (tv < p.Sq) {
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) vg[(int64_t)tv * p.v_rs + dt * 16 + li] = v4[dt][i];
        }
      }
    }
  }
  __syncthreads();

  // ---- core (attn_q_kernel MODE 0, bf16) with the Q fragments from registers ----
  const float c2 = p.scale * 1.4426950408889634f;
  const int ts = attn_ts(p.dthresh);
  const uint32_t ts2 = ((uint32_t)ts & 0xffffu) * 0x10001u;
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int qt = wave + r * QW;
    if (qt >= nqt) continue;
    const int q = qt * 16 + li;
    const bool qok = q < p.Sq;
    const bf16x8 qh = qreg[r];
    f32x4 s[NT];
#pragma unroll
    for (int j0 = 0; j0 < NT; j0 += 4) {
      if (j0 < ntr) {
#pragma unroll
        for (int j = j0; j < j0 + 4; ++j) {
          s[j] = *reinterpret_cast<const f32x4*>(kbias + j * 16 + g * 4);
          const bf16x8 kh = *reinterpret_cast<const bf16x8*>(Kh + (j * 16 + li) * KP + g * 8);
          s[j] = mfma16(kh, qh, s[j]);
        }
      } else {
#pragma unroll
        for (int j = j0; j < j0 + 4; ++j) s[j] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) mx = fmaxf(mx, s[j][i]);
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float nm = mx == -INFINITY ? 0.f : -mx * c2;
    float lsum = 0.f;
#pragma unroll
    for (int j0 = 0; j0 < NT; j0 += 4) {
      if (j0 < ntr) {
#pragma unroll
        for (int j = j0; j < j0 + 4; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i) { const float e = __builtin_amdgcn_exp2f(fmaf(s[j][i], c2, nm)); s[j][i] = e; lsum += e; }
      }
    }
    lsum += __shfl_xor(lsum, 16);
    lsum += __shfl_xor(lsum, 32);
    if (g == 0 && qok && p.lse) p.lse[((int64_t)b * p.H + h) * p.Sq + q] = (mx == -INFINITY ? 0.f : mx * p.scale) + logf(lsum);
    const uint32_t gb = p.dthresh ? attn_row_seed(p.seed, ((uint64_t)b * p.H + h) * p.Sq + q) + (uint32_t)(g * 2) * ATTN_PAIR_STEP : 0u;
    f32x4 oacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j0 = 0; j0 < NT; j0 += 4) {
      __builtin_amdgcn_sched_barrier(0);
      if (j0 < ntr) {
#pragma unroll
        for (int kb = j0 / 2; kb < j0 / 2 + 2; ++kb) {
          bf16x8 ph;
#pragma unroll
          for (int i = 0; i < 4; ++i) { ph[i] = (bf16)s[2 * kb][i]; ph[4 + i] = (bf16)s[2 * kb + 1][i]; }
          if (p.dthresh) {
            u32x4 pw = __builtin_bit_cast(u32x4, ph);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              const uint32_t pb = gb + (uint32_t)((2 * kb + t) * 8) * ATTN_PAIR_STEP;
              pw[2 * t] &= ~attn_drop_bits(attn_pair_bits(pb), ts2);
              pw[2 * t + 1] &= ~attn_drop_bits(attn_pair_bits(pb + ATTN_PAIR_STEP), ts2);
            }
            ph = __builtin_bit_cast(bf16x8, pw);
          }
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            const int off = (dt * 16 + li) * vtp + kb * 32 + g * 4;
            const bf16x8 xh = ld_pair64(Th + off, Th + off + 16);
            oacc[dt] = mfma16(xh, ph, oacc[dt]);
          }
        }
      }
    }
    if (!qok) continue;
    bf16* outp = reinterpret_cast<bf16*>(p.o) + b * p.o_bs + (int64_t)q * p.o_rs + h * DH;
    const float inv = lsum > 0.f ? (p.dthresh ? p.dscale : 1.f) / lsum : 0.f;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      bf16x4 o4;
#pragma unroll
      for (int i = 0; i < 4; ++i) o4[i] = (bf16)(oacc[dt][i] * inv);
      *reinterpret_cast<bf16x4*>(outp + dt * 16 + g * 4) = o4;
    }
  }
}

// dK / dV: workgroup = 64 keys of one (b,h); wave = 16 keys; queries streamed in chunks of 64.
template <typename T, int DHK, int DHV>
__global__ __launch_bounds__(256) void attn_kv_kernel(AttnK p) {
  if (p.dthresh) p.seed = eff_seed(p.seed, p.seed_dev);
  constexpr bool PRECISE = sizeof(T) == 4;
  constexpr int KP = DHK + 8, KC = DHK / 32, DT = DHV / 16, QC = 64, QTP = QC + 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16* sm = reinterpret_cast<bf16*>(smem_raw);
  bf16* Qh = sm;                 bf16* Ql = Qh + (PRECISE ? QC * KP : 0);
  bf16* Dh = Ql + QC * KP;       bf16* Dl = Dh + (PRECISE ? QC * KP : 0);
  bf16* QTh = Dl + QC * KP;      bf16* QTl = QTh + (PRECISE ? DHV * QTP : 0);
  bf16* DTh = QTl + DHV * QTP;   bf16* DTl = DTh + (PRECISE ? DHV * QTP : 0);
  float* lse_s = reinterpret_cast<float*>(DTl + DHV * QTP);
  float* del_s = lse_s + QC;
  uint32_t* rs_s = reinterpret_cast<uint32_t*>(del_s + QC);       // dropout row seeds of the staged queries

  const int b = blockIdx.z, h = blockIdx.y, k0 = blockIdx.x * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
  const int key = k0 + wave * 16 + (lane & 15);
  const bool kok = key < p.Sk;
  bool kdead = !kok;
  if (kok && p.kpm) kdead = p.kpm[(int64_t)b * p.Sk + key] != 0;
  const float c2k = p.scale * 1.4426950408889634f;
  const uint32_t kpair = (uint32_t)(key >> 1) * ATTN_PAIR_STEP, kshift = (key & 1) * 16;
  const int ts = attn_ts(p.dthresh);

  bf16x8 kh[KC], kl[KC], vh[KC], vl[KC];
  {
    const T* kp_ = reinterpret_cast<const T*>(p.k) + b * p.k_bs + (int64_t)key * p.k_rs + h * p.dh;
    const T* vp_ = reinterpret_cast<const T*>(p.v) + b * p.v_bs + (int64_t)key * p.v_rs + h * p.dh;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      const int d0 = kc * 32 + g * 8;
      R8<T> x, y;
      if (kok && d0 < p.dh) { x.load(kp_ + d0); y.load(vp_ + d0); } else { x.zero(); y.zero(); }
      kh[kc] = x.hi(); kl[kc] = x.lo(); vh[kc] = y.hi(); vl[kc] = y.lo();
    }
  }
  f32x4 dkacc[DT], dvacc[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) { dkacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dvacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  const T* qg = reinterpret_cast<const T*>(p.q) + b * p.q_bs + h * p.dh;
  const T* dog = reinterpret_cast<const T*>(p.dout) + b * p.do_bs + h * p.dh;
  const T* og = reinterpret_cast<const T*>(p.o) + b * p.o_bs + h * p.dh;

  for (int qc0 = 0; qc0 < p.Sq; qc0 += QC) {
    const int nq = min(QC, p.Sq - qc0);
    __syncthreads();
    stage_rows<T, PRECISE, DHK>(qg + (int64_t)qc0 * p.q_rs, p.q_rs, nq, QC, p.dh, Qh, Ql);
    stage_rows<T, PRECISE, DHK>(dog + (int64_t)qc0 * p.do_rs, p.do_rs, nq, QC, p.dh, Dh, Dl);
    stage_cols<T, PRECISE, DHV>(qg + (int64_t)qc0 * p.q_rs, p.q_rs, nq, QC, QTP, QTh, QTl);
    stage_cols<T, PRECISE, DHV>(dog + (int64_t)qc0 * p.do_rs, p.do_rs, nq, QC, QTP, DTh, DTl);
    if (threadIdx.x < QC) {
      const int r = threadIdx.x;
      float dl = 0.f, ls = INFINITY;
      if (r < nq) {
        const T* a = dog + (int64_t)(qc0 + r) * p.do_rs;
        const T* c = og + (int64_t)(qc0 + r) * p.o_rs;
        for (int d = 0; d < p.dh; d += 8) {
          float u[8], w[8];
          Ld8<T>::ld(a + d, u); Ld8<T>::ld(c + d, w);
#pragma unroll
          for (int e = 0; e < 8; ++e) dl += u[e] * w[e];
        }
        ls = p.lse[((int64_t)b * p.H + h) * p.Sq + qc0 + r];
      }
      lse_s[r] = -ls * 1.4426950408889634f; del_s[r] = dl;
      rs_s[r] = p.dthresh ? attn_row_seed(p.seed, ((uint64_t)b * p.H + h) * p.Sq + qc0 + r) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int qb = 0; qb < QC / 32; ++qb) {
      if (qb * 32 >= nq) break;
      f32x4 pr[2], ds[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f32x4 sa = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
          const int off = (qb * 32 + t * 16 + (lane & 15)) * KP + kc * 32 + g * 8;
          bf16x8 qh_ = *reinterpret_cast<const bf16x8*>(Qh + off);
          bf16x8 dh_ = *reinterpret_cast<const bf16x8*>(Dh + off);
          sa = mfma16(qh_, kh[kc], sa);
          dp = mfma16(dh_, vh[kc], dp);
          if (PRECISE) {
            // SAME accumulation order as the forward / dQ kernels (hi*hi, Klo*Qhi, Khi*Qlo): the recomputed
            // scores must be bit-identical to the forward's, otherwise exp(s - lse) drifts from the forward
            // probabilities when |s| is large (ulp(s) ~ 0.1 at |s| ~ 1e6) and the (dP - delta) cancellation breaks.
            bf16x8 ql_ = *reinterpret_cast<const bf16x8*>(Ql + off);
            bf16x8 dl_ = *reinterpret_cast<const bf16x8*>(Dl + off);
            sa = mfma16(qh_, kl[kc], sa); sa = mfma16(ql_, kh[kc], sa);
            dp = mfma16(dh_, vl[kc], dp); dp = mfma16(dl_, vh[kc], dp);
          }
        }
        const int qr = qb * 32 + t * 16 + g * 4;   // local query row of element i = qr + i
        const float4 l4 = *reinterpret_cast<const float4*>(lse_s + qr);
        const float4 d4 = *reinterpret_cast<const float4*>(del_s + qr);
        const float ls[4] = {l4.x, l4.y, l4.z, l4.w};             // -lse * log2(e); -inf for padded queries -> P = 0
        const float dl[4] = {d4.x, d4.y, d4.z, d4.w};
        uint32_t rsd[4] = {0u, 0u, 0u, 0u};
        if (p.dthresh) {
          const uint4 r4 = *reinterpret_cast<const uint4*>(rs_s + qr);
          rsd[0] = r4.x; rsd[1] = r4.y; rsd[2] = r4.z; rsd[3] = r4.w;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int qq = qc0 + qr + i;
          const bool dead = kdead || (p.causal && key > qq);
          float pv = __builtin_amdgcn_exp2f(fmaf(sa[i], c2k, ls[i]));
          pv = dead ? 0.f : pv;
          float d = dp[i], pd = pv;
          if (p.dthresh) {
            const uint32_t w = attn_pair_bits(rsd[i] + kpair);
            const bool keep = (int)(short)((w >> kshift) & 0xffffu) >= ts;
            d = keep ? d * p.dscale : 0.f;
            pd = keep ? pv * p.dscale : 0.f;
          }
          pr[t][i] = pd;
          ds[t][i] = pv * (d - dl[i]) * p.scale;
        }
      }
      bf16x8 ph, pl, sh, sl;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ph[i] = (bf16)pr[0][i]; ph[4 + i] = (bf16)pr[1][i];
        pl[i] = (bf16)(pr[0][i] - (float)ph[i]); pl[4 + i] = (bf16)(pr[1][i] - (float)ph[4 + i]);
        sh[i] = (bf16)ds[0][i]; sh[4 + i] = (bf16)ds[1][i];
        sl[i] = (bf16)(ds[0][i] - (float)sh[i]); sl[4 + i] = (bf16)(ds[1][i] - (float)sh[4 + i]);
      }
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const int off = (dt * 16 + (lane & 15)) * QTP + qb * 32 + g * 4;
        bf16x8 dth = ld_pair64(DTh + off, DTh + off + 16);
        bf16x8 qth = ld_pair64(QTh + off, QTh + off + 16);
        dvacc[dt] = mfma16(dth, ph, dvacc[dt]);
        dkacc[dt] = mfma16(qth, sh, dkacc[dt]);
        if (PRECISE) {
          bf16x8 dtl = ld_pair64(DTl + off, DTl + off + 16);
          bf16x8 qtl = ld_pair64(QTl + off, QTl + off + 16);
          dvacc[dt] = mfma16(dtl, ph, dvacc[dt]); dvacc[dt] = mfma16(dth, pl, dvacc[dt]);
          dkacc[dt] = mfma16(qtl, sh, dkacc[dt]); dkacc[dt] = mfma16(qth, sl, dkacc[dt]);
        }
      }
    }
  }
  if (!kok) return;
  T* dkp = reinterpret_cast<T*>(p.dk) + b * p.k_bs + (int64_t)key * p.k_rs + h * p.dh;
  T* dvp = reinterpret_cast<T*>(p.dv) + b * p.v_bs + (int64_t)key * p.v_rs + h * p.dh;
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) {
    const int d = dt * 16 + g * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) { dkp[d + i] = (T)dkacc[dt][i]; dvp[d + i] = (T)dvacc[dt][i]; }
  }
}

// dK / dV, bf16, all queries of one (batch, head) resident: workgroup = 8 waves on one (batch, head) [x a split of the key tiles],
// Q and dO staged ONCE (row-major for the score MFMAs, transposed for the dK / dV MFMAs) together with every query's
// -lse*log2(e), delta = dO.O and dropout row seed; a wave then owns 16 keys at a time and walks all queries in pairs of tiles.
// (attn_kv_kernel above re-staged each 64-query chunk in every 64-key block behind two barriers: five dependent load ->
//  LDS -> barrier round trips per block, 65 us on the encoder shape with the MFMAs idle 90 % of the time.)
// NQT = query tiles the loops are unrolled for; CAUSAL compiles the per-score query test in.
template <int DHK, int DHV, int NQT, bool CAUSAL>
__global__ __launch_bounds__(QTHR) void attn_kv2_kernel(AttnK p) {
  using T = bf16;
  if (p.dthresh) p.seed = eff_seed(p.seed, p.seed_dev);
  constexpr int KP = DHK + 8, KC = DHK / 32, DT = DHV / 16;
  const int sqp = p.sqp, QTP = sqp + 8;              // sqp: Sq rounded up to 32
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16* Qh = reinterpret_cast<bf16*>(smem_raw);
  bf16* Dh = Qh + sqp * KP;
  bf16* QTh = Dh + sqp * KP;
  bf16* DTh = QTh + DHV * QTP;
  float* lse_s = reinterpret_cast<float*>(DTh + DHV * QTP);
  float* del_s = lse_s + sqp;
  uint32_t* rs_s = reinterpret_cast<uint32_t*>(del_s + sqp);

  int b, h, xs;
  {
    const int nsp = p.nsplit, total = (int)gridDim.x;
    const int qd = total >> 3, r = total & 7, xcd = (int)blockIdx.x & 7, loc = (int)blockIdx.x >> 3;
    const int v = (xcd < r ? xcd * (qd + 1) : r * (qd + 1) + (xcd - r) * qd) + loc;
    xs = v % nsp;
    const int bh = v / nsp;
    h = bh % p.H; b = bh / p.H;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
  const int nkt = (p.Sk + 15) >> 4;
  const int per = (nkt + p.nsplit - 1) / p.nsplit;
  const int kt_end = min(nkt, (xs + 1) * per);
  const int kt_first = xs * per + wave;
  // this wave's K / V rows (MFMA B operands), fetched one key tile ahead
  Raw8<T> nk[KC], nv[KC];
  bool ndead = true;
  auto fetch = [&](int kt_) {
    const int key_ = kt_ * 16 + (lane & 15);
    const bool ok = kt_ < kt_end && key_ < p.Sk;
    const T* kp_ = reinterpret_cast<const T*>(p.k) + b * p.k_bs + (int64_t)key_ * p.k_rs + h * p.dh;
    const T* vp_ = reinterpret_cast<const T*>(p.v) + b * p.v_bs + (int64_t)key_ * p.v_rs + h * p.dh;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      const int d0 = kc * 32 + g * 8;
      if (ok && d0 < p.dh) { nk[kc].load(kp_ + d0); nv[kc].load(vp_ + d0); } else { nk[kc].zero(); nv[kc].zero(); }
    }
    ndead = !ok || (p.kpm && p.kpm[(int64_t)b * p.Sk + key_] != 0);
  };
  fetch(kt_first);

  const T* qg = reinterpret_cast<const T*>(p.q) + b * p.q_bs + h * p.dh;
  const T* dog = reinterpret_cast<const T*>(p.dout) + b * p.do_bs + h * p.dh;
  const T* og = reinterpret_cast<const T*>(p.o) + b * p.o_bs + h * p.dh;
  {
    RowBatch<T, DHK, NQT * 16, QTHR> qr, dr;
    ColBatch<T, DHV, NQT * 16, QTHR> qc, dc;
    qr.load(qg, p.q_rs, p.Sq, sqp, p.dh);
    dr.load(dog, p.do_rs, p.Sq, sqp, p.dh);
    qc.load(qg, p.q_rs, p.Sq, sqp);
    dc.load(dog, p.do_rs, p.Sq, sqp);
    qr.template store<false>(sqp, Qh, nullptr);
    dr.template store<false>(sqp, Dh, nullptr);
    qc.template store<false>(sqp, QTP, QTh, nullptr);
    dc.template store<false>(sqp, QTP, DTh, nullptr);
  }
  for (int r = threadIdx.x; r < sqp; r += QTHR) {
    float dl = 0.f, ls = INFINITY;                   // padded queries: -lse = -inf -> P = 0
    if (r < p.Sq) {
      const T* a = dog + (int64_t)r * p.do_rs;
      const T* c = og + (int64_t)r * p.o_rs;
      for (int d = 0; d < p.dh; d += 8) {
        float u[8], w[8];
        Ld8<T>::ld(a + d, u); Ld8<T>::ld(c + d, w);
#pragma unroll
        for (int e = 0; e < 8; ++e) dl += u[e] * w[e];
      }
      ls = p.lse[((int64_t)b * p.H + h) * p.Sq + r];
    }
    lse_s[r] = -ls * 1.4426950408889634f; del_s[r] = dl;
    rs_s[r] = p.dthresh ? attn_row_seed(p.seed, ((uint64_t)b * p.H + h) * p.Sq + r) : 0u;
  }
  __syncthreads();

  const float c2k = p.scale * 1.4426950408889634f;
  const int ts = attn_ts(p.dthresh);
  const int nqb = sqp >> 5;                          // pairs of query tiles
  for (int kt = kt_first; kt < kt_end; kt += QW) {
    const int key = kt * 16 + (lane & 15);
    const bool kok = key < p.Sk;
    const bool kdead = ndead;
    bf16x8 kh[KC], vh[KC];
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) { kh[kc] = nk[kc].get().hi(); vh[kc] = nv[kc].get().hi(); }
    fetch(kt + QW);
    const uint32_t kpair = (uint32_t)(key >> 1) * ATTN_PAIR_STEP, kshift = (key & 1) * 16;
    f32x4 dkacc[DT], dvacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) { dkacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dvacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int qb = 0; qb < NQT / 2; ++qb) {
      if (qb >= nqb) break;
      f32x4 pr[2], ds[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f32x4 sa = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
          const int off = (qb * 32 + t * 16 + (lane & 15)) * KP + kc * 32 + g * 8;
          sa = mfma16(*reinterpret_cast<const bf16x8*>(Qh + off), kh[kc], sa);
          dp = mfma16(*reinterpret_cast<const bf16x8*>(Dh + off), vh[kc], dp);
        }
        const int qr = qb * 32 + t * 16 + g * 4;     // query row of element i = qr + i
        const float4 l4 = *reinterpret_cast<const float4*>(lse_s + qr);
        const float4 d4 = *reinterpret_cast<const float4*>(del_s + qr);
        const float ls[4] = {l4.x, l4.y, l4.z, l4.w};
        const float dl[4] = {d4.x, d4.y, d4.z, d4.w};
        uint32_t rsd[4] = {0u, 0u, 0u, 0u};
        if (p.dthresh) {
          const uint4 r4 = *reinterpret_cast<const uint4*>(rs_s + qr);
          rsd[0] = r4.x; rsd[1] = r4.y; rsd[2] = r4.z; rsd[3] = r4.w;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float pv = __builtin_amdgcn_exp2f(fmaf(sa[i], c2k, ls[i]));
          bool dead = kdead;
          if (CAUSAL) dead = dead || (p.causal && key > qr + i);
          pv = dead ? 0.f : pv;
          float d = dp[i], pd = pv;
          if (p.dthresh) {
            const uint32_t w = attn_pair_bits(rsd[i] + kpair);
            const bool keep = (int)(short)((w >> kshift) & 0xffffu) >= ts;
            d = keep ? d * p.dscale : 0.f;
            pd = keep ? pv * p.dscale : 0.f;
          }
          pr[t][i] = pd;
          ds[t][i] = pv * (d - dl[i]) * p.scale;
        }
      }
      bf16x8 ph, sh;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ph[i] = (bf16)pr[0][i]; ph[4 + i] = (bf16)pr[1][i];
        sh[i] = (bf16)ds[0][i]; sh[4 + i] = (bf16)ds[1][i];
      }
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const int off = (dt * 16 + (lane & 15)) * QTP + qb * 32 + g * 4;
        dvacc[dt] = mfma16(ld_pair64(DTh + off, DTh + off + 16), ph, dvacc[dt]);
        dkacc[dt] = mfma16(ld_pair64(QTh + off, QTh + off + 16), sh, dkacc[dt]);
      }
    }
    if (!kok) continue;
    T* dkp = reinterpret_cast<T*>(p.dk) + b * p.k_bs + (int64_t)key * p.k_rs + h * p.dh;
    T* dvp = reinterpret_cast<T*>(p.dv) + b * p.v_bs + (int64_t)key * p.v_rs + h * p.dh;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const int d = dt * 16 + g * 4;
      if (d < p.dh) {
        bf16x4 k4, v4;
#pragma unroll
        for (int i = 0; i < 4; ++i) { k4[i] = (bf16)dkacc[dt][i]; v4[i] = (bf16)dvacc[dt][i]; }
        *reinterpret_cast<bf16x4*>(dkp + d) = k4;
        *reinterpret_cast<bf16x4*>(dvp + d) = v4;
      }
    }
  }
}

// The whole backward (dQ, dK, dV) in ONE launch, bf16: attn_kv2_kernel's structure -- one workgroup of 8 waves per (batch, head),
// Q / dO resident in both orientations -- with two changes.  (1) A wave keeps ALL its key tiles (kt = wave, wave + 8, ...: K / V
// fragments and the dK / dV accumulators, KTW of them) in registers and walks the queries in pairs of tiles ONCE: the Q / dO
// fragments of a pair are read from LDS once for all of the wave's key tiles.  (2) The dS tile a wave has just formed (lane = key,
// registers = queries: the B operand of dK += Q^T dS) is ALSO written transposed into a [32 queries][keys] strip in LDS; after
// the pair's barrier the strip is complete over all keys and dQ^T[d][q] = K^T[d][:] dS^T[:][q] of those 32 queries is a K = skp
// product per 16 x 16 output tile, owned by ONE wave (the waves with the fewest key tiles) and stored straight from the
// accumulator: S is formed once (5 products instead of the 7 of the dQ + dK/dV pair of launches), no second exponentiation /
// dropout pass, and no cross-wave sum at all -- the summation order is fixed, the result reproducible.  The strip is double
// buffered: one barrier per pair of query tiles.
// NQT / NKT = query / key tiles the loops are unrolled for (8 or 20); needs nkt <= 8 KTW.
template <int DHK, int DHV, int NQT, int NKT, bool CAUSAL>
__global__ __launch_bounds__(QTHR) void attn_bwd1_kernel(AttnK p) {
  using T = bf16;
  if (p.dthresh) p.seed = eff_seed(p.seed, p.seed_dev);
  constexpr int KP = DHK + 8, KC = DHK / 32, DT = DHV / 16, KTW = (NKT + QW - 1) / QW;
  const int sqp = p.sqp, QTP = sqp + 8, skp = p.skp, vtp = skp + 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16* Qh = reinterpret_cast<bf16*>(smem_raw);
  bf16* Dh = Qh + sqp * KP;
  bf16* QTh = Dh + sqp * KP;
  bf16* DTh = QTh + DHV * QTP;
  bf16* KTh = DTh + DHV * QTP;                       // K^T [DHV][vtp]
  bf16* dSb = KTh + DHV * vtp;                       // two strips dS [32 queries][vtp]
  float* lse_s = reinterpret_cast<float*>(dSb + 64 * vtp);
  float* del_s = lse_s + sqp;
  uint32_t* rs_s = reinterpret_cast<uint32_t*>(del_s + sqp);

  int b, h;
  {
    const int total = (int)gridDim.x;
    const int qd = total >> 3, r = total & 7, xcd = (int)blockIdx.x & 7, loc = (int)blockIdx.x >> 3;
    const int bh = (xcd < r ? xcd * (qd + 1) : r * (qd + 1) + (xcd - r) * qd) + loc;
    h = bh % p.H; b = bh / p.H;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
  const int nkt = (p.Sk + 15) >> 4;
  const T* kg = reinterpret_cast<const T*>(p.k) + b * p.k_bs + h * p.dh;
  const T* vg = reinterpret_cast<const T*>(p.v) + b * p.v_bs + h * p.dh;
  // this wave's K / V rows (MFMA B operands of the score products)
  bf16x8 kh[KTW][KC], vh[KTW][KC];
  bool kdead[KTW];
#pragma unroll
  for (int j = 0; j < KTW; ++j) {
    const int kt_ = wave + j * QW, key_ = kt_ * 16 + (lane & 15);
    const bool ok = kt_ < nkt && key_ < p.Sk;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      const int d0 = kc * 32 + g * 8;
      Raw8<T> a, c;
      if (ok && d0 < p.dh) { a.load(kg + (int64_t)key_ * p.k_rs + d0); c.load(vg + (int64_t)key_ * p.v_rs + d0); } else { a.zero(); c.zero(); }
      kh[j][kc] = a.r; vh[j][kc] = c.r;
    }
    kdead[j] = !ok || (p.kpm && p.kpm[(int64_t)b * p.Sk + key_] != 0);
  }

  const T* qg = reinterpret_cast<const T*>(p.q) + b * p.q_bs + h * p.dh;
  const T* dog = reinterpret_cast<const T*>(p.dout) + b * p.do_bs + h * p.dh;
  const T* og = reinterpret_cast<const T*>(p.o) + b * p.o_bs + h * p.dh;
  {
    RowBatch<T, DHK, NQT * 16, QTHR> qr, dr;
    ColBatch<T, DHV, NQT * 16, QTHR> qc, dc;
    qr.load(qg, p.q_rs, p.Sq, sqp, p.dh);
    dr.load(dog, p.do_rs, p.Sq, sqp, p.dh);
    qc.load(qg, p.q_rs, p.Sq, sqp);
    dc.load(dog, p.do_rs, p.Sq, sqp);
    qr.template store<false>(sqp, Qh, nullptr);
    dr.template store<false>(sqp, Dh, nullptr);
    qc.template store<false>(sqp, QTP, QTh, nullptr);
    dc.template store<false>(sqp, QTP, DTh, nullptr);
  }
  {
    ColBatch<T, DHV, NKT * 16, QTHR> kc_;
    kc_.load(kg, p.k_rs, p.Sk, skp);
    kc_.template store<false>(skp, vtp, KTh, nullptr);
  }
  for (int idx = threadIdx.x; idx < 8 * vtp; idx += QTHR)       // key tiles nobody owns (>= nkt) stay zero in both strips
    reinterpret_cast<bf16x8*>(dSb)[idx] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
  for (int r = threadIdx.x; r < sqp; r += QTHR) {
    float dl = 0.f, ls = INFINITY;                   // padded queries: -lse = -inf -> P = 0
    if (r < p.Sq) {
      const T* a = dog + (int64_t)r * p.do_rs;
      const T* c = og + (int64_t)r * p.o_rs;
      for (int d = 0; d < p.dh; d += 8) {
        float u[8], w[8];
        Ld8<T>::ld(a + d, u); Ld8<T>::ld(c + d, w);
#pragma unroll
        for (int e = 0; e < 8; ++e) dl += u[e] * w[e];
      }
      ls = p.lse[((int64_t)b * p.H + h) * p.Sq + r];
    }
    lse_s[r] = -ls * 1.4426950408889634f; del_s[r] = dl;
    rs_s[r] = p.dthresh ? attn_row_seed(p.seed, ((uint64_t)b * p.H + h) * p.Sq + r) : 0u;
  }
  __syncthreads();

  const float c2k = p.scale * 1.4426950408889634f;
  const int ts = attn_ts(p.dthresh);
  const int nqb = sqp >> 5;                          // pairs of query tiles
  f32x4 dkacc[KTW][DT], dvacc[KTW][DT];
#pragma unroll
  for (int j = 0; j < KTW; ++j)
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) { dkacc[j][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dvacc[j][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  T* dqg = reinterpret_cast<T*>(p.dq) + b * p.q_bs + h * p.dh;

#pragma unroll 1
  for (int qb = 0; qb < nqb; ++qb) {
    bf16* dsw = dSb + (qb & 1) * 32 * vtp;
    // operands of this pair of query tiles, shared by all of the wave's key tiles
    bf16x8 qf[2][KC], df[2][KC], tq[DT], td[DT];
    float ls[2][4], dl[2][4];
    uint32_t rsd[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) {
        const int off = (qb * 32 + t * 16 + (lane & 15)) * KP + kc * 32 + g * 8;
        qf[t][kc] = *reinterpret_cast<const bf16x8*>(Qh + off);
        df[t][kc] = *reinterpret_cast<const bf16x8*>(Dh + off);
      }
      const int qr = qb * 32 + t * 16 + g * 4;       // query row of element i = qr + i
      const float4 l4 = *reinterpret_cast<const float4*>(lse_s + qr);
      const float4 d4 = *reinterpret_cast<const float4*>(del_s + qr);
      ls[t][0] = l4.x; ls[t][1] = l4.y; ls[t][2] = l4.z; ls[t][3] = l4.w;
      dl[t][0] = d4.x; dl[t][1] = d4.y; dl[t][2] = d4.z; dl[t][3] = d4.w;
      rsd[t][0] = rsd[t][1] = rsd[t][2] = rsd[t][3] = 0u;
      if (p.dthresh) {
        const uint4 r4 = *reinterpret_cast<const uint4*>(rs_s + qr);
        rsd[t][0] = r4.x; rsd[t][1] = r4.y; rsd[t][2] = r4.z; rsd[t][3] = r4.w;
      }
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const int off = (dt * 16 + (lane & 15)) * QTP + qb * 32 + g * 4;
      tq[dt] = ld_pair64(QTh + off, QTh + off + 16);
      td[dt] = ld_pair64(DTh + off, DTh + off + 16);
    }
#pragma unroll
    for (int j = 0; j < KTW; ++j) {
      const int kt = wave + j * QW;
      if (kt < nkt) {                                // (wave-uniform)
        const int key = kt * 16 + (lane & 15);
        const uint32_t kpair = (uint32_t)(key >> 1) * ATTN_PAIR_STEP, kshift = (key & 1) * 16;
        bf16x8 ph, sh;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          f32x4 sa = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kc = 0; kc < KC; ++kc) {
            sa = mfma16(qf[t][kc], kh[j][kc], sa);
            dp = mfma16(df[t][kc], vh[j][kc], dp);
          }
          const int qr = qb * 32 + t * 16 + g * 4;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float pv = __builtin_amdgcn_exp2f(fmaf(sa[i], c2k, ls[t][i]));
            bool dead = kdead[j];
            if (CAUSAL) dead = dead || (p.causal && key > qr + i);
            pv = dead ? 0.f : pv;
            float d = dp[i], pd = pv;
            if (p.dthresh) {
              const uint32_t w = attn_pair_bits(rsd[t][i] + kpair);
              const bool keep = (int)(short)((w >> kshift) & 0xffffu) >= ts;
              d = keep ? d * p.dscale : 0.f;
              pd = keep ? pv * p.dscale : 0.f;
            }
            ph[t * 4 + i] = (bf16)pd;
            sh[t * 4 + i] = (bf16)(pv * (d - dl[t][i]) * p.scale);
          }
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          dvacc[j][dt] = mfma16(td[dt], ph, dvacc[j][dt]);
          dkacc[j][dt] = mfma16(tq[dt], sh, dkacc[j][dt]);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int i = 0; i < 4; ++i) dsw[(t * 16 + g * 4 + i) * vtp + key] = sh[t * 4 + i];
      }
    }
    __syncthreads();
    // dQ^T of these 32 queries: output tile o = (query tile t, d tile dt), all keys, one wave each
    for (int o = QW - 1 - wave; o < 2 * DT; o += QW) {
      const int t = o / DT, dt = o - t * DT;
      const bf16* ap = KTh + (dt * 16 + (lane & 15)) * vtp + g * 8;
      const bf16* bp = dsw + (t * 16 + (lane & 15)) * vtp + g * 8;
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < NKT / 2; ++kb) {
        if (kb * 32 >= skp) break;
        acc = mfma16(*reinterpret_cast<const bf16x8*>(ap + kb * 32), *reinterpret_cast<const bf16x8*>(bp + kb * 32), acc);
      }
      const int q = qb * 32 + t * 16 + (lane & 15), d = dt * 16 + g * 4;
      if (q < p.Sq && d < p.dh) {
        bf16x4 o4;
#pragma unroll
        for (int i = 0; i < 4; ++i) o4[i] = (bf16)acc[i];
        *reinterpret_cast<bf16x4*>(dqg + (int64_t)q * p.q_rs + d) = o4;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < KTW; ++j) {
    const int key = (wave + j * QW) * 16 + (lane & 15);
    if (key >= p.Sk) continue;
    T* dkp = reinterpret_cast<T*>(p.dk) + b * p.k_bs + (int64_t)key * p.k_rs + h * p.dh;
    T* dvp = reinterpret_cast<T*>(p.dv) + b * p.v_bs + (int64_t)key * p.v_rs + h * p.dh;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const int d = dt * 16 + g * 4;
      if (d < p.dh) {
        bf16x4 k4, v4;
#pragma unroll
        for (int i = 0; i < 4; ++i) { k4[i] = (bf16)dkacc[j][dt][i]; v4[i] = (bf16)dvacc[j][dt][i]; }
        *reinterpret_cast<bf16x4*>(dkp + d) = k4;
        *reinterpret_cast<bf16x4*>(dvp + d) = v4;
      }
    }
  }
}

template <typename T, int DHK, int DHV, int NT, int MODE, bool MASKED, bool FULL>
int launch_q_f(AttnK p, hipStream_t st) {
  constexpr bool PRECISE = sizeof(T) == 4;
  constexpr int KP = DHK + 8;
  const int vtp = p.skp + 8;
  size_t elems = (size_t)p.skp * KP + (size_t)DHV * vtp + (MODE ? (size_t)p.skp * KP : 0);
  size_t lds = elems * 2 * (PRECISE ? 2 : 1) + (size_t)p.skp * sizeof(float);
  auto fn = attn_q_kernel<T, DHK, DHV, NT, MODE, MASKED, FULL>;
  static size_t attr = 0;
  if (lds > 160 * 1024) return (int)hipErrorInvalidValue;      // more LDS than a CU has: this shape is outside the kernel's range
  if (lds > 64 * 1024 && lds > attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }      // (leave no stale error behind for the next launch check)
    attr = lds;
  }
  // one block of 8 waves per (batch, head) when those fill the chip; fewer query tiles per block (down to one per wave) otherwise
  const int nqt = (p.Sq + 15) / 16;
  const int bh = p.B * p.H;
  int nsplit = (256 + bh - 1) / bh;
  if (nsplit > (nqt + QW - 1) / QW) nsplit = (nqt + QW - 1) / QW;
  if (nsplit < 1) nsplit = 1;
  {
    static const int force = tune_env("GPV_ATTN_SPLIT", 0);
    if (force > 0) nsplit = force < nqt ? force : nqt;
  }
  p.nsplit = nsplit;
  hipLaunchKernelGGL(fn, dim3(nsplit * bh), dim3(QTHR), lds, st, p);
  GPV_CHECK_LAUNCH();
  return 0;
}
template <typename T, int DHK, int DHV, int NT, int MODE>
int launch_q(const AttnK& p, hipStream_t st) {
  // the causal mask is a template parameter (its per-score tests are compiled out of the encoder / decoder / co-attention
  // launches); key padding costs nothing per score (the kbias row in LDS), so masked and unmasked batches share one kernel
  const bool full = p.skp == NT * 16;
  if (p.causal) return full ? launch_q_f<T, DHK, DHV, NT, MODE, true, true>(p, st) : launch_q_f<T, DHK, DHV, NT, MODE, true, false>(p, st);
  return full ? launch_q_f<T, DHK, DHV, NT, MODE, false, true>(p, st) : launch_q_f<T, DHK, DHV, NT, MODE, false, false>(p, st);
}
template <typename T, int DHK, int DHV>
int launch_kv(const AttnK& p, hipStream_t st) {
  constexpr bool PRECISE = sizeof(T) == 4;
  constexpr int KP = DHK + 8, QC = 64, QTP = QC + 8;
  size_t lds = ((size_t)2 * QC * KP + (size_t)2 * DHV * QTP) * 2 * (PRECISE ? 2 : 1) + 3 * QC * sizeof(float);
  auto fn = attn_kv_kernel<T, DHK, DHV>;
  static size_t attr = 0;
  if (lds > 160 * 1024) return (int)hipErrorInvalidValue;      // more LDS than a CU has: this shape is outside the kernel's range
  if (lds > 64 * 1024 && lds > attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }      // (leave no stale error behind for the next launch check)
    attr = lds;
  }
  dim3 grid((p.Sk + 63) / 64, p.H, p.B);
  hipLaunchKernelGGL(fn, grid, dim3(256), lds, st, p);
  GPV_CHECK_LAUNCH();
  return 0;
}

// resident-queries dK/dV (bf16): returns -1 when the shape does not fit (Sq > 320, or the staged queries exceed the LDS)
template <int DHK, int DHV, int NQT, bool CAUSAL>
int launch_kv2_c(AttnK p, hipStream_t st) {
  constexpr int KP = DHK + 8;
  const int QTP = p.sqp + 8;
  const size_t lds = ((size_t)2 * p.sqp * KP + (size_t)2 * DHV * QTP) * 2 + (size_t)3 * p.sqp * sizeof(float);
  if (lds > 160 * 1024) return -1;
  auto fn = attn_kv2_kernel<DHK, DHV, NQT, CAUSAL>;
  static size_t attr = 0;
  if (lds > 160 * 1024) return (int)hipErrorInvalidValue;      // more LDS than a CU has: this shape is outside the kernel's range
  if (lds > 64 * 1024 && lds > attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }      // (leave no stale error behind for the next launch check)
    attr = lds;
  }
  const int nkt = (p.Sk + 15) / 16;
  const int bh = p.B * p.H;
  int nsplit = (256 + bh - 1) / bh;
  if (nsplit > (nkt + QW - 1) / QW) nsplit = (nkt + QW - 1) / QW;
  if (nsplit < 1) nsplit = 1;
  p.nsplit = nsplit;
  hipLaunchKernelGGL(fn, dim3(nsplit * bh), dim3(QTHR), lds, st, p);
  GPV_CHECK_LAUNCH();
  return 0;
}
template <int DHK, int DHV>
int launch_kv2(AttnK p, hipStream_t st) {
  if (p.Sq > 320) return -1;
  p.sqp = ((p.Sq + 31) / 32) * 32;
  if (p.sqp <= 128) return p.causal ? launch_kv2_c<DHK, DHV, 8, true>(p, st) : launch_kv2_c<DHK, DHV, 8, false>(p, st);
  return p.causal ? launch_kv2_c<DHK, DHV, 20, true>(p, st) : launch_kv2_c<DHK, DHV, 20, false>(p, st);
}

// single-launch backward (bf16): returns -1 when the shape is outside the instantiated range or the LDS
template <int DHK, int DHV, int NQT, int NKT, bool CAUSAL>
int launch_bwd1_c(AttnK p, hipStream_t st) {
  constexpr int KP = DHK + 8;
  const int QTP = p.sqp + 8, vtp = p.skp + 8;
  const size_t lds = ((size_t)2 * p.sqp * KP + (size_t)2 * DHV * QTP + (size_t)DHV * vtp + (size_t)64 * vtp) * 2 + (size_t)3 * p.sqp * sizeof(float);
  if (lds > 160 * 1024) return -1;
  auto fn = attn_bwd1_kernel<DHK, DHV, NQT, NKT, CAUSAL>;
  static size_t attr = 0;
  if (lds > 64 * 1024 && lds > attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }
    attr = lds;
  }
  p.nsplit = 1;
  hipLaunchKernelGGL(fn, dim3(p.B * p.H), dim3(QTHR), lds, st, p);
  GPV_CHECK_LAUNCH();
  ++g_bwd1_launches;
  return 0;
}
template <int DHK, int DHV>
int launch_bwd1(AttnK p, hipStream_t st) {
  if (p.Sq > 320) return -1;
  p.sqp = ((p.Sq + 31) / 32) * 32;
  const bool q8 = p.sqp <= 128, k8 = p.skp <= 128;
  if (p.causal) return (q8 && k8) ? launch_bwd1_c<DHK, DHV, 8, 8, true>(p, st) : -1;
  if (q8 && k8) return launch_bwd1_c<DHK, DHV, 8, 8, false>(p, st);
  if constexpr (DHK == 32) {       // the DETR encoder / decoder shapes (300 keys): three key tiles per wave in registers
    if (q8) return launch_bwd1_c<DHK, DHV, 8, 20, false>(p, st);
    if (!k8) return launch_bwd1_c<DHK, DHV, 20, 20, false>(p, st);
  }
  return -1;
}
int dispatch_bwd1(const AttnK& p, hipStream_t st) {
  if (g_bwd1_mode == 0 || (g_bwd1_mode == 1 && p.B * p.H < 128)) return -1;   // few (batch, head) pairs: the split launches fill the chip better
  switch (p.dh) {
    case 32: return launch_bwd1<32, 32>(p, st);
    case 48: return launch_bwd1<64, 48>(p, st);
    case 64: return launch_bwd1<64, 64>(p, st);
    case 96: return launch_bwd1<96, 96>(p, st);
  }
  return -1;
}

template <typename T, int MODE>
int dispatch_q(const AttnK& p, hipStream_t st) {
  const bool small = p.skp <= 128;
#define GO(DHK, DHV) return small ? launch_q<T, DHK, DHV, 8, MODE>(p, st) : launch_q<T, DHK, DHV, 20, MODE>(p, st)
  switch (p.dh) {
    case 32: GO(32, 32);
    case 48: GO(64, 48);
    case 64: GO(64, 64);
    case 96: GO(96, 96);
  }
#undef GO
  return (int)hipErrorInvalidValue;
}
template <typename T>
int dispatch_kv(const AttnK& p, hipStream_t st) {
  if constexpr (sizeof(T) == 2) {
    static const bool old_only = tune_env("GPV_ATTN_KV_OLD", 0) == 1;
    int r = -1;
    if (!old_only) {
      switch (p.dh) {
        case 32: r = launch_kv2<32, 32>(p, st); break;
        case 48: r = launch_kv2<64, 48>(p, st); break;
        case 64: r = launch_kv2<64, 64>(p, st); break;
        case 96: r = launch_kv2<96, 96>(p, st); break;
      }
    }
    if (r >= 0) return r;
  }
  switch (p.dh) {
    case 32: return launch_kv<T, 32, 32>(p, st);
    case 48: return launch_kv<T, 64, 48>(p, st);
    case 64: return launch_kv<T, 64, 64>(p, st);
    case 96: return launch_kv<T, 96, 96>(p, st);
  }
  return (int)hipErrorInvalidValue;
}

int fill(const gpv_attn_args* a, AttnK& p) {
  if (!a || !a->q || !a->k || !a->v || a->Sk <= 0 || a->Sq <= 0 || a->Sk > 320) return (int)hipErrorInvalidValue;
  p.q = a->q; p.k = a->k; p.v = a->v; p.o = a->o;
  p.q_bs = a->q_bs; p.q_rs = a->q_rs; p.k_bs = a->k_bs; p.k_rs = a->k_rs; p.v_bs = a->v_bs; p.v_rs = a->v_rs;
  p.o_bs = a->o_bs; p.o_rs = a->o_rs;
  p.B = a->B; p.H = a->H; p.Sq = a->Sq; p.Sk = a->Sk; p.dh = a->dh;
  p.skp = ((a->Sk + 63) / 64) * 64;      // key tiles come in groups of four (one uniform test per group in the kernels)
  p.scale = a->scale; p.kpm = a->kpm; p.causal = a->causal;
  p.dthresh = a->drop_p > 0.f ? drop_thresh(a->drop_p) : 0u;
  p.dscale = a->drop_p > 0.f ? 1.f / (1.f - a->drop_p) : 1.f;
  p.seed = a->seed; p.seed_dev = gpvk::g_seed_dev; p.lse = a->lse;
  p.dout = a->dout; p.do_bs = a->do_bs; p.do_rs = a->do_rs; p.dq = a->dq; p.dk = a->dk; p.dv = a->dv;
  return 0;
}
}  // namespace

namespace gpvk {
int attn_bwd1_mode(int set) { const int prev = g_bwd1_mode; g_bwd1_mode = set; return prev; }
long attn_bwd1_launches(long set) { const long prev = g_bwd1_launches; if (set >= 0) g_bwd1_launches = set; return prev; }
}  // namespace gpvk

extern "C" int gpv_attention_fwd(const gpv_attn_args* a, void* stream) {
  AttnK p{};
  int e = fill(a, p);
  if (e) return e;
  if (!a->o) return (int)hipErrorInvalidValue;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  return a->dtype == GPV_F32 ? dispatch_q<float, 0>(p, st) : dispatch_q<bf16, 0>(p, st);
}

template <int NT, bool FULL>
int launch_qkv(AttnK p, const QkvK& xk, hipStream_t st) {
  constexpr int KP = 40;
  const size_t lds = ((size_t)p.skp * KP + (size_t)32 * (p.skp + 8) + (size_t)96 * 256) * 2 + (size_t)p.skp * 4 + 96 * 4;
  auto fn = attn_qkv_kernel<NT, FULL>;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024);
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }
    attr = true;
  }
  hipLaunchKernelGGL(fn, dim3(p.B * p.H), dim3(QTHR), lds, st, p, xk);
  GPV_CHECK_LAUNCH();
  return 0;
}

/* Self-attention with the in-projection inside the launch (attn_qkv_kernel): a->q / a->k / a->v are OUTPUTS here (the projected
 * rows, written where the projection GEMMs would have written them: the backward reads them), xp / x: [B, S, 256] rows (x_bs / x_rs in
 * elements), w: in_proj_weight [768, 256] bf16, bias: [768] fp32 or NULL.  bf16, dh = 32, H * dh = 256, Sq == Sk <= 320, not causal. */
extern "C" int gpv_attention_qkv_fwd(const gpv_attn_args* a, const void* xp, const void* x, int64_t x_bs, int64_t x_rs, const void* w,
                                     const float* bias, void* stream) {
  AttnK p{};
  int e = fill(a, p);
  if (e) return e;
  if (!a->o || !xp || !x || !w || a->dtype != GPV_BF16 || a->dh != 32 || a->H * a->dh != 256 || a->Sq != a->Sk || a->causal) return (int)hipErrorInvalidValue;
  if ((x_rs & 7) || (x_bs & 7) || (a->q_rs & 3) || (a->k_rs & 3) || (a->o_rs & 3)) return (int)hipErrorInvalidValue;
  if ((reinterpret_cast<uintptr_t>(xp) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) & 15) return (int)hipErrorInvalidValue;
  if ((reinterpret_cast<uintptr_t>(a->q) | reinterpret_cast<uintptr_t>(a->k) | reinterpret_cast<uintptr_t>(a->o)) & 7) return (int)hipErrorInvalidValue;
  // v is WRITTEN (element stores: 2-byte alignment is all the kernel needs -- the core re-reads V^T from LDS, not from here); the bias is
  // read as single floats.  Both are checked so that everything the header promises is refused before a launch.
  if (!a->v || (reinterpret_cast<uintptr_t>(a->v) & 1) || (reinterpret_cast<uintptr_t>(bias) & 3) || a->v_rs < a->H * a->dh || a->q_rs < a->H * a->dh ||
      a->k_rs < a->H * a->dh) return (int)hipErrorInvalidValue;
  QkvK xk{reinterpret_cast<const bf16*>(xp), reinterpret_cast<const bf16*>(x), x_bs, x_rs, reinterpret_cast<const bf16*>(w), bias};
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (p.skp <= 128) return p.skp == 128 ? launch_qkv<8, true>(p, xk, st) : launch_qkv<8, false>(p, xk, st);
  return p.skp == 320 ? launch_qkv<20, true>(p, xk, st) : launch_qkv<20, false>(p, xk, st);
}

extern "C" int gpv_attention_bwd(const gpv_attn_args* a, void* stream) {
  AttnK p{};
  int e = fill(a, p);
  if (e) return e;
  if (!a->o || !a->dout || !a->dq || !a->dk || !a->dv || !a->lse) return (int)hipErrorInvalidValue;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (a->dtype != GPV_F32) {
    e = dispatch_bwd1(p, st);
    if (e >= 0) return e;
  }
  e = a->dtype == GPV_F32 ? dispatch_q<float, 1>(p, st) : dispatch_q<bf16, 1>(p, st);
  if (e) return e;
  return a->dtype == GPV_F32 ? dispatch_kv<float>(p, st) : dispatch_kv<bf16>(p, st);
}
