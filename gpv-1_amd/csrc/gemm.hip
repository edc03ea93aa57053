// MFMA bf16 GEMM / implicit-GEMM convolution for gfx950.
//
//   C[M,N] = epilogue( A[M,K] x B[N,K]^T )      256 threads = 4 waves (2x2), BMxBNx32 tiles,
//   register-staged double-buffered LDS, one barrier per k-tile, mfma_f32_16x16x32_bf16.
//
// Operand staging modes
//   A: PLAIN (k contiguous) | TRANS (reduction index is the slow dim; transposed through registers
//      on the way into LDS) | CONV (NHWC implicit-GEMM gather, forward or dgrad geometry)
//   B: PLAIN | TRANS | CONVT (wgrad: gathered activations, reduction = pixels)
// TIn = bf16 : operands are bf16 in HBM.
// TIn = float: "precise" mode, operands are fp32 in HBM and are split hi+lo bf16 while staged;
//              3 MFMAs per product (hi*hi + hi*lo + lo*hi) -> ~1e-6 relative error.
// The MFMA is issued with swapped operands (B fragment as the MFMA "A") so that a lane ends up
// holding 4 CONSECUTIVE columns of one C row -> 8/16-byte epilogue stores.
#include "gemm_common.h"
#include <cstdlib>

using namespace gpvk;

#ifndef GPV_PF
#define GPV_PF 2      /* register prefetch ring depth (tiles). Measured on the ResNet-50 conv shapes (tools/bench_conv.py): 1: 1968 us, 2: 1877 us, 3 (occupancy 3->2): 2136 us */
#endif

namespace {

constexpr int BK = 32;
constexpr int LDK = 40;  // LDS row pitch in elements (80 B: keeps ds_read_b128 16-B aligned, spreads banks)

// ---- 8 staged elements of one operand row -------------------------------------------------
template <typename T> struct Raw8;
__device__ __forceinline__ uint32_t bf_bits(float x) { return (uint32_t)__builtin_bit_cast(unsigned short, (bf16)x); }
template <> struct Raw8<bf16> {
  uint32_t w[4];   // 8 bf16 packed; kept as dwords so that element extraction stays in registers
  __device__ __forceinline__ void zero() { w[0] = w[1] = w[2] = w[3] = 0u; }
  __device__ __forceinline__ void load(const bf16* p) {
    uint4 t = *reinterpret_cast<const uint4*>(p);
    w[0] = t.x; w[1] = t.y; w[2] = t.z; w[3] = t.w;
  }
  __device__ __forceinline__ void load_n(const bf16* p, int n) {
    const unsigned short* q = reinterpret_cast<const unsigned short*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint32_t a = (2 * i < n) ? q[2 * i] : 0u, b = (2 * i + 1 < n) ? q[2 * i + 1] : 0u;
      w[i] = a | (b << 16);
    }
  }
  __device__ __forceinline__ uint32_t hi_bits(int i) const { return (w[i >> 1] >> ((i & 1) * 16)) & 0xffffu; }
  __device__ __forceinline__ float val(int i) const { return __builtin_bit_cast(float, (i & 1) ? (w[i >> 1] & 0xffff0000u) : (w[i >> 1] << 16)); }
  __device__ __forceinline__ uint32_t lo_bits(int) const { return 0u; }
  __device__ __forceinline__ void write(bf16* h, bf16*) const { *reinterpret_cast<uint4*>(h) = make_uint4(w[0], w[1], w[2], w[3]); }
};
template <> struct Raw8<float> {
  float v[8];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = 0.0f;
  }
  __device__ __forceinline__ void load(const float* p) {
    float4 a = *reinterpret_cast<const float4*>(p);
    float4 b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
  __device__ __forceinline__ void load_n(const float* p, int n) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = i < n ? p[i] : 0.0f;
  }
  __device__ __forceinline__ uint32_t hi_bits(int i) const { return bf_bits(v[i]); }
  __device__ __forceinline__ float val(int i) const { return v[i]; }
  __device__ __forceinline__ uint32_t lo_bits(int i) const { return bf_bits(v[i] - (float)(bf16)v[i]); }
  __device__ __forceinline__ void write(bf16* h, bf16* l) const {
    uint32_t a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      a[i] = hi_bits(2 * i) | (hi_bits(2 * i + 1) << 16);
      b[i] = lo_bits(2 * i) | (lo_bits(2 * i + 1) << 16);
    }
    *reinterpret_cast<uint4*>(h) = make_uint4(a[0], a[1], a[2], a[3]);
    *reinterpret_cast<uint4*>(l) = make_uint4(b[0], b[1], b[2], b[3]);
  }
};

// masking helpers (branch-free: select after an unconditional load from a safe address)
template <typename T> __device__ __forceinline__ void mask_raw(Raw8<T>& r, bool ok);
template <> __device__ __forceinline__ void mask_raw<bf16>(Raw8<bf16>& r, bool ok) {
  const uint32_t m = ok ? 0xffffffffu : 0u;
  r.w[0] &= m; r.w[1] &= m; r.w[2] &= m; r.w[3] &= m;
}
template <> __device__ __forceinline__ void mask_raw<float>(Raw8<float>& r, bool ok) {
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = ok ? r.v[i] : 0.0f;
}

// ---- K-contiguous operand (PLAIN or CONV gather): tile [ROWS][32] -------------------------
template <typename T, int ROWS, int MODE, bool VEC>
struct KStage {
  static constexpr int NIT = ROWS / 64;
  struct Buf { Raw8<T> raw[NIT]; bool okf[NIT]; };
  // bf16 16-byte path of the plain operand: buffer loads -- the per-lane byte offset is fixed for the tile (row * ld + slot),
  // the k-tile offset rides in the scalar soffset, rows beyond the operand get an out-of-range offset = hardware zeros
  // (no 64-bit address add, no safe-address select per load, no zero-masking pass before the LDS write)
  static constexpr bool BUF = VEC && MODE == OP_PLAIN && sizeof(T) == 2;
  static constexpr int OOB = 0x7ffffff0;
  const T* base[NIT];
  const T* safe;
  int oh[NIT], ow[NIT];
  bool rowok[NIT];
  __amdgpu_buffer_rsrc_t rs;
  int vo[NIT];

  __device__ __forceinline__ void init(const T* ptr, int64_t ld, int row0, int nrows, int k_begin, const ConvGeom& g) {
    const int tid = threadIdx.x;
    safe = ptr;
    if constexpr (BUF) {
      rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(ptr), (short)0, OOB, 0x00020000);
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int row = (tid + it * 256) >> 2, slot = (tid + it * 256) & 3;
        vo[it] = row0 + row < nrows ? ((row0 + row) * (int)ld + slot * 8) * 2 : OOB;
      }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      int row = (tid + it * 256) >> 2;
      int r = row0 + row;
      rowok[it] = r < nrows;
      if (MODE == OP_CONV) {
        int rr = conv_row_to_pixel(rowok[it] ? r : 0, g);
        int b = rr / (g.OH * g.OW);
        int rem = rr - b * (g.OH * g.OW);
        oh[it] = rem / g.OW;
        ow[it] = rem - oh[it] * g.OW;
        base[it] = ptr + (int64_t)b * g.IH * g.IW * g.Cs;
      } else {
        base[it] = ptr + (int64_t)(rowok[it] ? r : 0) * ld;
        oh[it] = ow[it] = 0;
      }
    }
  }
  __device__ __forceinline__ void load(int k0, int K, const ConvGeom& g, Buf& bf) {
    Raw8<T>(&raw)[NIT] = bf.raw; bool(&okf)[NIT] = bf.okf;
    const int tid = threadIdx.x;
    if constexpr (BUF) {
      typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
      const bool full = k0 + BK <= K;                      // uniform; K % 8 == 0 or finite padding (host): whole chunks
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int slot = (tid + it * 256) & 3;
        const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, (full || k0 + slot * 8 < K) ? vo[it] : OOB, k0 * 2, 0);
        raw[it].w[0] = t[0]; raw[it].w[1] = t[1]; raw[it].w[2] = t[2]; raw[it].w[3] = t[3];
        okf[it] = true;
      }
      return;
    }
    int tr = 0, ts = 0, c0 = k0;
    if (MODE == OP_CONV) {
      int tap = k0 / g.Cin;
      c0 = k0 - tap * g.Cin;
      tr = tap / g.KW;
      ts = tap - tr * g.KW;
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      int slot = (tid + it * 256) & 3;
      int k = k0 + slot * 8;
      bool ok = rowok[it] && k < K;
      const T* p;
      if (MODE == OP_CONV) {
        int ih, iw;
        if (g.dgrad) {                       // strides are 1 or 2 (checked on the host)
          int th = oh[it] + g.PH - tr, tw = ow[it] + g.PW - ts;
          int sh = g.SH - 1, sw = g.SW - 1;  // shift amount == mask for stride in {1,2}
          ih = th >> sh; iw = tw >> sw;
          ok = ok && th >= 0 && tw >= 0 && ((th & sh) == 0) && ((tw & sw) == 0) && ih < g.IH && iw < g.IW;
        } else {
          ih = oh[it] * g.SH + tr - g.PH; iw = ow[it] * g.SW + ts - g.PW;
          ok = ok && ih >= 0 && iw >= 0 && ih < g.IH && iw < g.IW;
        }
        p = base[it] + ((int64_t)(ih * g.IW + iw)) * g.Cs + c0 + slot * 8;
      } else {
        p = base[it] + k;
      }
      if (VEC) {
        // unconditional 16-B load (from a safe address when masked); the zero-select is applied in
        // store(), AFTER the MFMAs of the current tile, so the load latency overlaps the math
        raw[it].load(ok ? p : safe);
        okf[it] = ok;
      } else {
        if (!ok) raw[it].zero();
        else raw[it].load_n(p, K - k);
      }
    }
  }
  __device__ __forceinline__ void store(bf16* hi, bf16* lo, bool, Buf& bf) {
    Raw8<T>(&raw)[NIT] = bf.raw; bool(&okf)[NIT] = bf.okf;
    const int tid = threadIdx.x;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      int i = tid + it * 256;
      int off = (i >> 2) * LDK + (i & 3) * 8;
      if (VEC) mask_raw<T>(raw[it], okf[it]);
      raw[it].write(hi + off, lo + off);
    }
  }
};

// ---- reduction-major operand (TRANS, or CONVT gather): memory tile [32 red][COLS] -------------------------
// Staged ROW-MAJOR exactly as it is loaded (16-B global loads, conflict-free ds_write_b128, pitch COLS+16), and
// consumed with the gfx950 LDS transpose read: one ds_read_b64_tr_b16 gives a lane 4 consecutive reduction rows of
// ITS column (verified on hardware by tools/probe/trread.hip); two of them = one 16x16x32 MFMA operand fragment.
template <typename T, int COLS, int MODE, bool VEC, bool SUM = false>
struct TStage {
  static constexpr int CHK = COLS / 8;            // 16-B chunks per reduction row
  static constexpr int NIT = (BK * CHK) / 256;    // chunks per thread
  static constexpr int PITCH = COLS + 16;         // elements; (PITCH/2) % 64 in {8, 40}: 8 rows hit distinct bank octets
  struct Buf { Raw8<T> raw[NIT]; bool okf[NIT]; };
  static constexpr bool BUF = VEC && MODE == OP_PLAIN && sizeof(T) == 2;    // see KStage
  static constexpr int OOB = 0x7ffffff0;
  __amdgpu_buffer_rsrc_t rs;
  int vo[NIT];
  const T* ptr; int64_t ld; int col0, ncols;
  int tap_r, tap_s, c0;
  int pb[NIT], poh[NIT], pow_[NIT];      // CONVT: (batch, oh, ow) of this thread's reduction rows, advanced per k-tile
  float csum[SUM ? NIT : 1][8];          // SUM: running sum over the reduction of this thread's 8 columns (bias gradient)
  bool do_sum;

  __device__ __forceinline__ void init(const T* p, int64_t ld_, int col0_, int ncols_, int k_begin, const ConvGeom& g) {
    ptr = p; ld = ld_; col0 = col0_; ncols = ncols_;
    tap_r = tap_s = c0 = 0;
    do_sum = false;
    if constexpr (BUF) {
      rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(p), (short)0, OOB, 0x00020000);
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int idx = threadIdx.x + it * 256;
        const int kr = idx / CHK, ch = idx - kr * CHK;
        vo[it] = col0 + ch * 8 < ncols ? (kr * (int)ld + col0 + ch * 8) * 2 : OOB;
      }
    }
    if constexpr (SUM) {
#pragma unroll
      for (int it = 0; it < NIT; ++it)
#pragma unroll
        for (int e = 0; e < 8; ++e) csum[it][e] = 0.f;
    }
    if (MODE == OP_CONV) {
      int tap = col0 / g.Cin;
      c0 = col0 - tap * g.Cin;
      tap_r = tap / g.KW;
      tap_s = tap - tap_r * g.KW;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        int k = k_begin + (threadIdx.x + it * 256) / CHK;
        pb[it] = k / (g.OH * g.OW);
        int rem = k - pb[it] * (g.OH * g.OW);
        poh[it] = rem / g.OW;
        pow_[it] = rem - poh[it] * g.OW;
      }
    }
  }
  __device__ __forceinline__ void load(int k0, int K, const ConvGeom& g, Buf& bf) {
    Raw8<T>(&raw)[NIT] = bf.raw; bool(&okf)[NIT] = bf.okf;
    if constexpr (BUF) {
      typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
      const bool full = k0 + BK <= K;                      // uniform
      const int soff = k0 * (int)ld * 2;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int kr = (threadIdx.x + it * 256) / CHK;
        const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, (full || k0 + kr < K) ? vo[it] : OOB, soff, 0);
        raw[it].w[0] = t[0]; raw[it].w[1] = t[1]; raw[it].w[2] = t[2]; raw[it].w[3] = t[3];
        okf[it] = true;
      }
      return;
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = threadIdx.x + it * 256;
      const int kr = idx / CHK, ch = idx - kr * CHK;
      const int k = k0 + kr, col = col0 + ch * 8;
      bool ok = k < K && col < ncols;
      const T* src;
      if (MODE == OP_CONV) {
        int ih = poh[it] * g.SH + tap_r - g.PH, iw = pow_[it] * g.SW + tap_s - g.PW;
        ok = ok && ih >= 0 && iw >= 0 && ih < g.IH && iw < g.IW;
        src = ptr + ((int64_t)((pb[it] * g.IH + ih) * g.IW + iw)) * g.Cs + c0 + ch * 8;
        pow_[it] += BK;                           // advance this reduction row by BK pixels for the next k-tile
        while (pow_[it] >= g.OW) { pow_[it] -= g.OW; poh[it] += 1; }
        while (poh[it] >= g.OH) { poh[it] -= g.OH; pb[it] += 1; }
      } else {
        src = ptr + (int64_t)k * ld + col;
      }
      if (VEC) {                                  // host guarantees ncols % 8 == 0 for the VEC variant
        raw[it].load(ok ? src : ptr);
        okf[it] = ok;
      } else {
        if (!ok) raw[it].zero();
        else raw[it].load_n(src, ncols - col);
      }
    }
  }
  __device__ __forceinline__ void store(bf16* hi, bf16* lo, bool, Buf& bf) {
    Raw8<T>(&raw)[NIT] = bf.raw; bool(&okf)[NIT] = bf.okf;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = threadIdx.x + it * 256;
      const int kr = idx / CHK, ch = idx - kr * CHK;
      if (VEC) mask_raw<T>(raw[it], okf[it]);
      if constexpr (SUM) {
        if (do_sum) {
#pragma unroll
          for (int e = 0; e < 8; ++e) csum[it][e] += raw[it].val(e);
        }
      }
      raw[it].write(hi + kr * PITCH + ch * 8, lo + kr * PITCH + ch * 8);
    }
  }
  // SUM: add this block's column sums (over its reduction range) to out[col0 + c]; threads with equal column chunk
  // are reduced with lane shuffles first, then one atomic per column and wave
  __device__ __forceinline__ void flush_sums(float* out) {
    if constexpr (SUM) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int ch = (threadIdx.x + it * 256) % CHK;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float v = csum[it][e];
#pragma unroll
          for (int s = CHK; s < 64; s <<= 1) v += __shfl_xor(v, s);
          const int col = col0 + ch * 8 + e;
          if ((threadIdx.x & 63) < CHK && col < ncols) atomicAdd(out + col, v);
        }
      }
    }
  }
  // fragment for the 16 columns starting at `cbase` of the staged tile: lane (col = lane&15, g = lane>>4) receives
  // reduction rows 8g..8g+7
  static __device__ __forceinline__ bf16x8 frag(const bf16* tile, int cbase, int lane) {
    typedef short __attribute__((ext_vector_type(4))) s16x4;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const int g = lane >> 4, i = lane & 15;
    const bf16* p0 = tile + (8 * g + (i >> 2)) * PITCH + cbase + (i & 3) * 4;
    s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0));
    s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0 + 4 * PITCH));
    typedef short __attribute__((ext_vector_type(8))) s16x8;
    s16x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, r);
  }
};

template <typename T> struct Vec8IO;
template <> struct Vec8IO<bf16> {
  static __device__ __forceinline__ void ld(const bf16* p, float* o) { Ld8<bf16>::ld(p, o); }
  static __device__ __forceinline__ void st(bf16* p, const float* o) { Ld8<bf16>::st(p, o); }
  // non-temporal forms (streaming: do not displace what the next kernels will read from L2 / MALL)
  static __device__ __forceinline__ void ld_nt(const bf16* p, float* o) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const bf16x8 v = __builtin_bit_cast(bf16x8, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)));
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (float)v[i];
  }
  static __device__ __forceinline__ void st_nt(bf16* p, const float* o) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    bf16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (bf16)o[i];
    __builtin_nontemporal_store(__builtin_bit_cast(u32x4, v), reinterpret_cast<u32x4*>(p));
  }
};
template <> struct Vec8IO<float> {
  static __device__ __forceinline__ void ld(const float* p, float* o) { Ld8<float>::ld(p, o); }
  static __device__ __forceinline__ void st(float* p, const float* o) { Ld8<float>::st(p, o); }
  static __device__ __forceinline__ void ld_nt(const float* p, float* o) { Ld8<float>::ld(p, o); }
  static __device__ __forceinline__ void st_nt(float* p, const float* o) { Ld8<float>::st(p, o); }
};

template <typename TIn, typename TOut, int AMODE, int BMODE, int BM, int BN, bool VEC, bool NTIO = false>
__device__ __forceinline__ void gemm_body(const GemmK& p) {
  constexpr bool PRECISE = sizeof(TIn) == 4;
  constexpr int FM = BM / 32, FN = BN / 32;
  constexpr int A_ELEMS = BM * LDK, B_ELEMS = BN * LDK;
  constexpr int STAGE_ELEMS = (A_ELEMS + B_ELEMS) * (PRECISE ? 2 : 1);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16* smem = reinterpret_cast<bf16*>(smem_raw);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order: workgroup b runs on XCD b % 8 (observed dispatch rule, used for speed only).  Give every
  // XCD a CONTIGUOUS range of tiles so that the column tiles of one row panel (same A rows / same conv pixels) share
  // one L2 instead of being fetched from HBM once per XCD.  Bijective for any grid size.
  // With a split reduction (gridDim.y > 1) the (tile, split) plane is remapped as a whole, split-major: an XCD then
  // runs ALL tiles of one reduction slice together, so the slice's dy / x rows are fetched from HBM once and shared
  // through that XCD's L2 (wgrad of a 1x1 conv has only 4..32 output tiles: tile-only remapping left every XCD
  // with one tile and all slices, i.e. no sharing at all: TCC hit rate 0.6 %).
  int tile, ksplit;
  {
    const bool plane = gridDim.z == 1;
    const int gx = gridDim.x;
    const int nwg = plane ? gx * (int)gridDim.y : gx, bid = plane ? (int)blockIdx.x + gx * (int)blockIdx.y : (int)blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    ksplit = plane ? v / gx : (int)blockIdx.y;
    tile = plane ? v - ksplit * gx : v;
  }
  int tm = tile / p.tilesN;
  const int tn = tile - tm * p.tilesN;
  if constexpr (AMODE == OP_CONV) {
    // parity-class-major dgrad: the four classes have different tap counts (0..4 taps); cycle the classes
    // through consecutive row panels so every XCD's contiguous tile range gets the same mix of work
    if (p.cg.cm && p.cg.cls_rows % BM == 0 && p.M == 4 * p.cg.cls_rows) {
      const int tpc = p.cg.cls_rows / BM;
      tm = (tm & 3) * tpc + (tm >> 2);
    }
  }
  const int row0 = tm * BM, col0 = tn * BN;
  const int batch = blockIdx.z;

  int kt_total = (p.K + BK - 1) / BK;
  // stride-2 dgrad with parity-class-major rows: a class-uniform tile only visits its own taps
  int cm_r0 = 0, cm_s0 = 0, cm_nS = 1, cm_cpt = 1;
  bool cm_on = false, cm_empty = false;
  __shared__ int s_rowpix[AMODE == OP_CONV ? BM : 1];   // class-major row -> output pixel (divisions done once per tile row)
  if constexpr (AMODE == OP_CONV) {
    if (p.cg.cm && threadIdx.x < BM) s_rowpix[threadIdx.x] = conv_row_to_pixel(min(row0 + (int)threadIdx.x, p.M - 1), p.cg);
  }
  if constexpr (AMODE == OP_CONV) {
    if (p.cg.cm) {
      const int c_lo = row0 / p.cg.cls_rows, c_hi = min(row0 + BM - 1, p.M - 1) / p.cg.cls_rows;
      if (c_lo == c_hi) {
        cm_on = true;
        cm_r0 = ((c_lo >> 1) + p.cg.PH) & 1;
        cm_s0 = ((c_lo & 1) + p.cg.PW) & 1;
        const int nR = (p.cg.KH - cm_r0 + 1) / 2;
        cm_nS = (p.cg.KW - cm_s0 + 1) / 2;
        cm_cpt = p.cg.Cin / BK;
        kt_total = nR * cm_nS * cm_cpt;
        if (kt_total == 0) { cm_empty = true; kt_total = 1; }   // no tap reaches this class: one fully masked k-tile, dx = res * mask
      }
    }
  }
  auto kmap = [&](int kt) -> int {      // k-tile index -> offset on the full [taps x Cin] reduction axis
    if (!cm_on) return kt * BK;
    if (cm_empty) return p.K;
    const int t = kt / cm_cpt, c = kt - t * cm_cpt;
    const int ri = t / cm_nS, si = t - ri * cm_nS;
    return ((cm_r0 + 2 * ri) * p.cg.KW + cm_s0 + 2 * si) * p.cg.Cin + c * BK;
  };
  const int kt0 = ksplit * p.kt_per_split;
  const int kt1 = min(kt_total, kt0 + p.kt_per_split);
  if (kt0 >= kt1) return;

  const TIn* Ap = reinterpret_cast<const TIn*>(p.A) + (int64_t)batch * p.sA;
  const TIn* Bp = reinterpret_cast<const TIn*>(p.B) + (int64_t)batch * p.sB;

  constexpr bool ASUM = AMODE == OP_TRANS && BMODE == OP_TRANS;     // linear wgrad: bias gradient = row sums of A (= dY^T)
  typedef typename std::conditional<AMODE == OP_TRANS, TStage<TIn, BM, OP_PLAIN, VEC, ASUM>, KStage<TIn, BM, AMODE, VEC>>::type AStage;
  typedef typename std::conditional<BMODE == OP_PLAIN, KStage<TIn, BN, OP_PLAIN, VEC>,
                                    TStage<TIn, BN, (BMODE == OP_CONV ? OP_CONV : OP_PLAIN), VEC>>::type BStage;
  AStage as; BStage bs;
  as.init(Ap, p.lda, row0, p.M, kt0 * BK, p.cg);
  if constexpr (ASUM) as.do_sum = p.a_rowsum != nullptr && tn == 0;
  bs.init(Bp, p.ldb, col0, p.N, kt0 * BK, p.cg);

  auto stage_ptr = [&](int s, int which) -> bf16* {  // which: 0 Ahi 1 Alo 2 Bhi 3 Blo
    bf16* b = smem + s * STAGE_ELEMS;
    if (PRECISE) {
      switch (which) { case 0: return b; case 1: return b + A_ELEMS; case 2: return b + 2 * A_ELEMS; default: return b + 2 * A_ELEMS + B_ELEMS; }
    }
    return which < 2 ? b : b + A_ELEMS;
  };
  // register prefetch ring, PF tiles deep: the global loads of tile t+PF-1 are issued while tile t is being
  // multiplied, so a load has PF-1 whole iterations (not one) to come back before its ds_write needs it.
  constexpr int PF = GPV_PF;
  typename AStage::Buf abuf[PF];
  typename BStage::Buf bbuf[PF];

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int d = 0; d < PF - 1; ++d) {
    if (kt0 + d < kt1) {
      as.load(kmap(kt0 + d), p.K, p.cg, abuf[d]);
      bs.load(kmap(kt0 + d), p.K, p.cg, bbuf[d]);
    }
  }
  as.store(stage_ptr(0, 0), stage_ptr(0, 1), PRECISE, abuf[0]);
  bs.store(stage_ptr(0, 2), stage_ptr(0, 3), PRECISE, bbuf[0]);
  __syncthreads();

  int cur = 0;
  const int a_off = (wm * (BM / 2) + (lane & 15)) * LDK + (lane >> 4) * 8;
  const int b_off = (wn * (BN / 2) + (lane & 15)) * LDK + (lane >> 4) * 8;
  auto compute = [&](int st) {
    bf16x8 af[FM], bfr[FN];
    const bf16* At = stage_ptr(st, 0);
    const bf16* Bt = stage_ptr(st, 2);
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      if constexpr (AMODE == OP_TRANS) af[i] = AStage::frag(At, wm * (BM / 2) + i * 16, lane);
      else af[i] = *reinterpret_cast<const bf16x8*>(At + a_off + i * 16 * LDK);
    }
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      if constexpr (BMODE != OP_PLAIN) bfr[j] = BStage::frag(Bt, wn * (BN / 2) + j * 16, lane);
      else bfr[j] = *reinterpret_cast<const bf16x8*>(Bt + b_off + j * 16 * LDK);
    }
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = mfma16(bfr[j], af[i], acc[i][j]);
    if constexpr (PRECISE) {
      const bf16* Alt = stage_ptr(st, 1);
      const bf16* Blt = stage_ptr(st, 3);
      bf16x8 al[FM], bl[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        if constexpr (AMODE == OP_TRANS) al[i] = AStage::frag(Alt, wm * (BM / 2) + i * 16, lane);
        else al[i] = *reinterpret_cast<const bf16x8*>(Alt + a_off + i * 16 * LDK);
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        if constexpr (BMODE != OP_PLAIN) bl[j] = BStage::frag(Blt, wn * (BN / 2) + j * 16, lane);
        else bl[j] = *reinterpret_cast<const bf16x8*>(Blt + b_off + j * 16 * LDK);
      }
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          acc[i][j] = mfma16(bl[j], af[i], acc[i][j]);
          acc[i][j] = mfma16(bfr[j], al[i], acc[i][j]);
        }
    }
  };
  for (int kt = kt0; kt < kt1; kt += PF) {
#pragma unroll
    for (int d = 0; d < PF; ++d) {       // statically indexed ring slots (runtime indices would spill to scratch)
      const int t = kt + d;
      if (t < kt1) {
        if (t + PF - 1 < kt1) {
          as.load(kmap(t + PF - 1), p.K, p.cg, abuf[(d + PF - 1) % PF]);
          bs.load(kmap(t + PF - 1), p.K, p.cg, bbuf[(d + PF - 1) % PF]);
        }
        compute(cur);
        if (t + 1 < kt1) {
          as.store(stage_ptr(cur ^ 1, 0), stage_ptr(cur ^ 1, 1), PRECISE, abuf[(d + 1) % PF]);
          bs.store(stage_ptr(cur ^ 1, 2), stage_ptr(cur ^ 1, 3), PRECISE, bbuf[(d + 1) % PF]);
        }
        __syncthreads();
        cur ^= 1;
      }
    }
  }

  TOut* Cp = reinterpret_cast<TOut*>(p.C) + (int64_t)batch * p.sC;
  int64_t ldc = p.ldc;
  if (p.ws != nullptr) {            // split reduction through a workspace: this split's partial product is a dense [M,N] slab
    Cp = reinterpret_cast<TOut*>(p.ws) + (int64_t)ksplit * p.M * p.N;
    ldc = p.N;
  }
  const TOut* Rp = p.res ? reinterpret_cast<const TOut*>(p.res) + (int64_t)batch * p.sR : nullptr;
  const TOut* Mp = reinterpret_cast<const TOut*>(p.mask);

  if constexpr (ASUM) {
    if (as.do_sum) as.flush_sums(p.a_rowsum);
  }
  if (p.accumulate) {
    // ---------------- split-K / accumulate epilogue: fp32 atomics straight from the fragments ----------------
    if constexpr (sizeof(TOut) == 4) {
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int m = row0 + wm * (BM / 2) + i * 16 + (lane & 15);
        if (m >= p.M) continue;
        const float rs = p.rowscale ? p.rowscale[m] * p.alpha : p.alpha;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const int n = col0 + wn * (BN / 2) + j * 16 + (lane >> 4) * 4;
          float* dst = reinterpret_cast<float*>(Cp) + (int64_t)m * ldc + n;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n + r < p.N) atomicAdd(dst + r, acc[i][j][r] * rs);
        }
      }
    }
    return;
  }

  // ---------------- coalesced epilogue: fragments -> LDS (fp32, half a tile at a time) -> full rows ----------------
  // (a lane holds 4 consecutive columns of 16 different rows; writing that straight to HBM gives 32-byte
  //  row segments.  Through LDS every wave-store instruction covers whole 128/256-byte row runs.)
  float* ep = reinterpret_cast<float*>(smem_raw);
  constexpr int EPITCH = BN + 4;
  constexpr int HR = BM / 2;
  constexpr int CH = BN / 8;
  constexpr int NCH = (HR * CH) / 256;
  const bool v_st = (ldc % 8 == 0) && ((reinterpret_cast<uintptr_t>(Cp) & 15) == 0);
  const bool v_res = Rp && (p.ldr % 8 == 0) && ((reinterpret_cast<uintptr_t>(Rp) & 15) == 0);
  const bool v_msk = Mp && (p.ldm % 8 == 0) && ((reinterpret_cast<uintptr_t>(Mp) & 15) == 0);
  // every chunk this thread finishes covers the SAME 8 columns (256 % CH == 0): fetch their bias once
  // (per-element bias loads inside the readback loop cost more than the whole GEMM on the K=64 convs)
  float bv[8];
  {
    const int nb = col0 + (tid % CH) * 8;
    const bool vb = p.bias && nb + 8 <= p.N && ((reinterpret_cast<uintptr_t>(p.bias) & 15) == 0);
    if (vb) {
      const float4 b0 = *reinterpret_cast<const float4*>(p.bias + nb);
      const float4 b1 = *reinterpret_cast<const float4*>(p.bias + nb + 4);
      bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) bv[e] = (p.bias && nb + e < p.N) ? p.bias[nb + e] : 0.f;
    }
  }
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if (half == 1) __syncthreads();      // (the main loop ended with a barrier)
    if (wm == half) {
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int ml = i * 16 + (lane & 15);
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const int nl = wn * (BN / 2) + j * 16 + (lane >> 4) * 4;
          *reinterpret_cast<float4*>(ep + ml * EPITCH + nl) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
      }
    }
    __syncthreads();
    // readback in groups of EPG chunks: the residual / mask loads of a group are issued together (independent
    // loads in flight), then the staged accumulators are finished and stored as whole row runs.  Small groups
    // keep the epilogue's register footprint below the main loop's, so it does not cost occupancy.
    constexpr int EPG = NCH >= 2 ? 2 : 1;
#pragma unroll
    for (int c0 = 0; c0 < NCH; c0 += EPG) {
      float rv[EPG][8], mv[EPG][8];
#pragma unroll
      for (int g = 0; g < EPG; ++g) {
        const int idx = tid + (c0 + g) * 256;
        const int r = idx / CH, c8 = idx - r * CH;
        const int m = row0 + half * HR + r;
        const int n = col0 + c8 * 8;
        const bool inb = m < p.M && n < p.N;
        const bool full = n + 8 <= p.N;
        const int64_t mp = (AMODE == OP_CONV && p.cg.cm && inb) ? s_rowpix[m - row0] : m;     // output pixel of this GEMM row
        if (Rp) {
          if (inb && v_res && full) { if constexpr (NTIO) Vec8IO<TOut>::ld_nt(Rp + mp * p.ldr + n, rv[g]); else Vec8IO<TOut>::ld(Rp + mp * p.ldr + n, rv[g]); }
          else {
#pragma unroll
            for (int e = 0; e < 8; ++e) rv[g][e] = (inb && n + e < p.N) ? (float)Rp[mp * p.ldr + n + e] : 0.f;
          }
        }
        if (Mp) {
          if (inb && v_msk && full) Vec8IO<TOut>::ld(Mp + mp * p.ldm + n, mv[g]);
          else {
#pragma unroll
            for (int e = 0; e < 8; ++e) mv[g][e] = (inb && n + e < p.N) ? (float)Mp[mp * p.ldm + n + e] : 0.f;
          }
        }
      }
#pragma unroll
      for (int g = 0; g < EPG; ++g) {
        const int idx = tid + (c0 + g) * 256;
        const int r = idx / CH, c8 = idx - r * CH;
        const int m = row0 + half * HR + r;
        const int n = col0 + c8 * 8;
        if (m >= p.M || n >= p.N) continue;
        float v[8];
        {
          const float4 a = *reinterpret_cast<const float4*>(ep + r * EPITCH + c8 * 8);
          const float4 b = *reinterpret_cast<const float4*>(ep + r * EPITCH + c8 * 8 + 4);
          v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        }
        const float rs = p.rowscale ? p.rowscale[m] * p.alpha : p.alpha;
        const bool full = n + 8 <= p.N;
        const uint32_t keep8 = p.dthresh ? drop_mask<8>(p.seed, ((uint64_t)batch * p.M + m) * (uint64_t)p.N + n, p.dthresh) : 0xffu;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float x = v[e] * rs;
          x += bv[e];
          if (Rp) x += rv[g][e];
          if (p.act == GPV_ACT_RELU) x = fmaxf(x, 0.f);
          else if (p.act == GPV_ACT_GELU) x = gelu_erf(x);
          if (p.dthresh) x = ((keep8 >> e) & 1u) ? x * p.dscale : 0.f;
          if (Mp) x = mv[g][e] > 0.f ? x : 0.f;
          v[e] = x;
        }
        const int64_t mp = (AMODE == OP_CONV && p.cg.cm && m < p.M) ? s_rowpix[m - row0] : m;
        TOut* dst = Cp + mp * ldc + n;
        if (v_st && full) { if constexpr (NTIO) Vec8IO<TOut>::st_nt(dst, v); else Vec8IO<TOut>::st(dst, v); }   // (a run-time choice: the two stores are merged and lose the hint)
        else {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (n + e < p.N) dst[e] = (TOut)v[e];
        }
      }
    }
  }
}

template <typename TIn, typename TOut, int AMODE, int BMODE, int BM, int BN, bool VEC>
__global__ __launch_bounds__(256) void gemm_kernel(GemmK p) {
  if (p.dthresh) p.seed = eff_seed(p.seed, p.seed_dev);
  gemm_body<TIn, TOut, AMODE, BMODE, BM, BN, VEC>(p);
}
// the same body under its own name for the backbone's 1x1 stride-1 convolutions (plain GEMMs over the pixel rows):
// profiles and PMC passes can then tell them from the transformer's Linear layers
template <typename TIn, typename TOut, int BM, int BN, bool VEC>
__global__ __launch_bounds__(256) void conv1x1_kernel(GemmK p) {
  if (p.dthresh) p.seed = eff_seed(p.seed, p.seed_dev);
  gemm_body<TIn, TOut, OP_PLAIN, OP_PLAIN, BM, BN, VEC>(p);
}
// ... with non-temporal epilogue stores / residual loads (GemmK::nt_io: outputs far larger than the MALL)
template <int BM, int BN>
__global__ __launch_bounds__(256) void conv1x1_nt_kernel(GemmK p) {
  gemm_body<bf16, bf16, OP_PLAIN, OP_PLAIN, BM, BN, true, true>(p);
}

// C[m, n] += sum_s ws[s][m][n]   (second pass of a workspace split reduction).  A group of G threads shares one run
// of 4 columns and strides over the splits (a layer2 1x1 wgrad has 96 slabs of 64K outputs: one thread per output
// quad walking 96 slabs was 64 blocks of serial latency), partial sums meet in LDS.
template <int G>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int nsplit, int M, int N, float* __restrict__ C,
                                                            int64_t ldc) {
  constexpr int QB = 256 / G;                      // quads per block
  __shared__ float4 part[G > 1 ? 256 : 1];
  const int64_t slab = (int64_t)M * N;
  const int nq = N >> 2;                           // host guarantees N % 4 == 0, ldc % 4 == 0, C 16-byte aligned
  const int64_t total = (int64_t)M * nq;
  const int ql = threadIdx.x % QB, g = threadIdx.x / QB;
  const int64_t i = (int64_t)blockIdx.x * QB + ql;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  int m = 0, n = 0;
  if (i < total) {
    m = (int)(i / nq); n = (int)(i - (int64_t)m * nq) * 4;
    const float* src = ws + (int64_t)m * N + n;
    for (int s = g; s < nsplit; s += G) {
      const float4 v = *reinterpret_cast<const float4*>(src + s * slab);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
  }
  if constexpr (G > 1) {
    part[threadIdx.x] = a;
    __syncthreads();
    if (g == 0) {
#pragma unroll
      for (int k = 1; k < G; ++k) {
        const float4 v = part[k * QB + ql];
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
      }
    }
  }
  if (g == 0 && i < total) {
    float4* dst = reinterpret_cast<float4*>(C + (int64_t)m * ldc + n);
    float4 c = *dst;
    c.x += a.x; c.y += a.y; c.z += a.z; c.w += a.w;
    *dst = c;
  }
}

}  // namespace

namespace gpvk {
// C[m,n] += sum over `split` dense [M,N] slabs of ws (N % 4 == 0, ldc % 4 == 0, C 16-byte aligned)
int launch_splitk_reduce(const float* ws, int split, int M, int N, float* C, int64_t ldc, hipStream_t st) {
  const int64_t quads = (int64_t)M * (N / 4);
  if (split >= 32) {
    hipLaunchKernelGGL(splitk_reduce_kernel<16>, dim3((unsigned)((quads + 15) / 16)), dim3(256), 0, st, ws, split, M, N, C, ldc);
  } else if (split >= 4) {
    hipLaunchKernelGGL(splitk_reduce_kernel<4>, dim3((unsigned)((quads + 63) / 64)), dim3(256), 0, st, ws, split, M, N, C, ldc);
  } else {
    hipLaunchKernelGGL(splitk_reduce_kernel<1>, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, ws, split, M, N, C, ldc);
  }
  GPV_CHECK_LAUNCH();
  return 0;
}
}  // namespace gpvk

namespace {

template <typename TIn, typename TOut, int AMODE, int BMODE, int BM, int BN, bool VEC>
int launch_cfg_v(const GemmK& k, int batch, hipStream_t st) {
  constexpr bool PRECISE = sizeof(TIn) == 4;
  constexpr size_t lds = (size_t)2 * (BM + BN) * LDK * 2 * (PRECISE ? 2 : 1);
  GemmK p = k;
  const int tilesM = (p.M + BM - 1) / BM;
  p.tilesN = (p.N + BN - 1) / BN;
  const int kt_total = (p.K + BK - 1) / BK;
  int split = p.split_k < 1 ? 1 : p.split_k;
  if (split > kt_total) split = kt_total;
  p.kt_per_split = (kt_total + split - 1) / split;
  split = (kt_total + p.kt_per_split - 1) / p.kt_per_split;
  // Split reduction without atomics when the caller lent a workspace: every split writes its partial [M,N] product
  // with the normal coalesced epilogue, a second tiny kernel adds the slabs to C.  fp32 atomics on C cost
  // outputs x splits / ~94e9 s (67 us for the 128x512 gradient of a layer2 1x1 conv at its best split of 96).
  p.ws = nullptr;
  bool two_pass = false;
  if (p.accumulate && split > 1 && batch == 1 && k.ws_base != nullptr && sizeof(TOut) == 4 &&
      (int64_t)split * p.M * p.N * 4 <= k.ws_bytes && p.N % 4 == 0 && p.ldc % 4 == 0 &&
      (reinterpret_cast<uintptr_t>(p.C) & 15) == 0) {
    two_pass = true;
    p.ws = reinterpret_cast<float*>(k.ws_base);
    p.accumulate = 0; p.res = nullptr; p.mask = nullptr; p.bias = nullptr; p.act = 0; p.dthresh = 0;
  }
  auto fn = gemm_kernel<TIn, TOut, AMODE, BMODE, BM, BN, VEC>;
  if constexpr (AMODE == OP_PLAIN && BMODE == OP_PLAIN) {
    if (k.conv1x1) fn = conv1x1_kernel<TIn, TOut, BM, BN, VEC>;
    if constexpr (std::is_same<TIn, bf16>::value && std::is_same<TOut, bf16>::value && VEC) {
      if (k.conv1x1 && k.nt_io && !p.dthresh && !two_pass) fn = conv1x1_nt_kernel<BM, BN>;
    }
  }
  static bool attr_done[2] = {false, false};
  if (lds > 64 * 1024 && !attr_done[k.conv1x1 ? 1 : 0]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }      // (leave no stale error behind for the next launch check)
    attr_done[k.conv1x1 ? 1 : 0] = true;
  }
  dim3 grid(tilesM * p.tilesN, split, batch);
  hipLaunchKernelGGL(fn, grid, dim3(256), lds, st, p);
  GPV_CHECK_LAUNCH();
  if (two_pass) {
    const int e = launch_splitk_reduce(p.ws, split, p.M, p.N, reinterpret_cast<float*>(p.C), p.ldc, st);
    if (e) return e;
  }
  return 0;
}

template <typename TIn, typename TOut, int AMODE, int BMODE, int BM, int BN>
int launch_cfg(const GemmK& k, int batch, hipStream_t st) {
  if (k.vecA && k.vecB) return launch_cfg_v<TIn, TOut, AMODE, BMODE, BM, BN, true>(k, batch, st);
  return launch_cfg_v<TIn, TOut, AMODE, BMODE, BM, BN, false>(k, batch, st);
}

// Second pass of a split FORWARD convolution: y[m, n] = act(sum_s ws[s][m][n] + bias[n] + res[m, n]), four columns per thread.
__global__ __launch_bounds__(256) void conv_split_epilogue_kernel(const float* __restrict__ ws, int nsplit, int M, int N, const float* __restrict__ bias,
                                                                  const bf16* __restrict__ res, int act, bf16* __restrict__ y) {
  const int nq = N >> 2;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)M * nq) return;
  const int m = (int)(i / nq), n = (int)(i - (int64_t)m * nq) * 4;
  const int64_t slab = (int64_t)M * N, off = (int64_t)m * N + n;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int sidx = 0; sidx < nsplit; ++sidx) {
    const float4 v = *reinterpret_cast<const float4*>(ws + sidx * slab + off);
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  float o[4] = {a.x, a.y, a.z, a.w};
  if (bias) {
    const float4 b = *reinterpret_cast<const float4*>(bias + n);
    o[0] += b.x; o[1] += b.y; o[2] += b.z; o[3] += b.w;
  }
  if (res) {
    const bf16x4 r = *reinterpret_cast<const bf16x4*>(res + off);
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] += (float)r[e];
  }
  bf16x4 out;
#pragma unroll
  for (int e = 0; e < 4; ++e) out[e] = (bf16)(act == GPV_ACT_RELU ? fmaxf(o[e], 0.f) : o[e]);
  *reinterpret_cast<bf16x4*>(y + off) = out;
}

// Forward convolutions over a few hundred to a few thousand output pixels (inference at batch 1: layer2-4 see 4800 / 1200 / 300
// pixels): 12..40 tiles of a 128-wide kernel walk K = 1152..4608 alone on a 256-CU chip.  The reduction is split over grid.y
// into fp32 slabs of the caller's workspace (64 x 64 tiles, ~300 workgroups of >= 8 k-tiles), a second launch sums the slabs and
// applies bias / residual / ReLU.  Returns 0 = launched, -1 = not applicable.
inline bool a16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
int conv_split_try_launch(const GemmK& k, int dtype_in, int dtype_out, hipStream_t st) {
  static const int on = tune_env("GPV_CONV_SPLIT", 1);
  if (!on || g_kernel_forced || dtype_in != GPV_BF16 || dtype_out != GPV_BF16 || !k.vecA || !k.vecB || !k.ws_base) return -1;
  if (k.mask || k.rowscale || k.dthresh || k.accumulate || k.cg.dgrad || k.alpha != 1.0f || (k.act != 0 && k.act != GPV_ACT_RELU)) return -1;
  if (k.N % 4 != 0 || k.ldc != k.N || (k.res && k.ldr != k.N) || !a16(k.C) || (k.res && !a16(k.res)) || (k.bias && !a16(k.bias))) return -1;
  const int tiles = ((k.M + 63) / 64) * ((k.N + 63) / 64);
  const int kt_total = (k.K + BK - 1) / BK;
  if (tiles > 160 || kt_total < 32) return -1;
  int split = (384 + tiles - 1) / tiles;
  if (split > kt_total / 8) split = kt_total / 8;
  if (split > 16) split = 16;
  while (split > 1 && (int64_t)split * k.M * k.N * 4 > k.ws_bytes) --split;
  if (split < 2) return -1;
  GemmK p = k;
  p.tilesN = (p.N + 63) / 64;
  p.kt_per_split = (kt_total + split - 1) / split;
  split = (kt_total + p.kt_per_split - 1) / p.kt_per_split;
  p.split_k = split;
  p.ws = reinterpret_cast<float*>(k.ws_base);
  p.res = nullptr; p.bias = nullptr; p.act = 0;
  constexpr size_t lds = (size_t)2 * (64 + 64) * LDK * 2;
  hipLaunchKernelGGL((gemm_kernel<bf16, float, OP_CONV, OP_PLAIN, 64, 64, true>), dim3(tiles, split, 1), dim3(256), lds, st, p);
  GPV_CHECK_LAUNCH();
  const int64_t quads = (int64_t)k.M * (k.N / 4);
  hipLaunchKernelGGL(conv_split_epilogue_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, p.ws, split, k.M, k.N, k.bias,
                     reinterpret_cast<const bf16*>(k.res), k.act, reinterpret_cast<bf16*>(k.C));
  GPV_CHECK_LAUNCH();
  return 0;
}

template <typename TIn, typename TOut, int AMODE, int BMODE>
int launch_tiles(const GemmK& k, int batch, hipStream_t st) {
  const int64_t sk = (k.split_k < 1 ? 1 : k.split_k);
  const int64_t t128 = (int64_t)((k.M + 127) / 128) * ((k.N + 127) / 128) * batch * sk;
  const int64_t t12864 = (int64_t)((k.M + 127) / 128) * ((k.N + 63) / 64) * batch * sk;
  static const int force = tune_env("GPV_FORCE_TILE", 0);   // tuning only
  int cfg;   // 0: 128x128, 1: 128x64, 2: 64x64
  if (force) cfg = force - 1;
  else if (k.K <= 256) cfg = (k.N >= 1024 && t128 >= 384 && !k.conv1x1) ? 0 : 2;   // <= 8 k-tiles: HBM/latency bound, 64x64 keeps more bytes in flight
                                                                      // (tools/bench_c3.py); the wide Linear outputs (FFN expansion) stay at 128x128, the
                                                                      // 1x1 convs since the buffer-load loaders do not: layer3 conv3 66 -> 61, conv1 dgrad 78 -> 65 us
  else if (k.N <= 64 && t12864 >= 384) cfg = 1;                       // N = 64 (layer1): 128x64, measured 100 vs 105 / 144 vs 151 us
  else if (k.N > 64 && t128 >= 384) cfg = 0;
  else if (t12864 >= 384) cfg = 1;
  else cfg = 2;
  if (cfg == 0) return launch_cfg<TIn, TOut, AMODE, BMODE, 128, 128>(k, batch, st);
  if (cfg == 1) return launch_cfg<TIn, TOut, AMODE, BMODE, 128, 64>(k, batch, st);
  return launch_cfg<TIn, TOut, AMODE, BMODE, 64, 64>(k, batch, st);
}

template <int AMODE, int BMODE>
int launch_dtype(const GemmK& k, int batch, int dt_in, int dt_out, hipStream_t st) {
  if (dt_in == GPV_BF16 && dt_out == GPV_BF16) return launch_tiles<bf16, bf16, AMODE, BMODE>(k, batch, st);
#ifndef GPV_EXPERIMENT      /* trimmed instantiation set for kernel-tuning builds */
  if (dt_in == GPV_BF16 && dt_out == GPV_F32) return launch_tiles<bf16, float, AMODE, BMODE>(k, batch, st);
  if (dt_in == GPV_F32 && dt_out == GPV_F32) return launch_tiles<float, float, AMODE, BMODE>(k, batch, st);
#endif
  return (int)hipErrorInvalidValue;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Backward-data of a 1x1 stride-2 convolution (the downsample projections): only output pixels with even row AND column receive a
// product; the other three quarters are dx = res * (mask > 0) (or 0).  The GEMM kernels take the first quarter (parity class 0 of
// the class-major row order); this element-wise kernel streams the rest.  (Routed through the GEMM epilogue -- one dummy k-tile
// per 64/128-row tile, LDS round trip -- those three quarters ran at 2 TB/s and were most of a 245 us launch.)
__global__ __launch_bounds__(256) void s2_dgrad_fill_kernel(bf16* __restrict__ dx, const bf16* __restrict__ res, const bf16* __restrict__ mask,
                                                            int64_t total_chunks, int OW, int OH, int C8) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total_chunks) return;
  const int64_t px = i / C8;
  const int ow = (int)(px % OW), oh = (int)((px / OW) % OH);
  if (((oh | ow) & 1) == 0) return;                    // class 0: written by the GEMM
  bf16x8 o = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
  if (res) {
    o = *reinterpret_cast<const bf16x8*>(res + i * 8);
    if (mask) {
      const bf16x8 m = *reinterpret_cast<const bf16x8*>(mask + i * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (float)m[e] > 0.f ? o[e] : (bf16)0.f;
    }
  }
  *reinterpret_cast<bf16x8*>(dx + i * 8) = o;
}

}  // namespace

extern "C" int gpv_abi_version(void) { return 1; }

extern "C" int gpv_gemm(const gpv_gemm_args* a, void* stream) {
  if (!a || !a->A || !a->B || !a->C || a->M <= 0 || a->N <= 0 || a->K <= 0 || a->batch <= 0) return (int)hipErrorInvalidValue;
  if ((a->split_k > 1 || a->accumulate) && a->dtype_out != GPV_F32) return (int)hipErrorInvalidValue;
  if (a->split_k > 1 && (!a->accumulate || a->bias || a->res || a->act || a->relu_mask || a->drop_p > 0.f)) return (int)hipErrorInvalidValue;
  GemmK k{};
  k.A = a->A; k.B = a->B; k.C = a->C; k.M = a->M; k.N = a->N; k.K = a->K;
  k.lda = a->lda; k.ldb = a->ldb; k.ldc = a->ldc; k.sA = a->sA; k.sB = a->sB; k.sC = a->sC;
  k.alpha = a->alpha; k.rowscale = a->rowscale; k.bias = a->bias;
  k.res = a->res; k.ldr = a->ldr; k.sR = a->sR; k.mask = a->relu_mask; k.ldm = a->ldm;
  k.act = a->act; k.seed = a->seed; k.seed_dev = gpvk::g_seed_dev;
  k.dthresh = a->drop_p > 0.f ? drop_thresh(a->drop_p) : 0u;
  k.dscale = a->drop_p > 0.f ? 1.0f / (1.0f - a->drop_p) : 1.0f;
  k.accumulate = a->accumulate; k.split_k = a->split_k;
  k.a_rowsum = a->a_rowsum;
  k.no_small = (a->flags & GPV_GEMM_NO_PIPE_SMALL) ? 1 : 0;
  k.ws_base = a->workspace; k.ws_bytes = a->workspace ? a->workspace_bytes : 0;
  if (k.a_rowsum && !(a->layoutA == GPV_TRANS && a->layoutB == GPV_TRANS && a->batch == 1)) return (int)hipErrorInvalidValue;
  if (a->accumulate && a->split_k <= 1 && !a->res) {
    // one block owns every output element: C += acc as a coalesced read-modify-write through the LDS epilogue
    // (fp32 atomics run at ~70 G/s: a 10000x768 gradient costs 110 us in atomics alone)
    k.accumulate = 0; k.res = a->C; k.ldr = a->ldc; k.sR = a->sC;
  }
  const int esz = a->dtype_in == GPV_F32 ? 4 : 2;
  const int64_t vecel = 16 / esz;  // elements per 16 B
  auto vec_ok = [&](const void* ptr, int64_t ld, int64_t bs, int layout, int extent_contig) {
    bool ok = aligned16(ptr) && (ld % vecel == 0) && (a->batch == 1 || bs % vecel == 0);
    // 16-byte chunks: the contiguous extent must be a multiple of 8, or end inside the row pitch -- a reduction-major
    // operand's last chunk then only feeds outputs beyond M / N (discarded); a k-major operand's last chunk multiplies
    // masked rows of the other operand, so its padding must be finite (caller's promise: GPV_GEMM_KPAD_FINITE)
    if (layout == GPV_KMAJOR) ok = ok && (a->K % 8 == 0 || ((a->flags & GPV_GEMM_KPAD_FINITE) && ld >= ((int64_t)a->K + 7) / 8 * 8));
    else ok = ok && (extent_contig % 8 == 0 || ld >= ((int64_t)extent_contig + 7) / 8 * 8);
    return ok ? 1 : 0;
  };
  k.vecA = vec_ok(a->A, a->lda, a->sA, a->layoutA, a->M);
  k.vecB = vec_ok(a->B, a->ldb, a->sB, a->layoutB, a->N);
  {   // the 16-byte path addresses an operand (one batch element) with 32-bit byte offsets
    const int64_t lim = 0x7ffffff0ll / esz;
    if ((int64_t)(a->layoutA == GPV_KMAJOR ? a->M : a->K) * a->lda >= lim) k.vecA = 0;
    if ((int64_t)(a->layoutB == GPV_KMAJOR ? a->N : a->K) * a->ldb >= lim) k.vecB = 0;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int la = a->layoutA, lb = a->layoutB;
  if (la == GPV_KMAJOR && lb == GPV_KMAJOR) {
    const int gv = gemv_try_launch(k, a->dtype_in, a->dtype_out, a->batch, st);                // M <= 8: matrix-vector products of the decode step
    if (gv >= 0) return gv;
    static const bool c1s_linear = tune_env("GPV_C1S_LINEAR", 1) != 0;     // 0: A/B only
    if (a->batch == 1 && !g_kernel_forced && c1s_linear) {
      const int cs = c1s_try_launch(k, a->dtype_in, a->dtype_out, st, true);               // K = 256 -> >= 1024 features: weights resident in LDS, rows streamed
      if (cs >= 0) return cs;
    }
    const int pp = pipe_try_launch(k, OP_PLAIN, a->dtype_in, a->dtype_out, a->batch, st);     // pipelined direct-to-LDS kernel (big tiles, and the small-M 6-8 stage tiles)
    if (pp >= 0) return pp;
    const int sk = skinny_try_launch(k, 0, a->dtype_in, a->dtype_out, a->batch, st);             // few tiles, long reduction: what the pipelined kernel does not take (fp32 outputs, ragged N)
    if (sk >= 0) return sk;
    const int g = glds_try_launch(k, OP_PLAIN, a->dtype_in, a->dtype_out, a->batch, st);      // 8-wave direct-to-LDS kernel when it qualifies
    if (g >= 0) return g;
    return launch_dtype<OP_PLAIN, OP_PLAIN>(k, a->batch, a->dtype_in, a->dtype_out, st);
  }
  if (la == GPV_KMAJOR && lb == GPV_TRANS) {
    const int sk = skinny_try_launch(k, 1, a->dtype_in, a->dtype_out, a->batch, st);             // small backward-data GEMMs
    if (sk >= 0) return sk;
    return launch_dtype<OP_PLAIN, OP_TRANS>(k, a->batch, a->dtype_in, a->dtype_out, st);
  }
  if (la == GPV_TRANS && lb == GPV_TRANS) {
    if (a->accumulate) {                                  // small weight gradients: reduction split over the block's waves
      GemmK kk = k;
      kk.accumulate = 1; kk.res = nullptr; kk.ldr = 0; kk.sR = 0;      // (the split==1 rewrite above turned C += into res = C)
      const int gt = glds_tt_try_launch(kk, a->dtype_in, a->dtype_out, a->batch, st);        // 128-multiples, long reduction
      if (gt >= 0) return gt;
      const int sk = skinny_tt_try_launch(kk, a->dtype_in, a->dtype_out, a->batch, st);
      if (sk >= 0) return sk;
    }
    return launch_dtype<OP_TRANS, OP_TRANS>(k, a->batch, a->dtype_in, a->dtype_out, st);
  }
  return (int)hipErrorInvalidValue;
}

// dry: check only (gpv_conv2d_mask_bits_ok) -- valid for calls that ask for mask bits, which return before any other launch path
static int conv2d_impl(const gpv_conv_args* a, hipStream_t st, bool dry) {
  if (!a || !a->x || !a->w || !a->y) return (int)hipErrorInvalidValue;
  if (a->Cin % 32 != 0) return (int)hipErrorInvalidValue;
  if ((a->SH != 1 && a->SH != 2) || (a->SW != 1 && a->SW != 2)) return (int)hipErrorInvalidValue;
  const bool want_bits = a->y_mask_bits != nullptr || a->relu_mask_bits != nullptr;
  if (want_bits) {
    // one-bit ReLU masks: the streaming 1x1 kernel only (stride 1, pointwise), forward with ReLU writes them, backward-data reads them
    const bool pw1 = a->KH == 1 && a->KW == 1 && a->SH == 1 && a->SW == 1 && a->PH == 0 && a->PW == 0 && a->IH == a->OH && a->IW == a->OW;
    const bool d3s2 = a->mode == 1 && a->KH == 3 && a->KW == 3 && a->SH == 2 && a->SW == 2 && a->PH == 1 && a->PW == 1 && a->Cout == 128 && !a->y_mask_bits;
    if (!pw1 && !d3s2) return (int)hipErrorInvalidValue;
    if ((pw1 && a->Cout % 256 != 0) || (a->y_mask_bits && (a->mode != 0 || a->act != GPV_ACT_RELU || a->relu_mask)) || (a->relu_mask_bits && a->mode != 1)) return (int)hipErrorInvalidValue;
    if ((reinterpret_cast<uintptr_t>(a->y_mask_bits) | reinterpret_cast<uintptr_t>(a->relu_mask_bits)) & 15) return (int)hipErrorInvalidValue;
  } else if (dry) {
    return (int)hipErrorInvalidValue;
  }
  GemmK k{};
  k.alpha = 1.0f; k.rowscale = a->rowscale; k.bias = a->bias; k.act = a->act;
  k.cg = ConvGeom{a->IH, a->IW, a->Cs, a->Cin, a->OH, a->OW, a->KH, a->KW, a->SH, a->SW, a->PH, a->PW, a->mode == 1, 0, 0};
  if (a->mode == 1 && a->SH == 2 && a->SW == 2 && a->OH % 2 == 0 && a->OW % 2 == 0) {
    // stride-2 dgrad: only taps with r = (ih+PH) mod 2, s = (iw+PW) mod 2 contribute.  Order the rows by pixel parity
    // class so that a tile is class-uniform and skips the other taps (2.25 of 9 on average for a 3x3, 1 of 4 pixels for a 1x1)
    k.cg.cm = 1;
    k.cg.cls_rows = a->B * (a->OH / 2) * (a->OW / 2);
  }
  const int T = a->KH * a->KW;
  const int esz = a->dtype_in == GPV_F32 ? 4 : 2;
  const int64_t vecel = 16 / esz;
  if (a->mode == 0 || a->mode == 1) {
    // rows = B*OH*OW output positions, N = Cout, K = T*Cin ; A = gather(x), B = w [Cout][T*Cin]
    k.A = a->x; k.B = a->w; k.C = a->y;
    k.M = a->B * a->OH * a->OW; k.N = a->Cout; k.K = T * a->Cin;
    k.lda = 0; k.ldb = k.K; k.ldc = a->Cout;
    k.res = a->res; k.ldr = a->Cout; k.mask = a->relu_mask; k.ldm = a->Cout;
    k.vecA = aligned16(a->x) && (a->Cs % vecel == 0 || (a->Cs * esz) % 8 == 0) ? 1 : 0;
    // the stem reads 16-B runs at pixel granularity (Cs = 4 bf16 = 8 B): require 16-B aligned runs
    if ((a->Cs % vecel) != 0) k.vecA = ((a->SW * a->Cs) % vecel == 0 && (a->IW * a->Cs) % vecel == 0 && a->PW == 0) ? k.vecA : 0;
    k.vecB = aligned16(a->w) && (k.K % 8 == 0) ? 1 : 0;
    if (a->mode == 0 && a->KH == 1 && a->KW == 1 && a->SH == 2 && a->SW == 2 && a->PH == 0 && a->PW == 0 &&
        aligned16(a->x) && a->Cs % vecel == 0 && (int64_t)a->B * a->IH * a->IW * a->Cs * esz < 0x7ffffff0ll * 4ll) {
      // stride-2 pointwise projection (the downsample branches), forward: the streaming kernel reads every other pixel of every
      // other row (conv1x1_stream.hip); anything it does not take goes down the generic gather path below
      GemmK ks = k;
      ks.lda = a->Cs; ks.vecA = 1;
      const int cs = c1s_try_launch(ks, a->dtype_in, a->dtype_out, st);
      if (cs >= 0) return cs;
    }
    if (a->KH == 1 && a->KW == 1 && a->SH == 1 && a->SW == 1 && a->PH == 0 && a->PW == 0 && a->IH == a->OH && a->IW == a->OW) {
      // a 1x1 stride-1 convolution IS a GEMM over the pixel rows (row pitch Cs): no tap / pixel decoding per thread
      // (three integer divisions per staged row -- a third of the instructions of a K = 64 tile), and the GEMM-side
      // kernel choices apply
      k.lda = a->Cs;
      k.cg = ConvGeom{};
      k.cg.SH = k.cg.SW = 1;
      k.conv1x1 = 1;
      {
        // outputs far beyond the 256 MB MALL (layer1's 314 MB maps) are stored -- and their residual read -- non-temporally:
        // 64 -> 256 without a residual 150 -> 106 us, with one 177 -> 168 us; smaller outputs are better left cacheable for
        // the next convolution (layer3 conv3, 79 MB: 58 -> 72 us with non-temporal stores)   [tools/bench_c1.py]
        static const int64_t nt_min = (int64_t)tune_env("GPV_NT_MIN_MB", 200) << 20;
        k.nt_io = (int64_t)k.M * k.N * esz >= nt_min ? 1 : 0;
      }
      k.vecA = aligned16(a->x) && (a->Cs % vecel == 0) && (int64_t)k.M * a->Cs * esz < 0x7ffffff0ll ? 1 : 0;
      if (want_bits) {
        if (!k.vecA) return (int)hipErrorInvalidValue;
        k.out_bits = reinterpret_cast<uint32_t*>(a->y_mask_bits);
        k.mask_bits = reinterpret_cast<const uint32_t*>(a->relu_mask_bits);
        if (k.mask_bits) { k.mask = a->relu_mask_bits; k.ldm = a->Cout; }      // (selects the masked instances; never dereferenced as bf16)
        const int cs = c1s_try_launch(k, a->dtype_in, a->dtype_out, st, false, dry);
        return cs >= 0 ? cs : (int)hipErrorInvalidValue;
      }
      if (k.vecA) {
        const int cs = c1s_try_launch(k, a->dtype_in, a->dtype_out, st);
        if (cs >= 0) return cs;
      }
      const int pp = pipe_try_launch(k, OP_PLAIN, a->dtype_in, a->dtype_out, 1, st);
      if (pp >= 0) return pp;
      if (k.M <= 8192 && !g_kernel_forced) {               // a few thousand pixels (inference at batch 1): 64x64 tiles, reduction split over the block's waves
        const int sk = skinny_try_launch(k, 0, a->dtype_in, a->dtype_out, 1, st);
        if (sk >= 0) return sk;
      }
      const int g = glds_try_launch(k, OP_PLAIN, a->dtype_in, a->dtype_out, 1, st);
      if (g >= 0) return g;
      return launch_dtype<OP_PLAIN, OP_PLAIN>(k, 1, a->dtype_in, a->dtype_out, st);
    }
    static const bool s2_split = tune_env("GPV_S2_DGRAD_SPLIT", 1) != 0;
    if (s2_split && a->mode == 1 && k.cg.cm && a->KH == 1 && a->KW == 1 && a->PH == 0 && a->PW == 0 && a->dtype_in == GPV_BF16 &&
        a->dtype_out == GPV_BF16 && a->Cout % 8 == 0 && aligned16(a->y) && (!a->res || aligned16(a->res)) &&
        (!a->relu_mask || aligned16(a->relu_mask)) && k.vecA && k.vecB) {
      // pointwise stride-2 backward-data: class 0 (even row, even column) through the GEMM kernels, the rest element-wise
      GemmK k0 = k;
      k0.M = k.cg.cls_rows;
      int e = pipe_try_launch(k0, OP_CONV, a->dtype_in, a->dtype_out, 1, st);
      if (e < 0) e = glds_try_launch(k0, OP_CONV, a->dtype_in, a->dtype_out, 1, st);
      if (e < 0) e = launch_dtype<OP_CONV, OP_PLAIN>(k0, 1, a->dtype_in, a->dtype_out, st);
      if (e) return e;
      const int C8 = a->Cout / 8;
      const int64_t chunks = (int64_t)a->B * a->OH * a->OW * C8;
      hipLaunchKernelGGL(s2_dgrad_fill_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, st, reinterpret_cast<bf16*>(a->y),
                         reinterpret_cast<const bf16*>(a->res), reinterpret_cast<const bf16*>(a->relu_mask), chunks, a->OW, a->OH, C8);
      GPV_CHECK_LAUNCH();
      return 0;
    }
    if (want_bits) {
      // (not the pointwise form, which returned above: the stride-2 3x3 backward-data over 128 channels on the streaming kernel, or nothing)
      if (!(a->KH == 3 && a->KW == 3 && k.vecA && k.vecB)) return (int)hipErrorInvalidValue;
      k.mask_bits = reinterpret_cast<const uint32_t*>(a->relu_mask_bits);
      k.mask = a->relu_mask_bits; k.ldm = a->Cout;
      const int c3 = c3s_try_launch(k, a->dtype_in, a->dtype_out, st, dry);
      return c3 >= 0 ? c3 : (int)hipErrorInvalidValue;
    }
    if (a->KH == 3 && a->KW == 3 && k.vecA && k.vecB) {
      // 3x3 with 64 / 128 input channels over the layer1 / layer2 maps: weights resident in LDS, barrier-free streaming kernel
      const int c3 = c3s_try_launch(k, a->dtype_in, a->dtype_out, st);
      if (c3 >= 0) return c3;
    }
    if (k.vecA && k.vecB) {
      if (a->mode == 0 && a->workspace) {                  // few output pixels, long reduction (inference at batch 1): split + second pass
        GemmK ks = k;
        ks.ws_base = a->workspace; ks.ws_bytes = a->workspace_bytes;
        const int cs = conv_split_try_launch(ks, a->dtype_in, a->dtype_out, st);
        if (cs >= 0) return cs;
      }
      const int pp = pipe_try_launch(k, OP_CONV, a->dtype_in, a->dtype_out, 1, st);
      if (pp >= 0) return pp;
      const int g = glds_try_launch(k, OP_CONV, a->dtype_in, a->dtype_out, 1, st);
      if (g >= 0) return g;
    }
    return launch_dtype<OP_CONV, OP_PLAIN>(k, 1, a->dtype_in, a->dtype_out, st);
  }
  if (a->mode == 2) {
    // dw[Cout][T*Cin] += dy^T x_gather : M = Cout, N = T*Cin, K = B*OH*OW
    if (a->dtype_out != GPV_F32) return (int)hipErrorInvalidValue;
    if (a->Cin % 64 != 0) return (int)hipErrorInvalidValue;
    k.A = a->w /* dy */; k.B = a->x; k.C = a->y /* dw */;
    k.M = a->Cout; k.N = T * a->Cin; k.K = a->B * a->OH * a->OW;
    k.lda = a->Cout; k.ldb = 0; k.ldc = k.N;
    k.accumulate = 1;
    int split = a->split_k;
    if (split < 1) {
      const int kt = (k.K + BK - 1) / BK;
      int64_t want;
      if (a->workspace && a->Cin % 128 == 0 && k.M >= 128) {
        // two-pass reduction (no atomics): ~2 blocks of 128x128 per CU is the measured optimum on the layer2-4 shapes
        // (tools/bench_split_conv.py: 64..96 / 64 / 32 / 16 / 8 / 4 splits for 4 / 9 / 16 / 36 / 64 / 144 tiles)
        const int64_t t128 = (int64_t)((k.M + 127) / 128) * (k.N / 128);
        want = (544 + t128 / 2) / t128;
        if (want > 288) want = 288;
        while (want > 1 && want * (int64_t)k.M * k.N * 4 > a->workspace_bytes) --want;
      } else {
        const int64_t tiles = (int64_t)((k.M + 63) / 64) * ((k.N + 63) / 64);
        want = 1536 / (tiles > 0 ? tiles : 1);
      }
      if (want > kt / 8) want = kt / 8;
      split = want < 1 ? 1 : (int)want;
    }
    k.split_k = split;
    if (split <= 1) { k.accumulate = 0; k.res = a->y; k.ldr = k.ldc; }
    k.ws_base = a->workspace; k.ws_bytes = a->workspace ? a->workspace_bytes : 0;
    k.vecA = aligned16(a->w) && (a->Cout % vecel == 0) ? 1 : 0;
    k.vecB = aligned16(a->x) && (a->Cs % vecel == 0) ? 1 : 0;
    // CONVT needs every N tile inside one tap: BN divides Cin (64 always does here; 128 when Cin % 128 == 0)
    {   // direct-to-LDS kernel (gemm_glds_tt.hip): 128-multiples of Cout / Cin with a workspace split reduction
      const int gw = glds_wgrad_try_launch(k, a->dtype_in, a->dtype_out, st);
      if (gw >= 0) return gw;
    }
    static const int wforce = tune_env("GPV_FORCE_WGRAD_TILE", 0);   // tuning only
    if (a->dtype_in == GPV_BF16) {
      const bool can128 = (a->Cin % 128 == 0) && k.M >= 128;
      int cfg;   // 0: 128x128, 1: 128x64, 2: 64x64
      if (wforce) cfg = wforce - 1;
      else if (can128 && (int64_t)((k.M + 127) / 128) * (k.N / 128) * split >= 256) cfg = 0;
      else if (k.M >= 128 && (int64_t)((k.M + 127) / 128) * (k.N / 64) * split >= 384) cfg = 1;
      else cfg = 2;
      if (cfg == 0 && can128) return launch_cfg<bf16, float, OP_TRANS, OP_CONV, 128, 128>(k, 1, st);
      if (cfg <= 1 && k.M >= 128) return launch_cfg<bf16, float, OP_TRANS, OP_CONV, 128, 64>(k, 1, st);
      return launch_cfg<bf16, float, OP_TRANS, OP_CONV, 64, 64>(k, 1, st);
    }
    return launch_cfg<float, float, OP_TRANS, OP_CONV, 64, 64>(k, 1, st);
  }
  return (int)hipErrorInvalidValue;
}

extern "C" int gpv_conv2d(const gpv_conv_args* a, void* stream) { return conv2d_impl(a, reinterpret_cast<hipStream_t>(stream), false); }

extern "C" int gpv_conv2d_mask_bits_ok(const gpv_conv_args* a) { return conv2d_impl(a, nullptr, true) == 0 ? 1 : 0; }
