// Small-M GEMM with the reduction split across the four waves of a workgroup (gfx950).
//
//   C[M,N] = epilogue( A[M,K] x B[N,K]^T ),  bf16, both operands K-major, 64x64 output tile per 256-thread block.
//
// The frozen BERT encoder runs at M = B*T_l = 192 rows and the text decoder at 640: 36..144 tiles of 64x64 cannot fill
// 256 CUs, so gemm.hip's kernel spends its time in the serial k-loop of a few resident blocks (K = 768: 24 steps of 32,
// ~0.5 us each: 11.9 us; K = 3072: 34 us).  Here one k-step is 128 deep: every wave multiplies the WHOLE 64x64 tile
// over its own 32-deep quarter (4x fewer serial steps, 16 MFMAs per wave and step), and the four partial tiles are
// summed through LDS once at the end.  Same epilogue contract as gpv_gemm (alpha, rowscale, bias, residual, ReLU/GELU,
// dropout, ReLU mask).
#include "gemm_common.h"

namespace gpvk {

int g_skinny_mode = 1;        // gpv_set_option(GPV_OPT_SKINNY, .): 0 never, 1 heuristic, 2 wherever legal

namespace {

constexpr int SBM = 64, SBN = 64, SBK = 128, SPITCH = SBK + 8;     // LDS row pitch 272 B: 16 rows cover all 64 banks
constexpr int TPITCH = SBN + 16;      // reduction-major B tile [128 red][64 cols]: pitch 160 B (the transpose-read rule of gemm.hip)

// fragment of a reduction-major tile for the 16 columns starting at cbase: lane (col = lane&15, g = lane>>4) receives the
// reduction rows 8g..8g+7 of its column (two ds_read_b64_tr_b16, see gemm.hip TStage::frag)
__device__ __forceinline__ bf16x8 tr_frag(const bf16* tile, int cbase, int lane) {
  typedef short __attribute__((ext_vector_type(4))) s16x4;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const int g = lane >> 4, i = lane & 15;
  const bf16* p0 = tile + (8 * g + (i >> 2)) * TPITCH + cbase + (i & 3) * 4;
  s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0));
  s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0 + 4 * TPITCH));
  typedef short __attribute__((ext_vector_type(8))) s16x8;
  s16x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, r);
}

// PF = k-steps of global loads in flight per thread (register ring), BM x BN = output tile (64 | 32 each; reduction-major B: BN = 64).
// Round 5, measured (tools/bench_skinny_pf.py): a k-step of the 64 x 64 tile costs 0.75 - 0.85 us WHATEVER the prefetch depth
// (300 x 256 x 2048: 16.2 / 14.3 / 15.2 / 13.7 us at PF = 1 / 2 / 3 / 4) -- the 32 KB a step pulls through ONE CU arrive at
// ~ 18 B / clk, the per-CU operand delivery rate every GEMM in this tree meets; the loop was never latency-bound.  What shortens it
// is fewer bytes per CU: with 20 - 120 tiles of 64 x 64 on 256 CUs, 32-row / 32-column tiles put 2 - 4 x as many CUs on the problem and
// each pulls (BM + BN) / 128 of the bytes (the extra L2 reads are free here).  PF = 2 is kept: the loop is branch-free -- the
// step count is rounded up to a multiple of PF and steps beyond the reduction fetch through out-of-range offsets (hardware zeros, no
// memory traffic) -- so that hipcc's counted vmcnt keeps PF - 1 steps in flight across every barrier.
template <typename TOut, bool BT, int PF, int BM, int BN>
__global__ __launch_bounds__(256) void skinny_kernel(GemmK p) {
  static_assert(!BT || BN == 64, "reduction-major B tiles are [128][64]");
  if (p.dthresh) p.seed = eff_seed(p.seed, p.seed_dev);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16* lds = reinterpret_cast<bf16*>(smem);
  constexpr int FM = BM / 16, FN = BN / 16;           // MFMA tiles per wave (every wave multiplies the WHOLE tile over its k-quarter)
  constexpr int ATILE = BM * SPITCH;                  // elements of a K-major operand tile
  constexpr int BTILE = BT ? SBK * TPITCH : BN * SPITCH;      // reduction-major B tile is [128][80]
  constexpr int STAGE = ATILE + BTILE;
  constexpr int NB = BT ? 4 : FN;                     // loads per thread and step for B
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tn = blockIdx.x % p.tilesN, tm = blockIdx.x / p.tilesN;
  const int row0 = tm * BM, col0 = tn * BN;
  const bf16* A = reinterpret_cast<const bf16*>(p.A);
  const bf16* B = reinterpret_cast<const bf16*>(p.B);
  const int nk = (p.K + SBK - 1) / SBK;

  // loader: rows x 16 chunks of 16 B per operand -> BM / 16 (BN / 16) per thread (same chunk column, rows +16)
  const int lc = tid & 15, lr = tid >> 4;
  // operands through buffer loads: per-thread 32-bit byte offsets fixed for the tile, the k-step offset in the scalar soffset,
  // an out-of-range offset (reduction tail, columns beyond N) = hardware zeros -- no 64-bit address add, no safe-address
  // select and no zeroing selects per load (this loader issued ~64 VALU instructions per 16 MFMAs)
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  constexpr int OOB = 0x7ffffff0;
  const auto rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(A), (short)0, OOB, 0x00020000);
  const auto rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(B), (short)0, OOB, 0x00020000);
  int avo[FM], bvo[NB];
  // reduction-major B (BT): tile rows are reduction indices, a row is 64 output columns = 8 chunks -> thread (row tr + 32 i, chunk tc)
  const int tc = tid & 7, tr = tid >> 3;
  const bool bcol_ok = col0 + tc * 8 < p.N;            // N % 8 == 0 (host)
#pragma unroll
  for (int i = 0; i < FM; ++i) avo[i] = (min(row0 + lr + 16 * i, p.M - 1) * (int)p.lda + lc * 8) * 2;
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    if constexpr (BT) bvo[i] = bcol_ok ? ((tr + 32 * i) * (int)p.ldb + col0 + tc * 8) * 2 : OOB;
    else bvo[i] = (min(col0 + lr + 16 * i, p.N - 1) * (int)p.ldb + lc * 8) * 2;
  }
  uint4 ra[PF][FM], rb[PF][NB];
  auto bl = [&](const decltype(rsA)& rs, int vo, int so) {
    const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so, 0);
    return make_uint4(t[0], t[1], t[2], t[3]);
  };
  auto load = [&](int kt, uint4 (&qa)[FM], uint4 (&qb)[NB]) {
    const bool in = kt < nk;                           // uniform; beyond the reduction: zeros through the descriptor
    const bool full = in && (kt + 1) * SBK <= p.K;
    const bool ok = full || (in && kt * SBK + lc * 8 < p.K);   // K % 8 == 0 (host): a chunk is entirely inside or outside
    const int so = in ? kt * SBK * 2 : 0;
#pragma unroll
    for (int i = 0; i < FM; ++i) qa[i] = bl(rsA, ok ? avo[i] : OOB, so);
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      if constexpr (BT) qb[i] = bl(rsB, (full || (in && kt * SBK + tr + 32 * i < p.K)) ? bvo[i] : OOB, so * (int)p.ldb);
      else qb[i] = bl(rsB, ok ? bvo[i] : OOB, so);
    }
  };
  auto store = [&](int stage, const uint4 (&qa)[FM], const uint4 (&qb)[NB]) {
    bf16* sa = lds + stage * STAGE;
    bf16* sb = sa + ATILE;
#pragma unroll
    for (int i = 0; i < FM; ++i) *reinterpret_cast<uint4*>(sa + (lr + 16 * i) * SPITCH + lc * 8) = qa[i];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      if constexpr (BT) *reinterpret_cast<uint4*>(sb + (tr + 32 * i) * TPITCH + tc * 8) = qb[i];
      else *reinterpret_cast<uint4*>(sb + (lr + 16 * i) * SPITCH + lc * 8) = qb[i];
    }
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int foff = (lane & 15) * SPITCH + wave * 32 + (lane >> 4) * 8;      // this wave's quarter of the k-step

#pragma unroll
  for (int s = 0; s < PF; ++s) load(s, ra[s], rb[s]);
  store(0, ra[0], rb[0]);
  __syncthreads();
  const int nkp = (nk + PF - 1) / PF * PF;
  for (int t0 = 0; t0 < nkp; t0 += PF) {
#pragma unroll
    for (int s = 0; s < PF; ++s) {
      const int t = t0 + s;
      load(t + PF, ra[s], rb[s]);                      // slot s held tile t: in LDS since the previous step
      const bf16* sa = lds + (t & 1) * STAGE;
      const bf16* sb = sa + ATILE;
      bf16x8 af[FM], bfr[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(sa + i * 16 * SPITCH + foff);
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        if constexpr (BT) bfr[j] = tr_frag(sb + wave * 32 * TPITCH, j * 16, lane);
        else bfr[j] = *reinterpret_cast<const bf16x8*>(sb + j * 16 * SPITCH + foff);
      }
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = mfma16(bfr[j], af[i], acc[i][j]);    // swapped: lane holds 4 consecutive columns
      store((t + 1) & 1, ra[(s + 1) % PF], rb[(s + 1) % PF]);                           // tile t + 1 (PF - 1 younger steps stay in flight)
      __syncthreads();
    }
  }

  // ---- sum the four waves' partial tiles through LDS (fragment-native order: every access is lane-contiguous) ----
  float* red = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
      *reinterpret_cast<f32x4*>(red + wave * (BM * BN) + (i * FN + j) * 256 + lane * 4) = acc[i][j];
  __syncthreads();

  TOut* Cp = reinterpret_cast<TOut*>(p.C);
  const TOut* Rp = reinterpret_cast<const TOut*>(p.res);
  const TOut* Mp = reinterpret_cast<const TOut*>(p.mask);
#pragma unroll
  for (int r = 0; r < (FM * FN) / 4; ++r) {
    const int q = tid + 256 * r;
    const int ij = q >> 6, l = q & 63;
    f32x4 v = *reinterpret_cast<const f32x4*>(red + ij * 256 + l * 4);
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const f32x4 u = *reinterpret_cast<const f32x4*>(red + w * (BM * BN) + ij * 256 + l * 4);
      v[0] += u[0]; v[1] += u[1]; v[2] += u[2]; v[3] += u[3];
    }
    const int m = row0 + (ij / FN) * 16 + (l & 15);
    const int n = col0 + (ij % FN) * 16 + (l >> 4) * 4;
    if (m >= p.M || n >= p.N) continue;
    const float rs = p.rowscale ? p.rowscale[m] * p.alpha : p.alpha;
    // the 4 columns' bias / residual / mask as one vector load each when they are whole and aligned (per-element global
    // loads in an epilogue loop are what made the LayerNorm forward 2x too slow)
    const bool full = n + 4 <= p.N;
    float bv[4] = {0.f, 0.f, 0.f, 0.f}, rv[4] = {0.f, 0.f, 0.f, 0.f}, mv[4] = {1.f, 1.f, 1.f, 1.f};
    if (p.bias) {
      if (full && (reinterpret_cast<uintptr_t>(p.bias) & 15) == 0) {
        const float4 b4 = *reinterpret_cast<const float4*>(p.bias + n);
        bv[0] = b4.x; bv[1] = b4.y; bv[2] = b4.z; bv[3] = b4.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[e] = n + e < p.N ? p.bias[n + e] : 0.f;
      }
    }
    auto ld4 = [&](const TOut* base, int64_t ld, float* out) {
      const TOut* q = base + (int64_t)m * ld + n;
      if (full && (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(base) & (4 * sizeof(TOut) - 1)) == 0) {
        if constexpr (sizeof(TOut) == 2) {
          const bf16x4 t = *reinterpret_cast<const bf16x4*>(q);
#pragma unroll
          for (int e = 0; e < 4; ++e) out[e] = (float)t[e];
        } else {
          const float4 t = *reinterpret_cast<const float4*>(q);
          out[0] = t.x; out[1] = t.y; out[2] = t.z; out[3] = t.w;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) out[e] = n + e < p.N ? (float)q[e] : 0.f;
      }
    };
    if (Rp) ld4(Rp, p.ldr, rv);
    if (Mp) ld4(Mp, p.ldm, mv);
    float o[4];
    const uint32_t keep4 = p.dthresh ? drop_mask<4>(p.seed, (uint64_t)m * (uint64_t)p.N + n, p.dthresh) : 0xfu;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (n + e >= p.N) { o[e] = 0.f; continue; }
      float x = v[e] * rs + bv[e];
      if (Rp) x += rv[e];
      if (p.act == GPV_ACT_RELU) x = fmaxf(x, 0.f);
      else if (p.act == GPV_ACT_GELU) x = gelu_erf(x);
      if (p.dthresh) x = ((keep4 >> e) & 1u) ? x * p.dscale : 0.f;
      if (Mp) x = mv[e] > 0.f ? x : 0.f;
      o[e] = x;
    }
    TOut* dst = Cp + (int64_t)m * p.ldc + n;
    if (n + 4 <= p.N && (p.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(Cp) & 7) == 0) {
      if constexpr (sizeof(TOut) == 2) {
        bf16x4 o4;
#pragma unroll
        for (int e = 0; e < 4; ++e) o4[e] = (bf16)o[e];
        *reinterpret_cast<bf16x4*>(dst) = o4;
      } else {
        if ((reinterpret_cast<uintptr_t>(Cp) & 15) == 0) *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
        else { dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2]; dst[3] = o[3]; }
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (n + e < p.N) dst[e] = (TOut)o[e];
    }
  }
}

int g_skinny_tile = tune_env("GPV_SKINNY_TILE", 0);     // tuning build: 0 = by cost, 1 = 64 x 64, 2 = 32 x 64, 3 = 32 x 32

template <typename TOut, bool BT, int BM, int BN>
int launch_skinny_tile(const GemmK& k, hipStream_t st) {
  constexpr size_t stage = (size_t)2 * (BM * SPITCH + (BT ? SBK * TPITCH : BN * SPITCH)) * 2;      // two stages x (A, B) tiles
  constexpr size_t redb = (size_t)4 * BM * BN * 4;
  constexpr size_t lds = stage > redb ? stage : redb;
  GemmK p = k;
  p.tilesN = (p.N + BN - 1) / BN;
  const int tilesM = (p.M + BM - 1) / BM;
  const int nk = (p.K + SBK - 1) / SBK;
  auto fn = nk >= 2 ? skinny_kernel<TOut, BT, 2, BM, BN> : skinny_kernel<TOut, BT, 1, BM, BN>;
  static bool attr_done = false;
  if (!attr_done) {
    for (auto f : {skinny_kernel<TOut, BT, 1, BM, BN>, skinny_kernel<TOut, BT, 2, BM, BN>}) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(f), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }      // (leave no stale error behind for the next launch check)
    }
    attr_done = true;
  }
  hipLaunchKernelGGL(fn, dim3(tilesM * p.tilesN), dim3(256), lds, st, p);
  GPV_CHECK_LAUNCH();
  return 0;
}

// Tile by measurement (tools/bench_skinny_pf.py, tools/ab_skinny_pf.sh; us at 64 x 64 | 32 x 64 | 32 x 32):
//   300 x 256 x 2048 (20 tiles of 64 x 64) 14.2 | 10.4 | 7.8     100 x 768 x 3072 (24) 18.3 | 13.3 | 9.9     192 x 768 x 3072 (36) 18.3 | 13.2 | 11.3
//   640 x 768 x 768 (120) 8.6 | 9.5 | 8.4     640 x 768 x 2048 (120) 15.3 | 15.6 | 15.3 (126 MB through L2 at 32 x 32: the L2s bound it)
//   640 x 2304 x 768 (360) 12.5 | 11.0 | 12.3     640 x 2048 x 768 (320) 12.0 | 10.6 | 10.8     192 x 3072 x 768 (144) 8.4 | 9.1 | 8.4
//   reduction-major B: 640 x 768 x 2048 16.8 | 12.6     640 x 768 x 2304 18.1 | 13.6     3200 x 256 x 2048 (200) 20.1 | 16.3
// i.e. up to ~ 160 tiles of 64 x 64 the smallest tile (4 x as many CUs, half the bytes each), above that 32 x 64 (two co-resident blocks
// per CU); batch-1 inference 4.88 -> 4.5 - 4.7 ms, train step -0.2 ms (same box, alternating twice).
template <typename TOut, bool BT>
int launch_skinny(const GemmK& k, hipStream_t st) {
  int pick = g_skinny_tile;
  if (pick == 0) {
    const int64_t t64 = (int64_t)((k.M + 63) / 64) * ((k.N + 63) / 64);
    pick = (!BT && t64 <= 160) ? 3 : 2;
  }
  if (pick == 2) return launch_skinny_tile<TOut, BT, 32, 64>(k, st);
  if constexpr (!BT) { if (pick == 3) return launch_skinny_tile<TOut, BT, 32, 32>(k, st); }
  return launch_skinny_tile<TOut, BT, 64, 64>(k, st);
}

// ---- weight-gradient form: C[M,N] (+)= A^T B with A [K][M], B [K][N] both reduction-major (dW = dY^T X) -------------
// grid = (tiles, splits): split s covers k-steps [s*per, (s+1)*per) of 128 and writes its partial product to the
// workspace slab s (or, with one split, adds straight into C); a_rowsum: column sums of the A tile (bias gradient),
// accumulated by the loader threads of the tn == 0 blocks.
__global__ __launch_bounds__(256) void skinny_tt_kernel(GemmK p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16* lds = reinterpret_cast<bf16*>(smem);
  constexpr int TT = SBK * TPITCH;                    // [128 red][64 cols + 16]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tn = blockIdx.x % p.tilesN, tm = blockIdx.x / p.tilesN;
  const int row0 = tm * SBM, col0 = tn * SBN;
  const bf16* A = reinterpret_cast<const bf16*>(p.A);
  const bf16* B = reinterpret_cast<const bf16*>(p.B);
  const int nk_total = (p.K + SBK - 1) / SBK;
  const int kt0 = blockIdx.y * p.kt_per_split, kt1 = min(nk_total, kt0 + p.kt_per_split);
  const int tc = tid & 7, tr = tid >> 3;
  const bool a_ok = row0 + tc * 8 < p.M, b_ok = col0 + tc * 8 < p.N;      // M % 8 == 0, N % 8 == 0 (host)
  // buffer loads (see skinny_kernel): fixed per-thread offsets, the k-step offset in soffset, out-of-range = zeros
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  constexpr int OOB = 0x7ffffff0;
  const auto rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(A), (short)0, OOB, 0x00020000);
  const auto rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(B), (short)0, OOB, 0x00020000);
  int avo[4], bvo[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    avo[i] = a_ok ? ((tr + 32 * i) * (int)p.lda + row0 + tc * 8) * 2 : OOB;
    bvo[i] = b_ok ? ((tr + 32 * i) * (int)p.ldb + col0 + tc * 8) * 2 : OOB;
  }
  const bool do_sum = p.a_rowsum != nullptr && tn == 0;
  float csum[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) csum[e] = 0.f;
  uint4 ra[4], rb[4];
  auto load = [&](int kt) {
    const bool full = (kt + 1) * SBK <= p.K;           // uniform
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = full || kt * SBK + tr + 32 * i < p.K;
      const u32x4 ta = __builtin_amdgcn_raw_buffer_load_b128(rsA, ok ? avo[i] : OOB, kt * SBK * (int)p.lda * 2, 0);
      const u32x4 tb = __builtin_amdgcn_raw_buffer_load_b128(rsB, ok ? bvo[i] : OOB, kt * SBK * (int)p.ldb * 2, 0);
      ra[i] = make_uint4(ta[0], ta[1], ta[2], ta[3]);
      rb[i] = make_uint4(tb[0], tb[1], tb[2], tb[3]);
    }
  };
  auto store = [&](int stage) {
    bf16* sa = lds + stage * 2 * TT;
    bf16* sb = sa + TT;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<uint4*>(sa + (tr + 32 * i) * TPITCH + tc * 8) = ra[i];
      *reinterpret_cast<uint4*>(sb + (tr + 32 * i) * TPITCH + tc * 8) = rb[i];
      if (do_sum) {
        const uint32_t w[4] = {ra[i].x, ra[i].y, ra[i].z, ra[i].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          csum[2 * e] += __builtin_bit_cast(float, w[e] << 16);
          csum[2 * e + 1] += __builtin_bit_cast(float, w[e] & 0xffff0000u);
        }
      }
    }
  };
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  load(kt0);
  store(0);
  __syncthreads();
  for (int t = kt0; t < kt1; ++t) {
    if (t + 1 < kt1) load(t + 1);
    const bf16* sa = lds + ((t - kt0) & 1) * 2 * TT + wave * 32 * TPITCH;
    const bf16* sb = sa + TT;
    bf16x8 af[4], bfr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { af[i] = tr_frag(sa, i * 16, lane); bfr[i] = tr_frag(sb, i * 16, lane); }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(bfr[j], af[i], acc[i][j]);
    if (t + 1 < kt1) store(((t + 1 - kt0) & 1));
    __syncthreads();
  }
  if (do_sum) {                                       // threads with equal tc hold partial sums of the same 8 rows of dW
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = csum[e];
      v += __shfl_xor(v, 8); v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
      const int m = row0 + tc * 8 + e;
      if (lane < 8 && m < p.M) atomicAdd(p.a_rowsum + m, v);
    }
  }
  float* red = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<f32x4*>(red + wave * (SBM * SBN) + (i * 4 + j) * 256 + lane * 4) = acc[i][j];
  __syncthreads();
  float* Cp = p.ws ? p.ws + (int64_t)blockIdx.y * p.M * p.N : reinterpret_cast<float*>(p.C);
  const int64_t ldc = p.ws ? p.N : p.ldc;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int q = tid + 256 * r;
    const int ij = q >> 6, l = q & 63;
    f32x4 v = *reinterpret_cast<const f32x4*>(red + ij * 256 + l * 4);
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const f32x4 u = *reinterpret_cast<const f32x4*>(red + w * (SBM * SBN) + ij * 256 + l * 4);
      v[0] += u[0]; v[1] += u[1]; v[2] += u[2]; v[3] += u[3];
    }
    const int m = row0 + (ij >> 2) * 16 + (l & 15);
    const int n = col0 + (ij & 3) * 16 + (l >> 4) * 4;
    if (m >= p.M || n >= p.N) continue;
    const float rs = p.rowscale ? p.rowscale[m] * p.alpha : p.alpha;
    float4* dst = reinterpret_cast<float4*>(Cp + (int64_t)m * ldc + n);     // N % 8 == 0, ldc % 4 == 0, 16-byte aligned (host)
    float4 o = make_float4(v[0] * rs, v[1] * rs, v[2] * rs, v[3] * rs);
    if (!p.ws) { const float4 c = *dst; o.x += c.x; o.y += c.y; o.z += c.z; o.w += c.w; }      // single split: C += ...
    *dst = o;
  }
}

inline bool al16s(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

}  // namespace

// returns 0 = launched, -1 = not applicable, > 0 = hipError_t.  b_trans: B is reduction-major (GPV_TRANS): dX = dY W.
int skinny_try_launch(const GemmK& k, int b_trans, int dtype_in, int dtype_out, int batch, hipStream_t st) {
  if (g_skinny_mode == 0 || dtype_in != GPV_BF16 || batch != 1 || k.accumulate || k.split_k > 1) return -1;
  if (k.K % 8 != 0 || k.lda % 8 != 0 || k.ldb % 8 != 0 || !al16s(k.A) || !al16s(k.B)) return -1;
  if (b_trans && k.N % 8 != 0) return -1;
  {   // 32-bit byte offsets into each operand
    const int64_t lim = 0x7ffffff0ll / 2;
    if ((int64_t)k.M * k.lda >= lim || (int64_t)(b_trans ? k.K : k.N) * k.ldb >= lim) return -1;
  }
  if (g_skinny_mode == 1) {
    // measured on every forward GEMM shape of the step (tools/bench_step_gemms.py, SK=0 vs 2): it wins whenever the
    // 64x64 tiles cannot fill the chip (<= ~1.4 per CU), and up to 2.5 per CU when the reduction is long
    const int64_t tiles = (int64_t)((k.M + SBM - 1) / SBM) * ((k.N + SBN - 1) / SBN);
    // (the 360..640-tile, K >= 2048 launches -- 9600x256x2048, 3200x768x3072 -- go to the 4-wave direct-to-LDS kernel when B
    // is k-major: 54 -> 30 us, 52 -> 40 us; that kernel has no reduction-major B form, so dX = dY W keeps them here)
    if (!((tiles <= 360 && k.K >= 128) || (b_trans && tiles <= 640 && k.K >= 2048))) return -1;
  }
  if (b_trans) return dtype_out == GPV_BF16 ? launch_skinny<bf16, true>(k, st) : launch_skinny<float, true>(k, st);
  return dtype_out == GPV_BF16 ? launch_skinny<bf16, false>(k, st) : launch_skinny<float, false>(k, st);
}

// weight-gradient form; returns 0 = launched, -1 = not applicable
int skinny_tt_try_launch(const GemmK& k, int dtype_in, int dtype_out, int batch, hipStream_t st) {
  if (g_skinny_mode == 0 || dtype_in != GPV_BF16 || dtype_out != GPV_F32 || batch != 1 || !k.accumulate) return -1;
  if (k.M % 8 != 0 || k.N % 8 != 0 || k.lda % 8 != 0 || k.ldb % 8 != 0 || k.ldc % 4 != 0) return -1;
  if (!al16s(k.A) || !al16s(k.B) || !al16s(k.C) || k.res || k.mask || k.bias || k.act || k.dthresh) return -1;
  if ((int64_t)k.K * k.lda >= 0x7ffffff0ll / 2 || (int64_t)k.K * k.ldb >= 0x7ffffff0ll / 2) return -1;   // 32-bit byte offsets
  const int64_t tiles = (int64_t)((k.M + SBM - 1) / SBM) * ((k.N + SBN - 1) / SBN);
  const int nk = (k.K + SBK - 1) / SBK;
  if (g_skinny_mode == 1 && (tiles > 256 || nk < 4)) return -1;
  // enough splits for ~512 blocks, at least 3 k-steps of 128 each, within the lent workspace
  int split = (int)((512 + tiles - 1) / tiles);
  if (split > nk / 3) split = nk / 3;
  if (split < 1) split = 1;
  while (split > 1 && (k.ws_base == nullptr || (int64_t)split * k.M * k.N * 4 > k.ws_bytes)) --split;
  GemmK p = k;
  p.tilesN = (p.N + SBN - 1) / SBN;
  p.kt_per_split = (nk + split - 1) / split;
  split = (nk + p.kt_per_split - 1) / p.kt_per_split;
  p.ws = split > 1 ? reinterpret_cast<float*>(k.ws_base) : nullptr;
  constexpr size_t stage = (size_t)2 * 2 * SBK * TPITCH * 2;
  constexpr size_t redb = (size_t)4 * SBM * SBN * 4;
  constexpr size_t lds = stage > redb ? stage : redb;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(skinny_tt_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }      // (leave no stale error behind for the next launch check)
    attr_done = true;
  }
  hipLaunchKernelGGL(skinny_tt_kernel, dim3((unsigned)tiles, split), dim3(256), lds, st, p);
  GPV_CHECK_LAUNCH();
  if (split > 1) return launch_splitk_reduce(p.ws, split, p.M, p.N, reinterpret_cast<float*>(p.C), p.ldc, st);
  return 0;
}

}  // namespace gpvk
