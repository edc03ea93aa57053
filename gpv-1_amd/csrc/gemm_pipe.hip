// Pipelined direct-to-LDS bf16 GEMM / implicit-GEMM convolution (forward and dgrad) for gfx950: the kernel the B=32 ResNet-50
// body and the wide transformer GEMMs run on since round 2.
//
//   C[M,N] = epilogue( A[M,K] x B[N,K]^T ),  k-tiles of 64, mfma_f32_16x16x32_bf16, 8 waves, ONE block per CU
//
// What changed against gemm_glds.hip (two LDS stages, `s_waitcnt vmcnt(0)` + `__syncthreads()` per k-tile, 128x128 or 256-row tiles):
//   * THREE LDS stages and a counted `s_waitcnt vmcnt(N)`: the loads of k-tile t+2 are issued before the MFMAs of k-tile t, the wait at
//     the top of an iteration only retires the pieces of tile t+1's predecessor -- one whole tile stays in flight across the raw
//     `s_barrier` (PMC of the two-stage loop: waves parked 42 % of their cycles on vmcnt(0) + barrier, MFMA pipe busy 19 %;
//     a k-tile's operands take ~1 us to arrive from L2, one tile of look-ahead is less than that).  `__syncthreads()` would drain the
//     LDS-DMA queue (its fence carries vmcnt(0)), hence the raw barrier; the only LDS object is the dynamic array (a second
//     __shared__ object makes hipcc wait vmcnt(0) before every fragment read).
//   * Tile heights chosen per problem so that the tiles fill the 256 CUs in whole rounds: the B=32 body has 150-1200 tiles of
//     128x128; at two blocks per CU a launch of 600 tiles ran 2 rounds for 1.17 rounds of work.  BM in {96,128,160,192,256} x
//     BN in {128,256}; layer3's 38400 x 256 output is 240 tiles of 160x256 = one round on 94 % of the CUs.
//   * 8 waves = two per SIMD in one block: the second wave hides the first one's fragment reads; wave tiles of (BM/WM) x (BN/WN).
// Operand fetch, LDS image and swizzle are those of gemm_glds.hip: `buffer_load_dwordx4 ... lds` pieces of 8 rows x 128 B, lane
// (row r, slot s) fetches logical 16-byte chunk s ^ (r & 7), the fragment read of chunk c of row r looks at slot c ^ (r & 7)
// (conflict-free ds_read_b128); out-of-range rows / padding taps deliver zeros through the buffer descriptor's bounds check.
// Epilogue: alpha, rowscale, bias, residual, ReLU/GELU, dropout, ReLU-mask; fp32 tile staged through LDS in two halves.
#include "gemm_common.h"
#include <type_traits>
#include <cstdlib>
#ifndef GPV_PIPE_SWP
#define GPV_PIPE_SWP 1
#endif

namespace gpvk {
namespace {

constexpr int GBK = 64;
constexpr int ROWB = 128;
constexpr int NW = 8, NT = 512;
long g_pipe_launches = 0;

typedef __attribute__((address_space(3))) void lds_void_t;

template <int N> __device__ __forceinline__ void wait_vm_c() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// wave-uniform n (scalar branches)
__device__ __forceinline__ void wait_vm(int n) {
  switch (n) {
    case 0: wait_vm_c<0>(); break;
    case 1: wait_vm_c<1>(); break;
    case 2: wait_vm_c<2>(); break;
    case 3: wait_vm_c<3>(); break;
    case 4: wait_vm_c<4>(); break;
    case 5: wait_vm_c<5>(); break;
    case 6: wait_vm_c<6>(); break;
    case 7: wait_vm_c<7>(); break;
    case 8: wait_vm_c<8>(); break;
    case 9: wait_vm_c<9>(); break;
    case 10: wait_vm_c<10>(); break;
    case 11: wait_vm_c<11>(); break;
    case 12: wait_vm_c<12>(); break;
    case 13: wait_vm_c<13>(); break;
    case 14: wait_vm_c<14>(); break;
    case 15: wait_vm_c<15>(); break;
    case 16: wait_vm_c<16>(); break;
    case 17: wait_vm_c<17>(); break;
    case 18: wait_vm_c<18>(); break;
    default: wait_vm_c<0>(); break;
  }
}

template <int AMODE, int BM, int BN, int WM, int NS, typename TOut>
__device__ __forceinline__ void pipe_body(const GemmK& p) {
  constexpr int WN = NW / WM;
  constexpr bool SWP = GPV_PIPE_SWP != 0 && (BM / WM / 16 + BN / (NW / WM) / 16) * 8 + (BM / WM / 16) * (BN / (NW / WM) / 16) * 4 <= 200;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int FM = WTM / 16, FN = WTN / 16;
  static_assert(WTM % 16 == 0 && WTN % 16 == 0 && BM % 16 == 0 && BN % 64 == 0, "tile shape");
  constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
  constexpr int PA = BM / 8, PB = BN / 8;                   // 1-KiB pieces (8 rows) per operand and k-tile
  constexpr int AI = (PA + NW - 1) / NW, BI = (PB + NW - 1) / NW;
  constexpr bool A_EVEN = PA % NW == 0, B_EVEN = PB % NW == 0;
  extern __shared__ __attribute__((aligned(128))) unsigned char smem[];
  int* s_rowpix = reinterpret_cast<int*>(smem + NS * STAGE);  // [BM], stride-2 dgrad only (the launcher adds the bytes)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave - wm * WN;
  int tile;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  int tm = __builtin_amdgcn_readfirstlane(tile / p.tilesN);
  const int tn = tile - tm * p.tilesN;
  if constexpr (AMODE == OP_CONV) {
    if (p.cg.cm && p.cg.cls_rows % BM == 0 && p.M == 4 * p.cg.cls_rows) {
      const int tpc = p.cg.cls_rows / BM;
      tm = (tm & 3) * tpc + (tm >> 2);
    }
  }
  const int row0 = tm * BM, col0 = tn * BN;
  const int batch = blockIdx.z;

  int nk = p.K / GBK;
  int cm_r0 = 0, cm_s0 = 0, cm_nS = 1, cm_cpt = 1;
  bool cm_on = false, cm_empty = false;
  if constexpr (AMODE == OP_CONV) {
    if (p.cg.cm) {
      if (tid < BM) s_rowpix[tid] = conv_row_to_pixel(min(row0 + tid, p.M - 1), p.cg);
      const int c_lo = __builtin_amdgcn_readfirstlane(row0 / p.cg.cls_rows);
      const int c_hi = __builtin_amdgcn_readfirstlane(min(row0 + BM - 1, p.M - 1) / p.cg.cls_rows);
      if (c_lo == c_hi) {
        cm_on = true;
        cm_r0 = ((c_lo >> 1) + p.cg.PH) & 1;
        cm_s0 = ((c_lo & 1) + p.cg.PW) & 1;
        const int nR = (p.cg.KH - cm_r0 + 1) / 2;
        cm_nS = (p.cg.KW - cm_s0 + 1) / 2;
        cm_cpt = __builtin_amdgcn_readfirstlane(p.cg.Cin / GBK);
        nk = nR * cm_nS * cm_cpt;
        if (nk == 0) { cm_empty = true; nk = 1; }
      }
    }
  }
  constexpr int OOB = 0x7ffffff0;
  const int lrow = lane >> 3;
  const int lchunk = (lane & 7) ^ lrow;
  int a_vo[AI], a_oh[AI], a_ow[AI], b_vo[BI];
  const bf16* Ab = reinterpret_cast<const bf16*>(p.A) + (int64_t)batch * p.sA;
  const bf16* Bb = reinterpret_cast<const bf16*>(p.B) + (int64_t)batch * p.sB;
  const auto rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(Ab), (short)0, OOB, 0x00020000);
  const auto rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(Bb), (short)0, OOB, 0x00020000);
#pragma unroll
  for (int j = 0; j < AI; ++j) {
    const int r = (j * NW + wave) * 8 + lrow;              // piece j*NW+wave (pieces beyond PA are never issued)
    const int m = min(row0 + r, p.M - 1);
    if constexpr (AMODE == OP_CONV) {
      const int pix = conv_row_to_pixel(m, p.cg);
      const int b = pix / (p.cg.OH * p.cg.OW);
      const int rem = pix - b * (p.cg.OH * p.cg.OW);
      a_oh[j] = rem / p.cg.OW;
      a_ow[j] = rem - a_oh[j] * p.cg.OW;
      a_vo[j] = (b * p.cg.IH * p.cg.IW * p.cg.Cs + lchunk * 8) * 2;
    } else {
      a_oh[j] = a_ow[j] = 0;
      a_vo[j] = (m * (int)p.lda + lchunk * 8) * 2;
    }
  }
  const bool bz = (AMODE == OP_CONV) && cm_empty;
  // register epilogue over permuted output columns (gemm_glds.hip, round 4): wave tiles with an even number of 16-column fragments,
  // whole column tiles, 16-byte aligned operands
  const bool depi = (FN % 2 == 0) && p.depi && std::is_same<TOut, bf16>::value && p.N % BN == 0 && p.ldc % 8 == 0 &&
                    ((reinterpret_cast<uintptr_t>(p.C) | (uintptr_t)(p.sC * 2)) & 15) == 0 &&
                    (!p.res || (p.ldr % 8 == 0 && ((reinterpret_cast<uintptr_t>(p.res) | (uintptr_t)(p.sR * 2)) & 15) == 0)) &&
                    (!p.mask || (p.ldm % 8 == 0 && (reinterpret_cast<uintptr_t>(p.mask) & 15) == 0)) &&
                    (!p.bias || (reinterpret_cast<uintptr_t>(p.bias) & 15) == 0);
#pragma unroll
  for (int j = 0; j < BI; ++j) {
    const int r = (j * NW + wave) * 8 + lrow;
    const int rp = depi ? (r & ~31) + ((r & 15) >> 2) * 8 + ((r >> 4) & 1) * 4 + (r & 3) : r;
    const int n = min(col0 + rp, p.N - 1);
    b_vo[j] = bz ? OOB : (n * (int)p.ldb + lchunk * 8) * 2;
  }
  // loads this wave issues per k-tile (the count the vmcnt waits are built from)
  int nl = 0;
#pragma unroll
  for (int j = 0; j < AI; ++j) nl += (A_EVEN || j * NW + wave < PA) ? 1 : 0;
#pragma unroll
  for (int j = 0; j < BI; ++j) nl += (B_EVEN || j * NW + wave < PB) ? 1 : 0;
  nl = __builtin_amdgcn_readfirstlane(nl);

  auto bload = [&](const decltype(rsA)& rs, int voff, int soff, unsigned char* l) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)l, 16, voff, soff, 0, 0);
  };
  int it_c0 = 0, it_tr = cm_on ? cm_r0 : 0, it_ts = cm_on ? cm_s0 : 0, it_kt = 0;
  int tap_off[AI];
#pragma unroll
  for (int j = 0; j < AI; ++j) tap_off[j] = OOB;

  auto issue = [&](int stage) {
    unsigned char* sa = smem + stage * STAGE + wave * 1024;
    unsigned char* sb = sa + A_BYTES;
    int k0;
    if constexpr (AMODE == OP_CONV) {
      const ConvGeom& g = p.cg;
      if (it_c0 == 0) {
#pragma unroll
        for (int j = 0; j < AI; ++j) {
          int ih, iw;
          bool ok = !cm_empty;
          if (g.dgrad) {
            const int th = a_oh[j] + g.PH - it_tr, tw = a_ow[j] + g.PW - it_ts;
            const int sh = g.SH - 1, sw = g.SW - 1;
            ih = th >> sh; iw = tw >> sw;
            ok = ok && th >= 0 && tw >= 0 && ((th & sh) == 0) && ((tw & sw) == 0) && ih < g.IH && iw < g.IW;
          } else {
            ih = a_oh[j] * g.SH + it_tr - g.PH; iw = a_ow[j] * g.SW + it_ts - g.PW;
            ok = ok && ih >= 0 && iw >= 0 && ih < g.IH && iw < g.IW;
          }
          tap_off[j] = ok ? a_vo[j] + (ih * g.IW + iw) * g.Cs * 2 : OOB;
        }
      }
      k0 = (it_tr * g.KW + it_ts) * g.Cin + it_c0;
#pragma unroll
      for (int j = 0; j < AI; ++j)
        if (A_EVEN || j * NW + wave < PA) bload(rsA, tap_off[j], it_c0 * 2, sa + j * (NW * 1024));
      it_c0 += GBK;
      if (it_c0 == g.Cin) {
        it_c0 = 0;
        if (cm_on) { it_ts += 2; if (it_ts >= g.KW) { it_ts = cm_s0; it_tr += 2; } }
        else { ++it_ts; if (it_ts == g.KW) { it_ts = 0; ++it_tr; } }
      }
    } else {
      k0 = it_kt * GBK;
      ++it_kt;
#pragma unroll
      for (int j = 0; j < AI; ++j)
        if (A_EVEN || j * NW + wave < PA) bload(rsA, a_vo[j], k0 * 2, sa + j * (NW * 1024));
    }
#pragma unroll
    for (int j = 0; j < BI; ++j)
      if (B_EVEN || j * NW + wave < PB) bload(rsB, b_vo[j], k0 * 2, sb + j * (NW * 1024));
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fkg = lane >> 4, fsw = lane & 7;
  const int a_off = (wm * WTM + frow) * ROWB;
  const int b_off = A_BYTES + (wn * WTN + frow) * ROWB;

  if constexpr (AMODE == OP_CONV) {
    if (p.cg.cm) __syncthreads();          // s_rowpix visible before anything is in flight (a plain barrier: no LDS-DMA outstanding yet)
  }
#pragma unroll
  for (int s0 = 0; s0 < NS - 1; ++s0)
    if (s0 < nk) issue(s0);                 // NS-1 tiles in flight before the first MFMA (NS = 3 for the big tiles; 6-8 for the
  int sc = 0, si = NS - 1;                  // small-M tiles, whose whole cost is the latency of their short k-loop)
  for (int t = 0; t < nk; ++t) {
    // tile t has landed once at most the tiles issued after it are still outstanding for this wave ...
    const int ahead = min(nk, t + NS - 1) - (t + 1);
    wait_vm(ahead * nl);
    __builtin_amdgcn_s_barrier();           // ... for every wave; and everyone is done reading stage (t-1)%NS (tile t-1's)
    asm volatile("" ::: "memory");
    if (t + NS - 1 < nk) issue(si);
    const unsigned char* st = smem + sc * STAGE;
    if constexpr (SWP) {
      // all fragment reads of the k-tile (both 32-deep halves) are requested before its first MFMA: left to itself the
      // scheduler keeps ~6 fragments live and re-loads just in time -- ten `s_waitcnt lgkmcnt` stops per k-tile with 1..4 MFMAs
      // (16..64 clocks) of cover each against ~150 clocks of LDS latency.  Here one exposed latency per k-tile; the second
      // half's reads land under the first half's MFMAs.
      bf16x8 af[2][FM], bfr[2][FN];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int slot = ((kk * 4 + fkg) ^ fsw) << 4;
#pragma unroll
        for (int j = 0; j < FN; ++j) bfr[kk][j] = *reinterpret_cast<const bf16x8*>(st + b_off + j * 16 * ROWB + slot);
#pragma unroll
        for (int i = 0; i < FM; ++i) af[kk][i] = *reinterpret_cast<const bf16x8*>(st + a_off + i * 16 * ROWB + slot);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) acc[i][j] = mfma16(bfr[kk][j], af[kk][i], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int slot = ((kk * 4 + fkg) ^ fsw) << 4;
      bf16x8 af[FM], bfr[FN];
#pragma unroll
      for (int j = 0; j < FN; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(st + b_off + j * 16 * ROWB + slot);
#pragma unroll
      for (int i = 0; i < FM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(st + a_off + i * 16 * ROWB + slot);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = mfma16(bfr[j], af[i], acc[i][j]);
    }
    }
    sc = sc == NS - 1 ? 0 : sc + 1;
    si = si == NS - 1 ? 0 : si + 1;
  }

  // ---------------- epilogue: fragments -> LDS (fp32, half the rows at a time) -> whole rows ----------------
  TOut* Cp = reinterpret_cast<TOut*>(p.C) + (int64_t)batch * p.sC;
  const TOut* Rp = p.res ? reinterpret_cast<const TOut*>(p.res) + (int64_t)batch * p.sR : nullptr;
  const TOut* Mp = reinterpret_cast<const TOut*>(p.mask);
  float* ep = reinterpret_cast<float*>(smem);
  constexpr int EPITCH = BN + 4;
  constexpr int NHALF = (BM * BN >= 2 * NT * 8) ? 2 : 1;      // small tiles: the whole fp32 image in one pass
  constexpr int HR = BM / NHALF;
  constexpr int CH = BN / 8;
  static_assert(HR % WTM == 0 || WTM % HR == 0, "a wave's rows lie in one half");
  static_assert((size_t)HR * EPITCH * 4 <= (size_t)NS * STAGE, "epilogue image fits the stages");
  constexpr int NCH = (HR * CH + NT - 1) / NT;
  const bool v_st = (p.ldc % 8 == 0) && ((reinterpret_cast<uintptr_t>(Cp) & 15) == 0);
  const bool v_res = Rp && (p.ldr % 8 == 0) && ((reinterpret_cast<uintptr_t>(Rp) & 15) == 0);
  const bool v_msk = Mp && (p.ldm % 8 == 0) && ((reinterpret_cast<uintptr_t>(Mp) & 15) == 0);
  if constexpr (FN % 2 == 0) {
    if (depi) {
      const int mrow = wm * WTM + frow;
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int ml = mrow + i * 16, m = row0 + ml;
        if (m >= p.M) continue;
        int64_t mp = m;
        if constexpr (AMODE == OP_CONV) { if (p.cg.cm) mp = s_rowpix[ml]; }
        const float rs = p.rowscale ? p.rowscale[m] * p.alpha : p.alpha;
#pragma unroll
        for (int t = 0; t < FN / 2; ++t) {
          const int n = col0 + wn * WTN + t * 32 + fkg * 8;
          float bq[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, rv[8], mv[8];
          if (p.bias) Ld8<float>::ld(p.bias + n, bq);
          if (Rp) Ld8<TOut>::ld(Rp + mp * p.ldr + n, rv);
          if (Mp) Ld8<TOut>::ld(Mp + mp * p.ldm + n, mv);
          const uint32_t keep8 = p.dthresh ? drop_mask<8>(p.seed, ((uint64_t)batch * p.M + m) * (uint64_t)p.N + n, p.dthresh) : 0xffu;
          float v[8] = {acc[i][2 * t][0], acc[i][2 * t][1], acc[i][2 * t][2], acc[i][2 * t][3],
                        acc[i][2 * t + 1][0], acc[i][2 * t + 1][1], acc[i][2 * t + 1][2], acc[i][2 * t + 1][3]};
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float x = v[e] * rs;
            x += bq[e];
            if (Rp) x += rv[e];
            if (p.act == GPV_ACT_RELU) x = fmaxf(x, 0.f);
            else if (p.act == GPV_ACT_GELU) x = gelu_erf(x);
            if (p.dthresh) x = ((keep8 >> e) & 1u) ? x * p.dscale : 0.f;
            if (Mp) x = mv[e] > 0.f ? x : 0.f;
            v[e] = x;
          }
          Ld8<TOut>::st(Cp + mp * p.ldc + n, v);
        }
      }
      return;
    }
  }
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
  for (int half = 0; half < NHALF; ++half) {
    __syncthreads();
    if ((wm * WTM) / HR == half) {
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int r = wm * WTM + i * 16 + frow - half * HR;
#pragma unroll
        for (int j = 0; j < FN; ++j)
          *reinterpret_cast<f32x4*>(ep + r * EPITCH + wn * WTN + j * 16 + fkg * 4) = acc[i][j];
      }
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < NCH; ++g) {
      const int idx = tid + g * NT;
      if ((HR * CH) % NT != 0 && idx >= HR * CH) continue;
      const int r = idx / CH, c8 = idx - r * CH;
      const int m = row0 + half * HR + r;
      const int n = col0 + c8 * 8;
      if (m >= p.M || n >= p.N) continue;
      const bool full = n + 8 <= p.N;
      int64_t mp = m;
      if constexpr (AMODE == OP_CONV) { if (p.cg.cm) mp = s_rowpix[half * HR + r]; }
      float v[8], rv[8], mv[8];
      {
        const float4 a = *reinterpret_cast<const float4*>(ep + r * EPITCH + c8 * 8);
        const float4 b = *reinterpret_cast<const float4*>(ep + r * EPITCH + c8 * 8 + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
      }
      if (Rp) {
        if (v_res && full) Ld8<TOut>::ld(Rp + mp * p.ldr + n, rv);
        else {
#pragma unroll
          for (int e = 0; e < 8; ++e) rv[e] = (n + e < p.N) ? (float)Rp[mp * p.ldr + n + e] : 0.f;
        }
      }
      if (Mp) {
        if (v_msk && full) Ld8<TOut>::ld(Mp + mp * p.ldm + n, mv);
        else {
#pragma unroll
          for (int e = 0; e < 8; ++e) mv[e] = (n + e < p.N) ? (float)Mp[mp * p.ldm + n + e] : 0.f;
        }
      }
      const float rs = p.rowscale ? p.rowscale[m] * p.alpha : p.alpha;
      const uint32_t keep8 = p.dthresh ? drop_mask<8>(p.seed, ((uint64_t)batch * p.M + m) * (uint64_t)p.N + n, p.dthresh) : 0xffu;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float x = v[e] * rs;
        x += bv[e];
        if (Rp) x += rv[e];
        if (p.act == GPV_ACT_RELU) x = fmaxf(x, 0.f);
        else if (p.act == GPV_ACT_GELU) x = gelu_erf(x);
        if (p.dthresh) x = ((keep8 >> e) & 1u) ? x * p.dscale : 0.f;
        if (Mp) x = mv[e] > 0.f ? x : 0.f;
        v[e] = x;
      }
      TOut* dst = Cp + mp * p.ldc + n;
      if (v_st && full) Ld8<TOut>::st(dst, v);
      else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (n + e < p.N) dst[e] = (TOut)v[e];
      }
    }
  }
}

template <int AMODE, int BM, int BN, int WM, int NS, typename TOut>
__global__ __launch_bounds__(NT) void pipe_kernel(GemmK p) {
  if (p.dthresh) p.seed = eff_seed(p.seed, p.seed_dev);
  pipe_body<AMODE, BM, BN, WM, NS, TOut>(p);
}
// 1x1 stride-1 convolutions launched as plain GEMMs keep a name of their own (rocprofv3 attribution to the backbone)
template <int BM, int BN, int WM, int NS, typename TOut>
__global__ __launch_bounds__(NT) void pipe_conv1x1_kernel(GemmK p) {
  if (p.dthresh) p.seed = eff_seed(p.seed, p.seed_dev);
  pipe_body<OP_PLAIN, BM, BN, WM, NS, TOut>(p);
}

template <int AMODE, int BM, int BN, int WM, int NS>
int launch_pipe(const GemmK& k, int batch, hipStream_t st) {
  constexpr size_t stages = (size_t)NS * (BM + BN) * ROWB;
  constexpr size_t lds = stages + (AMODE == OP_CONV ? (size_t)BM * 4 : 0);
  static_assert(lds <= 160 * 1024, "LDS budget");
  GemmK p = k;
  const int tilesM = (p.M + BM - 1) / BM;
  p.tilesN = (p.N + BN - 1) / BN;
  const bool c11 = AMODE == OP_PLAIN && p.conv1x1;
  void (*fn)(GemmK);
  if constexpr (AMODE == OP_PLAIN) fn = c11 ? pipe_conv1x1_kernel<BM, BN, WM, NS, bf16> : pipe_kernel<AMODE, BM, BN, WM, NS, bf16>;
  else fn = pipe_kernel<AMODE, BM, BN, WM, NS, bf16>;
  static bool attr_done[2] = {false, false};
  if (!attr_done[c11]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }      // (leave no stale error behind for the next launch check)
    attr_done[c11] = true;
  }
  dim3 grid(tilesM * p.tilesN, 1, batch);
  ++g_pipe_launches;
  static const int depi = tune_env("GPV_GLDS_DEPI", 1);
  p.depi = depi;
  hipLaunchKernelGGL(fn, grid, dim3(NT), lds, st, p);
  GPV_CHECK_LAUNCH();
  return 0;
}

// tile configurations: index -> (BM, BN, WM)
struct PipeCfg { int bm, bn; };
constexpr PipeCfg kCfgs[] = {{256, 128}, {192, 128}, {128, 128}, {160, 256}, {128, 256}, {96, 256},
                             {64, 64}, {32, 64}};     // the last two: small-M tiles, 6 / 8 stages, plain GEMMs only
constexpr int kNumBig = 6;
constexpr int kNumCfgs = sizeof(kCfgs) / sizeof(kCfgs[0]);

template <int AMODE>
int launch_cfg_idx(int idx, const GemmK& k, int batch, hipStream_t st) {
  switch (idx) {
    case 0: return launch_pipe<AMODE, 256, 128, 4, 3>(k, batch, st);
    case 1: return launch_pipe<AMODE, 192, 128, 4, 3>(k, batch, st);
    case 2: return launch_pipe<AMODE, 128, 128, 2, 3>(k, batch, st);
    case 3: return launch_pipe<AMODE, 160, 256, 2, 3>(k, batch, st);
    case 4: return launch_pipe<AMODE, 128, 256, 2, 3>(k, batch, st);
    case 5: return launch_pipe<AMODE, 96, 256, 2, 3>(k, batch, st);
  }
  if constexpr (AMODE == OP_PLAIN) {
    if (idx == 6) return launch_pipe<OP_PLAIN, 64, 64, 2, 6>(k, batch, st);
    if (idx == 7) return launch_pipe<OP_PLAIN, 32, 64, 2, 8>(k, batch, st);
  }
  return -1;
}

int g_pipe_small = tune_env("GPV_PIPE_SMALL", 1);   // the small-M configurations (64 x 64 / 32 x 64, 6 / 8 stages)
int g_pipe_mode = tune_env("GPV_PIPE", 1);   // 0 off, 1 heuristic, 100+i: force configuration i wherever legal

inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

// rounds of 256 CUs a configuration needs, weighted by the tile's MFMA work; ties go to the larger tile (operand reuse)
int pick_cfg(const GemmK& k, int batch) {
  double best = 1e300;
  int bi = -1;
  for (int i = 0; i < kNumBig; ++i) {
    const int bm = kCfgs[i].bm, bn = kCfgs[i].bn;
    if (k.N % bn != 0) continue;
    const int64_t tiles = (int64_t)((k.M + bm - 1) / bm) * (k.N / bn) * batch;
    const int64_t rounds = (tiles + 255) / 256;
    // cost of a round ~ MFMA work of a tile + a term for the operand bytes it pulls from L2 (both per k-tile) + fixed per-tile overhead
    const double per_kt = (double)bm * bn / 16384.0 * 4.0 + (double)(bm + bn) * 128.0 / 56.0 / 16.0 * 0.5;
    const double cost = (double)rounds * (per_kt * (k.K / 64) + 40.0);
    if (cost < best - 1e-9) { best = cost; bi = i; }
  }
  return bi;
}

}  // namespace

int pipe_try_launch(const GemmK& k, int amode, int dtype_in, int dtype_out, int batch, hipStream_t st) {
  const int mode = g_pipe_mode;
  if (mode == 0 || dtype_in != GPV_BF16 || dtype_out != GPV_BF16) return -1;
  if (k.accumulate || k.split_k > 1) return -1;
  if (k.K % GBK != 0 || k.K < 2 * GBK || k.N % 64 != 0) return -1;
  if (!al16(k.A) || !al16(k.B) || k.ldb % 8 != 0 || (batch > 1 && (k.sA % 8 != 0 || k.sB % 8 != 0))) return -1;
  const int64_t lim = 0x7ffffff0ll / 2;
  if ((int64_t)k.N * k.ldb >= lim) return -1;
  if (amode == OP_CONV) {
    if ((int64_t)k.cg.IH * k.cg.IW * k.cg.Cs * ((int64_t)k.M / ((int64_t)k.cg.OH * k.cg.OW) + 1) >= lim) return -1;
    if (k.cg.Cin % GBK != 0 || k.cg.Cs % 8 != 0) return -1;
  } else if (amode == OP_PLAIN) {
    if (k.lda % 8 != 0 || (int64_t)k.M * k.lda >= lim) return -1;
  } else {
    return -1;
  }
  if (mode < 100 && glds_two_per_cu(k, batch)) return -1;       // (gemm_glds.hip takes these: two half-width tiles per CU)
  int idx;
  if (mode >= 100) {
    idx = mode - 100;
    if (idx >= kNumCfgs || k.N % kCfgs[idx].bn != 0 || (idx >= kNumBig && amode != OP_PLAIN)) return -1;
  } else if (g_pipe_small && !k.no_small && amode == OP_PLAIN && !k.conv1x1 && k.K >= 256 &&
             (int64_t)((k.M + 63) / 64) * (k.N / 64) * batch <= 250 &&
             ((int64_t)((k.M + 63) / 64) * (k.N / 64) * batch >= 100 || k.K <= 1024)) {
    // small-M GEMMs (BERT and the co-attention text stream at M = 192, the text decoder at 640 rows): too few 64x64 tiles to
    // fill the chip.  6-8 k-tiles in flight through LDS-DMA instead of gemm_skinny.hip's one register-staged tile of
    // look-ahead: 9.0 -> 8.4 us (192x768x768), 10.6 -> 8.8 (192x2304x768), 19.0 -> 16.5 (3200x256x2048) -- modest: what
    // bounds these launches is the ~35 GB/s per CU of the operand path times the few CUs they occupy, not the look-ahead
    // (tools/bench_pipe.py small).  Very few tiles with a long reduction (192x768x3072) keep the in-block k-split of gemm_skinny.
    const int64_t t64 = (int64_t)((k.M + 63) / 64) * (k.N / 64) * batch;
    idx = t64 < 128 ? 7 : 6;
  } else {
    // Where it wins (tools/bench_pipe.py, B=32 shapes, round 2): reductions of >= 512 over 256..768-wide outputs of <= 40 k rows
    // (layer3/4 3x3 and long-K 1x1 convs fwd + dgrad, the DETR / co-attention GEMMs) and any width below 9600 rows.  N = 128
    // (layer2) and the >= 1024-wide outputs of >= 9600 rows stay on the two-blocks-per-CU 128x128 kernel (its 600-2400 tiles
    // balance better than 256-row rounds); the stride-2 dgrad parity classes are too uneven for one block per CU.
    if (k.K < 512 || k.M < 2048 || k.M > 40000 || k.N < 256) return -1;
    if (amode == OP_CONV && k.cg.cm) return -1;
    if (k.N >= 1024 && k.M >= 9600) return -1;
    if ((int64_t)k.M * k.N * batch > (int64_t)24 << 20) return -1;          // >= 1024 tiles of 256x256: the 8-wave 256x256 kernel (128 flop per operand byte)
    idx = pick_cfg(k, batch);
    if (idx < 0) return -1;
  }
  return amode == OP_CONV ? launch_cfg_idx<OP_CONV>(idx, k, batch, st) : launch_cfg_idx<OP_PLAIN>(idx, k, batch, st);
}

int pipe_set_mode(int v) { const int prev = g_pipe_mode; g_pipe_mode = v; return prev; }
int pipe_set_small(int v) { const int prev = g_pipe_small; g_pipe_small = v; return prev; }
long pipe_launches(long set) { const long prev = g_pipe_launches; if (set >= 0) g_pipe_launches = set; return prev; }

}  // namespace gpvk
