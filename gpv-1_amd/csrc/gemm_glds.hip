// Direct-to-LDS bf16 GEMM / implicit-GEMM convolution (forward and dgrad) for gfx950, in two shapes:
//
//   C[M,N] = epilogue( A[M,K] x B[N,K]^T ),  k-tiles of 64, mfma_f32_16x16x32_bf16
//   * 8 waves, 256 x BN (256 | 128) tiles, wave tile 128x64 (waves 2x4) or 64x64 (waves 4x2), one block per CU:
//     the launches that fill the chip several times over (960-985 TFLOP/s on 4096^3 / 8192^3);
//   * 4 waves, 128 x 128 tiles, wave tile 64x64 (waves 2x2), 64 KB of LDS -> two blocks per CU: the B=32 ResNet-50 /
//     transformer launches with K >= 512..1024 (150-1200 tiles), where a second resident block hides the other's
//     prologue/epilogue and the tail is finer (layer4 3x3 fwd 150 -> 76 us, 9600x256x2048 54 -> 30 us).
//
// Why a second kernel next to gemm.hip: the 4-wave 128x128x32 register-staged kernel tops out at ~550-650 TFLOP/s on
// large GEMMs (tools/bench_gemm_sq.py): a barrier and a ds_read restart every 32-deep step, 8 LDS fragment reads per
// 16 MFMAs, the staging registers + ds_write pass in the loop.  Here
//   * operand rows go HBM/L2 -> LDS with global_load_lds_dwordx4 (no staging VGPRs, no ds_write pass); a k-tile is one
//     full 128-byte line per operand row, so every request is a whole cache line;
//   * the LDS image is the lane-linear image the DMA writes (wave instruction = 8 rows x 128 B); bank conflicts are
//     removed by an XOR swizzle applied on BOTH sides: lane (row r, slot s) FETCHES logical 16-B chunk s^(r&7), the
//     MFMA fragment read of chunk c of row r looks at slot c^(r&7)  (ds_read_b128 conflict-free, checked by hand for
//     the hardware's 16-lane service groups);
//   * two LDS stages, one barrier per 64-deep k-tile; tile t+1 is in flight while tile t is multiplied;
//   * 128x64 wave tiles: 12 fragment reads per 32 MFMAs.
// Conv gathers (NHWC, forward or dgrad geometry, stride-2 dgrad parity classes) only change the per-lane SOURCE
// address; padding taps fetch from a 128-byte line of zeros.
// The epilogue is the one of gemm.hip (alpha, rowscale, bias, residual, ReLU/GELU, dropout, ReLU-mask; fp32 tile staged
// through LDS in two halves so that stores and residual/mask loads cover whole rows).
// bf16 operands only: the fp32 "precise" mode needs the hi/lo split on the way into LDS and stays on gemm.hip.
#include "gemm_common.h"
#include <cstdlib>
#include <type_traits>

namespace gpvk {
namespace {

constexpr int GBK = 64;       // k-tile in elements
constexpr int ROWB = 128;     // bytes per staged operand row
constexpr int GBM = 256;
long g_glds_launches = 0;     // gpv_set_option(GPV_OPT_GLDS_LAUNCHES, .)

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

template <int AMODE, int BM, int BN, typename TOut>
__device__ __forceinline__ void glds_body(const GemmK& p) {
  constexpr int NT = BM == 256 ? 512 : 256, NW = NT / 64;     // 512 threads = 8 waves (BM 256) | 256 threads = 4 waves (BM 96 / 128 / 160, two blocks per CU)
  constexpr int WN = BN / 64, WM = NW / WN;
  constexpr int WTM = BM / WM;                 // 128 | 64
  constexpr int FM = WTM / 16, FN = 4;
  constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
  constexpr int AI = BM / (8 * NW), BI = BN / (8 * NW);    // global_load_lds instructions per wave and k-tile (8 rows each)
  extern __shared__ __attribute__((aligned(128))) unsigned char smem[];
  __shared__ int s_rowpix[AMODE == OP_CONV ? BM : 1];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform, but only provably so this way: LDS-DMA bases (M0) in SGPRs
  const int wm = wave / WN, wn = wave - wm * WN;
  // XCD-aware tile order (see gemm.hip): contiguous tile range per XCD
  int tile;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  if constexpr (AMODE == OP_CONV) {
    if (p.cg.cm && p.cg.cls_rows % BM == 0 && p.M == 4 * p.cg.cls_rows) {
      // Stride-2 backward-data: row panel tm is parity class tm & 3 (next remap), so an XCD's contiguous tile range holds the four
      // classes of the same dy pixels (shared L2 lines) -- but in launch order every resident slot kept drawing the SAME class: the
      // slots on the 4-tap class (4 x the k-tiles of the 1-tap one) finished last.  Inside its range an XCD now takes the LONGEST
      // tiles first: class 3 (four taps), then 1 and 2 (two), then 0 (one); list scheduling of the same 1920 tiles on 512 slots:
      // makespan 80 -> 68 in units of k-tiles (ramp 8).
      const int nwg = gridDim.x, bid = blockIdx.x;
      const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
      const int s0 = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q, cnt = q + (xcd < r ? 1 : 0);
      const int tN = p.tilesN, G = 4 * tN;
      auto below = [&](int x, int c) { const int m = x / G, u = x - m * G - c * tN; return m * tN + (u < 0 ? 0 : (u > tN ? tN : u)); };   // tiles < x of class c
      int left = loc, pick = 0, before = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = k == 0 ? 3 : k == 1 ? 1 : k == 2 ? 2 : 0;
        const int b = below(s0, c), n = below(s0 + cnt, c) - b;
        if (k == 3 || left < n) { pick = c; before = b; break; }
        left -= n;
      }
      const int ord = before + left;                    // the ord-th tile of class `pick` in the whole launch
      tile = (ord / tN) * G + pick * tN + (ord - (ord / tN) * tN);
    }
  }
  int tm = __builtin_amdgcn_readfirstlane(tile / p.tilesN);      // (the division is done on the VALU)
  const int tn = tile - tm * p.tilesN;
  if constexpr (AMODE == OP_CONV) {
    if (p.cg.cm && p.cg.cls_rows % BM == 0 && p.M == 4 * p.cg.cls_rows) {           // cycle the stride-2 dgrad parity classes through the row panels
      const int tpc = p.cg.cls_rows / BM;
      tm = (tm & 3) * tpc + (tm >> 2);
    }
  }
  const int row0 = tm * BM, col0 = tn * BN;
  const int batch = blockIdx.z;

  // ---- reduction axis: k-tiles of 64; a class-uniform stride-2 dgrad tile only visits its own taps ----
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
        if (nk == 0) { cm_empty = true; nk = 1; }       // no tap reaches this class: one all-zero k-tile, dx = res * mask
      }
    }
  }
  // ---- loader state: this lane's rows and its (swizzled) 16-byte chunk ----
  // Operands are fetched with buffer_load_dwordx4 ... lds through one descriptor per operand: the per-lane part of an address
  // is a 32-bit byte offset fixed for the whole tile (conv: for a whole tap), the k-tile offset rides in the scalar soffset,
  // and an out-of-range voffset (padding tap, empty parity class) makes the hardware deliver zeros -- no 64-bit address
  // arithmetic and no zero-line select per load in the loop (PMC: the loop issued 4.9 VALU instructions per MFMA).
  constexpr int OOB = 0x7ffffff0;                      // == num_records: voffset + 16 > num_records -> zeros
  const int lrow = lane >> 3;
  const int lchunk = (lane & 7) ^ lrow;                 // logical chunk fetched into slot (lane & 7) of row lrow
  int a_vo[AI];                                        // plain: row * lda bytes ; conv: image base bytes
  int a_oh[AI], a_ow[AI];
  int b_vo[BI];
  const bf16* Ab = reinterpret_cast<const bf16*>(p.A) + (int64_t)batch * p.sA;
  const bf16* Bb = reinterpret_cast<const bf16*>(p.B) + (int64_t)batch * p.sB;
  const auto rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(Ab), (short)0, OOB, 0x00020000);
  const auto rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(Bb), (short)0, OOB, 0x00020000);
#pragma unroll
  for (int j = 0; j < AI; ++j) {
    const int r = wave * (AI * 8) + j * 8 + lrow;
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
  // Register epilogue (round 4): when the tile's columns are whole and every access is 16-byte aligned, the B rows
  // (= output columns) are staged PERMUTED -- LDS row L of a 32-row group holds column 8 (r / 4) + 4 jj + (r % 4), jj = L / 16 % 2,
  // r = L % 16 (conv1x1_stream.hip's c1s_chan) -- so that the accumulator tiles 2t, 2t + 1 leave lane (row, g) with the 8 consecutive
  // columns 32 t + 8 g .. + 7 of its row: bias / residual / mask / output are plain 16-byte accesses in the accumulator layout, and
  // the fp32 LDS image with its four barriers per tile (16 us of serial ramp + store burst around the main loop) goes.
  const bool depi = p.depi && std::is_same<TOut, bf16>::value && p.N % BN == 0 && p.ldc % 8 == 0 &&
                    ((reinterpret_cast<uintptr_t>(p.C) | (uintptr_t)(p.sC * 2)) & 15) == 0 &&
                    (!p.res || (p.ldr % 8 == 0 && ((reinterpret_cast<uintptr_t>(p.res) | (uintptr_t)(p.sR * 2)) & 15) == 0)) &&
                    (!p.mask || (p.ldm % 8 == 0 && (reinterpret_cast<uintptr_t>(p.mask) & 15) == 0)) &&
                    (!p.bias || (reinterpret_cast<uintptr_t>(p.bias) & 15) == 0);
#pragma unroll
  for (int j = 0; j < BI; ++j) {
    const int r = wave * (BI * 8) + j * 8 + lrow;
    const int rp = depi ? (r & ~31) + ((r & 15) >> 2) * 8 + ((r >> 4) & 1) * 4 + (r & 3) : r;
    const int n = min(col0 + rp, p.N - 1);
    b_vo[j] = bz ? OOB : (n * (int)p.ldb + lchunk * 8) * 2;
  }
  auto bload = [&](const decltype(rsA)& rs, int voff, int soff, unsigned char* l) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)l, 16, voff, soff, 0, 0);
  };

  // conv gather state: k-tiles walk (tap, channel block) with the channel block innermost, so the per-row source offset and
  // its bounds check are recomputed once per TAP (Cin/64 k-tiles), not per k-tile; the loop itself only changes soffset.
  int it_c0 = 0, it_tr = cm_on ? cm_r0 : 0, it_ts = cm_on ? cm_s0 : 0;
  int tap_off[AI];
#pragma unroll
  for (int j = 0; j < AI; ++j) tap_off[j] = OOB;

  auto issue = [&](int kt, int stage) {
    unsigned char* sa = smem + stage * STAGE + wave * (AI * 1024);
    unsigned char* sb = smem + stage * STAGE + A_BYTES + wave * (BI * 1024);
    int k0;
    if constexpr (AMODE == OP_CONV) {
      const ConvGeom& g = p.cg;
      if (it_c0 == 0) {                              // new tap (uniform branch)
#pragma unroll
        for (int j = 0; j < AI; ++j) {
          int ih, iw;
          bool ok = !cm_empty;
          if (g.dgrad) {                             // strides are 1 or 2 (checked on the host)
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
      for (int j = 0; j < AI; ++j) bload(rsA, tap_off[j], it_c0 * 2, sa + j * 1024);
      it_c0 += GBK;
      if (it_c0 == g.Cin) {                          // next tap; a class-uniform stride-2 dgrad tile steps by 2
        it_c0 = 0;
        if (cm_on) { it_ts += 2; if (it_ts >= g.KW) { it_ts = cm_s0; it_tr += 2; } }
        else { ++it_ts; if (it_ts == g.KW) { it_ts = 0; ++it_tr; } }
      }
    } else {
      k0 = kt * GBK;
#pragma unroll
      for (int j = 0; j < AI; ++j) bload(rsA, a_vo[j], k0 * 2, sa + j * 1024);
    }
#pragma unroll
    for (int j = 0; j < BI; ++j) bload(rsB, b_vo[j], k0 * 2, sb + j * 1024);
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment addresses: row base + swizzled slot; (row & 7) == (lane & 7) because every fragment starts on a multiple of 16 rows
  const int frow = lane & 15, fkg = lane >> 4, fsw = lane & 7;
  const int a_off = (wm * WTM + frow) * ROWB;
  const int b_off = A_BYTES + (wn * 64 + frow) * ROWB;

  issue(0, 0);
  for (int t = 0; t < nk; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // tile t has landed (this wave's pieces) ...
    __syncthreads();                                      // ... everyone's pieces; and stage (t+1)&1 is no longer being read
    if (t + 1 < nk) issue(t + 1, (t + 1) & 1);
    const unsigned char* st = smem + (t & 1) * STAGE;
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
        for (int j = 0; j < FN; ++j) acc[i][j] = mfma16(bfr[j], af[i], acc[i][j]);   // swapped: lane holds 4 consecutive columns
    }
  }

  // ---------------- epilogue: fragments -> LDS (fp32, 128 rows at a time) -> whole rows ----------------
  TOut* Cp = reinterpret_cast<TOut*>(p.C) + (int64_t)batch * p.sC;
  const TOut* Rp = p.res ? reinterpret_cast<const TOut*>(p.res) + (int64_t)batch * p.sR : nullptr;
  const TOut* Mp = reinterpret_cast<const TOut*>(p.mask);
  float* ep = reinterpret_cast<float*>(smem);
  constexpr int EPITCH = BN + 4;
  constexpr int HR = BM / 2;                     // rows staged at a time: the fp32 image must fit the two operand stages
  constexpr int CH = BN / 8;
  constexpr int NCH = (HR * CH) / NT;            // chunks of 8 columns per thread and half
  const bool v_st = (p.ldc % 8 == 0) && ((reinterpret_cast<uintptr_t>(Cp) & 15) == 0);
  const bool v_res = Rp && (p.ldr % 8 == 0) && ((reinterpret_cast<uintptr_t>(Rp) & 15) == 0);
  const bool v_msk = Mp && (p.ldm % 8 == 0) && ((reinterpret_cast<uintptr_t>(Mp) & 15) == 0);
  if (depi) {
    const int mrow = wm * WTM + frow;                // + i * 16: this lane's row inside the tile
    float bq[2][8];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int n = col0 + wn * 64 + t * 32 + fkg * 8;
      if (p.bias) {
        const float4 b0 = *reinterpret_cast<const float4*>(p.bias + n), b1 = *reinterpret_cast<const float4*>(p.bias + n + 4);
        bq[t][0] = b0.x; bq[t][1] = b0.y; bq[t][2] = b0.z; bq[t][3] = b0.w; bq[t][4] = b1.x; bq[t][5] = b1.y; bq[t][6] = b1.z; bq[t][7] = b1.w;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) bq[t][e] = 0.f;
      }
    }
    // The residual / mask operands are requested two accumulator rows (eight 16-byte loads) at a time, unconditionally, from rows clamped
    // into the matrix, BEFORE the first of them is used.  The loop this replaces loaded inside `if (m >= M) continue; if (Rp) ...;
    // if (Mp) ...` and used each value at once: load -> s_waitcnt vmcnt(0) -> load -> s_waitcnt vmcnt(0) -> compute -> store, FM x 2
    // times -- up to twenty dependent round trips at the end of every tile (the tile kernels' ramp).  Not all FM rows at
    // once: with the 80 accumulator registers (AGPRs) that is 284 registers = ONE block per CU instead of two (measured: +50 %).
    if constexpr (std::is_same<TOut, bf16>::value) {
      constexpr int CHK = 2;
#pragma unroll
      for (int i0 = 0; i0 < FM; i0 += CHK) {
        bf16x8 rraw[CHK][2], mraw[CHK][2];
        int64_t mpv[CHK];
        float rsv[CHK];
#pragma unroll
        for (int ii = 0; ii < CHK; ++ii) {
          const int i = i0 + ii;
          if (i < FM) {
            const int ml = mrow + i * 16, m = min(row0 + ml, p.M - 1);
            rsv[ii] = p.rowscale ? p.rowscale[m] : 1.0f;
            int64_t mp = m;
            if constexpr (AMODE == OP_CONV) { if (p.cg.cm) mp = s_rowpix[ml]; }     // (filled from rows clamped the same way)
            mpv[ii] = mp;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              const int n = col0 + wn * 64 + t * 32 + fkg * 8;
              if (Rp) rraw[ii][t] = *reinterpret_cast<const bf16x8*>(Rp + mp * p.ldr + n);
              if (Mp) mraw[ii][t] = *reinterpret_cast<const bf16x8*>(Mp + mp * p.ldm + n);
            }
          }
        }
#pragma unroll
        for (int ii = 0; ii < CHK; ++ii) {
          const int i = i0 + ii;
          if (i < FM) {
            const int ml = mrow + i * 16, m = row0 + ml;
            const int mc = min(m, p.M - 1);
            const float rs = rsv[ii] * p.alpha;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              const int n = col0 + wn * 64 + t * 32 + fkg * 8;
              const uint32_t keep8 = p.dthresh ? drop_mask<8>(p.seed, ((uint64_t)batch * p.M + mc) * (uint64_t)p.N + n, p.dthresh) : 0xffu;
              float v[8] = {acc[i][2 * t][0], acc[i][2 * t][1], acc[i][2 * t][2], acc[i][2 * t][3],
                            acc[i][2 * t + 1][0], acc[i][2 * t + 1][1], acc[i][2 * t + 1][2], acc[i][2 * t + 1][3]};
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                float x = v[e] * rs;
                x += bq[t][e];
                if (Rp) x += (float)rraw[ii][t][e];
                if (p.act == GPV_ACT_RELU) x = fmaxf(x, 0.f);
                else if (p.act == GPV_ACT_GELU) x = gelu_erf(x);
                if (p.dthresh) x = ((keep8 >> e) & 1u) ? x * p.dscale : 0.f;
                if (Mp) x = (float)mraw[ii][t][e] > 0.f ? x : 0.f;
                v[e] = x;
              }
              if (m < p.M) Ld8<TOut>::st(Cp + mpv[ii] * p.ldc + n, v);
            }
          }
        }
      }
    }
    return;
  }
  float bv[8];
  {
    const int nb = col0 + (tid % CH) * 8;          // NT % CH == 0: a thread always finishes the same 8 columns
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
    __syncthreads();                                 // main loop / previous half done with the LDS
    if ((wm * WTM) / HR == half) {
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int r = wm * WTM + i * 16 + frow - half * HR;
#pragma unroll
        for (int j = 0; j < FN; ++j)
          *reinterpret_cast<f32x4*>(ep + r * EPITCH + wn * 64 + j * 16 + fkg * 4) = acc[i][j];
      }
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < NCH; ++g) {
      const int idx = tid + g * NT;
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

template <int AMODE, int BM, int BN, typename TOut>
__global__ __launch_bounds__(BM == 256 ? 512 : 256) void glds_kernel(GemmK p) {
  if (p.dthresh) p.seed = eff_seed(p.seed, p.seed_dev);
  glds_body<AMODE, BM, BN, TOut>(p);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Stride-1 3x3 convolution (forward / backward-data), two blocks per CU, with ONE HALO IMAGE per channel block instead of nine tap
// tiles (round 5).  The tile kernel above fetches its A operand once per tap: 9 x BM rows per 64 channels, although the nine tiles are
// the same BM + 2 (W + 1) consecutive pixel rows (x is pixel-major) shifted by r W + s.  Its main loop is bound by L2 -> LDS delivery
// (~ 20 B/clk/CU at two blocks per CU), so here:
//   * the halo image [BM + 96 rows][64 channels] of channel block c is ONE contiguous LDS-DMA (rows row0 - (W + 1) ..; out of the
//     tensor: zeros), staged once for nine taps; the weights keep their two stages (one tap x 64 channels each);
//   * the k-loop runs (channel block, tap); tap (r, s) reads its A fragments at row offset r W + s (backward-data: mirrored), with the
//     swizzle key of the SHIFTED row, and the lanes whose pixel has no such neighbour (image border) zero their fragment -- what an
//     out-of-range DMA offset did per tap;
//   * delivered bytes per 64 channels: (BM + 96) x 128 + 9 x 16 KB instead of 9 x (BM x 128 + 16 KB).
// Register epilogue, XCD-aware tile order, loaders and fragment layout are glds_body's.  W <= 47, Cin % 64 == 0, N % 128 == 0.
int g_halo_mode = tune_env("GPV_C3_HALO", 1);       // gpv_set_option(GPV_OPT_C3_HALO, .): 0 never, 1 the 160-row tiles (where it wins), 2 the 96-row tiles as well
long g_halo_launches = 0;

template <int BM>
__global__ __launch_bounds__(256) void glds_halo_kernel(GemmK p) {
  constexpr int BN = 128, NW = 4, WN = 2, WTM = BM / 2, FM = WTM / 16, FN = 4;
  constexpr int HROWS = BM + 96;
  constexpr int A_BYTES = HROWS * ROWB, B_BYTES = BN * ROWB;
  constexpr int AIH = HROWS / (8 * NW), BI = BN / (8 * NW);
  static_assert(HROWS % (8 * NW) == 0 && WTM % 16 == 0, "tile shape");
  extern __shared__ __attribute__((aligned(128))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave - wm * WN;
  int tile;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int tm = __builtin_amdgcn_readfirstlane(tile / p.tilesN);
  const int tn = tile - tm * p.tilesN;
  const int row0 = tm * BM, col0 = tn * BN;
  const ConvGeom& g = p.cg;
  const int W = g.IW, HW = g.IH * g.IW;
  const int nC = g.Cin / GBK, nsteps = nC * 9;
  const bool dg = g.dgrad != 0;

  constexpr int OOB = 0x7ffffff0;
  const int lrow = lane >> 3;
  const int lchunk = (lane & 7) ^ lrow;
  const bf16* Ab = reinterpret_cast<const bf16*>(p.A);
  const bf16* Bb = reinterpret_cast<const bf16*>(p.B);
  const auto rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(Ab), (short)0, OOB, 0x00020000);
  const auto rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(Bb), (short)0, OOB, 0x00020000);
  int a_vo[AIH], b_vo[BI];
#pragma unroll
  for (int j = 0; j < AIH; ++j) {
    const int h = wave * (AIH * 8) + j * 8 + lrow;           // halo row = pixel row0 - (W + 1) + h
    const int px = row0 - (W + 1) + h;
    a_vo[j] = (px >= 0 && px < p.M) ? (px * g.Cs + lchunk * 8) * 2 : OOB;
  }
#pragma unroll
  for (int j = 0; j < BI; ++j) {
    const int r = wave * (BI * 8) + j * 8 + lrow;
    const int rp = (r & ~31) + ((r & 15) >> 2) * 8 + ((r >> 4) & 1) * 4 + (r & 3);      // permuted output columns (glds_body's register epilogue)
    b_vo[j] = ((col0 + rp) * (int)p.ldb + lchunk * 8) * 2;
  }
  auto bload = [&](const decltype(rsA)& rs, int voff, int soff, unsigned char* l) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)l, 16, voff, soff, 0, 0);
  };
  auto issueA = [&](int c) {
    unsigned char* sa = smem + wave * (AIH * 1024);
#pragma unroll
    for (int j = 0; j < AIH; ++j) bload(rsA, a_vo[j], c * GBK * 2, sa + j * 1024);
  };
  auto issueB = [&](int c, int tap, int stage) {
    unsigned char* sb = smem + A_BYTES + stage * B_BYTES + wave * (BI * 1024);
    const int k0 = tap * g.Cin + c * GBK;
#pragma unroll
    for (int j = 0; j < BI; ++j) bload(rsB, b_vo[j], k0 * 2, sb + j * 1024);
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fkg = lane >> 4, fsw = lane & 7;
  const int a_row = wm * WTM + frow;
  const int b_off = A_BYTES + (wn * 64 + frow) * ROWB;
  // image-border bits of this lane's FM output pixels: 1 top row, 2 bottom row, 4 left column, 8 right column
  int bord[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int m = min(row0 + a_row + i * 16, p.M - 1);
    const int rem = m % HW, oh = rem / W, ow = rem - oh * W;
    bord[i] = (oh == 0 ? 1 : 0) | (oh == g.IH - 1 ? 2 : 0) | (ow == 0 ? 4 : 0) | (ow == W - 1 ? 8 : 0);
  }

  issueA(0);
  issueB(0, 0, 0);
  int c = 0, tap = 0;
  for (int t = 0; t < nsteps; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int tap_n = tap == 8 ? 0 : tap + 1, c_n = tap == 8 ? c + 1 : c;
    if (t + 1 < nsteps) issueB(c_n, tap_n, (t + 1) & 1);
    const int r = tap / 3, s2 = tap - r * 3;
    const int rr = dg ? 2 - r : r, ss = dg ? 2 - s2 : s2;            // offset of the input pixel: (rr - 1, ss - 1)
    const int shift = rr * W + ss;
    const int tmask = (rr == 0 ? 1 : 0) | (rr == 2 ? 2 : 0) | (ss == 0 ? 4 : 0) | (ss == 2 ? 8 : 0);
    const int hkey = (frow + shift) & 7;                             // swizzle key of the shifted halo row (WTM, 16 i: multiples of 8)
    const unsigned char* sa = smem + (a_row + shift) * ROWB;
    const unsigned char* sb = smem + (t & 1) * B_BYTES + b_off;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int slot_b = ((kk * 4 + fkg) ^ fsw) << 4, slot_a = ((kk * 4 + fkg) ^ hkey) << 4;
      bf16x8 af[FM], bfr[FN];
#pragma unroll
      for (int j = 0; j < FN; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(sb + j * 16 * ROWB + slot_b);
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        af[i] = *reinterpret_cast<const bf16x8*>(sa + i * 16 * ROWB + slot_a);
        if (bord[i] & tmask) af[i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      }
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = mfma16(bfr[j], af[i], acc[i][j]);
    }
    if (tap == 8 && c + 1 < nC) {                                    // every wave is done with this channel block's halo image
      __syncthreads();
      issueA(c + 1);
    }
    tap = tap_n; c = c_n;
  }

  // ---------------- register epilogue (glds_body's, bf16, whole column tiles) ----------------
  bf16* Cp = reinterpret_cast<bf16*>(p.C);
  const bf16* Rp = reinterpret_cast<const bf16*>(p.res);
  const bf16* Mp = reinterpret_cast<const bf16*>(p.mask);
  float bq[2][8];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int n = col0 + wn * 64 + t * 32 + fkg * 8;
    if (p.bias) {
      const float4 b0 = *reinterpret_cast<const float4*>(p.bias + n), b1 = *reinterpret_cast<const float4*>(p.bias + n + 4);
      bq[t][0] = b0.x; bq[t][1] = b0.y; bq[t][2] = b0.z; bq[t][3] = b0.w; bq[t][4] = b1.x; bq[t][5] = b1.y; bq[t][6] = b1.z; bq[t][7] = b1.w;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) bq[t][e] = 0.f;
    }
  }
  constexpr int CHK = 2;
#pragma unroll
  for (int i0 = 0; i0 < FM; i0 += CHK) {
    bf16x8 rraw[CHK][2], mraw[CHK][2];
    float rsv[CHK];
#pragma unroll
    for (int ii = 0; ii < CHK; ++ii) {
      const int i = i0 + ii;
      if (i < FM) {
        const int64_t m = min(row0 + a_row + i * 16, p.M - 1);
        rsv[ii] = p.rowscale ? p.rowscale[m] : 1.0f;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int n = col0 + wn * 64 + t * 32 + fkg * 8;
          if (Rp) rraw[ii][t] = *reinterpret_cast<const bf16x8*>(Rp + m * p.ldr + n);
          if (Mp) mraw[ii][t] = *reinterpret_cast<const bf16x8*>(Mp + m * p.ldm + n);
        }
      }
    }
#pragma unroll
    for (int ii = 0; ii < CHK; ++ii) {
      const int i = i0 + ii;
      if (i < FM) {
        const int m = row0 + a_row + i * 16;
        const float rs = rsv[ii] * p.alpha;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int n = col0 + wn * 64 + t * 32 + fkg * 8;
          float v[8] = {acc[i][2 * t][0], acc[i][2 * t][1], acc[i][2 * t][2], acc[i][2 * t][3],
                        acc[i][2 * t + 1][0], acc[i][2 * t + 1][1], acc[i][2 * t + 1][2], acc[i][2 * t + 1][3]};
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float x = v[e] * rs + bq[t][e];
            if (Rp) x += (float)rraw[ii][t][e];
            if (p.act == GPV_ACT_RELU) x = fmaxf(x, 0.f);
            if (Mp) x = (float)mraw[ii][t][e] > 0.f ? x : 0.f;
            v[e] = x;
          }
          if (m < p.M) Ld8<bf16>::st(Cp + (int64_t)m * p.ldc + n, v);
        }
      }
    }
  }
}

template <int BM>
int launch_halo(const GemmK& k, hipStream_t st) {
  constexpr size_t lds = (size_t)(BM + 96) * ROWB + (size_t)2 * 128 * ROWB;
  GemmK p = k;
  const int tilesM = (p.M + BM - 1) / BM;
  p.tilesN = p.N / 128;
  auto fn = glds_halo_kernel<BM>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }
    attr_done = true;
  }
  ++g_halo_launches;
  ++g_glds_launches;                  // (a launch of the two-per-CU tile family: tests that ask whether a shape took it count these too)
  hipLaunchKernelGGL(fn, dim3(tilesM * p.tilesN), dim3(256), lds, st, p);
  GPV_CHECK_LAUNCH();
  return 0;
}

inline bool halo_ok(const GemmK& k, int dtype_out, int batch) {
  const ConvGeom& g = k.cg;
  auto a16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  return g_halo_mode != 0 && dtype_out == GPV_BF16 && batch == 1 && g.KH == 3 && g.KW == 3 && g.SH == 1 && g.SW == 1 && g.PH == 1 && g.PW == 1 &&
         !g.cm && g.OH == g.IH && g.OW == g.IW && g.IW <= 47 && g.Cin % GBK == 0 && k.N % 128 == 0 && k.K == 9 * g.Cin &&
         k.M % (g.IH * g.IW) == 0 && k.ldc % 8 == 0 && a16(k.C) && (!k.res || (k.ldr % 8 == 0 && a16(k.res))) &&
         (!k.mask || (k.ldm % 8 == 0 && a16(k.mask))) && (!k.bias || a16(k.bias)) && !k.dthresh && k.act != GPV_ACT_GELU;
}

// 1x1 stride-1 convolutions are plain GEMMs over the NHWC rows; their own kernel name keeps them attributable to the
// backbone in rocprofv3 traces and PMC passes (like conv1x1_kernel in gemm.hip)
template <int BM, int BN, typename TOut>
__global__ __launch_bounds__(BM == 256 ? 512 : 256) void glds_conv1x1_kernel(GemmK p) {
  if (p.dthresh) p.seed = eff_seed(p.seed, p.seed_dev);
  glds_body<OP_PLAIN, BM, BN, TOut>(p);
}

template <int AMODE, int BM, int BN, typename TOut>
int launch_glds(const GemmK& k, int batch, hipStream_t st) {
  constexpr size_t stage = (size_t)2 * (BM + BN) * ROWB;
  constexpr size_t epi = (size_t)(BM / 2) * (BN + 4) * 4;
  constexpr size_t lds = stage > epi ? stage : epi;
  GemmK p = k;
  const int tilesM = (p.M + BM - 1) / BM;
  p.tilesN = (p.N + BN - 1) / BN;
  const bool c11 = AMODE == OP_PLAIN && p.conv1x1;
  auto fn = c11 ? glds_conv1x1_kernel<BM, BN, TOut> : glds_kernel<AMODE, BM, BN, TOut>;
  static bool attr_done[2] = {false, false};
  if (!attr_done[c11]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }      // (leave no stale error behind for the next launch check)
    attr_done[c11] = true;
  }
  static const int depi = tune_env("GPV_GLDS_DEPI", 1);
  p.depi = depi;
  dim3 grid(tilesM * p.tilesN, 1, batch);
  ++g_glds_launches;
  hipLaunchKernelGGL(fn, grid, dim3(BM == 256 ? 512 : 256), lds, st, p);
  GPV_CHECK_LAUNCH();
  return 0;
}

template <int AMODE, int BM, int BN>
int launch_glds_out(const GemmK& k, int dtype_out, int batch, hipStream_t st) {
  if (dtype_out == GPV_BF16) return launch_glds<AMODE, BM, BN, bf16>(k, batch, st);
  return launch_glds<AMODE, BM, BN, float>(k, batch, st);
}

}  // namespace
int g_kernel_forced = 0;      // bit 0: GPV_OPT_GLDS >= 2, bit 1: GPV_OPT_PIPE >= 100 -- a kernel family is being forced (tests / tuning): the small-problem side paths step aside
int g_two_per_cu = tune_env("GPV_TWO_PER_CU", 1);
int g_bm96_fill = tune_env("GPV_BM96", 1);       // 96-row tiles for <= 1024-row GEMMs (gemm_common.h two_per_cu_bm)
namespace {
int g_glds_mode = tune_env("GPV_GLDS", 1);   // gpv_set_option(GPV_OPT_GLDS, .)

inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

}  // namespace

int glds_try_launch(const GemmK& k, int amode, int dtype_in, int dtype_out, int batch, hipStream_t st) {
  const int mode = g_glds_mode;     // 0 = never, 1 (default) = where it is expected to win, 2 / 3 = the 8-wave / 4-wave variant wherever legal
  if (mode == 0 || dtype_in != GPV_BF16) return -1;
  if (k.accumulate || k.split_k > 1) return -1;
  if (k.K % GBK != 0 || k.K < GBK || k.N <= 64) return -1;
  if (!al16(k.A) || !al16(k.B) || k.ldb % 8 != 0 || (batch > 1 && (k.sA % 8 != 0 || k.sB % 8 != 0))) return -1;
  // 32-bit byte offsets into each operand (one buffer descriptor per operand and batch element)
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
  // one-tile-per-CU launches (layer3 / layer4 at B = 32: 240 tiles of 160 x 256) serialise load ramp, main loop and store burst:
  // ~16 us of fixed cost on a 33-57 us kernel (tools/bench_ktile.py).  Half-width tiles at two blocks per CU (160 x 128: 480
  // tiles, 96 x 128 for the 9600-row maps) let one block's epilogue run under the other's main loop.
  if (glds_two_per_cu(k, batch) && mode != 2 && mode != 3) {
    const int bm2 = two_per_cu_bm(k, batch);
    if (amode == OP_CONV && halo_ok(k, dtype_out, batch)) {
      if (bm2 == 160) return launch_halo<160>(k, st);
      if (bm2 == 96 && g_halo_mode >= 2) return launch_halo<96>(k, st);     // 96-row tiles (layer4): forward 57.2 against 55.0 us, backward-data equal (tools/bench_c3_halo.py) -- tests only
    }
    if (bm2 == 160) return amode == OP_CONV ? launch_glds_out<OP_CONV, 160, 128>(k, dtype_out, batch, st) : launch_glds_out<OP_PLAIN, 160, 128>(k, dtype_out, batch, st);
    if (bm2 == 96) return amode == OP_CONV ? launch_glds_out<OP_CONV, 96, 128>(k, dtype_out, batch, st) : launch_glds_out<OP_PLAIN, 96, 128>(k, dtype_out, batch, st);
  }
  const int bn = k.N > 128 ? 256 : 128;
  bool small_tiles = mode == 3;
  if (mode == 1) {
    // Measured per shape on the ResNet-50 / transformer launches at B=32 (tools/bench_conv.py, bench_dgrad.py,
    // bench_gemm_sq.py with GPV_GLDS=0/2/3).  At 1-2 256-row tiles per CU the 8-wave kernel loses to kernels with 2-3
    // co-resident blocks (epilogues overlapped with other blocks' main loops, finer tail); it keeps the launches that fill the
    // chip several times over.  The 4-wave 128x128 variant wins wherever the reduction is long enough to amortise its
    // prologue/epilogue: every K >= 1024 conv / GEMM (layer3/4 3x3 fwd 94->82 / 150->76 us, their stride-2 dgrads
    // 194->139 / 161->101 us) and the K = 512 forward 1x1s; ReLU-mask (dgrad) epilogues need K >= 1024.
    const int64_t tiles = (int64_t)((k.M + GBM - 1) / GBM) * ((k.N + bn - 1) / bn) * batch;
    const bool huge = tiles >= 1024 && k.K >= 512 && !(amode == OP_CONV && k.cg.cm);   // parity classes: too unbalanced for 1 block/CU
    small_tiles = !huge && (k.K >= 1024 || (k.K >= 512 && !k.mask));
    if (!huge && !small_tiles) return -1;
  }
  if (small_tiles)   // 4-wave 128x128 variant, two blocks per CU
    return amode == OP_CONV ? launch_glds_out<OP_CONV, 128, 128>(k, dtype_out, batch, st) : launch_glds_out<OP_PLAIN, 128, 128>(k, dtype_out, batch, st);
  if (amode == OP_CONV) return bn == 256 ? launch_glds_out<OP_CONV, 256, 256>(k, dtype_out, batch, st) : launch_glds_out<OP_CONV, 256, 128>(k, dtype_out, batch, st);
  return bn == 256 ? launch_glds_out<OP_PLAIN, 256, 256>(k, dtype_out, batch, st) : launch_glds_out<OP_PLAIN, 256, 128>(k, dtype_out, batch, st);
}

}  // namespace gpvk

namespace gpvk { int attn_bwd1_mode(int set); long attn_bwd1_launches(long set); }   // attention.hip

extern "C" int gpv_set_option(int option, int value) {
  if (option == GPV_OPT_C1S_LAUNCHES) {
    const long prev = gpvk::g_c1s_launches;
    gpvk::g_c1s_launches = value;
    return (int)prev;
  }
  if (option == GPV_OPT_ATTN_BWD1) return gpvk::attn_bwd1_mode(value);
  if (option == GPV_OPT_ATTN_BWD1_LAUNCHES) return (int)gpvk::attn_bwd1_launches(value);
  if (option == GPV_OPT_GLDS) {
    const int prev = gpvk::g_glds_mode;
    gpvk::g_glds_mode = value;
    gpvk::g_kernel_forced = (gpvk::g_kernel_forced & ~1) | (value >= 2 ? 1 : 0);
    return prev;
  }
  if (option == GPV_OPT_SKINNY) {
    const int prev = gpvk::g_skinny_mode;
    gpvk::g_skinny_mode = value;
    return prev;
  }
  if (option == GPV_OPT_GEMV) {
    const int prev = gpvk::g_gemv_mode;
    gpvk::g_gemv_mode = value;
    return prev;
  }
  if (option == GPV_OPT_GEMV_LAUNCHES) {
    const long prev = gpvk::g_gemv_launches;
    gpvk::g_gemv_launches = value;
    return (int)prev;
  }
  if (option == GPV_OPT_WG8) {
    const int prev = gpvk::g_wg8_mode;
    gpvk::g_wg8_mode = value;
    return prev;
  }
  if (option == GPV_OPT_PIPE_SMALL) return gpvk::pipe_set_small(value);
  if (option == GPV_OPT_WG8H) {
    const int prev = gpvk::g_wg8h_mode;
    gpvk::g_wg8h_mode = value;
    return prev;
  }
  if (option == GPV_OPT_W8L) {
    const int prev = gpvk::g_w8l_mode;
    gpvk::g_w8l_mode = value;
    return prev;
  }
  if (option == GPV_OPT_WG8_LAUNCHES) {
    const long prev = gpvk::g_wg8_launches;
    gpvk::g_wg8_launches = value;
    return (int)prev;
  }
  if (option == GPV_OPT_GLDS_WGRAD) {
    const int prev = gpvk::g_wgrad_mode;
    gpvk::g_wgrad_mode = value;
    return prev;
  }
  if (option == GPV_OPT_C1S) {
    const int prev = gpvk::g_c1s_mode;
    gpvk::g_c1s_mode = value;
    return prev;
  }
  if (option == GPV_OPT_C3S) {
    const int prev = gpvk::g_c3s_mode;
    gpvk::g_c3s_mode = value;
    return prev;
  }
  if (option == GPV_OPT_C3S_LAUNCHES) {
    const long prev = gpvk::g_c3s_launches;
    gpvk::g_c3s_launches = value;
    return (int)prev;
  }
  if (option == GPV_OPT_PIPE) {
    gpvk::g_kernel_forced = (gpvk::g_kernel_forced & ~2) | (value >= 100 ? 2 : 0);
    return gpvk::pipe_set_mode(value);
  }
  if (option == GPV_OPT_PIPE_LAUNCHES) return (int)gpvk::pipe_launches(value);
  if (option == GPV_OPT_C3_HALO) {
    const int prev = gpvk::g_halo_mode;
    gpvk::g_halo_mode = value;
    return prev;
  }
  if (option == GPV_OPT_C3_HALO_LAUNCHES) {
    const long prev = gpvk::g_halo_launches;
    gpvk::g_halo_launches = value;
    return (int)prev;
  }
  if (option == GPV_OPT_GLDS_LAUNCHES) {
    const long prev = gpvk::g_glds_launches;
    gpvk::g_glds_launches = value;
    return (int)prev;
  }
  return -1;
}
