// Streaming 3x3 convolution (gfx950): the stride-1 3x3 convolutions of ResNet layer1 / layer2 (64 or 128 input channels, 10^5 ..
// 6 10^5 output pixels at the training batch), forward and backward-data, and the stride-2 backward-data of layer2's first block
// (exp/gpv/models/backbone.py:93-95 -> torchvision Bottleneck.conv2).
//
//   y[px, n] = epilogue( sum_{r,s,c} x[pix(px, r, s), c] W[n, r, s, c] ),   bf16, Cin in {64, 128}, 64 output channels per block
//   epilogue = + bias[n] -> ReLU -> * (mask[px, n] > 0)
//
// The tile kernels run these launches at 0.16-0.25 of their own roofline (layer1: 119 us for 157 MB / 45 GFLOP, layer2 73 / 83 us):
// 128x64 tiles with one barrier per k-tile, and -- what bounds every implicit-GEMM formulation here -- nine trips of every input
// pixel through the vector-memory path, which sustains 12-20 B/clk/CU from L2 (a first version of this file gathered the taps
// straight from global memory per 32-pixel tile, weights in LDS, no barriers: 93-103 us, the same wall).
//   * the weights of 64 output channels (all 9 taps: 72 KB at Cin = 64, 144 KB at Cin = 128) are staged ONCE per block into LDS;
//     a layer with 128 output channels is two 64-channel slices on gridDim.y that share the input through L2;
//   * v_mfma_f32_32x32x16_bf16 with the weights as the row operand: one ds_read_b128 of weights per MFMA (8 passes), half the
//     LDS bytes per flop of the 16x16x32 shape; no barrier after the prologue;
//   * the input passes through the vector-memory path ONCE (see c3r_kernel): rows live in registers, tap columns are DPP shifts;
//   * padding rows / columns and the tails are hardware zero fill (buffer loads with an out-of-range offset);
//   * the output channels are PERMUTED when W is staged so that a lane ends up with two runs of 8 consecutive channels per
//     32-channel MFMA tile: mask and output are plain 16-byte accesses in the accumulator layout, no LDS round trip.
// fp32 accumulation order per output: taps in (r, kc, s) order -- results differ from the tile kernels by summation order only.
// Measured (B = 32, tools/bench_body.py): layer1 conv2 119 -> 51 us, layer2 conv2 73 -> 51 us, its backward-data 83 -> 54 us.
#include "gemm_common.h"

namespace gpvk {
namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int C3_OOB = 0x7ffffff0;
constexpr int C3_NSL = 64;              // output channels per block

// MFMA row m (0..31) of a 32-channel tile <-> channel offset: lane half h = (m >> 2) & 1 owns rows 8 q + 4 h + j, which must be
// the channels 16 (q >> 1) + 8 h + 4 (q & 1) + j (two runs of 8 consecutive channels per lane): swap bits 2 and 3 of m
__device__ __forceinline__ int c3_perm(int m) { return (m & 0x13) | ((m & 4) << 1) | ((m & 8) >> 1); }

// A 16-byte store the compiler cannot see.  With a visible store in the row loop the waitcnt pass holds loads and stores pending together,
// treats the vm counter as out of order and waits vmcnt(0) before every tile's stores -- for the ReLU-mask loads of the whole row AND for
// the previous tile's stores (four serialised store round trips per dy row in the stride-2 backward-data).  s_nop 1: the two wait states a
// 16-byte store's data registers need before a VALU write (conv1x1_dual.hip).
__device__ __forceinline__ void c3_store16(bf16* q, const bf16x8& o) {
  typedef unsigned int c3_u32x4 __attribute__((ext_vector_type(4)));
  const c3_u32x4 ov = __builtin_bit_cast(c3_u32x4, o);
  asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(q), "v"(ov) : "memory");
}

// Row-walking kernel (stride 1, forward and backward-data).  The input passes through the vector memory path ONCE: a wave owns a strip of 30 output columns and walks DOWN it; the three input rows a 3x3 needs live in REGISTERS
// (32 lanes = input columns x0-1 .. x0+30, two 8-channel groups per 16-channel chunk on the two lane halves), each new output
// row loads one new input row (Cin/16 loads per lane) into the slot of the row that just left the window, and the column
// shifts of the taps are DPP wave shifts of those registers (v_mov_b32 wave_shl:1 -- lane i takes lane i+1; the lanes that
// pick up a neighbour from the wrong half only feed the two discarded columns 30, 31).  Weights: LDS, as above.
template <int CIN, bool DGRAD, bool MASK, int WAVES>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu((CIN >= 128 && WAVES == 8) ? 2 : 1, (CIN >= 128 && WAVES == 8) ? 2 : 8)))
void c3r_kernel(GemmK p, int rows_per_item, int nstrip, int nseg, int nitems) {
  constexpr int KTOT = 9 * CIN, KP = KTOT + 8, KCN = CIN / 16, SWO = 30;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16* Wl = reinterpret_cast<bf16*>(smem_raw);
  float* bias_l = reinterpret_cast<float*>(Wl + (size_t)C3_NSL * KP);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5, pl = lane & 31;
  const int cbase = (int)blockIdx.y * C3_NSL;
  const ConvGeom& g = p.cg;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), (short)0, C3_OOB, 0x00020000);
  bf16* Y = reinterpret_cast<bf16*>(p.C) + cbase;
  const bf16* Mk = reinterpret_cast<const bf16*>(p.mask) + cbase;

  int i_hi, i_step, item;
  {
    const int nb = gridDim.x;
    if ((nb & 7) == 0) {
      const int per = (nitems + 7) >> 3, xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, nbx = nb >> 3;
      i_hi = min(nitems, (xcd + 1) * per);
      i_step = nbx * WAVES;
      item = xcd * per + lb * WAVES + wave;
    } else {
      i_hi = nitems; i_step = nb * WAVES;
      item = blockIdx.x * WAVES + wave;
    }
  }
  {
    const bf16* Wg = reinterpret_cast<const bf16*>(p.B);
    constexpr int SL = KTOT / 8;
    stage_chunks16<WAVES * 64, 8>(C3_NSL * SL, tid,
        [&](int idx) { const int L = idx / SL, sl = idx - L * SL; return Wg + (int64_t)(cbase + (L & ~31) + c3_perm(L & 31)) * KTOT + sl * 8; },
        [&](int idx) { const int L = idx / SL, sl = idx - L * SL; return Wl + L * KP + sl * 8; });
    for (int L = tid; L < C3_NSL; L += WAVES * 64) bias_l[L] = p.bias ? p.bias[cbase + (L & ~31) + c3_perm(L & 31)] : 0.f;
  }
  __syncthreads();

  const bf16* wlane = Wl + pl * KP + h * 8;
  u32x4 row[3][KCN];
  for (; item < i_hi; item += i_step) {
    // item -> (image, strip, row segment); segments of a strip are consecutive items (they share two halo rows through L2)
    const int seg = item % nseg, t1 = item / nseg, strip = t1 % nstrip, b = t1 / nstrip;
    const int oh0 = seg * rows_per_item, x0 = strip * SWO;
    const int xin = x0 - 1 + pl;                                     // this lane's input column
    const bool xok = (unsigned)xin < (unsigned)g.IW;
    const int colo = ((b * g.IH) * g.IW + xin) * g.Cs * 2 + h * 16;       // byte offset of (b, row 0, xin), this lane's 8-channel group
    const int rowb = g.IW * g.Cs * 2;
    auto load_row = [&](int ih, int slot) {
      const int vo = (xok && (unsigned)ih < (unsigned)g.IH) ? colo + ih * rowb : C3_OOB;
#pragma unroll
      for (int kc = 0; kc < KCN; ++kc) row[slot][kc] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, kc * 32, 0);
    };
    load_row(oh0 - 1, 0);
    load_row(oh0, 1);
    load_row(oh0 + 1, 2);
    const bool cok = pl < SWO && x0 + pl < g.OW;
    const int rows_here = min(rows_per_item, g.OH - oh0);
    for (int j0 = 0; j0 < rows_here; j0 += 3) {
#pragma unroll
      for (int jj = 0; jj < 3; ++jj) {
        const int oh = oh0 + j0 + jj;
        const bool sok = cok && j0 + jj < rows_here;
        // the weight fragments are the same LDS words for every output row: without this the compiler hoists all 72-144 reads
        // out of the row loops into registers (200-700 spilled VGPRs)
        int wvo = 0;
        asm volatile("" : "+v"(wvo));
        const bf16* wl = wlane + wvo;
        const int opix = (b * g.OH + oh) * g.OW + x0 + pl;
        f32x16 acc[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 bv = *reinterpret_cast<const float4*>(bias_l + nt * 32 + q * 8 + h * 4);
            acc[nt][4 * q] = bv.x; acc[nt][4 * q + 1] = bv.y; acc[nt][4 * q + 2] = bv.z; acc[nt][4 * q + 3] = bv.w;
          }
        }
        bf16x8 mv[MASK ? 4 : 1];
        if constexpr (MASK) {
          // unconditional, from pixel 0 for the lanes that store nothing: `sok ? load : 0` is a branch per load
          const int64_t mpix = sok ? opix : 0;
#pragma unroll
          for (int c = 0; c < 4; ++c)
            mv[c] = *reinterpret_cast<const bf16x8*>(Mk + mpix * p.ldm + (c >> 1) * 32 + (c & 1) * 16 + h * 8);
        }
        if constexpr (CIN >= 128 && WAVES == 8) {
          // One block per CU = two waves per SIMD: the six weight fragments of the NEXT (tap row, channel chunk) are read while the six
          // MFMAs of this one run, pinned with sched_barrier.  Left alone hipcc reads a fragment, waits lgkmcnt(0) and multiplies --
          // 115 of the 144 MFMAs of an output row started behind a full LDS round trip (halving the reads changed nothing: it is the
          // latency, not the LDS bandwidth; waves_per_eu(2, 2) alone did not change the schedule either).
          constexpr int NGR = 3 * KCN;
          bf16x8 wfg[2][3][2];
          auto ldg = [&](int grp, int b) {
            const int r = grp / KCN, kc = grp - r * KCN;
#pragma unroll
            for (int s2 = 0; s2 < 3; ++s2) {
              const int tap = DGRAD ? 8 - (r * 3 + s2) : r * 3 + s2;
#pragma unroll
              for (int nt = 0; nt < 2; ++nt) wfg[b][s2][nt] = *reinterpret_cast<const bf16x8*>(wl + nt * 32 * KP + tap * CIN + kc * 16);
            }
          };
          ldg(0, 0);
#pragma unroll
          for (int grp = 0; grp < NGR; ++grp) {
            const int r = grp / KCN, kc = grp - r * KCN, slot = (jj + r) % 3;
            if (grp + 1 < NGR) ldg(grp + 1, (grp + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            u32x4 a = row[slot][kc];
#pragma unroll
            for (int s2 = 0; s2 < 3; ++s2) {
              if (s2 > 0) {
#pragma unroll
                for (int d = 0; d < 4; ++d) a[d] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)a[d], 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
              }
#pragma unroll
              for (int nt = 0; nt < 2; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfg[grp & 1][s2][nt], __builtin_bit_cast(bf16x8, a), acc[nt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (grp == KCN - 1) load_row(oh + 2, jj % 3);   // the row that left the window makes room for the next one
          }
        } else {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const int slot = (jj + r) % 3;
#pragma unroll
          for (int kc = 0; kc < KCN; ++kc) {
            u32x4 a = row[slot][kc];
#pragma unroll
            for (int s2 = 0; s2 < 3; ++s2) {
              if (s2 > 0) {
#pragma unroll
                for (int d = 0; d < 4; ++d) a[d] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)a[d], 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
              }
              const int tap = DGRAD ? 8 - (r * 3 + s2) : r * 3 + s2;
#pragma unroll
              for (int nt = 0; nt < 2; ++nt) {
                const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wl + nt * 32 * KP + tap * CIN + kc * 16);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, __builtin_bit_cast(bf16x8, a), acc[nt], 0, 0, 0);
              }
            }
          }
          if (r == 0) load_row(oh + 2, jj % 3);           // the row that left the window makes room for the next one
        }
        }
        if (sok) {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int nt = c >> 1, hi = c & 1;
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              float x = acc[nt][hi * 8 + e];
              if (p.act == GPV_ACT_RELU) x = fmaxf(x, 0.f);
              if constexpr (MASK) x = (float)mv[c][e] > 0.f ? x : 0.f;
              o[e] = (bf16)x;
            }
            c3_store16(Y + (int64_t)opix * p.ldc + nt * 32 + hi * 16 + h * 8, o);
          }
        }
      }
    }
  }
}

template <int CIN, bool DGRAD, bool MASK, int WAVES>
int c3r_launch_w(const GemmK& k, hipStream_t st) {
  constexpr int KP = 9 * CIN + 8;
  const size_t lds = (size_t)C3_NSL * KP * 2 + C3_NSL * sizeof(float);
  auto fn = c3r_kernel<CIN, DGRAD, MASK, WAVES>;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }      // (leave no stale error behind for the next launch check)
    attr = true;
  }
  const ConvGeom& g = k.cg;
  const int nsl = k.N / C3_NSL, Bn = k.M / (g.OH * g.OW);
  const int nstrip = (g.OW + 29) / 30;
  static const int env_blocks = tune_env("GPV_C3S_BLOCKS", 0);
  static const int env_rows = tune_env("GPV_C3R_ROWS", 0);
  int blocks = env_blocks > 0 ? env_blocks : ((CIN == 64 ? 512 : 256) / nsl);
  // rows per item: a multiple of 3 such that the items fill the resident waves about twice (each segment re-reads two halo rows)
  int rows = env_rows > 0 ? env_rows : 3;
  if (env_rows <= 0) {
    const int64_t waves = (int64_t)blocks * WAVES;
    while (rows < 30 && (int64_t)Bn * nstrip * ((g.OH + rows + 2) / (rows + 3)) >= 2 * waves) rows += 3;
  }
  const int nseg = (g.OH + rows - 1) / rows;
  const int nitems = Bn * nstrip * nseg;
  while (blocks > 8 && (int64_t)(blocks - 8) * WAVES >= nitems) blocks -= 8;
  hipLaunchKernelGGL(fn, dim3(blocks, nsl), dim3(WAVES * 64), lds, st, k, rows, nstrip, nseg, nitems);
  GPV_CHECK_LAUNCH();
  return 0;
}

template <int CIN, bool DGRAD, bool MASK>
int c3r_launch(const GemmK& k, hipStream_t st) {
  static const int env_waves = tune_env("GPV_C3S_WAVES", 0);
  if (env_waves == 4) return c3r_launch_w<CIN, DGRAD, MASK, 4>(k, st);
  if (env_waves == 16) return c3r_launch_w<CIN, DGRAD, MASK, 16>(k, st);
  return c3r_launch_w<CIN, DGRAD, MASK, 8>(k, st);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Stride-2 backward-data (the 3x3 / 2 of a stage's first block: dx [2 IH x 2 IW] from dy [IH x IW]), same row-walking scheme.
// dx(oh, ow) only receives the taps with (oh + 1 - r) and (ow + 1 - s) even:
//   even row 2y   : r = 1 <- dy row y                   odd row 2y+1  : r = 2 <- dy row y,  r = 0 <- dy row y + 1
//   even col 2c   : s = 1 <- dy col c                   odd col 2c+1  : s = 2 <- dy col c,  s = 0 <- dy col c + 1 (lane shift)
// A wave holds 32 dy columns of the dy rows y and y + 1 in registers and writes, for every y, the four parity tiles of the dx rows
// 2y, 2y+1 (1 + 2 + 2 + 4 = 9 tap products, 31 of the 32 columns: lane 31 only lends its column to lane 30's odd outputs).  The
// tile kernels ran this launch (layer2.0 conv2, B = 32) at 213 us by parity-class tiles with gathered rows; its bytes (dy 39 MB,
// mask + dx 2 x 157 MB) take 56 us at the measured 6.3 TB/s.
// BITS (round 6): the ReLU mask as one bit per element (gpv_conv_args.relu_mask_bits, 128 channels: bit c & 7 of byte 4 ((c % 32) / 8) + c / 32
// of the pixel's 16 bytes): ONE 16-byte load per output pixel and lane instead of four.
template <int CIN, bool MASK, int WAVES, bool BITS = false>
__global__ __launch_bounds__(WAVES * 64) void c3d2_kernel(GemmK p, int rows_per_item, int nstrip, int nseg, int nitems) {
  constexpr int KTOT = 9 * CIN, KP = KTOT + 8, KCN = CIN / 16, SWD = 31;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16* Wl = reinterpret_cast<bf16*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5, pl = lane & 31;
  const int cbase = (int)blockIdx.y * C3_NSL;
  const ConvGeom& g = p.cg;                      // IH x IW = dy, OH x OW = dx = 2 IH x 2 IW
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), (short)0, C3_OOB, 0x00020000);
  bf16* Y = reinterpret_cast<bf16*>(p.C) + cbase;
  const bf16* Mk = reinterpret_cast<const bf16*>(p.mask) + cbase;
  int i_hi, i_step, item;
  {
    const int nb = gridDim.x;
    if ((nb & 7) == 0) {
      const int per = (nitems + 7) >> 3, xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, nbx = nb >> 3;
      i_hi = min(nitems, (xcd + 1) * per);
      i_step = nbx * WAVES;
      item = xcd * per + lb * WAVES + wave;
    } else {
      i_hi = nitems; i_step = nb * WAVES;
      item = blockIdx.x * WAVES + wave;
    }
  }
  {
    const bf16* Wg = reinterpret_cast<const bf16*>(p.B);
    constexpr int SL = KTOT / 8;
    stage_chunks16<WAVES * 64, 8>(C3_NSL * SL, tid,
        [&](int idx) { const int L = idx / SL, sl = idx - L * SL; return Wg + (int64_t)(cbase + (L & ~31) + c3_perm(L & 31)) * KTOT + sl * 8; },
        [&](int idx) { const int L = idx / SL, sl = idx - L * SL; return Wl + L * KP + sl * 8; });
  }
  __syncthreads();
  const bf16* wlane = Wl + pl * KP + h * 8;
  u32x4 row[2][KCN];
  for (; item < i_hi; item += i_step) {
    const int seg = item % nseg, t1 = item / nseg, strip = t1 % nstrip, b = t1 / nstrip;
    const int y0 = seg * rows_per_item, x0 = strip * SWD;
    const int xd = x0 + pl;                                           // this lane's dy column
    const bool xok = xd < g.IW;
    const int colo = ((b * g.IH) * g.IW + xd) * g.Cs * 2 + h * 16;
    const int rowb = g.IW * g.Cs * 2;
    auto load_row = [&](int ih, int slot) {
      const int vo = (xok && (unsigned)ih < (unsigned)g.IH) ? colo + ih * rowb : C3_OOB;
#pragma unroll
      for (int kc = 0; kc < KCN; ++kc) row[slot][kc] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, kc * 32, 0);
    };
    load_row(y0, 0);
    load_row(y0 + 1, 1);
    const bool cok = pl < SWD && xd < g.IW;
    const int rows_here = min(rows_per_item, g.IH - y0);
    for (int j0 = 0; j0 < rows_here; j0 += 2) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        if (j0 + jj < rows_here) {                                    // (uniform)
          const int y = y0 + j0 + jj;
          // four parity tiles: (py, px); taps of tile = {r in R(py)} x {s in S(px)}; r = 1 | {2, 0} reads slot jj | {jj, jj ^ 1}
          // the ReLU masks of all four tiles are requested up front: the first tile has only 16 MFMAs to hide an HBM read behind
          bf16x8 mv[(MASK && !BITS) ? 16 : 1];
          u32x4 mbits[(MASK && BITS) ? 4 : 1];
          if constexpr (MASK && BITS) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const int opix = (b * g.OH + 2 * y + (t >> 1)) * g.OW + 2 * xd + (t & 1);
              mbits[t] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(p.mask_bits) + (int64_t)(cok ? opix : 0) * 16);
            }
          } else if constexpr (MASK) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const int opix = (b * g.OH + 2 * y + (t >> 1)) * g.OW + 2 * xd + (t & 1);
#pragma unroll
              for (int c = 0; c < 4; ++c)       // (unconditional: pixel 0 for the lanes that store nothing)
                mv[t * 4 + c] = *reinterpret_cast<const bf16x8*>(Mk + (int64_t)(cok ? opix : 0) * p.ldm + (c >> 1) * 32 + (c & 1) * 16 + h * 8);
            }
          }
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int py = t >> 1, px = t & 1;
            int wvo = 0;
            asm volatile("" : "+v"(wvo));                             // (keeps the weight reads inside the loops, see c3r_kernel)
            const bf16* wl = wlane + wvo;
            const int opix = (b * g.OH + 2 * y + py) * g.OW + 2 * xd + px;
            f32x16 acc[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
              for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;
#pragma unroll
            for (int ri = 0; ri <= py; ++ri) {
              const int r = py ? (ri ? 0 : 2) : 1;                    // odd rows: r = 2 (dy row y), r = 0 (dy row y + 1)
              const int slot = (py && ri) ? (jj ^ 1) : jj;
#pragma unroll
              for (int kc = 0; kc < KCN; ++kc) {
                u32x4 a = row[slot][kc];
#pragma unroll
                for (int si = 0; si <= px; ++si) {
                  const int sx = px ? (si ? 0 : 2) : 1;               // odd cols: s = 2 (dy col c), s = 0 (dy col c + 1: lane shift)
                  if (si > 0) {
#pragma unroll
                    for (int d = 0; d < 4; ++d) a[d] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)a[d], 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
                  }
                  const int tap = r * 3 + sx;
#pragma unroll
                  for (int nt = 0; nt < 2; ++nt) {
                    const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wl + nt * 32 * KP + tap * CIN + kc * 16);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, __builtin_bit_cast(bf16x8, a), acc[nt], 0, 0, 0);
                  }
                }
              }
            }
            if (cok) {
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                const int nt = c >> 1, hi = c & 1;
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                  float x = acc[nt][hi * 8 + e];
                  if constexpr (MASK && BITS) {
                    // channels cbase + 32 nt + 16 hi + 8 h .. + 7: byte 4 (2 hi + h) + cbase / 32 + nt of the pixel's 16
                    const int bidx = 4 * (2 * hi + h) + (cbase >> 5) + nt;
                    const uint32_t wsel = (bidx >> 2) == 0 ? mbits[t][0] : ((bidx >> 2) == 1 ? mbits[t][1] : ((bidx >> 2) == 2 ? mbits[t][2] : mbits[t][3]));
                    x = ((wsel >> ((bidx & 3) * 8 + e)) & 1u) ? x : 0.f;
                  } else if constexpr (MASK) x = (float)mv[t * 4 + c][e] > 0.f ? x : 0.f;
                  o[e] = (bf16)x;
                }
                c3_store16(Y + (int64_t)opix * p.ldc + nt * 32 + hi * 16 + h * 8, o);
              }
            }
          }
          load_row(y + 2, jj);                                        // dy row y is done: its slot takes row y + 2
        }
      }
    }
  }
}

template <int CIN, bool MASK, bool BITS = false>
int c3d2_launch(const GemmK& k, hipStream_t st) {
  constexpr int WAVES = 8, KP = 9 * CIN + 8;
  const size_t lds = (size_t)C3_NSL * KP * 2;
  auto fn = c3d2_kernel<CIN, MASK, WAVES, BITS>;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }      // (leave no stale error behind for the next launch check)
    attr = true;
  }
  const ConvGeom& g = k.cg;
  const int nsl = k.N / C3_NSL, Bn = k.M / (g.OH * g.OW);
  const int nstrip = (g.IW + 30) / 31;
  static const int env_rows = tune_env("GPV_C3D2_ROWS", 0);
  int blocks = 256 / nsl;
  int rows = env_rows > 0 ? env_rows : 2;
  if (env_rows <= 0) {
    const int64_t waves = (int64_t)blocks * WAVES;
    while (rows < 16 && (int64_t)Bn * nstrip * ((g.IH + rows + 1) / (rows + 2)) >= 2 * waves) rows += 2;
  }
  const int nseg = (g.IH + rows - 1) / rows;
  const int nitems = Bn * nstrip * nseg;
  while (blocks > 8 && (int64_t)(blocks - 8) * WAVES >= nitems) blocks -= 8;
  hipLaunchKernelGGL(fn, dim3(blocks, nsl), dim3(WAVES * 64), lds, st, k, rows, nstrip, nseg, nitems);
  GPV_CHECK_LAUNCH();
  return 0;
}

template <int CIN>
int c3r_mode(const GemmK& k, bool dgrad, hipStream_t st) {
  const bool m = k.mask != nullptr;
  if (dgrad) return m ? c3r_launch<CIN, true, true>(k, st) : c3r_launch<CIN, true, false>(k, st);
  return m ? c3r_launch<CIN, false, true>(k, st) : c3r_launch<CIN, false, false>(k, st);
}

inline bool al16c(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

}  // namespace

int g_c3s_mode = 1;          // 0 never, 1 heuristic, 2 wherever legal (tests)
long g_c3s_launches = 0;

// k: the GemmK gpv_conv2d prepared for the implicit-GEMM path (A = input pixels, B = [N][9][Cin] weights, cg = geometry).
// 0 = launched, -1 = not applicable, > 0 = hipError_t
int c3s_try_launch(const GemmK& k, int dtype_in, int dtype_out, hipStream_t st, bool dry) {
  static const int env = tune_env("GPV_C3S", -1);
  const int mode = env >= 0 ? env : g_c3s_mode;
  const ConvGeom& g = k.cg;
  if (mode == 0 || dtype_in != GPV_BF16 || dtype_out != GPV_BF16) return -1;
  if (g.KH != 3 || g.KW != 3 || g.PH != 1 || g.PW != 1 || g.SH != g.SW) return -1;
  if ((g.Cin != 64 && g.Cin != 128) || k.N % C3_NSL != 0 || k.N > 256 || k.K != 9 * g.Cin) return -1;
  if (g.Cs % 8 != 0 || k.ldc % 8 != 0 || (k.mask && k.ldm % 8 != 0) || k.ldb != k.K) return -1;
  if (k.res || k.rowscale || k.alpha != 1.0f || k.dthresh || k.accumulate || k.split_k > 1 || k.a_rowsum) return -1;
  if (k.act != GPV_ACT_NONE && k.act != GPV_ACT_RELU) return -1;
  if (!al16c(k.A) || !al16c(k.B) || !al16c(k.C) || (k.mask && !al16c(k.mask))) return -1;
  const bool d2 = g.SH == 2 && g.dgrad && g.Cin == 128 && g.OH == 2 * g.IH && g.OW == 2 * g.IW && k.act == GPV_ACT_NONE && !k.bias;
  if (g.SH != 1 && !d2) return -1;       // (the stride-2 forward stays on the tile kernels)
  // input extent addressed with 32-bit byte offsets (dgrad: the "input" is dy)
  const int64_t in_px = (int64_t)(k.M / (g.OH * g.OW)) * g.IH * g.IW;
  if (in_px * g.Cs * 2 >= (int64_t)C3_OOB) return -1;
  // a streaming regime needs rows: the layer1 / layer2 maps at training batch sizes -- and layer1's 64-channel map of ONE image (19200
  // pixels, forward): 12.4 us as a graph node against 15.2 for the 64 x 64 tile kernel (tools/bench_c3_bs1.py); layer2's 4800 pixels: 20.9 against 18.2
  if (mode == 1 && k.M < 65536 && !(g.Cin == 64 && !g.dgrad && k.M >= 16384)) return -1;
  if (k.out_bits || (k.mask_bits && !(d2 && k.N == 128 && (reinterpret_cast<uintptr_t>(k.mask_bits) & 15) == 0))) return -1;      // mask bits: the stride-2 backward-data over 128 channels only
  if (dry) return 0;
  int e;
  if (d2 && k.mask_bits) e = c3d2_launch<128, true, true>(k, st);
  else if (d2) e = k.mask ? c3d2_launch<128, true>(k, st) : c3d2_launch<128, false>(k, st);
  else e = g.Cin == 64 ? c3r_mode<64>(k, g.dgrad != 0, st) : c3r_mode<128>(k, g.dgrad != 0, st);
  if (e == 0) ++g_c3s_launches;
  return e;
}

}  // namespace gpvk
