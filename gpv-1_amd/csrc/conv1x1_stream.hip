// Streaming 1x1 convolution (gfx950): the K <= 512 pointwise convolutions of ResNet layer1 / layer2 over 10^5 .. 6 10^5 pixels,
// forward and backward-data (exp/gpv/models/backbone.py:93-95 -> torchvision Bottleneck conv1 / conv3 / downsample).
//
//   C[px, n] = epilogue( sum_k A[px, k] W[n, k] ),   bf16, K in {64, 128, 256, 512}, N in {64, 128, 256, 512}, N (K + 8) <= 75 K,
//   epilogue = + bias[n] -> + res[px, n] -> ReLU -> * (mask[px, n] > 0)
//
// These launches move 0.7 .. 1.2 KB per pixel for 2 K N flops: they are HBM-streaming, and the tile kernels (conv1x1_kernel,
// 64x64 tiles) reach 3.2-4.3 TB/s of the 5.3-6.2 TB/s an element-wise kernel gets on the same tensors, because a tile block has
// loads in flight for only a third of its life (load -> barrier -> MFMA -> barrier -> LDS round trip -> residual load -> store).
// Here nothing waits on a barrier after the prologue:
//   * the whole weight matrix sits in LDS (<= 139 KB), staged once per block;
//   * a WAVE owns 16 pixels at a time: its A fragments come straight from global (16 B per lane, next tile fetched a tile
//     ahead), the W fragments from LDS, the accumulators for up to 256 output channels stay in registers;
//   * the output channels are PERMUTED when W is staged so that the two MFMA tiles 2t, 2t+1 leave lane (pixel, g) with the 8
//     consecutive channels 32 t + 8 g .. + 7: residual, mask and output are then plain 16-byte accesses in the accumulator
//     layout -- no LDS round trip for the epilogue, four lanes cover a 64-byte run of a pixel's row;
//   * waves are independent, 8-16 per CU, each with its A prefetch and 8 residual loads in flight.
#include "gemm_common.h"

namespace gpvk {
namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma16s(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// LDS row L of the staged weight matrix holds output channel chan(L): pass h = L / NH (NH channels per register pass), MFMA
// tile j = (L % NH) / 16, row r = L % 16 -> channel h NH + 32 (j / 2) + 8 (r / 4) + 4 (j & 1) + (r % 4)
template <int NH>
__device__ __forceinline__ int c1s_chan(int L) {
  const int h = L / NH, w = L - h * NH, j = w >> 4, r = w & 15;
  return h * NH + (j >> 1) * 32 + (r >> 2) * 8 + (j & 1) * 4 + (r & 3);
}

// LIN: the kernel as a plain linear layer's GEMM (gpv_gemm, K = 256 -> 2048 outputs: the DETR feed-forward 256 -> 2048 and its
// backward-data product with the ReLU mask): alpha on the accumulator and the GEMM kernels' dropout
// epilogue (same element index -> same keep pattern as gemm.hip / gemm_glds.hip / gemm_pipe.hip); the convolution instances are unchanged
// amdgpu_waves_per_eu(2, 2) on the instances whose weights leave room for ONE 8-wave block per CU (> 72 KB of LDS): without the hint
// hipcc schedules for three or four waves per SIMD, keeps one fragment register and serialises ds_read -> s_waitcnt lgkmcnt(0) -> MFMA
// (97 of the 128 MFMAs of the 256 x 256 instance waited for a read issued right before them).
// BITS (round 6): ReLU masks as one bit per element (include/gpv_hip.h gpv_conv_args.y_mask_bits / relu_mask_bits).  The byte order inside a
// pixel's row of bits follows THIS kernel's accumulator layout, so that neither side needs a cross-lane operation: channel c of a pixel is
// bit (c & 7) of byte  32 (c / 256) + 8 ((c % 32) / 8) + (c % 256) / 32  -- lane (pixel, g) of a 256-channel pass holds the 8 channels
// 32 t + 8 g .. + 7 of its eight tile pairs t: their eight bytes are CONSECUTIVE (one 8-byte store / load per lane and pass; the K = 512
// instances' 128-channel passes read four of them).  !MASK: the launch also writes (output > 0) of its ReLU outputs; MASK: the mask
// operand is p.mask_bits, 8 (4) bytes per lane and pass instead of 128 (64).  (First build: channel-linear words, two cross-row shuffles
// and a 4-byte store per tile pair in the producer -- the forward launches lost 13 / 6 us of the 25 / 11 the backward-data ones gained.)
template <int K, int NH, bool RES, bool MASK, bool NT, bool LIN = false, int NP = 1, bool BITS = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu((NP == 1 && NH * K * 2 > 72 * 1024) ? 2 : 1, (NP == 1 && NH * K * 2 > 72 * 1024) ? 2 : 8)))
void c1s_kernel(GemmK p, int ncols) {
  // ncols: output channels per block row (gridDim.y slices of a wide layer: layer3's 1024 channels as four 256-channel
  // problems that share A -- the weights of one slice fit the LDS, A is small next to the output)
  constexpr int KC = K / 32, NTL = NH / 16, NG = NH / 32, SL = K / 8, ROWB = K * 2;      // LDS row = K bf16, no padding: XOR swizzle (below)
  constexpr bool BIG = NP == 1 && NH * K * 2 > 72 * 1024;                                 // one block per CU (the two-pass K = 128 instance measured 3 % slower this way)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16* Wl = reinterpret_cast<bf16*>(smem_raw);
  float* bias_l = reinterpret_cast<float*>(Wl + (size_t)ncols * K);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, pl = lane & 15;
  const int cbase = (int)blockIdx.y * ncols;
  const bf16* A = reinterpret_cast<const bf16*>(p.A);
  const bf16* Wg = reinterpret_cast<const bf16*>(p.B) + (int64_t)cbase * p.ldb;
  bf16* C = reinterpret_cast<bf16*>(p.C) + cbase;
  const bf16* R = reinterpret_cast<const bf16*>(p.res) + cbase;
  const bf16* Mk = reinterpret_cast<const bf16*>(p.mask) + cbase;
  const float* bias_g = p.bias ? p.bias + cbase : nullptr;
  const int nfull = p.N;
  const int bpp = nfull >> 3;                              // BITS: mask bytes per pixel
  if constexpr (LIN) { if (p.dthresh) p.seed = eff_seed(p.seed, p.seed_dev); }
  p.N = ncols;
  const int ntile = (p.M + 15) >> 4;
  const int nw = (int)gridDim.x * 8;
  int tile = (int)blockIdx.x * 8 + wave;

  // this wave's first A fragments are requested before the weights are staged
  bf16x8 an[KC];
  // stride 2 (the downsample projections, forward): output pixel (b, oh, ow) reads input pixel (b, 2 oh, 2 ow)
  const int ow_ = p.cg.OW, ohw = p.cg.OH * p.cg.OW, ihw = p.cg.IH * p.cg.IW, iw_ = p.cg.IW, sxy = p.cg.SH;
  // (rows beyond the map -- the last tile's tail, the prefetch past a wave's last tile -- re-read the LAST row: no predicate per load;
  //  with `if (ok)` around them hipcc branched around every pair of loads and waited for the pair before the next one)
  auto fetch = [&](int t) {
    const int px = min(t * 16 + pl, p.M - 1);
    int64_t ipx = px;
    if (sxy == 2) {
      const int b = px / ohw, r = px - b * ohw, oh = r / ow_, ow = r - oh * ow_;
      ipx = (int64_t)b * ihw + (int64_t)(2 * oh) * iw_ + 2 * ow;
    }
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) an[kc] = *reinterpret_cast<const bf16x8*>(A + ipx * p.lda + kc * 32 + g * 8);
  };
  fetch(tile);
  // The weights go L2 -> LDS by DMA (buffer_load ... lds: no staging registers, a wave's whole share in flight at once -- through
  // registers the 128 KB of a 256 x 256 slice were dependent round trips of eight loads per thread: 16 K cycles before round 5's
  // unconditional loads, ~ 8 K after, 4 K by DMA, timed with s_memtime in linear_ln.hip).  The DMA writes lane-linear 1 KB images
  // (instruction i = LDS rows i R .. i R + R - 1, R = 512 / K; lane = (row, 16-byte slot)), so the bank-conflict-free layout is made on
  // the SOURCE side: slot s of LDS row L receives chunk s ^ swz(L) of channel chan(L), swz(L) = L & 15 (K >= 128: a row covers all 64
  // banks at least once) | (L >> 1) & 7 (K = 64: two rows per 256 bytes), and the fragment read of chunk c of row L looks at slot
  // c ^ swz(L) -- checked against the ds_read_b128 lane groups of MI355X_MICROARCH.md ({0-3, 12-15, 20-27}, ...: g pairs (0,1), (2,3)).
  {
    typedef __attribute__((address_space(3))) void lds_void_t;
    constexpr int OOB = 0x7ffffff0, R = 512 / K, SPR = K / 8;
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(Wg), (short)0, OOB, 0x00020000);
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int ninst = p.N * K / 512;                         // 1 KB each; a multiple of 8 (N, K multiples of 64)
    for (int i = wv; i < ninst; i += 8) {
      const int L = i * R + (R > 1 ? lane / SPR : 0), sl = lane % SPR;
      const int sw = K >= 128 ? (L & 15) : ((L >> 1) & 7);
      const int voff = (c1s_chan<NH>(L) * (int)p.ldb + ((sl ^ sw) * 8)) * 2;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_void_t*)(smem_raw + i * 1024), 16, voff, 0, 0, 0);
    }
  }
  for (int c = tid; c < p.N; c += 512) bias_l[c] = bias_g ? bias_g[c] : 0.f;
  // this wave's DMA pieces have landed; the barrier orders everyone's for the fragment reads.  The BUILTIN, not inline asm: the waitcnt
  // pass must know that nothing is pending at the tile loop's header, or it merges the prologue's unknown state into it and drains the
  // queue at the top of every tile (conv1x1_dual.hip)
  __builtin_amdgcn_s_waitcnt(0x0F70);
  asm volatile("" ::: "memory");
  __syncthreads();

  // LIN dropout (round 5): the keep words of common.h's drop_pair_bits, bit for bit, at a tenth of the instructions.  The epilogue below
  // WAS the launch: 42 VALU instructions per output element (two quarter-rate 32-bit multiplies and 64-bit index arithmetic per PAIR,
  // shift / and / compare / select / multiply per element) against 2 MFMAs per 64 elements -- 13 K cycles of VALU per 16-row tile
  // beside 2 K of MFMA (ISA count; the 256 -> 2048 feed-forward GEMM ran at 29 - 37 us for 4 us of matrix work).  A lane's pairs in a
  // tile are pair0 + const: the first multiply is ONE per tile (the rest are literal adds), the second (high words) is constant while the
  // low word does not wrap (checked per tile, wave-uniform; the general path stays for the wrap), and the keep test is applied to the
  // PACKED bf16 pairs: flip the sign bits of the two 16-bit fields, saturating packed subtract of the threshold, arithmetic shift ->
  // 0xffff per dropped half, and-not (attention.hip's attn_drop_bits).
  const uint32_t t16 = LIN ? (p.dthresh >> 16) : 0u;
  const uint32_t ts2 = ((t16 - 32768u) & 0xffffu) * 0x10001u;
  for (; tile < ntile; tile += nw) {
    bf16x8 af[KC];
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) af[kc] = an[kc];
    fetch(tile + nw);
    const int px = tile * 16 + pl;
    const int pxc = min(px, p.M - 1);
    uint32_t hbase = 0u;
    bool hfast = false;
    if constexpr (LIN) {
      if (p.dthresh) {
        const uint64_t pr = ((uint64_t)pxc * (uint64_t)nfull + (uint64_t)(cbase + g * 8)) >> 1;       // pair index of this lane's first group (rows beyond M: the last row's, see the stores)
        hbase = (uint32_t)pr * 0x9E3779B9u + (uint32_t)p.seed + ((uint32_t)(pr >> 32) ^ (uint32_t)(p.seed >> 32)) * 0x85EBCA6Bu;
        hfast = !__any((uint32_t)pr > 0xffffffffu - (uint32_t)(p.N / 2 + 8));
      }
    }
    // Every LOAD of the tile loop is unconditional (rows beyond the map read the last row) and the pass loop is unrolled (NP = channels
    // per block / NH, 1 | 2).  With `if (pok)` around the loads and a run-time pass loop hipcc could not count what is in flight and
    // drained the queue -- `s_waitcnt vmcnt(0)` -- between the prefetch and the first MFMA of every tile (seen in the ISA).
#pragma unroll
    for (int h = 0; h < NP; ++h) {
      // epilogue operands of this pass: requested before the MFMAs, consumed after them
      bf16x8 rv[RES ? NG : 1], mv[MASK ? NG : 1];
      if constexpr (RES) {
#pragma unroll
        for (int t = 0; t < NG; ++t) {
          const bf16* q = R + (int64_t)pxc * p.ldr + h * NH + t * 32 + g * 8;
          rv[t] = NT ? __builtin_bit_cast(bf16x8, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(q)))
                     : *reinterpret_cast<const bf16x8*>(q);
        }
      }
      // BITS: this pass covers channels c0p .. c0p + NH - 1 of the pixel: byte 32 (c0p / 256) + 8 g + (c0p % 256) / 32 onwards, NG bytes
      const int c0p = cbase + h * NH;
      const int64_t boff = (int64_t)pxc * bpp + (c0p >> 8) * 32 + g * 8 + ((c0p & 255) >> 5);
      uint32_t mw[2] = {0u, 0u};
      if constexpr (MASK && BITS) {
        const unsigned char* mq = reinterpret_cast<const unsigned char*>(p.mask_bits) + boff;
        if constexpr (NG == 8) {
          const uint2 w2 = *reinterpret_cast<const uint2*>(mq);
          mw[0] = w2.x; mw[1] = w2.y;
        } else {
          static_assert(NG == 4 || !(MASK && BITS), "mask bits: 256- or 128-channel passes");
          mw[0] = *reinterpret_cast<const uint32_t*>(mq);
        }
      } else if constexpr (MASK) {
#pragma unroll
        for (int t = 0; t < NG; ++t) {
          mv[t] = *reinterpret_cast<const bf16x8*>(Mk + (int64_t)pxc * p.ldm + h * NH + t * 32 + g * 8);
        }
      }
      f32x4 acc[NTL];
#pragma unroll
      for (int j = 0; j < NTL; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      int woff = (h * NH + pl) * ROWB;               // bytes
      asm volatile("" : "+v"(woff));                 // opaque per tile: with the pass loop unrolled the fragment reads are loop-invariant and hipcc hoists all 128 of them out of the tile loop (into scratch)
      const unsigned char* wrow = smem_raw + woff;
      const int tx = (g ^ (K >= 128 ? pl : (pl >> 1))) * 16;       // chunk 4 kc + g of row L sits in slot (4 kc + g) ^ swz(L) = 4 kc ^ (g ^ swz)
      if constexpr (BIG) {
        // one block per CU = two waves per SIMD: the fragments are read a group of eight column tiles (8 MFMAs) ahead of their use,
        // pinned with sched_barrier -- left alone hipcc keeps one fragment register and waits for every read right before its MFMA
        constexpr int JG = NTL >= 8 ? (((RES && MASK) || LIN) ? 4 : 8) : NTL, NGJ = NTL / JG, NGRP = KC * NGJ;   // (residual + mask operands in flight: 256 registers are reached with groups of eight)
        bf16x8 wf[2][JG];
        auto ldw = [&](int grp, int b) {
          const int kc = grp / NGJ, jh = grp - kc * NGJ;
          const unsigned char* wk = wrow + ((kc * 64) ^ tx);
#pragma unroll
          for (int jj = 0; jj < JG; ++jj) wf[b][jj] = *reinterpret_cast<const bf16x8*>(wk + (jh * JG + jj) * 16 * ROWB);
        };
        ldw(0, 0);
#pragma unroll
        for (int grp = 0; grp < NGRP; ++grp) {
          const int kc = grp / NGJ, jh = grp - kc * NGJ;
          if (grp + 1 < NGRP) ldw(grp + 1, (grp + 1) & 1);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int jj = 0; jj < JG; ++jj) acc[jh * JG + jj] = mfma16s(wf[grp & 1][jj], af[kc], acc[jh * JG + jj]);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
          const unsigned char* wk = wrow + ((kc * 64) ^ tx);
#pragma unroll
          for (int j = 0; j < NTL; ++j) {
            const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wk + j * 16 * ROWB);
            acc[j] = mfma16s(wf, af[kc], acc[j]);
          }
        }
      }
      uint32_t bits_acc[2] = {0u, 0u};
#pragma unroll
      for (int t = 0; t < NG; ++t) {
        const int c0 = h * NH + t * 32 + g * 8;
        const float4 b0 = *reinterpret_cast<const float4*>(bias_l + c0), b1 = *reinterpret_cast<const float4*>(bias_l + c0 + 4);
        if constexpr (LIN) {
#pragma unroll
          for (int i = 0; i < 4; ++i) { acc[2 * t][i] *= p.alpha; acc[2 * t + 1][i] *= p.alpha; }
        }
        float v[8] = {acc[2 * t][0] + b0.x, acc[2 * t][1] + b0.y, acc[2 * t][2] + b0.z, acc[2 * t][3] + b0.w,
                      acc[2 * t + 1][0] + b1.x, acc[2 * t + 1][1] + b1.y, acc[2 * t + 1][2] + b1.z, acc[2 * t + 1][3] + b1.w};
        bf16x8 o;
        bool masked = false;
        if constexpr (LIN) {
          if (p.dthresh) {
            masked = true;
            uint32_t dropw[4];                      // 0xffff in every dropped half of pair q (elements c0 + 2 q, c0 + 2 q + 1)
            if (hfast) {
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                uint32_t x = hbase + (uint32_t)(h * (NH / 2) + t * 16 + q) * 0x9E3779B9u;
                x ^= x >> 16; x = __umul24(x, 0x85EBCBu);
                x ^= x >> 13; x = __umul24(x, 0xC2B2AFu);
                x ^= x >> 16;
                typedef short s16x2 __attribute__((ext_vector_type(2)));
                const s16x2 d = __builtin_elementwise_sub_sat(__builtin_bit_cast(s16x2, x ^ 0x80008000u), __builtin_bit_cast(s16x2, ts2));
                dropw[q] = __builtin_bit_cast(uint32_t, (s16x2)(d >> (s16x2){15, 15}));
              }
            } else {
              const uint32_t keep8 = drop_mask<8>(p.seed, (uint64_t)pxc * (uint64_t)nfull + (uint64_t)(cbase + c0), p.dthresh);
#pragma unroll
              for (int q = 0; q < 4; ++q)
                dropw[q] = (((keep8 >> (2 * q)) & 1u) ? 0u : 0xffffu) | (((keep8 >> (2 * q + 1)) & 1u) ? 0u : 0xffff0000u);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              float x = v[e];
              if constexpr (RES) x += (float)rv[t][e];
              if (p.act == GPV_ACT_RELU) x = fmaxf(x, 0.f);
              x *= p.dscale;
              if constexpr (MASK) x = (float)mv[t][e] > 0.f ? x : 0.f;
              o[e] = (bf16)x;
            }
            u32x4 ow = __builtin_bit_cast(u32x4, o);
#pragma unroll
            for (int q = 0; q < 4; ++q) ow[q] &= ~dropw[q];
            o = __builtin_bit_cast(bf16x8, ow);
          }
        }
        if (!masked) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float x = v[e];
            if constexpr (RES) x += (float)rv[t][e];
            if (p.act == GPV_ACT_RELU) x = fmaxf(x, 0.f);
            if constexpr (MASK && BITS) x = ((mw[t >> 2] >> ((t & 3) * 8 + e)) & 1u) ? x : 0.f;
            else if constexpr (MASK) x = (float)mv[t][e] > 0.f ? x : 0.f;
            o[e] = (bf16)x;
          }
        }
        if constexpr (BITS && !MASK) {
          // (output > 0) of the STORED bf16 values -- as 16-bit integers: > 0 (a stored -0 or negative is not; what a later
          // (float)mask > 0.f test sees) -- eight of them into one byte in 13 instructions: packed min(., 1) / max(., 0) leave 0 | 1 per
          // half, a byte permute gathers four halves' low bytes, a 4 x 8-bit dot product with (1, 2, 4, 8) makes the nibble
          // (inline asm: written with __builtin_elementwise_min / max on short2 values hipcc 7.2 derived all four words' flags from
          //  the FIRST word -- v_cmp_lt_i16 on elements 0 and 1 only, seen in the ISA -- and every byte came out as 0x00 / 55 / aa / ff)
          const u32x4 ow = __builtin_bit_cast(u32x4, o);
          const uint32_t one2 = 0x00010001u, zero2 = 0u;
          uint32_t m[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            uint32_t tq;
            asm("v_pk_min_i16 %0, %1, %2" : "=v"(tq) : "v"(ow[q]), "v"(one2));
            asm("v_pk_max_i16 %0, %1, %2" : "=v"(m[q]) : "v"(tq), "v"(zero2));
          }
          const uint32_t b03 = __builtin_amdgcn_perm(m[1], m[0], 0x06040200u);      // bytes (m0.b0, m0.b2, m1.b0, m1.b2) = elements 0..3
          const uint32_t b47 = __builtin_amdgcn_perm(m[3], m[2], 0x06040200u);
          const uint32_t byte = __builtin_amdgcn_udot4(b03, 0x08040201u, 0u, false) | (__builtin_amdgcn_udot4(b47, 0x08040201u, 0u, false) << 4);
          bits_acc[t >> 2] |= byte << ((t & 3) * 8);
        }
        // No branch and no store hipcc can see in the tile loop (conv1x1_dual.hip): rows beyond M were LOADED from row M - 1 (A,
        // residual, mask, dropout index), so they hold row M - 1's results and store them there again; the store is inline asm because
        // a visible one makes the waitcnt pass treat the vm counter as out of order and drain it -- the previous tile's stores included
        // -- before the first MFMA of every tile.  s_nop 1: the two wait states a 16-byte store's data registers need before a VALU
        // write (the hazard recognizer cannot see into the asm).  (A buffer store with a dropped out-of-range offset: 25 - 35 % slower.)
        {
          bf16* q = C + (int64_t)pxc * p.ldc + c0;
          const u32x4 ov = __builtin_bit_cast(u32x4, o);
          if constexpr (NT) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" :: "v"(q), "v"(ov) : "memory");
          else asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(q), "v"(ov) : "memory");
        }
      }
      if constexpr (BITS && !MASK) {
        // the pass's eight mask bytes of this lane: one 8-byte store, invisible to the waitcnt pass like the data stores (rows beyond M
        // hold row M - 1's values and store them there again)
        static_assert(NG == 8 || !(BITS && !MASK), "mask bits are written by 256-channel passes");
        unsigned char* bq = reinterpret_cast<unsigned char*>(p.out_bits) + boff;
        const uint2 bv = make_uint2(bits_acc[0], bits_acc[1]);
        asm volatile("global_store_dwordx2 %0, %1, off\n\ts_nop 0" :: "v"(bq), "v"(bv) : "memory");
      }
    }
  }
}

// output channels per block: the whole layer when its weights fit the LDS (K <= 128: up to 512 channels), else slices on gridDim.y
inline int c1s_cols(int K, int N) {
  const int cap = K <= 128 ? 512 : (K == 256 ? 256 : 128);
  return N < cap ? N : cap;
}

template <int K, int NH, bool RES, bool MASK, bool NT, bool LIN = false, int NP = 1, bool BITS = false>
int c1s_launch_np(const GemmK& k, hipStream_t st) {
  const int ncols = c1s_cols(k.K, k.N), nsl = k.N / ncols;
  const size_t lds = (size_t)ncols * K * 2 + (size_t)ncols * sizeof(float);
  auto fn = c1s_kernel<K, NH, RES, MASK, NT, LIN, NP, BITS>;
  static size_t attr = 0;
  if (lds > 64 * 1024 && lds > attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }      // (leave no stale error behind for the next launch check)
    attr = lds;
  }
  // persistent waves: two blocks of 8 waves per CU when the weights leave room for them, one otherwise
  const int ntile = (k.M + 15) / 16;
  static const int bpc = tune_env("GPV_C1S_BLOCKS", 0);
  int blocks = bpc > 0 ? bpc : (lds <= 72 * 1024 ? 512 : 256);
  blocks = (blocks + nsl - 1) / nsl;
  if (blocks * 8 > ntile) blocks = (ntile + 7) / 8;
  hipLaunchKernelGGL(fn, dim3(blocks, nsl), dim3(512), lds, st, k, ncols);
  GPV_CHECK_LAUNCH();
  return 0;
}

template <int K, int NH, bool RES, bool MASK, bool NT, bool LIN = false, bool BITS = false>
int c1s_launch(const GemmK& k, hipStream_t st) {
  const int np = c1s_cols(k.K, k.N) / NH;
  if (np == 1) return c1s_launch_np<K, NH, RES, MASK, NT, LIN, 1, BITS>(k, st);
  if constexpr (K <= 128 && NH == 256 && !LIN) { if (np == 2) return c1s_launch_np<K, NH, RES, MASK, NT, LIN, 2, BITS>(k, st); }
  return -1;
}

template <int K, int NH>
int c1s_flags(const GemmK& k, hipStream_t st, bool linear) {
  const bool r = k.res != nullptr, m = k.mask != nullptr, nt = k.nt_io != 0;
  // gpv_gemm's calls always take the LIN instances (alpha = 1 and no dropout are exact no-ops there): a linear layer's launch is then
  // told from a convolution's by its NAME -- profiles and bench.py's live HBM-traffic passes (tools/pmc_traffic.py) classify by it
  if (linear || k.alpha != 1.0f || k.dthresh) {     // linear-layer extras: instantiated for the 256 -> n x 256 shapes only
    if constexpr (K == 256 && NH == 256) {
      if (nt) return -1;
      if (r && m) return c1s_launch<K, NH, true, true, false, true>(k, st);
      if (r) return c1s_launch<K, NH, true, false, false, true>(k, st);
      if (m) return c1s_launch<K, NH, false, true, false, true>(k, st);
      return c1s_launch<K, NH, false, false, false, true>(k, st);
    } else {
      return -1;
    }
  }
  if (k.mask_bits || k.out_bits) {
    // one-bit ReLU masks: the instances the ResNet body uses -- conv3 + residual + ReLU of layer2 / layer3 writes them (K = 128 | 256,
    // 256 channels per pass), conv1's backward-data (residual + mask) of layer2 / layer3 / layer4.0 reads them (+ K = 512 -> 128-wide passes)
    if (nt || !r || (k.mask_bits && k.out_bits)) return -1;
    if constexpr ((K == 128 || K == 256) && NH == 256) {
      if (k.out_bits) return (m || k.act != GPV_ACT_RELU) ? -1 : c1s_launch<K, NH, true, false, false, false, true>(k, st);
      return c1s_launch<K, NH, true, true, false, false, true>(k, st);
    } else if constexpr (K == 512 && NH == 128) {
      if (k.out_bits) return -1;
      return c1s_launch<K, NH, true, true, false, false, true>(k, st);
    } else {
      return -1;
    }
  }
  if (nt) {      // (non-temporal: the layer1 forwards -- never with a mask)
    if (m) return -1;
    return r ? c1s_launch<K, NH, true, false, true>(k, st) : c1s_launch<K, NH, false, false, true>(k, st);
  }
  if (r && m) return c1s_launch<K, NH, true, true, false>(k, st);
  if (r) return c1s_launch<K, NH, true, false, false>(k, st);
  if (m) return c1s_launch<K, NH, false, true, false>(k, st);
  return c1s_launch<K, NH, false, false, false>(k, st);
}

template <int K>
int c1s_n(const GemmK& k, hipStream_t st, bool linear) {
  switch (c1s_cols(K, k.N)) {          // channels per block -> channels per register pass
    case 64: return c1s_flags<K, 64>(k, st, linear);
    case 128: return c1s_flags<K, 128>(k, st, linear);
    case 256: case 512: if constexpr (K <= 256) return c1s_flags<K, 256>(k, st, linear); else return -1;
  }
  return -1;
}

inline bool al16s(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

}  // namespace

int g_c1s_mode = 1;          // 0 never, 1 heuristic, 2 wherever legal (tests)
long g_c1s_launches = 0;     // gpv_set_option(GPV_OPT_C1S_LAUNCHES, .)

// 0 = launched, -1 = not applicable, > 0 = hipError_t
int c1s_try_launch(const GemmK& k, int dtype_in, int dtype_out, hipStream_t st, bool linear, bool dry) {
  static const int env = tune_env("GPV_C1S", -1);
  const int mode = env >= 0 ? env : g_c1s_mode;
  if (mode == 0 || dtype_in != GPV_BF16 || dtype_out != GPV_BF16) return -1;
  if (k.K != 64 && k.K != 128 && k.K != 256 && k.K != 512) return -1;
  if (k.N != 64 && k.N != 128 && k.N != 256 && k.N != 512 && k.N != 1024 && k.N != 2048 && !(linear && k.N % 256 == 0 && k.N <= 2048)) return -1;
  const int ncols = c1s_cols(k.K, k.N), nsl = k.N / ncols;
  if ((size_t)ncols * (k.K + 8) * 2 + (size_t)ncols * 4 > 150 * 1024 || nsl > 8) return -1;
  if (k.rowscale || k.accumulate || k.split_k > 1 || k.a_rowsum) return -1;
  if ((k.alpha != 1.0f || k.dthresh) && !(linear && k.K == 256 && k.N >= 256 && k.N % 256 == 0)) return -1;
  if (linear && (k.cg.SH == 2 || k.cg.cm || k.nt_io)) return -1;
  if (k.act != GPV_ACT_NONE && k.act != GPV_ACT_RELU) return -1;
  if (k.lda % 8 || k.ldb != k.K || k.ldc % 8 || (k.res && k.ldr % 8) || (k.mask && k.ldm % 8)) return -1;
  if (!al16s(k.A) || !al16s(k.B) || !al16s(k.C) || (k.res && !al16s(k.res)) || (k.mask && !al16s(k.mask))) return -1;
  if (linear) {      // gpv_gemm: 256 -> 2048 features over >= 2048 rows (the DETR feed-forward; tools/bench_c1s_linear.py: 1024 / 1536 outputs are faster on the tile kernels)
    if (mode == 1 && (k.K != 256 || k.N < 1536 || k.M < 2048)) return -1;      // (1536: as graph nodes 20.8 against 26.8 us at 9600 rows, tools/tune_gemms.py --chain; 1024 stays on the tile kernels)
  } else if (mode == 1 && ((int64_t)k.M * nsl < 65536 || k.M < 32768)) return -1;   // a streaming regime needs rows (x slices): the layer1-3 maps at training batch sizes
  if (k.mask_bits || k.out_bits) {
    // instances that exist with mask bits (c1s_flags) -- decided here so that a dry run (gpv_conv2d_mask_bits_ok) answers what a launch would do
    const int nh = ncols >= 256 ? 256 : ncols;
    const bool inst = ((k.K == 128 || k.K == 256) && nh == 256) || (k.K == 512 && nh == 128 && !k.out_bits);
    if (linear || !inst || !k.res || k.nt_io || (k.mask_bits && k.out_bits) || k.N % 256 != 0 || (k.out_bits && (k.mask && !k.mask_bits))) return -1;
    if (k.out_bits && k.act != GPV_ACT_RELU) return -1;
    if (k.K <= 128 && ncols / nh > 2) return -1;
    if (k.K > 128 && ncols / nh != 1) return -1;
  }
  if (dry) return 0;
  int e = -1;
  switch (k.K) {
    case 64: e = c1s_n<64>(k, st, linear); break;
    case 128: e = c1s_n<128>(k, st, linear); break;
    case 256: e = c1s_n<256>(k, st, linear); break;
    case 512: e = c1s_n<512>(k, st, linear); break;       // (N <= 128: the weights must fit the LDS)
  }
  if (e == 0) ++g_c1s_launches;
  return e;
}

}  // namespace gpvk
