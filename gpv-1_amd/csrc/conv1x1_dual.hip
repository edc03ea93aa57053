// The tail of a ResNet stage's FIRST bottleneck in one launch (gfx950):
//
//     y = ReLU( conv3(a2) * bn3  +  downsample(x) * bn_d )          (torchvision Bottleneck.forward with a downsample branch;
//                                                                    exp/gpv/models/backbone.py:93-95 builds resnet50 from it)
//       = ReLU( [a2 | x_s] . [W3 | Wd]^T + (shift3 + shift_d) ),     x_s = x sampled at the block's stride
//
// Two pointwise convolutions that write / re-read the same [pixels, 4 planes] map: as separate launches the downsample branch
// writes it (314 MB in layer1 at B = 32), conv3 reads it back as its residual and writes it again.  Both are reductions over the
// channels of ONE pixel, so they are one GEMM over the concatenated channels K1 + K2 with two row pointers.  The kernel is
// conv1x1_stream.hip's design: the concatenated weights of a block of output channels resident in LDS (layer1: 256 x (64 + 64),
// 70 KB; layer2: 128-channel slices of 512 x (128 + 256), 100 KB each), a wave per 16 pixels, A fragments straight from global
// memory a tile ahead, output channels permuted at staging so that the epilogue is plain 16-byte stores, no barrier after the
// prologue.  layer3 / layer4 (K1 + K2 = 768 / 1536 into 1024 / 2048 channels) do not fit and stay two launches.
#include "gemm_common.h"

namespace gpvk {
namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct DualK {
  const void* a1; const void* a2; const void* w1; const void* w2; const float* bias; void* y;
  int M, N;                       // output pixels, output channels
  int OH, OW, IH2, IW2, S2;       // a2's spatial extent and stride (a1 has the output's)
  int ld1, ld2;                   // channel strides of a1 / a2 pixels
  int relu, nt;
  void* bits;                     // round 6: (y > 0) as one bit per element in conv1x1_stream.hip's byte order (gpv_conv_args.y_mask_bits), or NULL
};

template <int NH>
__device__ __forceinline__ int c1d_chan(int L) {      // (conv1x1_stream.hip c1s_chan)
  const int hh = L / NH, w = L - hh * NH, j = w >> 4, r = w & 15;
  return hh * NH + (j >> 1) * 32 + (r >> 2) * 8 + (j & 1) * 4 + (r & 3);
}

template <int K1, int K2, int NH, bool NT, bool BITS = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void c1d_kernel(DualK p, int ncols) {
  // A wave owns 32 pixels (two MFMA row tiles): every weight fragment read from LDS feeds TWO MFMAs -- with 16 pixels per wave the
  // launch was bound by the LDS read pipe (96 ds_read_b128 per 96 MFMAs and wave: 109 us for layer2's 60 GFLOP).  The A fragments of
  // the NEXT tile are loaded straight into the registers of the k-chunk that has just been consumed (rolling prefetch, no second
  // register image); ncols == NH (one pass per tile, checked by the launch).  The weight fragments are read one group of four
  // column tiles (8 MFMAs) ahead, pinned with sched_barrier: left alone hipcc reads two fragments and waits for them.  No branch in
  // the tile loop (rows beyond M are clamped to row M - 1 on load, so they hold row M - 1's results and store them again): with
  // conditional stores in the loop the waitcnt pass gives up counting and waits vmcnt(0) for the rolling loads.  amdgpu_waves_per_eu(2, 2): one 8-wave block per CU is
  // all the LDS image allows, and without the hint hipcc schedules for three waves per SIMD and serialises read -> wait -> MFMA.
  constexpr int KT = K1 + K2, KP = KT + 8, KC1 = K1 / 32, KC = KT / 32, NTL = NH / 16, NG = NH / 32, SL1 = K1 / 8, SL = KT / 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16* Wl = reinterpret_cast<bf16*>(smem_raw);
  float* bias_l = reinterpret_cast<float*>(Wl + (size_t)ncols * KP);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, pl = lane & 15;
  const int cbase = (int)blockIdx.y * ncols;
  const bf16* A1 = reinterpret_cast<const bf16*>(p.a1);
  const bf16* A2 = reinterpret_cast<const bf16*>(p.a2);
  bf16* C = reinterpret_cast<bf16*>(p.y) + cbase;
  const int ntile = (p.M + 31) >> 5;
  const int nw = (int)gridDim.x * 8;
  int tile = (int)blockIdx.x * 8 + wave;
  const int ohw = p.OH * p.OW, ihw = p.IH2 * p.IW2;
  // rows of tile t as this lane reads them: unconditional loads from a clamped pixel (a predicate around a load makes hipcc branch
  // and wait vmcnt(0) at the join -- DESIGN section 0, finding 1); rows beyond M are computed and not stored
  const bf16* r1[2];
  const bf16* r2[2];
  auto rows = [&](int t) {
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int px = min(t * 32 + m * 16 + pl, p.M - 1);
      int64_t ipx = px;
      if (p.S2 == 2) {
        const int b = px / ohw, r = px - b * ohw, oh = r / p.OW, ow = r - oh * p.OW;
        ipx = (int64_t)b * ihw + (int64_t)(2 * oh) * p.IW2 + 2 * ow;
      }
      r1[m] = A1 + (int64_t)px * p.ld1 + g * 8;
      r2[m] = A2 + ipx * p.ld2 + g * 8;
    }
  };
  bf16x8 af[2][KC];
  auto fetch = [&](int m, int kc) {
    af[m][kc] = *reinterpret_cast<const bf16x8*>(kc < KC1 ? r1[m] + kc * 32 : r2[m] + (kc - KC1) * 32);
  };
  rows(tile);
#pragma unroll
  for (int kc = 0; kc < KC; ++kc) { fetch(0, kc); fetch(1, kc); }
  {
    const bf16* W1 = reinterpret_cast<const bf16*>(p.w1);
    const bf16* W2 = reinterpret_cast<const bf16*>(p.w2);
    stage_chunks16<512, 8>(ncols * SL, tid,
        [&](int idx) { const int L = idx / SL, sl = idx - L * SL; const int c = cbase + c1d_chan<NH>(L);
                       return sl < SL1 ? W1 + (int64_t)c * K1 + sl * 8 : W2 + (int64_t)c * K2 + (sl - SL1) * 8; },
        [&](int idx) { const int L = idx / SL, sl = idx - L * SL; return Wl + L * KP + sl * 8; });
    for (int c = tid; c < ncols; c += 512) bias_l[c] = p.bias ? p.bias[cbase + c] : 0.f;
  }
  __syncthreads();
  __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0): nothing pending at the loop header, or the waitcnt pass -- which merges the
                                                       // staging loop's unknown state into it -- waits vmcnt(0) at the top of EVERY tile
  const bf16* wrow = Wl + pl * KP + g * 8;
  for (; tile < ntile; tile += nw) {
    const int px0 = tile * 32 + pl;
    rows(min(tile + nw, ntile - 1));                   // (the last tile of a wave re-reads a valid tile; nobody uses it)
    int woff = 0;
    asm volatile("" : "+v"(woff));                     // the fragment reads are loop-invariant: opaque, or hipcc hoists all of them into scratch
    const bf16* wr = wrow + woff;
    f32x4 acc[2][NTL];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int j = 0; j < NTL; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int JG = 4, NGJ = NTL / JG, NGRP = KC * NGJ;
    bf16x8 wf[2][JG];
    auto ldw = [&](int grp, int b) {
      const int kc = grp / NGJ, jh = grp - kc * NGJ;
#pragma unroll
      for (int jj = 0; jj < JG; ++jj) wf[b][jj] = *reinterpret_cast<const bf16x8*>(wr + (jh * JG + jj) * 16 * KP + kc * 32);
    };
    ldw(0, 0);
#pragma unroll
    for (int grp = 0; grp < NGRP; ++grp) {
      const int kc = grp / NGJ, jh = grp - kc * NGJ;
      if (grp + 1 < NGRP) ldw(grp + 1, (grp + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int jj = 0; jj < JG; ++jj) {
        acc[0][jh * JG + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[grp & 1][jj], af[0][kc], acc[0][jh * JG + jj], 0, 0, 0);
        acc[1][jh * JG + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[grp & 1][jj], af[1][kc], acc[1][jh * JG + jj], 0, 0, 0);
      }
      if (jh == NGJ - 1) { fetch(0, kc); fetch(1, kc); }   // this k-chunk of the next tile, into the registers just consumed
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int px = min(px0 + m * 16, p.M - 1);
      uint32_t bits_acc = 0u;
#pragma unroll
      for (int t = 0; t < NG; ++t) {
        const int c0 = t * 32 + g * 8;
        const float4 b0 = *reinterpret_cast<const float4*>(bias_l + c0), b1 = *reinterpret_cast<const float4*>(bias_l + c0 + 4);
        float v[8] = {acc[m][2 * t][0] + b0.x, acc[m][2 * t][1] + b0.y, acc[m][2 * t][2] + b0.z, acc[m][2 * t][3] + b0.w,
                      acc[m][2 * t + 1][0] + b1.x, acc[m][2 * t + 1][1] + b1.y, acc[m][2 * t + 1][2] + b1.z, acc[m][2 * t + 1][3] + b1.w};
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)(p.relu ? fmaxf(v[e], 0.f) : v[e]);
        // the store as inline asm: a store hipcc can see makes its waitcnt pass treat the vm counter as out of order (loads and stores
        // pending together) and wait vmcnt(0) -- for the previous tile's stores -- before the first MFMA of every tile.  Invisible
        // stores only make its counted waits for the rolling loads later than necessary, never earlier (they add to the counter).
        bf16* q = C + (int64_t)px * p.N + c0;
        const u32x4 ov = __builtin_bit_cast(u32x4, o);
        // (s_nop 1: a 16-byte store's data registers must not be written by the VALU for two wait states -- the hazard recognizer
        //  cannot see into the asm; without it rows 12..15 of a tile's last store carried the next tile's values)
        if constexpr (NT) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" :: "v"(q), "v"(ov) : "memory");
        else asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(q), "v"(ov) : "memory");
        if constexpr (BITS) {
          // (output > 0) of the stored bf16 values, eight into a byte (conv1x1_stream.hip: packed min / max as inline asm, byte permute, 4 x 8-bit dot)
          const uint32_t one2 = 0x00010001u, zero2 = 0u;
          uint32_t mq[4];
#pragma unroll
          for (int q2 = 0; q2 < 4; ++q2) {
            uint32_t tq;
            asm("v_pk_min_i16 %0, %1, %2" : "=v"(tq) : "v"(ov[q2]), "v"(one2));
            asm("v_pk_max_i16 %0, %1, %2" : "=v"(mq[q2]) : "v"(tq), "v"(zero2));
          }
          const uint32_t b03 = __builtin_amdgcn_perm(mq[1], mq[0], 0x06040200u), b47 = __builtin_amdgcn_perm(mq[3], mq[2], 0x06040200u);
          bits_acc |= (__builtin_amdgcn_udot4(b03, 0x08040201u, 0u, false) | (__builtin_amdgcn_udot4(b47, 0x08040201u, 0u, false) << 4)) << (t * 8);
        }
      }
      if constexpr (BITS) {
        // this lane's four mask bytes of the pixel's 128-channel slice: bytes 32 (c / 256) + 8 g + (c % 256) / 32 .. + 3, c = cbase
        static_assert(!BITS || NG == 4, "mask bits: 128-channel passes");
        unsigned char* bq = reinterpret_cast<unsigned char*>(p.bits) + (int64_t)px * (p.N >> 3) + (cbase >> 8) * 32 + g * 8 + ((cbase & 255) >> 5);
        asm volatile("global_store_dword %0, %1, off\n\ts_nop 0" :: "v"(bq), "v"(bits_acc) : "memory");
      }
    }
  }
}

template <int K1, int K2, int NH, bool NT, bool BITS = false>
int c1d_launch_nt(const DualK& p, int ncols, hipStream_t st) {
  constexpr int KP = K1 + K2 + 8;
  const size_t lds = (size_t)ncols * KP * 2 + (size_t)ncols * sizeof(float);
  auto fn = c1d_kernel<K1, K2, NH, NT, BITS>;
  static size_t attr = 0;
  if (lds > 64 * 1024 && lds > attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }      // (leave no stale error behind for the next launch check)
    attr = lds;
  }
  if (ncols != NH) return (int)hipErrorInvalidValue;             // one pass per tile (c1d_kernel's rolling prefetch)
  const int nsl = p.N / ncols, ntile = (p.M + 31) / 32;
  int blocks = (lds <= 72 * 1024 ? 512 : 256);
  blocks = (blocks + nsl - 1) / nsl;
  if (blocks * 8 > ntile) blocks = (ntile + 7) / 8;
  hipLaunchKernelGGL(fn, dim3(blocks, nsl), dim3(512), lds, st, p, ncols);
  GPV_CHECK_LAUNCH();
  return 0;
}

template <int K1, int K2, int NH>
int c1d_launch(const DualK& p, int ncols, hipStream_t st) {
  if (p.bits) {
    if constexpr (NH == 128) { if (!p.nt && p.relu && p.N % 256 == 0) return c1d_launch_nt<K1, K2, NH, false, true>(p, ncols, st); }
    return (int)hipErrorNotSupported;
  }
  return p.nt ? c1d_launch_nt<K1, K2, NH, true>(p, ncols, st) : c1d_launch_nt<K1, K2, NH, false>(p, ncols, st);
}

}  // namespace
}  // namespace gpvk

// y[B,OH,OW,N] = act( a1[B,OH,OW,K1] . w1[N,K1]^T + a2[B,IH2,IW2,K2](stride s2) . w2[N,K2]^T + bias[N] ),  bf16.
// Supported: (K1, K2) in {(64, 64), (128, 256)}: the first bottlenecks of layer1 / layer2; returns hipErrorNotSupported otherwise
// (the caller then runs the two convolutions one after the other).
extern "C" int gpv_conv1x1_dual(const void* a1, const void* w1, const void* a2, const void* w2, const float* bias, void* y, int B, int OH,
                                int OW, int K1, int IH2, int IW2, int K2, int s2, int N, int act, void* stream) {
  return gpv_conv1x1_dual_bits(a1, w1, a2, w2, bias, y, B, OH, OW, K1, IH2, IW2, K2, s2, N, act, nullptr, stream);
}

// + y_mask_bits (or NULL): (y > 0) as one bit per element in gpv_conv_args.y_mask_bits' layout (the (128, 256) -> 512 shape with ReLU only:
// hipErrorNotSupported otherwise, nothing launched)
extern "C" int gpv_conv1x1_dual_bits(const void* a1, const void* w1, const void* a2, const void* w2, const float* bias, void* y, int B, int OH,
                                     int OW, int K1, int IH2, int IW2, int K2, int s2, int N, int act, void* y_mask_bits, void* stream) {
  using namespace gpvk;
  if (!a1 || !a2 || !w1 || !w2 || !y || B <= 0) return (int)hipErrorInvalidValue;
  if (reinterpret_cast<uintptr_t>(y_mask_bits) & 15) return (int)hipErrorInvalidValue;
  if ((s2 != 1 && s2 != 2) || (OH - 1) * s2 >= IH2 || (OW - 1) * s2 >= IW2) return (int)hipErrorInvalidValue;
  if (act != GPV_ACT_NONE && act != GPV_ACT_RELU) return (int)hipErrorInvalidValue;
  if ((reinterpret_cast<uintptr_t>(a1) | reinterpret_cast<uintptr_t>(a2) | reinterpret_cast<uintptr_t>(w1) | reinterpret_cast<uintptr_t>(w2) |
       reinterpret_cast<uintptr_t>(y)) & 15) return (int)hipErrorInvalidValue;
  DualK p{};
  p.a1 = a1; p.a2 = a2; p.w1 = w1; p.w2 = w2; p.bias = bias; p.y = y;
  p.M = B * OH * OW; p.N = N; p.OH = OH; p.OW = OW; p.IH2 = IH2; p.IW2 = IW2; p.S2 = s2; p.ld1 = K1; p.ld2 = K2;
  p.relu = act == GPV_ACT_RELU;
  p.bits = y_mask_bits;
  p.nt = (int64_t)p.M * N * 2 >= ((int64_t)200 << 20);          // outputs beyond the 256 MB MALL are stored non-temporally (conv1x1_stream.hip)
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (K1 == 64 && K2 == 64 && N == 256) return c1d_launch<64, 64, 256>(p, 256, st);
  if (K1 == 128 && K2 == 256 && N == 512) return c1d_launch<128, 256, 128>(p, 128, st);
  return (int)hipErrorNotSupported;
}
