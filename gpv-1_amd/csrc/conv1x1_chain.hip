// A layer1 bottleneck's tail AND the next bottleneck's conv1 in one launch (gfx950):
//
//     y = ReLU( conv3(a) * bn3 [+ downsample(x) * bn_d | + identity] )        [pixels, 256]   (stored: the next block's identity)
//     z = ReLU( conv1_next(y) * bn1_next )                                    [pixels, 64 | 128]
//
// (torchvision Bottleneck.forward, exp/gpv/models/backbone.py:93-95; conv1 / layer1 are frozen, :61-63.)  As two launches the 256-
// channel map of layer1 -- 314 MB at B = 32, more than the MALL holds -- is written by the first and read back by the second.
// Here the second GEMM runs on registers: in conv1x1_stream.hip's layout (a wave owns 16 pixels, the output channels permuted when
// the weights are staged) the two accumulator tiles 2t, 2t + 1 leave lane (pixel, g) with the 8 consecutive channels 32 t + 8 g ..
// + 7 of its pixel -- after bias / residual / ReLU and rounding to bf16 that IS the B fragment of mfma_f32_16x16x32_bf16 for the
// k-slice t of the next convolution (K = 256 = 8 slices).  So after storing y's 16 bytes a lane feeds them to N2 / 16 more MFMAs
// against the next conv1's weights (LDS resident next to this block's).  Same products, same fp32 order as c1s_kernel<256, N2>:
// bit-identical to the two launches (tests/test_kernels_gpu.py).
#include "gemm_common.h"

namespace gpvk {
namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct ChainK {
  const void* a1; const void* a2; const void* w1; const void* w2; const float* bias; const void* res; void* y;
  const void* wn; const float* bias_n; void* z;
  int M;                          // output pixels
  int OH, OW, IH2, IW2, S2;       // a2's spatial extent and stride (a1 has the output's)
  int nt;
  void* zbits;                    // round 6: (z > 0) as one bit per element, N2 = 128: bit c & 7 of byte 4 ((c % 32) / 8) + c / 32 of the pixel's 16 (gpv_conv_args.y_mask_bits' order for 128 channels), or NULL
};

template <int NH>
__device__ __forceinline__ int c1c_chan(int L) {      // (conv1x1_stream.hip c1s_chan)
  const int hh = L / NH, w = L - hh * NH, j = w >> 4, r = w & 15;
  return hh * NH + (j >> 1) * 32 + (r >> 2) * 8 + (j & 1) * 4 + (r & 3);
}

// K1 = 64 channels of a1 (+ K2 = 64 of a2: the downsample branch of a stage's first block), N = 256, N2 = 64 | 128
template <int K1, int K2, int N2, bool RES, bool BITS = false>
__global__ __launch_bounds__(512) void c1c_kernel(ChainK p) {
  constexpr int N = 256, KT = K1 + K2, KP = KT + 8, KC1 = K1 / 32, KC = KT / 32, NTL = N / 16, NG = N / 32, SL1 = K1 / 8, SL = KT / 8;
  constexpr int KP2 = N + 8, NTL2 = N2 / 16, NG2 = N2 / 32, SL2 = N / 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16* Wl = reinterpret_cast<bf16*>(smem_raw);                       // [256][KP]
  bf16* Wn = Wl + (size_t)N * KP;                                      // [N2][KP2]
  float* bias_l = reinterpret_cast<float*>(Wn + (size_t)N2 * KP2);     // [256] then [N2]
  float* bias_n = bias_l + N;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, pl = lane & 15;
  const bf16* A1 = reinterpret_cast<const bf16*>(p.a1);
  const bf16* A2 = reinterpret_cast<const bf16*>(p.a2);
  const bf16* R = reinterpret_cast<const bf16*>(p.res);
  bf16* C = reinterpret_cast<bf16*>(p.y);
  bf16* Z = reinterpret_cast<bf16*>(p.z);
  const int ntile = (p.M + 15) >> 4;
  const int nw = (int)gridDim.x * 8;
  int tile = (int)blockIdx.x * 8 + wave;
  const int ohw = p.OH * p.OW, ihw = p.IH2 * p.IW2;
  bf16x8 an[KC];
  auto fetch = [&](int t) {
    const int px = t * 16 + pl;
    const bool ok = t < ntile && px < p.M;
    int64_t ipx = px;
    if (K2 > 0 && p.S2 == 2) {
      const int b = px / ohw, r = px - b * ohw, oh = r / p.OW, ow = r - oh * p.OW;
      ipx = (int64_t)b * ihw + (int64_t)(2 * oh) * p.IW2 + 2 * ow;
    }
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      const bf16* src = kc < KC1 ? A1 + (int64_t)px * K1 + kc * 32 + g * 8 : A2 + ipx * K2 + (kc - KC1) * 32 + g * 8;
      if (ok) an[kc] = *reinterpret_cast<const bf16x8*>(src);
      else an[kc] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
  };
  fetch(tile);
  {
    const bf16* W1 = reinterpret_cast<const bf16*>(p.w1);
    const bf16* W2 = reinterpret_cast<const bf16*>(p.w2);
    const bf16* WN = reinterpret_cast<const bf16*>(p.wn);
    stage_chunks16<512, 8>(N * SL, tid,
        [&](int idx) { const int L = idx / SL, sl = idx - L * SL; const int c = c1c_chan<N>(L);
                       return sl < SL1 ? W1 + (int64_t)c * K1 + sl * 8 : W2 + (int64_t)c * K2 + (sl - SL1) * 8; },
        [&](int idx) { const int L = idx / SL, sl = idx - L * SL; return Wl + L * KP + sl * 8; });
    stage_chunks16<512, 8>(N2 * SL2, tid,
        [&](int idx) { const int L = idx / SL2, sl = idx - L * SL2; return WN + (int64_t)c1c_chan<N2>(L) * N + sl * 8; },
        [&](int idx) { const int L = idx / SL2, sl = idx - L * SL2; return Wn + L * KP2 + sl * 8; });
    for (int c = tid; c < N; c += 512) bias_l[c] = p.bias ? p.bias[c] : 0.f;
    for (int c = tid; c < N2; c += 512) bias_n[c] = p.bias_n ? p.bias_n[c] : 0.f;
  }
  __syncthreads();
  for (; tile < ntile; tile += nw) {
    bf16x8 af[KC];
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) af[kc] = an[kc];
    fetch(tile + nw);
    const int px = tile * 16 + pl;
    const bool pok = px < p.M;
    bf16x8 rv[RES ? NG : 1];
    if constexpr (RES) {
#pragma unroll
      for (int t = 0; t < NG; ++t) {
        const bf16* q = R + (int64_t)px * N + t * 32 + g * 8;
        if (pok) rv[t] = p.nt ? __builtin_bit_cast(bf16x8, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(q)))
                              : *reinterpret_cast<const bf16x8*>(q);
        else rv[t] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      }
    }
    f32x4 acc[NTL];
#pragma unroll
    for (int j = 0; j < NTL; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // (the LDS offsets are laundered once per tile: otherwise LICM hoists all (N + N2) / 16 x K / 32 weight fragments out of the
    //  tile loop -- 96 .. 192 of them -- and spills hundreds of registers)
    int woff = pl * KP + g * 8, noff = pl * KP2 + g * 8;
    asm volatile("" : "+v"(woff), "+v"(noff));
    const bf16* wrow = Wl + woff;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
#pragma unroll
      for (int j = 0; j < NTL; ++j) {
        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wrow + j * 16 * KP + kc * 32);
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af[kc], acc[j], 0, 0, 0);
      }
    }
    f32x4 acc2[NTL2];
#pragma unroll
    for (int j = 0; j < NTL2; ++j) acc2[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16* nrow = Wn + noff;
#pragma unroll
    for (int t = 0; t < NG; ++t) {
      const int c0 = t * 32 + g * 8;
      const float4 b0 = *reinterpret_cast<const float4*>(bias_l + c0), b1 = *reinterpret_cast<const float4*>(bias_l + c0 + 4);
      float v[8] = {acc[2 * t][0] + b0.x, acc[2 * t][1] + b0.y, acc[2 * t][2] + b0.z, acc[2 * t][3] + b0.w,
                    acc[2 * t + 1][0] + b1.x, acc[2 * t + 1][1] + b1.y, acc[2 * t + 1][2] + b1.z, acc[2 * t + 1][3] + b1.w};
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float x = v[e];
        if constexpr (RES) x += (float)rv[t][e];
        o[e] = (bf16)fmaxf(x, 0.f);
      }
      if (pok) {
        bf16* q = C + (int64_t)px * N + c0;
        if (p.nt) __builtin_nontemporal_store(__builtin_bit_cast(u32x4, o), reinterpret_cast<u32x4*>(q));
        else *reinterpret_cast<bf16x8*>(q) = o;
      }
      // k-slice t of the next convolution: channels 32 t + 8 g .. + 7 of this lane's pixel
#pragma unroll
      for (int j = 0; j < NTL2; ++j) {
        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(nrow + j * 16 * KP2 + t * 32);
        acc2[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, o, acc2[j], 0, 0, 0);
      }
    }
    uint32_t zb = 0u;
#pragma unroll
    for (int u = 0; u < NG2; ++u) {
      const int c0 = u * 32 + g * 8;
      const float4 b0 = *reinterpret_cast<const float4*>(bias_n + c0), b1 = *reinterpret_cast<const float4*>(bias_n + c0 + 4);
      const float v[8] = {acc2[2 * u][0] + b0.x, acc2[2 * u][1] + b0.y, acc2[2 * u][2] + b0.z, acc2[2 * u][3] + b0.w,
                          acc2[2 * u + 1][0] + b1.x, acc2[2 * u + 1][1] + b1.y, acc2[2 * u + 1][2] + b1.z, acc2[2 * u + 1][3] + b1.w};
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16)fmaxf(v[e], 0.f);
      if (pok) *reinterpret_cast<bf16x8*>(Z + (int64_t)px * N2 + c0) = o;
      if constexpr (BITS) {
        // (z > 0) of the stored bf16 values, eight into a byte (conv1x1_stream.hip: packed min / max as inline asm, byte permute, 4 x 8-bit dot)
        const u32x4 ow = __builtin_bit_cast(u32x4, o);
        const uint32_t one2 = 0x00010001u, zero2 = 0u;
        uint32_t mq[4];
#pragma unroll
        for (int q2 = 0; q2 < 4; ++q2) {
          uint32_t tq;
          asm("v_pk_min_i16 %0, %1, %2" : "=v"(tq) : "v"(ow[q2]), "v"(one2));
          asm("v_pk_max_i16 %0, %1, %2" : "=v"(mq[q2]) : "v"(tq), "v"(zero2));
        }
        const uint32_t b03 = __builtin_amdgcn_perm(mq[1], mq[0], 0x06040200u), b47 = __builtin_amdgcn_perm(mq[3], mq[2], 0x06040200u);
        zb |= (__builtin_amdgcn_udot4(b03, 0x08040201u, 0u, false) | (__builtin_amdgcn_udot4(b47, 0x08040201u, 0u, false) << 4)) << (u * 8);
      }
    }
    if constexpr (BITS) {
      static_assert(!BITS || NG2 == 4, "mask bits: the 128-channel conv1 of layer2.0");
      if (pok) *reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(p.zbits) + (int64_t)px * (N2 / 8) + g * 4) = zb;
    }
  }
}

template <int K1, int K2, int N2, bool RES, bool BITS = false>
int c1c_launch(const ChainK& p, hipStream_t st) {
  constexpr int N = 256;
  const size_t lds = (size_t)N * (K1 + K2 + 8) * 2 + (size_t)N2 * (N + 8) * 2 + (size_t)(N + N2) * sizeof(float);
  auto fn = c1c_kernel<K1, K2, N2, RES, BITS>;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }      // (leave no stale error behind for the next launch check)
    attr = true;
  }
  const int ntile = (p.M + 15) / 16;
  static const int bpc = tune_env("GPV_C1C_BLOCKS", 0);
  int blocks = bpc > 0 ? bpc : (lds <= 80 * 1024 ? 512 : 256);        // persistent waves: two 8-wave blocks per CU when the LDS allows
  if (blocks * 8 > ntile) blocks = (ntile + 7) / 8;
  hipLaunchKernelGGL(fn, dim3(blocks), dim3(512), lds, st, p);
  GPV_CHECK_LAUNCH();
  return 0;
}

}  // namespace
}  // namespace gpvk

// y[B,OH,OW,256] = relu( a1[B,OH,OW,K1] . w1[256,K1]^T  (+ a2[B,IH2,IW2,K2](stride s2) . w2[256,K2]^T)  (+ res[B,OH,OW,256]) + bias[256] )
// z[B,OH,OW,N2]  = relu( y . wn[N2,256]^T + bias_n[N2] ),   bf16, both stored.
// Supported: K1 = 64, K2 in {0, 64} (a2 / w2 NULL when 0), N2 in {64, 128}: layer1's three bottleneck tails with the conv1 of the
// bottleneck that follows (layer1.1, layer1.2, layer2.0); hipErrorNotSupported otherwise (the caller launches the convolutions
// one by one).  res and a2 are mutually exclusive (identity branch | downsample branch).
extern "C" int gpv_conv1x1_chain(const void* a1, const void* w1, int K1, const void* a2, const void* w2, int K2, int IH2, int IW2, int s2,
                                 const void* res, const float* bias, void* y, int B, int OH, int OW, int N, const void* wn,
                                 const float* bias_n, void* z, int N2, void* stream) {
  return gpv_conv1x1_chain_bits(a1, w1, K1, a2, w2, K2, IH2, IW2, s2, res, bias, y, B, OH, OW, N, wn, bias_n, z, N2, nullptr, stream);
}

// + z_mask_bits (or NULL): (z > 0) as one bit per element (gpv_conv_args.y_mask_bits' layout for 128 channels); the identity-branch form with
// N2 = 128 only (layer1's last tail + layer2.0's conv1, whose output is the ReLU mask of layer2.0's stride-2 3x3 backward-data): 801 otherwise
extern "C" int gpv_conv1x1_chain_bits(const void* a1, const void* w1, int K1, const void* a2, const void* w2, int K2, int IH2, int IW2, int s2,
                                      const void* res, const float* bias, void* y, int B, int OH, int OW, int N, const void* wn,
                                      const float* bias_n, void* z, int N2, void* z_mask_bits, void* stream) {
  using namespace gpvk;
  if (!a1 || !w1 || !y || !wn || !z || B <= 0) return (int)hipErrorInvalidValue;
  if (z_mask_bits && (K2 != 0 || !res || N2 != 128)) return (int)hipErrorNotSupported;
  if (reinterpret_cast<uintptr_t>(z_mask_bits) & 15) return (int)hipErrorInvalidValue;
  if (K1 != 64 || N != 256 || (K2 != 0 && K2 != 64) || (N2 != 64 && N2 != 128)) return (int)hipErrorNotSupported;
  if ((K2 != 0) != (a2 != nullptr && w2 != nullptr) || (K2 != 0 && res != nullptr)) return (int)hipErrorInvalidValue;
  if (K2 != 0 && ((s2 != 1 && s2 != 2) || (OH - 1) * s2 >= IH2 || (OW - 1) * s2 >= IW2)) return (int)hipErrorInvalidValue;
  if ((reinterpret_cast<uintptr_t>(a1) | reinterpret_cast<uintptr_t>(a2) | reinterpret_cast<uintptr_t>(w1) | reinterpret_cast<uintptr_t>(w2) |
       reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(res) | reinterpret_cast<uintptr_t>(wn) | reinterpret_cast<uintptr_t>(z)) & 15)
    return (int)hipErrorInvalidValue;
  ChainK p{};
  p.a1 = a1; p.a2 = a2; p.w1 = w1; p.w2 = w2; p.bias = bias; p.res = res; p.y = y; p.wn = wn; p.bias_n = bias_n; p.z = z;
  p.M = B * OH * OW; p.OH = OH; p.OW = OW; p.IH2 = IH2; p.IW2 = IW2; p.S2 = s2;
  p.nt = (int64_t)p.M * N * 2 >= ((int64_t)200 << 20);          // outputs beyond the 256 MB MALL are stored (and their residual read) non-temporally
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (K2 == 64) return N2 == 64 ? c1c_launch<64, 64, 64, false>(p, st) : c1c_launch<64, 64, 128, false>(p, st);
  p.zbits = z_mask_bits;
  if (z_mask_bits) return c1c_launch<64, 0, 128, true, true>(p, st);
  if (res) return N2 == 64 ? c1c_launch<64, 0, 64, true>(p, st) : c1c_launch<64, 0, 128, true>(p, st);
  return N2 == 64 ? c1c_launch<64, 0, 64, false>(p, st) : c1c_launch<64, 0, 128, false>(p, st);
}
