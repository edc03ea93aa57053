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
};

template <int NH>
__device__ __forceinline__ int c1d_chan(int L) {      // (conv1x1_stream.hip c1s_chan)
  const int hh = L / NH, w = L - hh * NH, j = w >> 4, r = w & 15;
  return hh * NH + (j >> 1) * 32 + (r >> 2) * 8 + (j & 1) * 4 + (r & 3);
}

template <int K1, int K2, int NH>
__global__ __launch_bounds__(512) void c1d_kernel(DualK p, int ncols) {
  constexpr int KT = K1 + K2, KP = KT + 8, KC1 = K1 / 32, KC = KT / 32, NTL = NH / 16, NG = NH / 32, SL1 = K1 / 8, SL = KT / 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16* Wl = reinterpret_cast<bf16*>(smem_raw);
  float* bias_l = reinterpret_cast<float*>(Wl + (size_t)ncols * KP);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, pl = lane & 15;
  const int cbase = (int)blockIdx.y * ncols;
  const bf16* A1 = reinterpret_cast<const bf16*>(p.a1);
  const bf16* A2 = reinterpret_cast<const bf16*>(p.a2);
  bf16* C = reinterpret_cast<bf16*>(p.y) + cbase;
  const int npass = ncols / NH;
  const int ntile = (p.M + 15) >> 4;
  const int nw = (int)gridDim.x * 8;
  int tile = (int)blockIdx.x * 8 + wave;
  const int ohw = p.OH * p.OW, ihw = p.IH2 * p.IW2;
  bf16x8 an[KC];
  auto fetch = [&](int t) {
    const int px = t * 16 + pl;
    const bool ok = t < ntile && px < p.M;
    int64_t ipx = px;
    if (p.S2 == 2) {
      const int b = px / ohw, r = px - b * ohw, oh = r / p.OW, ow = r - oh * p.OW;
      ipx = (int64_t)b * ihw + (int64_t)(2 * oh) * p.IW2 + 2 * ow;
    }
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      const bf16* src = kc < KC1 ? A1 + (int64_t)px * p.ld1 + kc * 32 + g * 8 : A2 + ipx * p.ld2 + (kc - KC1) * 32 + g * 8;
      if (ok) an[kc] = *reinterpret_cast<const bf16x8*>(src);
      else an[kc] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
  };
  fetch(tile);
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
  for (; tile < ntile; tile += nw) {
    bf16x8 af[KC];
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) af[kc] = an[kc];
    fetch(tile + nw);
    const int px = tile * 16 + pl;
    const bool pok = px < p.M;
    for (int hh = 0; hh < npass; ++hh) {
      f32x4 acc[NTL];
#pragma unroll
      for (int j = 0; j < NTL; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      const bf16* wrow = Wl + (hh * NH + pl) * KP + g * 8;
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) {
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
          const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wrow + j * 16 * KP + kc * 32);
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af[kc], acc[j], 0, 0, 0);
        }
      }
#pragma unroll
      for (int t = 0; t < NG; ++t) {
        const int c0 = hh * NH + t * 32 + g * 8;
        const float4 b0 = *reinterpret_cast<const float4*>(bias_l + c0), b1 = *reinterpret_cast<const float4*>(bias_l + c0 + 4);
        float v[8] = {acc[2 * t][0] + b0.x, acc[2 * t][1] + b0.y, acc[2 * t][2] + b0.z, acc[2 * t][3] + b0.w,
                      acc[2 * t + 1][0] + b1.x, acc[2 * t + 1][1] + b1.y, acc[2 * t + 1][2] + b1.z, acc[2 * t + 1][3] + b1.w};
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)(p.relu ? fmaxf(v[e], 0.f) : v[e]);
        if (pok) {
          bf16* q = C + (int64_t)px * p.N + c0;
          if (p.nt) __builtin_nontemporal_store(__builtin_bit_cast(u32x4, o), reinterpret_cast<u32x4*>(q));
          else *reinterpret_cast<bf16x8*>(q) = o;
        }
      }
    }
  }
}

template <int K1, int K2, int NH>
int c1d_launch(const DualK& p, int ncols, hipStream_t st) {
  constexpr int KP = K1 + K2 + 8;
  const size_t lds = (size_t)ncols * KP * 2 + (size_t)ncols * sizeof(float);
  auto fn = c1d_kernel<K1, K2, NH>;
  static size_t attr = 0;
  if (lds > 64 * 1024 && lds > attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }      // (leave no stale error behind for the next launch check)
    attr = lds;
  }
  const int nsl = p.N / ncols, ntile = (p.M + 15) / 16;
  int blocks = (lds <= 72 * 1024 ? 512 : 256);
  blocks = (blocks + nsl - 1) / nsl;
  if (blocks * 8 > ntile) blocks = (ntile + 7) / 8;
  hipLaunchKernelGGL(fn, dim3(blocks, nsl), dim3(512), lds, st, p, ncols);
  GPV_CHECK_LAUNCH();
  return 0;
}

}  // namespace
}  // namespace gpvk

// y[B,OH,OW,N] = act( a1[B,OH,OW,K1] . w1[N,K1]^T + a2[B,IH2,IW2,K2](stride s2) . w2[N,K2]^T + bias[N] ),  bf16.
// Supported: (K1, K2) in {(64, 64), (128, 256)}: the first bottlenecks of layer1 / layer2; returns hipErrorNotSupported otherwise
// (the caller then runs the two convolutions one after the other).
extern "C" int gpv_conv1x1_dual(const void* a1, const void* w1, const void* a2, const void* w2, const float* bias, void* y, int B, int OH,
                                int OW, int K1, int IH2, int IW2, int K2, int s2, int N, int act, void* stream) {
  using namespace gpvk;
  if (!a1 || !a2 || !w1 || !w2 || !y || B <= 0) return (int)hipErrorInvalidValue;
  if ((s2 != 1 && s2 != 2) || (OH - 1) * s2 >= IH2 || (OW - 1) * s2 >= IW2) return (int)hipErrorInvalidValue;
  if (act != GPV_ACT_NONE && act != GPV_ACT_RELU) return (int)hipErrorInvalidValue;
  if ((reinterpret_cast<uintptr_t>(a1) | reinterpret_cast<uintptr_t>(a2) | reinterpret_cast<uintptr_t>(w1) | reinterpret_cast<uintptr_t>(w2) |
       reinterpret_cast<uintptr_t>(y)) & 15) return (int)hipErrorInvalidValue;
  DualK p{};
  p.a1 = a1; p.a2 = a2; p.w1 = w1; p.w2 = w2; p.bias = bias; p.y = y;
  p.M = B * OH * OW; p.N = N; p.OH = OH; p.OW = OW; p.IH2 = IH2; p.IW2 = IW2; p.S2 = s2; p.ld1 = K1; p.ld2 = K2;
  p.relu = act == GPV_ACT_RELU;
  p.nt = (int64_t)p.M * N * 2 >= ((int64_t)200 << 20);          // outputs beyond the 256 MB MALL are stored non-temporally (conv1x1_stream.hip)
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (K1 == 64 && K2 == 64 && N == 256) return c1d_launch<64, 64, 256>(p, 256, st);
  if (K1 == 128 && K2 == 256 && N == 512) return c1d_launch<128, 256, 128>(p, 128, st);
  return (int)hipErrorNotSupported;
}
