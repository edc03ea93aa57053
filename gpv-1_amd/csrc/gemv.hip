// Few-row GEMM (M <= 8) and the greedy token pick: the decode step at small batch.
//
// One new token per sequence goes through the text decoder per step (exp/gpv/models/gpv.py:178-196 re-runs the whole prefix; the
// KV-cached schedule of decode.py feeds one row).  At B = 1 every projection of the step is a matrix-vector product: the tile
// kernels put N / 64 = 12 workgroups on a 768 x 768 weight and take 8 us per launch, 25 launches per token.  Here a WAVE owns
// 1..4 output columns: its 64 lanes walk the weight rows in 16-byte pieces (a 768-wide row is 1.5 coalesced requests), keep
// fp32 partial sums per (row, column) and fold them with cross-lane adds; N / (4 * columns) workgroups of four waves cover the
// chip, nothing goes through LDS, no barrier.  Epilogue as gpv_gemm documents it (alpha, bias, residual, activation).
#include "gemm_common.h"

namespace gpvk {

int g_gemv_mode = 1;          // gpv_set_option(GPV_OPT_GEMV, .): 0 never, 1 (default) wherever legal
long g_gemv_launches = 0;

namespace {

template <typename TI, typename TO, int MR, int NW>
__global__ __launch_bounds__(256) void gemv_kernel(GemmK p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = (blockIdx.x * 4 + wave) * NW;
  if (n0 >= p.N) return;
  const TI* __restrict__ A = reinterpret_cast<const TI*>(p.A);
  const TI* __restrict__ B = reinterpret_cast<const TI*>(p.B);
  float acc[MR][NW];
#pragma unroll
  for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int j = 0; j < NW; ++j) acc[m][j] = 0.f;
  const TI* brow[NW];
#pragma unroll
  for (int j = 0; j < NW; ++j) brow[j] = B + (int64_t)min(n0 + j, p.N - 1) * p.ldb;
#pragma unroll 2
  for (int k = lane * 8; k < p.K; k += 512) {
    float w[NW][8];
#pragma unroll
    for (int j = 0; j < NW; ++j) Ld8<TI>::ld(brow[j] + k, w[j]);
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      if (m < p.M) {                                   // uniform
        float x[8];
        Ld8<TI>::ld(A + (int64_t)m * p.lda + k, x);
#pragma unroll
        for (int j = 0; j < NW; ++j)
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[m][j] = fmaf(x[e], w[j][e], acc[m][j]);
      }
    }
  }
  float mine = 0.f;                                    // lane m * NW + j keeps the sum of (row m, column n0 + j)
#pragma unroll
  for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const float s = wave_sum(acc[m][j]);
      if (lane == m * NW + j) mine = s;
    }
  const int m = lane / NW, j = lane - m * NW, n = n0 + j;
  if (lane >= MR * NW || m >= p.M || n >= p.N) return;
  float v = p.alpha * mine;
  if (p.bias) v += p.bias[n];
  if (p.res) v += (float)reinterpret_cast<const TO*>(p.res)[(int64_t)m * p.ldr + n];
  if (p.act == GPV_ACT_RELU) v = fmaxf(v, 0.f);
  else if (p.act == GPV_ACT_GELU) v = gelu_erf(v);
  reinterpret_cast<TO*>(p.C)[(int64_t)m * p.ldc + n] = (TO)v;
}

template <typename TI, typename TO, int MR>
int launch_nw(const GemmK& k, hipStream_t st) {
  // columns per wave: as few as keep >= 256 workgroups (one per CU) busy
  const int nw = k.N >= 4096 ? 4 : (k.N >= 2048 ? 2 : 1);
  const dim3 block(256);
  if (nw == 4) gemv_kernel<TI, TO, MR, 4><<<dim3((k.N + 15) / 16), block, 0, st>>>(k);
  else if (nw == 2) gemv_kernel<TI, TO, MR, 2><<<dim3((k.N + 7) / 8), block, 0, st>>>(k);
  else gemv_kernel<TI, TO, MR, 1><<<dim3((k.N + 3) / 4), block, 0, st>>>(k);
  ++g_gemv_launches;
  const hipError_t e = hipGetLastError();
  return (int)e;
}

template <typename TI, typename TO>
int launch_mr(const GemmK& k, hipStream_t st) {
  if (k.M == 1) return launch_nw<TI, TO, 1>(k, st);
  if (k.M == 2) return launch_nw<TI, TO, 2>(k, st);
  if (k.M <= 4) return launch_nw<TI, TO, 4>(k, st);
  return launch_nw<TI, TO, 8>(k, st);
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// one workgroup per row: the index of the largest x[r, v] + addend[v]; equal values -> the lowest index; NaNs never win
template <typename T>
__global__ __launch_bounds__(256) void argmax_rows_kernel(const T* __restrict__ x, int64_t ld, const float* __restrict__ addend, int V,
                                                          int64_t* o0, int64_t s0, int64_t* o1, int64_t s1) {
  const int r = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const T* row = x + (int64_t)r * ld;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int v = threadIdx.x; v < V; v += 256) {
    float f = (float)row[v];
    if (addend) f += addend[v];
    if (f > best || (f == best && v < bi)) { best = f; bi = v; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o);
    const int oi = __shfl_xor(bi, o);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  __shared__ float sb[4];
  __shared__ int si[4];
  if (lane == 0) { sb[wave] = best; si[wave] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w)
      if (sb[w] > best || (sb[w] == best && si[w] < bi)) { best = sb[w]; bi = si[w]; }
    if (bi == 0x7fffffff) bi = 0;                     // a row of NaNs / -inf only
    if (o0) o0[(int64_t)r * s0] = bi;
    if (o1) o1[(int64_t)r * s1] = bi;
  }
}

}  // namespace

// returns 0 = launched, -1 = not applicable, > 0 = hipError_t
int gemv_try_launch(const GemmK& k, int dtype_in, int dtype_out, int batch, hipStream_t st) {
  if (g_gemv_mode == 0 || k.M > 8 || batch != 1 || k.accumulate || k.split_k > 1) return -1;
  if (k.rowscale || k.mask || k.dthresh || k.a_rowsum) return -1;
  if (k.K % 8 != 0 || k.lda % 8 != 0 || k.ldb % 8 != 0 || !al16(k.A) || !al16(k.B)) return -1;
  if (dtype_in == GPV_BF16 && dtype_out == GPV_BF16) return launch_mr<bf16, bf16>(k, st);
  if (dtype_in == GPV_BF16 && dtype_out == GPV_F32) return launch_mr<bf16, float>(k, st);
  if (dtype_in == GPV_F32 && dtype_out == GPV_F32) return launch_mr<float, float>(k, st);
  return -1;
}

}  // namespace gpvk

extern "C" int gpv_argmax_rows(const void* x, int64_t ld, const float* addend, int rows, int V, int dtype,
                               int64_t* out0, int64_t stride0, int64_t* out1, int64_t stride1, void* stream) {
  if (!x || rows <= 0 || V <= 0 || (!out0 && !out1)) return (int)hipErrorInvalidValue;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == GPV_BF16)
    gpvk::argmax_rows_kernel<bf16><<<dim3(rows), dim3(256), 0, st>>>(reinterpret_cast<const bf16*>(x), ld, addend, V, out0, stride0, out1, stride1);
  else if (dtype == GPV_F32)
    gpvk::argmax_rows_kernel<float><<<dim3(rows), dim3(256), 0, st>>>(reinterpret_cast<const float*>(x), ld, addend, V, out0, stride0, out1, stride1);
  else
    return (int)hipErrorInvalidValue;
  return (int)hipGetLastError();
}
