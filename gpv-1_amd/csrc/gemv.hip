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
  const TI* arow[MR];
#pragma unroll
  for (int m = 0; m < MR; ++m) arow[m] = A + (int64_t)min(m, p.M - 1) * p.lda;      // rows beyond M repeat the last one (never stored)
#pragma unroll 2
  for (int k = lane * 8; k < p.K; k += 512) {
    float w[NW][8], x[MR][8];
#pragma unroll
    for (int j = 0; j < NW; ++j) Ld8<TI>::ld(brow[j] + k, w[j]);
#pragma unroll
    for (int m = 0; m < MR; ++m) Ld8<TI>::ld(arow[m] + k, x[m]);
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
      for (int j = 0; j < NW; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[m][j] = fmaf(x[m][e], w[j][e], acc[m][j]);
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

// one workgroup of 1024 threads per row: the index of the largest x[r, v] + addend[v]; equal values -> the lowest index; NaNs
// never win.  16-byte pieces of the row when its base and pitch allow (V = 10000 bf16: 1250 pieces, 1.2 per thread).
template <typename T>
__global__ __launch_bounds__(1024) void argmax_rows_kernel(const T* __restrict__ x, int64_t ld, const float* __restrict__ addend, int V, int vec,
                                                           int64_t* o0, int64_t s0, int64_t* o1, int64_t s1,
                                                           const T* __restrict__ table, int64_t ldt, const T* __restrict__ pos_row, T* xnext, int D) {
  // table (optional): the picked index also selects the NEXT input row, xnext[r, :] = table[index, :] (+ pos_row[:]) -- the
  // greedy decoder's embedding gather, input transform (precomputed per vocabulary entry) and position add of the next token
  // ride in the launch that picks the token (decode.py: three graph nodes per token fewer)
  const int r = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const T* row = x + (int64_t)r * ld;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  const int V8 = vec ? (V & ~7) : 0;
  for (int v = threadIdx.x * 8; v < V8; v += 8192) {
    float f[8], a[8];
    Ld8<T>::ld(row + v, f);
    if (addend) {
      Ld8<float>::ld(addend + v, a);
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] += a[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (f[e] > best) { best = f[e]; bi = v + e; }           // ascending v within a thread: the first of equal values stays
  }
  for (int v = V8 + threadIdx.x; v < V; v += 1024) {
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
  __shared__ float sb[16];
  __shared__ int si[16];
  if (lane == 0) { sb[wave] = best; si[wave] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w)
      if (sb[w] > best || (sb[w] == best && si[w] < bi)) { best = sb[w]; bi = si[w]; }
    if (bi == 0x7fffffff) bi = 0;                     // a row of NaNs / -inf only
    if (o0) o0[(int64_t)r * s0] = bi;
    if (o1) o1[(int64_t)r * s1] = bi;
    si[0] = bi;
  }
  if (table) {                                        // (uniform)
    __syncthreads();
    const int idx = si[0];
    for (int d = threadIdx.x * 8; d < D; d += 8192) {
      float f[8], a[8];
      Ld8<T>::ld(table + (int64_t)idx * ldt + d, f);
      if (pos_row) {
        Ld8<T>::ld(pos_row + d, a);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] += a[e];
      }
      Ld8<T>::st(xnext + (int64_t)r * D + d, f);
    }
  }
}

// LayerNorm -> Linear for <= 4 rows of <= 1024 columns: xn = LayerNorm(x + s) * gamma + beta, y = act(xn W^T + bias).
// Every wave normalises the rows itself (the same lane layout, sums and rounding as ln_fwd_kernel: the fused result equals the
// two launches bit for bit), wave 0 of workgroup 0 also stores xn -- the residual input of the next LayerNorm.
struct LnGemvK {
  const void* x; const void* s; const float* gamma; const float* beta; float eps; void* xn;
  const void* W; int64_t ldw; const float* bias; void* y; int64_t ldy;
  int rows, N, K, act;
  const float* s_part; int s_parts; const float* s_bias;      // s[r, c] = (T)(sum_h s_part[(r * s_parts + h) * K + c] + s_bias[c])  instead of `s`
};

template <typename T, int MR, int NW>
__global__ __launch_bounds__(256) void ln_gemv_kernel(LnGemvK p) {
  constexpr int NV = 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = (blockIdx.x * 4 + wave) * NW;
  // Round 6: with per-head partial rows as the sublayer output (attn1_proj_kernel) every WAVE used to fetch all of them -- 8 heads x
  // 768 floats = 24 KB per wave, 96 KB per workgroup through ONE CU's vector-memory path (12 - 18 B/clk: ~2 us; a plain 768 x 768
  // matrix-vector node is 2.3 us, this one was 5.0 -- tools/probe_decode_nodes.py).  Now the workgroup sums them ONCE: thread c owns
  // the 8-column piece c of a row, adds its eight partial pieces in head order (the same additions in the same order as before) and
  // leaves the rounded sublayer output in LDS; every wave then runs the unchanged LayerNorm on it.  Bit-identical results.
  __shared__ float s_sum[MR][NV * 64 * 8];
  const int cols = p.K;
  if (p.s_part) {
    const int pieces = cols >> 3;
    for (int idx = threadIdx.x; idx < p.rows * pieces; idx += 256) {
      const int m = idx / pieces, c = (idx - m * pieces) * 8;
      float t[8];
      if (p.s_bias) Ld8<float>::ld(p.s_bias + c, t);
      else {
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = 0.f;
      }
      for (int h0 = 0; h0 < p.s_parts; h0 += 8) {                 // eight partial rows requested at once, added in head order
        float uu[8][8];
#pragma unroll
        for (int hh = 0; hh < 8; ++hh)
          if (h0 + hh < p.s_parts) Ld8<float>::ld(p.s_part + ((int64_t)m * p.s_parts + h0 + hh) * cols + c, uu[hh]);
#pragma unroll
        for (int hh = 0; hh < 8; ++hh)
          if (h0 + hh < p.s_parts) {
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] += uu[hh][e];
          }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) s_sum[m][c + e] = (float)(T)t[e];      // (what a stored sublayer output would hold)
    }
  }
  const bool active = n0 < p.N;
  const bool writer = blockIdx.x == 0 && wave == 0;
  // the weight pieces and gamma / beta are requested first: their latency passes under the LayerNorm arithmetic
  typedef typename std::conditional<std::is_same<T, float>::value, float __attribute__((ext_vector_type(8))), bf16x8>::type RawT;
  const T* Wp = reinterpret_cast<const T*>(p.W);
  RawT wraw[NV][NW];
  float gm[NV][8], bt[NV][8];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 8;
    if (c < cols) {
#pragma unroll
      for (int j = 0; j < NW; ++j) wraw[i][j] = *reinterpret_cast<const RawT*>(Wp + (int64_t)min(min(n0, p.N - 1) + j, p.N - 1) * p.ldw + c);
      if (p.gamma) { Ld8<float>::ld(p.gamma + c, gm[i]); Ld8<float>::ld(p.beta + c, bt[i]); }
    }
  }
  if (p.s_part) __syncthreads();                    // (uniform: a kernel argument)
  if (!active) return;
  float xn[MR][NV][8];
#pragma unroll
  for (int m = 0; m < MR; ++m) {
    if (m >= p.rows) {
#pragma unroll
      for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) xn[m][i][e] = 0.f;
      continue;
    }
    const T* xr = reinterpret_cast<const T*>(p.x) + (int64_t)m * cols;
    const T* sr = p.s ? reinterpret_cast<const T*>(p.s) + (int64_t)m * cols : nullptr;
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * 8;
      if (c < cols) {
        Ld8<T>::ld(xr + c, xn[m][i]);
        if (sr) {
          float t[8];
          Ld8<T>::ld(sr + c, t);
#pragma unroll
          for (int e = 0; e < 8; ++e) xn[m][i][e] += t[e];
        } else if (p.s_part) {                       // the sublayer output as per-head partial sums (attn1_proj_kernel), summed above
#pragma unroll
          for (int e = 0; e < 8; ++e) xn[m][i][e] += s_sum[m][c + e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) sum += xn[m][i][e];
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) xn[m][i][e] = 0.f;
      }
    }
    sum = wave_sum(sum);
    const float mu = sum / cols;
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * 8;
      if (c < cols) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = xn[m][i][e] - mu; var += d * d; }
      }
    }
    var = wave_sum(var) / cols;
    const float rs = rsqrtf(var + p.eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * 8;
      if (c < cols) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float n = (xn[m][i][e] - mu) * rs;
          o[e] = (float)(T)(p.gamma ? n * gm[i][e] + bt[i][e] : n);    // the value the LayerNorm kernel stores
          xn[m][i][e] = o[e];
        }
        if (writer) Ld8<T>::st(reinterpret_cast<T*>(p.xn) + (int64_t)m * cols + c, o);
      }
    }
  }
  float acc[MR][NW];
#pragma unroll
  for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int j = 0; j < NW; ++j) acc[m][j] = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 8;
    if (c < cols) {
#pragma unroll
      for (int j = 0; j < NW; ++j) {
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[m][j] = fmaf(xn[m][i][e], (float)wraw[i][j][e], acc[m][j]);
      }
    }
  }
  float mine = 0.f;
#pragma unroll
  for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const float t = wave_sum(acc[m][j]);
      if (lane == m * NW + j) mine = t;
    }
  const int m = lane / NW, j = lane - m * NW, n = n0 + j;
  if (lane >= MR * NW || m >= p.rows || n >= p.N) return;
  float v = mine;
  if (p.bias) v += p.bias[n];
  if (p.act == GPV_ACT_RELU) v = fmaxf(v, 0.f);
  else if (p.act == GPV_ACT_GELU) v = gelu_erf(v);
  reinterpret_cast<T*>(p.y)[(int64_t)m * p.ldy + n] = (T)v;
}

// Single-query attention with the output projection folded in (the decode step: one new token attends over its t + 1 cached keys
// or over the memory).  Workgroup (4 waves) = (sequence b, head h, slab of 64 output columns): it recomputes the head's
// attention -- scores of the Sk keys (16 lanes per key row, 16 keys per pass), softmax through LDS, o_h = P V -- and
// multiplies o_h (rounded like a stored attention output) with its 64 x dh block of the out-projection weight.  The per-head
// partial rows part[b][h][:] are summed in head order by the LayerNorm + Linear kernel that consumes them (ln_gemv_kernel): the
// out-projection, its bias and its output round trip cost no launch.  K / V of a head are 2 * Sk * dh elements (44 KB for 116 keys):
// recomputing them per slab is cheaper than a node of the graph.
struct Attn1K {
  const void* q; int64_t q_bs; const void* k; int64_t k_bs, k_rs; const void* v; int64_t v_bs, v_rs;
  const void* Wo; int64_t ldw; float* part; int B, H, Sk, dh, slabs; float scale;
};

template <typename T>
__global__ __launch_bounds__(256) void attn1_proj_kernel(Attn1K p) {
  __shared__ float sc_s[256];          // scores, then probabilities (Sk <= 256)
  __shared__ float red_s[4][128];      // per-wave partial o_h
  __shared__ float o_s[128];
  __shared__ float wred[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int slab = blockIdx.x % p.slabs, h = (blockIdx.x / p.slabs) % p.H, b = blockIdx.x / (p.slabs * p.H);
  const int dh = p.dh, D = p.H * dh, pieces = dh >> 3;
  const T* q = reinterpret_cast<const T*>(p.q) + (int64_t)b * p.q_bs + h * dh;
  const T* K = reinterpret_cast<const T*>(p.k) + (int64_t)b * p.k_bs + h * dh;
  const T* V = reinterpret_cast<const T*>(p.v) + (int64_t)b * p.v_bs + h * dh;
  // a key row is `pieces` 16-byte pieces: 16 lanes per key (pieces <= 16), 4 keys per wave and pass, 16 per workgroup.
  // Every K / V / Wo piece a thread will use is requested before anything is computed (the cache rows were written by other
  // XCDs' kernels: each dependent load is a ~1 us round trip to the MALL) -- one latency instead of 2 * Sk / 16 + 3.
  typedef typename std::conditional<std::is_same<T, float>::value, float __attribute__((ext_vector_type(8))), bf16x8>::type RawT;
  constexpr int MAXIT = 16;                               // Sk <= 256
  const int kl = lane >> 4, pc = lane & 15;
  const bool live = pc < pieces;
  const int j0 = wave * 4 + kl;
  RawT qraw, kraw[MAXIT], vraw[MAXIT], wraw[4];
  if (live) qraw = *reinterpret_cast<const RawT*>(q + pc * 8);
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    const int j = j0 + it * 16;
    if (live && j < p.Sk) kraw[it] = *reinterpret_cast<const RawT*>(K + (int64_t)j * p.k_rs + pc * 8);
  }
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    const int j = j0 + it * 16;
    if (live && j < p.Sk) vraw[it] = *reinterpret_cast<const RawT*>(V + (int64_t)j * p.v_rs + pc * 8);
  }
  const int n = slab * 64 + (tid >> 2), quarter = tid & 3;
  {
    const T* w = reinterpret_cast<const T*>(p.Wo) + (int64_t)min(n, D - 1) * p.ldw + h * dh;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (quarter + 4 * i < pieces) wraw[i] = *reinterpret_cast<const RawT*>(w + (quarter + 4 * i) * 8);
  }
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    const int j = j0 + it * 16;
    if (it * 16 >= p.Sk) break;                            // uniform
    float a = 0.f;
    if (live && j < p.Sk) {
#pragma unroll
      for (int e = 0; e < 8; ++e) a = fmaf((float)qraw[e], (float)kraw[it][e], a);
    }
    a += __shfl_xor(a, 1); a += __shfl_xor(a, 2); a += __shfl_xor(a, 4); a += __shfl_xor(a, 8);
    if (pc == 0 && j < p.Sk) sc_s[j] = a * p.scale;
  }
  __syncthreads();
  const float sv = tid < p.Sk ? sc_s[tid] : -INFINITY;
  float mx = wave_max(sv);
  if (lane == 0) wred[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(wred[0], wred[1]), fmaxf(wred[2], wred[3]));
  const float ev = tid < p.Sk ? __expf(sv - mx) : 0.f;
  float l = wave_sum(ev);
  if (lane == 0) wred[4 + wave] = l;
  if (tid < p.Sk) sc_s[tid] = (float)(T)ev;               // the probabilities enter P V rounded, as in the MFMA kernels
  __syncthreads();
  const float inv = 1.f / (wred[4] + wred[5] + wred[6] + wred[7]);
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    const int j = j0 + it * 16;
    if (it * 16 >= p.Sk) break;
    if (live && j < p.Sk) {
      const float pj = sc_s[j];
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = fmaf(pj, (float)vraw[it][e], acc[e]);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) { acc[e] += __shfl_xor(acc[e], 16); acc[e] += __shfl_xor(acc[e], 32); }
  if (kl == 0 && live) {
#pragma unroll
    for (int e = 0; e < 8; ++e) red_s[wave][pc * 8 + e] = acc[e];
  }
  __syncthreads();
  if (tid < dh) o_s[tid] = (float)(T)((red_s[0][tid] + red_s[1][tid] + red_s[2][tid] + red_s[3][tid]) * inv);   // rounded like a stored attention output
  __syncthreads();
  // this slab's 64 columns of the out-projection: 4 lanes per column, pieces dealt round-robin
  float a = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = quarter + 4 * i;
    if (c < pieces) {
#pragma unroll
      for (int e = 0; e < 8; ++e) a = fmaf(o_s[c * 8 + e], (float)wraw[i][e], a);
    }
  }
  a += __shfl_xor(a, 1); a += __shfl_xor(a, 2);
  if (n < D && quarter == 0) p.part[((int64_t)b * p.H + h) * D + n] = a;
}

template <typename T, int MR>
int launch_ln_nw(const LnGemvK& k, hipStream_t st) {
  const int nw = k.N >= 4096 ? 4 : (k.N >= 2048 ? 2 : 1);
  const dim3 block(256);
  if (nw == 4) ln_gemv_kernel<T, MR, 4><<<dim3((k.N + 15) / 16), block, 0, st>>>(k);
  else if (nw == 2) ln_gemv_kernel<T, MR, 2><<<dim3((k.N + 7) / 8), block, 0, st>>>(k);
  else ln_gemv_kernel<T, MR, 1><<<dim3((k.N + 3) / 4), block, 0, st>>>(k);
  return (int)hipGetLastError();
}
template <typename T>
int launch_ln_mr(const LnGemvK& k, hipStream_t st) {
  if (k.rows == 1) return launch_ln_nw<T, 1>(k, st);
  if (k.rows == 2) return launch_ln_nw<T, 2>(k, st);
  return launch_ln_nw<T, 4>(k, st);
}

}  // namespace

// returns 0 = launched, -1 = not applicable, > 0 = hipError_t
int gemv_try_launch(const GemmK& k, int dtype_in, int dtype_out, int batch, hipStream_t st) {
  if (g_gemv_mode == 0 || g_kernel_forced || k.M > 8 || batch != 1 || k.accumulate || k.split_k > 1) return -1;
  if (k.rowscale || k.mask || k.dthresh || k.a_rowsum) return -1;
  if (k.M > 2 && k.N >= 4096) return -1;            // 4 columns x > 2 rows per wave: the tile kernels are faster (tools/bench_gemv.py)
  if (k.K % 8 != 0 || k.lda % 8 != 0 || k.ldb % 8 != 0 || !al16(k.A) || !al16(k.B)) return -1;
  if (dtype_in == GPV_BF16 && dtype_out == GPV_BF16) return launch_mr<bf16, bf16>(k, st);
  if (dtype_in == GPV_BF16 && dtype_out == GPV_F32) return launch_mr<bf16, float>(k, st);
  if (dtype_in == GPV_F32 && dtype_out == GPV_F32) return launch_mr<float, float>(k, st);
  return -1;
}

}  // namespace gpvk

extern "C" int gpv_argmax_rows_embed(const void* x, int64_t ld, const float* addend, int rows, int V, int dtype,
                                     int64_t* out0, int64_t stride0, int64_t* out1, int64_t stride1,
                                     const void* table, int64_t ldt, const void* pos_row, void* xnext, int D, void* stream) {
  if (!x || rows <= 0 || V <= 0 || (!out0 && !out1)) return (int)hipErrorInvalidValue;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int esz = dtype == GPV_F32 ? 4 : 2;
  if (table) {
    if (!xnext || D <= 0 || D % 8 != 0 || (ldt * esz) % 16 != 0) return (int)hipErrorInvalidValue;
    for (const void* q : {table, pos_row, (const void*)xnext})
      if (reinterpret_cast<uintptr_t>(q) & 15) return (int)hipErrorInvalidValue;
  }
  const int vec = (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (ld * esz) % 16 == 0 && (!addend || (reinterpret_cast<uintptr_t>(addend) & 15) == 0);
  if (dtype == GPV_BF16)
    gpvk::argmax_rows_kernel<bf16><<<dim3(rows), dim3(1024), 0, st>>>(reinterpret_cast<const bf16*>(x), ld, addend, V, vec, out0, stride0, out1, stride1,
                                                                     reinterpret_cast<const bf16*>(table), ldt, reinterpret_cast<const bf16*>(pos_row), reinterpret_cast<bf16*>(xnext), D);
  else if (dtype == GPV_F32)
    gpvk::argmax_rows_kernel<float><<<dim3(rows), dim3(1024), 0, st>>>(reinterpret_cast<const float*>(x), ld, addend, V, vec, out0, stride0, out1, stride1,
                                                                      reinterpret_cast<const float*>(table), ldt, reinterpret_cast<const float*>(pos_row), reinterpret_cast<float*>(xnext), D);
  else
    return (int)hipErrorInvalidValue;
  return (int)hipGetLastError();
}

extern "C" int gpv_argmax_rows(const void* x, int64_t ld, const float* addend, int rows, int V, int dtype,
                               int64_t* out0, int64_t stride0, int64_t* out1, int64_t stride1, void* stream) {
  return gpv_argmax_rows_embed(x, ld, addend, rows, V, dtype, out0, stride0, out1, stride1, nullptr, 0, nullptr, nullptr, 0, stream);
}

extern "C" int gpv_attention_row_proj(const void* q, int64_t q_bs, const void* k, int64_t k_bs, int64_t k_rs, const void* v, int64_t v_bs,
                                      int64_t v_rs, const void* Wo, int64_t ldw, float* partial, int B, int H, int Sk, int dh, float scale,
                                      int dtype, void* stream) {
  if (!q || !k || !v || !Wo || !partial || B <= 0 || H <= 0 || Sk <= 0 || Sk > 256 || dh <= 0 || dh > 128 || dh % 8 != 0) return (int)hipErrorInvalidValue;
  const int esz = dtype == GPV_F32 ? 4 : 2;
  for (const void* ptr : {q, k, v, Wo})
    if (reinterpret_cast<uintptr_t>(ptr) & 15) return (int)hipErrorInvalidValue;
  for (int64_t st : {q_bs, k_bs, k_rs, v_bs, v_rs, ldw})
    if ((st * esz) % 16 != 0) return (int)hipErrorInvalidValue;
  const int D = H * dh, slabs = (D + 63) / 64;
  gpvk::Attn1K a{q, q_bs, k, k_bs, k_rs, v, v_bs, v_rs, Wo, ldw, partial, B, H, Sk, dh, slabs, scale};
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == GPV_BF16) gpvk::attn1_proj_kernel<bf16><<<dim3(B * H * slabs), dim3(256), 0, st>>>(a);
  else if (dtype == GPV_F32) gpvk::attn1_proj_kernel<float><<<dim3(B * H * slabs), dim3(256), 0, st>>>(a);
  else return (int)hipErrorInvalidValue;
  return (int)hipGetLastError();
}

extern "C" int gpv_ln_linear_rows(const void* x, const void* s, const float* gamma, const float* beta, float eps, void* xn,
                                  const void* W, int64_t ldw, const float* bias, void* y, int64_t ldy,
                                  int rows, int N, int K, int act, int dtype, const float* s_partial, int s_parts, const float* s_bias,
                                  void* stream) {
  if (s_partial && (s || s_parts <= 0 || (reinterpret_cast<uintptr_t>(s_partial) & 15) || (reinterpret_cast<uintptr_t>(s_bias) & 15))) return (int)hipErrorInvalidValue;
  if (!x || !xn || !W || !y || rows <= 0 || rows > 4 || N <= 0 || K <= 0 || K > 1024 || K % 8 != 0 || ldw % 8 != 0) return (int)hipErrorInvalidValue;
  if ((gamma == nullptr) != (beta == nullptr) || xn == x || xn == s) return (int)hipErrorInvalidValue;
  for (const void* q : {x, s, (const void*)xn, W, (const void*)gamma, (const void*)beta})
    if (reinterpret_cast<uintptr_t>(q) & 15) return (int)hipErrorInvalidValue;
  gpvk::LnGemvK k{x, s, gamma, beta, eps, xn, W, ldw, bias, y, ldy, rows, N, K, act, s_partial, s_partial ? s_parts : 0, s_bias};
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == GPV_BF16) return gpvk::launch_ln_mr<bf16>(k, st);
  if (dtype == GPV_F32) return gpvk::launch_ln_mr<float>(k, st);
  return (int)hipErrorInvalidValue;
}
