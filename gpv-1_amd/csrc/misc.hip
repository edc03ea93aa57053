// Memory-bound kernels: LayerNorm(+residual+dropout) fwd/bwd, vocabulary softmax-CE, max-pool,
// image layout conversion, RoI separable weights, casts / weight prep, element-wise helpers,
// fused AdamW.  All use 16-byte vector accesses where the layout allows (gfx950 HBM-bound rules).
#include "common.h"
#include <cstdlib>
#include "../../include/gpv_hip.h"

namespace {

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline int grid1d(int64_t n, int per_block) {
  int64_t g = cdiv(n, per_block);
  return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

// ------------------------------------------------------------------------------------------
// LayerNorm forward: one wave per row (cols <= 8*64*MAXV), values kept in registers.
// y = LN(x + drop(s)) * g + b
// HALF (cols <= 256, the DETR width): a row is 32 lanes x 8 elements, so a wave takes TWO rows -- with one row per wave half
// the lanes of every DETR LayerNorm (9600 x 256, 3200 x 256) sat idle and a wave had half the bytes in flight.
// ------------------------------------------------------------------------------------------
template <bool HALF> __device__ __forceinline__ float row_sum(float v) {
  if (HALF) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
  }
  return wave_sum(v);
}

template <typename T, int NV, bool HALF>   // NV = number of 8-element vectors per lane (cols <= NV*512)
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const T* __restrict__ s,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     T* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd,
                                                     int rows, int cols, float eps, uint32_t dthresh, float dscale,
                                                     uint64_t seed, const uint64_t* seed_dev, const T* __restrict__ pos, int pos_rows,
                                                     T* __restrict__ y2) {
  // y2 (optional second output) = y + pos[row % pos_rows]: the `x + pos` / `tgt + query_pos` sums the DETR layers feed their q / k
  // projections (transformer.py:150,216,221) leave the LayerNorm that produces x instead of being a launch of their own; computed
  // from the ROUNDED y, so it is bit-identical to gpv_add on the stored output
  if (dthresh) seed = eff_seed(seed, seed_dev);
  constexpr int LW = HALF ? 32 : 64;                       // lanes per row
  const int lane = threadIdx.x & (LW - 1);
  const int row = HALF ? blockIdx.x * 8 + (threadIdx.x >> 5) : blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* xr = x + (int64_t)row * cols;
  const T* sr = s ? s + (int64_t)row * cols : nullptr;
  float v[NV][8];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * LW + lane) * 8;
    if (c < cols) {
      Ld8<T>::ld(xr + c, v[i]);
      if (sr) {
        float t[8];
        Ld8<T>::ld(sr + c, t);
        const uint32_t keep8 = dthresh ? drop_mask<8>(seed, (uint64_t)row * cols + c, dthresh) : 0xffu;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float sv = t[e];
          if (dthresh) sv = ((keep8 >> e) & 1u) ? sv * dscale : 0.f;
          v[i][e] += sv;
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) sum += v[i][e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
    }
  }
  sum = row_sum<HALF>(sum);
  const float mu = sum / cols;
  float var = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * LW + lane) * 8;
    if (c < cols) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { float d = v[i][e] - mu; var += d * d; }
    }
  }
  var = row_sum<HALF>(var) / cols;
  const float rs = rsqrtf(var + eps);
  if (lane == 0) { if (mean) mean[row] = mu; if (rstd) rstd[row] = rs; }
  T* yr = y + (int64_t)row * cols;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * LW + lane) * 8;
    if (c < cols) {
      float o[8], gm[8], bt[8];
      if (gamma) { Ld8<float>::ld(gamma + c, gm); Ld8<float>::ld(beta + c, bt); }      // cols % 8 == 0: two 16-byte loads each
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float n = (v[i][e] - mu) * rs;
        o[e] = gamma ? n * gm[e] + bt[e] : n;
      }
      Ld8<T>::st(yr + c, o);
      if (y2) {
        float pv[8];
        Ld8<T>::ld(pos + (int64_t)(row % pos_rows) * cols + c, pv);
#pragma unroll
        for (int e = 0; e < 8; ++e) pv[e] += (float)(T)o[e];
        Ld8<T>::st(y2 + (int64_t)row * cols + c, pv);
      }
    }
  }
}

// LayerNorm backward: one wave (HALF: half a wave) per row. z = x + drop(s) is recomputed.
// dz = rstd * (g*dy - mean(g*dy) - zhat*mean(g*dy*zhat)); dx = dz ; ds = dz * dropmask
// dgamma/dbeta accumulated per block in LDS then atomics.  NW waves per block: the atomics (blocks x 2 cols of them on cols / 16
// cache lines) were 11 us of a 19.5 us launch on the 9600 x 256 DETR shapes with 512 blocks of 4 waves; 16 waves per block keep
// the rows in flight and quarter the blocks.
template <typename T, int NV, bool HALF, int NW>
__global__ __launch_bounds__(NW * 64) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ s,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, T* __restrict__ dx, T* __restrict__ ds,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta, int rows, int cols,
                                                     int rows_per_block, uint32_t dthresh, float dscale, uint64_t seed,
                                                     const uint64_t* seed_dev, const T* __restrict__ dy2, float* __restrict__ partials) {
  // dy2 (optional): a second gradient of the same output (the consumer of the forward's y2 = y + pos), summed on load in fp32
  // partials (optional, instead of dgamma / dbeta): the block's column sums go to partials[blockIdx.x][dgamma | dbeta] with plain
  // stores and gpv_colsum_fold_group adds them up later, off the backward chain -- the same-address atomics of gridDim.x blocks
  // were 4.5 - 6.8 us of a 14.3 us launch on the 9600 x 256 DETR shape (tools/bench_ln.py)
  if (dthresh) seed = eff_seed(seed, seed_dev);
  extern __shared__ float lds[];   // [4 waves][2][cols]: every wave parks its partial dgamma | dbeta, no LDS atomics
  constexpr int LW = HALF ? 32 : 64, RPI = HALF ? 2 * NW : NW;      // lanes per row, rows per block iteration
  const int lane = threadIdx.x & (LW - 1), wave = threadIdx.x >> 6;
  const int slot = HALF ? threadIdx.x >> 5 : wave;            // which of the block's RPI concurrent rows
  float ag[NV][8], ab[NV][8];
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) ag[i][e] = ab[i][e] = 0.f;

  const int r_begin = blockIdx.x * rows_per_block;
  const int r_end = min(rows, r_begin + rows_per_block);
  for (int row = r_begin + slot; row < r_end; row += RPI) {      // (HALF: the two halves of a wave may run one iteration apart;
    const float mu = mean[row], rs = rstd[row];                  //  every shuffle below stays inside a half)
    float zh[NV][8], gy[NV][8];
    uint32_t keep[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * LW + lane) * 8;
      keep[i] = 0xffu;
      if (c < cols) {
        float xv[8], dv[8];
        Ld8<T>::ld(x + (int64_t)row * cols + c, xv);
        Ld8<T>::ld(dy + (int64_t)row * cols + c, dv);
        if (dy2) {
          float d2[8];
          Ld8<T>::ld(dy2 + (int64_t)row * cols + c, d2);
#pragma unroll
          for (int e = 0; e < 8; ++e) dv[e] += d2[e];
        }
        if (dthresh) keep[i] = drop_mask<8>(seed, (uint64_t)row * cols + c, dthresh);
        if (s) {
          float t[8];
          Ld8<T>::ld(s + (int64_t)row * cols + c, t);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float sv = t[e];
            if (dthresh) sv = ((keep[i] >> e) & 1u) ? sv * dscale : 0.f;
            xv[e] += sv;
          }
        }
        float gm[8];
        if (gamma) Ld8<float>::ld(gamma + c, gm);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          zh[i][e] = (xv[e] - mu) * rs;
          float g = gamma ? gm[e] : 1.f;
          gy[i][e] = dv[e] * g;
          s1 += gy[i][e];
          s2 += gy[i][e] * zh[i][e];
          ag[i][e] += dv[e] * zh[i][e];
          ab[i][e] += dv[e];
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) zh[i][e] = gy[i][e] = 0.f;
      }
    }
    s1 = row_sum<HALF>(s1) / cols;
    s2 = row_sum<HALF>(s2) / cols;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * LW + lane) * 8;
      if (c < cols) {
        float o[8], o2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          o[e] = rs * (gy[i][e] - s1 - zh[i][e] * s2);
          if (dthresh) o2[e] = ((keep[i] >> e) & 1u) ? o[e] * dscale : 0.f;
        }
        Ld8<T>::st(dx + (int64_t)row * cols + c, o);
        if (dthresh && ds) Ld8<T>::st(ds + (int64_t)row * cols + c, o2);
      }
    }
  }
  if (dgamma || partials) {
    if (HALF) {                                        // the two halves of a wave hold partials of the same columns
#pragma unroll
      for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) { ag[i][e] += __shfl_xor(ag[i][e], 32); ab[i][e] += __shfl_xor(ab[i][e], 32); }
    }
    float* wg = lds + (wave & 3) * 2 * cols;         // four [dgamma | dbeta] rows, wave w adds into row w % 4 in round w / 4
#pragma unroll
    for (int rnd = 0; rnd < NW / 4; ++rnd) {
      if ((wave >> 2) == rnd) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          const int c = (i * LW + lane) * 8;
          if (c < cols && (!HALF || (threadIdx.x & 32) == 0)) {
            float4 g0 = make_float4(ag[i][0], ag[i][1], ag[i][2], ag[i][3]), g1 = make_float4(ag[i][4], ag[i][5], ag[i][6], ag[i][7]);
            float4 b0 = make_float4(ab[i][0], ab[i][1], ab[i][2], ab[i][3]), b1 = make_float4(ab[i][4], ab[i][5], ab[i][6], ab[i][7]);
            if (rnd > 0) {
              const float4 p0 = *reinterpret_cast<float4*>(wg + c), p1 = *reinterpret_cast<float4*>(wg + c + 4);
              const float4 q0 = *reinterpret_cast<float4*>(wg + cols + c), q1 = *reinterpret_cast<float4*>(wg + cols + c + 4);
              g0.x += p0.x; g0.y += p0.y; g0.z += p0.z; g0.w += p0.w; g1.x += p1.x; g1.y += p1.y; g1.z += p1.z; g1.w += p1.w;
              b0.x += q0.x; b0.y += q0.y; b0.z += q0.z; b0.w += q0.w; b1.x += q1.x; b1.y += q1.y; b1.z += q1.z; b1.w += q1.w;
            }
            *reinterpret_cast<float4*>(wg + c) = g0; *reinterpret_cast<float4*>(wg + c + 4) = g1;
            *reinterpret_cast<float4*>(wg + cols + c) = b0; *reinterpret_cast<float4*>(wg + cols + c + 4) = b1;
          }
        }
      }
      __syncthreads();
    }
    for (int c = threadIdx.x; c < 2 * cols; c += NW * 64) {
      const float v = lds[c] + lds[2 * cols + c] + lds[4 * cols + c] + lds[6 * cols + c];
      if (partials) partials[(int64_t)blockIdx.x * 2 * cols + c] = v;
      else atomicAdd(c < cols ? dgamma + c : dbeta + (c - cols), v);
    }
  }
}

// ------------------------------------------------------------------------------------------
// softmax cross-entropy: one block per row, V up to any size (3 passes over L2-resident row).
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void ce_kernel(const T* __restrict__ logits, int64_t ld, const int64_t* __restrict__ target,
                                                 float* __restrict__ loss, T* __restrict__ dlogits, const float* __restrict__ gscale,
                                                 int V) {
  __shared__ float red[8];
  const int row = blockIdx.x;
  const T* lr = logits + (int64_t)row * ld;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float mx = -INFINITY;
  for (int c = tid; c < V; c += 256) mx = fmaxf(mx, (float)lr[c]);
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
  for (int c = tid; c < V; c += 256) sum += __expf((float)lr[c] - mx);
  sum = wave_sum(sum);
  if (lane == 0) red[4 + wave] = sum;
  __syncthreads();
  sum = red[4] + red[5] + red[6] + red[7];
  const float lse = mx + logf(sum);
  const int64_t t = target[row];
  const bool valid = t >= 0 && t < V;
  if (tid == 0) loss[row] = valid ? lse - (float)lr[t] : 0.f;
  if (dlogits) {
    const float gs = valid ? (gscale ? gscale[row] : 1.f) : 0.f;
    T* dr = dlogits + (int64_t)row * ld;
    for (int c = tid; c < V; c += 256) {
      float p = __expf((float)lr[c] - lse);
      dr[c] = (T)((p - (c == t ? 1.f : 0.f)) * gs);
    }
  }
}

// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void image_to_nhwc4_kernel(const float* __restrict__ img, T* __restrict__ out, int B, int H, int W, int pad, int Hp,
                                      int Wp) {
  const int64_t total = (int64_t)B * Hp * Wp;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int xp = (int)(i % Wp);
    int64_t r = i / Wp;
    int yp = (int)(r % Hp);
    int b = (int)(r / Hp);
    int x = xp - pad, y = yp - pad;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (x >= 0 && x < W && y >= 0 && y < H) {
      const float* s = img + ((int64_t)b * 3 * H + y) * W + x;
      v[0] = s[0]; v[1] = s[(int64_t)H * W]; v[2] = s[2 * (int64_t)H * W];
    }
    T* d = out + i * 4;
    d[0] = (T)v[0]; d[1] = (T)v[1]; d[2] = (T)v[2]; d[3] = (T)v[3];
  }
}

template <typename T>
__global__ void maxpool_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C, int OH, int OW) {
  const int C8 = C / 8;
  const int64_t total = (int64_t)B * OH * OW * C8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c8 = (int)(i % C8);
    int64_t r = i / C8;
    int ow = (int)(r % OW); r /= OW;
    int oh = (int)(r % OH);
    int b = (int)(r / OH);
    float m[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
    for (int dy = 0; dy < 3; ++dy) {
      int ih = oh * 2 - 1 + dy;
      if (ih < 0 || ih >= H) continue;
      for (int dx = 0; dx < 3; ++dx) {
        int iw = ow * 2 - 1 + dx;
        if (iw < 0 || iw >= W) continue;
        float v[8];
        Ld8<T>::ld(x + (((int64_t)b * H + ih) * W + iw) * C + c8 * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], v[e]);
      }
    }
    Ld8<T>::st(y + (((int64_t)b * OH + oh) * OW + ow) * C + c8 * 8, m);
  }
}

// ------------------------------------------------------------------------------------------
// RoI separable weights (oracle.roi_axis_weights): one wave per roi.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void roi_axis(float start, float length, int size, float* acc /* LDS [size] */, int lane) {
  // accumulate bilinear weights of all 7*grid sample points into acc (lanes cooperate by sample)
  const int pooled = 7;
  const int grid = (int)ceilf(length / pooled);
  if (grid <= 0) return;
  const float bin = length / pooled;
  const float norm = 1.0f / (pooled * grid);
  for (int sidx = lane; sidx < pooled * grid; sidx += 64) {
    int p = sidx / grid, i = sidx - p * grid;
    float c = start + p * bin + (i + 0.5f) * bin / grid;
    if (c < -1.0f || c > (float)size) continue;
    if (c <= 0.f) c = 0.f;
    int lo = (int)c, hi;
    if (lo >= size - 1) { hi = lo = size - 1; c = (float)lo; } else hi = lo + 1;
    float l = c - lo;
    atomicAdd(&acc[lo], (1.f - l) * norm);
    atomicAdd(&acc[hi], l * norm);
  }
}

template <typename T>
__global__ __launch_bounds__(64) void roi_weights_kernel(const float* __restrict__ boxes, T* __restrict__ wgt, int n_roi, int H,
                                                         int W, int64_t ldw) {
  __shared__ float ay[64], ax[64];
  const int roi = blockIdx.x, lane = threadIdx.x;
  ay[lane] = 0.f; ax[lane] = 0.f;
  __syncthreads();
  const float cx = boxes[roi * 4 + 0], cy = boxes[roi * 4 + 1], w = boxes[roi * 4 + 2], h = boxes[roi * 4 + 3];
  const float x1 = W * (cx - 0.5f * w) - 0.5f, x2 = W * (cx + 0.5f * w) - 0.5f;
  const float y1 = H * (cy - 0.5f * h) - 0.5f, y2 = H * (cy + 0.5f * h) - 0.5f;
  roi_axis(y1, y2 - y1, H, ay, lane);
  roi_axis(x1, x2 - x1, W, ax, lane);
  __syncthreads();
  T* out = wgt + (int64_t)roi * ldw;
  for (int i = lane; i < ldw; i += 64) {
    float v = 0.f;
    if (i < H * W) { int y = i / W, x = i - y * W; v = ay[y] * ax[x]; }
    out[i] = (T)v;
  }
}

// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y, int64_t n8, int64_t nb8) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    float u[8], v[8];
    Ld8<T>::ld(a + i * 8, u);
    Ld8<T>::ld(b + (i % nb8) * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) u[e] += v[e];
    Ld8<T>::st(y + i * 8, u);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, float* __restrict__ out, int rows, int cols, int64_t ld,
                                                     int rows_per_block) {
  // block handles a [rows_per_block x 256-col] slab; thread = column
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  float s = 0.f;
  for (int r = r0; r < r1; ++r) s += (float)x[(int64_t)r * ld + c];
  atomicAdd(&out[c], s);
}

// 16-byte form: a thread owns 8 consecutive columns, the block's 256 threads are CT column threads x (256 / CT) row lanes, every
// thread adds its partial sums with 8 atomics (the few-row / many-column sums of the query_pos gradients -- 32 x 25600 -- ran 9 us
// on 100 blocks of the scalar kernel)
template <typename T>
__global__ __launch_bounds__(256) void colsum8_kernel(const T* __restrict__ x, float* __restrict__ out, int rows, int cols, int64_t ld,
                                                      int ct, int rows_per_block) {
  const int cx = threadIdx.x % ct, ry = threadIdx.x / ct, rl = 256 / ct;
  const int c = ((int)blockIdx.x * ct + cx) * 8;
  if (c >= cols) return;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int r = r0 + ry; r < r1; r += rl) {
    float v[8];
    Ld8<T>::ld(x + (int64_t)r * ld + c, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] += v[e];
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) atomicAdd(&out[c + e], s[e]);
}

template <typename TS, typename TD>
__global__ void cast_kernel(const TS* __restrict__ s, TD* __restrict__ d, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) d[i] = (TD)(float)s[i];
}

// dst[r][c] = src[r][c]*scale[r]; dstT[c][r] likewise: 32x32 tiles through LDS
template <typename TD>
__global__ __launch_bounds__(256) void cast_rowscale_t_kernel(const float* __restrict__ src, const float* __restrict__ scale,
                                                              TD* __restrict__ dst, TD* __restrict__ dstT, int rows, int cols) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int j = ty; j < 32; j += 8) {
    int r = r0 + j, c = c0 + tx;
    float v = 0.f;
    if (r < rows && c < cols) {
      v = src[(int64_t)r * cols + c] * (scale ? scale[r] : 1.f);
      if (dst) dst[(int64_t)r * cols + c] = (TD)v;
    }
    tile[j][tx] = v;
  }
  if (!dstT) return;
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    int c = c0 + j, r = r0 + tx;
    if (r < rows && c < cols) dstT[(int64_t)c * rows + r] = (TD)tile[tx][j];
  }
}

// many cast-transposes in one launch: the problem table travels in the kernel argument, block -> (problem, 32x32 tile) by a
// linear scan of the tile prefix sums
struct TcGroup {
  int n;
  int tile_start[GPV_TC_GROUP_MAX + 1];
  gpv_tc_problem prob[GPV_TC_GROUP_MAX];
};
template <typename TD>
__global__ __launch_bounds__(256) void cast_transpose_group_kernel(const TcGroup g) {
  // 64x64 tiles: a thread reads 16 consecutive floats of a source row (four threads = one 256-byte run) and writes 16 consecutive
  // outputs of a destination row; the tile crosses through LDS with a 65-float pitch (conflict-free both ways)
  __shared__ float tile[64][65];
  int pi = 0;
  while (pi + 1 < g.n && (int)blockIdx.x >= g.tile_start[pi + 1]) ++pi;
  const gpv_tc_problem q = g.prob[pi];
  const int t = (int)blockIdx.x - g.tile_start[pi];
  const int tc = (q.cols + 63) >> 6;
  const int r0 = (t / tc) * 64, c0 = (t % tc) * 64;
  const int lr = threadIdx.x >> 2, seg = (threadIdx.x & 3) * 16;
  {
    const int r = r0 + lr, c = c0 + seg;
    const float* sp = q.src + (int64_t)r * q.cols + c;
    if (r < q.rows && c + 16 <= q.cols && (q.cols & 3) == 0) {
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const float4 x = *reinterpret_cast<const float4*>(sp + 4 * v);
        tile[lr][seg + 4 * v] = x.x; tile[lr][seg + 4 * v + 1] = x.y; tile[lr][seg + 4 * v + 2] = x.z; tile[lr][seg + 4 * v + 3] = x.w;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) tile[lr][seg + e] = (r < q.rows && c + e < q.cols) ? sp[e] : 0.f;
    }
  }
  __syncthreads();
  {
    const int c = c0 + lr, r = r0 + seg;          // destination row c (a source column), 16 consecutive source rows
    if (c >= q.cols) return;
    TD* dp = reinterpret_cast<TD*>(q.dstT) + (int64_t)c * q.rows + r;
    float v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = tile[seg + e][lr];
    if (r + 16 <= q.rows && (q.rows & 7) == 0) {
      Ld8<TD>::st(dp, v);
      Ld8<TD>::st(dp + 8, v + 8);
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e)
        if (r + e < q.rows) dp[e] = (TD)v[e];
    }
  }
}

// src [Cout][T][Cin] fp32 -> wf [Cout][T][Cin] scaled, wd [Cin][T][Cout] scaled
template <typename TD>
__global__ __launch_bounds__(256) void prep_conv_w_kernel(const float* __restrict__ src, const float* __restrict__ scale,
                                                          TD* __restrict__ wf, TD* __restrict__ wd, int Cout, int T, int Cin) {
  __shared__ float tile[32][33];
  const int t = blockIdx.z;
  const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8) {
    int co = co0 + j, ci = ci0 + tx;
    float v = 0.f;
    if (co < Cout && ci < Cin) {
      v = src[((int64_t)co * T + t) * Cin + ci] * (scale ? scale[co] : 1.f);
      if (wf) wf[((int64_t)co * T + t) * Cin + ci] = (TD)v;
    }
    tile[j][tx] = v;
  }
  if (!wd) return;
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    int ci = ci0 + j, co = co0 + tx;
    if (co < Cout && ci < Cin) wd[((int64_t)ci * T + t) * Cout + co] = (TD)tile[tx][j];
  }
}

template <typename TT, typename TO>
__global__ void embedding_kernel(const TT* __restrict__ table, const int64_t* __restrict__ ids, TO* __restrict__ out, int64_t n_ids,
                                 int dim) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_ids * dim; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = i / dim;
    int c = (int)(i - r * dim);
    out[i] = (TO)(float)table[ids[r] * dim + c];
  }
}

template <typename T>
__global__ void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n, uint32_t dthresh, float dscale, uint64_t seed,
                               const uint64_t* seed_dev) {
  seed = eff_seed(seed, seed_dev);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = (T)(drop_keep(seed, (uint64_t)i, dthresh) ? (float)x[i] * dscale : 0.f);
}

template <typename T>
__global__ void relevance_condition_kernel(const T* __restrict__ x, const float* __restrict__ logits, const float* __restrict__ tok,
                                           T* __restrict__ y, int rows, int dim) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (int64_t)rows * dim; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = i / dim;
    int c = (int)(i - r * dim);
    float l0 = logits[r * 2], l1 = logits[r * 2 + 1];
    float m = fmaxf(l0, l1);
    float e0 = __expf(l0 - m), e1 = __expf(l1 - m);
    float p0 = e0 / (e0 + e1), p1 = e1 / (e0 + e1);
    y[i] = (T)((float)x[i] + p0 * tok[c] + p1 * tok[dim + c]);
  }
}

// y = act(x) (act: 1 relu, 2 gelu-erf)
template <typename T>
__global__ void act_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n8, int act) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    float v[8];
    Ld8<T>::ld(x + i * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = act == GPV_ACT_RELU ? fmaxf(v[e], 0.f) : gelu_erf(v[e]);
    Ld8<T>::st(y + i * 8, v);
  }
}
// dx = dy * act'(ref) * alpha ; relu: ref = OUTPUT (ref > 0), gelu: ref = pre-activation
template <typename T>
__global__ void act_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ ref, T* __restrict__ dx, int64_t n8, int act,
                               float alpha) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    float g[8], r[8];
    Ld8<T>::ld(dy + i * 8, g);
    Ld8<T>::ld(ref + i * 8, r);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (act == GPV_ACT_RELU) g[e] = r[e] > 0.f ? g[e] * alpha : 0.f;
      else {
        const float z = r[e];
        const float cdf = 0.5f * (1.0f + erff(z * 0.70710678118654752f));
        const float pdf = 0.3989422804014327f * __expf(-0.5f * z * z);
        g[e] = g[e] * (cdf + z * pdf) * alpha;
      }
    }
    Ld8<T>::st(dx + i * 8, g);
  }
}

__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             bf16* __restrict__ plow, int64_t n, float lr, float b1, float b2, float eps, float wd, float bc1,
                             float bc2, const float* __restrict__ gscale, const uint16_t* __restrict__ seg_id,
                             const int32_t* __restrict__ seg_live) {
  const float gs = gscale ? *gscale : 1.f;
  const float l1 = logf(b1), l2 = logf(b2);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if (seg_id) {
      // seg_live[parameter] = the parameter's own Adam step count INCLUDING this step, 0 = it never received a gradient
      // (untouched, like torch 1.6).  torch keeps `step` per parameter, starting at its first gradient: a head that is first
      // touched at global step 100 takes its first update with the bias corrections of step 1.
      const int t = seg_live[seg_id[i >> 3]];
      if (!t) continue;
      bc1 = -expm1f((float)t * l1);
      bc2 = -expm1f((float)t * l2);
    }
    float gi = g[i] * gs;
    float pi = p[i] * (1.f - lr * wd);
    float mi = b1 * m[i] + (1.f - b1) * gi;
    float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    float denom = sqrtf(vi) / sqrtf(bc2) + eps;
    pi -= (lr / bc1) * mi / denom;
    p[i] = pi;
    if (plow) plow[i] = (bf16)pi;
  }
}

// The same update, 4 consecutive parameters per thread: every load / store instruction of a wave is one contiguous kilobyte
// (16 bytes per lane).  Same expressions per element as adamw_kernel: bit-identical results.  Measured in the training step
// (GPV_ADAMW_VEC = 0 scalar / 1 this / 2 this with non-temporal loads and stores): 18.64 / 18.46 / 18.62 ms -- the pass streams
// 30 bytes per parameter, but part of it still sits in the 256 MB MALL from the weight-gradient writes, which the non-temporal
// form gives up.  (8 parameters per thread -- one seg_id chunk -- was slower than the scalar kernel: two 16-byte loads per lane at
// a 32-byte lane stride touch every cache line twice.)
template <bool NT>
__global__ __launch_bounds__(256) void adamw_vec4_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                         float* __restrict__ v, bf16* __restrict__ plow, int64_t nquads, float lr, float b1,
                                                         float b2, float eps, float wd, float bc1, float bc2, const float* __restrict__ gscale,
                                                         const uint16_t* __restrict__ seg_id, const int32_t* __restrict__ seg_live) {
  const float gs = gscale ? *gscale : 1.f;
  const float l1 = logf(b1), l2 = logf(b2);
  const float decay = 1.f - lr * wd;
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < nquads; q += (int64_t)gridDim.x * blockDim.x) {
    if (seg_id) {
      const int t = seg_live[seg_id[q >> 1]];
      if (!t) continue;
      bc1 = -expm1f((float)t * l1);
      bc2 = -expm1f((float)t * l2);
    }
    const float sq2 = sqrtf(bc2), step = lr / bc1;
    const int64_t i = q * 4;
    auto ld = [](const float* a) { return NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a)) : *reinterpret_cast<const f32x4*>(a); };
    auto st = [](f32x4 x, float* a) { if (NT) __builtin_nontemporal_store(x, reinterpret_cast<f32x4*>(a)); else *reinterpret_cast<f32x4*>(a) = x; };
    const f32x4 gv = ld(g + i);
    f32x4 pv = ld(p + i), mv = ld(m + i), vv = ld(v + i);
    bf16x4 lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gi = gv[e] * gs;
      float pi = pv[e] * decay;
      const float mi = b1 * mv[e] + (1.f - b1) * gi;
      const float vi = b2 * vv[e] + (1.f - b2) * gi * gi;
      mv[e] = mi; vv[e] = vi;
      const float denom = sqrtf(vi) / sq2 + eps;
      pi -= step * mi / denom;
      pv[e] = pi;
      lo[e] = (bf16)pi;
    }
    st(pv, p + i); st(mv, m + i); st(vv, v + i);
    if (plow) *reinterpret_cast<bf16x4*>(plow + i) = lo;          // (read by the next step's forward: stays cacheable)
  }
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) s += x[i] * x[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

// ---- gradient-norm clip factor, deterministic (train_distr.py:423-425: clip_grad_norm_(detr params, 0.1)) ----------------------
// Every rank must derive the SAME bits from the same all-reduced gradient, or the replicas drift apart one ulp of the clip factor per
// step -- so no atomics (sumsq_kernel's float atomics arrive in a different order every run): pass 1 = CLIP_BLOCKS blocks, block b
// sums x^2 over ITS contiguous chunk in a fixed order (thread-strided partial sums, shuffle tree, the four waves in order) ->
// partial[b]; pass 2 = one block sums the partials in a fixed order (double), writes gscale = min(1, max_norm / (norm + 1e-6)) and,
// riding along, adds the per-parameter liveness flags into the per-parameter Adam step counts (pstep += live).
constexpr int CLIP_BLOCKS = 1024;
__global__ __launch_bounds__(256) void clip_partial_kernel(const float* __restrict__ x, int64_t nquads, float* __restrict__ partial) {
  __shared__ float red[4];
  const int64_t per = (nquads + CLIP_BLOCKS - 1) / CLIP_BLOCKS;
  const int64_t q0 = (int64_t)blockIdx.x * per, q1 = q0 + per < nquads ? q0 + per : nquads;
  float s = 0.f;
  for (int64_t q = q0 + threadIdx.x; q < q1; q += 256) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + q * 4);
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}
__global__ __launch_bounds__(256) void clip_final_kernel(const float* __restrict__ partial, float max_norm, float* __restrict__ gscale,
                                                         int32_t* __restrict__ pstep, const int32_t* __restrict__ live, int nparam) {
  __shared__ double red[256];
  if (partial) {
    double s = 0.0;
    for (int i = threadIdx.x; i < CLIP_BLOCKS; i += 256) s += (double)partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      const float norm = (float)sqrt(red[0]);
      const float c = max_norm / (norm + 1e-6f);
      *gscale = c < 1.f ? c : 1.f;
    }
  }
  if (pstep)
    for (int i = threadIdx.x; i < nparam; i += 256) pstep[i] += live[i];
}

#define ST(s) reinterpret_cast<hipStream_t>(s)
}  // namespace

extern "C" int gpv_layernorm_pos_fwd(const void* x, const void* s, const float* gamma, const float* beta, void* y, float* mean,
                                     float* rstd, int rows, int cols, float eps, float drop_p, uint64_t seed, const void* pos,
                                     int pos_rows, void* y2, int dtype, void* stream) {
  if (cols % 8 != 0 || cols > 4096 || rows <= 0) return (int)hipErrorInvalidValue;
  if ((pos == nullptr) != (y2 == nullptr) || (pos && (pos_rows <= 0 || rows % pos_rows != 0))) return (int)hipErrorInvalidValue;
  const uint32_t th = drop_p > 0.f ? drop_thresh(drop_p) : 0u;
  const float sc = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  const bool half = cols <= 256;                         // two rows per wave (see ln_fwd_kernel)
  dim3 grid(half ? (rows + 7) / 8 : (rows + 3) / 4), block(256);
#define LN_F(T, NV, H) hipLaunchKernelGGL((ln_fwd_kernel<T, NV, H>), grid, block, 0, ST(stream), (const T*)x, (const T*)s, gamma, beta, (T*)y, mean, rstd, rows, cols, eps, th, sc, seed, gpvk::g_seed_dev, (const T*)pos, pos_rows, (T*)y2)
  const int nv = (cols + 511) / 512;
  if (dtype == GPV_BF16) { if (half) LN_F(bf16, 1, true); else if (nv <= 1) LN_F(bf16, 1, false); else if (nv <= 2) LN_F(bf16, 2, false); else if (nv <= 5) LN_F(bf16, 5, false); else LN_F(bf16, 8, false); }
  else { if (half) LN_F(float, 1, true); else if (nv <= 1) LN_F(float, 1, false); else if (nv <= 2) LN_F(float, 2, false); else if (nv <= 5) LN_F(float, 5, false); else LN_F(float, 8, false); }
#undef LN_F
  GPV_CHECK_LAUNCH();
  return 0;
}

extern "C" int gpv_layernorm_fwd(const void* x, const void* s, const float* gamma, const float* beta, void* y, float* mean,
                                 float* rstd, int rows, int cols, float eps, float drop_p, uint64_t seed, int dtype, void* stream) {
  return gpv_layernorm_pos_fwd(x, s, gamma, beta, y, mean, rstd, rows, cols, eps, drop_p, seed, nullptr, 0, nullptr, dtype, stream);
}

namespace {
// launch geometry of ln_bwd_kernel (shared by the launch and gpv_layernorm_bwd_blocks)
struct LnBwdCfg { int nv, rpb, blocks; bool half, wide; };
LnBwdCfg ln_bwd_cfg(int rows, int cols, bool partial) {
  LnBwdCfg c;
  c.nv = (cols + 511) / 512;
  c.half = cols <= 256;
  c.wide = c.nv <= 2 && rows >= 2048;                    // 16 waves per block (register budget: the NV <= 2 bodies; few rows: no gain)
  static const int rpb_div = tune_env("GPV_LN_BWD_BLOCKS", 0);   // tuning only
  // few blocks: fewer same-address global atomics on dgamma; with per-block partials (no atomics) one block per CU
  const int target = rpb_div > 0 ? rpb_div : (c.wide ? (partial ? 256 : 160) : 512);
  const int rpi = (c.wide ? 16 : 4) * (c.half ? 2 : 1);
  c.rpb = (rows + target - 1) / target;
  if (c.rpb < rpi) c.rpb = rpi;
  c.blocks = (rows + c.rpb - 1) / c.rpb;
  return c;
}
}  // namespace

extern "C" int gpv_layernorm_bwd_blocks(int rows, int cols) {
  if (cols % 8 != 0 || cols > 4096 || rows <= 0) return -1;
  return ln_bwd_cfg(rows, cols, true).blocks;
}

extern "C" int gpv_layernorm_bwd3(const void* dy, const void* dy2, const void* x, const void* s, const float* gamma, const float* mean,
                                  const float* rstd, void* dx, void* ds, float* dgamma, float* dbeta, float* partials, int rows, int cols,
                                  float drop_p, uint64_t seed, int dtype, void* stream) {
  if (cols % 8 != 0 || cols > 4096 || rows <= 0 || !dy) return (int)hipErrorInvalidValue;
  if (partials && (dgamma || dbeta)) return (int)hipErrorInvalidValue;      // one or the other
  const uint32_t th = drop_p > 0.f ? drop_thresh(drop_p) : 0u;
  const float sc = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  const LnBwdCfg c = ln_bwd_cfg(rows, cols, partials != nullptr);
  const int nv = c.nv, rpb = c.rpb;
  const bool half = c.half, wide = c.wide;
  dim3 grid(c.blocks), block(wide ? 1024 : 256);
  const size_t lds = (dgamma || partials) ? 8 * (size_t)cols * sizeof(float) : 0;       // 4 waves x [dgamma | dbeta]; cols <= 4096 -> <= 128 KB
#define LN_B(T, NV, H, NW) hipLaunchKernelGGL((ln_bwd_kernel<T, NV, H, NW>), grid, block, lds, ST(stream), (const T*)dy, (const T*)x, (const T*)s, gamma, mean, rstd, (T*)dx, (T*)ds, dgamma, dbeta, rows, cols, rpb, th, sc, seed, gpvk::g_seed_dev, (const T*)dy2, partials)
#define LN_BW(T, NV, H) do { if (wide) LN_B(T, NV, H, 16); else LN_B(T, NV, H, 4); } while (0)
  if (dtype == GPV_BF16) { if (half) LN_BW(bf16, 1, true); else if (nv <= 1) LN_BW(bf16, 1, false); else if (nv <= 2) LN_BW(bf16, 2, false); else if (nv <= 5) LN_B(bf16, 5, false, 4); else LN_B(bf16, 8, false, 4); }
  else { if (half) LN_BW(float, 1, true); else if (nv <= 1) LN_BW(float, 1, false); else if (nv <= 2) LN_BW(float, 2, false); else if (nv <= 5) LN_B(float, 5, false, 4); else LN_B(float, 8, false, 4); }
#undef LN_BW
#undef LN_B
  GPV_CHECK_LAUNCH();
  return 0;
}

extern "C" int gpv_layernorm_bwd2(const void* dy, const void* dy2, const void* x, const void* s, const float* gamma, const float* mean,
                                  const float* rstd, void* dx, void* ds, float* dgamma, float* dbeta, int rows, int cols,
                                  float drop_p, uint64_t seed, int dtype, void* stream) {
  return gpv_layernorm_bwd3(dy, dy2, x, s, gamma, mean, rstd, dx, ds, dgamma, dbeta, nullptr, rows, cols, drop_p, seed, dtype, stream);
}

// out0[c] += sum_b partials[b][c], out1[c] += sum_b partials[b][cols + c] for a group of problems in one launch: a workgroup owns 64
// consecutive columns of one problem's [nblk][2 cols] partials, its four waves take every fourth row (coalesced 256-byte reads,
// eight in flight), LDS sum in a fixed order -- the result does not depend on the schedule.
namespace {
constexpr int FOLD_MAX = 64;
struct FoldG { gpv_fold_problem p[FOLD_MAX]; int first[FOLD_MAX + 1]; int n; };
__global__ __launch_bounds__(256) void colsum_fold_kernel(FoldG g) {
  __shared__ float red[4][64];
  int pi = 0;
  while (pi + 1 < g.n && (int)blockIdx.x >= g.first[pi + 1]) ++pi;
  const gpv_fold_problem q = g.p[pi];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = ((int)blockIdx.x - g.first[pi]) * 64 + lane, w2 = 2 * q.cols;
  float acc[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) acc[u] = 0.f;
  if (c < w2) {
    int b = wave;
    for (; b + 28 < q.nblk; b += 32) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] += q.partials[(int64_t)(b + 4 * u) * w2 + c];
    }
    for (; b < q.nblk; b += 4) acc[0] += q.partials[(int64_t)b * w2 + c];
  }
  red[wave][lane] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  if (wave == 0 && c < w2) {
    const float v = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    atomicAdd(c < q.cols ? q.out0 + c : q.out1 + (c - q.cols), v);       // (uncontended; atomic only because two problems may share a parameter)
  }
}
}  // namespace

extern "C" int gpv_colsum_fold_group(const gpv_fold_problem* problems, int n, void* stream) {
  if (n < 0 || (n > 0 && !problems)) return (int)hipErrorInvalidValue;
  for (int i0 = 0; i0 < n; i0 += FOLD_MAX) {
    FoldG g;
    g.n = n - i0 < FOLD_MAX ? n - i0 : FOLD_MAX;
    int blocks = 0;
    for (int i = 0; i < g.n; ++i) {
      const gpv_fold_problem& q = problems[i0 + i];
      if (!q.partials || !q.out0 || !q.out1 || q.nblk <= 0 || q.cols <= 0) return (int)hipErrorInvalidValue;
      g.p[i] = q;
      g.first[i] = blocks;
      blocks += (2 * q.cols + 63) / 64;
    }
    g.first[g.n] = blocks;
    hipLaunchKernelGGL(colsum_fold_kernel, dim3(blocks), dim3(256), 0, ST(stream), g);
    GPV_CHECK_LAUNCH();
  }
  return 0;
}

extern "C" int gpv_layernorm_bwd(const void* dy, const void* x, const void* s, const float* gamma, const float* mean,
                                 const float* rstd, void* dx, void* ds, float* dgamma, float* dbeta, int rows, int cols,
                                 float drop_p, uint64_t seed, int dtype, void* stream) {
  return gpv_layernorm_bwd2(dy, nullptr, x, s, gamma, mean, rstd, dx, ds, dgamma, dbeta, rows, cols, drop_p, seed, dtype, stream);
}

extern "C" int gpv_softmax_ce(const void* logits, int64_t ld, const int64_t* target, float* loss, void* dlogits, const float* gscale,
                              int rows, int V, int dtype, void* stream) {
  if (rows <= 0 || V <= 0) return (int)hipErrorInvalidValue;
  if (dtype == GPV_BF16) hipLaunchKernelGGL((ce_kernel<bf16>), dim3(rows), dim3(256), 0, ST(stream), (const bf16*)logits, ld, target, loss, (bf16*)dlogits, gscale, V);
  else hipLaunchKernelGGL((ce_kernel<float>), dim3(rows), dim3(256), 0, ST(stream), (const float*)logits, ld, target, loss, (float*)dlogits, gscale, V);
  GPV_CHECK_LAUNCH();
  return 0;
}

extern "C" int gpv_image_to_nhwc4(const float* img, void* out, int B, int H, int W, int pad, int Hp, int Wp, int dtype_out,
                                  void* stream) {
  const int64_t n = (int64_t)B * Hp * Wp;
  if (dtype_out == GPV_BF16) hipLaunchKernelGGL((image_to_nhwc4_kernel<bf16>), dim3(grid1d(n, 256)), dim3(256), 0, ST(stream), img, (bf16*)out, B, H, W, pad, Hp, Wp);
  else hipLaunchKernelGGL((image_to_nhwc4_kernel<float>), dim3(grid1d(n, 256)), dim3(256), 0, ST(stream), img, (float*)out, B, H, W, pad, Hp, Wp);
  GPV_CHECK_LAUNCH();
  return 0;
}

extern "C" int gpv_maxpool3x3s2(const void* x, void* y, int B, int H, int W, int C, int OH, int OW, int dtype, void* stream) {
  if (C % 8) return (int)hipErrorInvalidValue;
  const int64_t n = (int64_t)B * OH * OW * (C / 8);
  if (dtype == GPV_BF16) hipLaunchKernelGGL((maxpool_kernel<bf16>), dim3(grid1d(n, 256)), dim3(256), 0, ST(stream), (const bf16*)x, (bf16*)y, B, H, W, C, OH, OW);
  else hipLaunchKernelGGL((maxpool_kernel<float>), dim3(grid1d(n, 256)), dim3(256), 0, ST(stream), (const float*)x, (float*)y, B, H, W, C, OH, OW);
  GPV_CHECK_LAUNCH();
  return 0;
}

extern "C" int gpv_roi_weights(const float* boxes, void* wgt, int n_roi, int H, int W, int64_t ldw, int dtype, void* stream) {
  if (H > 64 || W > 64 || ldw < (int64_t)H * W) return (int)hipErrorInvalidValue;
  if (dtype == GPV_BF16) hipLaunchKernelGGL((roi_weights_kernel<bf16>), dim3(n_roi), dim3(64), 0, ST(stream), boxes, (bf16*)wgt, n_roi, H, W, ldw);
  else hipLaunchKernelGGL((roi_weights_kernel<float>), dim3(n_roi), dim3(64), 0, ST(stream), boxes, (float*)wgt, n_roi, H, W, ldw);
  GPV_CHECK_LAUNCH();
  return 0;
}

extern "C" int gpv_add(const void* a, const void* b, void* y, int64_t n, int dtype, void* stream) {
  if (n % 8) return (int)hipErrorInvalidValue;
  if (dtype == GPV_BF16) hipLaunchKernelGGL((add_kernel<bf16>), dim3(grid1d(n / 8, 256)), dim3(256), 0, ST(stream), (const bf16*)a, (const bf16*)b, (bf16*)y, n / 8, n / 8);
  else hipLaunchKernelGGL((add_kernel<float>), dim3(grid1d(n / 8, 256)), dim3(256), 0, ST(stream), (const float*)a, (const float*)b, (float*)y, n / 8, n / 8);
  GPV_CHECK_LAUNCH();
  return 0;
}

extern "C" int gpv_add_rowbcast(const void* a, const void* b, void* y, int64_t rows_total, int64_t rows_b, int cols, int dtype,
                                void* stream) {
  if (cols % 8) return (int)hipErrorInvalidValue;
  const int64_t n8 = rows_total * cols / 8, nb8 = rows_b * cols / 8;
  if (dtype == GPV_BF16) hipLaunchKernelGGL((add_kernel<bf16>), dim3(grid1d(n8, 256)), dim3(256), 0, ST(stream), (const bf16*)a, (const bf16*)b, (bf16*)y, n8, nb8);
  else hipLaunchKernelGGL((add_kernel<float>), dim3(grid1d(n8, 256)), dim3(256), 0, ST(stream), (const float*)a, (const float*)b, (float*)y, n8, nb8);
  GPV_CHECK_LAUNCH();
  return 0;
}

extern "C" int gpv_colsum(const void* x, float* out, int rows, int cols, int64_t ld, int dtype, void* stream) {
  if (cols % 8 == 0 && ld % 8 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
    const int c8 = cols / 8;
    int ct = 256;
    while (ct > 1 && ct / 2 >= c8) ct /= 2;                       // column threads per block: the power of two that covers cols / 8 (<= 256)
    const int gx = (c8 + ct - 1) / ct, rl = 256 / ct;
    int gy = (768 + gx - 1) / gx;                                  // ~768 blocks, at least rl rows each
    if (gy > (rows + rl - 1) / rl) gy = (rows + rl - 1) / rl;
    if ((int64_t)gy * rl * cols > (1 << 17)) gy = (int)((1 << 17) / ((int64_t)rl * cols));      // <= 128 K atomics (~1.5 us of them)
    if (gy < 1) gy = 1;
    const int rpb = (rows + gy - 1) / gy;
    dim3 grid(gx, (rows + rpb - 1) / rpb);
    if (dtype == GPV_BF16) hipLaunchKernelGGL((colsum8_kernel<bf16>), grid, dim3(256), 0, ST(stream), (const bf16*)x, out, rows, cols, ld, ct, rpb);
    else hipLaunchKernelGGL((colsum8_kernel<float>), grid, dim3(256), 0, ST(stream), (const float*)x, out, rows, cols, ld, ct, rpb);
    GPV_CHECK_LAUNCH();
    return 0;
  }
  int rpb = (rows + 255) / 256;
  if (rpb < 32) rpb = 32;
  dim3 grid((cols + 255) / 256, (rows + rpb - 1) / rpb);
  if (dtype == GPV_BF16) hipLaunchKernelGGL((colsum_kernel<bf16>), grid, dim3(256), 0, ST(stream), (const bf16*)x, out, rows, cols, ld, rpb);
  else hipLaunchKernelGGL((colsum_kernel<float>), grid, dim3(256), 0, ST(stream), (const float*)x, out, rows, cols, ld, rpb);
  GPV_CHECK_LAUNCH();
  return 0;
}

extern "C" int gpv_cast(const void* src, void* dst, int64_t n, int ds, int dd, void* stream) {
  dim3 g(grid1d(n, 256)), b(256);
  if (ds == GPV_F32 && dd == GPV_BF16) hipLaunchKernelGGL((cast_kernel<float, bf16>), g, b, 0, ST(stream), (const float*)src, (bf16*)dst, n);
  else if (ds == GPV_BF16 && dd == GPV_F32) hipLaunchKernelGGL((cast_kernel<bf16, float>), g, b, 0, ST(stream), (const bf16*)src, (float*)dst, n);
  else if (ds == GPV_F32 && dd == GPV_F32) hipLaunchKernelGGL((cast_kernel<float, float>), g, b, 0, ST(stream), (const float*)src, (float*)dst, n);
  else hipLaunchKernelGGL((cast_kernel<bf16, bf16>), g, b, 0, ST(stream), (const bf16*)src, (bf16*)dst, n);
  GPV_CHECK_LAUNCH();
  return 0;
}

extern "C" int gpv_cast_rowscale_t(const float* src, const float* scale, void* dst, void* dstT, int rows, int cols, int dtype_dst,
                                   void* stream) {
  dim3 grid((cols + 31) / 32, (rows + 31) / 32);
  if (dtype_dst == GPV_BF16) hipLaunchKernelGGL((cast_rowscale_t_kernel<bf16>), grid, dim3(256), 0, ST(stream), src, scale, (bf16*)dst, (bf16*)dstT, rows, cols);
  else hipLaunchKernelGGL((cast_rowscale_t_kernel<float>), grid, dim3(256), 0, ST(stream), src, scale, (float*)dst, (float*)dstT, rows, cols);
  GPV_CHECK_LAUNCH();
  return 0;
}

extern "C" int gpv_cast_transpose_group(const gpv_tc_problem* problems, int n, int dtype_dst, void* stream) {
  if (n < 0 || (n > 0 && !problems)) return (int)hipErrorInvalidValue;
  for (int i0 = 0; i0 < n; i0 += GPV_TC_GROUP_MAX) {
    TcGroup g;
    g.n = n - i0 < GPV_TC_GROUP_MAX ? n - i0 : GPV_TC_GROUP_MAX;
    int tiles = 0;
    for (int i = 0; i < g.n; ++i) {
      const gpv_tc_problem& q = problems[i0 + i];
      if (!q.src || !q.dstT || q.rows <= 0 || q.cols <= 0) return (int)hipErrorInvalidValue;
      g.prob[i] = q;
      g.tile_start[i] = tiles;
      tiles += ((q.rows + 63) / 64) * ((q.cols + 63) / 64);
    }
    g.tile_start[g.n] = tiles;
    if (dtype_dst == GPV_BF16) hipLaunchKernelGGL((cast_transpose_group_kernel<bf16>), dim3(tiles), dim3(256), 0, ST(stream), g);
    else hipLaunchKernelGGL((cast_transpose_group_kernel<float>), dim3(tiles), dim3(256), 0, ST(stream), g);
    GPV_CHECK_LAUNCH();
  }
  return 0;
}

extern "C" int gpv_prep_conv_weight(const float* src, const float* scale, void* wf, void* wd, int Cout, int T, int Cin, int dtype_dst,
                                    void* stream) {
  dim3 grid((Cin + 31) / 32, (Cout + 31) / 32, T);
  if (dtype_dst == GPV_BF16) hipLaunchKernelGGL((prep_conv_w_kernel<bf16>), grid, dim3(256), 0, ST(stream), src, scale, (bf16*)wf, (bf16*)wd, Cout, T, Cin);
  else hipLaunchKernelGGL((prep_conv_w_kernel<float>), grid, dim3(256), 0, ST(stream), src, scale, (float*)wf, (float*)wd, Cout, T, Cin);
  GPV_CHECK_LAUNCH();
  return 0;
}

extern "C" int gpv_embedding(const void* table, const int64_t* ids, void* out, int64_t n_ids, int dim, int dt, int dout, void* stream) {
  dim3 g(grid1d(n_ids * dim, 256)), b(256);
  if (dt == GPV_F32 && dout == GPV_BF16) hipLaunchKernelGGL((embedding_kernel<float, bf16>), g, b, 0, ST(stream), (const float*)table, ids, (bf16*)out, n_ids, dim);
  else if (dt == GPV_F32 && dout == GPV_F32) hipLaunchKernelGGL((embedding_kernel<float, float>), g, b, 0, ST(stream), (const float*)table, ids, (float*)out, n_ids, dim);
  else if (dt == GPV_BF16 && dout == GPV_BF16) hipLaunchKernelGGL((embedding_kernel<bf16, bf16>), g, b, 0, ST(stream), (const bf16*)table, ids, (bf16*)out, n_ids, dim);
  else return (int)hipErrorInvalidValue;
  GPV_CHECK_LAUNCH();
  return 0;
}

namespace gpvk { const uint64_t* g_seed_dev = nullptr; }
extern "C" int gpv_set_seed_device(const uint64_t* epoch) {
  gpvk::g_seed_dev = epoch;
  return 0;
}

extern "C" int gpv_dropout(const void* x, void* y, int64_t n, float p, uint64_t seed, int dtype, void* stream) {
  const uint32_t th = drop_thresh(p);
  const float sc = 1.f / (1.f - p);
  dim3 g(grid1d(n, 256)), b(256);
  if (dtype == GPV_BF16) hipLaunchKernelGGL((dropout_kernel<bf16>), g, b, 0, ST(stream), (const bf16*)x, (bf16*)y, n, th, sc, seed, gpvk::g_seed_dev);
  else hipLaunchKernelGGL((dropout_kernel<float>), g, b, 0, ST(stream), (const float*)x, (float*)y, n, th, sc, seed, gpvk::g_seed_dev);
  GPV_CHECK_LAUNCH();
  return 0;
}

extern "C" int gpv_relevance_condition(const void* x, const float* logits, const float* tokens, void* y, int rows, int dim, int dtype,
                                       void* stream) {
  dim3 g(grid1d((int64_t)rows * dim, 256)), b(256);
  if (dtype == GPV_BF16) hipLaunchKernelGGL((relevance_condition_kernel<bf16>), g, b, 0, ST(stream), (const bf16*)x, logits, tokens, (bf16*)y, rows, dim);
  else hipLaunchKernelGGL((relevance_condition_kernel<float>), g, b, 0, ST(stream), (const float*)x, logits, tokens, (float*)y, rows, dim);
  GPV_CHECK_LAUNCH();
  return 0;
}

extern "C" int gpv_act_fwd(const void* x, void* y, int64_t n, int act, int dtype, void* stream) {
  if (n % 8 || (act != GPV_ACT_RELU && act != GPV_ACT_GELU)) return (int)hipErrorInvalidValue;
  dim3 g(grid1d(n / 8, 256)), b(256);
  if (dtype == GPV_BF16) hipLaunchKernelGGL((act_fwd_kernel<bf16>), g, b, 0, ST(stream), (const bf16*)x, (bf16*)y, n / 8, act);
  else hipLaunchKernelGGL((act_fwd_kernel<float>), g, b, 0, ST(stream), (const float*)x, (float*)y, n / 8, act);
  GPV_CHECK_LAUNCH();
  return 0;
}

extern "C" int gpv_act_bwd(const void* dy, const void* ref, void* dx, int64_t n, int act, float alpha, int dtype, void* stream) {
  if (n % 8 || (act != GPV_ACT_RELU && act != GPV_ACT_GELU)) return (int)hipErrorInvalidValue;
  dim3 g(grid1d(n / 8, 256)), b(256);
  if (dtype == GPV_BF16) hipLaunchKernelGGL((act_bwd_kernel<bf16>), g, b, 0, ST(stream), (const bf16*)dy, (const bf16*)ref, (bf16*)dx, n / 8, act, alpha);
  else hipLaunchKernelGGL((act_bwd_kernel<float>), g, b, 0, ST(stream), (const float*)dy, (const float*)ref, (float*)dx, n / 8, act, alpha);
  GPV_CHECK_LAUNCH();
  return 0;
}

extern "C" int gpv_adamw(float* p, const float* g, float* m, float* v, void* p_lowp, int64_t n, float lr, float beta1, float beta2,
                         float eps, float wd, float bc1, float bc2, const float* gscale, const uint16_t* seg_id,
                         const int32_t* seg_live, void* stream) {
  if ((seg_id == nullptr) != (seg_live == nullptr)) return (int)hipErrorInvalidValue;
  static const int vec = tune_env("GPV_ADAMW_VEC", 1);
  const auto al = [](const void* q, uintptr_t a) { return (reinterpret_cast<uintptr_t>(q) & (a - 1)) == 0; };
  if (vec && n % 8 == 0 && al(p, 16) && al(g, 16) && al(m, 16) && al(v, 16) && (!p_lowp || al(p_lowp, 8))) {
    if (vec != 2) hipLaunchKernelGGL(adamw_vec4_kernel<false>, dim3(grid1d(n / 4, 256)), dim3(256), 0, ST(stream), p, g, m, v, (bf16*)p_lowp, n / 4, lr, beta1, beta2, eps, wd, bc1, bc2, gscale, seg_id, seg_live);
    else hipLaunchKernelGGL(adamw_vec4_kernel<true>, dim3(grid1d(n / 4, 256)), dim3(256), 0, ST(stream), p, g, m, v, (bf16*)p_lowp, n / 4, lr, beta1, beta2, eps, wd, bc1, bc2, gscale, seg_id, seg_live);
    GPV_CHECK_LAUNCH();
    return 0;
  }
  hipLaunchKernelGGL(adamw_kernel, dim3(grid1d(n, 256)), dim3(256), 0, ST(stream), p, g, m, v, (bf16*)p_lowp, n, lr, beta1, beta2, eps, wd, bc1, bc2, gscale, seg_id, seg_live);
  GPV_CHECK_LAUNCH();
  return 0;
}

extern "C" int gpv_clip_scale(const float* g, int64_t n, float max_norm, float* partial, float* gscale, int32_t* pstep,
                              const int32_t* live, int nparam, void* stream) {
  if (g && (n % 4 != 0 || (reinterpret_cast<uintptr_t>(g) & 15) != 0 || !partial || !gscale)) return (int)hipErrorInvalidValue;
  if ((pstep == nullptr) != (live == nullptr) || (!g && !pstep)) return (int)hipErrorInvalidValue;
  if (g) {
    hipLaunchKernelGGL(clip_partial_kernel, dim3(CLIP_BLOCKS), dim3(256), 0, ST(stream), g, n / 4, partial);
    GPV_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(clip_final_kernel, dim3(1), dim3(256), 0, ST(stream), g ? partial : nullptr, max_norm, gscale, pstep, live, nparam);
  GPV_CHECK_LAUNCH();
  return 0;
}

extern "C" int gpv_sumsq(const float* x, int64_t n, float* out, void* stream) {
  hipLaunchKernelGGL(sumsq_kernel, dim3(grid1d(n, 1024)), dim3(256), 0, ST(stream), x, n, out);
  GPV_CHECK_LAUNCH();
  return 0;
}
