// Fused ResNet stem (gfx950): 7x7 / stride 2 convolution (3 -> 64 channels) + FrozenBatchNorm shift + ReLU + 3x3 / stride 2
// max-pool in ONE kernel  (exp/gpv/models/backbone.py:93-95 -> torchvision resnet50 conv1 / bn1 / relu / maxpool).
//
// Before: gpv_conv2d on the generic 128x64-tile kernel (290 us at B = 32: K = 7 x 32 is seven 32-deep k-tiles with a barrier
// each) wrote the 314 MB conv map, gpv_maxpool3x3s2 read it back (83 us).  Here the conv map never exists:
//   * input: the zero-padded NHWC4 bf16 image gpv_image_to_nhwc4 writes (8 B per pixel: one tap ROW of the 7x7 = 8 pixels x 4
//     channels = 64 contiguous bytes, the 8th pixel and 4th channel meet zero weights) -- K = 7 rows x 32;
//   * v_mfma_f32_32x32x16_bf16, weights (64 x 224, 28 KB, BN scale folded in) resident in LDS as the row operand, the output
//     channels permuted as in conv3x3_stream.hip so that a lane ends with runs of 8 consecutive channels;
//   * a WAVE owns 32 conv columns (30 new ones + 2 of overlap = 15 pooled columns) and walks DOWN the conv rows of its row
//     segment: conv row j needs padded input rows 2j .. 2j+6, kept in an 8-slot register ring (2 x 16 B per lane and row: lane
//     (column c, half h) holds pixels 2c + 4kc + 2h, +1) -- every conv row loads only its two new input rows;
//   * epilogue of a conv row, all in registers: + shift -> ReLU -> bf16 -> zero the conv positions outside the map (the pool pads
//     with -inf; after the ReLU 0 is as good) -> horizontal 3-max with DPP wave shifts (non-negative bf16 order like their
//     bit patterns: v_pk_max_u16) -> vertical 3-max against the two previous rows -> odd conv rows emit a pooled row from the
//     odd lanes (pooled column = lane / 2), 16-byte stores.
// Same bf16 rounding points as the two-kernel path (conv output rounded to bf16, then the max): results differ from it by the
// fp32 summation order of the 147 products only.
#include "gemm_common.h"

namespace gpvk {
namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

constexpr int ST_OOB = 0x7ffffff0;
constexpr int ST_K = 224, ST_KP = ST_K + 8;
constexpr int ST_PCOLS = 15;            // pooled columns per wave strip

__device__ __forceinline__ int st_perm(int m) { return (m & 0x13) | ((m & 4) << 1) | ((m & 8) >> 1); }
__device__ __forceinline__ unsigned pk_max_u16(unsigned a, unsigned b) {
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ unsigned dpp_shl(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true); }   // lane i <- lane i + 1
__device__ __forceinline__ unsigned dpp_shr(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true); }   // lane i <- lane i - 1

struct StemK {
  const void* x; const void* w; const float* shift; void* y;
  int B, Hp, Wp, CH, CW, PH, PW;       // padded input rows / row pitch in pixels, conv map, pooled map
  int prows, nstrip, nseg, nitems;     // pooled rows per item, strips per row, segments per column, items in all
};

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void stem_pool_kernel(StemK p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16* Wl = reinterpret_cast<bf16*>(smem_raw);
  float* bias_l = reinterpret_cast<float*>(Wl + 64 * ST_KP);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5, pl = lane & 31;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), (short)0, ST_OOB, 0x00020000);
  bf16* Y = reinterpret_cast<bf16*>(p.y);

  int i_hi, i_step, item;
  {
    const int nb = gridDim.x;
    if ((nb & 7) == 0) {
      const int per = (p.nitems + 7) >> 3, xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, nbx = nb >> 3;
      i_hi = min(p.nitems, (xcd + 1) * per);
      i_step = nbx * WAVES;
      item = xcd * per + lb * WAVES + wave;
    } else {
      i_hi = p.nitems; i_step = nb * WAVES;
      item = blockIdx.x * WAVES + wave;
    }
  }
  {
    const bf16* Wg = reinterpret_cast<const bf16*>(p.w);
    constexpr int SL = ST_K / 8;
    stage_chunks16<WAVES * 64, 8>(64 * SL, tid,
        [&](int idx) { const int L = idx / SL, sl = idx - L * SL; return Wg + ((L & ~31) + st_perm(L & 31)) * ST_K + sl * 8; },
        [&](int idx) { const int L = idx / SL, sl = idx - L * SL; return Wl + L * ST_KP + sl * 8; });
    for (int L = tid; L < 64; L += WAVES * 64) bias_l[L] = p.shift ? p.shift[(L & ~31) + st_perm(L & 31)] : 0.f;
  }
  __syncthreads();

  const bf16* wlane = Wl + pl * ST_KP + h * 8;
  u32x4 ring[8][2];
  for (; item < i_hi; item += i_step) {
    const int seg = item % p.nseg, t1 = item / p.nseg, strip = t1 % p.nstrip, b = t1 / p.nstrip;
    const int p0 = seg * p.prows;                        // first pooled row of the item
    const int np = min(p.prows, p.PH - p0);
    const int c0 = 2 * p0 - 1;                           // first conv row (may be -1: outside the map, zeroed)
    const int nrows = 2 * np + 1;
    const int cc = strip * 2 * ST_PCOLS - 1 + pl;        // this lane's conv column (may be -1 / >= CW: zeroed)
    const bool colok = (unsigned)cc < (unsigned)p.CW;
    // byte offset of padded pixel (b, row 0, 2 cc + 2 h); conv row c reads padded rows 2 c + r
    const int colo = ((b * p.Hp) * p.Wp + 2 * cc + 2 * h) * 8;
    const int rowb = p.Wp * 8;
    auto load_row = [&](int prow, int slot) {            // padded input row prow -> ring[slot]
      const int vo = (colok && (unsigned)prow < (unsigned)p.Hp) ? colo + prow * rowb : ST_OOB;
      ring[slot][0] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, 0, 0);
      ring[slot][1] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, 32, 0);
    };
    const int pr0 = 2 * c0;                              // padded row of (conv row c0, tap row 0); negative rows are out of range
#pragma unroll
    for (int r = 0; r < 7; ++r) load_row(pr0 + r, r);
    unsigned prev[16], cur[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) prev[i] = cur[i] = 0u;
    const bool pooled_lane = (pl & 1) && pl < 2 * ST_PCOLS && strip * ST_PCOLS + (pl >> 1) < p.PW;
    for (int j0 = 0; j0 < nrows; j0 += 4) {
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int j = j0 + jj;
        if (j < nrows) {                                 // (uniform)
          const int c = c0 + j;
          int wvo = 0;
          asm volatile("" : "+v"(wvo));                  // keeps the weight-fragment reads inside the row loop (see conv3x3_stream.hip)
          const bf16* wl = wlane + wvo;
          f32x16 acc[2];
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float4 bv = *reinterpret_cast<const float4*>(bias_l + nt * 32 + q * 8 + h * 4);
              acc[nt][4 * q] = bv.x; acc[nt][4 * q + 1] = bv.y; acc[nt][4 * q + 2] = bv.z; acc[nt][4 * q + 3] = bv.w;
            }
          }
          load_row(pr0 + 2 * j + 7, (2 * jj + 7) & 7);   // first new row of the NEXT conv row: its slot has been free since row j - 1
#pragma unroll
          for (int r = 0; r < 7; ++r) {
            const int slot = (2 * jj + r) & 7;
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
#pragma unroll
              for (int nt = 0; nt < 2; ++nt) {
                const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wl + nt * 32 * ST_KP + r * 32 + kc * 16);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, __builtin_bit_cast(bf16x8, ring[slot][kc]), acc[nt], 0, 0, 0);
              }
            }
            if (r == 0) load_row(pr0 + 2 * j + 8, (2 * jj) & 7);   // second new row: into the slot tap row 0 just left
          }
          // ReLU -> bf16 pairs -> zero outside the conv map -> horizontal 3-max
          const bool live = colok && (unsigned)c < (unsigned)p.CH;
          unsigned hm[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int nt = i >> 3, e = (i & 7) * 2;
            bf16x2 pr;
            pr[0] = (bf16)fmaxf(acc[nt][e], 0.f);
            pr[1] = (bf16)fmaxf(acc[nt][e + 1], 0.f);
            const unsigned v = live ? __builtin_bit_cast(unsigned, pr) : 0u;
            hm[i] = pk_max_u16(v, pk_max_u16(dpp_shl(v), dpp_shr(v)));
          }
          if ((jj & 1) == 0) {                           // odd conv row 2 ph + 1 (c0 is odd): closes pooled row ph, opens ph + 1
            if (j > 0 && pooled_lane) {
              const int ph = p0 + (j >> 1) - 1;
              bf16* dst = Y + ((int64_t)(b * p.PH + ph) * p.PW + strip * ST_PCOLS + (pl >> 1)) * 64 + h * 8;
#pragma unroll
              for (int cch = 0; cch < 4; ++cch) {        // chunk = 8 channels at (cch >> 1) * 32 + (cch & 1) * 16 + 8 h
                u32x4 o;
#pragma unroll
                for (int d = 0; d < 4; ++d) o[d] = pk_max_u16(cur[cch * 4 + d], hm[cch * 4 + d]);
                *reinterpret_cast<u32x4*>(dst + (cch >> 1) * 32 + (cch & 1) * 16) = o;
              }
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) prev[i] = hm[i];
          } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) cur[i] = pk_max_u16(prev[i], hm[i]);
          }
        }
      }
    }
  }
}

}  // namespace
}  // namespace gpvk

// y[B, PH, PW, 64] = maxpool3x3s2p1( relu( conv7x7s2(x) + shift ) ), x = padded NHWC4 bf16 image [B, Hp, Wp, 4] (gpv_image_to_nhwc4,
// pad 3), w = [64][7][8 px][4 ch] bf16 with the FrozenBN scale folded in.  bf16 only.
extern "C" int gpv_stem_pool(const void* x, const void* w, const float* shift, void* y, int B, int Hp, int Wp, int CH, int CW, int PH,
                             int PW, void* stream) {
  using namespace gpvk;
  if (!x || !w || !y || B <= 0) return (int)hipErrorInvalidValue;
  if (PH != (CH + 2 - 3) / 2 + 1 || PW != (CW + 2 - 3) / 2 + 1) return (int)hipErrorInvalidValue;
  if (2 * (CH - 1) + 7 > Hp || 2 * (CW - 1) + 8 > Wp || Wp % 2) return (int)hipErrorInvalidValue;
  if ((int64_t)B * Hp * Wp * 8 >= (int64_t)ST_OOB) return (int)hipErrorInvalidValue;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(y)) & 15) return (int)hipErrorInvalidValue;
  constexpr int WAVES = 8;
  StemK p{};
  p.x = x; p.w = w; p.shift = shift; p.y = y;
  p.B = B; p.Hp = Hp; p.Wp = Wp; p.CH = CH; p.CW = CW; p.PH = PH; p.PW = PW;
  p.nstrip = (PW + ST_PCOLS - 1) / ST_PCOLS;
  static const int env_rows = tune_env("GPV_STEM_ROWS", 0);
  static const int env_blocks = tune_env("GPV_STEM_BLOCKS", 0);
  int blocks = env_blocks > 0 ? env_blocks : 512;
  // pooled rows per item: the choice that minimises (rounds of the resident waves) x (conv rows per item, 2 rows + 1)
  int best = 1;
  {
    const int64_t waves = (int64_t)blocks * WAVES;
    int64_t best_cost = -1;
    for (int r = 1; r <= PH && r <= 32; ++r) {
      const int64_t items = (int64_t)B * p.nstrip * ((PH + r - 1) / r);
      const int64_t cost = ((items + waves - 1) / waves) * (2 * r + 1);
      if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = r; }
    }
  }
  p.prows = env_rows > 0 ? env_rows : best;
  p.nseg = (PH + p.prows - 1) / p.prows;
  p.nitems = B * p.nstrip * p.nseg;
  while (blocks > 8 && (int64_t)(blocks - 8) * WAVES >= p.nitems) blocks -= 8;
  const size_t lds = (size_t)64 * ST_KP * 2 + 64 * sizeof(float);
  hipLaunchKernelGGL(stem_pool_kernel<WAVES>, dim3(blocks), dim3(WAVES * 64), lds, reinterpret_cast<hipStream_t>(stream), p);
  GPV_CHECK_LAUNCH();
  return 0;
}
