// Linear (width 256 -> 256) + bias + dropout + residual + LayerNorm (+ position sum) in ONE launch (gfx950):
//
//   s  = a W^T + b                                   (the attention sublayer's out-projection, rounded to bf16 as the GEMM stores it)
//   y  = LayerNorm(x + dropout(s)) * gamma + beta    (ref: exp/gpv/models/transformer.py:156-157, 218-219, 224-226 -- norm1 / norm2 of the
//   y2 = y + pos[row % pos_rows]                      post-norm layers; y2 as gpv_layernorm_pos_fwd)
//
// The two launches it replaces (out-projection GEMM 9600 x 256 x 256: 9.4 us, LayerNorm: 6.6 us inside the step's graph) are both
// latency-shaped, and the 256-wide row is whole in ONE MFMA tile row: the wave that multiplied 16 rows holds them complete -- lane
// (row, g) has 64 of the row's 256 values, the other three quarters sit in the lanes row + 16, + 32, + 48 -- so the LayerNorm is two
// xor-shuffles away from the accumulators.  Structure of conv1x1_stream.hip: the weight matrix resident in LDS (staged once per
// workgroup, output channels permuted so that a lane's values are runs of 8 consecutive channels = 16-byte accesses), a wave per 16-row
// tile, A fragments straight from global memory.  s is written too: the backward (gpv_layernorm_bwd3, the projection's weight gradient
// and backward-data GEMM) is unchanged and reads it.  Same arithmetic as ln_fwd_kernel on the same rounded s (same dropout words: the
// (seed, flat index) pairs of common.h); the row sums are taken in another order (64 values per lane instead of 8), so mean / rstd may
// differ from the two launches in the last bit.
#include "common.h"
#include "../../include/gpv_hip.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct LinLnK {
  const bf16* a; const bf16* w; const float* bias; const bf16* x; const float* gamma; const float* beta;
  bf16* s; bf16* y; bf16* y2; const bf16* pos; float* mean; float* rstd;
  int M, pos_rows; float eps; uint32_t dthresh; float dscale; uint64_t seed; const uint64_t* seed_dev;
};

constexpr int LK = 256, LN_ = 256, LKC = LK / 32, LNT = LN_ / 16, LNG = LN_ / 32, LROW = LK * 2;     // LDS row = 512 bytes, no padding: XOR swizzle
typedef __attribute__((address_space(3))) void lds_void_t;

// LDS row L holds output channel chan(L): MFMA tile j = L / 16, row r = L % 16 -> channel 32 (j / 2) + 8 (r / 4) + 4 (j & 1) + (r % 4)
// (conv1x1_stream.hip's c1s_chan): tiles 2 t, 2 t + 1 leave lane (row, g) with the 8 consecutive channels 32 t + 8 g .. + 7
__device__ __forceinline__ int lin_chan(int L) {
  const int j = L >> 4, r = L & 15;
  return (j >> 1) * 32 + (r >> 2) * 8 + (j & 1) * 4 + (r & 3);
}

template <bool FULLM, bool DROP, bool POS>
__device__ __forceinline__ void linear_ln_tiles(const LinLnK& p, unsigned char* smem_raw, int tile, int ntile, int nw, bf16x8 (&an)[LKC]) {
  // STRAIGHT-LINE tile body: whole 16-row tiles only (FULLM: no predicate on loads or stores), dropout / position output as template
  // tags, gamma / beta / bias from LDS.  The first build had `if (row < M)`, `if (gamma)`, `if (y2)`, `if (dthresh)` around its loads
  // and stores: hipcc drains the memory queue (s_waitcnt vmcnt(0)) at every such join -- a store's write acknowledgement, then the
  // next load's round trip, sixteen times per tile: ONE tile took 17 K cycles (7 us, timed with s_memtime) for 2 K cycles of MFMA.
  const int lane = threadIdx.x & 63, g = lane >> 4, pl = lane & 15;
  const float* bias_l = reinterpret_cast<const float*>(smem_raw + (size_t)LN_ * LROW);
  const float* gamma_l = bias_l + LN_;
  const float* beta_l = gamma_l + LN_;
  const uint32_t t16 = p.dthresh >> 16;
  const uint32_t hseed = (uint32_t)p.seed + (uint32_t)(p.seed >> 32) * 0x85EBCA6Bu;     // drop_pair_bits with pair index < 2^32 (host-checked)
  for (; tile < ntile; tile += nw) {
    bf16x8 af[LKC];
#pragma unroll
    for (int kc = 0; kc < LKC; ++kc) af[kc] = an[kc];
    {
      const int rown = min((tile + nw) * 16 + pl, p.M - 1);
#pragma unroll
      for (int kc = 0; kc < LKC; ++kc) an[kc] = *reinterpret_cast<const bf16x8*>(p.a + (int64_t)rown * LK + kc * 32 + g * 8);
    }
    const int row = tile * 16 + pl;
    const bool rok = FULLM || row < p.M;
    const int rowc = FULLM ? row : min(row, p.M - 1);
    // the residual row and the position row (this lane's 8 runs of 8 channels each) are requested before the MFMAs
    bf16x8 xv[LNG], pv[POS ? LNG : 1];
#pragma unroll
    for (int t = 0; t < LNG; ++t) xv[t] = *reinterpret_cast<const bf16x8*>(p.x + (int64_t)rowc * LN_ + t * 32 + g * 8);
    if constexpr (POS) {
      const int prow = rowc % p.pos_rows;
#pragma unroll
      for (int t = 0; t < LNG; ++t) pv[t] = *reinterpret_cast<const bf16x8*>(p.pos + (int64_t)prow * LN_ + t * 32 + g * 8);
    }
    f32x4 acc[LNT];
#pragma unroll
    for (int j = 0; j < LNT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    int woff = pl * LROW;                              // bytes; row L = 16 j + pl, chunk 4 kc + g sits in slot (4 kc + g) ^ pl = 4 kc ^ (g ^ pl)
    asm volatile("" : "+v"(woff));                     // (keeps the fragment reads inside the tile loop: see conv1x1_stream.hip)
    const unsigned char* wbase = smem_raw + woff;
    const int tx = (g ^ pl) * 16;
#pragma unroll
    for (int kc = 0; kc < LKC; ++kc) {
      const unsigned char* wk = wbase + ((kc * 64) ^ tx);
#pragma unroll
      for (int j = 0; j < LNT; ++j) {
        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wk + j * 16 * LROW);
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af[kc], acc[j], 0, 0, 0);
      }
    }
    // s (rounded, stored), v = x + dropout(s), row sum
    float v[LNG][8];
    float sum = 0.f;
    uint32_t hbase = 0u;
    if constexpr (DROP) hbase = (uint32_t)(((uint64_t)row * LN_ + (uint64_t)(g * 8)) >> 1) * 0x9E3779B9u + hseed;
#pragma unroll
    for (int t = 0; t < LNG; ++t) {
      const int c0 = t * 32 + g * 8;
      const float4 b0 = *reinterpret_cast<const float4*>(bias_l + c0), b1 = *reinterpret_cast<const float4*>(bias_l + c0 + 4);
      const float sv[8] = {acc[2 * t][0] + b0.x, acc[2 * t][1] + b0.y, acc[2 * t][2] + b0.z, acc[2 * t][3] + b0.w,
                           acc[2 * t + 1][0] + b1.x, acc[2 * t + 1][1] + b1.y, acc[2 * t + 1][2] + b1.z, acc[2 * t + 1][3] + b1.w};
      bf16x8 sb;
#pragma unroll
      for (int e = 0; e < 8; ++e) sb[e] = (bf16)sv[e];
      if (rok) *reinterpret_cast<bf16x8*>(p.s + (int64_t)row * LN_ + c0) = sb;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float q0 = (float)sb[2 * q], q1 = (float)sb[2 * q + 1];
        if constexpr (DROP) {                          // the keep words of common.h's drop_pair_bits (pair = flat index / 2), element kept iff its 16 bits >= t16
          uint32_t r = hbase + (uint32_t)(t * 16 + q) * 0x9E3779B9u;
          r ^= r >> 16; r = __umul24(r, 0x85EBCBu);
          r ^= r >> 13; r = __umul24(r, 0xC2B2AFu);
          r ^= r >> 16;
          q0 = (r & 0xffffu) >= t16 ? q0 * p.dscale : 0.f;
          q1 = (r >> 16) >= t16 ? q1 * p.dscale : 0.f;
        }
        v[t][2 * q] = (float)xv[t][2 * q] + q0;
        v[t][2 * q + 1] = (float)xv[t][2 * q + 1] + q1;
        sum += v[t][2 * q];
        sum += v[t][2 * q + 1];
      }
    }
    sum += __shfl_xor(sum, 16); sum += __shfl_xor(sum, 32);
    const float mu = sum / LN_;
    float var = 0.f;
#pragma unroll
    for (int t = 0; t < LNG; ++t)
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[t][e] - mu; var += d * d; }
    var += __shfl_xor(var, 16); var += __shfl_xor(var, 32);
    var /= LN_;
    const float rs = rsqrtf(var + p.eps);
    if (g == 0 && rok) { p.mean[row] = mu; p.rstd[row] = rs; }
#pragma unroll
    for (int t = 0; t < LNG; ++t) {
      const int c0 = t * 32 + g * 8;
      float o[8], gm[8], bt[8];
      Ld8<float>::ld(gamma_l + c0, gm); Ld8<float>::ld(beta_l + c0, bt);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (v[t][e] - mu) * rs * gm[e] + bt[e];
      bf16x8 ob;
#pragma unroll
      for (int e = 0; e < 8; ++e) ob[e] = (bf16)o[e];
      if (rok) *reinterpret_cast<bf16x8*>(p.y + (int64_t)row * LN_ + c0) = ob;
      if constexpr (POS) {
        bf16x8 o2;
#pragma unroll
        for (int e = 0; e < 8; ++e) o2[e] = (bf16)((float)pv[t][e] + (float)ob[e]);
        if (rok) *reinterpret_cast<bf16x8*>(p.y2 + (int64_t)row * LN_ + c0) = o2;
      }
    }
  }
}

__global__ __launch_bounds__(256) void linear_ln_kernel(LinLnK p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* bias_l = reinterpret_cast<float*>(smem_raw + (size_t)LN_ * LROW);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, pl = lane & 15;
  if (p.dthresh) p.seed = eff_seed(p.seed, p.seed_dev);
  const int ntile = (p.M + 15) >> 4;
  const int nw = (int)gridDim.x * 4;
  const int tile = (int)blockIdx.x + wave * (int)gridDim.x;     // waves of a block take tiles a grid apart: every block has work for its first waves

  bf16x8 an[LKC];
  {
    const int row = min(tile * 16 + pl, p.M - 1);
#pragma unroll
    for (int kc = 0; kc < LKC; ++kc) an[kc] = *reinterpret_cast<const bf16x8*>(p.a + (int64_t)row * LK + kc * 32 + g * 8);
  }
  // The weight matrix goes L2 -> LDS by DMA (buffer_load ... lds, no staging registers: all 32 KB of a wave's share in flight at
  // once; through registers the 128 KB took four dependent round trips of eight loads per thread).  The DMA writes lane-linear
  // images -- instruction i of the block = LDS rows 2 i, 2 i + 1, lane = (row parity, 16-byte slot) -- so the bank-conflict-free
  // layout is made on the SOURCE side: slot s of LDS row L receives chunk s ^ (L & 15) of output channel chan(L), and the fragment
  // read of chunk c of row L looks at slot c ^ (L & 15) (ds_read_b128 lane groups of MI355X_MICROARCH.md checked).  4 K cycles.
  {
    constexpr int OOB = 0x7ffffff0;
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.w), (short)0, OOB, 0x00020000);
    const int wv = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
    for (int i = 0; i < 32; ++i) {                       // 128 instructions of 1 KB, 32 per wave
      const int inst = wv * 32 + i;
      const int L = inst * 2 + (lane >> 5), sl = lane & 31;
      const int voff = (lin_chan(L) * LK + ((sl ^ (L & 15)) * 8)) * 2;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_void_t*)(smem_raw + inst * 1024), 16, voff, 0, 0, 0);
    }
  }
  bias_l[tid] = p.bias ? p.bias[tid] : 0.f;
  bias_l[LN_ + tid] = p.gamma ? p.gamma[tid] : 1.f;       // (no affine: gamma 1, beta 0 -- n * 1 + 0 is n exactly)
  bias_l[2 * LN_ + tid] = p.beta ? p.beta[tid] : 0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const bool fullm = (p.M & 15) == 0;
#define GO(F, D, P) linear_ln_tiles<F, D, P>(p, smem_raw, tile, ntile, nw, an)
  if (fullm) {
    if (p.dthresh) { if (p.y2) GO(true, true, true); else GO(true, true, false); }
    else { if (p.y2) GO(true, false, true); else GO(true, false, false); }
  } else {
    if (p.dthresh) { if (p.y2) GO(false, true, true); else GO(false, true, false); }
    else { if (p.y2) GO(false, false, true); else GO(false, false, false); }
  }
#undef GO
}

inline bool al16l(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

}  // namespace

extern "C" int gpv_linear_layernorm_fwd(const void* a, const void* w, const float* bias, const void* x, const float* gamma,
                                        const float* beta, void* s, void* y, float* mean, float* rstd, int rows, int K, int N, float eps,
                                        float drop_p, uint64_t seed, const void* pos, int pos_rows, void* y2, void* stream) {
  if (!a || !w || !x || !s || !y || !mean || !rstd || rows <= 0 || rows > (1 << 23)) return (int)hipErrorInvalidValue;
  if (K != LK || N != LN_) return (int)hipErrorInvalidValue;                      // the DETR width only
  if ((gamma == nullptr) != (beta == nullptr) || (pos == nullptr) != (y2 == nullptr) || (pos && pos_rows <= 0)) return (int)hipErrorInvalidValue;
  if (!al16l(a) || !al16l(w) || !al16l(x) || !al16l(s) || !al16l(y) || (pos && (!al16l(pos) || !al16l(y2))) || (gamma && (!al16l(gamma) || !al16l(beta))) ||
      (bias && !al16l(bias)))
    return (int)hipErrorInvalidValue;
  LinLnK p{};
  p.a = reinterpret_cast<const bf16*>(a); p.w = reinterpret_cast<const bf16*>(w); p.bias = bias; p.x = reinterpret_cast<const bf16*>(x);
  p.gamma = gamma; p.beta = beta; p.s = reinterpret_cast<bf16*>(s); p.y = reinterpret_cast<bf16*>(y); p.y2 = reinterpret_cast<bf16*>(y2);
  p.pos = reinterpret_cast<const bf16*>(pos); p.mean = mean; p.rstd = rstd; p.M = rows; p.pos_rows = pos ? pos_rows : 1; p.eps = eps;
  p.dthresh = drop_p > 0.f ? drop_thresh(drop_p) : 0u;
  p.dscale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  p.seed = seed; p.seed_dev = gpvk::g_seed_dev;
  const size_t lds = (size_t)LN_ * LROW + 3 * LN_ * sizeof(float);
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(linear_ln_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }
    attr = true;
  }
  const int ntile = (rows + 15) / 16;
  const int blocks = ntile < 256 ? ntile : 256;
  hipLaunchKernelGGL(linear_ln_kernel, dim3(blocks), dim3(256), lds, reinterpret_cast<hipStream_t>(stream), p);
  GPV_CHECK_LAUNCH();
  return 0;
}
