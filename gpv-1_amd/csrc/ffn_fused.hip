// The post-norm feed-forward sub-layer of a DETR encoder / decoder layer as ONE launch (gfx950):
//
//     h   = dropout( relu( x W1^T + b1 ) )                    [M, F]      (stored: the backward's ReLU mask and dW2 operand)
//     y   = h W2^T + b2                                       [M, 256]    (stored, bf16: the LayerNorm backward recomputes x + drop(y))
//     out = LayerNorm( x + dropout(y) ) ; out2 = out + pos    [M, 256]
//
// (exp/gpv/models/transformer.py:156-160 encoder, :226-231 decoder: linear2(dropout(activation(linear1(src)))), src + dropout2(.),
// norm2.)  As three launches -- K = 256 GEMM with a 39 MB output, K = 2048 GEMM on 150 tiles, LayerNorm -- the [M, 2048] hidden
// activation is written by the first and read back by the second (79 us per encoder layer at M = 9600).  Here a workgroup owns
// 64 rows of x for the whole sub-layer and walks the hidden dimension in chunks of 64 features:
//
//   * x never enters LDS: a wave keeps its 32 rows as the 16 B-operand fragments of the first product in registers for the whole
//     kernel (64 VGPRs);
//   * the first product is computed TRANSPOSED, C1[f][m] = W1c[f][:] . x[m][:]  -- the accumulator tiles 2u, 2u + 1 then leave lane
//     (m, g) with 8 hidden features of its row m, and because W1's rows are staged PERMUTED (LDS row 16 j + 4 a + b of a 32-row group
//     holds feature 8 a + 4 j + b: conv1x1_stream.hip's c1s_chan) they are the 8 CONSECUTIVE features 8 g .. 8 g + 7: after bias /
//     ReLU / dropout / rounding that is one 16-byte store of h AND, unchanged, the B fragment (k = 8 g .. + 7) of the second
//     product's MFMA -- the hidden activation goes from accumulator to operand without touching LDS (conv1x1_chain.hip's trick);
//   * the weights stream L2 -> LDS with buffer_load ... lds (no staging registers), two stages of 64 KB: chunk t + 1 is in flight
//     while chunk t is multiplied; W2's rows (= output columns) are staged permuted as well, so the epilogue's bias / residual /
//     y / out accesses are 16 bytes in the accumulator layout (the register epilogue of gemm_glds.hip);
//   * waves = (row half) x (feature half of the chunk): a wave multiplies its 32 rows by ITS 32 features in both products
//     (64 MFMAs and 32 ds_read_b128 per chunk: half the LDS read bandwidth), the two partial sums over the feature halves meet once,
//     at the end, through LDS; the LayerNorm runs on the finished rows in registers.
//
// Per workgroup 2 MB of weights pass through one CU; 150 workgroups at M = 9600.  Not bit-identical to the three launches (the
// K = 2048 sum is split in two halves), same rounding points: h and y are rounded to bf16 exactly where the launches round them,
// dropout masks are the same (seed, flat index) masks.
//
// MEASURED (one MI355X, tools-free timing loop, dropout 0.1): M = 9600: 71 us against 71 us for the three launches; M = 3200: 66 against
// 42.  It does not pay, and ops.FFNBlockFn keeps the three launches unless GPV_FFN_FUSED=1.  Where the time goes (ablations, r04):
// 16 us of prologue + epilogue per workgroup, then 1.5 us per 64-feature chunk where the MFMAs need 0.43: with 400 VGPRs there is ONE wave
// per SIMD, and inside a chunk the first product, its bias / ReLU / dropout / rounding (VALU, ~250 instructions of 4 cycles) and the second
// product depend on each other in that order -- nothing overlaps; removing the fragment reads, the barrier or the h stores changes
// nothing (1.24 - 1.34 us), removing the weight loads 0.38 us: the 64 KB in flight per CU arrive at 18 B/clk.  And overlapping the phases would not be enough: the 2 MB of
// weights a workgroup streams through its CU at those 18 B/clk are 48 us by themselves -- rows stationary with all of F per workgroup is
// bound by per-CU weight delivery; the three launches win by moving the [M, 2048] activation through HBM with all CUs at once.
#include "gemm_common.h"

namespace gpvk {
namespace {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct FfnK {
  const void* x; const void* w1; const float* b1; const void* w2; const float* b2; const float* gamma; const float* beta;
  void* h; void* y; void* out; float* mean; float* rstd; const void* pos; void* out2;
  int M, F, pos_rows;
  float eps;
  uint32_t dthresh; float dscale; uint64_t seed1, seed2; const uint64_t* seed_dev;
};

constexpr int D = 256, BM = 64, FC = 64, MAXF = 8192;
constexpr int W1C_BYTES = FC * D * 2, W2C_BYTES = D * FC * 2, STAGE = W1C_BYTES + W2C_BYTES;      // 32 KB + 32 KB

__device__ __forceinline__ int perm32(int r) {      // LDS row r of a 32-row group -> the source row it holds
  return (r & ~31) + ((r & 15) >> 2) * 8 + ((r >> 4) & 1) * 4 + (r & 3);
}

// drop_mask<8> (common.h) for an EVEN flat index below 2^33, from the running word xb = (index / 2) * 0x9E3779B9 + seed terms: the four
// pair hashes of drop_pair_bits without its two 32-bit multiplies (quarter rate: 16 cycles each on a wave that has nothing to hide
// them behind) -- consecutive pairs differ by the constant, consecutive chunks by 32 of them.
__device__ __forceinline__ uint32_t drop_mask8_x(uint32_t xb, uint32_t t16) {
  uint32_t m = 0u;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    uint32_t x = xb + (uint32_t)q * 0x9E3779B9u;
    x ^= x >> 16; x = __umul24(x, 0x85EBCBu);
    x ^= x >> 13; x = __umul24(x, 0xC2B2AFu);
    x ^= x >> 16;
    m |= ((x & 0xffffu) >= t16 ? 1u : 0u) << (2 * q);
    m |= ((x >> 16) >= t16 ? 1u : 0u) << (2 * q + 1);
  }
  return m;
}

__global__ __launch_bounds__(256) void ffn_fwd_kernel(FfnK p) {
  extern __shared__ __attribute__((aligned(128))) unsigned char smem[];
  // (b1 behind the weight stages.  Its reads in the loop are bf16x8-typed like the fragment reads: in front of a float-typed LDS read the
  //  compiler puts s_waitcnt vmcnt(0) -- every LDS-DMA load in flight, it cannot tell the destinations apart -- and chunk t + 1's loads
  //  and chunk t's products serialise.)
  float* b1s = reinterpret_cast<float*>(smem + 2 * STAGE);                  // [F]
  float* red = reinterpret_cast<float*>(smem);                              // end of kernel: partial sums of the upper feature halves
  if (p.dthresh) { p.seed1 = eff_seed(p.seed1, p.seed_dev); p.seed2 = eff_seed(p.seed2, p.seed_dev); }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mh = wave & 1, fh = wave >> 1;
  const int ml = lane & 15, g = lane >> 4;
  const int row0 = (int)blockIdx.x * BM + mh * 32;
  const bf16* X = reinterpret_cast<const bf16*>(p.x);
  const int F = p.F, nchunk = F / FC;

  // ---- weight loaders: 1 KB per wave instruction = 8 LDS rows x 128 B, 16-byte chunks XOR-swizzled with the row (gemm_glds.hip) ----
  constexpr int OOB = 0x7ffffff0;
  const auto rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w1), (short)0, OOB, 0x00020000);
  const auto rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w2), (short)0, OOB, 0x00020000);
  const int lrow = lane >> 3, lchunk = (lane & 7) ^ lrow;
  // W1 chunk: four k-panels (64 of the 256 input features each) of [64 rows][128 B]; wave w stages panel w.  Row L <- feature perm32(L)
  // W2 chunk: [256 rows = output columns][128 B = the chunk's 64 features]; wave w stages rows 64 w .. + 63.  Row L <- column perm32(L)
  int v1[8], v2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    v1[j] = (perm32(j * 8 + lrow) * D + wave * 64 + lchunk * 8) * 2;
    v2[j] = (perm32(wave * 64 + j * 8 + lrow) * F + lchunk * 8) * 2;
  }
  auto issue = [&](int t, int stage) {
    unsigned char* s1 = smem + stage * STAGE + wave * 8192;
    unsigned char* s2 = smem + stage * STAGE + W1C_BYTES + wave * 8192;
    const int so1 = t * (FC * D * 2), so2 = t * (FC * 2);
#pragma unroll
    for (int j = 0; j < 8; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (lds_void_t*)(s1 + j * 1024), 16, v1[j], so1, 0, 0);
#pragma unroll
    for (int j = 0; j < 8; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs2, (lds_void_t*)(s2 + j * 1024), 16, v2[j], so2, 0, 0);
  };
  issue(0, 0);
  for (int i = tid; i < F; i += 256) b1s[i] = p.b1[i];
  b1s[F + tid] = p.b2[tid]; b1s[F + D + tid] = p.gamma[tid]; b1s[F + 2 * D + tid] = p.beta[tid];      // (256 threads = 256 columns)

  // ---- this wave's 32 rows of x as MFMA fragments: xf[j][ks] = x[row0 + 16 j + ml][32 ks + 8 g .. + 7] ----
  bf16x8 xf[2][8];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int m = min(row0 + j * 16 + ml, p.M - 1);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) xf[j][ks] = *reinterpret_cast<const bf16x8*>(X + (int64_t)m * D + ks * 32 + g * 8);
  }
  f32x4 acc[16][2];
#pragma unroll
  for (int dt = 0; dt < 16; ++dt) { acc[dt][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[dt][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  const int fsw = lane & 7;
  int a1_off = (fh * 32 + ml) * 128;                       // + T * 2048 + panel * 8192 + slot * 16
  int a2_off = W1C_BYTES + ml * 128 + (((fh * 4 + g) ^ fsw) << 4);      // + dt * 2048
  int b_off = 2 * STAGE + (fh * 32 + g * 8) * 4;                        // + t * FC * 4
  // h goes out through a buffer descriptor: rows beyond M get an out-of-range offset (the hardware drops the store), so that EVERY
  // iteration issues exactly two vector-memory stores per wave.  vmcnt counts stores as well as loads (in issue order on gfx9): the
  // top-of-loop wait is vmcnt(2) -- everything but this wave's two newest operations, i.e. the next chunk's 16 LDS-DMA loads have
  // landed while the previous iteration's two h stores may still be on their way (waiting for them too cost 1.4 us per chunk).
  const auto rsH = __builtin_amdgcn_make_buffer_rsrc(p.h, (short)0, OOB, 0x00020000);
  int hvo[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int m = row0 + j * 16 + ml;
    hvo[j] = m < p.M ? (m * F + fh * 32 + g * 8) * 2 : OOB;
  }

  uint32_t xd[2] = {0u, 0u};                               // dropout: pair counter of (row, first feature of this lane) x constant + seed terms
  const uint32_t t16 = p.dthresh >> 16;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const uint64_t pair = ((uint64_t)(row0 + j * 16 + ml) * (uint64_t)F + fh * 32 + g * 8) >> 1;
    xd[j] = (uint32_t)pair * 0x9E3779B9u + (uint32_t)p.seed1 + ((uint32_t)(pair >> 32) ^ (uint32_t)(p.seed1 >> 32)) * 0x85EBCA6Bu;
  }

  for (int t = 0; t < nchunk; ++t) {
    if (t == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    __syncthreads();
    if (t + 1 < nchunk) issue(t + 1, (t + 1) & 1);
    const unsigned char* st = smem + (t & 1) * STAGE;
    asm volatile("" : "+v"(a1_off), "+v"(a2_off), "+v"(b_off));         // (keep the fragment reads inside the loop: no hoisting into 100s of VGPRs)
    // ---- first product, transposed: c1[T][j] = W1c[f-tile T of this wave's half] . x[m-tile j] over the 256 inputs ----
    // One wave per SIMD: nothing but this wave's own instruction order hides an LDS round trip.  All 16 fragment reads of a product go out
    // back to back (they return in order, the MFMAs follow them one lgkmcnt step behind), the second product's 16 before the epilogue
    // arithmetic of the first; sched_barriers keep the compiler from pairing every read with its MFMAs again (2 reads, wait, 4 MFMAs: 1.9 us
    // per chunk for 0.43 us of MFMA).
    bf16x8 wf[16];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int slot = (((ks & 1) * 4 + g) ^ fsw) << 4;
      const unsigned char* pa = st + (ks >> 1) * 8192 + a1_off + slot;
      wf[2 * ks] = *reinterpret_cast<const bf16x8*>(pa);
      wf[2 * ks + 1] = *reinterpret_cast<const bf16x8*>(pa + 2048);
    }
    const f32x4 ba = __builtin_bit_cast(f32x4, *reinterpret_cast<const bf16x8*>(smem + b_off + t * (FC * 4)));       // (bf16x8-typed: see b1s)
    const f32x4 bb = __builtin_bit_cast(f32x4, *reinterpret_cast<const bf16x8*>(smem + b_off + t * (FC * 4) + 16));
    __builtin_amdgcn_sched_barrier(0);
    f32x4 c1[2][2];
#pragma unroll
    for (int T = 0; T < 2; ++T) { c1[T][0] = f32x4{0.f, 0.f, 0.f, 0.f}; c1[T][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      c1[0][0] = mfma16(wf[2 * ks], xf[0][ks], c1[0][0]);
      c1[0][1] = mfma16(wf[2 * ks], xf[1][ks], c1[0][1]);
      c1[1][0] = mfma16(wf[2 * ks + 1], xf[0][ks], c1[1][0]);
      c1[1][1] = mfma16(wf[2 * ks + 1], xf[1][ks], c1[1][1]);
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned char* pb = st + a2_off;
#pragma unroll
    for (int dt = 0; dt < 16; ++dt) wf[dt] = *reinterpret_cast<const bf16x8*>(pb + dt * 2048);
    __builtin_amdgcn_sched_barrier(0);
    // ---- bias, ReLU, dropout, round: lane (m, g) holds features f0 .. f0 + 7 of rows m (j = 0, 1) ----
    const float bq[8] = {ba[0], ba[1], ba[2], ba[3], bb[0], bb[1], bb[2], bb[3]};
    bf16x8 hf[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const uint32_t keep8 = p.dthresh ? drop_mask8_x(xd[j], t16) : 0xffu;
      xd[j] += (uint32_t)(FC / 2) * 0x9E3779B9u;
      const float v[8] = {c1[0][j][0], c1[0][j][1], c1[0][j][2], c1[0][j][3], c1[1][j][0], c1[1][j][1], c1[1][j][2], c1[1][j][3]};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float u = fmaxf(v[e] + bq[e], 0.f);
        if (p.dthresh) u = ((keep8 >> e) & 1u) ? u * p.dscale : 0.f;
        hf[j][e] = (bf16)u;
      }
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hf[j]), rsH, hvo[j], t * (FC * 2), 0);
    }
    // ---- second product on the rounded accumulators: acc[dt][j] += W2c[column tile dt][this wave's 32 features] . h ----
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int dt = 0; dt < 16; ++dt) {
      acc[dt][0] = mfma16(wf[dt], hf[0], acc[dt][0]);
      acc[dt][1] = mfma16(wf[dt], hf[1], acc[dt][1]);
    }
  }

  // ---- the two feature halves meet: the upper half's waves park their partial sums, the lower half's finish the rows ----
  const bf16* P = reinterpret_cast<const bf16*>(p.pos);
  bf16x8 pf[2][8];                                            // pos rows of the second output: on their way during the exchange
  if (p.out2 && fh == 0) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int m = min(row0 + j * 16 + ml, p.M - 1);
#pragma unroll
      for (int u = 0; u < 8; ++u) pf[j][u] = *reinterpret_cast<const bf16x8*>(P + (int64_t)(m % p.pos_rows) * D + u * 32 + g * 8);
    }
  }
  __syncthreads();                                            // every wave is done with the weight stages
  float4* rp = reinterpret_cast<float4*>(red) + (size_t)mh * (32 * 64) + lane;       // [mh][dt * 2 + j][lane]
  if (fh == 1) {
#pragma unroll
    for (int dt = 0; dt < 16; ++dt)
#pragma unroll
      for (int j = 0; j < 2; ++j) rp[(dt * 2 + j) * 64] = make_float4(acc[dt][j][0], acc[dt][j][1], acc[dt][j][2], acc[dt][j][3]);
  }
  __syncthreads();
  if (fh == 1) return;
#pragma unroll
  for (int dt = 0; dt < 16; ++dt)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float4 q = rp[(dt * 2 + j) * 64];
      acc[dt][j][0] += q.x; acc[dt][j][1] += q.y; acc[dt][j][2] += q.z; acc[dt][j][3] += q.w;
    }
  // lane (m, g), tile pair u: columns 32 u + 8 g .. + 7 of row m.  z = x + dropout(bf16(y)) stays in the accumulator registers.
  bf16* Y = reinterpret_cast<bf16*>(p.y);
  bf16* O = reinterpret_cast<bf16*>(p.out);
  bf16* O2 = reinterpret_cast<bf16*>(p.out2);
  const unsigned char* prm = smem + 2 * STAGE + F * 4;       // b2 | gamma | beta, fp32 [256] each (bf16x8-typed reads: see b1s)
  auto ld8f = [&](int off, float* o) {
    const f32x4 a = __builtin_bit_cast(f32x4, *reinterpret_cast<const bf16x8*>(prm + off));
    const f32x4 b = __builtin_bit_cast(f32x4, *reinterpret_cast<const bf16x8*>(prm + off + 16));
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
  };
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int m = row0 + j * 16 + ml;
    const bool ok = m < p.M;
    float sum = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int c = u * 32 + g * 8;
      float bq[8], xv[8];
      ld8f(c * 4, bq);
#pragma unroll
      for (int e = 0; e < 8; ++e) xv[e] = (float)xf[j][u][e];          // the residual is the first product's operand: already here
      const uint32_t keep8 = p.dthresh ? drop_mask<8>(p.seed2, (uint64_t)m * D + c, p.dthresh) : 0xffu;
      bf16x8 yb;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float yv = acc[2 * u + (e >> 2)][j][e & 3] + bq[e];
        yb[e] = (bf16)yv;
        float s = (float)yb[e];
        if (p.dthresh) s = ((keep8 >> e) & 1u) ? s * p.dscale : 0.f;
        const float z = xv[e] + s;
        acc[2 * u + (e >> 2)][j][e & 3] = z;
        sum += z;
      }
      if (ok) *reinterpret_cast<bf16x8*>(Y + (int64_t)m * D + c) = yb;
    }
    sum += __shfl_xor(sum, 16); sum += __shfl_xor(sum, 32);
    const float mu = sum * (1.f / D);
    float var = 0.f;
#pragma unroll
    for (int dt = 0; dt < 16; ++dt)
#pragma unroll
      for (int i = 0; i < 4; ++i) { const float d = acc[dt][j][i] - mu; var += d * d; }
    var += __shfl_xor(var, 16); var += __shfl_xor(var, 32);
    const float rs = rsqrtf(var * (1.f / D) + p.eps);
    if (ok && g == 0) { p.mean[m] = mu; p.rstd[m] = rs; }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int c = u * 32 + g * 8;
      float gm[8], bt[8], o[8];
      ld8f((D + c) * 4, gm); ld8f((2 * D + c) * 4, bt);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (acc[2 * u + (e >> 2)][j][e & 3] - mu) * rs * gm[e] + bt[e];
      if (ok) {
        Ld8<bf16>::st(O + (int64_t)m * D + c, o);
        if (O2) {
          float pv[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) pv[e] = (float)pf[j][u][e] + (float)(bf16)o[e];
          Ld8<bf16>::st(O2 + (int64_t)m * D + c, pv);
        }
      }
    }
  }
}

}  // namespace
}  // namespace gpvk

// out[M,256] = LayerNorm(x + dropout(y)), y = h W2^T + b2, h[M,F] = dropout(relu(x W1^T + b1)); bf16 x / w1[F,256] / w2[256,F] / h / y /
// out, fp32 biases, gamma, beta, mean[M], rstd[M].  Optional second output out2 = out + pos[row % pos_rows] (gpv_layernorm_pos_fwd).
// drop_p applies to both dropouts (transformer.py:137-139 builds them from one rate), masks = (seed1 | seed2, flat index) as in
// gpv_gemm's epilogue and gpv_layernorm_fwd.  Supported: model width 256, F a multiple of 64 (the DETR layers: 256 -> 2048 -> 256);
// hipErrorNotSupported otherwise (the caller launches gpv_gemm, gpv_gemm, gpv_layernorm_pos_fwd).
extern "C" int gpv_ffn_fused_fwd(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, const float* gamma,
                                 const float* beta, void* h, void* y, void* out, float* mean, float* rstd, int M, int Dm, int F, float eps,
                                 float drop_p, uint64_t seed1, uint64_t seed2, const void* pos, int pos_rows, void* out2, void* stream) {
  using namespace gpvk;
  if (!x || !w1 || !b1 || !w2 || !b2 || !gamma || !beta || !h || !y || !out || !mean || !rstd || M <= 0) return (int)hipErrorInvalidValue;
  if (Dm != D || F <= 0 || F % FC != 0 || F > MAXF) return (int)hipErrorNotSupported;
  if ((pos == nullptr) != (out2 == nullptr) || (pos && (pos_rows <= 0 || M % pos_rows != 0))) return (int)hipErrorInvalidValue;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w1) | reinterpret_cast<uintptr_t>(w2) | reinterpret_cast<uintptr_t>(h) |
       reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(out2) | reinterpret_cast<uintptr_t>(pos) |
       reinterpret_cast<uintptr_t>(b1) | reinterpret_cast<uintptr_t>(b2) | reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta)) & 15)
    return (int)hipErrorInvalidValue;
  if ((int64_t)F * D * 2 >= 0x7ffffff0 || (int64_t)M * F * 2 >= 0x7ffffff0) return (int)hipErrorNotSupported;
  FfnK p{};
  p.x = x; p.w1 = w1; p.b1 = b1; p.w2 = w2; p.b2 = b2; p.gamma = gamma; p.beta = beta; p.h = h; p.y = y; p.out = out; p.mean = mean; p.rstd = rstd;
  p.pos = pos; p.out2 = out2; p.M = M; p.F = F; p.pos_rows = pos ? pos_rows : 1; p.eps = eps;
  p.dthresh = drop_p > 0.f ? drop_thresh(drop_p) : 0u;
  p.dscale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  p.seed1 = seed1; p.seed2 = seed2; p.seed_dev = g_seed_dev;
  const size_t lds = 2 * (size_t)STAGE + (size_t)(F + 3 * D) * sizeof(float);
  static size_t attr = 0;
  if (lds > attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ffn_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }
    attr = lds;
  }
  hipLaunchKernelGGL(ffn_fwd_kernel, dim3((M + BM - 1) / BM), dim3(256), lds, reinterpret_cast<hipStream_t>(stream), p);
  GPV_CHECK_LAUNCH();
  return 0;
}
