// Shared device helpers for the gfx950 kernels (wave = 64 lanes, MFMA 16x16x32 bf16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

typedef __bf16 bf16;
typedef bf16 __attribute__((ext_vector_type(8))) bf16x8;
typedef bf16 __attribute__((ext_vector_type(4))) bf16x4;
typedef bf16 __attribute__((ext_vector_type(2))) bf16x2;
typedef float __attribute__((ext_vector_type(4))) f32x4;

// Environment knobs of the kernels' launch heuristics (kernel family on / off, blocks per CU, rows per strip, forced tiles, split targets,
// timing ablations) exist for tools/ only: they are read iff the library was compiled with -DGPV_TUNING (`make tuning` ->
// libgpv_hip_tuning.so, loaded by gpv1_amd.hip when GPV_TUNING_LIB=1).  The production library (libgpv_hip.so) reads NO environment:
// every knob is its default, and the run-time switches tests need go through gpv_set_option ("no global state").
#include <stdlib.h>
#ifdef GPV_TUNING
static inline int tune_env(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
#else
static inline int tune_env(const char*, int dflt) { return dflt; }
#endif

#define GPV_CHECK_LAUNCH() \
  do {                     \
    hipError_t e_ = hipGetLastError(); \
    if (e_ != hipSuccess) return (int)e_; \
  } while (0)

__device__ __forceinline__ float bf2f(bf16 x) { return (float)x; }
__device__ __forceinline__ bf16 f2bf(float x) { return (bf16)x; }

// split an fp32 value into hi + lo bf16 (hi = RNE(x), lo = RNE(x - hi)): |x - hi - lo| <~ 2^-17 |x|
__device__ __forceinline__ void split_bf16(float x, bf16& hi, bf16& lo) {
  hi = (bf16)x;
  lo = (bf16)(x - (float)hi);
}

// D(16x16) += A(16x32, row = lane&15, k = 8*(lane>>4)..+7) * B(32x16, col = lane&15, same k)
// result: lane holds D[row = 4*(lane>>4)+i][col = lane&15], i = 0..3
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// counter-based RNG for dropout: one 32-bit hash per element index.  32-bit arithmetic only: the previous fmix64
// (three 64x64 multiplies = 12 quarter-rate v_mul_lo/hi_u32 per element) made every dropout consumer VALU-bound --
// the encoder attention kernels spent 54 % of their wave cycles issuing ~3700 VALU instructions per wave for 20 MFMAs
// (PMC, tools/pmc_attn.py), the FFN GEMM epilogue hashes 19.6 M elements per layer.  This is the 2-multiply
// "lowbias32" finalizer on (index low word ^ seed), the high words folded in with one more multiply.
__device__ __forceinline__ uint32_t hash_u32(uint64_t seed, uint64_t idx) {
  uint32_t x = (uint32_t)idx ^ (uint32_t)seed;
  x ^= ((uint32_t)(idx >> 32) ^ (uint32_t)(seed >> 32)) * 0x9E3779B9u;
  x ^= x >> 16; x *= 0x7feb352du;
  x ^= x >> 15; x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
// Dropout keep decisions come in PAIRS of consecutive flat element indices: one 32-bit mix per pair, 16 bits per element
// (drop probability = floor(p * 65536) / 65536), and only one quarter-rate 32-bit multiply per pair -- the rest is shifts, xors
// and full-rate 24-bit multiplies.  (Round 1 hashed the 64-bit index of every element with three 32-bit multiplies: the
// LayerNorm kernels and the K = 256 FFN GEMM epilogues spent more on the mask than on their arithmetic.)  Every consumer of an
// (seed, flat index) mask -- GEMM epilogues, LayerNorm forward / backward, gpv_dropout -- uses these, so a mask drawn by one
// kernel is the mask another one reconstructs.
__device__ __forceinline__ uint32_t drop_pair_bits(uint64_t seed, uint64_t pair) {
  uint32_t x = (uint32_t)pair * 0x9E3779B9u + (uint32_t)seed;
  x += ((uint32_t)(pair >> 32) ^ (uint32_t)(seed >> 32)) * 0x85EBCA6Bu;
  x ^= x >> 16; x = __umul24(x, 0x85EBCBu);
  x ^= x >> 13; x = __umul24(x, 0xC2B2AFu);
  x ^= x >> 16;
  return x;
}
// element kept iff its 16 bits >= p * 2^16
__device__ __forceinline__ bool drop_keep(uint64_t seed, uint64_t idx, uint32_t thresh) {
  const uint32_t w = drop_pair_bits(seed, idx >> 1);
  return ((idx & 1) ? (w >> 16) : (w & 0xffffu)) >= (thresh >> 16);
}
// keep bits of the N (4 or 8) consecutive elements base .. base+N-1 (bit e = element base + e); N/2 mixes when base is even
template <int N>
__device__ __forceinline__ uint32_t drop_mask(uint64_t seed, uint64_t base, uint32_t thresh) {
  const uint32_t t16 = thresh >> 16;
  uint32_t m = 0u;
  if ((base & 1) == 0) {
#pragma unroll
    for (int q = 0; q < N / 2; ++q) {
      const uint32_t w = drop_pair_bits(seed, (base >> 1) + q);
      m |= ((w & 0xffffu) >= t16 ? 1u : 0u) << (2 * q);
      m |= ((w >> 16) >= t16 ? 1u : 0u) << (2 * q + 1);
    }
  } else {
#pragma unroll
    for (int e = 0; e < N; ++e) m |= (drop_keep(seed, base + e, thresh) ? 1u : 0u) << e;
  }
  return m;
}
// Device-resident seed epoch (gpv_set_seed_device): when a caller replays captured launches (hipGraph) the `seed` argument of a
// launch is frozen in the graph; the effective seed of every dropout consumer is then seed ^ mix(*epoch), the caller bumps the
// word once per step, and the forward / backward launches of one step see the same value.  NULL = the seed argument as given.
namespace gpvk { extern const uint64_t* g_seed_dev; }
__device__ __forceinline__ uint64_t eff_seed(uint64_t seed, const uint64_t* epoch) {
  return epoch ? seed ^ (*epoch * 0x9E3779B97F4A7C15ull) : seed;
}
__host__ __device__ __forceinline__ uint32_t drop_thresh(float p) {
  double t = (double)p * 4294967296.0;
  if (t < 0) t = 0;
  if (t > 4294967295.0) t = 4294967295.0;
  return (uint32_t)t;
}

// Weights resident in LDS, staged once per workgroup (the streaming kernels): U 16-byte loads of every thread are issued before the
// first LDS store.  The plain `for (idx ...) lds[dst(idx)] = glob[src(idx)]` loop compiles to load -> s_waitcnt vmcnt(0) -> ds_write
// per iteration: 16 DEPENDENT L2 round trips for a 128 KB matrix on 512 threads -- 10-20 us at the head of every launch of
// conv1x1_stream / _chain / _dual, conv3x3_stream and stem_pool (seen in the ISA; the A-fragment prefetch shares the counter).
// Round 5: the loads are UNCONDITIONAL (a thread beyond the end re-reads the last chunk); only the LDS stores are predicated.  With a
// `if (idx < total)` around each load hipcc branched around every load and put `s_waitcnt vmcnt(0)` in front of the next one (seen in
// the ISA of every user): the "eight loads in flight" were eight -- sixteen for a 128 KB matrix -- dependent round trips, 16 K cycles
// (7 us, timed with s_memtime) at the head of every streaming launch.
template <int NTHR, int U, typename SrcF, typename DstF>
__device__ __forceinline__ void stage_chunks16(int total, int tid, SrcF src, DstF dst) {
  for (int base = 0; base < total; base += NTHR * U) {
    bf16x8 tmp[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = base + u * NTHR + tid;
      tmp[u] = *reinterpret_cast<const bf16x8*>(src(idx < total ? idx : total - 1));
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = base + u * NTHR + tid;
      if (idx < total) *reinterpret_cast<bf16x8*>(dst(idx)) = tmp[u];
    }
  }
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

template <typename T> struct Ld8;   // load 8 consecutive elements as float[8]
template <> struct Ld8<bf16> {
  static __device__ __forceinline__ void ld(const bf16* p, float* o) {
    bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (float)v[i];
  }
  static __device__ __forceinline__ void st(bf16* p, const float* o) {
    bf16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (bf16)o[i];
    *reinterpret_cast<bf16x8*>(p) = v;
  }
};
template <> struct Ld8<float> {
  static __device__ __forceinline__ void ld(const float* p, float* o) {
    float4 a = *reinterpret_cast<const float4*>(p);
    float4 b = *reinterpret_cast<const float4*>(p + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
  }
  static __device__ __forceinline__ void st(float* p, const float* o) {
    *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(o[4], o[5], o[6], o[7]);
  }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
