// Direct-to-LDS convolution weight gradient for gfx950 (bf16 operands, fp32 partial products).
//
//   dW[Cout][tap*Cin + c] += sum over pixels k of  dY[k][Cout]  *  X[pixel k shifted by tap][c]
//   M = Cout, N = taps*Cin, reduction K = B*OH*OW output pixels; both operands are REDUCTION-major in memory
//   (a row of dY / of the NHWC input is one pixel), so a k-tile is [64 pixels][128 columns] = 256 B per pixel and operand.
//
// Same idea as gemm_glds.hip, for the operand form gemm.hip's register-staged TRANS x CONV kernel handles:
//   * 128 x 128 tiles, 4 waves (2x2, wave tile 64x64), k-tiles of 64 pixels, two LDS stages of 32 KB -> two blocks per CU;
//   * pixel rows go L2 -> LDS with global_load_lds_dwordx4 (a wave instruction = 4 pixel rows x 256 B, lane-linear in LDS):
//     no staging VGPRs, no zero-masking pass, no ds_write;
//   * MFMA operand fragments come out of the pixel-major LDS image with ds_read_b64_tr_b16 (one read = 4 consecutive pixels
//     of the lane's column).  A pixel row is exactly 64 banks wide, so the 8 rows a 32-lane phase touches (r, r+8; r = 0..3)
//     would all hit the same banks: the 32-byte column pairs of row r are stored at pair index P ^ q(r),
//     q(r) = (r & 3) | ((r >> 3) & 1) << 2, applied when the lane picks its SOURCE chunk and again in the fragment address;
//   * the gather arithmetic (pixel -> (image, oh, ow) -> input offset, bounds) is done ONCE per pixel row and k-tile: lane l
//     owns row l of the k-tile and advances its (oh, ow, image) state by 64 pixels without divisions; the four loads a lane
//     issues fetch their row's offset from the owning lane with a wave shuffle.  (The register-staged kernel recomputes it
//     for every 16-byte load: ~100 VALU instructions per 16 MFMAs and wave; here ~60 per 32.)
// The reduction is split over gridDim.y; every split writes its partial [M,N] product (alpha, per-row BN scale applied)
// to its slab of the caller's workspace, gemm.hip's splitk_reduce_kernel adds the slabs to dW (two-pass, no atomics).
#include "gemm_common.h"

namespace gpvk {
namespace {

constexpr int TBM = 128, TBN = 128, TBK = 64;
constexpr int ROWBYTES = 256;                      // one pixel row of a 128-column operand tile
constexpr int OP_BYTES = TBK * ROWBYTES;           // 16 KB
constexpr int TSTAGE = 2 * OP_BYTES;

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* p) {
  typedef short __attribute__((ext_vector_type(4))) s16x4;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * ROWBYTES));
  typedef short __attribute__((ext_vector_type(8))) s16x8;
  s16x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, r);
}

// ABL: timing-only ablations of the k-loop (results are wrong): 1 = no MFMAs, 2 = no fragment reads and no MFMAs (loads + barriers
// only), 3 = no global loads (fragment reads + MFMAs on whatever the LDS holds), 4 = as 2 but every iteration loads the FIRST k-tile
// (the loop's rate when every line hits the L2) -- tools/bench_wgrad_ablate.py
template <bool CONV, int ABL = 0>
__device__ __forceinline__ void glds_tt_core(const GemmK& p, int tile, int ksplit) {
  extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const ConvGeom& g = p.cg;
  const int tm = tile / p.tilesN, tn = tile - tm * p.tilesN;
  const int row0 = tm * TBM, col0 = tn * TBN;
  const int kt_total = (p.K + TBK - 1) / TBK;
  const int kt0 = ksplit * p.kt_per_split;
  const int kt1 = min(kt_total, kt0 + p.kt_per_split);
  if (kt0 >= kt1) return;                            // (the host sizes the split so that this does not happen)

  // CONV: this tile's tap and channel block (TBN divides Cin: checked on the host)
  int c0 = 0, dh = 0, dw = 0;
  if constexpr (CONV) {
    const int tap = col0 / g.Cin;
    c0 = col0 - tap * g.Cin;
    const int tap_r = tap / g.KW, tap_s = tap - tap_r * g.KW;
    dh = tap_r - g.PH; dw = tap_s - g.PW;
  }

  // ---- CONV pixel-row state: lane l owns reduction row l of every k-tile ----
  int px_b = 0, px_oh = 0, px_ow = 0, adv_q = 0, adv_r = 0;
  if constexpr (CONV) {
    const int k = kt0 * TBK + lane;
    px_b = k / (g.OH * g.OW);
    const int rem = k - px_b * (g.OH * g.OW);
    px_oh = rem / g.OW;
    px_ow = rem - px_oh * g.OW;
    adv_q = TBK / g.OW; adv_r = TBK - adv_q * g.OW;              // 64 pixels = adv_q rows + adv_r columns (adv_q + 2 <= OH: host)
  }

  // ---- loader: instruction j of this wave covers rows wave*16 + j*4 + (lane >> 4), slot lane & 15 ----
  // buffer_load_dwordx4 ... lds through one descriptor per operand: per-lane 32-bit byte offsets fixed for the whole split, the
  // k-tile offset in the scalar soffset, an out-of-range voffset (tap outside the input, row beyond the last pixel) = zeros.
  constexpr int OOB = 0x7ffffff0;                                // == num_records
  const int lrow = lane >> 4;
  const int chunk01 = (lane & 15) ^ (2 * lrow);                  // logical 16-B chunk for j = 0, 1; j = 2, 3: ^ 8
  const auto rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), (short)0, OOB, 0x00020000);
  const auto rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), (short)0, OOB, 0x00020000);
  int a_vo[4], b_vo[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = wave * 16 + j * 4 + lrow;
    const int ch = chunk01 ^ ((j >> 1) * 8);
    a_vo[j] = (r * (int)p.lda + row0 + ch * 8) * 2;
    b_vo[j] = CONV ? (c0 + ch * 8) * 2 : (r * (int)p.ldb + col0 + ch * 8) * 2;
  }
  auto bload = [&](const decltype(rsA)& rs, int voff, int soff, unsigned char* l) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)l, 16, voff, soff, 0, 0);
  };

  auto issue = [&](int kt, int stage) {
    unsigned char* sa = smem + stage * TSTAGE + wave * (4 * 1024);
    unsigned char* sb = sa + OP_BYTES;
    const bool full = (kt + 1) * TBK <= p.K;                     // uniform
    if constexpr (CONV) {
      // my row's gather offset (bytes into x), or OOB when the tap falls outside the input / beyond the last pixel
      int off;
      {
        const int ih = px_oh * g.SH + dh, iw = px_ow * g.SW + dw;
        const bool ok = (unsigned)ih < (unsigned)g.IH && (unsigned)iw < (unsigned)g.IW && kt * TBK + lane < p.K;
        off = ok ? ((px_b * g.IH + ih) * g.IW + iw) * g.Cs * 2 : OOB;
        px_ow += adv_r;
        const int c = px_ow >= g.OW ? 1 : 0;
        px_ow -= c ? g.OW : 0;
        px_oh += adv_q + c;
        const int c2 = px_oh >= g.OH ? 1 : 0;
        px_oh -= c2 ? g.OH : 0;
        px_b += c2;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)                                  // (OOB + a chunk offset is still out of range)
        bload(rsB, __shfl(off, wave * 16 + j * 4 + lrow) + b_vo[j], 0, sb + j * 1024);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool ok = full || kt * TBK + wave * 16 + j * 4 + lrow < p.K;
        bload(rsB, ok ? b_vo[j] : OOB, kt * TBK * (int)p.ldb * 2, sb + j * 1024);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool ok = full || kt * TBK + wave * 16 + j * 4 + lrow < p.K;
      bload(rsA, ok ? a_vo[j] : OOB, kt * TBK * (int)p.lda * 2, sa + j * 1024);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment addresses: lane (li = lane & 15, fg = lane >> 4) points at row 8*fg + (li >> 2), 8 bytes (li & 1) of logical
  // chunk 2*P + ((li & 3) >> 1), P = 16-column group; stored pair index = P ^ q, q = (li >> 2) | (fg & 1) << 2
  const int li = lane & 15, fg = lane >> 4;
  const int fq = (li >> 2) | ((fg & 1) << 2);
  const int f_row = (8 * fg + (li >> 2)) * ROWBYTES + ((li & 3) >> 1) * 16 + (li & 1) * 8;
  int a_sl[4], b_sl[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a_sl[i] = f_row + (((wm * 4 + i) ^ fq) << 5);
    b_sl[i] = OP_BYTES + f_row + (((wn * 4 + i) ^ fq) << 5);
  }

  // plain form, bias gradient: a_rowsum[m] += sum_k A[k][m].  The A fragments are already in registers: one more MFMA per
  // fragment against an all-ones operand gives every lane the column sum of ITS m (lane & 15), on the waves of the first
  // column tile only
  const bool do_sum = !CONV && p.a_rowsum != nullptr && tn == 0 && wn == 0;
  f32x4 sacc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) sacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;

  if (ABL == 5 || ABL == 6) {
    // loads-only with TWO (5) / THREE (6) k-tiles in flight per block (the LDS image is overwritten at will: nobody reads it):
    // what a deeper pipeline could deliver -- 1.65 ms either way against 1.87 ms with one (profiles/r04_wgrad_ablations.txt)
    constexpr int AHEAD = ABL == 5 ? 2 : 3;
    for (int a = 0; a < AHEAD && kt0 + a < kt1; ++a) issue(kt0 + a, a & 1);
    for (int kt = kt0; kt < kt1; ++kt) {
      if (ABL == 5) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      __syncthreads();
      if (kt + AHEAD < kt1) issue(kt + AHEAD, kt & 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  if (ABL != 3) issue(kt0, 0);
  for (int kt = kt0; kt < kt1; ++kt) {
    const int t = kt - kt0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (ABL != 3 && kt + 1 < kt1) issue(ABL == 4 ? kt0 : kt + 1, (t + 1) & 1);
    const unsigned char* st = smem + (t & 1) * TSTAGE;
    if (ABL == 2 || ABL == 4) continue;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = tr_frag(st + b_sl[j] + kk * 32 * ROWBYTES);
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = tr_frag(st + a_sl[i] + kk * 32 * ROWBYTES);
      if (ABL == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(af[i]), "v"(bfr[i]));
        continue;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(bfr[j], af[i], acc[i][j]);   // swapped: lane holds 4 consecutive columns
      if constexpr (!CONV) {
        if (do_sum) {
#pragma unroll
          for (int i = 0; i < 4; ++i) sacc[i] = mfma16(ones, af[i], sacc[i]);
        }
      }
    }
  }
  if constexpr (!CONV) {
    if (do_sum && fg == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) atomicAdd(p.a_rowsum + row0 + wm * 64 + i * 16 + li, sacc[i][0]);
    }
  }

  // ---------------- epilogue: fragments -> LDS (fp32, 64 rows at a time) -> whole rows of this split's slab ----------------
  // p.ws: this split's slab of the workspace (two-pass reduction); NULL (grouped launch, one block owns the whole reduction of
  // its tile): C += acc as a coalesced read-modify-write
  const bool direct = p.ws == nullptr;
  float* Cp = direct ? reinterpret_cast<float*>(p.C) : p.ws + (int64_t)ksplit * p.M * p.N;
  const int64_t cpitch = direct ? p.ldc : p.N;
  float* ep = reinterpret_cast<float*>(smem);
  constexpr int EPITCH = TBN + 4;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    __syncthreads();
    if (wm == half) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          *reinterpret_cast<f32x4*>(ep + (i * 16 + li) * EPITCH + wn * 64 + j * 16 + fg * 4) = acc[i][j];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; ++q) {                    // 64 rows x 32 quads / 256 threads
      const int idx = tid + q * 256;
      const int r = idx >> 5, c4 = idx & 31;
      const int m = row0 + half * 64 + r;
      if (m >= p.M) continue;
      const float rs = p.rowscale ? p.rowscale[m] * p.alpha : p.alpha;
      float4 v = *reinterpret_cast<const float4*>(ep + r * EPITCH + c4 * 4);
      v.x *= rs; v.y *= rs; v.z *= rs; v.w *= rs;
      float4* dst = reinterpret_cast<float4*>(Cp + (int64_t)m * cpitch + col0 + c4 * 4);
      if (direct) { const float4 c = *dst; v.x += c.x; v.y += c.y; v.z += c.z; v.w += c.w; }
      *dst = v;
    }
  }
}

// ================================================================================================================================
// Round 5: the same product on 256 x 256 tiles with an EIGHT-PHASE schedule (8 waves, one block per CU).
//
// The 128 x 128 core above is the "two barriers per k-tile, vmcnt(0) before the barrier" structure: every wave of a block loads,
// waits, multiplies in lockstep, and two such blocks per CU overlap only by accident -- 615-690 TFLOP/s on these shapes, 64 flop per
// byte moved L2 -> LDS (20 GB per training step, the measured bound: profiles/r04_wgrad_ablations.txt).  Here:
//   * 256 x 256 x 64 tiles (128 flop per L2 -> LDS byte), waves 2 (M) x 4 (N), wave tile 128 x 64 = 8 x 4 MFMA tiles, 128 accumulator
//     registers per lane;
//   * a k-tile is staged as FOUR half-tile images of [64 pixels][128 columns] (exactly the 16 KB image of the core above, same
//     swizzle, same transpose reads): Ah0 | Ah1 = output rows 0..127 | 128..255 of dY, Bh0 | Bh1 = columns 0..127 | 128..255 of x;
//     wave (wm, wn) owns rows {64 wm .. + 63} of BOTH row halves and columns {32 wn .. + 31} of BOTH column halves, so that the
//     quadrant a phase multiplies reads ONE A image and ONE B image and an image is dead as soon as its phase is over;
//   * a k-tile is four phases of 16 MFMAs per wave: (A0,B0) (A0,B1) (A1,B1) (A1,B0); a phase = [fragment reads of the sub-tile that
//     is new | issue ONE half-tile image of a later k-tile (2 x 1 KB LDS-DMA pieces per wave) | counted s_waitcnt vmcnt(8) |
//     s_waitcnt lgkmcnt(0) | s_barrier | 16 MFMAs at s_setprio 1 | s_barrier].  The two wave groups (wm = 0 | 1; a SIMD hosts one wave
//     of each) run ONE BARRIER apart: while one group's 16 MFMAs occupy the SIMD's matrix pipe the other group does its reads and
//     DMA issue;
//   * half-tiles are issued in the order they are read (Ah0, Bh0, Bh1, Ah1), five phases ahead: an image is overwritten two phases
//     after its last read (both groups are past it by then) and waited for one phase before its first read -- four half-tiles
//     (64 KB per CU) stay in flight across every barrier; two 64 KB buffers, no third stage.
// Work units, slabs and the grouped reduce pass are those of the 128 x 128 grouped launch.
// Measured (B = 32 bench shapes, tools/bench_wgrad_ablate.py, same box): layer4's ten gradients 456 -> 322 us, layer3's nineteen
// 859 -> 615 us.  PMC (tools/pmc_wg8.sh): MFMA pipe 51 % busy, no LDS bank conflicts, L2 hit rate 62 %, L2 misses = 1.1 x the
// operands' unique bytes -- what bounds it is operand delivery, which follows tools/probe/dma_rate.hip's additive model (a byte
// that hits L2 costs a CU 1/40 clock, one that misses 1/10: 15 B/clk/CU here); every schedule variant tried (reads waited for
// before / after the barrier, DMA issue before / after the reads, no s_setprio) measured the same +- 1 %, and an L2 prefetch from a
// dedicated wave group (touches of k-tile t + 2 | t + 4, cooperative among the tiles that share lines) LOST 15 %: the touches hold
// the same per-CU miss slots the real pieces wait for.
constexpr int W8_HALF = TBK * ROWBYTES;            // 16 KB: [64 pixels][128 columns]
constexpr int W8_BUF = 4 * W8_HALF;                // Ah0 | Ah1 | Bh0 | Bh1
constexpr int W8_LDS = 2 * W8_BUF;                 // 128 KB

// One transpose read as inline asm: the builtin carries an LDS memory operand, and in front of every such read hipcc waits
// vmcnt(0) for ALL LDS-DMA pieces in flight (it cannot tell their destinations from the read's source) -- the pipeline would drain in
// every phase (seen in the ISA).  The asm form is invisible to that pass; its results are ordered by the explicit
// `s_waitcnt lgkmcnt(0)` + sched_barrier in front of the MFMAs that consume them.
typedef short __attribute__((ext_vector_type(4))) w8_s16x4;
typedef short __attribute__((ext_vector_type(8))) w8_s16x8;
template <int OFF>
__device__ __forceinline__ bf16x8 tr_frag_asm(uint32_t addr) {
  w8_s16x4 a, b;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(a) : "v"(addr), "n"(OFF) : "memory");
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(b) : "v"(addr), "n"(OFF + 4 * ROWBYTES) : "memory");
  w8_s16x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, r);
}

template <bool CONV>
__device__ __forceinline__ void wg8_core(const GemmK& p, int tile, int ksplit) {
  extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const ConvGeom& g = p.cg;
  const int tm = tile / p.tilesN, tn = tile - tm * p.tilesN;
  const int row0 = tm * 256, col0 = tn * 256;
  const int kt_total = (p.K + TBK - 1) / TBK;
  const int kt0 = ksplit * p.kt_per_split;
  const int kt1 = min(kt_total, kt0 + p.kt_per_split);
  if (kt0 >= kt1) return;                            // (uniform; the host sizes the split so that this does not happen)
  const int nkt = kt1 - kt0;

  constexpr int OOB = 0x7ffffff0;
  const int lrow = lane >> 4;
  const int myrow = wave * 8 + lrow;                 // piece j of this wave covers pixel rows myrow + 4 j of a k-tile, j = 0, 1
  const int chunk = (lane & 15) ^ (2 * lrow) ^ (8 * (wave & 1));   // q(r) of row r = 8 wave + 4 j + lrow: (r & 3) = lrow, bit 3 = wave & 1

  // CONV: the tile's tap and channel block (256 divides Cin: checked on the host -> one tap per tile); the lane walks the two
  // pixel rows it fetches from k-tile to k-tile without divisions
  int c0 = 0, dh = 0, dw = 0, adv_q = 0, adv_r = 0;
  int px_b[2] = {0, 0}, px_oh[2] = {0, 0}, px_ow[2] = {0, 0}, px_k[2] = {0, 0};
  if constexpr (CONV) {
    const int tap = col0 / g.Cin;
    c0 = col0 - tap * g.Cin;
    const int tap_r = tap / g.KW, tap_s = tap - tap_r * g.KW;
    dh = tap_r - g.PH; dw = tap_s - g.PW;
    adv_q = TBK / g.OW; adv_r = TBK - adv_q * g.OW;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      px_k[j] = kt0 * TBK + myrow + 4 * j;
      px_b[j] = px_k[j] / (g.OH * g.OW);
      const int rem = px_k[j] - px_b[j] * (g.OH * g.OW);
      px_oh[j] = rem / g.OW;
      px_ow[j] = rem - px_oh[j] * g.OW;
    }
  }

  const auto rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), (short)0, OOB, 0x00020000);
  const auto rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), (short)0, OOB, 0x00020000);
  const int a_vo = (myrow * (int)p.lda + row0 + chunk * 8) * 2;
  const int a_j = 8 * (int)p.lda;                                  // piece j = 1: four rows further (bytes)
  const int b_vo = CONV ? (c0 + chunk * 8) * 2 : (myrow * (int)p.ldb + col0 + chunk * 8) * 2;
  const int b_j = 8 * (int)p.ldb;
  auto bload = [&](const decltype(rsA)& rs, int voff, int soff, unsigned char* l) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)l, 16, voff, soff, 0, 0);
  };
  // half-tile image h of relative k-tile tx -> buffer buf; beyond the unit's last k-tile the pieces are still ISSUED (out of range:
  // the DMA writes zeros nobody reads) so that every wave's vmcnt arithmetic is the same in every phase
  auto issueA = [&](int tx, int h, int buf) {
    const int kt = kt0 + tx;
    const bool tv = tx < nkt;
    const bool full = (kt + 1) * TBK <= p.K;
    unsigned char* dst = smem + buf * W8_BUF + h * W8_HALF + wave * 2048;
    const int so = tv ? kt * TBK * (int)p.lda * 2 : 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bool ok = tv && (full || kt * TBK + myrow + 4 * j < p.K);
      bload(rsA, ok ? a_vo + j * a_j + h * 256 : OOB, so, dst + j * 1024);
    }
  };
  int b_off[2] = {OOB, OOB};                          // CONV: gather offsets of my two rows for the k-tile whose B halves are being issued
  auto issueB = [&](int tx, int h, int buf) {
    const int kt = kt0 + tx;
    const bool tv = tx < nkt;
    unsigned char* dst = smem + buf * W8_BUF + (2 + h) * W8_HALF + wave * 2048;
    if constexpr (CONV) {
      if (h == 0) {                                   // (Bh0 is issued one phase before Bh1 of the same k-tile)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int ih = px_oh[j] * g.SH + dh, iw = px_ow[j] * g.SW + dw;
          const bool ok = tv && (unsigned)ih < (unsigned)g.IH && (unsigned)iw < (unsigned)g.IW && px_k[j] < p.K;
          b_off[j] = ok ? ((px_b[j] * g.IH + ih) * g.IW + iw) * g.Cs * 2 + b_vo : OOB;
          px_k[j] += TBK;
          px_ow[j] += adv_r;
          const int c = px_ow[j] >= g.OW ? 1 : 0;
          px_ow[j] -= c ? g.OW : 0;
          px_oh[j] += adv_q + c;
          const int c2 = px_oh[j] >= g.OH ? 1 : 0;
          px_oh[j] -= c2 ? g.OH : 0;
          px_b[j] += c2;
        }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)                       // (OOB + 256 is still out of range)
        bload(rsB, b_off[j] + h * 256, 0, dst + j * 1024);
    } else {
      const bool full = (kt + 1) * TBK <= p.K;
      const int so = tv ? kt * TBK * (int)p.ldb * 2 : 0;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const bool ok = tv && (full || kt * TBK + myrow + 4 * j < p.K);
        bload(rsB, ok ? b_vo + j * b_j + h * 256 : OOB, so, dst + j * 1024);
      }
    }
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment addresses inside a half-tile image (see the 128 x 128 core): A sub-tile = 16-column groups 4 wm + i, B = 2 wn + j
  const int li = lane & 15, fg = lane >> 4;
  const int fq = (li >> 2) | ((fg & 1) << 2);
  const int f_row = (8 * fg + (li >> 2)) * ROWBYTES + ((li & 3) >> 1) * 16 + (li & 1) * 8;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void_t*)smem;
  uint32_t a_sl[4], b_sl[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) a_sl[i] = lds0 + f_row + (((wm * 4 + i) ^ fq) << 5);
#pragma unroll
  for (int j = 0; j < 2; ++j) b_sl[j] = lds0 + f_row + (((wn * 2 + j) ^ fq) << 5);

  // plain form, bias gradient: a_rowsum[m] += sum_k A[k][m], on the waves of the first column tile.  The A fragments are in
  // registers: one more MFMA against an all-ones operand gives every lane the column sum of ITS m (lane & 15).  The four waves of a
  // row (wn = 0..3) hold the same A fragments: wave wn takes MFMA row tile wn of either half (two accumulators, not eight)
  const bool do_sum = !CONV && p.a_rowsum != nullptr && tn == 0;
  f32x4 sacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;

  bf16x8 af[4][2], b0f[2][2], b1f[2][2];
  // 16 MFMAs: accumulator rows i0 .. i0 + 3 x columns j0, j0 + 1, both 32-pixel halves of the k-tile (operands swapped: a lane
  // ends up with 4 consecutive columns of one row); the fragment reads are waited for BEFORE the barrier: their latency passes
  // under the other group's MFMAs
  auto mma = [&](int i0, int j0, const bf16x8 (&bf)[2][2], bool sum) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i0 + i][j0 + j] = mfma16(bf[j][kk], af[i][kk], acc[i0 + i][j0 + j]);
    if constexpr (!CONV) {
      if (sum && do_sum) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const bf16x8 a = wn == 0 ? af[0][kk] : wn == 1 ? af[1][kk] : wn == 2 ? af[2][kk] : af[3][kk];
          sacc[i0 >> 2] = mfma16(ones, a, sacc[i0 >> 2]);
        }
      }
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
#define W8_READ_A(HALF)                                                                        \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                              \
    af[i][0] = tr_frag_asm<(HALF) * W8_HALF>(a_sl[i] + bufo);                                  \
    af[i][1] = tr_frag_asm<(HALF) * W8_HALF + 32 * ROWBYTES>(a_sl[i] + bufo);                  \
  }
#define W8_READ_B(HALF, BF)                                                                    \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                              \
    BF[j][0] = tr_frag_asm<(2 + (HALF)) * W8_HALF>(b_sl[j] + bufo);                            \
    BF[j][1] = tr_frag_asm<(2 + (HALF)) * W8_HALF + 32 * ROWBYTES>(b_sl[j] + bufo);            \
  }

  // ---- prologue: k-tile 0 whole, Ah0 / Bh0 of k-tile 1 (12 pieces per wave); Ah0, Bh0 of k-tile 0 landed before the first read ----
  issueA(0, 0, 0); issueB(0, 0, 0); issueB(0, 1, 0); issueA(0, 1, 0); issueA(1, 0, 1); issueB(1, 0, 1);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();         // the second wave group runs one barrier behind the first
  asm volatile("" ::: "memory");

  for (int t = 0; t < nkt; ++t) {
    const int buf = t & 1;
    const uint32_t bufo = buf * W8_BUF;
    // phase 1: (A0, B0)
    W8_READ_B(0, b0f)
    W8_READ_A(0)
    __builtin_amdgcn_sched_barrier(0);
    issueB(t + 1, 1, buf ^ 1);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // Bh1 of this k-tile (phase 2) has landed: 4 half-tiles stay in flight
    mma(0, 0, b0f, true);
    // phase 2: (A0, B1)
    W8_READ_B(1, b1f)
    __builtin_amdgcn_sched_barrier(0);
    issueA(t + 1, 1, buf ^ 1);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // Ah1 of this k-tile (phase 3)
    mma(0, 2, b1f, false);
    // phase 3: (A1, B1)
    W8_READ_A(1)
    __builtin_amdgcn_sched_barrier(0);
    issueA(t + 2, 0, buf);                            // over Ah0 of this k-tile: last read two phases ago by either group
    mma(4, 2, b1f, true);
    // phase 4: (A1, B0)
    issueB(t + 2, 0, buf);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // Ah0, Bh0 of the next k-tile (its phase 1)
    mma(4, 0, b0f, false);
  }
#undef W8_READ_A
#undef W8_READ_B
  if (wm == 0) __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the trailing zero-fill pieces)

  if constexpr (!CONV) {
    if (do_sum && fg == 0) {
#pragma unroll
      for (int h = 0; h < 2; ++h) atomicAdd(p.a_rowsum + row0 + h * 128 + wm * 64 + wn * 16 + li, sacc[h][0]);
    }
  }

  // ---------------- epilogue: straight from the accumulators (a lane holds 4 consecutive columns of row li of each MFMA tile) --------
  const bool direct = p.ws == nullptr;
  float* Cp = direct ? reinterpret_cast<float*>(p.C) : p.ws + (int64_t)ksplit * p.M * p.N;
  const int64_t cpitch = direct ? p.ldc : p.N;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = row0 + (i >> 2) * 128 + wm * 64 + (i & 3) * 16 + li;
    const float rs = p.rowscale ? p.rowscale[m] * p.alpha : p.alpha;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = col0 + (j >> 1) * 128 + wn * 32 + (j & 1) * 16 + fg * 4;
      float4* dst = reinterpret_cast<float4*>(Cp + (int64_t)m * cpitch + n);
      float4 v = make_float4(acc[i][j][0] * rs, acc[i][j][1] * rs, acc[i][j][2] * rs, acc[i][j][3] * rs);
      if (direct) { const float4 c = *dst; v.x += c.x; v.y += c.y; v.z += c.z; v.w += c.w; }
      *dst = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Half-width eight-phase tiles (round 5): layer2's weight gradients have a 128-wide side (Cout = 128, or Cin = 128 with one tap per
// 128 columns) and ran on the 128 x 128 core above at 570 TFLOP/s -- exactly what 15 B/clk/CU of operand delivery gives a tile that
// does 64 flop per delivered byte.  The same pipeline on 128 x 256 (AH = 1, BH = 2) or 256 x 128 (AH = 2, BH = 1) tiles does 85:
//   * a k-tile is THREE half-tile images, in the order they are read: H0 = Ah0, H1 = Bh0, H2 = Bh1 | Ah1; two phases of 16 MFMAs
//     per wave: (A0, B0) then (A0, B1) | (A1, B0);
//   * three 48 KB stages (144 KB): phase 1 of k-tile t issues H0, H1 of k-tile t + 2, phase 2 issues its H2 -- six half-tiles
//     (96 KB) in flight across every barrier; a stage is overwritten a whole k-tile after its last read;
//   * a 256-column tile over Cin = 128 is TWO taps (each half gathers with its own tap offsets); the ragged last column tile of
//     9 x 128 columns issues its missing half out of range (zeros) and does not store it.
// Wave groups, barriers, fragment reads and the epilogue are those of wg8_core.
constexpr int W8H_BUF = 3 * W8_HALF;               // H0 | H1 | H2
constexpr int W8H_LDS = 3 * W8H_BUF;               // 144 KB

template <int AH, int BH>
__device__ __forceinline__ void wg8h_core(const GemmK& p, int tile, int ksplit) {
  static_assert(AH + BH == 3, "three half-tile images per k-tile");
  extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const ConvGeom& g = p.cg;
  const int tm = tile / p.tilesN, tn = tile - tm * p.tilesN;
  const int row0 = tm * (128 * AH), col0 = tn * (128 * BH);
  const int kt_total = (p.K + TBK - 1) / TBK;
  const int kt0 = ksplit * p.kt_per_split;
  const int kt1 = min(kt_total, kt0 + p.kt_per_split);
  if (kt0 >= kt1) return;
  const int nkt = kt1 - kt0;

  constexpr int OOB = 0x7ffffff0;
  const int lrow = lane >> 4;
  const int myrow = wave * 8 + lrow;
  const int chunk = (lane & 15) ^ (2 * lrow) ^ (8 * (wave & 1));

  // the tap and channel block of either column half (128 divides Cin: one tap per half); a half beyond N is never fetched
  int hc0[BH], hdh[BH], hdw[BH];
  bool hval[BH];
#pragma unroll
  for (int h = 0; h < BH; ++h) {
    const int c = col0 + h * 128;
    hval[h] = c < p.N;
    const int tap = min(c, p.N - 128) / g.Cin;
    hc0[h] = min(c, p.N - 128) - tap * g.Cin;
    const int tap_r = tap / g.KW, tap_s = tap - tap_r * g.KW;
    hdh[h] = tap_r - g.PH; hdw[h] = tap_s - g.PW;
  }
  const int adv_q = TBK / g.OW, adv_r = TBK - adv_q * g.OW;
  int px_b[2], px_oh[2], px_ow[2], px_k[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    px_k[j] = kt0 * TBK + myrow + 4 * j;
    px_b[j] = px_k[j] / (g.OH * g.OW);
    const int rem = px_k[j] - px_b[j] * (g.OH * g.OW);
    px_oh[j] = rem / g.OW;
    px_ow[j] = rem - px_oh[j] * g.OW;
  }

  const auto rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), (short)0, OOB, 0x00020000);
  const auto rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), (short)0, OOB, 0x00020000);
  const int a_vo = (myrow * (int)p.lda + row0 + chunk * 8) * 2;
  const int a_j = 8 * (int)p.lda;
  auto bload = [&](const decltype(rsA)& rs, int voff, int soff, unsigned char* l) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)l, 16, voff, soff, 0, 0);
  };
  // image slots inside a stage: Ah0 -> 0, Bh0 -> 1, the third image (Bh1 | Ah1) -> 2
  auto issueA = [&](int tx, int h, int buf) {
    const int kt = kt0 + tx;
    const bool tv = tx < nkt;
    const bool full = (kt + 1) * TBK <= p.K;
    unsigned char* dst = smem + buf * W8H_BUF + (h == 0 ? 0 : 2) * W8_HALF + wave * 2048;
    const int so = tv ? kt * TBK * (int)p.lda * 2 : 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bool ok = tv && (full || kt * TBK + myrow + 4 * j < p.K);
      bload(rsA, ok ? a_vo + j * a_j + h * 256 : OOB, so, dst + j * 1024);
    }
  };
  int b_off[BH][2];
#pragma unroll
  for (int h = 0; h < BH; ++h) b_off[h][0] = b_off[h][1] = OOB;
  auto issueB = [&](int tx, int h, int buf) {
    const bool tv = tx < nkt;
    unsigned char* dst = smem + buf * W8H_BUF + (1 + h) * W8_HALF + wave * 2048;
    if (h == 0) {                                     // (Bh0 is issued before Bh1 of the same k-tile: the gather of both, then the walk)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int hh = 0; hh < BH; ++hh) {
          const int ih = px_oh[j] * g.SH + hdh[hh], iw = px_ow[j] * g.SW + hdw[hh];
          const bool ok = tv && hval[hh] && (unsigned)ih < (unsigned)g.IH && (unsigned)iw < (unsigned)g.IW && px_k[j] < p.K;
          b_off[hh][j] = ok ? ((px_b[j] * g.IH + ih) * g.IW + iw) * g.Cs * 2 + (hc0[hh] + chunk * 8) * 2 : OOB;
        }
        px_k[j] += TBK;
        px_ow[j] += adv_r;
        const int c = px_ow[j] >= g.OW ? 1 : 0;
        px_ow[j] -= c ? g.OW : 0;
        px_oh[j] += adv_q + c;
        const int c2 = px_oh[j] >= g.OH ? 1 : 0;
        px_oh[j] -= c2 ? g.OH : 0;
        px_b[j] += c2;
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) bload(rsB, b_off[h][j], 0, dst + j * 1024);
  };
  auto issue01 = [&](int tx, int buf) { issueA(tx, 0, buf); issueB(tx, 0, buf); };
  auto issue2 = [&](int tx, int buf) {
    if constexpr (BH == 2) issueB(tx, 1, buf); else issueA(tx, 1, buf);
  };

  constexpr int NI = 4 * AH, NJ = 2 * BH;
  f32x4 acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int li = lane & 15, fg = lane >> 4;
  const int fq = (li >> 2) | ((fg & 1) << 2);
  const int f_row = (8 * fg + (li >> 2)) * ROWBYTES + ((li & 3) >> 1) * 16 + (li & 1) * 8;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void_t*)smem;
  uint32_t a_sl[4], b_sl[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) a_sl[i] = lds0 + f_row + (((wm * 4 + i) ^ fq) << 5);
#pragma unroll
  for (int j = 0; j < 2; ++j) b_sl[j] = lds0 + f_row + (((wn * 2 + j) ^ fq) << 5);

  bf16x8 af[4][2], bfr[2][2];
  auto mma = [&](int i0, int j0) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i0 + i][j0 + j] = mfma16(bfr[j][kk], af[i][kk], acc[i0 + i][j0 + j]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
#define W8H_READ_A(SLOT)                                                                       \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                              \
    af[i][0] = tr_frag_asm<(SLOT) * W8_HALF>(a_sl[i] + bufo);                                  \
    af[i][1] = tr_frag_asm<(SLOT) * W8_HALF + 32 * ROWBYTES>(a_sl[i] + bufo);                  \
  }
#define W8H_READ_B(SLOT)                                                                       \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                              \
    bfr[j][0] = tr_frag_asm<(SLOT) * W8_HALF>(b_sl[j] + bufo);                                 \
    bfr[j][1] = tr_frag_asm<(SLOT) * W8_HALF + 32 * ROWBYTES>(b_sl[j] + bufo);                 \
  }

  // ---- prologue: k-tiles 0 and 1 whole (12 pieces per wave); H0, H1 of k-tile 0 landed before the first read ----
  issue01(0, 0); issue2(0, 0); issue01(1, 1); issue2(1, 1);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();         // the second wave group runs one barrier behind the first
  asm volatile("" ::: "memory");

  int buf = 0, buf2 = 2;                             // stage of k-tile t | of k-tile t + 2
  for (int t = 0; t < nkt; ++t) {
    const uint32_t bufo = buf * W8H_BUF;
    // phase 1: (A0, B0)
    W8H_READ_B(1)
    W8H_READ_A(0)
    __builtin_amdgcn_sched_barrier(0);
    issue01(t + 2, buf2);
    asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); // H2 of this k-tile (phase 2) has landed: five half-tiles stay in flight
    mma(0, 0);
    // phase 2: (A0, B1) | (A1, B0)
    if constexpr (BH == 2) { W8H_READ_B(2) } else { W8H_READ_A(2) }
    __builtin_amdgcn_sched_barrier(0);
    issue2(t + 2, buf2);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // H0, H1 of the next k-tile (its phase 1)
    if constexpr (BH == 2) mma(0, 2); else mma(4, 0);
    buf = buf == 2 ? 0 : buf + 1;
    buf2 = buf2 == 2 ? 0 : buf2 + 1;
  }
#undef W8H_READ_A
#undef W8H_READ_B
  if (wm == 0) __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  const bool direct = p.ws == nullptr;
  float* Cp = direct ? reinterpret_cast<float*>(p.C) : p.ws + (int64_t)ksplit * p.M * p.N;
  const int64_t cpitch = direct ? p.ldc : p.N;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int m = row0 + (i >> 2) * 128 + wm * 64 + (i & 3) * 16 + li;
    const float rs = p.rowscale ? p.rowscale[m] * p.alpha : p.alpha;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (!hval[j >> 1]) continue;
      const int n = col0 + (j >> 1) * 128 + wn * 32 + (j & 1) * 16 + fg * 4;
      float4* dst = reinterpret_cast<float4*>(Cp + (int64_t)m * cpitch + n);
      float4 v = make_float4(acc[i][j][0] * rs, acc[i][j][1] * rs, acc[i][j][2] * rs, acc[i][j][3] * rs);
      if (direct) { const float4 c = *dst; v.x += c.x; v.y += c.y; v.z += c.z; v.w += c.w; }
      *dst = v;
    }
  }
}

template <bool CONV>
__device__ __forceinline__ void glds_tt_body(const GemmK& p) {
  // (tile, split) plane, split-major, contiguous range per XCD (see gemm.hip): an XCD runs all tiles of one reduction slice
  const int gx = gridDim.x, nwg = gx * (int)gridDim.y, bid = (int)blockIdx.x + gx * (int)blockIdx.y;
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
  const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  const int ksplit = v / gx;
  glds_tt_core<CONV>(p, v - ksplit * gx, ksplit);
}
__global__ __launch_bounds__(256) void glds_wgrad_kernel(GemmK p) { glds_tt_body<true>(p); }       // conv: B gathered from NHWC x
__global__ __launch_bounds__(256) void glds_tt_kernel(GemmK p) { glds_tt_body<false>(p); }         // linear: B = x [K][N]

// ---- grouped launch: many independent linear weight gradients in ONE grid ---------------------------------------------------
// The model body's ~140 weight-gradient GEMMs (0.44 TFLOP in all) are independent of each other and of everything else until
// the optimizer; launched one by one they cost 3.2 ms (17 us each: launch, first load, a 3-50-step k-loop on 4-144 blocks, a
// split reduction).  Here up to GPV_TT_GROUP_MAX problems travel in the kernel argument, every 128x128 tile of every problem is
// one workgroup that walks its whole reduction and adds its tile into the gradient (no split, no workspace, no second pass):
// tens of thousands of workgroups, the chip stays full.
struct GroupK {
  int n;
  int tile_start[GPV_TT_GROUP_MAX + 1];           // prefix sums of the problems' tile counts
  gpv_tt_problem prob[GPV_TT_GROUP_MAX];
};
static_assert(sizeof(GroupK) <= 4096, "kernel argument segment");

__global__ __launch_bounds__(256) void glds_tt_group_kernel(GroupK g) {
  const int bid = blockIdx.x;
  int pi = 0;
  while (pi + 1 < g.n && bid >= g.tile_start[pi + 1]) ++pi;          // uniform scalar walk (<= 48 steps)
  const gpv_tt_problem& q = g.prob[pi];
  GemmK p{};
  p.A = q.A; p.B = q.B; p.C = q.C; p.a_rowsum = q.a_rowsum;
  p.M = q.M; p.N = q.N; p.K = q.K; p.lda = q.lda; p.ldb = q.ldb; p.ldc = q.ldc;
  p.alpha = 1.0f; p.ws = nullptr;
  p.tilesN = q.N / TBN;
  p.kt_per_split = (q.K + TBK - 1) / TBK;
  glds_tt_core<false>(p, bid - g.tile_start[pi], 0);
}

// ---- grouped launch: ALL conv weight gradients of a backward pass in one or two grids --------------------------------------------
// Launched one by one (42 per step) every weight gradient sizes its split to fill the chip alone: 14..64 slabs of partial products
// per gradient -- as many bytes as the operands -- a reduction pass each, a ramp and a tail each.  Nobody needs a weight gradient
// before the optimizer, so the backward pass hands all of them over at its end: the work unit is (problem, split, tile), the split
// of a problem is sized so that every unit walks ~GPV_WGRAD_GROUP_KT (150) k-tiles of 64 pixels -- layer4 needs NO split at all
// (a tile adds itself into the gradient), layer3 4 slabs instead of 14..32 -- thousands of equal units keep the chip full, and
// one grouped pass adds the remaining slabs.
struct WgProb {
  const void* A; const void* B; float* C; const float* rowscale;
  int M, N, K, lda, ldc;
  int kt_per_split, tilesN, tiles;
  int unit_start;                        // prefix sum of tiles * split
  int direct;                            // split == 1: the tile is added into C
  int64_t ws_off;                        // floats into the workspace
  int IH, IW, Cs, Cin, OH, OW, KH, KW, SH, SW, PH, PW;
};
static_assert(sizeof(WgProb) == 128, "descriptor size");
constexpr int WG_MAX = 28;
struct WgGroupK { int n; int pad; float* ws; WgProb prob[WG_MAX]; };
static_assert(sizeof(WgGroupK) <= 4096, "kernel argument segment");

template <int ABL>
__device__ __forceinline__ void wgrad_group_body(const WgGroupK& g) {
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, loc = bid >> 3;
  const int v = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;     // contiguous unit range per XCD
  int pi = 0;
  while (pi + 1 < g.n && v >= g.prob[pi + 1].unit_start) ++pi;
  const WgProb& q = g.prob[pi];
  GemmK p{};
  p.A = q.A; p.B = q.B; p.C = q.C; p.rowscale = q.rowscale;
  p.M = q.M; p.N = q.N; p.K = q.K; p.lda = q.lda; p.ldb = 0; p.ldc = q.ldc;
  p.alpha = 1.0f;
  p.ws = q.direct ? nullptr : g.ws + q.ws_off;
  p.tilesN = q.tilesN; p.kt_per_split = q.kt_per_split;
  p.cg.IH = q.IH; p.cg.IW = q.IW; p.cg.Cs = q.Cs; p.cg.Cin = q.Cin; p.cg.OH = q.OH; p.cg.OW = q.OW;
  p.cg.KH = q.KH; p.cg.KW = q.KW; p.cg.SH = q.SH; p.cg.SW = q.SW; p.cg.PH = q.PH; p.cg.PW = q.PW;
  const int u = v - q.unit_start;
  const int ksplit = u / q.tiles;
  glds_tt_core<true, ABL>(p, u - ksplit * q.tiles, ksplit);
}
__global__ __launch_bounds__(256) void glds_wgrad_group_kernel(WgGroupK g) { wgrad_group_body<0>(g); }
// the problems whose Cout and Cin are multiples of 256 (layer3, layer4): the same units on 256 x 256 tiles, eight-phase core
__global__ __launch_bounds__(512) void wg8_group_kernel(WgGroupK g) {
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, loc = bid >> 3;
  const int v = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;     // contiguous unit range per XCD
  int pi = 0;
  while (pi + 1 < g.n && v >= g.prob[pi + 1].unit_start) ++pi;
  const WgProb& q = g.prob[pi];
  GemmK p{};
  p.A = q.A; p.B = q.B; p.C = q.C; p.rowscale = q.rowscale;
  p.M = q.M; p.N = q.N; p.K = q.K; p.lda = q.lda; p.ldb = 0; p.ldc = q.ldc;
  p.alpha = 1.0f;
  p.ws = q.direct ? nullptr : g.ws + q.ws_off;
  p.tilesN = q.tilesN; p.kt_per_split = q.kt_per_split;
  p.cg.IH = q.IH; p.cg.IW = q.IW; p.cg.Cs = q.Cs; p.cg.Cin = q.Cin; p.cg.OH = q.OH; p.cg.OW = q.OW;
  p.cg.KH = q.KH; p.cg.KW = q.KW; p.cg.SH = q.SH; p.cg.SW = q.SW; p.cg.PH = q.PH; p.cg.PW = q.PW;
  const int u = v - q.unit_start;
  const int ksplit = u / q.tiles;
  wg8_core<true>(p, u - ksplit * q.tiles, ksplit);
}
// layer2's problems (a 128-wide side): the same units on 128 x 256 | 256 x 128 tiles, wg8h_core; WgProb::direct carries the tile
// shape in bit 1 (0 = 128 x 256, 1 = 256 x 128)
__global__ __launch_bounds__(512) void wg8h_group_kernel(WgGroupK g) {
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, loc = bid >> 3;
  const int v = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;     // contiguous unit range per XCD
  int pi = 0;
  while (pi + 1 < g.n && v >= g.prob[pi + 1].unit_start) ++pi;
  const WgProb& q = g.prob[pi];
  GemmK p{};
  p.A = q.A; p.B = q.B; p.C = q.C; p.rowscale = q.rowscale;
  p.M = q.M; p.N = q.N; p.K = q.K; p.lda = q.lda; p.ldb = 0; p.ldc = q.ldc;
  p.alpha = 1.0f;
  p.ws = (q.direct & 1) ? nullptr : g.ws + q.ws_off;
  p.tilesN = q.tilesN; p.kt_per_split = q.kt_per_split;
  p.cg.IH = q.IH; p.cg.IW = q.IW; p.cg.Cs = q.Cs; p.cg.Cin = q.Cin; p.cg.OH = q.OH; p.cg.OW = q.OW;
  p.cg.KH = q.KH; p.cg.KW = q.KW; p.cg.SH = q.SH; p.cg.SW = q.SW; p.cg.PH = q.PH; p.cg.PW = q.PW;
  const int u = v - q.unit_start;
  const int ksplit = u / q.tiles;
  if (q.direct & 2) wg8h_core<2, 1>(p, u - ksplit * q.tiles, ksplit);
  else wg8h_core<1, 2>(p, u - ksplit * q.tiles, ksplit);
}
#ifdef GPV_TUNING        // timing-ablation instances (wrong results by construction): tuning build only, never in libgpv_hip.so
__global__ __launch_bounds__(256) void glds_wgrad_group_abl1_kernel(WgGroupK g) { wgrad_group_body<1>(g); }
__global__ __launch_bounds__(256) void glds_wgrad_group_abl2_kernel(WgGroupK g) { wgrad_group_body<2>(g); }
__global__ __launch_bounds__(256) void glds_wgrad_group_abl3_kernel(WgGroupK g) { wgrad_group_body<3>(g); }
__global__ __launch_bounds__(256) void glds_wgrad_group_abl4_kernel(WgGroupK g) { wgrad_group_body<4>(g); }
__global__ __launch_bounds__(256) void glds_wgrad_group_abl5_kernel(WgGroupK g) { wgrad_group_body<5>(g); }
__global__ __launch_bounds__(256) void glds_wgrad_group_abl6_kernel(WgGroupK g) { wgrad_group_body<6>(g); }
#endif

// ---- grouped launch, linear weight gradients on the eight-phase core (round 5) ------------------------------------------------------
// gpv_gemm_tt_group's problems whose M and N are multiples of 256 (every nn.Linear of the DETR transformer, the co-attention layers
// and the text decoder at the shipped widths 256 / 768 / 2048 / 3072).  Unlike the conv groups the reductions differ (9600 / 3392 /
// 3200 / 640 / 192 rows) and a 256 x 256 output is ONE tile: long reductions are cut into slices of ~W8L_KT k-tiles (partial tiles
// through the caller's workspace + the grouped reduce pass), and the work units -- (problem, slice, tile), between 3 and ~75 k-tiles
// long -- are dealt to the 8 XCDs as whole (problem, slice) GROUPS, longest first to the least loaded XCD: the tiles of a group share
// operand rows (one L2), and an XCD's units start in descending length, so the grid drains through its short units.  Block b runs on
// XCD b % 8: XCD x owns units [x Q, x Q + cnt[x]) of the table, the surplus blocks of the shorter lists exit at once.
constexpr int W8L_PMAX = GPV_TT_GROUP_MAX, W8L_GMAX = 200;
struct W8LProb {                                    // 64 bytes: 48 of them + the group table fit the 4 KB kernel-argument segment
  const void* A; const void* B; float* C; float* a_rowsum;
  int MN, N, K, lda, ldb, ldc;
  int kt_per_split;
  int ws_off;                                       // floats into the workspace; < 0: the tile is added into C (no slices)
};
struct W8LGrp { unsigned short v_start; unsigned char prob, slice; };
struct W8LGroupK { int n_prob, n_grp, Q, pad; int cnt[8]; float* ws; W8LProb prob[W8L_PMAX]; W8LGrp grp[W8L_GMAX]; };
static_assert(sizeof(W8LProb) == 64 && sizeof(W8LGrp) == 4 && sizeof(W8LGroupK) <= 4096, "kernel argument segment");

__global__ __launch_bounds__(512) void wg8_tt_group_kernel(W8LGroupK g) {
  const int bid = blockIdx.x, xcd = bid & 7, loc = bid >> 3;
  if (loc >= g.cnt[xcd]) return;
  const int v = xcd * g.Q + loc;
  int lo = 0, hi = g.n_grp - 1;                       // last group whose v_start <= v (uniform scalar search)
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if ((int)g.grp[mid].v_start <= v) lo = mid; else hi = mid - 1;
  }
  const W8LGrp gr = g.grp[lo];
  const W8LProb& q = g.prob[gr.prob];
  GemmK p{};
  p.A = q.A; p.B = q.B; p.C = q.C; p.a_rowsum = q.a_rowsum;
  p.M = q.MN / q.N; p.N = q.N; p.K = q.K; p.lda = q.lda; p.ldb = q.ldb; p.ldc = q.ldc;
  p.alpha = 1.0f;
  p.ws = q.ws_off < 0 ? nullptr : g.ws + q.ws_off;
  p.tilesN = q.N / 256; p.kt_per_split = q.kt_per_split;
  wg8_core<false>(p, v - gr.v_start, gr.slice);
}

struct WgRed { const float* ws; float* C; int64_t MN; int split, N, ldc, blk_start; };
struct WgRedK { int n; int pad; WgRed r[WG_MAX]; };

__global__ __launch_bounds__(256) void wgrad_group_reduce_kernel(WgRedK g) {
  const int bid = blockIdx.x;
  int pi = 0;
  while (pi + 1 < g.n && bid >= g.r[pi + 1].blk_start) ++pi;
  const WgRed& q = g.r[pi];
  const int64_t idx = (int64_t)(bid - q.blk_start) * 256 + threadIdx.x;     // one quad of 4 columns per thread
  if (idx * 4 >= q.MN) return;
  const int nq = q.N >> 2;
  const int m = (int)(idx / nq), c4 = (int)(idx - (int64_t)m * nq);
  const float* src = q.ws + (int64_t)m * q.N + c4 * 4;
  float4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int sidx = 0; sidx < q.split; ++sidx) {
    const float4 v = *reinterpret_cast<const float4*>(src + (int64_t)sidx * q.MN);
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  float4* dst = reinterpret_cast<float4*>(q.C + (int64_t)m * q.ldc + c4 * 4);
  float4 c = *dst;
  c.x += a.x; c.y += a.y; c.z += a.z; c.w += a.w;
  *dst = c;
}

// shared launch tail: split sizing, workspace slabs, kernel, reduction
template <typename F>
int launch_tt(F fn, const GemmK& k, bool& attr_done, hipStream_t st) {
  GemmK p = k;
  const int kt_total = (p.K + TBK - 1) / TBK;
  // two 64 KB blocks per CU = 512 resident blocks: size the split so that (tiles x splits) fills them once -- the 544 the
  // register-staged kernel (three blocks per CU) is tuned for would leave a second, almost empty round
  static const int target = tune_env("GPV_WGRAD_TARGET", 512);
  const int tiles = (p.M / TBM) * (p.N / TBN);
  int split = target / tiles;
  if (split > kt_total / 4) split = kt_total / 4;
  if (split < 2) return -1;
  while (split > 2 && (int64_t)split * p.M * p.N * 4 > k.ws_bytes) --split;
  p.kt_per_split = (kt_total + split - 1) / split;
  split = (kt_total + p.kt_per_split - 1) / p.kt_per_split;
  if (split < 2 || (int64_t)split * p.M * p.N * 4 > k.ws_bytes) return -1;
  p.ws = reinterpret_cast<float*>(k.ws_base);
  p.tilesN = p.N / TBN;
  constexpr int lds = 2 * TSTAGE;                    // 64 KB (the fp32 epilogue image needs 33 KB)
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }      // (leave no stale error behind for the next launch check)
    attr_done = true;
  }
  dim3 grid((p.M / TBM) * p.tilesN, split, 1);
  hipLaunchKernelGGL(fn, grid, dim3(256), lds, st, p);
  GPV_CHECK_LAUNCH();
  return launch_splitk_reduce(p.ws, split, p.M, p.N, reinterpret_cast<float*>(p.C), p.ldc, st);
}

inline bool al16t(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

}  // namespace

int g_wgrad_mode = tune_env("GPV_GLDS_WGRAD", 1);   // gpv_set_option(GPV_OPT_GLDS_WGRAD, .)
int g_wg8_mode = tune_env("GPV_WG8", 1);            // gpv_set_option(GPV_OPT_WG8, .): eight-phase 256 x 256 weight-gradient kernel
long g_wg8_launches = 0;
int g_wg8h_mode = tune_env("GPV_WG8H", 1);          // gpv_set_option(GPV_OPT_WG8H, .): the 128 x 256 | 256 x 128 eight-phase tiles for the problems with a 128-wide side (layer2)
int g_w8l_mode = tune_env("GPV_W8L", 0);            // gpv_set_option(GPV_OPT_W8L, .): the same kernel on the linear weight-gradient groups -- OFF by default: alone on the chip the step's 123
                                                    // linear gradients take 810 instead of 885 us with it, inside the step nothing (16.63 / 16.60 ms off, 16.66 / 16.77 on, same box): its
                                                    // 128 KB blocks take whole CUs from the backward chain they run beside

// conv weight gradient (k as prepared by gpv_conv2d mode 2: A = dy [K][M], B = x NHWC, C = dw [M][N] fp32, split chosen).
// returns 0 = launched (partial products + reduction), -1 = not applicable, > 0 = hipError_t
int glds_wgrad_try_launch(const GemmK& k, int dtype_in, int dtype_out, hipStream_t st) {
  if (g_wgrad_mode == 0 || dtype_in != GPV_BF16 || dtype_out != GPV_F32) return -1;
  const ConvGeom& g = k.cg;
  if (k.M % TBM != 0 || g.Cin % TBN != 0 || k.N % TBN != 0) return -1;
  if (g.Cs % 8 != 0 || k.lda % 8 != 0 || !al16t(k.A) || !al16t(k.B)) return -1;
  if (TBK / g.OW + 2 > g.OH || (int64_t)g.IH * g.IW * g.Cs * (k.K / (g.OH * g.OW)) >= (1ll << 30) ||
      (int64_t)k.K * k.lda >= (1ll << 30)) return -1;
  if (!k.accumulate || k.ws_base == nullptr || !al16t(k.C) || k.ldc % 4 != 0) return -1;
  static bool attr_done = false;
  return launch_tt(glds_wgrad_kernel, k, attr_done, st);
}

// linear weight gradient dW[M][N] += dY^T X (A = dy [K][M], B = x [K][N], both reduction-major), optional a_rowsum (bias
// gradient).  Same return convention.
int glds_tt_try_launch(const GemmK& k, int dtype_in, int dtype_out, int batch, hipStream_t st) {
  if (g_wgrad_mode == 0 || dtype_in != GPV_BF16 || dtype_out != GPV_F32 || batch != 1) return -1;
  if (k.M % TBM != 0 || k.N % TBN != 0 || k.K < 8 * TBK) return -1;
  if ((int64_t)k.K * k.lda >= (1ll << 30) || (int64_t)k.K * k.ldb >= (1ll << 30)) return -1;
  // measured on every linear weight-gradient shape of the step (tools/bench_step_gemms.py, GPV_GLDS_WGRAD=0 vs 2): it wins
  // once there is enough work to fill the chip -- 768x3072 / 3072x768 over 3200 rows 52 -> 35 us, 1536x768 36 -> 26 us,
  // 2048x256 over 9600 rows 33 -> 27 us -- and loses on the 4..36-tile gradients of short reductions (skinny_tt's territory)
  if (g_wgrad_mode == 1 && (int64_t)(k.M / TBM) * (k.N / TBN) * ((k.K + TBK - 1) / TBK) < 3000) return -1;
  if (k.lda % 8 != 0 || k.ldb % 8 != 0 || !al16t(k.A) || !al16t(k.B)) return -1;
  if (!k.accumulate || k.ws_base == nullptr || !al16t(k.C) || k.ldc % 4 != 0) return -1;
  if (k.res || k.mask || k.bias || k.act || k.dthresh) return -1;
  static bool attr_done = false;
  return launch_tt(glds_tt_kernel, k, attr_done, st);
}

}  // namespace gpvk

extern "C" int gpv_gemm_tt_group(const gpv_tt_problem* problems, int n, void* stream) {
  using namespace gpvk;
  if (!problems || n <= 0) return (int)hipErrorInvalidValue;
  static bool attr_done = false;
  constexpr int lds = 2 * TSTAGE;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(glds_tt_group_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }      // (leave no stale error behind for the next launch check)
    attr_done = true;
  }
  for (int i0 = 0; i0 < n; i0 += GPV_TT_GROUP_MAX) {
    GroupK g{};
    g.n = n - i0 < GPV_TT_GROUP_MAX ? n - i0 : GPV_TT_GROUP_MAX;
    int tiles = 0;
    for (int i = 0; i < g.n; ++i) {
      const gpv_tt_problem& q = problems[i0 + i];
      if (!q.A || !q.B || !q.C || q.M <= 0 || q.N <= 0 || q.K <= 0 || q.M % TBM != 0 || q.N % TBN != 0 || q.lda % 8 != 0 ||
          q.ldb % 8 != 0 || q.ldc % 4 != 0 || !al16t(q.A) || !al16t(q.B) || !al16t(q.C) ||
          (int64_t)q.K * q.lda >= (1ll << 30) || (int64_t)q.K * q.ldb >= (1ll << 30))
        return (int)hipErrorInvalidValue;
      g.prob[i] = q;
      g.tile_start[i] = tiles;
      tiles += (q.M / TBM) * (q.N / TBN);
    }
    g.tile_start[g.n] = tiles;
    hipLaunchKernelGGL(glds_tt_group_kernel, dim3(tiles), dim3(256), lds, reinterpret_cast<hipStream_t>(stream), g);
    GPV_CHECK_LAUNCH();
  }
  return 0;
}

extern "C" int gpv_conv_wgrad_group(const gpv_conv_wgrad_problem* probs, int n, void* workspace, int64_t workspace_bytes, void* stream) {
  using namespace gpvk;
  if (!probs || n <= 0) return (int)hipErrorInvalidValue;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // k-tiles of 64 pixels per work unit: 150 = layer4's whole reduction at B = 32 (measured: 75 / 100 / 150 / 250 / 300 ->
  // 2.06 / 2.03 / 2.00 / 2.37 / 2.45 ms for the 42 gradients of the training step)
  static const int target_kt = [] { const int v = tune_env("GPV_WGRAD_GROUP_KT", 150); return v < 8 ? 8 : v; }();
  static bool attr_done = false;
  constexpr int lds = 2 * TSTAGE;
  // GPV_WG_ABL=1|2|3 (timing experiments only, wrong results): see glds_tt_core
  typedef void (*wg_fn)(WgGroupK);
#ifdef GPV_TUNING
  static const int abl = tune_env("GPV_WG_ABL", 0);
  const wg_fn fn = abl == 1 ? glds_wgrad_group_abl1_kernel : abl == 2 ? glds_wgrad_group_abl2_kernel : abl == 3 ? glds_wgrad_group_abl3_kernel : abl == 4 ? glds_wgrad_group_abl4_kernel : abl == 5 ? glds_wgrad_group_abl5_kernel : abl == 6 ? glds_wgrad_group_abl6_kernel
                                                                                                        : glds_wgrad_group_kernel;
#else
  const wg_fn fn = glds_wgrad_group_kernel;
#endif
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }      // (leave no stale error behind for the next launch check)
    attr_done = true;
  }
  // the eight-phase 256 x 256 kernel takes the problems whose Cout and Cin are multiples of 256 (one tap per column tile) -- when
  // the call has enough of them to fill the chip: the two kernels run one after the other, and a handful of 256 x 256 units alone
  // on 256 CUs (layer2's call: its down-sample projection only) costs more than it saves (695 -> 955 us measured)
  static bool attr8_done = false;
  if (!attr8_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wg8_group_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, W8_LDS);
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }
    attr8_done = true;
  }
  static const int target_kt8 = [] { const int v = tune_env("GPV_WG8_KT", 150); return v < 8 ? 8 : v; }();
  auto eligible8 = [&](const gpv_conv_wgrad_problem& q) {
    const int64_t K64 = (int64_t)q.B * q.OH * q.OW;
    return q.Cout % 256 == 0 && q.Cin % 256 == 0 && q.Cs % 8 == 0 && al16t(q.dy) && al16t(q.x) && al16t(q.dw) &&
           TBK / q.OW + 2 <= q.OH && (int64_t)q.IH * q.IW * q.Cs * q.B < (1ll << 30) && K64 * q.Cout < (1ll << 30) && K64 >= 8 * TBK;
  };
  static bool attr8h_done = false;
  if (!attr8h_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wg8h_group_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, W8H_LDS);
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }
    attr8h_done = true;
  }
  static const int forced_kt8h = tune_env("GPV_WG8H_KT", 0);       // 0: chosen per call (below)
  static const int ramp_kt8h = tune_env("GPV_WG8H_C", 15);
  // a 128-wide side: Cout or Cin an odd multiple of 128 (or a 256-multiple problem the 256 x 256 launch did not take)
  auto eligible8h = [&](const gpv_conv_wgrad_problem& q) {
    const int64_t K64 = (int64_t)q.B * q.OH * q.OW;
    return g_wg8h_mode != 0 && q.Cout % 128 == 0 && q.Cin % 128 == 0 && q.Cs % 8 == 0 && al16t(q.dy) && al16t(q.x) && al16t(q.dw) &&
           TBK / q.OW + 2 <= q.OH && (int64_t)q.IH * q.IW * q.Cs * q.B < (1ll << 30) && K64 * q.Cout < (1ll << 30) && K64 >= 8 * TBK;
  };
  bool use8 = g_wg8_mode != 0;
  if (use8 && g_wg8_mode == 1) {
    int64_t u8 = 0;
    for (int i = 0; i < n; ++i) {
      const gpv_conv_wgrad_problem& q = probs[i];
      if (!eligible8(q)) continue;
      const int64_t kt = ((int64_t)q.B * q.OH * q.OW + TBK - 1) / TBK;
      u8 += (int64_t)(q.Cout / 256) * (q.KH * q.KW * q.Cin / 256) * ((kt + target_kt8 / 2) / target_kt8 > 0 ? (kt + target_kt8 / 2) / target_kt8 : 1);
    }
    use8 = u8 >= 128;
  }
  // work-unit length of the half-width launch: one 144 KB block per CU, so the launch runs in whole rounds of 256 units and its
  // makespan is rounds x (unit length + ramp) -- measured on layer2's call (B = 32): 60 / 75 / 100 / 120 / 150 / 200 / 300 k-tiles
  // per unit = 7 / 6 / 4 / 4 / 3 / 2 / 2 rounds -> 676 / 698 / 617 / 700 / 644 / 590 / 807 us.  Pick the length that minimises it.
  int target_kt8h = forced_kt8h >= 8 ? forced_kt8h : 150;
  if (forced_kt8h < 8 && g_wg8h_mode != 0) {
    int64_t best = -1;
    for (int kt = 64; kt <= 640; kt += 4) {
      int64_t u = 0;
      int longest = 0;
      for (int i = 0; i < n; ++i) {
        const gpv_conv_wgrad_problem& q = probs[i];
        if ((use8 && eligible8(q)) || !eligible8h(q)) continue;
        const int M = q.Cout, N = q.KH * q.KW * q.Cin;
        const int kt_total = (int)(((int64_t)q.B * q.OH * q.OW + TBK - 1) / TBK);
        int split = (kt_total + kt / 2) / kt;
        if (split < 1) split = 1;
        const int kps = (kt_total + split - 1) / split;
        split = (kt_total + kps - 1) / kps;
        u += (int64_t)(M % 256 == 0 ? (M / 256) * (N / 128) : (M / 128) * ((N + 255) / 256)) * split;
        if (kps > longest) longest = kps;
      }
      if (u == 0) break;
      const int64_t cost = ((u + 255) / 256) * (longest + ramp_kt8h);
      if (best < 0 || cost < best) { best = cost; target_kt8h = kt; }
    }
  }
  WgGroupK g{}, g8{}, gh{};
  WgRedK rk{};
  int units = 0, units8 = 0, unitsh = 0, rblocks = 0;
  int64_t ws_used = 0;
  const int64_t ws_floats = workspace ? workspace_bytes / 4 : 0;
  auto flush = [&]() -> int {
    if (g.n == 0 && g8.n == 0 && gh.n == 0) return 0;
    if (g8.n > 0) {
      g8.ws = reinterpret_cast<float*>(workspace);
      hipLaunchKernelGGL(wg8_group_kernel, dim3(units8), dim3(512), W8_LDS, st, g8);
      GPV_CHECK_LAUNCH();
      ++g_wg8_launches;
    }
    if (gh.n > 0) {
      gh.ws = reinterpret_cast<float*>(workspace);
      hipLaunchKernelGGL(wg8h_group_kernel, dim3(unitsh), dim3(512), W8H_LDS, st, gh);
      GPV_CHECK_LAUNCH();
      ++g_wg8_launches;
    }
    if (g.n > 0) {
      g.ws = reinterpret_cast<float*>(workspace);
      hipLaunchKernelGGL(fn, dim3(units), dim3(256), lds, st, g);
      GPV_CHECK_LAUNCH();
    }
    if (rk.n > 0) {
      hipLaunchKernelGGL(wgrad_group_reduce_kernel, dim3(rblocks), dim3(256), 0, st, rk);
      GPV_CHECK_LAUNCH();
    }
    g.n = 0; g8.n = 0; gh.n = 0; rk.n = 0; units = 0; units8 = 0; unitsh = 0; rblocks = 0; ws_used = 0;
    return 0;
  };
  for (int i = 0; i < n; ++i) {
    const gpv_conv_wgrad_problem& q = probs[i];
    if (!q.x || !q.dy || !q.dw || q.B <= 0) return (int)hipErrorInvalidValue;
    const int M = q.Cout, N = q.KH * q.KW * q.Cin;
    const int64_t K64 = (int64_t)q.B * q.OH * q.OW;
    const bool ok = M % TBM == 0 && q.Cin % TBN == 0 && q.Cs % 8 == 0 && al16t(q.dy) && al16t(q.x) && al16t(q.dw) &&
                    TBK / q.OW + 2 <= q.OH && (int64_t)q.IH * q.IW * q.Cs * q.B < (1ll << 30) && K64 * M < (1ll << 30) && K64 >= 8 * TBK;
    if (!ok) {                                       // a shape the direct-to-LDS kernel does not take: its own launch
      gpv_conv_args a{};
      a.mode = 2; a.x = q.x; a.w = q.dy; a.y = q.dw;
      a.B = q.B; a.IH = q.IH; a.IW = q.IW; a.Cs = q.Cs; a.Cin = q.Cin; a.OH = q.OH; a.OW = q.OW; a.Cout = q.Cout;
      a.KH = q.KH; a.KW = q.KW; a.SH = q.SH; a.SW = q.SW; a.PH = q.PH; a.PW = q.PW;
      a.dtype_in = GPV_BF16; a.dtype_out = GPV_F32; a.rowscale = q.rowscale;
      int e = flush();                               // (the workspace is busy with this group's slabs until it is flushed)
      if (e) return e;
      a.workspace = workspace; a.workspace_bytes = workspace_bytes;
      e = gpv_conv2d(&a, stream);
      if (e) return e;
      continue;
    }
    const bool big = use8 && eligible8(q);
    const bool half = !big && eligible8h(q);
    const bool tall = half && M % 256 == 0;          // 256 x 128 tiles | 128 x 256
    const int tbm = big ? 256 : half ? (tall ? 256 : 128) : TBM, tbn = big ? 256 : half ? (tall ? 128 : 256) : TBN;
    const int K = (int)K64;
    const int kt_total = (K + TBK - 1) / TBK;
    const int tkt = big ? target_kt8 : half ? target_kt8h : target_kt;
    int split = (kt_total + tkt / 2) / tkt;
    if (split < 1) split = 1;
    const int64_t MN = (int64_t)M * N;
    while (split > 1 && (int64_t)split * MN > ws_floats) --split;
    const int kps = (kt_total + split - 1) / split;
    split = (kt_total + kps - 1) / kps;
    WgGroupK& gg = big ? g8 : half ? gh : g;
    if (gg.n == WG_MAX || (split > 1 && (rk.n == WG_MAX || ws_used + (int64_t)split * MN > ws_floats))) {
      const int e = flush();
      if (e) return e;
    }
    int& uu = big ? units8 : half ? unitsh : units;
    WgProb& d = gg.prob[gg.n];
    d.A = q.dy; d.B = q.x; d.C = q.dw; d.rowscale = q.rowscale;
    d.M = M; d.N = N; d.K = K; d.lda = M; d.ldc = N;
    d.kt_per_split = kps; d.tilesN = (N + tbn - 1) / tbn; d.tiles = (M / tbm) * d.tilesN;
    d.unit_start = uu; d.direct = (split == 1 ? 1 : 0) | (tall ? 2 : 0); d.ws_off = ws_used;
    d.IH = q.IH; d.IW = q.IW; d.Cs = q.Cs; d.Cin = q.Cin; d.OH = q.OH; d.OW = q.OW;
    d.KH = q.KH; d.KW = q.KW; d.SH = q.SH; d.SW = q.SW; d.PH = q.PH; d.PW = q.PW;
    uu += d.tiles * split;
    if (split > 1) {
      WgRed& r = rk.r[rk.n];
      r.ws = reinterpret_cast<const float*>(workspace) + ws_used; r.C = q.dw; r.MN = MN; r.split = split; r.N = N; r.ldc = N;
      r.blk_start = rblocks;
      rblocks += (int)((MN / 4 + 255) / 256);
      ++rk.n;
      ws_used += (int64_t)split * MN;
    }
    ++gg.n;
  }
  return flush();
}

extern "C" int gpv_gemm_tt_group_ws(const gpv_tt_problem* problems, int n, void* workspace, int64_t workspace_bytes, void* stream) {
  using namespace gpvk;
  if (!problems || n <= 0) return (int)hipErrorInvalidValue;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // k-tiles per work unit (16 / 24 / 32 / 48 / 64 / 100 -> 927 / 818 / 835 / 761 / 815 / 809 us for the step's 123 problems alone on the
  // chip, tools/bench_tt_group.py; the 128 x 128 grouped launches: 880): these gradients move 128 flop per operand byte at best -- a
  // 256 x 256 output is ONE tile that reads both operands once -- and a lone tile pulls its misses at ~10 B/clk (3 us per k-tile):
  // shorter units balance the CUs, longer ones write fewer partial tiles
  static const int target_kt = [] { const int v = tune_env("GPV_W8L_KT", 48); return v < 8 ? 8 : v; }();
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wg8_tt_group_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, W8_LDS);
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }
    attr_done = true;
  }
  const int64_t ws_floats = workspace ? workspace_bytes / 4 : 0;
  // the problems the 256 x 256 kernel does not take keep the 128 x 128 grouped launch (after the big ones: they are the short ones)
  gpv_tt_problem rest[GPV_TT_GROUP_MAX];
  int n_rest = 0;
  auto flush_rest = [&]() -> int {
    if (n_rest == 0) return 0;
    const int e = gpv_gemm_tt_group(rest, n_rest, stream);
    n_rest = 0;
    return e;
  };
  struct Grp { int prob, slice, tiles, kps; };
  W8LGroupK g{};
  WgRedK rk{};
  Grp grp[W8L_GMAX];
  int n_grp = 0, rblocks = 0;
  int64_t ws_used = 0;
  auto flush = [&]() -> int {
    if (g.n_prob == 0) return 0;
    // longest groups first, each to the XCD with the least work so far
    for (int i = 1; i < n_grp; ++i) {                   // (insertion sort: <= 160 entries, mostly sorted already)
      const Grp x = grp[i];
      int j = i - 1;
      while (j >= 0 && grp[j].kps < x.kps) { grp[j + 1] = grp[j]; --j; }
      grp[j + 1] = x;
    }
    int64_t load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int bin_of[W8L_GMAX], cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n_grp; ++i) {
      int b = 0;
      for (int x = 1; x < 8; ++x) if (load[x] < load[b]) b = x;
      bin_of[i] = b;
      load[b] += (int64_t)grp[i].tiles * grp[i].kps;
      cnt[b] += grp[i].tiles;
    }
    int Q = 0;
    for (int x = 0; x < 8; ++x) { g.cnt[x] = cnt[x]; if (cnt[x] > Q) Q = cnt[x]; }
    if (8 * Q >= 65536) return (int)hipErrorInvalidValue;
    g.Q = Q; g.n_grp = n_grp;
    int k = 0;
    for (int x = 0; x < 8; ++x) {
      int off = 0;
      for (int i = 0; i < n_grp; ++i) {
        if (bin_of[i] != x) continue;
        g.grp[k].v_start = (unsigned short)(x * Q + off); g.grp[k].prob = (unsigned char)grp[i].prob; g.grp[k].slice = (unsigned char)grp[i].slice;
        off += grp[i].tiles;
        ++k;
      }
    }
    g.ws = reinterpret_cast<float*>(workspace);
    hipLaunchKernelGGL(wg8_tt_group_kernel, dim3(8 * Q), dim3(512), W8_LDS, st, g);
    GPV_CHECK_LAUNCH();
    ++g_wg8_launches;
    if (rk.n > 0) {
      hipLaunchKernelGGL(wgrad_group_reduce_kernel, dim3(rblocks), dim3(256), 0, st, rk);
      GPV_CHECK_LAUNCH();
    }
    g.n_prob = 0; n_grp = 0; rk.n = 0; rblocks = 0; ws_used = 0;
    return 0;
  };
  for (int i = 0; i < n; ++i) {
    const gpv_tt_problem& q = problems[i];
    const bool big = g_w8l_mode != 0 && q.A && q.B && q.C && q.M > 0 && q.N > 0 && q.M % 256 == 0 && q.N % 256 == 0 && q.K >= 2 * TBK && (int64_t)q.M * q.N < (1ll << 30) &&
                     q.lda % 8 == 0 && q.ldb % 8 == 0 && q.ldc % 4 == 0 && al16t(q.A) && al16t(q.B) && al16t(q.C) &&
                     (int64_t)q.K * q.lda < (1ll << 30) && (int64_t)q.K * q.ldb < (1ll << 30);
    if (!big) {
      if (n_rest == GPV_TT_GROUP_MAX) { const int e = flush_rest(); if (e) return e; }
      rest[n_rest++] = q;
      continue;
    }
    const int kt_total = (q.K + TBK - 1) / TBK;
    int split = (kt_total + target_kt / 2) / target_kt;
    if (split < 1) split = 1;
    const int64_t MN = (int64_t)q.M * q.N;
    while (split > 1 && (int64_t)split * MN > ws_floats) --split;
    const int kps = (kt_total + split - 1) / split;
    split = (kt_total + kps - 1) / kps;
    if (split > 255) return (int)hipErrorInvalidValue;
    if (g.n_prob == W8L_PMAX || n_grp + split > W8L_GMAX || ws_used + (int64_t)split * MN >= (1ll << 31) || (split > 1 && (rk.n == WG_MAX || ws_used + (int64_t)split * MN > ws_floats))) {
      const int e = flush();
      if (e) return e;
    }
    W8LProb& d = g.prob[g.n_prob];
    d.A = q.A; d.B = q.B; d.C = q.C; d.a_rowsum = q.a_rowsum;
    d.MN = (int)MN; d.N = q.N; d.K = q.K; d.lda = q.lda; d.ldb = q.ldb; d.ldc = q.ldc;
    d.kt_per_split = kps; d.ws_off = split == 1 ? -1 : (int)ws_used;
    const int tiles = (q.M / 256) * (q.N / 256);
    for (int sidx = 0; sidx < split; ++sidx) {
      const int len = (sidx + 1) * kps <= kt_total ? kps : kt_total - sidx * kps;
      grp[n_grp++] = Grp{g.n_prob, sidx, tiles, len};
    }
    if (split > 1) {
      WgRed& r = rk.r[rk.n];
      r.ws = reinterpret_cast<const float*>(workspace) + ws_used; r.C = q.C; r.MN = MN; r.split = split; r.N = q.N; r.ldc = q.ldc;
      r.blk_start = rblocks;
      rblocks += (int)((MN / 4 + 255) / 256);
      ++rk.n;
      ws_used += (int64_t)split * MN;
    }
    ++g.n_prob;
  }
  int e = flush();
  if (e) return e;
  return flush_rest();
}
