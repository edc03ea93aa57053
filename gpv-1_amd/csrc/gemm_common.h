// Shared between gemm.hip (register-staged 4-wave kernel, every mode incl. fp32 "precise") and gemm_glds.hip
// (8-wave direct-to-LDS kernel for the large bf16 GEMM / conv forward / dgrad launches).
#pragma once
#include "common.h"
#include "../../include/gpv_hip.h"

namespace gpvk {

enum { OP_PLAIN = 0, OP_TRANS = 1, OP_CONV = 2 };

struct ConvGeom {
  int IH, IW, Cs, Cin, OH, OW, KH, KW, SH, SW, PH, PW, dgrad;
  int cm;          // stride-2 dgrad: GEMM rows are ordered parity-class-major (4 classes of (OH/2)*(OW/2) pixels per image)
  int cls_rows;    // rows per parity class = batch * (OH/2) * (OW/2)
};

struct GemmK {
  const void* A; const void* B; void* C;
  int M, N, K;
  int64_t lda, ldb, ldc, sA, sB, sC;
  float alpha;
  const float* rowscale; const float* bias;
  const void* res; int64_t ldr, sR;
  const void* mask; int64_t ldm;
  int act; uint32_t dthresh; float dscale; uint64_t seed; const uint64_t* seed_dev;
  int accumulate, split_k, kt_per_split, tilesN;
  int vecA, vecB;
  int conv1x1;         // launched from gpv_conv2d as a plain GEMM: use the conv1x1_kernel name
  int depi;            // gemm_glds.hip: register epilogue over permuted output columns (set by launch_glds)
  int no_small;        // host side: gpv_gemm_args.flags & GPV_GEMM_NO_PIPE_SMALL (pipe_try_launch skips its small-M configurations for this call)
  int nt_io;           // epilogue stores / residual loads non-temporal (outputs far larger than the 256 MB MALL: layer1's 314 MB maps)
  void* ws_base; int64_t ws_bytes;   // host side: caller workspace for split reductions
  float* ws;                         // kernel side: != NULL -> split s writes its partial product to ws[s][M][N]
  const uint32_t* mask_bits;   // conv1x1_stream.hip: the ReLU mask as ONE BIT per element, word [row][n >> 5] bit n & 31 (read instead of `mask`, which is then only non-NULL)
  uint32_t* out_bits;          // conv1x1_stream.hip: (output > 0) of a ReLU forward written in the same layout
  float* a_rowsum;     // TRANS x TRANS only: a_rowsum[m] += sum_k A[m,k]  (bias gradient fused into the weight-gradient GEMM)
  ConvGeom cg;
};

// GEMM row -> output pixel (row of the NHWC output) ; identity unless the rows are parity-class-major
__device__ __forceinline__ int conv_row_to_pixel(int m, const ConvGeom& g) {
  if (!g.cm) return m;
  const int cls = m / g.cls_rows, rem = m - cls * g.cls_rows;
  const int hw2 = (g.OH >> 1) * (g.OW >> 1), w2 = g.OW >> 1;
  const int b = rem / hw2, r2 = rem - b * hw2;
  const int y2 = r2 / w2, x2 = r2 - y2 * w2;
  return (b * g.OH + 2 * y2 + (cls >> 1)) * g.OW + 2 * x2 + (cls & 1);
}


// gemm_glds.hip: launches the 8-wave kernel when the problem qualifies.  Returns 0 = launched, -1 = not applicable
// (caller falls back to the 4-wave kernel), > 0 = hipError_t.
int glds_try_launch(const GemmK& k, int amode, int dtype_in, int dtype_out, int batch, hipStream_t st);

// gemm_pipe.hip: three-stage pipelined direct-to-LDS kernel, 8 waves, one block per CU, tile height picked per problem (bf16 -> bf16).
// Same return convention; tried before glds_try_launch.
int pipe_try_launch(const GemmK& k, int amode, int dtype_in, int dtype_out, int batch, hipStream_t st);
int pipe_set_mode(int v);
int pipe_set_small(int v);     // gpv_set_option(GPV_OPT_PIPE_SMALL, .)
long pipe_launches(long set);

// conv1x1_stream.hip: streaming kernel for the K <= 256 pointwise convolutions over >= 65536 pixels (weights resident in LDS,
// a wave per 16 pixels, register epilogue in a permuted channel order).  Same return convention; tried first on the 1x1 path.
int c1s_try_launch(const GemmK& k, int dtype_in, int dtype_out, hipStream_t st, bool linear = false, bool dry = false);   // linear: called from gpv_gemm (plain rows, alpha / dropout allowed)
extern int g_c1s_mode;
extern long g_c1s_launches;

// conv3x3_stream.hip: streaming kernel for the 3x3 convolutions with 64 / 128 input channels over >= 65536 output pixels (weights
// of 64 output channels resident in LDS, a wave per 32 pixels, 32x32x16 MFMA, no barriers): forward, backward-data stride 1 and
// the stride-2 backward-data parity classes.  Same return convention; tried first on the 3x3 path.
int c3s_try_launch(const GemmK& k, int dtype_in, int dtype_out, hipStream_t st, bool dry = false);
extern int g_c3s_mode;
extern long g_c3s_launches;

// tile height (160 | 96) with which (rows / h) x (N / 128) tiles fill the 512 two-per-CU slots in whole rounds (>= 90 % of the last
// round's slots, more than 384 tiles), 0 = none
extern int g_two_per_cu;
extern int g_kernel_forced;
extern int g_bm96_fill;
inline int two_per_cu_bm(const GemmK& k, int batch) {
  if (k.N % 128 != 0 || batch != 1) return 0;
  if (g_bm96_fill && k.M <= 1024) {
    // round 5, measured (tools/bench_skinny_pf.py, GPV_BM96 in the tuning build): at <= 1024 rows the two-per-CU 96 x 128 direct-to-LDS kernel beats the
    // small-M kernel wherever 96-row tiles are more tiles than 128-row ones -- 640 x 768 x 2048 15.4 -> 10.6 us, 640 x 768 x 768 8.5 -> 6.6,
    // 100 x 768 x 768 7.9 -> 6.5 -- and loses on the 3200- / 9600-row GEMMs (9600 x 256 x 2048 23.6 -> 25.1, 3200 x 768 x 3072 32.1 -> 33.3)
    const int64_t t128 = (int64_t)((k.M + 127) / 128) * (k.N / 128), t96 = (int64_t)((k.M + 95) / 96) * (k.N / 128);
    if (t96 <= 256 && t96 * 10 >= t128 * 12) return 96;
  }
  int best = 0;
  double best_u = 0.0;
  for (int bm : {160, 96}) {
    const int64_t t = (int64_t)((k.M + bm - 1) / bm) * (k.N / 128);
    if (t <= 384) continue;
    const double u = (double)t / (512.0 * ((t + 511) / 512));
    if ((t <= 512 || u >= 0.9) && u > best_u) { best = bm; best_u = u; }     // one round: every tile is resident at once, whatever the fill
  }
  return best;
}
inline bool glds_two_per_cu(const GemmK& k, int batch) {
  return g_two_per_cu && k.K >= 512 && (!k.cg.cm || k.cg.KH == 1) && two_per_cu_bm(k, batch) != 0;      // (K = 256: measured, no gain)
}

// gemm_skinny.hip: 64x64 tiles with the reduction split across the block's four waves, for GEMMs whose tiles cannot
// fill the chip (M = 192..640 rows).  Same return convention.
int skinny_try_launch(const GemmK& k, int b_trans, int dtype_in, int dtype_out, int batch, hipStream_t st);
extern int g_skinny_mode;

// gemv.hip: M <= 8 rows (the greedy decode step at small batch): a wave per 1..4 output columns, no LDS.  Same return convention.
int gemv_try_launch(const GemmK& k, int dtype_in, int dtype_out, int batch, hipStream_t st);
extern int g_gemv_mode;
extern long g_gemv_launches;

// gemm.hip: second pass of a workspace split reduction, C[m,n] += sum_s ws[s][m][n]
int launch_splitk_reduce(const float* ws, int split, int M, int N, float* C, int64_t ldc, hipStream_t st);
// gemm_skinny.hip: weight-gradient form (both operands reduction-major), fp32 accumulate into C, optional a_rowsum
int skinny_tt_try_launch(const GemmK& k, int dtype_in, int dtype_out, int batch, hipStream_t st);

// gemm_glds_tt.hip: direct-to-LDS conv weight gradient (two-pass split reduction through k.ws_base).  Same return convention.
int glds_wgrad_try_launch(const GemmK& k, int dtype_in, int dtype_out, hipStream_t st);
int glds_tt_try_launch(const GemmK& k, int dtype_in, int dtype_out, int batch, hipStream_t st);   // linear form (+ a_rowsum)
extern int g_wgrad_mode;
extern int g_wg8_mode;
extern long g_wg8_launches;
extern int g_w8l_mode;
extern int g_wg8h_mode;

}  // namespace gpvk
