/* gpv_hip.h -- C ABI of libgpv_hip.so: hand-written HIP kernels (gfx950 / CDNA4) for the GPV-1
 * encoder-decoder hot path.
 *
 * The reference (allenai/gpv-1) is 100 % Python and has no FFI/plugin layer: every device kernel
 * it runs is implicit (cuDNN / cuBLAS / torchvision / apex through torch 1.6).  The boundary this
 * library replaces is therefore "the torch op the reference calls at file:line"; each entry point
 * below cites that call site (paths relative to the reference checkout).  INTEGRATION.md shows the
 * ctypes stub a maintainer of the reference would add to route those call sites here.
 *
 * Conventions
 *   - every function returns hipError_t as int (0 = success); never throws, never allocates,
 *     no global state; work is enqueued asynchronously on `stream` (a hipStream_t passed as void*).
 *   - all pointers are DEVICE pointers; sizes / leading dimensions are in ELEMENTS.
 *   - dtype codes: GPV_BF16 = 0 (bfloat16, fp32 accumulate on MFMA), GPV_F32 = 1
 *     ("precise" mode: fp32 in HBM, operands split hi+lo bf16 on the fly, 3 MFMAs per product,
 *     ~1e-6 relative error; used for parity tests and for the fp32 box/class heads).
 *   - matrices are row-major.  Activations are (rows = tokens/pixels, cols = channels); images /
 *     feature maps are NHWC.
 */
#ifndef GPV_HIP_H
#define GPV_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define GPV_BF16 0
#define GPV_F32 1

#define GPV_ACT_NONE 0
#define GPV_ACT_RELU 1
#define GPV_ACT_GELU 2 /* erf form, vilbert.py:111-117 */

/* operand storage for gpv_gemm */
#define GPV_KMAJOR 0 /* element (row r, red k) at ptr[r*ld + k]  (nn.Linear weight, activations) */
#define GPV_TRANS 1  /* element (row r, red k) at ptr[k*ld + r]  (reduction index is the slow dim) */

int gpv_abi_version(void); /* = 1 */
/* writes the NUL-terminated build id (sha256 prefix of the sources the library was compiled from: csrc/Makefile BUILD_ID) into the
 * HOST buffer `buf` of `cap` bytes; 0 = ok.  No reference counterpart: it lets a loader refuse a library that is older than the
 * tree next to it (__graft_entry__.build()). */
int gpv_build_id(char* buf, int cap);

/* ---------------------------------------------------------------------------------------------
 * Grouped weight-gradient GEMM:  for every problem i:  C_i[M_i, N_i] += A_i^T B_i   (fp32 accumulate into C)
 *   A_i: K_i x M_i (dY rows, row pitch lda), B_i: K_i x N_i (layer input rows, row pitch ldb), both bf16 and reduction-major;
 *   a_rowsum_i (optional): += column sums of A_i (the bias gradient).  M_i, N_i multiples of 128; lda, ldb multiples of 8;
 *   ldc multiple of 4; 16-byte aligned bases.  One launch per GPV_TT_GROUP_MAX problems, one workgroup per 128x128 output
 *   tile walking the whole reduction: no split, no workspace.  `problems` is HOST memory (copied into the kernel arguments).
 *   Replaces the per-layer weight-gradient halves of loss.backward() (exp/gpv/train_distr.py:421: every nn.Linear of the DETR
 *   transformer, the co-attention layers and the text decoder) where the caller has all their operands at once. */
#define GPV_TT_GROUP_MAX 48
typedef struct {
  const void* A; const void* B; float* C; float* a_rowsum;
  int M, N, K;
  int lda, ldb, ldc;
} gpv_tt_problem;
int gpv_gemm_tt_group(const gpv_tt_problem* problems, int n, void* stream);
/* The same products with a caller-lent workspace (fp32 partial tiles of the reductions the library slices; workspace may be NULL:
 * no slicing): the problems whose M and N are multiples of 256 run on 256 x 256 tiles with the eight-phase schedule
 * (gemm_glds_tt.hip wg8_*), long reductions cut into slices so that every (problem, slice, tile) unit walks <= ~64 k-tiles of 64
 * rows and the units of all problems fill the chip together; the rest goes through gpv_gemm_tt_group.  Same reference call sites. */
int gpv_gemm_tt_group_ws(const gpv_tt_problem* problems, int n, void* workspace, int64_t workspace_bytes, void* stream);

/* Kernel-selection knob (process-wide; tests and tuning only; never changes results beyond fp32 summation order).
 *   option GPV_OPT_GLDS: 0 = 4-wave register-staged GEMM/conv kernel only, 1 (default) = use the direct-to-LDS
 *   kernels where they are expected to win, 2 / 3 = the 8-wave 256-row / 4-wave 128x128 variant wherever it is legal.
 *   Returns the previous value, or -1 for an unknown option.  (No reference counterpart: the reference delegates kernel choice to cuDNN/cuBLAS.) */
#define GPV_OPT_GLDS 0
#define GPV_OPT_SKINNY 2 /* small-M GEMM kernel (reduction split over the block's waves): 0 never, 1 (default) heuristic, 2 wherever legal */
#define GPV_OPT_GLDS_LAUNCHES 1 /* returns the number of direct-to-LDS GEMM/conv launches so far, then sets the counter to value */
#define GPV_OPT_GLDS_WGRAD 3 /* direct-to-LDS weight-gradient kernel: 0 never, 1 (default) conv wherever legal + linear where it wins, 2 both wherever legal */
#define GPV_OPT_PIPE 4 /* three-stage pipelined direct-to-LDS GEMM/conv kernel (gemm_pipe.hip): 0 never, 1 (default) heuristic, 100 + i = tile configuration i wherever legal */
#define GPV_OPT_C1S 6 /* streaming kernel for the K <= 256 1x1 convolutions (conv1x1_stream.hip): 0 never, 1 (default) >= 65536 pixel rows, 2 wherever legal */
#define GPV_OPT_C3S 7 /* streaming kernel for the 3x3 convolutions with 64 / 128 input channels (conv3x3_stream.hip): 0 never, 1 (default) >= 65536 output pixels, 2 wherever legal */
#define GPV_OPT_C3S_LAUNCHES 8 /* returns the number of streaming-3x3 launches so far, then sets the counter to value */
#define GPV_OPT_GEMV 9 /* few-row kernel (gemv.hip) for M <= 8: 0 never, 1 (default) wherever legal */
#define GPV_OPT_GEMV_LAUNCHES 10 /* returns the number of few-row launches so far, then sets the counter to value */
#define GPV_OPT_PIPE_LAUNCHES 5 /* returns the number of pipelined-kernel launches so far, then sets the counter to value */
#define GPV_OPT_ATTN_BWD1 11 /* single-launch attention backward (bf16; attention.hip attn_bwd1_kernel): 0 never, 1 (default) when B * H >= 128, 2 wherever legal */
#define GPV_OPT_ATTN_BWD1_LAUNCHES 12 /* returns the number of single-launch attention backwards so far, then sets the counter to value (value >= 0) */
#define GPV_OPT_WG8 14 /* eight-phase 256 x 256 weight-gradient kernel (gemm_glds_tt.hip wg8_*) in gpv_conv_wgrad_group, problems with Cout, Cin multiples of 256: 0 never, 1 (default) when the call holds >= 128 such work units, 2 wherever legal */
#define GPV_OPT_WG8_LAUNCHES 15 /* returns the number of eight-phase weight-gradient launches (conv and linear) so far, then sets the counter to value */
#define GPV_OPT_PIPE_SMALL 17 /* gemm_pipe.hip's small-M configurations (64 x 64 / 32 x 64 tiles, 6 / 8 stages): 1 (default) by heuristic, 0 never -- the inference paths switch them off
                                 (greedy batch 64: 15.0 -> 14.05 ms per batch; the training step's backward shapes prefer them: B1 +0.1 ms without) */
#define GPV_OPT_WG8H 18 /* the eight-phase kernel on 128 x 256 | 256 x 128 tiles (gemm_glds_tt.hip wg8h_*) in gpv_conv_wgrad_group, problems with Cout, Cin multiples of 128 that the 256 x 256 launch does not take (layer2): 1 (default) wherever legal, 0 never */
#define GPV_OPT_C3_HALO 19 /* stride-1 3x3 convolutions (forward / backward-data) of the two-blocks-per-CU tile kernel with ONE halo image per channel block instead of nine tap tiles (gemm_glds.hip glds_halo_kernel; image width <= 47, Cin % 64 == 0, N % 128 == 0): 0 never, 1 (default) the 160-row tiles, 2 the 96-row tiles as well (no gain there: tests) */
#define GPV_OPT_C3_HALO_LAUNCHES 20 /* returns the number of halo-image launches so far, then sets the counter to value */
#define GPV_OPT_W8L 16 /* the same kernel in gpv_gemm_tt_group_ws (problems with M, N multiples of 256): 0 (default) never, 1 wherever legal */
#define GPV_OPT_C1S_LAUNCHES 13 /* returns the number of streaming-1x1 launches so far (convolutions and the K = 256 linear GEMMs), then sets the counter to value */
int gpv_set_option(int option, int value);

/* ---------------------------------------------------------------------------------------------
 * GEMM with fused epilogue:   C[b] = epi( alpha * A[b] x B[b]^T )        b = 0..batch-1
 *   A: M x K (layoutA), B: N x K (layoutB), C: M x N row-major (ldc), dtype_out.
 *   epi (in this order): * rowscale[m] -> + bias[n] -> + res[m,n] -> act -> dropout(p, seed)
 *                        -> * (relu_mask[m,n] > 0) -> store  (or atomic += when accumulate != 0)
 *   split_k > 1 requires accumulate = 1, dtype_out = GPV_F32 and a linear epilogue.
 * Replaces: nn.Linear forward/backward everywhere (transformer.py:131-135, vilbert.py:748-766,
 *   gpv.py:69-88, answer_head.py:31-33), torch.matmul, 1x1 Conv2d (detr_roi_head.py:39),
 *   and (through the separable form, see gpv_roi_weights) torchvision.ops.roi_align+mean
 *   (detr_roi_head.py:44-56).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  const void* A; const void* B; void* C;
  int M, N, K, batch;
  int64_t lda, ldb, ldc;          /* leading dims (elements) */
  int64_t sA, sB, sC;             /* batch strides (elements) */
  int layoutA, layoutB;           /* GPV_KMAJOR / GPV_TRANS */
  int dtype_in, dtype_out;        /* A,B dtype ; C (and res) dtype */
  float alpha;
  const float* rowscale;          /* [M] or NULL */
  const float* bias;              /* [N] or NULL */
  const void* res; int64_t ldr, sR; /* [M,N] dtype_out or NULL */
  const void* relu_mask; int64_t ldm; /* bf16/f32 (dtype_out) [M,N] or NULL */
  int act;
  float drop_p; uint64_t seed;    /* inverted dropout after act; drop_p = 0 disables */
  int accumulate, split_k;
  void* workspace; int64_t workspace_bytes; /* optional scratch for accumulate != 0 with split_k > 1: when it holds
                                     split_k * M * N * 4 bytes the splits write partial products there and a second
                                     pass adds them to C (no fp32 atomics on C).  Contents are don't-care before and
                                     after; one workspace may be shared by all launches of ONE stream. */
  float* a_rowsum;                /* layoutA = layoutB = GPV_TRANS only, or NULL: a_rowsum[m] += sum_k A[m,k]  (atomic).
                                     Weight-gradient GEMMs pass dY as A, so this is the bias gradient
                                     (sum over tokens), fused instead of a separate gpv_colsum launch. */
  int flags;                      /* GPV_GEMM_KPAD_FINITE: every row of a GPV_KMAJOR operand is readable and holds finite values
                                     (zero padding) up to the next multiple of 8 elements of K -- lets a K that is not a multiple
                                     of 8 (RoI pooling: K = H*W = 300 of a 320-pitch weight row) use the 16-byte load path */
} gpv_gemm_args;
#define GPV_GEMM_KPAD_FINITE 1
#define GPV_GEMM_NO_PIPE_SMALL 2  /* this call never takes gemm_pipe.hip's small-M configurations (what GPV_OPT_PIPE_SMALL = 0 does process-wide):
                                     the inference paths pass it per call (round 6, ADVICE r5: the process-wide switch raced with launches of other threads) */
int gpv_gemm(const gpv_gemm_args* a, void* stream);

/* ---------------------------------------------------------------------------------------------
 * NHWC convolution as implicit GEMM (ResNet-50 body, backbone.py:93-95 + FrozenBatchNorm2d
 * backbone.py:44-54: its per-channel scale is folded into the WEIGHT copy the caller passes
 * (gpv_cast_rowscale_t: w[co, :] *= scale[co]; for the weight gradient it is `rowscale`, see
 * mode 2), its shift is `bias`; ReLU and the bottleneck residual fused).
 *   mode 0 forward : y[b,oh,ow,co] = epi( sum_{r,s,ci} x[b, oh*SH+r-PH, ow*SW+s-PW, ci] w[co,r,s,ci] )
 *   mode 1 dgrad   : dx[b,ih,iw,ci] = epi( sum_{r,s,co} dy[b,(ih+PH-r)/SH,(iw+PW-s)/SW,co] wd[ci,r,s,co] )
 *                    (taps whose division is inexact / out of range contribute 0)
 *   mode 2 wgrad   : dw[co,r,s,ci] += rowscale[co] * sum_{b,oh,ow} dy[b,oh,ow,co] x[b,oh*SH+r-PH,..,ci]
 *                    (fp32 atomics, split over the pixel dimension)
 * `x` may have a pixel stride Cs != Cin (the 7x7 stem reads 8-pixel x 4-channel runs of a
 * zero-padded NHWC4 image as one 32-wide "channel" chunk).
 * epilogue as in gpv_gemm (bias = folded BN bias, res = identity branch, act = ReLU; for dgrad:
 * res = gradient arriving through the identity branch, relu_mask = the saved block output).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  int mode;
  const void* x; const void* w; void* y;   /* roles per mode: (x,w,y) / (dy,wd,dx) / (x,dy,dw) */
  int B, IH, IW, Cs, Cin;                  /* gathered tensor: B x IH x IW pixels, pixel stride Cs, Cin used */
  int OH, OW, Cout;                        /* the other tensor: B x OH x OW x Cout */
  int KH, KW, SH, SW, PH, PW;
  int dtype_in, dtype_out;
  const float* rowscale; const float* bias; /* EXTENTS (plain pointers: the library cannot check them, a short vector is an out-of-bounds read):
                                               rowscale is indexed by the output ROW as in gpv_gemm -- modes 0 / 1: B * OH * OW floats, one per output
                                               PIXEL (not per channel; the ResNet path passes NULL), mode 2: Cout floats (the rows of dw);
                                               bias by the output column -- modes 0 / 1: Cout floats; either may be NULL */
  const void* res; const void* relu_mask;  /* same shape as y */
  int act;
  int split_k;                             /* wgrad only */
  void* workspace; int64_t workspace_bytes; /* optional split-reduction scratch, same contract as gpv_gemm_args: wgrad, and forward
                                               convolutions over <= 8192 output pixels (fp32 slabs + a second pass with the epilogue) */
  /* Round 6: ReLU masks as ONE BIT per element (both optional; NULL = as before).  The backward pass needs a post-ReLU activation only
     as "was this element > 0" (the dgrad epilogue's * (relu_mask > 0)); read as bf16 that is 16 bits per element of the widest tensors
     of a bottleneck (157 MB per layer2 block at B = 32: a third of such a launch's bytes).  Both: Cout / 8 bytes per pixel, Cout a multiple
     of 256, (y[pixel, c] > 0) of the bf16 value as stored = bit (c & 7) of byte  32 (c / 256) + 8 ((c % 32) / 8) + (c % 256) / 32  of the pixel's
     row (the streaming kernel's accumulator order inside a 256-channel group: neither side needs a cross-lane operation); 16-byte aligned.
     Cout = 128 (the stride-2 3x3 backward-data of layer2's first block reads them; gpv_conv1x1_chain_bits writes them): 16 bytes per pixel,
     bit (c & 7) of byte  4 ((c % 32) / 8) + c / 32.
       y_mask_bits    (mode 0, act = GPV_ACT_RELU): the launch ALSO writes the bits of its output;
       relu_mask_bits (mode 1): read INSTEAD of relu_mask (relu_mask may then be NULL); results are bit-identical to the bf16 mask's.
     Only the streaming kernels serve them (pointwise stride-1 forward / backward-data; relu_mask_bits also the 3x3 stride-2 backward-data
     over 128 channels): gpv_conv2d_mask_bits_ok() tells whether a call would be served, gpv_conv2d returns hipErrorInvalidValue (nothing
     launched) for one that would not. */
  void* y_mask_bits; const void* relu_mask_bits;
} gpv_conv_args;
int gpv_conv2d(const gpv_conv_args* a, void* stream);
/* 1 when gpv_conv2d would serve these arguments' y_mask_bits / relu_mask_bits (same checks, nothing launched; pointers must be the real ones:
   alignment counts), else 0 */
int gpv_conv2d_mask_bits_ok(const gpv_conv_args* a);

/* All conv weight gradients of a backward pass in one call (the 42 trainable convolutions of ResNet-50 layer2-4,
 * exp/gpv/models/backbone.py:61-63: what autograd's 42 cudnn_convolution_backward_weight calls compute):
 *   dw[co,r,s,ci] += rowscale[co] * sum_{b,oh,ow} dy[b,oh,ow,co] x[b,oh*SH+r-PH,ow*SW+s-PW,ci]      for every problem
 * bf16 operands (x: B x IH x IW pixels of pixel stride Cs; dy: B x OH x OW x Cout), fp32 gradients [Cout][KH][KW][Cin] that are
 * ACCUMULATED into.  The problems run as one or two grids of equal work units (problem, reduction slice, 128 x 128 tile);
 * workspace: scratch for the partial products of the sliced problems (256 MB covers the training shapes), busy until the call's
 * work has run on `stream`.  A problem the grouped kernel does not take (Cout or Cin not a multiple of 128, ...) is issued as
 * its own gpv_conv2d mode-2 launch.  Results equal gpv_conv2d's up to fp32 summation order. */
typedef struct gpv_conv_wgrad_problem {
  const void* x; const void* dy; float* dw; const float* rowscale;
  int B, IH, IW, Cs, Cin, OH, OW, Cout, KH, KW, SH, SW, PH, PW;
} gpv_conv_wgrad_problem;
int gpv_conv_wgrad_group(const gpv_conv_wgrad_problem* problems, int n, void* workspace, int64_t workspace_bytes, void* stream);

/* NCHW fp32 image -> zero-padded NHWC4 (bf16 or f32) with `pad` pixels on every side and the
 * row length rounded up to Wp pixels (nested_tensor images, detr_misc.py:282-299 -> backbone). */
int gpv_image_to_nhwc4(const float* img, void* out, int B, int H, int W, int pad, int Hp, int Wp,
                       int dtype_out, void* stream);
/* 3x3 stride-2 pad-1 max-pool, NHWC (torchvision resnet maxpool). */
int gpv_maxpool3x3s2(const void* x, void* y, int B, int H, int W, int C, int OH, int OW, int dtype,
                     void* stream);
/* Device-side input pipeline (datasets/coco_generic_dataset.py:49-62 resize(.., (480, 640), anti_aliasing=True);
 * datasets/coco_datasets.py:26-38,137-150 (255 img).astype(uint8) -> ColorJitter / RandomHorizontalFlip / RandomGrayscale -> ToTensor
 * -> Normalize): B decoded uint8 HWC images of any size -> the stem's zero-padded NHWC4 batch out[B,Hp,Wp,4] (what
 * gpv_image_to_nhwc4 produces from a normalised fp32 batch).  descs: DEVICE array of B descriptors (the host draws the random
 * parameters); scratch_u8: B*OH*OW*3 bytes; grey_sum: B 32-bit words of scratch (an exact integer sum lives there).  order[] = the four jitter steps in the sample's drawn order
 * (0 brightness, 1 contrast, 2 saturation, 3 hue); jitter = 0 skips them.  Source side length / output side length <= 9. */
typedef struct gpv_image_desc {
  const unsigned char* src;   /* [H][W][3] uint8, device */
  int H, W;
  int flip, gray, jitter;
  int order[4];
  float brightness, contrast, saturation, hue;
  int reserved;
} gpv_image_desc;
int gpv_image_pipeline(const gpv_image_desc* descs, int B, void* scratch_u8, float* grey_sum, void* out, int OH, int OW, int pad,
                       int Hp, int Wp, int dtype_out, void* stream);
/* Baseline JPEG decoding for the device-side input pipeline: `skio.imread(img_path)` of the reference's loader workers
 * (datasets/coco_generic_dataset.py:54, datasets/coco_datasets.py:157, inference_util.py:10 -> Pillow -> libjpeg-turbo defaults:
 * accurate integer IDCT, fancy chroma upsampling, YCbCr -> RGB), bit-exact against that decoder.
 *   gpv_jpeg_parse  (HOST, thread-safe, no GPU work): header + Huffman decoding of one file into quantised coefficient blocks
 *                   coefs[component][block row][block col][64] (int16, natural order; whole MCUs).  coefs == NULL: header pass,
 *                   fills `info` only (coef_count = the int16 capacity the second call needs).
 *                   Returns 0, hipErrorInvalidValue (malformed) or hipErrorNotSupported (801: progressive / arithmetic / 12-bit /
 *                   CMYK / non-interleaved / other sampling factors than luma 1x1, 2x1, 2x2 over 1x1 chroma).
 *   gpv_jpeg_decode (DEVICE, two launches for the whole batch): descs = DEVICE array of B descriptors; per image the coefficients
 *                   (device), a scratch of sum_c bh[c]*8 * bw[c]*8 bytes for the component planes, and the output
 *                   out[height][width][3] uint8 RGB (a single-component file is replicated to three channels, as
 *                   coco_generic_dataset.py:55-56 does) -- the `src` of gpv_image_pipeline.  max_blocks / max_pixels: the largest
 *                   block count (all components) / pixel count of a batch member (grid sizing). */
typedef struct gpv_jpeg_info {
  int width, height, ncomp;            /* ncomp 1 (grey) or 3 (YCbCr) */
  int hmax, vmax;                      /* luma sampling factors (chroma is 1x1) */
  int mcus_x, mcus_y;
  int bh[3], bw[3];                    /* blocks per component */
  int reserved;
  int64_t coef_offset[3];              /* int16 element offset of the component's blocks in `coefs` */
  int64_t coef_count;
  unsigned short quant[3][64];         /* quantisation table per component, natural order */
} gpv_jpeg_info;
typedef struct gpv_jpeg_desc {
  const short* coefs; unsigned char* planes; unsigned char* out;
  int width, height, ncomp, hmax, vmax;
  int bh[3], bw[3];
  int coef_off[3];                     /* int16 element offsets into coefs */
  int plane_off[3];                    /* byte offsets into planes */
  unsigned short quant[3][64];
} gpv_jpeg_desc;
int gpv_jpeg_parse(const unsigned char* data, int64_t nbytes, gpv_jpeg_info* info, short* coefs, int64_t coefs_capacity);
int gpv_jpeg_decode(const gpv_jpeg_desc* descs, int B, int max_blocks, int64_t max_pixels, void* stream);
/* The tail of a stage's first bottleneck in one launch (torchvision Bottleneck.forward with a downsample branch,
 * exp/gpv/models/backbone.py:93-95): y = act(conv3(a1) + downsample(a2 at stride s2) + bias), both pointwise, FrozenBN scales folded
 * into w1 [N,K1] / w2 [N,K2], bias = the two shifts added.  bf16; (K1, K2, N) in {(64, 64, 256), (128, 256, 512)} (layer1 / layer2);
 * hipErrorNotSupported (801) for anything else -- the caller then issues the two convolutions. */
int gpv_conv1x1_dual(const void* a1, const void* w1, const void* a2, const void* w2, const float* bias, void* y, int B, int OH, int OW,
                     int K1, int IH2, int IW2, int K2, int s2, int N, int act, void* stream);
/* the same launch, also writing y_mask_bits = (y > 0) as one bit per element in gpv_conv_args.y_mask_bits' layout (round 6: the mask
 * of the next bottleneck's conv1 backward-data, backbone.py:61-63 trains layer2): (K1, K2, N) = (128, 256, 512) with ReLU only -- 801
 * otherwise, nothing launched; y_mask_bits == NULL is gpv_conv1x1_dual */
int gpv_conv1x1_dual_bits(const void* a1, const void* w1, const void* a2, const void* w2, const float* bias, void* y, int B, int OH, int OW,
                          int K1, int IH2, int IW2, int K2, int s2, int N, int act, void* y_mask_bits, void* stream);

/* A layer1 bottleneck tail AND the conv1 of the bottleneck that follows, one launch (torchvision Bottleneck.forward twice:
 * exp/gpv/models/backbone.py:93-95; conv1 / layer1 are frozen, :61-63, so nothing in between is needed by a backward pass):
 *   y[B,OH,OW,256] = relu( a1[B,OH,OW,K1] . w1[256,K1]^T (+ a2[B,IH2,IW2,K2] sampled at stride s2 . w2[256,K2]^T) (+ res) + bias )
 *   z[B,OH,OW,N2]  = relu( y . wn[N2,256]^T + bias_n )
 * bf16 operands and outputs, fp32 biases (FrozenBN shifts; the scales are folded into the weights).  The second GEMM consumes the
 * rounded y out of registers: y is written once and not read back (314 MB at B = 32 in layer1).  Bit-identical to the two
 * launches.  K1 = 64, K2 in {0, 64} (a2 = w2 = NULL when 0), N = 256, N2 in {64, 128}; res (identity branch) and a2 (downsample
 * branch) are mutually exclusive; anything else: hipErrorNotSupported (801), the caller then launches the convolutions separately. */
int gpv_conv1x1_chain(const void* a1, const void* w1, int K1, const void* a2, const void* w2, int K2, int IH2, int IW2, int s2,
                      const void* res, const float* bias, void* y, int B, int OH, int OW, int N, const void* wn,
                      const float* bias_n, void* z, int N2, void* stream);
/* the same launch, also writing z_mask_bits = (z > 0) as one bit per element (gpv_conv_args.y_mask_bits' layout for 128 channels; round 6):
 * the identity-branch form with N2 = 128 only (layer1's last tail + layer2.0's conv1) -- 801 otherwise, nothing launched; NULL is gpv_conv1x1_chain */
int gpv_conv1x1_chain_bits(const void* a1, const void* w1, int K1, const void* a2, const void* w2, int K2, int IH2, int IW2, int s2,
                      const void* res, const float* bias, void* y, int B, int OH, int OW, int N, const void* wn,
                      const float* bias_n, void* z, int N2, void* z_mask_bits, void* stream);
#ifdef GPV_TUNING
/* TUNING BUILD ONLY (libgpv_hip_tuning.so, `make -C gpv-1_amd/csrc tuning`; not part of the production ABI): built, correct and not faster
 * than the three launches it replaces -- DESIGN.md section 0 / 8 -- kept compilable for the next attempt. */
/* The post-norm feed-forward sub-layer of a DETR encoder / decoder layer in ONE launch (exp/gpv/models/transformer.py:156-160
 * encoder, :226-231 decoder: src + dropout2(linear2(dropout(relu(linear1(src))))) -> norm2):
 *   h[M,F]   = dropout(relu(x[M,D] . w1[F,D]^T + b1))        stored: the backward's ReLU mask and the dW2 operand
 *   y[M,D]   = h . w2[D,F]^T + b2                            stored (bf16): the LayerNorm backward recomputes x + dropout(y)
 *   out[M,D] = LayerNorm(x + dropout(y)) * gamma + beta ;  mean[M], rstd[M] fp32 ;  out2 = out + pos[row % pos_rows] (optional, both
 *              or neither of pos / out2, as gpv_layernorm_pos_fwd)
 * replaces gpv_gemm (ReLU + dropout epilogue), gpv_gemm, gpv_layernorm_pos_fwd: the [M,F] activation is not read back from HBM, the
 * second product consumes it from the first one's accumulator registers.  bf16 x / w1 / w2 / h / y / out / pos, fp32 the rest; one
 * drop_p for both dropouts, masks = (seed1 | seed2, flat element index) exactly as the replaced launches draw them.  Same rounding
 * points as the three launches (h, y rounded to bf16), the F-sum split in two halves.  D = 256, F % 64 == 0, 16-byte aligned pointers;
 * anything else: hipErrorNotSupported (801) and the caller launches the three kernels. */
int gpv_ffn_fused_fwd(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, const float* gamma,
                      const float* beta, void* h, void* y, void* out, float* mean, float* rstd, int M, int D, int F, float eps,
                      float drop_p, uint64_t seed1, uint64_t seed2, const void* pos, int pos_rows, void* out2, void* stream);
#endif
/* The whole ResNet stem in one launch (exp/gpv/models/backbone.py:93-95 -> torchvision resnet50 conv1 + bn1 (frozen: scale folded
 * into w, shift here) + relu + maxpool):  y[B,PH,PW,64] = maxpool3x3s2p1(relu(conv7x7s2(x) + shift)).  x = the zero-padded NHWC4
 * bf16 image gpv_image_to_nhwc4 writes with pad 3 ([B,Hp,Wp,4], Wp even, >= 2 (CW - 1) + 8), w = [64][7][8 px][4 ch] bf16 (8th pixel /
 * 4th channel zero), CH x CW = conv map, PH x PW = pooled map.  bf16 only; the conv map is never written. */
int gpv_stem_pool(const void* x, const void* w, const float* shift, void* y, int B, int Hp, int Wp, int CH, int CW, int PH, int PW,
                  void* stream);

/* ---------------------------------------------------------------------------------------------
 * Multi-head attention core  O = dropout(softmax(Q K^T * scale + masks)) V   per (batch, head).
 * Q/K/V/O are addressed as ptr[b*bs + t*rs + h*dh + d] so they can be slices of fused projection
 * buffers.  Sk <= 320.  dh in {32,48,64,96}.  kpm: uint8 [B,Sk] (1 = ignore key) or NULL.
 * causal: key j > query i masked.  lse: fp32 [B,H,Sq] (saved for backward).
 * Range: Sk <= 320; a head's K and V (backward: + Q) stay in LDS -- in GPV_F32 ("precise": hi + lo bf16 pairs) that caps
 * Sk x dh at about 300 x 32 / 190 x 64 / 120 x 96 (more than covers the model's shapes); beyond it the call returns
 * hipErrorInvalidValue before anything is launched.  A row whose keys are ALL masked has no defined output.
 * Replaces nn.MultiheadAttention's core (transformer.py:153-155,218-226; gpv.py:38-43), the
 * two co-attention products of BertBiAttention (vilbert.py:770-810) and BERT self-attention.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  const void* q; const void* k; const void* v; void* o;
  int64_t q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs;
  int B, H, Sq, Sk, dh;
  float scale;
  const uint8_t* kpm; int causal;
  float drop_p; uint64_t seed;
  float* lse;
  int dtype;                       /* GPV_BF16 / GPV_F32 (precise) for q,k,v,o (+ grads) */
  /* backward only */
  const void* dout; int64_t do_bs, do_rs;
  void* dq; void* dk; void* dv;    /* same addressing as q / k / v */
} gpv_attn_args;
int gpv_attention_fwd(const gpv_attn_args* a, void* stream);
int gpv_attention_bwd(const gpv_attn_args* a, void* stream);
/* Self-attention with nn.MultiheadAttention's in-projection INSIDE the launch (transformer.py:148-155: q = k = src + pos, value = src;
 * :216-219 the decoder's self-attention; torch F.multi_head_attention_forward's in_proj):
 *   q = xp Wq^T + bq,  k = xp Wk^T + bk,  v = x Wv^T + bv,  o = dropout(softmax(q k^T scale + key padding)) v
 * a->q / a->k / a->v are OUTPUTS here: the projected rows, written where the projection GEMMs would have written them (addressing as
 * above) -- gpv_attention_bwd and the backward GEMMs read them; a->o, a->lse as in gpv_attention_fwd.  xp, x: [B, Sq, 256] rows
 * (x_bs / x_rs in elements, multiples of 8), w: in_proj_weight [768, 256] bf16 contiguous, bias: [768] fp32 or NULL.
 * bf16 only, dh = 32, H * dh = 256, Sq == Sk <= 320, not causal; anything else: hipErrorInvalidValue before anything is launched. */
int gpv_attention_qkv_fwd(const gpv_attn_args* a, const void* xp, const void* x, int64_t x_bs, int64_t x_rs, const void* w,
                          const float* bias, void* stream);

/* ---------------------------------------------------------------------------------------------
 * y = LayerNorm(x + dropout(s)) * gamma + beta     (post-norm residual blocks:
 * transformer.py:156-160,227-231; BertBiOutput/BertOutput vilbert.py:845-856,510-516;
 * F.layer_norm without affine detr_roi_head.py:91 when gamma == NULL; s == NULL -> plain LN).
 * Saves mean / rstd (fp32 [rows]) for backward.  Backward returns dx (gradient w.r.t. x, also the
 * gradient w.r.t. the pre-dropout s when drop_p == 0), ds (only written when drop_p > 0), and
 * accumulates dgamma / dbeta (fp32 atomics).
 * ------------------------------------------------------------------------------------------- */
int gpv_layernorm_fwd(const void* x, const void* s, const float* gamma, const float* beta, void* y,
                      float* mean, float* rstd, int rows, int cols, float eps, float drop_p,
                      uint64_t seed, int dtype, void* stream);
int gpv_layernorm_bwd(const void* dy, const void* x, const void* s, const float* gamma,
                      const float* mean, const float* rstd, void* dx, void* ds, float* dgamma,
                      float* dbeta, int rows, int cols, float drop_p, uint64_t seed, int dtype,
                      void* stream);
/* The same LayerNorm with a second output  y2 = y + pos[row % pos_rows]  (pos: [pos_rows, cols] in the activation dtype, rows a
 * multiple of pos_rows; pos == y2 == NULL: gpv_layernorm_fwd): the sums `src + pos` / `tgt + query_pos` that DETR's layers feed their
 * q / k projections (transformer.py:150-151,216-217,221-222) leave the LayerNorm that produces src / tgt instead of being an
 * element-wise launch each; y2 is computed from the stored (rounded) y, bit-identical to gpv_add(y, pos).
 * gpv_layernorm_bwd2: dy2 (nullable) is a second gradient of the same output -- the gradient that came back through y2 -- summed with
 * dy on load. */
int gpv_layernorm_pos_fwd(const void* x, const void* s, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                          int rows, int cols, float eps, float drop_p, uint64_t seed, const void* pos, int pos_rows, void* y2,
                          int dtype, void* stream);
int gpv_layernorm_bwd2(const void* dy, const void* dy2, const void* x, const void* s, const float* gamma, const float* mean,
                       const float* rstd, void* dx, void* ds, float* dgamma, float* dbeta, int rows, int cols, float drop_p,
                       uint64_t seed, int dtype, void* stream);

/* The attention sublayer's out-projection AND the LayerNorm behind it in one launch (transformer.py:156-157 norm1(src + dropout1(src2)),
 * 218-219 / 224-226 in the decoder; nn.MultiheadAttention's out_proj):
 *   s = a w^T + bias  (rounded to bf16, WRITTEN: the backward -- gpv_layernorm_bwd3, the projection's weight gradient and backward-data
 *   GEMM -- reads it),  y = LayerNorm(x + dropout(s)) * gamma + beta,  y2 = y + pos[row % pos_rows]  (pos == y2 == NULL: no second output).
 * a, x, s, y, y2, pos: bf16 [rows, 256] contiguous; w: [256, 256] bf16 (rows = output features); bias, gamma, beta: fp32 [256] (gamma ==
 * beta == NULL: no affine); mean / rstd: fp32 [rows].  Same dropout words as gpv_layernorm_fwd for (seed, row * 256 + column).
 * K == N == 256 only (hipErrorInvalidValue otherwise, before anything is launched). */
int gpv_linear_layernorm_fwd(const void* a, const void* w, const float* bias, const void* x, const float* gamma, const float* beta,
                             void* s, void* y, float* mean, float* rstd, int rows, int K, int N, float eps, float drop_p, uint64_t seed,
                             const void* pos, int pos_rows, void* y2, void* stream);
/* gpv_layernorm_bwd3: the same backward with the column sums taken OFF the launch.  partials != NULL (then dgamma == dbeta == NULL):
 * every workgroup stores its partial [dgamma | dbeta] row to partials[block][2 * cols] (fp32; gpv_layernorm_bwd_blocks(rows, cols)
 * rows -- the grid this library launches for the shape, -1 for a shape it refuses) instead of adding it to dgamma / dbeta with
 * fp32 atomics; gpv_colsum_fold_group adds the rows of a GROUP of such buffers to their dgamma / dbeta in one launch, in a fixed
 * order (reproducible), wherever the caller puts it -- the gradient of a LayerNorm's affine parameters is needed by nobody until
 * the optimizer (reference: torch.nn.LayerNorm's backward inside transformer.py:156-160,227-231, vilbert.py:845-856). */
typedef struct gpv_fold_problem {
  const float* partials;          /* [nblk][2 * cols] fp32 */
  float* out0;                    /* [cols] += column sums of partials[:, :cols]   (dgamma) */
  float* out1;                    /* [cols] += column sums of partials[:, cols:]   (dbeta) */
  int nblk, cols;
} gpv_fold_problem;
int gpv_layernorm_bwd_blocks(int rows, int cols);
int gpv_layernorm_bwd3(const void* dy, const void* dy2, const void* x, const void* s, const float* gamma, const float* mean,
                       const float* rstd, void* dx, void* ds, float* dgamma, float* dbeta, float* partials, int rows, int cols,
                       float drop_p, uint64_t seed, int dtype, void* stream);
int gpv_colsum_fold_group(const gpv_fold_problem* problems, int n, void* stream);   /* problems: HOST memory */

/* softmax cross-entropy over the vocabulary (losses.py:20-26, nn.CrossEntropyLoss reduction none).
 * logits [rows, V] (ld), target int64 [rows] (ignore_index < 0 rows give loss 0);
 * loss[rows] fp32; if dlogits != NULL writes dlogits = (softmax - onehot) * gscale[row]. */
int gpv_softmax_ce(const void* logits, int64_t ld, const int64_t* target, float* loss, void* dlogits,
                   const float* gscale, int rows, int V, int dtype, void* stream);

/* RoIAlign(7x7, aligned, adaptive sampling) + mean over bins in separable form
 * (detr_roi_head.py:44-56): wgt[b,q,y*W+x] = Ay[b,q,y]*Ax[b,q,x]; pooled = wgt x feat via gpv_gemm.
 * boxes: fp32 [B*Q,4] normalised cxcywh.  wgt row stride ldw >= H*W (zero filled up to ldw). */
int gpv_roi_weights(const float* boxes, void* wgt, int n_roi, int H, int W, int64_t ldw, int dtype,
                    void* stream);

/* ---- element-wise / reduction helpers (all NHWC / row-major contiguous) ---- */
int gpv_add(const void* a, const void* b, void* y, int64_t n, int dtype, void* stream);             /* y = a + b  */
int gpv_add_rowbcast(const void* a, const void* b, void* y, int64_t rows_total, int64_t rows_b,
                     int cols, int dtype, void* stream);                    /* y[r] = a[r] + b[r % rows_b] */
int gpv_colsum(const void* x, float* out, int rows, int cols, int64_t ld, int dtype, void* stream);  /* out[c] += sum_r x[r,c] */
int gpv_cast(const void* src, void* dst, int64_t n, int dtype_src, int dtype_dst, void* stream);
/* dst[r,c] = cast(src[r,c] * scale[r]);  dstT[c,r] = same (optional) */
int gpv_cast_rowscale_t(const float* src, const float* scale, void* dst, void* dstT, int rows, int cols,
                        int dtype_dst, void* stream);
/* Many cast-transposes in one launch:  dstT_i[c, r] = cast(src_i[r, c])   (src fp32 [rows_i, cols_i] row-major, dstT [cols_i, rows_i]
 * row-major in dtype_dst).  `problems` is HOST memory, at most GPV_TC_GROUP_MAX per launch (more: several launches).
 * Replaces nothing in the reference (its autograd reads W as it is); here it refreshes the W^T mirrors of the Linear weights
 * once per optimizer step so that the backward-data GEMMs dX = dY W run as K-major x K-major GEMMs. */
#define GPV_TC_GROUP_MAX 128
typedef struct { const float* src; void* dstT; int rows, cols; } gpv_tc_problem;
int gpv_cast_transpose_group(const gpv_tc_problem* problems, int n, int dtype_dst, void* stream);
/* conv weight prep: src fp32 [Cout][T][Cin] -> wf [Cout][T][Cin] (x scale[Cout]) and wd [Cin][T][Cout] */
int gpv_prep_conv_weight(const float* src, const float* scale, void* wf, void* wd, int Cout, int T,
                         int Cin, int dtype_dst, void* stream);
int gpv_embedding(const void* table, const int64_t* ids, void* out, int64_t n_ids, int dim, int dtype_table,
                  int dtype_out, void* stream);
/* Greedy token pick: for every row r the index of the largest x[r*ld + v] + addend[v] (addend: fp32 [V] or NULL; the sum is
 * formed in fp32), written as int64 to out0[r*stride0] and / or out1[r*stride1] (either may be NULL).  Equal values: the lowest
 * index; NaNs never win.  Replaces `answer_logits + vocab_mask` followed by torch.topk(k=1) in the sampling loop of
 * GPV.forward (exp/gpv/models/gpv.py:185-188): the next input token and the id row of the step in one launch. */
int gpv_argmax_rows(const void* x, int64_t ld, const float* addend, int rows, int V, int dtype,
                    int64_t* out0, int64_t stride0, int64_t* out1, int64_t stride1, void* stream);
/* The same pick, and the NEXT decoder input row in the same launch: xnext[r, :D] = table[index_r * ldt + :D] (+ pos_row[:D], NULL: no
 * position term), dtype as x, sums in fp32 rounded once.  table = the transformed input embedding of every vocabulary entry
 * (AnswerInputEmbedding: gpv.py:46-55 applied to the whole vocabulary once per weights), pos_row = the sinusoidal row of position
 * t + 1 (gpv.py:178-196 builds both per token): embedding gather + transform + position add of the next token leave the chain of
 * dependent launches of a decode step.  table == NULL: gpv_argmax_rows.  D % 8 == 0, 16-byte aligned bases and pitches. */
int gpv_argmax_rows_embed(const void* x, int64_t ld, const float* addend, int rows, int V, int dtype,
                          int64_t* out0, int64_t stride0, int64_t* out1, int64_t stride1,
                          const void* table, int64_t ldt, const void* pos_row, void* xnext, int D, void* stream);
/* LayerNorm -> Linear on at most 4 rows (the decode step at small batch; inference only, no statistics are kept):
 *   xn[r, :] = LayerNorm(x[r, :] + s[r, :]) * gamma + beta      (s, gamma / beta may be NULL; K <= 1024 columns, K % 8 == 0)
 *   y[r*ldy + n] = act(sum_k xn[r, k] W[n*ldw + k] + bias[n])    n < N
 * x, s, xn: [rows, K] contiguous, dtype; W: [N, K] rows of pitch ldw (multiple of 8), dtype; y: dtype; 16-byte aligned bases;
 * xn must not alias x or s.  Bit-identical to gpv_layernorm_fwd followed by gpv_gemm on the same operands.
 * Replaces norm{1,2,3}(x + sublayer) followed by the next nn.Linear of torch's TransformerDecoderLayer in the sampling loop
 * (exp/gpv/models/gpv.py:183-184 -> decode_text).
 * The sublayer output may instead arrive as fp32 partial rows (gpv_attention_row_proj): with s == NULL and s_partial != NULL,
 *   s[r, c] = round_dtype(sum_{h < s_parts} s_partial[(r*s_parts + h)*K + c] + s_bias[c])     (s_bias may be NULL) */
int gpv_ln_linear_rows(const void* x, const void* s, const float* gamma, const float* beta, float eps, void* xn,
                       const void* W, int64_t ldw, const float* bias, void* y, int64_t ldy,
                       int rows, int N, int K, int act, int dtype, const float* s_partial, int s_parts, const float* s_bias,
                       void* stream);
/* Single-query attention with the output projection folded in (one new token per sequence over Sk <= 256 keys; no mask, no
 * dropout: the sampling loop):  o[b, h, :] = softmax(scale * q[b, h] K[b, :, h]^T) V[b, :, h];
 *   partial[(b*H + h)*D + n] = sum_d round_dtype(o[b, h, d]) * Wo[n*ldw + h*dh + d]          n < D = H*dh
 * q: element (b, h*dh + d) at q[b*q_bs + h*dh + d]; K / V: (b, j, h*dh + d) at k[b*k_bs + j*k_rs + h*dh + d]; strides in
 * elements, multiples of 8 (fp32: 4); dh % 8 == 0, dh <= 128.  The sum over h (+ the projection's bias) is formed by the
 * consumer (gpv_ln_linear_rows).  Replaces self_attn / multihead_attn of torch's TransformerDecoderLayer incl. their out_proj at
 * one query row (exp/gpv/models/gpv.py:183-184 -> decode_text). */
int gpv_attention_row_proj(const void* q, int64_t q_bs, const void* k, int64_t k_bs, int64_t k_rs, const void* v, int64_t v_bs,
                           int64_t v_rs, const void* Wo, int64_t ldw, float* partial, int B, int H, int Sk, int dh, float scale,
                           int dtype, void* stream);
/* y = act(x), act in {RELU, GELU};  dx = dy * act'(ref) * alpha with ref = OUTPUT for relu (works through a
 * fused inverted dropout: alpha = 1/(1-p)), ref = PRE-activation for gelu.  n % 8 == 0. */
int gpv_act_fwd(const void* x, void* y, int64_t n, int act, int dtype, void* stream);
int gpv_act_bwd(const void* dy, const void* ref, void* dx, int64_t n, int act, float alpha, int dtype, void* stream);
int gpv_dropout(const void* x, void* y, int64_t n, float p, uint64_t seed, int dtype, void* stream);
/* Device-resident seed epoch for captured launches (process-wide, like gpv_set_option).  The reference draws fresh dropout masks
 * from torch's device RNG state on every step (nn.Dropout / F.dropout in exp/gpv/models/transformer.py:156-160, vilbert.py);
 * a launch recorded in a hipGraph has its `seed` argument frozen, so the caller lends ONE device word: every dropout consumer
 * (gpv_gemm epilogue, gpv_attention_*, gpv_layernorm_*, gpv_dropout) then uses seed ^ mix(*epoch), read at kernel run time.
 * The caller bumps the word once per step, before the forward; forward and backward of a step see the same value.
 * epoch == NULL (default) restores the plain `seed` argument.  The pointer must stay valid while launches can run. */
int gpv_set_seed_device(const uint64_t* epoch);
/* relevance conditioning gpv.py:364-375: y = x + softmax(logits)[.,0]*tok[0] + softmax(logits)[.,1]*tok[1] */
int gpv_relevance_condition(const void* x, const float* logits, const float* tokens, void* y, int rows,
                            int dim, int dtype, void* stream);

/* fused AdamW over a flat fp32 parameter buffer (torch.optim.AdamW semantics, train_distr.py:247-253),
 * optionally writing the bf16 compute copy; grad is multiplied by *gscale (device scalar, clip factor)
 * when gscale != NULL. */
int gpv_adamw(float* p, const float* g, float* m, float* v, void* p_lowp, int64_t n, float lr, float beta1,
              float beta2, float eps, float wd, float bc1, float bc2, const float* gscale,
              const uint16_t* seg_id, const int32_t* seg_live, void* stream);
/* seg_id / seg_live (both or neither): element i belongs to parameter seg_id[i / 8] (parameters start on multiples of 8
 * elements); seg_live[that id] is the parameter's OWN Adam step count including this step, 0 = it has never received a
 * gradient and is left untouched (torch-1.6 optimizers skip parameters whose .grad is None, incl. their weight decay).
 * When given, the bias corrections are computed per parameter from that count (torch keeps `step` per parameter, starting
 * at its first gradient) and bc1 / bc2 are ignored.  The counts live on the device (agreed across ranks with a MAX
 * all-reduce of the "ever touched" flags), so picking the live parameters costs no host round trip. */
int gpv_sumsq(const float* x, int64_t n, float* out /* += */, void* stream);
/* Clip factor of torch.nn.utils.clip_grad_norm_(params, max_norm) (exp/gpv/train_distr.py:423-425, the DETR parameters at 0.1) over
 * the contiguous fp32 gradient range g[0, n) (n a multiple of 4, 16-byte aligned):  *gscale = min(1, max_norm / (||g||_2 + 1e-6)).
 * Deterministic -- fixed summation order, no atomics: every data-parallel rank derives the same bits from the same all-reduced
 * gradient.  `partial`: >= GPV_CLIP_PARTIALS floats of scratch.  g == NULL: no norm (gscale untouched).
 * pstep / live (both or neither, `nparam` entries): pstep[i] += live[i] in the same launch -- the per-parameter Adam step counts
 * gpv_adamw reads (seg_live), advanced for the parameters that have ever received a gradient.  Two launches. */
#define GPV_CLIP_PARTIALS 1024
int gpv_clip_scale(const float* g, int64_t n, float max_norm, float* partial, float* gscale, int32_t* pstep, const int32_t* live,
                   int nparam, void* stream);

#ifdef __cplusplus
}
#endif
#endif
