// Device-side input pipeline (gfx950): decoded uint8 HWC images of any size -> the stem's zero-padded NHWC4 bf16 batch, in two
// small launches (reference: datasets/coco_generic_dataset.py:49-62 `resize(img, (480, 640), anti_aliasing=True)`,
// datasets/coco_datasets.py:26-38,137-150 `(255 * img).astype(uint8)` -> ToPILImage -> RandomApply([ColorJitter(.4,.4,.4,.1)], .8)
// -> [RandomHorizontalFlip] -> RandomGrayscale(.2) -> ToTensor -> Normalize(ImageNet)).
//
// Why on the device: the reference feeds 120 images per step from 30 CPU workers (configs/exp/gpv.yaml:126); at 1600 images/s per
// GPU the host side (resize + PIL colour ops + a 118 MB fp32 batch over PCIe every 20 ms) becomes the bound.  Here the host only
// decodes the JPEG and draws the augmentation parameters; 0.9 MB of uint8 per image crosses PCIe instead of 3.7 MB of fp32, and
// the result is written directly in the layout the stem kernel reads (gpv_stem_pool), so gpv_image_to_nhwc4 disappears as well.
//
//   pass 1  resize_kernel:  Gaussian pre-filter (sigma = (scale - 1) / 2 per shrinking axis, truncated at 4 sigma, mirrored borders)
//           + bilinear sampling on the pixel-centre grid (skimage.transform.resize, order 1) -> uint8 by TRUNCATION (the reference's
//           astype) -> [B, OH, OW, 3] uint8 scratch; and, for samples whose colour jitter contains the contrast step, the sum of the
//           grey values of the image as it enters that step (contrast blends with the image's mean grey: a global reduction)
//   pass 2  color_kernel:   the four jitter steps in the sample's drawn order (each rounds to uint8 like a PIL image does),
//           grayscale, horizontal flip, (x / 255 - mean) / std -> bf16 -> padded NHWC4
// Semantics of the colour steps (PIL / torchvision 0.7 arithmetic, restated; torchvision and PIL are not in the image -> the
// oracle restates the same formulas and the step is "parity unpinned" against the real libraries):
//   grey L = (19595 R + 38470 G + 7471 B + 32768) >> 16;  brightness b: round(clip(b x));  contrast c: round(clip(m + c (x - m))),
//   m = floor(mean(L) + 0.5);  saturation s: round(clip(L + s (x - L)));  hue h: RGB -> HSV (float), H += h mod 1, -> RGB, round.
#include "common.h"
#include "../../include/gpv_hip.h"

namespace {

__device__ __forceinline__ int mirror(int i, int n) {          // scipy.ndimage mode 'mirror' (d c b | a b c d | c b a)
  if (n == 1) return 0;
  const int p = 2 * n - 2;
  i = i % p;
  if (i < 0) i += p;
  return i < n ? i : p - i;
}
__device__ __forceinline__ float clip255(float x) { return fminf(fmaxf(x, 0.f), 255.f); }
__device__ __forceinline__ float grey(float r, float g, float b) {
  return (float)((19595u * (unsigned)r + 38470u * (unsigned)g + 7471u * (unsigned)b + 32768u) >> 16);
}

// one colour step on uint8-valued floats; mgrey = the image's mean grey for the contrast step
__device__ __forceinline__ void color_step(int op, const gpv_image_desc& d, float mgrey, float& r, float& g, float& b) {
  if (op == 0) {
    r = rintf(clip255(r * d.brightness)); g = rintf(clip255(g * d.brightness)); b = rintf(clip255(b * d.brightness));
  } else if (op == 1) {
    r = rintf(clip255(mgrey + d.contrast * (r - mgrey)));
    g = rintf(clip255(mgrey + d.contrast * (g - mgrey)));
    b = rintf(clip255(mgrey + d.contrast * (b - mgrey)));
  } else if (op == 2) {
    const float l = grey(r, g, b);
    r = rintf(clip255(l + d.saturation * (r - l))); g = rintf(clip255(l + d.saturation * (g - l))); b = rintf(clip255(l + d.saturation * (b - l)));
  } else {
    const float R = r / 255.f, G = g / 255.f, B = b / 255.f;
    const float mx = fmaxf(R, fmaxf(G, B)), mn = fminf(R, fminf(G, B)), df = mx - mn;
    float h = 0.f;
    if (df > 0.f) {
      if (mx == R) h = (G - B) / df;
      else if (mx == G) h = 2.f + (B - R) / df;
      else h = 4.f + (R - G) / df;
      h = h / 6.f;
      h = h - floorf(h);
    }
    const float s = mx > 0.f ? df / mx : 0.f, v = mx;
    h = h + d.hue;
    h = h - floorf(h);
    const float h6 = h * 6.f;
    const int i = (int)floorf(h6) % 6;
    const float f = h6 - floorf(h6), p = v * (1.f - s), q = v * (1.f - s * f), t = v * (1.f - s * (1.f - f));
    float o0, o1, o2;
    switch (i) {
      case 0: o0 = v; o1 = t; o2 = p; break;
      case 1: o0 = q; o1 = v; o2 = p; break;
      case 2: o0 = p; o1 = v; o2 = t; break;
      case 3: o0 = p; o1 = q; o2 = v; break;
      case 4: o0 = t; o1 = p; o2 = v; break;
      default: o0 = v; o1 = p; o2 = q; break;
    }
    r = rintf(clip255(o0 * 255.f)); g = rintf(clip255(o1 * 255.f)); b = rintf(clip255(o2 * 255.f));
  }
}

__global__ __launch_bounds__(256) void resize_kernel(const gpv_image_desc* __restrict__ descs, uint8_t* __restrict__ tmp,
                                                     float* __restrict__ grey_sum, int OH, int OW) {
  const int b = blockIdx.y;
  const gpv_image_desc d = descs[b];
  const int n = OH * OW;
  const float sy = (float)d.H / OH, sx = (float)d.W / OW;
  const float sgy = fmaxf(0.f, (sy - 1.f) * 0.5f), sgx = fmaxf(0.f, (sx - 1.f) * 0.5f);
  const int ry = sgy > 0.f ? (int)(4.f * sgy + 0.5f) : 0, rx = sgx > 0.f ? (int)(4.f * sgx + 0.5f) : 0;
  float wy[33], wx[33];                           // radius <= 16 (scale <= 9: host checks)
  {
    float s = 0.f;
    for (int k = -ry; k <= ry; ++k) { wy[k + ry] = sgy > 0.f ? __expf(-0.5f * k * k / (sgy * sgy)) : 1.f; s += wy[k + ry]; }
    for (int k = 0; k <= 2 * ry; ++k) wy[k] /= s;
    s = 0.f;
    for (int k = -rx; k <= rx; ++k) { wx[k + rx] = sgx > 0.f ? __expf(-0.5f * k * k / (sgx * sgx)) : 1.f; s += wx[k + rx]; }
    for (int k = 0; k <= 2 * rx; ++k) wx[k] /= s;
  }
  unsigned lsum = 0u;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int oy = i / OW, ox = i - oy * OW;
    // pixel-centre grid: input coordinate of output pixel centre, clamped like an order-1 warp with mirrored borders
    float fy = (oy + 0.5f) * sy - 0.5f, fx = (ox + 0.5f) * sx - 0.5f;
    const int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
    const float ay = fy - y0, ax = fx - x0;
    float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int ty = 0; ty < 2; ++ty) {
      const float wyy = ty ? ay : 1.f - ay;
      if (wyy == 0.f) continue;
#pragma unroll
      for (int tx = 0; tx < 2; ++tx) {
        const float w = wyy * (tx ? ax : 1.f - ax);
        if (w == 0.f) continue;
        float v[3] = {0.f, 0.f, 0.f};
        for (int ky = -ry; ky <= ry; ++ky) {
          const int yy = mirror(y0 + ty + ky, d.H);
          float rowv[3] = {0.f, 0.f, 0.f};
          for (int kx = -rx; kx <= rx; ++kx) {
            const int xx = mirror(x0 + tx + kx, d.W);
            const uint8_t* px = d.src + ((int64_t)yy * d.W + xx) * 3;
            const float wk = wx[kx + rx];
            rowv[0] += wk * px[0]; rowv[1] += wk * px[1]; rowv[2] += wk * px[2];
          }
          const float wk = wy[ky + ry];
          v[0] += wk * rowv[0]; v[1] += wk * rowv[1]; v[2] += wk * rowv[2];
        }
        acc[0] += w * v[0]; acc[1] += w * v[1]; acc[2] += w * v[2];
      }
    }
    // the reference: resize() returns floats in [0, 1]; (255 * img).astype(np.uint8) truncates
    float r = floorf(clip255(acc[0])), g = floorf(clip255(acc[1])), bl = floorf(clip255(acc[2]));
    uint8_t* o = tmp + ((int64_t)b * n + i) * 3;
    o[0] = (uint8_t)r; o[1] = (uint8_t)g; o[2] = (uint8_t)bl;
    if (d.jitter) {                                // grey of the image as it enters the contrast step
      for (int k = 0; k < 4 && d.order[k] != 1; ++k) color_step(d.order[k], d, 0.f, r, g, bl);
      lsum += (unsigned)grey(r, g, bl);
    }
  }
  if (d.jitter) {
    // integer sum (grey levels are integers, 255 x 480 x 640 < 2^27): exact and independent of the order of the atomics -- a float
    // sum passes 2^24 at this size and made floor(mean + 0.5) depend on the run
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lsum += __shfl_xor(lsum, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(reinterpret_cast<unsigned*>(grey_sum) + b, lsum);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void color_kernel(const gpv_image_desc* __restrict__ descs, const uint8_t* __restrict__ tmp,
                                                    const float* __restrict__ grey_sum, T* __restrict__ out, int OH, int OW, int pad,
                                                    int Hp, int Wp) {
  const int b = blockIdx.y;
  const gpv_image_desc d = descs[b];
  const int n = Hp * Wp;
  const float mgrey = d.jitter ? (float)floor((double)reinterpret_cast<const unsigned*>(grey_sum)[b] / (double)(OH * OW) + 0.5) : 0.f;
  const float mean[3] = {0.485f, 0.456f, 0.406f}, istd[3] = {1.f / 0.229f, 1.f / 0.224f, 1.f / 0.225f};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int yp = i / Wp, xp = i - yp * Wp;
    const int y = yp - pad, x = xp - pad;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (y >= 0 && y < OH && x >= 0 && x < OW) {
      const int xs = d.flip ? OW - 1 - x : x;
      const uint8_t* px = tmp + ((int64_t)b * OH * OW + (int64_t)y * OW + xs) * 3;
      float r = px[0], g = px[1], bl = px[2];
      if (d.jitter) {
        for (int k = 0; k < 4; ++k) color_step(d.order[k], d, mgrey, r, g, bl);
      }
      if (d.gray) { const float l = grey(r, g, bl); r = g = bl = l; }
      v[0] = (r / 255.f - mean[0]) * istd[0];
      v[1] = (g / 255.f - mean[1]) * istd[1];
      v[2] = (bl / 255.f - mean[2]) * istd[2];
    }
    T* o = out + ((int64_t)b * n + i) * 4;
    o[0] = (T)v[0]; o[1] = (T)v[1]; o[2] = (T)v[2]; o[3] = (T)0.f;
  }
}

}  // namespace

extern "C" int gpv_image_pipeline(const gpv_image_desc* descs, int B, void* scratch_u8, float* grey_sum, void* out, int OH, int OW,
                                  int pad, int Hp, int Wp, int dtype_out, void* stream) {
  if (!descs || !scratch_u8 || !grey_sum || !out || B <= 0 || OH <= 0 || OW <= 0) return (int)hipErrorInvalidValue;
  if (Hp < OH + 2 * pad || Wp < OW + 2 * pad) return (int)hipErrorInvalidValue;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipError_t e = hipMemsetAsync(grey_sum, 0, sizeof(float) * B, st);
  if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }      // (leave no stale error behind for the next launch check)
  const int n = OH * OW;
  dim3 g1((unsigned)((n + 255) / 256 < 480 ? (n + 255) / 256 : 480), (unsigned)B);
  hipLaunchKernelGGL(resize_kernel, g1, dim3(256), 0, st, descs, reinterpret_cast<uint8_t*>(scratch_u8), grey_sum, OH, OW);
  GPV_CHECK_LAUNCH();
  const int np = Hp * Wp;
  dim3 g2((unsigned)((np + 255) / 256 < 480 ? (np + 255) / 256 : 480), (unsigned)B);
  if (dtype_out == GPV_BF16) hipLaunchKernelGGL((color_kernel<bf16>), g2, dim3(256), 0, st, descs, reinterpret_cast<const uint8_t*>(scratch_u8), grey_sum, reinterpret_cast<bf16*>(out), OH, OW, pad, Hp, Wp);
  else hipLaunchKernelGGL((color_kernel<float>), g2, dim3(256), 0, st, descs, reinterpret_cast<const uint8_t*>(scratch_u8), grey_sum, reinterpret_cast<float*>(out), OH, OW, pad, Hp, Wp);
  GPV_CHECK_LAUNCH();
  return 0;
}
