// Baseline JPEG decoding for the device-side input pipeline: what `skio.imread(img_path)` does in the reference's
// 30 data-loader workers (datasets/coco_generic_dataset.py:54, datasets/coco_datasets.py:157 -> Pillow -> libjpeg-turbo with its
// defaults: accurate integer IDCT, "fancy" chroma upsampling, YCbCr -> RGB).
//
// The split follows the data: entropy decoding is a serial walk over a bit stream (one image per host thread, gpv_jpeg_parse: a
// few ms per COCO image), everything after it is per-block / per-pixel integer arithmetic and runs on the GPU for the whole batch
// in two launches (gpv_jpeg_decode):
//   jpeg_idct_kernel   quantised coefficients -> dequantise -> IJG jidctint.c 8x8 inverse DCT (8 lanes per block: a lane owns a
//                      column in pass 1, a row in pass 2, the transpose goes through LDS) -> component planes (uint8)
//   jpeg_color_kernel  per output pixel: chroma through the IJG triangle filter (h2v1 / h2v2 "fancy" upsampling, evaluated on the
//                      fly from the chroma planes) -> jdcolor.c fixed-point YCbCr -> RGB -> [H][W][3] uint8, the input of
//                      gpv_image_pipeline.  A single-component file is replicated to three channels (coco_generic_dataset.py:55-56).
// Bit-exact against the Pillow decoder on the fixtures of tests/golden/jpeg (tests/test_jpeg_gpu.py); oracle/jpeg_oracle.py is the
// CPU restatement.  Scope: baseline sequential (SOF0/SOF1 Huffman), 8 bit, one interleaved scan, 1 or 3 components, luma 1x1 / 2x1 /
// 2x2 over 1x1 chroma, restart intervals; anything else returns hipErrorNotSupported and the caller must convert the file.
#include "common.h"
#include "../../include/gpv_hip.h"
#include <string.h>

namespace {

const unsigned char kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                   41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                   30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

constexpr int LOOK = 9;
struct HuffTab {
  bool present;
  unsigned char look_nbits[1 << LOOK], look_sym[1 << LOOK];
  int maxcode[18];        // largest code of length l, -1 if none; maxcode[17] = sentinel
  int valoffset[17];      // symbol index of the first code of length l, minus that code
  unsigned char syms[256];
};

bool build_table(const unsigned char* counts, const unsigned char* syms, int nsyms, HuffTab& t) {
  memset(&t, 0, sizeof(t));
  memcpy(t.syms, syms, nsyms);
  int code = 0, k = 0;
  for (int l = 1; l <= 16; ++l) {
    t.valoffset[l] = k - code;
    if (counts[l - 1]) {
      if (code + counts[l - 1] > (1 << l)) return false;
      for (int i = 0; i < counts[l - 1]; ++i, ++k, ++code) {
        if (l <= LOOK) {
          const int first = code << (LOOK - l);
          for (int f = 0; f < (1 << (LOOK - l)); ++f) { t.look_nbits[first + f] = (unsigned char)l; t.look_sym[first + f] = syms[k]; }
        }
      }
      t.maxcode[l] = code - 1;
    } else {
      t.maxcode[l] = -1;
    }
    code <<= 1;
  }
  t.maxcode[17] = 0x7fffffff;
  t.present = true;
  return k == nsyms;
}

struct BitReader {
  const unsigned char* p; const unsigned char* end;
  uint64_t acc; int n; bool marker;
  inline void fill() {
    while (n <= 56) {
      unsigned b = 0;
      if (!marker && p < end) {
        b = *p;
        if (b == 0xFF) {
          if (p + 1 < end && p[1] == 0) { p += 2; }
          else { marker = true; b = 0; }               // a marker: the stream continues with zero bits (T.81 F.2.2.5)
        } else {
          ++p;
        }
      }
      acc = (acc << 8) | b;
      n += 8;
    }
  }
  inline int decode(const HuffTab& t) {
    if (n < 16) fill();
    const unsigned look = (unsigned)(acc >> (n - LOOK)) & ((1u << LOOK) - 1);
    const int nb = t.look_nbits[look];
    if (nb) { n -= nb; return t.look_sym[look]; }
    int l = LOOK + 1;
    int code = (int)((acc >> (n - l)) & ((1u << l) - 1));
    while (code > t.maxcode[l]) {                       // (never shifts by n - 17: a code longer than 16 bits does not exist)
      if (++l > 16) return -1;
      code = (int)((acc >> (n - l)) & ((1u << l) - 1));
    }
    n -= l;
    return t.syms[(code + t.valoffset[l]) & 255];
  }
  inline int receive_extend(int s) {
    if (n < s) fill();
    const int r = (int)((acc >> (n - s)) & ((1u << s) - 1));
    n -= s;
    return r < (1 << (s - 1)) ? r - (1 << s) + 1 : r;
  }
  bool restart() {
    acc = 0; n = 0; marker = false;
    while (p + 1 < end && !(p[0] == 0xFF && p[1] >= 0xD0 && p[1] <= 0xD7)) ++p;
    if (p + 1 >= end) return false;
    p += 2;
    return true;
  }
};

inline int rd16(const unsigned char* q) { return (q[0] << 8) | q[1]; }

}  // namespace

extern "C" int gpv_jpeg_parse(const unsigned char* data, int64_t nbytes, gpv_jpeg_info* info, short* coefs, int64_t coefs_capacity) {
  if (!data || !info || nbytes < 4 || data[0] != 0xFF || data[1] != 0xD8) return (int)hipErrorInvalidValue;
  const unsigned char* end = data + nbytes;
  const unsigned char* p = data + 2;
  static thread_local HuffTab dc[4], ac[4];
  for (int i = 0; i < 4; ++i) dc[i].present = ac[i].present = false;
  unsigned short qt[4][64];
  bool qt_ok[4] = {false, false, false, false};
  int comp_id[3] = {0, 0, 0}, comp_h[3] = {1, 1, 1}, comp_v[3] = {1, 1, 1}, comp_tq[3] = {0, 0, 0}, comp_td[3], comp_ta[3];
  int W = 0, H = 0, nf = 0, ri = 0;
  bool have_frame = false, adobe_rgb = false;
  for (;;) {
    while (p < end && *p != 0xFF) ++p;
    while (p < end && *p == 0xFF) ++p;
    if (p >= end) return (int)hipErrorInvalidValue;
    const int m = *p++;
    if (m == 0xD8 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
    if (m == 0xD9) return (int)hipErrorInvalidValue;
    if (p + 2 > end) return (int)hipErrorInvalidValue;
    const int ln = rd16(p);
    if (ln < 2 || p + ln > end) return (int)hipErrorInvalidValue;
    const unsigned char* seg = p + 2;
    const int sl = ln - 2;
    if (m == 0xDB) {
      int q = 0;
      while (q < sl) {
        const int pq = seg[q] >> 4, tq = seg[q] & 15;
        if (pq) return (int)hipErrorNotSupported;
        if (tq > 3 || q + 65 > sl) return (int)hipErrorInvalidValue;
        for (int i = 0; i < 64; ++i) qt[tq][kZigzag[i]] = seg[q + 1 + i];
        qt_ok[tq] = true;
        q += 65;
      }
    } else if (m == 0xC4) {
      int q = 0;
      while (q < sl) {
        if (q + 17 > sl) return (int)hipErrorInvalidValue;
        const int tc = seg[q] >> 4, th = seg[q] & 15;
        int ns = 0;
        for (int i = 0; i < 16; ++i) ns += seg[q + 1 + i];
        if (tc > 1 || th > 3 || ns > 256 || q + 17 + ns > sl) return (int)hipErrorInvalidValue;
        if (!build_table(seg + q + 1, seg + q + 17, ns, tc ? ac[th] : dc[th])) return (int)hipErrorInvalidValue;
        q += 17 + ns;
      }
    } else if (m == 0xC0 || m == 0xC1) {
      if (sl < 6) return (int)hipErrorInvalidValue;
      if (seg[0] != 8) return (int)hipErrorNotSupported;
      H = rd16(seg + 1); W = rd16(seg + 3); nf = seg[5];
      if (nf != 1 && nf != 3) return (int)hipErrorNotSupported;
      if (sl < 6 + 3 * nf || W <= 0 || H <= 0) return (int)hipErrorInvalidValue;
      for (int i = 0; i < nf; ++i) {
        comp_id[i] = seg[6 + 3 * i]; comp_h[i] = seg[7 + 3 * i] >> 4; comp_v[i] = seg[7 + 3 * i] & 15; comp_tq[i] = seg[8 + 3 * i];
        if (comp_tq[i] > 3) return (int)hipErrorInvalidValue;
      }
      have_frame = true;
    } else if (m == 0xC2 || m == 0xC3 || (m >= 0xC5 && m <= 0xC7) || (m >= 0xC9 && m <= 0xCB) || (m >= 0xCD && m <= 0xCF)) {
      return (int)hipErrorNotSupported;                 // progressive / lossless / arithmetic coding
    } else if (m == 0xEE) {
      // Adobe APP14: transform 0 = the components are RGB (3) / CMYK (4), not YCbCr -- libjpeg then skips the colour conversion
      if (sl >= 12 && memcmp(seg, "Adobe", 5) == 0 && seg[11] == 0) adobe_rgb = true;
    } else if (m == 0xDD) {
      if (sl < 2) return (int)hipErrorInvalidValue;
      ri = rd16(seg);
    } else if (m == 0xDA) {
      if (!have_frame || sl < 1) return (int)hipErrorInvalidValue;
      const int ns = seg[0];
      if (ns != nf) return (int)hipErrorNotSupported;   // non-interleaved scans
      if (sl < 1 + 2 * ns + 3) return (int)hipErrorInvalidValue;
      for (int i = 0; i < ns; ++i) {
        if (seg[1 + 2 * i] != comp_id[i]) return (int)hipErrorNotSupported;
        comp_td[i] = seg[2 + 2 * i] >> 4; comp_ta[i] = seg[2 + 2 * i] & 15;
        if (comp_td[i] > 3 || comp_ta[i] > 3) return (int)hipErrorInvalidValue;
      }
      p += ln;
      break;
    }
    p += ln;
  }
  if (nf == 3 && adobe_rgb) return (int)hipErrorNotSupported;       // RGB-coded file (no YCbCr transform)
  if (nf == 3) {
    const bool luma_ok = (comp_h[0] == 1 && comp_v[0] == 1) || (comp_h[0] == 2 && comp_v[0] == 1) || (comp_h[0] == 2 && comp_v[0] == 2);
    if (!luma_ok || comp_h[1] != 1 || comp_v[1] != 1 || comp_h[2] != 1 || comp_v[2] != 1) return (int)hipErrorNotSupported;
  } else {
    comp_h[0] = comp_v[0] = 1;                           // a single-component scan is never interleaved (T.81 A.2.2)
  }
  const int hmax = comp_h[0], vmax = comp_v[0];
  const int mx = (W + 8 * hmax - 1) / (8 * hmax), my = (H + 8 * vmax - 1) / (8 * vmax);
  memset(info, 0, sizeof(*info));
  info->width = W; info->height = H; info->ncomp = nf; info->hmax = hmax; info->vmax = vmax; info->mcus_x = mx; info->mcus_y = my;
  int64_t off = 0;
  for (int i = 0; i < nf; ++i) {
    if (!qt_ok[comp_tq[i]]) return (int)hipErrorInvalidValue;
    info->bh[i] = my * comp_v[i]; info->bw[i] = mx * comp_h[i];
    info->coef_offset[i] = off;
    off += (int64_t)info->bh[i] * info->bw[i] * 64;
    for (int k = 0; k < 64; ++k) info->quant[i][k] = qt[comp_tq[i]][k];
  }
  info->coef_count = off;
  if (!coefs) return 0;                                  // header pass: the caller sizes its buffers
  if (coefs_capacity < off) return (int)hipErrorInvalidValue;
  for (int i = 0; i < nf; ++i)
    if (!dc[comp_td[i]].present || !ac[comp_ta[i]].present) return (int)hipErrorInvalidValue;
  memset(coefs, 0, (size_t)off * sizeof(short));
  BitReader br{p, end, 0, 0, false};
  int pred[3] = {0, 0, 0};
  const int nmcu = mx * my;
  for (int mcu = 0; mcu < nmcu; ++mcu) {
    if (ri && mcu && mcu % ri == 0) {
      if (!br.restart()) return (int)hipErrorInvalidValue;
      pred[0] = pred[1] = pred[2] = 0;
    }
    const int y0 = mcu / mx, x0 = mcu - y0 * mx;
    for (int ci = 0; ci < nf; ++ci) {
      const HuffTab& td = dc[comp_td[ci]];
      const HuffTab& ta = ac[comp_ta[ci]];
      for (int v = 0; v < comp_v[ci]; ++v)
        for (int h = 0; h < comp_h[ci]; ++h) {
          short* blk = coefs + info->coef_offset[ci] + ((int64_t)(y0 * comp_v[ci] + v) * info->bw[ci] + x0 * comp_h[ci] + h) * 64;
          const int t = br.decode(td);
          if (t < 0 || t > 15) return (int)hipErrorInvalidValue;
          if (t) pred[ci] += br.receive_extend(t);
          blk[0] = (short)pred[ci];
          int k = 1;
          while (k < 64) {
            const int rs = br.decode(ta);
            if (rs < 0) return (int)hipErrorInvalidValue;
            const int r = rs >> 4, s = rs & 15;
            if (s == 0) {
              if (r != 15) break;
              k += 16;
              continue;
            }
            k += r;
            if (k > 63) return (int)hipErrorInvalidValue;
            blk[kZigzag[k]] = (short)br.receive_extend(s);
            ++k;
          }
        }
    }
  }
  return 0;
}

// ------------------------------------------------------------------ device side ------------------------------------------------------
namespace {

constexpr int CB = 13, P1 = 2;
constexpr int F_0_298 = 2446, F_0_390 = 3196, F_0_541 = 4433, F_0_765 = 6270, F_0_899 = 7373, F_1_175 = 9633;
constexpr int F_1_501 = 12299, F_1_847 = 15137, F_1_961 = 16069, F_2_053 = 16819, F_2_562 = 20995, F_3_072 = 25172;

// IJG jidctint.c, one 1-D pass (32-bit arithmetic as in the C source: 8-bit samples keep every intermediate below 2^31)
template <int SHIFT>
__device__ __forceinline__ void idct_1d(const int* in, int* out) {
  int z1 = (in[2] + in[6]) * F_0_541;
  const int t2 = z1 - in[6] * F_1_847;
  const int t3 = z1 + in[2] * F_0_765;
  const int t0 = (in[0] + in[4]) << CB;
  const int t1 = (in[0] - in[4]) << CB;
  const int t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
  int o0 = in[7], o1 = in[5], o2 = in[3], o3 = in[1];
  z1 = o0 + o3;
  int z2 = o1 + o2, z3 = o0 + o2, z4 = o1 + o3;
  const int z5 = (z3 + z4) * F_1_175;
  o0 *= F_0_298; o1 *= F_2_053; o2 *= F_3_072; o3 *= F_1_501;
  z1 *= -F_0_899; z2 *= -F_2_562; z3 = z3 * -F_1_961 + z5; z4 = z4 * -F_0_390 + z5;
  o0 += z1 + z3; o1 += z2 + z4; o2 += z2 + z3; o3 += z1 + z4;
  constexpr int R = 1 << (SHIFT - 1);
  out[0] = (t10 + o3 + R) >> SHIFT; out[7] = (t10 - o3 + R) >> SHIFT;
  out[1] = (t11 + o2 + R) >> SHIFT; out[6] = (t11 - o2 + R) >> SHIFT;
  out[2] = (t12 + o1 + R) >> SHIFT; out[5] = (t12 - o1 + R) >> SHIFT;
  out[3] = (t13 + o0 + R) >> SHIFT; out[4] = (t13 - o0 + R) >> SHIFT;
}

// 256 threads = 32 blocks of 8x8; blockIdx.y = image
__global__ __launch_bounds__(256) void jpeg_idct_kernel(const gpv_jpeg_desc* __restrict__ descs) {
  __shared__ int ws[32][8][9];
  const gpv_jpeg_desc& d = descs[blockIdx.y];
  const int tid = threadIdx.x, lb = tid >> 3, t = tid & 7;
  int nb[3], total = 0;
  for (int c = 0; c < d.ncomp; ++c) { nb[c] = d.bh[c] * d.bw[c]; total += nb[c]; }
  int b = blockIdx.x * 32 + lb;
  const bool live = b < total;
  int c = 0;
  if (live) { while (b >= nb[c]) { b -= nb[c]; ++c; } }
  if (live) {
    const short* blk = d.coefs + d.coef_off[c] + (int64_t)b * 64;
    int in[8], out[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) in[k] = (int)blk[k * 8 + t] * (int)d.quant[c][k * 8 + t];      // column t
    idct_1d<CB - P1>(in, out);
#pragma unroll
    for (int k = 0; k < 8; ++k) ws[lb][k][t] = out[k];
  }
  __syncthreads();
  if (live) {
    int in[8], out[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) in[k] = ws[lb][t][k];                                            // row t
    idct_1d<CB + P1 + 3>(in, out);
    const int by = b / d.bw[c], bx = b - by * d.bw[c];
    unsigned char* dst = d.planes + d.plane_off[c] + (int64_t)(by * 8 + t) * (d.bw[c] * 8) + bx * 8;
    uint2 v;
    unsigned char* q = reinterpret_cast<unsigned char*>(&v);
#pragma unroll
    for (int k = 0; k < 8; ++k) { const int s = out[k] + 128; q[k] = (unsigned char)(s < 0 ? 0 : (s > 255 ? 255 : s)); }
    *reinterpret_cast<uint2*>(dst) = v;                                                          // (rows of a plane are 8-byte aligned)
  }
}

__device__ __forceinline__ int chroma_at(const unsigned char* pl, int pitch, int cw, int ch, int hs, int vs, int y, int x) {
  if (hs == 1) return pl[(int64_t)y * pitch + x];
  const int c = x >> 1;
  if (cw <= 2) return pl[(int64_t)(vs == 2 ? y >> 1 : y) * pitch + c];     // jdsample.c: fancy upsampling needs downsampled_width > 2; replication below
  if (vs == 1) {                                       // h2v1 fancy: 3/4 nearer + 1/4 further column
    const unsigned char* r = pl + (int64_t)y * pitch;
    const int a = r[c];
    if (x & 1) return c == cw - 1 ? a : (a * 3 + r[c + 1] + 2) >> 2;
    return c == 0 ? a : (a * 3 + r[c - 1] + 1) >> 2;
  }
  const int r0 = y >> 1;                               // h2v2 fancy: the triangle filter in both directions
  int r1 = (y & 1) ? r0 + 1 : r0 - 1;
  r1 = r1 < 0 ? 0 : (r1 > ch - 1 ? ch - 1 : r1);
  const unsigned char* a = pl + (int64_t)r0 * pitch;
  const unsigned char* n = pl + (int64_t)r1 * pitch;
  const int cs = a[c] * 3 + n[c];
  if (x & 1) return c == cw - 1 ? (cs * 4 + 7) >> 4 : (cs * 3 + a[c + 1] * 3 + n[c + 1] + 7) >> 4;
  return c == 0 ? (cs * 4 + 8) >> 4 : (cs * 3 + a[c - 1] * 3 + n[c - 1] + 8) >> 4;
}

__global__ __launch_bounds__(256) void jpeg_color_kernel(const gpv_jpeg_desc* __restrict__ descs) {
  const gpv_jpeg_desc& d = descs[blockIdx.y];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)d.width * d.height) return;
  const int y = (int)(i / d.width), x = (int)(i - (int64_t)y * d.width);
  const int Y = d.planes[d.plane_off[0] + (int64_t)y * (d.bw[0] * 8) + x];
  unsigned char* o = d.out + i * 3;
  if (d.ncomp == 1) { o[0] = o[1] = o[2] = (unsigned char)Y; return; }
  const int cw = (d.width + d.hmax - 1) / d.hmax, ch = (d.height + d.vmax - 1) / d.vmax;
  const int cb = chroma_at(d.planes + d.plane_off[1], d.bw[1] * 8, cw, ch, d.hmax, d.vmax, y, x) - 128;
  const int cr = chroma_at(d.planes + d.plane_off[2], d.bw[2] * 8, cw, ch, d.hmax, d.vmax, y, x) - 128;
  // jdcolor.c: FIX(1.40200) = 91881, FIX(1.77200) = 116130, FIX(0.71414) = 46802, FIX(0.34414) = 22554, ONE_HALF = 32768
  const int r = Y + ((91881 * cr + 32768) >> 16);
  const int g = Y + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
  const int bl = Y + ((116130 * cb + 32768) >> 16);
  o[0] = (unsigned char)(r < 0 ? 0 : (r > 255 ? 255 : r));
  o[1] = (unsigned char)(g < 0 ? 0 : (g > 255 ? 255 : g));
  o[2] = (unsigned char)(bl < 0 ? 0 : (bl > 255 ? 255 : bl));
}

}  // namespace

extern "C" int gpv_jpeg_decode(const gpv_jpeg_desc* descs, int B, int max_blocks, int64_t max_pixels, void* stream) {
  if (!descs || B <= 0 || max_blocks <= 0 || max_pixels <= 0) return (int)hipErrorInvalidValue;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(jpeg_idct_kernel, dim3((max_blocks + 31) / 32, B), dim3(256), 0, st, descs);
  GPV_CHECK_LAUNCH();
  hipLaunchKernelGGL(jpeg_color_kernel, dim3((unsigned)((max_pixels + 255) / 256), B), dim3(256), 0, st, descs);
  GPV_CHECK_LAUNCH();
  return 0;
}
