// Probe: how fast can ONE CU pull operand bytes global -> LDS with buffer_load ... lds (the direct-to-LDS kernels' loader), as a
// function of (a) where the bytes come from (L2-resident re-reads vs a stream no CU reads twice), (b) how many loads a wave
// keeps in flight, (c) a helper wave that touches one dword per 128-byte line of the data `ahead` iterations early (an L2
// prefetch that costs one vector-memory instruction per 8 KB instead of eight).
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/dma_rate tools/probe/dma_rate.hip ; ./tools/probe/dma_rate       (GPU box)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

typedef __attribute__((address_space(3))) void lds_void_t;

// 8 loader waves (+1 prefetch wave when PF): every loader wave moves 1 KB per instruction, DEPTH instructions in flight
template <int DEPTH, bool PF>
__global__ __launch_bounds__(PF ? 576 : 512) void dma_kernel(const unsigned char* src, int64_t bytes_per_block, int64_t wrap, int iters, int ahead, unsigned* sink) {
  extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t base = ((int64_t)blockIdx.x * bytes_per_block) % wrap;          // wrap small: every block re-reads an L2-resident window
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(src + base), (short)0, 0x7ffffff0, 0x00020000);
  unsigned acc = 0;
  if (PF && wave == 8) {
    for (int it = 0; it < iters; ++it) {
      const int tgt = it + ahead;
      if (tgt < iters) {
        const unsigned v = __builtin_amdgcn_raw_buffer_load_b32(rs, (tgt * 8192 + lane * 128) % (int)bytes_per_block, 0, 0);
        acc += v;                                                                // (consumed at the end only: no wait inside the loop)
      }
      if ((it & 3) == 3) __builtin_amdgcn_s_sleep(2);
    }
  } else {
    for (int it = 0; it < iters; ++it) {
      const int off = (it * 8192 + wave * 1024 + lane * 16) % (int)bytes_per_block;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(smem + (it % DEPTH) * 8192 + wave * 1024), 16, off, 0, 0, 0);
      if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      else if (DEPTH == 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else if (DEPTH == 8) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc = smem[threadIdx.x];
  }
  if (acc == 0xdeadbeef) sink[0] = acc;
}

// mixed: of every 8 iterations `miss8` read the block's private stream (HBM), the others a window all blocks share (L2)
template <int DEPTH>
__global__ __launch_bounds__(512) void mix_kernel(const unsigned char* src, int64_t bytes_per_block, int iters, int miss8, unsigned* sink) {
  extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const auto rp = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(src + (64 << 20) + (int64_t)blockIdx.x * bytes_per_block), (short)0, 0x7ffffff0, 0x00020000);
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(src), (short)0, 0x7ffffff0, 0x00020000);
  int ip = 0, is = 0;
  for (int it = 0; it < iters; ++it) {
    unsigned char* dst = smem + (it % DEPTH) * 8192 + wave * 1024;
    if ((it & 7) < miss8) { __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, (lds_void_t*)dst, 16, (ip * 8192 + wave * 1024 + lane * 16) % (int)bytes_per_block, 0, 0, 0); ++ip; }
    else { __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)dst, 16, (is * 8192 + wave * 1024 + lane * 16) & ((1 << 20) - 1), 0, 0, 0); ++is; }
    if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if (DEPTH == 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (smem[threadIdx.x] == 0xde && iters < 0) sink[0] = 1;
}

template <int DEPTH>
static void run_mix(const unsigned char* d, int blocks, int miss8, unsigned* sink) {
  const int64_t bpb = 2 << 20;
  const int iters = 2048;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute(reinterpret_cast<const void*>(mix_kernel<DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, DEPTH * 8192);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((mix_kernel<DEPTH>), dim3(blocks), dim3(512), DEPTH * 8192, 0, d, bpb, iters, miss8, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)iters * 8192;
  const double bpc = bytes / (ms * 1e-3 * 2.4e9) * (blocks > 256 ? blocks / 256.0 : 1.0);
  const double model = 1.0 / (miss8 / 8.0 / 10.0 + (1 - miss8 / 8.0) / 40.0);
  printf("mixed %d/8 from HBM, %3d blocks, depth %d: %8.1f us  %6.1f B/clk/CU   (additive model 10 | 40: %5.1f)   HBM side %5.2f TB/s\n", miss8, blocks, DEPTH, ms * 1e3, bpc, model,
         bytes * blocks * miss8 / 8.0 / ms / 1e9);
}

// GEMM-shaped requests: a wave instruction fetches 8 ROWS x 128 bytes (row pitch `pitch` bytes) instead of one contiguous kilobyte;
// the block walks k (128-byte steps along the rows) like an operand tile of `rows` rows: L2-resident window shared by all blocks
template <int DEPTH>
__global__ __launch_bounds__(512) void rows_kernel(const unsigned char* src, int pitch, int rows, int ksteps, int iters, int private_rows, unsigned* sink) {
  extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // private_rows: every block reads its own row range (an A operand: rows = pixels of this tile); otherwise all share (a B operand)
  const int64_t base = private_rows ? (int64_t)blockIdx.x * rows * pitch : 0;
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(src + base), (short)0, 0x7ffffff0, 0x00020000);
  const int pieces = rows / 8;                                         // 1-KB pieces (8 rows x 128 B) per k-step
  int n = 0;
  for (int it = 0; it < iters; ++it) {
    const int k = it % ksteps;
    for (int pc = wave; pc < pieces; pc += 8) {
      const int off = ((pc * 8 + (lane >> 3)) * pitch) + k * 128 + (lane & 7) * 16;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(smem + (n % DEPTH) * 8192 + wave * 1024), 16, off, 0, 0, 0);
      ++n;
      if (DEPTH == 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (smem[threadIdx.x] == 0xde && iters < 0) sink[0] = 1;
}

static void run_rows(const unsigned char* d, int pitch, int rows, int ksteps, int private_rows, unsigned* sink, const char* what) {
  const int iters = 512;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute(reinterpret_cast<const void*>(rows_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 8192);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((rows_kernel<8>), dim3(256), dim3(512), 8 * 8192, 0, d, pitch, rows, ksteps, iters, private_rows, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)iters * rows * 128;
  printf("rows: %-34s pitch %6d B, %3d rows, %3d k-steps, %s : %8.1f us  %6.1f B/clk/CU\n", what, pitch, rows, ksteps, private_rows ? "private rows" : "shared rows ", ms * 1e3,
         bytes / (ms * 1e-3 * 2.4e9));
}

// The skeleton of the pipelined GEMM loop, ingredient by ingredient: per k-tile a block of 8 waves requests (BM + BN) rows x 128 B
// (shared L2-resident rows), and optionally reads fragments back from LDS (ds_read_b128, the 160 x 256 tile's 18 per wave),
// issues its 40 MFMAs, and synchronises like the real loop (counted vmcnt + s_barrier, three stages).
typedef __bf16 pbf16x8 __attribute__((ext_vector_type(8)));
typedef float pf32x4 __attribute__((ext_vector_type(4)));
template <bool LDSR, bool MFMA, bool BAR>
__global__ __launch_bounds__(512) void loop_kernel(const unsigned char* src, int pitch, int nk, unsigned* sink) {
  extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
  constexpr int ROWS = 160 + 256, STAGE = ROWS * 128, PIECES = ROWS / 8;          // 52 pieces of 1 KB: 6.5 per wave
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(src), (short)0, 0x7ffffff0, 0x00020000);
  const int nl = (PIECES - wave + 7) / 8;
  auto issue = [&](int kt, int stage) {
    for (int pc = wave; pc < PIECES; pc += 8)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(smem + stage * STAGE + pc * 1024), 16, (pc * 8 + (lane >> 3)) * pitch + (kt % 32) * 128 + (lane & 7) * 16, 0, 0, 0);
  };
  pf32x4 acc[5][4];
  for (int i = 0; i < 5; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = pf32x4{0.f, 0.f, 0.f, 0.f};
  issue(0, 0); issue(1, 1);
  int sc = 0, si = 2;
  for (int t = 0; t < nk; ++t) {
    if (BAR) {
      if (nl == 7) { if (t + 1 < nk) asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
      else { if (t + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
      __builtin_amdgcn_s_barrier();
    } else {
      asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
    }
    if (t + 2 < nk) issue(t + 2, si);
    const unsigned char* st = smem + sc * STAGE;
    if (LDSR) {
      pbf16x8 a[2][5], b[2][4];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int slot = ((kk * 4 + (lane >> 4)) ^ (lane & 7)) << 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) b[kk][j] = *reinterpret_cast<const pbf16x8*>(st + 160 * 128 + ((wave & 3) * 64 + j * 16 + (lane & 15)) * 128 + slot);
#pragma unroll
        for (int i = 0; i < 5; ++i) a[kk][i] = *reinterpret_cast<const pbf16x8*>(st + ((wave >> 2) * 80 + i * 16 + (lane & 15)) * 128 + slot);
      }
      if (MFMA) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[kk][j], a[kk][i], acc[i][j], 0, 0, 0);
      } else {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) { for (int i = 0; i < 5; ++i) acc[i][0][0] += (float)a[kk][i][0]; for (int j = 0; j < 4; ++j) acc[0][j][1] += (float)b[kk][j][0]; }
      }
    }
    sc = sc == 2 ? 0 : sc + 1;
    si = si == 2 ? 0 : si + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float sum = 0.f;
  for (int i = 0; i < 5; ++i) for (int j = 0; j < 4; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  if (sum == 123.456f) sink[0] = 1;
}

template <bool LDSR, bool MFMA, bool BAR>
static void run_loop(const unsigned char* d, unsigned* sink, const char* what) {
  const int nk = 360;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int lds = 3 * (160 + 256) * 128;
  hipFuncSetAttribute(reinterpret_cast<const void*>(loop_kernel<LDSR, MFMA, BAR>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((loop_kernel<LDSR, MFMA, BAR>), dim3(256), dim3(512), lds, 0, d, 4608, nk, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double fl = 2.0 * 160 * 256 * 64 * nk * 256;
  printf("loop 160x256x64 x %d k-tiles: %-46s %8.1f us = %5.2f us per k-tile, %5.1f B/clk/CU delivered, MFMA-equivalent %6.0f TFLOP/s\n", nk, what, ms * 1e3, ms * 1e3 / nk,
         (double)nk * (160 + 256) * 128 / (ms * 1e-3 * 2.4e9), fl / ms / 1e9);
}

template <int DEPTH, bool PF>
static void run(const char* what, const unsigned char* d, int64_t bpb, int64_t wrap, int blocks, int ahead, unsigned* sink) {
  const int iters = (int)(bpb / 8192) * 4;                                       // four passes over the block's region
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int lds = DEPTH * 8192;
  hipFuncSetAttribute(reinterpret_cast<const void*>(dma_kernel<DEPTH, PF>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((dma_kernel<DEPTH, PF>), dim3(blocks), dim3(PF ? 576 : 512), lds, 0, d, bpb, wrap, iters, ahead, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)blocks * iters * 8192;
  printf("%-46s depth %2d pf %d ahead %2d : %8.1f us  %7.2f TB/s  %6.1f B/clk/CU (2.4 GHz, %d blocks)\n", what, DEPTH, (int)PF, ahead, ms * 1e3,
         bytes / ms / 1e9, bytes / blocks / (ms * 1e-3 * 2.4e9) * (blocks > 256 ? blocks / 256.0 : 1.0), blocks);
}

int main() {
  const int64_t total = 2ll << 30;
  unsigned char* d; unsigned* sink;
  hipMalloc(&d, total); hipMalloc(&sink, 64);
  hipMemset(d, 1, total);
  const int64_t bpb = 4 << 20;                                                   // 4 MB per block and pass
  // (a) stream: 256 blocks x 4 MB = 1 GB, no byte shared between CUs (HBM / MALL-sourced after the first pass: 1 GB > 256 MB MALL)
  run<1, false>("stream 1 GB (HBM)", d, bpb, total, 256, 0, sink);
  run<2, false>("stream 1 GB (HBM)", d, bpb, total, 256, 0, sink);
  run<4, false>("stream 1 GB (HBM)", d, bpb, total, 256, 0, sink);
  run<8, false>("stream 1 GB (HBM)", d, bpb, total, 256, 0, sink);
  run<16, false>("stream 1 GB (HBM)", d, bpb, total, 256, 0, sink);
  run<2, true>("stream 1 GB (HBM) + prefetch wave", d, bpb, total, 256, 4, sink);
  run<2, true>("stream 1 GB (HBM) + prefetch wave", d, bpb, total, 256, 8, sink);
  run<4, true>("stream 1 GB (HBM) + prefetch wave", d, bpb, total, 256, 8, sink);
  run<4, true>("stream 1 GB (HBM) + prefetch wave", d, bpb, total, 256, 16, sink);
  // (b) MALL-sized: 256 blocks x 512 KB = 128 MB
  run<2, false>("stream 128 MB (MALL)", d, 512 << 10, total, 256, 0, sink);
  run<4, false>("stream 128 MB (MALL)", d, 512 << 10, total, 256, 0, sink);
  run<8, false>("stream 128 MB (MALL)", d, 512 << 10, total, 256, 0, sink);
  run<2, true>("stream 128 MB (MALL) + prefetch wave", d, 512 << 10, total, 256, 8, sink);
  // (c) L2-resident: every block reads the same 1 MB window
  run<2, false>("shared 1 MB window (L2)", d, 1 << 20, 1 << 20, 256, 0, sink);
  run<4, false>("shared 1 MB window (L2)", d, 1 << 20, 1 << 20, 256, 0, sink);
  run<8, false>("shared 1 MB window (L2)", d, 1 << 20, 1 << 20, 256, 0, sink);
  run<16, false>("shared 1 MB window (L2)", d, 1 << 20, 1 << 20, 256, 0, sink);
  // two blocks per CU
  run<4, false>("stream 1 GB, 512 blocks (2 per CU)", d, 2 << 20, total, 512, 0, sink);
  run<4, false>("shared 1 MB window, 512 blocks", d, 1 << 20, 1 << 20, 512, 0, sink);
  run_loop<false, false, false>(d, sink, "DMA only, no barrier");
  run_loop<false, false, true>(d, sink, "DMA + counted vmcnt + barrier");
  run_loop<true, false, false>(d, sink, "DMA + fragment reads");
  run_loop<true, false, true>(d, sink, "DMA + fragment reads + barrier");
  run_loop<true, true, false>(d, sink, "DMA + fragment reads + MFMAs");
  run_loop<true, true, true>(d, sink, "DMA + fragment reads + MFMAs + barrier");
  run_rows(d, 128, 256, 1, 0, sink, "contiguous (pitch = 128)");
  run_rows(d, 512, 256, 4, 0, sink, "K = 256 weights");
  run_rows(d, 2048, 256, 16, 0, sink, "K = 1024 weights");
  run_rows(d, 4608, 256, 36, 0, sink, "layer3 3x3 weights (K = 2304)");
  run_rows(d, 4608 + 128, 256, 36, 0, sink, "same, pitch + 128");
  run_rows(d, 4096, 256, 32, 0, sink, "K = 2048 weights");
  run_rows(d, 4096 + 128, 256, 32, 0, sink, "same, pitch + 128");
  run_rows(d, 512, 160, 4, 1, sink, "layer3 pixels (256 ch), per-block rows");
  run_rows(d, 2048, 160, 16, 1, sink, "1024-ch pixels, per-block rows");
  run_rows(d, 2048 + 128, 160, 16, 1, sink, "same, pitch + 128");
  for (int m = 0; m <= 8; ++m) run_mix<4>(d, 256, m, sink);
  for (int m = 0; m <= 8; m += 2) run_mix<4>(d, 512, m, sink);
  // how fast can FEW CUs stream when HBM is not saturated
  run<4, false>("stream, 32 blocks only", d, bpb, total, 32, 0, sink);
  run<8, false>("stream, 32 blocks only", d, bpb, total, 32, 0, sink);
  run<16, false>("stream, 32 blocks only", d, bpb, total, 32, 0, sink);
  run<8, false>("stream, 128 blocks", d, bpb, total, 128, 0, sink);
  return 0;
}
