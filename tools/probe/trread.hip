// probe: semantics of ds_read_b64_tr_b16 on gfx950.  LDS holds u16 values lds[i] = i; every lane reads with its own address.
#include <hip/hip_runtime.h>
#include <stdint.h>
extern "C" __global__ void trread_probe(const int* __restrict__ lane_addr_bytes, uint16_t* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  uint32_t addr = (uint32_t)(uintptr_t)lds + (uint32_t)lane_addr_bytes[threadIdx.x];
  uint64_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  out[threadIdx.x * 4 + 0] = (uint16_t)(v & 0xffff);
  out[threadIdx.x * 4 + 1] = (uint16_t)((v >> 16) & 0xffff);
  out[threadIdx.x * 4 + 2] = (uint16_t)((v >> 32) & 0xffff);
  out[threadIdx.x * 4 + 3] = (uint16_t)((v >> 48) & 0xffff);
}
extern "C" int run_probe(const int* addr, uint16_t* out, void* stream) {
  hipLaunchKernelGGL(trread_probe, dim3(1), dim3(64), 0, (hipStream_t)stream, addr, out);
  return (int)hipGetLastError();
}
