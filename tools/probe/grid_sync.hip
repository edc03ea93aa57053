// Probe: what does ONE grid-wide dependency cost on this chip -- as a hipGraph node boundary (today's decode step: ~21 dependent
// launches per token, 5-7 us each) and as an in-kernel synchronisation of a persistent kernel?  Every round, every workgroup needs
// the WHOLE 768-element vector the previous round produced (each workgroup produces one slice of it): the dependency structure of a
// decoded token's stages (decode.py).  Protocols:
//   graph    : R dependent kernel nodes in one hipGraph (the baseline)
//   counter  : persistent kernel, one agent-scope atomic counter per round (arrive = atomicAdd, wait = spin on a load)
//   tree     : persistent kernel, counters per group of G workgroups + one top counter
//   flags    : persistent kernel, no read-modify-write at all: producer w stores flag[w] = round after its slice (vmcnt(0) between),
//              a consumer's lanes poll all flags
// Inter-workgroup data and flags move with agent-scope (sc1) loads / stores: correct wherever the workgroups are placed (the 8 XCD
// L2s are not coherent with each other), no L2 write-back / invalidate.  Every spin is bounded (a lost wake-up ends the kernel with
// an error code instead of hanging the box).
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/grid_sync tools/probe/grid_sync.hip ; ./tools/probe/grid_sync      (GPU box)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int VEC = 768;
constexpr unsigned SPIN_MAX = 4000000u;       // ~ a second of polling: bail out instead of hanging

__device__ __forceinline__ unsigned ld_agent(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ unsigned mix(unsigned s, unsigned i) { return (s ^ (i * 0x9E3779B9u)) * 0x85EBCA6Bu + 0x165667B1u; }

// the work of a round: sum the whole previous vector (every workgroup needs all of it), write this workgroup's slice of the next one
__device__ __forceinline__ void round_body(const unsigned* prev, unsigned* next, int wg, int nwg, unsigned* sh) {
  unsigned s = 0;
  for (int i = threadIdx.x; i < VEC; i += blockDim.x) s += ld_agent(prev + i);
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  unsigned t = 0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sh[w];
  __syncthreads();
  const int per = (VEC + nwg - 1) / nwg;
  for (int i = threadIdx.x; i < per; i += blockDim.x) {
    const int j = wg * per + i;
    if (j < VEC) st_agent(next + j, mix(t, (unsigned)j));
  }
}

__global__ __launch_bounds__(256) void node_kernel(const unsigned* prev, unsigned* next) {
  __shared__ unsigned sh[4];
  round_body(prev, next, blockIdx.x, gridDim.x, sh);
}

struct SyncK { unsigned* buf0; unsigned* buf1; unsigned* counters; unsigned* flags; int rounds; int group; unsigned* err; };

// MODE 0 counter, 1 tree, 2 flags
template <int MODE>
__global__ __launch_bounds__(256) void persist_kernel(SyncK p) {
  __shared__ unsigned sh[4];
  __shared__ unsigned ok;
  const int wg = blockIdx.x, nwg = gridDim.x;
  for (int r = 0; r < p.rounds; ++r) {
    const unsigned* prev = (r & 1) ? p.buf1 : p.buf0;
    unsigned* next = (r & 1) ? p.buf0 : p.buf1;
    round_body(prev, next, wg, nwg, sh);
    // ---- publish this workgroup's slice, wait for everybody's ----
    __builtin_amdgcn_s_waitcnt(0);                  // every lane's slice stores acknowledged (vmcnt(0) / lgkmcnt(0))
    __syncthreads();
    const unsigned target = (unsigned)(r + 1);
    if (threadIdx.x == 0) ok = 1;
    if (MODE == 0) {
      if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(p.counters, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned n = 0;
        while (ld_agent(p.counters) < target * (unsigned)nwg) { if (++n > SPIN_MAX) { ok = 0; break; } }
      }
    } else if (MODE == 1) {
      if (threadIdx.x == 0) {
        const int G = p.group, grp = wg / G, ngrp = (nwg + G - 1) / G;
        const int gsize = min(G, nwg - grp * G);
        unsigned* gc = p.counters + 64 * (1 + grp);                  // one counter per 256-byte line
        const unsigned old = __hip_atomic_fetch_add(gc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == target * (unsigned)gsize) __hip_atomic_fetch_add(p.counters, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // last of the group
        unsigned n = 0;
        while (ld_agent(p.counters) < target * (unsigned)ngrp) { if (++n > SPIN_MAX) { ok = 0; break; } }
      }
    } else {
      if (threadIdx.x == 0) st_agent(p.flags + wg, target);
      // lanes poll: thread t watches flags t, t + 256, ...
      unsigned n = 0;
      bool all = false;
      while (!all) {
        all = true;
        for (int i = threadIdx.x; i < nwg; i += blockDim.x) all = all && (ld_agent(p.flags + i) >= target);
        all = __syncthreads_and(all);
        if (++n > SPIN_MAX / 64) { if (threadIdx.x == 0) ok = 0; break; }
      }
    }
    __syncthreads();
    if (!ok) { if (threadIdx.x == 0) atomicAdd(p.err, 1u); return; }
  }
}

static void host_ref(std::vector<unsigned>& v, int rounds, int nwg) {
  std::vector<unsigned> nx(VEC);
  for (int r = 0; r < rounds; ++r) {
    unsigned t = 0;
    for (int i = 0; i < VEC; ++i) t += v[i];
    for (int j = 0; j < VEC; ++j) nx[j] = (t ^ ((unsigned)j * 0x9E3779B9u)) * 0x85EBCA6Bu + 0x165667B1u;
    v.swap(nx);
  }
}

int main(int argc, char** argv) {
  const int R = argc > 1 ? atoi(argv[1]) : 460;
  unsigned *b0, *b1, *cnt, *flg, *err;
  CK(hipMalloc(&b0, VEC * 4)); CK(hipMalloc(&b1, VEC * 4)); CK(hipMalloc(&cnt, 64 * 64 * 4)); CK(hipMalloc(&flg, 1024 * 4)); CK(hipMalloc(&err, 4));
  std::vector<unsigned> init(VEC), ref(VEC), got(VEC);
  for (int i = 0; i < VEC; ++i) init[i] = 1000003u * i + 7;
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto check = [&](const char* what, int nwg, float ms) {
    ref = init; host_ref(ref, R, nwg);
    CK(hipMemcpy(got.data(), (R & 1) ? b1 : b0, VEC * 4, hipMemcpyDeviceToHost));
    unsigned e; CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
    bool same = true; for (int i = 0; i < VEC; ++i) same = same && got[i] == ref[i];
    printf("%-10s nwg %4d  %8.2f us total  %6.2f us per round   %s%s\n", what, nwg, ms * 1000.f, ms * 1000.f / R, same ? "ok" : "WRONG", e ? "  (spin bail-out!)" : "");
    fflush(stdout);
  };
  // ---- baseline: a hipGraph of R dependent kernel nodes ----
  for (int nwg : {32, 64, 128, 256}) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int r = 0; r < R; ++r) node_kernel<<<nwg, 256, 0, st>>>((r & 1) ? b1 : b0, (r & 1) ? b0 : b1);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    float best = 1e9f;
    for (int it = 0; it < 5; ++it) {
      CK(hipMemcpy(b0, init.data(), VEC * 4, hipMemcpyHostToDevice)); CK(hipMemset(err, 0, 4));
      CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    check("graph", nwg, best);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  // ---- persistent kernels ----
  for (int mode = 0; mode < 3; ++mode)
    for (int nwg : {32, 64, 128, 256}) {
      for (int group : {8, 16}) {
        if (mode != 1 && group != 8) continue;
        SyncK p{b0, b1, cnt, flg, R, group, err};
        float best = 1e9f;
        for (int it = 0; it < 5; ++it) {
          CK(hipMemcpy(b0, init.data(), VEC * 4, hipMemcpyHostToDevice)); CK(hipMemset(err, 0, 4));
          CK(hipMemset(cnt, 0, 64 * 64 * 4)); CK(hipMemset(flg, 0, 1024 * 4));
          CK(hipEventRecord(e0, st));
          if (mode == 0) persist_kernel<0><<<nwg, 256, 0, st>>>(p);
          else if (mode == 1) persist_kernel<1><<<nwg, 256, 0, st>>>(p);
          else persist_kernel<2><<<nwg, 256, 0, st>>>(p);
          CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        char name[32]; snprintf(name, sizeof name, mode == 0 ? "counter" : (mode == 1 ? "tree/%d" : "flags"), group);
        check(name, nwg, best);
      }
    }
  return 0;
}
