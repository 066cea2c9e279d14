// persist_probe.hip -- how much of the per-launch ramp a persistent (one launch per frame) slow-layer kernel
// could recover on MI355X.  Streams the four weight matrices of a Dual-AR slow layer (w_qkv 31.5 MB, w_o 21 MB,
// w1|w3 99.6 MB, w2 49.8 MB) for L layers
//   mode 0: one kernel launch per matrix (what the library does today, minus the arithmetic)
//   mode 1: ONE persistent launch, a grid barrier after every matrix, no prefetch across the barrier
//   mode 2: as 1, but each workgroup issues the first loads of the NEXT matrix before it waits at the barrier
//           (legal in the real kernel: weights do not depend on the activations the barrier orders)
//   mode 3: barrier only (no streaming) -- the cost of one grid barrier incl. the release/acquire fences
// Every phase also writes 1 KiB per workgroup and, after the barrier, reads 40 KiB written by other workgroups
// (the activation exchange of a GEMV chain), so the fences have real work.
// build: hipcc --offload-arch=gfx950 -O3 -o persist_probe tools/persist_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x)                                                                     \
  do {                                                                            \
    hipError_t e = (x);                                                           \
    if (e != hipSuccess) {                                                        \
      printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__);            \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

constexpr int THREADS = 512;
constexpr int UNR = 8;  // 16 B loads in flight per thread

// BAR selects the barrier flavour (argv[3]): 0 seq_cst fences + acquire spin, 1 release/acquire fences + relaxed
// spin with s_sleep, 2 as 1 without sleep, 3 no fences at all (atomics only: the floor)
__constant__ int BAR;
__device__ inline void grid_barrier(unsigned* cnt, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    if (BAR == 0) {
      __threadfence();
      atomicAdd(cnt, 1u);
      while (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
      __threadfence();
    } else {
      if (BAR != 3) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target)
        if (BAR == 1) __builtin_amdgcn_s_sleep(1);
      if (BAR != 3) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
  }
  __syncthreads();
}

// stream [lo, hi) 16-byte words of w, grid-strided inside the workgroup's private contiguous share
__device__ inline u32x4 stream_share(const u32x4* __restrict__ w, size_t n16, u32x4 acc, const u32x4* pre, bool has_pre) {
  const size_t per = (n16 + gridDim.x - 1) / gridDim.x;
  const size_t lo = (size_t)blockIdx.x * per;
  const size_t hi = lo + per < n16 ? lo + per : n16;
  size_t i = lo + threadIdx.x;
  if (has_pre) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) acc ^= pre[u];
    i += (size_t)UNR * THREADS;
  }
  for (; i + (size_t)(UNR - 1) * THREADS < hi; i += (size_t)UNR * THREADS) {
    u32x4 v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) v[u] = __builtin_nontemporal_load(w + i + (size_t)u * THREADS);
#pragma unroll
    for (int u = 0; u < UNR; ++u) acc ^= v[u];
  }
  for (; i < hi; i += THREADS) acc ^= __builtin_nontemporal_load(w + i);
  return acc;
}

__device__ inline void prefetch_share(const u32x4* __restrict__ w, size_t n16, u32x4* pre) {
  const size_t per = (n16 + gridDim.x - 1) / gridDim.x;
  const size_t lo = (size_t)blockIdx.x * per;
  const size_t i = lo + threadIdx.x;
#pragma unroll
  for (int u = 0; u < UNR; ++u) pre[u] = __builtin_nontemporal_load(w + i + (size_t)u * THREADS);
}

struct Phases {
  size_t off[4];
  size_t n16[4];
};

__global__ __launch_bounds__(THREADS) void one_matrix(const u32x4* w, size_t n16, float* act, unsigned* sink) {
  u32x4 acc = {0, 0, 0, 0};
  // activation read (40 KiB, written by the previous launch)
  float a = 0.f;
  for (int i = threadIdx.x; i < 10240; i += THREADS) a += act[i];
  acc = stream_share(w, n16, acc, nullptr, false);
  if (threadIdx.x < 256) act[10240 + blockIdx.x * 256 + threadIdx.x] = a + (float)acc.x;
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

template <int MODE>
__global__ __launch_bounds__(THREADS) void persistent(const u32x4* base, size_t layer_n16, Phases rel, int layers,
                                                       float* act, unsigned* cnt, unsigned* sink) {
  u32x4 acc = {0, 0, 0, 0};
  u32x4 pre[UNR];
  bool has_pre = false;
  unsigned phase = 0;
  float a = 0.f;
  for (int l = 0; l < layers; ++l) {
    for (int p = 0; p < 4; ++p) {
      const u32x4* w = base + (size_t)l * layer_n16 + rel.off[p];
      for (int i = threadIdx.x; i < 10240; i += THREADS) a += __builtin_nontemporal_load(act + i);
      if (MODE != 3) acc = stream_share(w, rel.n16[p], acc, pre, has_pre);
      if (threadIdx.x < 256) act[10240 + blockIdx.x * 256 + threadIdx.x] = a + (float)acc.x;
      if (MODE == 2) {
        int np = p + 1, nl = l;
        if (np == 4) np = 0, nl = l + 1;
        if (nl < layers) {
          prefetch_share(base + (size_t)nl * layer_n16 + rel.off[np], rel.n16[np], pre);
          has_pre = true;
        } else {
          has_pre = false;
        }
      }
      ++phase;
      grid_barrier(cnt, phase * gridDim.x);
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u && a == 1.2345f) sink[0] = 1;
}

int main(int argc, char** argv) {
  const int layers = argc > 1 ? atoi(argv[1]) : 12;
  const int grid = argc > 2 ? atoi(argv[2]) : 256;
  const size_t mb[4] = {31457280, 20971520, 99614720, 49807360};
  Phases rel;
  size_t off = 0;
  for (int p = 0; p < 4; ++p) {
    rel.off[p] = off;
    rel.n16[p] = mb[p] / 16;
    off += mb[p] / 16;
  }
  const size_t layer_n16 = off;
  const size_t total = layer_n16 * 16 * (size_t)layers;
  u32x4* w;
  float* act;
  unsigned *cnt, *sink;
  CK(hipMalloc(&w, total));
  CK(hipMemset(w, 1, total));
  CK(hipMalloc(&act, (10240 + 1024 * 256) * 4));
  CK(hipMemset(act, 0, (10240 + 1024 * 256) * 4));
  CK(hipMalloc(&cnt, 4));
  CK(hipMalloc(&sink, 4));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int bar = argc > 3 ? atoi(argv[3]) : 0;
  CK(hipMemcpyToSymbol(HIP_SYMBOL(BAR), &bar, 4));
  const double gb = (double)total / 1e9;
  printf("layers %d  grid %d  barrier flavour %d  bytes/pass %.2f GB\n", layers, grid, bar, gb);
  for (int mode = 0; mode < 4; ++mode) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipMemsetAsync(cnt, 0, 4, s));
      CK(hipEventRecord(e0, s));
      if (mode == 0) {
        for (int l = 0; l < layers; ++l)
          for (int p = 0; p < 4; ++p)
            one_matrix<<<grid, THREADS, 0, s>>>(w + (size_t)l * layer_n16 + rel.off[p],
                                                 rel.n16[p], act, sink);
      } else if (mode == 1) {
        persistent<1><<<grid, THREADS, 0, s>>>(w, layer_n16, rel, layers, act, cnt, sink);
      } else if (mode == 2) {
        persistent<2><<<grid, THREADS, 0, s>>>(w, layer_n16, rel, layers, act, cnt, sink);
      } else {
        persistent<3><<<grid, THREADS, 0, s>>>(w, layer_n16, rel, layers, act, cnt, sink);
      }
      CK(hipEventRecord(e1, s));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    const char* names[4] = {"launch per matrix", "persistent, barrier", "persistent, barrier + prefetch", "barrier only"};
    if (mode < 3)
      printf("mode %d %-32s %8.3f ms  %6.2f TB/s  (%.1f us per matrix)\n", mode, names[mode], best, gb / best,
             best * 1e3 / (layers * 4));
    else
      printf("mode %d %-32s %8.3f ms  (%.2f us per barrier)\n", mode, names[mode], best, best * 1e3 / (layers * 4));
  }
  return 0;
}
