// mall_resident_probe.hip -- can part of the fast transformer's weights stay resident in the 256 MiB Infinity Cache while
// the rest of a pass (~830 MB) streams by?  The frame re-streams the four fast layers nine times; with every weight load
// non-temporal (the shipped GEMV) a pass is a cyclic 830 MB sweep through an LRU of 256 MiB: no hits.  If non-temporal
// loads do not ALLOCATE in the Infinity Cache, a subset read with the default policy could survive the sweep.
//   1. read A (size a MB, policy pa) cold (after a 1 GiB default-policy flush)
//   2. read A again at once (hot)
//   3. read A, then sweep S (s MB) with policy ps, then time A again  <- the question
// for a in {24, 48, 96, 192} MB, ps in {nt, default}.  Bare streaming reads (16 B per lane, 8 in flight), 1024 x 256.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <bool NT>
__global__ __launch_bounds__(256) void read_kernel(const u32x4* __restrict__ p, size_t n, uint32_t* sink) {
  u32x4 acc = {0, 0, 0, 0};
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 7 * stride < n; i += 8 * stride) {
    u32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = NT ? __builtin_nontemporal_load(p + i + j * stride) : p[i + j * stride];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc ^= v[j];
  }
  for (; i < n; i += stride) acc ^= NT ? __builtin_nontemporal_load(p + i) : p[i];
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345) sink[0] = 1;
}

static void rd(bool nt, const void* p, size_t bytes, uint32_t* sink) {
  if (nt) hipLaunchKernelGGL(read_kernel<true>, dim3(1024), dim3(256), 0, 0, (const u32x4*)p, bytes / 16, sink);
  else hipLaunchKernelGGL(read_kernel<false>, dim3(1024), dim3(256), 0, 0, (const u32x4*)p, bytes / 16, sink);
}

int main() {
  uint32_t* sink; CK(hipMalloc((void**)&sink, 4));
  const size_t MB = 1 << 20;
  void *A, *S, *F;
  CK(hipMalloc(&A, 256 * MB)); CK(hipMemset(A, 1, 256 * MB));
  CK(hipMalloc(&S, 1024 * MB)); CK(hipMemset(S, 2, 1024 * MB));
  CK(hipMalloc(&F, 1024 * MB)); CK(hipMemset(F, 3, 1024 * MB));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timed = [&](bool nt, const void* p, size_t bytes) {
    CK(hipEventRecord(e0)); rd(nt, p, bytes, sink); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms * 1e3f;
  };
  printf("%6s %4s | %8s %8s | %14s %14s %14s %14s\n", "A MB", "polA", "cold us", "hot us", "sweep200 nt", "sweep600 nt", "sweep600 dflt", "sweep800 nt");
  for (int pa = 0; pa < 2; ++pa)
    for (size_t a : {24, 48, 96, 192}) {
      const size_t bytes = a * MB;
      float cold = 0, hot = 0, r[4] = {0, 0, 0, 0};
      const int reps = 6;
      for (int it = 0; it < reps; ++it) {
        rd(false, F, 1024 * MB, sink);
        cold += timed(pa, A, bytes);
        hot += timed(pa, A, bytes);
        const size_t sweeps[4] = {200, 600, 600, 800};
        const bool snt[4] = {true, true, false, true};
        for (int k = 0; k < 4; ++k) {
          rd(false, F, 1024 * MB, sink);
          rd(pa, A, bytes, sink);
          rd(pa, A, bytes, sink);
          rd(snt[k], S, sweeps[k] * MB, sink);
          r[k] += timed(pa, A, bytes);
        }
      }
      printf("%6zu %4s | %8.2f %8.2f | %14.2f %14.2f %14.2f %14.2f\n", a, pa ? "nt" : "dflt", cold / reps, hot / reps, r[0] / reps,
             r[1] / reps, r[2] / reps, r[3] / reps);
    }
  return 0;
}
