// overlap_bench.hip -- can the launch ramp of a chain of dependent GEMVs be hidden WITHOUT a grid barrier?
// Model: the slow transformer's 36 x [wqkv, wo, w13, w2] weight streams (S2 shapes) as 144 launches of 256
// work-groups (one per CU, persistent over its share of the rows) x 512 threads, weights loaded straight into registers
// (4 x 16 B per lane in flight, like linear_skinny_kernel), each launch reading a 40 KiB activation vector first.
//   mode 0: one stream, one hipGraph chain -- kernel k starts when kernel k-1 has drained (what the decode step does)
//   mode 1: two graph branches, kernel k on branch k % 2: it is resident while k-1 still runs, loads its first weight
//           chunk, then waits on k-1's completion counter (256 arrivals) before it touches the activations.  Two
//           launches of 256 work-groups are always co-resident (2 x 512 threads per CU), so the wait cannot deadlock;
//           spins are bounded anyway.
//   mode 2: as mode 1 without the wait (upper bound: pure overlap, wrong results in a real kernel)
//   modes 3-5 (round 3): the same three WITHOUT a graph -- eager launches, kernel k on stream k % 2.  Two streams are two
//           hardware queues, so kernel k may really be resident while k-1 runs (stream order keeps k-2 finished before k
//           starts: at most two launches co-resident, 2 x 512 threads per CU, no deadlock); the host pays ~4 us per launch.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/overlap_bench.hip -o /tmp/overlap && timeout 120 /tmp/overlap
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int UNR = 4, SPIN_CAP = 1 << 16;

__global__ __launch_bounds__(512, 2) void gemv_like(const u32x4* __restrict__ w, int64_t n16_per_wg, const u32x4* __restrict__ x,
                                                    int xn16, unsigned* done, int k, int wait, unsigned* sink, unsigned* err) {
  const int tid = threadIdx.x;
  const u32x4* p = w + (int64_t)blockIdx.x * n16_per_wg;
  u32x4 acc = {0, 0, 0, 0};
  u32x4 buf[UNR];
  int64_t i = tid;
#pragma unroll
  for (int u = 0; u < UNR; ++u) buf[u] = i + u * 512 < n16_per_wg ? __builtin_nontemporal_load(p + i + u * 512) : acc;
  if (wait && k > 0) {   // the producer of our activations: all of its work-groups have arrived
    if (tid == 0) {
      int it = 0;
      while (__hip_atomic_load(&done[k - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x && ++it < SPIN_CAP)
        __builtin_amdgcn_s_sleep(2);
      if (it >= SPIN_CAP) atomicAdd(err, 1u);
    }
    __syncthreads();
  }
  for (int j = tid; j < xn16; j += 512) acc ^= x[j];          // activation vector (L2)
  for (;;) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) acc ^= buf[u];
    i += UNR * 512;
    if (i >= n16_per_wg) break;
#pragma unroll
    for (int u = 0; u < UNR; ++u) buf[u] = i + u * 512 < n16_per_wg ? __builtin_nontemporal_load(p + i + u * 512) : (u32x4){0, 0, 0, 0};
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[0] = 1;
  __syncthreads();
  if (tid == 0) {
    __threadfence();
    __hip_atomic_fetch_add(&done[k], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

int main(int argc, char** argv) {
  const int layers = argc > 1 ? atoi(argv[1]) : 36;
  const int nwg = 256;
  struct G { const char* name; int64_t N, K; };
  const G gemv[4] = {{"wqkv", 6144, 2560}, {"wo", 2560, 4096}, {"w13", 19456, 2560}, {"w2", 2560, 9728}};
  struct Seg { int64_t off16, n16_per_wg; int xn16; };
  std::vector<Seg> segs;
  int64_t off16 = 0;
  for (int l = 0; l < layers; ++l)
    for (const G& g : gemv) {
      const int64_t n16 = g.N * g.K * 2 / 16, per = (n16 + nwg - 1) / nwg;
      segs.push_back({off16, per, (int)(g.K * 16 / 16)});
      off16 += per * nwg;
    }
  const int n = (int)segs.size();
  const double bytes = off16 * 16.0;
  printf("chain of %d launches, %.3f GB, %d work-groups x 512 threads each\n", n, bytes * 1e-9, nwg);
  u32x4* w; CK(hipMalloc((void**)&w, (size_t)off16 * 16)); CK(hipMemset(w, 1, (size_t)off16 * 16));
  u32x4* x; CK(hipMalloc((void**)&x, 1 << 20)); CK(hipMemset(x, 2, 1 << 20));
  unsigned *done, *sink, *err;
  CK(hipMalloc((void**)&done, n * 4)); CK(hipMalloc((void**)&sink, 4)); CK(hipMalloc((void**)&err, 4));
  CK(hipMemset(sink, 0, 4)); CK(hipMemset(err, 0, 4));
  hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  hipEvent_t fork, join, e0, e1;
  CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int mode = (argc > 2 ? 3 : 0); mode < 3; ++mode) {   // (a second argument skips the graph modes)
    hipGraph_t graph; hipGraphExec_t gexec;
    CK(hipStreamBeginCapture(s0, hipStreamCaptureModeGlobal));
    CK(hipMemsetAsync(done, 0, n * 4, s0));
    if (mode > 0) { CK(hipEventRecord(fork, s0)); CK(hipStreamWaitEvent(s1, fork, 0)); }
    for (int k = 0; k < n; ++k) {
      hipStream_t st = (mode > 0 && (k & 1)) ? s1 : s0;
      hipLaunchKernelGGL(gemv_like, dim3(nwg), dim3(512), 0, st, w + segs[k].off16, segs[k].n16_per_wg, x, segs[k].xn16, done, k,
                         mode == 1 ? 1 : 0, sink, err);
    }
    if (mode > 0) { CK(hipEventRecord(join, s1)); CK(hipStreamWaitEvent(s0, join, 0)); }
    CK(hipStreamEndCapture(s0, &graph));
    CK(hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0));
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipEventRecord(e0, s0));
      CK(hipGraphLaunch(gexec, s0));
      CK(hipEventRecord(e1, s0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
      if (rep) printf("mode %d: %8.3f ms  %7.1f GB/s  %.2f us per launch%s\n", mode, ms, bytes / ms * 1e-6, ms * 1e3 / n, herr ? "  ** SPIN CAP HIT **" : "");
      fflush(stdout);
    }
    CK(hipGraphExecDestroy(gexec)); CK(hipGraphDestroy(graph));
  }
  for (int mode = 3; mode < 6; ++mode) {   // eager: 3 = one stream, 4 = two streams + completion-counter wait, 5 = two streams, no wait
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipMemsetAsync(done, 0, n * 4, s0));
      CK(hipMemsetAsync(err, 0, 4, s0));
      CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1));
      CK(hipEventRecord(e0, s0));
      if (mode > 3) { CK(hipEventRecord(fork, s0)); CK(hipStreamWaitEvent(s1, fork, 0)); }
      for (int k = 0; k < n; ++k) {
        hipStream_t st = (mode > 3 && (k & 1)) ? s1 : s0;
        hipLaunchKernelGGL(gemv_like, dim3(nwg), dim3(512), 0, st, w + segs[k].off16, segs[k].n16_per_wg, x, segs[k].xn16, done, k,
                           mode == 4 ? 1 : 0, sink, err);
      }
      if (mode > 3) { CK(hipEventRecord(join, s1)); CK(hipStreamWaitEvent(s0, join, 0)); }
      CK(hipEventRecord(e1, s0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
      if (rep) printf("mode %d (eager): %8.3f ms  %7.1f GB/s  %.2f us per launch%s\n", mode, ms, bytes / ms * 1e-6, ms * 1e3 / n, herr ? "  ** SPIN CAP HIT **" : "");
      fflush(stdout);
    }
  }
  return 0;
}
