// runahead_bench.hip -- can ONE persistent launch stream the weights of a chain of dependent GEMVs without the
// per-launch ramp?  Model of the slow transformer's decode step (36 layers x [wqkv, attention, wo, w13, w2] at the S2
// shapes, batch 8): every GEMV is a region of weight bytes split evenly over 256 work-groups (one per CU); a GEMV may
// only be CONSUMED after a grid barrier (everybody finished the previous one) and a read of its activation vector,
// but its weights do not depend on anything, so producer waves keep up to 128 KiB per CU of LDS-DMA loads in flight
// across the barriers (run-ahead).  No arithmetic: the consumer waves read every staged byte out of LDS once.
//   mode 0: one launch per GEMV (hipGraph of 180 launches) -- what the shipped decode step does
//   mode 1: one persistent launch, barriers (one atomic counter) + activation reads, run-ahead ring
//   mode 3: the same with one flag per work-group (stores; every wave polls the 256 flags)
//   mode 2: the same launch without barriers and activation reads (streaming ceiling of this structure)
// Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/runahead_bench.hip -o /tmp/runahead && timeout 120 /tmp/runahead
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int NPROD = 4, NCONS = 4;          // waves
constexpr int STAGE = 16 * 1024;             // bytes per ring slot: each producer wave brings 4 KiB (4 DMA instructions)
constexpr int SPIN_CAP = 1 << 18;

struct Seg {
  int64_t off;      // byte offset of the region (work-group w owns [off + w * nst * STAGE, ...))
  int nst;          // stages per work-group (0: a barrier-only step, e.g. attention)
  int xbytes;       // activation bytes every work-group reads after the barrier
};

template <int D>
__global__ __launch_bounds__(512, 1) void chain_kernel(const char* __restrict__ w, const Seg* __restrict__ segs, int seg0,
                                                       int nseg, const u32x4* __restrict__ xbuf, unsigned* bar,
                                                       int use_bar, unsigned* sink, unsigned* err) {
  extern __shared__ __attribute__((aligned(16))) char ring[];   // D slots + flags
  volatile int* fill = reinterpret_cast<volatile int*>(ring + D * STAGE);   // [NPROD] stages landed
  volatile int* cons = fill + NPROD;                                       // [NCONS] stages consumed
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wg = blockIdx.x, nwg = gridDim.x;
  if (tid < NPROD + NCONS + 1) fill[tid] = 0;
  __syncthreads();

  if (wave < NPROD) {
    // ---------------- producer: stream this work-group's share of every GEMV through the ring
    const int p = wave;
    int g = 0;
    for (int s = seg0; s < seg0 + nseg; ++s) {
      const Seg sg = segs[s];
      const char* base = w + sg.off + (int64_t)wg * sg.nst * STAGE + p * 4096 + lane * 16;
      for (int c = 0; c < sg.nst; ++c, ++g) {
        if (g >= D) {   // slot free?  (stage g - D consumed by every consumer wave)
          bool free_ = true;
          for (int k = 0; k < NCONS; ++k) free_ &= cons[k] >= g - D + 1;
          if (!free_) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            fill[p] = g;   // everything issued so far has landed
            int it = 0;
            do {
              __builtin_amdgcn_s_sleep(1);
              free_ = true;
              for (int k = 0; k < NCONS; ++k) free_ &= cons[k] >= g - D + 1;
            } while (!free_ && ++it < SPIN_CAP);
            if (!free_) { if (lane == 0) atomicAdd(err, 1u); return; }
          }
        }
        char* dst = ring + (g % D) * STAGE + p * 4096;
        const char* src = base + (int64_t)c * STAGE;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          __builtin_amdgcn_global_load_lds((glb_void*)(src + j * 1024), (lds_void*)(dst + j * 1024), 16, 0, 0);
        // all but the newest D - 1 stages of this wave have landed
        if constexpr (D == 8) asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
        else if constexpr (D == 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        if (g + 1 - (D - 1) > 0) fill[p] = g + 1 - (D - 1);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    fill[p] = g;
  } else {
    // ---------------- consumer
    const int c4 = wave - NPROD;
    u32x4 acc = {0, 0, 0, 0};
    int g = 0;
    for (int s = seg0; s < seg0 + nseg; ++s) {
      const Seg sg = segs[s];
      if (use_bar > 0 && s > seg0) {   // everybody is done with the previous GEMV (its output is this one's input)
        int it = 0;
        if (use_bar == 1) {            // one counter, one atomic per wave
          while (__hip_atomic_load(&bar[s - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(nwg * NCONS) && ++it < SPIN_CAP)
            __builtin_amdgcn_s_sleep(1);
        } else {                       // one flag per work-group (plain stores), a wave polls all of them
          const unsigned* fl = bar + (int64_t)(s - 1) * nwg;
          for (;;) {
            unsigned v = 1;
            for (int i = lane; i < nwg; i += 64) v &= __hip_atomic_load(fl + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all(v != 0) || ++it >= SPIN_CAP) break;
            __builtin_amdgcn_s_sleep(1);
          }
          __atomic_thread_fence(__ATOMIC_ACQUIRE);
        }
        if (it >= SPIN_CAP) { if (lane == 0) atomicAdd(err, 1u); return; }
      }
      if (use_bar >= 0) {          // the activation vector (L2-resident): every work-group reads all of it
        const int n16 = sg.xbytes >> 4;
        for (int i = c4 * 64 + lane; i < n16; i += NCONS * 64) acc ^= xbuf[i];
      }
      for (int c = 0; c < sg.nst; ++c, ++g) {
        int it = 0;
        bool ok;
        do {
          ok = true;
          for (int k = 0; k < NPROD; ++k) ok &= fill[k] > g;
          if (!ok) __builtin_amdgcn_s_sleep(1);
        } while (!ok && ++it < SPIN_CAP);
        if (!ok) { if (lane == 0) atomicAdd(err, 1u); return; }
        const char* src = ring + (g % D) * STAGE + (c4 * 64 + lane) * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc ^= *reinterpret_cast<const u32x4*>(src + j * 4096);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) cons[c4] = g + 1;
      }
      if (use_bar > 0) {           // "write the outputs", then arrive
        __threadfence();
        if (use_bar == 1) {
          if (lane == 0) __hip_atomic_fetch_add(&bar[s], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else if (lane == 0) {      // the last consumer wave of the work-group raises its flag
          const int prev = __hip_atomic_fetch_add(const_cast<int*>(cons) + NCONS, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (prev == (s - seg0) * NCONS + NCONS - 1)
            __hip_atomic_store(bar + (int64_t)s * nwg + wg, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[0] = 1;
  }
}

int main(int argc, char** argv) {
  const int layers = argc > 1 ? atoi(argv[1]) : 36;
  const int nwg = argc > 2 ? atoi(argv[2]) : 256;
  constexpr int D = 8;
  // S2 slow-layer shapes (N x K bf16): wqkv 6144x2560, wo 2560x4096, w13 19456x2560, w2 2560x9728; batch 8 activations
  struct G { const char* name; int64_t N, K; };
  const G gemv[5] = {{"wqkv", 6144, 2560}, {"attn", 0, 256}, {"wo", 2560, 4096}, {"w13", 19456, 2560}, {"w2", 2560, 9728}};
  std::vector<Seg> segs;
  int64_t off = 0, real_bytes = 0;
  for (int l = 0; l < layers; ++l)
    for (const G& gm : gemv) {
      const int64_t bytes = gm.N * gm.K * 2;
      const int nst = (int)((bytes + (int64_t)nwg * STAGE - 1) / ((int64_t)nwg * STAGE));
      segs.push_back({off, nst, (int)(gm.K * 16)});
      off += (int64_t)nst * STAGE * nwg;
      real_bytes += bytes;
    }
  const int nseg = (int)segs.size();
  printf("chain: %d layers, %d steps, %.3f GB streamed (%.3f GB of real weights), %d work-groups, ring %d x %d KiB\n", layers, nseg,
         off * 1e-9, real_bytes * 1e-9, nwg, D, STAGE / 1024);
  char* w; CK(hipMalloc((void**)&w, (size_t)off));
  CK(hipMemset(w, 1, (size_t)off));
  Seg* dsegs; CK(hipMalloc((void**)&dsegs, nseg * sizeof(Seg)));
  CK(hipMemcpy(dsegs, segs.data(), nseg * sizeof(Seg), hipMemcpyHostToDevice));
  u32x4* xbuf; CK(hipMalloc((void**)&xbuf, 1 << 20)); CK(hipMemset(xbuf, 2, 1 << 20));
  unsigned *bar, *sink, *err;
  CK(hipMalloc((void**)&bar, (size_t)nseg * nwg * 4)); CK(hipMalloc((void**)&sink, 4)); CK(hipMalloc((void**)&err, 4));
  CK(hipMemset(err, 0, 4)); CK(hipMemset(sink, 0, 4));
  const size_t smem = (size_t)D * STAGE + 64;
  CK(hipFuncSetAttribute((const void*)chain_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

  // mode 0: a graph of one launch per step
  hipGraph_t graph; hipGraphExec_t gexec;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int s = 0; s < nseg; ++s)
    hipLaunchKernelGGL(chain_kernel<D>, dim3(nwg), dim3(512), smem, st, w, dsegs, s, 1, xbuf, bar, 0, sink, err);
  CK(hipStreamEndCapture(st, &graph));
  CK(hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0));
  auto report = [&](const char* what, float ms) {
    unsigned herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    printf("%-58s %8.3f ms  %7.1f GB/s  (%.2f us per step)%s\n", what, ms, off / ms * 1e-6, ms * 1e3 / nseg, herr ? "  ** SPIN CAP HIT **" : "");
    fflush(stdout);
    return herr;
  };
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, st));
    CK(hipGraphLaunch(gexec, st));
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep) report("mode 0: one launch per step (hipGraph)", ms);
  }
  for (int mode : {2, 1, 3}) {
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemsetAsync(bar, 0, (size_t)nseg * nwg * 4, st));
      CK(hipEventRecord(e0, st));
      hipLaunchKernelGGL(chain_kernel<D>, dim3(nwg), dim3(512), smem, st, w, dsegs, 0, nseg, xbuf, bar, mode == 1 ? 1 : mode == 3 ? 2 : -1, sink, err);
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep && report(mode == 1 ? "mode 1: persistent, counter barriers + activation reads" : mode == 3 ? "mode 3: persistent, flag barriers + activation reads" : "mode 2: persistent, no dependencies (ceiling)", ms)) return 1;
    }
  }
  return 0;
}
