// engine_probe.hip -- prices of the two primitives a persistent decode layer is built from, at the
// BATCH-8 edge sizes of the S2-Pro fast layer, on MI355X:
//   barrier   XCD-hierarchical grid barrier (8 group counters of 32 arrivals -> one top counter -> 8 generation words
//             polled by 32 work-groups each), host-paired, 256 work-groups = one per CU, with / without fences
//   edge      one all-to-all activation edge: every work-group publishes its slice of a P-byte vector with
//             write-through (sc1) 8-byte stores, drains, arrives at the barrier; after it one agent acquire, then every
//             work-group reads the WHOLE vector (its eight consumer waves one k-slice each) with plain 16-byte loads.
//             P = 40 KiB (x: 8 x 2560 bf16), 64 KiB (attention output 8 x 4096), 152 KiB (SwiGLU output 8 x 9728);
//             16 KiB / 4 KiB = batch-1-sized edges, where the guide's price list was measured at
//   stream    the same edge while a loader wave per CU streams weights through an 8 x 16 KiB LDS ring with
//             global_load_lds (nt), as the engine's loader would: 'streaming' vs 'parked' prices
// The numbers go to profiles/r06_engine_probe.txt.  Every spin is bounded; a timeout prints FAILED instead of hanging.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/engine_probe.hip -o tools/bin/engine_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned gu32;
typedef __attribute__((address_space(1))) unsigned long long gu64;

constexpr int NWG = 256, GROUPS = 8, PER_GROUP = NWG / GROUPS;
constexpr unsigned SPIN_MAX = 1u << 22;

struct Sync {               // every word on its own 128-byte line
  unsigned grp[GROUPS][32];
  unsigned top[32];
  unsigned gen[GROUPS][32];
  unsigned fail[32];
};

__device__ __forceinline__ unsigned ld_rlx(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// one lane per work-group.  release / acquire: 0 = none (payload was stored write-through and drained / nothing to read)
__device__ __forceinline__ bool xcd_barrier_lane0(Sync* sy, unsigned e, bool release, bool acquire) {
  const int g = blockIdx.x & (GROUPS - 1);
  if (release) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  const unsigned prev = __hip_atomic_fetch_add(&sy->grp[g][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (prev + 1 == e * PER_GROUP) {
    const unsigned p2 = __hip_atomic_fetch_add(&sy->top[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (p2 + 1 == e * GROUPS) {
#pragma unroll
      for (int j = 0; j < GROUPS; ++j) __hip_atomic_store(&sy->gen[j][0], e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  bool ok = true;
  unsigned spins = 0;
  while (ld_rlx(&sy->gen[g][0]) < e) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > SPIN_MAX || ld_rlx(&sy->fail[0])) { __hip_atomic_store(&sy->fail[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = false; break; }
  }
  if (acquire) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  return ok;
}

// ---- barrier alone
__global__ __launch_bounds__(256) void barrier_kernel(Sync* sy, int iters, int fences) {
  for (int i = 1; i <= iters; ++i) {
    __syncthreads();
    if (threadIdx.x == 0) xcd_barrier_lane0(sy, (unsigned)i, fences != 0, fences != 0);
    __syncthreads();
  }
}

// ---- edges.  9 waves: wave 8 = the loader (stream != 0), waves 0-7 consumers / producers.
// buf: two P-byte vectors (ping-pong by epoch).  Work-group b owns bytes [b * P / 256, (b + 1) * P / 256) of the vector.
template <bool STREAM>
__global__ __launch_bounds__(576) void edge_kernel(Sync* sy, unsigned long long* buf, int P, int iters, const u32x4* weights,
                                                   size_t wbytes_per_wg, unsigned* sink, unsigned long long* bad) {
  extern __shared__ __attribute__((aligned(1024))) char ring[];   // 8 x 16 KiB
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  __shared__ int s_stop;
  __shared__ unsigned s_arr, s_go;
  if (tid == 0) { s_stop = 0; s_arr = 0; s_go = 0; }
  __syncthreads();   // (the only s_barrier: all nine waves are still here)
  if (wave == 8) {
    if (!STREAM) return;
    // loader: 16 KiB fills (16 x 1 KiB global_load_lds) round the ring until the consumers are done; at most 8 fills
    // (128 KiB) in flight -- nobody consumes, the ring only stands for the engine's weight stream
    const char* src = reinterpret_cast<const char*>(weights) + (size_t)blockIdx.x * wbytes_per_wg;
    size_t off = 0;
    unsigned fills = 0;
    while (!__hip_atomic_load(&s_stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
      char* dst = ring + (fills & 7) * 16384;
#pragma unroll
      for (int j = 0; j < 16; ++j)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + off + j * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, 0, 2 /* nt */);
      off += 16384;
      if (off + 16384 > wbytes_per_wg) off = 0;
      ++fills;
      asm volatile("s_waitcnt vmcnt(48)" ::: "memory");   // <= 3 older fills outstanding (vmcnt counts to 63)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) sink[blockIdx.x] = fills;
    return;
  }
  const int slice8 = P / NWG / 8;                   // 8-byte words this work-group publishes
  unsigned long long errs = 0;
  u32x4 acc = {0, 0, 0, 0};
  // the eight consumer waves meet through LDS words (s_barrier would wait for the loader wave too)
  auto lds_ld = [](unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
  for (int e = 1; e <= iters; ++e) {
    unsigned long long* v = buf + (size_t)(e & 1) * (P / 8);
    // produce: word w of my slice = (epoch << 32) | global word index
    if (tid < slice8) {
      const unsigned w = blockIdx.x * slice8 + tid;
      __hip_atomic_store((gu64*)(v + w), ((unsigned long long)e << 32) | w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its write-through stores
    if (lane == 0) __hip_atomic_fetch_add(&s_arr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (tid == 0) {
      unsigned spins = 0;
      bool ok = true;
      while (lds_ld(&s_arr) < 8u * (unsigned)e) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > SPIN_MAX) { ok = false; break; }
      }
      ok = ok && xcd_barrier_lane0(sy, (unsigned)e, false, true);
      __hip_atomic_store(&s_go, ok ? (unsigned)e : 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    unsigned go;
    while ((go = lds_ld(&s_go)) < (unsigned)e) __builtin_amdgcn_s_sleep(1);
    if (go == 0xffffffffu) break;
    if (wave != 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   // (wave 0 issued the agent acquire: L1 is per CU)
    // consume: wave w reads its k-slice (P / 8 bytes) of the vector, 16 bytes per lane per load
    const u32x4* src = reinterpret_cast<const u32x4*>(v) + (size_t)wave * (P / 8 / 16);
    const int n16 = P / 8 / 16;
    for (int i = lane; i < n16; i += 64 * 4) {
      u32x4 r[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) r[u] = i + u * 64 < n16 ? src[i + u * 64] : (u32x4){0, 0, 0, 0};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (i + u * 64 < n16) {
          const unsigned w0 = (unsigned)((size_t)wave * (P / 8 / 8) + (size_t)(i + u * 64) * 2);
          errs += (r[u].x != w0) + (r[u].y != (unsigned)e) + (r[u].z != w0 + 1) + (r[u].w != (unsigned)e);
        }
        acc ^= r[u];
      }
    }
  }
  if (tid == 0) __hip_atomic_store(&s_stop, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if (errs) atomicAdd(bad, errs);
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[0] = 1;
}


// ---- layer skeleton: the persistent fast layer WITHOUT its arithmetic -- what the structure alone costs.
// Per CU: one loader wave streams this work-group's share of the layer's weights (wqkv 8, wo 5, w1|w3 25, w2 12 fills of
// 16 KiB = 800 KiB per layer, the S2-Pro fast layer over 256 CUs) through the 8-slot ring, back to back across ops and
// layers (it runs ahead over every edge as far as the ring allows); three consumer waves read every landed slot out of
// LDS (ds_read_b128, a third each) and free it; at the end of an op they publish the work-group's output slice
// write-through, meet, cross the grid barrier, acquire, and read the next op's WHOLE input vector (k-slices over the
// three waves) before they touch that op's weights.  Edges per layer: qkv -> attention (1 KiB read per work-group),
// attention -> wo (64 KiB), wo -> w1|w3 (40 KiB), w1|w3 -> w2 (152 KiB), w2 -> next wqkv (40 KiB).
// edges = 0: the same stream with the consumers never waiting for anyone (the floor: what the loader/consumer ring
// itself sustains); edges = 1: barriers but nothing published / read; edges = 2: everything.
constexpr int NCONS = 3;
struct LayerPlan { int fills[5]; int in_bytes[5]; int out_bytes[5]; };   // op 1 = attention (no weights)

__global__ __launch_bounds__(256) void layer_kernel(Sync* sy, unsigned long long* buf, LayerPlan plan, int layers, int edges,
                                                    const u32x4* weights, size_t wbytes_per_wg, unsigned* sink) {
  extern __shared__ __attribute__((aligned(1024))) char ring[];   // 8 x 16 KiB
  __shared__ unsigned s_landed, s_cons[NCONS], s_arr, s_go, s_fail;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (tid == 0) { s_landed = 0; s_arr = 0; s_go = 0; s_fail = 0; for (int i = 0; i < NCONS; ++i) s_cons[i] = 0; }
  __syncthreads();
  auto lds_ld = [](unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
  auto lds_st = [](unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
  int fills_per_layer = 0;
  for (int o = 0; o < 5; ++o) fills_per_layer += plan.fills[o];
  const unsigned total = (unsigned)(fills_per_layer * layers);
  if (wave == NCONS) {   // ---- loader
    const char* src = reinterpret_cast<const char*>(weights) + (size_t)blockIdx.x * wbytes_per_wg;
    size_t off = 0;
    for (unsigned k = 0; k < total; ++k) {
      // slot k % 8 is free once every consumer is done with fill k - 8
      unsigned spins = 0;
      while (k >= 8) {
        unsigned m = lds_ld(&s_cons[0]);
#pragma unroll
        for (int i = 1; i < NCONS; ++i) { const unsigned c = lds_ld(&s_cons[i]); m = c < m ? c : m; }
        if (m + 8 > k) break;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > SPIN_MAX || lds_ld(&s_fail)) { lds_st(&s_fail, 1); break; }
      }
      if (lds_ld(&s_fail)) break;
      char* dst = ring + (k & 7) * 16384;
#pragma unroll
      for (int j = 0; j < 16; ++j)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + off + j * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, 0, 2 /* nt */);
      off += 16384;
      if (off + 16384 > wbytes_per_wg) off = 0;
      asm volatile("s_waitcnt vmcnt(32)" ::: "memory");   // fills <= k - 2 have landed
      if (k >= 2) lds_st(&s_landed, k - 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_st(&s_landed, total);
    return;
  }
  // ---- consumers
  u32x4 acc = {0, 0, 0, 0};
  unsigned k = 0, e = 0;
  bool ok = true;
  for (int l = 0; l < layers && ok; ++l) {
    for (int o = 0; o < 5 && ok; ++o) {
      for (int f = 0; f < plan.fills[o] && ok; ++f, ++k) {
        unsigned spins = 0;
        while (lds_ld(&s_landed) <= k) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > SPIN_MAX || lds_ld(&s_fail)) { lds_st(&s_fail, 1); ok = false; break; }
        }
        if (!ok) break;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const u32x4* slot = reinterpret_cast<const u32x4*>(ring + (k & 7) * 16384);
        // a third of the slot per wave: 1024 x 16 B / 3 -> lanes stride 64
        for (int i = wave * 64 + lane; i < 1024; i += NCONS * 64) acc ^= slot[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) lds_st(&s_cons[wave], k + 1);
      }
      if (!ok || edges == 0) continue;
      // ---- edge after op o
      ++e;
      unsigned long long* v = buf + (size_t)(e & 1) * (160 * 1024 / 8);
      if (edges == 2) {
        const int slice8 = plan.out_bytes[o] / NWG / 8;
        if (tid < slice8) {
          const unsigned w = blockIdx.x * slice8 + tid;
          __hip_atomic_store((gu64*)(v + w), ((unsigned long long)e << 32) | w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (lane == 0) __hip_atomic_fetch_add(&s_arr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (tid == 0) {
        unsigned spins = 0;
        bool good = true;
        while (lds_ld(&s_arr) < (unsigned)NCONS * e) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > SPIN_MAX) { good = false; break; }
        }
        good = good && xcd_barrier_lane0(sy, e, false, edges == 2);
        if (!good) lds_st(&s_fail, 1);
        lds_st(&s_go, good ? e : 0xffffffffu);
      }
      unsigned go;
      while ((go = lds_ld(&s_go)) < e) __builtin_amdgcn_s_sleep(1);
      if (go == 0xffffffffu) { ok = false; break; }
      if (edges == 2) {
        if (wave != 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const int nin = plan.in_bytes[(o + 1) % 5];     // the next op's input vector, k-sliced over the consumer waves
        const int n16 = nin / 16;
        const u32x4* src = reinterpret_cast<const u32x4*>(v);
        for (int i = wave * 64 + lane; i < n16; i += NCONS * 64 * 4) {
          u32x4 r[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) r[u] = i + u * NCONS * 64 < n16 ? src[i + u * NCONS * 64] : (u32x4){0, 0, 0, 0};
#pragma unroll
          for (int u = 0; u < 4; ++u) acc ^= r[u];
        }
      }
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[0] = 1;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 200;
  Sync* sy; CK(hipMalloc((void**)&sy, sizeof(Sync)));
  unsigned long long* buf; CK(hipMalloc((void**)&buf, 2 * 160 * 1024));
  unsigned* sink; CK(hipMalloc((void**)&sink, 4 * NWG));
  unsigned long long* bad; CK(hipMalloc((void**)&bad, 8));
  const size_t wbytes_per_wg = 4u << 20;
  u32x4* weights; CK(hipMalloc((void**)&weights, wbytes_per_wg * NWG)); CK(hipMemset(weights, 1, wbytes_per_wg * NWG));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto failed = [&]() { Sync h; CK(hipMemcpy(&h, sy, sizeof(Sync), hipMemcpyDeviceToHost)); return h.fail[0] != 0; };

  for (int fences = 0; fences <= 1; ++fences) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipMemset(sy, 0, sizeof(Sync)));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(barrier_kernel, dim3(NWG), dim3(256), 0, 0, sy, iters, fences);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best;
    }
    printf("barrier-xcd, 256 WGs, %s: %.2f us per barrier%s\n", fences ? "release + acquire fences" : "no fences",
           best * 1e3f / iters, failed() ? "  FAILED (timeout)" : "");
  }
  CK(hipFuncSetAttribute((const void*)edge_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  for (int stream = 0; stream <= 1; ++stream)
    for (int P : {4096, 16384, 40960, 65536, 155648}) {
      float best = 1e9f;
      unsigned long long hbad = 0;
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipMemset(sy, 0, sizeof(Sync)));
        CK(hipMemset(bad, 0, 8));
        CK(hipMemset(buf, 0, 2 * 160 * 1024));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        if (stream) hipLaunchKernelGGL(edge_kernel<true>, dim3(NWG), dim3(576), 131072, 0, sy, buf, P, iters, weights, wbytes_per_wg, sink, bad);
        else hipLaunchKernelGGL(edge_kernel<false>, dim3(NWG), dim3(576), 0, 0, sy, buf, P, iters, weights, wbytes_per_wg, sink, bad);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
        unsigned long long b; CK(hipMemcpy(&b, bad, 8, hipMemcpyDeviceToHost));
        hbad += b;
      }
      unsigned fills[NWG]; CK(hipMemcpy(fills, sink, sizeof(fills), hipMemcpyDeviceToHost));
      printf("edge %6d B (%4d B per WG), %s: %.2f us per edge (publish + barrier + acquire + read-all), stale/wrong words %llu%s",
             P, P / NWG, stream ? "loader streaming" : "parked          ", best * 1e3f / iters, hbad, failed() ? "  FAILED (timeout)" : "");
      if (stream) printf("  [loader: %.1f GB/s per CU]", (double)fills[1] * 16384 / (best * 1e-3) * 1e-9);
      printf("\n");
    }
  // layer skeleton: S2-Pro fast layer, 201.9 MB over 256 CUs = 50 fills of 16 KiB per work-group
  CK(hipFuncSetAttribute((const void*)layer_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  {
    LayerPlan plan{};
    const int fills[5] = {8, 0, 5, 25, 12};                       // wqkv, attention, wo, w1|w3, w2
    const int out_b[5] = {98304, 65536, 40960, 155648, 40960};    // what the op publishes (bytes, all work-groups together)
    const int in_b[5] = {40960, 1024, 65536, 40960, 155648};      // what a work-group reads before the op
    for (int o = 0; o < 5; ++o) { plan.fills[o] = fills[o]; plan.out_bytes[o] = out_b[o]; plan.in_bytes[o] = in_b[o]; }
    const int layers = 8;
    for (int edges = 0; edges <= 2; ++edges) {
      float best = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipMemset(sy, 0, sizeof(Sync)));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(layer_kernel, dim3(NWG), dim3(256), 131072, 0, sy, buf, plan, layers, edges, weights, wbytes_per_wg, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
      }
      const double bytes = 50.0 * 16384 * NWG;
      printf("layer skeleton (%d layers in one launch), %s: %.2f us per layer, %.0f GB/s%s\n", layers,
             edges == 0 ? "no edges (ring floor)           " : edges == 1 ? "5 barriers per layer, no payload" : "5 edges per layer, full payload ",
             best * 1e3f / layers, bytes / (best * 1e-3 / layers) * 1e-9, failed() ? "  FAILED (timeout)" : "");
    }
    printf("(the same layer as five hipGraph'd launches in the frame: 54 us, profiles/r05_kernels_step.txt)\n");
  }
  return 0;
}
