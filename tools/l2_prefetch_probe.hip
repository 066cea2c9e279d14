// l2_prefetch_probe.hip -- would a decode GEMV run faster if ITS work-group's weight tile already sat in ITS XCD's L2?
// The frame's attention kernels (8 / 6 us, 256 / 64 work-groups, a few MB of traffic) leave HBM idle right before every `wo`
// GEMV (21 MB = 2.6 MB per XCD, under the 4 MiB L2); a work-group that also READS the 80 KB tile the same-numbered GEMV
// work-group will stream would leave it in the L2 of the XCD both run on -- if work-groups of consecutive launches map to
// XCDs alike (linear id mod 8).  Measured here: the shipped wo / w2 / head GEMV (M = 8) cold, after a prefetch launch with
// the SAME work-group -> tile map (1-D grid, the attention kernel's 3-D grid, the fast attention's 64 work-groups), and with
// the map shifted by one work-group (then the tile lands in the neighbouring XCD's L2: the control).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I fish_speech_amd/csrc tools/l2_prefetch_probe.hip fish_speech_amd/csrc/common.cpp -o tools/bin/l2_prefetch_probe
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../fish_speech_amd/csrc/dualar_kernels.hip"
using namespace fmi;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void fill_kernel(bf16_t* p, size_t n, uint32_t seed, float scale) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t x = (uint32_t)i * 0x9E3779B1u + seed;
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  p[i] = f2bf(((float)(x & 0xffff) / 32768.f - 1.f) * scale);
}

// work-group with LINEAR id i reads the tiles (i + shift) % ntiles, + nwg, + 2 nwg ... (tile = tile_bytes contiguous)
template <bool NT>
__global__ void prefetch_kernel(const char* __restrict__ w, int tile_bytes, int ntiles, int shift, uint32_t* sink) {
  const int nthreads = blockDim.x;
  const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  const int nwg = gridDim.x * gridDim.y * gridDim.z;
  u32x4 acc = {0, 0, 0, 0};
  for (int t = lin; t < ntiles; t += nwg) {
    const u32x4* p = reinterpret_cast<const u32x4*>(w + (int64_t)((t + shift) % ntiles) * tile_bytes);
    const int n16 = tile_bytes / 16;
    for (int i = threadIdx.x; i < n16; i += nthreads * 4) {
      u32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = i + u * nthreads < n16 ? (NT ? __builtin_nontemporal_load(p + i + u * nthreads) : p[i + u * nthreads]) : acc;
#pragma unroll
      for (int u = 0; u < 4; ++u) acc ^= v[u];
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[0] = 1;
}

int main() {
  struct Shape { const char* name; int N, K, epi; bool norm; } shapes[] = {
      {"wo   N=2560 K=4096 (row-balanced copy, 10-row tiles)", 2560, 4096, EPI_RESIDUAL, false},
      {"w2   N=2560 K=9728 (row-balanced copy, 10-row tiles)", 2560, 9728, EPI_RESIDUAL, false},
      {"head N=4096 K=2560 norm (16-row tiles)", 4096, 2560, EPI_STORE, true}};
  uint32_t* sink; CK(hipMalloc((void**)&sink, 4));
  const int iters = 40, M = 8;
  for (const Shape& sh : shapes) {
    const size_t elems = (size_t)sh.N * sh.K;
    const double bytes = (double)elems * 2;
    const int nbuf = (int)(1.5e9 / bytes) + 2;
    const bool rows = skinny_rows_supported(sh.N, sh.K, sh.epi, sh.norm);
    const RowPlan plan = skinny_row_plan(sh.N, sh.K, sh.epi);
    bf16_t* rowmajor; CK(hipMalloc((void**)&rowmajor, elems * 2));
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, 0, rowmajor, elems, 7u, 0.05f);
    std::vector<bf16_t*> w16(nbuf), wr(nbuf, nullptr);
    for (int i = 0; i < nbuf; ++i) {
      CK(hipMalloc((void**)&w16[i], elems * 2));
      launch_pack_weight(rowmajor, w16[i], sh.N, sh.K, 0, 0);
      if (rows) { CK(hipMalloc((void**)&wr[i], (size_t)plan.elems * 2)); launch_repack_rows(w16[i], wr[i], sh.N, sh.K, sh.epi, plan, 0); }
    }
    const int ntiles = 256, tile_bytes = (int)(bytes / ntiles);
    bf16_t *x, *nw, *res, *out;
    CK(hipMalloc((void**)&x, (size_t)16 * sh.K * 2)); CK(hipMalloc((void**)&nw, (size_t)sh.K * 2));
    CK(hipMalloc((void**)&res, (size_t)16 * sh.N * 2)); CK(hipMalloc((void**)&out, (size_t)16 * sh.N * 2));
    hipLaunchKernelGGL(fill_kernel, dim3((16 * sh.K + 255) / 256), dim3(256), 0, 0, x, (size_t)16 * sh.K, 11u, 1.0f);
    hipLaunchKernelGGL(fill_kernel, dim3((sh.K + 255) / 256), dim3(256), 0, 0, nw, (size_t)sh.K, 13u, 1.0f);
    hipLaunchKernelGGL(fill_kernel, dim3((16 * sh.N + 255) / 256), dim3(256), 0, 0, res, (size_t)16 * sh.N, 17u, 1.0f);
    CK(hipDeviceSynchronize());
    LinearArgs a{};
    a.x = x; a.ldx = sh.K; a.norm_w = sh.norm ? nw : nullptr; a.eps = 1e-6f; a.res = res; a.ldr = sh.N; a.out = out; a.ldo = sh.N;
    a.M = M; a.N = sh.N; a.K = sh.K; a.epi = sh.epi;
    hipEvent_t e0, e1, p0; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&p0));
    printf("%s  (%.1f MB = %.2f MB per XCD, %d KB per tile)\n", sh.name, bytes / 1e6, bytes / 8e6, tile_bytes / 1024);
    struct Mode { const char* name; dim3 grid; int threads; int shift; bool on; bool nt; } modes[] = {
        {"cold (no prefetch)", dim3(1), 0, 0, false, false},
        {"prefetch (default policy), 1-D grid 256 x 512, same map", dim3(256), 512, 0, true, false},
        {"prefetch (default policy), map shifted by 1", dim3(256), 512, 1, true, false},
        {"prefetch (default policy), map shifted by 8", dim3(256), 512, 8, true, false},
        {"prefetch (default policy), map shifted by 128", dim3(256), 512, 128, true, false},
        {"prefetch (default policy), 3-D grid (8,8,4) x 512, same map", dim3(8, 8, 4), 512, 0, true, false},
        {"prefetch (default policy), 2-D grid (8,8) x 256, same map", dim3(8, 8), 256, 0, true, false},
        {"prefetch (NON-TEMPORAL), 1-D grid 256 x 512, same map", dim3(256), 512, 0, true, true},
        {"prefetch (NON-TEMPORAL), map shifted by 1", dim3(256), 512, 1, true, true},
        {"prefetch (NON-TEMPORAL), map shifted by 8", dim3(256), 512, 8, true, true},
        {"prefetch (NON-TEMPORAL), map shifted by 128", dim3(256), 512, 128, true, true},
        {"prefetch of ANOTHER copy (default policy): TLB / clock control", dim3(256), 512, 0, true, false}};
    for (auto& m : modes) {
      float tot = 0, ptot = 0;
      for (int it = 0; it < iters + 2; ++it) {
        const int k = it % nbuf;
        a.wp = w16[k]; a.wr = rows ? wr[k] : nullptr;
        const char* wsrc = reinterpret_cast<const char*>(rows ? wr[k] : w16[k]);
        CK(hipEventRecord(p0));
        const bool other = m.name[12] == 'A';   // the last mode touches the NEXT copy instead
        const char* psrc = other ? reinterpret_cast<const char*>(rows ? wr[(k + 1) % nbuf] : w16[(k + 1) % nbuf]) : wsrc;
        if (m.on && m.nt) hipLaunchKernelGGL(prefetch_kernel<true>, m.grid, dim3(m.threads), 0, 0, psrc, tile_bytes, ntiles, m.shift, sink);
        else if (m.on) hipLaunchKernelGGL(prefetch_kernel<false>, m.grid, dim3(m.threads), 0, 0, psrc, tile_bytes, ntiles, m.shift, sink);
        CK(hipEventRecord(e0));
        launch_linear_skinny(a, 0);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms, pms; CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipEventElapsedTime(&pms, p0, e0));
        if (it >= 2) { tot += ms; ptot += pms; }
      }
      printf("  %-72s GEMV %6.2f us   (prefetch launch %6.2f us)\n", m.name, tot * 1e3f / iters, ptot * 1e3f / iters);
      fflush(stdout);
    }
    for (int i = 0; i < nbuf; ++i) { hipFree(w16[i]); if (wr[i]) hipFree(wr[i]); }
    hipFree(rowmajor); hipFree(x); hipFree(nw); hipFree(res); hipFree(out);
  }
  return 0;
}
