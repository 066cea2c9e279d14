// gemv_ksplit_probe.hip -- VERDICT r04 #3b, built once and measured: the decomposition DESIGN named as "what is left" for the
// decode GEMVs whose 10-row work-groups spend their requests on ACTIVATIONS (wo, w2: 8 rows x K of activations per 10 rows
// x K of weights) -- fewer, larger work-groups per matrix (TILES x 10 rows) that each take 1 / KSL of K, so that a
// work-group's activation requests shrink by KSL x TILES / 1 per weight byte, and a fixed-order cross-work-group reduction:
// partial sums through a global buffer, an arrival counter per row block, the LAST arriver adds the slices in slice order
// and runs the epilogue (linear_skinny_kernel<..., KSL>).  Next to it: the shipped launcher (256 work-groups x 10 rows x
// all of K) and the same decomposition WITHOUT the fix-up (partials only: the upper bound of what it could win).
// Every launch streams a different weight copy (no cache reuse), M = 8.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I fish_speech_amd/csrc tools/gemv_ksplit_probe.hip fish_speech_amd/csrc/common.cpp -o tools/bin/gemv_ksplit_probe
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
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

template <int UNR, int TILES, int KSL>
static void launch_ksl(LinearArgs a, hipStream_t s) {
  a.wp = a.wr;
  hipLaunchKernelGGL((linear_skinny_kernel<8, EPI_RESIDUAL, false, UNR, TILES, 8, true, false, 10, 0, KSL>), dim3(a.N / (10 * TILES), KSL), dim3(512), 0, s, a);
}

typedef void (*Launch)(LinearArgs, hipStream_t);

int main() {
  struct Shape { const char* name; int N, K; } shapes[] = {{"wo   N=2560 K=4096", 2560, 4096}, {"w2   N=2560 K=9728", 2560, 9728}};
  const int iters = 200, M = 8;
  for (const Shape& sh : shapes) {
    const size_t elems = (size_t)sh.N * sh.K;
    const double bytes = (double)elems * 2;
    const int nbuf = (int)(2.0e9 / bytes) + 1;
    const RowPlan plan = skinny_row_plan(sh.N, sh.K, EPI_RESIDUAL);
    if (!plan.ok || plan.rows != 10) { printf("unexpected row plan\n"); return 1; }
    bf16_t* rowmajor; CK(hipMalloc((void**)&rowmajor, elems * 2));
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, 0, rowmajor, elems, 7u, 0.05f);
    bf16_t* w16; CK(hipMalloc((void**)&w16, elems * 2));
    launch_pack_weight(rowmajor, w16, sh.N, sh.K, 0, 0);
    std::vector<bf16_t*> wr(nbuf);
    for (auto& p : wr) { CK(hipMalloc((void**)&p, (size_t)plan.elems * 2)); launch_repack_rows(w16, p, sh.N, sh.K, EPI_RESIDUAL, plan, 0); }
    bf16_t *x, *res, *out, *ref;
    CK(hipMalloc((void**)&x, (size_t)16 * sh.K * 2)); CK(hipMalloc((void**)&res, (size_t)16 * sh.N * 2));
    CK(hipMalloc((void**)&out, (size_t)16 * sh.N * 2)); CK(hipMalloc((void**)&ref, (size_t)16 * sh.N * 2));
    hipLaunchKernelGGL(fill_kernel, dim3((16 * sh.K + 255) / 256), dim3(256), 0, 0, x, (size_t)16 * sh.K, 11u, 1.0f);
    hipLaunchKernelGGL(fill_kernel, dim3((16 * sh.N + 255) / 256), dim3(256), 0, 0, res, (size_t)16 * sh.N, 17u, 1.0f);
    float* part; CK(hipMalloc((void**)&part, (size_t)8 * 256 * 4 * 256 * 4)); unsigned* cnt; CK(hipMalloc((void**)&cnt, 4096)); CK(hipMemset(cnt, 0, 4096));
    CK(hipDeviceSynchronize());
    LinearArgs a{};
    a.x = x; a.ldx = sh.K; a.eps = 1e-6f; a.res = res; a.ldr = sh.N; a.out = ref; a.ldo = sh.N; a.M = M; a.N = sh.N; a.K = sh.K; a.epi = EPI_RESIDUAL;
    a.wp = w16; a.wr = wr[0]; a.part = part;
    launch_linear_skinny(a, 0);   // reference result: the shipped kernel
    CK(hipDeviceSynchronize());
    std::vector<bf16_t> href((size_t)M * sh.N), hgot((size_t)M * sh.N);
    CK(hipMemcpy(href.data(), ref, href.size() * 2, hipMemcpyDeviceToHost));
    a.out = out;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](auto&& launch) {
      for (int i = 0; i < 3; ++i) { a.wr = wr[i % nbuf]; launch(a); }
      CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
      for (int i = 0; i < iters; ++i) { a.wr = wr[i % nbuf]; launch(a); }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      return ms * 1e3f / iters;
    };
    printf("%s (%.1f MB, M = %d)\n", sh.name, bytes / 1e6, M);
    float us = timeit([&](LinearArgs b) { launch_linear_skinny(b, 0); });
    printf("  shipped: 256 work-groups x 10 rows x K                      %7.2f us  %5.0f GB/s\n", us, bytes / us * 1e-3);
    struct V { const char* name; Launch fn; } vs[] = {
        {"TILES 2 x KSL 2 (128 x 2 work-groups), 2 pairs in flight", launch_ksl<2, 2, 2>},
        {"TILES 2 x KSL 2 (128 x 2 work-groups), 4 pairs in flight", launch_ksl<4, 2, 2>},
        {"TILES 4 x KSL 4 ( 64 x 4 work-groups), 1 pair  in flight", launch_ksl<1, 4, 4>},
        {"TILES 4 x KSL 4 ( 64 x 4 work-groups), 2 pairs in flight", launch_ksl<2, 4, 4>},
        {"TILES 4 x KSL 8 ( 64 x 8 work-groups), 1 pair  in flight", launch_ksl<1, 4, 8>},
        {"TILES 2 x KSL 4 (128 x 4 work-groups), 2 pairs in flight", launch_ksl<2, 2, 4>}};
    for (auto& v : vs) {
      a.cnt = nullptr;
      const float ub = timeit([&](LinearArgs b) { v.fn(b, 0); });
      a.cnt = cnt;
      a.part_proto = 0;
      const float fenced = timeit([&](LinearArgs b) { v.fn(b, 0); });
      a.part_proto = 1;
      const float real = timeit([&](LinearArgs b) { v.fn(b, 0); });
      a.wr = wr[0];
      CK(hipMemset(out, 0, (size_t)16 * sh.N * 2));
      v.fn(a, 0);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(hgot.data(), out, hgot.size() * 2, hipMemcpyDeviceToHost));
      double worst = 0; size_t diff = 0;
      for (size_t i = 0; i < hgot.size(); ++i) { const double d = fabs((double)bf2f(hgot[i]) - (double)bf2f(href[i])); worst = d > worst ? d : worst; diff += hgot[i] != href[i]; }
      // twice more: the bits must not depend on which work-group arrives last
      std::vector<bf16_t> again(hgot.size()); bool stable = true;
      for (int r = 0; r < 2; ++r) { v.fn(a, 0); CK(hipDeviceSynchronize()); CK(hipMemcpy(again.data(), out, again.size() * 2, hipMemcpyDeviceToHost)); stable &= memcmp(again.data(), hgot.data(), again.size() * 2) == 0; }
      printf("  %s: partials only %7.2f us, fix-up with release/acquire fences %7.2f us, fix-up with agent-scope stores / loads %7.2f us  (%5.0f GB/s); vs shipped: %zu of %zu values differ, max |d| %.4f; repeatable %s\n",
             v.name, ub, fenced, real, bytes / real * 1e-3, diff, hgot.size(), worst, stable ? "yes" : "NO");
      fflush(stdout);
    }
    hipFree(rowmajor); hipFree(w16); for (auto p : wr) hipFree(p); hipFree(x); hipFree(res); hipFree(out); hipFree(ref); hipFree(part); hipFree(cnt);
  }
  return 0;
}
