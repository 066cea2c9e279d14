// gemm_bench.hip -- the prefill GEMM variants (launch_linear_tiled: 'l' 4-wave LDS-staged, 'w' wave-specialised, 'x' the
// 8-compute-wave 128 x 256 tile) on the S2-Pro prefill shapes, M = 8 x 200 and 8 x 2048 rows (round 3).
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DFMI_WS_ABLATE=n] tools/gemm_bench.hip fish_speech_amd/csrc/dualar_kernels.hip fish_speech_amd/csrc/common.cpp -o tools/bin/gemm_bench
// Usage:  [GEMM_MS=m1,m2,...] gemm_bench [variants, default lw] [shape name] [M]
//   FMI_WS_ABLATE (resource ablation of the 'w' kernel, results are garbage): 1 = the loader waves issue only the first
//   two stages (no DMA in the steady state), 2 = no MFMA (operand reads only), 3 = no operand reads (MFMA on stale registers)
// Every variant is checked bit for bit against 'l' (same products, same order) and timed over NBUF weight copies.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <map>
#include <algorithm>
#include "../fish_speech_amd/csrc/dualar_gemm.hip"

using namespace fmi;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Shape { const char* name; int N, K, epi; };

static uint32_t rng_state = 12345;
static inline uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state; }
static inline bf16_t rnd_bf16(float scale) {
  float f = ((int)(rnd() >> 8) - (1 << 23)) * (scale / (1 << 23));
  uint32_t u; memcpy(&u, &f, 4);
  return (bf16_t)(u >> 16);
}

// register-only MFMA loop: what the matrix cores deliver on this box (clock under load included) -- the ceiling the GEMM
// numbers below are to be read against.  WPS waves per SIMD, each with 4 independent accumulators.
__global__ __launch_bounds__(1024) void mfma_peak_kernel(float* out, int iters, long long* cycles) {
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f + i * 0.5f); }
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  const long long t1 = __builtin_readcyclecounter();
  float v = 0.f;
  for (int i = 0; i < 4; ++i) v += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = v;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

static void mfma_peak() {
  float* out; long long* cyc; CK(hipMalloc(&out, 256 * 8 * 1024 * 4)); CK(hipMalloc(&cyc, 8));
  for (int wps : {1, 2, 4}) {
    const int iters = 20000;
    dim3 grid(256 * 2), block(wps * 2 * 64);     // 2 work-groups per CU x (2 wps) waves = 4 wps waves per CU
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(mfma_peak_kernel, grid, block, 0, 0, out, iters, cyc);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(mfma_peak_kernel, grid, block, 0, 0, out, iters, cyc);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    const double flop = (double)grid.x * (block.x / 64) * iters * 16.0 * 16384.0;
    printf("mfma peak, %d wave(s)/SIMD: %.3f ms, %.0f TF/s, %lld counter ticks = %.1f per MFMA of the wave, tick rate %.0f MHz\n", wps, ms,
           flop / ms * 1e-9, c, (double)c / (iters * 16.0), c / (ms * 1e3));
  }
  hipFree(out); hipFree(cyc);
}

int main(int argc, char** argv) {
  mfma_peak();
  const Shape shapes[] = {{"wqkv", 6144, 2560, EPI_STORE}, {"wo", 2560, 4096, EPI_RESIDUAL}, {"w1|w3", 19456, 2560, EPI_SILU}, {"w2", 2560, 9728, EPI_RESIDUAL}};
  std::vector<int> Ms = {1600, 16384};
  if (getenv("GEMM_MS")) {      // e.g. GEMM_MS=200,400,800 : sweep of row counts
    Ms.clear();
    for (const char* p = getenv("GEMM_MS"); *p;) { Ms.push_back(atoi(p)); while (*p && *p != ',') ++p; if (*p) ++p; }
  }
  const char* variants = argc > 1 ? argv[1] : "lw";
  const char* only_shape = argc > 2 ? argv[2] : "";      // e.g. "w2"; "" = all
  const int only_m = argc > 3 ? atoi(argv[3]) : 0;       // 0 = both row counts
  const int NBUF = 3;
  for (int M : Ms) {
    if (only_m && M != only_m) continue;
    double total[8] = {0};
    for (const Shape& sh : shapes) {
      if (only_shape[0] && strcmp(only_shape, sh.name)) continue;
      const int n_out = sh.epi == EPI_SILU ? sh.N / 2 : sh.N;
      std::vector<bf16_t> hw((size_t)sh.N * sh.K), hx((size_t)M * sh.K), hr((size_t)M * n_out);
      for (auto& v : hw) v = rnd_bf16(0.05f);
      for (auto& v : hx) v = rnd_bf16(1.0f);
      for (auto& v : hr) v = rnd_bf16(1.0f);
      bf16_t *raw, *x, *res, *out, *ref; std::vector<bf16_t*> w(NBUF);
      CK(hipMalloc(&raw, hw.size() * 2)); CK(hipMalloc(&x, hx.size() * 2)); CK(hipMalloc(&res, hr.size() * 2));
      CK(hipMalloc(&out, hr.size() * 2)); CK(hipMalloc(&ref, hr.size() * 2));
      CK(hipMemcpy(raw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
      CK(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
      CK(hipMemcpy(res, hr.data(), hr.size() * 2, hipMemcpyHostToDevice));
      for (auto& p : w) {
        CK(hipMalloc(&p, hw.size() * 2));
        int rc;
        if (sh.epi == EPI_SILU) {   // gate rows and up rows interleaved per 16-row tile, like the w1 / w3 loads
          rc = launch_pack_weight(raw, p, sh.N / 2, sh.K, 1, 0);
          if (!rc) rc = launch_pack_weight(raw + (size_t)(sh.N / 2) * sh.K, p, sh.N / 2, sh.K, 2, 0);
        } else {
          rc = launch_pack_weight(raw, p, sh.N, sh.K, 0, 0);
        }
        if (rc) { printf("pack failed\n"); return 1; }
      }
      CK(hipDeviceSynchronize());
      LinearArgs a{};
      a.x = x; a.ldx = sh.K; a.res = res; a.ldr = n_out; a.out = ref; a.ldo = n_out; a.M = M; a.N = sh.N; a.K = sh.K; a.epi = sh.epi; a.wp = w[0];
      if (launch_linear_tiled(a, 0, false, 1)) { printf("launch failed: %s\n", g_last_error.c_str()); return 1; }
      CK(hipDeviceSynchronize());
      const double flop = 2.0 * M * sh.N * sh.K;
      printf("M=%5d %-6s N=%5d K=%5d :", M, sh.name, sh.N, sh.K);
      int vi = 0;
      for (const char* v = variants; *v; ++v, ++vi) {
        // 'a' = what the library selects for the shape (variant 0): the 256-column tile by row count, and for w2 the
        // contraction split in three + reduce pass (round 6: another fp32 order, so "DIFF" vs the reference is expected there)
        const int variant = *v == 'a' ? 0 : *v == 'l' ? 1 : *v == 'w' ? 2 : *v == 'x' ? 3 : *v == 'y' ? 4 : *v == 'p' ? 7 : *v == 'q' ? 8 : *v == 'r' ? 9 : *v == 's' ? 10 : *v == 't' ? 11 : 12;
        a.out = out; a.wp = w[0];
        float* part = nullptr;
        if (variant == 0 && linear_tiled_part_floats(M, sh.N, sh.K) > 0) CK(hipMalloc((void**)&part, (size_t)linear_tiled_part_floats(M, sh.N, sh.K) * 4));
        a.part = part;
        CK(hipMemset(out, 0xff, hr.size() * 2));
        if (launch_linear_tiled(a, 0, false, variant)) { printf("launch failed: %s\n", g_last_error.c_str()); return 1; }
        CK(hipDeviceSynchronize());
        std::vector<bf16_t> got(hr.size()), want(hr.size());
        CK(hipMemcpy(got.data(), out, got.size() * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(want.data(), ref, want.size() * 2, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < got.size(); ++i) bad += got[i] != want[i];
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int iters = M > 4000 ? 10 : 40;
        for (int i = 0; i < 3; ++i) { a.wp = w[i % NBUF]; launch_linear_tiled(a, 0, false, variant); }
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < iters; ++i) { a.wp = w[i % NBUF]; launch_linear_tiled(a, 0, false, variant); }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / iters;
        total[vi] += us;
        printf("  %c %8.1f us %6.0f TF/s%s", *v, us, flop / us * 1e-6, bad ? (part ? " (split)" : " DIFF") : "");
        if (part) { CK(hipDeviceSynchronize()); hipFree(part); a.part = nullptr; }
#if defined(FMI_Y_TIMING)
        if (variant >= 4) {
          long long t[8]; CK(hipMemcpyFromSymbol(t, HIP_SYMBOL(g_ytime), sizeof(t)));
          printf(" [wg(0,0) us: prologue %.2f loop %.2f epilogue %.2f]", (t[1] - t[0]) * 0.01, (t[2] - t[1]) * 0.01, (t[3] - t[2]) * 0.01);
          if (getenv("GEMM_WGLOG")) {   // per-CU timeline of the LAST launch: busy ticks, gaps between consecutive work-groups
            const int nwg = ((sh.N / 16 + 15) / 16) * ((M + 255) / 256);
            std::vector<long long> lg((size_t)8192 * 4);
            CK(hipMemcpyFromSymbol(lg.data(), HIP_SYMBOL(g_ylog), lg.size() * 8));
            std::map<long long, std::vector<std::pair<long long, long long>>> cu;
            long long tmin = 1LL << 62, tmax = 0;
            for (int i = 0; i < nwg && i < 8192; ++i) {
              const long long hw = lg[i * 4 + 2], xcc = lg[i * 4 + 3] & 0xf;
              const long long key = (xcc << 16) | (hw & 0xff00);   // xcc | se, sh, cu bits
              cu[key].push_back({lg[i * 4], lg[i * 4 + 1]});
              if (lg[i * 4]) tmin = std::min(tmin, lg[i * 4]);
              tmax = std::max(tmax, lg[i * 4 + 1]);
            }
            double busy = 0, gaps = 0; long long ngap = 0; size_t mx = 0, mn = 1 << 30;
            for (auto& kv : cu) {
              auto& v = kv.second; std::sort(v.begin(), v.end());
              mx = std::max(mx, v.size()); mn = std::min(mn, v.size());
              for (size_t j = 0; j < v.size(); ++j) { busy += v[j].second - v[j].first; if (j) { gaps += v[j].first - v[j - 1].second; ++ngap; } }
            }
            printf("\n      [%d wgs on %zu CUs (%zu..%zu per CU): span %.1f us, mean busy %.2f us per wg, mean gap %.2f us]",
                   nwg, cu.size(), mn, mx, (tmax - tmin) * 0.01, busy / nwg * 0.01, ngap ? gaps / ngap * 0.01 : 0.0);
          }
        }
#endif
      }
      printf("\n");
      hipFree(raw); hipFree(x); hipFree(res); hipFree(out); hipFree(ref); for (auto p : w) hipFree(p);
    }
    printf("M=%5d layer total:", M);
    int vi = 0;
    for (const char* v = variants; *v; ++v, ++vi) printf("  %c %8.1f us (%.3f of 2.5 PF)", *v, total[vi], 2.0 * M * 101007360.0 / total[vi] * 1e-6 / 2.5e3);
    printf("\n");
  }
  return 0;
}
