// gemv_rows_bench.hip -- the row-balanced decode GEMV (linear_skinny_kernel<..., ROWS < 16>) against the shipped
// 16-row tiling, on the S2-Pro decode shapes, M = 8 (round 3).
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemv_rows_bench.hip fish_speech_amd/csrc/common.cpp -o tools/bin/gemv_rows_bench
// Every variant is (1) checked BIT FOR BIT against the 16-row kernel on random weights (the row-balanced copy holds
// the same products in the same order) and (2) timed over NBUF distinct weight copies round-robin (no cache reuse).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../fish_speech_amd/csrc/dualar_kernels.hip"

using namespace fmi;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Shape { const char* name; int N, K, epi; bool norm; };

static uint32_t rng_state = 12345;
static inline uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state; }
static inline bf16_t rnd_bf16(float scale) {   // uniform in (-scale, scale), bf16-truncated
  float f = ((int)(rnd() >> 8) - (1 << 23)) * (scale / (1 << 23));
  uint32_t u; memcpy(&u, &f, 4);
  return (bf16_t)(u >> 16);
}

struct Bufs {
  std::vector<bf16_t*> w16, wr;   // 16-row packed copies and their row-balanced repacks
  bf16_t *x, *nw, *res, *out, *out_ref;
  RowPlan plan;
};

template <int WAVES, int EPI, bool NORM, int UNR, int TILES, int ROWS>
static float time_rows(const Shape& sh, Bufs& b, int M, int iters, bool check) {
  const RowPlan& p = b.plan;
  LinearArgs a{};
  a.x = b.x; a.ldx = sh.K; a.norm_w = NORM ? b.nw : nullptr; a.eps = 1e-6f; a.res = b.res; a.M = M; a.N = sh.N; a.K = sh.K; a.epi = EPI;
  const int n_out = EPI == EPI_SILU ? sh.N / 2 : sh.N;
  a.ldr = n_out; a.out = b.out; a.ldo = n_out;
  // the work-group count follows from ROWS x TILES (variants may regroup the plan's tiles)
  const int total_tiles = p.wgs * p.tiles;
  if (ROWS != p.rows || total_tiles % TILES) return -1.f;
  dim3 grid(total_tiles / TILES), block(WAVES * 64);
  if (check) {
    CK(hipMemset(b.out, 0xff, (size_t)16 * n_out * 2));
    a.wp = b.wr[0];
    hipLaunchKernelGGL((linear_skinny_kernel<WAVES, EPI, NORM, UNR, TILES, 8, true, false, ROWS>), grid, block, 0, 0, a);
    CK(hipDeviceSynchronize());
    std::vector<bf16_t> got((size_t)M * n_out), want((size_t)M * n_out);
    CK(hipMemcpy(got.data(), b.out, got.size() * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(want.data(), b.out_ref, want.size() * 2, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < got.size(); ++i) bad += got[i] != want[i];
    if (bad) { printf("  !! W=%d UNR=%d TILES=%d ROWS=%d: %zu of %zu outputs differ from the 16-row kernel\n", WAVES, UNR, TILES, ROWS, bad, got.size()); return -2.f; }
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) { a.wp = b.wr[w % b.wr.size()]; hipLaunchKernelGGL((linear_skinny_kernel<WAVES, EPI, NORM, UNR, TILES, 8, true, false, ROWS>), grid, block, 0, 0, a); }
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) { a.wp = b.wr[i % b.wr.size()]; hipLaunchKernelGGL((linear_skinny_kernel<WAVES, EPI, NORM, UNR, TILES, 8, true, false, ROWS>), grid, block, 0, 0, a); }
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / iters;
}

static float time_shipped16(const Shape& sh, Bufs& b, int M, int iters, bool write_ref) {
  LinearArgs a{};
  a.x = b.x; a.ldx = sh.K; a.norm_w = sh.norm ? b.nw : nullptr; a.eps = 1e-6f; a.res = b.res; a.M = M; a.N = sh.N; a.K = sh.K; a.epi = sh.epi;
  const int n_out = sh.epi == EPI_SILU ? sh.N / 2 : sh.N;
  a.ldr = n_out; a.out = write_ref ? b.out_ref : b.out; a.ldo = n_out;
  if (write_ref) { a.wp = b.w16[0]; launch_linear_skinny(a, 0); CK(hipDeviceSynchronize()); return 0.f; }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) { a.wp = b.w16[w % b.w16.size()]; launch_linear_skinny(a, 0); }
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) { a.wp = b.w16[i % b.w16.size()]; launch_linear_skinny(a, 0); }
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / iters;
}

// the split-K boundaries follow the wave count: only the 8-wave variants can (and must) equal the shipped kernel's bits
#define R(W, U, T, ROWS_, EPI_, NORM_) do { float us = time_rows<W, EPI_, NORM_, U, T, ROWS_>(sh, b, M, iters, W == 8); \
    if (us > 0) { printf("  rows=%2d W=%2d PAIRS=%d TILES=%d wgs=%4d : %7.2f us  %6.0f GB/s%s\n", ROWS_, W, U, T, b.plan.wgs * b.plan.tiles / T, us, bytes / us * 1e-3, W == 8 ? "  (bit-identical to the 16-row kernel)" : ""); fflush(stdout); } } while (0)

int main() {
  const int M = 8, iters = 200;
  Shape shapes[] = {{"wo   N=2560 K=4096 residual", 2560, 4096, EPI_RESIDUAL, false}, {"w2   N=2560 K=9728 residual", 2560, 9728, EPI_RESIDUAL, false},
                    {"wqkv N=6144 K=2560 store+norm", 6144, 2560, EPI_STORE, true}};
  // (w1|w3 with 19 + 19 rows x 512 work-groups was measured in round 3 and dropped: 21.65 us against 20.9-22.0 us for
  // the 16-row tiles, profiles/r03_gemv_rows_bench.txt)
  for (const Shape& sh : shapes) {
    const double bytes = (double)sh.N * sh.K * 2;
    const int nbuf = (int)(1.5e9 / bytes) + 1;
    Bufs b;
    b.plan = skinny_row_plan(sh.N, sh.K, sh.epi);
    if (!b.plan.ok) { printf("%s: no row plan\n", sh.name); continue; }
    // host: random row-major weights -> device -> pack (16-row) -> repack (rows)
    std::vector<bf16_t> hw((size_t)sh.N * sh.K);
    for (auto& v : hw) v = rnd_bf16(0.05f);
    bf16_t* raw; CK(hipMalloc((void**)&raw, hw.size() * 2));
    CK(hipMemcpy(raw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    b.w16.resize(nbuf); b.wr.resize(nbuf);
    for (int i = 0; i < nbuf; ++i) {
      CK(hipMalloc((void**)&b.w16[i], (size_t)bytes));
      CK(hipMalloc((void**)&b.wr[i], (size_t)b.plan.elems * 2));
      if (sh.epi == EPI_SILU) {   // w1 = first half of the rows, w3 = second half, interleaved in 16-row blocks
        if (launch_pack_weight(raw, b.w16[i], sh.N / 2, sh.K, 1, 0) || launch_pack_weight(raw + (size_t)(sh.N / 2) * sh.K, b.w16[i], sh.N / 2, sh.K, 2, 0)) { printf("pack failed: %s\n", g_last_error.c_str()); return 1; }
      } else if (launch_pack_weight(raw, b.w16[i], sh.N, sh.K, 0, 0)) { printf("pack failed\n"); return 1; }
      if (launch_repack_rows(b.w16[i], b.wr[i], sh.N, sh.K, sh.epi, b.plan, 0)) { printf("repack failed: %s\n", g_last_error.c_str()); return 1; }
    }
    CK(hipDeviceSynchronize());
    CK(hipFree(raw));
    std::vector<bf16_t> hx((size_t)16 * sh.K), hn(sh.K), hr((size_t)16 * sh.N);
    for (auto& v : hx) v = rnd_bf16(2.0f);
    for (auto& v : hn) v = rnd_bf16(1.5f);
    for (auto& v : hr) v = rnd_bf16(1.0f);
    CK(hipMalloc((void**)&b.x, hx.size() * 2)); CK(hipMemcpy(b.x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc((void**)&b.nw, hn.size() * 2)); CK(hipMemcpy(b.nw, hn.data(), hn.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc((void**)&b.res, hr.size() * 2)); CK(hipMemcpy(b.res, hr.data(), hr.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc((void**)&b.out, (size_t)16 * sh.N * 2)); CK(hipMalloc((void**)&b.out_ref, (size_t)16 * sh.N * 2));
    printf("%s  (%.1f MB, M=%d; plan: %d rows x %d tiles x %d work-groups, copy %.1f MB)\n", sh.name, bytes / 1e6, M, b.plan.rows, b.plan.tiles, b.plan.wgs, b.plan.elems * 2 / 1e6);
    time_shipped16(sh, b, M, iters, true);
    { float us = time_shipped16(sh, b, M, iters, false); printf("  shipped 16-row launcher                 : %7.2f us  %6.0f GB/s\n", us, bytes / us * 1e-3); fflush(stdout); }
    if (sh.epi == EPI_RESIDUAL) {
      R(8, 1, 1, 10, EPI_RESIDUAL, false); R(8, 2, 1, 10, EPI_RESIDUAL, false); R(8, 4, 1, 10, EPI_RESIDUAL, false); R(8, 8, 1, 10, EPI_RESIDUAL, false);
      R(16, 1, 1, 10, EPI_RESIDUAL, false); R(16, 2, 1, 10, EPI_RESIDUAL, false); R(16, 4, 1, 10, EPI_RESIDUAL, false);
      R(4, 4, 1, 10, EPI_RESIDUAL, false); R(4, 8, 1, 10, EPI_RESIDUAL, false);
    } else if (sh.epi == EPI_STORE) {
      R(8, 1, 2, 12, EPI_STORE, true); R(8, 2, 2, 12, EPI_STORE, true); R(8, 5, 2, 12, EPI_STORE, true);
      R(8, 1, 1, 12, EPI_STORE, true); R(8, 2, 1, 12, EPI_STORE, true); R(8, 5, 1, 12, EPI_STORE, true);
      R(16, 1, 2, 12, EPI_STORE, true); R(16, 2, 2, 12, EPI_STORE, true); R(4, 2, 2, 12, EPI_STORE, true); R(4, 5, 2, 12, EPI_STORE, true);
    }
    { float us = time_shipped16(sh, b, M, iters, false); printf("  shipped 16-row launcher (again)         : %7.2f us  %6.0f GB/s\n", us, bytes / us * 1e-3); fflush(stdout); }
    for (auto p : b.w16) hipFree(p);
    for (auto p : b.wr) hipFree(p);
    hipFree(b.x); hipFree(b.nw); hipFree(b.res); hipFree(b.out); hipFree(b.out_ref);
  }
  return 0;
}
