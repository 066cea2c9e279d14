// mall_probe.hip -- does a GEMV run faster when its weights were just read by another kernel
// (Infinity Cache / L2 residency)?  Decides whether a concurrent weight-prefetch stream can pay.
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../fish_speech_amd/csrc/dualar_kernels.hip"
using namespace fmi;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void touch_kernel(const u32x4* __restrict__ p, size_t n, uint32_t* sink) {
  u32x4 acc = {0, 0, 0, 0};
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) acc ^= p[i];
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345) sink[0] = 1;
}

template <int EPI, bool NORM, int UNR, int TILES, bool NT>
float gemv_time(const bf16_t* w, const bf16_t* warm, size_t bytes, int N, int K, bf16_t* x, bf16_t* nw, bf16_t* res, bf16_t* out, uint32_t* sink, int reps) {
  LinearArgs a{};
  a.x = x; a.ldx = K; a.norm_w = NORM ? nw : nullptr; a.eps = 1e-6f; a.res = res; a.M = 8; a.N = N; a.K = K; a.epi = EPI;
  const int n_out = EPI == EPI_SILU ? N / 2 : N;
  a.ldr = n_out; a.out = out; a.ldo = n_out; a.wp = w;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float tot = 0;
  for (int r = 0; r < reps; ++r) {
    hipLaunchKernelGGL(touch_kernel, dim3(2048), dim3(256), 0, 0, (const u32x4*)warm, bytes / 16, sink);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((linear_skinny_kernel<8, EPI, NORM, UNR, TILES, 8, NT>), dim3(N / (16 * TILES)), dim3(512), 0, 0, a);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tot += ms;
  }
  return tot * 1e3f / reps;
}

int main() {
  uint32_t* sink; CK(hipMalloc((void**)&sink, 4));
  struct S { const char* name; int N, K; } shapes[] = {{"w13", 19456, 2560}, {"w2", 2560, 9728}, {"wqkv", 6144, 2560}, {"wo", 2560, 4096}};
  void* flush; const size_t fl = (size_t)1 << 30; CK(hipMalloc(&flush, fl)); CK(hipMemset(flush, 1, fl));
  for (auto& sh : shapes) {
    const size_t bytes = (size_t)sh.N * sh.K * 2;
    bf16_t *w, *other, *x, *nw, *res, *out;
    CK(hipMalloc((void**)&w, bytes)); CK(hipMemset(w, 0x11, bytes));
    CK(hipMalloc((void**)&other, bytes)); CK(hipMemset(other, 0x11, bytes));
    CK(hipMalloc((void**)&x, 16 * sh.K * 2)); CK(hipMemset(x, 0x3c, 16 * sh.K * 2));
    CK(hipMalloc((void**)&nw, sh.K * 2)); CK(hipMemset(nw, 0x3c, sh.K * 2));
    CK(hipMalloc((void**)&res, 16 * sh.N * 2)); CK(hipMemset(res, 0, 16 * sh.N * 2));
    CK(hipMalloc((void**)&out, 16 * sh.N * 2));
    float cold_nt, warm_nt, cold, warm;
    if (sh.N == 19456) {
      cold_nt = gemv_time<EPI_SILU, true, 2, 2, true>(w, (bf16_t*)flush, fl, sh.N, sh.K, x, nw, res, out, sink, 20);
      warm_nt = gemv_time<EPI_SILU, true, 2, 2, true>(w, w, bytes, sh.N, sh.K, x, nw, res, out, sink, 20);
      cold = gemv_time<EPI_SILU, true, 2, 2, false>(w, (bf16_t*)flush, fl, sh.N, sh.K, x, nw, res, out, sink, 20);
      warm = gemv_time<EPI_SILU, true, 2, 2, false>(w, w, bytes, sh.N, sh.K, x, nw, res, out, sink, 20);
    } else if (sh.K == 2560) {
      cold_nt = gemv_time<EPI_STORE, true, 2, 2, true>(w, (bf16_t*)flush, fl, sh.N, sh.K, x, nw, res, out, sink, 20);
      warm_nt = gemv_time<EPI_STORE, true, 2, 2, true>(w, w, bytes, sh.N, sh.K, x, nw, res, out, sink, 20);
      cold = gemv_time<EPI_STORE, true, 2, 2, false>(w, (bf16_t*)flush, fl, sh.N, sh.K, x, nw, res, out, sink, 20);
      warm = gemv_time<EPI_STORE, true, 2, 2, false>(w, w, bytes, sh.N, sh.K, x, nw, res, out, sink, 20);
    } else {
      cold_nt = gemv_time<EPI_RESIDUAL, false, 4, 1, true>(w, (bf16_t*)flush, fl, sh.N, sh.K, x, nw, res, out, sink, 20);
      warm_nt = gemv_time<EPI_RESIDUAL, false, 4, 1, true>(w, w, bytes, sh.N, sh.K, x, nw, res, out, sink, 20);
      cold = gemv_time<EPI_RESIDUAL, false, 4, 1, false>(w, (bf16_t*)flush, fl, sh.N, sh.K, x, nw, res, out, sink, 20);
      warm = gemv_time<EPI_RESIDUAL, false, 4, 1, false>(w, w, bytes, sh.N, sh.K, x, nw, res, out, sink, 20);
    }
    printf("%-5s %6.1f MB: GEMV after flushing 1 GiB  nt %.2f us / plain %.2f us ; after touching its own weights  nt %.2f us / plain %.2f us\n",
           sh.name, bytes / 1e6, cold_nt, cold, warm_nt, warm);
    hipFree(w); hipFree(other); hipFree(x); hipFree(nw); hipFree(res); hipFree(out);
  }
  return 0;
}
