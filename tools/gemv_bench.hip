// gemv_bench.hip -- micro-benchmark of the skinny (decode) linear on MI355X: the SHIPPED launcher per S2-Pro shape
// (row-balanced copies included, as the frame runs them) at M = 1 / 8 / 12 / 16 / 24 / 32, next to a bare streaming read of the
// same bytes.  Kernel variants are selected by the launcher's environment switches, so A/B = two runs:
//   FMI_GEMV_NOHOLD=1    norm-fused variants with the round-3 prologue (statistics pass + fragments)
//   FMI_GEMV_LATE_EPI=1  residual / scale / bias loaded after the last barrier (rounds 1-3)
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemv_bench.hip \
//                               fish_speech_amd/csrc/common.cpp -o tools/bin/gemv_bench && tools/bin/gemv_bench
// Every launch streams a different weight copy (round-robin over ~2 GB: no cache reuse).  GEMV_CHECK=1 also compares
// every row of the M = 8 ... 32 results with the M = 1 result of that row (batch invariance, bit for bit).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../fish_speech_amd/csrc/dualar_kernels.hip"

using namespace fmi;

__global__ __launch_bounds__(256) void stream_read_kernel(const u32x4* __restrict__ p, size_t n, uint32_t* sink) {
  u32x4 acc = {0, 0, 0, 0};
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll 8
  for (; i < n; i += stride) {
    u32x4 v = __builtin_nontemporal_load(p + i);
    acc ^= v;
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345) sink[0] = 1;
}

__global__ void fill_kernel(bf16_t* p, size_t n, uint32_t seed, float scale) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t x = (uint32_t)i * 0x9E3779B1u + seed;
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  p[i] = f2bf(((float)(x & 0xffff) / 32768.f - 1.f) * scale);
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Shape { const char* name; int N, K, epi; bool norm; };

static float time_launches(LinearArgs a, const std::vector<bf16_t*>& w16, const std::vector<bf16_t*>& wr, int iters) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) { a.wp = w16[w % w16.size()]; a.wr = wr.empty() ? nullptr : wr[w % wr.size()]; launch_linear_skinny(a, 0); }
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) { a.wp = w16[i % w16.size()]; a.wr = wr.empty() ? nullptr : wr[i % wr.size()]; launch_linear_skinny(a, 0); }
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / iters;
}

int main() {
  const int iters = 200;
  const bool check = getenv("GEMV_CHECK") != nullptr;
  const size_t stream_bytes = (size_t)1 << 30;
  void* big; CK(hipMalloc(&big, stream_bytes)); CK(hipMemset(big, 1, stream_bytes));
  uint32_t* sink; CK(hipMalloc((void**)&sink, 4));
  for (size_t mb : {21, 32, 50, 100}) {
    size_t bytes = mb << 20; hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(stream_read_kernel, dim3(2048), dim3(256), 0, 0, (const u32x4*)((char*)big + (size_t)(i % 8) * (128 << 20)), bytes / 16, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("stream read %zu MiB per launch: %.2f us/launch, %.0f GB/s\n", mb, ms * 10, bytes * 100 / (ms * 1e-3) * 1e-9);
  }
  hipFree(big);
  Shape shapes[] = {{"w13  N=19456 K=2560 swiglu+norm", 19456, 2560, EPI_SILU, true}, {"wqkv N=6144 K=2560 store+norm", 6144, 2560, EPI_STORE, true},
                    {"wo   N=2560 K=4096 residual", 2560, 4096, EPI_RESIDUAL, false}, {"w2   N=2560 K=9728 residual", 2560, 9728, EPI_RESIDUAL, false},
                    {"head N=4096 K=2560 store+norm (fast_output / live LM head)", 4096, 2560, EPI_STORE, true}};
  printf("switches: FMI_GEMV_NOHOLD=%s FMI_GEMV_LATE_EPI=%s\n", getenv("FMI_GEMV_NOHOLD") ? getenv("FMI_GEMV_NOHOLD") : "0",
         getenv("FMI_GEMV_LATE_EPI") ? getenv("FMI_GEMV_LATE_EPI") : "0");
  for (const Shape& sh : shapes) {
    const double bytes = (double)sh.N * sh.K * 2;
    const size_t elems = (size_t)sh.N * sh.K;
    const int nbuf = (int)(2.0e9 / bytes) + 1;
    std::vector<bf16_t*> w16(nbuf), wr;
    bf16_t* rowmajor; CK(hipMalloc((void**)&rowmajor, elems * 2));
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, 0, rowmajor, elems, 7u, 0.05f);
    for (auto& p : w16) {
      CK(hipMalloc((void**)&p, elems * 2));
      if (sh.epi == EPI_SILU) { launch_pack_weight(rowmajor, p, sh.N / 2, sh.K, 1, 0); launch_pack_weight(rowmajor + elems / 2, p, sh.N / 2, sh.K, 2, 0); }
      else launch_pack_weight(rowmajor, p, sh.N, sh.K, 0, 0);
    }
    const bool rows = skinny_rows_supported(sh.N, sh.K, sh.epi, sh.norm);
    const RowPlan plan = skinny_row_plan(sh.N, sh.K, sh.epi);
    if (rows) {
      wr.resize(nbuf);
      for (int i = 0; i < nbuf; ++i) { CK(hipMalloc((void**)&wr[i], (size_t)plan.elems * 2)); launch_repack_rows(w16[i], wr[i], sh.N, sh.K, sh.epi, plan, 0); }
    }
    CK(hipDeviceSynchronize());
    hipFree(rowmajor);
    const int n_out = sh.epi == EPI_SILU ? sh.N / 2 : sh.N;
    bf16_t *x, *nw, *res, *out, *out1;
    CK(hipMalloc((void**)&x, (size_t)32 * sh.K * 2)); CK(hipMalloc((void**)&nw, (size_t)sh.K * 2));
    CK(hipMalloc((void**)&res, (size_t)32 * n_out * 2)); CK(hipMalloc((void**)&out, (size_t)32 * n_out * 2)); CK(hipMalloc((void**)&out1, (size_t)32 * n_out * 2));
    hipLaunchKernelGGL(fill_kernel, dim3((32 * sh.K + 255) / 256), dim3(256), 0, 0, x, (size_t)32 * sh.K, 11u, 1.0f);
    hipLaunchKernelGGL(fill_kernel, dim3((sh.K + 255) / 256), dim3(256), 0, 0, nw, (size_t)sh.K, 13u, 1.0f);
    hipLaunchKernelGGL(fill_kernel, dim3((32 * n_out + 255) / 256), dim3(256), 0, 0, res, (size_t)32 * n_out, 17u, 1.0f);
    printf("%s  (%.1f MB%s)\n", sh.name, bytes / 1e6, rows ? ", row-balanced copy" : "");
    LinearArgs a{};
    a.x = x; a.ldx = sh.K; a.norm_w = sh.norm ? nw : nullptr; a.eps = 1e-6f; a.res = res; a.N = sh.N; a.K = sh.K; a.epi = sh.epi;
    a.ldr = n_out; a.out = out; a.ldo = n_out;
    for (int M : {1, 8, 12, 16, 24, 32}) {
      a.M = M;
      const float us = time_launches(a, w16, wr, iters);
      printf("  M=%2d shipped launcher : %7.2f us  %6.0f GB/s\n", M, us, bytes / us * 1e-3);
      fflush(stdout);
    }
    if (check) {  // row r of the M-row result == the M = 1 result of that row
      std::vector<bf16_t> ref((size_t)32 * n_out), got((size_t)32 * n_out);
      for (int r = 0; r < 32; ++r) {
        LinearArgs b = a; b.M = 1; b.x = x + (size_t)r * sh.K; b.res = res + (size_t)r * n_out; b.out = out1 + (size_t)r * n_out;
        b.wp = w16[0]; b.wr = rows ? wr[0] : nullptr;
        launch_linear_skinny(b, 0);
      }
      CK(hipMemcpy(ref.data(), out1, ref.size() * 2, hipMemcpyDeviceToHost));
      for (int M : {8, 12, 16, 17, 24, 32}) {
        LinearArgs b = a; b.M = M; b.wp = w16[0]; b.wr = rows ? wr[0] : nullptr;
        CK(hipMemset(out, 0, (size_t)32 * n_out * 2));
        launch_linear_skinny(b, 0);
        CK(hipMemcpy(got.data(), out, got.size() * 2, hipMemcpyDeviceToHost));
        const bool same = memcmp(ref.data(), got.data(), (size_t)M * n_out * 2) == 0;
        printf("  M=%2d rows == their M=1 results: %s\n", M, same ? "yes (bit for bit)" : "NO");
      }
    }
    for (auto p : w16) hipFree(p);
    for (auto p : wr) hipFree(p);
    hipFree(x); hipFree(nw); hipFree(res); hipFree(out); hipFree(out1);
  }
  return 0;
}
