// gemv_bench.hip -- micro-benchmark of the skinny (decode) linear kernel variants on MI355X.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemv_bench.hip \
//                               fish_speech_amd/csrc/common.cpp -o /tmp/gemv_bench && /tmp/gemv_bench
// Each variant streams NBUF distinct weight copies round-robin (no cache reuse) and is timed with HIP
// events over many launches; a pure streaming-read kernel gives the achievable-HBM ceiling of the box.
#include <stdio.h>
#include <stdlib.h>
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

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Shape { const char* name; int N, K, epi; bool norm; };

template <int WAVES, int EPI, bool NORM, int UNR, int TILES, bool NT, bool PX = true>
float run_variant(const Shape& sh, std::vector<bf16_t*>& wbufs, bf16_t* x, bf16_t* nw, bf16_t* res, bf16_t* out, int M, int iters) {
  LinearArgs a{};
  a.x = x; a.ldx = sh.K; a.norm_w = NORM ? nw : nullptr; a.eps = 1e-6f; a.res = res; a.M = M; a.N = sh.N; a.K = sh.K; a.epi = EPI;
  const int n_out = EPI == EPI_SILU ? sh.N / 2 : sh.N;
  a.ldr = n_out; a.out = out; a.ldo = n_out;
  if ((sh.N / 16) % TILES) return -1.f;
  dim3 grid(sh.N / (16 * TILES)), block(WAVES * 64);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) { a.wp = wbufs[w % wbufs.size()]; hipLaunchKernelGGL((linear_skinny_kernel<WAVES, EPI, NORM, UNR, TILES, PX, NT>), grid, block, 0, 0, a); }
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) { a.wp = wbufs[i % wbufs.size()]; hipLaunchKernelGGL((linear_skinny_kernel<WAVES, EPI, NORM, UNR, TILES, PX, NT>), grid, block, 0, 0, a); }
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / iters;
}

#define V(W, U, T, NTF) do { float us = (sh.epi == EPI_SILU) ? run_variant<W, EPI_SILU, true, U, (T < 2 ? 2 : T), NTF>(sh, wbufs, x, nw, res, out, M, iters) \
    : (sh.epi == EPI_RESIDUAL ? run_variant<W, EPI_RESIDUAL, false, U, T, NTF>(sh, wbufs, x, nw, res, out, M, iters) \
    : run_variant<W, EPI_STORE, true, U, T, NTF>(sh, wbufs, x, nw, res, out, M, iters)); \
    if (us > 0) { printf("  W=%2d PAIRS=%d TILES=%d nt=%d : %7.2f us  %6.0f GB/s\n", W, U, (sh.epi == EPI_SILU && T < 2) ? 2 : T, (int)NTF, us, bytes / us * 1e-3); fflush(stdout); } } while (0)

static float run_shipped(const Shape& sh, std::vector<bf16_t*>& wbufs, bf16_t* x, bf16_t* nw, bf16_t* res, bf16_t* out, int M, int iters) {
  LinearArgs a{};
  a.x = x; a.ldx = sh.K; a.norm_w = sh.norm ? nw : nullptr; a.eps = 1e-6f; a.res = res; a.M = M; a.N = sh.N; a.K = sh.K; a.epi = sh.epi;
  const int n_out = sh.epi == EPI_SILU ? sh.N / 2 : sh.N;
  a.ldr = n_out; a.out = out; a.ldo = n_out;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) { a.wp = wbufs[w % wbufs.size()]; launch_linear_skinny(a, 0); }
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) { a.wp = wbufs[i % wbufs.size()]; launch_linear_skinny(a, 0); }
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / iters;
}

#define VX(W, U, T) do { float a_ = (sh.epi == EPI_SILU) ? run_variant<W, EPI_SILU, true, U, (T < 2 ? 2 : T), true, false>(sh, wbufs, x, nw, res, out, M, iters) \
    : (sh.epi == EPI_RESIDUAL ? run_variant<W, EPI_RESIDUAL, false, U, T, true, false>(sh, wbufs, x, nw, res, out, M, iters) \
    : run_variant<W, EPI_STORE, true, U, T, true, false>(sh, wbufs, x, nw, res, out, M, iters)); \
    if (a_ > 0) { printf("  W=%2d PAIRS=%d TILES=%d per-tile activation loads : %7.2f us\n", W, U, T, a_); fflush(stdout); } } while (0)

int main() {
  const int M = 8, iters = 200;
  const size_t stream_bytes = (size_t)1 << 30;
  void* big; CK(hipMalloc(&big, stream_bytes)); CK(hipMemset(big, 1, stream_bytes));
  uint32_t* sink; CK(hipMalloc((void**)&sink, 4));
  const bool quick = getenv("GEMV_QUICK") != nullptr;   // only the GEMV variants of w13 / wo
  for (int blocks : {1024, 2048, 4096, 8192}) {
    if (quick) break;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(stream_read_kernel, dim3(blocks), dim3(256), 0, 0, (const u32x4*)big, stream_bytes / 16, sink);
    CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(stream_read_kernel, dim3(blocks), dim3(256), 0, 0, (const u32x4*)big, stream_bytes / 16, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("stream read 1 GiB, %d blocks: %.0f GB/s\n", blocks, stream_bytes * 10 / (ms * 1e-3) * 1e-9);
  }
  // small streaming reads (the size of one GEMV) to see the fixed per-kernel cost
  for (size_t mb : {21, 32, 50, 100}) {
    if (quick) break;
    size_t bytes = mb << 20; hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(stream_read_kernel, dim3(2048), dim3(256), 0, 0, (const u32x4*)((char*)big + (size_t)(i % 8) * (128 << 20)), bytes / 16, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("stream read %zu MiB per launch: %.2f us/launch, %.0f GB/s\n", mb, ms * 10, bytes * 100 / (ms * 1e-3) * 1e-9);
  }
  Shape shapes[] = {{"w13  N=19456 K=2560 swiglu+norm", 19456, 2560, EPI_SILU, true}, {"wqkv N=6144 K=2560 store+norm", 6144, 2560, EPI_STORE, true},
                    {"wo   N=2560 K=4096 residual", 2560, 4096, EPI_RESIDUAL, false}, {"w2   N=2560 K=9728 residual", 2560, 9728, EPI_RESIDUAL, false},
                    {"head N=4096 K=2560 store+norm (fast_output / live LM head)", 4096, 2560, EPI_STORE, true}};
  for (const Shape& sh : shapes) {
    if (quick && sh.N != 19456 && !(sh.N == 2560 && sh.K == 4096)) continue;
    const double bytes = (double)sh.N * sh.K * 2;
    const int nbuf = (int)(2.0e9 / bytes) + 1;
    std::vector<bf16_t*> wbufs(nbuf);
    for (auto& p : wbufs) { CK(hipMalloc((void**)&p, (size_t)bytes)); CK(hipMemset(p, 0x11, (size_t)bytes)); }
    bf16_t *x, *nw, *res, *out;
    CK(hipMalloc((void**)&x, (size_t)16 * sh.K * 2)); CK(hipMemset(x, 0x3c, (size_t)16 * sh.K * 2));
    CK(hipMalloc((void**)&nw, (size_t)sh.K * 2)); CK(hipMemset(nw, 0x3c, (size_t)sh.K * 2));
    CK(hipMalloc((void**)&res, (size_t)16 * sh.N * 2)); CK(hipMemset(res, 0, (size_t)16 * sh.N * 2));
    CK(hipMalloc((void**)&out, (size_t)16 * sh.N * 2));
    printf("%s  (%.1f MB, M=%d)\n", sh.name, bytes / 1e6, M);
    // paired activation loads (the M <= 8 path) over pairs-in-flight x tiles, then the per-tile-load path (M > 8) of the shipped shapes
    V(8, 1, 2, true); V(8, 2, 2, true); V(8, 1, 1, true); V(8, 2, 1, true); V(8, 4, 1, true); V(16, 1, 2, true); V(16, 2, 1, true);
    // one chunk = the wave's whole k-slice in flight (K = 2560: 5 pairs per wave of 8, 10 per wave of 4; K = 4096: 8):
    // a single exposed round trip per work-group instead of one per chunk (round-3 candidates)
    V(8, 5, 2, true); V(8, 5, 1, true); V(4, 10, 1, true); V(4, 5, 2, true); V(8, 8, 1, true); V(8, 3, 2, true);
    VX(8, 1, 2); VX(8, 2, 1);
    { float us = run_shipped(sh, wbufs, x, nw, res, out, M, iters); printf("  shipped launcher (launch_linear_skinny) : %7.2f us  %6.0f GB/s\n", us, bytes / us * 1e-3); }
    for (auto p : wbufs) hipFree(p);
    hipFree(x); hipFree(nw); hipFree(res); hipFree(out);
  }
  return 0;
}
