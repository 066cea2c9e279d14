// gemv_q8_bench.hip -- the int8 weight stream of the decode GEMV against the bf16 stream, per S2 shape (M = 8).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemv_q8_bench.hip fish_speech_amd/csrc/common.cpp -o /tmp/q8 && /tmp/q8
//   (-DFMI_Q8_PLAIN_CVT: the compiler's own int8 -> float sequence instead of the SDWA convert)
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../fish_speech_amd/csrc/dualar_kernels.hip"

using namespace fmi;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Shape { const char* name; int N, K, epi; bool norm; };

template <int WAVES, int EPI, bool NORM, int UNR, int TILES, bool Q8>
float run_variant(const Shape& sh, std::vector<void*>& wbufs, bf16_t* x, bf16_t* nw, bf16_t* res, bf16_t* out, bf16_t* scale, int M, int iters) {
  LinearArgs a{};
  a.x = x; a.ldx = sh.K; a.norm_w = NORM ? nw : nullptr; a.eps = 1e-6f; a.res = res; a.M = M; a.N = sh.N; a.K = sh.K; a.epi = EPI;
  const int n_out = EPI == EPI_SILU ? sh.N / 2 : sh.N;
  a.ldr = n_out; a.out = out; a.ldo = n_out;
  a.scale = Q8 ? scale : nullptr;
  if ((sh.N / 16) % TILES) return -1.f;
  dim3 grid(sh.N / (16 * TILES)), block(WAVES * 64);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto go = [&](int i) {
    if (Q8) a.wq = (const int8_t*)wbufs[i % wbufs.size()]; else a.wp = (const bf16_t*)wbufs[i % wbufs.size()];
    hipLaunchKernelGGL((linear_skinny_kernel<WAVES, EPI, NORM, UNR, TILES, 8, true, Q8>), grid, block, 0, 0, a);
  };
  for (int w = 0; w < 3; ++w) go(w);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) go(i);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / iters;
}

#define V(Q, W, U, T) do { float us = (sh.epi == EPI_SILU) ? run_variant<W, EPI_SILU, true, U, (T < 2 ? 2 : T), Q>(sh, wbufs, x, nw, res, out, scale, M, iters) \
    : (sh.epi == EPI_RESIDUAL ? run_variant<W, EPI_RESIDUAL, false, U, T, Q>(sh, wbufs, x, nw, res, out, scale, M, iters) \
    : run_variant<W, EPI_STORE, true, U, T, Q>(sh, wbufs, x, nw, res, out, scale, M, iters)); \
    if (us > 0) { printf("  %s W=%2d PAIRS=%d TILES=%d : %7.2f us\n", Q ? "int8" : "bf16", W, U, (sh.epi == EPI_SILU && T < 2) ? 2 : T, us); fflush(stdout); } } while (0)

int main() {
  const int M = 8, iters = 200;
  Shape shapes[] = {{"w13  N=19456 K=2560 swiglu+norm", 19456, 2560, EPI_SILU, true}, {"wqkv N=6144 K=2560 store+norm", 6144, 2560, EPI_STORE, true},
                    {"wo   N=2560 K=4096 residual", 2560, 4096, EPI_RESIDUAL, false}, {"w2   N=2560 K=9728 residual", 2560, 9728, EPI_RESIDUAL, false},
                    {"head N=4096 K=2560 store+norm", 4096, 2560, EPI_STORE, true}};
  for (const Shape& sh : shapes) {
    const double bytes = (double)sh.N * sh.K * 2;
    const int nbuf = (int)(2.0e9 / bytes) + 1;
    std::vector<void*> wbufs(nbuf);
    for (auto& p : wbufs) { CK(hipMalloc(&p, (size_t)bytes)); CK(hipMemset(p, 0x11, (size_t)bytes)); }
    bf16_t *x, *nw, *res, *out, *scale;
    CK(hipMalloc((void**)&x, (size_t)16 * sh.K * 2)); CK(hipMemset(x, 0x3c, (size_t)16 * sh.K * 2));
    CK(hipMalloc((void**)&nw, (size_t)sh.K * 2)); CK(hipMemset(nw, 0x3c, (size_t)sh.K * 2));
    CK(hipMalloc((void**)&res, (size_t)16 * sh.N * 2)); CK(hipMemset(res, 0, (size_t)16 * sh.N * 2));
    CK(hipMalloc((void**)&out, (size_t)16 * sh.N * 2));
    CK(hipMalloc((void**)&scale, (size_t)sh.N * 2)); CK(hipMemset(scale, 0x3c, (size_t)sh.N * 2));
    printf("%s  (%.1f MB bf16, M=%d)\n", sh.name, bytes / 1e6, M);
    V(false, 8, 1, 2); V(false, 8, 2, 1);
    V(true, 8, 1, 2); V(true, 8, 2, 2); V(true, 8, 4, 2); V(true, 8, 2, 4); V(true, 8, 4, 4);
    V(true, 8, 2, 1); V(true, 8, 4, 1); V(true, 8, 8, 1); V(true, 4, 4, 2); V(true, 16, 2, 2); V(true, 16, 4, 1);
    for (auto p : wbufs) hipFree(p);
    hipFree(x); hipFree(nw); hipFree(res); hipFree(out); hipFree(scale);
  }
  return 0;
}
