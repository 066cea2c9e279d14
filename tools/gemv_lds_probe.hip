// gemv_lds_probe.hip -- can a weight-streaming GEMV get more bytes in flight by staging through LDS?
// The shipped kernel (linear_skinny_kernel) holds in-flight weight tiles in VGPRs (UNR x TILES KiB per wave).
// Here every wave streams its K range through a private ring of D KiB-slots filled by LDS-DMA
// (global_load_lds_dwordx4, no VGPRs), consumed by ds_read_b128 + MFMA with manual vmcnt accounting.
// Arithmetic is the real one (v_mfma_f32_16x16x32_bf16 against a constant activation fragment); norm,
// activations loads and epilogue are left out: this measures the streaming part only.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gemv_lds_probe tools/gemv_lds_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int MROWS = 8;  // live batch rows: lanes of the other fragment rows issue no request
template <int N> __device__ inline void wait_vmcnt() {
  // gfx9 s_waitcnt: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt_hi[15:14]
  __builtin_amdgcn_s_waitcnt((N & 0xF) | ((N >> 4) << 14) | (0x7 << 4) | (0xF << 8));
}

// XMODE: how the activation fragment of a k-tile (16 rows x 32 k, 1 KiB) reaches the MFMA
//   0 constant register (no traffic: the streaming ceiling)   1 LDS-DMA from row-major x (16 x 64-byte pieces)
//   2 LDS-DMA from x pre-packed in fragment order (1 KiB linear)   3 plain global loads, D-deep register ring
template <int WAVES, int TILES, int D, int XMODE>
__global__ __launch_bounds__(WAVES * 64) void gemv_lds(const u32x4* __restrict__ wp, int KT, float* out,
                                                        const unsigned short* __restrict__ x, int ldx) {
  extern __shared__ __attribute__((aligned(1024))) char ring[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int tile0 = blockIdx.x * TILES;
  const int kbeg = (int)((int64_t)wave * KT / WAVES), kend = (int)((int64_t)(wave + 1) * KT / WAVES);
  const int n = kend - kbeg;
  constexpr int OPS = TILES + ((XMODE == 1 || XMODE == 2) ? 1 : 0);
  constexpr int VOPS = TILES + (XMODE ? 1 : 0);   // vm ops per slot incl. register loads
  char* wbase = ring + wave * (D * OPS * 1024);
  const int b = lane & 15, g = lane >> 4;
  const unsigned short* xrow = x + (size_t)b * ldx + g * 8 + (size_t)kbeg * 32;         // row-major
  const u32x4* xpk = reinterpret_cast<const u32x4*>(x) + (size_t)kbeg * 64 + lane;      // packed
  u32x4 xr[D];
#pragma unroll
  for (int s = 0; s < D; ++s) xr[s] = (u32x4){0, 0, 0, 0};
  const u32x4* wrow[TILES];
#pragma unroll
  for (int t = 0; t < TILES; ++t) wrow[t] = wp + ((int64_t)(tile0 + t) * KT + kbeg) * 64 + lane;

  auto dma = [&](int i) {
    const int slot = i % D;
#pragma unroll
    for (int t = 0; t < TILES; ++t)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wrow[t] + (int64_t)i * 64),
                                       (__attribute__((address_space(3))) void*)(wbase + (slot * OPS + t) * 1024), 16, 0, 0);
    if (XMODE == 1 && b < MROWS)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xrow + (size_t)i * 32),
                                       (__attribute__((address_space(3))) void*)(wbase + (slot * OPS + TILES) * 1024), 16, 0, 0);
    if (XMODE == 2)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xpk + (size_t)i * 64),
                                       (__attribute__((address_space(3))) void*)(wbase + (slot * OPS + TILES) * 1024), 16, 0, 0);
  };
  f32x4 acc[TILES];
#pragma unroll
  for (int t = 0; t < TILES; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 xb;
#pragma unroll
  for (int j = 0; j < 8; ++j) xb[j] = (__bf16)(0.01f * (lane + j));
  const unsigned lds_lane = (unsigned)(size_t)(__attribute__((address_space(3))) char*)wbase + lane * 16;

  auto consume = [&](int i, const u32x4& xreg) {
    const int slot = i % D;
    u32x4 w[TILES], xv = xreg;
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
      const unsigned addr = lds_lane + (slot * OPS + t) * 1024;
      asm volatile("ds_read_b128 %0, %1" : "=v"(w[t]) : "v"(addr) : "memory");
    }
    if (XMODE == 1 || XMODE == 2) {
      const unsigned addr = lds_lane + (slot * OPS + TILES) * 1024;
      asm volatile("ds_read_b128 %0, %1" : "=v"(xv) : "v"(addr) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const bf16x8 xf = XMODE ? *reinterpret_cast<bf16x8*>(&xv) : xb;
#pragma unroll
    for (int t = 0; t < TILES; ++t)
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&w[t]), xf, acc[t], 0, 0, 0);
  };

  // n is a multiple of D for the shapes probed (K/32/WAVES in {10, 16, 38}: D = 2 only) -> generic tail below
  const int pro = n < D ? n : D;
#pragma unroll
  for (int s = 0; s < D; ++s)
    if (s < pro) {
      dma(s);
      if (XMODE == 3 && b < MROWS) xr[s] = *reinterpret_cast<const u32x4*>(xrow + (size_t)s * 32);
    }
  int i = 0;
  for (; i + 2 * D <= n; i += D) {   // steady state, unrolled by D so that the register ring is static
#pragma unroll
    for (int s = 0; s < D; ++s) {
      wait_vmcnt<(D - 1) * VOPS>();
      consume(i + s, xr[s]);
      dma(i + s + D);
      if (XMODE == 3 && b < MROWS) xr[s] = *reinterpret_cast<const u32x4*>(xrow + (size_t)(i + s + D) * 32);
    }
  }
  wait_vmcnt<0>();
  // drain: slots i .. min(i + D, n) - 1 are in the ring; anything beyond is fetched synchronously
#pragma unroll
  for (int s = 0; s < D; ++s)
    if (i + s < n) consume(i + s, xr[s]);
  for (int j = i + D; j < n; ++j) {
    dma(j);
    u32x4 xt = xr[0];
    if (XMODE == 3 && b < MROWS) xt = *reinterpret_cast<const u32x4*>(xrow + (size_t)j * 32);
    wait_vmcnt<0>();
    consume(j, xt);
  }

  float s = 0.f;
#pragma unroll
  for (int t = 0; t < TILES; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  if (s == 1.2345f) out[0] = s;
}

template <int WAVES, int TILES, int D, int XMODE>
void run(const char* name, int N, int K, std::vector<u32x4*>& bufs) {
  if ((N / 16) % TILES) return;
  const size_t smem = (size_t)WAVES * D * (TILES + ((XMODE == 1 || XMODE == 2) ? 1 : 0)) * 1024;
  if (smem > 160 * 1024) return;
  CK(hipFuncSetAttribute((const void*)gemv_lds<WAVES, TILES, D, XMODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  printf("  [W=%d TILES=%d D=%d x-mode %d]\n", WAVES, TILES, D, XMODE);
  float* out; CK(hipMalloc(&out, 4));
  unsigned short* xbuf; CK(hipMalloc((void**)&xbuf, (size_t)16 * K * 2)); CK(hipMemset(xbuf, 0x3c, (size_t)16 * K * 2));
  dim3 grid(N / (16 * TILES)), block(WAVES * 64);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 200;
  for (int w = 0; w < 3; ++w) gemv_lds<WAVES, TILES, D, XMODE><<<grid, block, smem>>>(bufs[w % bufs.size()], K / 32, out, xbuf, K);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int it = 0; it < iters; ++it) gemv_lds<WAVES, TILES, D, XMODE><<<grid, block, smem>>>(bufs[it % bufs.size()], K / 32, out, xbuf, K);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / iters, bytes = (double)N * K * 2;
  printf("  %-5s LDS ring: W=%2d TILES=%d D=%2d x-mode %d (%3zu KiB LDS) : %7.2f us  %6.0f GB/s\n", name, WAVES, TILES, D, XMODE, smem >> 10, us, bytes / us * 1e-3);
  CK(hipFree(out)); CK(hipFree(xbuf));
}

int main() {
  setvbuf(stdout, NULL, _IONBF, 0);
  struct { const char* name; int N, K; } shapes[] = {{"w13", 19456, 2560}, {"wqkv", 6144, 2560}, {"wo", 2560, 4096}, {"w2", 2560, 9728}};
  for (auto& sh : shapes) {
    const double bytes = (double)sh.N * sh.K * 2;
    const int nbuf = (int)(2.0e9 / bytes) + 1;
    std::vector<u32x4*> bufs(nbuf);
    for (auto& p : bufs) { CK(hipMalloc((void**)&p, (size_t)bytes)); CK(hipMemset(p, 0x11, (size_t)bytes)); }
    printf("%s N=%d K=%d (%.1f MB)\n", sh.name, sh.N, sh.K, bytes / 1e6);
    // (TILES = 2 with the lane-masked activation DMA faults on gfx950 / ROCm 7.2 -- not pursued)
    run<8, 1, 4, 0>(sh.name, sh.N, sh.K, bufs); run<8, 1, 4, 1>(sh.name, sh.N, sh.K, bufs); run<8, 1, 2, 1>(sh.name, sh.N, sh.K, bufs);
    run<8, 1, 4, 3>(sh.name, sh.N, sh.K, bufs); run<4, 1, 4, 1>(sh.name, sh.N, sh.K, bufs); run<16, 1, 2, 1>(sh.name, sh.N, sh.K, bufs);
    for (auto p : bufs) CK(hipFree(p));
  }
  return 0;
}
