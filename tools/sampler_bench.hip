// sampler_bench.hip -- cumulative stage timing of sample_small_kernel (profiling aid).
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../fish_speech_amd/csrc/dualar_sample.hip"
using namespace fmi;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
int main() {
  const int B = 8, n = 4097, ld = 4128;
  std::vector<bf16_t> h((size_t)B * ld);
  srand(1);
  for (auto& v : h) v = f2bf(((rand() % 2000) - 1000) / 250.0f);
  bf16_t* lg; CK(hipMalloc((void**)&lg, h.size() * 2)); CK(hipMemcpy(lg, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  int32_t* out; CK(hipMalloc((void**)&out, B * 4));
  int32_t* ids; CK(hipMalloc((void**)&ids, n * 4)); std::vector<int32_t> hi(n); for (int i = 0; i < n; ++i) hi[i] = 100 + i; CK(hipMemcpy(ids, hi.data(), n * 4, hipMemcpyHostToDevice));
  SampleArgs a{};
  a.logits = lg; a.B = B; a.n = n; a.ld = ld; a.ids = ids; a.mode = 2; a.temperature = rbf(0.7f); a.top_p = rbf(0.7f); a.top_k = 30; a.seed = 7;
  a.frame = 3; a.draw = 0; a.out_tok = out; a.small_k = 1;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int stage : {1, 2, 3, 4, 5, 6, 7, 0}) {
    a.dbg_stop = stage;
    for (int i = 0; i < 5; ++i) launch_sample(a, 0);
    CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
    for (int i = 0; i < 200; ++i) launch_sample(a, 0);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("stop after stage %d: %.2f us per launch\n", stage, ms * 1e3 / 200);
  }
  a.small_k = 0; a.dbg_stop = 0;
  CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
  for (int i = 0; i < 200; ++i) launch_sample(a, 0);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("generic sample_kernel: %.2f us per launch\n", ms * 1e3 / 200);
  return 0;
}
