// anyorder_probe.hip - does hipExtAnyOrderLaunch let two independent kernels of ONE stream run side by side on gfx950?
// (hip_ext.h says the flag is "not supported on AMD GFX9xx boards"; measured rather than believed.)
//   hipcc --offload-arch=gfx950 -O3 tools/anyorder_probe.hip -o tools/_anyorder_probe && tools/_anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin(unsigned long long ticks, unsigned int *out) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) out[blockIdx.x] = 1u;
}
int main() {
  unsigned int *d;
  hipMalloc(&d, 4096);
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode) {
    float best = 1e9f;
    for (int rep = 0; rep < 20; ++rep) {
      hipEventRecord(e0, s);
      spin<<<30, 1024, 0, s>>>(2000ull, d);  // 20 us on 30 CUs
      if (mode == 0) spin<<<256, 256, 0, s>>>(2000ull, d + 64);
      else hipExtLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, 2000ull, d + 64);
      spin<<<1, 64, 0, s>>>(100ull, d + 512);  // an ordinary launch behind them: must wait for both
      hipEventRecord(e1, s);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("%s: two 20 us kernels + 1 us kernel = %.1f us\n", mode ? "second launch with hipExtAnyOrderLaunch" : "ordinary launches", best * 1e3f);
  }
  return 0;
}
