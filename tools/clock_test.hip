// development aid: effective shader clock, empty-kernel duration, dependent-FMA latency, double-division cost
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void empty_k() {}
__global__ void fma_chain(float *out, int n, unsigned long long *stamps) {
  float x = threadIdx.x * 1e-9f, a = 1.000001f, b = 1e-7f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < n; ++i) x = fmaf(x, a, b);
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[threadIdx.x + blockIdx.x * blockDim.x] = x;
  if (threadIdx.x == 0 && blockIdx.x == 0) { stamps[0] = t1 - t0; stamps[1] = r1 - r0; }
}
__global__ void ddiv_chain(double *out, int n) {
  double x = 1.0 + threadIdx.x, y = 3.0;
  for (int i = 0; i < n; ++i) x = x / y + 1.0;
  out[threadIdx.x + blockIdx.x * blockDim.x] = x;
}
__global__ void lmod_chain(long *out, int n, int m) {
  long x = 123456789 + threadIdx.x; long acc = 0;
  for (int i = 0; i < n; ++i) { acc += x % m; x += acc; }
  out[threadIdx.x + blockIdx.x * blockDim.x] = acc;
}
template <class F> float timeit(F f, int reps) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); for (int i = 0; i < reps; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms * 1e3f / reps;
}
int main() {
  float *o; hipMalloc(&o, 1 << 22); unsigned long long *st; hipMalloc(&st, 16);
  printf("empty kernel back-to-back: %.2f us each\n", timeit([&] { empty_k<<<1, 64>>>(); }, 1000));
  printf("empty kernel 1024 blocks : %.2f us each\n", timeit([&] { empty_k<<<1024, 128>>>(); }, 1000));
  for (int rep = 0; rep < 3; ++rep) {
    float us = timeit([&] { fma_chain<<<1, 64>>>(o, 100000, st); }, 20);
    unsigned long long h[2]; hipMemcpy(h, st, 16, hipMemcpyDeviceToHost);
    printf("100k dependent fma, 1 wave: %.1f us -> %.2f ns/fma ; memtime ticks %llu, realtime ticks(100MHz) %llu -> clock %.0f MHz, %.2f cyc/fma\n", us, us * 1e3 / 100000,
           h[0], h[1], (double)h[0] / ((double)h[1] / 100.0), (double)h[0] / 100000);
  }
  printf("1000 dependent double divs, 1024x128 threads: %.1f us\n", timeit([&] { ddiv_chain<<<1024, 128>>>((double *)o, 1000); }, 20));
  printf("100 long %% int, 1024x128 threads: %.1f us\n", timeit([&] { lmod_chain<<<1024, 128>>>((long *)o, 100, 7); }, 20));
  printf("100k fma 1024x128 threads: %.1f us\n", timeit([&] { fma_chain<<<1024, 128>>>(o, 100000, st); }, 5));
  return 0;
}
