// mailbox_probe.hip - what a device-resident control tick costs beyond its arithmetic (round 5).
// A persistent grid of G x 1024 lanes: workgroup 0 polls a pinned-host mailbox for (seq, state), relays it to the other workgroups
// through T replicated device lines, every workgroup invalidates its caches (agent-scope acquire: what a tick needs before it reads
// exchange buffers that an earlier tick of the same launch used), writes 16 bytes of output to pinned host memory, and the LAST
// workgroup to arrive on a device counter publishes the sequence number to the host.  The host measures the round trip.
//   hipcc --offload-arch=gfx950 -O3 tools/mailbox_probe.hip -o tools/_mailbox_probe && tools/_mailbox_probe [G] [iters] [inv]
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Mbox {  // pinned host memory
  volatile unsigned int seq_in;  float state[4];  unsigned int stop;  unsigned int pad0[26];
  volatile unsigned int seq_out; unsigned int status; unsigned int pad1[30];
  float out[4096];
};
enum { NREP = 8, LINE = 32 };

__global__ __launch_bounds__(1024) void serve(Mbox *mb, unsigned int *relay /* [NREP][LINE]: seq, stop, x0..x3 */, unsigned int *done, int do_inv, unsigned long long idle_ticks) {
  const int b = blockIdx.x, G = gridDim.x, tid = threadIdx.x;
  __shared__ unsigned int sh[8];
  for (unsigned int tick = 1;; ++tick) {
    if (tid == 0) {
      unsigned int stop = 0;
      float x[4] = {0, 0, 0, 0};
      if (b == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        unsigned int spins = 0;
        while (__hip_atomic_load(&mb->seq_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != tick) {
          if ((++spins & 63u) == 0u && __builtin_amdgcn_s_memrealtime() - t0 > idle_ticks) { stop = 2; break; }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        if (!stop) {
          stop = __hip_atomic_load(&mb->stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          for (int k = 0; k < 4; ++k) x[k] = __hip_atomic_load(&mb->state[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        for (int r = 0; r < NREP; ++r) {
          unsigned int *l = relay + r * LINE;
          __hip_atomic_store(l + 1, stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          for (int k = 0; k < 4; ++k) __hip_atomic_store(l + 2 + k, __float_as_uint(x[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        for (int r = 0; r < NREP; ++r) __hip_atomic_store(relay + r * LINE, tick, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        unsigned int *l = relay + (b % NREP) * LINE;
        while (__hip_atomic_load(l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != tick) __builtin_amdgcn_s_sleep(1);
        stop = __hip_atomic_load(l + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int k = 0; k < 4; ++k) x[k] = __uint_as_float(__hip_atomic_load(l + 2 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      }
      sh[0] = stop;
      sh[1] = __float_as_uint(x[0] + x[1]);
    }
    __syncthreads();
    if (sh[0]) return;
    if (do_inv) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    // "outputs": 4 floats per workgroup straight to pinned host memory
    if (tid < 4) mb->out[b * 4 + tid] = __uint_as_float(sh[1]) + (float)tick;
    __syncthreads();
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // system scope
      const unsigned int old = __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old == tick * (unsigned int)G - 1u) __hip_atomic_store(&mb->seq_out, tick, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

int main(int argc, char **argv) {
  const int G = argc > 1 ? atoi(argv[1]) : 256, iters = argc > 2 ? atoi(argv[2]) : 20000, inv = argc > 3 ? atoi(argv[3]) : 1;
  Mbox *mb;
  CHECK(hipHostMalloc((void **)&mb, sizeof(Mbox), hipHostMallocCoherent | hipHostMallocMapped));
  memset((void *)mb, 0, sizeof(Mbox));
  unsigned int *relay, *done;
  CHECK(hipMalloc((void **)&relay, NREP * LINE * 4));
  CHECK(hipMalloc((void **)&done, 128));
  CHECK(hipMemset(relay, 0, NREP * LINE * 4));
  CHECK(hipMemset(done, 0, 128));
  hipStream_t s;
  CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  serve<<<G, 1024, 0, s>>>(mb, relay, done, inv, 100000000ull /* 1 s */);
  CHECK(hipGetLastError());
  std::vector<double> us(iters);
  for (int i = 1; i <= iters; ++i) {
    const auto t0 = std::chrono::steady_clock::now();
    mb->state[0] = (float)i;
    std::atomic_thread_fence(std::memory_order_release);
    mb->seq_in = (unsigned int)i;
    while (mb->seq_out != (unsigned int)i) __builtin_ia32_pause();
    std::atomic_thread_fence(std::memory_order_acquire);
    us[i - 1] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    if (mb->out[(G - 1) * 4 + 3] != (float)i + (float)i) { printf("bad output at %d: %f\n", i, mb->out[(G - 1) * 4 + 3]); return 2; }
  }
  mb->stop = 1;
  std::atomic_thread_fence(std::memory_order_release);
  mb->seq_in = (unsigned int)(iters + 1);
  CHECK(hipStreamSynchronize(s));
  std::sort(us.begin() + iters / 10, us.end());  // (drop the first tenth: clock ramp)
  std::vector<double> v(us.begin() + iters / 10, us.end());
  std::sort(v.begin(), v.end());
  double sum = 0;
  for (double x : v) sum += x;
  printf("G=%d inv=%d iters=%d: round trip mean %.2f us, median %.2f, p90 %.2f, p99 %.2f, min %.2f\n", G, inv, iters, sum / v.size(), v[v.size() / 2],
         v[(size_t)(v.size() * 0.9)], v[(size_t)(v.size() * 0.99)], v[0]);
  return 0;
}
