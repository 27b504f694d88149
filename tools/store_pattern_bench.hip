// Development micro-benchmark: what HBM write rate do the store patterns of the stored-states rollout reach on their own (no
// compute)?  Writes R = M*S*N trajectories of ROWB bytes each ([M][S][N][(H+1)*ds] layout, as rollout_body stores them).
//   hipcc --offload-arch=gfx950 -O3 tools/store_pattern_bench.hip -o /tmp/store_pattern_bench && /tmp/store_pattern_bench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

// P0: flat fill
__global__ void fill_flat(v4f *out, size_t n16) {
  const v4f v = {1.f, 2.f, 3.f, 4.f};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) out[i] = v;
}

// P1: the rollout's pattern.  Block = particle n, wave w = dynamics group, lane = sample s; per chunk c (CH bytes) the wave
// writes CH/16 lanes per trajectory.  Rows r = (m*S + s)*N + n.
template <int CH>
__global__ void __launch_bounds__(256) fill_rollout(char *out, int M, int S, int N, int rowb, int G, int pair) {
  const int n = blockIdx.x, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int PC = CH / 16;
  const int piece = lane & (PC - 1);
  const v4f v = {1.f, 2.f, 3.f, 4.f};
  const int nch = (rowb + CH - 1) / CH + 1;  // chunks sit on CH-byte lines of the output (phase ph), first / last partial
  for (int m = w; m < M; m += G * pair) {
    for (int c = 0; c < nch; ++c) {
      for (int q = 0; q < pair; ++q) {
#pragma unroll
        for (int i = 0; i < PC; ++i) {
          const int j = lane / PC + (64 / PC) * i;  // trajectory (sample s = j)
          const size_t r = ((size_t)(m + q * G) * S + j) * N + n;
          const int ph = (int)((r * rowb) % CH), o = c * CH - ph + piece * 16;
          if (o >= 0 && o + 16 <= rowb) *reinterpret_cast<v4f *>(out + r * rowb + o) = v;
        }
      }
      __builtin_amdgcn_s_sleep(8);
    }
  }
}

// P2: wave = 64 ADJACENT particles n of one (m, s): rows contiguous (64 * rowb bytes per wave and rollout)
template <int CH>
__global__ void __launch_bounds__(256) fill_adjacent(char *out, int M, int S, int N, int rowb, int G) {
  const int nb = blockIdx.x % (N / 64), s = blockIdx.x / (N / 64), w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int PC = CH / 16;
  const int piece = lane & (PC - 1);
  const v4f v = {1.f, 2.f, 3.f, 4.f};
  const int nch = (rowb + CH - 1) / CH + 1;
  for (int m = w; m < M; m += G) {
    for (int c = 0; c < nch; ++c) {
#pragma unroll
      for (int i = 0; i < PC; ++i) {
        const int j = lane / PC + (64 / PC) * i;
        const size_t r = ((size_t)m * S + s) * N + nb * 64 + j;
        const int ph = (int)((r * rowb) % CH), o = c * CH - ph + piece * 16;
        if (o >= 0 && o + 16 <= rowb) *reinterpret_cast<v4f *>(out + r * rowb + o) = v;
      }
      __builtin_amdgcn_s_sleep(8);
    }
  }
}

// P3: adjacent rows, whole-row streaming: the wave's 64 rows are one contiguous 64*rowb region: write it linearly in time order
// of a rollout that stages T steps: per flush, each row's chunk (CH bytes) -> as P2; (control: fully linear per wave)
__global__ void __launch_bounds__(256) fill_wave_linear(char *out, int M, int S, int N, int rowb, int G) {
  const int nb = blockIdx.x % (N / 64), s = blockIdx.x / (N / 64), w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const v4f v = {1.f, 2.f, 3.f, 4.f};
  for (int m = w; m < M; m += G) {
    const size_t r0 = ((size_t)m * S + s) * N + nb * 64;
    char *base = out + r0 * rowb;
    const int total = 64 * rowb;
    for (int o = lane * 16; o < total; o += 64 * 16) *reinterpret_cast<v4f *>(base + o) = v;
  }
}

// P5: wave = 8 samples s x 8 ADJACENT particles n of one m: each (m, s) group of 8 rows is 8 * 656 = 41 whole lines.  Whole
// lines only: per chunk step every trajectory writes one line lying fully inside its row; the 7 lines per group that straddle two
// rows are written at the end (the rollout would keep each row's head in LDS until its neighbour's tail exists).
__global__ void __launch_bounds__(256) fill_group8(char *out, int M, int S, int N, int rowb, int G) {
  const int nb = blockIdx.x % (N / 8), sb = blockIdx.x / (N / 8), w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int piece = lane & 7;
  const v4f v = {1.f, 2.f, 3.f, 4.f};
  for (int m = w; m < M; m += G) {
    for (int c = 0; c < 5; ++c) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int t = lane / 8 + 8 * i;          // trajectory of the wave: s_sub = t / 8, n_sub = t % 8
        const int s = sb * 8 + t / 8, j = t & 7;  // row j of the group
        const size_t g0 = (((size_t)m * S + s) * N + (size_t)nb * 8) * rowb;  // group base (line-aligned)
        const int first = (rowb * j + 127) / 128, last = (rowb * (j + 1)) / 128 - 1;  // full lines of row j
        const int ln = first + c;
        if (ln <= last) *reinterpret_cast<v4f *>(out + g0 + (size_t)ln * 128 + piece * 16) = v;
      }
      __builtin_amdgcn_s_sleep(8);
    }
    // straddling lines: 7 per group, 8 groups per wave
    for (int q = lane / 8; q < 56; q += 8) {
      const int grp = q / 7, k = q % 7 + 1;  // boundary between rows k-1 and k
      const int s = sb * 8 + grp;
      const size_t g0 = (((size_t)m * S + s) * N + (size_t)nb * 8) * rowb;
      const int ln = (rowb * k) / 128;
      *reinterpret_cast<v4f *>(out + g0 + (size_t)ln * 128 + piece * 16) = v;
    }
  }
}

int main() {
  const int M = 64, S = 64, N = 4096, H = 40, ds = 4, rowb = (H + 1) * ds * 4;
  const size_t R = (size_t)M * S * N, bytes = R * rowb;
  char *out;
  CK(hipMalloc(&out, bytes));
  CK(hipMemset(out, 0, bytes));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto timeit = [&](const char *name, auto launch, double frac) {
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 3;
    printf("%-34s %8.1f us  %6.0f GB/s\n", name, ms * 1e3, bytes * frac / ms / 1e6);
    fflush(stdout);
  };
  timeit("flat fill", [&] { fill_flat<<<256 * 16, 256>>>(reinterpret_cast<v4f *>(out), bytes / 16); }, 1.0);
  const double f128 = 1.0, f64 = 1.0, f256 = 1.0;  // (whole 16-byte pieces: 656 = 41 * 16 - everything is written)
  timeit("rollout pattern, 128-B chunks", [&] { fill_rollout<128><<<N, 256>>>(out, M, S, N, rowb, 4, 1); }, f128);
  timeit("rollout pattern, 64-B chunks", [&] { fill_rollout<64><<<N, 256>>>(out, M, S, N, rowb, 4, 1); }, f64);
  timeit("rollout pattern, 64-B, pairs", [&] { fill_rollout<64><<<N, 256>>>(out, M, S, N, rowb, 4, 2); }, f64);
  timeit("rollout pattern, 256-B chunks", [&] { fill_rollout<256><<<N, 256>>>(out, M, S, N, rowb, 4, 1); }, f256);
  timeit("adjacent rows, 128-B chunks", [&] { fill_adjacent<128><<<S * (N / 64), 256>>>(out, M, S, N, rowb, 4); }, f128);
  timeit("adjacent rows, 64-B chunks", [&] { fill_adjacent<64><<<S * (N / 64), 256>>>(out, M, S, N, rowb, 4); }, f64);
  timeit("adjacent rows, wave-linear", [&] { fill_wave_linear<<<S * (N / 64), 256>>>(out, M, S, N, rowb, 4); }, 1.0);
  timeit("8 adjacent rows, whole lines only", [&] { fill_group8<<<(S / 8) * (N / 8), 256>>>(out, M, S, N, rowb, 4); }, 1.0);
  {  // control: the rollout pattern with 640-byte rows (5 whole lines, no phase, no partial lines)
    const size_t b640 = R * 640;
    auto t640 = [&](const char *name, auto launch) {
      launch();
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for (int i = 0; i < 3; ++i) launch();
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      ms /= 3;
      printf("%-34s %8.1f us  %6.0f GB/s\n", name, ms * 1e3, b640 / ms / 1e6);
    };
    t640("rollout pattern, 640-B rows, 128", [&] { fill_rollout<128><<<N, 256>>>(out, M, S, N, 640, 4, 1); });
    t640("rollout pattern, 640-B rows, 64", [&] { fill_rollout<64><<<N, 256>>>(out, M, S, N, 640, 4, 1); });
  }
  return 0;
}
