// Where the 256-lane bandwidth role's time goes: hipcc --offload-arch=gfx950 -O3 -DK2_STAMPS -Iinclude -Idust_amd/csrc tools/k2_bw_probe.hip -o /tmp/k2_bw_probe
// Stamps (s_memrealtime, 100 MHz): 0 start, 1 loads in, 2 sorted, 3 warm start done, 4 narrowed, 5 end; [8] narrowing rounds.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
#include "dust_amd.h"
#include "bandwidth.hpp"
using namespace dust;
int main() {
  const int N = 1024, D = 30;
  std::mt19937 g(1);
  std::normal_distribution<float> nd(0.f, 2.2f);
  std::vector<float> th((size_t)N * D);
  for (auto &x : th) x = nd(g);
  float *dth, *dh;
  hipMalloc(&dth, th.size() * 4);
  hipMalloc(&dh, D * 4);
  hipMemset(dh, 0, D * 4);
  hipMemcpy(dth, th.data(), th.size() * 4, hipMemcpyHostToDevice);
  K2Args a{};
  a.N = N; a.D = D; a.H = D; a.da = 1; a.n_local = N; a.bw_scale = 1.f; a.min_bw = 1e-5f; a.theta = dth; a.h = dh; a.h_prev = dh;
  a.log_n1 = (float)std::log((double)N + 1.0);
  for (int it = 0; it < 4; ++it) {
    if (it >= 2) {  // nudge the particles as an SVGD step does (the warm start then brackets the answer)
      for (auto &x : th) x *= 1.003f;
      hipMemcpy(dth, th.data(), th.size() * 4, hipMemcpyHostToDevice);
    }
    k2_bandwidth256_kernel<<<D, 256, K2_BW256_LDS * sizeof(float)>>>(a);
    hipDeviceSynchronize();
    unsigned long long st[64 * 16];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(k2_stamps), sizeof st);
    printf("call %d:", it);
    for (int c = 0; c < 3; ++c) {
      const unsigned long long *s = st + c * 16;
      printf("  [dim %d] load %.2f sort %.2f warm %.2f narrow %.2f (%llu rounds) tail %.2f us", c, (s[1] - s[0]) * 0.01, (s[2] - s[1]) * 0.01,
             (s[3] - s[2]) * 0.01, (s[4] - s[3]) * 0.01, s[8], (s[5] - s[4]) * 0.01);
    }
    printf("\n");
  }
  return 0;
}
