// one-off: what do row_ror:8, v_permlane16_swap and v_permlane32_swap do per lane?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int v2u __attribute__((ext_vector_type(2)));
__global__ void k(unsigned *o) {
  const unsigned lane = threadIdx.x;
  unsigned a = lane, b = 100 + lane;
  o[lane] = __builtin_amdgcn_update_dpp(0, (int)a, 0x128, 0xf, 0xf, false);
  v2u s = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  o[64 + lane] = s.x; o[128 + lane] = s.y;
  v2u t = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  o[192 + lane] = t.x; o[256 + lane] = t.y;
}
int main() {
  unsigned *d, h[320];
  hipMalloc(&d, sizeof h);
  k<<<1, 64>>>(d);
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  const char *names[5] = {"row_ror:8(a)", "p16swap.x", "p16swap.y", "p32swap.x", "p32swap.y"};
  for (int r = 0; r < 5; ++r) { printf("%s:", names[r]); for (int l = 0; l < 64; ++l) printf(" %u", h[r * 64 + l]); printf("\n"); }
  return 0;
}
