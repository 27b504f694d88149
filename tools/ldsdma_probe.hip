// LDS-DMA addressing probe (gfx950): where do the bytes of `buffer_load_dwordx4 ... offen offset:X lds` land, and does M0 reach LDS
// addresses beyond 64 KB?   hipcc --offload-arch=gfx950 -O2 tools/ldsdma_probe.hip -o tools/_ldsdma_probe && tools/_ldsdma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void *lds_ptr;
__global__ void probe(const float *g, float *out, int base_floats, int imm16) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x;
  for (int i = lane; i < 40960; i += 64) lds[i] = -1.f;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(g), 0, 1 << 20, 0x00020000);
  if (imm16) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)(lds + base_floats), 16, lane * 32, 0, 16, 0);
  else __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)(lds + base_floats), 16, lane * 32, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = lane; i < 40960; i += 64) out[i] = lds[i];
}
int main() {
  std::vector<float> h(1 << 18);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)i;
  float *g, *o;
  hipMalloc(&g, h.size() * 4);
  hipMalloc(&o, 40960 * 4);
  hipMemcpy(g, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int base : {0, 8192, 20000, 30000})
    for (int imm : {0, 1}) {
      probe<<<1, 64, 160 * 1024>>>(g, o, base, imm);
      std::vector<float> r(40960);
      hipMemcpy(r.data(), o, 40960 * 4, hipMemcpyDeviceToHost);
      int first = -1, cnt = 0;
      for (int i = 0; i < 40960; ++i)
        if (r[i] != -1.f) { if (first < 0) first = i; ++cnt; }
      printf("base %6d floats imm16 %d: first written float %6d (count %d) values lane0: %g %g %g %g lane1: %g\n", base, imm, first, cnt,
             first >= 0 ? r[first] : -1.f, first >= 0 ? r[first + 1] : -1.f, first >= 0 ? r[first + 2] : -1.f, first >= 0 ? r[first + 3] : -1.f, first >= 0 ? r[first + 4] : -1.f);
    }
  return 0;
}
