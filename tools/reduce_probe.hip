// reduce_probe.hip - checks the lane-group reduction used by tick2.hpp (reduce_u) on the device: NV values per lane are summed
// over the 8 key sub-slices of a wave (lane bits 3..5) with one DPP row rotate and the gfx950 lane-swap instructions
// (v_permlane16_swap_b32 / v_permlane32_swap_b32); lane (u, c) ends with NV / 8 of the sums for its column group c.
//   hipcc --offload-arch=gfx950 -O3 -I dust_amd/csrc -I include tools/reduce_probe.hip -o tools/_reduce_probe && tools/_reduce_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "tick2_reduce.hpp"

template <int NV>
__global__ void probe(const float *in, float *out) {
  const int lane = threadIdx.x & 63;
  float v[NV], r[NV / 8];
  for (int i = 0; i < NV; ++i) v[i] = in[(size_t)lane * NV + i];
  dust::reduce_u<NV>(v, r, lane);
  for (int i = 0; i < NV / 8; ++i) {
    const int idx = dust::reduce_u_index<NV>(i, lane);
    out[(size_t)idx * 8 + (lane & 7)] = r[i];  // [NV][8 column groups]
  }
}

template <int NV>
__global__ void probe16(const float *in, float *out) {
  const int lane = threadIdx.x & 63;
  float v[NV], r[NV / 16];
  for (int i = 0; i < NV; ++i) v[i] = in[(size_t)lane * NV + i];
  dust::reduce_u16<NV>(v, r, lane);
  for (int i = 0; i < NV / 16; ++i) out[(size_t)dust::reduce_u16_index<NV>(i, lane) * 4 + (lane & 3)] = r[i];  // [NV][4 column groups]
}

template <int NV>
static int run16() {
  std::vector<float> h(64 * NV), o(NV * 4, -1.f);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 8.0f;
  float *di, *dout;
  hipMalloc(&di, h.size() * 4);
  hipMalloc(&dout, o.size() * 4);
  hipMemcpy(di, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemset(dout, 0xff, o.size() * 4);
  probe16<NV><<<1, 64>>>(di, dout);
  hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int idx = 0; idx < NV; ++idx)
    for (int c = 0; c < 4; ++c) {
      double ref = 0;
      for (int u = 0; u < 16; ++u) ref += h[(size_t)(u * 4 + c) * NV + idx];
      if (fabs(ref - o[idx * 4 + c]) > 1e-3 * (1 + fabs(ref))) {
        if (bad < 5) printf("NV=%d idx %d c %d: got %g want %g\n", NV, idx, c, o[idx * 4 + c], ref);
        ++bad;
      }
    }
  printf("reduce_u16<%d>: %s (%d mismatches)\n", NV, bad ? "FAIL" : "ok", bad);
  hipFree(di);
  hipFree(dout);
  return bad;
}

template <int NV>
static int run() {
  std::vector<float> h(64 * NV), o(NV * 8, -1.f);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 8.0f;
  float *di, *dout;
  hipMalloc(&di, h.size() * 4);
  hipMalloc(&dout, o.size() * 4);
  hipMemcpy(di, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemset(dout, 0xff, o.size() * 4);
  probe<NV><<<1, 64>>>(di, dout);
  hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int idx = 0; idx < NV; ++idx)
    for (int c = 0; c < 8; ++c) {
      double ref = 0;
      for (int u = 0; u < 8; ++u) ref += h[(size_t)(u * 8 + c) * NV + idx];
      if (fabs(ref - o[idx * 8 + c]) > 1e-3 * (1 + fabs(ref))) {
        if (bad < 5) printf("NV=%d idx %d c %d: got %g want %g\n", NV, idx, c, o[idx * 8 + c], ref);
        ++bad;
      }
    }
  printf("reduce_u<%d>: %s (%d mismatches)\n", NV, bad ? "FAIL" : "ok", bad);
  hipFree(di);
  hipFree(dout);
  return bad;
}

int main() { return (run<40>() + run<16>() + run<8>() + run16<48>() + run16<32>() + run16<16>()) ? 1 : 0; }
