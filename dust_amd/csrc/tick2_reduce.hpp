// tick2_reduce.hpp - sum NV per-lane values over the 8 key sub-slices of a wave (lane = (u, c): u = lane >> 3 the key
// sub-slice, c = lane & 7 the column group) without LDS: a transposing butterfly.  Each step halves the number of values a
// lane carries - a lane keeps the half selected by one bit of u and hands the other half to its partner:
//   bit 3 (partner 8 lanes away, same 16-lane row): DPP row_ror:8
//   bit 4 (partner row):   v_permlane16_swap_b32 (gfx950) - swaps the odd rows of one register with the even rows of another
//   bit 5 (partner half):  v_permlane32_swap_b32 (gfx950) - swaps the upper half of one register with the lower half of another
// After the three steps lane (u, c) holds NV / 8 complete sums: result i is the sum of input index reduce_u_index<NV>(i, lane).
// ~2 VALU operations per input value, against 6 DPP steps per value for a plain wave reduction.
#pragma once
#include <hip/hip_runtime.h>

namespace dust {

typedef unsigned int t2_v2u __attribute__((ext_vector_type(2)));

template <int NV>
__device__ __forceinline__ int reduce_u_index(const int i, const int lane) {
  return i + (NV / 8) * ((lane >> 5) & 1) + (NV / 4) * ((lane >> 4) & 1) + (NV / 2) * ((lane >> 3) & 1);
}

// The two swaps as inline assembly: hipcc (ROCm 7.2) folds `s.x + s.y` of __builtin_amdgcn_permlane16_swap's result pair into
// `v_add_f32 v1, v1, v1` (both results taken from the first register; seen in tools/reduce_probe.hip).  s_nop 1 in front: the
// hazard recogniser does not see inside the asm (VALU write -> lane-swap read needs wait states).
__device__ __forceinline__ void swap16(float &a, float &b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void swap32(float &a, float &b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }

// One butterfly step on ONE element pair without selects: the lanes of the banks in MASK_LO keep element a (their partner's a comes in
// through the DPP operand), the lanes of the other banks keep element b - two bank-masked `v_add_f32_dpp` into the same register.  The
// select form (keep = bit ? b : a; give = bit ? a : b; keep + dpp(give)) costs two v_cndmask + two v_mov_dpp + one add per pair - 20
// issue cycles against 8.7 (tools/valu_rate_probe.hip: v_cndmask / DPP 4.4 cycles each, a plain add 2.6) - and the reductions were 10 %
// of the tick's vector issue.  Four pairs per statement: the leading s_nop covers the VALU-write -> DPP-read hazard, which the
// compiler's hazard recogniser cannot see inside the asm.
#define DUST_DPP_PAIR4(CTRL_LO, MASK_LO, CTRL_HI, MASK_HI)                                                                                   \
  asm volatile("s_nop 1\n\t"                                                                                                                 \
               "v_add_f32_dpp %0, %4, %4 " CTRL_LO " row_mask:0xf bank_mask:" MASK_LO "\n\t"                                                   \
               "v_add_f32_dpp %1, %5, %5 " CTRL_LO " row_mask:0xf bank_mask:" MASK_LO "\n\t"                                                   \
               "v_add_f32_dpp %2, %6, %6 " CTRL_LO " row_mask:0xf bank_mask:" MASK_LO "\n\t"                                                   \
               "v_add_f32_dpp %3, %7, %7 " CTRL_LO " row_mask:0xf bank_mask:" MASK_LO "\n\t"                                                   \
               "v_add_f32_dpp %0, %8, %8 " CTRL_HI " row_mask:0xf bank_mask:" MASK_HI "\n\t"                                                   \
               "v_add_f32_dpp %1, %9, %9 " CTRL_HI " row_mask:0xf bank_mask:" MASK_HI "\n\t"                                                   \
               "v_add_f32_dpp %2, %10, %10 " CTRL_HI " row_mask:0xf bank_mask:" MASK_HI "\n\t"                                                 \
               "v_add_f32_dpp %3, %11, %11 " CTRL_HI " row_mask:0xf bank_mask:" MASK_HI                                                        \
               : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3)                                                                                  \
               : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3))
// lane bit 3 (partner 8 lanes away in the 16-lane row: banks 0-1 keep a, banks 2-3 keep b)
__device__ __forceinline__ void dpp_step8(const float a0, const float a1, const float a2, const float a3, const float b0, const float b1, const float b2,
                                          const float b3, float &w0, float &w1, float &w2, float &w3) {
  DUST_DPP_PAIR4("row_ror:8", "0x3", "row_ror:8", "0xc");
}
// lane bit 2 (partner 4 lanes away: banks 0 / 2 keep a - their partner is 4 lanes up -, banks 1 / 3 keep b)
__device__ __forceinline__ void dpp_step4(const float a0, const float a1, const float a2, const float a3, const float b0, const float b1, const float b2,
                                          const float b3, float &w0, float &w1, float &w2, float &w3) {
  DUST_DPP_PAIR4("row_shl:4", "0x5", "row_shr:4", "0xa");
}

template <int NV>
__device__ __forceinline__ void reduce_u(const float (&v)[NV], float (&out)[NV / 8], const int lane) {
  static_assert(NV % 8 == 0, "NV must be a multiple of 8");
  float w[NV / 2];
  if constexpr ((NV / 2) % 4 == 0) {
#pragma unroll
    for (int i = 0; i < NV / 2; i += 4)
      dpp_step8(v[i], v[i + 1], v[i + 2], v[i + 3], v[i + NV / 2], v[i + 1 + NV / 2], v[i + 2 + NV / 2], v[i + 3 + NV / 2], w[i], w[i + 1], w[i + 2], w[i + 3]);
  } else {
    const bool b3 = (lane & 8) != 0;
#pragma unroll
    for (int i = 0; i < NV / 2; ++i) {
      const float keep = b3 ? v[i + NV / 2] : v[i];
      const float give = b3 ? v[i] : v[i + NV / 2];
      const float got = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, give), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
      w[i] = keep + got;
    }
  }
  float x[NV / 4];
#pragma unroll
  for (int i = 0; i < NV / 4; ++i) {
    float a = w[i], b = w[i + NV / 4];
    swap16(a, b);
    x[i] = a + b;
  }
#pragma unroll
  for (int i = 0; i < NV / 8; ++i) {
    float a = x[i], b = x[i + NV / 8];
    swap32(a, b);
    out[i] = a + b;
  }
}

// The same over SIXTEEN sub-slices (lane = (u, c): u = lane >> 2, c = lane & 3): one more butterfly step in front, over lane bit 2
// (partner 4 lanes away: row_shl:4 into banks 0 / 2 and row_shr:4 into banks 1 / 3 of every 16-lane row).  Result i of lane l is the
// sum of input index reduce_u16_index<NV>(i, l); NV / 16 results per lane.
template <int NV>
__device__ __forceinline__ int reduce_u16_index(const int i, const int lane) {
  return i + (NV / 16) * ((lane >> 5) & 1) + (NV / 8) * ((lane >> 4) & 1) + (NV / 4) * ((lane >> 3) & 1) + (NV / 2) * ((lane >> 2) & 1);
}
template <int NV>
__device__ __forceinline__ void reduce_u16(const float (&v)[NV], float (&out)[NV / 16], const int lane) {
  static_assert(NV % 16 == 0, "NV must be a multiple of 16");
  float h[NV / 2];
  if constexpr ((NV / 2) % 4 == 0) {
#pragma unroll
    for (int i = 0; i < NV / 2; i += 4)
      dpp_step4(v[i], v[i + 1], v[i + 2], v[i + 3], v[i + NV / 2], v[i + 1 + NV / 2], v[i + 2 + NV / 2], v[i + 3 + NV / 2], h[i], h[i + 1], h[i + 2], h[i + 3]);
  } else {
    const bool b2 = (lane & 4) != 0;
#pragma unroll
    for (int i = 0; i < NV / 2; ++i) {
      const float keep = b2 ? v[i + NV / 2] : v[i];
      const int give = __builtin_bit_cast(int, b2 ? v[i] : v[i + NV / 2]);
      int got = __builtin_amdgcn_update_dpp(0, give, 0x104 /* row_shl:4 */, 0xf, 0x5, false);
      got = __builtin_amdgcn_update_dpp(got, give, 0x114 /* row_shr:4 */, 0xf, 0xa, false);
      h[i] = keep + __builtin_bit_cast(float, got);
    }
  }
  reduce_u<NV / 2>(h, out, lane);
}

}  // namespace dust
