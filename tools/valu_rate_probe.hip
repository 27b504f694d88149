// VALU issue rates on gfx950: wave64 v_fma_f32 vs v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32, by waves per SIMD (1, 2, 4), independent
// accumulators.  One workgroup on one CU; cycles from s_memtime around an unrolled loop.
//   hipcc --offload-arch=gfx950 -O2 tools/valu_rate_probe.hip -o tools/_valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ void k(float *out, long long *cyc, int iters) {
  float a[8];
  v2f p[8];
  const float x = out[threadIdx.x & 7], y = out[8 + (threadIdx.x & 7)];
  for (int i = 0; i < 8; ++i) { a[i] = x + i; p[i] = v2f{x + i, y - i}; }
  const v2f xv = {x, y}, yv = {y, x};
  typedef float v4f_ __attribute__((ext_vector_type(4)));
  v4f_ m4[4] = {{x, y, x, y}, {y, x, y, x}, {x, x, y, y}, {y, y, x, x}};
  v2f xs;
  xs.x = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x)));
  xs.y = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, y)));
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
        if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(xv), "v"(yv));
        if (KIND == 2) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[i]) : "v"(xv));
        if (KIND == 3) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[i]) : "v"(xv));
        if (KIND == 4) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
        if (KIND == 5) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
        if (KIND == 6) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(x));
        if (KIND == 7) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(*(double *)&p[i]) : "v"(x));
        if (KIND == 8) asm volatile("v_add_f64 %0, %0, %1" : "+v"(*(double *)&p[i]) : "v"(*(const double *)&xv));
        if (KIND == 9) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(*(unsigned long long *)&p[i]) : "v"(__float_as_uint(x)), "v"(__float_as_uint(y)) : "vcc");
        if (KIND == 10) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
        if (KIND == 11) asm volatile("v_mul_hi_u32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
        if (KIND == 12) asm volatile("v_sin_f32 %0, %0" : "+v"(a[i]));
        if (KIND == 13) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
        if (KIND == 14) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(x));
        if (KIND == 15) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
        if (KIND == 16) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(a[i]) : "v"(x) : "s10", "s11");
        if (KIND == 17) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(x) : "vcc");
        if (KIND == 18) asm volatile("v_cmp_lt_f32_e64 s[10:11], %0, %1" : : "v"(a[i]), "v"(x) : "s10", "s11");
        if (KIND == 19) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
        if (KIND == 20) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3c0888c1" : "+v"(a[i]) : "v"(x));
        if (KIND == 21) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
        if (KIND == 22) asm volatile("v_add_f32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(x));
        if (KIND == 23) asm volatile("v_add_f32 %0, s12, %0" : "+v"(a[i]) : : "s12");
        if (KIND == 24) asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
        if (KIND == 25) asm volatile("v_rndne_f32 %0, %0" : "+v"(a[i]));
        if (KIND == 26) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[i]));
        if (KIND == 27) asm volatile("v_and_b32 %0, 1, %0" : "+v"(a[i]));
        if (KIND == 28) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(x), "v"(y));
        if (KIND == 29) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
        if (KIND == 30) asm volatile("v_pk_add_f32 %0, %1, %0 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(p[i]) : "v"(xv));
        if (KIND == 31) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(p[i]) : "v"(xv), "v"(yv));
        if (KIND == 32) asm volatile("v_pk_add_f32 %0, %1, %0 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(p[i]) : "s"(xs));   // packed, one operand a scalar register pair
        if (KIND == 33) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "s"(xs), "v"(yv));
        if (KIND == 34) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(m4[i & 3]) : "v"(x), "v"(y));
        if (KIND == 35) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(m4[i & 3]) : "v"(x), "v"(y));
      }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = m4[0].x + m4[1].y + m4[2].z + m4[3].w;
  for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
  out[64 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}
template <int KIND>
void run(const char *name, float *o, long long *c) {
  const int iters = 2000;
  printf("%-16s", name);
  for (int nt : {64, 256, 512, 1024}) {
    k<KIND><<<1, nt>>>(o, c, iters);
    long long h[16];
    hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
    long long mx = 0;
    for (int w = 0; w < nt / 64; ++w) mx = h[w] > mx ? h[w] : mx;
    // cycles of the SIMD per wave-instruction: elapsed / (instructions per wave * waves per SIMD)
    const double per_wave = (double)mx / (iters * 32.0);
    const int wps = nt <= 256 ? 1 : nt / 256;
    printf("  %4d thr: %5.2f cyc/instr/wave (%5.2f per SIMD slot)", nt, per_wave, per_wave / wps);
  }
  printf("\n");
}
int main() {
  float *o; long long *c;
  hipMalloc(&o, 4096 * 4); hipMalloc(&c, 16 * 8);
  hipMemset(o, 0, 4096 * 4);
  run<0>("v_fma_f32", o, c); run<1>("v_pk_fma_f32", o, c); run<2>("v_pk_add_f32", o, c); run<3>("v_pk_mul_f32", o, c);
  run<4>("v_add_f32", o, c); run<5>("v_exp_f32", o, c); run<6>("v_mov_dpp quad", o, c); run<7>("v_cvt_f64_f32", o, c);
  run<8>("v_add_f64", o, c); run<9>("v_mad_u64_u32", o, c); run<10>("v_mul_lo_u32", o, c); run<11>("v_mul_hi_u32", o, c);
  run<12>("v_sin_f32", o, c); run<13>("v_med3_f32", o, c); run<14>("v_cndmask_b32", o, c); run<15>("v_xor_b32", o, c);
  run<16>("v_cndmask e64 s", o, c); run<17>("v_cmp vcc", o, c); run<18>("v_cmp e64 s", o, c); run<19>("v_fmac_f32", o, c);
  run<20>("v_fmaak_f32", o, c); run<21>("v_mul_f32", o, c); run<22>("v_add_f32_dpp", o, c); run<23>("v_add_f32 sgpr", o, c);
  run<24>("cndmask+add", o, c); run<25>("v_rndne_f32", o, c); run<26>("v_cvt_i32_f32", o, c); run<27>("v_and_b32", o, c);
  run<28>("v_bfi_b32", o, c); run<29>("v_max_f32", o, c); run<30>("v_pk_add neg", o, c); run<31>("v_pk_fma opsel", o, c);
  run<32>("v_pk_add sgpr pair", o, c); run<33>("v_pk_fma sgpr pair", o, c); run<34>("v_mfma_f32_16x16x4_f32", o, c); run<35>("v_mfma_f32_4x4x1_16b", o, c);
  return 0;
}
