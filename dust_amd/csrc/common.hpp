// common.hpp - shared host/device declarations for libdust_amd (gfx950 / CDNA4 only).
//
// Build flags matter: -ffp-contract=off.  The rollout follows the reference's fp32 operation order exactly (each torch
// CPU elementwise op rounds once; the pendulum is chaotic and the particle map is discontinuous), so the compiler must
// not fuse a*b+c on its own.  Kernels that WANT fused multiply-adds (pairwise passes) call fmaf() explicitly.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "dust_amd.h"

#define DUST_WAVE 64

// In-kernel phase stamps (s_memtime) exist only in the diagnostic build (-DDUST_STAMPS, tools/kprof.py); the product
// build compiles them to nothing.
#ifdef DUST_STAMPS
#define DUST_STAMP(p, k)                                                                             \
  do {                                                                                               \
    if ((p) && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) (p)[k] = __builtin_amdgcn_s_memtime(); \
  } while (0)
// launch timeline of the one-launch SVGD iteration: 4 words per workgroup, the 100 MHz wall clock at its role's boundaries
#define DUST_TL(p, k)                                                                                                      \
  do {                                                                                                                      \
    if ((p) && threadIdx.x == 0) (p)[4 * blockIdx.x + (k)] = (unsigned long long)__builtin_amdgcn_s_memrealtime();          \
  } while (0)
// persistent tick (persist.hpp): wall-clock stamp into slot k of the workgroup's current iteration block
#define DUST_TLP(p, k)                                                                                     \
  do {                                                                                                      \
    if ((p) && threadIdx.x == 0) (p)[k] = (unsigned long long)__builtin_amdgcn_s_memrealtime();             \
  } while (0)
#else
#define DUST_TLP(p, k) \
  do {                 \
  } while (0)
#define DUST_STAMP(p, k) \
  do {                   \
  } while (0)
#define DUST_TL(p, k) \
  do {                \
  } while (0)
#endif

namespace dust {

// Workgroup barrier.  `s_barrier` does not wait for LDS instructions of the issuing wave that are still queued: a ds_write
// issued just before it can land AFTER another wave's post-barrier ds_read.  __syncthreads() normally carries the
// `s_waitcnt lgkmcnt(0)` that closes this, but the compiler drops it where it believes nothing is pending - observed at a loop
// header whose back edge ends in ds_write_b128 (pairwise_fused_kernel's key commit: 1 launch in ~10 read 4 stale key rows,
// tools/fused_race.hip).  The wait is therefore spelled out; tools/barrier_audit.py checks the generated code.
__device__ __forceinline__ void wg_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------
// Device-side model description (by-value kernel argument; lives in SGPRs / constant cache).
struct DevParam {
  int kind;  // dust_param_kind
  int col;
  double value;
};

typedef float v2f __attribute__((ext_vector_type(2)));  // packed fp32 math: v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 (2 flops per lane-op)

struct DevModel {
  int model;
  int P;
  int log_space;
  int interleave;
  double dt;
  // pendulum: dust/models/pendulum.py:61-100
  DevParam g, mass, length;
  float max_torque, max_speed_pend, w_cos, w_vel;
  // particle: dust/models/particle.py:117-225, dust/utils/obstacle_map.py:64-93
  float max_speed, max_acc;
  int can_crash, with_obstacle;
  float inv_cell, off_x, off_y;
  int nx, ny;
  const uint32_t *grid_bits;  // bit-packed [nx][ny] occupancy (nx*ny bits, row-major), 6 KB for the demo map
  float target[4], w_state[4], w_term[4], w_ctrl[2], w_obs;
};

// Python-float vs fp32-tensor arithmetic, as the interpreter evaluates pendulum.py:93-96.
struct Val {
  int t;
  double d;
  float f;
};
__device__ __forceinline__ float tof(Val v) { return v.t ? v.f : (float)v.d; }
__device__ __forceinline__ Val v_py(double d) { return Val{0, d, 0.f}; }
__device__ __forceinline__ Val v_t(float f) { return Val{1, 0.0, f}; }
__device__ __forceinline__ Val v_mul(Val a, Val b) {
  if (!a.t && !b.t) return v_py(a.d * b.d);
  return v_t(tof(a) * tof(b));
}
__device__ __forceinline__ Val v_div(Val a, Val b) {
  if (!a.t && !b.t) return v_py(a.d / b.d);
  if (a.t && !b.t) return v_t(a.f / (float)b.d);
  if (!a.t && b.t) return v_t((1.0f / b.f) * (float)a.d);  // Tensor.__rtruediv__ = reciprocal() * other
  return v_t(a.f / b.f);
}
__device__ __forceinline__ Val v_sq(Val a) { return a.t ? v_t(a.f * a.f) : v_py(a.d * a.d); }

__device__ __forceinline__ float clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }

__device__ __forceinline__ Val load_param(const DevModel &dm, const DevParam &p, const float *prow) {
  if (p.kind == DUST_PARAM_SAMPLED && prow) {
    float v = prow[p.col];
    if (dm.log_space) v = expf(v);
    return v_t(v);
  }
  if (p.kind == DUST_PARAM_TENSOR0D) return v_t((float)p.value);
  return v_py(p.value);
}

// Per-rollout constants derived from the (sampled) model parameters.
struct Coef {
  float c0, c1;  // pendulum: -3g/(2l), 3/(m l^2) ; particle: mass, unused
};

__device__ __forceinline__ Coef make_coef(const DevModel &dm, const float *prow) {
  Coef c;
  if (dm.model == DUST_MODEL_PENDULUM) {
    Val g = load_param(dm, dm.g, prow), m = load_param(dm, dm.mass, prow), l = load_param(dm, dm.length, prow);
    c.c0 = tof(v_div(v_mul(v_py(-3.0), g), v_mul(v_py(2.0), l)));
    c.c1 = tof(v_div(v_py(3.0), v_mul(m, v_sq(l))));
  } else {
    c.c0 = tof(load_param(dm, dm.mass, prow));
    c.c1 = 0.f;
  }
  return c;
}

// ObstacleMap.get_collisions obstacle_map.py:64-93: floor(x * (1/cell) + offset) -> int64 -> clamp -> gather
__device__ __forceinline__ float collision(const DevModel &dm, float px, float py) {
  float fx = floorf(px * dm.inv_cell + dm.off_x);
  float fy = floorf(py * dm.inv_cell + dm.off_y);
  // float -> int64 of NaN/out-of-range is INT64_MIN on the reference's x86 host, which then clamps to 0
  int ix = (fx >= 0.f && fx <= 9.2e18f) ? (fx > (float)(dm.nx - 1) ? dm.nx - 1 : (int)fx) : 0;
  int iy = (fy >= 0.f && fy <= 9.2e18f) ? (fy > (float)(dm.ny - 1) ? dm.ny - 1 : (int)fy) : 0;
  int bit = ix * dm.ny + iy;
  return (float)((dm.grid_bits[bit >> 5] >> (bit & 31)) & 1u);
}

#define PI_F 3.14159274101257324f /* (float)math.pi */

// sin / cos for the rollout: 3-term Cody-Waite reduction by pi/2 (exact for |x| < 1e5) + degree-7/8 minimax polynomials
// on [-pi/4, pi/4], <= 1.5 ulp - the same class as the reference's vectorised torch kernels (Sleef u10).  ocml's
// sinf/cosf cost ~10x more instructions (Payne-Hanek path compiled in); huge arguments fall back to them.
__device__ __forceinline__ float trig_reduce(float x, int *q) {
  const float k = rintf(x * 0.636619747f);
  *q = (int)k;
  float r = fmaf(k, -1.57079601e+00f, x);
  r = fmaf(k, -3.13916473e-07f, r);
  r = fmaf(k, -5.39030253e-15f, r);
  return r;
}
__device__ __forceinline__ float poly_sin(float r) {
  const float s = r * r;
  float p = 2.86567956e-6f;
  p = fmaf(p, s, -1.98559923e-4f);
  p = fmaf(p, s, 8.33338592e-3f);
  p = fmaf(p, s, -1.66666672e-1f);
  const float t = r * s;
  return fmaf(p, t, r);
}
__device__ __forceinline__ float poly_cos(float r) {
  const float s = r * r;
  float p = 2.44677067e-5f;
  p = fmaf(p, s, -1.38877297e-3f);
  p = fmaf(p, s, 4.16666567e-2f);
  p = fmaf(p, s, -5.00000000e-1f);
  return fmaf(p, s, 1.0f);
}
__device__ __forceinline__ float fast_sinf(float x) {
  if (!(fabsf(x) < 1.0e5f)) return sinf(x);
  int q;
  const float r = trig_reduce(x, &q);
  const float v = (q & 1) ? poly_cos(r) : poly_sin(r);
  return (q & 2) ? -v : v;
}
__device__ __forceinline__ float fast_cosf(float x) {
  if (!(fabsf(x) < 1.0e5f)) return cosf(x);
  int q;
  const float r = trig_reduce(x, &q);
  const float v = (q & 1) ? poly_sin(r) : poly_cos(r);
  return ((q + 1) & 2) ? -v : v;
}

// Sum of a trajectory's instantaneous costs.  The oracle - and rounds 1-3 here - add every step's cost to a double: a v_cvt_f64_f32 and
// a v_add_f64 per step, 9 of the 134 issue cycles of the Pendulum step (tools/valu_rate_probe.hip).  The Pendulum kernels now add FOUR
// consecutive steps in fp32 and that partial sum to the double: 4.2 instead of 8.9 cycles per step.  Three fp32 additions of
// non-negative terms inside a group of four: the group's relative error is <= 1.5 eps, the sum's <= 1.5 eps * (4 / H) of the total plus
// the final rounding - indistinguishable from the double sum at the 1e-5 the costs are held to, and small enough for the tests that
// watch the softmax amplify cost ulps (a plain fp32 sum over H = 30 steps - the reference's own torch.sum, disco.py:325-330 - was
// tried: -2 us per cfg2 tick instead of -1.5, but 2-3 ulp of cost error pushed two amplification-limited checks over their bounds).
// Every Pendulum rollout kernel, and both its trig paths, sums this way: their costs still agree bit for bit.  The Particle kernels'
// packed two-sample loops carry a Kahan-compensated fp32 sum (PairKahan below: obstacle weights of 1e6 beside terms of 1e-2 want the
// compensation); the general one-sample loop and skid-steer keep the plain double sum.
template <int MODEL>
struct CostSum {
  double tot = 0.0;
  float part = 0.f, comp = 0.f;  // Pendulum: the open group of four; Particle: Kahan sum / compensation (PairKahan's arithmetic, one sample)
  __device__ __forceinline__ void add(const float c, const int t) {
    if (MODEL == DUST_MODEL_PENDULUM) {
      part += c;
      if ((t & 3) == 3) {
        tot += (double)part;
        part = 0.f;
      }
    } else {
      const float y = c - comp;
      const float s = part + y;
      comp = (s - part) - y;
      part = s;
    }
  }
  __device__ __forceinline__ void add_weighted(const double w, const float c) { tot += w * (double)c; }  // (sigma-point rollouts: rare)
  __device__ __forceinline__ double total() const { return tot + (double)part; }
};

template <int MODEL>
__device__ __forceinline__ void model_step(const DevModel &dm, const Coef &c, float *x, const float *a) {
  const float dt = (float)dm.dt;
  if (MODEL == DUST_MODEL_PENDULUM) {
    float u = clampf(a[0], -dm.max_torque, dm.max_torque);
    float s = fast_sinf(x[0] + PI_F);
    float t1 = c.c0 * s;
    float t2 = c.c1 * u;
    float thd = x[1] + dt * (t1 + t2);
    thd = clampf(thd, -dm.max_speed_pend, dm.max_speed_pend);
    x[0] = x[0] + thd * dt;
    x[1] = thd;
  } else {
    float ax = clampf(a[0] / c.c0, -dm.max_acc, dm.max_acc);
    float ay = clampf(a[1] / c.c0, -dm.max_acc, dm.max_acc);
    float xd[4] = {x[2], x[3], ax, ay};
    if (dm.can_crash && dm.with_obstacle) {
      float om = 1.0f - collision(dm, x[0], x[1]);
#pragma unroll
      for (int k = 0; k < 4; ++k) x[k] = x[k] + (xd[k] * dt) * om;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) x[k] = x[k] + xd[k] * dt;
    }
    x[2] = clampf(x[2], -dm.max_speed, dm.max_speed);
    x[3] = clampf(x[3], -dm.max_speed, dm.max_speed);
  }
}

// sum_k w_k (x_k - target_k)^2 over the four state dimensions - cost4_pair's sum: every Particle path agrees bit for bit
__device__ __forceinline__ float particle_state_cost(const float *x, const float *target, const float *w) {
  const float d0 = x[0] - target[0], d1 = x[1] - target[1], d2 = x[2] - target[2], d3 = x[3] - target[3];
  return ((d0 * d0) * w[0] + (d1 * d1) * w[1]) + ((d2 * d2) * w[2] + (d3 * d3) * w[3]);
}

template <int MODEL>
__device__ __forceinline__ float inst_cost(const DevModel &dm, const float *x, const float *a) {
  if (MODEL == DUST_MODEL_PENDULUM) {
    float cm = fast_cosf(x[0]) - 1.0f;
    float t1 = dm.w_cos * (cm * cm);
    float t2 = dm.w_vel * (x[1] * x[1]);
    return t1 + t2;
  } else {
    const float sc = particle_state_cost(x, dm.target, dm.w_state);
    double cc = 0.0;
#pragma unroll
    for (int k = 0; k < 2; ++k) cc += (double)((a[k] * a[k]) * dm.w_ctrl[k]);
    float ob = dm.with_obstacle ? dm.w_obs * collision(dm, x[0], x[1]) : 0.0f;
    return (sc + (float)cc) + ob;
  }
}

// inst_cost of the state BEFORE the action, then the step: the Particle model looks the same cell up for the obstacle
// cost and for the crash mask - one lookup serves both.
template <int MODEL>
__device__ __forceinline__ float step_with_cost(const DevModel &dm, const Coef &c, float *x, const float *a,
                                                const float *un = nullptr /* Particle with control noise: the action that drives the dynamics */) {
  if (MODEL == DUST_MODEL_PENDULUM) {
    const float cost = inst_cost<MODEL>(dm, x, a);
    model_step<MODEL>(dm, c, x, a);
    return cost;
  } else {
    const bool crash = dm.can_crash && dm.with_obstacle;
    const float coll = (dm.with_obstacle || crash) ? collision(dm, x[0], x[1]) : 0.f;
    const float sc = particle_state_cost(x, dm.target, dm.w_state);
    double cc = 0.0;
#pragma unroll
    for (int k = 0; k < 2; ++k) cc += (double)((a[k] * a[k]) * dm.w_ctrl[k]);
    const float ob = dm.with_obstacle ? dm.w_obs * coll : 0.0f;
    const float cost = (sc + (float)cc) + ob;
    const float dt = (float)dm.dt;
    const float *ad = un ? un : a;
    float ax = clampf(ad[0] / c.c0, -dm.max_acc, dm.max_acc);
    float ay = clampf(ad[1] / c.c0, -dm.max_acc, dm.max_acc);
    float xd[4] = {x[2], x[3], ax, ay};
    if (crash) {
      float om = 1.0f - coll;
#pragma unroll
      for (int k = 0; k < 4; ++k) x[k] = x[k] + (xd[k] * dt) * om;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) x[k] = x[k] + xd[k] * dt;
    }
    x[2] = clampf(x[2], -dm.max_speed, dm.max_speed);
    x[3] = clampf(x[3], -dm.max_speed, dm.max_speed);
    return cost;
  }
}

// ---- Particle fast path: FINITE operands only (the caller checks, wave-uniformly: finite start state and actions, a finite normal
// non-zero mass).  Same fp32 operations in the same order as step_with_cost<PARTICLE> / term_cost; what changes is the code
// shape: the occupancy lookup is branch-free (selects instead of four exec-mask branches: two independent rollouts of one lane
// then interleave in the scheduler), clamps are single v_med3_f32 (it would swallow a NaN that torch.clamp propagates - hence the
// precondition), and a / mass is formed from the hoisted reciprocal with one correction on the exact residual (Markstein: with
// r = RN(1/m), q = RN(a r), e = a - q m exactly (fma), RN(q + e r) is the correctly rounded quotient - the value the IEEE
// division of the reference produces - for normal operands) instead of the 11-instruction v_div_scale / v_div_fmas sequence.
__device__ __forceinline__ float collision_bf(const DevModel &dm, float px, float py) {
  const float fx = floorf(px * dm.inv_cell + dm.off_x), fy = floorf(py * dm.inv_cell + dm.off_y);
  // negative -> 0 (fmaxf also sends NaN to 0); > 9.2e18 (int64 overflow on the reference's host -> INT64_MIN -> clamp) -> 0
  float cx = fminf(fmaxf(fx, 0.f), (float)(dm.nx - 1)), cy = fminf(fmaxf(fy, 0.f), (float)(dm.ny - 1));
  cx = fx <= 9.2e18f ? cx : 0.f;
  cy = fy <= 9.2e18f ? cy : 0.f;
  const int bit = (int)cx * dm.ny + (int)cy;
  return (float)((dm.grid_bits[bit >> 5] >> (bit & 31)) & 1u);
}
__device__ __forceinline__ float div_by_const(const float a, const float m, const float r /* RN(1/m) */) {
  const float q = a * r;
  return fmaf(fmaf(-q, m, a), r, q);
}
// OBST / CRASH are the (wave-uniform, run-time) flags with_obstacle / can_crash && with_obstacle as TEMPLATE arguments: the caller
// picks the instance outside its time loop, so the loop body is one basic block
template <bool OBST, bool CRASH>
__device__ __forceinline__ float particle_step_cost_fast(const DevModel &dm, const float mass, const float rmass, float *x, const float *a) {
  constexpr bool crash = CRASH;
  const float coll = OBST ? collision_bf(dm, x[0], x[1]) : 0.f;
  const float sc = particle_state_cost(x, dm.target, dm.w_state);
  double cc = 0.0;
#pragma unroll
  for (int k = 0; k < 2; ++k) cc += (double)((a[k] * a[k]) * dm.w_ctrl[k]);
  const float ob = OBST ? dm.w_obs * coll : 0.0f;
  const float cost = (sc + (float)cc) + ob;
  const float dt = (float)dm.dt;
  const float ax = __builtin_amdgcn_fmed3f(div_by_const(a[0], mass, rmass), -dm.max_acc, dm.max_acc);
  const float ay = __builtin_amdgcn_fmed3f(div_by_const(a[1], mass, rmass), -dm.max_acc, dm.max_acc);
  const float xd[4] = {x[2], x[3], ax, ay};
  if (crash) {
    const float om = 1.0f - coll;
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = x[k] + (xd[k] * dt) * om;
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = x[k] + xd[k] * dt;
  }
  x[2] = __builtin_amdgcn_fmed3f(x[2], -dm.max_speed, dm.max_speed);
  x[3] = __builtin_amdgcn_fmed3f(x[3], -dm.max_speed, dm.max_speed);
  return cost;
}
template <bool OBST>
__device__ __forceinline__ float particle_term_cost_fast(const DevModel &dm, const float *x) {
  double sc = 0.0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float d = x[k] - dm.target[k];
    sc += (double)((d * d) * dm.w_term[k]);
  }
  const float ob = OBST ? dm.w_obs * collision_bf(dm, x[0], x[1]) : 0.0f;
  return (float)sc + ob;
}

// ---- two dynamics samples per lane in the halves of packed-fp32 registers (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 process
// both in one issue slot; the rollout loop is VALU-issue-bound - measured 143 instructions per pair-step at ~100 % issue).
// .x = sample A, .y = sample B; same fp32 operations per sample, in the same order, as particle_step_cost_fast.
// Extra precondition (checked by the caller): positions stay small enough that floor(p / cell + off) cannot reach the
// int64-overflow branch, so the cell clamp is one v_med3_f32 (which also sends NaN to 0, as the reference's cast does).
__device__ __forceinline__ v2f collision_pair(const DevModel &dm, const uint32_t *grid, const v2f px, const v2f py) {
  const v2f fx = __builtin_elementwise_floor(px * dm.inv_cell + dm.off_x), fy = __builtin_elementwise_floor(py * dm.inv_cell + dm.off_y);
  const float hx = (float)(dm.nx - 1), hy = (float)(dm.ny - 1), fny = (float)dm.ny;
  v2f cx, cy;
  cx.x = __builtin_amdgcn_fmed3f(fx.x, 0.f, hx);
  cx.y = __builtin_amdgcn_fmed3f(fx.y, 0.f, hx);
  cy.x = __builtin_amdgcn_fmed3f(fy.x, 0.f, hy);
  cy.y = __builtin_amdgcn_fmed3f(fy.y, 0.f, hy);
  const v2f cell = __builtin_elementwise_fma(cx, (v2f){fny, fny}, cy);  // exact: < 2^24 cells
  const uint32_t ia = (uint32_t)(int)cell.x, ib = (uint32_t)(int)cell.y;
  v2f c;
  c.x = (float)__builtin_amdgcn_ubfe(grid[ia >> 5], ia, 1u);  // bit (ia & 31): v_bfe_u32 takes the offset modulo 32
  c.y = (float)__builtin_amdgcn_ubfe(grid[ib >> 5], ib, 1u);
  return c;
}
// The state cost sum_k (d_k d_k) w_k of both samples: the reference's own fp32 products ((d * d) * w, each rounded - particle.py:180-181),
// summed pairwise in fp32, ((t0 + t1) + (t2 + t3)): 8 packed multiplies + 3 packed adds per step pair.  (Until round 5 this was the
// correctly rounded exact sum of the four products - 8 v_cvt_f64_f32 + 6 v_add_f64 + 2 v_cvt_f32_f64, 16 of the ~100 VALU instructions of
// a Particle step pair and the half-rate ones.  The reference's torch fp32 `.sum(-1)` is a sum of this class; the result stays within
// 2 ulp of the exact value: 1e-7 against the 1e-5 bound.  An FMA chain over the squares saves 3 more instructions and was tried: its
// singly rounded terms differ from the reference's doubly rounded ones by an ulp often enough to move the softmax(-cost) of the Particle
// goldens - costs of 6.5e5 at temperature 1, 6 % of a weight per ulp - so the terms stay the reference's.  particle_state_cost below
// is the same sum for one sample: every Particle path agrees bit for bit.)
__device__ __forceinline__ v2f cost4_pair(const v2f *d, const v2f *w) {
  const v2f t0 = (d[0] * d[0]) * w[0], t1 = (d[1] * d[1]) * w[1], t2 = (d[2] * d[2]) * w[2], t3 = (d[3] * d[3]) * w[3];
  return (t0 + t1) + (t2 + t3);
}
// Running sum over the time steps of both samples' step costs: Kahan-compensated in packed fp32 (4 v_pk ops per step pair instead of
// 2 cvt + 2 v_add_f64): the total is the fp32 step costs' sum to ~1 ulp, as the double accumulator's rounded value was.
struct PairKahan {
  v2f sum = {0.f, 0.f}, comp = {0.f, 0.f};
  __device__ __forceinline__ void add(const v2f c) {
    const v2f y = c - comp;
    const v2f t = sum + y;
    comp = (t - sum) - y;
    sum = t;
  }
};
// the per-dimension target / state-cost weights as packed pairs (both halves equal): built once per kernel by the caller - in
// SGPR pairs when the compiler has them to spare, in VGPR pairs (PairK::pin) when it does not (a splat of an odd-indexed SGPR
// otherwise goes through a stack slot: a scratch load, and with it a vmcnt wait behind every store in flight, per time step)
struct PairK {
  v2f target[4], w_state[4];
  __device__ __forceinline__ void load(const DevModel &dm) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      target[k] = (v2f){dm.target[k], dm.target[k]};
      w_state[k] = (v2f){dm.w_state[k], dm.w_state[k]};
    }
  }
  __device__ __forceinline__ void pin() {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      asm volatile("" : "+v"(target[k]));
      asm volatile("" : "+v"(w_state[k]));
    }
  }
};
// coll_io (optional): the occupancy of the CURRENT state on entry (looked up by the previous call), of the NEXT state on exit - the
// lookup's LDS round trip then overlaps whatever the caller does between two steps instead of opening every step
template <bool OBST, bool CRASH>
__device__ __forceinline__ v2f particle_pair_step(const DevModel &dm, const PairK &pk, const uint32_t *grid, const v2f mass, const v2f rmass, v2f *x, const float a0,
                                                  const float a1, const float cc /* control cost of (a0, a1), shared by both samples */,
                                                  v2f *coll_io = nullptr,
                                                  const v2f *un = nullptr /* [2] the actions that DRIVE the two samples when the model adds control noise
                                                                             (particle.py:144-153: the cost sees the raw action) */) {
  v2f coll = {0.f, 0.f};
  if (OBST) coll = coll_io ? *coll_io : collision_pair(dm, grid, x[0], x[1]);
  v2f dk[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) dk[k] = x[k] - pk.target[k];
  v2f cost = cost4_pair(dk, pk.w_state) + cc;
  if (OBST) cost = cost + dm.w_obs * coll;  // (without obstacles the reference adds +0 to a non-negative sum: identity)
  const float dt = (float)dm.dt;
  const v2f av[2] = {un ? un[0] : v2f{a0, a0}, un ? un[1] : v2f{a1, a1}};
  v2f acc[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const v2f q = av[k] * rmass;
    const v2f qq = __builtin_elementwise_fma(__builtin_elementwise_fma(-q, mass, av[k]), rmass, q);
    acc[k].x = __builtin_amdgcn_fmed3f(qq.x, -dm.max_acc, dm.max_acc);
    acc[k].y = __builtin_amdgcn_fmed3f(qq.y, -dm.max_acc, dm.max_acc);
  }
  const v2f xd[4] = {x[2], x[3], acc[0], acc[1]};
  if (CRASH) {
    const v2f om = 1.0f - coll;
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = x[k] + (xd[k] * dt) * om;
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = x[k] + xd[k] * dt;
  }
#pragma unroll
  for (int k = 2; k < 4; ++k) {
    x[k].x = __builtin_amdgcn_fmed3f(x[k].x, -dm.max_speed, dm.max_speed);
    x[k].y = __builtin_amdgcn_fmed3f(x[k].y, -dm.max_speed, dm.max_speed);
  }
  if (OBST && coll_io) *coll_io = collision_pair(dm, grid, x[0], x[1]);
  return cost;
}
__device__ __forceinline__ float particle_ctrl_cost(const DevModel &dm, const float a0, const float a1) {
  const double cc = (double)((a0 * a0) * dm.w_ctrl[0]) + (double)((a1 * a1) * dm.w_ctrl[1]);
  return (float)cc;
}
template <bool OBST>
__device__ __forceinline__ v2f particle_pair_term(const DevModel &dm, const uint32_t *grid, const v2f *x, const v2f *coll_in = nullptr) {
  // (once per trajectory: scalar - packed math would splat the odd-indexed SGPRs of dm.target / dm.w_term through a stack slot)
  const float xa[4] = {x[0].x, x[1].x, x[2].x, x[3].x}, xb[4] = {x[0].y, x[1].y, x[2].y, x[3].y};
  v2f c = {particle_state_cost(xa, dm.target, dm.w_term), particle_state_cost(xb, dm.target, dm.w_term)};
  if (OBST) c = c + dm.w_obs * (coll_in ? *coll_in : collision_pair(dm, grid, x[0], x[1]));
  return c;
}

template <int MODEL>
__device__ __forceinline__ float term_cost(const DevModel &dm, const float *x) {
  if (MODEL == DUST_MODEL_PENDULUM) {
    return inst_cost<MODEL>(dm, x, nullptr);
  } else {
    const float sc = particle_state_cost(x, dm.target, dm.w_term);
    float ob = dm.with_obstacle ? dm.w_obs * collision(dm, x[0], x[1]) : 0.0f;
    return sc + ob;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Wave / block reductions (64-lane wavefronts).  __shfl_xor compiles to ds_bpermute_b32 (an LDS-crossbar round trip,
// ~100+ cycles per step); these use DPP modifiers instead: two quad permutes, row_half_mirror and row_mirror leave every
// lane with its 16-lane row's result, row_bcast:15 / row_bcast:31 (gfx9) carry it across the four rows into lane 63, and
// a v_readlane broadcasts it.  `identity` fills lanes a DPP step does not write (bound_ctrl off keeps `old`).
template <typename F>
__device__ __forceinline__ float wave_reduce_dpp(float v, const float identity, F op) {
  auto dpp = [&](float x, auto ctrl, auto row_mask) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, identity), __builtin_bit_cast(int, x),
                                                                   decltype(ctrl)::value, decltype(row_mask)::value, 0xf, false));
  };
  using I = std::integral_constant<int, 0>;
  (void)sizeof(I);
  v = op(v, dpp(v, std::integral_constant<int, 0xB1>{}, std::integral_constant<int, 0xf>{}));   // quad_perm [1,0,3,2]
  v = op(v, dpp(v, std::integral_constant<int, 0x4E>{}, std::integral_constant<int, 0xf>{}));   // quad_perm [2,3,0,1]
  v = op(v, dpp(v, std::integral_constant<int, 0x141>{}, std::integral_constant<int, 0xf>{}));  // row_half_mirror
  v = op(v, dpp(v, std::integral_constant<int, 0x140>{}, std::integral_constant<int, 0xf>{}));  // row_mirror
  v = op(v, dpp(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{}));  // row_bcast:15 -> rows 1, 3
  v = op(v, dpp(v, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{}));  // row_bcast:31 -> rows 2, 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// the same over each 16-lane row only (every lane of a row gets its row's result)
template <typename F>
__device__ __forceinline__ float row16_reduce(float v, const float identity, F op) {
  auto dpp = [&](float x, auto ctrl) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, identity), __builtin_bit_cast(int, x),
                                                                   decltype(ctrl)::value, 0xf, 0xf, false));
  };
  v = op(v, dpp(v, std::integral_constant<int, 0xB1>{}));   // quad_perm [1,0,3,2]
  v = op(v, dpp(v, std::integral_constant<int, 0x4E>{}));   // quad_perm [2,3,0,1]
  v = op(v, dpp(v, std::integral_constant<int, 0x141>{}));  // row_half_mirror
  v = op(v, dpp(v, std::integral_constant<int, 0x140>{}));  // row_mirror
  return v;
}
// max over aligned groups of 2 / 4 consecutive lanes
__device__ __forceinline__ float pair_max(float v) {
  return fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, -INFINITY), __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false)));
}
__device__ __forceinline__ float quad_max(float v) {
  v = pair_max(v);
  return fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, -INFINITY), __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false)));
}
// sum over aligned groups of 2 / 4 consecutive lanes (every lane of the group gets it)
__device__ __forceinline__ float pair_sum(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
}
__device__ __forceinline__ float quad_sum(float v) {
  v = pair_sum(v);
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));
}
// sum over each aligned group of 8 consecutive lanes
__device__ __forceinline__ float oct_sum(float v) {
  auto dpp = [&](float x, auto ctrl) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, false));
  };
  v += dpp(v, std::integral_constant<int, 0xB1>{});
  v += dpp(v, std::integral_constant<int, 0x4E>{});
  v += dpp(v, std::integral_constant<int, 0x141>{});
  return v;
}
// max over each aligned group of 8 consecutive lanes (all 8 lanes get it)
__device__ __forceinline__ float oct_max(float v) {
  auto dpp = [&](float x, auto ctrl) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, -INFINITY), __builtin_bit_cast(int, x),
                                                                   decltype(ctrl)::value, 0xf, 0xf, false));
  };
  v = fmaxf(v, dpp(v, std::integral_constant<int, 0xB1>{}));   // lane ^ 1
  v = fmaxf(v, dpp(v, std::integral_constant<int, 0x4E>{}));   // lane ^ 2
  v = fmaxf(v, dpp(v, std::integral_constant<int, 0x141>{}));  // row_half_mirror: the other quad of the 8
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  return wave_reduce_dpp(v, 0.f, [](float a, float b) { return a + b; });
}
__device__ __forceinline__ float wave_max(float v) {
  return wave_reduce_dpp(v, -INFINITY, [](float a, float b) { return fmaxf(a, b); });
}
__device__ __forceinline__ float wave_min(float v) {
  return wave_reduce_dpp(v, INFINITY, [](float a, float b) { return fminf(a, b); });
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Block-wide reductions through a small LDS scratch (>= 16 floats). All threads get the result.
enum { RED_SUM = 0, RED_MAX = 1, RED_MIN = 2 };
template <int OP>
__device__ __forceinline__ float block_reduce(float v, float *scratch) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = OP == RED_SUM ? wave_sum(v) : (OP == RED_MAX ? wave_max(v) : wave_min(v));
  wg_sync();  // scratch may still be read from a previous reduction
  if (lane == 0) scratch[wid] = v;
  wg_sync();
  // second level in registers: lane w < nw picks up wave w's value, one more wave reduction (a serial walk over the
  // nw LDS words costs nw dependent reads: ~1.5k cycles per reduction in the 1024-lane single-workgroup kernels)
  float r = lane < nw ? scratch[lane] : (OP == RED_SUM ? 0.f : (OP == RED_MAX ? -INFINITY : INFINITY));
  return OP == RED_SUM ? wave_sum(r) : (OP == RED_MAX ? wave_max(r) : wave_min(r));
}

// ---------------------------------------------------------------------------------------------------------------
// Philox4x32 counter RNG (Salmon et al., SC'11) + Box-Muller.  Used only when the caller supplies no noise.
// 7 rounds: the smallest round count the paper reports as Crush-resistant (10 is its conservative default); the rollout
// kernel is instruction-issue bound and the integer multiplies are quarter rate, so rounds are not free here.
template <int ROUNDS>
__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0, p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;  // one 32x32->64 multiply each
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
  philox4x32<10>(c0, c1, c2, c3, k0, k1, out);
}

// four standard normals from one Philox block.  u in (0,1) from the top 24 bits; radius and angle with the hardware
// transcendentals (v_log_f32 is log2, v_sin/v_cos take revolutions): sqrt(-2 ln u0) cos(2 pi u1), ... (relative error
// ~1e-6: statistical noise generation, never compared bitwise).
__device__ __forceinline__ void philox_normal4(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, float z[4]) {
  uint32_t r[4];
  philox4x32<7>(c0, c1, c2, c3, (uint32_t)seed, (uint32_t)(seed >> 32), r);
  const float k = 5.9604644775390625e-08f, h = 2.98023223876953125e-08f;  // 2^-24, 2^-25
  const float u0 = fmaf((float)(r[0] >> 8), k, h), u1 = fmaf((float)(r[1] >> 8), k, h);
  const float u2 = fmaf((float)(r[2] >> 8), k, h), u3 = fmaf((float)(r[3] >> 8), k, h);
  const float ra = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u0));  // -2 ln 2 * log2 u
  const float rb = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u2));
  z[0] = ra * __builtin_amdgcn_cosf(u1);
  z[1] = ra * __builtin_amdgcn_sinf(u1);
  z[2] = rb * __builtin_amdgcn_cosf(u3);
  z[3] = rb * __builtin_amdgcn_sinf(u3);
}

// EIGHT standard normals from one Philox block: the policy-noise stream of the rollout kernels (rollout.hpp, persist.hpp, tick2.hpp,
// skid.hpp - element j of a row comes from block j >> 3, lane element j & 7; all four draw the same stream).  Each 32-bit word gives one
// Box-Muller pair from two 16-bit uniforms - radius from the low half, angle from the high half: 65 536 radii up to 4.85 sigma times
// 65 536 angles per pair.  Against four normals per block from 24-bit uniforms (rounds 1-3) this halves the Philox rounds per normal -
// the integer multiplies are a third of the draw - at the price of a tail cut at 4.85 sigma (2.4e-6 of the mass of a pair) and a
// polar grid of 2^32 points: policy noise for a sampling controller, compared with the reference statistically only (the reference
// draws torch normals, a different stream on any device).  45 instead of 61 issue cycles per normal (tools/valu_rate_probe.hip).
__device__ __forceinline__ void philox_normal8(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, float z[8]) {
  uint32_t r[4];
  philox4x32<7>(c0, c1, c2, c3, (uint32_t)seed, (uint32_t)(seed >> 32), r);
  const float k = 1.52587890625e-05f, h = 7.62939453125e-06f;  // 2^-16, 2^-17: u = (n + 1/2) / 65536 in (0, 1)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float u = fmaf((float)(r[i] & 0xffffu), k, h), a = fmaf((float)(r[i] >> 16), k, h);
    const float ra = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u));  // sqrt(-2 ln u): v_log_f32 is log2
    z[2 * i] = ra * __builtin_amdgcn_cosf(a);                                                   // v_cos / v_sin take revolutions
    z[2 * i + 1] = ra * __builtin_amdgcn_sinf(a);
  }
}

}  // namespace dust
