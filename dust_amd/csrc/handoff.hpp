// handoff.hpp - small device helpers shared by every kernel header AND by the second translation unit (tick2.hip), which
// must not see the non-template kernels of stein.hpp / forward.hpp: kernel-mode enums, the bounded counter wait, write-through
// scalar loads / stores, the optimiser step, the pendulum's shared-reduction trig, wave priorities.
#pragma once
#include "common.hpp"

namespace dust {

enum { PAIR_PRIOR = 0, PAIR_K1 = 1, PAIR_IMQ = 2, PAIR_LOGP = 3 };  // LOGP: the prior pass of SVMPC.forward - log p only, no gradient
typedef float v4f __attribute__((ext_vector_type(4)));

// Wall-clock bound of every in-launch wait (s_memrealtime ticks, 100 MHz): 50 ms.  The start barriers prove that a grid IS resident, not
// that it STAYS resident: when another PROCESS computes on the device the hardware scheduler time-slices the two processes' queues with
// wave save / restore, and a partially restored grid spins on peers that are still saved while the other process's grid does the same
// (tools/two_process_ticks.py: two processes ticking the product shape deadlock within a second; raising this bound to 2 s only
// delays the report).  Giving up is what resolves it - the other process's grid can then be restored whole - so the bound stays short;
// the host turns the time-out into an error for that tick and takes the context off every kernel that spins on its own grid.
#ifndef DUST_SPIN_TIMEOUT_TICKS
#define DUST_SPIN_TIMEOUT_TICKS 5000000ull
#endif

// Bounded wait of ONE lane on a monotonic arrival counter (wrap-safe compare).  A spin gives up after DUST_SPIN_TIMEOUT_TICKS of wall clock
// (s_memrealtime, 100 MHz) or as soon as another waiter has given up, and raises the flag: the host reports it as an error.
__device__ __forceinline__ bool spin_until(const unsigned int *p, const unsigned int target, unsigned int *flag) {
  unsigned int spins = 0;
  unsigned long long t0 = 0;
#ifndef DUST_SPIN_SLEEP
#define DUST_SPIN_SLEEP 2
#endif
  while ((int)(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
    __builtin_amdgcn_s_sleep(DUST_SPIN_SLEEP);
    if ((++spins & 255u) == 0u) {
      const unsigned long long now = __builtin_amdgcn_s_memrealtime();
      if (!t0) t0 = now;
      else if (now - t0 > DUST_SPIN_TIMEOUT_TICKS || __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        // (one more look: a wait that is complete by now has not failed, whatever the clock says - a wave can find its time gone
        //  because its process was switched out, with every arrival in)
        if ((int)(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) >= 0) return true;
        __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
      }
    }
  }
  return true;
}

// torch.optim.Adam, single-tensor CPU path (svgd.py:115 is the reference's class default), on grad = -phi: lerp_ for exp_avg
// (vectorised form: fmadd(w, grad - m, m)), mul_ + addcmul_ for exp_avg_sq ((value * g) * g), bias corrections / step size /
// sqrt(bias_correction2) as Python floats (double), addcdiv_ as self + (value * m) / denom.  `t` is the 1-based step count
// since the last roll (the optimiser state restarts at every forward(): forward.hpp RollArgs).
__device__ __forceinline__ float adam_step(float th, const float g, float &m, float &v, const float lr, const float beta1, const float beta2,
                                           const float eps, const float t) {
  const float w1 = (float)(1.0 - (double)beta1), w2 = (float)(1.0 - (double)beta2);
  const double bc1 = 1.0 - pow((double)beta1, (double)t), bc2 = 1.0 - pow((double)beta2, (double)t);
  const float value = (float)(-((double)lr / bc1)), bc2s = (float)sqrt(bc2);
  m = fmaf(w1, g - m, m);
  v = v * beta2;
  v = v + (w2 * g) * g;
  const float denom = sqrtf(v) / bc2s + eps;
  return th + (value * m) / denom;
}


// sin(fl(theta + pi_f)) and cos(theta) from ONE Cody-Waite reduction of theta.  fl(theta + pi_f) = theta + pi + e with
// e = (pi_f - pi) - err, err the rounding error of the fp32 add recovered exactly by TwoSum; then
// sin(theta + pi + e) = -(sin theta cos e + cos theta sin e) = -(sin theta + e cos theta) up to e^2 < 1e-13.
// Round 4: the reduction is by PI, not pi / 2 - theta = k pi + r, |r| <= pi / 2: sin theta = (-1)^k S(r), cos theta = (-1)^k C(r) with an
// odd degree-9 / even degree-10 polynomial pair (tools/trig_fit.py: fitted to 5e-9 / 3e-10 - a term more in each changes nothing once the
// polynomials are evaluated in fp32 -, they stay within
// 1.2e-7 / 1.5e-7 ABSOLUTE of sin / cos for |theta| <= 5e4 - the [-pi/4, pi/4] pair of rounds 1-3: 7e-8; one fp32 ulp of the angular
// velocity the value feeds is 1.2e-7 .. 4.8e-7).  Gone with it: the two polynomial swaps and three selects on the quadrant
// (v_cmp / v_cndmask / VOP3 negations: 41 of the step's 171 issue cycles, tools/valu_rate_probe.hip) - the sign is the parity bit of
// k, taken from the mantissa of fl(theta / pi + 1.5 * 2^23), which is also how k is rounded (no v_rndne / v_cvt).  The rollout step of the
// product tick: 56 -> 45 instructions, 171 -> 129 issue cycles.  Valid for |theta| < 2^22 pi (the callers hold |theta| <= 5e4).
// The constants that sit in the ADDEND slot beside a literal multiplier (v_fmamk takes one literal): a caller whose translation unit is
// built without machine LICM (tick2.hip) pins them in registers in front of its step loop - trig_consts_pinned() - instead of
// re-creating them in every step.
struct TrigConsts {
  float magic, s1, c1, pif;
};
__device__ __forceinline__ TrigConsts trig_consts() { return TrigConsts{12582912.0f, -1.980484958e-04f, 2.475986184e-05f, PI_F}; }
__device__ __forceinline__ TrigConsts trig_consts_pinned() {
  TrigConsts k = trig_consts();
  asm volatile("" : "+v"(k.magic), "+v"(k.s1), "+v"(k.c1), "+v"(k.pif));
  return k;
}
__device__ __forceinline__ void pendulum_trig(float th, float *sin_tp, float *cos_th, const TrigConsts K = trig_consts()) {
#ifdef DUST_OLD_TRIG  // A/B only: the pi / 2 reduction of rounds 1-3
  int q_;
  const float r_ = trig_reduce(th, &q_);
  const float ps = poly_sin(r_), pc = poly_cos(r_);
  float sn_ = (q_ & 1) ? pc : ps;
  float cs_ = (q_ & 1) ? ps : pc;
  sn_ = (q_ & 2) ? -sn_ : sn_;
  cs_ = ((q_ + 1) & 2) ? -cs_ : cs_;
  const float tp_ = th + PI_F;
  const float bb_ = tp_ - th;
  const float err_ = (th - (tp_ - bb_)) + (PI_F - bb_);
  const float e_ = 8.742278000372485e-8f - err_;
  *sin_tp = -fmaf(e_, cs_, sn_);
  *cos_th = cs_;
  return;
#endif
  const float magic = K.magic;  // 1.5 * 2^23: fl(x + magic) - magic = rint(x), and bit 0 of fl(x + magic) is the parity of rint(x)
  const float t = fmaf(th, 0.318309886183790671538f, magic);
  const float kf = t - magic;
  const unsigned int sgn = __float_as_uint(t) << 31;
  float r = fmaf(kf, -(2.0f * 1.57079601e+00f), th);  // the first two terms of trig_reduce's pi / 2, doubled (exact); the third,
  r = fmaf(kf, -(2.0f * 3.13916473e-07f), r);         // 1.1e-14 k, stays below 2e-10 for |theta| <= 5e4 (tools/trig_fit.py)
  const float s = r * r;
  float p = 2.596175364e-06f;
  p = fmaf(p, s, K.s1);
  p = fmaf(p, s, 8.332992904e-03f);
  p = fmaf(p, s, -1.666665673e-01f);
  float sn = fmaf(p, r * s, r);
  float q = -2.604016061e-07f;
  q = fmaf(q, s, K.c1);
  q = fmaf(q, s, -1.388836536e-03f);
  q = fmaf(q, s, 4.166663811e-02f);
  q = fmaf(q, s, -5.000000000e-01f);
  float cs = fmaf(q, s, 1.0f);
  sn = __uint_as_float(__float_as_uint(sn) ^ sgn);
  cs = __uint_as_float(__float_as_uint(cs) ^ sgn);
  const float tp = th + K.pif;
  const float bb = tp - th;
  const float err = (th - (tp - bb)) + (K.pif - bb);  // exact: th + PI_F = tp + err
  const float e = 8.742278000372485e-8f - err;       // pi_f - pi
  *sin_tp = -fmaf(e, cs, sn);
  *cos_th = cs;
}

// pendulum_trig for TWO angles in packed fp32 (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32: the same IEEE operations, two per lane-op) -
// the two dynamics samples a rollout lane carries side by side (rollout.hpp, round 6).  Operation for operation the scalar function.
__device__ __forceinline__ void pendulum_trig2(const v2f th, v2f *sin_tp, v2f *cos_th, const TrigConsts K = trig_consts()) {
  auto sp = [](const float v) { return v2f{v, v}; };
  const v2f magic = sp(K.magic);
  const v2f t = __builtin_elementwise_fma(th, sp(0.318309886183790671538f), magic);
  const v2f kf = t - magic;
  const unsigned int sg0 = __float_as_uint(t.x) << 31, sg1 = __float_as_uint(t.y) << 31;
  v2f r = __builtin_elementwise_fma(kf, sp(-(2.0f * 1.57079601e+00f)), th);
  r = __builtin_elementwise_fma(kf, sp(-(2.0f * 3.13916473e-07f)), r);
  const v2f s = r * r;
  v2f p = sp(2.596175364e-06f);
  p = __builtin_elementwise_fma(p, s, sp(K.s1));
  p = __builtin_elementwise_fma(p, s, sp(8.332992904e-03f));
  p = __builtin_elementwise_fma(p, s, sp(-1.666665673e-01f));
  v2f sn = __builtin_elementwise_fma(p, r * s, r);
  v2f q = sp(-2.604016061e-07f);
  q = __builtin_elementwise_fma(q, s, sp(K.c1));
  q = __builtin_elementwise_fma(q, s, sp(-1.388836536e-03f));
  q = __builtin_elementwise_fma(q, s, sp(4.166663811e-02f));
  q = __builtin_elementwise_fma(q, s, sp(-5.000000000e-01f));
  v2f cs = __builtin_elementwise_fma(q, s, sp(1.0f));
  sn.x = __uint_as_float(__float_as_uint(sn.x) ^ sg0);
  sn.y = __uint_as_float(__float_as_uint(sn.y) ^ sg1);
  cs.x = __uint_as_float(__float_as_uint(cs.x) ^ sg0);
  cs.y = __uint_as_float(__float_as_uint(cs.y) ^ sg1);
  const v2f pif = sp(K.pif);
  const v2f tp = th + pif;
  const v2f bb = tp - th;
  const v2f err = (th - (tp - bb)) + (pif - bb);  // exact: th + PI_F = tp + err
  const v2f e = sp(8.742278000372485e-8f) - err;  // pi_f - pi
  *sin_tp = -__builtin_elementwise_fma(e, cs, sn);
  *cos_th = cs;
}

// Workgroup barrier that orders LDS only: global loads issued before it stay in flight across it (wg_sync()
// drains vmcnt as well, which would serialise the prefetches of the tail behind every reduction step).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// v[j % DA] for a kernel-ARGUMENT array without a per-lane index: a lane-varying index into the argument block becomes a
// vector memory load (whose in-order vmcnt wait covers every prefetch issued before it); DA is 1 or 2, so a select does.
template <int DA>
__device__ __forceinline__ float pick_da(const float (&v)[4], int j) {
  return DA == 1 ? v[0] : ((j & 1) ? v[1] : v[0]);
}


__device__ __forceinline__ float ld_sc1(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// an opaque copy of a lane index: everything derived from it is recomputed where it is used instead of being hoisted out of the
// iteration loop and kept in registers across it (the three pair bodies' hoisted addresses alone spilled ~100 VGPRs)
__device__ __forceinline__ int opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}

__device__ __forceinline__ int opaque_s(int v) {  // the same for a wave-uniform value (stays in an SGPR)
  asm volatile("" : "+s"(v));
  return v;
}


}  // namespace dust

// wave priorities by phase (s_setprio): 0 = background.  -DDUST_NO_PRIO compiles them out (A/B measurements).
#ifdef DUST_NO_PRIO
#define DUST_PRIO(x) \
  do {               \
  } while (0)
#else
#define DUST_PRIO(x) __builtin_amdgcn_s_setprio(x)
#endif
// measured at cfg2 (us per tick): owners 3 / prior 2: 155; none: 155; owners 1 / prior 1 (both above the background phases -
// noise drawing, the theta-only half of the Stein tile): 148.5; owners 2 / prior 3: 150
#ifndef DUST_PRIO_OWNER
#define DUST_PRIO_OWNER 1
#endif
#ifndef DUST_PRIO_PRIOR
#define DUST_PRIO_PRIOR 1
#endif

