// tick2.hpp - the control tick as ONE launch with TWO all-to-all hand-offs per SVGD iteration (owner-computes form).
//
// Replaces (reference file:line): the loop of SVMPC.optimize svmpc.py:97-126 around SVMPC.step svmpc.py:87-95 (likelihood.sample
// likelihoods.py:81-101 -> MultiDISCO._rollout / _compute_cost / forward disco.py:139-394 -> SVMPC.phi svmpc.py:38-83 ->
// optimiser step) and SVMPC.forward svmpc.py:172-200, i.e. the call pair `optimize(); forward()` of
// dust/utils/simulations.py:104-123, for N <= 1024 particles, H * d_a <= 32, prior means aliasing the particles (every tick after
// the first forward: svgd.py:87) and an isotropic prior scale.  Everything else keeps persist.hpp / the launch-per-iteration path.
//
// Why a second form (MI355X): persist.hpp cuts the pairwise work into 32 x 64 tiles owned by separate workgroups, which costs
// four all-to-all hand-offs per iteration (theta -> tiles, prior partials -> owners, score -> tiles, Stein partials -> owners) at
// ~4 us each - 80 of its 144 us per tick at cfg2.  An SVGD iteration needs only two exchanges: every particle's theta (for the
// pair distances) and every particle's score (for sum_j k_ij s_j).  Here the workgroup that OWNS 4 particles also computes their
// rows of every pairwise sum against all N keys, so nothing but "theta published" and "score published" crosses workgroups:
//   * one 1024-lane workgroup per CU, N / 4 workgroups, all resident (start barrier below);
//   * waves 0-7 (R): rollouts of the 4 particles, lane = action sample (tick_owner's arithmetic), then - wave-locally - their
//     samples' softmax pieces and weighted sums; policy noise kept in an LDS tile, every wave redrawing the rows of its own samples
//     (Philox) half while the score hop is in flight and half while the owner lanes update the particles;
//   * waves 8-15 (P): the theta-only half of SVMPC.phi in two passes over all N keys (exact differences to two query rows per
//     wave): the PRIOR pass underneath the rollouts - lane = (key of a 32-key step, column half), squared distances -> LDS, the
//     prior's softmax-weighted sum and mass in registers (t2_prior_pass_w) - and, while the score rows travel, the STEIN pass -
//     lane = (key of 16, column quarter), k_ij -> LDS in place of the distances, repulsion in registers (t2_pair_pass).  Sums over
//     the key lanes leave a wave through the butterflies of tick2_reduce.hpp;
//   * the owner lanes (waves 8-9, lane = (particle, column)) finish the score rows behind the prior pass (at wave priority 3:
//     every other workgroup waits for them) and publish them; after the score hop all 16 waves stream the score rows:
//     sum_j k_ij s_j, then phi and the optimiser step by the owner lanes, who publish the new particles.
//   The theta hop of iteration k+1 hides under its rollouts (which need only the workgroup's own particles): ONE exposed hop
//   per iteration.  DESIGN.md 4.R has the timeline (15.9 us per iteration at cfg2) and what was tried around it.
// Hand-offs: write-through rows (one 128-byte line per particle, 16-byte sc1 stores) -> every storing wave drains vmcnt -> one
// agent-scope add per storing wave on the workgroup's counter shard (in each of T2_NREP replicas); consumers: one wave polls the
// T2_NSH shard lines of ITS replica with sc1 loads, then a workgroup barrier (or an LDS word) (cdna_hip_programming.md Guideline 16).  The exchange
// buffers hold one [N][32] block PER GENERATION (particles: n_iters + 1, score rows: n_iters), which is what lets the consumers
// use plain (L1 / L2-cached) loads - see t2_ld16.
// Start barrier: the first hand-off (the particles of generation 0) doubles as it: workgroup 0 waits (bounded) for every
// workgroup's arrival and publishes go / no-go; the others start their first pass on their own observation and pick "go" up in
// front of barrier B1.  Nothing outside the exchange buffers is written before "go", so a launch whose workgroups cannot all be
// resident (another context on the device) - or that stands behind an earlier launch awaiting its replay (expect_aborts) -
// leaves the state untouched and the host runs that tick, in order, on the launch-per-iteration path instead.
// Arithmetic: element-wise the operations of rollout.hpp / stein.hpp; sums over keys and samples are taken in a different order
// and the sample softmax is merged from two wave-local pieces, so a tick agrees with the other paths to ~1e-6, not bitwise.
#pragma once
#include "handoff.hpp"
#include "tick2_args.hpp"
#include "tick2_reduce.hpp"

namespace dust {

typedef unsigned int t2_v4u __attribute__((ext_vector_type(4)));

// The argument block is read through the kernel-argument segment pointer (scalar loads), and that pointer is made OPAQUE once per
// SVGD iteration: everything derived from the arguments is then recomputed inside the iteration instead of being hoisted out of
// the iteration loop by LICM and kept - i.e. spilled: 270 scalars to VGPR lanes and 60-330 vector registers to scratch, inside
// every phase of the loop - across it.
typedef const Tick2Args __attribute__((address_space(4))) *T2ArgPtr;
__device__ __forceinline__ T2ArgPtr t2_args() {
  T2ArgPtr p = (T2ArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return p;
}

#ifdef DUST_STAMPS
#define T2_TL(w, k)                                                                                                       \
  do {                                                                                                                    \
    if (f->tl && (int)threadIdx.x == 64 * (w) && (k) < 128) f->tl[128 * blockIdx.x + (k)] = (unsigned long long)__builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define T2_TL(w, k) \
  do {              \
  } while (0)
#endif

__device__ __forceinline__ __amdgpu_buffer_rsrc_t t2_rsrc(const float *p, const int n_floats) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, n_floats * 4, 0x00020000);
}
// 16-byte stores that write through to memory (aux 16 = sc1 on gfx950), and PLAIN 16-byte loads of the exchanged rows.  Plain loads
// are coherent here by construction: every generation of particles / score rows has a buffer of its own (T2 exchange buffers are
// [generation][N][32]), so no line of it can sit in this CU's L1 or this XCD's L2 before the generation was written through by its
// owner - and nobody loads it before the arrival counters say so (L1 / L2 start a launch invalidated, as for any kernel that reads
// what an earlier kernel wrote).  Against sc1 loads of ONE reused buffer: 3.2 instead of 4.5 us for a pass over 256 KB per CU
// (tools/allgather_probe.hip, no stale or torn row in 200 rounds), and the rows two waves of a CU share come out of its L1.
__device__ __forceinline__ v4f t2_ld16(__amdgpu_buffer_rsrc_t r, const int byte_off) {
  return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
}
__device__ __forceinline__ void t2_st16(__amdgpu_buffer_rsrc_t r, const int byte_off, const v4f v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(t2_v4u, v), r, byte_off, 0, 16);
}
__device__ __forceinline__ unsigned int t2_lds_ld(const unsigned int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void t2_lds_st(unsigned int *p, unsigned int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ unsigned int t2_shard_wgs(const int sh, const int G) { return (unsigned int)(G / T2_NSH + (sh < G % T2_NSH ? 1 : 0)); }

// COMMIT of a tick.  A workgroup that has passed the tick's last wait reads the time-out flag ONCE (one lane, handed round through
// LDS) and writes persistent state only if it is clear; otherwise it counts itself in status[2].  The host replays a tick nobody
// committed.  This read is not a linearisation point - a wait can give up a moment before the arrival it was waiting for while
// another workgroup sees that arrival and commits - but the host SEES that case (0 < status[2] < workgroups) and reports the tick lost,
// as it did every timed-out tick before.  One compare-and-swap word deciding for everybody closes the window and costs 2.4 us per
// tick (256 agent-scope RMWs on one line: 118.9 against 116.5 us); the window is ~1 us wide at the end of a 50 ms wait.
__device__ __forceinline__ bool t2_commit(unsigned int *status, const int b) {
  const bool ok = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;
  if (!ok) {
    __hip_atomic_fetch_add(status + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (b == 0) __hip_atomic_fetch_add(status + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // the host replays it
  }
  return ok;
}
// lanes [0, T2_NSH) of the calling wave wait until every shard has seen `per_wg * phase` arrivals of each of its workgroups
__device__ __forceinline__ void t2_poll(const unsigned int *lines, const unsigned int per_wg, const unsigned int phase, const int G, const int lane,
                                        unsigned int *flag) {
  if (lane < T2_NSH) {
    const unsigned int target = t2_shard_wgs(lane, G) * per_wg * phase;
    if (target) spin_until(lines + (size_t)lane * T2_CNT_STRIDE, target, flag);
  }
}
// every storing wave for itself: drain its write-through stores, then one agent-scope add
__device__ __forceinline__ void t2_arrive_wave(unsigned int *line /* the workgroup's shard in replica 0 */, const int lane) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane < T2_NREP) __hip_atomic_fetch_add(line + (size_t)lane * T2_NSH * T2_CNT_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#if defined(__HIP_DEVICE_COMPILE__)  // (the host pass sees only the kernel's signature: it cannot read through constant-address-space pointers)
// Pairwise passes of a P wave over its share of the keys.  Lane = (key sub-slice u = lane >> 2, column group c = lane & 3): a step
// is 16 key rows; a lane holds 8 columns (two 16-byte pieces) of its key row.  The per-(query, key) scalar work - the reduction of
// the squared distance over the column lanes, the exponential, the mask - is shared by 4 lanes here; with 8 column lanes of
// 4 columns it made up 60 % of the pass (2 560 instructions per wave against 960; measured again on the split passes below:
// 150 against 130 us per tick, although the 8 x 8 form reads whole lines and every row once).
// The theta-only half of SVMPC.phi (svmpc.py:38-41, 76-83) is cut in two at the point the SCORE needs:
//   T2_PASS_PRIOR  (phase 1, underneath the rollouts; needed for the score rows): prior softmax mass L, weighted sum
//                  a = sum_j e_ij (y_j - x_i); the squared distances go to LDS (dk[key][4]).  The tick runs this pass in the
//                  wide layout since round 4 - t2_prior_pass_w below; the form here is the A/B partner.
//   T2_PASS_STEIN  (phase 4, while the score rows of the other workgroups are in flight - the CU has nothing else to do then but
//                  draw the next noise): k_ij from the kept distances -> LDS (the same dk[key][4], in place), repulsion b = sum_j k'_ij (y_j - x_i)
//                  with k' = k (K1) or k^3 (IMQ).  Re-reads the key rows and re-forms the differences: ~15 % more work in
//                  all, a third of it off the path that the score rows wait for (146 -> 130 us per tick).
//   In both a wave takes TWO of the workgroup's four queries (waves 8-11: queries 0-1, waves 12-15: queries 2-3) and a quarter of
//   the keys: four queries per wave put the accumulators alone over the 128-register budget of a 16-wave workgroup.
//   T2_PASS_LOGP   the log-density pass of SVMPC.forward (svmpc.py:137): L only, four queries per wave, all 16 waves share the keys.
// Keys: the padded rows of the exchange buffer (sc1 loads); every workgroup publishes its particles there at the start of the tick.
// f->steps (a multiple of 16, the host rounds up) = steps of a two-query pass; steps past the last key run on clamped rows, weight 0.
enum { T2_PASS_PRIOR = 0, T2_PASS_STEIN = 1, T2_PASS_LOGP = 2 };
#ifndef T2_PRIO_ROLL
#define T2_PRIO_ROLL 1
#endif
#ifndef T2_PRIO_PPASS
#define T2_PRIO_PPASS 0
#endif
#ifndef T2_NOISE_SPLIT
#define T2_NOISE_SPLIT 1
#endif
#ifndef T2_NOISE_P4
#define T2_NOISE_P4 2  // quarters of a row's Philox blocks drawn in phase 4 (the rest: phase 6)
#endif
#ifndef T2_PRIO_OWNERSTAGE
#define T2_PRIO_OWNERSTAGE 3
#endif

__device__ __forceinline__ float t2_quad_sum(float v) {  // quad permutes never read an invalid lane: bound_ctrl spares the `old` operand
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
}
template <int MODE, int PASS, bool MASK /* the steps cover more than the N keys: those past the last one carry no weight */>
__device__ __forceinline__ void t2_pair_pass(const T2ArgPtr f, const int gen /* theta generation */, const float *thq /* LDS [4][32] own query rows */,
                                             float *dk /* LDS [keys][4]: |y_j - x_q|^2 (PRIOR) -> k_qj (STEIN) */, const float *lml /* LDS [keys] log pi_j */,
                                             const int pw /* unit: (query half, key quarter), or the wave (LOGP) */, const int lane, const float lm_ref,
                                             float (&red)[4] /* PRIOR: reduce_u16<32> of a | L; STEIN: reduce_u16<16> of b; LOGP: L of the 4 queries */,
                                             const int t_begin = 0, const int t_end = -1 /* the unit's steps [t_begin, t_end): even bounds; -1: all */) {
  constexpr int NQ = PASS == T2_PASS_LOGP ? 4 : 2;   // queries per wave
  constexpr int NKW = PASS == T2_PASS_LOGP ? 16 : 4;  // waves that share the keys (forward: all 16 waves of the workgroup)
  const int u = lane >> 2, c = lane & 3, N = f->N;
  const int kw = PASS == T2_PASS_LOGP ? pw : (pw & 3), q0 = PASS == T2_PASS_LOGP ? 0 : (pw >> 2) * 2;
  v2f xq[NQ][4];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const float4 xa = *reinterpret_cast<const float4 *>(&thq[(q0 + q) * T2_ROW + 8 * c]);
    const float4 xb = *reinterpret_cast<const float4 *>(&thq[(q0 + q) * T2_ROW + 8 * c + 4]);
    xq[q][0] = v2f{xa.x, xa.y};
    xq[q][1] = v2f{xa.z, xa.w};
    xq[q][2] = v2f{xb.x, xb.y};
    xq[q][3] = v2f{xb.z, xb.w};
  }
  v2f acc[NQ][4];
  float accL[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
#pragma unroll
    for (int h = 0; h < 4; ++h) acc[q][h] = v2f{0.f, 0.f};
    accL[q] = 0.f;
  }
  const __amdgpu_buffer_rsrc_t rx = t2_rsrc(f->xq + (size_t)gen * N * T2_ROW, N * T2_ROW);
  const int steps = PASS == T2_PASS_LOGP ? f->steps / 4 : f->steps;
  const float cP = f->cP, cS = f->cS;
  // one 16-key step of the pass on the lane's 8 columns (y0 | y1) of key row j
  auto step = [&](const int t, const v4f y0, const v4f y1, const v2f dq /* STEIN: the distances of pass PRIOR */) {
    const int j = (t * NKW + kw) * 16 + u;
    const bool valid = !MASK || j < N;
    const v2f yv[4] = {{y0[0], y0[1]}, {y0[2], y0[3]}, {y1[0], y1[1]}, {y1[2], y1[3]}};
    if (PASS == T2_PASS_STEIN) {
      const float dd[2] = {dq.x, dq.y};
      float kq[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float k;
        if (MODE == PAIR_K1) k = valid ? __builtin_amdgcn_exp2f(dd[q] * cS) : 0.f;
        else k = valid ? __builtin_amdgcn_rsqf(fmaf(dd[q], cS, 1.0f)) : 0.f;
        kq[q] = k;
        const float kp = MODE == PAIR_K1 ? k : (k * k) * k;
        const v2f kk = {kp, kp};
#pragma unroll
        for (int h = 0; h < 4; ++h) acc[q][h] = __builtin_elementwise_fma(kk, yv[h] - xq[q][h], acc[q][h]);
      }
      if (c == 0) *reinterpret_cast<float2 *>(&dk[(size_t)j * 4 + q0]) = float2{kq[0], kq[1]};  // (in place: the quad has read its distances)
    } else {
      const float lm2 = (lml[j] - lm_ref) * 1.44269504088896340736f;
      float dsq[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        v2f z[4];
#pragma unroll
        for (int h = 0; h < 4; ++h) z[h] = yv[h] - xq[q][h];
        v2f d2 = z[0] * z[0];
#pragma unroll
        for (int h = 1; h < 4; ++h) d2 = __builtin_elementwise_fma(z[h], z[h], d2);
        const float dd = t2_quad_sum(d2.x + d2.y);  // |y_j - x_q|^2 over all columns, in every lane of the 4
        dsq[q] = dd;
        const float e = valid ? __builtin_amdgcn_exp2f(fmaf(dd, cP, lm2)) : 0.f;  // prior weight pi_j N(x_q; y_j) / exp(lm_ref)
        accL[q] += e;
        if (PASS == T2_PASS_PRIOR) {
          const v2f ee = {e, e};
#pragma unroll
          for (int h = 0; h < 4; ++h) acc[q][h] = __builtin_elementwise_fma(ee, z[h], acc[q][h]);
        }
      }
      if (PASS == T2_PASS_PRIOR && c == 0) *reinterpret_cast<float2 *>(&dk[(size_t)j * 4 + q0]) = float2{dsq[0], dsq[1]};
    }
  };
  {
    constexpr int PF = 2;  // key steps in flight (deeper prefetch - four steps in registers, an LDS-DMA ring of three: tools/ldsdma_probe.hip -
    v4f ya[PF], yb[PF];    // was measured and bought nothing: the passes follow the bytes they pull through the CU's vector memory path)
    const int tb = t_begin, te = t_end < 0 ? steps : t_end;
    auto issue = [&](const int t, v4f &y0, v4f &y1) {
      const int j = min((t * NKW + kw) * 16 + u, N - 1);
      y0 = t2_ld16(rx, (j * T2_ROW + 8 * c) * 4);
      y1 = t2_ld16(rx, (j * T2_ROW + 8 * c + 4) * 4);
    };
#pragma unroll
    for (int p = 0; p < PF; ++p) issue(min(tb + p, te - 1), ya[p], yb[p]);
    for (int t0 = tb; t0 < te; t0 += PF) {
#pragma unroll
      for (int p = 0; p < PF; ++p) {
        const int t = t0 + p;
        const v4f y0 = ya[p], y1 = yb[p];
        issue(min(t + PF, te - 1), ya[p], yb[p]);  // (the last group re-reads its last rows: no branch in the loop)
        v2f dq = {0.f, 0.f};
        if (PASS == T2_PASS_STEIN) {
          const float2 d = *reinterpret_cast<const float2 *>(&dk[(size_t)((t * NKW + kw) * 16 + u) * 4 + q0]);
          dq = v2f{d.x, d.y};
        }
        step(t, y0, y1, dq);
      }
    }
  }
  if (PASS == T2_PASS_PRIOR) {
    float v[32], r2[2];  // a[2][8] | L[2] | padding
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        v[q * 8 + 2 * h] = acc[q][h].x;
        v[q * 8 + 2 * h + 1] = acc[q][h].y;
      }
      v[16 + q] = accL[q];
    }
#pragma unroll
    for (int i = 18; i < 32; ++i) v[i] = 0.f;
    reduce_u16<32>(v, r2, lane);
    red[0] = r2[0];
    red[1] = r2[1];
  } else if (PASS == T2_PASS_STEIN) {
    float v[16], r1[1];  // b[2][8]
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        v[q * 8 + 2 * h] = acc[q][h].x;
        v[q * 8 + 2 * h + 1] = acc[q][h].y;
      }
    reduce_u16<16>(v, r1, lane);
    red[0] = r1[0];
  } else {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {  // 4 values: a plain all-reduce over u (row shifts, then the LDS crossbar for the rows)
      float s = accL[q];
      s += __shfl_xor(s, 4, 64);
      s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x128, 0xf, 0xf, false));
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      red[q] = s;
    }
  }
}

// The PRIOR pass in the WIDE layout: lane = (key u = lane >> 1, column half c = lane & 1) - a step is 32 key rows, a lane holds 16
// columns of its key row.  The per-(query, key) scalar chain (the column sum of the squared distance, the exponential, the mask) is
// shared by 2 lanes instead of 4 and covers twice the columns: 109 instructions per (32 keys, 2 queries) against 139 in the 16 x 4
// layout (98.5 -> 95.5 us per tick; phase 1 is bound by vector issue - tools/valu_rate_probe.hip - and this pass shares it with the
// rollouts).  Registers: 32 of accumulators, two key steps of 16 (one in flight), 16 of differences; the query rows do NOT fit beside
// them (128 registers at 16 waves per CU) and are read from LDS for every (step, query) - two broadcast addresses per read.
// The Stein pass stays in the 16 x 4 layout: it runs in phase 4, where vector issue is not the bound - the wide form of it (correct,
// with the query rows in registers) cost +9 us, a four-query form that loads every key row once per workgroup +1.3 us; so did a
// prefetch of the next query row here (+1.5 .. 4 us: one spilled register puts a scratch load, i.e. a vmcnt(0), into the key loop).
// Results: red[0] = the complete sum over the unit's keys of value index (lane >> 2 & 1) * 16 + t2_wide_col(lane) of column half c
// (value index = query * 16 + column within the half); red[1] = L of query (lane >> 1) & 1 (every lane).
__device__ __forceinline__ int t2_wide_col(const int lane) {
  return ((lane >> 1) & 1) + 2 * ((lane >> 5) & 1) + 4 * ((lane >> 4) & 1) + 8 * ((lane >> 3) & 1);
}
template <int MODE, bool MASK>
__device__ __forceinline__ void t2_prior_pass_w(const T2ArgPtr f, const int gen, const float *thq, float *dk, const float *lml, const int pw, const int lane,
                                                const float lm_ref, float (&red)[2]) {
  const int u = lane >> 1, c = lane & 1, N = f->N;
  const int kw = pw & 3, q0 = (pw >> 2) * 2;
  const int xoff = opaque((q0 * T2_ROW + 16 * c) * 4);
  auto xrow = [&](const int q, v2f (&x)[8]) {  // (opaque offset: re-read for every (step, query), not hoisted into 32 registers)
    const char *b = reinterpret_cast<const char *>(thq) + opaque(xoff) + q * T2_ROW * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 xa = *reinterpret_cast<const float4 *>(b + 16 * i);
      x[2 * i] = v2f{xa.x, xa.y};
      x[2 * i + 1] = v2f{xa.z, xa.w};
    }
  };
  v2f acc[2][8];
  float accL[2] = {0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int h = 0; h < 8; ++h) acc[q][h] = v2f{0.f, 0.f};
  const __amdgpu_buffer_rsrc_t rx = t2_rsrc(f->xq + (size_t)gen * N * T2_ROW, N * T2_ROW);
  const int steps = f->steps / 2;
  const float cP = f->cP;
  auto step = [&](const int t, const v4f (&y)[4]) {
    const int j = (t * 4 + kw) * 32 + u;
    const bool valid = !MASK || j < N;
    v2f yv[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      yv[2 * i] = v2f{y[i][0], y[i][1]};
      yv[2 * i + 1] = v2f{y[i][2], y[i][3]};
    }
    const float lm2 = (lml[j] - lm_ref) * 1.44269504088896340736f;
    float dsq[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      v2f z[8];
      xrow(q, z);
#pragma unroll
      for (int h = 0; h < 8; ++h) z[h] = yv[h] - z[h];
      v2f d2 = z[0] * z[0];
#pragma unroll
      for (int h = 1; h < 8; ++h) d2 = __builtin_elementwise_fma(z[h], z[h], d2);
      float dd = d2.x + d2.y;
      asm volatile("" : "+v"(dd));  // (keeps the two queries' chains apart: fused into packed operations they are live together)
      dd += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, dd), 0xB1, 0xf, 0xf, true));  // the other column half
      dsq[q] = dd;
      const float e = valid ? __builtin_amdgcn_exp2f(fmaf(dd, cP, lm2)) : 0.f;  // prior weight pi_j N(x_q; y_j) / exp(lm_ref)
      accL[q] += e;
      const v2f ee = {e, e};
#pragma unroll
      for (int h = 0; h < 8; ++h) acc[q][h] = __builtin_elementwise_fma(ee, z[h], acc[q][h]);
      // one (step, query) after the other - left alone, the compiler sinks the accumulation of both steps and both queries behind the
      // four distance chains: 64 registers of differences, the accumulators in scratch
#pragma unroll
      for (int h = 0; h < 8; ++h) asm volatile("" : "+v"(acc[q][h]));
    }
    if (c == 0) *reinterpret_cast<float2 *>(&dk[(size_t)j * 4 + q0]) = float2{dsq[0], dsq[1]};
  };
  auto issue = [&](const int t, v4f (&y)[4]) {
    const int j = min((t * 4 + kw) * 32 + u, N - 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i] = t2_ld16(rx, (j * T2_ROW + 16 * c + 4 * i) * 4);
  };
  {
    v4f ya[4], yb[4];
    issue(0, ya);
    for (int t = 0; t < steps; t += 2) {  // (steps is a multiple of 8; the last group re-reads its last rows: no branch in the loop)
      issue(t + 1, yb);
      step(t, ya);
      issue(min(t + 2, steps - 1), ya);
      step(t + 1, yb);
    }
  }
  float v[32], r2[2];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int h = 0; h < 8; ++h) {
      v[q * 16 + 2 * h] = acc[q][h].x;
      v[q * 16 + 2 * h + 1] = acc[q][h].y;
    }
  reduce_u16<32>(v, r2, lane);  // over lane bits 2-5 ...
  const bool b1 = (lane & 2) != 0;
  {  // ... and over lane bit 1 (partner 2 lanes away: no bank mask tells them apart - the select form, on the last element pair only)
    const float keep = b1 ? r2[1] : r2[0], give = b1 ? r2[0] : r2[1];
    red[0] = keep + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, give), 0x4E, 0xf, 0xf, true));
  }
  {
    const float keep = b1 ? accL[1] : accL[0], give = b1 ? accL[0] : accL[1];
    float sL = keep + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, give), 0x4E, 0xf, 0xf, true));
    sL += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sL), 0x124 /* row_ror:4 */, 0xf, 0xf, false));
    sL += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sL), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
    sL += __shfl_xor(sL, 16, 64);
    sL += __shfl_xor(sL, 32, 64);
    red[1] = sL;
  }
}

// The log-density pass of SVMPC.forward in the same wide layout (lane = (key of a 32-key step, column half)): L only - no accumulators,
// so the four query rows stay in registers.  All 16 waves share the keys: f->steps / 8 steps per wave.  red[q] = L of query q, every lane.
template <bool MASK>
__device__ __forceinline__ void t2_logp_pass_w(const T2ArgPtr f, const int gen, const float *thq, const float *lml, const int wave, const int lane,
                                               const float lm_ref, float (&red)[4]) {
  const int u = lane >> 1, c = lane & 1, N = f->N;
  v2f xq[T2_PW][8];
#pragma unroll
  for (int q = 0; q < T2_PW; ++q)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 xa = *reinterpret_cast<const float4 *>(&thq[q * T2_ROW + 16 * c + 4 * i]);
      xq[q][2 * i] = v2f{xa.x, xa.y};
      xq[q][2 * i + 1] = v2f{xa.z, xa.w};
    }
  float accL[T2_PW] = {0.f, 0.f, 0.f, 0.f};
  const __amdgpu_buffer_rsrc_t rx = t2_rsrc(f->xq + (size_t)gen * N * T2_ROW, N * T2_ROW);
  const int steps = f->steps / 8;
  const float cP = f->cP;
  auto issue = [&](const int t, v4f (&y)[4]) {
    const int j = min((t * 16 + wave) * 32 + u, N - 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i] = t2_ld16(rx, (j * T2_ROW + 16 * c + 4 * i) * 4);
  };
  auto step = [&](const int t, const v4f (&y)[4]) {
    const int j = (t * 16 + wave) * 32 + u;
    const bool valid = !MASK || j < N;
    const float lm2 = (lml[j] - lm_ref) * 1.44269504088896340736f;
#pragma unroll
    for (int q = 0; q < T2_PW; ++q) {
      v2f d2 = v2f{0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const v2f z0 = v2f{y[i][0], y[i][1]} - xq[q][2 * i], z1 = v2f{y[i][2], y[i][3]} - xq[q][2 * i + 1];
        d2 = i == 0 ? z0 * z0 : __builtin_elementwise_fma(z0, z0, d2);
        d2 = __builtin_elementwise_fma(z1, z1, d2);
      }
      float dd = d2.x + d2.y;
      dd += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, dd), 0xB1, 0xf, 0xf, true));  // the other column half
      accL[q] += valid ? __builtin_amdgcn_exp2f(fmaf(dd, cP, lm2)) : 0.f;
    }
  };
  {
    v4f ya[4], yb[4];
    issue(0, ya);
    for (int t = 0; t < steps; t += 2) {  // (steps is even: f->steps is a multiple of 16)
      issue(t + 1, yb);
      step(t, ya);
      issue(min(t + 2, steps - 1), ya);
      step(t + 1, yb);
    }
  }
#pragma unroll
  for (int q = 0; q < T2_PW; ++q) {  // all-reduce over the 32 keys of the lanes (the two column halves hold the same value)
    float sL = accL[q];
    sL += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sL), 0x4E, 0xf, 0xf, true));
    sL += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sL), 0x124 /* row_ror:4 */, 0xf, 0xf, false));
    sL += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sL), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
    sL += __shfl_xor(sL, 16, 64);
    sL += __shfl_xor(sL, 32, 64);
    red[q] = sL;
  }
}

// the general (reference-order) step / terminal cost with the Particle's occupancy grid read from LDS: the model block is copied
// only inside the Particle instance, where it is used
template <int MODEL>
__device__ __forceinline__ float t2_step_with_cost(const T2ArgPtr f, const float *grid_lds, const Coef &cf, float *x, const float *a) {
  DevModel dml = f->dm;  // (a private copy: only the fields the instance uses survive)
  if (MODEL == DUST_MODEL_PARTICLE) dml.grid_bits = reinterpret_cast<const uint32_t *>(grid_lds);
  return step_with_cost<MODEL>(dml, cf, x, a);
}
template <int MODEL>
__device__ __forceinline__ float t2_term_cost(const T2ArgPtr f, const float *grid_lds, const float *x) {
  DevModel dml = f->dm;
  if (MODEL == DUST_MODEL_PARTICLE) dml.grid_bits = reinterpret_cast<const uint32_t *>(grid_lds);
  return term_cost<MODEL>(dml, x);
}

#endif  // __HIP_DEVICE_COMPILE__

template <int MODEL, int MODE>
__global__ __launch_bounds__(T2_NT, 4) void svmpc_tick2_kernel(const Tick2Args f_by_value) {
  (void)f_by_value;
#if defined(__HIP_DEVICE_COMPILE__)
  const T2ArgPtr f0 = t2_args();
  T2ArgPtr f = f0;
  constexpr int DS = MODEL == DUST_MODEL_PENDULUM ? 2 : 4;
  constexpr int DA = MODEL == DUST_MODEL_PENDULUM ? 1 : 2;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid0 = (int)threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int S = f->S, D = f->D, H = f->H, N = f->N, M = f->M, Dp = D | 1;
  const int G = (int)gridDim.x, b = (int)blockIdx.x, n_first = b * T2_PW, sh = b % T2_NSH;
  const Tick2Lds L = tick2_lds(S, D, f->steps, f->grid_words);
  float *tile = lds + L.tile;    // [4][S][Dp] standard normals of the current iteration
  float *cst = lds + L.cst;      // [4][S] costs -> softmax weights
  float *omg = lds + L.omg;      // [4][S] omega weights (alpha * temp != 1)
  float *th = lds + T2_L_TH;     // [4][32] the workgroup's particles, zero padded
  float *misc = lds + T2_L_MISC;
  float *red_w = misc;           // [4][2][4] per (particle, wave): cost min, sum exp, sum exp (omega), cost sum
  float *ll = misc + 40;         // [4] log-likelihood of the last sample (SVMPC.forward, fast_pred)
  float *flag_th = misc + 44;    // [4] non-finite particle
  float *flag_eps = misc + 48;   // [2][4] non-finite caller-supplied noise, by iteration parity (iteration k + 1's noise is staged while
                                 // iteration k's flags are still to be cleared: phase 6)
  float *mbx = misc + 32;        // [4] plant state of an armed launch, as it arrived in the mailbox
  // (misc[36..39], misc[56..59] unused)
  unsigned int *sig = reinterpret_cast<unsigned int *>(misc + 60);  // [0] go (1) / abort (2)  [1] theta generations arrived  [2] COMMIT (1) / not (0)
                                                                    // [3] armed launch: state arrived (1) / cancelled or never came (2)
  float *wred = misc + 64;       // [128] block-reduction scratch
  float *coefs = lds + T2_L_COEFS;
  float *ksl = lds + L.ksl;
  float *ppart = lds + T2_L_PPART;
  float *rpart = lds + T2_L_RP;
  float *lml = lds + L.lml;      // [steps * 64] log pi_j of the tick's prior (read by every pair pass; constant over the tick)
  float *wpart = lds + T2_L_WPART;
  float *scl = lds + T2_L_SCL;
  unsigned int *cnt_start = f->cnt;  // (replica 0 only: one arrival per workgroup, polled by workgroup 0)
  unsigned int *cnt_theta = f->cnt + (size_t)1 * T2_KIND * T2_CNT_STRIDE;
  unsigned int *cnt_score = f->cnt + (size_t)2 * T2_KIND * T2_CNT_STRIDE;
  unsigned int *cnt_lw = f->cnt + (size_t)3 * T2_KIND * T2_CNT_STRIDE;
  unsigned int *go = f->cnt + (size_t)4 * T2_KIND * T2_CNT_STRIDE;
  const size_t rep_off = (size_t)(b % T2_NREP) * T2_NSH * T2_CNT_STRIDE;  // the replica this workgroup polls
  unsigned int *tflag = f->status;

  if (b == 0)
    for (int t = tid0; t < T2_SETS; t += T2_NT) f->zero_base[(size_t)t * T2_CNT_STRIDE] = 0u;
  if (tid0 == 15 * 64) __hip_atomic_fetch_add(cnt_start + (size_t)sh * T2_CNT_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  T2_TL(0, 120);

  // owner lanes: waves 8-9, lane -> (particle op, column od); they keep theta / a_mat / Adam moments in registers for the tick
  const bool isown = wave == 8 || wave == 9;
  const int ol = tid0 - 512, op = (ol >> 5) & 3, od = ol & 31;
  const bool ownv = isown && od < D;
  const size_t no = (size_t)(n_first + op) * D + (ownv ? od : 0);
  const bool adam = f->optimizer == DUST_OPT_ADAM;
  float thv = 0.f, amv = 0.f, adm = 0.f, adv = 0.f;
  if (ownv) {
    thv = f->theta[no];
    if (f->update_a_mat) amv = f->a_mat[no];
    if (adam) {
      adm = f->adam_m[no];
      adv = f->adam_v[no];
    }
  }
  const uint32_t ctr_tick = f->ctr[0], ctr_iter0 = f->ctr[1], adam0 = f->ctr[2];
  float x0[DS];
#pragma unroll
  for (int k = 0; k < DS; ++k) x0[k] = f->x0[k];
  if (tid0 < 16) misc[44 + tid0] = 0.f;  // flags
  if (tid0 == 16) {
    sig[0] = 0u;
    sig[1] = 0u;
    sig[3] = 0u;
  }
  float ll0 = 0.f;
  if (tid0 >= 32 && tid0 < 32 + T2_PW) ll0 = f->logl[n_first + tid0 - 32];
  float lm_max = -INFINITY;  // max_j log pi_j: an upper bound of every prior logit (the exponent reference of the pair passes)
  for (int i = tid0; i < f->steps * 64; i += T2_NT) {
    const float lmi = f->logmix[min(i, N - 1)];
    lml[i] = lmi;
    lm_max = fmaxf(lm_max, lmi);
  }
  if (MODEL == DUST_MODEL_PARTICLE) {  // occupancy grid -> LDS
    uint32_t *gridl = reinterpret_cast<uint32_t *>(lds + L.grid);
    const int words = f->dm.with_obstacle ? (f->dm.nx * f->dm.ny + 31) >> 5 : 0;
    for (int w = tid0; w < words; w += T2_NT) gridl[w] = f->dm.grid_bits[w];
  }
  lds_barrier();  // orders the flag reset against the first noise staging; the loads above stay in flight across it

  // publish rows of the workgroup's particles held in LDS ([4][32]) as whole 128-byte lines: waves 8 / 9 take two rows each
  auto publish_rows = [&](const float *src, const float *dst, unsigned int *lines) {
    const __amdgpu_buffer_rsrc_t r = t2_rsrc(dst, N * T2_ROW);
    const int lane = tid0 & 63;
    if (lane < 16) {
      const int pl = (wave - 8) * 2 + (lane >> 3), c = lane & 7;
      const float4 v = *reinterpret_cast<const float4 *>(&src[pl * T2_ROW + 4 * c]);
      t2_st16(r, ((n_first + pl) * T2_ROW + 4 * c) * 4, v4f{v.x, v.y, v.z, v.w});
    }
    t2_arrive_wave(lines + (size_t)sh * T2_CNT_STRIDE, lane);
  };


  // dynamics coefficients of iteration k (rollout_body stage 1); wave 11
  auto make_coefs = [&](const T2ArgPtr f, const int k) {
    const DevModel dmc = f->dm;
    for (int m = (tid0 & 63); m < M; m += 64) {
      if (f->coef_given) {
        coefs[2 * m] = f->coef_host[0];
        coefs[2 * m + 1] = f->coef_host[1];
      } else {
        const Coef cf = make_coef(dmc, f->params ? f->params + ((size_t)k * M + m) * f->dm.P : nullptr);
        coefs[2 * m] = cf.c0;
        coefs[2 * m + 1] = cf.c1;
      }
    }
  };
  // standard normals of iteration k into the tile: `nl` lanes (lane index `lid`) share the 4 S rows; with twice as many lanes as
  // rows two lanes split a row's Philox blocks.  Same counter layout as rollout.hpp / persist.hpp: element (s, n, j).
  auto draw_noise = [&](const T2ArgPtr f, const int k, const int lid, const int nl) {
    const int rows = T2_PW * S;
    if (f->eps == nullptr) {
      const int split = nl >= 2 * rows ? 2 : 1;
      const int half = split == 2 ? (lid & 1) : 0;
      for (int r = split == 2 ? (lid >> 1) : lid; r < rows; r += nl / split) {
        const int p = r / S, s = r - p * S;
        float *row = tile + (size_t)r * Dp;
        for (int j8 = half; j8 * 8 < D; j8 += split) {
          float z[8];
          philox_normal8(f->seed, (uint32_t)j8, (uint32_t)(s * N + n_first + p), ctr_iter0 + (uint32_t)k, ctr_tick, z);
#pragma unroll
          for (int q = 0; q < 8; ++q)
            if (j8 * 8 + q < D) row[j8 * 8 + q] = z[q];
        }
      }
    } else {
      const float *base = f->eps + (size_t)k * f->eps_stride;
      const int total = rows * D;
      for (int e0 = lid; e0 < total; e0 += 8 * nl) {
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int e = min(e0 + q * nl, total - 1);
          const int r = e / D, j = e - r * D, p = r / S, s = r - p * S;
          v[q] = base[((size_t)s * N + n_first + p) * D + j];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int e = e0 + q * nl;
          if (e < total) {
            const int r = e / D, j = e - r * D;
            tile[(size_t)r * Dp + j] = v[q];
            if (!(fabsf(v[q]) <= 3.0e38f)) flag_eps[(k & 1) * T2_PW + r / S] = 1.f;
          }
        }
      }
    }
  };
  // device noise of iteration k, by the rollout waves: every wave draws the rows of ITS OWN samples (particle wave >> 1, samples
  // (wave & 1) * 64 + lane + 128 i), so a wave may go from its draw into its rollouts - and from its weighted sums into the next draw -
  // without a barrier in between
  auto draw_own_rows = [&](const T2ArgPtr f, const int k, const int wave, const int lane, const int part = 2 /* 0 / 1: the first / second half of the row's Philox blocks; 2: all */) {
    const int rp = wave >> 1;
    const int nb = (D + 7) >> 3, nh = T2_NOISE_SPLIT ? (nb * T2_NOISE_P4) >> 2 : nb;
    const int jb = part == 1 ? nh : 0, je = part == 0 ? nh : nb;
    for (int s = (wave & 1) * 64 + lane; s < S; s += 128) {
      float *row = tile + (size_t)(rp * S + s) * Dp;
      for (int j8 = jb; j8 < je; ++j8) {
        float z[8];
        philox_normal8(f->seed, (uint32_t)j8, (uint32_t)(s * N + n_first + rp), ctr_iter0 + (uint32_t)k, ctr_tick, z);
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (j8 * 8 + q < D) row[j8 * 8 + q] = z[q];
      }
    }
  };
  // the workgroup's particles into LDS and out to the other workgroups FIRST (generation 0; the exchange buffer is scratch: harmless if
  // the tick does not start): their hop runs under the first noise draw (116.35 -> 115.8 us per tick, A/B)
  if (isown) {
    th[op * T2_ROW + od] = thv;
    const unsigned long long badm = __ballot(ownv && !(fabsf(thv) <= 3.0e38f));
    if ((tid0 & 31) == 0) flag_th[op] = ((badm >> (tid0 & 32)) & 0xffffffffull) ? 1.f : 0.f;
    publish_rows(th, f->xq, cnt_theta);
  }
  T2_TL(0, 122);
  const bool own_first = f->n_iters > 0 && f->eps == nullptr;  // device noise: drawn behind the barrier, each rollout wave its own rows
  if (f->n_iters > 0) {
    if (!own_first) draw_noise(f, 0, tid0, T2_NT);
    if (wave == 11) make_coefs(f, 0);
  }
  // ... and now what the other loads brought: the likelihood of the last sample, the logit reference
  if (tid0 >= 32 && tid0 < 32 + T2_PW) ll[tid0 - 32] = ll0;
  lm_max = wave_max(lm_max);
  if ((tid0 & 63) == 0) wred[wave] = lm_max;
  // start barrier of a launch WITHOUT iterations (SVMPC.forward alone; wave 15): workgroup 0 collects the arrivals - bounded: ~200 us -
  // and publishes go / abort
  auto start_protocol = [&]() {
    const int lane = tid0 & 63;
    if (b == 0) {
      bool ok = true;
      if (lane < T2_NSH) {
        const unsigned int target = t2_shard_wgs(lane, G);
        const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
        while (target && (int)(__hip_atomic_load(cnt_start + (size_t)lane * T2_CNT_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
          __builtin_amdgcn_s_sleep(2);
          if (__builtin_amdgcn_s_memrealtime() - t_start > 20000ull) {
            ok = false;
            break;
          }
        }
      }
      const bool all_ok = __all(ok ? 1 : 0) && __hip_atomic_load(tflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && f->test_abort != 1 &&
                          __hip_atomic_load(f->status + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == f->expect_aborts;  // (an earlier tick awaits its replay)
      if (lane == 0) {
        if (!all_ok) __hip_atomic_fetch_add(f->status + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(go, all_ok ? 1u : 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t2_lds_st(sig, all_ok ? 1u : 2u);
      }
    } else if (lane == 0) {
      unsigned int g = 0u, spins = 0u;
      unsigned long long t_start = 0;
      for (;;) {
        g = __hip_atomic_load(go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (g) break;
        __builtin_amdgcn_s_sleep(2);
        if ((++spins & 255u) == 0u) {
          const unsigned long long now = __builtin_amdgcn_s_memrealtime();
          if (!t_start) t_start = now;
          else if (now - t_start > DUST_SPIN_TIMEOUT_TICKS) {  // workgroup 0 never came: give up (reported as a time-out)
            __hip_atomic_store(tflag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            g = 2u;
            break;
          }
        }
      }
      t2_lds_st(sig, g);
    }
  };
  // With SVGD iterations in the launch the FIRST HAND-OFF is the start barrier: every workgroup publishes its particles (generation 0)
  // before anything else, so "all generation-0 rows have arrived" proves that all workgroups are resident.  Workgroup 0 decides - it
  // waits (bounded: ~200 us) for the arrivals and publishes go / abort; the others start their pair pass over generation 0 as soon as
  // THEY see the arrivals complete (the pass reads the exchange buffer and writes LDS only) and pick the go word up behind it, in front
  // of barrier B1, the first point where a workgroup writes anything another one reads.  Against a start barrier of its own in front of
  // the first hand-off (round 3) this takes two round trips out of the head of the tick (~3 us).
  auto gen0_wait = [&]() {  // wave 15
    const int lane = tid0 & 63;
    bool ok = true;
    if (lane < T2_NSH) {
      const unsigned int target = t2_shard_wgs(lane, G) * 2u;
      const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
      unsigned int spins = 0u;
      while (target && (int)(__hip_atomic_load(cnt_theta + rep_off + (size_t)lane * T2_CNT_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
        __builtin_amdgcn_s_sleep(2);
        if ((++spins & 15u) == 0u) {
          const unsigned long long waited = __builtin_amdgcn_s_memrealtime() - t_start;
          if (b == 0 ? waited > 20000ull
                     : (waited > DUST_SPIN_TIMEOUT_TICKS || __hip_atomic_load(go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 2u)) {
            ok = false;
            break;
          }
        }
      }
    }
    const bool all_in = __all(ok ? 1 : 0);
    if (b == 0) {
      const bool all_ok = all_in && __hip_atomic_load(tflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && f->test_abort != 1 &&
                          __hip_atomic_load(f->status + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == f->expect_aborts;  // (an earlier tick awaits its replay)
      if (lane == 0) {
        if (!all_ok) __hip_atomic_fetch_add(f->status + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(go, all_ok ? 1u : 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t2_lds_st(sig, all_ok ? 1u : 2u);
      }
    } else if (!all_in && lane == 0) {  // workgroup 0 said abort, or nothing came for 50 ms (reported as a time-out)
      if (__hip_atomic_load(go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 2u) __hip_atomic_store(tflag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      t2_lds_st(sig, 2u);
    }
  };
  auto go_wait = [&]() {  // wave 15 of workgroups 1 ..: the go word, published by workgroup 0 when it saw generation 0 complete
    if ((tid0 & 63) == 0 && t2_lds_ld(sig) == 0u) {
      unsigned int g = 0u, spins = 0u;
      unsigned long long t_start = 0;
      for (;;) {
        g = __hip_atomic_load(go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (g) break;
        __builtin_amdgcn_s_sleep(2);
        if ((++spins & 255u) == 0u) {
          const unsigned long long now = __builtin_amdgcn_s_memrealtime();
          if (!t_start) t_start = now;
          else if (now - t_start > DUST_SPIN_TIMEOUT_TICKS) {  // workgroup 0 never came: give up (reported as a time-out)
            __hip_atomic_store(tflag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            g = 2u;
            break;
          }
        }
      }
      t2_lds_st(sig, g);
    }
  };
  wg_sync();
  T2_TL(0, 121);
  const float lm_ref = wave_max((tid0 & 63) < T2_NT / 64 ? wred[tid0 & 63] : -INFINITY);
  const bool same_w = (f->alpha * f->temp == 1.0f);
  const int lane0 = tid0 & 63;
  if (own_first && wave < 8) draw_own_rows(f, 0, wave, lane0);  // (the pair waves are on their way to the first hand-off meanwhile)

  const int wave0 = wave;
  for (int k = 0; k < f->n_iters; ++k) {
    // opaque copies of the lane and wave index: everything derived from them is recomputed where it is used instead of being hoisted
    // out of the iteration loop and kept (or spilled) across it
    T2ArgPtr f = t2_args();  // (made opaque again behind every barrier: a phase recomputes what it needs from the arguments)
    const int S = f->S, D = f->D, H = f->H, N = f->N, M = f->M, Dp = D | 1;
    const int tid = opaque(tid0);
    const int wave = opaque_s(wave0);
    const int lane = tid & 63;
    const bool isown = wave == 8 || wave == 9;
    const int ol = tid - 512, op = (ol >> 5) & 3, od = ol & 31;
    const bool ownv = isown && od < D;
    const size_t no = (size_t)(n_first + op) * D + (ownv ? od : 0);
    T2_TL(0, 16 * k + 0);
    bool mb_cancel = false;
    if (k == 0 && f->mbox != nullptr && wave < 8) {
      // armed launch (tick2_args.hpp T2Mbox): the plant state comes through the pinned-host mailbox.  Workgroup 0 polls it - one aligned
      // 16-byte system-scope load per round trip (~2 us over PCIe) - and relays state and verdict through device memory; in every
      // workgroup wave 0 fetches them and the other rollout waves wait on an LDS word.  Bounded: a host that never supplies the state
      // leaves the launch as one that did not start.
      if (wave == 0) {
        unsigned int verdict = 0u;
        float xs[4] = {0.f, 0.f, 0.f, 0.f};
        if (lane == 0) {
          // relay lines: kind 0 of this launch's counter set uses replica 0 (start arrivals) and line T2_NSH (done arrivals); lines
          // 2 T2_NSH .. 2 T2_NSH + T2_NREP - 1 carry {verdict, x0..x3} from workgroup 0 to the others, workgroup b reading line b % T2_NREP
          // (256 workgroups polling the HOST line themselves serialise at the root complex: 121 us per tick instead of 93)
          unsigned int *relay = cnt_start + (size_t)2 * T2_NSH * T2_CNT_STRIDE;
          const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
          if (b == 0) {
            const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned int *>(f->mbox), 0, 32, 0x00020000);
            for (;;) {
              const t2_v4u a = __builtin_amdgcn_raw_buffer_load_b128(rm, 0, 0, 17);  // sc0 sc1: system scope
              if (a[0] == f->launch_seq && a[1] != 0u) {
                verdict = a[1] == 1u ? 1u : 2u;
                xs[0] = __uint_as_float(a[2]);
                xs[1] = __uint_as_float(a[3]);
                if (DS > 2 && verdict == 1u) {
                  const t2_v4u bq = __builtin_amdgcn_raw_buffer_load_b128(rm, 16, 0, 17);
                  if (bq[2] != f->launch_seq) continue;  // (the halves of two different posts: look again)
                  xs[2] = __uint_as_float(bq[0]);
                  xs[3] = __uint_as_float(bq[1]);
                }
                break;
              }
              if (__builtin_amdgcn_s_memrealtime() - t_start > f->mbox_wait) {
                verdict = 2u;
                break;
              }
            }
#pragma unroll
            for (int r = 0; r < T2_NREP; ++r)
#pragma unroll
              for (int q = 0; q < 4; ++q) __hip_atomic_store(relay + (size_t)r * T2_CNT_STRIDE + 1 + q, __float_as_uint(xs[q]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int r = 0; r < T2_NREP; ++r) __hip_atomic_store(relay + (size_t)r * T2_CNT_STRIDE, verdict, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } else {
            unsigned int *line = relay + (size_t)(b % T2_NREP) * T2_CNT_STRIDE;
            unsigned int spins = 0u;
            for (;;) {
              verdict = __hip_atomic_load(line, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (verdict) break;
              __builtin_amdgcn_s_sleep(2);
              if ((++spins & 255u) == 0u && __builtin_amdgcn_s_memrealtime() - t_start > f->mbox_wait + 200000ull) {  // (workgroup 0 never came)
                verdict = 2u;
                break;
              }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) xs[q] = __uint_as_float(__hip_atomic_load(line + 1 + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
          }
          mbx[0] = xs[0];
          mbx[1] = xs[1];
          mbx[2] = xs[2];
          mbx[3] = xs[3];
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          t2_lds_st(sig + 3, verdict);
        }
      }
      while (t2_lds_ld(sig + 3) == 0u) __builtin_amdgcn_s_sleep(1);
      mb_cancel = t2_lds_ld(sig + 3) == 2u;
#pragma unroll
      for (int q = 0; q < DS; ++q) x0[q] = mbx[q];
      T2_TL(0, 123);
    }
    if (wave < 8 && mb_cancel) {
      // (cancelled: nothing to roll out; the launch ends at barrier B1)
    } else if (wave < 8) {
      // ================= R waves, phase 1: rollouts (tick_owner stage 2; lane = sample) =================
      DUST_PRIO(T2_PRIO_ROLL);
      const int rp = wave >> 1;
      const int n = n_first + rp;
      const float *tile_p = tile + (size_t)rp * S * Dp;
      float *cst_p = cst + rp * S, *omg_p = omg + rp * S;
      const float *th_p = th + rp * T2_ROW;
      const bool bad = flag_th[rp] != 0.f || flag_eps[(k & 1) * T2_PW + rp] != 0.f;
      const long SN = (long)S * N;
      const bool fast_trig = MODEL == DUST_MODEL_PENDULUM && !bad && fabsf(x0[1]) <= 3.0e38f &&
                             (fabsf(x0[0]) + f->dm.max_speed_pend * (float)f->dm.dt * (float)H < 5.0e4f);
      for (int s = (wave & 1) * 64 + lane; s < S; s += 128) {
        const float *act = tile_p + s * Dp;
        double acc_m = 0.0;
        for (int m = 0; m < M; ++m) {
          const long r = (long)m * SN + (long)s * N + n;
          const int pidx = f->dm.interleave ? (int)(r % M) : m;
          Coef cf;
          cf.c0 = coefs[2 * pidx];
          cf.c1 = coefs[2 * pidx + 1];
          float x[DS];
#pragma unroll
          for (int q = 0; q < DS; ++q) x[q] = x0[q];
          CostSum<MODEL> tot;  // (Pendulum: fp32 within groups of four steps - common.hpp)
          float traj;
          if (fast_trig && fabsf(cf.c0) <= 3.0e38f && fabsf(cf.c1) <= 3.0e38f) {
            const float dt = (float)f->dm.dt, mt = f->dm.max_torque, ms = f->dm.max_speed_pend;
            float chol0 = f->chol_a[0];
            asm volatile("" : "+v"(chol0));  // (a vector register: a VOP2 with a scalar operand issues in 4.3 cycles instead of 2.6)
            const v2f W = {f->dm.w_cos, f->dm.w_vel};
            float sn, cs;
            const TrigConsts K = trig_consts_pinned();
#pragma unroll 4
            for (int t = 0; t < H; ++t) {
              pendulum_trig(x[0], &sn, &cs, K);
              v2f q = {cs - 1.0f, x[1]};
              q = W * (q * q);
              tot.add(q.x + q.y, t);
              const float uu = __builtin_amdgcn_fmed3f(th_p[t] + chol0 * act[t], -mt, mt);
              float thd = x[1] + dt * (cf.c0 * sn + cf.c1 * uu);
              thd = __builtin_amdgcn_fmed3f(thd, -ms, ms);
              x[0] = x[0] + thd * dt;
              x[1] = thd;
            }
            pendulum_trig(x[0], &sn, &cs, K);
            v2f q = {cs - 1.0f, x[1]};
            q = W * (q * q);
            traj = (float)tot.total() + (q.x + q.y);
          } else {
            for (int t = 0; t < H; ++t) {
              float at[DA];
#pragma unroll
              for (int q = 0; q < DA; ++q) at[q] = th_p[t * DA + q] + f->chol_a[q] * act[t * DA + q];
              const float ci = t2_step_with_cost<MODEL>(f, lds + L.grid, cf, x, at);
              tot.add(ci, t);
            }
            traj = (float)tot.total() + t2_term_cost<MODEL>(f, lds + L.grid, x);
          }
          acc_m += (double)traj;
        }
        const float cost = (M == 1) ? (float)acc_m : (float)(acc_m / M);
        cst_p[s] = cost;
      }
      T2_TL(0, 16 * k + 1);
      // (the costs are an output of the tick's last iteration only, rewritten by the replay of a tick that does not start: they may leave
      //  the workgroup before "go" is known)
      // Stage outputs (costs, log-likelihood, score halves, phi) are what the getters return for the LAST iteration: only that one
      // stores them - the stores of the other iterations sat in front of the barriers' vmcnt waits (phi: right in front of B6)
      const bool last_iter = k + 1 == f->n_iters;
      bool store_costs = last_iter;
      if (last_iter && k == 0) {
        // a ONE-iteration tick reaches this store before barrier B1, i.e. possibly before workgroup 0 has decided whether the launch
        // starts (an aborted launch with caller-supplied noise is not replayed and must leave dust_get_costs' record alone):
        // wait for the go word (published ~2 us into the launch; the rollouts above took longer) and skip the store on "abort"
        unsigned int g = 0u;
        if (lane == 0) {
          g = t2_lds_ld(sig);
          unsigned int spins = 0u;
          while (g == 0u) {
            g = __hip_atomic_load(go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (g == 0u) g = t2_lds_ld(sig);
            if (g == 0u) {
              __builtin_amdgcn_s_sleep(2);
              if ((++spins & 0xffffu) == 0u && __hip_atomic_load(tflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) g = 2u;  // (a waiter gave up)
            }
          }
        }
        g = (unsigned int)__builtin_amdgcn_readfirstlane((int)g);
        store_costs = g == 1u;
      }
      if (store_costs)
        for (int s = (wave & 1) * 64 + lane; s < S; s += 128) f->costsT[(size_t)n * S + s] = cst_p[s];
      // wave-local softmax pieces over the wave's own samples, and the wave's share of the weighted sums over them (likelihoods.py:127-135,
      // svmpc.py:50-53, disco.py:387-392): nothing here needs the partner wave or the prior pass, so it runs underneath the P waves' pass
      // instead of behind barrier B1; the owner lanes merge the two waves' pieces of a particle (scale factors exp(-alpha (m_w - min)))
      float m_w = INFINITY, cs_w = 0.f;
      for (int s = (wave & 1) * 64 + lane; s < S; s += 128) {
        m_w = fminf(m_w, cst_p[s]);
        cs_w += cst_p[s];
      }
      m_w = wave_min(m_w);
      cs_w = wave_sum(cs_w);
      float zw = 0.f, zo = 0.f;
      for (int s = (wave & 1) * 64 + lane; s < S; s += 128) {
        const float cc = cst_p[s];
        const float ew = expf(-cc * f->alpha - (-m_w * f->alpha));
        cst_p[s] = ew;
        zw += ew;
        if (!same_w) {
          const float eo = expf((-1.0f * (cc - m_w)) / f->temp);
          omg_p[s] = eo;
          zo += eo;
        }
      }
      zw = wave_sum(zw);
      zo = wave_sum(zo);
      if (lane == 0) {
        float *rw = red_w + (rp * 2 + (wave & 1)) * 4;
        rw[0] = m_w;
        rw[1] = zw;
        rw[2] = zo;
        rw[3] = cs_w;
      }
      DUST_PRIO(0);
      {  // lane = (sample parity q, column j): 32 of the wave's 64 samples each (its own LDS writes: in order, no barrier)
        const int q = lane >> 5, j = lane & 31;
        float g = 0.f, am = 0.f;
        if (j < D) {
          const float thj = th_p[j], lj = pick_da<DA>(f->chol_a, j);
          const float is2 = 1.0f / (pick_da<DA>(f->sigma_a, j) * pick_da<DA>(f->sigma_a, j));
          const float base = f->eps_base_mode ? thj : f->a_seq[j];
          const float *tp = tile_p + j;
          const float *op_ = same_w ? cst_p : omg_p;
#ifndef T2_WSUM_PER_SAMPLE
          // The sums run over the RAW draws and the per-column factors are applied once: sum_s w_s (a_s - theta) / sigma^2 =
          // (L / sigma^2) sum_s w_s eps_s and sum_s omega_s (a_s - base) = (theta - base) sum_s omega_s + L sum_s omega_s eps_s, up to the
          // rounding of a_s = fl(theta + L eps_s) that the per-sample form (below, -DT2_WSUM_PER_SAMPLE) carries through (<= 1e-6 relative per
          // term; all parity tests unchanged): 2 instead of 6 vector instructions per sample - 94.4 -> 91.2 us per cfg2 tick (round 5)
          {
            float ge = 0.f, oe = 0.f, so = 0.f;
            for (int s0 = (wave & 1) * 64; s0 < S; s0 += 128) {
              const int s1 = min(s0 + 64, S);
#pragma unroll 8
              for (int s = s0 + q; s < s1; s += 2) {
                const float e = tp[s * Dp], c = cst_p[s];
                ge = fmaf(c, e, ge);
                if (!same_w) {
                  oe = fmaf(op_[s], e, oe);
                  so += op_[s];
                } else {
                  so += c;
                }
              }
            }
            if (same_w) oe = ge;
            g = ge * (lj * is2);
            am = fmaf(thj - base, so, lj * oe);
          }
#else
          {
            for (int s0 = (wave & 1) * 64; s0 < S; s0 += 128) {
              const int s1 = min(s0 + 64, S);
#pragma unroll 4
              for (int s = s0 + q; s < s1; s += 2) {
                const float av = thj + lj * tp[s * Dp];
                g = fmaf(cst_p[s], (av - thj) * is2, g);
                am = fmaf(op_[s], av - base, am);
              }
            }
          }
#endif
        }
        wpart[((rp * 2 + (wave & 1)) * 2 + q) * T2_ROW + j] = g;
        wpart[(T2_PW * 4 + (rp * 2 + (wave & 1)) * 2 + q) * T2_ROW + j] = am;
      }
      T2_TL(0, 16 * k + 5);
    } else {
      // ================= P waves, phase 1: the theta-only half of SVMPC.phi against all N keys =================
      const int pw = wave - 8;
      if (wave == 15) {  // theta generation k: published at the start of the tick (k = 0) or by iteration k - 1
        if (k == 0) gen0_wait();
        else t2_poll(cnt_theta + rep_off, 2u, (unsigned int)(k + 1), G, lane, tflag);
        if (lane == 0) t2_lds_st(sig + 1, (unsigned int)(k + 1));
      }
      while (t2_lds_ld(sig + 1) < (unsigned int)(k + 1)) __builtin_amdgcn_s_sleep(1);
      T2_TL(8, 16 * k + 2);
      float rw[2];
      DUST_PRIO(T2_PRIO_PPASS);
      if (N != f->steps * 64) t2_prior_pass_w<MODE, true>(f, k, th, ksl, lml, pw, lane, lm_ref, rw);
      else t2_prior_pass_w<MODE, false>(f, k, th, ksl, lml, pw, lane, lm_ref, rw);
      ppart[(pw * 2 + ((lane >> 2) & 1)) * 32 + 16 * (lane & 1) + t2_wide_col(lane)] = rw[0];  // [unit][query][column]
      if ((lane & ~2) == 0) ppart[512 + pw * 2 + (lane >> 1)] = rw[1];                            // [unit][query] L
      DUST_PRIO(0);
      T2_TL(8, 16 * k + 3);
      T2_TL(15, 16 * k + 15);
      if (k == 0 && wave == 15) go_wait();  // (the pass above read and wrote nothing outside the workgroup)
    }
    wg_sync();  // B1
    f = t2_args();
    T2_TL(0, 16 * k + 4);
    if (sig[0] == 2u || (k == 0 && sig[3] == 2u)) {  // (uniform: no workgroup wrote anything)
      if (b == 0 && tid0 == 0) {
        // an armed launch whose state never came (cancelled by the host, or the bound ran out) counts as one that did not start
        if (sig[0] != 2u) __hip_atomic_fetch_add(f->status + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (f->host_done) __hip_atomic_store(f->host_done, f->launch_seq | 0x80000000u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      return;
    }
    // ================= phase 4: score rows out (waves 8-9) | score arrivals (wave 10) | next noise (R) =================
    // The owner lanes finish both halves of the score themselves (no barrier between the prior pass and the published row): grad_pri from
    // the pass partials of the four waves that hold the query, grad_lik from the two R waves' pieces of the particle's sample softmax.
    float gs_keep = 0.f, gp_keep = 0.f;
    if (isown) {
      DUST_PRIO(T2_PRIO_OWNERSTAGE);  // the score rows are what every other workgroup waits for: ahead of the Stein pass and the noise draw
      const float *r0 = red_w + (op * 2) * 4, *r1 = r0 + 4;
      const float m0 = r0[0], m1 = r1[0], cmin = fminf(m0, m1);
      const float f0 = expf(-m0 * f->alpha - (-cmin * f->alpha)), f1 = expf(-m1 * f->alpha - (-cmin * f->alpha));
      const float zw = r0[1] * f0 + r1[1] * f1;
      float zo = zw, g0 = f0, g1 = f1;
      if (!same_w) {
        g0 = expf((-1.0f * (m0 - cmin)) / f->temp);
        g1 = expf((-1.0f * (m1 - cmin)) / f->temp);
        zo = r0[2] * g0 + r1[2] * g1;
      }
      if (od == 0) {
        const int n = n_first + op;
        float last_logl;
        if (f->lik == DUST_LIK_EXP_UTILITY) last_logl = ((-cmin * f->alpha) + logf(zw)) - logf((float)S);
        else last_logl = -f->alpha * ((r0[3] + r1[3]) / (float)S);
        ll[op] = last_logl;
        if (k + 1 == f->n_iters) {
          f->logl[n] = last_logl;
          f->eta[n] = (-cmin / f->temp) + logf(zo);
        }
      }
      const float *wg = wpart + (op * 4) * T2_ROW + od, *wa = wpart + (T2_PW * 4 + op * 4) * T2_ROW + od;
      const float g = (wg[0] + wg[T2_ROW]) * f0 + (wg[2 * T2_ROW] + wg[3 * T2_ROW]) * f1;
      const float am = (wa[0] + wa[T2_ROW]) * g0 + (wa[2 * T2_ROW] + wa[3 * T2_ROW]) * g1;
      const float gs = g / zw;
      const float as = am / zo;
      if (f->update_a_mat) amv = amv + as;
      {
        float sp = 0.f, l = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {  // the four waves that hold this query's half: their key quarters, in order
          const int pwq = (op >> 1) * 4 + w;
          sp += ppart[(pwq * 2 + (op & 1)) * 32 + od];
          l += ppart[512 + pwq * 2 + (op & 1)];
        }
        gp_keep = (sp / l) * f->inv_sp2;
      }
      gs_keep = gs;
      scl[op * T2_ROW + od] = ownv ? gs + gp_keep : 0.f;
      publish_rows(scl, f->sq + (size_t)k * N * T2_ROW, cnt_score);
      DUST_PRIO(0);
      T2_TL(8, 16 * k + 7);
      if (ownv && k + 1 == f->n_iters) {
        f->score[no] = gs_keep + gp_keep;
        f->grad_lik[no] = gs_keep;
        f->grad_pri[no] = gp_keep;
      }
    }
    if (wave >= 8) {  // the Stein half of the theta-only work, while the score rows travel
      // (tried: the owner waves' two units - they start 1.7 us late and finish last - cut in four and handed to the rollout waves behind
      //  their noise draw, with the owner waves idle: +4.5 us per tick, the workgroups drift apart; an idle wave polling the score
      //  counters from the moment the rows are published: the hop takes 5.8 instead of 2.5 us - the polls sit on the arrivals' lines)
      float rb[4];
      if (N != f->steps * 64) t2_pair_pass<MODE, T2_PASS_STEIN, true>(f, k, th, ksl, lml, wave - 8, lane, lm_ref, rb);
      else t2_pair_pass<MODE, T2_PASS_STEIN, false>(f, k, th, ksl, lml, wave - 8, lane, lm_ref, rb);
      rpart[((wave - 8) * 16 + reduce_u16_index<16>(0, lane)) * 4 + (lane & 3)] = rb[0];
      T2_TL(8, 16 * k + 11);
      T2_TL(15, 16 * k + 6);
    }
    if (wave == 10) {
      t2_poll(cnt_score + rep_off, 2u, (unsigned int)(k + 1), G, lane, tflag);
      T2_TL(10, 16 * k + 8);
    } else if (wave < 8) {
      if (k + 1 < f->n_iters) {
        if (f->eps == nullptr) draw_own_rows(f, k + 1, wave, lane, 0);  // (the second half: phase 6)
        else draw_noise(f, k + 1, tid, 512);
      }
      T2_TL(0, 16 * k + 9);
    } else if (wave == 11) {
      if (k + 1 < f->n_iters) make_coefs(f, k + 1);
    }
    wg_sync();  // B4
    f = t2_args();
    T2_TL(0, 16 * k + 10);
    // ================= phase 5: sum_j k_ij s_j (all 16 waves stream the score rows) =================
    {  // (the R waves have drawn their noise in phase 4 and would idle here: 16 waves instead of the 8 P waves, -1.1 us per tick)
      const int pw = wave, u = lane >> 2, c = lane & 3;
      const __amdgpu_buffer_rsrc_t rsq = t2_rsrc(f->sq + (size_t)k * N * T2_ROW, N * T2_ROW);
      v2f acc[T2_PW][4];
#pragma unroll
      for (int q = 0; q < T2_PW; ++q)
#pragma unroll
        for (int h = 0; h < 4; ++h) acc[q][h] = v2f{0.f, 0.f};
      constexpr int NB = 4;
      for (int t0 = 0; t0 < f->steps / 4; t0 += NB) {
        v4f sa[NB], sb[NB];
#pragma unroll
        for (int p = 0; p < NB; ++p) {
          const int j = min(((t0 + p) * 16 + pw) * 16 + u, N - 1);
          sa[p] = t2_ld16(rsq, (j * T2_ROW + 8 * c) * 4);
          sb[p] = t2_ld16(rsq, (j * T2_ROW + 8 * c + 4) * 4);
        }
#pragma unroll
        for (int p = 0; p < NB; ++p) {
          const int j = ((t0 + p) * 16 + pw) * 16 + u;
          const float4 kq = *reinterpret_cast<const float4 *>(&ksl[(size_t)j * 4]);  // (keys past N hold k = 0)
          const v2f sv[4] = {{sa[p][0], sa[p][1]}, {sa[p][2], sa[p][3]}, {sb[p][0], sb[p][1]}, {sb[p][2], sb[p][3]}};
          const float kk[T2_PW] = {kq.x, kq.y, kq.z, kq.w};
#pragma unroll
          for (int q = 0; q < T2_PW; ++q) {
            const v2f kv = {kk[q], kk[q]};
#pragma unroll
            for (int h = 0; h < 4; ++h) acc[q][h] = __builtin_elementwise_fma(kv, sv[h], acc[q][h]);
          }
        }
      }
      float v[32], r2[2];
#pragma unroll
      for (int q = 0; q < T2_PW; ++q)
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          v[q * 8 + 2 * h] = acc[q][h].x;
          v[q * 8 + 2 * h + 1] = acc[q][h].y;
        }
      reduce_u16<32>(v, r2, lane);
      float *kp16 = lds + T2_L_WPART;  // (the weighted-sum partials are dead by now: [16 waves][32][4] fits their 2 x 4 x 8 x 32 floats)
#pragma unroll
      for (int i = 0; i < 2; ++i) kp16[(pw * 32 + reduce_u16_index<32>(i, lane)) * 4 + c] = r2[i];
    }
    wg_sync();  // B5
    f = t2_args();
    T2_TL(0, 16 * k + 12);
    // ================= phase 6: phi, optimiser step, theta rows out (waves 8-9) =================
    float phi_keep = 0.f;
    if (isown) {
      float sa = 0.f;
      {
        const float *kp16 = lds + T2_L_WPART;
#pragma unroll
        for (int w = 0; w < 16; ++w) sa += kp16[(w * 32 + op * 8 + (od & 7)) * 4 + (od >> 3)];
      }

      float sb = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) sb += rpart[(((op >> 1) * 4 + w) * 16 + (op & 1) * 8 + (od & 7)) * 4 + (od >> 3)];
      const float phi = sb * f->inv_l2 + sa * f->inv_n;
      const float gr = -phi;
      if (!adam) thv = fmaf(-f->lr, gr, thv);
      else thv = adam_step(thv, gr, adm, adv, f->lr, f->beta1, f->beta2, f->adam_eps, (float)(adam0 + (uint32_t)k + 1u));
      if (!ownv) thv = 0.f;
      th[op * T2_ROW + od] = thv;
      const unsigned long long badm = __ballot(ownv && !(fabsf(thv) <= 3.0e38f));
      if ((lane & 31) == 0) {
        flag_th[op] = ((badm >> (lane & 32)) & 0xffffffffull) ? 1.f : 0.f;
        flag_eps[(k & 1) * T2_PW + op] = 0.f;  // read by this iteration's rollouts; iteration k + 2 stages its noise behind barrier B6
      }
      phi_keep = phi;
    }
    if (isown) {
      if (k + 1 < f->n_iters || f->do_forward) publish_rows(th, f->xq + (size_t)(k + 1) * N * T2_ROW, cnt_theta);
      T2_TL(8, 16 * k + 13);
      if (ownv && k + 1 == f->n_iters) f->phi[no] = phi_keep;
    } else if (T2_NOISE_SPLIT && wave < 8 && k + 1 < f->n_iters && f->eps == nullptr) {
      // the second half of the next iteration's noise, by the rollout waves, while the owner lanes update and publish: in phase 4 the whole
      // draw competed with the Stein pass for issue slots, here nothing else runs
      draw_own_rows(f, k + 1, wave, lane, 1);
    }
    wg_sync();  // B6  (tried twice: the barrier in front of the publish, so that the rollout waves start ~1 us earlier - +1.3 us per tick; with the
                // publish at wave priority 3: no gain either)
    T2_TL(0, 16 * k + 14);
  }

  const int kf = f->n_iters;
  const int lane = tid0 & 63;
  // COMMIT (t2_commit above).  Everything persistent - particles, a_mat, optimiser moments, prior mixture, stream counters - is written
  // only from here on, behind the tick's last wait, and only if no wait of the launch has given up: a workgroup that has passed the last
  // wait knows that every workgroup published its last hand-off, i.e. passed every earlier wait.  A tick whose waits timed out (another
  // process's compute on the device: handoff.hpp) therefore leaves the state untouched and the host replays it on plain kernels.
  if (!f->do_forward) {  // SVMPC.optimize alone: particles and optimiser state stay (svmpc.py:97-126)
    if (tid0 == 0) t2_lds_st(sig + 2, t2_commit(f->status, b) ? 1u : 0u);
    wg_sync();
    if (sig[2] == 0u) return;
    if (ownv) {
      f->theta[no] = thv;
      if (f->update_a_mat) f->a_mat[no] = amv;
      if (adam) {
        f->adam_m[no] = adm;
        f->adam_v[no] = adv;
      }
    }
    if (b == 0 && tid0 == 0) {
      f->ctr[1] = ctr_iter0 + (uint32_t)kf;
      f->ctr[2] = adam0 + (uint32_t)kf;
    }
    return;
  }

  // ================= SVMPC.forward (svmpc.py:172-200), fast_pred: the last iteration's costs =================
  if (kf == 0) {
    if (wave == 15) start_protocol();
    while (t2_lds_ld(sig) == 0u) __builtin_amdgcn_s_sleep(1);
    if (sig[0] == 2u) {
      if (b == 0 && tid0 == 0 && f->host_done) __hip_atomic_store(f->host_done, f->launch_seq | 0x80000000u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
  }
  {  // log p(theta_n) under the tick's prior, whose means ARE the particles (svmpc.py:137; svgd.py:87): all 16 waves
    if (wave == 15) {
      t2_poll(cnt_theta + rep_off, 2u, (unsigned int)(kf + 1), G, lane, tflag);
      if (lane == 0) t2_lds_st(sig + 1, (unsigned int)(kf + 1));
    }
    while (t2_lds_ld(sig + 1) < (unsigned int)(kf + 1)) __builtin_amdgcn_s_sleep(1);
    float red[4];
#ifdef T2_LOGP_NARROW
    t2_pair_pass<MODE, T2_PASS_LOGP, true>(f, kf, th, ksl, lml, wave, lane, lm_ref, red);
#else
    t2_logp_pass_w<true>(f, kf, th, lml, wave, lane, lm_ref, red);
#endif
    if (lane < T2_PW) {
      float s = red[0];
      s = lane == 1 ? red[1] : s;
      s = lane == 2 ? red[2] : s;
      s = lane == 3 ? red[3] : s;
      ppart[wave * 4 + lane] = s;
    }
  }
  wg_sync();
  T2_TL(0, 16 * kf + 0);
  if (wave == 8) {
    if (lane < T2_PW) {
      float l = 0.f;
#pragma unroll
      for (int w = 0; w < 16; ++w) l += ppart[w * 4 + lane];
      const float lp = (lm_ref + logf(l)) + f->log_norm;
      const int n = n_first + lane;
      f->logp[n] = lp;
      const float lwv = ll[lane] + lp;
      f->lw[n] = lwv;
      st_sc1(f->lwq + n, lwv);
    }
    if (f->test_abort == 2 && b == G - 1 && lane == 0) {  // test hook: "a wait of this workgroup gave up"
      __hip_atomic_store(tflag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    t2_arrive_wave(cnt_lw + (size_t)sh * T2_CNT_STRIDE, lane);
    t2_poll(cnt_lw + rep_off, 1u, 1u, G, lane, tflag);
    if (lane == 0) t2_lds_st(sig + 2, t2_commit(f->status, b) ? 1u : 0u);
  }
  wg_sync();
  T2_TL(0, 16 * kf + 1);
  if (sig[2] == 0u) return;  // no COMMIT (see above)
  // softmax over all particles, first-index argmax (finalize_body), computed by every workgroup for itself
  bool reports = false;  // closed-loop serving: this workgroup owns the best particle and writes the outputs to pinned host memory
  {
    const int t = tid0;
    const float lwr = t < N ? ld_sc1(f->lwq + t) : -INFINITY;
    // per wave (max, sum exp) pieces, merged by every wave for itself: one barrier for the softmax
    const float m_w = wave_max(lwr);
    const float z_w = wave_sum(t < N ? expf(lwr - m_w) : 0.f);
    if (lane == 0) {
      wred[wave] = m_w;
      wred[16 + wave] = z_w;
    }
    wg_sync();
    const float mm = lane < T2_NT / 64 ? wred[lane] : -INFINITY;
    const float m = wave_max(mm);
    const float z = wave_sum((lane < T2_NT / 64 && mm != -INFINITY) ? wred[16 + lane] * expf(mm - m) : 0.f);
    const float lz = m + logf(z);
    const float p = t < N ? expf(lwr - lz) : 0.f;
    if (t >= n_first && t < n_first + T2_PW) f->pw[t] = p;
    // first-index argmax of p: per wave the largest value and its first lane, then the first wave that holds the largest
    const float pb_w = wave_max(t < N ? p : -INFINITY);
    const unsigned long long mk = __ballot(t < N && p == pb_w);
    int *redi = reinterpret_cast<int *>(wred + 48);
    if (lane == 0) {
      wred[32 + wave] = pb_w;
      redi[wave] = mk ? wave * 64 + (__ffsll((long long)mk) - 1) : 0x7fffffff;
    }
    wg_sync();
    const float pv = lane < T2_NT / 64 ? wred[32 + lane] : -INFINITY;
    const float best = wave_max(pv);
    const unsigned long long mk2 = __ballot(lane < T2_NT / 64 && pv == best);
    const int bi = redi[mk2 ? (__ffsll((long long)mk2) - 1) : 0];
    if (bi >= n_first && bi < n_first + T2_PW) {  // the owner of the best particle hands out its action sequence
      if (t == 0) *f->istar = bi;
      if (t < D) f->a_seq_out[t] = th[(bi - n_first) * T2_ROW + t];
      if (f->host_out) {
        // closed-loop serving: THIS workgroup - the owner of the best particle - also writes the outputs straight into pinned host
        // memory: the chosen sequence and, when the caller wants them, all N particle weights (every workgroup computed the whole
        // softmax: lane t holds p_t).  The stores are posted here and acknowledged ~3 us later; the workgroup goes on with the mixture
        // and the roll meanwhile and tells the host at its very end (below).  No other workgroup touches host memory.
        if (t < D) f->host_out[t] = th[(bi - n_first) * T2_ROW + t];
        if (f->host_pw && t < N) f->host_out[(f->pw - f->a_seq_out) + t] = p;  // (the pinned mirror has outblk's layout: a_seq | p_weights)
        reports = true;  // (uniform over the workgroup: every wave found the same best particle)
      }
    }
    // new prior mixture (finalize_body): Categorical(probs) clamps, then log_softmax
    if (!f->weighted_prior) {
      const float l = logf(fminf(fmaxf(1.0f / (float)N, 1.1920929e-07f), 1.0f - 1.1920929e-07f));
      const float lzz = l + logf((float)N);
      if (t < T2_PW) {
        f->mixw[n_first + t] = 1.0f;
        f->logmix[n_first + t] = l - lzz;
      }
    } else {
      const float psum = block_reduce<RED_SUM>(p, wred);
      float pc = p / psum;
      pc = fminf(fmaxf(pc, 1.1920929e-07f), 1.0f - 1.1920929e-07f);
      const float l = t < N ? logf(pc) : -INFINITY;
      const float lm = block_reduce<RED_MAX>(l, wred);
      const float zs = block_reduce<RED_SUM>(t < N ? expf(l - lm) : 0.f, wred);
      const float lzz = lm + logf(zs);
      if (t >= n_first && t < n_first + T2_PW) {
        f->mixw[t] = p;
        f->logmix[t] = l - lzz;
      }
    }
  }
  // roll (svmpc.py:142-158): shift left along H, last row per strategy.  Every reader of theta in this launch has finished: the
  // log-weights of ALL particles needed every workgroup's log-density pass.
  if (isown) {
    float outv = 0.f;
    if (f->roll_strategy == DUST_ROLL_MEAN) {  // each policy's mean over H, per control dimension (roll_kernel's order: a wave sum)
      for (int c = 0; c < DA; ++c) {
        float v = (ownv && od % DA == c) ? thv : 0.f;
        // sum over the 32 lanes of this particle
        v += __shfl_xor(v, 16, 64);
        v = row16_reduce(v, 0.f, [](float x, float y) { return x + y; });
        if (ownv && od + DA >= D && od % DA == c) outv = v / (float)H;
      }
    }
    if (ownv) {
      float out = (od + DA < D) ? th[op * T2_ROW + od + DA] : thv;
      if (f->roll_strategy == DUST_ROLL_MEAN && od + DA >= D) out = outv;
      f->theta[no] = out;
      if (f->update_a_mat) f->a_mat[no] = amv;
      if (adam) {  // SVMPC.roll makes a NEW parameter tensor: torch's optimiser state restarts
        f->adam_m[no] = 0.f;
        f->adam_v[no] = 0.f;
      }
    }
  }
  if (b == 0 && tid0 == 0) {
    f->ctr[0] = ctr_tick + 1u;
    f->ctr[1] = 0u;
    f->ctr[2] = 0u;
  }
  if (reports) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores to host memory are acknowledged
    wg_sync();
    if (tid0 == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
      __hip_atomic_store(f->host_done, f->launch_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  T2_TL(0, 16 * kf + 2);
#endif  // __HIP_DEVICE_COMPILE__
}

}  // namespace dust
