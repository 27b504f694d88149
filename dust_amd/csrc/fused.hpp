// fused.hpp - one launch for the two halves of the score that only READ theta: the prior pass (pairwise_body<PRIOR>)
// and the rollout / likelihood-score kernel (rollout_body).
//
// Why: at cfg2 sizes (N = 1024, S = 128) the rollout kernel is 2 waves per SIMD and instruction-issue / latency bound
// (~25 % of the issue slots), and the prior pass is another 2 waves per SIMD of the same kind; run back to back they cost
// 20.6 + 12.2 us.  Two streams (eager or inside a hipGraph) were SLOWER than back-to-back on this stack (2 970 vs
// 3 660 ticks/s: cross-queue dependencies cost more than the overlap gains), so the overlap is done inside one launch:
// workgroups [0, n_pair) run the prior tiles, workgroups [n_pair, ...) run the rollouts (256 / nt particles each).
//
// Hand-off inside the launch (the rollout role needs the prior partials of its particle only at its very end): per
// query tile a counter in HBM, zeroed by the update kernel of the previous iteration.  Producer = cdna_hip_programming.md
// Guideline 16, write-through form: partials stored with sc1 (agent-scope relaxed atomic stores) -> every wave
// s_waitcnt vmcnt(0) -> barrier -> lane 0 relaxed agent atomic add.  Consumer (rollout_body): ONE lane polls relaxed with
// s_sleep, barrier, then EVERY load of the partials is an sc1 load.  No L2 write-back / L1 invalidate fences: the
// release+acquire fence form measured 30 % SLOWER than two separate launches (512 workgroups each flushing an XCD L2).  Results do not depend on placement or timing; the
// prior workgroups have the LOWER block indices and never wait on anything, so the rollout workgroups can only wait on
// work that has already been dispatched (and the spin is bounded).
#pragma once
#include "bandwidth.hpp"
#include "rollout.hpp"
#include "stein.hpp"

namespace dust {

struct FusedArgs {
  PairArgs pa;
  RolloutArgs ra;
  int tiles, n_pair_blocks;
  int sub_nt, per_block;     // rollout role: lanes per particle, particles per 256-lane workgroup
  int lds_roll_floats;       // LDS floats per particle sub-block
  unsigned int *cnt;         // [tiles][CNT_STRIDE] arrival counters, one per 128-byte line
  unsigned int *timeout_flag;
  // kernel branch K2 (round 6): the per-dimension median bandwidths read the particles only, like the two roles above - workgroups
  // [0, n_k2_blocks) of the launch (the lowest indices: latency-bound chains of ~10 us, dispatched first; a multiple of 8, so that the
  // other roles keep their XCD mapping; workgroup c < k2.D finds h_c, the rest return at once).  0: no such role.
  int n_k2_blocks;
  K2Args k2;
};

template <int MODEL, int CPT, bool GROUPS>
__global__ __launch_bounds__(PAIR_NT, (CPT <= 4 ? 4 : 2)) void fused_prior_rollout_kernel(const FusedArgs f) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if ((int)blockIdx.x < f.n_k2_blocks) {
    if ((int)blockIdx.x < f.k2.D) k2_bandwidth256(f.k2, (int)blockIdx.x, lds);  // (s_setprio(3) around it: no change, 186 us per cfg2 / K2 tick either way)
    return;
  }
  const int bx = (int)blockIdx.x - f.n_k2_blocks;
  if (bx < f.n_pair_blocks) {
    const int tile_x = bx % f.tiles, js = bx / f.tiles;
    pairwise_body<PAIR_PRIOR, CPT>(f.pa, lds, tile_x, js, /*write_through=*/true);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its sc1 stores ...
    wg_sync();                                    // ... before the one lane that signals for the workgroup
    if (threadIdx.x == 0) __hip_atomic_fetch_add(f.cnt + tile_x * CNT_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    const int nb = (int)gridDim.x - f.n_k2_blocks - f.n_pair_blocks;
    const int b = (f.n_pair_blocks & 7) ? bx - f.n_pair_blocks : xcd_contiguous(bx - f.n_pair_blocks, nb);
    const int sub = (int)threadIdx.x / f.sub_nt, tid = (int)threadIdx.x - sub * f.sub_nt;
    const FusedWait fw{f.cnt, (unsigned int)f.pa.JS, f.timeout_flag, nullptr, 1, 0, nullptr, nullptr};
    rollout_body<MODEL, 12, GROUPS, true>(f.ra, lds + (size_t)sub * f.lds_roll_floats, tid, f.sub_nt, b * f.per_block + sub, &fw);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Stein pass + optimiser update in one launch.  Workgroups [0, n_pair) run the Stein tiles (pairwise_body<K1|IMQ>) and
// publish their slice partials write-through; workgroups [n_pair, ...) are the update role: 256 consecutive (particle,
// dim) elements each.  theta is BOTH an input of every Stein tile (queries and keys) and the output of the update: updated
// IN PLACE, the update role may not start before EVERY Stein workgroup has finished reading (arrivals are counted per query
// tile, one 128-byte line each; the last arrival of a tile bumps one global line and the update role polls that line - a
// flat counter with tiles*JS arrivals on one line serialises the producers' atomics).  Unsharded contexts ping-pong theta
// instead (the update writes the other buffer), and the update role of a query tile starts as soon as that tile's JS
// slices have arrived.
// The update role has the HIGHER block indices: it only ever waits on work dispatched before it; the spin is bounded.
// The counters are re-armed by the next rollout launch (RolloutArgs::rearm), as fused_cnt is by the update role.
struct SteinUpdateArgs {
  PairArgs pa;
  UpdateArgs ua;
  int tiles, n_pair_blocks;
  int wait_all;       // 1: theta is updated in place, the update role waits for EVERY Stein tile; 0: ping-pong, per query tile
  unsigned int *cnt;  // [tiles + 1][CNT_STRIDE]: per-tile arrivals, then the global "tiles done" line
  unsigned int *timeout_flag;
};

template <int MODE, int CPT>
__global__ __launch_bounds__(PAIR_NT, (CPT <= 4 ? 4 : 2)) void stein_update_kernel(const SteinUpdateArgs f) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if ((int)blockIdx.x < f.n_pair_blocks) {
    const int tile_x = (int)blockIdx.x % f.tiles, js = (int)blockIdx.x / f.tiles;
    pairwise_body<MODE, CPT>(f.pa, lds, tile_x, js, /*write_through=*/true);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wg_sync();
    if (threadIdx.x == 0) {
      const unsigned int prev = __hip_atomic_fetch_add(f.cnt + tile_x * CNT_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (prev + 1u == (unsigned int)f.pa.JS)
        __hip_atomic_fetch_add(f.cnt + f.tiles * CNT_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else {
    const int b = (int)blockIdx.x - f.n_pair_blocks;
    const int idx = b * PAIR_NT + (int)threadIdx.x;
    if (threadIdx.x == 0) {
      unsigned int spins = 0;
      const int total = f.ua.n_local * f.ua.D;
      // in place: the global line (tiles arrivals); ping-pong: the (at most two) query tiles this block's elements belong to
      const int t0 = f.wait_all ? f.tiles : (min(b * PAIR_NT, total - 1) / f.ua.D) / PAIR_TI;
      const int t1 = f.wait_all ? f.tiles : (min(b * PAIR_NT + PAIR_NT - 1, total - 1) / f.ua.D) / PAIR_TI;
      const unsigned int target = f.wait_all ? (unsigned int)f.tiles : (unsigned int)f.pa.JS;
      for (int t = t0; t <= t1; ++t)
        while (__hip_atomic_load(f.cnt + t * CNT_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
          __builtin_amdgcn_s_sleep(8);
          if (++spins > (1u << 24)) {
            *f.timeout_flag = 1u;
            break;
          }
        }
    }
    wg_sync();
    update_body<true>(f.ua, idx);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// One launch per SVGD iteration (K1 / IMQ, D <= 64, N < 2048, single-chunk key slices, ping-ponged theta).  Roles by
// workgroup index, producers before consumers:
//   [0, P)            prior tiles              -> prior partials   (cnt_prior[tile], JS arrivals)
//   [P, P + R)        rollouts                 -> score rows       (cnt_score[key slice], `slice` arrivals)
//   [P + R, 2P + R)   Stein tiles              -> Stein partials   (cnt_stein[tile], JS arrivals)
//   [2P + R, ...)     optimiser update (256 elements each), writes the OTHER theta buffer
// Every input of the launch that is written inside it travels write-through (sc1 stores, sc1 loads, one arrival counter per
// 128-byte line; cdna_hip_programming.md Guideline 16).  What this buys over the two-launch form: the Stein tiles' theta-only
// work (Gram values, repulsive term) runs underneath the rollouts, and one launch boundary per iteration disappears.
// A workgroup only ever waits on LOWER-indexed workgroups; the hardware dispatches workgroups in index order per XCD, so the
// lowest-indexed unfinished workgroup is always resident and never blocked (the grid need not be co-resident); every spin
// is bounded.  Counters: two sets - the launch uses one and zeroes the other (which the previous launch used).
struct IterArgs {
  PairArgs prior, stein;
  RolloutArgs ra;
  UpdateArgs ua;
  int tiles, n_pair_blocks, n_roll_blocks;
  int sub_nt, per_block, lds_roll_floats;
  unsigned int *cnt_prior, *cnt_score, *cnt_stein;  // this launch's set
  unsigned int *zero_base;                          // the other set ...
  int zero_lines;                                   // ... of this many 128-byte lines
  unsigned int *timeout_flag;
  unsigned long long *tl;                           // diagnostic build only: [grid][4] launch timeline (common.hpp DUST_TL)
  // score rows handed over as data (rollout.hpp FusedWait::score_pub): this launch's buffer, and the other one, which the
  // prior tiles fill with the sentinel again for the next launch (its readers finished with the previous launch)
  float *score_pub;
  unsigned int *score_reset;
  int score_elems;
};

template <int MODEL, int MODE, int CPT>
__global__ __launch_bounds__(PAIR_NT, (CPT <= 4 ? 4 : 2)) void svgd_iter_kernel(const IterArgs f) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int b0 = (int)blockIdx.x;
  DUST_TL(f.tl, 0);
  if (b0 < f.n_pair_blocks) {
    if (b0 == 0)
      for (int t = threadIdx.x; t < f.zero_lines; t += PAIR_NT) f.zero_base[t * CNT_STRIDE] = 0u;
    if (f.score_reset)
      for (int e = b0 * PAIR_NT + (int)threadIdx.x; e < f.score_elems; e += f.n_pair_blocks * PAIR_NT) f.score_reset[e] = SCORE_SENTINEL;
    const int tile_x = b0 % f.tiles, js = b0 / f.tiles;
    pairwise_body<PAIR_PRIOR, CPT>(f.prior, lds, tile_x, js, /*write_through=*/true);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wg_sync();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(f.cnt_prior + tile_x * CNT_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    DUST_TL(f.tl, 3);
  } else if (b0 < f.n_pair_blocks + f.n_roll_blocks) {
    const int br = b0 - f.n_pair_blocks;
    const int b = ((f.n_pair_blocks | f.n_roll_blocks) & 7) ? br : xcd_contiguous(br, f.n_roll_blocks);
    const int sub = (int)threadIdx.x / f.sub_nt, tid = (int)threadIdx.x - sub * f.sub_nt;
    const FusedWait fw{f.cnt_prior, (unsigned int)f.prior.JS, f.timeout_flag, f.cnt_score, f.stein.slice, f.per_block, f.tl, f.score_pub};
    rollout_body<MODEL, 12, false, true>(f.ra, lds + (size_t)sub * f.lds_roll_floats, tid, f.sub_nt, b * f.per_block + sub, &fw);
    DUST_TL(f.tl, 3);
  } else if (b0 < 2 * f.n_pair_blocks + f.n_roll_blocks) {
    const int bs = b0 - f.n_pair_blocks - f.n_roll_blocks;
    const int tile_x = bs % f.tiles, js = bs / f.tiles;
    stein_split_body<MODE, CPT>(f.stein, lds, tile_x, js, f.cnt_score, f.score_pub, f.timeout_flag, f.tl);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wg_sync();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(f.cnt_stein + tile_x * CNT_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    DUST_TL(f.tl, 3);
  } else {
    const int b = b0 - 2 * f.n_pair_blocks - f.n_roll_blocks;
    const int idx = b * PAIR_NT + (int)threadIdx.x;
    if (threadIdx.x == 0) {
      unsigned int spins = 0;
      const int total = f.ua.n_local * f.ua.D;
      const int t0 = (min(b * PAIR_NT, total - 1) / f.ua.D) / PAIR_TI, t1 = (min(b * PAIR_NT + PAIR_NT - 1, total - 1) / f.ua.D) / PAIR_TI;
      for (int t = t0; t <= t1; ++t)
        while (__hip_atomic_load(f.cnt_stein + t * CNT_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned int)f.stein.JS) {
          __builtin_amdgcn_s_sleep(8);
          if (++spins > (1u << 24)) {
            *f.timeout_flag = 1u;
            break;
          }
        }
    }
    DUST_TL(f.tl, 1);
    wg_sync();
    update_body<true>(f.ua, idx);
    DUST_TL(f.tl, 3);
  }
}

}  // namespace dust
