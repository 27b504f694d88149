// tick2_args.hpp - argument block and geometry of the owner-computes persistent tick (tick2.hpp), shared by the two translation
// units of libdust_amd.so (dust_amd.hip: eligibility, buffers, launch bookkeeping; tick2.hip: the kernel and its launcher).
#pragma once
#include "common.hpp"

namespace dust {

enum {
  T2_PW = 4,      // Stein particles per workgroup (4 x 128 action samples = 8 rollout waves)
  T2_NT = 1024,   // lanes per workgroup: waves 0-7 roll out, waves 8-15 run the pairwise passes
  T2_NSH = 16,    // shards of every arrival counter (workgroup b signals shard b % T2_NSH; one 128-byte line each)
  T2_ROW = 32,    // floats per row of the exchange buffers (D <= 32 padded to one 128-byte line)
  T2_CNT_STRIDE = 32,
  T2_SETS = 4 * T2_NSH + 1  // lines of one counter set: start | theta | score | lw shards, then the go word
};

struct Tick2Args {
  DevModel dm;
  int N, S, M, H, D;
  int n_iters, do_forward;
  int steps;             // ceil(N / 64): key steps of one pair wave (8 keys per step and wave, 8 waves)
  int lik, update_a_mat, eps_base_mode, optimizer, roll_strategy, weighted_prior;
  int coef_given;
  int grid_words;        // Particle: words of the bit-packed occupancy grid staged in LDS (multiple of 4) or 0
  float coef_host[2];
  float alpha, temp;
  float chol_a[4], sigma_a[4];
  float inv_sp2;         // 1 / sigma_p^2 (isotropic prior scale: host check)
  float cP;              // -0.5 log2(e) / sigma_p^2: exponent scale of the prior weights on the raw squared distance
  float cS;              // K1: -0.5 log2(e) / ell^2 (exponent scale); IMQ: 1 / ell^2
  float inv_l2, inv_n, log_norm;
  float lr, beta1, beta2, adam_eps;
  float x0[4];           // plant state (by value)
  uint64_t seed;
  uint32_t *ctr;         // device counters {tick, iter, adam_step}
  const float *eps;      // device [n_iters][S][N][D] standard normals or nullptr: Philox stream in registers
  size_t eps_stride;
  const float *params;   // device [n_iters][M][P] dynamics samples or nullptr
  const float *a_seq;    // [D]
  float *theta;          // [N][D] particles: read at the start of the tick, rolled (forward) / updated (optimize) at its end
  float *xq, *sq, *lwq;  // exchange buffers written through inside the launch: particles [N][32], score rows [N][32], log-weights [N]
  float *logmix, *mixw;  // [N] prior mixture (the means alias the particles: host check)
  float *a_mat, *adam_m, *adam_v;
  float *costsT, *grad_lik, *grad_pri, *score, *phi, *logl, *eta, *logp, *lw, *pw, *a_seq_out;
  int *istar;
  unsigned int *cnt;        // this tick's counter set (T2_SETS lines)
  unsigned int *zero_base;  // the other set, zeroed by workgroup 0 for the next tick
  unsigned int *status;     // [0] a hand-off wait timed out (sticky until the host clears it) [1] ticks that did not start (not all workgroups resident)
  unsigned long long *tl;   // diagnostic build only: [grid][128] wall-clock stamps
};

// LDS bytes of one workgroup (host and device agree through these helpers)
struct Tick2Lds {
  int tile, cst, omg, th, misc, coefs, grid, ksl, ppart, gp, rp, wpart, kpart, scl, total;  // float offsets
};
__host__ __device__ inline Tick2Lds tick2_lds(int S, int D, int M, int steps, int grid_words) {
  Tick2Lds l;
  const int Dp = D | 1;
  auto up4 = [](int x) { return (x + 3) & ~3; };
  int o = 0;
  l.tile = o;  o = up4(o + T2_PW * S * Dp);
  l.cst = o;   o = up4(o + T2_PW * S);
  l.omg = o;   o = up4(o + T2_PW * S);
  l.th = o;    o += T2_PW * T2_ROW;
  l.misc = o;  o += 192;
  l.coefs = o; o = up4(o + 2 * M);
  l.grid = o;  o += grid_words;
  l.ksl = o;   o += steps * 64 * 4;        // Stein kernel values k_ij of the workgroup's 4 queries: [key][4]
  l.ppart = o; o += 8 * 40 * 8;            // per pair wave: 40 reduced sums x 8 column groups
  l.gp = o;    o += T2_PW * T2_ROW;
  l.rp = o;    o += T2_PW * T2_ROW;
  l.wpart = o; o += 2 * T2_PW * 8 * T2_ROW;  // weighted-sum partials (likelihood score, a_mat update)
  l.kpart = o; o += 8 * 16 * 8;
  l.scl = o;   o += T2_PW * T2_ROW;
  l.total = o;
  return l;
}

// tick2.hip
int tick2_occupancy(int model, int mode, size_t lds_bytes, int *blocks_per_cu);
int tick2_launch(const Tick2Args &f, int model, int mode, int grid, size_t lds_bytes, hipStream_t stream);

}  // namespace dust
