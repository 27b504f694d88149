// tick2_args.hpp - argument block and geometry of the owner-computes persistent tick (tick2.hpp), shared by the two translation
// units of libdust_amd.so (dust_amd.hip: eligibility, buffers, launch bookkeeping; tick2.hip: the kernel and its launcher).
#pragma once
#include "common.hpp"

namespace dust {

#ifndef T2_NSH_N
#define T2_NSH_N 16
#endif
#ifndef T2_NREP_N
#define T2_NREP_N 8
#endif
enum {
  T2_PW = 4,      // Stein particles per workgroup (4 x 128 action samples = 8 rollout waves)
  T2_NT = 1024,   // lanes per workgroup: waves 0-7 roll out, waves 8-15 run the pairwise passes
  T2_NSH = T2_NSH_N,  // shards of every arrival counter (workgroup b signals shard b % T2_NSH; one 128-byte line each)
  T2_ROW = 32,    // floats per row of the exchange buffers (D <= 32 padded to one 128-byte line)
  T2_NREP = T2_NREP_N,  // replicas of every arrival counter: an arriving wave adds to all of them (one lane each), workgroup b polls replica
                        // b % T2_NREP only - 256 workgroups polling the SAME 16 lines took 2 us per poll round (agent-scope loads of one
                        // line are served one after the other); with 8 replicas a line has 32 pollers
  T2_CNT_STRIDE = 32,
  T2_KIND = T2_NSH * T2_NREP,  // lines of one counter kind: [replica][shard]
  T2_SETS = 4 * T2_KIND + 1    // lines of one counter set: start | theta | score | lw counters, then the go word
};

struct Tick2Args {
  DevModel dm;
  int N, S, M, H, D;
  int n_iters, do_forward;
  int steps;             // ceil(N / 64) rounded up to a multiple of 16: 16-key steps of a pair wave in the theta-only pass (4 waves share the keys)
  int lik, update_a_mat, eps_base_mode, optimizer, roll_strategy, weighted_prior;
  int coef_given;
  int test_abort;        // test hooks, every k-th launch.  1 (DUST_TICK2_TEST_ABORT=k): workgroup 0 publishes "abort" as if a peer were missing;
                         // 2 (DUST_TICK2_TEST_TIMEOUT=k): the last workgroup behaves as if its last wait had given up (no COMMIT)
  unsigned int expect_aborts;  // value of status[1] the host knew when it enqueued this launch: a different value on the device means an
                         // EARLIER one-launch tick of the context did not start or did not commit and will be replayed - this one must not
                         // run ahead of that replay: it aborts at its start as well and is replayed behind it, in order
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
  unsigned int *status;     // [0] a hand-off wait timed out (sticky until the host clears it) [1] ticks to replay: did not start (not all workgroups
                            // resident) or not committed (a wait gave up) [2] workgroups that did not commit
  unsigned long long *tl;   // diagnostic build only: [grid][128] wall-clock stamps
  // Closed-loop serving (dust_svmpc_serve_start): the tick's outputs go STRAIGHT to pinned host memory and the last workgroup to finish
  // publishes the launch's sequence number there - the host spins on that word instead of a device-to-host copy and a stream
  // synchronisation; and a tick may be launched AHEAD of its plant state ("armed"): it runs everything that does not need the state
  // (particle hand-off, noise draw, prior pass) and its rollout waves wait - bounded - for the state to arrive in a pinned-host mailbox.
  float *host_out;          // pinned host mirror of the output block (a_seq | p_weights, laid out as outblk) or nullptr
  unsigned int *host_done;  // pinned host word: launch_seq when the tick is committed and its outputs are in host_out (written, like them, by
                            // the workgroup that owns the best particle), launch_seq | 0x80000000 when the launch did not start (not
                            // resident, an earlier tick awaits its replay, cancelled, state never came)
  int host_pw;              // the caller wants the particle weights too (4 N bytes over PCIe; a control loop needs a_seq only)
  unsigned int launch_seq;
  const unsigned int *mbox; // pinned host mailbox T2Mbox or nullptr: the plant state is x0 (by value)
  unsigned long long mbox_wait;  // bound of the wait for the state, s_memrealtime ticks (100 MHz)
};

// Mailbox of an armed tick, pinned host memory, two 16-byte halves each written / read as ONE aligned 16-byte access:
//   a = {seq, verdict, x0, x1}   b = {x2, x3, seq, 0}        verdict 1: state of launch `seq`; 2: launch `seq` is cancelled
// The host writes b, then a (release); the kernel polls a and, for four-entry states, reads b behind it and checks its seq.
struct T2Mbox {
  unsigned int seq_a, verdict;
  float x01[2];
  float x23[2];
  unsigned int seq_b, pad;
};

// LDS layout of one workgroup (float offsets; host and device agree through these).  The fixed-size regions come first, at
// compile-time offsets (no registers held across the tick for them); the regions sized by S / D / N / the map follow.
enum {
  T2_MAXM = 64,                               // dynamics samples per rollout (coefficient pairs in LDS)
  T2_L_TH = 0,                                // [4][32] the workgroup's particles, zero padded
  T2_L_MISC = T2_L_TH + T2_PW * T2_ROW,       // [192] small words (see tick2.hpp)
  T2_L_PPART = T2_L_MISC + 192,               // [8 units][2 queries][32 columns] partials of the prior pass, then [8][2] of L (forward: [16][4])
  T2_L_RP = T2_L_PPART + 8 * 32 * 4,          // [8 waves][16 sums][4 column groups] partials of the Stein repulsion
  T2_L_WPART = T2_L_RP + 8 * 16 * 4,          // [2][4][8][32] weighted-sum partials (likelihood score, a_mat update); after barrier B4 the
                                              // same 2 048 floats hold [16 waves][32 sums][4 column groups] partials of sum_j k_ij s_j
  T2_L_SCL = T2_L_WPART + 2 * T2_PW * 8 * T2_ROW,  // [4][32] score rows on their way out
  T2_L_COEFS = T2_L_SCL + T2_PW * T2_ROW,     // [T2_MAXM][2] dynamics coefficients of the iteration
  T2_L_VAR = T2_L_COEFS + 2 * T2_MAXM         // cst | omg | ksl | lml | grid | tile
};
struct Tick2Lds {
  int cst, omg, ksl, lml, grid, tile, total;
};
__host__ __device__ inline Tick2Lds tick2_lds(int S, int D, int steps, int grid_words) {
  Tick2Lds l;
  const int Dp = D | 1, ps = (T2_PW * S + 3) & ~3;
  l.cst = T2_L_VAR;               // [4][S] costs -> softmax weights
  l.omg = l.cst + ps;             // [4][S] omega weights
  l.ksl = l.omg + ps;             // [steps * 64][4] squared distances |y_j - x_q|^2 of the workgroup's 4 queries (prior pass), replaced in
                                  // place by the Stein kernel values k_qj (Stein pass)
  l.lml = l.ksl + steps * 64 * 4;   // [steps * 64] log pi_j of the tick's prior
  l.grid = l.lml + steps * 64;
  l.tile = l.grid + grid_words;   // [4][S][Dp] standard normals of the current iteration
  l.total = l.tile + T2_PW * S * Dp;
  return l;
}

// tick2.hip
int tick2_occupancy(int model, int mode, size_t lds_bytes, int *blocks_per_cu);
int tick2_launch(const Tick2Args &f, int model, int mode, int grid, size_t lds_bytes, hipStream_t stream);

}  // namespace dust
