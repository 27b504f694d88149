// dust_amd.hip - C ABI of libdust_amd.so (see include/dust_amd.h).  Host-side orchestration of the HIP kernels in
// rollout.hpp / stein.hpp / bandwidth.hpp / forward.hpp / mpf.hpp.  gfx950 (MI355X) only; no CPU fallback.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <new>
#include <random>
#include <string>
#include <vector>
#include <atomic>
#include <chrono>
#include <mutex>
#if defined(__x86_64__) || defined(_M_X64)
#include <emmintrin.h>
#define DUST_CPU_PAUSE() _mm_pause()
#else
#define DUST_CPU_PAUSE() std::this_thread::yield()
#include <thread>
#endif

#include "bandwidth.hpp"
#include "common.hpp"
#include "forward.hpp"
#include "fused.hpp"
#include "pairwise_big.hpp"
#include "tick2_args.hpp"
#include "rollout_states.hpp"
#include "pairwise_fused.hpp"
#include "pairwise_logp_mfma.hpp"
#include "pairwise_far.hpp"
#include "pairwise_packed.hpp"
#include "peer_gather.hpp"
#include "skid.hpp"
#include "particle_general.hpp"
#include "rollout.hpp"
#include "stein.hpp"

using namespace dust;

static thread_local std::string g_err;
static int fail(int code, const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
#define HIP_TRY(x)                                                                                       \
  do {                                                                                                   \
    hipError_t e_ = (x);                                                                                 \
    if (e_ != hipSuccess) return fail(DUST_ERR_HIP, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)
#define TRY(x)               \
  do {                       \
    int s_ = (x);            \
    if (s_ != DUST_OK) return s_; \
  } while (0)

extern "C" const char *dust_last_error(void) { return g_err.c_str(); }
extern "C" int dust_abi_version(void) { return DUST_ABI_VERSION; }
extern "C" int dust_device_count(int *count) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    return fail(DUST_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
  }
  *count = n;
  return DUST_OK;
}

static const char *k_names[DUST_K_COUNT] = {"rollout_kernel", "pairwise_kernel<PRIOR>", "pairwise_kernel<STEIN>", "update_kernel",
                                            "forward(finalize+roll)", "bandwidth_kernel", "mpf_kernel", "states_kernel"};
extern "C" const char *dust_kernel_name(int id) { return (id >= 0 && id < DUST_K_COUNT) ? k_names[id] : ""; }

// peer_gather.hpp: the peers' buffers of the three exchanges (score rows, particles, log-weights) and arrival words, mapped through HIP IPC
struct PeerState {
  int world, rank;
  float *buf[PEER_BUFS][PEER_MAX];    // [which][rank]: entry `rank` is the context's own buffer
  unsigned int *flags[PEER_MAX];      // arrival words of every rank ([PEER_BUFS][PEER_MAX] each); entry `rank`: flags_local
  unsigned int *flags_local;          // [PEER_ROWS][PEER_MAX], then the store kernel's workgroup counters [PEER_MAX], then the error word
  void *opened[PEER_MAX][PEER_BUFS + 1];
  unsigned int seq[PEER_BUFS];
};

struct dust_ctx {
  dust_config cfg;
  int N, S, M, H, da, ds, P, D, n0, nloc;
  hipStream_t stream;
  bool own_stream;
  hipStream_t stream2;  // side stream: the prior pass runs beside the rollout kernel (both only read theta)
  hipEvent_t ev_fork, ev_join;
  hipEvent_t ev_order;   // behind the side stream's query_order_kernel (pairwise_packed.hpp); order_pending: nobody has waited for it yet
  bool order_pending;
  hipStream_t pair_stream;  // stream the next pairwise launch goes to (stream or stream2)
  // particles / prior / controller state
  float *theta, *thetaT, *mu, *muT, *logmix, *mixw;
  float *thetaT_alt;     // K2: the transposed particles of the next iteration, written by the phi + update launch (ping-pong with thetaT)
  bool k2_thetaT_fresh;  // thetaT is the transpose of theta as the last K2 update left it (valid from one iteration to the next of ONE loop only)
  bool mu_aliased;  // reference quirk: after update_prior the GMM means alias theta's storage (svgd.py:87, svmpc.py:160-170)
  float *a_mat, *a_seq, *a_mix, *eta;
  // per-iteration products
  float *costsT, *omegaT, *grad_lik, *grad_pri, *score, *phi, *logl, *logp, *lw, *pw, *a_seq_out, *bw;
  int *istar;
  float *adam_m, *adam_v;
  float *pA, *pB, *pM, *pL;  // slice partials of the tiled pairwise passes
  size_t pA_cap, pB_cap, pM_cap, pL_cap;
  float *pS;                 // Gram x score partials of the one-launch iteration (pA still holds the prior's while its Stein tiles run)
  size_t pS_cap;
  float *mw_dev;             // [M] unscented-transform weights of the dynamics samples (nullptr: mean)
  float *theta_w, *mu_w;     // full prior covariance: whitened copies L_p^-1 x of the particles / prior means, refreshed before each prior pass
  float *cz_dev;             // recorded control-noise draws [cz_sets][H][M*S*N][da] (dust_set_ctrl_noise), consumed one set per rollout launch
  size_t cz_cap;
  int cz_sets, cz_next;
  float *xpad;               // [N][DPB] zero-padded query rows of the large-N pairwise kernel
  size_t xpad_cap;
  float *kmat;               // [n_local][ldK] Stein kernel values of the current theta (pairwise_fused.hpp), valid while kmat_valid
  size_t kmat_cap;
  bool kmat_valid;
  bool t2_mu_aliased;        // the prior means aliased the particles BEFORE the last one-launch tick (its replay starts from that state)
  bool handoff_banned;       // an in-launch wait timed out once (the device is shared with another process): plain kernels from then on
  bool no_handoff;           // replay of a tick that found the device shared: plain kernels only, nothing that spins on its own grid
  unsigned long persist_declined;  // key of the (shape, state) for which the one-launch ticks last declined: no staging for it again
  SkidModel skid;            // DUST_MODEL_SKID_STEER: model parameters and the quadratic cost (dust_set_skid_steer)
  // pairwise_packed.hpp: the near-key lists of the last large-set pass 1 - key indices [tiles][chunks * 64], unit offsets
  // [tiles][chunks + 1], unit query masks [tiles][chunks][4], slice boundaries of pass 1 / pass 2, non-zero flags of the units' kernel blocks
  float *pk_idx, *pk_uoff, *pk_uq, *pk_soff, *pk_goff, *pk_nzu, *pk_perm, *pk_lead;
  size_t pk_idx_cap, pk_uoff_cap, pk_uq_cap, pk_soff_cap, pk_goff_cap, pk_nzu_cap, pk_perm_cap, pk_lead_cap;
  float *lp_idx, *lp_uoff, *lp_uq, *lp_soff, *lp_goff;  // the log-p pass' own run lists (64-query groups of the UPDATED particles)
  size_t lp_idx_cap, lp_uoff_cap, lp_uq_cap, lp_soff_cap, lp_goff_cap;
  bool params_staged;             // inside dust_dual_tick: the tick's dynamics samples already sit in params_dev (drawn there by the filter's kernel)
  bool stagewise;                 // inside dust_svmpc_phi (a stage-wise call on caller-supplied inputs): index order - the same inputs give the same bits, call after call
  bool pk_order;                  // pass 1 walks its queries in tile order (pk_perm) and notes their leaders (pk_lead) for the next order
  int pk_tiles, pk_umax, pk_jsg;  // geometry of those lists
  bool pk_masks, pk_dense;        // the units carry query masks / pass 2 visits every unit (DUST_DENSE)
  float *far_z, *far_n, *far_f;  // pairwise_far.hpp: binary16 rows, (norms, log weights), unit flags [tiles][chunks] bytes
  size_t far_z_cap, far_n_cap, far_f_cap;
  int far_tiles, far_chunks;  // geometry of the flags the last fused pass used (0: none)
  float *far_q;               // per-unit near-query masks of the fused pass [tiles][chunks][4] words
  size_t far_q_cap;
  size_t far_cnt_cap;
  float *far_cnt;             // [4] device counters {far, all} of the fused pass' and the log-p pass' last pre-pass
  unsigned int *far_cnt_host; // pinned copy (arrives a tick late at worst: it only steers whether the NEXT pre-pass is worth its launches)
  int far_logp_skip;          // log-p pre-pass: launches left to skip before the next probe
  unsigned int far_logp_issued;  // counted log-p pre-passes issued so far (the report the next decision reads carries this number)
  float *far_g;               // the flags of the last log-p pass [groups][chunks] bytes
  size_t far_g_cap;
  int far_groups, far_gchunks;
  int fused_js;  // slices of the repulsion partials of the last pairwise_packed_kernel launch (update_kernel merges them)
  int prior_js;  // slices of the prior partials when pairwise_packed_kernel wrote them (its own split); 0 = pair_geometry's
  // staging
  float *noise_stage, *actions, *states, *params_dev, *state_dev, *tmp, *costs_stage, *tile_scratch;
  size_t noise_cap, actions_cap, states_cap, params_cap, tmp_cap, tile_cap;
  float *wg_flags;  // one word per workgroup of the whole-line stored-states kernel: "take the general path" (rollout_states.hpp)
  size_t wg_flags_cap;
  uint32_t *grid_bits;
  int nx, ny;
  float off_x, off_y;
  uint32_t *ctr_dev;  // device counters {tick, iter, adam_step, pad}
  unsigned int *fused_cnt;  // [tiles + 1]: per-tile arrival counters of the fused launch, last word = spin-timeout flag
  int fused_tiles;
  bool fused_dirty;   // a fused launch ran and no update kernel has re-armed the counters yet
  float *theta_home, *theta_alt;  // theta ping-pong of the fused Stein+update launch (theta == one of the two)
  bool theta_pinned;              // dust_gather_buffers handed theta's address out: no ping-pong any more
  const float *graph_theta;       // theta at the start of the captured tick
  unsigned int *stein_cnt;  // [tiles + 1]: arrival counters of the Stein+update launch (re-armed by the next rollout launch)
  int stein_tiles;
  bool stein_dirty;
  // one-launch SVGD iteration (fused.hpp svgd_iter_kernel): two sets of [tiles | JS | tiles] counter lines, then the time-out flag
  unsigned int *iter_cnt;
  int iter_tiles, iter_js, iter_set;
  float *score_hs;  // [2][N][D] score rows handed over as data inside the one-launch iteration (sentinel-filled between uses)
  // C-side RCCL communicator of a sharded context (dust_comm_init): the sharded tick issues its all-gathers on the context's stream
  void *comm;  // ncclComm_t
  int comm_rank, comm_world;
  struct PeerState *peer;  // direct peer-store all-gathers (peer_gather.hpp), or nullptr: the collective library's
  // tick outputs: a_seq_out | p_weights | time-out word of the persistent tick live in ONE device block, copied with ONE
  // hipMemcpyAsync into a pinned host buffer (then one stream synchronisation per tick that returns outputs)
  float *outblk;
  size_t out_floats;   // a_seq (D rounded up to 32) + N + 32
  float *out_pinned;   // host, pinned: out_floats + 4 words for the hand-off flags of the launch-per-iteration paths
  int tick_occ;       // resident workgroups per CU of the instantiation in use (0: not queried yet)
  size_t tick_occ_lds;
  // owner-computes persistent tick (tick2.hpp): exchange buffers, two counter sets, bookkeeping of ticks that did not start
  float *t2_xq, *t2_sq, *t2_lwq;
  unsigned int *t2_cnt;
  int t2_set, t2_occ, t2_gens;
  size_t t2_occ_lds;
  unsigned int t2_aborts_seen;  // value of the device-side "did not start" counter already accounted for
  bool t2_inflight;             // a tick2 launch was enqueued since the last check
  int t2_grid;                  // workgroups of the last tick2 launch
  bool t2_transactional;        // ... of the owner-computes kernel, which commits nothing when one of its waits gives up
  float t2_state[4];            // inputs of the last tick2 launch (replayed on the launch-per-iteration path if it did not start)
  int t2_steps;
  bool t2_fwd, t2_replayable;
  // One entry per one-launch tick enqueued since the last settle: the inputs its replay needs.  A launch that does not start - or does
  // not commit - makes every LATER one-launch tick of the context abort at its start as well (the kernels compare the device's abort
  // count with the value the host knew when it launched them: `expect_aborts`), so the ticks that have to be replayed are always the
  // tail of this queue and are replayed in their order, each with its own inputs.
  // Development switches and test hooks (environment variables), read ONCE when the context is created or cloned - the tick entry
  // points called getenv() several times per control tick before.  -1: unset, otherwise atoi of the value.
  struct EnvSw {
    int comm_force, pair_big, pair_fused, states_form, dense, far, logp_mfma, no_fuse, no_persist, no_tick2,
        tick2_test_abort, tick2_test_timeout, no_comm_overlap;
    int noise_general;  // DUST_NOISE_GENERAL=1 (development switch): control-channel noise always through particle_general.hpp's first pass
    int k2_form;     // DUST_K2_FORM=0 (development switch): K2's bandwidth launch stays behind the prior + rollout launch (rounds 1-5);
                     // default: the bandwidths ride in that launch, phi reads the row-major particles (round 6)
    int logp_pack;   // DUST_LOGP_PACK=0 / 1: the log-p pass never / always walks run lists (default: from 8 192 local rows on)
    int pack_order;  // DUST_PACK_ORDER=0: the queries stay in index order (development switch)
    int pack_merge;  // DUST_PACK_MERGE=0: PLAIN run lists even below the exact-zero threshold (development switch)
    float far_t;  // DUST_FAR_T: the far pre-pass' threshold (pairwise_far.hpp), default DUST_FAR_T_DEFAULT
  } env;
  struct T2Replay {
    float state[4];
    int steps;
    bool fwd, replayable, mu_aliased;
    bool cancelled;             // an armed launch the host cancelled (or whose state never came): nobody asked for this tick - dropped, not replayed
    std::vector<float> params;  // host copy of the caller's [steps][M][P] dynamics samples (empty: none)
  };
  std::vector<T2Replay> *t2_queue;  // (a pointer: the context block is zero-filled as a whole when it is created)
  // closed-loop serving (dust_svmpc_serve_start; tick2_args.hpp): outputs straight to pinned host memory + a done word the host spins on,
  // and the NEXT tick launched ahead of its plant state ("armed": its rollout waves wait for the state in a pinned mailbox)
  bool serve_on;
  int serve_steps;
  unsigned long long serve_wait_ticks;  // bound of an armed launch's wait for its state (100 MHz ticks)
  float *serve_host;                    // pinned: [out_floats] output mirror | [32 words] done | [32 words] T2Mbox
  unsigned int serve_seq;               // sequence number of the last launch that reports to the done word
  bool serve_pw;                        // the served calls ask for the particle weights as well (decided by the first served call)
  int t2_launch_mode;                   // what the next launch_tick2 call adds: 0 nothing, 1 outputs + done word in pinned memory, 2 the same, armed
  bool armed;                           // an armed launch is in flight and its state has not been posted
  unsigned int armed_seq;
  bool cancel_unsettled;                // an armed launch was cancelled: the device's abort count runs ahead of t2_aborts_seen until the next settle
  int serve_misses;                     // consecutive armed launches whose state came too late (the loop is slower than the bound): stop arming at 3
  double armed_at;                      // host time the armed launch was enqueued (seconds, steady clock)
  long long n_served, n_armed_hit;      // sticky: ticks answered through the done word; of them, ticks that had been launched ahead
  const float *t2_params_host;  // the caller's params of the call being staged (valid inside try_persistent only)
  long long t2_replays;         // sticky: ticks replayed so far (dust_tick_stats)
  long long n_tick2, n_tick_other;  // sticky: optimize / tick calls served by tick2.hpp, the other paths
  // hipGraph replay of a whole tick (dust_svmpc_tick)
  hipGraph_t graph;
  hipGraphExec_t graph_exec;
  hipGraph_t graph_alt;          // the tick's OTHER captured variant (log-p pre-pass on / off: logp_far_decide), swapped in when the decision flips
  hipGraphExec_t graph_exec_alt;
  int graph_far;                 // the active capture's variant
  bool logp_far_on, logp_far_decided;  // this tick's decision (taken once per tick on the host: a capture freezes what it saw)
  int graph_steps;
  const void *graph_eps;
  int graph_seen;     // consecutive eager ticks with the same shape (capture on the 2nd)
  bool capturing;
  bool have_sample, actions_valid;
  bool states_valid, states_f16;  // the last rollout launch left states [M][S][N][H+1][ds] in `states` (binary16 when states_f16)
  bool k2_inline_want;  // K2, set by step_device around its score launch: the prior + rollout launch may carry the bandwidth role
  bool k2_bw_inline;    // K2: ... and did (the bandwidths of the current theta are in c->bw; no transposed copy was made)
  bool k2_bw_ahead;   // K2: transpose + bandwidths of the current theta are already in flight on the side stream
  float k2_fixed_h;   // > 0: fixed-bandwidth RBF (dust_set_k2_bandwidth), else the median trick
  float k2_min_bw;    // RBF(minimum_bw=): clamp of the median-trick bandwidths (0: the reference's default 1e-5)
  bool noise_f16;     // the eps / actions handed to the current call are binary16 (DUST_EPS_F16), set by the API entry points
  bool actions_f16;   // the kept actions were stored as binary16
  int graph_flags;
  unsigned long long *stamps_dev;  // diagnostic build only: [DUST_K_COUNT][16]
  unsigned long long *tl_dev;      // diagnostic build only: [8192][4] launch timeline of svgd_iter_kernel
  // profiling
  bool prof;
  hipEvent_t ev0, ev1;
  double prof_ms[DUST_K_COUNT];
  int64_t prof_n[DUST_K_COUNT];
};

static DevParam dev_param(const dust_param &p) { return DevParam{p.kind, p.column, p.value}; }

static DevModel make_dev_model(const dust_ctx *c) {
  const dust_config &g = c->cfg;
  DevModel m;
  memset(&m, 0, sizeof m);
  m.model = g.model;
  m.P = c->P;
  m.log_space = g.params_log_space;
  m.interleave = g.params_interleave;
  m.dt = g.dt;
  m.g = dev_param(g.g);
  m.mass = dev_param(g.mass);
  m.length = dev_param(g.length);
  m.max_torque = (float)g.max_torque;
  m.max_speed_pend = (float)g.max_speed_pend;
  m.w_cos = (float)g.w_cos;
  m.w_vel = (float)g.w_vel;
  m.max_speed = g.max_speed;
  m.max_acc = g.max_accel;
  m.can_crash = g.can_crash;
  m.with_obstacle = g.with_obstacle && c->grid_bits != nullptr;
  m.inv_cell = (float)(1.0 / g.cell_size);
  m.off_x = c->off_x;
  m.off_y = c->off_y;
  m.nx = c->nx;
  m.ny = c->ny;
  m.grid_bits = c->grid_bits;
  for (int k = 0; k < 4; ++k) {
    m.target[k] = g.target[k];
    m.w_state[k] = g.w_state[k];
    m.w_term[k] = g.w_term[k];
  }
  m.w_ctrl[0] = g.w_ctrl[0];
  m.w_ctrl[1] = g.w_ctrl[1];
  m.w_obs = g.w_obs;
  return m;
}

// Host evaluation of the model coefficients when no parameter is sampled (same Python-float / fp32-tensor rules as the
// device make_coef in common.hpp; pendulum.py:93-96, particle.py:152).
struct HVal {
  int t;
  double d;
  float f;
};
static float h_tof(HVal v) { return v.t ? v.f : (float)v.d; }
static HVal h_val(const dust_param &p) { return p.kind == DUST_PARAM_TENSOR0D ? HVal{1, 0.0, (float)p.value} : HVal{0, p.value, 0.f}; }
static HVal h_mul(HVal a, HVal b) { return (!a.t && !b.t) ? HVal{0, a.d * b.d, 0.f} : HVal{1, 0.0, h_tof(a) * h_tof(b)}; }
static HVal h_div(HVal a, HVal b) {
  if (!a.t && !b.t) return HVal{0, a.d / b.d, 0.f};
  if (a.t && !b.t) return HVal{1, 0.0, a.f / (float)b.d};
  if (!a.t && b.t) return HVal{1, 0.0, (1.0f / b.f) * (float)a.d};
  return HVal{1, 0.0, a.f / b.f};
}
static void host_coef(const dust_config &g, float out[2]) {
  if (g.model == DUST_MODEL_PENDULUM) {
    const HVal gg = h_val(g.g), m = h_val(g.mass), l = h_val(g.length);
    const HVal l2 = l.t ? HVal{1, 0.0, l.f * l.f} : HVal{0, l.d * l.d, 0.f};
    out[0] = h_tof(h_div(h_mul(HVal{0, -3.0, 0.f}, gg), h_mul(HVal{0, 2.0, 0.f}, l)));
    out[1] = h_tof(h_div(HVal{0, 3.0, 0.f}, h_mul(m, l2)));
  } else {
    out[0] = h_tof(h_val(g.mass));
    out[1] = 0.f;
  }
}

template <class T>
static int dalloc(T **p, size_t n) {
  *p = nullptr;
  if (n == 0) n = 1;
  HIP_TRY(hipMalloc((void **)p, n * sizeof(T)));
  return DUST_OK;
}
static int ensure(float **p, size_t *cap, size_t n) {
  if (*cap >= n) return DUST_OK;
  if (*p) HIP_TRY(hipFree(*p));
  *p = nullptr;
  *cap = 0;
  TRY(dalloc(p, n));
  *cap = n;
  return DUST_OK;
}

struct Prof {
  dust_ctx *c;
  int id;
  Prof(dust_ctx *c_, int id_) : c(c_), id(id_) {
    if (c->prof) (void)hipEventRecord(c->ev0, c->stream);
  }
  ~Prof() {
    if (c->prof) {
      (void)hipEventRecord(c->ev1, c->stream);
      (void)hipEventSynchronize(c->ev1);
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
      c->prof_ms[id] += ms;
      c->prof_n[id] += 1;
    }
  }
};

static int validate(const dust_config *g) {
  if (!g) return fail(DUST_ERR_INVALID, "null config");
  if (g->abi_version != DUST_ABI_VERSION) return fail(DUST_ERR_INVALID, "ABI version mismatch: got %d want %d", g->abi_version, DUST_ABI_VERSION);
  if (g->n_policies < 1 || g->n_samples < 1 || g->n_params < 1 || g->horizon < 1) return fail(DUST_ERR_INVALID, "sizes must be positive");
  if (g->model == DUST_MODEL_PENDULUM) {
    if (g->dim_a != 1 || g->dim_s != 2) return fail(DUST_ERR_INVALID, "PendulumModel has dim_a=1, dim_s=2");
    if (g->cost != DUST_COST_PENDULUM_QUADCOS) return fail(DUST_ERR_UNSUPPORTED, "pendulum model needs the quad-cos cost family");
  } else if (g->model == DUST_MODEL_PARTICLE) {
    if (g->control_type != DUST_CONTROL_ACCELERATION && g->control_type != DUST_CONTROL_VELOCITY) return fail(DUST_ERR_INVALID, "bad control_type %d", g->control_type);
    const int want_ds = g->control_type == DUST_CONTROL_VELOCITY ? 2 : 4;  // particle.py:41-60
    if (g->dim_a != 2 || g->dim_s != want_ds) return fail(DUST_ERR_INVALID, "Particle has dim_a=2 and dim_s=4 (acceleration control) / 2 (velocity control)");
    if (g->cost != DUST_COST_PARTICLE_DEFAULT) return fail(DUST_ERR_UNSUPPORTED, "particle model needs Particle.default_*_cost");
    if (g->ctrl_noise && (!(g->dyn_std[0] >= 0.f) || !(g->dyn_std[1] >= 0.f))) return fail(DUST_ERR_INVALID, "noise_std must be >= 0");
  } else if (g->model == DUST_MODEL_SKID_STEER) {
    if (g->dim_a != 2 || g->dim_s != 5) return fail(DUST_ERR_INVALID, "SkidSteerRobot has dim_a=2, dim_s=5");
    if (g->cost != DUST_COST_QUADRATIC) return fail(DUST_ERR_UNSUPPORTED, "the skid-steer model runs with the quadratic cost family (DUST_COST_QUADRATIC)");
    if (g->dim_p > 3) return fail(DUST_ERR_INVALID, "SkidSteerRobot has 3 parameters");
  } else {
    return fail(DUST_ERR_UNSUPPORTED, "unknown model id %d", g->model);
  }
  if (g->model != DUST_MODEL_PARTICLE && (g->control_type != 0 || g->ctrl_noise != 0))
    return fail(DUST_ERR_UNSUPPORTED, "control_type / ctrl_noise are Particle fields (particle.py:13-31); the other models have no control-channel noise");
  const int D = g->horizon * g->dim_a;
  if (D > 128) return fail(DUST_ERR_UNSUPPORTED, "H*da = %d > 128 not supported by the kernels", D);
  if ((double)g->n_samples * g->n_policies * D >= 1073741824.0)
    return fail(DUST_ERR_UNSUPPORTED, "n_samples * n_policies * H * da >= 2^30: the noise tile offsets are 32-bit");
  if (g->dim_p < 0 || g->dim_p > 4) return fail(DUST_ERR_INVALID, "dim_p out of range");
  if (g->kernel < 0 || g->kernel > DUST_KERNEL_IMQ) return fail(DUST_ERR_INVALID, "bad kernel id");
  if (g->optimizer != DUST_OPT_SGD && g->optimizer != DUST_OPT_ADAM) return fail(DUST_ERR_UNSUPPORTED, "optimizer must be SGD or Adam");
  if (g->n_policies > 16384) return fail(DUST_ERR_UNSUPPORTED, "n_policies > 16384 (finalize_kernel keeps all particles of the tick epilogue in one workgroup)");
  if (g->shard_size < 0 || g->shard_offset < 0 || g->shard_offset + g->shard_size > g->n_policies)
    return fail(DUST_ERR_INVALID, "bad shard [%d,+%d) of %d", g->shard_offset, g->shard_size, g->n_policies);
  for (int d = 0; d < g->dim_a; ++d)
    if (!(g->chol_a[d] > 0.f) || !(g->sigma_a[d] > 0.f) || !(g->sigma_p[d] > 0.f))
      return fail(DUST_ERR_INVALID, "covariance diagonals must be positive (only diagonal a_cov / prior covariances are supported)");
  if (!(g->temperature > 0.f)) return fail(DUST_ERR_INVALID, "temperature must be > 0");
  if (g->full_cov) {
    if (g->dim_a != 2) return fail(DUST_ERR_INVALID, "full covariances are 2 x 2: dim_a = %d", g->dim_a);
    if (!(g->chol_p[0] > 0.f) || !(g->chol_p[2] > 0.f)) return fail(DUST_ERR_INVALID, "chol_p must be the Cholesky factor of an SPD prior covariance (positive diagonal)");
  }
  return DUST_OK;
}

static void free_all(dust_ctx *c) {
  // {theta, theta_alt} are always the two particle buffers, whichever is current
  float **fp[] = {&c->theta, &c->theta_alt, &c->thetaT, &c->thetaT_alt, &c->mu, &c->muT, &c->logmix, &c->mixw, &c->a_mat, &c->a_seq, &c->a_mix, &c->eta,
                  &c->costsT, &c->omegaT, &c->grad_lik, &c->grad_pri, &c->score, &c->phi, &c->logl, &c->logp, &c->lw,
                  &c->outblk, &c->bw, &c->adam_m, &c->adam_v, &c->noise_stage, &c->actions, &c->states, &c->params_dev,
                  &c->state_dev, &c->tmp, &c->costs_stage, &c->tile_scratch, &c->wg_flags, &c->pA, &c->pB, &c->pM, &c->pL, &c->pS, &c->xpad, &c->kmat, &c->pk_idx, &c->pk_uoff, &c->pk_uq, &c->pk_soff, &c->pk_goff, &c->pk_nzu, &c->pk_perm, &c->pk_lead, &c->lp_idx, &c->lp_uoff, &c->lp_uq, &c->lp_soff, &c->lp_goff, &c->far_z, &c->far_n, &c->far_f, &c->far_g, &c->far_q, &c->far_cnt, &c->mw_dev, &c->cz_dev, &c->theta_w, &c->mu_w};
  for (auto p : fp)
    if (*p) (void)hipFree(*p);
  if (c->graph_exec) (void)hipGraphExecDestroy(c->graph_exec);
  if (c->graph) (void)hipGraphDestroy(c->graph);
  if (c->graph_exec_alt) (void)hipGraphExecDestroy(c->graph_exec_alt);
  if (c->graph_alt) (void)hipGraphDestroy(c->graph_alt);
  if (c->ctr_dev) (void)hipFree(c->ctr_dev);
  if (c->fused_cnt) (void)hipFree(c->fused_cnt);
  if (c->stein_cnt) (void)hipFree(c->stein_cnt);
  if (c->iter_cnt) (void)hipFree(c->iter_cnt);
  if (c->score_hs) (void)hipFree(c->score_hs);
  if (c->t2_cnt) (void)hipFree(c->t2_cnt);
  if (c->t2_xq) (void)hipFree(c->t2_xq);
  if (c->t2_sq) (void)hipFree(c->t2_sq);
  if (c->t2_lwq) (void)hipFree(c->t2_lwq);
  if (c->out_pinned) (void)hipHostFree(c->out_pinned);
  if (c->far_cnt_host) (void)hipHostFree(c->far_cnt_host);
  if (c->istar) (void)hipFree(c->istar);
  if (c->grid_bits) (void)hipFree(c->grid_bits);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->stream2) (void)hipStreamDestroy(c->stream2);
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  if (c->ev_order) (void)hipEventDestroy(c->ev_order);
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
}

// a communicator of one rank is the unsharded problem: it runs the single-GPU path (DUST_COMM_FORCE=1: development switch that
// sends a world-1 context through the sharded tick and its RCCL calls, for tests and per-collective latency measurements)
static int env_int(const char *name) {  // -1: unset; a set but empty / non-numeric value counts as 1 ("switch on")
  const char *v = getenv(name);
  if (!v) return -1;
  return (*v >= '0' && *v <= '9') ? atoi(v) : 1;
}
static void env_read(dust_ctx *c) {
  c->env.comm_force = env_int("DUST_COMM_FORCE");
  c->env.pair_big = env_int("DUST_PAIR_BIG");
  c->env.pair_fused = env_int("DUST_PAIR_FUSED");
  c->env.states_form = env_int("DUST_STATES_FORM");
  c->env.dense = env_int("DUST_DENSE");
  c->env.far = env_int("DUST_FAR");
  c->env.logp_mfma = env_int("DUST_LOGP_MFMA");
  c->env.no_fuse = env_int("DUST_NO_FUSE");
  c->env.no_persist = env_int("DUST_NO_PERSIST");
  c->env.no_tick2 = env_int("DUST_NO_TICK2");
  c->env.tick2_test_abort = env_int("DUST_TICK2_TEST_ABORT");
  c->env.tick2_test_timeout = env_int("DUST_TICK2_TEST_TIMEOUT");
  c->env.no_comm_overlap = env_int("DUST_NO_COMM_OVERLAP");
  c->env.pack_merge = env_int("DUST_PACK_MERGE");
  c->env.pack_order = env_int("DUST_PACK_ORDER");
  c->env.logp_pack = env_int("DUST_LOGP_PACK");
  c->env.k2_form = env_int("DUST_K2_FORM");
  c->env.noise_general = env_int("DUST_NOISE_GENERAL");
  const char *ft = getenv("DUST_FAR_T");
  c->env.far_t = (ft && *ft) ? (float)atof(ft) : DUST_FAR_T_DEFAULT;
}
// Particle configurations the specialised rollout kernels do not take (particle_general.hpp): velocity control, and control-channel noise
// that is not identically zero (the reference's constructor default is deterministic=False with noise_std = zeros(2): draws are made and
// multiplied by 0, which leaves every action as it is - that default runs on the fast kernels).
static bool particle_general(const dust_ctx *c) {
  if (c->cfg.model != DUST_MODEL_PARTICLE) return false;
  return c->cfg.control_type == DUST_CONTROL_VELOCITY || (c->cfg.ctrl_noise && (c->cfg.dyn_std[0] != 0.f || c->cfg.dyn_std[1] != 0.f));
}
// families whose rollouts are a launch of their own followed by the regular kernel in its injected-costs mode: launch-per-iteration path only
// ... and contexts with a FULL 2 x 2 action / prior covariance (disco.py:91-98, svgd.py:84-89): the one-launch and fused forms carry the
// diagonal arithmetic only - the launch-per-iteration kernels take the off-diagonal terms (rollout.hpp stage 1, the whitened prior pass)
static bool full_cov(const dust_ctx *c) { return c->cfg.full_cov != 0; }
static bool two_pass_family(const dust_ctx *c) { return c->cfg.model == DUST_MODEL_SKID_STEER || particle_general(c) || full_cov(c); }
static bool comm_active(const dust_ctx *c) { return c->comm && (c->comm_world > 1 || c->env.comm_force >= 0); }
static void comm_release(dust_ctx *c);
extern "C" int dust_comm_peer_gather(dust_ctx *c, int on);
static int peer_check(dust_ctx *c);
static int sharded_steps(dust_ctx *c, const float *state, int n_steps, const float *eps, const float *params, int flags);
static bool tick2_shape_ok(dust_ctx *c, int n_steps);
static int sharded_forward(dust_ctx *c);

// One-launch ticks (persist.hpp, tick2.hpp) spin on their own workgroups: every workgroup of the launch must be resident.  Two contexts
// of ONE process ticking on the same device from two threads would interleave their grids (tick2.hpp then aborts and replays - late
// but correct; persist.hpp has no such protocol and would time out).  So as soon as a second context lives on a device, the one-launch
// kernels of that device are chained across streams with an event: launch k + 1 waits for launch k, whichever context issued it.  A
// single context (the common case, the bench) pays nothing.  Other PROCESSES on the device remain tick2.hpp's start barrier's business.
enum { DUST_MAX_DEV = 64 };
static std::atomic<int> g_live_ctx[DUST_MAX_DEV];
static std::mutex g_persist_mu[DUST_MAX_DEV];
static hipEvent_t g_persist_ev[DUST_MAX_DEV];
static hipStream_t g_persist_last[DUST_MAX_DEV];
struct PersistChain {  // RAII around ONE one-launch kernel launch on c->stream
  dust_ctx *c;
  int dev;
  bool on, locked;
  explicit PersistChain(dust_ctx *c_);
  ~PersistChain();
};

// Armed launches (closed-loop serving) occupy every CU while they wait for their plant state: one per device at most, on record here so
// that whoever needs the device next - another context being created, the dynamics filter's kernels - can cancel it instead of waiting
// for its bound.
static std::mutex g_armed_mu[DUST_MAX_DEV];
static dust_ctx *g_armed[DUST_MAX_DEV];
static double host_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static T2Mbox *serve_mbox(dust_ctx *c) { return reinterpret_cast<T2Mbox *>(c->serve_host + c->out_floats + 32); }
static volatile unsigned int *serve_done(dust_ctx *c) { return reinterpret_cast<volatile unsigned int *>(c->serve_host + c->out_floats); }
// post a verdict for launch `seq`: half b first, then half a as ONE aligned 16-byte store (tick2_args.hpp T2Mbox)
static void serve_post(dust_ctx *c, unsigned int seq, unsigned int verdict, const float *state) {
  T2Mbox *m = serve_mbox(c);
  alignas(16) unsigned int hb[4] = {0u, 0u, seq, 0u}, ha[4] = {seq, verdict, 0u, 0u};
  if (state) {
    memcpy(&ha[2], &state[0], 4);
    if (c->ds > 1) memcpy(&ha[3], &state[1], 4);
    if (c->ds > 2) memcpy(&hb[0], &state[2], 4);
    if (c->ds > 3) memcpy(&hb[1], &state[3], 4);
  }
#if defined(__x86_64__) || defined(_M_X64)
  _mm_store_si128(reinterpret_cast<__m128i *>(&m->x23[0]), _mm_load_si128(reinterpret_cast<const __m128i *>(hb)));
  std::atomic_thread_fence(std::memory_order_release);
  _mm_store_si128(reinterpret_cast<__m128i *>(&m->seq_a), _mm_load_si128(reinterpret_cast<const __m128i *>(ha)));
#else
  // no single-copy-atomic 16-byte store to rely on: the sequence word of each half goes LAST (the reader takes a half only when its
  // sequence word matches, and reads the sequence word first - tick2_args.hpp T2Mbox)
  volatile unsigned int *pb = reinterpret_cast<volatile unsigned int *>(&m->x23[0]);
  pb[0] = hb[0]; pb[1] = hb[1]; pb[3] = hb[3];
  std::atomic_thread_fence(std::memory_order_release);
  pb[2] = hb[2];
  std::atomic_thread_fence(std::memory_order_release);
  volatile unsigned int *pa = reinterpret_cast<volatile unsigned int *>(&m->seq_a);
  pa[1] = ha[1]; pa[2] = ha[2]; pa[3] = ha[3];
  std::atomic_thread_fence(std::memory_order_release);
  pa[0] = ha[0];
#endif
  std::atomic_thread_fence(std::memory_order_seq_cst);
}
// cancel the armed launch of `c` (caller holds g_armed_mu of its device, or is the owning thread of a context that is not registered)
static void serve_cancel_locked(dust_ctx *c) {
  if (!c->armed) return;
  serve_post(c, c->armed_seq, 2u, nullptr);
  c->armed = false;
  c->cancel_unsettled = true;
  if (!c->t2_queue->empty()) c->t2_queue->back().cancelled = true;
  const int dev = c->cfg.device;
  if (dev >= 0 && dev < DUST_MAX_DEV && g_armed[dev] == c) g_armed[dev] = nullptr;
}
static void serve_cancel(dust_ctx *c) {
  if (!c || !c->armed) return;
  const int dev = c->cfg.device;
  if (dev >= 0 && dev < DUST_MAX_DEV) {
    std::lock_guard<std::mutex> lk(g_armed_mu[dev]);
    serve_cancel_locked(c);
  } else {
    serve_cancel_locked(c);
  }
}
// whoever is about to need the whole device (see above)
static void serve_cancel_device(int dev, const dust_ctx *except = nullptr) {
  if (dev < 0 || dev >= DUST_MAX_DEV) return;
  std::lock_guard<std::mutex> lk(g_armed_mu[dev]);
  if (g_armed[dev] && g_armed[dev] != except) serve_cancel_locked(g_armed[dev]);
}

extern "C" void dust_destroy(dust_ctx *c) {
  if (!c) return;
  serve_cancel(c);
  if (c->cfg.device >= 0 && c->cfg.device < DUST_MAX_DEV) {
    std::lock_guard<std::mutex> lk(g_persist_mu[c->cfg.device]);
    g_live_ctx[c->cfg.device].fetch_sub(1);
    if (g_persist_last[c->cfg.device] == c->stream) g_persist_last[c->cfg.device] = nullptr;  // (a later stream may reuse the address)
  }
  (void)hipSetDevice(c->cfg.device);
  (void)hipStreamSynchronize(c->stream);
  if (c->stream2) (void)hipStreamSynchronize(c->stream2);
  comm_release(c);
  free_all(c);
  if (c->serve_host) (void)hipHostFree(c->serve_host);
  delete c->t2_queue;
  delete c;
}

PersistChain::PersistChain(dust_ctx *c_) : c(c_), dev(c_->cfg.device), on(false), locked(false) {
  static const bool off = getenv("DUST_NO_CHAIN") != nullptr;  // development switch
  if (off || dev < 0 || dev >= DUST_MAX_DEV) return;
  // (the tenant count is tested UNDER the device's mutex, which dust_create holds while it counts itself in and drains the device:
  //  a launch cannot slip unchained between another thread's increment and its first kernel - ADVICE r3)
  g_persist_mu[dev].lock();
  locked = true;
  // (a single tenant needs no event, but KEEPS the mutex until its kernel is enqueued: released here, a dust_create of another thread
  //  could count itself in and drain the device between this test and the launch, and the two grids would meet unchained - ADVICE r4)
  if (g_live_ctx[dev].load() < 2) return;
  on = true;
  if (!g_persist_ev[dev] && hipEventCreateWithFlags(&g_persist_ev[dev], hipEventDisableTiming) != hipSuccess) {
    g_persist_ev[dev] = nullptr;
    (void)hipGetLastError();
    return;
  }
  if (g_persist_last[dev] && g_persist_last[dev] != c->stream) (void)hipStreamWaitEvent(c->stream, g_persist_ev[dev], 0);
}
PersistChain::~PersistChain() {
  if (on && g_persist_ev[dev]) {
    (void)hipEventRecord(g_persist_ev[dev], c->stream);
    g_persist_last[dev] = c->stream;
  }
  if (locked) g_persist_mu[dev].unlock();
}

static int create_impl(const dust_config *cfg, dust_ctx **out) {
  TRY(validate(cfg));
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
    return fail(DUST_ERR_NO_DEVICE, "no HIP device: libdust_amd has no CPU fallback");
  if (cfg->device < 0 || cfg->device >= ndev) return fail(DUST_ERR_NO_DEVICE, "device %d of %d", cfg->device, ndev);
  HIP_TRY(hipSetDevice(cfg->device));
  dust_ctx *c = new (std::nothrow) dust_ctx();
  if (!c) return fail(DUST_ERR_HIP, "out of host memory");
  memset((void *)c, 0, sizeof *c);
  env_read(c);
  c->t2_queue = new (std::nothrow) std::vector<dust_ctx::T2Replay>();
  if (!c->t2_queue) {
    delete c;
    return fail(DUST_ERR_HIP, "out of host memory");
  }
  c->cfg = *cfg;
  c->N = cfg->n_policies;
  c->S = cfg->n_samples;
  c->M = cfg->n_params;
  c->H = cfg->horizon;
  c->da = cfg->dim_a;
  c->ds = cfg->dim_s;
  if (cfg->model == DUST_MODEL_SKID_STEER) {  // SkidSteerRobot.__init__ defaults (skid_steer_robot.py:19-28); unit state weights, no goal
    memset(&c->skid, 0, sizeof c->skid);
    c->skid.x_icr = DevParam{DUST_PARAM_PYFLOAT, 0, 0.2};
    c->skid.wheel_radius = DevParam{DUST_PARAM_PYFLOAT, 0, 0.0625};
    c->skid.axial_distance = DevParam{DUST_PARAM_PYFLOAT, 0, 0.475};
    for (int d = 0; d < 2; ++d) {
      c->skid.lo[d] = -0.5f;
      c->skid.hi[d] = 0.5f;
    }
    for (int k = 0; k < 5; ++k) c->skid.w_state[k] = c->skid.w_term[k] = 1.0f;
  }
  c->P = cfg->dim_p > 0 ? cfg->dim_p : 1;
  c->D = c->H * c->da;
  c->n0 = cfg->shard_offset;
  c->nloc = cfg->shard_size > 0 ? cfg->shard_size : c->N;
  *out = c;
  HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  c->own_stream = true;
  HIP_TRY(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
  HIP_TRY(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&c->ev_order, hipEventDisableTiming));
  c->pair_stream = c->stream;
  HIP_TRY(hipEventCreate(&c->ev0));
  HIP_TRY(hipEventCreate(&c->ev1));
  const size_t ND = (size_t)c->N * c->D, SN = (size_t)c->S * c->N;
  float **nd[] = {&c->theta, &c->thetaT, &c->mu, &c->muT, &c->a_mat, &c->grad_lik, &c->grad_pri, &c->score, &c->phi};
  for (auto p : nd) {
    TRY(dalloc(p, ND));
    HIP_TRY(hipMemsetAsync(*p, 0, ND * sizeof(float), c->stream));
  }
  TRY(dalloc(&c->theta_alt, ND));
  if (cfg->kernel == DUST_KERNEL_K2_IIDMP || cfg->kernel == DUST_KERNEL_K2_SHARED) TRY(dalloc(&c->thetaT_alt, ND));
  c->theta_home = c->theta;
  float **nn[] = {&c->logmix, &c->mixw, &c->a_mix, &c->eta, &c->logl, &c->logp, &c->lw};
  for (auto p : nn) {
    TRY(dalloc(p, (size_t)c->N));
    HIP_TRY(hipMemsetAsync(*p, 0, c->N * sizeof(float), c->stream));
  }
  TRY(dalloc(&c->a_seq, (size_t)c->D));
  {
    const size_t dpad = ((size_t)c->D + 31) & ~(size_t)31;
    c->out_floats = dpad + (size_t)c->N + 32;
    TRY(dalloc(&c->outblk, c->out_floats));
    HIP_TRY(hipMemsetAsync(c->outblk, 0, c->out_floats * sizeof(float), c->stream));
    c->a_seq_out = c->outblk;
    c->pw = c->outblk + dpad;
    HIP_TRY(hipHostMalloc((void **)&c->out_pinned, (c->out_floats + 4) * sizeof(float), hipHostMallocDefault));
  }
  TRY(dalloc(&c->bw, (size_t)c->D));
  HIP_TRY(hipMemsetAsync(c->bw, 0, c->D * sizeof(float), c->stream));  // (no previous bandwidth: the sorted kernels' warm start stands down)
  HIP_TRY(hipMemsetAsync(c->a_seq, 0, c->D * sizeof(float), c->stream));
  TRY(dalloc(&c->costsT, SN));
  TRY(dalloc(&c->omegaT, SN));
  TRY(dalloc(&c->tmp, SN));
  c->tmp_cap = SN;
  TRY(dalloc(&c->costs_stage, SN));
  TRY(dalloc(&c->state_dev, (size_t)8));
  TRY(dalloc(&c->istar, (size_t)1));
  TRY(dalloc(&c->ctr_dev, (size_t)4));
  HIP_TRY(hipMemsetAsync(c->ctr_dev, 0, 4 * sizeof(uint32_t), c->stream));
  if (cfg->optimizer == DUST_OPT_ADAM) {
    TRY(dalloc(&c->adam_m, ND));
    TRY(dalloc(&c->adam_v, ND));
    HIP_TRY(hipMemsetAsync(c->adam_m, 0, ND * sizeof(float), c->stream));
    HIP_TRY(hipMemsetAsync(c->adam_v, 0, ND * sizeof(float), c->stream));
  }
  // a_mix starts as ones (disco.py:110); uniform prior weights until set_prior
  std::vector<float> ones((size_t)c->N, 1.0f);
  HIP_TRY(hipMemcpyAsync(c->a_mix, ones.data(), c->N * sizeof(float), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->mixw, ones.data(), c->N * sizeof(float), hipMemcpyHostToDevice, c->stream));
  logmix_kernel<<<1, 1024, 0, c->stream>>>(c->mixw, c->logmix, c->N);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(c->stream));
  return DUST_OK;
}

extern "C" int dust_create(const dust_config *cfg, dust_ctx **out) {
  if (!out) return fail(DUST_ERR_INVALID, "null out");
  *out = nullptr;
  if (cfg) serve_cancel_device(cfg->device);  // (an armed launch of another context holds every CU until its state arrives)
  int s = create_impl(cfg, out);
  if (s == DUST_OK && cfg->device >= 0 && cfg->device < DUST_MAX_DEV) {
    // counted in under the device's mutex (PersistChain tests the count under it, and holds it across its launch): a one-launch kernel
    // is either enqueued before this point - and drained here - or sees the second tenant and is chained
    std::lock_guard<std::mutex> lk(g_persist_mu[cfg->device]);
    if (g_live_ctx[cfg->device].fetch_add(1) + 1 == 2) (void)hipDeviceSynchronize();
  }
  if (s != DUST_OK && *out) {
    std::string keep = g_err;
    free_all(*out);
    delete *out;
    *out = nullptr;
    g_err = keep;
  }
  return s;
}

static void graph_drop(dust_ctx *c);
static int try_persistent(dust_ctx *c, const float *state, int n_steps, const float *eps, const float *params, int flags, bool do_forward,
                          bool *done);
static int tick_outputs(dust_ctx *c, float *a_seq, float *p_weights);
static int t2_settle(dust_ctx *c, unsigned int aborts_now, bool *replayed);
static void t2_queue_push(dust_ctx *c, const float *state4, int steps, bool fwd, bool replayable, bool mu_aliased);

// A wait inside a launch gave up.  The tick it belonged to is lost (its results are invalid and the particles may be partly updated);
// what can be saved is the future: no kernel of this context spins on its own grid any more (handoff.hpp DUST_SPIN_TIMEOUT_TICKS).
static void handoff_ban(dust_ctx *c) {
  c->handoff_banned = true;
  c->persist_declined = 0;
  if (c->graph_exec || c->graph_exec_alt) graph_drop(c);  // (a captured tick replays the fused launches)
}
// A wait of the owner-computes kernel gave up: clear the report, take the context off the spinning kernels, and tell whether the tick
// is whole - committed by every workgroup or by none (tick2.hpp t2_commit; a tick nobody committed is replayed by t2_settle).
#define T2_TORN_MSG "persistent tick kernel: a hand-off wait timed out while some workgroups were committing (results of this tick are invalid: re-seed the particles); the device seems to be shared with another process: this context runs plain kernels from here on"
static int t2_timed_out(dust_ctx *c, bool *whole) {
  unsigned int w[3] = {0u, 0u, 0u};
  HIP_TRY(hipMemcpy(w, c->outblk + c->out_floats - 32, sizeof w, hipMemcpyDeviceToHost));
  const unsigned int zero[3] = {0u, w[1], 0u};
  HIP_TRY(hipMemcpy(c->outblk + c->out_floats - 32, zero, sizeof zero, hipMemcpyHostToDevice));
  handoff_ban(c);
  *whole = c->t2_grid > 0 && (w[2] % (unsigned int)c->t2_grid) == 0u;
  return DUST_OK;
}
static int handoff_timeout(dust_ctx *c, const char *msg) {
  handoff_ban(c);
  return fail(DUST_ERR_HIP, "%s", msg);
}

extern "C" int dust_sync(dust_ctx *c) {
  if (!c) return fail(DUST_ERR_INVALID, "null ctx");
  serve_cancel(c);  // (an armed launch would wait for its plant state - and this synchronisation for it)
  c->cancel_unsettled = false;
  bool have_flag = false;
  unsigned int flag0 = 0u;
  if (c->t2_inflight) {  // owner-computes ticks that did not start (device shared with another context) run now, on the other path
    // (the status words ride behind the launches into pinned memory: one stream synchronisation instead of a synchronisation and two
    //  blocking copies of a few bytes - 128 -> 110 us per tick for a caller that synchronises after every open-loop tick)
    unsigned int *w = reinterpret_cast<unsigned int *>(c->out_pinned + c->out_floats);
    HIP_TRY(hipMemcpyAsync(w, c->outblk + c->out_floats - 32, 2 * sizeof(unsigned int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    have_flag = true;
    flag0 = w[0];
    if (w[0] && c->t2_transactional) {  // ... and so do ticks whose waits gave up: tick2.hpp commits nothing then (w[1] counts them too)
      bool whole = false;
      TRY(t2_timed_out(c, &whole));  // (reads and clears the words)
      if (!whole) return fail(DUST_ERR_HIP, "%s", T2_TORN_MSG);
      flag0 = 0u;
    }
    bool replayed = false;
    TRY(t2_settle(c, w[1], &replayed));
    if (replayed) {
      HIP_TRY(hipStreamSynchronize(c->stream));
      have_flag = false;  // (the replay's own launches may have raised it)
    }
  } else {
    HIP_TRY(hipStreamSynchronize(c->stream));
  }
  if (c->fused_cnt) {  // bounded spin of the fused launch's in-kernel hand-off: report instead of hanging
    unsigned int flag = 0;
    HIP_TRY(hipMemcpy(&flag, c->fused_cnt + (size_t)c->fused_tiles * CNT_STRIDE, sizeof flag, hipMemcpyDeviceToHost));
    if (flag) return fail(DUST_ERR_HIP, "fused prior+rollout launch: hand-off spin timed out (results of that tick are invalid)");
  }
  if (c->stein_cnt) {
    unsigned int flag = 0;
    HIP_TRY(hipMemcpy(&flag, c->stein_cnt + ((size_t)c->stein_tiles + 1) * CNT_STRIDE, sizeof flag, hipMemcpyDeviceToHost));
    if (flag) return handoff_timeout(c, "Stein+update launch: hand-off spin timed out (results of that tick are invalid); the device seems to be shared with another process: this context runs plain kernels from here on");
  }
  if (c->iter_cnt) {
    unsigned int flag = 0;
    HIP_TRY(hipMemcpy(&flag, c->iter_cnt + (size_t)2 * (2 * c->iter_tiles + c->iter_js) * CNT_STRIDE, sizeof flag, hipMemcpyDeviceToHost));
    if (flag) return handoff_timeout(c, "one-launch SVGD iteration: hand-off spin timed out (results of that tick are invalid); the device seems to be shared with another process: this context runs plain kernels from here on");
  }
  if (c->t2_cnt) {
    unsigned int flag = flag0;
    if (!have_flag) HIP_TRY(hipMemcpy(&flag, c->outblk + c->out_floats - 32, sizeof flag, hipMemcpyDeviceToHost));
    if (flag) {
      HIP_TRY(hipMemset(c->outblk + c->out_floats - 32, 0, sizeof flag));  // reported: clear
      return handoff_timeout(c, "persistent tick kernel: a hand-off wait timed out (results of that tick are invalid); the device seems to be shared with another process: this context runs plain kernels from here on");
    }
  }
  return peer_check(c);
}

extern "C" int dust_get_config(const dust_ctx *c, dust_config *out) {
  if (!c || !out) return fail(DUST_ERR_INVALID, "null argument");
  *out = c->cfg;
  return DUST_OK;
}

extern "C" int dust_set_stream(dust_ctx *c, void *s) {
  if (c) c->persist_declined = 0;  // (what the one-launch ticks declined may be eligible now - or the other way round)
  if (!c) return fail(DUST_ERR_INVALID, "null ctx");
  if (c->t2_inflight) TRY(dust_sync(c));  // (settle one-launch ticks on the stream they were enqueued on, replays included)
  if (c->graph_exec || c->graph_exec_alt) graph_drop(c);
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (c->own_stream) HIP_TRY(hipStreamDestroy(c->stream));
  c->stream = (hipStream_t)s;
  c->pair_stream = c->stream;
  c->own_stream = false;
  return DUST_OK;
}

static int d2d(dust_ctx *c, void *dst, const void *src, size_t bytes) {
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->stream));
  return DUST_OK;
}
// A one-launch tick enqueued earlier may not have started, or not committed (device shared): it - and every one-launch tick enqueued
// behind it, which then aborted as well (`expect_aborts`) - is replayed when the pending launches are SETTLED (dust_sync: stream
// synchronisation, status words, t2_settle).  Everything that reads or changes the particle / prior / optimiser state from the host, or
// runs plain kernels on it, settles first (a forward on un-optimised particles, getters ahead of the replay); the one-launch
// ticks themselves need no host round trip for that - the device-side chain keeps them in order.
static int settle_pending(dust_ctx *c) { return (c && (c->t2_inflight || c->armed || c->cancel_unsettled)) ? dust_sync(c) : DUST_OK; }

static int h2d(dust_ctx *c, void *dst, const void *src, size_t bytes) {
  TRY(settle_pending(c));
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));  // the caller owns (and may free/reuse) the host buffer
  return DUST_OK;
}
static int d2h(dust_ctx *c, void *dst, const void *src, size_t bytes) {
  TRY(settle_pending(c));
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return DUST_OK;
}

extern "C" int dust_clone(const dust_ctx *src, dust_ctx **out) {
  if (!src || !out) return fail(DUST_ERR_INVALID, "null argument");
  TRY(settle_pending(const_cast<dust_ctx *>(src)));
  TRY(dust_create(&src->cfg, out));
  dust_ctx *c = *out;
  HIP_TRY(hipStreamSynchronize(src->stream));
  const size_t ND = (size_t)c->N * c->D * sizeof(float), Nb = c->N * sizeof(float), SN = (size_t)c->S * c->N * sizeof(float);
  struct {
    float *dst;
    const float *s;
    size_t b;
  } cp[] = {{c->theta, src->theta, ND}, {c->thetaT, src->thetaT, ND}, {c->mu, src->mu, ND}, {c->muT, src->muT, ND},
            {c->a_mat, src->a_mat, ND}, {c->grad_lik, src->grad_lik, ND}, {c->score, src->score, ND}, {c->phi, src->phi, ND},
            {c->logmix, src->logmix, Nb}, {c->mixw, src->mixw, Nb}, {c->a_mix, src->a_mix, Nb}, {c->eta, src->eta, Nb},
            {c->logl, src->logl, Nb}, {c->a_seq, src->a_seq, c->D * sizeof(float)}, {c->costsT, src->costsT, SN}};
  for (auto &e : cp) TRY(d2d(c, e.dst, e.s, e.b));
  if (src->adam_m) {
    TRY(d2d(c, c->adam_m, src->adam_m, ND));
    TRY(d2d(c, c->adam_v, src->adam_v, ND));
  }
  TRY(d2d(c, c->ctr_dev, src->ctr_dev, 4 * sizeof(uint32_t)));
  c->mu_aliased = src->mu_aliased;
  c->have_sample = src->have_sample;
  c->skid = src->skid;
  c->k2_fixed_h = src->k2_fixed_h;  // (dust_set_k2_bandwidth: RBF(bandwidth=, minimum_bw=) - a deep-copied controller keeps its kernel)
  c->k2_min_bw = src->k2_min_bw;
  if (src->cz_dev && src->cz_sets > 0) {  // recorded control-noise draws not consumed yet (dust_set_ctrl_noise)
    const size_t n = (size_t)src->cz_sets * c->H * c->M * c->S * c->N * 2;
    TRY(ensure(&c->cz_dev, &c->cz_cap, n));
    TRY(d2d(c, c->cz_dev, src->cz_dev, n * sizeof(float)));
    c->cz_sets = src->cz_sets;
    c->cz_next = src->cz_next;
  }
  if (src->grid_bits) {
    const size_t words = ((size_t)src->nx * src->ny + 31) / 32;
    TRY(dalloc(&c->grid_bits, words));
    TRY(d2d(c, c->grid_bits, src->grid_bits, words * 4));
    c->nx = src->nx;
    c->ny = src->ny;
    c->off_x = src->off_x;
    c->off_y = src->off_y;
  }
  if (src->pk_perm && src->pk_lead && src->pk_perm_cap >= (size_t)c->nloc && src->pk_lead_cap >= (size_t)c->nloc) {
    // the tile order of the large-set passes and the leaders noted for the next one (pairwise_packed.hpp): a copy continues with the
    // same run lists - the same grouping of every sum - as its source
    if (src->stream2) HIP_TRY(hipStreamSynchronize(src->stream2));  // (the side stream's query_order_kernel)
    TRY(ensure(&c->pk_perm, &c->pk_perm_cap, (size_t)c->nloc));
    TRY(ensure(&c->pk_lead, &c->pk_lead_cap, (size_t)c->nloc));
    TRY(d2d(c, c->pk_perm, src->pk_perm, (size_t)c->nloc * sizeof(int)));
    TRY(d2d(c, c->pk_lead, src->pk_lead, (size_t)c->nloc * sizeof(int)));
  }
  if (src->mw_dev) {
    TRY(dalloc(&c->mw_dev, (size_t)c->M));
    TRY(d2d(c, c->mw_dev, src->mw_dev, (size_t)c->M * sizeof(float)));
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  return DUST_OK;
}

extern "C" int dust_set_model_param(dust_ctx *c, const char *name, double value, int kind) {
  if (!c || !name) return fail(DUST_ERR_INVALID, "null argument");
  TRY(settle_pending(c));  // (a one-launch tick that did not start is replayed with the dynamics it was enqueued with)
  if (c->graph_exec || c->graph_exec_alt) graph_drop(c);
  dust_param *p = nullptr;
  if (!strcmp(name, "g")) p = &c->cfg.g;
  else if (!strcmp(name, "mass")) p = &c->cfg.mass;
  else if (!strcmp(name, "length")) p = &c->cfg.length;
  else return fail(DUST_ERR_INVALID, "unknown model parameter '%s'", name);
  p->value = value;
  if (kind >= 0 && p->kind != DUST_PARAM_SAMPLED) p->kind = kind;
  return DUST_OK;
}

extern "C" int dust_set_param_weights(dust_ctx *c, const float *w) {
  if (c) c->persist_declined = 0;  // (what the one-launch ticks declined may be eligible now - or the other way round)
  if (!c) return fail(DUST_ERR_INVALID, "null ctx");
  HIP_TRY(hipSetDevice(c->cfg.device));
  TRY(settle_pending(c));  // (... and with the weights it was enqueued with)
  if (c->graph_exec || c->graph_exec_alt) graph_drop(c);
  if (!w) {
    if (c->mw_dev) {
      HIP_TRY(hipStreamSynchronize(c->stream));
      HIP_TRY(hipFree(c->mw_dev));
      c->mw_dev = nullptr;
    }
    return DUST_OK;
  }
  if (c->cfg.dim_p <= 0) return fail(DUST_ERR_STATE, "parameter weights need sampled parameters (dim_p > 0)");
  if (!c->mw_dev) TRY(dalloc(&c->mw_dev, (size_t)c->M));
  return h2d(c, c->mw_dev, w, (size_t)c->M * sizeof(float));
}

extern "C" int dust_set_ctrl_noise(dust_ctx *c, const float *z, int n_sets) {
  if (!c) return fail(DUST_ERR_INVALID, "null ctx");
  if (c->cfg.model != DUST_MODEL_PARTICLE || !c->cfg.ctrl_noise)
    return fail(DUST_ERR_STATE, "control noise belongs to a Particle(deterministic=False) context (dust_config.ctrl_noise)");
  TRY(settle_pending(c));
  if (c->graph_exec || c->graph_exec_alt) graph_drop(c);
  c->cz_sets = c->cz_next = 0;
  if (!z || n_sets <= 0) return DUST_OK;
  HIP_TRY(hipSetDevice(c->cfg.device));
  const size_t n = (size_t)n_sets * c->H * c->M * c->S * c->N * 2;
  TRY(ensure(&c->cz_dev, &c->cz_cap, n));
  TRY(h2d(c, c->cz_dev, z, n * sizeof(float)));
  c->cz_sets = n_sets;
  return DUST_OK;
}

extern "C" int dust_set_grid(dust_ctx *c, const float *grid, int nx, int ny, float off_x, float off_y) {
  if (c) c->persist_declined = 0;  // (what the one-launch ticks declined may be eligible now - or the other way round)
  if (!c || !grid || nx < 1 || ny < 1) return fail(DUST_ERR_INVALID, "bad grid");
  TRY(settle_pending(c));  // (before the old grid is freed: an armed launch may hold the device, a tick that did not start is replayed on the map it was enqueued with)
  if (c->graph_exec || c->graph_exec_alt) graph_drop(c);
  const size_t cells = (size_t)nx * ny, words = (cells + 31) / 32;
  std::vector<uint32_t> bits(words, 0u);
  for (size_t i = 0; i < cells; ++i) {
    if (grid[i] == 1.0f) bits[i >> 5] |= 1u << (i & 31);
    else if (grid[i] != 0.0f) return fail(DUST_ERR_UNSUPPORTED, "occupancy grid must be binary (ObstacleMap writes 0/1)");
  }
  HIP_TRY(hipSetDevice(c->cfg.device));
  if (c->grid_bits) HIP_TRY(hipFree(c->grid_bits));
  c->grid_bits = nullptr;
  TRY(dalloc(&c->grid_bits, words));
  TRY(h2d(c, c->grid_bits, bits.data(), words * 4));
  c->nx = nx;
  c->ny = ny;
  c->off_x = off_x;
  c->off_y = off_y;
  return DUST_OK;
}

static int launch_transpose(dust_ctx *c, const float *src, float *dst, int N, int D) {
  const int n = N * D;
  transpose_kernel<<<(n + 255) / 256, 256, 0, c->stream>>>(src, dst, N, D);
  HIP_TRY(hipGetLastError());
  return DUST_OK;
}

extern "C" int dust_set_theta(dust_ctx *c, const float *theta) {
  if (!c || !theta) return fail(DUST_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(c->cfg.device));
  TRY(h2d(c, c->theta, theta, (size_t)c->N * c->D * sizeof(float)));
  c->kmat_valid = false;
  if (c->adam_m) {
    HIP_TRY(hipMemsetAsync(c->adam_m, 0, (size_t)c->N * c->D * sizeof(float), c->stream));
    HIP_TRY(hipMemsetAsync(c->adam_v, 0, (size_t)c->N * c->D * sizeof(float), c->stream));
    HIP_TRY(hipMemsetAsync(c->ctr_dev + 2, 0, sizeof(uint32_t), c->stream));
  }
  return DUST_OK;
}
// RBF(bandwidth=, minimum_bw=) base_kernels.py:44-92 for the K2 kernels: bandwidth < 0 = the median trick (default); otherwise
// h = clip(bw_scale * bandwidth^2 / log(N + 1), minimum_bw), evaluated here in double as the reference's Python floats are,
// then used as an fp32 scalar by the tensor ops (K = exp(-d2 / h), dK = K (x - y) 2 / h).
extern "C" int dust_set_k2_bandwidth(dust_ctx *c, float bandwidth, float minimum_bw) {
  if (c) c->persist_declined = 0;  // (what the one-launch ticks declined may be eligible now - or the other way round)
  if (!c) return fail(DUST_ERR_INVALID, "null ctx");
  if (c->cfg.kernel != DUST_KERNEL_K2_IIDMP && c->cfg.kernel != DUST_KERNEL_K2_SHARED)
    return fail(DUST_ERR_STATE, "dust_set_k2_bandwidth: the context's kernel is not iid_mp(RBF)");
  if (!(minimum_bw > 0.f)) return fail(DUST_ERR_INVALID, "minimum_bw must be > 0");
  TRY(settle_pending(c));
  if (bandwidth < 0.f) {  // the median trick, clamped at minimum_bw (base_kernels.py:83-89)
    c->k2_fixed_h = 0.f;
    c->k2_min_bw = minimum_bw;
    if (c->graph_exec || c->graph_exec_alt) graph_drop(c);
    return DUST_OK;
  }
  double h = (double)bandwidth * (double)bandwidth;
  h = h / log((double)c->N + 1.0);
  h = (double)c->cfg.bw_scale * h;
  if (h < (double)minimum_bw) h = (double)minimum_bw;
  c->k2_fixed_h = (float)h;
  if (!(c->k2_fixed_h > 0.f)) return fail(DUST_ERR_INVALID, "bandwidth and minimum_bw give h = 0");
  if (c->graph_exec || c->graph_exec_alt) graph_drop(c);
  return DUST_OK;
}

extern "C" int dust_get_theta(dust_ctx *c, float *theta) {
  if (!c || !theta) return fail(DUST_ERR_INVALID, "null argument");
  return d2h(c, theta, c->theta, (size_t)c->N * c->D * sizeof(float));
}
extern "C" int dust_set_prior(dust_ctx *c, const float *means, const float *w) {
  if (!c || !means) return fail(DUST_ERR_INVALID, "null argument");
  if (c->graph_exec || c->graph_exec_alt) graph_drop(c);
  HIP_TRY(hipSetDevice(c->cfg.device));
  TRY(h2d(c, c->mu, means, (size_t)c->N * c->D * sizeof(float)));
  c->mu_aliased = false;
  if (w) {
    for (int i = 0; i < c->N; ++i)
      if (!(w[i] >= 0.f)) return fail(DUST_ERR_INVALID, "mixture weights must be >= 0 (torch.distributions.Categorical)");
    TRY(h2d(c, c->mixw, w, c->N * sizeof(float)));
  } else {
    std::vector<float> ones((size_t)c->N, 1.0f);
    TRY(h2d(c, c->mixw, ones.data(), c->N * sizeof(float)));
  }
  logmix_kernel<<<1, 1024, 0, c->stream>>>(c->mixw, c->logmix, c->N);
  HIP_TRY(hipGetLastError());
  return DUST_OK;
}
extern "C" int dust_get_prior(dust_ctx *c, float *means, float *probs) {
  if (!c) return fail(DUST_ERR_INVALID, "null ctx");
  if (means) TRY(d2h(c, means, c->mu_aliased ? c->theta : c->mu, (size_t)c->N * c->D * sizeof(float)));
  if (probs) {
    std::vector<float> w((size_t)c->N);
    TRY(d2h(c, w.data(), c->mixw, c->N * sizeof(float)));
    double s = 0;
    for (float v : w) s += v;
    for (int i = 0; i < c->N; ++i) probs[i] = w[i] / (float)s;
  }
  return DUST_OK;
}
extern "C" int dust_set_a_mat(dust_ctx *c, const float *a) {
  if (!c || !a) return fail(DUST_ERR_INVALID, "null argument");
  return h2d(c, c->a_mat, a, (size_t)c->N * c->D * sizeof(float));
}
extern "C" int dust_get_a_mat(dust_ctx *c, float *a) {
  if (!c || !a) return fail(DUST_ERR_INVALID, "null argument");
  return d2h(c, a, c->a_mat, (size_t)c->N * c->D * sizeof(float));
}
extern "C" int dust_get_a_mix(dust_ctx *c, float *a) {
  if (!c || !a) return fail(DUST_ERR_INVALID, "null argument");
  TRY(settle_pending(c));
  if (c->have_sample) {
    amix_kernel<<<1, 1024, 0, c->stream>>>(c->eta, c->a_mix, c->N);
    HIP_TRY(hipGetLastError());
  }
  return d2h(c, a, c->a_mix, c->N * sizeof(float));
}
extern "C" int dust_set_a_seq(dust_ctx *c, const float *a) {
  if (!c || !a) return fail(DUST_ERR_INVALID, "null argument");
  return h2d(c, c->a_seq, a, c->D * sizeof(float));
}
extern "C" int dust_get_a_seq(dust_ctx *c, float *a) {
  if (!c || !a) return fail(DUST_ERR_INVALID, "null argument");
  return d2h(c, a, c->a_seq, c->D * sizeof(float));
}

// ---------------------------------------------------------------------------------------------------------------
// kernel A
struct SampleOpts {
  int noise_mode;
  const float *noise_dev;  // device pointer [S][N][D] (eps or actions) or nullptr
  const float *base;       // theta or a_mat
  int eps_base_mode;
  int update_a_mat;
  const float *costs_in;   // device [S][N]: skip the rollouts and use these costs (stage-wise phi)
  bool costs_own;          // costs_in came from a first pass of this very sample (stored-states / skid-steer forms)
  bool want_actions, want_states, want_omega;
  bool store_f16;          // states / actions are stored as binary16 (DUST_STORE_F16)
  int merge_prior;         // a prior pass ran just before: fold its partials into grad_pri / score
  int bump_adam;           // an optimiser step follows this sample
};

// geometry of the tiled pairwise launches: i-tiles of PAIR_TI queries x JS key slices, >= ~512 workgroups when possible
// Large key sets take the register-blocked kernel of pairwise_big.hpp (its tile is TQ = 4096 / DPB queries).
// Measured (round 1, N = 16384 / 4096, D = 30): Stein 1107 / 82 us vs 1730 / 115 us, prior 1367 / 99 vs 1799 / 119 us; D = 40:
// 5-15 % faster; D = 80 (DPB = 128, 32-query tiles): 1630 / 1394 us vs 1060 / 905 us at N = 8192, so D > 64 keeps the 32 x 64 kernel.  Sharded
// contexts (few query tiles per rank) keep it too: it allows JS <= 16 and with that the fused launches.
// Round 2: once the prior means alias theta (every tick after the first) large sets take the fused pair of launches of
// pairwise_fused.hpp - one distance pass for prior + repulsion + the Gram matrix, then Gram x score as a GEMM - for D <= 80.
static int pair_dpb(int D) { return D <= 32 ? 32 : 64; }
static int fused_dpb(int D) { return D <= 32 ? 32 : (D <= 64 ? 64 : 80); }
static int fused_tq(int D) { return D <= 32 ? FusedGeom<32>::TQ : (D <= 64 ? FusedGeom<64>::TQ : FusedGeom<80>::TQ); }
// "large key set": none of the small-N launch fusions (fused.hpp, persist.hpp) applies
static bool pair_is_big(const dust_ctx *c) {
  // development switch DUST_PAIR_BIG: 0 forces the 32 x 64 kernel, 1 the large-set path from N = 2048 on
  if (c->env.pair_big == 0) return false;
  const bool forced = c->env.pair_big == 1;
  // (a rank of a sharded run qualifies from 512 local particles on: the fused pass slices the keys finely enough to fill the chip)
  // Just above the threshold the 32 x 64 kernels inside the fused launches (2 launches per iteration) still win while the rows are
  // short - measured, us per 5-iteration tick, large-set / small-set path: Pendulum N = 2048 D = 30: 434 / 355 (M = 8, cfg5: 742 / 620);
  // Particle N = 2048 D = 32: 467 / 376, D = 80: 946 / 1129; Pendulum N = 3072 D = 30: 722 / 725; N = 4096: 871 / 1044.
  if (!forced && c->nloc == c->N && (long)c->N * c->D <= 65536) return false;
  return c->N >= 2048 && c->D <= 80 && (c->nloc == c->N || c->nloc >= 512);
}
static bool pair_big_kernel(const dust_ctx *c) { return pair_is_big(c) && c->D <= 64 && c->nloc == c->N; }  // pairwise_big.hpp (unfused passes)
static bool pair_fused_ok(const dust_ctx *c) {
  if (c->cfg.full_cov) return false;  // (one distance serves prior and Stein kernel there: not with a prior metric of its own)
  if (c->env.pair_fused == 0) return false;  // development switch DUST_PAIR_FUSED: 0 keeps the two unfused passes
  return pair_is_big(c) && c->mu_aliased && (c->cfg.kernel == DUST_KERNEL_K1_RBF || c->cfg.kernel == DUST_KERNEL_IMQ);
}
// key slices: `tiles` is the query-tile count of the PRIMARY kernel of the current state; a launcher whose kernel has another
// tile size recomputes it (the partial-output layout [js][n_local][ldp] does not depend on the tile size)
static int device_cus(const dust_ctx *c) {
  static int n_cu = 0;
  if (!n_cu && hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, c->cfg.device) != hipSuccess) n_cu = 256;
  return n_cu;
}
static void pair_geometry(const dust_ctx *c, int *tiles, int *JS, int *slice) {
  const bool fz = pair_fused_ok(c);
  const int ti = fz ? fused_tq(c->D) : (pair_big_kernel(c) ? 4096 / pair_dpb(c->D) : PAIR_TI);
  *tiles = (c->nloc + ti - 1) / ti;
  const int chunks = (c->N + PAIR_JC - 1) / PAIR_JC;
  int js = (512 + *tiles - 1) / *tiles;
  js = std::max(1, std::min(js, chunks));
  if (fz) {
    // (the regular grid of the log-p-only kernel; pairwise_packed_kernel itself splits the work by fused_geometry below)
    // three workgroups per CU stay resident: pick the slice count whose grid fills whole rounds best (a small penalty per slice:
    // every slice repeats the tile prologue and adds a row of partial outputs)
    double best = -1.0;
    for (int cand = 1; cand <= std::min(64, chunks); ++cand) {
      const int cps = (chunks + cand - 1) / cand, real = (chunks + cps - 1) / cps;
      const long wgs = (long)*tiles * real, rounds = (wgs + 767) / 768;
      const double fill = (double)wgs / (double)(rounds * 768) - 2e-3 * real;  // (ties: fewer slices)
      if (fill > best) {
        best = fill;
        js = real;
      }
    }
  }
  const int cps = (chunks + js - 1) / js;  // chunks per slice
  *slice = cps * PAIR_JC;
  *JS = (c->N + *slice - 1) / *slice;
}

static int cpt_for(int D) { return D <= 32 ? 4 : (D <= 64 ? 8 : (D <= 96 ? 12 : 16)); }  // pass-B columns per lane

static PriorMerge prior_merge_args(const dust_ctx *c) {
  PriorMerge pm;
  memset(&pm, 0, sizeof pm);
  int tiles, slice;
  pair_geometry(c, &tiles, &pm.JS, &slice);
  if (c->prior_js) pm.JS = c->prior_js;  // (the partials came from pairwise_packed_kernel: its own slice count)
  pm.n_local = c->nloc;
  pm.ldp = 8 * cpt_for(c->D);
  pm.pA = c->pA;
  pm.pM = c->pM;
  pm.pL = c->pL;
  double logdet = 0;
  for (int d = 0; d < c->da; ++d) {
    pm.inv_s2[d] = 1.0f / (c->cfg.sigma_p[d] * c->cfg.sigma_p[d]);
    logdet += log((double)c->cfg.sigma_p[d]);
  }
  if (full_cov(c)) {  // the pass runs on whitened rows (prior_args): unit scale there, log det L_p in the constant
    pm.full = 1;
    for (int k = 0; k < 3; ++k) pm.Lp[k] = c->cfg.chol_p[k];
    pm.inv_s2[0] = pm.inv_s2[1] = 1.0f;
    logdet = log((double)c->cfg.chol_p[0]) + log((double)c->cfg.chol_p[2]);
  }
  pm.log_norm = (float)(-c->H * logdet - 0.5 * c->D * log(2.0 * M_PI));
  return pm;
}

static int rollout_args(dust_ctx *c, const SampleOpts &o, RolloutArgs &a, int *nt_out, size_t *lds_out) {
  memset(&a, 0, sizeof a);
  a.dm = make_dev_model(c);
  if (c->cfg.model == DUST_MODEL_PARTICLE && c->cfg.with_obstacle && !c->grid_bits)
    return fail(DUST_ERR_STATE, "with_obstacle is set but no occupancy grid was supplied (dust_set_grid)");
  a.N_total = c->N;
  a.n0 = c->n0;
  a.S = c->S;
  a.M = c->M;
  a.H = c->H;
  a.D = c->D;
  a.magicD = (uint32_t)((1ull << 32) / (uint64_t)c->D) + 1u;
  a.merge_prior = o.merge_prior;
  if (o.merge_prior) {
    a.pm = prior_merge_args(c);
    a.grad_pri = c->grad_pri;
    a.score = c->score;
  }
  a.noise_mode = o.noise_mode;
  a.noise_f16 = (o.noise_mode != NOISE_PHILOX && c->noise_f16) ? 1 : 0;
  a.store_f16 = o.store_f16 ? 1 : 0;
  a.lik = c->cfg.likelihood;
  a.eps_base_mode = o.eps_base_mode;
  a.update_a_mat = o.update_a_mat;
  a.alpha = c->cfg.alpha;
  a.temp = c->cfg.temperature;
  a.a_reg = c->cfg.a_reg;
  for (int d = 0; d < 4; ++d) {
    a.chol_a[d] = c->cfg.chol_a[d];
    a.sigma_a[d] = c->cfg.sigma_a[d];
    a.a_pre[d] = c->cfg.a_pre[d];
  }
  a.full_cov = c->cfg.full_cov ? 1 : 0;
  a.chol_off = c->cfg.full_cov ? c->cfg.chol_a_off : 0.f;
  a.a_pre_off = c->cfg.full_cov ? c->cfg.a_pre_off : 0.f;
  a.state = c->state_dev;
  a.theta = o.base;
  a.noise = o.noise_dev;
  a.params = (c->cfg.dim_p > 0 && c->params_dev) ? c->params_dev : nullptr;
  a.mw = c->mw_dev;
  a.a_seq = c->a_seq;
  a.a_mat = c->a_mat;
  a.costsT = c->costsT;
  a.costs_in = o.costs_in;
  a.costs_own = o.costs_own ? 1 : 0;
  a.grad_lik = c->grad_lik;
  a.logl = c->logl;
  a.eta = c->eta;
  a.omegaT = o.want_omega ? c->omegaT : nullptr;
  if (o.want_actions) {
    TRY(ensure(&c->actions, &c->actions_cap, ((size_t)c->S * c->N * c->D + (o.store_f16 ? 1 : 0)) >> (o.store_f16 ? 1 : 0)));
    a.actions_out = c->actions;
  }
  if (o.want_states) {
    TRY(ensure(&c->states, &c->states_cap, ((size_t)c->M * c->S * c->N * (c->H + 1) * c->ds) >> (o.store_f16 ? 1 : 0)));
    a.states_out = c->states;
  }
  a.seed = c->cfg.seed;
  a.ctr = c->ctr_dev;
  a.bump_adam = o.bump_adam;
  a.rearm = c->stein_cnt;
  a.rearm_n = c->stein_cnt ? c->stein_tiles + 1 : 0;  // per-tile lines + the global line
  if (a.params == nullptr) {
    a.coef_given = 1;
    host_coef(c->cfg, a.coef_host);
  }
  a.stamps = c->stamps_dev ? c->stamps_dev + 16 * DUST_K_ROLLOUT : nullptr;
  int nt = ((std::max(c->S, c->D) + 63) / 64) * 64;
  nt = std::min(std::max(nt, 64), 256);
  a.lgW = 0;
  while ((1 << a.lgW) < c->D) ++a.lgW;  // nt = roundup64(max(S, D)) >= 2^lgW for D <= 128
  a.G = 1;
  {  // several dynamics samples and few action samples: split the M loop over lane groups (rollout.hpp)
    const int sub = ((c->S + 63) / 64) * 64;
    int G = 1;
    // (Particle: keep >= 2 dynamics samples per lane - the packed pair path of rollout.hpp rolls (m, m + G) side by side; one
    // sample per lane falls back to the general loop: measured 590 vs 330 us at cfg4, M = 4)
    const bool part = c->cfg.model == DUST_MODEL_PARTICLE;
    while (2 * G <= c->M && sub * 2 * G <= 256 && (!part || 4 * G <= c->M)) G *= 2;
    if (G > 1 && !o.costs_in && !c->mw_dev) {
      a.G = G;
      nt = sub * G;  // >= 128 >= D
    }
  }
  if (c->cfg.model == DUST_MODEL_PARTICLE && a.dm.with_obstacle && c->grid_bits) {
    const int words = (c->nx * c->ny + 31) / 32;
    if (words <= 4096) a.grid_words = (words + 3) & ~3;  // <= 16 KB: the map rides in LDS
  }
  const bool stage_states = o.want_states;
  const int stage_b = stage_states ? rollout_stage_bytes_per_lane(c->cfg.model) : 0;
  size_t lds = rollout_lds_bytes(c->S, c->D, c->M, nt, true, stage_b, a.grid_words);
  if (lds > 96 * 1024) {  // keep >= 1 workgroup per CU resident with room to spare; larger tiles go to an HBM slab
    TRY(ensure(&c->tile_scratch, &c->tile_cap, (size_t)c->nloc * c->S * (c->D | 1)));
    a.tile_scratch = c->tile_scratch;
    lds = rollout_lds_bytes(c->S, c->D, c->M, nt, false, stage_b, a.grid_words);  // (HBM tile: full-feature kernel)
    if (lds > 160 * 1024) return fail(DUST_ERR_UNSUPPORTED, "n_samples too large for one workgroup (%zu B of LDS)", lds);
  }
  {
    const int Q = std::max(1, nt / c->D);
    a.wq_iters = (c->S + Q - 1) / Q;
  }
  *nt_out = nt;
  *lds_out = lds;
  return DUST_OK;
}

// The whole-line stored-states form (rollout_states.hpp): Particle, fp32 in and out, 8-particle groups that are whole lines.
static bool states_whole_lines(const dust_ctx *c, const SampleOpts &o, const RolloutArgs &a, int *gw_out, size_t *lds_out) {
  if (c->env.states_form == 0) return false;  // development switch DUST_STATES_FORM: 0 keeps the per-particle staging kernel
  if (c->cfg.model != DUST_MODEL_PARTICLE || !a.states_out || a.costs_in || a.mw || a.tile_scratch) return false;
  if (o.store_f16 || a.noise_f16 || a.noise_mode == NOISE_PHILOX || !a.noise || a.a_reg != 0.0f || a.dm.interleave) return false;
  if (((c->H + 1) & 1) == 0 || c->H + 1 < 9) return false;
  if ((c->N % 8) || (c->n0 % 8) || (c->nloc % 8)) return false;
  if (a.dm.with_obstacle && a.grid_words == 0) return false;
  int gw = 4;
  while (gw > 1 && c->M % (2 * gw)) gw >>= 1;
  if (c->M % (2 * gw)) return false;
  const size_t lds = particle_states_lds_bytes(c->D, c->M, a.grid_words, gw);
  if (lds > 80 * 1024) return false;
  *gw_out = gw;
  *lds_out = lds;
  return true;
}

// ... its binary16 form (DUST_STORE_F16): 8-byte states, 16-particle groups (rollout_states.hpp particle_states_f16_kernel)
static bool states_whole_lines_f16(const dust_ctx *c, const SampleOpts &o, const RolloutArgs &a, int *gw_out, size_t *lds_out) {
  if (c->env.states_form == 0) return false;
  if (c->cfg.model != DUST_MODEL_PARTICLE || !a.states_out || a.costs_in || a.mw || a.tile_scratch) return false;
  if (!o.store_f16 || a.noise_f16 || a.noise_mode == NOISE_PHILOX || !a.noise || a.a_reg != 0.0f || a.dm.interleave) return false;
  if (((c->H + 1) & 1) == 0 || c->H < 16) return false;
  if ((c->N % 16) || (c->n0 % 16) || (c->nloc % 16)) return false;
  if (a.dm.with_obstacle && a.grid_words == 0) return false;
  int gw = 4;
  while (gw > 1 && c->M % (2 * gw)) gw >>= 1;
  if (c->M % (2 * gw)) return false;
  const size_t lds = particle_states_f16_lds_bytes(c->D, c->M, a.grid_words, gw);
  if (lds > 80 * 1024) return false;
  *gw_out = gw;
  *lds_out = lds;
  return true;
}

// ... and its Pendulum counterpart: 16-particle groups of 8 (H+1)-byte rows
static bool states_whole_lines_pend(const dust_ctx *c, const SampleOpts &o, const RolloutArgs &a, size_t *lds_out) {
  if (c->env.states_form == 0) return false;
  if (c->cfg.model != DUST_MODEL_PENDULUM || !a.states_out || a.costs_in || a.mw || a.tile_scratch) return false;
  if (o.store_f16 || a.noise_f16 || a.noise_mode == NOISE_PHILOX || !a.noise || a.a_reg != 0.0f || a.dm.interleave) return false;
  if (((c->H + 1) & 1) == 0 || c->H < 16) return false;
  if ((c->N % 16) || (c->n0 % 16) || (c->nloc % 16)) return false;
  const size_t lds = pendulum_states_lds_bytes(c->D, c->M);
  if (lds > 80 * 1024) return false;
  *lds_out = lds;
  return true;
}

// ... and the binary16 form of the Pendulum states (DUST_STORE_F16): 32-particle groups of 4 (H+1)-byte rows
static bool states_whole_lines_pend_f16(const dust_ctx *c, const SampleOpts &o, const RolloutArgs &a, size_t *lds_out) {
  if (c->env.states_form == 0) return false;
  if (c->cfg.model != DUST_MODEL_PENDULUM || !a.states_out || a.costs_in || a.mw || a.tile_scratch) return false;
  if (!o.store_f16 || a.noise_f16 || a.noise_mode == NOISE_PHILOX || !a.noise || a.a_reg != 0.0f || a.dm.interleave) return false;
  if ((c->N % 32) || (c->n0 % 32) || (c->nloc % 32)) return false;
  const size_t lds = pendulum_states_f16_lds_bytes(c->D, c->M, c->H);
  if (lds > 80 * 1024) return false;
  *lds_out = lds;
  return true;
}

static int launch_rollout(dust_ctx *c, const SampleOpts &o_in) {
  SampleOpts o = o_in;
  RolloutArgs a;
  int nt;
  size_t lds;
  TRY(rollout_args(c, o, a, &nt, &lds));
  if (c->cfg.model == DUST_MODEL_SKID_STEER) {
    // pass 1 (skid.hpp): rollouts + costs (+ states); pass 2 (below): the regular kernel in its injected-costs mode
    if (a.noise_f16 || o.store_f16) return fail(DUST_ERR_UNSUPPORTED, "binary16 storage is not implemented for the skid-steer family");
    if (a.mw) return fail(DUST_ERR_UNSUPPORTED, "sigma-point weights are not implemented for the skid-steer family");
    if (o.costs_in == nullptr) {
      Prof ps(c, DUST_K_ROLLOUT_STATES);
      SkidArgs k;
      memset(&k, 0, sizeof k);
      k.sk = c->skid;
      k.N_total = c->N;
      k.n0 = c->n0;
      k.n_local = c->nloc;
      k.S = c->S;
      k.M = c->M;
      k.H = c->H;
      k.D = c->D;
      k.P = c->P;
      k.noise_mode = a.noise_mode;
      k.log_space = c->cfg.params_log_space;
      k.interleave = c->cfg.params_interleave;
      k.dt = (float)c->cfg.dt;
      k.chol_a[0] = a.chol_a[0];
      k.chol_a[1] = a.chol_a[1];
      k.chol_off = a.chol_off;
      k.seed = a.seed;
      k.ctr = a.ctr;
      k.noise = a.noise;
      k.theta = a.theta;
      k.state = a.state;
      k.params = a.params;
      k.costs_sn = c->costs_stage;
      k.costsT = c->costsT;
      k.states_out = a.states_out;
      const int nthr = c->nloc * c->S;
      skid_rollout_kernel<<<(nthr + 255) / 256, 256, 0, c->stream>>>(k);
      HIP_TRY(hipGetLastError());
      o.want_states = false;
      o.costs_in = c->costs_stage;
      o.costs_own = true;
      TRY(rollout_args(c, o, a, &nt, &lds));
    }
  }
  // Control-channel noise with acceleration control, drawn on the device, nothing but costs wanted: the regular kernel draws it inside
  // its own rollout loops (rollout.hpp, round 6: the packed pair path keeps its 40 instructions per step and sample and adds one
  // eight-normal Philox block per two steps of a pair) - no first pass.  Recorded draws (the goldens), stored states, velocity control:
  // particle_general.hpp below.
  const bool noise_inline = particle_general(c) && o.costs_in == nullptr && c->cfg.control_type != DUST_CONTROL_VELOCITY && !o.want_states &&
                            !a.mw && !(c->cz_dev && c->cz_next < c->cz_sets) && c->env.noise_general <= 0;
  if (noise_inline) {
    a.ctrl_noise = 1;
    a.dyn_std[0] = c->cfg.dyn_std[0];
    a.dyn_std[1] = c->cfg.dyn_std[1];
  }
  if (particle_general(c) && o.costs_in == nullptr && !noise_inline) {
    // pass 1 (particle_general.hpp): rollouts with control noise / velocity control + costs (+ states); pass 2: the regular kernel in
    // its injected-costs mode
    if (a.mw) return fail(DUST_ERR_UNSUPPORTED, "sigma-point weights are not implemented for Particle rollouts with control noise / velocity control");
    Prof ps(c, DUST_K_ROLLOUT_STATES);
    PartGenArgs k;
    memset(&k, 0, sizeof k);
    k.dm = a.dm;
    k.N_total = c->N;
    k.n0 = c->n0;
    k.n_local = c->nloc;
    k.S = c->S;
    k.M = c->M;
    k.H = c->H;
    k.D = c->D;
    k.noise_mode = a.noise_mode;
    k.noise_f16 = a.noise_f16;
    k.store_f16 = a.store_f16;
    k.velocity = c->cfg.control_type == DUST_CONTROL_VELOCITY;
    k.ctrl_noise = c->cfg.ctrl_noise && (c->cfg.dyn_std[0] != 0.f || c->cfg.dyn_std[1] != 0.f);
    for (int d = 0; d < 2; ++d) {
      k.dyn_std[d] = c->cfg.dyn_std[d];
      k.chol_a[d] = a.chol_a[d];
      k.a_pre[d] = a.a_pre[d];
    }
    k.a_reg = a.a_reg;
    k.chol_off = a.chol_off;
    k.a_pre_off = a.a_pre_off;
    k.seed = a.seed;
    k.ctr = a.ctr;
    k.noise = a.noise;
    k.theta = a.theta;
    k.state = a.state;
    k.params = a.params;
    if (k.ctrl_noise && c->cz_dev && c->cz_next < c->cz_sets) {  // recorded draws: one set per rollout launch
      if (c->capturing) return fail(DUST_ERR_STATE, "recorded control noise cannot be replayed from a captured graph");
      k.cz = c->cz_dev + (size_t)c->cz_next * c->H * c->M * c->S * c->N * 2;
      c->cz_next++;
    }
    k.a_seq = a.a_seq;
    k.a_mat = a.a_mat;
    k.costs_sn = c->costs_stage;
    k.costsT = c->costsT;
    k.states_out = a.states_out;
    const int nthr = c->nloc * c->S;
    const int pg_gw = (k.dm.with_obstacle && k.dm.grid_bits) ? (k.dm.nx * k.dm.ny + 31) / 32 : 0;
    k.mc = c->M >= 8 ? 8 : (c->M >= 4 ? 4 : (c->M >= 2 ? 2 : 1));
    const size_t pg_lds = particle_general_lds_bytes(c->D, pg_gw, k.mc);
    if (pg_lds > 160 * 1024) return fail(DUST_ERR_UNSUPPORTED, "occupancy grid of %d x %d cells: the control-noise / velocity-control rollouts keep it in LDS", k.dm.nx, k.dm.ny);
    if (pg_lds > 64 * 1024 && !c->capturing)
      HIP_TRY(hipFuncSetAttribute((const void *)particle_general_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pg_lds));
    const int pg_rows = PARTGEN_NT / k.mc;
    particle_general_kernel<<<(nthr + pg_rows - 1) / pg_rows, PARTGEN_NT, pg_lds, c->stream>>>(k);
    HIP_TRY(hipGetLastError());
    o.want_states = false;
    o.costs_in = c->costs_stage;
    o.costs_own = true;
    TRY(rollout_args(c, o, a, &nt, &lds));
  }
  {
    int gw;
    size_t lds_s;
    if (states_whole_lines(c, o, a, &gw, &lds_s)) {
      // pass 1: rollouts + states + costs; pass 2 (below): the regular kernel in its injected-costs mode - softmax, weights, score
      {
        Prof ps(c, DUST_K_ROLLOUT_STATES);
        const int blocks = (c->nloc / 8) * ((c->S + 7) / 8);
        TRY(ensure(&c->wg_flags, &c->wg_flags_cap, (size_t)blocks));
#define DUST_LAUNCH_STATES(MODE)                                                                                                             \
  do {                                                                                                                                       \
    if (lds_s > 64 * 1024 && !c->capturing)                                                                                                  \
      HIP_TRY(hipFuncSetAttribute((const void *)particle_states_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s));      \
    particle_states_kernel<MODE><<<blocks, 64 * gw, lds_s, c->stream>>>(a, c->costs_stage, gw, reinterpret_cast<unsigned int *>(c->wg_flags));                                 \
  } while (0)
        if (!a.dm.with_obstacle) DUST_LAUNCH_STATES(SP_FAST_FREE);
        else if (a.dm.can_crash) DUST_LAUNCH_STATES(SP_FAST_CRASH);
        else DUST_LAUNCH_STATES(SP_FAST_OBST);
        DUST_LAUNCH_STATES(SP_GENERAL);  // only the workgroups the fast kernel flagged (non-finite operands) do any work here
#undef DUST_LAUNCH_STATES
        HIP_TRY(hipGetLastError());
      }
      o.want_states = false;
      o.costs_in = c->costs_stage;
      o.costs_own = true;
      TRY(rollout_args(c, o, a, &nt, &lds));
    } else if (states_whole_lines_f16(c, o, a, &gw, &lds_s)) {
      {
        Prof ps(c, DUST_K_ROLLOUT_STATES);
        const int blocks = (c->nloc / 16) * ((c->S + 3) / 4);
        TRY(ensure(&c->wg_flags, &c->wg_flags_cap, (size_t)blocks));
#define DUST_LAUNCH_STATES(MODE)                                                                                                                 \
  do {                                                                                                                                           \
    if (lds_s > 64 * 1024 && !c->capturing)                                                                                                      \
      HIP_TRY(hipFuncSetAttribute((const void *)particle_states_f16_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s));      \
    particle_states_f16_kernel<MODE><<<blocks, 64 * gw, lds_s, c->stream>>>(a, c->costs_stage, gw, reinterpret_cast<unsigned int *>(c->wg_flags)); \
  } while (0)
        if (!a.dm.with_obstacle) DUST_LAUNCH_STATES(SP_FAST_FREE);
        else if (a.dm.can_crash) DUST_LAUNCH_STATES(SP_FAST_CRASH);
        else DUST_LAUNCH_STATES(SP_FAST_OBST);
        DUST_LAUNCH_STATES(SP_GENERAL);  // only the workgroups the fast kernel flagged (non-finite operands) do any work here
#undef DUST_LAUNCH_STATES
        HIP_TRY(hipGetLastError());
      }
      o.want_states = false;
      o.costs_in = c->costs_stage;
      o.costs_own = true;
      TRY(rollout_args(c, o, a, &nt, &lds));
    } else if (states_whole_lines_pend(c, o, a, &lds_s)) {
      {
        Prof ps(c, DUST_K_ROLLOUT_STATES);
        const int blocks = (c->nloc / 16) * ((c->S + 15) / 16);
        TRY(ensure(&c->wg_flags, &c->wg_flags_cap, (size_t)blocks));
        unsigned int *fl = reinterpret_cast<unsigned int *>(c->wg_flags);
        if (lds_s > 64 * 1024 && !c->capturing) {
          HIP_TRY(hipFuncSetAttribute((const void *)pendulum_states_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s));
          HIP_TRY(hipFuncSetAttribute((const void *)pendulum_states_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s));
        }
        pendulum_states_kernel<false><<<blocks, 256, lds_s, c->stream>>>(a, c->costs_stage, fl);
        pendulum_states_kernel<true><<<blocks, 256, lds_s, c->stream>>>(a, c->costs_stage, fl);  // flagged workgroups only
        HIP_TRY(hipGetLastError());
      }
      o.want_states = false;
      o.costs_in = c->costs_stage;
      o.costs_own = true;
      TRY(rollout_args(c, o, a, &nt, &lds));
    } else if (states_whole_lines_pend_f16(c, o, a, &lds_s)) {
      {
        Prof ps(c, DUST_K_ROLLOUT_STATES);
        const int blocks = (c->nloc / 32) * ((c->S + 7) / 8);
        TRY(ensure(&c->wg_flags, &c->wg_flags_cap, (size_t)blocks));
        unsigned int *fl = reinterpret_cast<unsigned int *>(c->wg_flags);
        if (lds_s > 64 * 1024 && !c->capturing) {
          HIP_TRY(hipFuncSetAttribute((const void *)pendulum_states_f16_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s));
          HIP_TRY(hipFuncSetAttribute((const void *)pendulum_states_f16_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s));
        }
        pendulum_states_f16_kernel<false><<<blocks, 256, lds_s, c->stream>>>(a, c->costs_stage, fl);
        pendulum_states_f16_kernel<true><<<blocks, 256, lds_s, c->stream>>>(a, c->costs_stage, fl);  // flagged workgroups only
        HIP_TRY(hipGetLastError());
      }
      o.want_states = false;
      o.costs_in = c->costs_stage;
      o.costs_own = true;
      TRY(rollout_args(c, o, a, &nt, &lds));
    }
  }
  Prof p(c, DUST_K_ROLLOUT);
#define DUST_LAUNCH_ROLLOUT(KERNEL)                                                                                            \
  do {                                                                                                                          \
    if (lds > 64 * 1024 && !c->capturing)                                                                                       \
      HIP_TRY(hipFuncSetAttribute((const void *)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                 \
    KERNEL<<<c->nloc, nt, lds, c->stream>>>(a);                                                                                 \
  } while (0)
  const bool stream_form = a.noise_mode == NOISE_EPS;
  // LEAN instances: the optional features compiled out (rollout.hpp); STATES: the stored-states form (states staged through LDS,
  // the Particle map in LDS); both need the occupancy grid in LDS when there is one
  const bool grid_ok = c->cfg.model != DUST_MODEL_PARTICLE || !a.dm.with_obstacle || a.grid_words > 0;
  const bool lean = !a.states_out && !a.actions_out && !a.costs_in && a.a_reg == 0.0f && !a.mw && !a.tile_scratch && !a.omegaT && grid_ok;
  const bool states_form = a.states_out && !a.costs_in && !a.mw && !a.tile_scratch && grid_ok && stream_form;
#define DUST_PICK_ROLLOUT3(MODEL, GR, LN)                                                            \
  do {                                                                                              \
    if (states_form) DUST_LAUNCH_ROLLOUT((rollout_stream_kernel<MODEL, GR, false, true>));           \
    else if (stream_form) DUST_LAUNCH_ROLLOUT((rollout_stream_kernel<MODEL, GR, LN>));               \
    else DUST_LAUNCH_ROLLOUT((rollout_kernel<MODEL, GR, LN>));                                       \
  } while (0)
#define DUST_PICK_ROLLOUT(MODEL)                                     \
  do {                                                               \
    if (a.G > 1) {                                                   \
      if (lean) DUST_PICK_ROLLOUT3(MODEL, true, true);               \
      else DUST_PICK_ROLLOUT3(MODEL, true, false);                   \
    } else {                                                         \
      if (lean) DUST_PICK_ROLLOUT3(MODEL, false, true);              \
      else DUST_PICK_ROLLOUT3(MODEL, false, false);                  \
    }                                                                \
  } while (0)
  if (c->cfg.model == DUST_MODEL_PENDULUM) DUST_PICK_ROLLOUT(DUST_MODEL_PENDULUM);
  else DUST_PICK_ROLLOUT(DUST_MODEL_PARTICLE);
#undef DUST_PICK_ROLLOUT3
#undef DUST_PICK_ROLLOUT
#undef DUST_LAUNCH_ROLLOUT
  HIP_TRY(hipGetLastError());
  c->actions_valid = o.want_actions;
  c->actions_f16 = o.want_actions && o.store_f16;
  c->states_valid = o_in.want_states;
  c->states_f16 = o_in.want_states && o_in.store_f16;
  c->stein_dirty = false;
  return DUST_OK;
}

struct StateWord {
  float v[8];
};
__global__ void set_state_kernel(float *dst, const StateWord w) { dst[threadIdx.x] = w.v[threadIdx.x]; }
// the same with the tick's dynamics samples ([n_sets][M][P], up to 240 floats) behind the state: one 256-lane launch instead of a launch
// and a host-to-device copy (4 us on the stream, and a staging copy on the host when the caller's array is pageable)
struct StateParamsWord {
  float v[8];
  float p[240];
};
__global__ void set_state_params_kernel(float *dst, float *pdst, const StateParamsWord w, int ns, int np) {
  const int t = threadIdx.x;
  if (t < ns) dst[t] = w.v[t];
  if (t < np) pdst[t] = w.p[t];
}

static int upload_state_params(dust_ctx *c, const float *state, const float *params, int n_sets) {
  if (c->params_staged) params = nullptr;  // (dust_dual_tick: the samples are in params_dev already)
  const size_t np = (c->cfg.dim_p > 0 && c->M >= 1 && params) ? (size_t)n_sets * c->M * c->P : 0;
  if (np > 0 && np <= 240) {
    TRY(ensure(&c->params_dev, &c->params_cap, np));
    StateParamsWord w;
    for (int k = 0; k < 8; ++k) w.v[k] = (state && k < c->ds) ? state[k] : 0.f;
    memcpy(w.p, params, np * sizeof(float));
    set_state_params_kernel<<<1, 256, 0, c->stream>>>(c->state_dev, c->params_dev, w, state ? 8 : 0, (int)np);
    HIP_TRY(hipGetLastError());
    return DUST_OK;
  }
  if (state) {
    // the 16-byte plant state travels as a kernel ARGUMENT of a 4-lane launch (the runtime copies arguments at launch
    // time: nothing to keep alive, no host wait); measured 2 us on the stream against 4 us for a 16-byte hipMemcpyAsync
    StateWord w;
    for (int k = 0; k < 8; ++k) w.v[k] = k < c->ds ? state[k] : 0.f;
    set_state_kernel<<<1, 8, 0, c->stream>>>(c->state_dev, w);
    HIP_TRY(hipGetLastError());
  }
  if (c->cfg.dim_p > 0 && c->M >= 1) {
    if (params) {
      TRY(ensure(&c->params_dev, &c->params_cap, (size_t)n_sets * c->M * c->P));
      TRY(h2d(c, c->params_dev, params, (size_t)n_sets * c->M * c->P * sizeof(float)));
    }
  }
  return DUST_OK;
}

static int stage_noise(dust_ctx *c, const float *src, int flags, const float **dev) {
  const size_t n = (size_t)c->S * c->N * c->D;
  if (!src) {
    *dev = nullptr;
    c->noise_f16 = false;
    return DUST_OK;
  }
  c->noise_f16 = (flags & DUST_EPS_F16) != 0;
  if (flags & DUST_PTR_DEVICE) {
    *dev = src;
    return DUST_OK;
  }
  TRY(ensure(&c->noise_stage, &c->noise_cap, n));
  TRY(h2d(c, c->noise_stage, src, n * ((flags & DUST_EPS_F16) ? 2 : sizeof(float))));
  *dev = c->noise_stage;
  return DUST_OK;
}

static int copy_out_SN(dust_ctx *c, const float *srcT, float *host) {  // device [N][S] -> host [S][N]
  const int n = c->N * c->S;
  transpose2_kernel<<<(n + 255) / 256, 256, 0, c->stream>>>(srcT, c->tmp, c->N, c->S);
  HIP_TRY(hipGetLastError());
  return d2h(c, host, c->tmp, (size_t)n * sizeof(float));
}

// rows[i] = (m*S + s)*N + n -> out row i ([H+1][ds] values); 64-bit row indices: the states of BASELINE configs[2] are 2.77 G values
__global__ void gather_rows_kernel(const void *src, void *dst, const long long *rows, int n_rows, int row_elems, int elem_bytes) {
  const int i = blockIdx.x;
  if (i >= n_rows) return;
  const size_t base = (size_t)rows[i] * (size_t)row_elems;
  for (int k = threadIdx.x; k < row_elems; k += blockDim.x) {
    if (elem_bytes == 2) reinterpret_cast<uint16_t *>(dst)[(size_t)i * row_elems + k] = reinterpret_cast<const uint16_t *>(src)[base + k];
    else reinterpret_cast<uint32_t *>(dst)[(size_t)i * row_elems + k] = reinterpret_cast<const uint32_t *>(src)[base + k];
  }
}
extern "C" int dust_get_states_rows(dust_ctx *c, const long long *rows, int n_rows, void *out) {
  if (!c || !rows || !out || n_rows < 1) return fail(DUST_ERR_INVALID, "bad argument");
  TRY(settle_pending(c));
  if (!c->states_valid || !c->states)
    return fail(DUST_ERR_STATE, "no stored states on the device: run a sample with DUST_STORE_STATES (or a forward that returns states) first");
  const long long R = (long long)c->M * c->S * c->N;
  for (int i = 0; i < n_rows; ++i)
    if (rows[i] < 0 || rows[i] >= R) return fail(DUST_ERR_INVALID, "row %lld outside [0, M*S*N = %lld)", rows[i], R);
  HIP_TRY(hipSetDevice(c->cfg.device));
  const int row_elems = (c->H + 1) * c->ds, eb = c->states_f16 ? 2 : 4;
  const size_t idx_floats = ((size_t)n_rows * sizeof(long long) + 3) / 4, out_floats = ((size_t)n_rows * row_elems * eb + 3) / 4;
  TRY(ensure(&c->tmp, &c->tmp_cap, idx_floats + 2 + out_floats));
  long long *idx_dev = reinterpret_cast<long long *>(c->tmp);  // (hipMalloc'd: 256-byte aligned)
  void *dst = c->tmp + ((idx_floats + 1) & ~(size_t)1);
  TRY(h2d(c, idx_dev, rows, (size_t)n_rows * sizeof(long long)));
  gather_rows_kernel<<<n_rows, 64, 0, c->stream>>>(c->states, dst, idx_dev, n_rows, row_elems, eb);
  HIP_TRY(hipGetLastError());
  return d2h(c, out, dst, (size_t)n_rows * row_elems * eb);
}

extern "C" int dust_get_costs(dust_ctx *c, float *costs) {
  if (!c || !costs) return fail(DUST_ERR_INVALID, "null argument");
  if (!c->have_sample) return fail(DUST_ERR_STATE, "no likelihood sample yet");
  return copy_out_SN(c, c->costsT, costs);
}
extern "C" int dust_get_actions(dust_ctx *c, float *actions) {
  if (!c || !actions) return fail(DUST_ERR_INVALID, "null argument");
  if (!c->actions_valid) return fail(DUST_ERR_STATE, "the last sample did not keep its actions (pass actions_out)");
  if (c->actions_f16) return fail(DUST_ERR_STATE, "the last sample kept its actions as binary16 (DUST_STORE_F16): they were returned by that call");
  return d2h(c, actions, c->actions, (size_t)c->S * c->N * c->D * sizeof(float));
}
extern "C" int dust_get_score(dust_ctx *c, float *s) {
  if (!c || !s) return fail(DUST_ERR_INVALID, "null argument");
  return d2h(c, s, c->score, (size_t)c->N * c->D * sizeof(float));
}
extern "C" int dust_get_score_parts(dust_ctx *c, float *gl, float *gp) {
  if (!c) return fail(DUST_ERR_INVALID, "null ctx");
  if (gl) TRY(d2h(c, gl, c->grad_lik, (size_t)c->N * c->D * sizeof(float)));
  if (gp) TRY(d2h(c, gp, c->grad_pri, (size_t)c->N * c->D * sizeof(float)));
  return DUST_OK;
}
extern "C" int dust_get_phi(dust_ctx *c, float *s) {
  if (!c || !s) return fail(DUST_ERR_INVALID, "null argument");
  return d2h(c, s, c->phi, (size_t)c->N * c->D * sizeof(float));
}
extern "C" int dust_get_log_weights(dust_ctx *c, float *ll, float *lp) {
  if (!c) return fail(DUST_ERR_INVALID, "null ctx");
  if (ll) TRY(d2h(c, ll, c->logl, c->N * sizeof(float)));
  if (lp) TRY(d2h(c, lp, c->logp, c->N * sizeof(float)));
  return DUST_OK;
}
extern "C" int dust_get_bandwidths(dust_ctx *c, float *h) {
  if (!c || !h) return fail(DUST_ERR_INVALID, "null argument");
  const int G = c->cfg.kernel == DUST_KERNEL_K2_SHARED ? c->H : c->D;
  return d2h(c, h, c->bw, G * sizeof(float));
}
extern "C" int dust_likelihood_log_prob(dust_ctx *c, float *ll) {
  if (!c || !ll) return fail(DUST_ERR_INVALID, "null argument");
  if (!c->have_sample) return fail(DUST_ERR_STATE, "no likelihood sample yet");
  return d2h(c, ll, c->logl, c->N * sizeof(float));
}

extern "C" int dust_disco_forward(dust_ctx *c, const float *state, const float *actions, const float *params, int flags,
                                  float *costs, float *states, float *actions_out, float *omega) {
  if (!c || !state) return fail(DUST_ERR_INVALID, "null argument");
  TRY(settle_pending(c));
  if (c->cfg.dim_p > 0 && !params) return fail(DUST_ERR_INVALID, "params_sampling is on: pass the [M][P] parameter samples");
  HIP_TRY(hipSetDevice(c->cfg.device));
  TRY(upload_state_params(c, state, params, 1));
  const float *nd = nullptr;
  TRY(stage_noise(c, actions, flags, &nd));
  SampleOpts o;
  memset(&o, 0, sizeof o);
  const bool internal = (actions == nullptr);
  const bool around_a_mat = internal || (flags & DUST_EPS_AROUND_A_MAT);
  o.noise_mode = internal ? NOISE_PHILOX : NOISE_ACTIONS;
  o.noise_dev = nd;
  o.base = around_a_mat ? c->a_mat : c->theta;
  o.eps_base_mode = around_a_mat ? 1 : 0;
  o.update_a_mat = 1;
  o.want_actions = actions_out != nullptr;
  o.want_states = states != nullptr;
  o.want_omega = omega != nullptr;
  o.store_f16 = (flags & DUST_STORE_F16) != 0;
  const size_t osz = o.store_f16 ? 2 : sizeof(float);
  TRY(launch_rollout(c, o));
  c->have_sample = true;
  bump_iter_kernel<<<1, 1, 0, c->stream>>>(c->ctr_dev);
  HIP_TRY(hipGetLastError());
  if (costs) TRY(copy_out_SN(c, c->costsT, costs));
  if (omega) TRY(copy_out_SN(c, c->omegaT, omega));
  if (actions_out) TRY(d2h(c, actions_out, c->actions, (size_t)c->S * c->N * c->D * osz));
  if (states) TRY(d2h(c, states, c->states, (size_t)c->M * c->S * c->N * (c->H + 1) * c->ds * osz));
  return DUST_OK;
}

extern "C" int dust_likelihood_sample(dust_ctx *c, const float *state, const float *eps, const float *params, int flags,
                                      float *costs, float *actions_out) {
  if (!c || !state) return fail(DUST_ERR_INVALID, "null argument");
  TRY(settle_pending(c));
  if (c->cfg.dim_p > 0 && !params) return fail(DUST_ERR_INVALID, "params_sampling is on: pass the [M][P] parameter samples");
  HIP_TRY(hipSetDevice(c->cfg.device));
  TRY(upload_state_params(c, state, params, 1));
  const float *nd = nullptr;
  TRY(stage_noise(c, eps, flags, &nd));
  SampleOpts o;
  memset(&o, 0, sizeof o);
  o.noise_mode = eps ? NOISE_EPS : NOISE_PHILOX;
  o.noise_dev = nd;
  o.base = c->theta;
  o.update_a_mat = 1;
  o.want_actions = actions_out != nullptr;
  o.want_states = (flags & DUST_STORE_STATES) != 0;
  o.store_f16 = (flags & DUST_STORE_F16) != 0;
  TRY(launch_rollout(c, o));
  c->have_sample = true;
  bump_iter_kernel<<<1, 1, 0, c->stream>>>(c->ctr_dev);
  HIP_TRY(hipGetLastError());
  if (costs) TRY(copy_out_SN(c, c->costsT, costs));
  if (actions_out) TRY(d2h(c, actions_out, c->actions, (size_t)c->S * c->N * c->D * (o.store_f16 ? 2 : sizeof(float))));
  return DUST_OK;
}

extern "C" int dust_likelihood_sample_at(dust_ctx *c, const float *state, const float *theta, const float *eps, const float *params, int flags,
                                         float *costs, float *actions_out) {
  if (!c || !state || !theta) return fail(DUST_ERR_INVALID, "null argument");
  TRY(settle_pending(c));
  if (c->cfg.dim_p > 0 && !params) return fail(DUST_ERR_INVALID, "params_sampling is on: pass the [M][P] parameter samples");
  HIP_TRY(hipSetDevice(c->cfg.device));
  TRY(upload_state_params(c, state, params, 1));
  const float *nd = nullptr;
  TRY(stage_noise(c, eps, flags, &nd));
  // the caller's theta rides in the spare particle buffer of the ping-pong: the optimiser's particles and moments stay untouched
  float *base = c->theta_alt && c->theta_alt != c->theta ? c->theta_alt : nullptr;
  if (!base) return fail(DUST_ERR_STATE, "no spare particle buffer");
  TRY(h2d(c, base, theta, (size_t)c->N * c->D * sizeof(float)));
  SampleOpts o;
  memset(&o, 0, sizeof o);
  o.noise_mode = eps ? NOISE_EPS : NOISE_PHILOX;
  o.noise_dev = nd;
  o.base = base;
  o.update_a_mat = 1;
  o.want_actions = actions_out != nullptr;
  o.want_states = (flags & DUST_STORE_STATES) != 0;
  o.store_f16 = (flags & DUST_STORE_F16) != 0;
  TRY(launch_rollout(c, o));
  c->have_sample = true;
  bump_iter_kernel<<<1, 1, 0, c->stream>>>(c->ctr_dev);
  HIP_TRY(hipGetLastError());
  if (costs) TRY(copy_out_SN(c, c->costsT, costs));
  if (actions_out) TRY(d2h(c, actions_out, c->actions, (size_t)c->S * c->N * c->D * (o.store_f16 ? 2 : sizeof(float))));
  return DUST_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// pairwise passes (tiled: PAIR_TI queries x key slices; partials combined by the next kernel in the chain)

template <int MODE>
static int launch_pair_big(dust_ctx *c, const PairArgs &a, int tiles);

template <int MODE>
static int launch_pair(dust_ctx *c, const PairArgs &a, int tiles) {
  if (pair_big_kernel(c)) return launch_pair_big<MODE>(c, a, (a.n_local + 4096 / pair_dpb(a.D) - 1) / (4096 / pair_dpb(a.D)));
  tiles = (a.n_local + PAIR_TI - 1) / PAIR_TI;  // (the caller's count is the primary kernel's: pair_geometry)
  const int cpt = cpt_for(a.D);
  const size_t lds = pairwise_lds_bytes(MODE, cpt);
  dim3 grid(tiles, a.JS);
#define DUST_LAUNCH_PAIR(CPT)                                                                                                    \
  do {                                                                                                                            \
    if (lds > 64 * 1024 && !c->capturing)                                                                                         \
      HIP_TRY(hipFuncSetAttribute((const void *)pairwise_kernel<MODE, CPT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    pairwise_kernel<MODE, CPT><<<grid, PAIR_NT, lds, c->pair_stream>>>(a);                                                             \
  } while (0)
  if (cpt == 4) DUST_LAUNCH_PAIR(4);
  else if (cpt == 8) DUST_LAUNCH_PAIR(8);
  else if (cpt == 12) DUST_LAUNCH_PAIR(12);
  else DUST_LAUNCH_PAIR(16);
#undef DUST_LAUNCH_PAIR
  HIP_TRY(hipGetLastError());
  return DUST_OK;
}

template <int MODE>
static int launch_pair_big(dust_ctx *c, const PairArgs &a, int tiles) {
  const int dpb = pair_dpb(a.D);
  TRY(ensure(&c->xpad, &c->xpad_cap, (size_t)c->N * dpb));
  {
    const int n = c->N * dpb;
    pad_rows_kernel<<<(n + 255) / 256, 256, 0, c->pair_stream>>>(a.X, c->xpad, c->N, a.D, dpb);
    HIP_TRY(hipGetLastError());
  }
  PairBigArgs b;
  memset(&b, 0, sizeof b);
  b.p = a;
  b.Xp = c->xpad;
  b.ldp = 8 * cpt_for(a.D);
  b.w[0] = a.inv_s[0] * a.inv_s[0];
  b.w[1] = a.da == 2 ? a.inv_s[1] * a.inv_s[1] : b.w[0];
  const size_t lds = pairwise_big_lds_bytes(MODE, dpb);
  dim3 grid(tiles, a.JS);
#define DUST_LAUNCH_BIG(DPB)                                                                                                        \
  do {                                                                                                                               \
    if (lds > 64 * 1024 && !c->capturing)                                                                                            \
      HIP_TRY(hipFuncSetAttribute((const void *)pairwise_big_kernel<MODE, DPB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    pairwise_big_kernel<MODE, DPB><<<grid, PAIR_NT, lds, c->pair_stream>>>(b);                                                       \
  } while (0)
  if (dpb == 32) DUST_LAUNCH_BIG(32);
  else DUST_LAUNCH_BIG(64);
#undef DUST_LAUNCH_BIG
  HIP_TRY(hipGetLastError());
  return DUST_OK;
}

static int ensure_partials(dust_ctx *c, int JS);
static int far_counts_alloc(dust_ctx *c);

// the side stream's query_order_kernel (launch_gram_score) must be done before its outputs are read or its inputs rewritten
static int order_join(dust_ctx *c) {
  if (!c->order_pending) return DUST_OK;
  HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_order, 0));
  if (c->pair_stream != c->stream) HIP_TRY(hipStreamWaitEvent(c->pair_stream, c->ev_order, 0));
  c->order_pending = false;
  return DUST_OK;
}

__global__ void iota_kernel(int *p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = i;
}

// slices per query tile of the two run-list passes (pairwise_packed.hpp): about two resident rounds of workgroups when every
// slice has units, at most 24 partial rows per particle for the merge kernels.  Static - a function of the shape alone: the
// summation order does not depend on the data, and a captured tick keeps its grid.
static int packed_slices(const dust_ctx *c, int tiles, int chunks) {
  static const char *env = getenv("DUST_PACK_JS");  // development switch
  if (env) return std::max(1, std::min(chunks, atoi(env)));
  const int want = (4 * device_cus(c) + tiles - 1) / tiles;
  // (fewer slices for the short merged lists were measured on a rank of 8: 8 slices 378-391 us per tick, 4 slices 456-460, against
  //  360-396 at 24 - a rank's tiles are less coherent than one GPU's, its lists longer)
  return std::max(1, std::min(std::min(chunks, 24), want));
}

// pass 1 of the fused pair (pairwise_packed.hpp): prior partials + repulsion partials + the kernel blocks of the current theta
static int launch_pair_fused(dust_ctx *c, const PairArgs &a, int tiles) {
  const int dpb = fused_dpb(a.D), tq = fused_tq(a.D);
  const int chunks = (c->N + PAIR_JC - 1) / PAIR_JC;
  TRY(ensure(&c->xpad, &c->xpad_cap, (size_t)c->N * dpb));
  TRY(ensure(&c->kmat, &c->kmat_cap, (size_t)tiles * chunks * tq * 64));  // [tiles][chunks][TQ][64]: touched only where units exist
  const bool whole_rows = a.D == dpb;  // (D = 80: the rows are their own padded copy - no pad_rows launch, 7 us per pass at N = 16 384)
  if (!whole_rows) {
    const int n = c->N * dpb;
    pad_rows_kernel<<<(n + 255) / 256, 256, 0, c->pair_stream>>>(a.X, c->xpad, c->N, a.D, dpb);
    HIP_TRY(hipGetLastError());
  }
  PairPackedArgs b;
  memset(&b, 0, sizeof b);
  b.p = a;
  b.Xp = whole_rows ? a.X : c->xpad;
  b.ldp = 8 * cpt_for(a.D);
  b.wP[0] = a.inv_s[0] * a.inv_s[0];
  b.wP[1] = a.da == 2 ? a.inv_s[1] * a.inv_s[1] : b.wP[0];
  const float ell = c->cfg.kernel == DUST_KERNEL_IMQ ? c->cfg.imq_ell : 0.69314718055994531f;
  b.wS[0] = b.wS[1] = (1.0f / ell) * (1.0f / ell);
  b.Kp = c->kmat;
  b.tiles = tiles;
  b.umax = chunks;
  const bool dense = c->env.dense >= 0;  // (development switch: evaluate everything)
  c->far_tiles = c->far_chunks = 0;
  PackArgs pk;
  memset(&pk, 0, sizeof pk);
  // tile order (pairwise_packed.hpp): merged lists only - K1 behind the far pre-pass below the exact-zero threshold
  const bool merge_mode = c->cfg.kernel == DUST_KERNEL_K1_RBF && !dense && c->env.far != 0 && c->env.far_t < DUST_FAR_T_EXACT && c->env.pack_merge != 0;
  const bool order = merge_mode && c->env.pack_order != 0 && !c->stagewise;
  if (order) {
    const bool fresh = c->pk_perm_cap < (size_t)c->nloc || c->pk_lead_cap < (size_t)c->nloc;
    if (fresh && c->capturing) return fail(DUST_ERR_STATE, "tile-order buffers must exist before a capture");
    TRY(ensure(&c->pk_perm, &c->pk_perm_cap, (size_t)c->nloc));
    TRY(ensure(&c->pk_lead, &c->pk_lead_cap, (size_t)c->nloc));
    if (fresh) {  // index order, no leaders yet
      iota_kernel<<<(c->nloc + 255) / 256, 256, 0, c->pair_stream>>>(reinterpret_cast<int *>(c->pk_perm), c->nloc);
      HIP_TRY(hipMemsetAsync(c->pk_lead, 0x7f, (size_t)c->nloc * sizeof(int), c->pair_stream));
    }
  }
  c->pk_order = order;
  TRY(order_join(c));
  if (c->cfg.kernel == DUST_KERNEL_K1_RBF) {  // pairwise_far.hpp: where the running maxima start (every mode), and the far units
    const bool flags = !dense && c->env.far != 0;  // (DUST_FAR=0 / DUST_DENSE=1: visit all)
    FarArgs f;
    memset(&f, 0, sizeof f);
    f.N = c->N;
    f.D = a.D;
    f.i0 = a.i0;
    f.n_local = c->nloc;
    f.tiles = tiles;
    f.q_rows = tiles * tq;
    f.lscale = 1.0f;
    f.chunks = chunks;
    f.X = a.X;
    f.logmix = a.logmix;
    f.sg[0] = sqrtf(std::min(b.wS[0], b.wP[0]));
    f.sg[1] = sqrtf(std::min(b.wS[1], b.wP[1]));
    TRY(ensure(&c->far_z, &c->far_z_cap, ((size_t)c->N * far_zh(dpb) + 1) / 2));
    TRY(ensure(&c->far_n, &c->far_n_cap, 3 * (size_t)c->N + chunks));
    TRY(ensure(&c->far_f, &c->far_f_cap, ((size_t)tiles * chunks + 3) / 4));
    f.Z = reinterpret_cast<_Float16 *>(c->far_z);
    f.nrm = c->far_n;
    f.lms = c->far_n + c->N;
    f.m0 = c->far_n + 2 * (size_t)c->N;
    f.T = c->env.far_t;
    f.Xp = b.Xp;
    f.wP[0] = b.wP[0];
    f.wP[1] = b.wP[1];
    f.far = reinterpret_cast<unsigned char *>(c->far_f);
    f.qperm = order ? reinterpret_cast<const int *>(c->pk_perm) : nullptr;
    if (flags) {
      TRY(ensure(&c->far_q, &c->far_q_cap, (size_t)tiles * chunks * 8));
      f.qmask = reinterpret_cast<unsigned int *>(c->far_q);
      TRY(far_counts_alloc(c));
      if (c->far_cnt_host) {  // {far units, all units} of this pass (dust_debug_far_units)
        f.count = reinterpret_cast<unsigned int *>(c->far_cnt);
        f.host_count = c->far_cnt_host;
      }
    }
    const int gx = (tiles + 3) / 4;
    const int want = std::max(1, (3 * device_cus(c) + gx - 1) / gx);  // (three resident workgroups per CU: the launch is bound by the latency of its key loads)
    f.cps = std::max(1, (chunks + want - 1) / want);
    dim3 fgrid(gx, (chunks + f.cps - 1) / f.cps);
    const int nq = std::min(c->N - a.i0, tiles * tq);
#define DUST_LAUNCH_FAR(DPB)                                                                         \
  do {                                                                                               \
    far_lb_kernel<DPB><<<(nq + 63) / 64, 64 * DUST_FAR_LB_WAVES, 0, c->pair_stream>>>(f);            \
    if (flags) {                                                                                     \
      far_prep_kernel<DPB><<<(c->N + 3) / 4, 256, 0, c->pair_stream>>>(f);                           \
      far_flags_kernel<DPB, FusedGeom<DPB>::TQ, true><<<fgrid, 256, far_flags_lds_bytes<DPB>(), c->pair_stream>>>(f); \
    }                                                                                                \
  } while (0)
    if (dpb == 32) DUST_LAUNCH_FAR(32);
    else if (dpb == 64) DUST_LAUNCH_FAR(64);
    else DUST_LAUNCH_FAR(80);
#undef DUST_LAUNCH_FAR
    HIP_TRY(hipGetLastError());
    b.m0 = f.m0;
    if (flags) {
      pk.far = f.far;
      pk.qmask = f.qmask;
      c->far_tiles = tiles;
      c->far_chunks = chunks;
    }
  }
  // the run lists: MERGED below the exact-zero threshold, PLAIN (bit-identical to visiting every chunk) otherwise
  const int JS = packed_slices(c, tiles, chunks);
  b.p.JS = JS;
  pk.N = c->N;
  pk.tiles = tiles;
  pk.chunks = chunks;
  pk.merge = (pk.qmask != nullptr && merge_mode) ? 1 : 0;
  pk.JS = JS;
  pk.JSG = JS;
  pk.ldi = chunks * 64;
  TRY(ensure(&c->pk_idx, &c->pk_idx_cap, (size_t)tiles * pk.ldi));
  TRY(ensure(&c->pk_uoff, &c->pk_uoff_cap, (size_t)tiles * (chunks + 1)));
  TRY(ensure(&c->pk_uq, &c->pk_uq_cap, (size_t)tiles * chunks * 4));
  TRY(ensure(&c->pk_soff, &c->pk_soff_cap, (size_t)tiles * (JS + 1)));
  TRY(ensure(&c->pk_goff, &c->pk_goff_cap, (size_t)tiles * (JS + 1)));
  TRY(ensure(&c->pk_nzu, &c->pk_nzu_cap, ((size_t)tiles * chunks + 3) / 4));
  pk.kidx = reinterpret_cast<int *>(c->pk_idx);
  pk.uoff = reinterpret_cast<int *>(c->pk_uoff);
  pk.uq = reinterpret_cast<unsigned int *>(c->pk_uq);
  pk.soff = reinterpret_cast<int *>(c->pk_soff);
  pk.goff = reinterpret_cast<int *>(c->pk_goff);
  far_pack_kernel<<<tiles, 1024, far_pack_lds_bytes(chunks), c->pair_stream>>>(pk);
  HIP_TRY(hipGetLastError());
  c->pk_tiles = tiles;
  c->pk_umax = chunks;
  c->pk_jsg = JS;
  c->pk_masks = pk.qmask != nullptr;
  c->pk_dense = dense || c->cfg.kernel != DUST_KERNEL_K1_RBF;
  b.kidx = pk.kidx;
  b.ldi = pk.ldi;
  b.uoff = pk.uoff;
  b.uq = c->pk_masks ? pk.uq : nullptr;
  b.soff = pk.soff;
  b.nzu = c->pk_dense ? nullptr : reinterpret_cast<unsigned char *>(c->pk_nzu);
  b.qperm = order ? reinterpret_cast<const int *>(c->pk_perm) : nullptr;
  b.lead = order ? reinterpret_cast<int *>(c->pk_lead) : nullptr;
  c->fused_js = JS;
  TRY(ensure_partials(c, std::max(a.JS, JS)));  // before anything is written: pass 2 adds its pA rows later
  b.p.pA = c->pA;
  b.p.pM = c->pM;
  b.p.pL = c->pL;
  c->prior_js = JS;
  // the kernel blocks are streamed past the caches when they exceed what L2 + Infinity Cache would hand to pass 2 anyway
  const bool stream_k = !pk.merge && (size_t)c->nloc * chunks * 64 * sizeof(float) > ((size_t)128 << 20);
  b.pB = c->pB;
  dim3 grid(tiles, JS);
#define DUST_LAUNCH_FUSED2(MODE, DPB, SK)                                                                                                      \
  do {                                                                                                                                         \
    const size_t lds = pairwise_packed_lds_bytes<DPB>();                                                                                       \
    if (lds > 64 * 1024 && !c->capturing)                                                                                                      \
      HIP_TRY(hipFuncSetAttribute((const void *)pairwise_packed_kernel<MODE, DPB, SK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    pairwise_packed_kernel<MODE, DPB, SK><<<grid, PAIR_NT, lds, c->pair_stream>>>(b);                                                          \
  } while (0)
#define DUST_LAUNCH_FUSED(MODE, DPB)                   \
  do {                                                 \
    if (stream_k) DUST_LAUNCH_FUSED2(MODE, DPB, true); \
    else DUST_LAUNCH_FUSED2(MODE, DPB, false);         \
  } while (0)
#define DUST_PICK_FUSED(MODE)                  \
  do {                                         \
    if (dpb == 32) DUST_LAUNCH_FUSED(MODE, 32); \
    else if (dpb == 64) DUST_LAUNCH_FUSED(MODE, 64); \
    else DUST_LAUNCH_FUSED(MODE, 80);          \
  } while (0)
  if (c->cfg.kernel == DUST_KERNEL_IMQ) DUST_PICK_FUSED(PAIR_IMQ);
  else DUST_PICK_FUSED(PAIR_K1);
#undef DUST_PICK_FUSED
#undef DUST_LAUNCH_FUSED
#undef DUST_LAUNCH_FUSED2
  HIP_TRY(hipGetLastError());
  c->kmat_valid = true;
  return DUST_OK;
}

// Is the log-p pass' far pre-pass (pairwise_far.hpp) worth its launches (~100 us at N = 16384)?  Only while it leaves most blocks out: a
// set with near-duplicates scattered through it (cfg4 after ~30 ticks) has a near pair in nearly every 64 x 64 block.  The pre-pass
// counts its far blocks on the device and its last workgroup reports {far, all, launch number} through pinned memory.  The decision is
// taken from the report of exactly the LAST pre-pass the host issued (it knows their number and waits for that report: by the time the
// next tick's log-p pass is decided, the previous tick's pre-pass is long done), so it is a function of the particle history alone -
// a pass with the pre-pass at the default threshold and one without it differ in their last bits, and whether it runs must not
// depend on when a counter write lands.  A pre-pass that found under 30 % is followed by 15 log-p passes without one,
// then probed again.  One decision per log-p pass; a graph-replayed tick takes it at its entry (dust_svmpc_tick keeps one capture per
// variant).  DUST_FAR=2: always on.
static int far_counts_alloc(dust_ctx *c) {
  if (c->far_cnt_host || c->capturing) return DUST_OK;
  HIP_TRY(hipHostMalloc((void **)&c->far_cnt_host, 8 * sizeof(unsigned int), hipHostMallocCoherent | hipHostMallocMapped));
  memset(c->far_cnt_host, 0, 8 * sizeof(unsigned int));
  TRY(ensure(&c->far_cnt, &c->far_cnt_cap, 8));
  HIP_TRY(hipMemsetAsync(c->far_cnt, 0, 8 * sizeof(unsigned int), c->pair_stream));
  return DUST_OK;
}
static bool logp_pack_ok(const dust_ctx *c);
static bool logp_far_decide(dust_ctx *c) {
  if (c->env.far == 0 || c->env.dense >= 0) return false;
  if (logp_pack_ok(c)) return true;  // (run lists: the pre-pass always pays there)
  if (c->env.far >= 2 || !c->far_cnt_host) return true;
  if (c->far_logp_skip > 0) {
    --c->far_logp_skip;
    return false;
  }
  volatile unsigned int *hc = c->far_cnt_host + 4;
  if (c->far_logp_issued == 0u) return true;
  const double t0 = host_now();
  unsigned int spins = 0u;
  while (hc[2] != c->far_logp_issued) {  // the last pre-pass issued has not reported yet (back-to-back ticks): wait for it
    DUST_CPU_PAUSE();
    if ((++spins & 1023u) == 0u && host_now() - t0 > 0.05) {  // (device stalled or gone: the next synchronisation will say)
      if (getenv("DUST_DEBUG_FAR")) fprintf(stderr, "logp_far_decide: report %u awaited, %u seen\n", c->far_logp_issued, hc[2]);
      return true;
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  const unsigned int fa = hc[0], al = hc[1];
  if (al && (double)fa < 0.3 * (double)al) {
    c->far_logp_skip = 15;
    return false;
  }
  return true;
}

// log p(theta) over run lists (pairwise_packed.hpp pairwise_logp_packed_kernel): the far pre-pass on the updated particles with key masks
// (64-query groups in pass 1's tile order), one list of near keys per group, the product-form log-sum-exp over the listed keys only.
// Large sets only: a rank's 2 048 rows cost 52 us dense and the pre-pass alone 45.
static bool logp_pack_ok(const dust_ctx *c) {
  if (c->env.far == 0 || c->env.dense >= 0 || c->env.pack_merge == 0 || !(c->env.far_t < DUST_FAR_T_EXACT)) return false;
  if (c->env.logp_pack >= 0) return c->env.logp_pack != 0;
  return c->nloc >= 8192;
}
static int launch_pair_logp_packed(dust_ctx *c, const PairArgs &a) {
  const int dpb = std::max(16, ((a.D + 15) / 16) * 16);
  TRY(ensure(&c->xpad, &c->xpad_cap, (size_t)c->N * dpb + 2 * (size_t)c->N));
  const int groups = (a.n_local + 63) / 64, chunks = (c->N + 63) / 64;
  LogpPackedArgs p;
  memset(&p, 0, sizeof p);
  LogpMfmaArgs &b = p.a;
  b.N = c->N;
  b.D = a.D;
  b.i0 = a.i0;
  b.n_local = a.n_local;
  const int JS = std::max(1, std::min(std::min(chunks, 8), (2 * device_cus(c) + groups - 1) / groups));
  b.JS = JS;
  TRY(ensure_partials(c, JS));
  c->prior_js = JS;
  b.X = a.X;
  b.logmix = a.logmix;
  b.sw[0] = a.inv_s[0] * 1.2011224087864498f;  // sqrt(log2 e) / sigma_p
  b.sw[1] = a.da == 2 ? a.inv_s[1] * 1.2011224087864498f : b.sw[0];
  b.Z = c->xpad;
  b.hq = c->xpad + (size_t)c->N * dpb;
  b.hj = b.hq + c->N;
  b.pM = c->pM;
  b.pL = c->pL;
  FarArgs f;
  memset(&f, 0, sizeof f);
  f.N = c->N;
  f.D = a.D;
  f.i0 = a.i0;
  f.n_local = a.n_local;
  f.tiles = groups;
  f.q_rows = groups * 64;
  f.lscale = 1.44269504088896340736f;
  f.chunks = chunks;
  f.X = a.X;
  f.logmix = a.logmix;
  f.sg[0] = b.sw[0];
  f.sg[1] = b.sw[1];
  f.T = c->env.far_t * 1.44269504088896340736f;
  TRY(ensure(&c->far_z, &c->far_z_cap, ((size_t)c->N * far_zh(dpb) + 1) / 2));
  TRY(ensure(&c->far_n, &c->far_n_cap, 3 * (size_t)c->N + chunks));
  TRY(ensure(&c->far_g, &c->far_g_cap, ((size_t)groups * chunks + 3) / 4));
  TRY(ensure(&c->far_q, &c->far_q_cap, (size_t)groups * chunks * 8));
  f.Z = reinterpret_cast<_Float16 *>(c->far_z);
  f.nrm = c->far_n;
  f.lms = c->far_n + c->N;
  f.m0 = c->far_n + 2 * (size_t)c->N;
  f.Xp = b.Z;  // the scaled rows: unit metric
  f.wP[0] = f.wP[1] = 1.0f;
  f.far = reinterpret_cast<unsigned char *>(c->far_g);
  f.qmask = reinterpret_cast<unsigned int *>(c->far_q);
  const bool order = c->pk_order && c->pk_perm && !c->stagewise;  // pass 1's tile order (the one in force this tick)
  f.qperm = order ? reinterpret_cast<const int *>(c->pk_perm) : nullptr;
  TRY(far_counts_alloc(c));
  if (c->far_cnt_host) {
    f.count = reinterpret_cast<unsigned int *>(c->far_cnt) + 4;
    f.host_count = c->far_cnt_host + 4;
    if (!c->capturing) c->far_logp_issued++;
  }
  const int gx = (groups + 3) / 4;
  const int want = std::max(1, (3 * device_cus(c) + gx - 1) / gx);
  f.cps = std::max(1, (chunks + want - 1) / want);
  dim3 fgrid(gx, (chunks + f.cps - 1) / f.cps);
  c->far_groups = groups;
  c->far_gchunks = chunks;
  PackArgs pk;
  memset(&pk, 0, sizeof pk);
  pk.N = c->N;
  pk.tiles = groups;
  pk.chunks = chunks;
  pk.merge = 1;
  pk.JS = JS;
  pk.JSG = JS;
  pk.far = f.far;
  pk.qmask = f.qmask;
  pk.ldi = chunks * 64;
  TRY(ensure(&c->lp_idx, &c->lp_idx_cap, (size_t)groups * pk.ldi));
  TRY(ensure(&c->lp_uoff, &c->lp_uoff_cap, (size_t)groups * (chunks + 1)));
  TRY(ensure(&c->lp_uq, &c->lp_uq_cap, (size_t)groups * chunks * 4));
  TRY(ensure(&c->lp_soff, &c->lp_soff_cap, (size_t)groups * (JS + 1)));
  TRY(ensure(&c->lp_goff, &c->lp_goff_cap, (size_t)groups * (JS + 1)));
  pk.kidx = reinterpret_cast<int *>(c->lp_idx);
  pk.uoff = reinterpret_cast<int *>(c->lp_uoff);
  pk.uq = reinterpret_cast<unsigned int *>(c->lp_uq);
  pk.soff = reinterpret_cast<int *>(c->lp_soff);
  pk.goff = reinterpret_cast<int *>(c->lp_goff);
  p.kidx = pk.kidx;
  p.ldi = pk.ldi;
  p.umax = chunks;
  p.uoff = pk.uoff;
  p.uq = pk.uq;
  p.soff = pk.soff;
  p.qperm = f.qperm;
  dim3 grid(groups, JS);
#define DUST_LAUNCH_LOGPP(DPB)                                                                                                       \
  do {                                                                                                                              \
    logp_prep_far_kernel<DPB><<<(c->N + 3) / 4, 256, 0, c->pair_stream>>>(b, f);                                                    \
    far_lb_kernel<DPB><<<(std::min(c->N - a.i0, f.q_rows) + 63) / 64, 64 * DUST_FAR_LB_WAVES, 0, c->pair_stream>>>(f);              \
    far_flags_kernel<DPB, 64, true><<<fgrid, 256, far_flags_lds_bytes<DPB>(), c->pair_stream>>>(f);                                 \
    far_pack_kernel<<<groups, 1024, far_pack_lds_bytes(chunks), c->pair_stream>>>(pk);                                               \
    pairwise_logp_packed_kernel<DPB><<<grid, 256, pairwise_logp_packed_lds_bytes<DPB>(), c->pair_stream>>>(p);                      \
  } while (0)
  if (dpb == 16) DUST_LAUNCH_LOGPP(16);
  else if (dpb == 32) DUST_LAUNCH_LOGPP(32);
  else if (dpb == 48) DUST_LAUNCH_LOGPP(48);
  else if (dpb == 64) DUST_LAUNCH_LOGPP(64);
  else DUST_LAUNCH_LOGPP(80);
#undef DUST_LAUNCH_LOGPP
  HIP_TRY(hipGetLastError());
  return DUST_OK;
}

// log p(theta) only, large aliased sets (SVMPC.forward): product-form distances on the matrix cores + log-sum-exp
// (pairwise_logp_mfma.hpp).  DUST_LOGP_MFMA=0 keeps the exact-difference pass of pairwise_fused.hpp (development switch).
static int launch_pair_logp_mfma(dust_ctx *c, const PairArgs &a) {
  const int dpb = std::max(16, ((a.D + 15) / 16) * 16);
  TRY(ensure(&c->xpad, &c->xpad_cap, (size_t)c->N * dpb + 2 * (size_t)c->N));
  LogpMfmaArgs b;
  memset(&b, 0, sizeof b);
  b.N = c->N;
  b.D = a.D;
  b.i0 = a.i0;
  b.n_local = a.n_local;
  const int tiles = (a.n_local + 255) / 256, chunks = (c->N + 63) / 64;
  // key slices: one resident round of two workgroups per CU when the set allows it
  int js = std::max(1, std::min(chunks, (2 * device_cus(c)) / tiles));
  const int cps = (chunks + js - 1) / js;
  b.slice = cps * 64;
  b.JS = (c->N + b.slice - 1) / b.slice;
  TRY(ensure_partials(c, b.JS));
  c->prior_js = b.JS;
  b.X = a.X;
  b.logmix = a.logmix;
  b.sw[0] = a.inv_s[0] * 1.2011224087864498f;  // sqrt(log2 e) / sigma_p
  b.sw[1] = a.da == 2 ? a.inv_s[1] * 1.2011224087864498f : b.sw[0];
  b.Z = c->xpad;
  b.hq = c->xpad + (size_t)c->N * dpb;
  b.hj = b.hq + c->N;
  b.pM = c->pM;
  b.pL = c->pL;
  dim3 grid(tiles, b.JS);
  // pairwise_far.hpp: (64-query group, key chunk) blocks whose terms are all negligible against the group's known max logits
  // (DUST_FAR=0 / DUST_DENSE=1: visit all).  Base-2 logits here: log weights and the threshold scaled by log2 e.
  bool flags = c->env.far != 0 && c->env.dense < 0;
  if (flags) {
    TRY(far_counts_alloc(c));
    flags = c->logp_far_decided ? c->logp_far_on : logp_far_decide(c);
  }
  FarArgs f;
  memset(&f, 0, sizeof f);
  c->far_groups = c->far_gchunks = 0;
  dim3 fgrid(1, 1);
  if (flags) {
    f.N = c->N;
    f.D = a.D;
    f.i0 = a.i0;
    f.n_local = a.n_local;
    f.tiles = (a.n_local + 63) / 64;
    f.q_rows = f.tiles * 64;
    f.lscale = 1.44269504088896340736f;
    f.chunks = chunks;
    f.X = a.X;
    f.logmix = a.logmix;
    f.sg[0] = b.sw[0];
    f.sg[1] = b.sw[1];
    f.T = c->env.far_t * 1.44269504088896340736f;
    TRY(ensure(&c->far_z, &c->far_z_cap, ((size_t)c->N * far_zh(dpb) + 1) / 2));
    TRY(ensure(&c->far_n, &c->far_n_cap, 3 * (size_t)c->N + chunks));
    TRY(ensure(&c->far_g, &c->far_g_cap, ((size_t)f.tiles * chunks + 3) / 4));
    f.Z = reinterpret_cast<_Float16 *>(c->far_z);
    f.nrm = c->far_n;
    f.lms = c->far_n + c->N;
    f.m0 = c->far_n + 2 * (size_t)c->N;
    f.Xp = b.Z;  // the scaled rows: unit metric
    f.wP[0] = f.wP[1] = 1.0f;
    f.far = reinterpret_cast<unsigned char *>(c->far_g);
    const int gx = (f.tiles + 3) / 4;
    const int want = std::max(1, (3 * device_cus(c) + gx - 1) / gx);  // (three resident workgroups per CU: the launch is bound by the latency of its key loads)
    f.cps = std::max(1, (chunks + want - 1) / want);
    fgrid = dim3(gx, (chunks + f.cps - 1) / f.cps);
    if (c->far_cnt_host) {
      f.count = reinterpret_cast<unsigned int *>(c->far_cnt) + 4;
      f.host_count = c->far_cnt_host + 4;
      if (!c->capturing) c->far_logp_issued++;  // (a captured launch reports when its graph runs: dust_svmpc_tick counts those)
    }
    b.far = f.far;
    b.groups = f.tiles;
    b.chunks = chunks;
    c->far_groups = f.tiles;
    c->far_gchunks = chunks;
  }
#define DUST_LAUNCH_LOGPM(DPB)                                                                                                 \
  do {                                                                                                                          \
    if (flags) {                                                                                                                \
      logp_prep_far_kernel<DPB><<<(c->N + 3) / 4, 256, 0, c->pair_stream>>>(b, f);                                              \
      far_lb_kernel<DPB><<<(std::min(c->N - a.i0, f.q_rows) + 63) / 64, 64 * DUST_FAR_LB_WAVES, 0, c->pair_stream>>>(f);        \
      far_flags_kernel<DPB, 64, false><<<fgrid, 256, far_flags_lds_bytes<DPB>(), c->pair_stream>>>(f);                                 \
    } else {                                                                                                                    \
      logp_prep_kernel<DPB><<<(c->N + 3) / 4, 256, 0, c->pair_stream>>>(b);                                                     \
    }                                                                                                                           \
    pairwise_logp_mfma_kernel<DPB><<<grid, 256, pairwise_logp_mfma_lds_bytes<DPB>(), c->pair_stream>>>(b);                      \
  } while (0)
  if (dpb == 16) DUST_LAUNCH_LOGPM(16);
  else if (dpb == 32) DUST_LAUNCH_LOGPM(32);
  else if (dpb == 48) DUST_LAUNCH_LOGPM(48);
  else if (dpb == 64) DUST_LAUNCH_LOGPM(64);
  else DUST_LAUNCH_LOGPM(80);
#undef DUST_LAUNCH_LOGPM
  HIP_TRY(hipGetLastError());

  return DUST_OK;
}

// log p(theta) only, large aliased sets (SVMPC.forward): the distance pass + log-sum-exp of pairwise_fused.hpp
static int launch_pair_logp_big(dust_ctx *c, const PairArgs &a, int tiles) {
  const int dpb = fused_dpb(a.D);
  TRY(ensure(&c->xpad, &c->xpad_cap, (size_t)c->N * dpb));
  const bool whole_rows = a.D == dpb;  // (D = 80: the rows are their own padded copy - no pad_rows launch, 7 us per pass at N = 16 384)
  if (!whole_rows) {
    const int n = c->N * dpb;
    pad_rows_kernel<<<(n + 255) / 256, 256, 0, c->pair_stream>>>(a.X, c->xpad, c->N, a.D, dpb);
    HIP_TRY(hipGetLastError());
  }
  PairFusedArgs b;
  memset(&b, 0, sizeof b);
  b.p = a;
  b.Xp = whole_rows ? a.X : c->xpad;
  b.ldp = 8 * cpt_for(a.D);
  b.wP[0] = a.inv_s[0] * a.inv_s[0];
  b.wP[1] = a.da == 2 ? a.inv_s[1] * a.inv_s[1] : b.wP[0];
  dim3 grid(tiles, a.JS);
#define DUST_LAUNCH_LOGP(DPB) pairwise_logp_big_kernel<DPB><<<grid, PAIR_NT, pairwise_logp_big_lds_bytes<DPB>(), c->pair_stream>>>(b)
  if (dpb == 32) DUST_LAUNCH_LOGP(32);
  else if (dpb == 64) DUST_LAUNCH_LOGP(64);
  else DUST_LAUNCH_LOGP(80);
#undef DUST_LAUNCH_LOGP
  HIP_TRY(hipGetLastError());
  return DUST_OK;
}

// pass 2: pA = K x score over the run lists of pass 1 (matrix cores; pairwise_packed.hpp)
static int launch_gram_score(dust_ctx *c, const PairArgs &a, int *JS_out) {
  GramPackedArgs g;
  memset(&g, 0, sizeof g);
  g.N = c->N;
  g.D = c->D;
  g.n_local = c->nloc;
  g.JS = c->pk_jsg;
  *JS_out = g.JS;
  g.ldp = 8 * cpt_for(a.D);
  g.tiles = c->pk_tiles;
  g.umax = c->pk_umax;
  g.Kp = c->kmat;
  g.V = c->score;
  g.pA = c->pA;
  g.kidx = reinterpret_cast<const int *>(c->pk_idx);
  g.ldi = c->pk_umax * 64;
  g.uoff = reinterpret_cast<const int *>(c->pk_uoff);
  g.uq = c->pk_masks ? reinterpret_cast<const unsigned int *>(c->pk_uq) : nullptr;
  g.goff = reinterpret_cast<const int *>(c->pk_goff);
  g.qperm = c->pk_order ? reinterpret_cast<const int *>(c->pk_perm) : nullptr;
  g.nzu = c->pk_dense ? nullptr : reinterpret_cast<const unsigned char *>(c->pk_nzu);
  dim3 grid(g.tiles, g.JS);
  // (its own column padding: whole 16-column MFMA tiles, not the 32 / 64 / 80 of pass 1 - D = 40 runs 3 column tiles instead of 4;
  //  the query tile is pass 1's: 128 / 112 / 96 rows at D <= 32 / 64 / 80)
#define DUST_LAUNCH_GS(DPG, TQ) gram_packed_kernel<DPG, TQ><<<grid, GramGeom<TQ>::NT, gram_packed_lds_bytes<DPG, TQ>(), c->stream>>>(g)
  const int dpg = ((a.D + 15) / 16) * 16;
  if (dpg == 16) DUST_LAUNCH_GS(16, FusedGeom<32>::TQ);
  else if (dpg == 32) DUST_LAUNCH_GS(32, FusedGeom<32>::TQ);
  else if (dpg == 48) DUST_LAUNCH_GS(48, FusedGeom<64>::TQ);
  else if (dpg == 64) DUST_LAUNCH_GS(64, FusedGeom<64>::TQ);
  else DUST_LAUNCH_GS(80, FusedGeom<80>::TQ);
#undef DUST_LAUNCH_GS
  HIP_TRY(hipGetLastError());
  if (c->pk_order) {
    // The next tick's tile order, from the leaders pass 1 noted.  One workgroup, 8 us at 2 048 rows and 42 us at 16 384, and nothing
    // in this tick needs its result: large sets run it on the side stream beside the update and the log-p pass; the tick's last
    // launches (forward_finish_device) - or the next pass 1, whichever comes first - wait for it.
    const size_t lds = query_order_lds_bytes(c->nloc);
    if (lds > 64 * 1024 && !c->capturing)
      HIP_TRY(hipFuncSetAttribute((const void *)query_order_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const bool side = c->stream2 && !c->prof && c->nloc >= 8192;  // (below that the kernel is shorter than the two cross-stream waits: 8 us at 2 048 rows)
    hipStream_t qs = side ? c->stream2 : c->stream;
    if (side) {
      HIP_TRY(hipEventRecord(c->ev_fork, c->stream));
      HIP_TRY(hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
    }
    query_order_kernel<<<1, 1024, lds, qs>>>(reinterpret_cast<int *>(c->pk_lead), reinterpret_cast<int *>(c->pk_perm), c->nloc, c->n0, c->N);
    HIP_TRY(hipGetLastError());
    if (side) {
      HIP_TRY(hipEventRecord(c->ev_order, c->stream2));
      c->order_pending = true;
    }
  }
  return DUST_OK;
}

static int ensure_partials(dust_ctx *c, int JS) {
  const size_t nd = (size_t)JS * c->nloc * 8 * cpt_for(c->D), nn = (size_t)JS * c->nloc;
  TRY(ensure(&c->pA, &c->pA_cap, nd));
  TRY(ensure(&c->pB, &c->pB_cap, nd));
  TRY(ensure(&c->pM, &c->pM_cap, nn));
  TRY(ensure(&c->pL, &c->pL_cap, nn));
  return DUST_OK;
}

// prior pass: writes slice partials (pA, pM, pL); combined by rollout_kernel (merge_prior) or prior_finish_kernel
static int ensure_whitened(dust_ctx *c) {
  const size_t nd = (size_t)c->N * c->D;
  if (!c->theta_w) TRY(dalloc(&c->theta_w, nd));
  if (!c->mu_w) TRY(dalloc(&c->mu_w, nd));
  return DUST_OK;
}
static int launch_whiten(dust_ctx *c) {
  TRY(ensure_whitened(c));
  const int n_pairs = c->N * c->H;
  const float *L = c->cfg.chol_p;
  whiten_rows_kernel<<<(n_pairs + 255) / 256, 256, 0, c->pair_stream>>>(c->theta, c->theta_w, n_pairs, L[0], L[1], L[2]);
  if (!c->mu_aliased) whiten_rows_kernel<<<(n_pairs + 255) / 256, 256, 0, c->pair_stream>>>(c->mu, c->mu_w, n_pairs, L[0], L[1], L[2]);
  HIP_TRY(hipGetLastError());
  return DUST_OK;
}

static int prior_args(dust_ctx *c, PairArgs &a, int *tiles) {
  memset(&a, 0, sizeof a);
  pair_geometry(c, tiles, &a.JS, &a.slice);
  TRY(ensure_partials(c, a.JS));
  a.N = c->N;
  a.D = c->D;
  a.da = c->da;
  a.H = c->H;
  a.i0 = c->n0;
  a.n_local = c->nloc;
  a.X = c->theta;
  a.Y = c->mu_aliased ? c->theta : c->mu;
  a.logmix = c->logmix;
  a.magicD = (uint32_t)((1ull << 32) / (uint64_t)c->D) + 1u;
  for (int d = 0; d < 4; ++d) a.inv_s[d] = 1.0f / c->cfg.sigma_p[d < c->da ? d : 0];
  if (full_cov(c)) {
    // full prior covariance: the pass sees z = L_p^-1 x (stein.hpp whiten_rows_kernel, launched by launch_prior) at unit scale - its
    // squared distance is the Mahalanobis distance and its weighted sum the gradient in whitened coordinates (prior_finish_kernel
    // turns it back: L_p^-T)
    TRY(ensure_whitened(c));
    a.X = c->theta_w;
    a.Y = c->mu_aliased ? c->theta_w : c->mu_w;
    for (int d = 0; d < 4; ++d) a.inv_s[d] = 1.0f;
  }
  a.pA = c->pA;
  a.pM = c->pM;
  a.pL = c->pL;
  a.stamps = c->stamps_dev ? c->stamps_dev + 16 * DUST_K_PRIOR_SCORE : nullptr;
  return DUST_OK;
}

static int launch_prior(dust_ctx *c, bool logp_only = false) {
  PairArgs a;
  int tiles;
  TRY(prior_args(c, a, &tiles));
  Prof p(c, DUST_K_PRIOR_SCORE);
  c->prior_js = 0;  // (launch_pair_fused sets its own slice count)
  if (full_cov(c)) TRY(launch_whiten(c));
  if (logp_only && pair_fused_ok(c)) {  // large aliased set
    if (c->env.logp_mfma != 0 && logp_pack_ok(c)) return launch_pair_logp_packed(c, a);  // ... over run lists of near keys (large sets)
    if (c->env.logp_mfma != 0) return launch_pair_logp_mfma(c, a);  // product-form distances on the matrix cores + log-sum-exp
    return launch_pair_logp_big(c, a, tiles);                         // exact-difference distance pass + log-sum-exp
  }
  if (logp_only && !pair_big_kernel(c)) return launch_pair<PAIR_LOGP>(c, a, tiles);  // SVMPC.forward needs log p(theta) only
  if (!logp_only && pair_fused_ok(c)) return launch_pair_fused(c, a, tiles);              // + repulsion + Gram matrix (pairwise_fused.hpp)
  return launch_pair<PAIR_PRIOR>(c, a, tiles);
}

static K2Args k2_args(dust_ctx *c);
// Fused prior pass + rollout kernel (fused.hpp).  Returns DUST_OK with *done = false when the shape does not qualify.
static int launch_fused(dust_ctx *c, const SampleOpts &o, bool *done) {
  *done = false;
  if (two_pass_family(c)) return DUST_OK;  // (these families run on the launch-per-iteration path: skid.hpp, particle_general.hpp)
  static const bool off = getenv("DUST_NO_FUSE") != nullptr;  // development switch
  if (off || c->no_handoff || c->handoff_banned || c->prof || o.want_actions || o.want_states || o.want_omega || o.costs_in || pair_is_big(c)) return DUST_OK;
  FusedArgs f;
  memset(&f, 0, sizeof f);
  int nt;
  size_t lds_r;
  TRY(prior_args(c, f.pa, &f.tiles));  // first: it sizes the partial buffers that rollout_args' combine descriptor points to
  if (f.pa.JS > 16) return DUST_OK;    // rollout_body's in-register combine holds 16 slices
  SampleOpts oo = o;
  oo.merge_prior = 1;
  TRY(rollout_args(c, oo, f.ra, &nt, &lds_r));
  if (f.ra.tile_scratch || (PAIR_NT % nt) != 0) return DUST_OK;
  if (f.ra.a_reg != 0.0f || f.ra.mw || f.ra.omegaT) return DUST_OK;  // the fused launch carries the LEAN rollout body only
  if (c->cfg.model == DUST_MODEL_PARTICLE && f.ra.dm.with_obstacle && f.ra.grid_words == 0) return DUST_OK;  // (LEAN: map in LDS)
  f.sub_nt = nt;
  f.per_block = PAIR_NT / nt;
  if (c->nloc % f.per_block || PAIR_TI % f.per_block) return DUST_OK;
  const int cpt = cpt_for(c->D);
  if (cpt > 8) return DUST_OK;  // D > 64: separate launches
  const size_t lds_p = pairwise_lds_bytes(PAIR_PRIOR, cpt);
  f.lds_roll_floats = (int)((lds_r / sizeof(float) + 3) & ~(size_t)3);
  size_t lds = std::max(lds_p, (size_t)f.per_block * f.lds_roll_floats * sizeof(float));
  if (lds > 72 * 1024) return DUST_OK;  // keep >= 2 workgroups per CU co-resident
  const bool k2_role = c->k2_inline_want;  // K2: the per-dimension bandwidths of THIS theta as a third role of the launch (bandwidth.hpp k2_bandwidth256)
  if (k2_role) {
    f.k2 = k2_args(c);
    f.n_k2_blocks = (c->D + 7) & ~7;
    lds = std::max(lds, (size_t)K2_BW256_LDS * sizeof(float));
  }
  if (!c->fused_cnt || c->fused_tiles != f.tiles) {
    if (c->capturing) return DUST_OK;
    if (c->fused_cnt) HIP_TRY(hipFree(c->fused_cnt));
    c->fused_cnt = nullptr;
    TRY(dalloc(&c->fused_cnt, ((size_t)f.tiles + 1) * CNT_STRIDE));
    HIP_TRY(hipMemsetAsync(c->fused_cnt, 0, ((size_t)f.tiles + 1) * CNT_STRIDE * sizeof(unsigned int), c->stream));  // incl. the time-out word
    c->fused_tiles = f.tiles;
    c->fused_dirty = true;
  }
  if (c->fused_dirty)  // previous fused launch was not followed by an update kernel (which re-arms the counters)
    HIP_TRY(hipMemsetAsync(c->fused_cnt, 0, (size_t)f.tiles * CNT_STRIDE * sizeof(unsigned int), c->stream));  // (not the time-out word behind the counters)
  f.n_pair_blocks = f.tiles * f.pa.JS;
  f.cnt = c->fused_cnt;
  f.timeout_flag = c->fused_cnt + (size_t)f.tiles * CNT_STRIDE;
  const int grid = f.n_k2_blocks + f.n_pair_blocks + c->nloc / f.per_block;
#define DUST_LAUNCH_FUSED2(MODEL, CPT, GR)                                                                                                \
  do {                                                                                                                                  \
    if (lds > 64 * 1024 && !c->capturing)                                                                                               \
      HIP_TRY(hipFuncSetAttribute((const void *)fused_prior_rollout_kernel<MODEL, CPT, GR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    PersistChain chain(c);                                                                                                              \
    fused_prior_rollout_kernel<MODEL, CPT, GR><<<grid, PAIR_NT, lds, c->stream>>>(f);                                                  \
  } while (0)
#define DUST_LAUNCH_FUSED(MODEL, CPT)                   \
  do {                                                   \
    if (f.ra.G > 1) DUST_LAUNCH_FUSED2(MODEL, CPT, true); \
    else DUST_LAUNCH_FUSED2(MODEL, CPT, false);           \
  } while (0)
  if (c->cfg.model == DUST_MODEL_PENDULUM) {
    if (cpt == 4) DUST_LAUNCH_FUSED(DUST_MODEL_PENDULUM, 4);
    else DUST_LAUNCH_FUSED(DUST_MODEL_PENDULUM, 8);
  } else {
    if (cpt == 4) DUST_LAUNCH_FUSED(DUST_MODEL_PARTICLE, 4);
    else DUST_LAUNCH_FUSED(DUST_MODEL_PARTICLE, 8);
  }
#undef DUST_LAUNCH_FUSED2
#undef DUST_LAUNCH_FUSED
  HIP_TRY(hipGetLastError());
  c->fused_dirty = true;
  c->stein_dirty = false;
  c->actions_valid = false;
  if (k2_role) c->k2_bw_ahead = c->k2_bw_inline = true;
  *done = true;
  return DUST_OK;
}

static int launch_prior_finish(dust_ctx *c, bool want_grad, bool want_logp, bool want_lw = false) {
  PriorFinishArgs f;
  memset(&f, 0, sizeof f);
  f.pm = prior_merge_args(c);
  f.D = c->D;
  f.da = c->da;
  f.i0 = c->n0;
  f.n_local = c->nloc;
  f.grad_lik = c->grad_lik;
  f.grad_pri = want_grad ? c->grad_pri : nullptr;
  f.score = want_grad ? c->score : nullptr;
  f.logp = want_logp ? c->logp : nullptr;
  f.logl = want_lw ? c->logl : nullptr;
  f.lw = want_lw ? c->lw : nullptr;
  const int n = c->nloc * c->D;
  Prof p(c, DUST_K_PRIOR_SCORE);
  prior_finish_kernel<<<(n + 255) / 256, 256, 0, c->stream>>>(f);
  HIP_TRY(hipGetLastError());
  return DUST_OK;
}

static UpdateArgs update_args(dust_ctx *c, int apply) {
  UpdateArgs u;
  memset(&u, 0, sizeof u);
  int tiles, slice;
  pair_geometry(c, &tiles, &u.JS, &slice);
  u.N = c->N;
  u.D = c->D;
  u.i0 = c->n0;
  u.n_local = c->nloc;
  u.optimizer = c->cfg.optimizer;
  u.apply = apply;
  u.lr = c->cfg.lr;
  u.beta1 = c->cfg.adam_beta1;
  u.beta2 = c->cfg.adam_beta2;
  u.eps = c->cfg.adam_eps;
  const float ell = c->cfg.kernel == DUST_KERNEL_IMQ ? c->cfg.imq_ell : 0.69314718055994531f;  // softplus(0) = ln 2 (svmpc.py:78 typo keeps it)
  u.inv_l2 = 1.0f / (ell * ell);
  u.inv_n = 1.0f / c->N;
  u.ldp = 8 * cpt_for(c->D);
  u.pA = c->pA;
  u.pB = c->pB;
  u.phi = c->phi;
  u.theta = c->theta;
  u.theta_out = c->theta;
  u.adam_m = c->adam_m;
  u.adam_v = c->adam_v;
  u.ctr = c->ctr_dev;
  u.fused_cnt = c->fused_cnt;
  u.fused_tiles = c->fused_cnt ? c->fused_tiles : 0;
  return u;
}

static K2Args k2_args(dust_ctx *c) {
  K2Args k;
  memset(&k, 0, sizeof k);
  k.N = c->N;
  k.H = c->H;
  k.da = c->da;
  k.D = c->D;
  k.shared = c->cfg.kernel == DUST_KERNEL_K2_SHARED;
  k.i0 = c->n0;
  k.n_local = c->nloc;
  k.bw_scale = c->cfg.bw_scale;
  k.min_bw = c->k2_min_bw > 0.f ? c->k2_min_bw : 1e-5f;
  k.fixed_h = c->k2_fixed_h;
  k.theta = c->theta;
  k.thetaT = c->thetaT;
  k.score = c->score;
  k.h = c->bw;
  k.h_prev = c->bw;
  k.log_n1 = (float)log((double)c->N + 1.0);
  k.phi = c->phi;
  return k;
}

// Stein pass (+ optimiser step when apply != 0).  K1 / IMQ: tiled partials -> update_kernel; K2: bandwidths + phi, then update.
static int launch_stein_update(dust_ctx *c, int apply, bool in_loop = false /* K2: called by step_device - thetaT may be fresh from iteration k - 1 */) {
  const int n = c->nloc * c->D;
  int jsa = 0;  // slices of pA when pass 2 of the fused pair wrote it
  if (c->cfg.kernel == DUST_KERNEL_K2_IIDMP || c->cfg.kernel == DUST_KERNEL_K2_SHARED) {
    {
      Prof p(c, DUST_K_BANDWIDTH);
      const K2Args k = k2_args(c);
      // the bandwidths came with the prior + rollout launch (k2_inline_want): phi needs no transposed copy either when the update can go
      // to the other particle buffer (unsharded contexts: nobody else holds theta's address)
      const bool inline_bw = c->k2_bw_ahead && c->k2_bw_inline;
      const bool rows = inline_bw && apply && c->nloc == c->N && !c->theta_pinned && c->theta_alt && !k.shared;
      if (!c->k2_bw_ahead) {  // (else: the bandwidths of this theta were computed beside the rollouts - in their launch, or step_device's forked form)
        if (!(in_loop && c->k2_thetaT_fresh)) TRY(launch_transpose(c, c->theta, c->thetaT, c->N, c->D));  // (fresh: the last update launch wrote it)
        TRY(launch_k2_bandwidth(c->stream, k));
      } else if (inline_bw && !rows) {
        if (!(in_loop && c->k2_thetaT_fresh)) TRY(launch_transpose(c, c->theta, c->thetaT, c->N, c->D));
      }
      c->k2_bw_ahead = c->k2_bw_inline = false;
      K2Args kp = k;
      if (apply) {  // the optimiser step rides in the phi kernel (no update_from_phi launch)
        const UpdateArgs u = update_args(c, 1);
        kp.apply = 1;
        kp.optimizer = u.optimizer;
        kp.lr = u.lr;
        kp.beta1 = u.beta1;
        kp.beta2 = u.beta2;
        kp.eps = u.eps;
        kp.theta_rw = u.theta;
        kp.adam_m = u.adam_m;
        kp.adam_v = u.adam_v;
        kp.ctr = u.ctr;
        kp.fused_cnt = u.fused_cnt;
        kp.fused_tiles = u.fused_tiles;
        if (rows) {
          kp.x_rows = c->theta;
          kp.theta_out = c->theta_alt;
        } else if (c->nloc == c->N && c->k2_fixed_h <= 0.f) {  // (unsharded: every column of the transposed copy is written here)
          kp.thetaT_out = c->thetaT_alt;  // (allocated with the context for the K2 kernels: nothing may be allocated inside a graph capture)
        }
      }
      TRY(launch_k2_phi(c->stream, kp));
      c->k2_thetaT_fresh = false;
      if (rows) std::swap(c->theta, c->theta_alt);
      if (kp.thetaT_out) {
        std::swap(c->thetaT, c->thetaT_alt);
        c->k2_thetaT_fresh = true;
      }
    }
    if (apply) c->fused_dirty = false;
    return DUST_OK;
  }
  {
    PairArgs a;
    memset(&a, 0, sizeof a);
    int tiles;
    pair_geometry(c, &tiles, &a.JS, &a.slice);
    TRY(ensure_partials(c, a.JS));
    a.N = c->N;
    a.D = c->D;
    a.da = c->da;
    a.H = c->H;
    a.i0 = c->n0;
    a.n_local = c->nloc;
    a.X = c->theta;
    a.Y = c->theta;
    a.V = c->score;
    a.magicD = (uint32_t)((1ull << 32) / (uint64_t)c->D) + 1u;
    const float ell = c->cfg.kernel == DUST_KERNEL_IMQ ? c->cfg.imq_ell : 0.69314718055994531f;
    for (int d = 0; d < 4; ++d) a.inv_s[d] = 1.0f / ell;
    a.pA = c->pA;
    a.pB = c->pB;
    a.stamps = c->stamps_dev ? c->stamps_dev + 16 * DUST_K_STEIN : nullptr;
    static const bool no_fuse = getenv("DUST_NO_FUSE") != nullptr;  // development switch
    const int cpt = cpt_for(a.D);
    const size_t lds = pairwise_lds_bytes(PAIR_K1, cpt);
    bool fuse = apply && !c->prof && !c->no_handoff && !c->handoff_banned && !no_fuse && cpt <= 8 && !pair_is_big(c);  // D <= 64: >= 2 workgroups per CU co-resident
    if (fuse && (!c->stein_cnt || c->stein_tiles != tiles)) {
      if (c->capturing) fuse = false;
      else {
        if (c->stein_cnt) HIP_TRY(hipFree(c->stein_cnt));
        c->stein_cnt = nullptr;
        TRY(dalloc(&c->stein_cnt, ((size_t)tiles + 2) * CNT_STRIDE));
        HIP_TRY(hipMemsetAsync(c->stein_cnt, 0, ((size_t)tiles + 2) * CNT_STRIDE * sizeof(unsigned int), c->stream));  // incl. the time-out word
        c->stein_tiles = tiles;
        c->stein_dirty = true;  // the rollout launch that preceded this call did not know the buffer
      }
    }
    if (fuse) {
      // Stein tiles + update role in ONE launch (fused.hpp): the update launch and its ramp disappear
      if (c->stein_dirty) HIP_TRY(hipMemsetAsync(c->stein_cnt, 0, ((size_t)tiles + 1) * CNT_STRIDE * sizeof(unsigned int), c->stream));  // (not the time-out word)
      // More key slices for the Stein tiles than for the prior tiles (which share their launch with the rollouts and are better off
      // few): a Stein workgroup is a chain of dependent chunk passes, and at ~500 of them the chip holds 2 waves per SIMD - cfg5
      // (64 tiles x 8 slices of 4 chunks): 26.5 us; 16 slices of 2: 24.7 (round 6).  The update role sums the slices this launch wrote.
      static const bool stein_fine = getenv("DUST_STEIN_JS") == nullptr || atoi(getenv("DUST_STEIN_JS")) != 0;  // development switch
      if (stein_fine) {
        int js2 = a.JS, sl2 = a.slice;
        while (tiles * js2 < 1024 && js2 * 2 <= 16 && sl2 >= 2 * PAIR_JC && (sl2 / 2) % PAIR_JC == 0) {
          sl2 /= 2;
          js2 = (c->N + sl2 - 1) / sl2;
        }
        const size_t need = (size_t)js2 * c->nloc * 8 * cpt;
        if (js2 != a.JS && js2 <= 16 && (!c->capturing || (c->pA_cap >= need && c->pB_cap >= need))) {
          TRY(ensure_partials(c, js2));
          a.JS = js2;
          a.slice = sl2;
          a.pA = c->pA;
          a.pB = c->pB;
        }
      }
      SteinUpdateArgs f;
      memset(&f, 0, sizeof f);
      f.pa = a;
      f.ua = update_args(c, 1);
      f.ua.JS = a.JS;
      // unsharded: the update writes the OTHER theta buffer, so its role may start per query tile while other Stein tiles
      // still read the current one; sharded contexts expose theta's address to the collectives and keep one buffer
      const bool pingpong = c->nloc == c->N && !c->theta_pinned;
      if (pingpong) f.ua.theta_out = c->theta_alt;
      f.wait_all = pingpong ? 0 : 1;
      f.tiles = tiles;
      f.n_pair_blocks = tiles * a.JS;
      f.cnt = c->stein_cnt;
      f.timeout_flag = c->stein_cnt + ((size_t)tiles + 1) * CNT_STRIDE;
      const int grid = f.n_pair_blocks + (n + PAIR_NT - 1) / PAIR_NT;
#define DUST_LAUNCH_SU(MODE, CPT)                                            \
  do {                                                                        \
    PersistChain chain(c);                                                    \
    stein_update_kernel<MODE, CPT><<<grid, PAIR_NT, lds, c->stream>>>(f);     \
  } while (0)
      if (c->cfg.kernel == DUST_KERNEL_IMQ) {
        if (cpt == 4) DUST_LAUNCH_SU(PAIR_IMQ, 4);
        else DUST_LAUNCH_SU(PAIR_IMQ, 8);
      } else {
        if (cpt == 4) DUST_LAUNCH_SU(PAIR_K1, 4);
        else DUST_LAUNCH_SU(PAIR_K1, 8);
      }
#undef DUST_LAUNCH_SU
      HIP_TRY(hipGetLastError());
      if (pingpong) std::swap(c->theta, c->theta_alt);
      c->stein_dirty = true;
      c->fused_dirty = false;
      return DUST_OK;
    }
    Prof p(c, DUST_K_STEIN);
    if (c->kmat_valid && pair_fused_ok(c)) {
      // the prior pass of this iteration already produced the repulsion partials and the Gram matrix of this theta
      TRY(launch_gram_score(c, a, &jsa));
    } else if (c->cfg.kernel == DUST_KERNEL_IMQ) TRY(launch_pair<PAIR_IMQ>(c, a, tiles));
    else TRY(launch_pair<PAIR_K1>(c, a, tiles));
  }
  c->kmat_valid = false;  // (theta moves below; a later Stein pass without a fresh prior pass recomputes)
  UpdateArgs u = update_args(c, apply);
  if (jsa) {  // pA from pass 2 (its own slices), pB from pairwise_packed_kernel
    u.JSA = jsa;
    u.JS = c->fused_js;
  }
  Prof p(c, DUST_K_UPDATE);
  update_kernel<<<(n + 255) / 256, 256, 0, c->stream>>>(u);
  HIP_TRY(hipGetLastError());
  c->fused_dirty = false;
  return DUST_OK;
}

extern "C" int dust_svmpc_phi(dust_ctx *c, const float *costs, const float *actions, float *phi, float *grad_lik, float *grad_pri) {
  if (!c) return fail(DUST_ERR_INVALID, "null ctx");
  TRY(settle_pending(c));
  if ((costs == nullptr) != (actions == nullptr)) return fail(DUST_ERR_INVALID, "pass both costs and actions, or neither");
  HIP_TRY(hipSetDevice(c->cfg.device));
  struct Stagewise {  // (the tile order of pairwise_packed.hpp follows the TICKS' history: a stage-wise call keeps index order)
    dust_ctx *c;
    ~Stagewise() { c->stagewise = false; }
  } stagewise_guard{c};
  c->stagewise = true;
  TRY(launch_prior(c));
  if (costs) {
    // a user-supplied log_p returned (costs, actions): only the weights / score reductions of kernel A run
    const float *nd = nullptr;
    TRY(stage_noise(c, actions, 0, &nd));
    TRY(h2d(c, c->costs_stage, costs, (size_t)c->S * c->N * sizeof(float)));
    SampleOpts o;
    memset(&o, 0, sizeof o);
    o.noise_mode = NOISE_ACTIONS;
    o.noise_dev = nd;
    o.base = c->theta;
    o.costs_in = c->costs_stage;
    o.merge_prior = full_cov(c) ? 0 : 1;
    TRY(launch_rollout(c, o));
    if (full_cov(c)) TRY(launch_prior_finish(c, true, false));
  } else {
    if (!c->have_sample) return fail(DUST_ERR_STATE, "phi without costs/actions needs a prior dust_likelihood_sample");
    TRY(launch_prior_finish(c, true, false));
  }
  TRY(launch_stein_update(c, 0));
  if (phi) TRY(d2h(c, phi, c->phi, (size_t)c->N * c->D * sizeof(float)));
  if (grad_lik) TRY(d2h(c, grad_lik, c->grad_lik, (size_t)c->N * c->D * sizeof(float)));
  if (grad_pri) TRY(d2h(c, grad_pri, c->grad_pri, (size_t)c->N * c->D * sizeof(float)));
  return DUST_OK;
}

// local half of one SVGD iteration: prior partials -> rollout (+ merge) -> score rows of this shard
static int local_score_device(dust_ctx *c, const float *noise_dev, int param_set) {
  // The prior pass and the rollout kernel both only READ theta: fork the prior pass onto the side stream, run the
  // rollout kernel on the main stream, join, then combine (prior_finish: score = grad_lik + grad_pri).  With per-kernel
  // event timing on (dust_profile_enable) everything stays on one stream so the timings remain attributable.
  // Measured on MI355X (round 1): the cross-stream event waits cost more than the overlap gains at cfg2 sizes
  // (2 695 vs 3 010 ticks/s), so the fork/join form is opt-in and the default folds the combine into rollout_kernel.
  static const bool env_on = getenv("DUST_OVERLAP") != nullptr;
  const bool overlap = !c->prof && env_on;
  if (!overlap) {
    SampleOpts of;
    memset(&of, 0, sizeof of);
    of.noise_mode = noise_dev ? NOISE_EPS : NOISE_PHILOX;
    of.noise_dev = noise_dev;
    of.base = c->theta;
    of.update_a_mat = 1;
    of.bump_adam = 1;
    float *sv = c->params_dev;
    if (c->params_dev) c->params_dev += (size_t)param_set * c->M * c->P;
    bool done = false;
    int sf = launch_fused(c, of, &done);
    c->params_dev = sv;
    TRY(sf);
    if (done) {
      c->have_sample = true;
      return DUST_OK;
    }
  }
  if (overlap) {
    HIP_TRY(hipEventRecord(c->ev_fork, c->stream));
    HIP_TRY(hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
    c->pair_stream = c->stream2;
  }
  int sp = launch_prior(c);
  c->pair_stream = c->stream;
  TRY(sp);
  if (overlap) HIP_TRY(hipEventRecord(c->ev_join, c->stream2));
  SampleOpts o;
  memset(&o, 0, sizeof o);
  o.noise_mode = noise_dev ? NOISE_EPS : NOISE_PHILOX;
  o.noise_dev = noise_dev;
  o.base = c->theta;
  o.update_a_mat = 1;
  o.bump_adam = 1;
  const bool split = overlap || full_cov(c);  // (full prior covariance: the merge epilogue of the rollout kernel has no back-substitution)
  o.merge_prior = split ? 0 : 1;
  float *save = c->params_dev;
  if (c->params_dev) c->params_dev += (size_t)param_set * c->M * c->P;
  int s = launch_rollout(c, o);
  c->params_dev = save;
  TRY(s);
  c->have_sample = true;
  if (overlap) HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_join, 0));
  if (split) TRY(launch_prior_finish(c, true, false));
  return DUST_OK;
}

// One launch per SVGD iteration (fused.hpp svgd_iter_kernel) when every role is eligible; otherwise *done stays false and
// the caller runs the two-launch form.
static int launch_iter(dust_ctx *c, const float *noise_dev, int param_set, bool *done) {
  *done = false;
  if (two_pass_family(c)) return DUST_OK;  // (these families run on the launch-per-iteration path: skid.hpp, particle_general.hpp)
  static const bool off = getenv("DUST_NO_FUSE") != nullptr || getenv("DUST_NO_ITER") != nullptr;  // development switches
  if (off || c->no_handoff || c->handoff_banned || c->prof || pair_is_big(c) || c->nloc != c->N || c->theta_pinned || !c->theta_alt) return DUST_OK;
  if (c->cfg.kernel != DUST_KERNEL_K1_RBF && c->cfg.kernel != DUST_KERNEL_IMQ) return DUST_OK;
  const int cpt = cpt_for(c->D);
  if (cpt > 8) return DUST_OK;
  IterArgs f;
  memset(&f, 0, sizeof f);
  TRY(prior_args(c, f.prior, &f.tiles));
  if (f.prior.JS > 16 || f.prior.slice > PAIR_JC) return DUST_OK;  // single-chunk key slices (stein_split_body)
  SampleOpts o;
  memset(&o, 0, sizeof o);
  o.noise_mode = noise_dev ? NOISE_EPS : NOISE_PHILOX;
  o.noise_dev = noise_dev;
  o.base = c->theta;
  o.update_a_mat = 1;
  o.bump_adam = 1;
  o.merge_prior = 1;
  int nt;
  size_t lds_r;
  float *sv = c->params_dev;
  if (c->params_dev) c->params_dev += (size_t)param_set * c->M * c->P;
  int sr = rollout_args(c, o, f.ra, &nt, &lds_r);
  c->params_dev = sv;
  TRY(sr);
  if (f.ra.tile_scratch || (PAIR_NT % nt) != 0 || f.ra.G > 1) return DUST_OK;
  if (f.ra.a_reg != 0.0f || f.ra.mw || f.ra.omegaT) return DUST_OK;  // LEAN rollout body only
  if (c->cfg.model == DUST_MODEL_PARTICLE && f.ra.dm.with_obstacle && f.ra.grid_words == 0) return DUST_OK;  // (LEAN: map in LDS)
  f.ra.rearm = nullptr;
  f.ra.rearm_n = 0;
  f.sub_nt = nt;
  f.per_block = PAIR_NT / nt;
  if (c->nloc % f.per_block || PAIR_JC % f.per_block) return DUST_OK;
  const size_t lds_p = pairwise_lds_bytes(PAIR_K1, cpt);  // >= the prior tile's
  f.lds_roll_floats = (int)((lds_r / sizeof(float) + 3) & ~(size_t)3);
  const size_t lds = std::max(lds_p, (size_t)f.per_block * f.lds_roll_floats * sizeof(float));
  if (lds > 72 * 1024) return DUST_OK;
  const int JS = f.prior.JS, lines = 2 * f.tiles + JS;
  {
    const size_t nd = (size_t)JS * c->nloc * 8 * cpt;
    if (c->pS_cap < nd && c->capturing) return DUST_OK;
    TRY(ensure(&c->pS, &c->pS_cap, nd));
  }
  if (!c->iter_cnt || c->iter_tiles != f.tiles || c->iter_js != JS) {
    if (c->capturing) return DUST_OK;
    if (c->iter_cnt) HIP_TRY(hipFree(c->iter_cnt));
    c->iter_cnt = nullptr;
    TRY(dalloc(&c->iter_cnt, ((size_t)2 * lines + 1) * CNT_STRIDE));
    HIP_TRY(hipMemsetAsync(c->iter_cnt, 0, ((size_t)2 * lines + 1) * CNT_STRIDE * sizeof(unsigned int), c->stream));
    c->iter_tiles = f.tiles;
    c->iter_js = JS;
    c->iter_set = 0;
    if (!c->score_hs) TRY(dalloc(&c->score_hs, (size_t)2 * c->N * c->D));
    HIP_TRY(hipMemsetAsync(c->score_hs, 0xFF, (size_t)2 * c->N * c->D * sizeof(float), c->stream));  // SCORE_SENTINEL in every word
  }
  // Stein role: same geometry as the prior pass (pair_geometry), keys = queries = theta, values = the score rows
  f.stein = f.prior;
  f.stein.Y = c->theta;
  f.stein.V = c->score;
  f.stein.logmix = nullptr;
  {
    const float ell = c->cfg.kernel == DUST_KERNEL_IMQ ? c->cfg.imq_ell : 0.69314718055994531f;
    for (int d = 0; d < 4; ++d) f.stein.inv_s[d] = 1.0f / ell;
  }
  f.stein.pA = c->pS;  // own buffer: pA holds the prior partials, which the rollout role may still be reading
  f.stein.pB = c->pB;
  f.stein.pM = nullptr;
  f.stein.pL = nullptr;
  f.stein.stamps = nullptr;
  f.ua = update_args(c, 1);
  f.ua.pA = c->pS;
  f.ua.theta_out = c->theta_alt;
  f.ua.fused_cnt = nullptr;
  f.ua.fused_tiles = 0;
  f.n_pair_blocks = f.tiles * JS;
  f.n_roll_blocks = c->nloc / f.per_block;
  unsigned int *set = c->iter_cnt + (size_t)c->iter_set * lines * CNT_STRIDE;
  f.cnt_prior = set;
  f.cnt_score = set + (size_t)f.tiles * CNT_STRIDE;
  f.cnt_stein = set + (size_t)(f.tiles + JS) * CNT_STRIDE;
  f.zero_base = c->iter_cnt + (size_t)(1 - c->iter_set) * lines * CNT_STRIDE;
  f.zero_lines = lines;
  f.timeout_flag = c->iter_cnt + (size_t)2 * lines * CNT_STRIDE;
  static const bool score_by_counter = getenv("DUST_SCORE_CNT") != nullptr;  // development switch: counter hand-off of the score rows
  if (!score_by_counter) {
    f.score_pub = c->score_hs + (size_t)c->iter_set * c->N * c->D;
    f.score_reset = reinterpret_cast<unsigned int *>(c->score_hs + (size_t)(1 - c->iter_set) * c->N * c->D);
    f.score_elems = c->N * c->D;
  }
  f.tl = c->tl_dev;
  const int n = c->nloc * c->D;
  const int grid = 2 * f.n_pair_blocks + f.n_roll_blocks + (n + PAIR_NT - 1) / PAIR_NT;
#define DUST_LAUNCH_ITER(MODEL, MODE, CPT)                                          \
  do {                                                                              \
    PersistChain chain(c);                                                          \
    svgd_iter_kernel<MODEL, MODE, CPT><<<grid, PAIR_NT, lds, c->stream>>>(f);       \
  } while (0)
#define DUST_PICK_ITER(MODEL)                                       \
  do {                                                              \
    if (c->cfg.kernel == DUST_KERNEL_IMQ) {                         \
      if (cpt == 4) DUST_LAUNCH_ITER(MODEL, PAIR_IMQ, 4);           \
      else DUST_LAUNCH_ITER(MODEL, PAIR_IMQ, 8);                    \
    } else {                                                        \
      if (cpt == 4) DUST_LAUNCH_ITER(MODEL, PAIR_K1, 4);            \
      else DUST_LAUNCH_ITER(MODEL, PAIR_K1, 8);                     \
    }                                                               \
  } while (0)
  if (c->cfg.model == DUST_MODEL_PENDULUM) DUST_PICK_ITER(DUST_MODEL_PENDULUM);
  else DUST_PICK_ITER(DUST_MODEL_PARTICLE);
#undef DUST_PICK_ITER
#undef DUST_LAUNCH_ITER
  HIP_TRY(hipGetLastError());
  c->iter_set ^= 1;
  std::swap(c->theta, c->theta_alt);
  c->actions_valid = false;
  c->have_sample = true;
  *done = true;
  return DUST_OK;
}

static int step_device(dust_ctx *c, const float *noise_dev, int param_set) {
  bool done = false;
  TRY(launch_iter(c, noise_dev, param_set, &done));
  if (done) return DUST_OK;
  // K2: the per-dimension median bandwidths read theta only.  Rounds 2-3 ran them on a side stream beside the prior + rollout launch;
  // the kernel trace of round 4 (profiles/round4_k2_kernel_stats.csv) shows what that bought: both kernels 30-40 % slower
  // side by side (24 / 24 us against 22 / 17 alone) and 10 us of cross-queue join in front of every phi launch - 55 us per iteration
  // either way.  One stream, launches back to back, is the default now (DUST_K2_FORK=1: the forked form).
  static const bool k2_fork = getenv("DUST_K2_FORK") != nullptr && getenv("DUST_NO_FUSE") == nullptr;  // development switch
  const bool k2 = c->cfg.kernel == DUST_KERNEL_K2_IIDMP || c->cfg.kernel == DUST_KERNEL_K2_SHARED;
  if (k2 && param_set == 0) c->k2_thetaT_fresh = false;  // (a new loop: whatever happened to theta since the last K2 update is not in thetaT)
  if (k2 && k2_fork && !c->prof && c->stream2 && c->own_stream) {
    HIP_TRY(hipEventRecord(c->ev_fork, c->stream));
    HIP_TRY(hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
    const int n = c->N * c->D;
    if (!c->k2_thetaT_fresh) {
      transpose_kernel<<<(n + 255) / 256, 256, 0, c->stream2>>>(c->theta, c->thetaT, c->N, c->D);
      HIP_TRY(hipGetLastError());
    }
    c->k2_thetaT_fresh = false;
    TRY(launch_k2_bandwidth(c->stream2, k2_args(c)));
    HIP_TRY(hipEventRecord(c->ev_join, c->stream2));
    int s = local_score_device(c, noise_dev, param_set);
    HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_join, 0));  // (join even on failure: the capture, if any, must see the side stream return)
    TRY(s);
    c->k2_bw_ahead = true;
    return launch_stein_update(c, 1, true);
  }
  // K2, one dimension per kernel (every demo), N <= 1024: the bandwidths as a role of the prior + rollout launch (when that launch is taken)
  c->k2_inline_want = c->cfg.kernel == DUST_KERNEL_K2_IIDMP && c->env.k2_form != 0 && c->env.k2_form != 3 && !c->prof && c->N <= 1024 &&
                      c->k2_fixed_h <= 0.f && c->da <= 2;
  int sl = local_score_device(c, noise_dev, param_set);
  c->k2_inline_want = false;
  if (sl != DUST_OK) c->k2_bw_ahead = c->k2_bw_inline = false;  // (no phi launch will consume the bandwidths of this theta)
  TRY(sl);
  TRY(launch_stein_update(c, 1, true));
  return DUST_OK;
}

extern "C" int dust_svmpc_optimize(dust_ctx *c, const float *state, int n_steps, const float *eps, const float *params, int flags) {
  if (!c || !state) return fail(DUST_ERR_INVALID, "null argument");
  if (n_steps < 0) return fail(DUST_ERR_INVALID, "n_steps < 0");
  if (c->cfg.dim_p > 0 && !params) return fail(DUST_ERR_INVALID, "params_sampling is on: pass [n_steps][M][P] parameter samples");
  if (comm_active(c)) {
    HIP_TRY(hipSetDevice(c->cfg.device));
    return sharded_steps(c, state, n_steps, eps, params, flags);
  }
  if (c->nloc != c->N) return fail(DUST_ERR_STATE, "sharded context without a communicator: dust_comm_init, or drive it with dust_svmpc_local_score / dust_svmpc_apply_phi");
  HIP_TRY(hipSetDevice(c->cfg.device));
  if (n_steps >= 1) {
    bool done = false;
    TRY(try_persistent(c, state, n_steps, eps, params, flags, /*do_forward=*/false, &done));
    if (done) return DUST_OK;
  }
  TRY(settle_pending(c));  // (plain kernels from here on)
  TRY(upload_state_params(c, state, params, n_steps));
  const size_t slice = ((size_t)c->S * c->N * c->D) >> ((flags & DUST_EPS_F16) ? 1 : 0);  // in floats (binary16: S*N*D is even or n_steps is 1)
  if ((flags & DUST_EPS_F16) && eps && n_steps > 1 && (((size_t)c->S * c->N * c->D) & 1))
    return fail(DUST_ERR_UNSUPPORTED, "binary16 eps for several steps needs an even S*N*D");
  for (int k = 0; k < n_steps; ++k) {
    const float *nd = nullptr;
    TRY(stage_noise(c, eps ? eps + (size_t)k * slice : nullptr, flags, &nd));
    TRY(step_device(c, nd, k));
  }
  return DUST_OK;
}

extern "C" int dust_svmpc_step(dust_ctx *c, const float *state, const float *eps, const float *params, int flags) {
  return dust_svmpc_optimize(c, state, 1, eps, params, flags);
}

static int forward_device(dust_ctx *c) {
  if (!c->have_sample) return fail(DUST_ERR_STATE, "forward(fast_pred=True) needs the costs of a previous optimize step");
  TRY(launch_prior(c, /*logp_only=*/true));
  // unsharded small sets: finalize_kernel combines the partials itself (one launch less); from 4 096 particles on the merge of the
  // slice partials is spread over many workgroups first (the single finalize workgroup took 40 us at N = 16 384 with it, 14 without)
  if (c->nloc == c->N && c->N < 4096) return DUST_OK;
  return launch_prior_finish(c, false, true, /*want_lw=*/true);  // log p and the log-weights logl + logp in one launch (logw_kernel's sum)
}

static FinalizeArgs finalize_args(dust_ctx *c, bool keep_prior) {
  FinalizeArgs f;
  memset(&f, 0, sizeof f);
  f.N = c->N;
  f.D = c->D;
  f.logl = c->logl;
  f.logp = c->logp;
  f.lw = c->lw;
  f.pw = c->pw;
  f.istar = c->istar;
  f.theta = c->theta;
  f.a_seq_out = c->a_seq_out;
  f.logmix = c->logmix;
  f.mixw = c->mixw;
  f.weighted_prior = c->cfg.weighted_prior;
  f.keep_prior = keep_prior ? 1 : 0;
  if (c->nloc == c->N && c->N < 4096) {  // (forward_device: larger sets arrive with lw formed)
    f.merge_logp = 1;
    f.logp_out = c->logp;
    f.pm = prior_merge_args(c);
  }
  return f;
}
static RollArgs roll_args(dust_ctx *c, int steps, int strategy, const float *last_row_dev) {
  RollArgs r;
  memset(&r, 0, sizeof r);
  r.theta = c->theta;
  r.theta_dst = c->theta_home;  // a tick always ends (and a captured tick always starts) on the home buffer
  r.N = c->N;
  r.H = c->H;
  r.da = c->da;
  r.strategy = strategy;
  r.steps = steps;
  r.last_row = last_row_dev;
  r.i0 = c->n0;
  r.n_local = c->nloc;
  r.ctr = c->ctr_dev;
  r.adam_m = c->adam_m;
  r.adam_v = c->adam_v;
  if (c->iter_cnt) {
    r.rearm = c->iter_cnt;
    r.rearm_lines = 2 * (2 * c->iter_tiles + c->iter_js);
    r.hs = reinterpret_cast<unsigned int *>(c->score_hs);
    r.hs_n = 2 * c->N * c->D;
  }
  return r;
}
static void roll_done(dust_ctx *c) {
  c->kmat_valid = false;  // theta rolled: the Gram matrix of pairwise_fused.hpp belongs to the old particles
  if (c->theta != c->theta_home) {
    c->theta_alt = c->theta;
    c->theta = c->theta_home;
  }
}

// weights + argmax (+ prior refresh), then the roll: SVMPC.forward's tail (svmpc.py:190-200)
// `roll_all`: a sharded context that holds every rank's particles (just gathered) rolls ALL N rows itself instead of rolling its
// own and gathering the others' - the roll is a pure function of the row (strategies "repeat" / "mean")
static int forward_finish_device(dust_ctx *c, int steps = -1, const float *last_row_dev = nullptr, bool roll_all = false) {
  TRY(order_join(c));  // (the side stream returns before the tick ends: a captured tick must see it come back)
  Prof p(c, DUST_K_FORWARD);
  const FinalizeArgs f = finalize_args(c, false);
  RollArgs r = roll_args(c, steps, c->cfg.roll_strategy, last_row_dev);
  if (roll_all) {
    r.i0 = 0;
    r.n_local = c->N;
  }
  if (c->theta != c->theta_home && c->cfg.roll_strategy == DUST_ROLL_REPEAT && steps == -1 && c->D <= 128 && !c->prof) {
    // out-of-place roll: independent of finalize (which gathers a_seq from the buffer the roll only reads) -> one launch
    finalize_roll_kernel<<<1 + (c->nloc + 7) / 8, 1024, 0, c->stream>>>(f, r);
    HIP_TRY(hipGetLastError());
  } else {
    finalize_kernel<<<1, 1024, 0, c->stream>>>(f);
    HIP_TRY(hipGetLastError());
    roll_kernel<<<r.n_local, 128, 0, c->stream>>>(r);
    HIP_TRY(hipGetLastError());
  }
  roll_done(c);
  c->mu_aliased = true;  // update_prior: the new GMM's means alias theta from here on (svmpc.py:160-170, svgd.py:87)
  return DUST_OK;
}

static int resample_rows(dust_ctx *c, const float *host_rows, const float **dev) {
  *dev = nullptr;
  if (c->cfg.roll_strategy != DUST_ROLL_RESAMPLE && !host_rows) return DUST_OK;
  if (!host_rows) return fail(DUST_ERR_INVALID, "roll strategy 'resample' needs the last action of a prior sample per particle ([N][da])");
  TRY(ensure(&c->tmp, &c->tmp_cap, (size_t)c->N * c->da));
  TRY(h2d(c, c->tmp, host_rows, (size_t)c->N * c->da * sizeof(float)));
  *dev = c->tmp;
  return DUST_OK;
}

extern "C" int dust_svmpc_get_weights(dust_ctx *c, float *p_weights) {
  if (!c) return fail(DUST_ERR_INVALID, "null ctx");
  TRY(settle_pending(c));
  if (c->nloc != c->N) return fail(DUST_ERR_STATE, "sharded context: get_weights is part of the sharded forward");
  HIP_TRY(hipSetDevice(c->cfg.device));
  TRY(forward_device(c));
  const FinalizeArgs f = finalize_args(c, true);
  finalize_kernel<<<1, 1024, 0, c->stream>>>(f);
  HIP_TRY(hipGetLastError());
  if (p_weights) TRY(tick_outputs(c, nullptr, p_weights));
  return DUST_OK;
}

extern "C" int dust_svmpc_roll(dust_ctx *c, int steps, int strategy, const float *last_row) {
  if (!c) return fail(DUST_ERR_INVALID, "null ctx");
  TRY(settle_pending(c));
  if (strategy < DUST_ROLL_REPEAT || strategy > DUST_ROLL_RESAMPLE) return fail(DUST_ERR_INVALID, "%d is an invalid roll strategy.", strategy);
  if (c->nloc != c->N) return fail(DUST_ERR_STATE, "sharded context: the roll is part of the sharded forward");
  HIP_TRY(hipSetDevice(c->cfg.device));
  if (c->graph_exec || c->graph_exec_alt) graph_drop(c);
  const float *lr = nullptr;
  if (strategy == DUST_ROLL_RESAMPLE) {
    if (!last_row) return fail(DUST_ERR_INVALID, "strategy 'resample' needs last_row [N][da]");
    TRY(ensure(&c->tmp, &c->tmp_cap, (size_t)c->N * c->da));
    TRY(h2d(c, c->tmp, last_row, (size_t)c->N * c->da * sizeof(float)));
    lr = c->tmp;
  }
  // SVMPC.roll rebinds self.theta to a NEW tensor (svmpc.py:142: theta.roll(...)); the prior built by the last update_prior keeps
  // the old storage as its means (svgd.py:87), so a stand-alone roll() leaves the prior where it was until update_prior() is
  // called again.  When the device prior aliases theta, give it its own copy of the pre-roll particles first.
  if (c->mu_aliased) {
    TRY(d2d(c, c->mu, c->theta, (size_t)c->N * c->D * sizeof(float)));
    c->mu_aliased = false;
  }
  const RollArgs r = roll_args(c, steps, strategy, lr);
  roll_kernel<<<c->nloc, 128, 0, c->stream>>>(r);
  HIP_TRY(hipGetLastError());
  roll_done(c);
  return DUST_OK;
}

extern "C" int dust_svmpc_update_prior(dust_ctx *c, const float *weights) {
  if (!c) return fail(DUST_ERR_INVALID, "null ctx");
  TRY(settle_pending(c));
  if (c->nloc != c->N) return fail(DUST_ERR_STATE, "sharded context: the prior refresh is part of the sharded forward");
  HIP_TRY(hipSetDevice(c->cfg.device));
  if (c->graph_exec || c->graph_exec_alt) graph_drop(c);
  std::vector<float> w((size_t)c->N, 1.0f);
  if (weights && c->cfg.weighted_prior) {  // svmpc.py:162-165: mix = ones unless weighted_prior
    for (int i = 0; i < c->N; ++i) {
      if (!(weights[i] >= 0.f)) return fail(DUST_ERR_INVALID, "mixture weights must be >= 0 (torch.distributions.Categorical)");
      w[i] = weights[i];
    }
  }
  TRY(h2d(c, c->mixw, w.data(), c->N * sizeof(float)));
  logmix_kernel<<<1, 1024, 0, c->stream>>>(c->mixw, c->logmix, c->N);
  HIP_TRY(hipGetLastError());
  c->mu_aliased = true;  // get_gmm(self.theta, ...): the means alias theta (svgd.py:87)
  return DUST_OK;
}

extern "C" int dust_svmpc_forward(dust_ctx *c, float *a_seq, float *p_weights) { return dust_svmpc_forward_ex(c, -1, nullptr, a_seq, p_weights); }

extern "C" int dust_svmpc_forward_ex(dust_ctx *c, int steps, const float *resample_last_row, float *a_seq, float *p_weights) {
  if (!c) return fail(DUST_ERR_INVALID, "null ctx");
  if (c->cfg.roll_strategy == DUST_ROLL_RESAMPLE && !resample_last_row)
    return fail(DUST_ERR_INVALID, "roll strategy 'resample': pass the last action of a prior sample per particle ([N][da])");
  if (comm_active(c)) {
    HIP_TRY(hipSetDevice(c->cfg.device));
    TRY(sharded_forward(c));
    if (a_seq || p_weights) TRY(tick_outputs(c, a_seq, p_weights));
    return DUST_OK;
  }
  if (c->nloc != c->N) return fail(DUST_ERR_STATE, "sharded context without a communicator: dust_comm_init, or use dust_svmpc_forward_local / _finish");
  HIP_TRY(hipSetDevice(c->cfg.device));
  if (steps == -1 && !resample_last_row && c->t2_inflight) {
    // behind a one-launch optimize(): the forward as a one-launch tick without iterations - it stays in the device-side order of the
    // one-launch ticks (an optimize() that did not start takes this forward with it into the replay) and needs no host round trip
    bool done = false;
    const float zero_state[4] = {0.f, 0.f, 0.f, 0.f};
    TRY(try_persistent(c, zero_state, 0, nullptr, nullptr, 0, /*do_forward=*/true, &done));
    if (done) {
      if (a_seq || p_weights) TRY(tick_outputs(c, a_seq, p_weights));
      return DUST_OK;
    }
  }
  TRY(settle_pending(c));  // (plain kernels from here on)
  if (steps != -1 && (c->graph_exec || c->graph_exec_alt)) graph_drop(c);
  const float *lr = nullptr;
  TRY(resample_rows(c, resample_last_row, &lr));
  TRY(forward_device(c));
  TRY(forward_finish_device(c, steps, lr));
  if (a_seq || p_weights) TRY(tick_outputs(c, a_seq, p_weights));
  return DUST_OK;
}


// ---------------------------------------------------------------------------------------------------------------
// One launch per control tick, owner-computes form (tick2.hpp): two all-to-all hand-offs per SVGD iteration instead of four.
// *done stays false when the shape / configuration does not qualify (persist.hpp's form or the launch-per-iteration path run).
// the static part of launch_tick2's eligibility (everything but "the prior means alias the particles")
static bool tick2_shape_ok(dust_ctx *c, int n_steps) {
  if (c->env.no_tick2 >= 0 || c->env.no_fuse >= 0 || c->env.no_persist >= 0) return false;
  if (c->prof || c->handoff_banned || c->nloc != c->N || c->theta_pinned || n_steps < 1) return false;
  if ((c->cfg.model != DUST_MODEL_PENDULUM && c->cfg.model != DUST_MODEL_PARTICLE) || two_pass_family(c)) return false;
  if (c->cfg.kernel != DUST_KERNEL_K1_RBF && c->cfg.kernel != DUST_KERNEL_IMQ) return false;
  if (c->cfg.roll_strategy == DUST_ROLL_RESAMPLE || c->cfg.a_reg != 0.0f || c->mw_dev) return false;
  if (c->N % T2_PW || c->D > T2_ROW || c->N / T2_PW > device_cus(c) || c->M > T2_MAXM) return false;
  for (int d = 1; d < c->da; ++d)
    if (c->cfg.sigma_p[d] != c->cfg.sigma_p[0]) return false;
  const int steps = (((c->N + 63) / 64) + 15) & ~15;
  int gw = 0;
  if (c->cfg.model == DUST_MODEL_PARTICLE && c->cfg.with_obstacle && c->grid_bits) {
    const int words = (c->nx * c->ny + 31) / 32;
    if (words > 4096) return false;
    gw = (words + 3) & ~3;
  }
  return (size_t)tick2_lds(c->S, c->D, steps, gw).total * sizeof(float) <= 160 * 1024;
}

static int launch_tick2(dust_ctx *c, const float *state, int n_steps, const float *eps_dev, bool do_forward, bool *done) {
  *done = false;
  if (two_pass_family(c)) return DUST_OK;  // (these families run on the launch-per-iteration path: skid.hpp, particle_general.hpp)
  if (c->env.no_tick2 >= 0 || c->env.no_fuse >= 0 || c->env.no_persist >= 0) return DUST_OK;  // development switches
  if (c->prof || c->nloc != c->N || c->theta_pinned || c->capturing || !c->mu_aliased) return DUST_OK;
  if (c->cfg.kernel != DUST_KERNEL_K1_RBF && c->cfg.kernel != DUST_KERNEL_IMQ) return DUST_OK;
  if (n_steps < 0 || (n_steps == 0 && !do_forward)) return DUST_OK;
  if (c->cfg.roll_strategy == DUST_ROLL_RESAMPLE) return DUST_OK;
  if (do_forward && !c->have_sample && n_steps == 0) return DUST_OK;
  if (c->noise_f16 && eps_dev) return DUST_OK;
  if (c->N % T2_PW || c->D > T2_ROW || c->N / T2_PW > device_cus(c)) return DUST_OK;
  for (int d = 1; d < c->da; ++d)
    if (c->cfg.sigma_p[d] != c->cfg.sigma_p[0]) return DUST_OK;  // isotropic prior scale: one squared distance serves both kernels
  SampleOpts o;
  memset(&o, 0, sizeof o);
  o.noise_mode = eps_dev ? NOISE_EPS : NOISE_PHILOX;
  o.noise_dev = eps_dev;
  o.base = c->theta;
  o.update_a_mat = 1;
  RolloutArgs ra;
  int nt;
  size_t lds_r;
  TRY(rollout_args(c, o, ra, &nt, &lds_r));
  if (ra.a_reg != 0.0f || ra.mw || ra.omegaT) return DUST_OK;
  if (c->cfg.model == DUST_MODEL_PARTICLE && ra.dm.with_obstacle && ra.grid_words == 0) return DUST_OK;
  Tick2Args f;
  memset(&f, 0, sizeof f);
  f.dm = ra.dm;
  f.N = c->N;
  f.S = c->S;
  f.M = c->M;
  f.H = c->H;
  f.D = c->D;
  f.n_iters = n_steps;
  f.do_forward = do_forward ? 1 : 0;
  f.steps = (((c->N + 63) / 64) + 15) & ~15;  // 16-key steps of a pair wave (theta-only pass), in whole groups of 16
  f.lik = ra.lik;
  f.update_a_mat = 1;
  f.eps_base_mode = ra.eps_base_mode;
  f.optimizer = c->cfg.optimizer;
  f.roll_strategy = c->cfg.roll_strategy;
  f.weighted_prior = c->cfg.weighted_prior;
  f.coef_given = ra.coef_given;
  {  // test hook DUST_TICK2_TEST_ABORT: exercises the replay path without a second tenant on the device
    const int every = c->env.tick2_test_abort;
    f.test_abort = every > 0 && ((c->n_tick2 + 1) % every) == 0;
  }
  {  // test hook DUST_TICK2_TEST_TIMEOUT: a tick whose last wait "gives up" - not committed, replayed, context banned
    const int every = c->env.tick2_test_timeout;
    if (every > 0 && ((c->n_tick2 + 1) % every) == 0) f.test_abort = 2;
  }
  f.expect_aborts = c->t2_aborts_seen;
  f.grid_words = c->cfg.model == DUST_MODEL_PARTICLE ? ra.grid_words : 0;
  f.coef_host[0] = ra.coef_host[0];
  f.coef_host[1] = ra.coef_host[1];
  f.alpha = ra.alpha;
  f.temp = ra.temp;
  for (int d = 0; d < 4; ++d) {
    f.chol_a[d] = ra.chol_a[d];
    f.sigma_a[d] = ra.sigma_a[d];
  }
  const size_t lds = (size_t)tick2_lds(c->S, c->D, f.steps, f.grid_words).total * sizeof(float);
  if (lds > 160 * 1024 || c->M > T2_MAXM) return DUST_OK;
  const int mode = c->cfg.kernel == DUST_KERNEL_IMQ ? PAIR_IMQ : PAIR_K1;
  HIP_TRY(hipSetDevice(c->cfg.device));
  if (!c->t2_occ || c->t2_occ_lds != lds) {
    int occ = 0;
    HIP_TRY((hipError_t)tick2_occupancy(c->cfg.model, mode, lds, &occ));
    c->t2_occ = occ > 0 ? occ : -1;
    c->t2_occ_lds = lds;
  }
  const int grid = c->N / T2_PW;
  if (c->t2_occ < 1 || grid > c->t2_occ * device_cus(c)) return DUST_OK;
  if (c->cfg.dim_p > 0 && c->M >= 1 && !c->params_dev) return fail(DUST_ERR_INVALID, "params_sampling is on: pass [n_steps][M][P] parameter samples");
  if (!c->t2_xq || c->t2_gens < n_steps + 1) {  // one [N][32] block per generation of particles / score rows (tick2.hpp t2_ld16)
    if (c->t2_xq) {
      HIP_TRY(hipStreamSynchronize(c->stream));
      (void)hipFree(c->t2_xq);
      (void)hipFree(c->t2_sq);
      c->t2_xq = c->t2_sq = nullptr;
    }
    const int gens = std::max(n_steps + 1, 8);
    TRY(dalloc(&c->t2_xq, (size_t)gens * c->N * T2_ROW));
    TRY(dalloc(&c->t2_sq, (size_t)gens * c->N * T2_ROW));
    c->t2_gens = gens;
  }
  if (!c->t2_cnt) {
    TRY(dalloc(&c->t2_lwq, (size_t)c->N));
    TRY(dalloc(&c->t2_cnt, (size_t)2 * T2_SETS * T2_CNT_STRIDE));
    HIP_TRY(hipMemsetAsync(c->t2_cnt, 0, (size_t)2 * T2_SETS * T2_CNT_STRIDE * sizeof(unsigned int), c->stream));
    c->t2_set = 0;
  }
  const float ell = c->cfg.kernel == DUST_KERNEL_IMQ ? c->cfg.imq_ell : 0.69314718055994531f;
  const float sp = c->cfg.sigma_p[0];
  f.inv_sp2 = 1.0f / (sp * sp);
  f.cP = (float)(-0.5 * 1.4426950408889634 / ((double)sp * (double)sp));
  f.cS = mode == PAIR_IMQ ? (float)(1.0 / ((double)ell * (double)ell)) : (float)(-0.5 * 1.4426950408889634 / ((double)ell * (double)ell));
  {
    const UpdateArgs ua = update_args(c, 1);
    f.inv_l2 = ua.inv_l2;
    f.inv_n = ua.inv_n;
    f.lr = ua.lr;
    f.beta1 = ua.beta1;
    f.beta2 = ua.beta2;
    f.adam_eps = ua.eps;
    const PriorMerge pm = prior_merge_args(c);
    f.log_norm = pm.log_norm;
  }
  for (int k = 0; k < 4; ++k) f.x0[k] = k < c->ds ? state[k] : 0.f;
  f.seed = c->cfg.seed;
  f.ctr = c->ctr_dev;
  f.eps = eps_dev;
  f.eps_stride = (size_t)c->S * c->N * c->D;
  f.params = (c->cfg.dim_p > 0 && c->params_dev) ? c->params_dev : nullptr;
  f.a_seq = c->a_seq;
  f.theta = c->theta;
  f.xq = c->t2_xq;
  f.sq = c->t2_sq;
  f.lwq = c->t2_lwq;
  f.logmix = c->logmix;
  f.mixw = c->mixw;
  f.a_mat = c->a_mat;
  f.adam_m = c->adam_m;
  f.adam_v = c->adam_v;
  f.costsT = c->costsT;
  f.grad_lik = c->grad_lik;
  f.grad_pri = c->grad_pri;
  f.score = c->score;
  f.phi = c->phi;
  f.logl = c->logl;
  f.eta = c->eta;
  f.logp = c->logp;
  f.lw = c->lw;
  f.pw = c->pw;
  f.a_seq_out = c->a_seq_out;
  f.istar = c->istar;
  f.cnt = c->t2_cnt + (size_t)c->t2_set * T2_SETS * T2_CNT_STRIDE;
  f.zero_base = c->t2_cnt + (size_t)(1 - c->t2_set) * T2_SETS * T2_CNT_STRIDE;
  f.status = reinterpret_cast<unsigned int *>(c->outblk + c->out_floats - 32);
  f.tl = c->tl_dev ? c->tl_dev + (size_t)(c->n_tick2 & 1) * 1024 * 128 : nullptr;  // (diagnostic build: consecutive launches stamp alternate halves)
  if (c->t2_launch_mode && c->serve_host && do_forward) {  // closed-loop serving: outputs and the done word go straight to pinned host memory
    f.host_out = c->serve_host;
    f.host_done = const_cast<unsigned int *>(serve_done(c));
    f.host_pw = c->serve_pw ? 1 : 0;
    f.launch_seq = ++c->serve_seq;
    if ((f.launch_seq & 0x7fffffffu) == 0u) f.launch_seq = c->serve_seq = 1u;  // (bit 31 is the "did not start" mark; 0 is never a launch)
    if (c->t2_launch_mode == 2) {  // armed: the state arrives later, through the mailbox
      f.mbox = reinterpret_cast<const unsigned int *>(serve_mbox(c));
      f.mbox_wait = c->serve_wait_ticks;
      for (int k = 0; k < 4; ++k) f.x0[k] = 0.f;
    }
  }
  {
    PersistChain chain(c);
    HIP_TRY((hipError_t)tick2_launch(f, c->cfg.model, mode, grid, lds, c->stream));
  }
  c->t2_set ^= 1;
  c->n_tick2++;
  c->t2_inflight = true;
  c->t2_transactional = true;
  c->t2_grid = grid;
  for (int k = 0; k < 4; ++k) c->t2_state[k] = f.x0[k];
  c->t2_steps = n_steps;
  c->t2_fwd = do_forward;
  c->t2_replayable = eps_dev == nullptr;
  c->t2_mu_aliased = true;
  t2_queue_push(c, c->t2_state, n_steps, do_forward, eps_dev == nullptr, true);
  if (do_forward) c->mu_aliased = true;
  c->actions_valid = false;
  c->have_sample = true;
  *done = true;
  return DUST_OK;
}


// ---------------------------------------------------------------------------------------------------------------
// RCCL (the NCCL API on ROCm), bound at run time: libdust_amd.so carries no link-time dependency on it (single-GPU hosts need
// none), and a process that already holds an RCCL (torch.distributed's) shares that copy instead of loading a second one.
namespace rccl {
typedef int (*get_unique_id_t)(void *);
struct UniqueId {
  char internal[128];
};
typedef int (*all_gather_t)(const void *, void *, size_t, int, void *, hipStream_t);
typedef int (*comm_destroy_t)(void *);
typedef const char *(*get_error_string_t)(int);
static void *handle = nullptr;
static get_unique_id_t get_unique_id = nullptr;
static int (*comm_init_rank)(void **, int, UniqueId, int) = nullptr;
static all_gather_t all_gather = nullptr;
static comm_destroy_t comm_destroy = nullptr;
static get_error_string_t get_error_string = nullptr;
enum { ncclFloat32 = 7 };
static int load() {
  if (all_gather) return DUST_OK;
  const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
  // DUST_RCCL_LIB: the collective library to bind instead (any library with the NCCL 2.x entry points used here: a site's own RCCL
  // build - or tests/fake_rccl, which runs the sharded tick of several PROCESSES on one GPU)
  if (const char *own = getenv("DUST_RCCL_LIB")) {
    handle = dlopen(own, RTLD_NOW | RTLD_LOCAL);
    if (!handle) return fail(DUST_ERR_UNSUPPORTED, "DUST_RCCL_LIB=%s: %s", own, dlerror());
  }
  for (const char *n : names) {  // a copy already mapped into the process first (torch's), then the ROCm installation's
    if (handle) break;
    handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
  }
  for (int i = 0; !handle && i < 3; ++i) handle = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
  if (!handle) return fail(DUST_ERR_UNSUPPORTED, "RCCL not found (librccl.so): %s", dlerror());
  get_unique_id = (get_unique_id_t)dlsym(handle, "ncclGetUniqueId");
  comm_init_rank = (int (*)(void **, int, UniqueId, int))dlsym(handle, "ncclCommInitRank");
  all_gather = (all_gather_t)dlsym(handle, "ncclAllGather");
  comm_destroy = (comm_destroy_t)dlsym(handle, "ncclCommDestroy");
  get_error_string = (get_error_string_t)dlsym(handle, "ncclGetErrorString");
  if (!get_unique_id || !comm_init_rank || !all_gather || !comm_destroy) {
    all_gather = nullptr;
    return fail(DUST_ERR_UNSUPPORTED, "librccl.so lacks the NCCL entry points");
  }
  // the binding assumes the NCCL 2.x ABI (ncclFloat32 = 7, a 128-byte ncclUniqueId passed by value)
  if (int (*get_version)(int *) = (int (*)(int *))dlsym(handle, "ncclGetVersion")) {
    int v = 0;
    if (get_version(&v) == 0 && v > 0 && (v < 20000 ? v / 1000 : v / 10000) != 2) {
      all_gather = nullptr;
      return fail(DUST_ERR_UNSUPPORTED, "RCCL reports version code %d: the binding is written against the NCCL 2.x ABI", v);
    }
  }
  return DUST_OK;
}
static int check(int r, const char *what) {
  if (r == 0) return DUST_OK;
  return fail(DUST_ERR_HIP, "%s failed: %s", what, get_error_string ? get_error_string(r) : "RCCL error");
}
}  // namespace rccl

static void peer_release(dust_ctx *c) {
  PeerState *p = c->peer;
  if (!p) return;
  for (int g = 0; g < PEER_MAX; ++g)
    for (int k = 0; k <= PEER_BUFS; ++k)
      if (p->opened[g][k]) (void)hipIpcCloseMemHandle(p->opened[g][k]);
  if (p->flags_local) (void)hipFree(p->flags_local);
  delete p;
  c->peer = nullptr;
}
static void comm_release(dust_ctx *c) {
  peer_release(c);
  if (c->comm && rccl::comm_destroy) (void)rccl::comm_destroy(c->comm);
  c->comm = nullptr;
}

extern "C" int dust_comm_unique_id(void *id) {
  if (!id) return fail(DUST_ERR_INVALID, "null id");
  TRY(rccl::load());
  return rccl::check(rccl::get_unique_id(id), "ncclGetUniqueId");
}

extern "C" int dust_comm_validate(dust_ctx *c, int rank, int world) {
  if (!c) return fail(DUST_ERR_INVALID, "null argument");
  if (world < 1 || rank < 0 || rank >= world) return fail(DUST_ERR_INVALID, "bad rank %d of %d", rank, world);
  if (c->N % world || c->nloc != c->N / world || c->n0 != rank * (c->N / world))
    return fail(DUST_ERR_INVALID, "context shard [%d,+%d) is not rank %d's equal share of %d particles over %d ranks", c->n0, c->nloc, rank, c->N, world);
  if (c->comm) return fail(DUST_ERR_STATE, "the context already has a communicator");
  return rccl::load();
}

extern "C" int dust_comm_init(dust_ctx *c, const void *id, int rank, int world) {
  if (!c || !id) return fail(DUST_ERR_INVALID, "null argument");
  TRY(dust_comm_validate(c, rank, world));  // (every rank's launcher has agreed on this beforehand: include/dust_amd.h, ABORT RULE)
  HIP_TRY(hipSetDevice(c->cfg.device));
  rccl::UniqueId uid;
  memcpy(uid.internal, id, sizeof uid.internal);
  void *comm = nullptr;
  TRY(rccl::check(rccl::comm_init_rank(&comm, world, uid, rank), "ncclCommInitRank"));
  c->comm = comm;
  c->comm_rank = rank;
  c->comm_world = world;
  // the collectives gather IN PLACE in the context's [N][D] buffers: theta stays in one buffer from here on
  if (c->theta != c->theta_home) {
    TRY(d2d(c, c->theta_home, c->theta, (size_t)c->N * c->D * sizeof(float)));
    c->theta_alt = c->theta;
    c->theta = c->theta_home;
  }
  if (c->graph_exec || c->graph_exec_alt) graph_drop(c);
  c->theta_pinned = true;
  // DUST_PEER_GATHER=1: the tick's all-gathers as direct peer stores (peer_gather.hpp; also dust_comm_peer_gather).  Every rank of a
  // run sees the same environment, so the set-up collective inside is entered by all of them or by none.
  if (env_int("DUST_PEER_GATHER") > 0) {  // (best effort: where the devices cannot map each other EVERY rank is told so and keeps the library's all-gathers)
    const int st = dust_comm_peer_gather(c, 1);
    if (st != DUST_OK && st != DUST_ERR_UNSUPPORTED) return st;
  }
  return DUST_OK;
}

extern "C" int dust_comm_destroy(dust_ctx *c) {
  if (!c) return fail(DUST_ERR_INVALID, "null ctx");
  if (c->comm) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    peer_release(c);
    (void)rccl::comm_destroy(c->comm);
    c->comm = nullptr;
  }
  return DUST_OK;
}

// in-place all-gather of every rank's `count` floats (rank r's piece at buf + r * count), on the context's stream
static int gather_inplace(dust_ctx *c, float *buf, size_t count) {
  return rccl::check(rccl::all_gather(buf + (size_t)c->comm_rank * count, buf, count, rccl::ncclFloat32, c->comm, c->stream), "ncclAllGather");
}

// ---- peer_gather.hpp: the same exchanges as direct peer stores --------------------------------------------------------------------
enum { GATHER_SCORE = 0, GATHER_THETA = 1, GATHER_LW = 2 };
static float *peer_own_buffer(dust_ctx *c, int which) { return which == GATHER_SCORE ? c->score : (which == GATHER_THETA ? c->theta_home : c->lw); }

// Map every peer's three buffers and arrival words (collective: the IPC handles travel through one all-gather of the communicator).
// The OUTCOME is collective too: a rank whose local step fails (no handle for a buffer, a peer's handle that does not open) still takes
// part in both all-gathers and says so in its status word, and every rank returns the same verdict - either all of them store to
// their peers from here on or none does.  (A rank that left before the second all-gather would leave the others hanging inside it.)
extern "C" int dust_comm_peer_gather(dust_ctx *c, int on) {
  if (!c) return fail(DUST_ERR_INVALID, "null ctx");
  if (!c->comm) return fail(DUST_ERR_STATE, "no communicator (dust_comm_init)");
  HIP_TRY(hipSetDevice(c->cfg.device));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (!on) {
    peer_release(c);
    return DUST_OK;
  }
  if (c->peer) return DUST_OK;
  if (c->comm_world > PEER_MAX) return fail(DUST_ERR_UNSUPPORTED, "peer-store all-gathers take up to %d ranks (one node)", (int)PEER_MAX);
  if (c->theta != c->theta_home) return fail(DUST_ERR_STATE, "the particles are not in their home buffer");
  // (the checks above depend on the shape and the call history alone: every rank of a run takes them the same way)
  PeerState *p = new (std::nothrow) PeerState();
  if (!p) return fail(DUST_ERR_HIP, "out of host memory");
  memset((void *)p, 0, sizeof *p);
  p->world = c->comm_world;
  p->rank = c->comm_rank;
  c->peer = p;
  const size_t nflag = (size_t)PEER_ROWS * PEER_MAX + PEER_MAX + 1;
  constexpr size_t HB = sizeof(hipIpcMemHandle_t), PER = (PEER_BUFS + 1) * HB + 64;  // four handles + a status word (padded)
  static_assert(HB % sizeof(float) == 0 && PER % sizeof(float) == 0, "handle block size");
  std::string why;  // this rank's first local failure
  auto note = [&](const char *what) {
    if (why.empty()) why = std::string(what) + ": " + hipGetErrorString(hipGetLastError());
  };
  std::vector<unsigned char> host((size_t)p->world * PER, 0);
  void *mine[PEER_BUFS + 1] = {c->score, c->theta_home, c->lw, nullptr};
  if (hipMalloc((void **)&p->flags_local, nflag * sizeof(unsigned int)) != hipSuccess) {
    p->flags_local = nullptr;
    note("hipMalloc (arrival words)");
  } else if (hipMemset(p->flags_local, 0, nflag * sizeof(unsigned int)) != hipSuccess) {
    note("hipMemset (arrival words)");
  }
  mine[PEER_BUFS] = p->flags_local;
  for (int k = 0; k <= PEER_BUFS && why.empty(); ++k) {
    hipIpcMemHandle_t h;
    if (!mine[k] || hipIpcGetMemHandle(&h, mine[k]) != hipSuccess) note("hipIpcGetMemHandle");
    else memcpy(&host[(size_t)p->rank * PER + k * HB], &h, HB);
  }
  unsigned int ok_word = why.empty() ? 1u : 0u;
  memcpy(&host[(size_t)p->rank * PER + (PEER_BUFS + 1) * HB], &ok_word, sizeof ok_word);
  // all-gather 1: handles + status.  (The staging buffer and the copies around a collective are unlikely to fail; if one does, the
  // communicator itself is in doubt and the error is returned as it is.)
  float *hbuf = nullptr;
  auto bail = [&](int code) {
    if (hbuf) (void)hipFree(hbuf);
    peer_release(c);
    return code;
  };
  if (hipMalloc((void **)&hbuf, host.size()) != hipSuccess) return bail(fail(DUST_ERR_HIP, "hipMalloc (handle exchange)"));
  if (hipMemcpy(hbuf, host.data(), host.size(), hipMemcpyHostToDevice) != hipSuccess) return bail(fail(DUST_ERR_HIP, "hipMemcpy (handle exchange)"));
  int st = rccl::check(rccl::all_gather(hbuf + (size_t)p->rank * (PER / sizeof(float)), hbuf, PER / sizeof(float), rccl::ncclFloat32, c->comm, c->stream), "ncclAllGather (IPC handles)");
  if (st != DUST_OK) return bail(st);
  if (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(host.data(), hbuf, host.size(), hipMemcpyDeviceToHost) != hipSuccess)
    return bail(fail(DUST_ERR_HIP, "handle exchange"));
  bool all_ok = true;
  for (int g = 0; g < p->world; ++g) {
    unsigned int w = 0u;
    memcpy(&w, &host[(size_t)g * PER + (PEER_BUFS + 1) * HB], sizeof w);
    all_ok = all_ok && w == 1u;
  }
  if (env_int("DUST_PEER_TEST_FAIL") == p->rank) why = "test hook DUST_PEER_TEST_FAIL: this rank acts as if a peer's handle had not opened";
  // open the peers' handles (only when every rank published valid ones)
  if (all_ok) {
    for (int g = 0; g < p->world && why.empty(); ++g) {
      for (int k = 0; k <= PEER_BUFS && why.empty(); ++k) {
        void *ptr = mine[k];
        if (g != p->rank) {
          hipIpcMemHandle_t h;
          memcpy(&h, &host[(size_t)g * PER + k * HB], HB);
          if (hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
            note("hipIpcOpenMemHandle");
            break;
          }
          p->opened[g][k] = ptr;
        }
        if (k < PEER_BUFS) p->buf[k][g] = (float *)ptr;
        else p->flags[g] = (unsigned int *)ptr;
      }
    }
  }
  // all-gather 2: the verdicts - and the barrier: nobody stores before every rank has mapped (and zeroed) everything
  ok_word = (all_ok && why.empty()) ? 1u : 0u;
  std::vector<float> verdict((size_t)p->world, 0.f);
  verdict[(size_t)p->rank] = ok_word ? 1.0f : 0.0f;
  if (hipMemcpy(hbuf, verdict.data(), verdict.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return bail(fail(DUST_ERR_HIP, "hipMemcpy (verdicts)"));
  st = rccl::check(rccl::all_gather(hbuf + p->rank, hbuf, 1, rccl::ncclFloat32, c->comm, c->stream), "ncclAllGather (verdicts)");
  if (st != DUST_OK) return bail(st);
  if (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(verdict.data(), hbuf, verdict.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
    return bail(fail(DUST_ERR_HIP, "verdict exchange"));
  int bad_rank = -1;
  for (int g = 0; g < p->world; ++g)
    if (verdict[(size_t)g] != 1.0f && bad_rank < 0) bad_rank = g;
  if (bad_rank >= 0) {
    const std::string mine_why = why;
    (void)bail(DUST_OK);
    return fail(DUST_ERR_UNSUPPORTED, "peer-store all-gathers are not available: rank %d could not map its peers%s%s (every rank keeps the collective library's all-gathers)",
                bad_rank, mine_why.empty() ? "" : " - this rank: ", mine_why.c_str());
  }
  (void)hipFree(hbuf);
  return DUST_OK;
}

static int peer_store(dust_ctx *c, int which, size_t count) {
  PeerState *p = c->peer;
  PeerStoreArgs a;
  memset(&a, 0, sizeof a);
  a.count = count;
  a.offset = (size_t)p->rank * count;
  a.src = peer_own_buffer(c, which) + a.offset;
  for (int g = 0; g < p->world; ++g) {
    a.dst[g] = p->buf[which][g];
    a.flags[g] = p->flags[g];
  }
  a.done = p->flags_local + PEER_ROWS * PEER_MAX;
  a.handshake = which == GATHER_THETA ? 1 : 0;
  a.timeout_ticks = 100000000ull;
  a.world = p->world;
  a.rank = p->rank;
  a.which = which;
  a.seq = ++p->seq[which];
  if (p->world < 2) return DUST_OK;
  const int gx = (int)std::max<size_t>(1, std::min<size_t>(16, ((count >> 2) + 255) / 256));  // (a link wants a few waves, not a full chip)
  peer_store_kernel<<<dim3(gx, p->world - 1), 256, 0, c->stream>>>(a);
  HIP_TRY(hipGetLastError());
  return DUST_OK;
}
static int peer_wait(dust_ctx *c, int which) {
  PeerState *p = c->peer;
  if (p->world < 2) return DUST_OK;
  PeerWaitArgs w;
  w.flags = p->flags_local;
  w.err = p->flags_local + PEER_ROWS * PEER_MAX + PEER_MAX;
  w.world = p->world;
  w.rank = p->rank;
  w.which = which;
  w.seq = p->seq[which];
  w.timeout_ticks = 100000000ull;  // 1 s
  peer_wait_kernel<<<1, 64, 0, c->stream>>>(w);
  HIP_TRY(hipGetLastError());
  return DUST_OK;
}
// a piece that never arrived (a rank died or fell out of step): reported at the next synchronisation, once
static int peer_check(dust_ctx *c) {
  if (!c->peer) return DUST_OK;
  unsigned int e = 0u;
  unsigned int *w = c->peer->flags_local + PEER_ROWS * PEER_MAX + PEER_MAX;
  HIP_TRY(hipMemcpy(&e, w, sizeof e, hipMemcpyDeviceToHost));
  if (!e) return DUST_OK;
  HIP_TRY(hipMemset(w, 0, sizeof e));
  return fail(DUST_ERR_HIP, "peer-store all-gather: a rank's piece did not arrive within 1 s (results of that tick are invalid)");
}
// the exchange `which` of the sharded tick: every rank's `count` floats, in place
static int gather(dust_ctx *c, int which, size_t count) {
  if (c->peer) {
    TRY(peer_store(c, which, count));
    return peer_wait(c, which);
  }
  return gather_inplace(c, peer_own_buffer(c, which), count);
}

// The collectives of one sharded control tick alone - per SVGD iteration the score and the particle all-gather, per tick the
// log-weights - `reps` times back to back on the context's stream between one pair of HIP events: their cost when nothing
// overlaps them (bench.py reports it as comm_us_per_tick beside the sharded rate).  A collective over all ranks.
extern "C" int dust_comm_probe(dust_ctx *c, int n_steps, int reps, double *us_per_tick) {
  if (!c || !us_per_tick || reps < 1 || n_steps < 0) return fail(DUST_ERR_INVALID, "bad argument");
  if (!c->comm) return fail(DUST_ERR_STATE, "no communicator (dust_comm_init)");
  HIP_TRY(hipSetDevice(c->cfg.device));
  const size_t shard = (size_t)c->nloc * c->D;
  if (c->peer) {
    // peer stores go to the mapped buffers themselves.  Between ticks every rank holds the same rows as every other (that is what the
    // exchanges are for), so sending them again changes nothing - the context's particles stay as they are here as well.
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int st = DUST_OK;
    for (int w = 0; w < 2 && st == DUST_OK; ++w) st = gather(c, GATHER_SCORE, shard);
    if (st == DUST_OK && (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)) st = fail(DUST_ERR_HIP, "hipEventCreate");
    if (st == DUST_OK) {
      (void)hipEventRecord(e0, c->stream);
      for (int r = 0; r < reps && st == DUST_OK; ++r) {
        for (int k = 0; k < n_steps && st == DUST_OK; ++k) {
          st = gather(c, GATHER_SCORE, shard);
          if (st == DUST_OK) st = gather(c, GATHER_THETA, shard);
        }
        if (st == DUST_OK) st = gather(c, GATHER_LW, (size_t)c->nloc);
      }
      (void)hipEventRecord(e1, c->stream);
      (void)hipEventSynchronize(e1);
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, e0, e1);
      *us_per_tick = 1e3 * (double)ms / reps;
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipStreamSynchronize(c->stream);
    if (st == DUST_OK) st = peer_check(c);
    return st;
  }
  float *scratch = nullptr;  // gathers into scratch: the context's particles stay untouched
  TRY(dalloc(&scratch, (size_t)c->N * c->D));
  auto gather = [&](size_t count) {
    return rccl::check(rccl::all_gather(scratch + (size_t)c->comm_rank * count, scratch, count, rccl::ncclFloat32, c->comm, c->stream), "ncclAllGather");
  };
  int st = DUST_OK;
  for (int w = 0; w < 2 && st == DUST_OK; ++w) st = gather(shard);  // warm-up
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (st == DUST_OK && (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)) st = fail(DUST_ERR_HIP, "hipEventCreate");
  if (st == DUST_OK) {
    (void)hipEventRecord(e0, c->stream);
    for (int r = 0; r < reps && st == DUST_OK; ++r) {
      for (int k = 0; k < n_steps && st == DUST_OK; ++k) {
        st = gather(shard);
        if (st == DUST_OK) st = gather(shard);
      }
      if (st == DUST_OK) st = gather((size_t)c->nloc);
    }
    (void)hipEventRecord(e1, c->stream);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    *us_per_tick = 1e3 * (double)ms / reps;
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipStreamSynchronize(c->stream);
  (void)hipFree(scratch);
  return st;
}

// One control tick of a SHARDED context with its own communicator (north_star: "an RCCL all-gather over xGMI of
// particle states before the pairwise kernel step"): rank-local rollouts / prior rows / score -> all-gather(score) -> Stein pass
// + update of the rank's rows -> all-gather(theta) (the prior means alias theta: the next prior pass needs every rank's new
// particles) ... -> local log-weights -> all-gather -> finalize + roll of ALL rows (strategy "resample": roll of the rank's rows ->
// all-gather(theta)).  Kernels and collectives share the context's stream: no host synchronisation inside the tick.
// The particle all-gather of iteration k runs on the side stream UNDER the rollouts of iteration k + 1, which read only the rank's own
// rows (an in-place all-gather writes the other ranks' rows of the same buffer); the prior pass - the first consumer of the other
// ranks' particles - waits for it.  Collectives of one communicator never overlap each other: the particle gather starts after the
// update (which follows the score gather) and the next score gather follows the prior pass that waited for it.  After the LAST
// iteration nothing local is left to run beside the gather (forward's log p needs every particle), so it stays on the main stream.
static int sharded_steps(dust_ctx *c, const float *state, int n_steps, const float *eps, const float *params, int flags) {
  if (c->cfg.dim_p > 0 && !params) return fail(DUST_ERR_INVALID, "params_sampling is on: pass [n_steps][M][P] parameter samples");
  TRY(upload_state_params(c, state, params, n_steps));
  const size_t slice = ((size_t)c->S * c->N * c->D) >> ((flags & DUST_EPS_F16) ? 1 : 0);
  const size_t shard = (size_t)c->nloc * c->D;
  const bool overlap = !c->prof && c->stream2 && c->env.no_comm_overlap < 0;  // (development switch)
  // 0: none; 1: the collective library's all-gather on the side stream (ev_join follows it); 2: peer stores issued, arrival not awaited yet
  int gather_in_flight = 0;
  for (int k = 0; k < n_steps; ++k) {
    const float *nd = nullptr;
    TRY(stage_noise(c, eps ? eps + (size_t)k * slice : nullptr, flags, &nd));
    if (gather_in_flight) {
      // rollouts first (own rows only, no merge of prior partials), then the prior pass once every rank's particles have landed
      SampleOpts o;
      memset(&o, 0, sizeof o);
      o.noise_mode = nd ? NOISE_EPS : NOISE_PHILOX;
      o.noise_dev = nd;
      o.base = c->theta;
      o.update_a_mat = 1;
      o.bump_adam = 1;
      float *save = c->params_dev;
      if (c->params_dev) c->params_dev += (size_t)k * c->M * c->P;
      const int st = launch_rollout(c, o);
      c->params_dev = save;
      TRY(st);
      c->have_sample = true;
      if (gather_in_flight == 2) TRY(peer_wait(c, GATHER_THETA));
      else HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_join, 0));
      gather_in_flight = 0;
      TRY(launch_prior(c));
      TRY(launch_prior_finish(c, true, false));
    } else {
      TRY(local_score_device(c, nd, k));
    }
    TRY(gather(c, GATHER_SCORE, shard));
    TRY(launch_stein_update(c, 1));
    if (overlap && k + 1 < n_steps && c->peer) {
      // peer stores need no stream of their own: the pieces travel while the next iteration's rollouts run, the arrival is awaited
      // in front of the prior pass
      TRY(peer_store(c, GATHER_THETA, shard));
      gather_in_flight = 2;
    } else if (overlap && k + 1 < n_steps) {
      HIP_TRY(hipEventRecord(c->ev_fork, c->stream));
      HIP_TRY(hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
      TRY(rccl::check(rccl::all_gather(c->theta + (size_t)c->comm_rank * shard, c->theta, shard, rccl::ncclFloat32, c->comm, c->stream2), "ncclAllGather"));
      HIP_TRY(hipEventRecord(c->ev_join, c->stream2));
      gather_in_flight = 1;
    } else {
      TRY(gather(c, GATHER_THETA, shard));
    }
  }
  return DUST_OK;
}
static int sharded_forward(dust_ctx *c) {
  TRY(forward_device(c));  // rank-local log p and log-weights
  TRY(gather(c, GATHER_LW, (size_t)c->nloc));
  // every rank holds all N particles of this tick (gathered after the last update): "repeat" / "mean" roll each row from itself,
  // so the rank rolls all of them (16 384 rows: ~10 us) and the tick ends without a third 5 MB all-gather
  const bool roll_all = c->cfg.roll_strategy != DUST_ROLL_RESAMPLE && c->theta == c->theta_home;
  TRY(forward_finish_device(c, -1, nullptr, roll_all));
  if (roll_all) return DUST_OK;
  return gather(c, GATHER_THETA, (size_t)c->nloc * c->D);  // the other ranks' rolled rows
}

// the replay inputs of the one-launch tick just enqueued (t2_settle)
static void t2_queue_push(dust_ctx *c, const float *state4, int steps, bool fwd, bool replayable, bool mu_aliased) {
  dust_ctx::T2Replay r;
  for (int k = 0; k < 4; ++k) r.state[k] = state4[k];
  r.steps = steps;
  r.fwd = fwd;
  r.replayable = replayable;
  r.mu_aliased = mu_aliased;
  r.cancelled = false;
  if (c->t2_params_host && c->cfg.dim_p > 0) r.params.assign(c->t2_params_host, c->t2_params_host + (size_t)steps * c->M * c->cfg.dim_p);
  c->t2_queue->push_back(std::move(r));
}

// stage the caller's inputs and try the persistent launch; *done = false -> nothing was launched
static int try_persistent(dust_ctx *c, const float *state, int n_steps, const float *eps, const float *params, int flags, bool do_forward,
                          bool *done) {
  *done = false;
  const bool off = c->env.no_fuse >= 0 || c->env.no_persist >= 0 || c->params_staged;
  if (off || c->no_handoff || c->handoff_banned || c->prof || c->nloc != c->N || c->theta_pinned || n_steps < 0 || (flags & DUST_EPS_F16)) return DUST_OK;
  if (c->cfg.kernel != DUST_KERNEL_K1_RBF && c->cfg.kernel != DUST_KERNEL_IMQ) return DUST_OK;
  if (c->N > 4096 || c->D > 64 || pair_is_big(c) || two_pass_family(c)) return DUST_OK;
  if (c->cfg.dim_p > 0 && !params && n_steps > 0) return fail(DUST_ERR_INVALID, "params_sampling is on: pass [n_steps][M][P] parameter samples");
  // The launchers decide eligibility (slice counts, occupancy, rollout form ...) only after the inputs are staged; what they decline
  // is static for a context in a given state, so a decline is remembered and the same call is not staged twice again
  const unsigned long key = 1ul + (unsigned long)n_steps * 16ul + (do_forward ? 8ul : 0ul) + (c->mu_aliased ? 4ul : 0ul) + (eps ? 2ul : 0ul) +
                            (c->have_sample ? 1ul : 0ul) * 1000003ul;
  if (c->persist_declined == key) return DUST_OK;
  HIP_TRY(hipSetDevice(c->cfg.device));
  if (c->t2_queue->size() >= 4096) TRY(dust_sync(c));  // (an open-loop caller that never reads outputs: settle now and then)
  TRY(upload_state_params(c, nullptr, params, n_steps));
  c->t2_params_host = params;
  const float *eps_dev = eps;
  c->noise_f16 = false;
  if (eps && !(flags & DUST_PTR_DEVICE)) {
    const size_t n = (size_t)n_steps * c->S * c->N * c->D;
    TRY(ensure(&c->noise_stage, &c->noise_cap, n));
    TRY(h2d(c, c->noise_stage, eps, n * sizeof(float)));
    eps_dev = c->noise_stage;
  }
  TRY(launch_tick2(c, state, n_steps, eps_dev, do_forward, done));
  if (!*done && !c->capturing) c->persist_declined = key;  // (a decline during graph capture says nothing about the eager call)
  return DUST_OK;
}

// a_seq / p_weights of the tick (or forward) just enqueued -> host: ONE device-to-host copy into pinned memory and ONE stream
// synchronisation.  A timed-out in-kernel hand-off (persistent tick, or any of the launch-per-iteration fused forms) is an ERROR
// here - the outputs of that tick are invalid and must not reach the plant.
static int tick_outputs(dust_ctx *c, float *a_seq, float *p_weights) {
  unsigned int *lf = reinterpret_cast<unsigned int *>(c->out_pinned + c->out_floats);
  for (int pass = 0; pass < 2; ++pass) {
    HIP_TRY(hipMemcpyAsync(c->out_pinned, c->outblk, c->out_floats * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    lf[0] = lf[1] = lf[2] = 0u;
    if (c->fused_cnt) HIP_TRY(hipMemcpyAsync(lf + 0, c->fused_cnt + (size_t)c->fused_tiles * CNT_STRIDE, 4, hipMemcpyDeviceToHost, c->stream));
    if (c->stein_cnt) HIP_TRY(hipMemcpyAsync(lf + 1, c->stein_cnt + ((size_t)c->stein_tiles + 1) * CNT_STRIDE, 4, hipMemcpyDeviceToHost, c->stream));
    if (c->iter_cnt)
      HIP_TRY(hipMemcpyAsync(lf + 2, c->iter_cnt + (size_t)2 * (2 * c->iter_tiles + c->iter_js) * CNT_STRIDE, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const unsigned int *st = reinterpret_cast<const unsigned int *>(c->out_pinned + c->out_floats - 32);
    if (st[0] && c->t2_inflight && c->t2_transactional) {  // the owner-computes kernel commits nothing when a wait gave up (tick2.hpp COMMIT): replayed below
      bool whole = false;
      TRY(t2_timed_out(c, &whole));
      if (!whole) {
        c->t2_inflight = false;
        return fail(DUST_ERR_HIP, "%s", T2_TORN_MSG);
      }
    } else if (st[0]) {
      HIP_TRY(hipMemset(c->outblk + c->out_floats - 32, 0, 4));
      c->t2_inflight = false;
      return handoff_timeout(c, "persistent tick kernel: a hand-off wait timed out (results of this tick are invalid); the device seems to be shared with another process: this context runs plain kernels from here on");
    }
    // an owner-computes tick whose workgroups were not all resident left the state untouched: run it on the other path, read again
    bool replayed = false;
    TRY(t2_settle(c, st[1], &replayed));
    if (!replayed) break;
  }
  if (lf[0] | lf[1] | lf[2]) {  // reported once: clear the device words (the kernels that set them are not launched again)
    if (c->fused_cnt) (void)hipMemsetAsync(c->fused_cnt + (size_t)c->fused_tiles * CNT_STRIDE, 0, 4, c->stream);
    if (c->stein_cnt) (void)hipMemsetAsync(c->stein_cnt + ((size_t)c->stein_tiles + 1) * CNT_STRIDE, 0, 4, c->stream);
    if (c->iter_cnt) (void)hipMemsetAsync(c->iter_cnt + (size_t)2 * (2 * c->iter_tiles + c->iter_js) * CNT_STRIDE, 0, 4, c->stream);
  }
  if (lf[0] | lf[1] | lf[2]) return handoff_timeout(c, "a fused launch's in-kernel hand-off timed out (results of this tick are invalid); the device seems to be shared with another process: this context runs plain kernels from here on");
  if (a_seq) memcpy(a_seq, c->out_pinned, c->D * sizeof(float));
  if (p_weights) memcpy(p_weights, c->out_pinned + (c->pw - c->outblk), c->N * sizeof(float));
  return DUST_OK;
}

// Ticks of the owner-computes kernel (tick2.hpp) that did not start - workgroup 0 did not see every workgroup arrive within its
// bound, i.e. the grid was not co-resident (another context or process on the device) - have changed nothing.  They are run here,
// late but on unchanged state, through the launch-per-iteration path, which needs no co-residency.  The controller loses no tick;
// dust_tick_stats() reports how often this happened.
static int t2_settle(dust_ctx *c, unsigned int aborts_now, bool *replayed) {
  *replayed = false;
  c->t2_inflight = false;
  const unsigned int n = aborts_now - c->t2_aborts_seen;
  c->t2_aborts_seen = aborts_now;
  std::vector<dust_ctx::T2Replay> q;
  q.swap(*c->t2_queue);
  if (!n) return DUST_OK;
  // every one-launch tick behind the first one that did not start (or commit) aborted as well (`expect_aborts`): the last n entries
  if (n > q.size()) return fail(DUST_ERR_STATE, "%u one-launch tick(s) reported as not started, %zu on record", n, q.size());
  for (size_t i = q.size() - n; i < q.size(); ++i)
    if (!q[i].replayable && !q[i].cancelled)
      return fail(DUST_ERR_HIP, "%u control tick(s) did not start (device shared with another context), one of them with caller-supplied noise: repeat them", n);
  // (the device is shared - that is why the tick did not start - so the replay uses plain kernels only: the fused launch forms spin on
  //  their own workgroups too and could meet the same tenant)
  c->no_handoff = true;
  int st = DUST_OK;
  for (size_t i = q.size() - n; i < q.size() && st == DUST_OK; ++i) {
    const dust_ctx::T2Replay &r = q[i];
    if (r.cancelled) continue;  // (an armed launch nobody supplied a state for: that tick was never asked for)
    if (!r.mu_aliased) c->mu_aliased = false;  // (a context's first tick: its prior means are still c->mu; the replay's forward aliases them)
    st = upload_state_params(c, r.state, r.params.empty() ? nullptr : r.params.data(), r.steps);
    c->noise_f16 = false;
    for (int k = 0; k < r.steps && st == DUST_OK; ++k) st = step_device(c, nullptr, k);
    if (r.fwd && st == DUST_OK) {
      st = forward_device(c);
      if (st == DUST_OK) st = forward_finish_device(c);
    }
    if (st == DUST_OK) c->t2_replays++;
  }
  c->no_handoff = false;
  TRY(st);
  *replayed = true;
  return DUST_OK;
}

extern "C" int dust_set_skid_steer(dust_ctx *c, const dust_skid_config *g) {
  if (!c || !g) return fail(DUST_ERR_INVALID, "null argument");
  if (c->cfg.model != DUST_MODEL_SKID_STEER) return fail(DUST_ERR_STATE, "the context's model is not DUST_MODEL_SKID_STEER");
  TRY(settle_pending(c));
  const dust_param *ps[3] = {&g->x_icr, &g->wheel_radius, &g->axial_distance};
  for (const dust_param *p : ps) {
    if (p->kind < 0 || p->kind > DUST_PARAM_TENSOR0D) return fail(DUST_ERR_INVALID, "bad parameter kind %d", p->kind);
    if (p->kind == DUST_PARAM_SAMPLED && (p->column < 0 || p->column >= c->P)) return fail(DUST_ERR_INVALID, "sampled parameter column %d outside dim_p = %d", p->column, c->P);
  }
  c->skid.x_icr = dev_param(g->x_icr);
  c->skid.wheel_radius = dev_param(g->wheel_radius);
  c->skid.axial_distance = dev_param(g->axial_distance);
  for (int d = 0; d < 2; ++d) {
    if (!(g->min_wheel_speed[d] <= g->max_wheel_speed[d])) return fail(DUST_ERR_INVALID, "wheel speed bounds: min > max");
    c->skid.lo[d] = g->min_wheel_speed[d];
    c->skid.hi[d] = g->max_wheel_speed[d];
    c->skid.w_ctrl[d] = g->w_ctrl[d];
  }
  for (int k = 0; k < 5; ++k) {
    c->skid.goal[k] = g->goal[k];
    c->skid.w_state[k] = g->w_state[k];
    c->skid.w_term[k] = g->w_term[k];
  }
  if (c->graph_exec || c->graph_exec_alt) graph_drop(c);  // (the captured kernel arguments hold the old model)
  return DUST_OK;
}

extern "C" int dust_tick_stats(dust_ctx *c, long long out[4]) {
  if (!c || !out) return fail(DUST_ERR_INVALID, "null argument");
  out[0] = c->n_tick2;
  out[1] = 0;  // (the tiled one-launch tick, persist.hpp: retired in round 6 - no BASELINE configuration took it)
  out[2] = c->n_served;  // (closed-loop serving: ticks answered through the pinned done word)
  out[3] = c->t2_replays;
  return DUST_OK;
}

static void graph_drop(dust_ctx *c) {
  if (c->graph_exec) (void)hipGraphExecDestroy(c->graph_exec);
  if (c->graph) (void)hipGraphDestroy(c->graph);
  if (c->graph_exec_alt) (void)hipGraphExecDestroy(c->graph_exec_alt);
  if (c->graph_alt) (void)hipGraphDestroy(c->graph_alt);
  c->graph_exec = nullptr;
  c->graph = nullptr;
  c->graph_exec_alt = nullptr;
  c->graph_alt = nullptr;
  c->graph_seen = 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Closed-loop serving (include/dust_amd.h dust_svmpc_serve_start).  A control loop is  state -> tick -> first action -> plant -> state:
// the tick's outputs are needed on the host before the next tick's input exists, so launch latency, the device-to-host copy and the
// stream synchronisation all sit on the loop's critical path (round 4: 93 us of kernel, 107 us per closed-loop tick).  While serving,
//  (1) the tick kernel writes a_seq / p_weights straight into pinned host memory and its last workgroup publishes a sequence number
//      there: the host spins on that word - no copy, no synchronisation, and the kernel's own end-of-launch work leaves the path;
//  (2) the NEXT tick is launched while the current one computes, "armed": it hands its particles round, draws its noise and runs the
//      prior pass - none of which needs the plant state - while its rollout waves wait for the state in a pinned mailbox the host
//      writes as soon as the plant has stepped.  Launch latency and the head of the tick overlap the host's share of the loop.
// An armed launch holds every CU while it waits, so: it is bounded (serve_wait_ticks; a launch whose state never comes aborts as one that
// did not start - nothing written), every other entry point of the context cancels it first (settle_pending), and so does whoever
// needs the device next (serve_cancel_device: dust_create, the dynamics filter).  A loop slower than the bound stops being armed.
static int serve_alloc(dust_ctx *c) {
  if (c->serve_host) return DUST_OK;
  // (the GPU polls the mailbox and the host polls the done word: fine-grained coherent memory whatever HIP_HOST_COHERENT says - ADVICE r5)
  HIP_TRY(hipHostMalloc((void **)&c->serve_host, (c->out_floats + 64) * sizeof(float), hipHostMallocCoherent | hipHostMallocMapped));
  memset(c->serve_host, 0, (c->out_floats + 64) * sizeof(float));
  return DUST_OK;
}
static bool serve_call_ok(const dust_ctx *c, int n_steps, const float *eps, const float *params, int flags, const float *a_seq, const float *p_weights) {
  return c->serve_on && n_steps == c->serve_steps && !eps && !params && flags == 0 && (a_seq || p_weights) && c->cfg.dim_p == 0 &&
         !c->handoff_banned && !c->prof && c->own_stream && c->mu_aliased;
}
static int serve_tick(dust_ctx *c, const float *state, float *a_seq, float *p_weights, bool *done) {
  *done = false;
  const int dev = c->cfg.device;
  if (c->cancel_unsettled) TRY(dust_sync(c));  // (the device's abort count must be known before the next launch: expect_aborts)
  HIP_TRY(hipSetDevice(dev));
  TRY(serve_alloc(c));
  if (c->armed && (p_weights != nullptr) && !c->serve_pw) TRY(dust_sync(c));  // (the armed launch will not deliver the weights this call asks for)
  c->serve_pw = p_weights != nullptr;
  unsigned int this_seq = 0u;
  bool was_armed = false;
  {
    std::unique_lock<std::mutex> lk(g_armed_mu[dev >= 0 && dev < DUST_MAX_DEV ? dev : 0]);
    if (c->armed) {  // the tick was launched ahead: hand it its state
      this_seq = c->armed_seq;
      serve_post(c, this_seq, 1u, state);
      c->armed = false;
      if (g_armed[dev] == c) g_armed[dev] = nullptr;
      dust_ctx::T2Replay &r = c->t2_queue->back();
      for (int k = 0; k < 4; ++k) r.state[k] = k < c->ds ? state[k] : 0.f;
      was_armed = true;
      // a state that comes after the launch's bound may find it gone (it then reports "did not start" and is replayed below): a loop
      // that is slower than the bound gains nothing from arming
      if ((host_now() - c->armed_at) * 1e8 > (double)c->serve_wait_ticks) c->serve_misses++;
      else c->serve_misses = 0;
    }
  }
  if (!was_armed) {
    c->t2_launch_mode = 1;
    c->t2_params_host = nullptr;
    int st = launch_tick2(c, state, c->serve_steps, nullptr, true, done);
    c->t2_launch_mode = 0;
    TRY(st);
    if (!*done) return DUST_OK;  // (not a shape / state the one-launch kernel takes: the caller goes on to the regular path)
    this_seq = c->serve_seq;
  }
  *done = true;
  // the next tick, armed, behind this one (single tenant only: a chained launch of another context could stand behind it)
  const bool tenants = dev >= 0 && dev < DUST_MAX_DEV && g_live_ctx[dev].load() >= 2;
  if (!tenants && c->serve_misses < 3 && c->serve_wait_ticks > 0 && c->t2_queue->size() < 4000) {
    bool d2 = false;
    c->t2_launch_mode = 2;
    c->t2_params_host = nullptr;
    const float zero[4] = {0.f, 0.f, 0.f, 0.f};
    int st = launch_tick2(c, zero, c->serve_steps, nullptr, true, &d2);
    c->t2_launch_mode = 0;
    TRY(st);
    if (d2) {
      std::lock_guard<std::mutex> lk(g_armed_mu[dev]);
      c->armed = true;
      c->armed_seq = c->serve_seq;
      c->armed_at = host_now();
      g_armed[dev] = c;
    }
  }
  // wait for this tick's word
  volatile unsigned int *dw = serve_done(c);
  const double t0 = host_now();
  unsigned int spins = 0u, v = 0u;
  bool ok = false, slow = false;
  for (;;) {
    v = *dw;
    if (v == this_seq) {
      ok = true;
      break;
    }
    if (v == (this_seq | 0x80000000u)) break;  // did not start
    DUST_CPU_PAUSE();
    if ((++spins & 1023u) == 0u && host_now() - t0 > 0.25) {  // (no word: a wait inside the launch gave up - 50 ms - or the device is gone)
      slow = true;
      break;
    }
  }
  if (ok) {
    std::atomic_thread_fence(std::memory_order_acquire);
    if (a_seq) memcpy(a_seq, c->serve_host, c->D * sizeof(float));
    if (p_weights) memcpy(p_weights, c->serve_host + (c->pw - c->outblk), c->N * sizeof(float));
    // this tick is committed: it leaves the replay queue (everything in front of it committed too - the launches are ordered).  The
    // launch BEHIND it stays on record while it is armed - or was cancelled by another thread (serve_cancel_device: dust_create, the
    // filter) while this one was spinning above: the device counts that launch as "did not start", and only dust_sync's settle takes the
    // count and the entry off together (dropped here, the next launch aborted on a count the host had never seen).  The
    // canceller writes these fields under the device's mutex, so the bookkeeping takes it too.
    std::lock_guard<std::mutex> lk(g_armed_mu[dev >= 0 && dev < DUST_MAX_DEV ? dev : 0]);
    const bool tail_pending = c->armed || c->cancel_unsettled || (!c->t2_queue->empty() && c->t2_queue->back().cancelled);
    const size_t keep = tail_pending ? 1u : 0u;
    if (c->t2_queue->size() > keep) c->t2_queue->erase(c->t2_queue->begin(), c->t2_queue->end() - keep);
    if (!tail_pending) c->t2_inflight = false;
    c->n_served++;
    if (was_armed) c->n_armed_hit++;
    return DUST_OK;
  }
  (void)slow;
  // the launch did not start (device shared, an earlier tick awaits its replay, its state came too late) or did not commit: the armed
  // launch behind it goes, then the regular machinery reads the status words, replays what has to be replayed and fetches the outputs
  serve_cancel(c);
  TRY(tick_outputs(c, a_seq, p_weights));
  c->cancel_unsettled = false;
  return DUST_OK;
}

extern "C" int dust_svmpc_serve_start(dust_ctx *c, int n_steps, double wait_us) {
  if (!c) return fail(DUST_ERR_INVALID, "null ctx");
  if (n_steps < 1) return fail(DUST_ERR_INVALID, "n_steps must be >= 1");
  if (!(wait_us >= 0.0) || wait_us > 45000.0) return fail(DUST_ERR_INVALID, "wait_us must be in [0, 45000] (below the 50 ms bound of the in-launch hand-offs)");
  TRY(settle_pending(c));
  if (c->cfg.dim_p > 0) return fail(DUST_ERR_UNSUPPORTED, "closed-loop serving takes no per-tick dynamics samples (dim_p = %d): they would have to be staged behind the armed launch", c->cfg.dim_p);
  if (!tick2_shape_ok(c, n_steps)) return fail(DUST_ERR_UNSUPPORTED, "closed-loop serving runs on the one-launch tick (tick2.hpp: K1 / IMQ, N %% 4 == 0, N / 4 <= CUs, H * da <= 32, isotropic prior scale, no control cost, one GPU)");
  HIP_TRY(hipSetDevice(c->cfg.device));
  TRY(serve_alloc(c));
  c->serve_on = true;
  c->serve_steps = n_steps;
  c->serve_wait_ticks = (unsigned long long)(wait_us * 100.0);
  c->serve_misses = 0;
  return DUST_OK;
}
extern "C" int dust_svmpc_serve_stop(dust_ctx *c) {
  if (!c) return fail(DUST_ERR_INVALID, "null ctx");
  c->serve_on = false;
  return settle_pending(c);
}

// One whole control tick.  The kernel chain of a tick is static once the context is warm (same kernels, same pointers:
// state, counters and dynamics samples live in device buffers that are refreshed by copies outside the graph), so from
// the third tick of a given shape on it is replayed as ONE hipGraph launch instead of ~22 kernel launches.
extern "C" int dust_svmpc_tick(dust_ctx *c, const float *state, int n_steps, const float *eps, const float *params, int flags,
                               float *a_seq, float *p_weights) {
  if (!c || !state) return fail(DUST_ERR_INVALID, "null argument");
  if (c->cfg.roll_strategy == DUST_ROLL_RESAMPLE)
    return fail(DUST_ERR_UNSUPPORTED, "roll strategy 'resample' draws from the prior on the host: call dust_svmpc_optimize, then dust_svmpc_forward_ex");
  if (comm_active(c)) {  // sharded context with its own RCCL communicator
    HIP_TRY(hipSetDevice(c->cfg.device));
    TRY(sharded_steps(c, state, n_steps, eps, params, flags));
    TRY(sharded_forward(c));
    if (a_seq || p_weights) {
      TRY(tick_outputs(c, a_seq, p_weights));
      TRY(peer_check(c));  // (a piece that never arrived: these outputs must not reach the plant)
    }
    return DUST_OK;
  }
  // The FIRST tick of a context whose later ticks the owner-computes kernel will serve (tick2.hpp needs the prior means aliased to the
  // particles, which this tick's forward establishes): plain kernels, nothing that spins on its own grid.  The tiled one-launch kernel
  // would serve it 0.4 ms faster - once per context - but has no residency proof: with another process on the device it would time
  // out; from the second tick on tick2.hpp's start barrier covers that case.
  if (!c->mu_aliased && tick2_shape_ok(c, n_steps)) {
    c->no_handoff = true;
    int st = dust_svmpc_optimize(c, state, n_steps, eps, params, flags);
    if (st == DUST_OK) st = forward_device(c);
    if (st == DUST_OK) st = forward_finish_device(c);
    c->no_handoff = false;
    TRY(st);
    if (a_seq || p_weights) TRY(tick_outputs(c, a_seq, p_weights));
    return DUST_OK;
  }
  if (serve_call_ok(c, n_steps, eps, params, flags, a_seq, p_weights)) {  // closed-loop serving: done word, next tick launched ahead
    bool done = false;
    TRY(serve_tick(c, state, a_seq, p_weights, &done));
    if (done) return DUST_OK;
  } else if (c->armed || c->cancel_unsettled) {
    TRY(settle_pending(c));  // (a call the armed launch was not made for)
  }
  {  // one persistent launch for the whole tick when the shape allows it (persist.hpp)
    bool done = false;
    TRY(try_persistent(c, state, n_steps, eps, params, flags, /*do_forward=*/true, &done));
    if (done) {
      if (a_seq || p_weights) TRY(tick_outputs(c, a_seq, p_weights));
      return DUST_OK;
    }
  }
  TRY(settle_pending(c));  // (plain kernels / a captured graph from here on)
  static const bool no_graph = getenv("DUST_NO_GRAPH") != nullptr;  // development switch
  // (a second context on the device: the in-launch hand-off kernels are chained across streams at every launch - PersistChain - which a
  //  replayed capture would not be)
  const bool tenants = c->cfg.device >= 0 && c->cfg.device < DUST_MAX_DEV && g_live_ctx[c->cfg.device].load() >= 2;
  const bool graphable = !no_graph && !c->prof && c->nloc == c->N && n_steps > 0 && (eps == nullptr || (flags & DUST_PTR_DEVICE)) &&
                         c->mu_aliased && c->own_stream && !tenants && c->cz_next >= c->cz_sets /* no recorded control noise pending */;
  struct FarTick {  // this tick's log-p pre-pass decision, taken once (see logp_far_decide)
    dust_ctx *c;
    ~FarTick() { c->logp_far_decided = false; }
  } far_tick{c};
  if (graphable) {
    c->logp_far_on = logp_far_decide(c);
    c->logp_far_decided = true;
  }
  if (!graphable || c->graph_steps != n_steps || c->graph_eps != (const void *)eps || c->graph_flags != flags ||
      ((c->graph_exec || c->graph_exec_alt) && c->graph_theta != c->theta)) {
    if (c->graph_exec || c->graph_exec_alt) graph_drop(c);
    c->graph_steps = n_steps;
    c->graph_eps = (const void *)eps;
    c->graph_flags = flags;
    c->graph_seen = 0;
  }
  if (graphable && (c->graph_exec || c->graph_exec_alt) && c->graph_far != (int)c->logp_far_on) {  // the other variant: swap it in (or capture it below)
    std::swap(c->graph, c->graph_alt);
    std::swap(c->graph_exec, c->graph_exec_alt);
    c->graph_far = (int)c->logp_far_on;
  }
  if (graphable && c->graph_exec) {
    if (n_steps < 0) return fail(DUST_ERR_INVALID, "n_steps < 0");
    if (c->cfg.dim_p > 0 && !params) return fail(DUST_ERR_INVALID, "params_sampling is on: pass [n_steps][M][P] parameter samples");
    HIP_TRY(hipSetDevice(c->cfg.device));
    TRY(upload_state_params(c, state, params, n_steps));
    HIP_TRY(hipGraphLaunch(c->graph_exec, c->stream));
    if (c->graph_far && c->far_cnt_host) c->far_logp_issued++;  // (the captured pre-pass reports once per replay)
  } else if (graphable && c->graph_seen >= 1) {
    // warm (every lazily sized buffer exists): capture this tick, then launch the capture
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (c->cfg.dim_p > 0 && !params) return fail(DUST_ERR_INVALID, "params_sampling is on: pass [n_steps][M][P] parameter samples");
    TRY(upload_state_params(c, state, params, n_steps));
    c->capturing = true;
    c->graph_theta = c->theta;
    c->graph_far = (int)c->logp_far_on;
    hipError_t e = hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal);
    int st = DUST_OK;
    if (e == hipSuccess) {
      const size_t slice = ((size_t)c->S * c->N * c->D) >> ((flags & DUST_EPS_F16) ? 1 : 0);
      c->noise_f16 = eps && (flags & DUST_EPS_F16);
      for (int k = 0; k < n_steps && st == DUST_OK; ++k) st = step_device(c, eps ? eps + (size_t)k * slice : nullptr, k);
      if (st == DUST_OK) st = forward_device(c);
      if (st == DUST_OK) st = forward_finish_device(c);
      hipGraph_t g = nullptr;
      hipError_t e2 = hipStreamEndCapture(c->stream, &g);
      if (st == DUST_OK && e2 == hipSuccess && g) {
        c->graph = g;
        if (hipGraphInstantiate(&c->graph_exec, c->graph, nullptr, nullptr, 0) != hipSuccess) {
          (void)hipGraphDestroy(c->graph);
          c->graph = nullptr;
          c->graph_exec = nullptr;
        }
      } else if (g) {
        (void)hipGraphDestroy(g);
      }
    }
    c->capturing = false;
    TRY(st);
    if (c->graph_exec) {
      HIP_TRY(hipGraphLaunch(c->graph_exec, c->stream));
      if (c->graph_far && c->far_cnt_host) c->far_logp_issued++;
    } else {  // capture unavailable: run the tick eagerly (nothing was executed during the failed capture)
      (void)hipGetLastError();
      c->graph_seen = -1000000;  // do not retry every tick
      const size_t slice = ((size_t)c->S * c->N * c->D) >> ((flags & DUST_EPS_F16) ? 1 : 0);
      c->noise_f16 = eps && (flags & DUST_EPS_F16);
      for (int k = 0; k < n_steps; ++k) TRY(step_device(c, eps ? eps + (size_t)k * slice : nullptr, k));
      TRY(forward_device(c));
      TRY(forward_finish_device(c));
    }
  } else {
    TRY(dust_svmpc_optimize(c, state, n_steps, eps, params, flags));
    TRY(forward_device(c));
    TRY(forward_finish_device(c));
    if (graphable) c->graph_seen++;
  }
  if (a_seq || p_weights) TRY(tick_outputs(c, a_seq, p_weights));
  return DUST_OK;
}

extern "C" int dust_disco_step(dust_ctx *c, int strategy, int steps, const float *ext, float *next) {
  if (!c || !next) return fail(DUST_ERR_INVALID, "null argument");
  TRY(settle_pending(c));
  if (strategy < 0 || strategy > 2 || (strategy == DUST_STEP_EXTERNAL && !ext)) return fail(DUST_ERR_INVALID, "Invalid value for strategy.");
  if (steps < 1 || steps > c->H) return fail(DUST_ERR_INVALID, "steps out of range");
  HIP_TRY(hipSetDevice(c->cfg.device));
  if (c->have_sample) {
    amix_kernel<<<1, 1024, 0, c->stream>>>(c->eta, c->a_mix, c->N);
    HIP_TRY(hipGetLastError());
  }
  StepArgs s;
  memset(&s, 0, sizeof s);
  s.N = c->N;
  s.H = c->H;
  s.da = c->da;
  s.strategy = strategy;
  s.steps = steps;
  for (int d = 0; d < 4; ++d) {
    s.min_a[d] = c->cfg.min_a[d];
    s.max_a[d] = c->cfg.max_a[d];
  }
  s.a_mix = c->a_mix;
  if (ext) {
    TRY(h2d(c, c->a_seq_out, ext, c->D * sizeof(float)));
    s.ext = c->a_seq_out;
  }
  s.a_mat = c->a_mat;
  s.a_seq = c->a_seq;
  s.next = c->tmp;
  disco_step_kernel<<<1, 1024, c->D * sizeof(float), c->stream>>>(s);
  HIP_TRY(hipGetLastError());
  const int n = c->N * c->da;
  amat_roll_kernel<<<(n + 255) / 256, 256, 0, c->stream>>>(c->a_mat, c->N, c->H, c->da, steps);
  HIP_TRY(hipGetLastError());
  return d2h(c, next, c->tmp, (size_t)steps * c->da * sizeof(float));
}

// ---------------------------------------------------------------------------------------------------------------
// multi-GPU split of one SVGD iteration around the RCCL all-gather of [theta | score]
extern "C" int dust_gather_buffers(dust_ctx *c, void **theta_all, void **score_all, size_t *shard_bytes) {
  if (!c) return fail(DUST_ERR_INVALID, "null ctx");
  // the caller keeps these addresses (in-place collectives): from here on theta stays in ONE buffer (no ping-pong)
  if (c->theta != c->theta_home) {
    HIP_TRY(hipSetDevice(c->cfg.device));
    TRY(d2d(c, c->theta_home, c->theta, (size_t)c->N * c->D * sizeof(float)));
    c->theta_alt = c->theta;
    c->theta = c->theta_home;
  }
  if (c->graph_exec || c->graph_exec_alt) graph_drop(c);
  c->theta_pinned = true;
  if (theta_all) *theta_all = c->theta;
  if (score_all) *score_all = c->score;
  if (shard_bytes) *shard_bytes = (size_t)c->nloc * c->D * sizeof(float);
  return DUST_OK;
}
extern "C" int dust_svmpc_local_score(dust_ctx *c, const float *state, const float *eps, const float *params, int flags) {
  if (!c || !state) return fail(DUST_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(c->cfg.device));
  TRY(upload_state_params(c, state, params, 1));
  const float *nd = nullptr;
  TRY(stage_noise(c, eps, flags, &nd));
  return local_score_device(c, nd, 0);
}
// The two halves of dust_svmpc_local_score as separate calls, for callers that overlap the all-gather of theta with the
// rollouts: the rollout / likelihood half reads only the LOCAL particles; the prior half reads every particle (keys).
extern "C" int dust_svmpc_local_rollout(dust_ctx *c, const float *state, const float *eps, const float *params, int flags) {
  if (!c || !state) return fail(DUST_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(c->cfg.device));
  TRY(upload_state_params(c, state, params, 1));
  const float *nd = nullptr;
  TRY(stage_noise(c, eps, flags, &nd));
  SampleOpts o;
  memset(&o, 0, sizeof o);
  o.noise_mode = nd ? NOISE_EPS : NOISE_PHILOX;
  o.noise_dev = nd;
  o.base = c->theta;
  o.update_a_mat = 1;
  o.bump_adam = 1;
  o.merge_prior = 0;
  TRY(launch_rollout(c, o));
  c->have_sample = true;
  return DUST_OK;
}
extern "C" int dust_svmpc_local_prior_score(dust_ctx *c) {
  if (!c) return fail(DUST_ERR_INVALID, "null ctx");
  if (!c->have_sample) return fail(DUST_ERR_STATE, "dust_svmpc_local_prior_score needs a preceding dust_svmpc_local_rollout");
  HIP_TRY(hipSetDevice(c->cfg.device));
  TRY(launch_prior(c));
  return launch_prior_finish(c, true, false);  // score = grad_lik + grad_pri for the local rows
}
extern "C" int dust_svmpc_apply_phi(dust_ctx *c) {
  if (!c) return fail(DUST_ERR_INVALID, "null ctx");
  HIP_TRY(hipSetDevice(c->cfg.device));
  TRY(launch_stein_update(c, 1));
  return DUST_OK;
}
extern "C" int dust_svmpc_forward_local(dust_ctx *c, void **log_w_all, size_t *shard_bytes) {
  if (!c) return fail(DUST_ERR_INVALID, "null ctx");
  HIP_TRY(hipSetDevice(c->cfg.device));
  TRY(forward_device(c));
  if (log_w_all) *log_w_all = c->lw;
  if (shard_bytes) *shard_bytes = (size_t)c->nloc * sizeof(float);
  return DUST_OK;
}
extern "C" int dust_svmpc_forward_finish(dust_ctx *c, float *a_seq, float *p_weights) {
  if (!c) return fail(DUST_ERR_INVALID, "null ctx");
  HIP_TRY(hipSetDevice(c->cfg.device));
  // a sharded context rolls every rank's rows (it holds them all: gathered after the last update), see sharded_forward
  const bool roll_all = c->nloc != c->N && c->cfg.roll_strategy != DUST_ROLL_RESAMPLE && c->theta == c->theta_home;
  TRY(forward_finish_device(c, -1, nullptr, roll_all));
  if (a_seq || p_weights) TRY(tick_outputs(c, a_seq, p_weights));
  return DUST_OK;
}

// ---------------------------------------------------------------------------------------------------------------
extern "C" int dust_profile_enable(dust_ctx *c, int on) {
  if (c) c->persist_declined = 0;  // (what the one-launch ticks declined may be eligible now - or the other way round)
  if (!c) return fail(DUST_ERR_INVALID, "null ctx");
  if (c->graph_exec || c->graph_exec_alt) graph_drop(c);
  c->prof = on != 0;
  return DUST_OK;
}
extern "C" int dust_profile_reset(dust_ctx *c) {
  if (!c) return fail(DUST_ERR_INVALID, "null ctx");
  memset(c->prof_ms, 0, sizeof c->prof_ms);
  memset(c->prof_n, 0, sizeof c->prof_n);
  return DUST_OK;
}
extern "C" int dust_profile_get(dust_ctx *c, int id, double *ms, int64_t *n) {
  if (!c || id < 0 || id >= DUST_K_COUNT) return fail(DUST_ERR_INVALID, "bad argument");
  if (ms) *ms = c->prof_ms[id];
  if (n) *n = c->prof_n[id];
  return DUST_OK;
}
// Back-to-back launches of the standalone rollout kernel in its HBM-streaming form (device-resident eps, a fresh slice
// per launch), bracketed by ONE pair of HIP events on the context's stream: the average is the kernel's launch-to-launch
// duration without per-launch event overhead (bench.py's roofline; compare rocprofv3 --kernel-trace --stats).
extern "C" int dust_profile_rollout(dust_ctx *c, const float *state, const float *eps_dev, int n_slices, int reps, int flags, double *avg_ms) {
  if (!c || !state || !eps_dev || !avg_ms || n_slices < 1 || reps < 1) return fail(DUST_ERR_INVALID, "bad argument");
  if (c->cfg.dim_p > 0 && !c->params_dev)
    return fail(DUST_ERR_STATE, "dust_profile_rollout: sampled parameters - run one dust_likelihood_sample with `params` first (they stay on the device)");
  HIP_TRY(hipSetDevice(c->cfg.device));
  TRY(upload_state_params(c, state, nullptr, 0));
  SampleOpts o;
  memset(&o, 0, sizeof o);
  o.noise_mode = NOISE_EPS;
  o.base = c->theta;
  o.update_a_mat = 1;
  o.merge_prior = 0;  // the rollout kernel as dust_likelihood_sample runs it (SURVEY 8d B_roll has no prior-partial traffic)
  o.want_states = (flags & DUST_STORE_STATES) != 0;  // the stored-states form ("for both store_states settings")
  o.store_f16 = (flags & DUST_STORE_F16) != 0;
  const size_t slice = ((size_t)c->S * c->N * c->D) >> ((flags & DUST_EPS_F16) ? 1 : 0);  // in floats
  c->noise_f16 = (flags & DUST_EPS_F16) != 0;
  const bool prof = c->prof;
  c->prof = false;
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  int st = DUST_OK;
  for (int r = 0; r < 3 && st == DUST_OK; ++r) {  // warm-up
    o.noise_dev = eps_dev + (size_t)(r % n_slices) * slice;
    st = launch_rollout(c, o);
  }
  if (st == DUST_OK) {
    (void)hipEventRecord(e0, c->stream);
    for (int r = 0; r < reps && st == DUST_OK; ++r) {
      o.noise_dev = eps_dev + (size_t)(r % n_slices) * slice;
      st = launch_rollout(c, o);
    }
    (void)hipEventRecord(e1, c->stream);
  }
  if (st == DUST_OK) {
    float ms = 0.f;
    if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) st = fail(DUST_ERR_HIP, "event timing failed");
    *avg_ms = (double)ms / reps;
  }
  if (st == DUST_OK && o.want_states) {
    // the stored-states form may be two launches (rollout_states.hpp + the injected-costs pass): time them apart as well, one event
    // pair per launch, into the per-kernel slots (dust_profile_get: DUST_K_ROLLOUT_STATES / DUST_K_ROLLOUT)
    c->prof = true;
    for (int k = 0; k < DUST_K_COUNT; ++k) {
      c->prof_ms[k] = 0.0;
      c->prof_n[k] = 0;
    }
    for (int r = 0; r < reps && st == DUST_OK; ++r) {
      o.noise_dev = eps_dev + (size_t)(r % n_slices) * slice;
      st = launch_rollout(c, o);
    }
  }
  c->prof = prof;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  c->have_sample = true;
  return st;
}

extern "C" int dust_rollout_algorithmic_bytes(const dust_ctx *c, int flags, double *bytes) {
  if (!c || !bytes) return fail(DUST_ERR_INVALID, "null argument");
  // B_roll = 4 [S N D (eps in) + N D (theta in) + M P (params) + S N (costs out)] (+ states when stored)
  double b = ((flags & DUST_EPS_F16) ? 2.0 : 4.0) * (double)c->S * c->nloc * c->D +
             4.0 * ((double)c->nloc * c->D + (double)c->M * c->P + (double)c->S * c->nloc);
  b += 4.0 * (double)c->nloc * c->D;  // grad_lik out
  if (flags & DUST_STORE_STATES) b += ((flags & DUST_STORE_F16) ? 2.0 : 4.0) * (double)c->M * c->S * c->nloc * (c->H + 1) * c->ds;
  *bytes = b;
  return DUST_OK;
}

__global__ void noise_fill_kernel(float *dst, size_t n, uint64_t seed) {
  const size_t i4 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i4 * 4 >= n) return;
  float z[4];
  philox_normal4(seed, (uint32_t)i4, (uint32_t)(i4 >> 32), 0x5eedu, 0xd057u, z);
  for (int q = 0; q < 4; ++q)
    if (i4 * 4 + q < n) dst[i4 * 4 + q] = z[q];
}
__global__ void narrow_f16_kernel(float *buf, size_t n) {  // in place: value i moves from byte 4i to byte 2i; chunks in order
  _Float16 *out = reinterpret_cast<_Float16 *>(buf);
  for (size_t base = 0; base < n; base += (size_t)gridDim.x * blockDim.x) {
    const size_t i = base + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const float v = i < n ? buf[i] : 0.f;
    wg_sync();
    if (i < n) out[i] = (_Float16)v;
  }
}

extern "C" int dust_device_noise_alloc(dust_ctx *c, size_t n, uint64_t seed, int flags, void **dptr) {
  if (!c || !dptr) return fail(DUST_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(c->cfg.device));
  float *p = nullptr;
  TRY(dalloc(&p, n));
  const size_t n4 = (n + 3) / 4;
  noise_fill_kernel<<<(unsigned)((n4 + 255) / 256), 256, 0, c->stream>>>(p, n, seed);
  HIP_TRY(hipGetLastError());
  if (flags & DUST_EPS_F16) {  // one workgroup walks the buffer front to back: a value's new place was read before it is written
    narrow_f16_kernel<<<1, 1024, 0, c->stream>>>(p, n);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  *dptr = p;
  return DUST_OK;
}
extern "C" int dust_device_free(dust_ctx *c, void *p) {
  if (!c) return fail(DUST_ERR_INVALID, "null ctx");
  if (p) HIP_TRY(hipFree(p));
  return DUST_OK;
}

// The tile order of the large-set passes (pairwise_packed.hpp query_order_kernel): position -> local row, n_local entries.  A test hook
// (not in include/dust_amd.h): the order must be a permutation of the rank's rows - a row that is listed twice leaves another one out of
// both passes.  Returns DUST_ERR_STATE when the context walks its queries in index order.
extern "C" int dust_debug_tile_order(dust_ctx *c, int *perm) {
  if (!c || !perm) return fail(DUST_ERR_INVALID, "null argument");
  if (!c->pk_order || !c->pk_perm) return fail(DUST_ERR_STATE, "no tile order (index order)");
  HIP_TRY(hipSetDevice(c->cfg.device));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (c->stream2) HIP_TRY(hipStreamSynchronize(c->stream2));
  HIP_TRY(hipMemcpy(perm, c->pk_perm, (size_t)c->nloc * sizeof(int), hipMemcpyDeviceToHost));
  return DUST_OK;
}

// The fixed cost of the sharded tick's three exchanges in their peer-store form, measured on ONE device (a test / measurement hook, not
// in include/dust_amd.h): this context plays rank 0 of `world`, its "peers" are scratch buffers of the same device, and a one-wave
// kernel plays the peers' arrival words and tokens.  Per exchange: the store kernel (the rank's piece copied world - 1 times, tokens,
// fences, arrival words) + the wait kernel - everything but the time the pieces spend on the links.
__global__ void peer_selftest_arrive_kernel(unsigned int *flags, int which, int world, unsigned int seq, int token) {
  const int r = threadIdx.x;
  if (r < 1 || r >= world) return;
  if (token) __hip_atomic_store(flags + PEER_BUFS * PEER_MAX + r, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  else __hip_atomic_store(flags + which * PEER_MAX + r, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
extern "C" int dust_debug_peer_selftest(dust_ctx *c, int world, int n_steps, int reps, double *us_per_tick) {
  if (!c || !us_per_tick || world < 2 || world > PEER_MAX || reps < 1 || n_steps < 1) return fail(DUST_ERR_INVALID, "bad argument");
  if (c->N % world) return fail(DUST_ERR_INVALID, "N %% world");
  if (c->peer) return fail(DUST_ERR_STATE, "the context has real peers");
  HIP_TRY(hipSetDevice(c->cfg.device));
  HIP_TRY(hipStreamSynchronize(c->stream));
  PeerState *p = new (std::nothrow) PeerState();
  if (!p) return fail(DUST_ERR_HIP, "out of host memory");
  memset((void *)p, 0, sizeof *p);
  p->world = world;
  p->rank = 0;
  const size_t nflag = (size_t)PEER_ROWS * PEER_MAX + PEER_MAX + 1, ND = (size_t)c->N * c->D;
  float *scratch = nullptr;
  unsigned int *sink = nullptr;
  int st = DUST_OK;
  if (hipMalloc((void **)&p->flags_local, nflag * 4) != hipSuccess || hipMalloc((void **)&sink, nflag * 4) != hipSuccess ||
      hipMalloc((void **)&scratch, ND * sizeof(float)) != hipSuccess)
    st = fail(DUST_ERR_HIP, "hipMalloc");
  if (st == DUST_OK) {
    (void)hipMemset(p->flags_local, 0, nflag * 4);
    (void)hipMemset(sink, 0, nflag * 4);
    for (int g = 0; g < world; ++g) {
      for (int k = 0; k < PEER_BUFS; ++k) p->buf[k][g] = g == 0 ? peer_own_buffer(c, k) : scratch;  // (every "peer" takes the piece at the same offset)
      p->flags[g] = g == 0 ? p->flags_local : sink;
    }
    c->peer = p;
    const int keep_rank = c->comm_rank;
    const size_t shard = (size_t)(c->N / world) * c->D;
    // the peers' side, once: every arrival word and token already stands at "arrived" (sequence numbers are compared as signed differences)
    for (int k = 0; k < PEER_ROWS; ++k) peer_selftest_arrive_kernel<<<1, 64, 0, c->stream>>>(p->flags_local, k, world, 0x3fffffffu, k == PEER_BUFS ? 1 : 0);
    auto exchange = [&](int which, size_t count) {
      int s2 = peer_store(c, which, count);
      if (s2 == DUST_OK) s2 = peer_wait(c, which);
      return s2;
    };
    hipEvent_t e0 = nullptr, e1 = nullptr;
    for (int w = 0; w < 3 && st == DUST_OK; ++w) st = exchange(GATHER_SCORE, shard);
    if (st == DUST_OK && (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)) st = fail(DUST_ERR_HIP, "hipEventCreate");
    if (st == DUST_OK) {
      (void)hipEventRecord(e0, c->stream);
      for (int r = 0; r < reps && st == DUST_OK; ++r) {
        for (int k = 0; k < n_steps && st == DUST_OK; ++k) {
          st = exchange(GATHER_SCORE, shard);
          if (st == DUST_OK) st = exchange(GATHER_THETA, shard);
        }
        if (st == DUST_OK) st = exchange(GATHER_LW, (size_t)(c->N / world));
      }
      (void)hipEventRecord(e1, c->stream);
      (void)hipEventSynchronize(e1);
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, e0, e1);
      *us_per_tick = 1e3 * (double)ms / reps;
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipStreamSynchronize(c->stream);
    if (st == DUST_OK) st = peer_check(c);
    (void)keep_rank;
    c->peer = nullptr;
  }
  if (scratch) (void)hipFree(scratch);
  if (sink) (void)hipFree(sink);
  if (p->flags_local) (void)hipFree(p->flags_local);
  delete p;
  return st;
}

// The run lists of the last large-set pass 1 (pairwise_packed.hpp), one query tile: *n_units, the unit offsets [n_units + 1], the key
// indices [uoff[n_units]], the unit query masks [n_units][4] and the slice boundaries [js + 1] of pass 1.  A test hook (not in
// include/dust_amd.h): tests/test_gpu_parity.py checks the lists against the masks they were built from.
extern "C" int dust_debug_pack_lists(dust_ctx *c, int tile, int *n_units, int *js, int *uoff, int *kidx, unsigned int *uq, int *soff) {
  if (!c || !n_units || !js) return fail(DUST_ERR_INVALID, "null argument");
  if (!c->pk_tiles || tile < -1 || tile >= c->pk_tiles) return fail(DUST_ERR_STATE, "no run lists (or no such tile)");
  HIP_TRY(hipSetDevice(c->cfg.device));
  HIP_TRY(hipStreamSynchronize(c->stream));
  const int JS = c->pk_jsg, um = c->pk_umax;
  if (tile == -1) {  // totals: *n_units = the units of all tiles, *js = tiles x chunks (what visiting every unit would walk)
    std::vector<int> all((size_t)c->pk_tiles * (JS + 1));
    HIP_TRY(hipMemcpy(all.data(), c->pk_soff, all.size() * sizeof(int), hipMemcpyDeviceToHost));
    long tot = 0;
    for (int t = 0; t < c->pk_tiles; ++t) tot += all[(size_t)t * (JS + 1) + JS];
    *n_units = (int)tot;
    *js = c->pk_tiles * um;
    return DUST_OK;
  }
  std::vector<int> so((size_t)JS + 1);
  HIP_TRY(hipMemcpy(so.data(), reinterpret_cast<int *>(c->pk_soff) + (size_t)tile * (JS + 1), so.size() * sizeof(int), hipMemcpyDeviceToHost));
  const int U = so[JS];
  *n_units = U;
  *js = JS;
  if (soff) memcpy(soff, so.data(), so.size() * sizeof(int));
  std::vector<int> uo((size_t)U + 1);
  HIP_TRY(hipMemcpy(uo.data(), reinterpret_cast<int *>(c->pk_uoff) + (size_t)tile * (um + 1), uo.size() * sizeof(int), hipMemcpyDeviceToHost));
  if (uoff) memcpy(uoff, uo.data(), uo.size() * sizeof(int));
  if (kidx && uo[U] > 0) HIP_TRY(hipMemcpy(kidx, reinterpret_cast<int *>(c->pk_idx) + (size_t)tile * um * 64, (size_t)uo[U] * sizeof(int), hipMemcpyDeviceToHost));
  if (uq && U > 0) HIP_TRY(hipMemcpy(uq, reinterpret_cast<unsigned int *>(c->pk_uq) + (size_t)tile * um * 4, (size_t)U * 4 * sizeof(unsigned int), hipMemcpyDeviceToHost));
  return DUST_OK;
}

extern "C" int dust_debug_far_units(dust_ctx *c, long long *out2) {
  if (!c || !out2) return fail(DUST_ERR_INVALID, "null argument");
  out2[0] = out2[1] = 0;
  const size_t n = (size_t)c->far_tiles * c->far_chunks;
  if (!n) return DUST_OK;
  HIP_TRY(hipSetDevice(c->cfg.device));
  HIP_TRY(hipDeviceSynchronize());
  std::vector<unsigned char> h(n);
  HIP_TRY(hipMemcpy(h.data(), c->far_f, n, hipMemcpyDeviceToHost));
  long long k = 0;
  for (size_t i = 0; i < n; ++i) k += h[i] != 0;
  out2[0] = k;
  out2[1] = (long long)n;
  return DUST_OK;
}
// the same for the last log-p pass (pairwise_logp_mfma.hpp): (64-query group, key chunk) blocks
extern "C" int dust_debug_far_logp(dust_ctx *c, long long *out2) {
  if (!c || !out2) return fail(DUST_ERR_INVALID, "null argument");
  out2[0] = out2[1] = 0;
  const size_t n = (size_t)c->far_groups * c->far_gchunks;
  if (!n) return DUST_OK;
  HIP_TRY(hipSetDevice(c->cfg.device));
  HIP_TRY(hipDeviceSynchronize());
  std::vector<unsigned char> h(n);
  HIP_TRY(hipMemcpy(h.data(), c->far_g, n, hipMemcpyDeviceToHost));
  long long k = 0;
  for (size_t i = 0; i < n; ++i) k += h[i] != 0;
  out2[0] = k;
  out2[1] = (long long)n;
  return DUST_OK;
}

#ifdef DUST_STAMPS
// diagnostic build only (not part of include/dust_amd.h): s_memtime phase stamps of block 0 of the last launch of a kernel
extern "C" int dust_debug_stamps(dust_ctx *c, int kernel_id, unsigned long long *out16) {
  if (!c->stamps_dev) {
    HIP_TRY(hipMalloc((void **)&c->stamps_dev, 16 * DUST_K_COUNT * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(c->stamps_dev, 0, 16 * DUST_K_COUNT * sizeof(unsigned long long)));
    HIP_TRY(hipMalloc((void **)&c->tl_dev, 2048 * 128 * sizeof(unsigned long long)));  // (the tick kernel stamps 128 words per workgroup)
    HIP_TRY(hipMemset(c->tl_dev, 0, 2048 * 128 * sizeof(unsigned long long)));
    return DUST_OK;
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (kernel_id < 0) {  // launch timeline of the one-launch iteration: out16 receives -kernel_id workgroups x 4 words
    HIP_TRY(hipMemcpy(out16, c->tl_dev, (size_t)(-kernel_id) * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return DUST_OK;
  }
  HIP_TRY(hipMemcpy(out16, c->stamps_dev + 16 * kernel_id, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return DUST_OK;
}
#endif

#include "mpf.hpp"
