/*
 * dust_amd.h - C ABI of libdust_amd.so: the MI355X-native (HIP, gfx950) SVGD-MPC inner loop.
 *
 * Drop-in boundary for the one hot path of lubaroli/dust.  The reference is pure Python and
 * has no FFI of its own; each entry point below replaces one reference METHOD (cited file:line into lubaroli/dust) and
 * is what a ctypes binding on the reference side would call (INTEGRATION.md shows that binding).
 *
 * Conventions
 *  - plain C, opaque handles, `int` status (0 = DUST_OK), no exceptions, no callbacks; dust_last_error() gives text;
 *  - all arrays are fp32, C-contiguous, in the reference's own layouts (e.g. actions [S][N][H][da]);
 *  - pointers are HOST pointers unless the call has a `flags` argument carrying DUST_PTR_DEVICE, in which case the
 *    bulk noise/action arrays are device pointers already resident in HBM (benchmarks, fused pipelines);
 *  - one HIP stream per context; a context is not thread-safe; every call returns after its results are on the host
 *    (calls with no host outputs are asynchronous on the context's stream; dust_sync() waits);
 *  - there is NO CPU fallback: without a usable HIP device dust_create fails with DUST_ERR_NO_DEVICE.
 *
 * Symbols: N = n_policies (Stein particles), S = n_samples (action samples per policy), M = n_params (dynamics
 * samples), H = horizon, da/ds = action/state dims, D = H*da, P = number of uncertain model parameters.
 */
#ifndef DUST_AMD_H
#define DUST_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DUST_ABI_VERSION 2

enum dust_status {
  DUST_OK = 0,
  DUST_ERR_INVALID = 1,     /* bad argument / shape (the reference raises ValueError / AssertionError) */
  DUST_ERR_UNSUPPORTED = 2, /* a configuration outside the HIP kernels' families (the reference would run Python) */
  DUST_ERR_NO_DEVICE = 3,
  DUST_ERR_HIP = 4,
  DUST_ERR_STATE = 5 /* call order (e.g. phi before any likelihood sample) */
};

enum dust_model { DUST_MODEL_PENDULUM = 0, DUST_MODEL_PARTICLE = 1, DUST_MODEL_SKID_STEER = 2 };
/* cost families: pendulum demo cost (demo/pendulum_example.py:21-28), Particle.default_*_cost (particle.py:170-225) */
enum dust_cost { DUST_COST_PENDULUM_QUADCOS = 0, DUST_COST_PARTICLE_DEFAULT = 1, DUST_COST_QUADRATIC = 2 };
/* K1: gpytorch RBFKernel semantics, lengthscale ln 2 (svmpc.py:76-83); K2: iid_mp(RBF) per-dimension median bandwidth
 * (svmpc.py:64-74, composite_kernels.py:33-64); K2_SHARED: indep_controls=False; IMQ: new, no reference. */
enum dust_kernel { DUST_KERNEL_K1_RBF = 0, DUST_KERNEL_K2_IIDMP = 1, DUST_KERNEL_K2_SHARED = 2, DUST_KERNEL_IMQ = 3 };
enum dust_likelihood { DUST_LIK_EXP_UTILITY = 0, DUST_LIK_EXPECTED_COST = 1 }; /* likelihoods.py:122-135 / 106-119 */
enum dust_optimizer { DUST_OPT_SGD = 0, DUST_OPT_ADAM = 1 };                   /* svgd.py:115, demos use SGD */
enum dust_roll { DUST_ROLL_REPEAT = 0, DUST_ROLL_MEAN = 1, DUST_ROLL_RESAMPLE = 2 }; /* svmpc.py:142-158 */
enum dust_step_strategy { DUST_STEP_ARGMAX = 0, DUST_STEP_AVERAGE = 1, DUST_STEP_EXTERNAL = 2 }; /* disco.py:396-417 */
/* how a model parameter enters the arithmetic: a Python float (double), a 0-dim fp32 tensor, or a sampled column */
enum dust_param_kind { DUST_PARAM_PYFLOAT = 0, DUST_PARAM_SAMPLED = 1, DUST_PARAM_TENSOR0D = 2 };
/* Particle(control_type=) particle.py:41-60, 149-153.  Velocity control is a TWO-state model (x, y; dim_s = 2): the action is clamped to
 * +-max_speed and integrated directly, and the closing clamp of the step (particle.py:165) lands on the positions. */
enum dust_control_type { DUST_CONTROL_ACCELERATION = 0, DUST_CONTROL_VELOCITY = 1 };
/* DUST_EPS_AROUND_A_MAT: the external actions were drawn around a_mat (MultiDISCO's own sampling, disco.py:155-160) */
enum dust_flags {
  DUST_PTR_DEVICE = 1,
  DUST_STORE_STATES = 2,
  DUST_EPS_AROUND_A_MAT = 4,
  /* fp16 STORAGE of the rollout's bulk data, fp32 arithmetic (BASELINE.json config 5 "fp16 rollout / fp32 SVGD"; the reference
   * itself is fp32 only).  DUST_EPS_F16: the `eps` / `actions` argument points to IEEE binary16 values (same layout, host or
   * device) - the rollout kernel's HBM read traffic halves.  DUST_STORE_F16: the stored `states` / `actions_out` are binary16
   * (the caller's output buffers receive binary16 values, half the bytes; 5.5 GB instead of 11 GB of states at config 3). */
  DUST_EPS_F16 = 8,
  DUST_STORE_F16 = 16
};

typedef struct dust_param {
  int32_t kind;   /* dust_param_kind */
  int32_t column; /* column of `params` when kind == DUST_PARAM_SAMPLED */
  double value;   /* default value otherwise */
} dust_param;

typedef struct dust_config {
  int32_t abi_version; /* DUST_ABI_VERSION */
  int32_t device;      /* HIP device ordinal */
  /* sizes: MultiDISCO(n_policies, action_samples, params_samples, hz_len) disco.py:16-137 */
  int32_t n_policies;       /* N, total over all shards */
  int32_t n_samples;        /* S */
  int32_t n_params;         /* M (1 when params_sampling is off) */
  int32_t horizon;          /* H */
  int32_t dim_a, dim_s, dim_p;
  /* data-parallel shard of the policy index owned by this context: [shard_offset, shard_offset + shard_size) */
  int32_t shard_offset, shard_size; /* shard_size 0 = all */
  int32_t model, cost, kernel, likelihood, optimizer, roll_strategy;
  int32_t weighted_prior;    /* SVMPC(weighted_prior=) svmpc.py:21 */
  int32_t params_log_space;  /* MultiDISCO(params_log_space=) disco.py:173 */
  int32_t params_interleave; /* scalar-event params_dist quirk, disco.py:177-179: rollout r uses params[r % M] */
  float alpha;               /* likelihood alpha */
  float temperature;         /* MultiDISCO temperature */
  float a_reg;               /* temperature * (1 - ctrl_penalty) disco.py:90 */
  float lr, adam_beta1, adam_beta2, adam_eps;
  float chol_a[4];  /* diagonal of cholesky(a_cov): policy-noise scale (likelihoods.py:85-90) */
  float sigma_a[4]; /* sqrt(diag(a_dist.covariance_matrix)) as svmpc.py:107-111 computes it */
  float a_pre[4];   /* diagonal of inverse(a_cov) disco.py:98 */
  float sigma_p[4]; /* sqrt of the prior component covariance diagonal (svgd.py:84-89) */
  float bw_scale;   /* RBF(bw_scale=) base_kernels.py:44 */
  float imq_ell;
  float min_a[4], max_a[4]; /* action_space bounds, disco.py:408-410 */
  uint64_t seed;            /* device Philox stream for internally drawn noise */
  /* model: dust/models/pendulum.py, dust/models/particle.py */
  double dt;
  dust_param g, mass, length; /* pendulum; `mass` is also the particle mass */
  double max_torque, max_speed_pend;
  double w_cos, w_vel;        /* pendulum cost weights */
  float max_speed, max_accel; /* particle */
  int32_t can_crash, with_obstacle;
  double cell_size;
  float target[4], w_state[4], w_term[4], w_ctrl[2], w_obs;
  /* ABI 2: Particle(control_type=, deterministic=, noise_std=) particle.py:13-31.  ctrl_noise = `not deterministic` (the reference's
   * constructor default is deterministic=False with noise_std = zeros(2)): every model step adds dyn_std * N(0, I) to the actions that
   * drive the dynamics (particle.py:145-148; the costs see the raw actions).  Draws: a device Philox stream, or the recorded tensors
   * handed to dust_set_ctrl_noise / dust_mpf_set_ctrl_noise (parity runs). */
  int32_t control_type; /* dust_control_type */
  int32_t ctrl_noise;
  float dyn_std[2];
  /* ABI 2: FULL 2 x 2 covariances (dim_a = 2).  MultiDISCO(a_cov=<any SPD matrix>) disco.py:91-98: policy noise actions = theta + L_a eps
   * with L_a = cholesky(a_cov) (likelihoods.py:85-90) - chol_a[0..1] stay its diagonal, chol_a_off = L_a[1][0]; a_pre = inverse(a_cov)
   * in the control cost (disco.py:338-346) - a_pre[0..1] stay its diagonal, a_pre_off = a_pre[0][1]; sigma_a stays sqrt(diag a_cov)
   * (svmpc.py:107-111).  Prior components with a full covariance Sigma_p = L_p L_p^T (get_gmm svgd.py:84-89): chol_p = {L_p[0][0],
   * L_p[1][0], L_p[1][1]}, sigma_p[0..1] stay sqrt(diag Sigma_p).  full_cov != 0 switches these on (all zero: the diagonal forms). */
  int32_t full_cov;
  float chol_a_off, a_pre_off;
  float chol_p[3];
} dust_config;

/* SkidSteerRobot (dust/models/skid_steer_robot.py:19-52, step :73-122; model = DUST_MODEL_SKID_STEER, dim_s = 5: x, y, theta, v,
 * omega; dim_a = 2: right / left wheel speed) with the quadratic cost family DUST_COST_QUADRATIC - the reference ships no cost for
 * this model, its MultiDISCO takes any callable (disco.py:294-346); the family is what dust_amd.costs.QuadraticCost evaluates:
 *   inst(x, a) = sum_k w_state[k] (x_k - goal_k)^2 + sum_d w_ctrl[d] a_d^2 ,   term(x) = sum_k w_term[k] (x_k - goal_k)^2 .
 * dust_create gives the reference's constructor defaults; dust_set_skid_steer replaces them.  Sampled parameters
 * (kind DUST_PARAM_SAMPLED) name columns of the `params` rows in the order of `uncertain_params`. */
typedef struct dust_skid_config {
  dust_param x_icr, wheel_radius, axial_distance;
  float min_wheel_speed[2], max_wheel_speed[2]; /* action_space bounds: the step clamps the wheel speeds to them */
  float goal[5], w_state[5], w_term[5], w_ctrl[2];
} dust_skid_config;

typedef struct dust_ctx dust_ctx;
typedef struct dust_mpf dust_mpf;

const char *dust_last_error(void);
int dust_abi_version(void);
int dust_device_count(int *count);

/* lifecycle.  dust_clone is what copy.deepcopy(controller/svmpc) maps to (simulations.py:62, particle_example.py:166-175) */
int dust_create(const dust_config *cfg, dust_ctx **out);
int dust_clone(const dust_ctx *src, dust_ctx **out);
void dust_destroy(dust_ctx *ctx);
int dust_sync(dust_ctx *ctx);
/* Which device path served the SVMPC.optimize / forward calls so far (svmpc.py:97-200; sticky counts since creation):
 * out[0] one-launch ticks, owner-computes form (tick2.hpp) - launched-ahead ticks of closed-loop serving included, also cancelled ones;
 * out[1] always 0 (the tiled one-launch tick of rounds 2-5 is retired); out[2] ticks answered through the pinned done word of closed-loop serving;
 * out[3] ticks whose one-launch kernel found the device shared with other work (its workgroups were not all resident) and that
 * were therefore run on the launch-per-iteration path instead - late, on unchanged state, never lost. */
int dust_tick_stats(dust_ctx *ctx, long long out[4]);
int dust_get_config(const dust_ctx *ctx, dust_config *out);
/* model.params_dict[...] = v after construction (particle_example.py:178-179) */
int dust_set_model_param(dust_ctx *ctx, const char *name, double value, int kind);
/* Unscented-transform rollouts (MultiDISCO(params_sampling=MerweScaledUTF), disco.py:211-292, 312-323): the n_params
 * dynamics samples passed to forward / optimize are the 2 n + 1 sigma points and the cost of a rollout is their weighted
 * combination with `w[n_params]` (utf.py loc_weights; the reference's (sigma, step) weight pattern is reproduced) instead
 * of the mean.  NULL switches back to the mean over sampled parameters. */
int dust_set_param_weights(dust_ctx *ctx, const float *w);
int dust_set_skid_steer(dust_ctx *ctx, const dust_skid_config *cfg);
/* Recorded control-channel noise for the NEXT rollouts of a Particle(deterministic=False) context (particle.py:145-148): z holds n_sets
 * tensors [H][M*S*N][da] of standard-normal draws in the reference's own order - one `torch.randn_like(acts)` per model.step call of
 * MultiDISCO._rollout (disco.py:193-200), rollout r = (m*S + s)*N + n, N = n_policies (all shards).  Every rollout launch that follows
 * (one per likelihood sample, i.e. per SVGD iteration) consumes one set; when they are used up - or after z = NULL - the draws come
 * from the context's Philox stream again.  Host pointer; copied before the call returns. */
int dust_set_ctrl_noise(dust_ctx *ctx, const float *z, int n_sets);
/* ObstacleMap occupancy grid [nx][ny] (obstacle_map.py:13-43); offsets are the map centre in cells */
int dust_set_grid(dust_ctx *ctx, const float *grid, int nx, int ny, float off_x, float off_y);

/* particle / prior / controller state (all [N][H][da] or [N]) */
int dust_set_theta(dust_ctx *ctx, const float *theta);                                /* SVMPC.theta svmpc.py:25 */
/* RBF(bandwidth=, minimum_bw=) of the K2 kernels (base_kernels.py:44-92): bandwidth < 0 = median trick (default), every
 * per-dimension h = max(bw_scale * median / log(N + 1), minimum_bw); otherwise the fixed h = clip(bw_scale * bandwidth^2 / log(N + 1),
 * minimum_bw) replaces the medians.  minimum_bw > 0 (the reference's default is 1e-5). */
int dust_set_k2_bandwidth(dust_ctx *ctx, float bandwidth, float minimum_bw);
int dust_get_theta(dust_ctx *ctx, float *theta);
int dust_set_prior(dust_ctx *ctx, const float *means, const float *mix_weights);      /* get_gmm svgd.py:84-89 */
int dust_get_prior(dust_ctx *ctx, float *means, float *mix_probs);
int dust_set_a_mat(dust_ctx *ctx, const float *a_mat);                                /* MultiDISCO.a_mat disco.py:101-109 */
int dust_get_a_mat(dust_ctx *ctx, float *a_mat);
int dust_get_a_mix(dust_ctx *ctx, float *a_mix);                                      /* disco.py:393 */
int dust_set_a_seq(dust_ctx *ctx, const float *a_seq);                                /* BaseController.a_seq base.py:34-37 */
int dust_get_a_seq(dust_ctx *ctx, float *a_seq);

/* MultiDISCO.forward(state, model, params_dist, ext_actions) disco.py:348-394.
 * actions [S][N][H][da] (NULL: drawn on device as a_mat + L z, disco.py:155-160); params [M][P] raw samples or NULL.
 * outputs may be NULL: costs [S][N], states [M][S][N][H+1][ds], actions_out [S][N][H][da], omega [S][N].
 * Side effects as in the reference: a_mat += sum_s omega eps, a_mix refreshed. */
int dust_disco_forward(dust_ctx *ctx, const float *state, const float *actions, const float *params, int flags,
                       float *costs, float *states, float *actions_out, float *omega);
/* MultiDISCO.step(strategy, steps, ext_actions) disco.py:396-417 -> next_actions [steps][da] */
int dust_disco_step(dust_ctx *ctx, int strategy, int steps, const float *ext_actions, float *next_actions);

/* CostLikelihood.sample(theta, state, params_dist) likelihoods.py:81-101: actions = theta + L eps, then forward.
 * eps [S][N][H][da] standard-normal draws (NULL: device Philox).  costs [S][N], actions_out optional. */
int dust_likelihood_sample(dust_ctx *ctx, const float *state, const float *eps, const float *params, int flags,
                           float *costs, float *actions_out);
/* likelihood.log_prob(costs) likelihoods.py:113-135 on the last sampled costs -> [N] */
int dust_likelihood_log_prob(dust_ctx *ctx, float *log_l);

/* SVMPC.phi(log_p, bw, sigma) svmpc.py:32-85 with the costs/actions a user-supplied log_p returned
 * (costs [S][N], actions [S][N][H][da]); NULL/NULL = use the last likelihood sample held on the device.
 * outputs optional: phi, grad_lik, grad_pri each [N][H][da]. */
int dust_svmpc_phi(dust_ctx *ctx, const float *costs, const float *actions, float *phi, float *grad_lik, float *grad_pri);
/* SVMPC.step(state, params_dist, bw, sigma) svmpc.py:87-95: sample, phi, optimiser step on theta */
int dust_svmpc_step(dust_ctx *ctx, const float *state, const float *eps, const float *params, int flags);
/* SVMPC.optimize(state, params_dist, n_steps) svmpc.py:97-126. eps [n_steps][S][N][H][da] or NULL, params [n_steps][M][P] or NULL */
int dust_svmpc_optimize(dust_ctx *ctx, const float *state, int n_steps, const float *eps, const float *params, int flags);
/* SVMPC.forward(state, params_dist, fast_pred=True) svmpc.py:172-200: weights, argmax, roll, prior refresh.
 * a_seq [H][da], p_weights [N] (either may be NULL) */
int dust_svmpc_forward(dust_ctx *ctx, float *a_seq, float *p_weights);
/* The pieces of SVMPC.forward as the reference exposes them (svmpc.py:128-170), for callers that use them one by one:
 * get_weights (fast_pred=True: the costs of the last sample; weights only - theta and the prior stay as they are),
 * roll(steps, strategy) - theta.roll(steps, dims=-2) is CIRCULAR, the strategy then rewrites the last row; strategy "resample" takes
 * the last action of a prior sample per particle, drawn by the caller (`last_row` [N][da]; the reference draws prior.sample([N])) -
 * and update_prior(weights) (weights NULL: ones).  dust_svmpc_forward_ex = forward(steps, ...) with the context's roll strategy. */
int dust_svmpc_get_weights(dust_ctx *ctx, float *p_weights);
int dust_svmpc_roll(dust_ctx *ctx, int steps, int strategy, const float *last_row);
int dust_svmpc_update_prior(dust_ctx *ctx, const float *weights);
int dust_svmpc_forward_ex(dust_ctx *ctx, int steps, const float *resample_last_row, float *a_seq, float *p_weights);
/* CostLikelihood.sample(theta, state, params_dist) for a theta that is NOT the optimiser's (likelihoods.py:81-101 takes theta as an
 * argument and touches neither SVMPC.theta nor the optimiser state): rollouts around `theta` [N][H][da], the context's particles,
 * Adam moments and a_mat side effects as in the reference (a_mat is MultiDISCO state and does change). */
int dust_likelihood_sample_at(dust_ctx *ctx, const float *state, const float *theta, const float *eps, const float *params, int flags,
                              float *costs, float *actions_out);
/* one whole control tick = optimize(n_steps) + forward(), enqueued without host round trips (one persistent launch where eligible) */
int dust_svmpc_tick(dust_ctx *ctx, const float *state, int n_steps, const float *eps, const float *params, int flags,
                    float *a_seq, float *p_weights);

/* Closed-loop serving: the loop of dust/utils/simulations.py:104-123 - optimize, forward, first action to the plant, new state, repeat -
 * with the host's share off the device's critical path.  Between dust_svmpc_serve_start and dust_svmpc_serve_stop a call
 *   dust_svmpc_tick(ctx, state, n_steps, NULL, NULL, 0, a_seq, p_weights)        (device noise, the n_steps given here)
 * (1) receives its outputs through pinned host memory - the kernel writes them there and publishes a sequence number the call spins on:
 * no device-to-host copy, no stream synchronisation - and (2) launches the NEXT tick ahead of its plant state: that launch exchanges
 * its particles, draws its noise and runs its prior pass while the caller steps the plant, and its rollouts start when the next call
 * posts the state to a pinned mailbox.  Results are those of the same calls without serving, bit for bit.
 * The launched-ahead tick waits at most wait_us microseconds for its state (it occupies the whole device while it waits): a state that
 * comes later finds a launch that has given up - nothing written - and the tick is run then, late but correctly; after three such
 * misses in a row the context stops launching ahead.  wait_us = 0: outputs through pinned memory only.  Every other entry point of
 * the context, dust_create on the same device and the dynamics filter's calls cancel a waiting launch first (it leaves no trace).
 * Needs the one-launch tick's shape (K1 / IMQ kernel, N % 4 == 0, N / 4 <= CUs, H * da <= 32, one GPU, no sampled dynamics). */
int dust_svmpc_serve_start(dust_ctx *ctx, int n_steps, double wait_us);
int dust_svmpc_serve_stop(dust_ctx *ctx);

/* stage outputs of the last call, for parity tests ([S][N] / [N][H][da] / [N]) */
int dust_get_costs(dust_ctx *ctx, float *costs);
int dust_get_actions(dust_ctx *ctx, float *actions);
/* Trajectories of the states the last stored-states sample left on the device (MultiDISCO.forward's `states` [M][S][N][H+1][ds],
 * disco.py:190-200, 394 - 11 GB at BASELINE configs[2]; kept in HBM by dust_likelihood_sample(flags & DUST_STORE_STATES) and by
 * dust_disco_forward): rows[i] = (m*S + s)*N + n selects one rollout, out receives n_rows x [H+1][ds] values - fp32, or binary16 when
 * the sample ran with DUST_STORE_F16.  For callers (and parity tests) that need some rollouts, not an 11 GB copy. */
int dust_get_states_rows(dust_ctx *ctx, const long long *rows, int n_rows, void *out);
int dust_get_score(dust_ctx *ctx, float *score);
int dust_get_phi(dust_ctx *ctx, float *phi);
/* the two halves of the score of the last SVGD iteration: grad_lik (svmpc.py:50-53) and grad_pri (svmpc.py:38-41), [N][H][da] each;
   either pointer may be NULL */
int dust_get_score_parts(dust_ctx *ctx, float *grad_lik, float *grad_pri);
int dust_get_log_weights(dust_ctx *ctx, float *log_l, float *log_p);
int dust_get_bandwidths(dust_ctx *ctx, float *h); /* K2: [H*da] or [H] */

/* multi-GPU: the pairwise stages read every shard's theta/score from context-owned gather buffers ([N][D] each).
 * The caller all-gathers them in place with RCCL between dust_svmpc_local_score and dust_svmpc_apply_phi. */
int dust_gather_buffers(dust_ctx *ctx, void **theta_all, void **score_all, size_t *shard_bytes);
int dust_svmpc_local_score(dust_ctx *ctx, const float *state, const float *eps, const float *params, int flags);
/* dust_svmpc_local_score in two calls, so that the all-gather of theta (issued after dust_svmpc_apply_phi) can run while
 * the rollouts - which read only this rank's particles - execute: local_rollout (costs, weights, grad_lik, a_mat), then,
 * once every rank's theta has arrived, local_prior_score (prior pass over all particles; score = grad_lik + grad_pri). */
int dust_svmpc_local_rollout(dust_ctx *ctx, const float *state, const float *eps, const float *params, int flags);
int dust_svmpc_local_prior_score(dust_ctx *ctx);
int dust_svmpc_apply_phi(dust_ctx *ctx);
int dust_svmpc_forward_local(dust_ctx *ctx, void **log_w_all, size_t *shard_bytes);
int dust_svmpc_forward_finish(dust_ctx *ctx, float *a_seq, float *p_weights);
/* C-side multi-GPU tick (BASELINE.json north_star: "particle batches shard across the 8 GPUs of one node with an
 * RCCL all-gather over xGMI of particle states before the pairwise kernel step").  One process per GPU, each with a context created
 * for its shard (shard_offset = rank * N / world, shard_size = N / world).  Rank 0 calls dust_comm_unique_id and hands the bytes to
 * every rank out of band (MPI, a file, torch.distributed.broadcast_object_list, ...); every rank then calls dust_comm_init
 * (ncclCommInitRank - collective).  From then on dust_svmpc_tick / _optimize / _forward run the sharded tick themselves: the
 * in-place all-gathers of score and theta (per SVGD iteration) and of the log-weights and rolled particles (per tick) are issued
 * on the context's stream between the kernels - no host synchronisation, no Python in the loop.  RCCL is bound at run time
 * (dlopen of librccl.so): single-GPU hosts need none. */
#define DUST_COMM_ID_BYTES 128
int dust_comm_unique_id(void *id /* DUST_COMM_ID_BYTES */);
/* dust_comm_validate: the LOCAL checks of dust_comm_init (rank / world sane, the context's shard is rank's equal share, no communicator
 * yet, librccl loadable) and nothing else - no collective.  ABORT RULE: ncclCommInitRank has no time-out, so a rank that fails its checks
 * while the others enter it leaves them hanging; a launcher therefore calls dust_comm_validate on every rank, lets the ranks AGREE on the
 * result out of band (an all-gather of the status over the same channel that carries the id), and calls dust_comm_init only when every
 * rank passed (dust_amd/parallel.py ShardedSVMPC does exactly that). */
int dust_comm_validate(dust_ctx *ctx, int rank, int world);
int dust_comm_init(dust_ctx *ctx, const void *id, int rank, int world);
int dust_comm_destroy(dust_ctx *ctx);
/* The all-gathers of one sharded tick alone (per SVGD iteration: score rows and particles; per tick: the log-weights), `reps` times
 * back to back on the context's stream between one pair of HIP events -> microseconds per tick when nothing overlaps them.
 * Collective over all ranks; the context's particles are not touched.  (Measurement aid; no reference counterpart.) */
int dust_comm_probe(dust_ctx *ctx, int n_steps, int reps, double *us_per_tick);
/* The tick's all-gathers as DIRECT PEER STORES (dust_amd/csrc/peer_gather.hpp): the GPUs of a node are one xGMI hop apart, the pieces are
 * small (a rank's score / particle rows, its log-weights), so every rank writes its piece straight into every peer's buffer - mapped once,
 * here, through HIP IPC - and raises one arrival word per peer; a one-wave kernel in front of the first consumer waits for the words.
 * on != 0: map the peers' buffers (COLLECTIVE: the IPC handles travel through one all-gather of the communicator - every rank calls it,
 * or none); on == 0: back to the collective library's all-gathers.  DUST_PEER_GATHER=1 in the environment makes dust_comm_init call it.
 * Its outcome is collective as well: if ANY rank cannot map its peers (no IPC between the devices), every rank returns DUST_ERR_UNSUPPORTED and
 * keeps the collective library's all-gathers.  A piece that does not arrive within 1 s is reported by the next dust_sync / tick output as DUST_ERR_HIP.  (No reference counterpart:
 * the reference is single-device.) */
int dust_comm_peer_gather(dust_ctx *ctx, int on);
/* run the context's kernels on an external HIP stream (hipStream_t), e.g. torch's current stream */
int dust_set_stream(dust_ctx *ctx, void *hip_stream);

/* timing / roofline support: HIP-event timing of each kernel family on the context's stream */
enum dust_kernel_id {
  DUST_K_ROLLOUT = 0, DUST_K_PRIOR_SCORE = 1, DUST_K_STEIN = 2, DUST_K_UPDATE = 3, DUST_K_FORWARD = 4, DUST_K_BANDWIDTH = 5,
  DUST_K_MPF = 6, DUST_K_ROLLOUT_STATES = 7 /* whole-line stored-states rollouts */, DUST_K_COUNT = 8
};
int dust_profile_enable(dust_ctx *ctx, int on);
int dust_profile_get(dust_ctx *ctx, int kernel_id, double *total_ms, int64_t *launches);
int dust_profile_reset(dust_ctx *ctx);
/* reps back-to-back launches of the standalone rollout kernel over device-resident eps [n_slices][S][N][D] (slice r %
 * n_slices per launch) between one pair of HIP events on the context's stream; *avg_ms = elapsed / reps.
 * flags: DUST_EPS_F16 when eps_dev holds binary16 values; DUST_STORE_STATES (| DUST_STORE_F16): the stored-states form - it may
 * be two launches (rollout_states.hpp + the injected-costs pass), so a second series of `reps` launches is then timed one event
 * pair per launch into the per-kernel slots (dust_profile_get: DUST_K_ROLLOUT_STATES, DUST_K_ROLLOUT; earlier totals are reset). */
int dust_profile_rollout(dust_ctx *ctx, const float *state, const float *eps_dev, int n_slices, int reps, int flags, double *avg_ms);
const char *dust_kernel_name(int kernel_id);
/* algorithmic bytes one launch of the rollout kernel moves (SURVEY.md section 8d B_roll) */
int dust_rollout_algorithmic_bytes(const dust_ctx *ctx, int flags, double *bytes);
/* device scratch for benchmarks: allocate/fill standard-normal noise in HBM with the context's Philox stream
 * (n values; flags: DUST_EPS_F16 -> binary16 values, n * 2 bytes) */
int dust_device_noise_alloc(dust_ctx *ctx, size_t n_floats, uint64_t seed, int flags, void **dptr);
int dust_device_free(dust_ctx *ctx, void *dptr);

/* ---- MPF: dynamics-parameter SVGD filter (mpf.py:13-86, likelihoods.py:12-64, svgd.py:92-99) ---- */
typedef struct dust_mpf_config {
  int32_t abi_version, device;
  int32_t n_particles; /* M_p */
  int32_t dim_p, dim_s, dim_a;
  int32_t model;
  int32_t log_space;   /* GaussianLikelihood(log_space=) */
  float obs_std, lr, bw_scale;
  float init_bw;       /* MPF(bw=): bandwidth of the initial prior; <= 0: bw_silverman of the particles (svgd.py:55-81) */
  dust_config model_cfg; /* only the model fields are read */
} dust_mpf_config;
int dust_mpf_create(const dust_mpf_config *cfg, const float *init_particles, const float *initial_obs, dust_mpf **out);
int dust_mpf_clone(const dust_mpf *src, dust_mpf **out);
/* MPF(optimizer_class=, **opt_args) svgd.py:108-122, mpf.py:24: DUST_OPT_SGD (default here, the demos' choice) or DUST_OPT_ADAM (the
 * reference's class default; betas / eps as torch.optim.Adam).  The optimiser state starts at zero and persists across
 * dust_mpf_optimize calls, as the reference's does (the optimiser is built once in MPF.__init__). */
int dust_mpf_set_optimizer(dust_mpf *mpf, int optimizer, float beta1, float beta2, float eps);
void dust_mpf_destroy(dust_mpf *mpf);
/* MPF.optimize(action, new_obs, bw, n_steps) mpf.py:64-86 -> grad_norms [n_steps] */
int dust_mpf_optimize(dust_mpf *mpf, const float *action, const float *new_obs, float bw, int n_steps, float *grad_norms);
int dust_mpf_phi(dust_mpf *mpf, float bw, float *phi); /* MPF.phi mpf.py:40-57 */
/* GaussianLikelihood.condition(action, new_obs) likelihoods.py:51-64 */
int dust_mpf_condition(dust_mpf *mpf, const float *action, const float *new_obs);
/* Control-channel noise inside the filter's one-step prediction (likelihoods.py:30-46 -> Particle.step particle.py:145-148 with
 * model_cfg.ctrl_noise): `acts` there is the bare past action, so ONE da-vector is drawn per MPF.phi call - per SVGD step - and shared
 * by all filter particles.  z [n][da]: recorded draws consumed one per step by the following dust_mpf_phi / dust_mpf_optimize calls;
 * when used up (or z = NULL) a host generator seeded from model_cfg.seed draws them. */
int dust_mpf_set_ctrl_noise(dust_mpf *mpf, const float *z, int n);
/* occupancy grid for the Particle model's crash mask inside the one-step prediction */
int dust_mpf_set_grid(dust_mpf *mpf, const float *grid, int nx, int ny, float off_x, float off_y);
int dust_mpf_get_particles(dust_mpf *mpf, float *x);
int dust_mpf_set_particles(dust_mpf *mpf, const float *x);
int dust_mpf_get_prior(dust_mpf *mpf, float *means, float *bw);
/* MPF(bw=None) with P > 1 parameters (mpf.py:29-38): bw_silverman (svgd.py:55-81) of the particle columns is a [P] vector and
 * `bw ** 2 * torch.eye(P)` makes it the covariance diag(bw_p^2) of the FIRST prior; n = 1 or P values.  Every later
 * update_prior(bw) (mpf.py:85, at the end of optimize) is scalar again.  dust_mpf_get_prior reports bw[0]; _get_prior_bw all P. */
int dust_mpf_set_prior_bw(dust_mpf *mpf, const float *bw, int n);
int dust_mpf_get_prior_bw(dust_mpf *mpf, float *bw);
/* How dust_mpf_optimize calls ran: out[0] calls served by a multi-workgroup kernel (>= 96 particles, >= 2 steps; mpf.hpp), out[1]
 * those among them that did not start or did not commit (the grid was not co-resident / a wait gave up) and were run by the
 * single-workgroup kernel instead - the caller saw DUST_OK either way. */
int dust_mpf_stats(dust_mpf *mpf, long long out[2]);
/* mpf.prior.sample([n]) / .log_prob(x): the controller draws its dynamics samples here (disco.py:171-172) */
int dust_mpf_prior_sample(dust_mpf *mpf, int n, uint64_t seed, float *samples);
int dust_mpf_prior_log_prob(dust_mpf *mpf, int n, const float *x, float *log_prob);
/* MPF.optimize's default bandwidth, `silvermans_rule(self.x.view(-1, 1)) * self.bw_scale` (mpf.py:68-73; KDEpy 1.1.0, third party: parity
 * unpinned), evaluated on the device (float64, numpy's linear-interpolation quartiles): one launch + a 4-byte read-back. */
int dust_mpf_silverman(dust_mpf *mpf, float *bw);
/* One control period of the dual loop (simulations.py:104-138; BASELINE north_star "DualSVMPC step() / forward()") in ONE call:
 * mpf.optimize(action_prev, state, bw, mpf_steps) (skipped when action_prev is NULL: the first period; mpf_bw <= 0: Silverman's rule of
 * the filter's particles, on the device) -> the controller's n_steps x [M][P] dynamics samples drawn from the filter's refreshed prior on
 * the device, into the controller's parameter buffer (dyn_dist = mpf.prior: simulations.py:79, disco.py:171) -> svmpc.optimize(n_steps) +
 * svmpc.forward().  a_seq [H][da], p_weights [N] or NULL; *bw_used (or NULL): the bandwidth of the filter update.  seed: Philox key of the
 * period's draws (the stream of dust_mpf_prior_sample). */
int dust_dual_tick(dust_ctx *ctx, dust_mpf *mpf, const float *state, const float *action_prev, int n_steps, int mpf_steps, float mpf_bw,
                   uint64_t seed, float *a_seq, float *p_weights, float *bw_used);

#ifdef __cplusplus
}
#endif
#endif
