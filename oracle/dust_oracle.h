/*
 * dust_oracle.h - CPU ORACLE for the SVGD-MPC hot path of lubaroli/dust.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library; the product path (libdust_amd.so, HIP) never links or calls it.
 *
 * Each function restates, in plain scalar C, one step of the reference's per-control-tick algorithm and cites the
 * reference file:line it follows (paths are into the upstream tree, lubaroli/dust @ v0).  Arithmetic is fp32 where
 * the reference's torch-CPU arithmetic is fp32; in the rollout (chaotic, discontinuous at obstacle cells) the
 * reference's operation ORDER is followed operation by operation; reductions accumulate in double.
 *
 * Pinning: checked against the golden vectors in tests/golden/*.npz, which were produced by importing the
 * reference itself (tests/golden/make_golden.py).  Two third-party boundaries are PARITY UNPINNED (the packages are
 * not vendored in the reference and not installable here): gpytorch 1.5.0 RBFKernel (kernel mode K1) and KDEpy
 * 1.1.0 silvermans_rule; see oracle/ref_shim.py.
 */
#ifndef DUST_ORACLE_H
#define DUST_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_MODEL_PENDULUM = 0, ORC_MODEL_PARTICLE = 1 };
enum { ORC_LIK_EXP_UTILITY = 0, ORC_LIK_EXPECTED_COST = 1 };
enum { ORC_KERNEL_K1 = 0, ORC_KERNEL_K2 = 1, ORC_KERNEL_K2_SHARED = 2, ORC_KERNEL_IMQ = 3 };
enum { ORC_ROLL_REPEAT = 0, ORC_ROLL_MEAN = 1 };

/* A model parameter that is either a Python float (kind 0, kept in double like the interpreter does) or a per-rollout
 * fp32 tensor column taken from the sampled `params` (kind 1, column index `col`). */
typedef struct {
  int is_tensor;
  int col;
  double value;
} orc_param;

typedef struct {
  int model;   /* ORC_MODEL_* */
  int N, S, M, H, da, ds, P;
  int params_interleave; /* scalar-event params_dist: rollout r uses params[r % M] (disco.py:177-179) */
  int params_log_space;  /* params = exp(sample) (disco.py:173-174) */
  double dt;
  /* pendulum (dust/models/pendulum.py:16,61-100) */
  orc_param g, mass, length;
  double max_torque, max_speed_pend;
  double w_cos, w_vel; /* demo cost 50 (cos th - 1)^2 + 1.0 thd^2 (demo/pendulum_example.py:21-28) */
  /* particle (dust/models/particle.py:117-225) */
  orc_param pmass;
  int mass_is_0dim_tensor; /* model mass given as a 0-dim fp32 tensor (particle_example.py:58-61) */
  float max_speed, max_acc;
  int can_crash, with_obstacle;
  double cell_size;
  int nx, ny;
  float off_x, off_y;
  const float *grid; /* [nx][ny] occupancy, row-major (obstacle_map.py:41) */
  float target[4], w_state[4], w_term[4], w_ctrl[2], w_obs;
  /* Particle(control_type="velocity") particle.py:41-48, 152-153: a two-state model (ds = 2): acts.clamp_(+-max_speed), x_dot = acts,
   * and the closing clamp (particle.py:165) lands on the positions */
  int velocity_ctrl;
  /* Particle(deterministic=False) particle.py:145-148: acts += dyn_std * randn_like(acts) in every model.step call.  ctrl_noise holds
   * the recorded standard-normal draws [H][M*S*N][da] of one MultiDISCO._rollout (rollout r = (m*S + s)*N + n), or NULL (no noise) */
  float dyn_std[2];
  const float *ctrl_noise;
  /* FULL 2 x 2 covariances (da = 2): MultiDISCO(a_cov=) disco.py:91-98 - policy noise actions = theta + L_a eps with L_a = cholesky(a_cov)
   * (likelihoods.py:85-90, MultivariateNormal.rsample), control cost through a_pre = inverse(a_cov) (disco.py:338-346) - and the prior
   * GMM's component covariance (svgd.py:84-89; MultivariateNormal.log_prob: Mahalanobis distance by a triangular solve with L_p,
   * minus sum log diag L_p).  full_cov != 0: the diagonal arguments (chol_a / a_pre_diag / sigma_p) of the functions below are ignored
   * in favour of these. */
  int full_cov;
  float chol_a_full[3]; /* l00, l10, l11 */
  float a_pre_full[3];  /* p00, p01 (= p10), p11 */
  float chol_p_full[3];
} orc_cfg;

/* a1  CostLikelihood.sample likelihoods.py:81-101: actions = theta + L eps (diagonal L) */
void orc_sample_actions(const orc_cfg *c, const float *theta, const float *eps, const float *chol_a, float *actions);

/* a2-a5  MultiDISCO._rollout + _compute_cost, disco.py:139-209, 294-346.
 * params: raw samples [M][P] (NULL when no sampling).  states_out may be NULL ([M][S][N][H+1][ds]).
 * a_mat/a_seq/a_pre_diag only used when a_reg != 0. */
void orc_rollout_cost(const orc_cfg *c, const float *state, const float *actions, const float *params,
                      float a_reg, const float *a_mat, const float *a_seq, const float *a_pre_diag,
                      float *states_out, float *costs);

/* a6  MultiDISCO.forward disco.py:380-393: omega, a_mat += sum_s omega eps, a_mix.
 * eps = actions - eps_base: eps_base = a_seq [H*da] for external actions (disco.py:161-164, base_per_policy 0),
 * or the pre-update a_mat [N][H*da] for internally sampled noise (disco.py:155-160, base_per_policy 1). */
void orc_rollout_cost_ut(const orc_cfg *c, const float *state, const float *actions, const float *params_sp, const float *w,
                         float a_reg, const float *a_mat, const float *a_seq, const float *a_pre_diag, float *costs);
void orc_disco_weights(const orc_cfg *c, const float *costs, const float *actions, const float *eps_base,
                       int base_per_policy, float temp, float *omega, float *a_mat, float *a_mix);

/* log mixture weights as torch builds them: log_softmax(log(clamp(w / sum w, eps, 1-eps)))
 * (torch.distributions.Categorical + MixtureSameFamily.log_prob) */
void orc_log_mix(int n, const float *weights, float *logmix);

/* a9  score part of SVMPC.phi svmpc.py:38-56 */
void orc_score(const orc_cfg *c, const float *theta, const float *mu, const float *logmix, const float *sigma_p,
               const float *costs, const float *actions, float alpha, const float *sigma_a, float *grad_lik,
               float *grad_pri, float *score);

/* a10 kernel branch K1 svmpc.py:76-83 with gpytorch-RBF semantics, lengthscale ln2.
 * variant 0: exact pairwise differences (double); variant 1: gpytorch's fp32 mean-centred matmul trick. */
void orc_phi_k1(int N, int D, const float *theta, const float *score, int variant, float *phi, float *gram);

/* IMQ kernel k = (1 + |x-y|^2/l^2)^(-1/2), same phi structure as K1 (new feature, no reference) */
void orc_phi_imq(int N, int D, const float *theta, const float *score, float ell, float *phi);

/* a11 kernel branch K2 svmpc.py:64-74 -> composite_kernels.py:33-64 -> base_kernels.py:53-108 */
void orc_phi_k2(int N, int H, int da, int indep, float bw_scale, float fixed_bw /* < 0: median trick */, float min_bw,
                const float *theta, const float *score, float *phi,
                float *h_out);

/* a8 SGD step svmpc.py:87-95 */
void orc_sgd(int n, float lr, const float *phi, float *theta);
/* torch.optim.Adam step on theta.grad = -phi (svgd.py:115, svmpc.py:87-95); step counts from 1 after every roll */
void orc_adam(int n, float lr, float beta1, float beta2, float eps, int step, const float *phi, float *theta, float *m, float *v);

/* a12 SVMPC.forward svmpc.py:128-200 + likelihoods.py:113-135 + svgd.py:84-89 */
void orc_forward(const orc_cfg *c, int lik_kind, float alpha, const float *costs, float *theta, float *mu,
                 float *mix_weights, const float *sigma_p, int weighted_prior, int roll_strategy, float *log_l,
                 float *log_p, float *p_weights, int *i_star, float *a_seq);

/* a14 MultiDISCO.step disco.py:396-417. strategy 0 argmax, 1 average, 2 external */
void orc_disco_step(int N, int H, int da, int strategy, int steps, const float *min_a, const float *max_a,
                    const float *ext, float *a_mat, const float *a_mix, float *a_seq, float *next_actions);

/* one model step for `n` rows (PendulumModel.step / Particle.step), params raw [n][P] or NULL */
void orc_model_step(const orc_cfg *c, int n, const float *states, const float *actions, int action_rows,
                    const float *params, float *next);

/* a13 MPF mpf.py:26-86 + GaussianLikelihood likelihoods.py:30-64 + default_kernel svgd.py:92-99.
 * x [Mp][P] updated in place by n_steps SGD steps; grad_norms[n_steps]. */
void orc_mpf_phi(const orc_cfg *c, int Mp, const float *x, const float *prior_means, float prior_bw, const float *past_obs,
                 const float *past_action, const float *obs, float obs_std, int log_space, float bw, float *phi);
void orc_mpf_optimize(const orc_cfg *c, int Mp, float *x, float *prior_means, float *prior_bw, const float *past_obs,
                      const float *past_action, const float *obs, float obs_std, int log_space, float bw, float lr,
                      int n_steps, float *grad_norms);
void orc_mpf_optimize_adam(const orc_cfg *c, int Mp, float *x, float *prior_means, float *prior_bw, const float *past_obs,
                           const float *past_action, const float *obs, float obs_std, int log_space, float bw, float lr, float beta1,
                           float beta2, float eps, float *m, float *v, int *step, int n_steps, float *grad_norms);
void orc_gmm_log_prob(int n, int K, int P, const float *x, const float *means, float bw, float *out);

/* occupancy lookup obstacle_map.py:64-93 */
void orc_get_collisions(const orc_cfg *c, int n, const float *xy, float *out);

/* Whole-tick driver used only for the cpu_baseline timing leg of bench.py (OpenMP over rollouts / particles):
 * n_iters x (sample actions, rollout+cost, disco weights, score, phi K1, SGD) then forward.  eps [n_iters][S][N][H][da]. */
void orc_tick_k1(const orc_cfg *c, const float *state, float *theta, float *mu, float *mix_weights, const float *sigma_p,
                 const float *sigma_a, const float *eps, int n_iters, float alpha, float lr, float *a_mat, float *a_seq_out,
                 float *p_weights, float *costs_out);
int orc_num_threads(void);
int orc_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
