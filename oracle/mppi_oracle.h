/*
 * mppi_oracle.h -- CPU restatement of the reference MPPI hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under autorally_amd/ (the product) may
 * include, link or call this.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker / baseline.
 *
 * Parity status: the reference has no tests, golden vectors or fixtures for
 * this path and its CUDA sources cannot be built here (nvcc, cuRAND, Eigen,
 * cnpy, ROS absent), so the rollout / cost / weighting / reduction parts of
 * this oracle are "parity unpinned": they restate the reference source line by
 * line.  The NN dynamics step (a4+a5) IS pinned: tests/golden/nn_dynamics_golden.npz
 * holds outputs of the reference's own Python restatement
 * (scripts/ml_pipeline/utils.py) and tests/test_oracle_golden.py checks this
 * file against them.  The costmap file format (costs.cu:190-232) is pinned by a file the reference's own
 * scripts/track_converter.py wrote (tests/golden/costmap_track_converter.npz).  The MRG32k3a recurrence is pinned by L'Ecuyer's published
 * RngStreams jump matrices (A1p76, A2p76, A1p127, A2p127).
 *
 * Paths below are relative to /root/reference/autorally_control/ :
 *   PI/  = include/autorally_control/path_integral/
 */
#ifndef MPPI_ORACLE_H_
#define MPPI_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_STATE_DIM 7
#define ORC_CONTROL_DIM 2
#define ORC_MAX_LAYERS 8

/* PI/costs.cuh:67-85 (CostParams) + l1_cost_ (costs.cuh:276). */
typedef struct {
  float desired_speed;
  float speed_coeff;
  float track_coeff;
  float max_slip_ang;
  float slip_penalty;
  float track_slop;
  float crash_coeff;
  float steering_coeff;
  float throttle_coeff;
  float boundary_threshold;
  float discount;
  int num_timesteps; /* unused by the reference kernels (Q16) */
  int grid_res;      /* unused (Q16) */
  float r_c1[3];
  float r_c2[3];
  float trs[3];
  int l1_cost;
} orc_cost_params;

typedef struct {
  int K;              /* rollouts (multiple of 64, mppi_controller.cuh:58-60) */
  int T;              /* num_timesteps */
  int n_layers;       /* entries of layers[] incl. input and output */
  int layers[ORC_MAX_LAYERS];
  const float *theta; /* packed [W1|b1|W2|b2|...], neural_net_model.cu:120-141 */
  float dt;           /* (float)(1.0/hz), path_integral_main.cu:100 */
  float nu[2];        /* exploration std */
  float u_lo[2], u_hi[2];
  int negate_yaw_der;
  int opt_delay;      /* optimization_stride, mppi_controller.cu:616 */
  float gamma;
  orc_cost_params P;
  int map_w, map_h;
  const float *map_rgba; /* float4[H][W], x fastest, costs.cu:206-216 */
  int fma_mode;       /* 1: nvcc-style FMA contraction (nominal), 0: none,
                         2: as 1 with the output layer summed in the row-tree kernels' order (mppi_oracle.c: out_tree_dot),
                         3: as 1 with the output layer in the 4x4x1-MFMA kernel's order (out_tree_dot_m44),
                         4: as 1 with the output layer in the multi4-tree kernel's order (out_tree_dot_multi),
                         5: as 3 with every 64-input hidden layer as two accumulation chains, even / odd k (hidden_split2_dot) */
  int nthreads;       /* OpenMP threads for the k loop; <=1 = serial */
  /* Second dynamics family: GeneralizedLinear<CarBasisFuncs,7,2,25,CarKinematics,3>,
   * PI/generalized_linear.cu:169-245 + PI/car_bfs.cuh:44-120.  bf_W != NULL selects it (theta/layers
   * are then unused): row-major [4][25], the .npz key "W" (generalized_linear.cu:99-106). */
  const float *bf_W;
} orc_problem;

#define ORC_NUM_BFS 25
/* CarBasisFuncs::basisFuncX, car_bfs.cuh:44-120 (source types kept: double where the literal is) */
float orc_basis_func(int idx, const float *s, const float *u);
void orc_set_bf_pow_products(int on); /* test knob, see mppi_oracle.c */

/* OpenMP team size used by loops that have no orc_problem (weighted reduction). */
void orc_set_num_threads(int n);
int orc_num_params(const int *layers, int n_layers);

/* neural_net_model.cu:357-410 (device computeDynamics): in[layers[0]] -> out[layers[L-1]] */
void orc_nn_forward(const float *theta, const int *layers, int n_layers,
                    const float *in, float *out, int fma_mode);

/* computeKinematics + computeDynamics: neural_net_model.cu:346-355, 357-410 */
void orc_state_deriv(const orc_problem *p, const float *s, const float *u, float *sd);

/* host updateState: clamp, deriv, Euler step. neural_net_model.cu:266-288 */
void orc_update_state(const orc_problem *p, float *s, float *u);

/* MPPICosts::computeCost, costs.cu:396-409; crash is in/out. */
float orc_compute_cost(const orc_problem *p, const float *s, const float *u,
                       const float *du, int *crash);

/* debugCostKernel, PI/debug_kernels.cuh:39-88: out[height_m*ppm][width_m*ppm], see mppi_oracle.c */
void orc_debug_cost_raster(const orc_problem *p, float x, float y, float heading, int width_m,
                           int height_m, int ppm, float *out);

/* rolloutKernel, mppi_controller.cu:72-184.
 * epsV: [K][T][2]; in = N(0,1) noise, out = applied (unclamped) controls (Q3).
 * costs[K]; crash_out[K] optional (final sticky crash flag). */
void orc_rollouts(const orc_problem *p, const float *state, const float *U,
                  float *epsV, float *costs, int *crash_out);

/* host baseline + normExpKernel + normaliser: mppi_controller.cu:627-652, 193-203.
 * w[K] out (un-normalised exp weights). */
void orc_weights(const float *costs, int K, float gamma, float *w, float *baseline,
                 float *eta, float *traj_cost);

/* weightedReductionKernel: mppi_controller.cu:219-267. */
void orc_weighted_reduction(const float *w, float eta, const float *V, int K, int T,
                            float *Unew, int fma_mode);

/* savitskyGolay: mppi_controller.cu:468-499. U in/out [T][2], hist[4]. */
void orc_savgol(float *U, const float *hist, int T);

/* computeNominalTraj: mppi_controller.cu:501-519. */
void orc_nominal_traj(const orc_problem *p, const float *state, const float *U,
                      float *state_seq, float *control_seq);

/* slideControlSeq: mppi_controller.cu:527-554 (Q15 reproduced). */
void orc_slide_control_seq(float *U, float *hist, const float *init_u, int T, int stride);

/* computeControl(state): mppi_controller.cu:600-675.
 * eps: [num_iters][K][T][2] (overwritten with applied controls). U in/out.
 * costs/w: [K] of the LAST iteration (optional). */
void orc_compute_control(const orc_problem *p, int num_iters, const float *state,
                         float *U, const float *hist, float *eps, float *traj_cost,
                         float *costs, float *w);

/* ---- feedback gains (ddp_oracle.c, parity UNPINNED -- see its header) ----
 * computeFeedbackGains -> DDP::run, mppi_controller.cu:402-441, ddp/ddp.h:49-157.
 * feedback [T][2][7], feedforward [T][2], xout [T][7], uout [T][2]. Returns 0, or 1 where the
 * reference exits on a failed LDLT. */
int orc_ddp_feedback_gains(const float *theta, const int *layers, int n_layers, int T, float dt,
                           const float *u_lo, const float *u_hi, int negate_yaw_der, const float *Q,
                           const float *R, const float *Qf, const float *x0, const float *target_x,
                           const float *target_u, float *feedback, float *feedforward, float *xout,
                           float *uout, float *total_cost, const float *bf_W /* NULL: network model */);

/* ---- noise generator spec (this build's own; cuRAND's XORWOW stream is not reproducible) ---- */
typedef struct { uint32_t s1[3]; uint32_t s2[3]; } orc_mrg_state;
void orc_mrg_seed(orc_mrg_state *st, uint64_t seed);
void orc_mrg_skip_subsequences(orc_mrg_state *st, uint64_t n); /* n * 2^76 draws */
void orc_mrg_skip(orc_mrg_state *st, uint64_t n);              /* n draws */
double orc_mrg_next_u01(orc_mrg_state *st);
uint32_t orc_mrg_next_z(orc_mrg_state *st);
/* 3x3 matrices of A1^(2^e) mod m1 and A2^(2^e) mod m2, row-major. */
void orc_mrg_jump_matrices(int e, uint32_t A1[9], uint32_t A2[9]);
void orc_box_muller(float u1, float u2, float *n0, float *n1);
/* eps[K][T][2]; rollout k uses subsequence k, starting at draw `offset`. */
void orc_generate_noise(uint64_t seed, uint64_t offset, int K, int T, float *eps);

#ifdef __cplusplus
}
#endif
#endif
