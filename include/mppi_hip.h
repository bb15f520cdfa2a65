/*
 * mppi_hip.h -- C ABI of libmppi_hip.so, the MI355X (gfx950) MPPI solver.
 *
 * Drop-in boundary for the rollout hot path of rdesc/autorally's
 * autorally_control/src/path_integral.  The reference has no FFI for this path:
 * its boundary is the C++ template class MPPIController<DYNAMICS_T, COSTS_T, ...>
 * plus two cnpy-loaded .npz formats.  Each entry point below names the reference
 * interface it replaces; paths are relative to /root/reference/autorally_control/,
 *   PI/ = include/autorally_control/path_integral/.
 *
 * Conventions: extern "C"; opaque handle; every call returns an int status
 * (MPPI_OK == 0); no exceptions cross the ABI; the caller owns every host
 * buffer and the library copies; a handle owns all its device memory and one HIP
 * stream; a handle is single-threaded, distinct handles may be driven from
 * distinct host threads (one per GPU).  On error outputs are left untouched
 * (the reference logs CUDA errors and continues, PI/gpu_err_chk.h:74).
 *
 * There is NO CPU fallback: every compute entry point fails with
 * MPPI_ERR_NO_DEVICE / MPPI_ERR_HIP when no gfx950 device is usable.
 */
#ifndef MPPI_HIP_H_
#define MPPI_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: + mppi_set_costmap_transform, mppi_savitsky_golay, mppi_compute_control_batch[_async], mppi_control_ticks_batch, mppi_nominal_traj_pair,
 *    mppi_debug_inject_handover_fault; mppi_slide_control_seq(h, 0) is MPPI_OK (was MPPI_ERR_INVALID); "fused" =
 *    four-wavefront workgroups; variant names "_fused_b256", "_3w", "_multiN", "_oct8w", "valu_row8w_*"; variants "row", "multi4u". */
/* 3: + mppi_set_host_threads, mppi_compute_feedback_gains_pair. */
/* 4: + mppi_debug_capture_iterations, mppi_debug_get_iterations, mppi_set_wait_timeout, mppi_debug_form_candidates; variants
 *    "row_tree", "row_exact", "row64[_r8|_r16]", "m44"; names "valu_row8w_tree_*", "valu_row64_r*_tree_*", "mfma4x4x1_*_m44_split_tree" (automatic), "mfma4x4x1_*_m44_tree" ("m44_chain"). */
/* 5: solves of more than 4096 rollouts run a one-launch tail (in-launch hand-overs with a deadline: fault roles 32-34 of
 *    mppi_debug_inject_handover_fault); after a wait timeout the lost solve is not waited for again (mppi_set_wait_timeout);
 *    "mfma" / "valu" / "valu_lds" drop a form forced by name; variants "multi1", "multi4u[_gen]", "row64_r8" removed;
 *    + mppi_debug_set_chained_ticks (mppi_control_ticks enqueues one solve ahead), + mppi_debug_min_cost. */
#define MPPI_ABI_VERSION 5
#define MPPI_STATE_DIM 7   /* [x, y, yaw, roll, u_x, u_y, yaw_mder]  NeuralNetModel<7,2,3,...> */
#define MPPI_CONTROL_DIM 2 /* [steering, throttle] */
#define MPPI_MAX_LAYERS 8

enum {
  MPPI_OK = 0,
  MPPI_ERR_INVALID = 1,   /* bad argument / size mismatch */
  MPPI_ERR_NO_DEVICE = 2, /* no usable gfx950 device */
  MPPI_ERR_HIP = 3,       /* a HIP runtime call failed; see mppi_last_error */
  MPPI_ERR_STATE = 4,     /* call order (e.g. solve before set_nn_params / set_costmap) */
  MPPI_ERR_UNSUPPORTED = 5
};

typedef struct mppi_handle mppi_handle;

/* Constructor arguments of MPPIController (PI/mppi_controller.cuh:101-102,
 * PI/mppi_controller.cu:321-363) + NeuralNetModel(dt, control_rngs)
 * (PI/neural_net_model.cu:38-67) + the template constants of
 * src/path_integral/path_integral_main.cu:65-78 made runtime values. */
typedef struct {
  int device;                 /* HIP device ordinal */
  int num_rollouts;           /* K; must be a multiple of 64 (mppi_controller.cuh:58-60) */
  int num_timesteps;          /* T */
  int hz;                     /* dt = (float)(1.0/hz) */
  int optimization_stride;    /* opt_delay of rolloutKernel */
  float gamma;
  int num_iters;
  int n_layers;               /* entries of layers[], input and output included; 0 selects the reference's
                               * other dynamics family, GeneralizedLinear<CarBasisFuncs,7,2,25,CarKinematics,3>
                               * (path_integral_main.cu:70-74): parameters by mppi_set_bf_params */
  int layers[MPPI_MAX_LAYERS];/* e.g. {6,32,32,4}; layers[0]==6, layers[n-1]==4 */
  float exploration_std[2];   /* nu */
  float init_control[2];      /* init_u */
  float control_min[2];       /* control_rngs_[i].x */
  float control_max[2];       /* control_rngs_[i].y */
  int negate_yaw_der;
  uint64_t seed;              /* reference: 1234 (mppi_controller.cu:331) */
} mppi_config;

/* MPPICosts::CostParams scalars (PI/costs.cuh:67-85) + l1_cost_ (costs.cuh:276).
 * The transform (r_c1, r_c2, trs) travels with the costmap, see mppi_set_costmap. */
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
  int l1_cost;
} mppi_cost_params;

/* Per-stage device times of the most recent solves (HIP events on the handle's stream). */
typedef struct {
  int n_solves;            /* solves accumulated since mppi_reset_stage_times */
  float noise_ms;          /* sums over n_solves */
  float rollout_ms;
  float weights_ms;
  float reduction_ms;      /* weighted reduction + Savitzky-Golay */
  float total_ms;          /* first launch -> last kernel end */
} mppi_stage_times;

int mppi_abi_version(void);
const char *mppi_strerror(int status);
/* Last HIP/validation error text of this handle ("" if none). */
const char *mppi_last_error(const mppi_handle *h);
/* Number of visible HIP devices whose architecture is gfx950 (0 if none / no runtime). */
int mppi_device_count(void);

/* MPPIController::MPPIController + allocateCudaMem (mppi_controller.cu:321-387). */
int mppi_create(const mppi_config *cfg, mppi_handle **out);
/* deallocateCudaMem (mppi_controller.cu:389-400), without the double free of Q14. */
int mppi_destroy(mppi_handle *h);

/* GeneralizedLinear::setParams / loadParams / paramsToDevice (PI/generalized_linear.cu:86-118):
 * W row-major [4][25] (the .npz key "W"), n == 100.  Only for handles created with n_layers == 0.
 * The basis functions are CarBasisFuncs (PI/car_bfs.cuh:44-120); the yaw rate is always negated
 * (generalized_linear.cu:216), negate_yaw_der is ignored. */
int mppi_set_bf_params(mppi_handle *h, const float *W, size_t n);
/* NeuralNetModel::paramsToDevice (neural_net_model.cu:120-150): theta is the packed
 * [W1|b1|W2|b2|...] blob, n == NUM_PARAMS. The .npz is read by the C++ host layer. */
int mppi_set_nn_params(mppi_handle *h, const float *theta, size_t n);
/* NeuralNetModel::updateModel (neural_net_model.cu:152-180): data = [W1|W2|..|b1|b2|..]. */
int mppi_update_model(mppi_handle *h, const int *description, int n_desc, const float *data, size_t n);
/* control_rngs_ (cutThrottle, mppi_controller.cu:460-466, sets control_max[1]=0). */
int mppi_set_control_limits(mppi_handle *h, const float umin[2], const float umax[2]);

/* MPPICosts::loadTrackData/updateTransform/costmapToTexture (costs.cu:190-232, 175-188,
 * 128-154): rgba = float4[H][W] (x fastest).  Texture semantics reproduced: point
 * filter, clamp, normalised coordinates. */
int mppi_set_costmap(mppi_handle *h, int width, int height, const float *rgba,
                     const float r_c1[3], const float r_c2[3], const float trs[3]);
/* MPPICosts::updateTransform + paramsToDevice (costs.cu:175-188, 234-238): the coordinate transform alone
 * (r_c1, r_c2, trs are public members of CostParams; the texture stays). */
int mppi_set_costmap_transform(mppi_handle *h, const float r_c1[3], const float r_c2[3], const float trs[3]);
/* costmapToTexture(float*, channel) (costs.cu:101-126). */
int mppi_set_costmap_channel(mppi_handle *h, int channel, const float *data, size_t n);
/* updateParams / updateParams_dcfg / paramsToDevice (costs.cu:156-173, 75-87, 234-238). */
int mppi_set_cost_params(mppi_handle *h, const mppi_cost_params *p);

/* U_ accessors (resetControls :448-458; setControlSequence-like). U is [T][2]. */
int mppi_reset_controls(mppi_handle *h);
int mppi_set_control_seq(mppi_handle *h, const float *U, size_t n);
int mppi_get_control_seq(mppi_handle *h, float *U, size_t n);
/* control_hist_ (mppi_controller.cu:347, 528-541): two controls executed before U_0. */
int mppi_set_control_hist(mppi_handle *h, const float hist[4]);
int mppi_get_control_hist(mppi_handle *h, float hist[4]);
/* savitskyGolay() (mppi_controller.cuh:134, mppi_controller.cu:468-499) as a call of its own: smooths the
 * handle's current U_ in place with control_hist_ as left padding (computeControl already ends with it). */
int mppi_savitsky_golay(mppi_handle *h);
/* slideControlSeq (mppi_controller.cu:527-554), flat-index quirk for stride>2 kept. */
int mppi_slide_control_seq(mppi_handle *h, int stride);

/* Replaces curandCreateGenerator/SetSeed (mppi_controller.cu:330-331): this build's
 * MRG32k3a generator, one 2^76-draw subsequence per rollout, `offset` draws skipped. */
int mppi_seed(mppi_handle *h, uint64_t seed, uint64_t offset);
/* Explicit-noise mode (what the parity tests use): eps = [num_iters][K][T][2] N(0,1),
 * consumed by the NEXT mppi_compute_control / mppi_rollout_only instead of the generator. */
int mppi_set_noise(mppi_handle *h, const float *eps, size_t n);
/* Runs the generator for one iteration worth of draws ([K][T][2]) and copies it out;
 * advances the stream exactly as a solve iteration would. */
int mppi_generate_noise(mppi_handle *h, float *eps_out, size_t n);

/* MPPIController::computeControl(state) up to and including savitskyGolay()
 * (mppi_controller.cu:600-671); computeNominalTraj is mppi_nominal_traj. Blocking. */
int mppi_compute_control(mppi_handle *h, const float state[MPPI_STATE_DIM]);
/* n_ticks times { mppi_compute_control(state); mppi_slide_control_seq(stride) } without returning to the
 * caller in between: the body of runControlLoop's tick for one controller (run_control_loop.cuh:207-219)
 * for callers -- bench.py -- whose own per-call overhead (an interpreter, an FFI) would otherwise sit
 * inside the solve-to-solve time.  stride = 0 skips the slide.  Stops at the first error. */
int mppi_control_ticks(mppi_handle *h, const float state[MPPI_STATE_DIM], int n_ticks, int stride);
/* Same solve, enqueued only; results are valid after mppi_synchronize. */
int mppi_compute_control_async(mppi_handle *h, const float state[MPPI_STATE_DIM]);
int mppi_synchronize(mppi_handle *h);
/* The solves of n independent controllers enqueued together -- runControlLoop's tick
 * (PI/run_control_loop.cuh:218-219: actual-state and predicted-state controller, both of path_integral_main.cu:119-122)
 * -- states is [n][7], handles[i] solves from states + 7 i.  Where the handles' rollout kernels can share a launch
 * (network model, the four-wavefront form, same layer list and num_iters, all groups of 16 rollouts together at
 * most one per CU: 2 x K=1920 on 256 CUs) the n solves cost TWO kernel launches in all, on a stream of the library
 * shared by the device's handles; otherwise this is n calls of mppi_compute_control_async.  Either way every
 * handle's results are bit for bit those of its own mppi_compute_control, collected per handle with
 * mppi_synchronize / mppi_get_results.  The blocking form waits for all of them. */
int mppi_compute_control_batch_async(mppi_handle *const *handles, const float *states, int n);
int mppi_compute_control_batch(mppi_handle *const *handles, const float *states, int n);
/* mppi_control_ticks for several controllers: n_ticks times { mppi_compute_control_batch; mppi_slide_control_seq
 * (handles[i], stride) for every i } -- the solve part of runControlLoop's tick for its two controllers. */
int mppi_control_ticks_batch(mppi_handle *const *handles, const float *states, int n, int n_ticks, int stride);
/* getComputedTrajectoryCost (:683-687) + optional per-rollout vectors of the last
 * iteration: costs[K] (traj_costs_ after the rollout), weights[K] (after normExpKernel). */
int mppi_get_results(mppi_handle *h, float *U, float *traj_cost, float *costs, float *weights);
/* The rewritten du_d buffer of the last iteration, [K][T][2] (applied, unclamped controls, Q3). */
int mppi_get_applied_controls(mppi_handle *h, float *V, size_t n);
/* Stage-level entry: noise (or explicit noise) + rolloutKernel only; costs[K]. */
int mppi_rollout_only(mppi_handle *h, const float state[MPPI_STATE_DIM], float *costs);
/* computeNominalTraj (mppi_controller.cu:501-519), host replay of U_ like the reference:
 * state_seq [T][7], control_seq [T][2] (clamped). */
int mppi_nominal_traj(mppi_handle *h, const float state[MPPI_STATE_DIM], float *state_seq,
                      float *control_seq);

/* computeNominalTraj of the two controllers of a control tick (run_control_loop.cuh:218-219: both end their
 * computeControl with it) in one call: two network replays of the same length advance in lockstep on the host, at about the
 * cost of one; results bit for bit those of two mppi_nominal_traj calls (which is what this does for any other pair). */
int mppi_nominal_traj_pair(mppi_handle *ha, const float state_a[MPPI_STATE_DIM], float *state_seq_a, float *control_seq_a,
                           mppi_handle *hb, const float state_b[MPPI_STATE_DIM], float *state_seq_b, float *control_seq_b);

/* computeFeedbackGains (PI/mppi_controller.cu:431-441) -> DDP::run (ddp/ddp.h:49-157), one
 * iteration, dt = 1/hz, from `state`, tracking target_state_seq [T][7] / target_control_seq [T][2]
 * -- the reference passes state_solution_ / control_solution_ of the last computeControl, which for
 * the predicted-state controller were NOT computed from `state` (run_control_loop.cuh:218-225).
 * Both NULL: the nominal trajectory of the current control sequence from `state`.
 * Host computation like the reference (T Jacobians of the network via computeGrad,
 * neural_net_model.cu:233-264, and T-1 sequential 7x7 Riccati steps).  Weights default to initDDP's
 * Q = diag(.5,.5,.25,0,.05,.01,.01), R = diag(10,10), Qf = 0 (mppi_controller.cu:410-417).
 * Returns MPPI_ERR_STATE where the reference exits (-3) on a failed LDLT. */
int mppi_set_ddp_weights(mppi_handle *h, const float Q[MPPI_STATE_DIM], const float R[MPPI_CONTROL_DIM],
                         const float Qf[MPPI_STATE_DIM]);
int mppi_compute_feedback_gains(mppi_handle *h, const float state[MPPI_STATE_DIM],
                                const float *target_state_seq, const float *target_control_seq);
/* getFeedbackGains (:443-446), OptimizerResult (ddp/result.h): feedback [T][2][7] (the last one is
 * zero), feedforward [T][2], state_traj [T][7], control_traj [T][2], total_cost; any may be NULL. */
int mppi_get_feedback_gains(mppi_handle *h, float *feedback, float *feedforward, float *state_traj,
                            float *control_traj, float *total_cost);

/* computeFeedbackGains of the two controllers of a control tick (run_control_loop.cuh:220-225: both, one after the other, on
 * the optimizer thread) in one call; results bit for bit those of two mppi_compute_feedback_gains calls.  With
 * mppi_set_host_threads(2) the two DDP passes run side by side. */
int mppi_compute_feedback_gains_pair(mppi_handle *ha, const float state_a[MPPI_STATE_DIM], const float *target_state_seq_a,
                                     const float *target_control_seq_a, mppi_handle *hb, const float state_b[MPPI_STATE_DIM],
                                     const float *target_state_seq_b, const float *target_control_seq_b);

/* Host threads the library may use for the host-side work of a tick that comes in pairs (mppi_nominal_traj_pair,
 * mppi_compute_feedback_gains_pair): 1 (default) = the caller's thread only (the reference's optimizer thread does everything,
 * path_integral_main.cu:136-138); 2 = one helper thread, process-wide, which sleeps between ticks and is woken by
 * mppi_compute_control_batch_async.  Results do not depend on the setting. */
int mppi_set_host_threads(int n);

/* MPPICosts::getDebugDisplay without the OpenCV display (costs.cu:272-285 -> debugCostKernel,
 * PI/debug_kernels.cuh:39-88): the costmap around (x, y), width_m x height_m metres at ppm pixels per
 * metre, car marker included; out is [height_m*ppm][width_m*ppm] (n floats), index
 * (H - (yi + 1)) * W + xi as in the reference.  The reference never writes row yi = 0 and flat index
 * 0 (they show whatever the buffer held); here they are 0.  debugDisplayInit's default is 10, 10, 50. */
int mppi_debug_cost_raster(mppi_handle *h, float x, float y, float heading, int width_m, int height_m,
                           int ppm, float *out, size_t n);

/* Measurement hooks.  on = 1: HIP events around every stage of every solve; on = N > 1: only on
 * every Nth solve (event packets between kernels lengthen the launch gaps, so sampling keeps the
 * measured run close to the unmeasured one); on = 0: off. */
int mppi_enable_stage_timing(mppi_handle *h, int on);
int mppi_reset_stage_times(mppi_handle *h);
int mppi_get_stage_times(mppi_handle *h, mppi_stage_times *out);
/* Name of the rollout kernel form in use, and mppi_set_rollout_variant's names for forcing one.  What is LOAD-BEARING:
 *
 *   automatic choice (csrc/abi_forms.hip: kFormRules, by model shape and 16-rollout groups per CU)
 *     "row_tree"         valu_row8w_tree_h32_l2             6-32-32-4, K <= 8192: the headline form (vector ALU; output layer a butterfly)
 *     "m44"              mfma4x4x1_h64_l<N>_m44_split_tree  64-wide nets, K <= 8192 (hidden layers as two chains, output a butterfly)
 *     "quad"             mfma16x16x4_h<H>_l<N>_quad4w       other shapes up to one group per CU (e.g. 6-32x4-4), and under "mfma"
 *     "multi2"           ..._multi2                         up to two groups per CU
 *     "multi4_tree_gen"  ..._multi4_tree_gen                beyond (BASELINE config 4; eps from the stand-alone generator kernel)
 *     "fused"            ..._fused_b256                     shapes without a multi form beyond two groups per CU (6-64x4-4, K > 8192)
 *     (generic)          valu_lds                           any other layer list: the only form for non-uniform nets
 *     basis functions    basis_funcs25_valu[_2w|_3w]        "fused" | "quad" | "bf3"
 *   the reference's summation order in EVERY layer (bit-identical to one another; "mfma" = the table restricted to them)
 *     "row_exact" (= "row") valu_row8w_h32_l2, "m44_chain" mfma4x4x1_*_m44_tree (hidden layers one chain; output a butterfly),
 *     "oct[_gen]" ..._oct8w, "quad", "multi2[_gen]", "multi4[_gen]", "fused" = "block256" | "block64" ..._fused_b256 / _b64
 *   A/B arms and cross-checks (never chosen automatically)
 *     "valu" valu_reg_lds (lane = rollout, the independent implementation every parity test also runs; config 4's untuned
 *     vector-ALU reference), "valu_lds" (the generic kernel on a standard shape), "row64" = "row64_r16"
 *     valu_row64_r16_tree_h64_l<N> (config 4's hand-scheduled vector-ALU arm), "multi4_tree" (in-kernel generator)
 *   "auto" restores the table.  "mfma", "valu", "valu_lds" also drop a form forced by name earlier.
 * The re-associated forms ("_tree", "_split") do NOT compute the reference's summation order: inside the 1e-4 tolerance on
 * the controls (profiles/r05_b_nominal_margin_*.txt), not bit-identical to the exact ones.  Removed in ABI 5 (no table row chose
 * them, no A/B needed them): "multi1", "multi4u[_gen]", "row64_r8".  MPPI_ERR_UNSUPPORTED if the handle's model has no such
 * form; a form name given to a handle whose model runs on the generic vector kernel is accepted and ignored. */
const char *mppi_rollout_variant(const mppi_handle *h);
int mppi_set_rollout_variant(mppi_handle *h, const char *name);

/* Test hook (not part of the drop-in surface): d/dt of n independent (state[7], control[2])
 * pairs through the SAME device functions the rollout kernel uses (computeKinematics +
 * computeDynamics, neural_net_model.cu:346-410); ders is [n][7]. */
int mppi_debug_dynamics(mppi_handle *h, int n, const float *states, const float *controls, float *ders);

/* Test hook (not part of the drop-in surface): from the next solve on, wavefront role `wave` of the
 * multi-wavefront rollout kernels starts with an exhausted poll budget, i.e. it never waits for its
 * partners -- the state of a wave that gave up on a hand-over -- and every other wave may spend
 * spin_budget + 64 T polls in total (0: the default).  Roles of the four-wavefront network kernel: 1, 2 =
 * dynamics waves, 3 = cost wave, 4 = control wave; of the two-wavefront basis-function kernel: 1 = dynamics,
 * 2 = cost.  The solve must then end in MPPI_ERR_HIP ("hand-over failed"), never in finite costs.
 * wave = 0 and spin_budget = 0 restore normal operation.
 * Roles of the row form and of the oct form: 1 .. 4 = dynamics waves, 5 = pose, 6 = cost, 7 = control, 8 = noise wave.
 * Roles of the multi form: 1 .. ND = dynamics waves, then the cost wave and the control wave (ND = 4: the pose
 * wave, the cost wave, the control wave).
 * Roles 32 .. 34: waits of the one-launch tail kernel of solves with more than 4096 rollouts -- the weights workgroup of
 * chunk 0 never publishes its chunk sum of weights (32), chunk 0 of every row never publishes its chain results (33), no
 * weights workgroup publishes its chunk sum, nor beta where the tail kernel takes the minimum itself (34) -- with a deadline of spin_budget microseconds for every wait (0: the default, 20 ms). */
int mppi_debug_inject_handover_fault(mppi_handle *h, int wave, int spin_budget);

/* Test hooks (not part of the drop-in surface): what EVERY iteration of a solve with num_iters > 1 left behind.  The
 * reference's iteration loop (PI/mppi_controller.cu:609-667) re-uses U_ -- the raw weighted mean, unsmoothed -- as the
 * nominal sequence of the next iteration, so from iteration 2 on a comparison with another implementation is a
 * comparison on identical inputs only if that implementation is started from THIS one's U of the iteration before.
 * mppi_debug_capture_iterations(h, 1): from the next solve on, keep them (two small device copies per iteration; such a
 * handle is solved on its own, never inside a batched launch); 0: off.
 * mppi_debug_get_iterations: U_raw [num_iters][T][2] = the weighted mean after iteration i before any smoothing
 * (weightedReductionKernel's output, :219-267), costs [num_iters][K], V [num_iters][K][T][2] = the applied controls
 * (explicit-noise solves only: MPPI_ERR_STATE otherwise); each may be NULL. */
int mppi_debug_capture_iterations(mppi_handle *h, int on);
int mppi_debug_get_iterations(mppi_handle *h, float *U_raw, float *costs, float *V);

/* Test / tooling hook (not part of the drop-in surface): the kernel forms the library's selection table knows for this
 * handle's model, as names for mppi_set_rollout_variant, the table's order; returns how many were written (<= max_n).
 * tests/test_form_selection_gpu.py times them against the automatic choice. */
int mppi_debug_form_candidates(const mppi_handle *h, const char **names, int max_n);

/* Test / tooling hook (not part of the drop-in surface): mppi_control_ticks on one handle in the row form enqueues every
 * solve but the first one tick AHEAD, gated on a word the host writes once it holds the previous result (csrc/abi_solve.hip:
 * chained control ticks; the launch call and the dispatch leave the step's critical path); on = 0 launches every solve when
 * its turn comes.  The results are bit for bit the same. */
int mppi_debug_set_chained_ticks(mppi_handle *h, int on);

/* Test / tooling hook.  In solves of more than 8192 rollouts (the rollout forms of csrc/rollout_multi.hip with the one-launch
 * tail kernel of K > 4096) the rollout kernel leaves beta = min_k costs[k]
 * (mppi_controller.cu:630-634, computeNormalizer's baseline: exact, order-free) behind as a tagged atomic minimum and the tail
 * kernel reads it instead of reducing the costs and handing the result round (csrc/mppi_device.hpp: publish_min_cost); a rollout
 * form that does not publish, or costs without one finite value, make the tail kernel reduce them itself.  Same bits either way.
 * on = 0 / 1 switches the publication off / on (default on; on < 0 leaves it); *from_rollout (may be NULL) receives whether the
 * LAST solve's tail kernel took beta from the rollout kernel (0 where the rollout form does not publish: by default every solve
 * of <= 8192 rollouts). */
int mppi_debug_min_cost(mppi_handle *h, int on, int *from_rollout);

/* How long a blocking call (mppi_compute_control, mppi_synchronize, mppi_get_results ...) polls for a solve's result block
 * before it gives up with MPPI_ERR_HIP; default 30 s.  (The reference blocks in cudaStreamSynchronize without a limit,
 * PI/mppi_controller.cu:667; at 50 Hz a controller may want a limit of a few periods.)  The clock is read every 256 polls,
 * so the limit holds to microseconds.  After a timeout the lost solve is never waited for again: the control sequence and
 * history the host holds (what mppi_get_control_seq returns: the state before the lost solve) are what the next solve
 * starts from; while the lost solve's device work is still running every solve entry and result getter returns
 * MPPI_ERR_HIP at once (one hipStreamQuery, no blocking), and the handle works again once that work has drained.
 * mppi_destroy synchronises the handle's streams: it may block for as long as that work runs. */
int mppi_set_wait_timeout(mppi_handle *h, double seconds);

#ifdef __cplusplus
}
#endif
#endif /* MPPI_HIP_H_ */
