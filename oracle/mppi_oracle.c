/*
 * mppi_oracle.c -- CPU restatement of the reference MPPI hot path (see mppi_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY: checker for tests/, smoke() and bench.py's
 * cpu_baseline leg.  Never linked into or called from the product library.
 *
 * Parity: "unpinned" for rollout bookkeeping and the cost terms (the reference has no
 * fixtures and its CUDA cannot be built here); NN dynamics pinned by
 * tests/golden/nn_dynamics_golden.npz; MRG32k3a pinned by published jump matrices; the
 * smoothing filter, the softmax weighting and the noise distribution by scipy's statements
 * of the published algorithms (tests/test_oracle_golden.py).
 *
 * Build: gcc -O2 -mavx2 -mfma -ffp-contract=off -fopenmp (oracle/Makefile).
 * -ffp-contract=off is REQUIRED: every fused multiply-add below is an explicit
 * fmaf() placed where nvcc's default contraction would put one (fma_mode=1), and
 * everything else must round exactly as written.
 *
 * Citations are relative to /root/reference/autorally_control/ ;
 *   PI/ = include/autorally_control/path_integral/
 */
#include "mppi_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_MAX_WIDTH 512

static inline float mac(float a, float b, float c, int fma_mode)
{
  if (fma_mode) return fmaf(a, b, c);
  float p = a * b;
  return p + c;
}

void orc_set_num_threads(int n)
{
#ifdef _OPENMP
  omp_set_num_threads(n > 0 ? n : 1);
#else
  (void)n;
#endif
}

int orc_num_params(const int *layers, int n_layers)
{
  /* PI/meta_math.h:38-51 param_counter */
  int n = 0;
  for (int i = 0; i + 1 < n_layers; i++) n += (layers[i] + 1) * layers[i + 1];
  return n;
}

/* fma_mode 2 ("tree"): the OUTPUT layer's dot product in the summation order of the row-tree rollout kernels
 * (autorally_amd/csrc/rollout_row.hip `row_tree` for 32-wide nets, rollout_row64.hip for 64-wide ones), everything else as
 * fma_mode 1.  NOT the reference's order (neural_net_model.cu:379-394 sums k ascending): the kernel forms that use it are
 * held against the nominal (fma_mode 1) oracle at the north-star tolerance and against THIS mode bit for bit.  The nin = 2 n
 * inputs (n = 16 or 32) are dealt to n lanes, lane p owning inputs 2p, 2p+1; a lane's partial is a product and a fused
 * multiply-add, and the n partials are summed by a fixed butterfly:
 *   n = 32 only: L0[p] = P[p] + P[p^16];
 *   L1[p] = L0[p] + L0[p^8];  L2[p] = L1[p] + L1[p^7];  L3[p] = L2[p] + L2[p^1];  sum = L3[0] + L3[2]
 * (IEEE addition is commutative, so every lane of the kernel that ends up with this output holds these bits). */
static float out_tree_dot(const float *w, const float *a, int nin)
{
  const int n = nin / 2;
  float P[32], L1[32], L2[32], L3[32];
  for (int p = 0; p < n; p++) P[p] = fmaf(w[2 * p + 1], a[2 * p + 1], w[2 * p] * a[2 * p]);
  if (n == 32) {
    float L0[32];
    for (int p = 0; p < 32; p++) L0[p] = P[p] + P[p ^ 16];
    for (int p = 0; p < 32; p++) P[p] = L0[p];
  }
  for (int p = 0; p < n; p++) L1[p] = P[p] + P[p ^ 8];
  for (int p = 0; p < n; p++) L2[p] = L1[p] + L1[p ^ 7];
  for (int p = 0; p < n; p++) L3[p] = L2[p] + L2[p ^ 1];
  return L3[0] + L3[2];
}

/* fma_mode 3: the OUTPUT layer in the order of the 4x4x1-MFMA kernel (autorally_amd/csrc/rollout_m44.hip: m44_out_tree),
 * 64 inputs only: 16 blocks of 4 consecutive inputs, a block's partial = a product and three fused multiply-adds ascending,
 * the blocks summed by the butterfly  L1[b] = P[b] + P[b^8];  L2[b] = L1[b] + L1[b^4];  L3[b] = L2[b] + L2[b^2];
 * sum = L3[0] + L3[1].  Like mode 2: everything else as fma_mode 1, and NOT the reference's order. */
static float out_tree_dot_m44(const float *w, const float *a)
{
  float P[16], L1[16], L2[16], L3[16];
  for (int b = 0; b < 16; b++) {
    float z = w[4 * b] * a[4 * b];
    for (int s = 1; s < 4; s++) z = fmaf(w[4 * b + s], a[4 * b + s], z);
    P[b] = z;
  }
  for (int b = 0; b < 16; b++) L1[b] = P[b] + P[b ^ 8];
  for (int b = 0; b < 16; b++) L2[b] = L1[b] + L1[b ^ 4];
  for (int b = 0; b < 16; b++) L3[b] = L2[b] + L2[b ^ 2];
  return L3[0] + L3[1];
}

/* fma_mode 4: the OUTPUT layer in the order of the multi4-tree kernel (autorally_amd/csrc/mfma_net.hpp: nn_last_tree): four
 * interleaved chains, chain g over the inputs 4 s + g, s ascending (a product, then fused multiply-adds), summed as
 * (P[0] + P[2]) + (P[1] + P[3]).  Like modes 2, 3: everything else as fma_mode 1, and NOT the reference's order. */
static float out_tree_dot_multi(const float *w, const float *a, int nin)
{
  float P[4];
  for (int g = 0; g < 4; g++) {
    float z = w[g] * a[g];
    for (int s = 1; s < nin / 4; s++) z = fmaf(w[4 * s + g], a[4 * s + g], z);
    P[g] = z;
  }
  return (P[0] + P[2]) + (P[1] + P[3]);
}

/* fma_mode 5: the 4x4x1-MFMA kernel's automatic form (rollout_m44.hip, SPLIT): as mode 3, and every 64-input HIDDEN layer as two
 * accumulation chains -- even k and odd k, each a fused multiply-add chain from 0, k ascending -- added at the end (even + odd).
 * Like modes 2-4 NOT the reference's order: held against the nominal mode at the north-star tolerance. */
static float hidden_split2_dot(const float *w, const float *a, int nin)
{
  float e = 0.0f, o = 0.0f;
  for (int k = 0; k < nin; k += 2) {
    e = fmaf(w[k], a[k], e);
    o = fmaf(w[k + 1], a[k + 1], o);
  }
  return e + o;
}

/* PI/neural_net_model.cu:357-410.  k ascending, bias added after the dot product (:389-394). */
void orc_nn_forward(const float *theta, const int *layers, int n_layers, const float *in,
                    float *out, int fma_mode)
{
  float buf0[ORC_MAX_WIDTH], buf1[ORC_MAX_WIDTH];
  float *cur = buf0, *nxt = buf1;
  for (int i = 0; i < layers[0]; i++) cur[i] = in[i];
  int off = 0;
  for (int l = 0; l + 1 < n_layers; l++) {
    const int nin = layers[l], nout = layers[l + 1];
    const float *W = theta + off;            /* stride_idcs_[2l]   (:123) */
    const float *b = theta + off + nout * nin; /* stride_idcs_[2l+1] (:131) */
    for (int j = 0; j < nout; j++) {
      float tmp = 0.0f;
      if (fma_mode == 2 && l == n_layers - 2 && (nin == 32 || nin == 64)) tmp = out_tree_dot(W + j * nin, cur, nin);
      else if ((fma_mode == 3 || fma_mode == 5) && l == n_layers - 2 && nin == 64) tmp = out_tree_dot_m44(W + j * nin, cur);
      else if (fma_mode == 5 && l >= 1 && l < n_layers - 2 && nin == 64) tmp = hidden_split2_dot(W + j * nin, cur, nin);
      else if (fma_mode == 4 && l == n_layers - 2 && nin % 4 == 0) tmp = out_tree_dot_multi(W + j * nin, cur, nin);
      else for (int k = 0; k < nin; k++) tmp = mac(W[j * nin + k], cur[k], tmp, fma_mode);
      tmp += b[j];
      if (l < n_layers - 2) tmp = tanhf(tmp); /* MPPI_NNET_NONLINEARITY, :35 */
      nxt[j] = tmp;
    }
    off += nout * nin + nout;
    float *t = cur; cur = nxt; nxt = t;
  }
  for (int i = 0; i < layers[n_layers - 1]; i++) out[i] = cur[i];
}

/* computeKinematics (:346-355) + computeDynamics (:357-410) */
/* CarBasisFuncs::basisFuncX, car_bfs.cuh:44-120.  Literals like 10.0 / .45 are double in the
 * source, so those sub-expressions are evaluated in double and rounded on assignment to the float
 * phi, exactly as a C compiler does; 1960000 / 2744000000 are integer literals (float division). */
/* Test knob: 1 = powf(x, 2|3) evaluated as the correctly rounded products x*x, (x*x)*x instead of the C
 * library's powf -- an ulp-level restatement of 7 of the 25 functions, used by tests/test_ddp.py to show how
 * far such a change moves the feedback gains through the fp32 numerical Jacobian (the product's host replays
 * follow the source and call powf; its device kernel takes the products). */
static int g_bf_pow_products = 0;
void orc_set_bf_pow_products(int on) { g_bf_pow_products = on; }
static inline float orc_powf(float x, int n)
{
  if (!g_bf_pow_products) return powf(x, n);
  return n == 2 ? x * x : (x * x) * x;
}

float orc_basis_func(int idx, const float *s, const float *u)
{
  float phi = 0;
  switch (idx) {
    case 0: phi = u[1]; break;
    case 1: phi = s[4] / 10.0; break;
    case 2: phi = (s[4] > .1) ? sinf(u[0]) * tanf(atanf(s[5] / s[4] + .45 * s[6] / s[4]) - u[0]) / 1200.0
                              : sinf(u[0]) * tanf(-u[0]) / 1200.0; break;
    case 3: phi = (s[4] > .1) ? sinf(u[0]) * tanf(atanf(s[5] / s[4] + .45 * s[6] / s[4]) - u[0]) *
                                    fabsf(tanf(atanf(s[5] / s[4] + .45 * s[6] / s[4]) - u[0])) / 1440000.0
                              : sinf(u[0]) * tanf(-u[0]) * fabsf(tanf(-u[0])) / 1440000.0; break;
    case 4: phi = (s[4] > .1) ? sinf(u[0]) * orc_powf(tanf(atanf(s[5] / s[4] + .45 * s[6] / s[4]) - u[0]), 3) / 1728000000.0
                              : sinf(u[0]) * orc_powf(tanf(-u[0]), 3) / 1728000000.0; break;
    case 5: phi = s[6] * s[5] / 25.0; break;
    case 6: phi = s[6] / 10.0; break;
    case 7: phi = s[5] / 10.0; break;
    case 8: phi = sinf(u[0]); break;
    case 9: phi = (s[4] > .1) ? s[5] / s[4] / 40.0 : 0; break;
    case 10: phi = (s[4] > .1) ? tanf(atanf(s[5] / s[4] + .45 * s[6] / s[4]) - u[0]) / 1400.0
                               : tanf(-u[0]) / 1400.0; break;
    case 11: phi = (s[4] > .1) ? tanf(atanf(s[5] / s[4] + .45 * s[6] / s[4]) - u[0]) *
                                     fabsf(tanf(atanf(s[5] / s[4] + .45 * s[6] / s[4]) - u[0])) / 1960000
                               : tanf(-u[0]) * fabsf(tanf(-u[0])) / 1960000; break;
    case 12: phi = (s[4] > .1) ? orc_powf(tanf(atanf(s[5] / s[4] + .45 * s[6] / s[4]) - u[0]), 3) / 2744000000
                               : orc_powf(tanf(-u[0]), 3) / 2744000000; break;
    case 13: phi = (s[4] > .1) ? (s[5] / s[4] - .35 * s[6] / s[4]) / 40.0 : 0; break;
    case 14: phi = (s[4] > .1) ? (s[5] / s[4] - .35 * s[6] / s[4]) * fabs(s[5] / s[4] - .35 * s[6] / s[4]) / 1600.0 : 0; break;
    case 15: phi = (s[4] > .1) ? orc_powf(s[5] / s[4] - .35 * s[6] / s[4], 3) / 64000.0 : 0; break;
    case 16: phi = s[6] * s[4] / 50.0; break;
    case 17: phi = s[3]; break;
    case 18: phi = s[3] * s[6]; break;
    case 19: phi = s[3] * s[4] / 3.0; break;
    case 20: phi = s[3] * s[4] * s[6] / 5.0; break;
    case 21: phi = orc_powf(s[4], 2) / 100.0; break;
    case 22: phi = orc_powf(s[4], 3) / 1000.0; break;
    case 23: phi = orc_powf(u[1], 2); break;
    case 24: phi = orc_powf(u[1], 3); break;
  }
  return phi;
}

/* GeneralizedLinear device computeStateDeriv (generalized_linear.cu:196-245): kinematics with the
 * yaw rate always negated (:216), then s_der[3+j] = sum_i W[j][i] phi_i.  The reference adds the
 * partial sums of its BLOCKSIZE_Y = 4 (path_integral_main.cu:73) y-threads with atomicAdd, i.e. in
 * no fixed order; this restatement takes the execution in which they land in thread order:
 * p_y = sum over i = y, y+4, ... (FMA-contracted like nvcc's `+=`), s_der = ((0+p_0)+p_1)+p_2)+p_3. */
static void bf_state_deriv(const orc_problem *p, const float *s, const float *u, float *sd)
{
  const float c = cosf(s[2]), sn = sinf(s[2]);
  if (p->fma_mode) {
    sd[0] = fmaf(c, s[4], -(sn * s[5]));
    sd[1] = fmaf(sn, s[4], c * s[5]);
  } else {
    sd[0] = c * s[4] - sn * s[5];
    sd[1] = sn * s[4] + c * s[5];
  }
  sd[2] = -s[6];
  float phi[ORC_NUM_BFS];
  for (int i = 0; i < ORC_NUM_BFS; i++) phi[i] = orc_basis_func(i, s, u);
  for (int j = 0; j < 4; j++) {
    float acc = 0.0f;
    for (int y = 0; y < 4; y++) {
      float part = 0.0f;
      for (int i = y; i < ORC_NUM_BFS; i += 4) part = mac(p->bf_W[j * ORC_NUM_BFS + i], phi[i], part, p->fma_mode);
      acc += part;
    }
    sd[3 + j] = acc;
  }
}

void orc_state_deriv(const orc_problem *p, const float *s, const float *u, float *sd)
{
  if (p->bf_W) {
    bf_state_deriv(p, s, u, sd);
    return;
  }
  const float c = cosf(s[2]), sn = sinf(s[2]);
  if (p->fma_mode) {
    sd[0] = fmaf(c, s[4], -(sn * s[5]));
    sd[1] = fmaf(sn, s[4], c * s[5]);
  } else {
    float a = c * s[4], b = sn * s[5];
    sd[0] = a - b;
    a = sn * s[4]; b = c * s[5];
    sd[1] = a + b;
  }
  sd[2] = p->negate_yaw_der ? -s[6] : s[6];
  float in[6] = { s[3], s[4], s[5], s[6], u[0], u[1] };
  float out[ORC_MAX_WIDTH];
  orc_nn_forward(p->theta, p->layers, p->n_layers, in, out, p->fma_mode);
  for (int i = 0; i < 4; i++) sd[3 + i] = out[i];
}

/* enforceConstraints, neural_net_model.cu:311-323 */
static inline void clamp_controls(const orc_problem *p, float *u)
{
  for (int i = 0; i < ORC_CONTROL_DIM; i++) {
    if (u[i] < p->u_lo[i]) u[i] = p->u_lo[i];
    else if (u[i] > p->u_hi[i]) u[i] = p->u_hi[i];
  }
}

/* incrementState, neural_net_model.cu:334-344 */
static inline void increment_state(const orc_problem *p, float *s, const float *sd)
{
  for (int i = 0; i < ORC_STATE_DIM; i++) s[i] = mac(sd[i], p->dt, s[i], p->fma_mode);
}

void orc_update_state(const orc_problem *p, float *s, float *u)
{
  float sd[ORC_STATE_DIM];
  clamp_controls(p, u);
  orc_state_deriv(p, s, u, sd);
  increment_state(p, s, sd);
}

/* point-sampled, clamped, normalised-coordinate float4 texture: costs.cu:128-154, 373, 377 */
static inline float texel_x(const orc_problem *p, float x, float y)
{
  const orc_cost_params *P = &p->P;
  /* coorTransform, costs.cu:351-357 */
  float u, v, w;
  if (p->fma_mode) {
    u = fmaf(P->r_c1[0], x, P->r_c2[0] * y) + P->trs[0];
    v = fmaf(P->r_c1[1], x, P->r_c2[1] * y) + P->trs[1];
    w = fmaf(P->r_c1[2], x, P->r_c2[2] * y) + P->trs[2];
  } else {
    float a, b;
    a = P->r_c1[0] * x; b = P->r_c2[0] * y; u = (a + b) + P->trs[0];
    a = P->r_c1[1] * x; b = P->r_c2[1] * y; v = (a + b) + P->trs[1];
    a = P->r_c1[2] * x; b = P->r_c2[2] * y; w = (a + b) + P->trs[2];
  }
  const float un = u / w, vn = v / w;
  float fi = floorf(un * (float)p->map_w);
  float fj = floorf(vn * (float)p->map_h);
  if (!(fi >= 0.0f)) fi = 0.0f; /* also catches NaN */
  if (!(fj >= 0.0f)) fj = 0.0f;
  if (fi > (float)(p->map_w - 1)) fi = (float)(p->map_w - 1);
  if (fj > (float)(p->map_h - 1)) fj = (float)(p->map_h - 1);
  const int i = (int)fi, j = (int)fj;
  return p->map_rgba[4 * ((size_t)j * (size_t)p->map_w + (size_t)i)];
}

/* debugCostKernel + launchDebugCostKernel, PI/debug_kernels.cuh:39-88 (MPPICosts::getDebugDisplay,
 * costs.cu:272-285): raster of the costmap around (x, y), width_m x height_m metres at ppm pixels per
 * metre, with a car marker.  out[(H - (yi + 1)) * W + xi]; pixels the kernel never writes (row yi = 0,
 * and flat index 0) keep the value the caller put there. */
void orc_debug_cost_raster(const orc_problem *p, float x, float y, float heading, int width_m,
                           int height_m, int ppm, float *out)
{
  const int W = width_m * ppm, H = height_m * ppm;
  for (int y_idx = 0; y_idx < H; y_idx++)
    for (int x_idx = 0; x_idx < W; x_idx++) {
      float x_pos = x_idx / (1.0 * ppm);
      float y_pos = y_idx / (1.0 * ppm);
      x_pos -= width_m / 2.0;
      y_pos -= height_m / 2.0;
      x_pos += x;
      y_pos += y;
      float cost = texel_x(p, x_pos, y_pos);
      if (x_idx < width_m * ppm && (height_m * ppm - y_idx) < height_m * ppm) {
        const float x_transformed = cosf(heading) * (x_pos - x) + sinf(heading) * (y_pos - y);
        const float y_transformed = -sinf(heading) * (x_pos - x) + cosf(heading) * (y_pos - y);
        const float dist = 0.25 * fabsf(x_transformed) + fabsf(y_transformed);
        if (dist < .15 && x_transformed > 0) {
          if (dist < .1 && x_transformed > 0.05) cost = 1;
          else cost = 0;
        }
        const int idx = (height_m * ppm - (y_idx + 1)) * (width_m * ppm) + x_idx;
        if ((idx > 0) && (idx < (width_m * ppm) * (height_m * ppm))) out[idx] = cost;
      }
    }
}

/* costs.cu:396-409 and the helpers it calls (:307-393) */
float orc_compute_cost(const orc_problem *p, const float *s, const float *u, const float *du,
                       int *crash)
{
  const orc_cost_params *P = &p->P;
  /* getControlCost :307-313 (u clamped, du unclamped) */
  float control_cost = 0.0f;
  control_cost += P->steering_coeff * du[0] * (u[0] - du[0]) / (p->nu[0] * p->nu[0]);
  control_cost += P->throttle_coeff * du[1] * (u[1] - du[1]) / (p->nu[1] * p->nu[1]);

  /* getTrackCost :359-393.  __cosf/__sinf (Q7) have no CPU equivalent; cosf/sinf used. */
  const float cpsi = cosf(s[2]), spsi = sinf(s[2]);
  const float x_front = mac(0.5f, cpsi, s[0], p->fma_mode);
  const float y_front = mac(0.5f, spsi, s[1], p->fma_mode);
  const float x_back = mac(-0.5f, cpsi, s[0], p->fma_mode);
  const float y_back = mac(-0.5f, spsi, s[1], p->fma_mode);
  const float tf = texel_x(p, x_front, y_front);
  const float tb = texel_x(p, x_back, y_back);
  float track_cost = (float)((double)(fabsf(tf) + fabsf(tb)) / 2.0);
  if (fabsf(track_cost) < P->track_slop) track_cost = 0.0f;
  else track_cost = P->track_coeff * track_cost;
  if (tf >= P->boundary_threshold || tb >= P->boundary_threshold) crash[0] = 1;

  /* getSpeedCost :315-326 */
  const float err = s[4] - P->desired_speed;
  const float speed_cost = P->speed_coeff * (P->l1_cost ? fabsf(err) : err * err);

  /* (1.0 - params_.discount)*getCrashCost :402, :328-335 (double, Q6) */
  const float crash_raw = (crash[0] > 0) ? P->crash_coeff : 0.0f;
  const float crash_cost = (float)((1.0 - (double)P->discount) * (double)crash_raw);

  /* getStabilizingCost :337-349 */
  float stabilizing_cost = 0.0f;
  if ((double)fabsf(s[4]) > 0.001) {
    const float slip = -atanf(s[5] / fabsf(s[4]));
    stabilizing_cost = P->slip_penalty * (slip * slip); /* powf(slip,2) */
    if (fabsf(slip) > P->max_slip_ang) stabilizing_cost += P->crash_coeff;
  }
  float cost = control_cost + speed_cost + crash_cost + track_cost + stabilizing_cost;
  if ((double)cost > 1e12 || isnan(cost)) cost = (float)1e12;
  return cost;
}

static void rollout_one(const orc_problem *p, const float *state, const float *U, float *epsV,
                        int k, float *cost_out, int *crash_out)
{
  const int T = p->T;
  float s[ORC_STATE_DIM], sd[ORC_STATE_DIM], u[2], du[2];
  int crash = 0;
  float running_cost = 0.0f;
  for (int i = 0; i < ORC_STATE_DIM; i++) s[i] = state[i];
  const int pure_noise = ((double)k >= .99 * (double)p->K); /* mppi_controller.cu:141 */
  for (int t = 0; t < T; t++) {
    for (int j = 0; j < 2; j++) {
      const size_t idx = (size_t)2 * T * k + 2 * t + j; /* :133 */
      if (k == 0 || t < p->opt_delay) {
        du[j] = 0.0f;
        u[j] = U[2 * t + j];
      } else if (pure_noise) {
        du[j] = epsV[idx] * p->nu[j];
        u[j] = du[j];
      } else {
        du[j] = epsV[idx] * p->nu[j];
        u[j] = U[2 * t + j] + du[j];
      }
      epsV[idx] = u[j]; /* stored BEFORE the clamp (Q3), :153 */
    }
    clamp_controls(p, u);
    if (t > 0) { /* crash[0] > -1 is always true (Q5) */
      const float c = orc_compute_cost(p, s, u, du, &crash);
      running_cost = (float)((double)running_cost + (double)(c - running_cost) / (1.0 * t));
    }
    orc_state_deriv(p, s, u, sd);
    increment_state(p, s, sd);
    if ((double)fabsf(s[3]) > 1.57) crash = 1; /* getCrash, costs.cu:301-305 */
  }
  *cost_out = running_cost + 0.0f; /* terminalCost = 0, costs.cu:411-414 */
  if (crash_out) *crash_out = crash;
}

void orc_rollouts(const orc_problem *p, const float *state, const float *U, float *epsV,
                  float *costs, int *crash_out)
{
  const int K = p->K;
#ifdef _OPENMP
  const int nt = p->nthreads > 1 ? p->nthreads : 1;
#pragma omp parallel for schedule(static) num_threads(nt) if (nt > 1)
#endif
  for (int k = 0; k < K; k++)
    rollout_one(p, state, U, epsV, k, &costs[k], crash_out ? &crash_out[k] : NULL);
}

void orc_weights(const float *costs, int K, float gamma, float *w, float *baseline_out,
                 float *eta_out, float *traj_cost_out)
{
  /* mppi_controller.cu:627-632 */
  float baseline = costs[0];
  for (int i = 0; i < K; i++)
    if (costs[i] < baseline) baseline = costs[i];
  /* normExpKernel :193-203 */
  for (int i = 0; i < K; i++) {
    const float cost2go = costs[i] - baseline;
    w[i] = expf(-gamma * cost2go);
  }
  /* :641-652, sequential fp32 sums (Q8) */
  float eta = 0.0f;
  for (int i = 0; i < K; i++) eta += w[i];
  float tc = 0.0f;
  for (int i = 0; i < K; i++) tc += w[i] * w[i] / eta;
  if (baseline_out) *baseline_out = baseline;
  if (eta_out) *eta_out = eta;
  if (traj_cost_out) *traj_cost_out = tc;
}

void orc_weighted_reduction(const float *w, float eta, const float *V, int K, int T, float *Unew,
                            int fma_mode)
{
  /* mppi_controller.cu:219-267: block t, thread m sums rollouts 64m..64m+63 in order,
   * thread 0 sums the partials in order. */
  const int nthr = (K - 1) / 64 + 1;
  /* weight = states_d[k]/normalizer is recomputed per (t,k) in the kernel (:244); the value is
   * the same every time, so it is hoisted here. */
  float *wn = (float *)malloc(sizeof(float) * (size_t)K);
  for (int k = 0; k < K; k++) wn[k] = w[k] / eta;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) if (K * T > 100000)
#endif
  for (int t = 0; t < T; t++) {
    float u0 = 0.0f, u1 = 0.0f;
    for (int m = 0; m < nthr; m++) {
      float a0 = 0.0f, a1 = 0.0f;
      for (int i = 0; i < 64; i++) {
        const int k = 64 * m + i;
        if (k < K) {
          a0 = mac(wn[k], V[(size_t)k * 2 * T + 2 * t + 0], a0, fma_mode);
          a1 = mac(wn[k], V[(size_t)k * 2 * T + 2 * t + 1], a1, fma_mode);
        }
      }
      u0 += a0; /* thread 0 adds the partials in order, :256-260 */
      u1 += a1;
    }
    Unew[2 * t] = u0;
    Unew[2 * t + 1] = u1;
  }
  free(wn);
}

void orc_savgol(float *U, const float *hist, int T)
{
  /* mppi_controller.cu:468-499 */
  const float coef[5] = { -3.0f, 12.0f, 17.0f, 12.0f, -3.0f };
  float f[5];
  for (int m = 0; m < 5; m++) f[m] = coef[m] / 35.0f;
  float *X = (float *)malloc(sizeof(float) * 2 * (size_t)(T + 4));
  for (int i = 0; i < T + 4; i++)
    for (int j = 0; j < 2; j++) {
      if (i < 2) X[2 * i + j] = hist[2 * i + j];
      else if (i < T + 2) X[2 * i + j] = U[2 * (i - 2) + j];
      else X[2 * i + j] = U[2 * (T - 1) + j];
    }
  for (int i = 0; i < T; i++)
    for (int j = 0; j < 2; j++) {
      float acc = f[0] * X[2 * i + j];
      for (int m = 1; m < 5; m++) {
        const float prod = f[m] * X[2 * (i + m) + j];
        acc = acc + prod;
      }
      U[2 * i + j] = acc;
    }
  free(X);
}

void orc_nominal_traj(const orc_problem *p, const float *state, const float *U, float *state_seq,
                      float *control_seq)
{
  /* mppi_controller.cu:501-519 */
  float s[ORC_STATE_DIM], u[2];
  for (int j = 0; j < ORC_STATE_DIM; j++) s[j] = state[j];
  for (int i = 0; i < p->T; i++) {
    for (int j = 0; j < ORC_STATE_DIM; j++) state_seq[i * ORC_STATE_DIM + j] = s[j];
    u[0] = U[2 * i];
    u[1] = U[2 * i + 1];
    orc_update_state(p, s, u);
    control_seq[2 * i] = u[0];
    control_seq[2 * i + 1] = u[1];
  }
}

void orc_slide_control_seq(float *U, float *hist, const float *init_u, int T, int stride)
{
  /* mppi_controller.cu:527-554 (flat-index quirk Q15 kept) */
  if (stride == 1) {
    hist[0] = hist[2];
    hist[1] = hist[3];
    hist[2] = U[0];
    hist[3] = U[1];
  } else {
    const int t = stride - 2;
    for (int i = 0; i < 4; i++) hist[i] = U[t + i];
  }
  for (int i = 0; i < T - stride; i++)
    for (int j = 0; j < 2; j++) U[i * 2 + j] = U[(i + stride) * 2 + j];
  for (int j = 1; j <= stride; j++)
    for (int i = 0; i < 2; i++) U[(T - j) * 2 + i] = init_u[i];
}

void orc_compute_control(const orc_problem *p, int num_iters, const float *state, float *U,
                         const float *hist, float *eps, float *traj_cost, float *costs_out,
                         float *w_out)
{
  /* mppi_controller.cu:600-675 */
  const int K = p->K, T = p->T;
  float *costs = (float *)malloc(sizeof(float) * (size_t)K);
  float *w = (float *)malloc(sizeof(float) * (size_t)K);
  float eta = 0.0f, tc = 0.0f;
  for (int it = 0; it < num_iters; it++) {
    float *V = eps + (size_t)it * K * T * 2;
    orc_rollouts(p, state, U, V, costs, NULL);
    orc_weights(costs, K, p->gamma, w, NULL, &eta, &tc);
    orc_weighted_reduction(w, eta, V, K, T, U, p->fma_mode);
  }
  orc_savgol(U, hist, T);
  if (traj_cost) *traj_cost = tc;
  if (costs_out) memcpy(costs_out, costs, sizeof(float) * (size_t)K);
  if (w_out) memcpy(w_out, w, sizeof(float) * (size_t)K);
  free(costs);
  free(w);
}

/* ======================= noise generator spec =======================
 * MRG32k3a (L'Ecuyer 1999, "Good parameters and implementations for combined
 * multiple recursive random number generators"), one subsequence of 2^76 draws
 * per rollout (the spacing cuRAND's MRG32k3a uses), draws turned into N(0,1)
 * pairs with a Box-Muller transform written ONLY in IEEE basic operations so
 * that the HIP generator reproduces it bit for bit.  Reference call sites being
 * replaced: mppi_controller.cu:330-331, 612 (cuRAND XORWOW, not reproducible).
 */
#define M1 4294967087ULL
#define M2 4294944443ULL
#define A12 1403580ULL
#define A13N 810728ULL
#define A21 527612ULL
#define A23N 1370589ULL

static inline uint32_t mulmod(uint64_t a, uint64_t b, uint64_t m) { return (uint32_t)((a * b) % m); }

void orc_mrg_seed(orc_mrg_state *st, uint64_t seed)
{
  for (int i = 0; i < 3; i++) { st->s1[i] = 12345u; st->s2[i] = 12345u; }
  if (seed != 0) {
    const uint32_t x1 = ((uint32_t)seed) ^ 0x55555555u;
    const uint32_t x2 = (uint32_t)((seed >> 32) ^ 0xAAAAAAAAu);
    st->s1[0] = mulmod(x1, st->s1[0], M1);
    st->s1[1] = mulmod(x2, st->s1[1], M1);
    st->s1[2] = mulmod(x1, st->s1[2], M1);
    st->s2[0] = mulmod(x2, st->s2[0], M2);
    st->s2[1] = mulmod(x1, st->s2[1], M2);
    st->s2[2] = mulmod(x2, st->s2[2], M2);
  }
}

uint32_t orc_mrg_next_z(orc_mrg_state *st)
{
  /* p1 = (a12*s1[1] - a13n*s1[0]) mod m1 ; p2 = (a21*s2[2] - a23n*s2[0]) mod m2 */
  /* each product is < 2^53; reduce before combining so nothing can wrap 2^64 */
  const uint64_t p1 = ((A12 * st->s1[1]) % M1 + M1 - (A13N * st->s1[0]) % M1) % M1;
  st->s1[0] = st->s1[1]; st->s1[1] = st->s1[2]; st->s1[2] = (uint32_t)p1;
  const uint64_t p2 = ((A21 * st->s2[2]) % M2 + M2 - (A23N * st->s2[0]) % M2) % M2;
  st->s2[0] = st->s2[1]; st->s2[1] = st->s2[2]; st->s2[2] = (uint32_t)p2;
  uint64_t z = (p1 >= p2) ? (p1 - p2) : (p1 + M1 - p2);
  if (z == 0) z = M1;
  return (uint32_t)z; /* in [1, m1] */
}

double orc_mrg_next_u01(orc_mrg_state *st)
{
  return (double)orc_mrg_next_z(st) * 2.328306549295727688e-10;
}

static void mat3_mul(const uint32_t *A, const uint32_t *B, uint32_t *C, uint64_t m)
{
  uint32_t R[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      uint64_t acc = 0;
      for (int k = 0; k < 3; k++) acc = (acc + (uint64_t)A[3 * i + k] * B[3 * k + j] % m) % m;
      R[3 * i + j] = (uint32_t)acc;
    }
  memcpy(C, R, sizeof(R));
}

static void mat3_vec(const uint32_t *A, uint32_t *v, uint64_t m)
{
  uint32_t r[3];
  for (int i = 0; i < 3; i++) {
    uint64_t acc = 0;
    for (int k = 0; k < 3; k++) acc = (acc + (uint64_t)A[3 * i + k] * v[k] % m) % m;
    r[i] = (uint32_t)acc;
  }
  memcpy(v, r, sizeof(r));
}

static void base_matrices(uint32_t *A1, uint32_t *A2)
{
  const uint32_t a1[9] = { 0, 1, 0, 0, 0, 1, (uint32_t)(M1 - A13N), (uint32_t)A12, 0 };
  const uint32_t a2[9] = { 0, 1, 0, 0, 0, 1, (uint32_t)(M2 - A23N), 0, (uint32_t)A21 };
  memcpy(A1, a1, sizeof(a1));
  memcpy(A2, a2, sizeof(a2));
}

void orc_mrg_jump_matrices(int e, uint32_t A1[9], uint32_t A2[9])
{
  base_matrices(A1, A2);
  for (int i = 0; i < e; i++) {
    mat3_mul(A1, A1, A1, M1);
    mat3_mul(A2, A2, A2, M2);
  }
}

/* advance by n * 2^e draws */
static void skip_pow2(orc_mrg_state *st, uint64_t n, int e)
{
  uint32_t A1[9], A2[9];
  orc_mrg_jump_matrices(e, A1, A2);
  while (n) {
    if (n & 1) { mat3_vec(A1, st->s1, M1); mat3_vec(A2, st->s2, M2); }
    mat3_mul(A1, A1, A1, M1);
    mat3_mul(A2, A2, A2, M2);
    n >>= 1;
  }
}

void orc_mrg_skip_subsequences(orc_mrg_state *st, uint64_t n) { skip_pow2(st, n, 76); }
void orc_mrg_skip(orc_mrg_state *st, uint64_t n) { skip_pow2(st, n, 0); }

/* log(x) for normal positive x; fdlibm e_logf algorithm, evaluated exactly as written. */
static float spec_logf(float x)
{
  const float ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f;
  const float Lg1 = 0.66666662693f, Lg2 = 0.40000972152f, Lg3 = 0.28498786688f,
              Lg4 = 0.24279078841f;
  uint32_t ix;
  memcpy(&ix, &x, 4);
  ix += 0x3f800000u - 0x3f3504f3u;
  const int k = (int)(ix >> 23) - 0x7f;
  ix = (ix & 0x007fffffu) + 0x3f3504f3u;
  memcpy(&x, &ix, 4);
  const float f = x - 1.0f;
  const float s = f / (2.0f + f);
  const float z = s * s;
  const float w = z * z;
  const float t1 = w * (Lg2 + w * Lg4);
  const float t2 = z * (Lg1 + w * Lg3);
  const float R = t2 + t1;
  const float hfsq = (0.5f * f) * f;
  const float dk = (float)k;
  return (((s * (hfsq + R) + dk * ln2_lo) - hfsq) + f) + dk * ln2_hi;
}

/* sin(2*pi*v), cos(2*pi*v) for v in (0,1]; octant reduction + fdlibm k_sinf/k_cosf polynomials. */
static void spec_sincos2pi(float v, float *sn, float *cs)
{
  const float S1 = -0.16666667163f, S2 = 0.0083333291113f, S3 = -0.00019839334709f,
              S4 = 2.7183114e-06f;
  const float C0 = -0.5f, C1 = 0.041666623205f, C2 = -0.0013886763947f, C3 = 2.4390449e-05f;
  const float x = v - 0.5f;
  const float kq = rintf(x * 4.0f);
  const float y = x - kq * 0.25f;
  const float a = y * 6.2831853071795864769f;
  const float a2 = a * a;
  const float ps = S1 + a2 * (S2 + a2 * (S3 + a2 * S4));
  const float sin_a = a + (a * a2) * ps;
  const float pc = C0 + a2 * (C1 + a2 * (C2 + a2 * C3));
  const float cos_a = 1.0f + a2 * pc;
  const int q = ((int)kq) & 3;
  float s2, c2; /* sin/cos of 2*pi*x */
  switch (q) {
    case 0: s2 = sin_a; c2 = cos_a; break;
    case 1: s2 = cos_a; c2 = -sin_a; break;
    case 2: s2 = -sin_a; c2 = -cos_a; break;
    default: s2 = -cos_a; c2 = sin_a; break;
  }
  *sn = -s2; /* v = x + 1/2 */
  *cs = -c2;
}

void orc_box_muller(float u1, float u2, float *n0, float *n1)
{
  const float r = sqrtf(-2.0f * spec_logf(u1));
  float sn, cs;
  spec_sincos2pi(u2, &sn, &cs);
  *n0 = r * sn;
  *n1 = r * cs;
}

void orc_generate_noise(uint64_t seed, uint64_t offset, int K, int T, float *eps)
{
  orc_mrg_state base;
  orc_mrg_seed(&base, seed);
  uint32_t A1[9], A2[9];
  orc_mrg_jump_matrices(76, A1, A2);
  orc_mrg_state sub = base; /* subsequence k: base advanced by k*2^76 */
  for (int k = 0; k < K; k++) {
    orc_mrg_state st = sub;
    orc_mrg_skip(&st, offset);
    for (int t = 0; t < T; t++) {
      const float u1 = (float)orc_mrg_next_z(&st) * 0x1p-32f;
      const float u2 = (float)orc_mrg_next_z(&st) * 0x1p-32f;
      orc_box_muller(u1, u2, &eps[(size_t)k * 2 * T + 2 * t], &eps[(size_t)k * 2 * T + 2 * t + 1]);
    }
    mat3_vec(A1, sub.s1, M1);
    mat3_vec(A2, sub.s2, M2);
  }
}
