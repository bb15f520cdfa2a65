/*
 * ddp_oracle.c -- CPU restatement of the reference's feedback-gain computation.
 *
 * TEST INFRASTRUCTURE ONLY (see mppi_oracle.h).
 *
 * Parity status: UNPINNED.  The reference holds no test, golden vector or fixture for this path
 * and cannot be built here (Eigen absent), so this file restates the source line by line:
 *   MPPIController::initDDP / computeFeedbackGains   PI/mppi_controller.cu:402-441
 *   DDP::run                                          ddp/ddp.h:49-157
 *   ModelWrapperDDP::f / df                           ddp/ddp_model_wrapper.h:57-79
 *   TrackingCostDDP / TrackingTerminalCost            ddp/ddp_tracking_costs.h:35-52, 98-111
 *   NeuralNetModel::computeKinematics/Dynamics/Grad   PI/neural_net_model.cu:191-264
 *   GeneralizedLinear (host) + numerical Jacobian     PI/generalized_linear.cu:140-175, ddp/ddp_dynamics.h:71-84
 * (paths relative to /root/reference/autorally_control/include/autorally_control/, PI = path_integral).
 * Eigen's internal summation order of the small products is not specified by the source; sums here
 * run in index order.  tests/test_ddp.py additionally checks the gains against a float64 Riccati
 * recursion built from finite-difference Jacobians of the oracle's own dynamics.
 */
#include "mppi_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define NS 7
#define NC 2
#define NZ (NS + NC)

typedef struct {
  const float *bf_W; /* != NULL: GeneralizedLinear basis-function model (W[4][25]); the network fields are unused */
  const float *theta;
  const int *layers;
  int n_layers;
  int woff[ORC_MAX_LAYERS], boff[ORC_MAX_LAYERS];
  float *z[ORC_MAX_LAYERS]; /* weighted_in_[l] */
  int width;
} ddp_net;

static void net_init(ddp_net *n, const float *theta, const int *layers, int n_layers, const float *bf_W)
{
  int off = 0, w = 0;
  n->bf_W = bf_W;
  if (bf_W) {
    n->n_layers = 0;
    return;
  }
  n->theta = theta;
  n->layers = layers;
  n->n_layers = n_layers;
  for (int l = 0; l < n_layers; l++)
    if (layers[l] > w) w = layers[l];
  n->width = w;
  for (int l = 0; l + 1 < n_layers; l++) {
    n->woff[l] = off;
    n->boff[l] = off + layers[l] * layers[l + 1];
    off = n->boff[l] + layers[l + 1];
    n->z[l] = (float *)malloc(sizeof(float) * (size_t)layers[l + 1]);
  }
}
static void net_free(ddp_net *n)
{
  for (int l = 0; l + 1 < n->n_layers; l++) free(n->z[l]);
}

/* computeDynamics (host), neural_net_model.cu:201-230: acts <- phi(W acts + b), tanh on all but the last */
static void net_forward(ddp_net *n, const float *x, const float *u, float *out4)
{
  float *a = (float *)malloc(sizeof(float) * (size_t)n->width);
  float *b = (float *)malloc(sizeof(float) * (size_t)n->width);
  const int L = n->n_layers - 1;
  a[0] = x[3]; a[1] = x[4]; a[2] = x[5]; a[3] = x[6]; a[4] = u[0]; a[5] = u[1];
  for (int l = 0; l < L; l++) {
    const int nin = n->layers[l], nout = n->layers[l + 1];
    for (int j = 0; j < nout; j++) {
      float acc = 0.0f;
      for (int k = 0; k < nin; k++) acc += n->theta[n->woff[l] + j * nin + k] * a[k];
      acc += n->theta[n->boff[l] + j];
      n->z[l][j] = acc;
      b[j] = (l < L - 1) ? tanhf(acc) : acc;
    }
    float *t = a; a = b; b = t;
  }
  for (int i = 0; i < 4; i++) out4[i] = a[i];
  free(a);
  free(b);
}

/* ModelWrapperDDP::f: computeKinematics (:191-199) then computeDynamics */
static void model_f(ddp_net *n, int negate_yaw_der, const float *x, const float *u, float *dx)
{
  dx[0] = cosf(x[2]) * x[4] - sinf(x[2]) * x[5];
  dx[1] = sinf(x[2]) * x[4] + cosf(x[2]) * x[5];
  if (n->bf_W) {
    /* GeneralizedLinear host computeKinematics / computeDynamics, generalized_linear.cu:151-167 */
    float phi[ORC_NUM_BFS];
    dx[2] = -x[6];
    for (int i = 0; i < ORC_NUM_BFS; i++) phi[i] = orc_basis_func(i, x, u);
    for (int j = 0; j < 4; j++) {
      float acc = 0.0f;
      for (int y = 0; y < 4; y++) { /* same summation as the rollout restatement (mppi_oracle.c bf_state_deriv) */
        float part = 0.0f;
        for (int i = y; i < ORC_NUM_BFS; i += 4) part = fmaf(n->bf_W[j * ORC_NUM_BFS + i], phi[i], part);
        acc += part;
      }
      dx[3 + j] = acc;
    }
    return;
  }
  dx[2] = negate_yaw_der ? -x[6] : x[6];
  net_forward(n, x, u, dx + 3);
}

/* Dynamics::df -> Eigen::NumericalDiff<..., Central> (ddp/ddp_dynamics.h:71-84) for a model without
 * computeGrad: h_j = sqrt(FLT_EPSILON) |z_j| (sqrt(FLT_EPSILON) if that is 0), central difference, fp32 */
static void model_jac_numeric(ddp_net *n, int negate_yaw_der, const float *x, const float *u, float *J)
{
  const float eps = sqrtf(1.1920928955078125e-07f);
  float z[NZ], v1[NS], v2[NS];
  memcpy(z, x, sizeof(float) * NS);
  memcpy(z + NS, u, sizeof(float) * NC);
  for (int j = 0; j < NZ; j++) {
    const float zj = z[j];
    float h = eps * fabsf(zj);
    if (h == 0.0f) h = eps;
    z[j] += h;
    model_f(n, negate_yaw_der, z, z + NS, v2);
    z[j] -= 2 * h;
    model_f(n, negate_yaw_der, z, z + NS, v1);
    z[j] = zj;
    for (int i = 0; i < NS; i++) J[i * NZ + j] = (v2[i] - v1[i]) / (2 * h);
  }
}

/* computeGrad, neural_net_model.cu:233-264: J is [NS][NZ] row-major */
static void model_jac(ddp_net *n, const float *x, const float *u, float *J)
{
  float out[4];
  memset(J, 0, sizeof(float) * NS * NZ);
  J[0 * NZ + 2] = -sinf(x[2]) * x[4] - cosf(x[2]) * x[5];
  J[0 * NZ + 4] = cosf(x[2]);
  J[0 * NZ + 5] = -sinf(x[2]);
  J[1 * NZ + 2] = cosf(x[2]) * x[4] - sinf(x[2]) * x[5];
  J[1 * NZ + 4] = sinf(x[2]);
  J[1 * NZ + 5] = cosf(x[2]);
  J[2 * NZ + 6] = -1.0f; /* :241, independent of negate_yaw_der */
  net_forward(n, x, u, out);
  const int L = n->n_layers - 1;
  /* ip_delta_: [rows][4], rows = width of the layer it currently refers to */
  float *d = (float *)calloc((size_t)n->width * 4, sizeof(float));
  float *e = (float *)calloc((size_t)n->width * 4, sizeof(float));
  for (int i = 0; i < 4; i++) d[i * 4 + i] = 1.0f;
  for (int i = L - 1; i > 0; i--) { /* reference: for (i = NUM_LAYERS-2; i > 0; i--) */
    const int nin = n->layers[i], nout = n->layers[i + 1];
    for (int r = 0; r < nin; r++)
      for (int c = 0; c < 4; c++) {
        float acc = 0.0f;
        for (int k = 0; k < nout; k++) acc += n->theta[n->woff[i] + k * nin + r] * d[k * 4 + c];
        e[r * 4 + c] = acc;
      }
    for (int c = 0; c < 4; c++)
      for (int r = 0; r < nin; r++) {
        const float zp = 1.0f - powf(tanhf(n->z[i - 1][r]), 2.0f);
        e[r * 4 + c] = e[r * 4 + c] * zp;
      }
    float *t = d; d = e; e = t;
  }
  {
    const int nin = n->layers[0], nout = n->layers[1];
    for (int r = 0; r < nin; r++)
      for (int c = 0; c < 4; c++) {
        float acc = 0.0f;
        for (int k = 0; k < nout; k++) acc += n->theta[n->woff[0] + k * nin + r] * d[k * 4 + c];
        e[r * 4 + c] = acc;
      }
  }
  /* jac_.bottomRightCorner(4, 6) += ip_delta_^T */
  for (int o = 0; o < 4; o++)
    for (int r = 0; r < 6; r++) J[(3 + o) * NZ + 3 + r] += e[r * 4 + o];
  free(d);
  free(e);
}

static float clampmm(float v, float lo, float hi)
{ /* cwiseMin(u_max).cwiseMax(u_min) */
  if (!(v < hi)) v = hi;
  if (!(v > lo)) v = lo;
  return v;
}

/* symmetric 2x2 solve through a pivoted LDL^T (largest |diagonal| first), as Eigen::LDLT */
static int solve2(const float A[4], const float *b, float *x)
{
  const int p = (fabsf(A[3]) > fabsf(A[0])) ? 1 : 0, q = 1 - p;
  const float d0 = A[p * 2 + p];
  if (d0 == 0.0f || !isfinite(d0)) return 1;
  const float l = A[q * 2 + p] / d0;
  const float d1 = A[q * 2 + q] - l * A[q * 2 + p];
  if (!isfinite(d1) || !isfinite(l)) return 1;
  const float z0 = b[p], z1 = b[q] - l * z0;
  const float w0 = z0 / d0, w1 = (d1 != 0.0f) ? z1 / d1 : 0.0f;
  x[q] = w1;
  x[p] = w0 - l * w1;
  return 0;
}

int orc_ddp_feedback_gains(const float *theta, const int *layers, int n_layers, int T, float dt,
                           const float *u_lo, const float *u_hi, int negate_yaw_der, const float *Q,
                           const float *R, const float *Qf, const float *x0, const float *target_x,
                           const float *target_u, float *feedback, float *feedforward, float *xout,
                           float *uout, float *total_cost, const float *bf_W)
{
  const int H = T;
  int rc = 0;
  ddp_net n;
  net_init(&n, theta, layers, n_layers, bf_W);
  float *x = (float *)calloc((size_t)H * NS, sizeof(float));
  float *u = (float *)malloc(sizeof(float) * (size_t)H * NC);
  float *df = (float *)malloc(sizeof(float) * (size_t)H * NS * NZ);
  float *dL = (float *)malloc(sizeof(float) * (size_t)H * NZ);
  float *cost = (float *)calloc((size_t)H, sizeof(float));
  memcpy(u, target_u, sizeof(float) * (size_t)H * NC);
  memset(feedback, 0, sizeof(float) * (size_t)H * NC * NS);
  memset(feedforward, 0, sizeof(float) * (size_t)H * NC);
  /* ddp.h:55-65 */
  memcpy(x, x0, sizeof(float) * NS);
  for (int i = 1; i < H; i++) {
    float dx[NS];
    if (i < H - 1)
      for (int j = 0; j < NC; j++) u[(i - 1) * NC + j] = clampmm(u[(i - 1) * NC + j], u_lo[j], u_hi[j]);
    model_f(&n, negate_yaw_der, x + (i - 1) * NS, u + (i - 1) * NC, dx);
    for (int s = 0; s < NS; s++) x[i * NS + s] = x[(i - 1) * NS + s] + dx[s] * dt;
  }
  /* ddp.h:71-80 */
  for (int k = 0; k < H; k++) {
    float *J = df + (size_t)k * NS * NZ;
    if (bf_W) model_jac_numeric(&n, negate_yaw_der, x + k * NS, u + k * NC, J);
    else model_jac(&n, x + k * NS, u + k * NC, J);
    for (int i = 0; i < NS * NZ; i++) J[i] = J[i] * dt;
    for (int i = 0; i < NS; i++) J[i * NZ + i] += 1.0f;
    for (int i = 0; i < NS; i++) dL[k * NZ + i] = Q[i] * (x[k * NS + i] - target_x[k * NS + i]);
    for (int j = 0; j < NC; j++) dL[k * NZ + NS + j] = R[j] * (u[k * NC + j] - target_u[k * NC + j]);
  }
  /* ddp.h:83-87 */
  float Vxx[NS * NS], Vx[NS], Vlast = 0.0f;
  memset(Vxx, 0, sizeof(Vxx));
  for (int i = 0; i < NS; i++) {
    const float e = x[(H - 1) * NS + i] - target_x[(H - 1) * NS + i];
    Vxx[i * NS + i] = Qf[i];
    Vx[i] = Qf[i] * e;
    Vlast += e * (Qf[i] * e);
  }
  /* ddp.h:90-123 */
  for (int k = H - 2; k >= 0 && rc == 0; k--) {
    const float *J = df + (size_t)k * NS * NZ;
    float qx[NS], qu[NC], BtV[NC * NS], PtV[NS * NS], qux[NC * NS], qxx[NS * NS], quu[NC * NC];
    for (int i = 0; i < NS; i++) {
      float acc = 0.0f;
      for (int m = 0; m < NS; m++) acc += J[m * NZ + i] * Vx[m]; /* Phi^T Vx */
      qx[i] = dL[k * NZ + i] * dt + acc;
    }
    for (int j = 0; j < NC; j++) {
      float acc = 0.0f;
      for (int m = 0; m < NS; m++) acc += J[m * NZ + NS + j] * Vx[m]; /* B^T Vx */
      qu[j] = dL[k * NZ + NS + j] * dt + acc;
    }
    for (int j = 0; j < NC; j++)
      for (int c = 0; c < NS; c++) {
        float acc = 0.0f;
        for (int m = 0; m < NS; m++) acc += J[m * NZ + NS + j] * Vxx[m * NS + c];
        BtV[j * NS + c] = acc;
      }
    for (int i = 0; i < NS; i++)
      for (int c = 0; c < NS; c++) {
        float acc = 0.0f;
        for (int m = 0; m < NS; m++) acc += J[m * NZ + i] * Vxx[m * NS + c];
        PtV[i * NS + c] = acc;
      }
    for (int j = 0; j < NC; j++)
      for (int c = 0; c < NS; c++) {
        float acc = 0.0f;
        for (int m = 0; m < NS; m++) acc += BtV[j * NS + m] * J[m * NZ + c];
        qux[j * NS + c] = 0.0f * dt + acc;
      }
    for (int i = 0; i < NS; i++)
      for (int c = 0; c < NS; c++) {
        float acc = 0.0f;
        for (int m = 0; m < NS; m++) acc += PtV[i * NS + m] * J[m * NZ + c];
        qxx[i * NS + c] = ((i == c) ? Q[i] : 0.0f) * dt + acc;
      }
    for (int j = 0; j < NC; j++)
      for (int c = 0; c < NC; c++) {
        float acc = 0.0f;
        for (int m = 0; m < NS; m++) acc += BtV[j * NS + m] * J[m * NZ + NS + c];
        quu[j * NC + c] = ((j == c) ? R[j] : 0.0f) * dt + acc;
      }
    float Lk[NC * NS], lk[NC];
    for (int c = 0; c < NS && rc == 0; c++) {
      const float b[2] = {-qux[0 * NS + c], -qux[1 * NS + c]};
      float s[2];
      rc = solve2(quu, b, s);
      Lk[0 * NS + c] = s[0];
      Lk[1 * NS + c] = s[1];
    }
    if (rc == 0) {
      const float b[2] = {-qu[0], -qu[1]};
      rc = solve2(quu, b, lk);
    }
    if (rc) break;
    memcpy(feedback + (size_t)k * NC * NS, Lk, sizeof(Lk));
    memcpy(feedforward + (size_t)k * NC, lk, sizeof(lk));
    float Vn[NS * NS];
    for (int i = 0; i < NS; i++)
      for (int c = 0; c < NS; c++) {
        float acc = 0.0f;
        for (int m = 0; m < NC; m++) acc += qux[m * NS + i] * Lk[m * NS + c]; /* qux^T Lk */
        Vn[i * NS + c] = qxx[i * NS + c] + acc;
      }
    for (int i = 0; i < NS; i++)
      for (int c = 0; c < NS; c++) Vxx[i * NS + c] = 0.5f * (Vn[i * NS + c] + Vn[c * NS + i]);
    for (int i = 0; i < NS; i++) {
      float acc = 0.0f;
      for (int m = 0; m < NC; m++) acc += qux[m * NS + i] * lk[m];
      Vx[i] = qx[i] + acc;
    }
  }
  /* ddp.h:125-152, alpha = 1, first iteration always accepted */
  if (rc == 0) {
    memset(xout, 0, sizeof(float) * (size_t)H * NS);
    memset(uout, 0, sizeof(float) * (size_t)H * NC);
    memcpy(xout, x, sizeof(float) * NS);
    for (int k = 0; k + 1 < H; k++) {
      float dx[NS], un[NC], fx[NS];
      for (int i = 0; i < NS; i++) dx[i] = xout[k * NS + i] - x[k * NS + i];
      for (int j = 0; j < NC; j++) {
        float acc = 0.0f;
        for (int i = 0; i < NS; i++) acc += feedback[((size_t)k * NC + j) * NS + i] * dx[i];
        un[j] = clampmm((u[k * NC + j] + 1.0f * feedforward[k * NC + j]) + acc, u_lo[j], u_hi[j]);
        uout[k * NC + j] = un[j];
      }
      model_f(&n, negate_yaw_der, xout + k * NS, un, fx);
      for (int i = 0; i < NS; i++) xout[(k + 1) * NS + i] = xout[k * NS + i] + fx[i] * dt;
      float sc = 0.0f, cc = 0.0f;
      for (int i = 0; i < NS; i++) {
        const float e = xout[k * NS + i] - target_x[k * NS + i];
        sc += e * (Q[i] * e);
      }
      for (int j = 0; j < NC; j++) {
        const float e = un[j] - target_u[k * NC + j];
        cc += e * (R[j] * e);
      }
      cost[k] = (sc + cc) * dt;
    }
    cost[H - 1] = Vlast;
    float tot = 0.0f;
    for (int k = 0; k < H; k++) tot += cost[k];
    *total_cost = tot;
  }
  free(x); free(u); free(df); free(dL); free(cost);
  net_free(&n);
  return rc;
}
