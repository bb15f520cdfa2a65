// ddp_feedback.cpp -- see ddp_feedback.hpp.  Plain fp32 host arithmetic (the reference is Eigen
// fp32 on the host); matrix products are written as k-ascending sums, association as in the source
// expressions of ddp.h (e.g. (B^T Vxx) Phi).
#include "ddp_feedback.hpp"
#include "basis_funcs.hpp"
#include "tanhf_vec.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <memory>

#include <immintrin.h>

namespace mppi {
namespace {

template <int R, int C>
struct Mat {
  float v[R][C];
  void zero() { std::memset(v, 0, sizeof(v)); }
};

template <int R, int K, int C>
Mat<R, C> mul(const Mat<R, K> &a, const Mat<K, C> &b)
{
  Mat<R, C> o;
  for (int i = 0; i < R; i++)
    for (int j = 0; j < C; j++) {
      float s = 0.0f;
      for (int k = 0; k < K; k++) s += a.v[i][k] * b.v[k][j];
      o.v[i][j] = s;
    }
  return o;
}
// The Riccati step's products on eight lanes: a matrix with up to 8 columns keeps every row in one AVX2 register (padded
// with zeros), and C[i][:] = sum_k A[i][k] * B[k][:] runs k ascending with a separate multiply and add per term (this file
// is compiled with -ffp-contract=off) -- element for element the operations of mul() above in the same order, hence the
// same bits; only seven of them advance per instruction (99 steps x ~2.5 products of 7 x 7 x 7 were 20 % of a pass).
template <int R>
struct Mat8 {
  alignas(32) float v[R][8];
};
template <int R, int C>
Mat8<R> pad8(const Mat<R, C> &a)
{
  static_assert(C <= 8, "one register per row");
  Mat8<R> o;
  for (int i = 0; i < R; i++) {
    for (int j = 0; j < C; j++) o.v[i][j] = a.v[i][j];
    for (int j = C; j < 8; j++) o.v[i][j] = 0.0f;
  }
  return o;
}
template <int R, int C>
Mat<R, C> unpad8(const Mat8<R> &a)
{
  Mat<R, C> o;
  for (int i = 0; i < R; i++)
    for (int j = 0; j < C; j++) o.v[i][j] = a.v[i][j];
  return o;
}
// A [R][K] (plain) times B [K][<= 8] (padded rows)
template <int R, int K>
Mat8<R> mul8(const float (&a)[R][K], const Mat8<K> &b)
{
  Mat8<R> o;
  for (int i = 0; i < R; i++) {
    __m256 s = _mm256_setzero_ps();
    for (int k = 0; k < K; k++) s = _mm256_add_ps(s, _mm256_mul_ps(_mm256_set1_ps(a[i][k]), _mm256_load_ps(b.v[k])));
    _mm256_store_ps(o.v[i], s);
  }
  return o;
}
template <int R, int K>
Mat8<R> mul8(const Mat8<R> &a, const Mat8<K> &b)  // the first K columns of a
{
  Mat8<R> o;
  for (int i = 0; i < R; i++) {
    __m256 s = _mm256_setzero_ps();
    for (int k = 0; k < K; k++) s = _mm256_add_ps(s, _mm256_mul_ps(_mm256_set1_ps(a.v[i][k]), _mm256_load_ps(b.v[k])));
    _mm256_store_ps(o.v[i], s);
  }
  return o;
}

template <int R, int C>
Mat<C, R> transpose(const Mat<R, C> &a)
{
  Mat<C, R> o;
  for (int i = 0; i < R; i++)
    for (int j = 0; j < C; j++) o.v[j][i] = a.v[i][j];
  return o;
}

// ModelWrapperDDP (ddp/ddp_model_wrapper.h:37-80): f and its Jacobian wrt [x | u]
struct HostModel {
  virtual ~HostModel() {}
  virtual void f(const DdpProblem &p, const float *x, const float *u, float *dx) = 0;
  virtual void jacobian(const DdpProblem &p, const float *x, const float *u, Mat<kDdpS, kDdpSC> &J) = 0;
  // the same, called right after f(p, x, u, .) on the same (x, u): a model may reuse that evaluation
  virtual void jacobian_after_f(const DdpProblem &p, const float *x, const float *u, Mat<kDdpS, kDdpSC> &J)
  {
    jacobian(p, x, u, J);
  }
};

// Host network: forward pass keeping the pre-activations (computeDynamics, neural_net_model.cu:201-230).
// Eight output neurons (forward) / eight input neurons (backward) per AVX2 register; every sum keeps its
// k-ascending order and its separate multiply and add (this file is compiled with -ffp-contract=off), so
// the values are those of the plain loops -- only eight of them advance per instruction.
struct HostNet : HostModel {
  const DdpNet &n;
  int L;                                // weight matrices
  std::vector<int> pin, pout;           // layer widths rounded up to 8
  std::vector<std::vector<float>> Wt;   // [l]: W^T, [nin][pout]   (forward: 8 outputs share a[k])
  std::vector<std::vector<float>> Wp;   // [l]: W,   [nout][pin]   (backward: 8 inputs share d[k][c])
  std::vector<std::vector<float>> bp;   // [l]: bias, [pout]
  std::vector<std::vector<float>> th;   // tanh(weighted_in_[l]) of the hidden layers, kept for computeGrad
  std::vector<float> a_, b_;            // activations (padded)
  std::vector<float> d_, dn_;           // delta, struct of arrays: [4][pmax]
  int pmax;
  bool vec_tanh;
  explicit HostNet(const DdpNet &net) : n(net), L(net.n_layers - 1)
  {
    static const bool vec_ok = tanhf_vec_selfcheck();
    vec_tanh = vec_ok;
    pmax = 8;
    size_t off = 0;
    for (int l = 0; l < L; l++) {
      const int nin = n.layers[l], nout = n.layers[l + 1];
      const int pi = (nin + 7) & ~7, po = (nout + 7) & ~7;
      pin.push_back(pi);
      pout.push_back(po);
      pmax = std::max(pmax, std::max(pi, po));
      const float *W = n.theta + off, *bias = W + (size_t)nin * nout;
      off += (size_t)nin * nout + nout;
      Wt.emplace_back((size_t)nin * po, 0.0f);
      Wp.emplace_back((size_t)nout * pi, 0.0f);
      bp.emplace_back(po, 0.0f);
      for (int j = 0; j < nout; j++) {
        bp[l][j] = bias[j];
        for (int k = 0; k < nin; k++) {
          Wt[l][(size_t)k * po + j] = W[(size_t)j * nin + k];
          Wp[l][(size_t)j * pi + k] = W[(size_t)j * nin + k];
        }
      }
      th.emplace_back(po, 0.0f);
    }
    a_.assign(pmax, 0.0f);
    b_.assign(pmax, 0.0f);
    d_.assign((size_t)4 * pmax, 0.0f);
    dn_.assign((size_t)4 * pmax, 0.0f);
  }
  // out[4] = network(s3..s6, u0, u1)
  void forward(const float *x, const float *u, float *out)
  {
    float *a = a_.data(), *b = b_.data();
    for (int i = 0; i < 4; i++) a[i] = x[3 + i];
    a[4] = u[0];
    a[5] = u[1];
    for (int l = 0; l < L; l++) {
      const int nin = n.layers[l], nout = n.layers[l + 1], po = pout[l];
      const float *W = Wt[l].data(), *bias = bp[l].data();
      // up to four registers of outputs advance together: four independent chains hide the add latency
      for (int j0 = 0; j0 < po; j0 += 32) {
        const int nb = std::min(4, (po - j0) / 8);
        __m256 s[4] = {_mm256_setzero_ps(), _mm256_setzero_ps(), _mm256_setzero_ps(), _mm256_setzero_ps()};
        for (int k = 0; k < nin; k++) {
          const __m256 ak = _mm256_set1_ps(a[k]);
          const float *w = W + (size_t)k * po + j0;
          for (int q = 0; q < nb; q++) s[q] = _mm256_add_ps(s[q], _mm256_mul_ps(_mm256_loadu_ps(w + 8 * q), ak));
        }
        for (int q = 0; q < nb; q++)
          _mm256_storeu_ps(b + j0 + 8 * q, _mm256_add_ps(s[q], _mm256_loadu_ps(bias + j0 + 8 * q)));
      }
      if (l < L - 1) {
        float *t = th[l].data();
        // tanhf as in the reference's host code (MPPI_NNET_NONLINEARITY): the Riccati recursion amplifies a
        // 1e-7 difference in these values to 1e-3 of the feedforward term at T = 250, so no merely accurate
        // substitute -- tanhf_vec.hpp is libm's own algorithm on eight lanes (the same bits, checked at start-up)
        if (vec_tanh) {
          tanhf_vec(b, po);  // padded lanes: tanhf(0) = 0
          for (int j = 0; j < nout; j++) t[j] = b[j];
        } else {
          for (int j = 0; j < nout; j++) b[j] = t[j] = std::tanh(b[j]);
        }
      }
      std::swap(a, b);
    }
    for (int i = 0; i < 4; i++) out[i] = a[i];
  }
  // f(x, u): computeKinematics + computeDynamics (ddp_model_wrapper.h:57-68)
  void f(const DdpProblem &p, const float *x, const float *u, float *dx) override
  {
    dx[0] = std::cos(x[2]) * x[4] - std::sin(x[2]) * x[5];
    dx[1] = std::sin(x[2]) * x[4] + std::cos(x[2]) * x[5];
    dx[2] = p.negate_yaw_der ? -x[6] : x[6];
    forward(x, u, dx + 3);
  }
  // computeGrad (neural_net_model.cu:233-264): 7 x 9 Jacobian of f wrt [x | u]
  void jacobian(const DdpProblem &p, const float *x, const float *u, Mat<kDdpS, kDdpSC> &J) override
  {
    float out[4];
    forward(x, u, out);  // "First do the forward pass", neural_net_model.cu:243-244
    jacobian_after_f(p, x, u, J);
  }
  // th holds the forward pass of this very (x, u): computeGrad's own forward pass would recompute it
  void jacobian_after_f(const DdpProblem &, const float *x, const float *, Mat<kDdpS, kDdpSC> &J) override
  {
    J.zero();
    const float sn = std::sin(x[2]), cs = std::cos(x[2]);
    J.v[0][2] = -sn * x[4] - cs * x[5]; J.v[0][4] = cs; J.v[0][5] = -sn;
    J.v[1][2] = cs * x[4] - sn * x[5];  J.v[1][4] = sn; J.v[1][5] = cs;
    J.v[2][6] = -1.0f;  // regardless of negate_yaw_der (reference quirk)
    // delta[c][k]: derivative of output c wrt the (pre-activation of) neuron k of the current layer;
    // starts as the 4 x 4 identity at the output
    float *d = d_.data(), *dn = dn_.data();
    const int P = pmax;
    for (int c = 0; c < 4; c++)
      for (int k = 0; k < 4; k++) d[c * P + k] = (c == k) ? 1.0f : 0.0f;
    for (int l = L - 1; l >= 0; l--) {
      // delta <- (W_l^T delta) [.* tanh'(z_{l-1}) for l > 0], accumulated over k in ascending order
      const int nout = n.layers[l + 1], pi = pin[l];
      const float *W = Wp[l].data();
      for (int i = 0; i < pi; i += 8) {
        // the four outputs' chains for these eight neurons advance together
        __m256 s[4] = {_mm256_setzero_ps(), _mm256_setzero_ps(), _mm256_setzero_ps(), _mm256_setzero_ps()};
        for (int k = 0; k < nout; k++) {
          const __m256 w = _mm256_loadu_ps(W + (size_t)k * pi + i);
          for (int c = 0; c < 4; c++) s[c] = _mm256_add_ps(s[c], _mm256_mul_ps(w, _mm256_set1_ps(d[c * P + k])));
        }
        if (l > 0) {
          // MPPI_NNET_NONLINEARITY_DERIV: 1 - powf(tanh(z), 2); tanh(z) is the value the forward pass
          // computed from the same z, and powf(x, 2) is the correctly rounded x*x
          const __m256 tz = _mm256_loadu_ps(th[l - 1].data() + i);
          const __m256 zp = _mm256_sub_ps(_mm256_set1_ps(1.0f), _mm256_mul_ps(tz, tz));
          for (int c = 0; c < 4; c++) s[c] = _mm256_mul_ps(s[c], zp);
        }
        for (int c = 0; c < 4; c++) _mm256_storeu_ps(dn + c * P + i, s[c]);
      }
      std::swap(d, dn);
    }
    // bottom-right 4 x 6 block += delta: rows = outputs, columns = [s3..s6, u0, u1]
    for (int o = 0; o < 4; o++)
      for (int i = 0; i < 6; i++) J.v[3 + o][3 + i] += d[o * P + i];
  }
};

// GeneralizedLinear on the host (generalized_linear.cu:140-175).  It has no computeGrad, so
// ModelWrapperDDP::df falls back to Dynamics::df = Eigen::NumericalDiff<..., Central>
// (ddp/ddp_dynamics.h:71-84): step h_j = sqrt(eps_fp32) |z_j| (sqrt(eps) when z_j == 0) in fp32.
struct HostBasis : HostModel {
  const float *W;
  explicit HostBasis(const float *w) : W(w) {}
  void f(const DdpProblem &, const float *x, const float *u, float *dx) override
  {
    dx[0] = std::cos(x[2]) * x[4] - std::sin(x[2]) * x[5];
    dx[1] = std::sin(x[2]) * x[4] + std::cos(x[2]) * x[5];
    dx[2] = -x[6];
    float phi[kNumBfs];
    basis_funcs(x, u[0], u[1], phi);
    basis_dynamics(W, phi, dx + 3);
  }
  void jacobian(const DdpProblem &p, const float *x, const float *u, Mat<kDdpS, kDdpSC> &J) override
  {
    const float eps = std::sqrt(std::numeric_limits<float>::epsilon());
    float z[kDdpSC];
    for (int i = 0; i < kDdpS; i++) z[i] = x[i];
    for (int j = 0; j < kDdpC; j++) z[kDdpS + j] = u[j];
    for (int j = 0; j < kDdpSC; j++) {
      const float zj = z[j];
      float h = eps * std::fabs(zj);
      if (h == 0.0f) h = eps;
      float v1[kDdpS], v2[kDdpS];
      z[j] += h;
      f(p, z, z + kDdpS, v2);
      z[j] -= 2 * h;
      f(p, z, z + kDdpS, v1);
      z[j] = zj;
      for (int i = 0; i < kDdpS; i++) J.v[i][j] = (v2[i] - v1[i]) / (2 * h);
    }
  }
};

inline float clamp_minmax(float v, float lo, float hi)
{  // cwiseMin(u_max).cwiseMax(u_min), ddp.h:62,131
  v = (v < hi) ? v : hi;
  v = (v > lo) ? v : lo;
  return v;
}

// x = A^{-1} b for symmetric 2x2 A via a pivoted LDL^T (Eigen::LDLT semantics: largest |diagonal| first)
struct Ldlt2 {
  int p;        // pivot index
  float l, d0, d1;
  bool ok;
  explicit Ldlt2(const Mat<2, 2> &A)
  {
    p = (std::fabs(A.v[1][1]) > std::fabs(A.v[0][0])) ? 1 : 0;
    const int q = 1 - p;
    d0 = A.v[p][p];
    l = A.v[q][p] / d0;
    d1 = A.v[q][q] - l * A.v[q][p];
    ok = std::isfinite(d0) && std::isfinite(d1) && std::isfinite(l) && d0 != 0.0f;
  }
  void solve(const float b[2], float x[2]) const
  {
    const int q = 1 - p;
    const float z0 = b[p];
    const float z1 = b[q] - l * z0;
    const float w0 = z0 / d0;
    const float w1 = (d1 != 0.0f) ? z1 / d1 : 0.0f;
    const float v1 = w1;
    const float v0 = w0 - l * v1;
    x[p] = v0;
    x[q] = v1;
  }
};

}  // namespace

int ddp_feedback_gains(const DdpNet &net, const DdpProblem &p, const float *x0, const float *target_x,
                       const float *target_u, DdpResult &out)
{
  const int H = p.T;
  const float dt = p.dt;
  std::unique_ptr<HostModel> model;
  if (net.n_layers == 0) model.reset(new HostBasis(net.theta));
  else model.reset(new HostNet(net));
  HostModel &nn = *model;
  std::vector<float> x((size_t)H * kDdpS), u(target_u, target_u + (size_t)H * kDdpC);
  // initial rollout, ddp.h:55-65
  for (int i = 0; i < kDdpS; i++) x[i] = x0[i];
  std::vector<Mat<kDdpS, kDdpSC>> df(H);
  for (int i = 1; i < H; i++) {
    float *up = &u[(size_t)(i - 1) * kDdpC];
    if (i < H - 1)
      for (int j = 0; j < kDdpC; j++) up[j] = clamp_minmax(up[j], p.u_lo[j], p.u_hi[j]);
    float dx[kDdpS];
    nn.f(p, &x[(size_t)(i - 1) * kDdpS], up, dx);
    // the Jacobian of ddp.h:73 at the same point, while its forward pass is at hand
    nn.jacobian_after_f(p, &x[(size_t)(i - 1) * kDdpS], up, df[i - 1]);
    for (int s = 0; s < kDdpS; s++) x[(size_t)i * kDdpS + s] = x[(size_t)(i - 1) * kDdpS + s] + dx[s] * dt;
  }
  nn.jacobian(p, &x[(size_t)(H - 1) * kDdpS], &u[(size_t)(H - 1) * kDdpC], df[H - 1]);
  // Jacobians (scaled, + I) and cost derivatives, ddp.h:71-80
  std::vector<float> dL((size_t)H * kDdpSC);
  for (int k = 0; k < H; k++) {
    for (int i = 0; i < kDdpS; i++)
      for (int j = 0; j < kDdpSC; j++) df[k].v[i][j] = df[k].v[i][j] * dt;
    for (int i = 0; i < kDdpS; i++) df[k].v[i][i] += 1.0f;
    for (int i = 0; i < kDdpS; i++) dL[(size_t)k * kDdpSC + i] = p.Q[i] * (x[(size_t)k * kDdpS + i] - target_x[(size_t)k * kDdpS + i]);
    for (int j = 0; j < kDdpC; j++) dL[(size_t)k * kDdpSC + kDdpS + j] = p.R[j] * (u[(size_t)k * kDdpC + j] - target_u[(size_t)k * kDdpC + j]);
  }
  // boundary condition, ddp.h:83-87 (target of the terminal cost = target_x[H-1], mppi_controller.cu:439)
  Mat<kDdpS, kDdpS> Vxx;
  Vxx.zero();
  for (int i = 0; i < kDdpS; i++) Vxx.v[i][i] = p.Qf[i];
  float Vx[kDdpS], Vlast = 0.0f;
  for (int i = 0; i < kDdpS; i++) {
    const float e = x[(size_t)(H - 1) * kDdpS + i] - target_x[(size_t)(H - 1) * kDdpS + i];
    Vx[i] = p.Qf[i] * e;
    Vlast += e * (p.Qf[i] * e);
  }
  out.feedback.assign((size_t)H * kDdpC * kDdpS, 0.0f);
  out.feedforward.assign((size_t)H * kDdpC, 0.0f);
  // backward pass, ddp.h:90-123
  for (int k = H - 2; k >= 0; k--) {
    Mat<kDdpS, kDdpS> Phi;
    Mat<kDdpS, kDdpC> B;
    for (int i = 0; i < kDdpS; i++) {
      for (int j = 0; j < kDdpS; j++) Phi.v[i][j] = df[k].v[i][j];
      for (int j = 0; j < kDdpC; j++) B.v[i][j] = df[k].v[i][kDdpS + j];
    }
    const Mat<kDdpS, kDdpS> PhiT = transpose(Phi);
    const Mat<kDdpC, kDdpS> BT = transpose(B);
    float qx[kDdpS], qu[kDdpC];
    for (int i = 0; i < kDdpS; i++) {
      float s = 0.0f;
      for (int m = 0; m < kDdpS; m++) s += PhiT.v[i][m] * Vx[m];
      qx[i] = dL[(size_t)k * kDdpSC + i] * dt + s;
    }
    for (int j = 0; j < kDdpC; j++) {
      float s = 0.0f;
      for (int m = 0; m < kDdpS; m++) s += BT.v[j][m] * Vx[m];
      qu[j] = dL[(size_t)k * kDdpSC + kDdpS + j] * dt + s;
    }
    const Mat8<kDdpS> Vxx8 = pad8(Vxx), Phi8 = pad8(Phi), B8 = pad8(B);
    const Mat8<kDdpC> BtV8 = mul8(BT.v, Vxx8);
    Mat<kDdpC, kDdpS> qux = unpad8<kDdpC, kDdpS>(mul8<kDdpC, kDdpS>(BtV8, Phi8));  // d2L's bottom-left block is zero
    Mat<kDdpS, kDdpS> qxx = unpad8<kDdpS, kDdpS>(mul8<kDdpS, kDdpS>(mul8(PhiT.v, Vxx8), Phi8));
    for (int i = 0; i < kDdpS; i++) qxx.v[i][i] = p.Q[i] * dt + qxx.v[i][i];
    Mat<kDdpC, kDdpC> quu = unpad8<kDdpC, kDdpC>(mul8<kDdpC, kDdpS>(BtV8, B8));
    for (int j = 0; j < kDdpC; j++) quu.v[j][j] = p.R[j] * dt + quu.v[j][j];
    const Ldlt2 ldlt(quu);
    if (!ldlt.ok) return 1;
    Mat<kDdpC, kDdpS> Lk;
    for (int c = 0; c < kDdpS; c++) {
      const float b[2] = {-qux.v[0][c], -qux.v[1][c]};
      float s[2];
      ldlt.solve(b, s);
      Lk.v[0][c] = s[0];
      Lk.v[1][c] = s[1];
    }
    float lk[2];
    {
      const float b[2] = {-qu[0], -qu[1]};
      ldlt.solve(b, lk);
    }
    for (int j = 0; j < kDdpC; j++) {
      for (int c = 0; c < kDdpS; c++) out.feedback[((size_t)k * kDdpC + j) * kDdpS + c] = Lk.v[j][c];
      out.feedforward[(size_t)k * kDdpC + j] = lk[j];
    }
    // value function, ddp.h:117-122
    const Mat<kDdpS, kDdpC> quxT = transpose(qux);
    const Mat<kDdpS, kDdpS> corr = unpad8<kDdpS, kDdpS>(mul8(quxT.v, pad8(Lk)));
    Mat<kDdpS, kDdpS> Vn;
    for (int i = 0; i < kDdpS; i++)
      for (int j = 0; j < kDdpS; j++) Vn.v[i][j] = qxx.v[i][j] + corr.v[i][j];
    for (int i = 0; i < kDdpS; i++)
      for (int j = 0; j < kDdpS; j++) Vxx.v[i][j] = 0.5f * (Vn.v[i][j] + Vn.v[j][i]);
    for (int i = 0; i < kDdpS; i++) Vx[i] = qx[i] + (quxT.v[i][0] * lk[0] + quxT.v[i][1] * lk[1]);
  }
  // forward pass with alpha = 1; iteration 0 is always accepted (ddp.h:125-152)
  out.x.assign((size_t)H * kDdpS, 0.0f);
  out.u.assign((size_t)H * kDdpC, 0.0f);
  out.cost.assign(H, 0.0f);
  for (int i = 0; i < kDdpS; i++) out.x[i] = x[i];
  for (int k = 0; k + 1 < H; k++) {
    float dx[kDdpS];
    for (int i = 0; i < kDdpS; i++) dx[i] = out.x[(size_t)k * kDdpS + i] - x[(size_t)k * kDdpS + i];
    float un[kDdpC];
    for (int j = 0; j < kDdpC; j++) {
      float s = 0.0f;
      for (int i = 0; i < kDdpS; i++) s += out.feedback[((size_t)k * kDdpC + j) * kDdpS + i] * dx[i];
      un[j] = (u[(size_t)k * kDdpC + j] + 1.0f * out.feedforward[(size_t)k * kDdpC + j]) + s;
      un[j] = clamp_minmax(un[j], p.u_lo[j], p.u_hi[j]);
      out.u[(size_t)k * kDdpC + j] = un[j];
    }
    float fx[kDdpS];
    nn.f(p, &out.x[(size_t)k * kDdpS], un, fx);
    for (int i = 0; i < kDdpS; i++) out.x[(size_t)(k + 1) * kDdpS + i] = out.x[(size_t)k * kDdpS + i] + fx[i] * dt;
    float sc = 0.0f, cc = 0.0f;
    for (int i = 0; i < kDdpS; i++) {
      const float e = out.x[(size_t)k * kDdpS + i] - target_x[(size_t)k * kDdpS + i];
      sc += e * (p.Q[i] * e);
    }
    for (int j = 0; j < kDdpC; j++) {
      const float e = un[j] - target_u[(size_t)k * kDdpC + j];
      cc += e * (p.R[j] * e);
    }
    out.cost[k] = (sc + cc) * dt;
  }
  out.cost[H - 1] = Vlast;
  float tot = 0.0f;
  for (int k = 0; k < H; k++) tot += out.cost[k];
  out.total_cost = tot;
  out.iterations = 1;
  return 0;
}

}  // namespace mppi
