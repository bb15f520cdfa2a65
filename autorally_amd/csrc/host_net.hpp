// host_net.hpp -- the network on the host, in the ROLLOUTS' arithmetic, eight output neurons per AVX2 register.
//
// computeNominalTraj (PI/mppi_controller.cu:501-519) replays U_ through the host model T times per solve; with two
// controllers per tick (run_control_loop.cuh:218-219) those replays were two thirds of the tick once the solves
// themselves shared a launch.  Per neuron the arithmetic is what mppi_nominal_traj always did -- z = 0; for k
// ascending z = fmaf(W[j][k], a[k], z); z += b[j]; tanhf(z) on hidden layers -- so state_solution_ does not change
// by a bit: a register holds eight neurons j, the activation a[k] is broadcast, _mm256_fmadd_ps is eight fmaf, and
// tanhf_vec.hpp returns libm's tanhf.
#pragma once
#include <algorithm>
#include <vector>

#include "tanhf_vec.hpp"

namespace mppi {

struct HostNetFma {
  int L = 0;                              // weight matrices
  std::vector<int> nin, nout, pout;       // widths; outputs rounded up to 8
  std::vector<std::vector<float>> Wt;     // [l]: W^T, [nin][pout], zero padded
  std::vector<std::vector<float>> bp;     // [l]: bias, [pout]
  std::vector<float> a_, b_;
  bool vec_tanh = false;                  // tanhf8 == libm's tanhf on this machine (checked once)

  // layers[n_layers], theta packed [W1|b1|W2|b2|..] (neural_net_model.cu:120-141)
  void init(const int *layers, int n_layers, const float *theta)
  {
    L = n_layers - 1;
    nin.clear(); nout.clear(); pout.clear(); Wt.clear(); bp.clear();
    int pmax = 8;
    size_t off = 0;
    for (int l = 0; l < L; l++) {
      const int ni = layers[l], no = layers[l + 1], po = (no + 7) & ~7;
      nin.push_back(ni); nout.push_back(no); pout.push_back(po);
      pmax = std::max(pmax, std::max((ni + 7) & ~7, po));
      const float *W = theta + off, *bias = W + (size_t)ni * no;
      off += (size_t)ni * no + no;
      Wt.emplace_back((size_t)ni * po, 0.0f);
      bp.emplace_back((size_t)po, 0.0f);
      for (int j = 0; j < no; j++) {
        bp[l][j] = bias[j];
        for (int k = 0; k < ni; k++) Wt[l][(size_t)k * po + j] = W[(size_t)j * ni + k];
      }
    }
    a_.assign((size_t)pmax, 0.0f);
    b_.assign((size_t)pmax, 0.0f);
    static const bool ok = tanhf_vec_selfcheck();
    vec_tanh = ok;
  }

  // out[4] = network([s3, s4, s5, s6, u0, u1])
  void forward(const float in[6], float out[4])
  {
    float *a = a_.data(), *b = b_.data();
    for (int i = 0; i < 6; i++) a[i] = in[i];
    for (int l = 0; l < L; l++) {
      const int ni = nin[l], no = nout[l], po = pout[l];
      const float *W = Wt[l].data(), *bias = bp[l].data();
      for (int j0 = 0; j0 < po; j0 += 32) {  // up to four registers of neurons advance together
        const int nb = std::min(4, (po - j0) / 8);
        __m256 s[4] = {_mm256_setzero_ps(), _mm256_setzero_ps(), _mm256_setzero_ps(), _mm256_setzero_ps()};
        for (int k = 0; k < ni; k++) {
          const __m256 ak = _mm256_set1_ps(a[k]);
          const float *w = W + (size_t)k * po + j0;
          for (int q = 0; q < nb; q++) s[q] = _mm256_fmadd_ps(_mm256_loadu_ps(w + 8 * q), ak, s[q]);
        }
        for (int q = 0; q < nb; q++)
          _mm256_storeu_ps(b + j0 + 8 * q, _mm256_add_ps(s[q], _mm256_loadu_ps(bias + j0 + 8 * q)));
      }
      if (l < L - 1) {
        if (vec_tanh) tanhf_vec(b, po);  // padded lanes hold tanhf(0) = 0
        else
          for (int j = 0; j < no; j++) b[j] = tanhf(b[j]);
      }
      std::swap(a, b);
    }
    for (int i = 0; i < 4; i++) out[i] = a[i];
  }
};

// Two replays in lockstep: a layer's k-ascending fmaf chains are latency-bound (four registers of neurons = four chains in
// flight per replay, the FMA units take eight), so the two controllers of a control tick -- two independent replays of the
// same length -- advance together at the cost of one.  Per replay the arithmetic is HostNetFma::forward's, instruction for
// instruction.  The two networks may be different objects (same layer list).
inline void host_net_forward2(HostNetFma &A, HostNetFma &B, const float inA[6], const float inB[6], float outA[4], float outB[4])
{
  float *a0 = A.a_.data(), *b0 = A.b_.data(), *a1 = B.a_.data(), *b1 = B.b_.data();
  for (int i = 0; i < 6; i++) { a0[i] = inA[i]; a1[i] = inB[i]; }
  for (int l = 0; l < A.L; l++) {
    const int ni = A.nin[l], no = A.nout[l], po = A.pout[l];
    const float *W0 = A.Wt[l].data(), *W1 = B.Wt[l].data(), *bias0 = A.bp[l].data(), *bias1 = B.bp[l].data();
    for (int j0 = 0; j0 < po; j0 += 32) {
      const int nb = std::min(4, (po - j0) / 8);
      __m256 s0[4] = {_mm256_setzero_ps(), _mm256_setzero_ps(), _mm256_setzero_ps(), _mm256_setzero_ps()};
      __m256 s1[4] = {_mm256_setzero_ps(), _mm256_setzero_ps(), _mm256_setzero_ps(), _mm256_setzero_ps()};
      for (int k = 0; k < ni; k++) {
        const __m256 ak0 = _mm256_set1_ps(a0[k]), ak1 = _mm256_set1_ps(a1[k]);
        const float *w0 = W0 + (size_t)k * po + j0, *w1 = W1 + (size_t)k * po + j0;
        for (int q = 0; q < nb; q++) {
          s0[q] = _mm256_fmadd_ps(_mm256_loadu_ps(w0 + 8 * q), ak0, s0[q]);
          s1[q] = _mm256_fmadd_ps(_mm256_loadu_ps(w1 + 8 * q), ak1, s1[q]);
        }
      }
      for (int q = 0; q < nb; q++) {
        _mm256_storeu_ps(b0 + j0 + 8 * q, _mm256_add_ps(s0[q], _mm256_loadu_ps(bias0 + j0 + 8 * q)));
        _mm256_storeu_ps(b1 + j0 + 8 * q, _mm256_add_ps(s1[q], _mm256_loadu_ps(bias1 + j0 + 8 * q)));
      }
    }
    if (l < A.L - 1) {
      if (A.vec_tanh) {
        for (int j = 0; j < po; j += 8) {  // the two replays' registers alternate: independent tanhf8 chains
          const __m256 t0 = tanhf8(_mm256_loadu_ps(b0 + j)), t1 = tanhf8(_mm256_loadu_ps(b1 + j));
          _mm256_storeu_ps(b0 + j, t0);
          _mm256_storeu_ps(b1 + j, t1);
        }
      } else {
        for (int j = 0; j < no; j++) { b0[j] = tanhf(b0[j]); b1[j] = tanhf(b1[j]); }
      }
    }
    std::swap(a0, b0);
    std::swap(a1, b1);
  }
  for (int i = 0; i < 4; i++) { outA[i] = a0[i]; outB[i] = a1[i]; }
}

}  // namespace mppi
