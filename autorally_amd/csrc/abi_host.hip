// abi_host.hip -- the host-side halves of a control tick: nominal trajectory replays (mppi_controller.cu:501-519), DDP
// feedback gains (:402-445), and the helper thread that takes one of a pair (mppi_set_host_threads).
#include "abi_internal.hpp"

#include <stdexcept>

using namespace mppi;
using namespace mppi_abi;

namespace mppi_abi {

// One helper thread for the host-side halves of a control tick that come in pairs (the two controllers' nominal replays,
// their two DDP passes: run_control_loop.cuh:218-225 -- independent work on two handles): the caller's thread does one, the
// helper the other.  Off unless mppi_set_host_threads(2) was called.  The helper sleeps on a condition variable; arm() wakes it
// (mppi_compute_control_batch_async does, a solve's length before the replays are due) and it then polls for work for 1 ms
// after the last job, so that the hand-over costs a cache line, not a futex wake.
class HostHelper {
 public:
  ~HostHelper()
  {
    if (!th_.joinable()) return;
    {
      std::lock_guard<std::mutex> lk(mu_);
      quit_ = true;
      wake_ = true;
    }
    cv_.notify_one();
    th_.join();
  }
  void arm()
  {
    std::call_once(started_, [this] { th_ = std::thread([this] { loop(); }); });
    if (spinning_.load()) return;  // (seq_cst, with the stores in run_pair / loop: a job is never posted to a helper going to sleep unseen)
    {
      std::lock_guard<std::mutex> lk(mu_);
      wake_ = true;
    }
    cv_.notify_one();
  }
  // runs `other` on the helper and `mine` on the caller's thread; returns when both are done.  One pair at a time: a second
  // caller (another control loop of the process) runs both halves itself.
  void run_pair(const std::function<void()> &other, const std::function<void()> &mine)
  {
    std::unique_lock<std::mutex> busy(pair_mu_, std::try_to_lock);
    if (!busy.owns_lock()) {
      mine();
      other();
      return;
    }
    job_ = &other;
    failed_.store(false);
    done_.store(false);
    posted_.store(true);
    arm();
    // the caller leaves only after the helper is through with `other` (which refers to the caller's frame), whatever
    // `mine` does: an exception of `mine` is rethrown after the wait, one of `other` (caught on the helper) afterwards
    struct Wait {
      std::atomic<bool> &d;
      ~Wait() { while (!d.load(std::memory_order_acquire)) __builtin_ia32_pause(); }
    };
    {
      Wait w{done_};
      mine();
    }
    if (failed_.load()) throw std::runtime_error("host helper: the paired job threw");
  }

 private:
  void loop()
  {
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [this] { return wake_; });
        wake_ = false;
        if (quit_) return;
      }
      do {
        spinning_.store(true);
        auto until = std::chrono::steady_clock::now() + std::chrono::milliseconds(1);
        unsigned spins = 0;
        for (;;) {
          if (posted_.load(std::memory_order_acquire)) {
            posted_.store(false);
            try {
              (*job_)();
            } catch (...) {  // (bad_alloc of a result vector ...): reported by run_pair on the caller's thread
              failed_.store(true);
            }
            done_.store(true, std::memory_order_release);
            until = std::chrono::steady_clock::now() + std::chrono::milliseconds(1);
          } else {
            __builtin_ia32_pause();
            if ((++spins & 0xFF) == 0 && std::chrono::steady_clock::now() > until) break;
          }
        }
        spinning_.store(false);
      } while (posted_.load());  // posted while this thread was on its way to sleep: its poster saw `spinning_` and sent no wake
    }
  }
  std::once_flag started_;
  std::thread th_;
  std::mutex mu_, pair_mu_;
  std::condition_variable cv_;
  bool wake_ = false, quit_ = false;
  std::atomic<bool> spinning_{false}, posted_{false}, done_{false}, failed_{false};
  const std::function<void()> *job_ = nullptr;
};
std::atomic<int> g_host_threads{1};
HostHelper &host_helper()
{
  static HostHelper hh;
  return hh;
}

void host_helper_arm() { host_helper().arm(); }

}  // namespace mppi_abi

extern "C" {

int mppi_nominal_traj(mppi_handle *h, const float state[MPPI_STATE_DIM], float *state_seq, float *control_seq)
{
  if (!h || !state || !state_seq || !control_seq) return MPPI_ERR_INVALID;
  if (!h->have_nn) return fail(h, MPPI_ERR_STATE, "mppi_set_nn_params has not been called");
  if (h->pending) {
    int rc = mppi_synchronize(h);
    if (rc) return rc;
  }
  // computeNominalTraj (mppi_controller.cu:501-519) -> host updateState (neural_net_model.cu:280-288):
  // like the reference this replay runs on the host (T sequential 1.4k-MAC steps).
  float s[kStateDim];
  for (int i = 0; i < kStateDim; i++) s[i] = state[i];
  for (int t = 0; t < h->T; t++) {
    for (int i = 0; i < kStateDim; i++) state_seq[t * kStateDim + i] = s[i];
    float u[2] = {h->U[2 * t], h->U[2 * t + 1]};
    for (int i = 0; i < 2; i++) {
      if (u[i] < h->u_lo[i]) u[i] = h->u_lo[i];
      else if (u[i] > h->u_hi[i]) u[i] = h->u_hi[i];
    }
    const float c = cosf(s[2]), sn = sinf(s[2]);
    float sd[kStateDim];
    sd[0] = fmaf(c, s[4], -(sn * s[5]));
    sd[1] = fmaf(sn, s[4], c * s[5]);
    sd[2] = h->cfg.negate_yaw_der ? -s[6] : s[6];
    if (h->basis) {  // GeneralizedLinear::updateState (generalized_linear.cu:140-167), yaw rate always negated
      float phi[kNumBfs];
      sd[2] = -s[6];
      basis_funcs(s, u[0], u[1], phi);
      basis_dynamics(h->theta.data(), phi, sd + 3);
      for (int i = 0; i < kStateDim; i++) s[i] = fmaf(sd[i], h->dt, s[i]);
      control_seq[2 * t] = u[0];
      control_seq[2 * t + 1] = u[1];
      continue;
    }
    // the network: per neuron the k-ascending fmaf chain, bias added afterwards, tanhf -- eight neurons per AVX2
    // register (host_net.hpp; the same values as the scalar loops, bit for bit)
    const float nin6[6] = {s[3], s[4], s[5], s[6], u[0], u[1]};
    h->hnet.forward(nin6, sd + 3);
    for (int i = 0; i < kStateDim; i++) s[i] = fmaf(sd[i], h->dt, s[i]);
    control_seq[2 * t] = u[0];
    control_seq[2 * t + 1] = u[1];
  }
  return MPPI_OK;
}

int mppi_nominal_traj_pair(mppi_handle *ha, const float state_a[MPPI_STATE_DIM], float *state_seq_a, float *control_seq_a,
                           mppi_handle *hb, const float state_b[MPPI_STATE_DIM], float *state_seq_b, float *control_seq_b)
{
  if (!ha || !hb || ha == hb) return MPPI_ERR_INVALID;
  // two network replays of the same length advance in lockstep (host_net_forward2); anything else: one after the other
  const bool lockstep = !ha->basis && !hb->basis && ha->have_nn && hb->have_nn && ha->T == hb->T &&
                        ha->net.n_layers == hb->net.n_layers &&
                        memcmp(ha->net.layers, hb->net.layers, sizeof(ha->net.layers)) == 0 && state_a && state_b &&
                        state_seq_a && state_seq_b && control_seq_a && control_seq_b;
  if (g_host_threads.load(std::memory_order_relaxed) >= 2) {  // one replay per thread (mppi_set_host_threads), any model
    for (mppi_handle *h : {ha, hb})
      if (h->pending) {
        const int rc = mppi_synchronize(h);
        if (rc) return rc;
      }
    int rca = MPPI_OK, rcb = MPPI_OK;
    try {
      host_helper().run_pair([&] { rcb = mppi_nominal_traj(hb, state_b, state_seq_b, control_seq_b); },
                             [&] { rca = mppi_nominal_traj(ha, state_a, state_seq_a, control_seq_a); });
    } catch (const std::exception &e) {
      return fail(ha, MPPI_ERR_HIP, e.what());
    }
    return rca ? rca : rcb;
  }
  if (!lockstep) {
    const int rc = mppi_nominal_traj(ha, state_a, state_seq_a, control_seq_a);
    return rc ? rc : mppi_nominal_traj(hb, state_b, state_seq_b, control_seq_b);
  }
  for (mppi_handle *h : {ha, hb})
    if (h->pending) {
      const int rc = mppi_synchronize(h);
      if (rc) return rc;
    }
  mppi_handle *hs[2] = {ha, hb};
  float *sseq[2] = {state_seq_a, state_seq_b}, *cseq[2] = {control_seq_a, control_seq_b};
  float s[2][kStateDim], sd[2][kStateDim], in6[2][6];
  for (int i = 0; i < kStateDim; i++) { s[0][i] = state_a[i]; s[1][i] = state_b[i]; }
  for (int t = 0; t < ha->T; t++) {
    for (int q = 0; q < 2; q++) {  // per replay exactly the statements of mppi_nominal_traj
      const mppi_handle *h = hs[q];
      for (int i = 0; i < kStateDim; i++) sseq[q][t * kStateDim + i] = s[q][i];
      float u[2] = {h->U[2 * t], h->U[2 * t + 1]};
      for (int i = 0; i < 2; i++) {
        if (u[i] < h->u_lo[i]) u[i] = h->u_lo[i];
        else if (u[i] > h->u_hi[i]) u[i] = h->u_hi[i];
      }
      const float c = cosf(s[q][2]), sn = sinf(s[q][2]);
      sd[q][0] = fmaf(c, s[q][4], -(sn * s[q][5]));
      sd[q][1] = fmaf(sn, s[q][4], c * s[q][5]);
      sd[q][2] = h->cfg.negate_yaw_der ? -s[q][6] : s[q][6];
      in6[q][0] = s[q][3]; in6[q][1] = s[q][4]; in6[q][2] = s[q][5]; in6[q][3] = s[q][6]; in6[q][4] = u[0]; in6[q][5] = u[1];
      cseq[q][2 * t] = u[0];
      cseq[q][2 * t + 1] = u[1];
    }
    host_net_forward2(ha->hnet, hb->hnet, in6[0], in6[1], sd[0] + 3, sd[1] + 3);
    for (int q = 0; q < 2; q++)
      for (int i = 0; i < kStateDim; i++) s[q][i] = fmaf(sd[q][i], hs[q]->dt, s[q][i]);
  }
  return MPPI_OK;
}

int mppi_set_ddp_weights(mppi_handle *h, const float Q[MPPI_STATE_DIM], const float R[MPPI_CONTROL_DIM],
                         const float Qf[MPPI_STATE_DIM])
{
  if (!h || !Q || !R || !Qf) return MPPI_ERR_INVALID;
  for (int i = 0; i < kStateDim; i++) {
    if (!(Q[i] >= 0.0f) || !(Qf[i] >= 0.0f)) return fail(h, MPPI_ERR_INVALID, "Q and Qf must be non-negative");
  }
  for (int j = 0; j < kControlDim; j++)
    if (!(R[j] > 0.0f)) return fail(h, MPPI_ERR_INVALID, "R must be positive");
  memcpy(h->ddp_Q, Q, sizeof(h->ddp_Q));
  memcpy(h->ddp_R, R, sizeof(h->ddp_R));
  memcpy(h->ddp_Qf, Qf, sizeof(h->ddp_Qf));
  return MPPI_OK;
}

int mppi_compute_feedback_gains(mppi_handle *h, const float state[MPPI_STATE_DIM],
                                const float *target_state_seq, const float *target_control_seq)
{
  if (!h || !state) return MPPI_ERR_INVALID;
  if ((target_state_seq == nullptr) != (target_control_seq == nullptr))
    return fail(h, MPPI_ERR_INVALID, "give both target sequences or neither");
  if (!h->have_nn) return fail(h, MPPI_ERR_STATE, "mppi_set_nn_params has not been called");
  const int T = h->T;
  std::vector<float> xs((size_t)T * kStateDim), us((size_t)T * kControlDim);
  if (target_state_seq) {
    memcpy(xs.data(), target_state_seq, sizeof(float) * xs.size());
    memcpy(us.data(), target_control_seq, sizeof(float) * us.size());
  } else {
    int rc = mppi_nominal_traj(h, state, xs.data(), us.data());  // state_solution_, control_solution_
    if (rc) return rc;
  }
  DdpNet net;
  net.n_layers = h->basis ? 0 : h->net.n_layers;  // 0: basis-function model, theta = W[4][25]
  net.layers = h->net.layers;
  net.theta = h->theta.data();
  net.max_width = h->net.max_width;
  DdpProblem p;
  p.T = T;
  p.dt = (float)(1.0 / h->cfg.hz);  // mppi_controller.cu:408
  for (int j = 0; j < kControlDim; j++) { p.u_lo[j] = h->u_lo[j]; p.u_hi[j] = h->u_hi[j]; p.R[j] = h->ddp_R[j]; }
  for (int i = 0; i < kStateDim; i++) { p.Q[i] = h->ddp_Q[i]; p.Qf[i] = h->ddp_Qf[i]; }
  p.negate_yaw_der = h->cfg.negate_yaw_der;
  h->have_ddp = false;
  if (ddp_feedback_gains(net, p, state, xs.data(), us.data(), h->ddp) != 0)
    return fail(h, MPPI_ERR_STATE, "DDP: control Hessian could not be factorised");
  h->have_ddp = true;
  return MPPI_OK;
}

int mppi_compute_feedback_gains_pair(mppi_handle *ha, const float state_a[MPPI_STATE_DIM], const float *target_state_seq_a,
                                     const float *target_control_seq_a, mppi_handle *hb, const float state_b[MPPI_STATE_DIM],
                                     const float *target_state_seq_b, const float *target_control_seq_b)
{
  if (!ha || !hb || ha == hb) return MPPI_ERR_INVALID;
  if (g_host_threads.load(std::memory_order_relaxed) >= 2) {
    // the nominal replays inside (no targets given) synchronise their handle: do that on this thread, which has the device
    for (mppi_handle *h : {ha, hb})
      if (h->pending) {
        const int rc = mppi_synchronize(h);
        if (rc) return rc;
      }
    int rca = MPPI_OK, rcb = MPPI_OK;
    try {  // no exception crosses the ABI: one thrown on either thread (an allocation of the result vectors) becomes a status
      host_helper().run_pair([&] { rcb = mppi_compute_feedback_gains(hb, state_b, target_state_seq_b, target_control_seq_b); },
                             [&] { rca = mppi_compute_feedback_gains(ha, state_a, target_state_seq_a, target_control_seq_a); });
    } catch (const std::exception &e) {
      return fail(ha, MPPI_ERR_HIP, e.what());
    }
    return rca ? rca : rcb;
  }
  const int rc = mppi_compute_feedback_gains(ha, state_a, target_state_seq_a, target_control_seq_a);
  return rc ? rc : mppi_compute_feedback_gains(hb, state_b, target_state_seq_b, target_control_seq_b);
}

int mppi_set_host_threads(int n)
{
  if (n < 1 || n > 2) return MPPI_ERR_INVALID;
  g_host_threads.store(n, std::memory_order_relaxed);
  if (n >= 2) host_helper().arm();  // starts the helper now, not inside the first tick
  return MPPI_OK;
}

int mppi_get_feedback_gains(mppi_handle *h, float *feedback, float *feedforward, float *state_traj,
                            float *control_traj, float *total_cost)
{
  if (!h) return MPPI_ERR_INVALID;
  if (!h->have_ddp) return fail(h, MPPI_ERR_STATE, "mppi_compute_feedback_gains has not succeeded yet");
  const DdpResult &r = h->ddp;
  if (feedback) memcpy(feedback, r.feedback.data(), sizeof(float) * r.feedback.size());
  if (feedforward) memcpy(feedforward, r.feedforward.data(), sizeof(float) * r.feedforward.size());
  if (state_traj) memcpy(state_traj, r.x.data(), sizeof(float) * r.x.size());
  if (control_traj) memcpy(control_traj, r.u.data(), sizeof(float) * r.u.size());
  if (total_cost) *total_cost = r.total_cost;
  return MPPI_OK;
}

}  // extern "C"
