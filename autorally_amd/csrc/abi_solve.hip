// abi_solve.hip -- one MPPI solve (PI/mppi_controller.cu:600-671) as launches on the handle's stream: noise source, rollout
// kernel, tail kernel, polling of the host-mapped result block; batched solves of several handles; result getters.
#include "abi_internal.hpp"

using namespace mppi;
using namespace mppi_abi;

namespace mppi_abi {

// Batched solves of several handles go to ONE stream per device, shared by all handles and never destroyed, so
// that the instances' kernels are one launch and a handle never holds another handle's stream.
std::mutex g_batch_mu;
hipStream_t g_batch_stream[64] = {};
hipStream_t batch_stream(int device)
{
  if (device < 0 || device >= 64) return nullptr;
  std::lock_guard<std::mutex> lk(g_batch_mu);
  if (!g_batch_stream[device] &&
      hipStreamCreateWithFlags(&g_batch_stream[device], hipStreamNonBlocking) != hipSuccess)
    g_batch_stream[device] = nullptr;
  return g_batch_stream[device];
}

void fill_cost_args(const mppi_handle *h, CostArgs &c)
{
  const mppi_cost_params &p = h->cost;
  c.desired_speed = p.desired_speed;
  c.speed_coeff = p.speed_coeff;
  c.track_coeff = p.track_coeff;
  c.max_slip_ang = p.max_slip_ang;
  c.slip_penalty = p.slip_penalty;
  c.track_slop = p.track_slop;
  c.crash_coeff = p.crash_coeff;
  c.steering_coeff = p.steering_coeff;
  c.throttle_coeff = p.throttle_coeff;
  c.boundary_threshold = p.boundary_threshold;
  c.crash_cost_discounted = (float)((1.0 - (double)p.discount) * (double)p.crash_coeff);
  c.l1_cost = p.l1_cost ? 1 : 0;
  for (int i = 0; i < 3; i++) {
    c.r_c1[i] = h->r_c1[i];
    c.r_c2[i] = h->r_c2[i];
    c.trs[i] = h->trs[i];
  }
  c.affine = (h->r_c1[2] == 0.0f && h->r_c2[2] == 0.0f && h->trs[2] == 1.0f) ? 1 : 0;
  const float n0 = h->cfg.exploration_std[0], n1 = h->cfg.exploration_std[1];
  const bool nu_ok = std::isfinite(n0) && std::isfinite(n1) && n0 * n0 > 0.0f && n1 * n1 > 0.0f &&
                     std::isfinite(n0 * n0) && std::isfinite(n1 * n1);
  c.need_control_cost = (p.steering_coeff != 0.0f || p.throttle_coeff != 0.0f || !nu_ok) ? 1 : 0;
  c.map_w = h->map_w;
  c.map_h = h->map_h;
  c.map = h->d_map;
}

void fill_rollout_args(const mppi_handle *h, const float *state, float *noise, RolloutArgs &a)
{
  for (int i = 0; i < kStateDim; i++) a.state[i] = state[i];
  a.U = h->d_in;
  a.noise = noise;
  a.costs = h->d_costs;
  const Form f = form_of(h);
  a.wpack = form_is_row(f) ? h->d_rowpack : form_is_row64(f) ? h->d_row64pack : (f == Form::M44 || f == Form::M44Chain) ? h->d_m44pack
            : f == Form::ValuReg ? h->d_theta_s : (f == Form::ValuLds || h->basis) ? h->d_theta : h->d_wpack;
  a.inv_t = h->d_invt;
  a.K = h->K;
  a.T = h->T;
  a.opt_delay = h->cfg.optimization_stride;
  a.k99 = h->k99;
  for (int i = 0; i < 2; i++) {
    a.nu[i] = h->cfg.exploration_std[i];
    a.u_lo[i] = h->u_lo[i];
    a.u_hi[i] = h->u_hi[i];
  }
  a.dt = h->dt;
  a.negate_yaw_der = h->cfg.negate_yaw_der ? 1 : 0;
  a.rng_in = nullptr;
  a.rng_out = nullptr;
  a.inline_noise = 0;
  a.gate = nullptr;
  a.gate_seq = 0;
  a.min_cost = nullptr;  // tag_min_cost
  a.min_cost_tag = 0;
  // roles 32-34 are waits of the streaming tail kernel (tail_launch): the rollout runs with its defaults
  a.spin_budget = h->fault_wave >= 32 ? 0 : h->spin_budget;
  a.fault_wave = h->fault_wave >= 32 ? 0 : h->fault_wave;
  fill_cost_args(h, a.cost);
}

// Where the tail stage is the streaming kernel (K > 4096) AND the rollout form is one of rollout_multi.hip's (the automatic ones
// beyond 8192 rollouts) beta comes out of the rollout kernel: -0.8 us of the step at K = 16 384, -1.7 at config 4.  With the row /
// m44 forms (every K <= 8192 by default) the same was measured a LOSS (headline 0.0393 -> 0.0403 ms: the device-scope atomic at the end of the
// cost wave costs the rollout kernel 0.6 us, and the tail's own block reduction was never on its critical path) -- not used there.
static unsigned long long *min_cost_keys(const mppi_handle *h)
{
  const Form f = form_of(h);
  const bool publishes = f == Form::Multi2 || f == Form::Multi4 || f == Form::Multi4Tree;  // rollout_multi.hip
  return (h->use_min_cost && tail_is_stream(h->K) && publishes) ? h->d_min_cost : nullptr;
}

int tag_min_cost(mppi_handle *h, RolloutArgs &a, hipStream_t stream)
{
  h->min_cost_tag++;  // also without the keys: whatever an earlier launch left there is not this launch's
  if (h->min_cost_tag == 0xFFFFFFFFu) {  // 4e9 launches on: ~tag has run out -- all keys back to "none", tags from 1 again
    HIPCHK(h, hipMemsetAsync(h->d_min_cost, 0xFF, sizeof(unsigned long long) * kMinCostLines * kMinCostStride, stream));
    h->min_cost_tag = 1;
  }
  a.min_cost = min_cost_keys(h);
  a.min_cost_tag = h->min_cost_tag;
  return MPPI_OK;
}

int launch_rollout(mppi_handle *h, const RolloutArgs &a)
{
  // basis-function model: the two-wave form while both waves of a group get a SIMD of their own
  const Form f = form_of(h);
  hipError_t e = hipSuccess;
  switch (f) {
    case Form::Bf1: case Form::Bf2: case Form::Bf3: e = launch_rollout_bf(a, form_bf_waves(f), h->stream); break;
    case Form::Multi2: case Form::Multi4: case Form::Multi4Tree:
      e = launch_rollout_multi(h->hidden, h->n_hidden, a, form_multi_nd(f), h->stream);
      break;
    case Form::Oct: e = launch_rollout_oct(h->hidden, h->n_hidden, a, h->stream); break;
    case Form::M44: e = launch_rollout_m44(h->hidden, h->n_hidden, a, true, h->stream); break;
    case Form::M44Chain: e = launch_rollout_m44(h->hidden, h->n_hidden, a, false, h->stream); break;
    case Form::Row64R16: e = launch_rollout_row64(h->hidden, h->n_hidden, a, 16, h->stream); break;
    case Form::Row: case Form::RowTree: e = launch_rollout_row(h->hidden, h->n_hidden, a, f == Form::RowTree, h->stream); break;
    case Form::Quad: case Form::Fused64: case Form::Fused256:
      e = launch_rollout_mfma(h->hidden, h->n_hidden, a, form_fused_threads(f), h->stream);
      break;
    case Form::ValuReg: e = launch_rollout_valu_reg(h->hidden, h->n_hidden, a, h->stream); break;
    default: e = launch_rollout_valu(h->net, a, h->stream); break;
  }
  if (e != hipSuccess) return fail(h, MPPI_ERR_HIP, "rollout launch", e);
  return MPPI_OK;
}

int check_ready(mppi_handle *h)
{
  if (!h) return MPPI_ERR_INVALID;
  if (!h->have_nn) return fail(h, MPPI_ERR_STATE, "mppi_set_nn_params has not been called");
  if (!h->have_map) return fail(h, MPPI_ERR_STATE, "mppi_set_costmap has not been called");
  if (!h->have_cost) return fail(h, MPPI_ERR_STATE, "mppi_set_cost_params has not been called");
  return MPPI_OK;
}

// Stand-alone generator (mppi_generate_noise, and solves with a rollout variant that has no noise wavefront,
// or any variant while prefetched draws are waiting): *buf_out holds the draws of the next solve iteration once
// the handle's stream has passed the wait enqueued here.
int launch_generator(mppi_handle *h, float *dst)
{
  const bool timed = h->gen_time_now && !h->gen_timed;
  if (timed) HIPCHK(h, hipEventRecord(h->ev_gt[0], h->gstream));
  HIPCHK(h, launch_noise(h->d_rng[h->rng_cur], h->d_rng[1 - h->rng_cur], h->d_jump, h->K, h->T, h->noise_L,
                         h->noise_C, dst, h->gstream));
  if (timed) {
    HIPCHK(h, hipEventRecord(h->ev_gt[1], h->gstream));
    h->gen_timed = true;
  }
  h->rng_cur = 1 - h->rng_cur;
  HIPCHK(h, hipEventRecord(h->ev_gen, h->gstream));
  return MPPI_OK;
}

int acquire_noise(mppi_handle *h, float **buf_out)
{
  if (!h->gen_async) {
    // small problems: the generator on the handle's own stream, in front of the rollout.  The two event waits of
    // the asynchronous path cost ~10 us per solve, more than a generator of K T < 2^20 pairs takes
    // (basis-function build, K=2560: 86 -> 92 us per solve with it; config 4, 2.4 M pairs: 355 -> 337 us)
    h->gen_cur = 1 - h->gen_cur;
    float *dst = h->d_gen[h->gen_cur];
    HIPCHK(h, launch_noise(h->d_rng[h->rng_cur], h->d_rng[1 - h->rng_cur], h->d_jump, h->K, h->T, h->noise_L,
                           h->noise_C, dst, h->stream));
    h->rng_cur = 1 - h->rng_cur;
    *buf_out = dst;
    return MPPI_OK;
  }
  if (!h->prefetch_valid) {
    // generate now: after everything enqueued on the handle's stream so far (the buffer may still be read by an
    // earlier iteration's tail kernel, the generator states may have been written by an in-kernel generator)
    HIPCHK(h, hipEventRecord(h->ev_s1, h->stream));
    HIPCHK(h, hipStreamWaitEvent(h->gstream, h->ev_s1, 0));
    int rc = launch_generator(h, h->d_gen[1 - h->gen_cur]);
    if (rc) return rc;
  }
  h->prefetch_valid = false;
  h->gen_cur = 1 - h->gen_cur;
  HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_gen, 0));
  *buf_out = h->d_gen[h->gen_cur];
  return MPPI_OK;
}

// Does the generator kernel of the NEXT solve start at once, beside this solve's rollout (on the lowest-priority stream: it
// gets what the dynamics waves leave), or behind it, beside the tail kernel?  Beside, while it does not compete with rollout
// workgroups that still wait for a slot (profiles/r05_p_generator_beside_rollout.txt, steps in ms, behind -> beside):
//   one dynamics wave per SIMD (K <= 16 384): round 3's measurement (config 4 0.317 -> 0.309);
//   64-wide nets at any K: K = 24 576 / T = 150 0.5245 -> 0.5090, K = 32 768 0.3684 -> 0.3584, K = 65 536 0.7121 -> 0.7014 (their
//     matrix-instruction chains leave the bubbles even with four rounds of workgroups);
//   32-wide nets while every rollout workgroup is resident at once (two per CU: K <= 32 768): K = 24 576 0.1709 -> 0.1551,
//     K = 32 768 0.1757 -> 0.1610, 6-32x4-4 K = 32 768 0.3146 -> 0.2985; at K = 65 536 (two rounds) the generator's workgroups
//     take the slots the second round waits for: 0.3112 -> 0.3639, so there it starts when the rollout ends.
// MPPI_GEN_BESIDE=0 / 1 (tools only) forces either.
bool gen_beside_rollout(const mppi_handle *h)
{
  static const int forced = [] { const char *e = getenv("MPPI_GEN_BESIDE"); return e ? (atoi(e) != 0 ? 1 : 0) : -1; }();
  if (forced >= 0) return forced == 1;
  if (h->K / kRolloutsPerWave <= h->num_simds) return true;
  if (h->hidden >= 64) return true;
  return h->K / 64 <= 2 * (h->num_simds / 4);
}

// The next solve's draws, requested right after this solve's rollout went out: they start when that rollout
// ends (ev_s1) and run beside the weights / tail kernels.  Their target is the buffer of the solve BEFORE this
// one, which the host has collected.  Only for single-iteration solves of a generator-kernel form.
int prefetch_noise(mppi_handle *h)
{
  // INVARIANT the target buffer relies on: d_gen[1 - gen_cur] holds the applied controls of the solve BEFORE the
  // one just enqueued; its readers were that solve's tail kernel -- every row workgroup published its row after
  // reading it, and the host has seen all rows (wait_pending at the top of enqueue_solve) -- and calls that
  // synchronise the stream themselves (mppi_get_applied_controls, mppi_rollout_only).  No device-side ordering
  // against h->stream is needed as long as no solve is pending here; a reader that does not synchronise would
  // have to be ordered explicitly (an event after the tail kernel, waited for by gstream).
  if (h->pending) return fail(h, MPPI_ERR_STATE, "prefetch with a solve pending");
  // beside the rollout or behind it: gen_beside_rollout
  if (!gen_beside_rollout(h)) HIPCHK(h, hipStreamWaitEvent(h->gstream, h->ev_s1, 0));
  int rc = launch_generator(h, h->d_gen[1 - h->gen_cur]);
  if (rc) return rc;
  h->prefetch_valid = true;
  return MPPI_OK;
}

int upload_controls_if_dirty(mppi_handle *h, hipStream_t stream)
{
  if (!h->u_dirty) return MPPI_OK;
  memcpy(h->h_in, h->U.data(), sizeof(float) * 2 * (size_t)h->T);
  memcpy(h->h_in + 2 * h->T, h->hist.data(), sizeof(float) * 4);
  HIPCHK(h, hipMemcpyAsync(h->d_in, h->h_in, sizeof(float) * (2 * (size_t)h->T + 4), hipMemcpyHostToDevice,
                           stream));
  h->u_dirty = false;
  return MPPI_OK;
}

#ifdef MPPI_HOSTPROF
static double hp_acc[8] = {0}, hp_n = 0;
static std::chrono::steady_clock::time_point hp_seen;
#define HP(i, t0) hp_acc[i] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - (t0)).count()
#endif

// savitskyGolay (mppi_controller.cu:468-499) on the host, the same operations in the same order as the tail
// kernel applies to the device copy (this file is compiled with -ffp-contract=off):
// X = [hist0, hist1, U_0 .. U_{T-1}, U_{T-1}, U_{T-1}], U_i = sum_m f_m X_{i+m}.  Row t of the unsmoothed
// sequence is (src[stride t], src[stride t + off1]); the result goes to h->U.
void savgol_host(mppi_handle *h, const float *src, int stride, int off1)
{
  const int T = h->T;
  std::vector<float> &X = h->sg_buf;
  X.resize((size_t)(T + 4) * 2);
  for (int j = 0; j < 4; j++) X[j] = h->hist[j];
  for (int t = 0; t < T; t++) {
    X[(t + 2) * 2 + 0] = src[stride * t + 0];
    X[(t + 2) * 2 + 1] = src[stride * t + off1];
  }
  for (int r = T + 2; r < T + 4; r++)
    for (int j = 0; j < 2; j++) X[r * 2 + j] = X[(T + 1) * 2 + j];
  const float f0 = -3.0f / 35.0f, f1 = 12.0f / 35.0f, f2 = 17.0f / 35.0f;
  for (int i = 0; i < 2 * T; i++) {
    float acc = f0 * X[i];
    float p = f1 * X[i + 2];
    acc = acc + p;
    p = f2 * X[i + 4];
    acc = acc + p;
    p = f1 * X[i + 6];
    acc = acc + p;
    p = f0 * X[i + 8];
    acc = acc + p;
    h->U[i] = acc;
  }
}

// Waits for the pending solve: polls the sequence number the tail kernel publishes (system-scope
// release) in the host-mapped result block; no stream synchronise on the fast path.
// A solve whose wait ran out of time (mppi_set_wait_timeout) leaves the handle "timed out": nothing waits for that solve
// again -- every later entry returns at once -- until its device work has drained (one hipStreamQuery per call, no
// blocking); then the handle works again from the host's copies of U / hist, which the failed solve never touched.
int recover_timed_out(mppi_handle *h)
{
  if (!h->timed_out) return MPPI_OK;
  const hipError_t q = hipStreamQuery(work_stream(h));
  if (q == hipErrorNotReady) return fail(h, MPPI_ERR_HIP, "an earlier solve timed out and its device work has not finished yet");
  if (q != hipSuccess) return fail(h, MPPI_ERR_HIP, "an earlier solve timed out; hipStreamQuery", q);
  h->timed_out = false;
  return MPPI_OK;
}

int wait_pending(mppi_handle *h)
{
  if (!h->pending) return recover_timed_out(h);  // nothing to wait for; a lost solve's device work must have drained
  // The tail kernel writes T+2 entries of 16 B into host-mapped memory -- row t: [u0, seq, u1, seq], then
  // [beta, seq, eta, seq] and [trajectory cost, seq, 0, seq] -- each as one store.  An entry is complete
  // once words 1 and 3 carry this solve's sequence number (either 8-byte half may land first); the solve
  // is complete for the host once every entry is.
  const volatile unsigned *words = reinterpret_cast<const volatile unsigned *>(h->h_res);
  const int n_entries = h->T + 2;
  const auto t0 = std::chrono::steady_clock::now();
  unsigned long spins = 0;
  double next_query_s = 0.01;  // the stream is asked whether it has drained every 10 ms of waiting, never on the fast path
  int next = 0;  // entries [0, next) have been seen with the sequence number
  // the host's copies are all that can be trusted after a failed wait: the next solve uploads them again, no slid copy is
  // offered, and nothing waits for this solve a second time (recover_timed_out)
  auto give_up = [&](const char *what, bool timed_out) {
    h->pending = false;
    h->pending_timed = false;
    h->u_dirty = true;
    h->slid_valid = false;
    h->timed_out = timed_out;
    h->no_result = true;  // until the next solve completes: the result getters refuse
    return fail(h, MPPI_ERR_HIP, what);
  };
  for (;;) {
    while (next < n_entries && __atomic_load_n(words + 4 * next + 1, __ATOMIC_ACQUIRE) == h->seq &&
           __atomic_load_n(words + 4 * next + 3, __ATOMIC_ACQUIRE) == h->seq)
      next++;
    if (next == n_entries) break;
    __builtin_ia32_pause();
    if ((++spins & 0xFF) == 0) {  // the clock every 256 polls (a vDSO read, ~20 ns): the limit holds to microseconds
      const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (el > h->wait_timeout_s) return give_up("timed out waiting for the solve", true);
      if (el > next_query_s) {
        next_query_s = el + 0.01;
        if (hipStreamQuery(work_stream(h)) == hipSuccess && (__atomic_load_n(words + 4 * next + 1, __ATOMIC_ACQUIRE) != h->seq ||
                                                        __atomic_load_n(words + 4 * next + 3, __ATOMIC_ACQUIRE) != h->seq))
          return give_up("solve finished without publishing its result block", false);
      }
    }
  }
#ifdef MPPI_HOSTPROF
  hp_seen = std::chrono::steady_clock::now();
#endif
  h->pending = false;
  h->no_result = false;
  h->baseline = h->h_res[4 * h->T + 0];
  h->eta = h->h_res[4 * h->T + 2];
  h->traj_cost = h->h_res[4 * (h->T + 1) + 0];
  savgol_host(h, h->h_res, 4, 2);  // rows [u0, seq, u1, seq] of the result block
#ifdef MPPI_HOSTPROF
  HP(4, hp_seen);
#endif
  // The minimum-cost rollout has weight 1 and costs are capped (never NaN), so eta >= 1 always.
  // Anything else means a rollout wavefront gave up on a hand-over (its spin budget) and poisoned
  // its costs: report it instead of returning a NaN control sequence.
  if (!(h->eta >= 1.0f)) return fail(h, MPPI_ERR_HIP, "solve produced a non-finite normaliser (device hand-over failed)");
  if (tail_is_stream(h->K)) {
    // the one-launch tail of many-chunk solves: a row workgroup whose wait for another workgroup's granule ran out of time
    // published NaN in place of what it waited for
    bool finite = std::isfinite(h->traj_cost);
    for (size_t i = 0; i < h->U.size() && finite; i++) finite = std::isfinite(h->U[i]);
    if (!finite) return fail(h, MPPI_ERR_HIP, "solve produced non-finite controls (device hand-over failed)");
  }
  if (h->pending_timed) {
    h->pending_timed = false;
    for (size_t it = 0; it < h->ev.size(); it++) {
      HIPCHK(h, hipEventSynchronize(h->ev[it].e[3]));
      float ms[3] = {0, 0, 0};
      for (int i = 0; i < 3; i++) (void)hipEventElapsedTime(&ms[i], h->ev[it].e[i], h->ev[it].e[i + 1]);
      // the rollout stage: the kernel's own dispatch stamps where the runtime delivered them
      float kms = 0.0f;
      if (hipEventElapsedTime(&kms, h->ev[it].e[4], h->ev[it].e[5]) == hipSuccess && kms > 0.0f) {
        ms[1] = kms;
      }
      h->acc.noise_ms += ms[0];
      h->acc.rollout_ms += ms[1];
      h->acc.reduction_ms += ms[2];
      h->acc.total_ms += ms[0] + ms[1] + ms[2];
    }
    if (h->gen_timed) {  // the generator launch enqueued during this solve (on gstream, beside the rollout / tail)
      h->gen_timed = false;
      float gms = 0.0f;
      HIPCHK(h, hipEventSynchronize(h->ev_gt[1]));
      if (hipEventElapsedTime(&gms, h->ev_gt[0], h->ev_gt[1]) == hipSuccess && gms > 0.0f) h->acc.noise_ms += gms;
    }
    h->acc.n_solves += 1;
  }
  return MPPI_OK;
}

// the tail kernel of the last iteration leaves [U | hist] slid by the optimization stride in the other buffer
bool wants_slid_copy(const mppi_handle *h)
{
  return h->cfg.optimization_stride >= 1 && h->cfg.optimization_stride < h->T;
}

TailLaunch tail_launch(const mppi_handle *h, const float *V, bool last)
{
  TailLaunch l;
  l.costs = h->d_costs; l.V = V; l.U = h->d_in; l.hist = h->d_in + 2 * h->T; l.w = h->d_w; l.scal = h->d_scal;
  l.res = h->d_res_map; l.counter = h->d_counter;
  l.K = h->K; l.T = h->T; l.gamma = h->cfg.gamma; l.last_iter = last ? 1 : 0; l.seq = h->seq;
  l.slid = (last && wants_slid_copy(h)) ? h->d_in_buf[1 - h->in_cur] : nullptr;
  l.slide_stride = h->cfg.optimization_stride;
  l.init0 = h->cfg.init_control[0]; l.init1 = h->cfg.init_control[1];
  // the one-launch streaming tail (K > 4096): granule buffers, this launch's tag (the caller advanced it), wait deadline; the
  // tests' fault roles 32-34 shorten the deadline to spin_budget x 1 us
  l.ug = h->d_ug;
  l.gx = h->d_gx; l.gpart = reinterpret_cast<unsigned long long *>(h->d_part);
  l.epoch = h->tail_epoch;
  l.fault = h->fault_wave >= 32 ? h->fault_wave : 0;
  l.poll_ticks = (l.fault && h->spin_budget > 0) ? (unsigned)h->spin_budget * 100u : h->tail_poll_ticks;
  l.min_cost = min_cost_keys(h);
  l.min_cost_tag = h->min_cost_tag;
  return l;
}

int enqueue_solve(mppi_handle *h, const float *state)
{
#ifdef MPPI_HOSTPROF
  const auto hp_t0 = std::chrono::steady_clock::now();
  if (hp_n > 0) hp_acc[0] += std::chrono::duration<double, std::micro>(hp_t0 - hp_seen).count();  // seen -> next enqueue entered
#endif
  int rc = check_ready(h);
  if (rc) return rc;
  if (!state) return fail(h, MPPI_ERR_INVALID, "state is NULL");
  rc = recover_timed_out(h);
  if (rc) return rc;
  rc = wait_pending(h);  // finish a previous asynchronous solve first
  if (rc) return rc;
  const int K = h->K, T = h->T, iters = h->cfg.num_iters;
  if (h->explicit_iters > 0 && h->explicit_iters != iters)
    return fail(h, MPPI_ERR_STATE, "explicit noise holds a different number of iterations");
  OWN(h);
  rc = upload_controls_if_dirty(h, h->stream);
  if (rc) return rc;
  const bool timed = h->timing && (h->timing_count++ % (unsigned)h->timing_every) == 0;
  const bool explicit_noise = h->explicit_iters > 0;
  const size_t slot_sz = (size_t)K * T * 2;
  h->seq++;
  for (int it = 0; it < iters; it++) {
    Events *ev = timed ? &h->ev[it] : nullptr;
    if (ev) HIPCHK(h, hipEventRecord(ev->e[0], h->stream));
    // source of eps: the explicit buffer (mppi_set_noise) > draws already prefetched > the rollout kernel's own
    // noise wavefront > the generator kernel, now
    const bool inline_noise = !explicit_noise && !h->prefetch_valid && has_noise_wave(h);
    float *noise = h->d_noise + (size_t)(explicit_noise ? it : 0) * slot_sz;
    if (!explicit_noise && !inline_noise) {
      rc = acquire_noise(h, &noise);
      if (rc) return rc;
    } else if (inline_noise) {
      noise = h->d_gen[h->gen_cur];  // receives the applied controls
    }
    h->v_buf = noise;
    if (ev) HIPCHK(h, hipEventRecord(ev->e[1], h->stream));
    RolloutArgs a;
    fill_rollout_args(h, state, noise, a);
    rc = tag_min_cost(h, a, h->stream);
    if (rc) return rc;
    if (inline_noise) {  // the rollout kernel's noise wavefront draws eps itself
      a.inline_noise = 1;
      a.rng_in = h->d_rng[h->rng_cur];
      a.rng_out = h->d_rng[1 - h->rng_cur];
      h->rng_cur = 1 - h->rng_cur;
    }
#ifdef MPPI_HOSTPROF
    HP(1, hp_t0);  // entry -> before the rollout launch
    const auto hp_t1 = std::chrono::steady_clock::now();
#endif
    if (ev) { tl_kernel_start = ev->e[4]; tl_kernel_stop = ev->e[5]; }
    rc = launch_rollout(h, a);
    tl_kernel_start = tl_kernel_stop = nullptr;
    if (rc) return rc;
#ifdef MPPI_HOSTPROF
    HP(2, hp_t1);  // the rollout launch call
    const auto hp_t2 = std::chrono::steady_clock::now();
#endif
    const bool prefetch = h->gen_async && iters == 1 && !explicit_noise && !has_noise_wave(h) && !h->prefetch_valid;
    if (prefetch) HIPCHK(h, hipEventRecord(h->ev_s1, h->stream));  // the generator starts when this rollout ends
    if (ev) HIPCHK(h, hipEventRecord(ev->e[2], h->stream));
    const bool last = (it == iters - 1);
    const bool want_slid = last && wants_slid_copy(h);
    if (++h->tail_epoch == 0) h->tail_epoch = 1;  // the tag of this launch's granules
    HIPCHK(h, launch_solve_tail(tail_launch(h, noise, last), h->stream));
    if (last) h->slid_valid = want_slid;
    if (h->capture) {  // test hook: what this iteration left (the last iteration's raw U is in the result block)
      float *c = h->d_cap + (size_t)it * (2 * (size_t)T + K);
      if (!last) HIPCHK(h, hipMemcpyAsync(c, h->d_in, sizeof(float) * 2 * (size_t)T, hipMemcpyDeviceToDevice, h->stream));
      HIPCHK(h, hipMemcpyAsync(c + 2 * (size_t)T, h->d_costs, sizeof(float) * (size_t)K, hipMemcpyDeviceToDevice, h->stream));
    }
    if (prefetch) {
      h->gen_time_now = timed;  // only the prefetch launch: a generator the stream waits for sits between e[0] and e[1]
      rc = prefetch_noise(h);
      h->gen_time_now = false;
      if (rc) return rc;
    }
#ifdef MPPI_HOSTPROF
    HP(3, hp_t2);  // the tail launch call
    hp_n += 1;
    if ((long)hp_n % 2000 == 0)
      fprintf(stderr, "hostprof n=%.0f: seen->enqueue %.2f us, entry->launch %.2f, rollout launch %.2f, tail launch %.2f, poll->smoothed %.2f\n",
              hp_n, hp_acc[0] / hp_n, hp_acc[1] / hp_n, hp_acc[2] / hp_n, hp_acc[3] / hp_n, hp_acc[4] / hp_n);
#endif
    if (ev) HIPCHK(h, hipEventRecord(ev->e[3], h->stream));
  }
  h->explicit_iters = 0;
  h->pending = true;
  h->pending_timed = timed;
  h->cap_valid = h->capture;
  h->cap_explicit = explicit_noise;
  return MPPI_OK;
}

// ---- chained control ticks -------------------------------------------------------------------------------------------------
// mppi_control_ticks knows that solve i+1 follows solve i at once.  Its launches -- 2.8 us of launch call and 1.5 us of
// dispatch for the rollout kernel alone, on the step's critical path between "result i on the host" and "first instruction of
// rollout i+1" -- are therefore made one tick AHEAD, while the host would otherwise only poll for result i: rollout i+1 (the
// gated form of the row kernel, rollout_row.hip) and tail i+1 go onto the handle's stream behind tail i, with the buffers
// solve i+1 will own after the slide (U: the slid copy tail i leaves; its own slid copy: the buffer solve i read).  The
// rollout starts as soon as tail i has ended, loads its weights, draws its first noise -- and waits for the host's gate: the
// state of solve i+1 and its sequence number, written into the gate block AFTER the host has seen, smoothed and slid result
// i.  The step keeps its meaning: solve i+1 computes nothing from the state before the host has result i in hand.
// tools/ub/gate_ub.hip (profiles/r05_c_gate_ub.txt): 46.3 -> 43.0 us per step for stand-in kernels of the headline's length
// with the gate in device memory written through the BAR, 44.5 with a host-mapped gate; two streams with the next rollout
// resident beside the tail: 54 (the cross-stream events cost more than the launches they hide).
// Results are bit for bit those of the unchained loop (tests/test_api_gpu.py).  Only for the cases that gain: one handle, a
// latency form with riders (the row form; the automatic m44 form of 64-wide nets) and its in-kernel generator, or the
// automatic multi4-tree form with its prefetched generator kernel; one iteration, no stage events, no capture, stride =
// optimization stride.
// 1: a latency form with riders and its in-kernel generator (the row form, the automatic m44 form); 2: the automatic multi4-tree
// form with the stand-alone generator kernel prefetched on a second stream; 0: not chained
static int chain_kind(const mppi_handle *h, int n_ticks, int stride)
{
  if (!(h->chain && n_ticks >= 2 && h->d_gate != nullptr && stride == h->cfg.optimization_stride && wants_slid_copy(h) &&
        h->cfg.num_iters == 1 && !h->timing && !h->capture && h->explicit_iters == 0 && !h->basis && h->fault_wave == 0 &&
        h->have_nn && h->have_map && h->have_cost && !h->timed_out))
    return 0;
  const Form f = form_of(h);
  if ((form_is_row(f) || f == Form::M44) && has_noise_wave(h) && !h->prefetch_valid) return 1;
  // (where the generator kernel runs beside the rollout.  Where it runs behind it -- 32-wide nets at K = 65 536 -- the phase
  // between two rollouts is the generator's own 41 us whatever the launches cost: chained 0.3139, unchained 0.3131 ms)
  if (f == Form::Multi4Tree && h->forced == Form::Auto && !has_noise_wave(h) && h->gen_async && gen_beside_rollout(h)) return 2;
  return 0;
}

// the gate block of solve `word`: its nominal sequence and history (the host's copies, smoothed and slid), its state in every
// replica, then (fenced) the gate word in every replica
static void write_gate(mppi_handle *h, const float *state, unsigned word)
{
  // (memcpy: wide stores -- the block is write-combining memory behind the PCIe BAR where gate_bar is set; the fences order the
  // payload before the gate words and push both out)
  unsigned *g = h->gate_cpu;
  const int T = h->T;
  memcpy(g + kGateUOffset, h->U.data(), sizeof(float) * 2 * (size_t)T);
  memcpy(g + gate_hist_offset(T), h->hist.data(), sizeof(float) * 4);
  for (int r = 0; r < kGateReplicas; r++) memcpy(g + 16 * r, state, sizeof(float) * kStateDim);
  asm volatile("" ::: "memory");
  __builtin_ia32_sfence();
  for (int r = 0; r < kGateReplicas; r++) __atomic_store_n(g + 16 * r + 7, word, __ATOMIC_RELAXED);
  __builtin_ia32_sfence();
}

// solve (h->seq + 1), gated, behind the pending solve h->seq.  Its rollout takes state AND nominal sequence from the gate
// block (the host has both in hand when it opens the gate: U smoothed and slid, as the reference uploads U_ with every
// computeControl, mppi_controller.cu:608-610).  Its tail: inside the chain only the publication (no workgroup smooths a device
// copy nobody reads: the kernel ends 2.5 us earlier, and the next rollout starts when it ends); the LAST solve of the chain
// smooths and leaves the slid copy as every ordinary solve does, with hist from the gate block, so that the handle's device
// state after the chain is the unchained loop's.
static int enqueue_ahead(mppi_handle *h, const float *state, bool last_of_chain, int kind)
{
  float *noise = h->d_gen[h->gen_cur];  // kind 1: the in-kernel generator's solves all leave their applied controls here
  int rc = MPPI_OK;
  if (kind == 2) {
    // The generator-kernel forms: this solve's eps were prefetched on gstream (into the other buffer) when the host opened the
    // gate of the solve before.  The NEXT prefetch is NOT enqueued ahead: it writes the buffer the pending solve's tail kernel
    // still reads, and -- measured -- a generator launch that becomes ready together with the rollout (both behind the same
    // tail kernel) takes the CUs first and costs its whole stand-alone time (config 4: 0.2651 -> 0.2834 ms, K = 16 384: 0.0978 ->
    // 0.1071; behind the rollout: 0.2837 / 0.1143; lowest stream priority for the generator: 0.2781).  The host launches it
    // when it opens this solve's gate (control_ticks_chained): a few microseconds behind the rollout's start, where it has
    // always been, filling the dynamics waves' bubbles.
    if (!h->prefetch_valid) return fail(h, MPPI_ERR_STATE, "chained ticks: no prefetched draws");
    rc = acquire_noise(h, &noise);
    if (rc) return rc;
  }
  RolloutArgs a;
  fill_rollout_args(h, state, noise, a);
  rc = tag_min_cost(h, a, h->stream);
  if (rc) return rc;
  a.U = reinterpret_cast<const float *>(h->d_gate) + kGateUOffset;
  if (kind == 1) {
    a.inline_noise = 1;
    a.rng_in = h->d_rng[h->rng_cur];
    a.rng_out = h->d_rng[1 - h->rng_cur];
    h->rng_cur = 1 - h->rng_cur;
  }
  a.gate = h->d_gate;
  a.gate_seq = h->seq + 1;
  rc = launch_rollout(h, a);
  if (rc) return rc;
  if (kind == 2) HIPCHK(h, hipEventRecord(h->ev_s1, h->stream));
  if (++h->tail_epoch == 0) h->tail_epoch = 1;  // the tag of this tail launch's granules (many-chunk solves)
  TailLaunch l = tail_launch(h, noise, true);
  l.seq = h->seq + 1;
  l.hist = reinterpret_cast<const float *>(h->d_gate) + gate_hist_offset(h->T);
  if (last_of_chain) {
    l.U = h->d_in;
    l.hist_out = h->d_in + 2 * h->T;
    l.slid = h->d_in_buf[1 - h->in_cur];
  } else {
    l.no_device_copy = 1;
    l.slid = nullptr;
  }
  HIPCHK(h, launch_solve_tail(l, h->stream));
  h->ahead = true;
  h->ahead_vbuf = noise;
  return MPPI_OK;
}

// the solve enqueued ahead is called off: its gate opens with the cancel bit (the kernels run through, poisoned); nothing on
// the device can be trusted afterwards -- the host copies are uploaded again by the next solve
static void cancel_ahead(mppi_handle *h, const float *state)
{
  write_gate(h, state, (h->seq + 1) | kGateCancel);
  (void)hipStreamSynchronize(h->stream);
  h->seq++;  // the called-off solve's tail kernel has published (poisoned) entries under that number: it is spent
  h->ahead = false;
  h->u_dirty = true;
  h->slid_valid = false;
}

static int control_ticks_chained(mppi_handle *h, const float *state, int n_ticks, int stride, int kind)
{
  HIPCHK(h, ensure_device(h->cfg.device));
  int rc = enqueue_solve(h, state);
  if (rc) return rc;
  for (int i = 0; i < n_ticks; i++) {
    const bool ahead = i + 1 < n_ticks, ahead_is_last = i + 2 == n_ticks;
    if (ahead) {
      rc = enqueue_ahead(h, state, ahead_is_last, kind);
      if (rc) {
        (void)wait_pending(h);
        return rc;
      }
    }
    rc = wait_pending(h);
    if (rc == MPPI_OK) {
      // inside the chain the device copy of U is not kept up (the next solve reads the host's through the gate block): the
      // slide is the host's alone; the last solve's tail restores it, and the slide behind it swaps to its slid copy as usual
      if (ahead) { h->u_dirty = true; h->slid_valid = false; }
      rc = mppi_slide_control_seq(h, stride);
    }
    if (rc) {
      if (ahead) cancel_ahead(h, state);
      return rc;
    }
    if (ahead) {  // the host has result i, smoothed and slid: solve i+1 may start
      h->seq++;
      write_gate(h, state, h->seq);
      if (kind == 2) {  // the draws of the solve after it, behind the opened gate (enqueue_solve: prefetch_noise behind the tail launch)
        rc = prefetch_noise(h);
        if (rc) return rc;
      }
      h->ahead = false;
      h->pending = true;
      h->pending_timed = false;
      h->v_buf = h->ahead_vbuf;
      if (ahead_is_last) { h->u_dirty = false; h->slid_valid = true; }  // its tail smooths h->d_in and leaves the slid copy
    }
  }
  return MPPI_OK;
}

}  // namespace mppi_abi

extern "C" {

int mppi_compute_control_async(mppi_handle *h, const float state[MPPI_STATE_DIM])
{
  if (!h) return MPPI_ERR_INVALID;
  HIPCHK(h, ensure_device(h->cfg.device));
  return enqueue_solve(h, state);
}

int mppi_synchronize(mppi_handle *h)
{
  if (!h) return MPPI_ERR_INVALID;
  return wait_pending(h);
}

int mppi_compute_control(mppi_handle *h, const float state[MPPI_STATE_DIM])
{
  int rc = mppi_compute_control_async(h, state);
  if (rc) return rc;
  return mppi_synchronize(h);
}

int mppi_compute_control_batch_async(mppi_handle *const *hs, const float *states, int n)
{
  if (!hs || !states || n < 1) return MPPI_ERR_INVALID;
#ifdef MPPI_HOSTPROF
  static double hb[8] = {0}; static double hbn = 0; static std::chrono::steady_clock::time_point hb_prev_end;
  auto hb_t = std::chrono::steady_clock::now();
  if (hbn > 0) hb[0] += std::chrono::duration<double, std::micro>(hb_t - hb_prev_end).count();  // previous batch_async returned -> this one entered (synchronize + slides of the caller)
  // (the first call creates the device's batch stream, ~3 ms: read these averages over many calls with that in mind)
#define HB(i) do { auto n__ = std::chrono::steady_clock::now(); hb[i] += std::chrono::duration<double, std::micro>(n__ - hb_t).count(); hb_t = n__; } while (0)
#else
#define HB(i) do { } while (0)
#endif
  for (int i = 0; i < n; i++) {
    if (!hs[i]) return MPPI_ERR_INVALID;
    for (int q = 0; q < i; q++)
      if (hs[q] == hs[i]) return fail(hs[i], MPPI_ERR_INVALID, "the same handle twice in one batch");
  }
  // the host-side halves of this tick (nominal replays, DDP passes) are a solve's length away: the helper thread wakes now
  if (n == 2 && g_host_threads.load(std::memory_order_relaxed) >= 2) host_helper_arm();
  // One launch for all instances where the quad form serves them together (every wave of every group still gets a
  // SIMD of its own: the sum of the groups fits the CUs); otherwise every solve on its own handle's stream, as
  // n calls of mppi_compute_control_async would do.
  mppi_handle *h0 = hs[0];
  bool together = n >= 2 && n <= kMaxBatch;
  int waves = 0;
  for (int i = 0; i < n && together; i++) {
    const mppi_handle *h = hs[i];
    // network model: the four-wavefront form; basis-function model: its three-wavefront form (in-kernel generator)
    const Form f = form_of(h);
    const bool form_ok = h->basis ? (h0->basis && f == Form::Bf3)
                                  : (!h0->basis && (f == Form::Quad || form_is_row(f)) && f == form_of(h0) &&
                                     h->hidden == h0->hidden && h->n_hidden == h0->n_hidden);
    // one kernel instance serves the whole batch: the instances must agree on what it is specialised for (affine / projective
    // costmap transform, control cost or none) -- a superset kernel would compute 0 * x where the single solve computes
    // nothing, which differs for a non-finite x; such a pair is solved one by one
    CostArgs ci, c0;
    fill_cost_args(h, ci);
    fill_cost_args(h0, c0);
    together = form_ok && ci.affine == c0.affine && ci.need_control_cost == c0.need_control_cost &&
               h->cfg.device == h0->cfg.device && h->cfg.num_iters == h0->cfg.num_iters &&
               h->K <= 4096 && !h->timing && !h->capture && !h->prefetch_valid && h->have_nn && h->have_map && h->have_cost;
    // waves of a group that need a SIMD each: quad 4, row 4 dynamics waves (its riders ride), basis functions 3
    waves += h->basis ? 3 * (h->K / 64) : 4 * (h->K / kRolloutsPerWave);
  }
  together = together && waves <= h0->num_simds;  // every wave of every group still gets a SIMD of its own
  if (!together) {
    for (int i = 0; i < n; i++) {
      const int rc = mppi_compute_control_async(hs[i], states + (size_t)MPPI_STATE_DIM * i);
      if (rc) return rc;
    }
    return MPPI_OK;
  }
  HB(1);  // the decision: one launch for all?
  // (the device's batch stream is looked up once per handle)
  HIPCHK(h0, ensure_device(h0->cfg.device));
  if (!h0->batch_s) h0->batch_s = batch_stream(h0->cfg.device);
  const hipStream_t S = h0->batch_s;
  if (!S) return fail(h0, MPPI_ERR_HIP, "no batch stream");
  const int iters = h0->cfg.num_iters;
  HB(2);  // current device + the batch stream
  // First pass: everything that can fail without having touched a handle -- every handle's previous solve collected,
  // every handle's explicit noise checked -- so that one handle's error does not leave its partners half-advanced.
  for (int i = 0; i < n; i++) {
    mppi_handle *h = hs[i];
    int rc = recover_timed_out(h);
    if (rc == MPPI_OK) rc = wait_pending(h);  // finish a previous asynchronous solve first
    if (rc) return rc;
    if (h->explicit_iters > 0 && h->explicit_iters != iters)
      return fail(h, MPPI_ERR_STATE, "explicit noise holds a different number of iterations");
  }
  // From here on handles change (sequence numbers, generator buffers, the slid copy's validity).  A HIP error below
  // leaves every handle of the batch in the state "nothing on the device can be trusted": the host copies of U / hist
  // are uploaded again by the next solve, no slid copy is offered to mppi_slide_control_seq.
  auto poison = [&](int rc_) {
    for (int i = 0; i < n; i++) {
      hs[i]->slid_valid = false;
      hs[i]->u_dirty = true;
      hs[i]->pending = false;
    }
    return rc_;
  };
  HB(3);  // wait_pending of every handle
  for (int i = 0; i < n; i++) {
    mppi_handle *h = hs[i];
    int rc = MPPI_OK;
    if (h->order_stream != S) {  // first batched solve after work on the handle's own streams: let that finish
      hipError_t e = hipStreamSynchronize(h->order_stream ? h->order_stream : h->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(h->gstream);
      if (e != hipSuccess) return poison(fail(h, MPPI_ERR_HIP, "batched solve: stream hand-over", e));
      h->order_stream = S;
    }
    rc = upload_controls_if_dirty(h, S);
    if (rc) return poison(rc);
    h->seq++;
  }
  for (int it = 0; it < iters; it++) {
    QuadBatchArgs qb;
    TailLaunch tl[kMaxBatch];
    qb.n = n;
    const bool last = (it == iters - 1);
    for (int i = 0; i < n; i++) {
      mppi_handle *h = hs[i];
      const bool explicit_noise = h->explicit_iters > 0;
      // eps: the explicit buffer, else the control wavefront's own generator; the buffer receives the applied controls
      float *noise = explicit_noise ? h->d_noise + (size_t)it * ((size_t)h->K * h->T * 2) : h->d_gen[h->gen_cur];
      h->v_buf = noise;
      RolloutArgs &a = qb.inst[i];
      fill_rollout_args(h, states + (size_t)MPPI_STATE_DIM * i, noise, a);
      if (int trc = tag_min_cost(h, a, S)) return trc;
      if (!explicit_noise) {
        a.inline_noise = 1;
        a.rng_in = h->d_rng[h->rng_cur];
        a.rng_out = h->d_rng[1 - h->rng_cur];
        h->rng_cur = 1 - h->rng_cur;
      }
      tl[i] = tail_launch(h, noise, last);
      if (last) h->slid_valid = wants_slid_copy(h);
    }
    for (int i = n; i < kMaxBatch; i++) qb.inst[i] = qb.inst[0];
    HB(4);  // uploads + argument blocks
    hipError_t e = h0->basis ? launch_rollout_bf_batch(qb, S)
                   : form_is_row(form_of(h0)) ? launch_rollout_row_batch(qb, form_of(h0) == Form::RowTree, S)
                                                : launch_rollout_quad_batch(h0->hidden, h0->n_hidden, qb, S);
    HB(5);  // rollout launch
    if (e == hipSuccess) e = launch_solve_tail_batch(tl, n, S);
    HB(6);  // tail launch
    if (e != hipSuccess) return poison(fail(h0, MPPI_ERR_HIP, "batched launch", e));
  }
  for (int i = 0; i < n; i++) {
    hs[i]->explicit_iters = 0;
    hs[i]->pending = true;
    hs[i]->pending_timed = false;
  }
#ifdef MPPI_HOSTPROF
  hbn += 1;
  hb_prev_end = std::chrono::steady_clock::now();
  if (((long)hbn % 1000) == 0)
    fprintf(stderr, "hostprof batch (us per call, %.0f calls): between calls %.2f | decide %.2f | device + stream %.2f | wait_pending %.2f | upload+args %.2f | rollout launch %.2f | tail launch %.2f\n",
            hbn, hb[0] / hbn, hb[1] / hbn, hb[2] / hbn, hb[3] / hbn, hb[4] / hbn, hb[5] / hbn, hb[6] / hbn);
#endif
  return MPPI_OK;
}

int mppi_compute_control_batch(mppi_handle *const *hs, const float *states, int n)
{
  int rc = mppi_compute_control_batch_async(hs, states, n);
  for (int i = 0; i < n && rc == MPPI_OK; i++) rc = mppi_synchronize(hs[i]);
  return rc;
}

int mppi_control_ticks_batch(mppi_handle *const *hs, const float *states, int n, int n_ticks, int stride)
{
  if (!hs || !states || n < 1 || n_ticks < 0 || stride < 0) return MPPI_ERR_INVALID;
  for (int t = 0; t < n_ticks; t++) {
    int rc = mppi_compute_control_batch(hs, states, n);
    if (rc) return rc;
    for (int i = 0; i < n && stride > 0; i++) {
      rc = mppi_slide_control_seq(hs[i], stride);
      if (rc) return rc;
    }
  }
  return MPPI_OK;
}

int mppi_control_ticks(mppi_handle *h, const float state[MPPI_STATE_DIM], int n_ticks, int stride)
{
  if (!h || n_ticks < 0 || stride < 0) return MPPI_ERR_INVALID;
  const int kind = state ? chain_kind(h, n_ticks, stride) : 0;
  if (kind) return control_ticks_chained(h, state, n_ticks, stride, kind);
  for (int i = 0; i < n_ticks; i++) {
    int rc = mppi_compute_control(h, state);
    if (rc) return rc;
    if (stride > 0) {
      rc = mppi_slide_control_seq(h, stride);
      if (rc) return rc;
    }
  }
  return MPPI_OK;
}

int mppi_get_results(mppi_handle *h, float *U, float *traj_cost, float *costs, float *weights)
{
  if (!h) return MPPI_ERR_INVALID;
  int rc = mppi_synchronize(h);
  if (rc) return rc;
  if (h->no_result) return fail(h, MPPI_ERR_HIP, "the last solve timed out: no result");
  if (U) memcpy(U, h->U.data(), sizeof(float) * 2 * (size_t)h->T);
  if (traj_cost) *traj_cost = h->traj_cost;
  if (costs || weights) {
    HIPCHK(h, hipSetDevice(h->cfg.device));
    OWN(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
  }
  if (costs) HIPCHK(h, hipMemcpy(costs, h->d_costs, sizeof(float) * h->K, hipMemcpyDeviceToHost));
  if (weights) HIPCHK(h, hipMemcpy(weights, h->d_w, sizeof(float) * h->K, hipMemcpyDeviceToHost));
  return MPPI_OK;
}

int mppi_get_applied_controls(mppi_handle *h, float *V, size_t n)
{
  if (!h || !V) return MPPI_ERR_INVALID;
  const size_t slot = (size_t)h->K * h->T * 2;
  if (n != slot) return fail(h, MPPI_ERR_INVALID, "n != K*T*2");
  int rc = mppi_synchronize(h);
  if (rc) return rc;
  if (h->no_result) return fail(h, MPPI_ERR_HIP, "the last solve timed out: no result");
  HIPCHK(h, hipSetDevice(h->cfg.device));
  OWN(h);
  HIPCHK(h, launch_tk_to_kt(h->v_buf, h->d_stage, h->K, h->T, h->stream));
  HIPCHK(h, hipMemcpyAsync(V, h->d_stage, slot * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return MPPI_OK;
}

int mppi_rollout_only(mppi_handle *h, const float state[MPPI_STATE_DIM], float *costs)
{
  int rc = check_ready(h);
  if (rc) return rc;
  if (!state || !costs) return fail(h, MPPI_ERR_INVALID, "NULL argument");
  HIPCHK(h, hipSetDevice(h->cfg.device));
  rc = recover_timed_out(h);
  if (rc) return rc;
  rc = mppi_synchronize(h);
  if (rc) return rc;
  OWN(h);
  rc = upload_controls_if_dirty(h, h->stream);
  if (rc) return rc;
  const bool explicit_noise = h->explicit_iters > 0;
  const bool inline_noise = !explicit_noise && !h->prefetch_valid && has_noise_wave(h);
  float *noise = h->d_noise;  // explicit: its first iteration
  if (!explicit_noise && !inline_noise) {
    rc = acquire_noise(h, &noise);
    if (rc) return rc;
  } else if (inline_noise) {
    noise = h->d_gen[h->gen_cur];
  }
  h->explicit_iters = 0;
  h->v_buf = noise;
  RolloutArgs a;
  fill_rollout_args(h, state, noise, a);
  if (inline_noise) {
    a.inline_noise = 1;
    a.rng_in = h->d_rng[h->rng_cur];
    a.rng_out = h->d_rng[1 - h->rng_cur];
    h->rng_cur = 1 - h->rng_cur;
  }
  rc = launch_rollout(h, a);
  if (rc) return rc;
  HIPCHK(h, hipMemcpyAsync(costs, h->d_costs, sizeof(float) * h->K, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return MPPI_OK;
}

/* Debug/test entries (not part of the drop-in surface): what every iteration of a multi-iteration solve left, so that
 * a test can hold iteration i against the oracle started from the SAME U (mppi_controller.cu:609-667: the loop re-uses
 * U_ without smoothing in between).  Capturing adds two small device copies per iteration and keeps the handle out of
 * batched launches. */
int mppi_debug_capture_iterations(mppi_handle *h, int on)
{
  if (!h) return MPPI_ERR_INVALID;
  int rc = mppi_synchronize(h);
  if (rc) return rc;
  if (on && !h->d_cap) {
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipMalloc(&h->d_cap, sizeof(float) * (size_t)h->cfg.num_iters * (2 * (size_t)h->T + h->K)));
  }
  h->capture = on != 0;
  h->cap_valid = false;
  return MPPI_OK;
}

int mppi_debug_get_iterations(mppi_handle *h, float *U_raw, float *costs, float *V)
{
  if (!h) return MPPI_ERR_INVALID;
  int rc = mppi_synchronize(h);
  if (rc) return rc;
  if (!h->cap_valid) return fail(h, MPPI_ERR_STATE, "no captured solve (mppi_debug_capture_iterations, then a solve on this handle alone)");
  if (V && !h->cap_explicit) return fail(h, MPPI_ERR_STATE, "applied controls of every iteration exist for explicit-noise solves only");
  const int iters = h->cfg.num_iters, T = h->T, K = h->K;
  HIPCHK(h, hipSetDevice(h->cfg.device));
  OWN(h);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  const size_t rec = 2 * (size_t)T + K;
  std::vector<float> buf((size_t)iters * rec);
  HIPCHK(h, hipMemcpy(buf.data(), h->d_cap, sizeof(float) * buf.size(), hipMemcpyDeviceToHost));
  for (int it = 0; it < iters; it++) {
    if (U_raw) {
      float *u = U_raw + (size_t)it * 2 * T;
      if (it < iters - 1) memcpy(u, buf.data() + (size_t)it * rec, sizeof(float) * 2 * (size_t)T);
      else for (int t = 0; t < T; t++) { u[2 * t] = h->h_res[4 * t]; u[2 * t + 1] = h->h_res[4 * t + 2]; }  // rows [u0, seq, u1, seq]
    }
    if (costs) memcpy(costs + (size_t)it * K, buf.data() + (size_t)it * rec + 2 * (size_t)T, sizeof(float) * (size_t)K);
    if (V) {
      const size_t slot = (size_t)K * T * 2;
      HIPCHK(h, launch_tk_to_kt(h->d_noise + (size_t)it * slot, h->d_stage, K, T, h->stream));
      HIPCHK(h, hipMemcpyAsync(V + (size_t)it * slot, h->d_stage, slot * sizeof(float), hipMemcpyDeviceToHost, h->stream));
      HIPCHK(h, hipStreamSynchronize(h->stream));
    }
  }
  return MPPI_OK;
}

/* How long a blocking call polls for a solve's result block before it reports MPPI_ERR_HIP (default 30 s).  The clock is
 * read every 256 polls: the limit holds to microseconds.  After a timeout the handle is "timed out": the host copies of U and
 * the control history are what the next solve starts from, no call waits for the lost solve again (each returns MPPI_ERR_HIP at
 * once while its device work is still running, one hipStreamQuery per call) and the handle works again once that work has
 * drained.  mppi_destroy synchronises the handle's streams and may block for as long as that work runs. */
int mppi_set_wait_timeout(mppi_handle *h, double seconds)
{
  if (!h || !(seconds > 0.0)) return MPPI_ERR_INVALID;
  h->wait_timeout_s = seconds;
  return MPPI_OK;
}

}  // extern "C"
