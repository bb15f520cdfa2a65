// mppi_abi.hip -- host side of libmppi_hip.so: the C ABI of include/mppi_hip.h.
//
// Owns one HIP stream and all device memory of a solver instance, enqueues one MPPI solve
// (PI/mppi_controller.cu:600-671) as: [H2D U|hist] -> noise -> rollout -> weights -> weighted
// reduction -> Savitzky-Golay -> [D2H scal|U], with ONE stream synchronisation per solve (the
// reference has three plus five blocking parameter uploads).
// There is no CPU fallback anywhere in this file: without a gfx950 device every compute entry
// point returns an error.
#include "../../include/mppi_hip.h"

#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <string>
#include <vector>

#include "mppi_kernels.hpp"
#include "ddp_feedback.hpp"
#include "basis_funcs.hpp"
#include "host_net.hpp"

using namespace mppi;

namespace {

constexpr uint64_t M1 = 4294967087ULL, M2 = 4294944443ULL;

struct Mat3 {
  uint32_t a[9];
};
Mat3 mat_mul(const Mat3 &A, const Mat3 &B, uint64_t m)
{
  Mat3 R;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      uint64_t acc = 0;
      for (int k = 0; k < 3; k++) acc = (acc + (uint64_t)A.a[3 * i + k] * B.a[3 * k + j] % m) % m;
      R.a[3 * i + j] = (uint32_t)acc;
    }
  return R;
}
Mat3 mat_identity()
{
  Mat3 R{{1, 0, 0, 0, 1, 0, 0, 0, 1}};
  return R;
}
Mat3 mat_pow(Mat3 A, uint64_t e, uint64_t m)
{
  Mat3 R = mat_identity();
  while (e) {
    if (e & 1) R = mat_mul(R, A, m);
    A = mat_mul(A, A, m);
    e >>= 1;
  }
  return R;
}
Mat3 base_A1() { return Mat3{{0, 1, 0, 0, 0, 1, (uint32_t)(M1 - 810728ULL), 1403580u, 0}}; }
Mat3 base_A2() { return Mat3{{0, 1, 0, 0, 0, 1, (uint32_t)(M2 - 1370589ULL), 0, 527612u}}; }

struct Events {
  // e[0..3]: markers on the handle's stream before noise / before rollout / after rollout / after tail;
  // e[4], e[5]: begin and end of the rollout kernel's own dispatch (hipExtLaunchKernelGGL, MPPI_LAUNCH_ROLLOUT)
  hipEvent_t e[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
};

}  // namespace

namespace mppi {
thread_local hipEvent_t tl_kernel_start = nullptr, tl_kernel_stop = nullptr;
}

struct mppi_handle {
  mppi_config cfg{};
  int K = 0, T = 0, k99 = 0;
  float dt = 0.0f;
  NetDesc net{};
  bool mfma_ok = false;
  int hidden = 0, n_hidden = 0;
  int variant_pref = 0;  // 0 auto, 1 mfma, 2 valu
  int block_threads = 0;    // 0: auto; 1040: multi form with ND = 4 and six waves (one cost wave); 512: quad (2 dynamics + cost + control waves per 16 rollouts); 800: oct (4 dynamics waves, one M tile of a
                            // 64-wide net each, + pose, cost, control, noise wave); 64, 256: single-wave form;
                            // 1000 + ND: multi form (ND dynamics waves of 16 rollouts + cost wave + control wave), ND = 1, 2, 4
  bool multi_standalone_noise = false;  // multi form: eps from the stand-alone generator kernel instead of the control wave
  int num_simds = 1024;     // 4 per CU
  hipStream_t stream = nullptr;
  // The stream the handle's most recent device work went to: its own, or the device's batch stream after a
  // batched solve (mppi_compute_control_batch).  nullptr: nothing outstanding anywhere but on `stream`.
  hipStream_t order_stream = nullptr;
  int n_slots = 1;  // explicit-noise slots in d_noise: one per iteration
  bool u_dirty = true;          // host copy of U/hist differs from the device copy in d_in
  unsigned seq = 0;             // sequence number of the last enqueued solve (last word of every h_res entry)
  std::vector<float> sg_buf;    // scratch of the host-side Savitzky-Golay pass
  bool basis = false;  // GeneralizedLinear basis-function dynamics (cfg.n_layers == 0); theta holds W[4][25]
  // DDP feedback gains (row f2): weights of initDDP (mppi_controller.cu:410-417) and the last result
  float ddp_Q[7] = {0.5f, 0.5f, 0.25f, 0.0f, 0.05f, 0.01f, 0.01f};
  float ddp_R[2] = {10.0f, 10.0f};
  float ddp_Qf[7] = {0, 0, 0, 0, 0, 0, 0};
  DdpResult ddp;
  bool have_ddp = false;
  unsigned *d_counter = nullptr;  // [1 + T] arrival counters of the tail kernel
  float *d_part = nullptr;        // [T][K/64][2] chain results of the tail kernel when K > 4096
  float *d_res_map = nullptr;   // device-side address of the host-mapped result block h_res

  std::vector<float> U, hist, theta, map_rgba;
  HostNetFma hnet;  // host twin of the network for computeNominalTraj (rebuilt with every mppi_set_nn_params)
  int map_w = 0, map_h = 0;
  mppi_cost_params cost{};
  float r_c1[3] = {0, 0, 0}, r_c2[3] = {0, 0, 0}, trs[3] = {0, 0, 1};
  float u_lo[2] = {0, 0}, u_hi[2] = {0, 0};
  bool have_nn = false, have_map = false, have_cost = false;

  float *d_in = nullptr, *d_scal = nullptr;
  float *d_in_buf[2] = {nullptr, nullptr};  // d_in points at one of them; the tail kernel leaves the
  int in_cur = 0;                            // stride-slid copy of [U | hist] in the other one
  bool slid_valid = false;
  float *d_noise = nullptr, *d_stage = nullptr;
  // Generator-kernel forms: eps of a solve is drawn by the stand-alone kernel into one of two buffers, on a
  // stream of its own (all generator launches, in order: the MRG32k3a states advance in launch order); the
  // draws of the NEXT solve are requested as soon as this solve's rollout has been enqueued and start when
  // that rollout ends, i.e. they overlap the weights / tail kernels, which leave the chip idle.
  float *d_gen[2] = {nullptr, nullptr};
  int gen_cur = 0;             // buffer of the most recent generator-mode solve (holds its applied controls V)
  bool gen_async = false;      // K T >= 2^20: generator on its own stream, next solve's draws prefetched
  bool prefetch_valid = false; // d_gen[1 - gen_cur] holds the next solve's draws (ev_gen marks their completion)
  float *v_buf = nullptr;      // where the last solve's applied controls are
  hipStream_t gstream = nullptr;
  hipEvent_t ev_gen = nullptr, ev_s1 = nullptr;
  // stage timing of the asynchronous generator: begin / end of the generator launch on gstream that was enqueued
  // during a timed solve (it runs BESIDE that solve's rollout or tail: reported as noise_ms, not additive)
  hipEvent_t ev_gt[2] = {nullptr, nullptr};
  bool gen_timed = false, gen_time_now = false;
  float *d_costs = nullptr, *d_w = nullptr;
  float *d_theta = nullptr, *d_wpack = nullptr, *d_map = nullptr;
  float *d_theta_s = nullptr;  // theta with hidden-layer biases * kTanhScale (register VALU kernel)
  float *d_rowpack = nullptr;  // 6-32-32-4: the weights in the register order of the row form (rollout_row.hip)
  float *d_row64pack = nullptr;  // 64-wide nets: register + LDS image of rollout_row64.hip
  bool valu_reg_ok = false;
  double *d_invt = nullptr;
  uint32_t *d_rng[2] = {nullptr, nullptr};
  uint32_t *d_jump = nullptr, *d_sub = nullptr, *d_one = nullptr;
  int rng_cur = 0;
  int noise_L = 1, noise_C = 1;
  float *h_in = nullptr, *h_res = nullptr;
  int explicit_iters = 0;  // >0: d_noise holds that many explicit iterations for the next solve
  bool pending = false;       // a solve is enqueued, results not yet collected
  bool pending_timed = false;
  float traj_cost = 0.0f, baseline = 0.0f, eta = 0.0f;

  int spin_budget = 0, fault_wave = 0;  // mppi_debug_inject_handover_fault (0, 0: kSpinBudget, no fault)
  // mppi_debug_capture_iterations: [num_iters][2T + K] -- the raw weighted mean U and the costs after every iteration
  float *d_cap = nullptr;
  bool capture = false, cap_valid = false, cap_explicit = false;
  double wait_timeout_s = 30.0;  // mppi_set_wait_timeout
  bool timing = false;
  int timing_every = 1;      // record stage events on every Nth solve only (events add launch gaps)
  unsigned timing_count = 0;
  std::vector<Events> ev;  // one set per iteration
  mppi_stage_times acc{};
  std::string err;
};

namespace {

int fail(mppi_handle *h, int code, const char *what, hipError_t e = hipSuccess)
{
  if (h) {
    h->err = what;
    if (e != hipSuccess) {
      h->err += ": ";
      h->err += hipGetErrorString(e);
    }
  }
  return code;
}

#define HIPCHK(h, call)                                                  \
  do {                                                                   \
    hipError_t e__ = (call);                                             \
    if (e__ != hipSuccess) return fail((h), MPPI_ERR_HIP, #call, e__);   \
  } while (0)

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
    done_.store(false);
    posted_.store(true);
    arm();
    mine();
    while (!done_.load(std::memory_order_acquire)) __builtin_ia32_pause();
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
            (*job_)();
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
  std::atomic<bool> spinning_{false}, posted_{false}, done_{false};
  const std::function<void()> *job_ = nullptr;
};
std::atomic<int> g_host_threads{1};
HostHelper &host_helper()
{
  static HostHelper hh;
  return hh;
}

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

// Called by every entry point that enqueues on the handle's OWN stream or relies on a synchronise of that stream
// having seen all of the handle's device work: if the handle's last work went to the batch stream, wait for it
// (only on a batch -> single transition: setup calls, result vectors, a stand-alone solve after a batched one).
int own_stream(mppi_handle *h)
{
  if (h->order_stream && h->order_stream != h->stream) HIPCHK(h, hipStreamSynchronize(h->order_stream));
  h->order_stream = h->stream;
  return MPPI_OK;
}
#define OWN(h)                      \
  do {                              \
    int rc__ = own_stream(h);       \
    if (rc__) return rc__;          \
  } while (0)
// where small follow-up work (upload of U, the slide kernel) goes: behind the handle's latest work, wherever it is
hipStream_t work_stream(const mppi_handle *h) { return h->order_stream ? h->order_stream : h->stream; }

int compute_k99(int K)
{
  // smallest k with (double)k >= .99*NUM_ROLLOUTS, mppi_controller.cu:141
  const double thr = .99 * (double)K;
  int k = 0;
  while (k < K && !((double)k >= thr)) k++;
  return k;
}

// A-operand / bias register image for rollout_mfma.hip (see the mapping comment there).
std::vector<float> pack_mfma_weights(const std::vector<float> &theta, int H, int NHID)
{
  const int MT = H / 16, KSH = H / 4;
  const int nA0 = MT * 2, nAH = MT * KSH, nAL = KSH;
  const int nA = nA0 + (NHID - 1) * nAH + nAL;
  const int nBias = NHID * MT * 4 + 4;
  std::vector<float> out((size_t)(nA + nBias) * 64, 0.0f);
  // offsets of W_l / b_l in theta, layers = 6, H x NHID, 4
  std::vector<int> wo, bo, nin, nout;
  int off = 0, prev = kNetIn;
  for (int l = 0; l <= NHID; l++) {
    const int no = (l < NHID) ? H : kNetOut;
    wo.push_back(off);
    bo.push_back(off + no * prev);
    nin.push_back(prev);
    nout.push_back(no);
    off += no * prev + no;
    prev = no;
  }
  for (int lane = 0; lane < 64; lane++) {
    const int row = lane & 15, kk = lane >> 4;  // A operand: A[row][k = kk]
    const int gq = row >> 2, rq = row & 3;
    // layer 0
    for (int m = 0; m < MT; m++)
      for (int s = 0; s < 2; s++) {
        const int n = 16 * m + 4 * rq + gq, kap = 4 * s + kk;
        out[(size_t)(m * 2 + s) * 64 + lane] = (kap < kNetIn) ? theta[wo[0] + n * kNetIn + kap] : 0.0f;
      }
    for (int l = 1; l < NHID; l++) {
      const int aoff = nA0 + (l - 1) * nAH;
      for (int m = 0; m < MT; m++)
        for (int s = 0; s < KSH; s++) {
          const int n = 16 * m + 4 * rq + gq, kap = 4 * s + kk;
          out[(size_t)(aoff + m * KSH + s) * 64 + lane] = theta[wo[l] + n * H + kap];
        }
    }
    {
      const int aoff = nA0 + (NHID - 1) * nAH;
      for (int s = 0; s < KSH; s++) {
        const int o = row & 3, kap = 4 * s + kk;
        out[(size_t)(aoff + s) * 64 + lane] = theta[wo[NHID] + o * H + kap];
      }
    }
    // biases: lane (j, g) register (m, r) holds D row 16m+4g+r = neuron 16m+4r+g
    const int g = lane >> 4;
    for (int l = 0; l < NHID; l++)
      for (int m = 0; m < MT; m++)
        for (int r = 0; r < 4; r++)
          out[(size_t)(nA + l * MT * 4 + m * 4 + r) * 64 + lane] = theta[bo[l] + 16 * m + 4 * r + g];
    for (int r = 0; r < 4; r++) out[(size_t)(nA + NHID * MT * 4 + r) * 64 + lane] = theta[bo[NHID] + r];
  }
  return out;
}

// Register image of the row form (rollout_row.hip: row_load): lane p of a rollout owns neurons 2p, 2p+1 of the hidden layers
// and outputs 2(p&1), 2(p&1)+1; entry i of lane p is float4 index i * 16 + p.  Hidden biases times kTanhScale.
std::vector<float> pack_row_weights(const std::vector<float> &theta)
{
  const int H = 32;
  const float *W1 = theta.data(), *B1 = W1 + H * kNetIn, *W2 = B1 + H, *B2 = W2 + H * H, *W3 = B2 + H, *B3 = W3 + kNetOut * H;
  std::vector<float> out((size_t)row_pack_floats(), 0.0f);
  for (int p = 0; p < 16; p++) {
    const int j0 = 2 * p, j1 = 2 * p + 1, o0 = 2 * (p & 1), o1 = o0 + 1;
    auto entry = [&](int i) { return &out[((size_t)i * 16 + p) * 4]; };
    for (int i = 0; i < 3; i++) {
      float *e = entry(i);
      e[0] = W1[j0 * kNetIn + 2 * i]; e[1] = W1[j1 * kNetIn + 2 * i];
      e[2] = W1[j0 * kNetIn + 2 * i + 1]; e[3] = W1[j1 * kNetIn + 2 * i + 1];
    }
    for (int i = 0; i < H / 2; i++) {
      float *e = entry(3 + i), *f = entry(3 + H / 2 + i);
      e[0] = W2[j0 * H + 2 * i]; e[1] = W2[j1 * H + 2 * i]; e[2] = W2[j0 * H + 2 * i + 1]; e[3] = W2[j1 * H + 2 * i + 1];
      f[0] = W3[o0 * H + 2 * i]; f[1] = W3[o1 * H + 2 * i]; f[2] = W3[o0 * H + 2 * i + 1]; f[3] = W3[o1 * H + 2 * i + 1];
    }
    float *b = entry(35), *c = entry(36);
    b[0] = B1[j0] * kTanhScale; b[1] = B1[j1] * kTanhScale; b[2] = B2[j0] * kTanhScale; b[3] = B2[j1] * kTanhScale;
    c[0] = B3[o0]; c[1] = B3[o1];
    // tree form (row_out_tree): this lane's own two activations into the four outputs, output (p >> 2) ^ i at position i
    const int o = p >> 2;
    float *t0 = entry(37), *t1 = entry(38), *t2 = entry(39);
    t0[0] = W3[o * H + j0]; t0[1] = W3[(o ^ 1) * H + j0]; t0[2] = W3[o * H + j1]; t0[3] = W3[(o ^ 1) * H + j1];
    t1[0] = W3[(o ^ 2) * H + j0]; t1[1] = W3[(o ^ 3) * H + j0]; t1[2] = W3[(o ^ 2) * H + j1]; t1[3] = W3[(o ^ 3) * H + j1];
    t2[0] = B3[o];
  }
  return out;
}

// Image of the 64-wide row form (rollout_row64.hip: row64_load + the LDS part): lane g of a 32-lane rollout owns neurons
// 2g, 2g+1 of every hidden layer; register entry i of lane g at float4 index i * 32 + g, then the 64 x 64 layers as
// [layer][k][g] pairs.  Hidden biases times kTanhScale.  Output layer in the order of row64_out_tree: Q = outputs {0, 1},
// P = outputs {2, 3}, inside each the output the lane keeps at the row_ror:8 level (bit 3 of g) first.
std::vector<float> pack_row64_weights(const std::vector<float> &theta, int NHID)
{
  const int H = 64, NB = (NHID + 1) / 2, NE = 3 + NB + 3;
  std::vector<float> out((size_t)row64_pack_floats(NHID), 0.0f);
  std::vector<const float *> Wl(NHID + 1), Bl(NHID + 1);
  {
    const float *p = theta.data();
    int prev = kNetIn;
    for (int l = 0; l <= NHID; l++) {
      const int no = (l < NHID) ? H : kNetOut;
      Wl[l] = p;
      Bl[l] = p + (size_t)no * prev;
      p += (size_t)no * prev + no;
      prev = no;
    }
  }
  for (int g = 0; g < 32; g++) {
    const int j0 = 2 * g, j1 = 2 * g + 1;
    auto entry = [&](int i) { return &out[((size_t)i * 32 + g) * 4]; };
    for (int i = 0; i < 3; i++) {
      float *e = entry(i);
      e[0] = Wl[0][j0 * kNetIn + 2 * i]; e[1] = Wl[0][j1 * kNetIn + 2 * i];
      e[2] = Wl[0][j0 * kNetIn + 2 * i + 1]; e[3] = Wl[0][j1 * kNetIn + 2 * i + 1];
    }
    for (int l = 0; l < NHID; l++) {
      float *e = entry(3 + l / 2) + 2 * (l & 1);
      e[0] = Bl[l][j0] * kTanhScale; e[1] = Bl[l][j1] * kTanhScale;
    }
    const int row = g >> 4, b3 = (g >> 3) & 1;
    const int qa = b3, qb = 1 - b3, pa = 2 + b3, pb = 3 - b3;
    const float *W3 = Wl[NHID];
    float *q = entry(3 + NB), *pp = entry(4 + NB), *c = entry(5 + NB);
    q[0] = W3[qa * H + j0]; q[1] = W3[qb * H + j0]; q[2] = W3[qa * H + j1]; q[3] = W3[qb * H + j1];
    pp[0] = W3[pa * H + j0]; pp[1] = W3[pb * H + j0]; pp[2] = W3[pa * H + j1]; pp[3] = W3[pb * H + j1];
    c[0] = Bl[NHID][2 * row + b3];
  }
  float *lds = out.data() + (size_t)NE * 32 * 4;
  for (int l = 1; l < NHID; l++)
    for (int k = 0; k < H; k++)
      for (int g = 0; g < 32; g++) {
        float *e = lds + (((size_t)(l - 1) * H + k) * 32 + g) * 2;
        e[0] = Wl[l][(2 * g) * H + k];
        e[1] = Wl[l][(2 * g + 1) * H + k];
      }
  return out;
}

bool use_mfma(const mppi_handle *h)
{
  if (h->basis || h->variant_pref == 2 || h->variant_pref == 3) return false;
  return h->mfma_ok;
}

// "valu" on a standard shape runs the register/scalar-operand kernel; "valu_lds" forces the generic one
bool use_valu_reg(const mppi_handle *h) { return !h->basis && !use_mfma(h) && h->valu_reg_ok && h->variant_pref != 3; }

// Kernel form for the MFMA path, by the number of 16-rollout groups against the machine (MI355X: 256 CUs of
// 4 SIMDs).  Measured rollout-kernel times (this file's forms are bit-identical, so only time decides):
//   * up to one group per CU (K <= 4096): the QUAD form -- the network itself split over two SIMDs, plus a
//     cost and a control wave; the T-step recurrence is latency bound and this is the shortest chain
//     (6-32-32-4, T=100, K=4096: quad 71 us, multi1 / multi2 83 us, single-wave 122 us).  64-wide nets: the
//     OCT form -- one M tile per dynamics wave, four of them, and four riders (rollout_oct.hip; T=100,
//     K=4096: 6-64-64-4 oct 108 us, quad 134 us; 6-64x4-4 oct 185 us, quad 279 us), also at two groups per
//     CU (K=8192: 6-64-64-4 oct 172 us, multi2 187 us; 6-64x4-4 oct 365 us, single-wave 503 us; at four
//     groups per CU it loses: 724 vs 508 us);
//   * up to two groups per CU (K <= 8192): MULTI2 -- two dynamics waves (whole network each), one cost wave,
//     one control wave with the in-kernel generator, every wave on a SIMD of its own (K=8192: 83 us; quad
//     112 us, single-wave 123 us; 6-64-64-4, T=150: 277 us vs 359 / 339 us);
//   * beyond: MULTI4 with eps from the stand-alone generator kernel -- four dynamics waves per workgroup, one
//     per SIMD, the cost and control waves riding along (K=16384: 106 us vs 124 us single-wave;
//     6-64-64-4, T=150: 306 us vs 341 us; the in-kernel generator would load one SIMD too much: 341 us).
// Shapes the multi form does not have (6-64x4-4: its weights do not fit a wave of a six-wave workgroup) run
// the single-wave form beyond one group per CU -- in
// workgroups of FOUR waves: the dispatcher spreads the waves of one workgroup over the four SIMDs of a CU,
// whereas 64-thread workgroups are placed one by one and -- at one wave per SIMD on paper (K = 16384) --
// sometimes two on one SIMD and none on its neighbour, which doubles the kernel time
// (tools/placement_probe.hip: 106 of 1024 SIMDs doubled on a first launch; rollout 601 us vs 341 us).
inline bool is_row64(int b) { return b == 908 || b == 916; }  // rollout_row64.hip, 8 / 16 rollouts per group
inline bool is_row(int b) { return b == 900 || b == 901; }  // 901: the tree form of the output layer (rollout_row.hip)
int effective_block(const mppi_handle *h)
{
  if (h->block_threads != 0) return h->block_threads;
  const int groups = h->K / kRolloutsPerWave;
  const int cus = h->num_simds / 4;
  if (groups <= 2 * cus && oct_variant_supported(h->hidden, h->n_hidden)) return 800;
  if (multi_variant_supported(h->hidden, h->n_hidden)) {
    // 6-32-32-4 at one group per CU: the vector-ALU ROW form (rollout_row.hip) -- the shortest recurrence of all
    // (K=4096, T=100: 56.8 us; quad 68.2 us)
    // ("mfma" asked for explicitly -- the A/B arm of SURVEY cfg 4 -- keeps the matrix-instruction forms)
    // -- in its TREE form (901: the output layer as per-lane partials + a butterfly, rollout 45.8 -> 36.7 us; inside the
    // north-star tolerance of the reference's summation order, tests/test_row_tree_gpu.py); "row_exact" keeps the
    // k-ascending output chain (900), bit-identical to every other form
    if (groups <= cus && row_variant_supported(h->hidden, h->n_hidden) && h->variant_pref != 1) return 901;
    if (groups <= cus) return 512;
    if (groups <= 2 * cus) return 1002;
    // (64-wide nets beyond one group per SIMD: the eight-wave form needs 172 VGPRs = one workgroup per CU, so K = 32768
    // runs in two rounds.  The six-wave form "multi4u" -- 168 VGPRs, three waves per SIMD, two workgroups per CU =
    // two dynamics waves + one rider per SIMD -- was measured against it: K=32768, T=150, 6-64-64-4 0.584 ms vs
    // 0.538 ms; K=16384 0.298 vs 0.272 ms.  Two f32-MFMA waves on one SIMD take the sum of their times (the f32
    // MFMA occupies the vector datapath), so co-residence buys nothing and the single cost wave is the
    // slower rider.  Not chosen automatically; kept as an A/B arm.)
    return 1004;
  }
  return (4 * groups <= h->num_simds) ? 512 : 256;
}

// multi form: eps from the stand-alone generator kernel (forced by "_gen", and the automatic choice for ND = 4)
bool multi_gen(const mppi_handle *h)
{
  if (h->block_threads != 0) return h->multi_standalone_noise;
  return effective_block(h) == 1004 || effective_block(h) == 1040;
}

// basis-function model, wavefronts per 64 rollouts: dynamics + cost + control wave (in-kernel generator) while
// each gets a SIMD of its own; dynamics + cost wave ("quad") up to twice that; one wave ("fused" / "block64")
int bf_waves(const mppi_handle *h)
{
  if (h->block_threads == 64 || h->block_threads == 256) return 1;
  if (h->block_threads == 512) return 2;
  if (h->block_threads == 768) return 3;
  if (3 * (h->K / 64) <= h->num_simds) return 3;
  return (2 * (h->K / 64) <= 2 * h->num_simds) ? 2 : 1;
}

// the quad and multi MFMA kernels carry their own control/noise wavefront
bool has_noise_wave(const mppi_handle *h)
{
  if (h->basis) return bf_waves(h) == 3;
  if (!use_mfma(h)) return false;
  const int b = effective_block(h);
  return b == 512 || is_row(b) || is_row64(b) || ((b == 800 || b > 1000) && !multi_gen(h));
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
  a.wpack = use_mfma(h) ? (is_row(effective_block(h)) ? h->d_rowpack : is_row64(effective_block(h)) ? h->d_row64pack : h->d_wpack) : (use_valu_reg(h) ? h->d_theta_s : h->d_theta);
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
  a.spin_budget = h->spin_budget;
  a.fault_wave = h->fault_wave;
  fill_cost_args(h, a.cost);
}

int launch_rollout(mppi_handle *h, const RolloutArgs &a)
{
  // basis-function model: the two-wave form while both waves of a group get a SIMD of their own
  hipError_t e = h->basis ? launch_rollout_bf(a, bf_waves(h), h->stream)
                 : (use_mfma(h) && effective_block(h) > 1000)
                     ? launch_rollout_multi(h->hidden, h->n_hidden, a, effective_block(h) - 1000, h->stream)
                 : (use_mfma(h) && effective_block(h) == 800) ? launch_rollout_oct(h->hidden, h->n_hidden, a, h->stream)
                 : (use_mfma(h) && is_row64(effective_block(h))) ? launch_rollout_row64(h->hidden, h->n_hidden, a, effective_block(h) - 900, h->stream)
                 : (use_mfma(h) && is_row(effective_block(h))) ? launch_rollout_row(h->hidden, h->n_hidden, a, effective_block(h) == 901, h->stream)
                 : use_mfma(h) ? launch_rollout_mfma(h->hidden, h->n_hidden, a, effective_block(h), h->stream)
                 : use_valu_reg(h) ? launch_rollout_valu_reg(h->hidden, h->n_hidden, a, h->stream)
                                   : launch_rollout_valu(h->net, a, h->stream);
  if (e != hipSuccess) return fail(h, MPPI_ERR_HIP, "rollout launch", e);
  return MPPI_OK;
}

int seed_device(mppi_handle *h, uint64_t seed, uint64_t offset)
{
  // base state: L'Ecuyer's default 12345 x 6, scrambled by the seed (DESIGN.md noise spec)
  uint32_t base[6] = {12345u, 12345u, 12345u, 12345u, 12345u, 12345u};
  if (seed != 0) {
    const uint32_t x1 = ((uint32_t)seed) ^ 0x55555555u;
    const uint32_t x2 = (uint32_t)((seed >> 32) ^ 0xAAAAAAAAu);
    base[0] = (uint32_t)((uint64_t)x1 * base[0] % M1);
    base[1] = (uint32_t)((uint64_t)x2 * base[1] % M1);
    base[2] = (uint32_t)((uint64_t)x1 * base[2] % M1);
    base[3] = (uint32_t)((uint64_t)x2 * base[3] % M2);
    base[4] = (uint32_t)((uint64_t)x1 * base[4] % M2);
    base[5] = (uint32_t)((uint64_t)x2 * base[5] % M2);
  }
  int sub_bits = 0;
  while ((1LL << sub_bits) < (long long)h->K) sub_bits++;
  HIPCHK(h, launch_noise_init(h->d_rng[0], h->K, base, h->d_sub, sub_bits, h->d_one, offset, h->stream));
  h->rng_cur = 0;
  h->cfg.seed = seed;
  return MPPI_OK;
}

int upload_rng_tables(mppi_handle *h)
{
  std::vector<uint32_t> sub(32 * 18), one(64 * 18), jump((size_t)h->noise_C * 18);
  Mat3 a1 = base_A1(), a2 = base_A2();
  for (int b = 0; b < 64; b++) {  // A^(2^b)
    memcpy(&one[(size_t)b * 18], a1.a, 36);
    memcpy(&one[(size_t)b * 18 + 9], a2.a, 36);
    a1 = mat_mul(a1, a1, M1);
    a2 = mat_mul(a2, a2, M2);
  }
  for (int b = 64; b < 76; b++) {
    a1 = mat_mul(a1, a1, M1);
    a2 = mat_mul(a2, a2, M2);
  }
  for (int b = 0; b < 32; b++) {  // A^(2^76 * 2^b)
    memcpy(&sub[(size_t)b * 18], a1.a, 36);
    memcpy(&sub[(size_t)b * 18 + 9], a2.a, 36);
    a1 = mat_mul(a1, a1, M1);
    a2 = mat_mul(a2, a2, M2);
  }
  const Mat3 j1 = mat_pow(base_A1(), 2ULL * (uint64_t)h->noise_L, M1);
  const Mat3 j2 = mat_pow(base_A2(), 2ULL * (uint64_t)h->noise_L, M2);
  Mat3 c1 = mat_identity(), c2 = mat_identity();
  for (int c = 0; c < h->noise_C; c++) {  // A^(2 L c)
    memcpy(&jump[(size_t)c * 18], c1.a, 36);
    memcpy(&jump[(size_t)c * 18 + 9], c2.a, 36);
    c1 = mat_mul(j1, c1, M1);
    c2 = mat_mul(j2, c2, M2);
  }
  HIPCHK(h, hipMemcpy(h->d_sub, sub.data(), sub.size() * 4, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(h->d_one, one.data(), one.size() * 4, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(h->d_jump, jump.data(), jump.size() * 4, hipMemcpyHostToDevice));
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
  // While every SIMD runs at most one dynamics wave (K <= 16 x #SIMDs) the generator starts at once, beside the
  // rollout: its instructions fit the dependency bubbles of the dynamics waves (config 4 0.317 -> 0.309 ms per
  // solve, K=16384 6-32-32-4 0.122 -> 0.117).  With several workgroups per CU there are no bubbles left
  // (K=65536: 0.380 -> 0.470 ms), so there it starts when the rollout ends, beside the weights / tail kernels.
  if (h->K / kRolloutsPerWave > h->num_simds) HIPCHK(h, hipStreamWaitEvent(h->gstream, h->ev_s1, 0));
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
int wait_pending(mppi_handle *h)
{
  if (!h->pending) return MPPI_OK;
  // The tail kernel writes T+2 entries of 16 B into host-mapped memory -- row t: [u0, seq, u1, seq], then
  // [beta, seq, eta, seq] and [trajectory cost, seq, 0, seq] -- each as one store.  An entry is complete
  // once words 1 and 3 carry this solve's sequence number (either 8-byte half may land first); the solve
  // is complete for the host once every entry is.
  const volatile unsigned *words = reinterpret_cast<const volatile unsigned *>(h->h_res);
  const int n_entries = h->T + 2;
  const auto t0 = std::chrono::steady_clock::now();
  unsigned long spins = 0;
  int next = 0;  // entries [0, next) have been seen with the sequence number
  for (;;) {
    while (next < n_entries && __atomic_load_n(words + 4 * next + 1, __ATOMIC_ACQUIRE) == h->seq &&
           __atomic_load_n(words + 4 * next + 3, __ATOMIC_ACQUIRE) == h->seq)
      next++;
    if (next == n_entries) break;
    __builtin_ia32_pause();
    if ((++spins & 0xFFFFF) == 0) {
      if (hipStreamQuery(work_stream(h)) == hipSuccess && (__atomic_load_n(words + 4 * next + 1, __ATOMIC_ACQUIRE) != h->seq ||
                                                      __atomic_load_n(words + 4 * next + 3, __ATOMIC_ACQUIRE) != h->seq))
        return fail(h, MPPI_ERR_HIP, "solve finished without publishing its result block");
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > h->wait_timeout_s)
        return fail(h, MPPI_ERR_HIP, "timed out waiting for the solve");
    }
  }
#ifdef MPPI_HOSTPROF
  hp_seen = std::chrono::steady_clock::now();
#endif
  h->pending = false;
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

int collect(mppi_handle *h) { return wait_pending(h); }

// the tail kernel of the last iteration leaves [U | hist] slid by the optimization stride in the other buffer
bool wants_slid_copy(const mppi_handle *h)
{
  return h->cfg.optimization_stride >= 1 && h->cfg.optimization_stride < h->T;
}

TailLaunch tail_launch(const mppi_handle *h, const float *V, bool last)
{
  TailLaunch l;
  l.costs = h->d_costs; l.V = V; l.U = h->d_in; l.hist = h->d_in + 2 * h->T; l.w = h->d_w; l.scal = h->d_scal;
  l.res = h->d_res_map; l.counter = h->d_counter; l.part = h->d_part;
  l.K = h->K; l.T = h->T; l.gamma = h->cfg.gamma; l.last_iter = last ? 1 : 0; l.seq = h->seq;
  l.slid = (last && wants_slid_copy(h)) ? h->d_in_buf[1 - h->in_cur] : nullptr;
  l.slide_stride = h->cfg.optimization_stride;
  l.init0 = h->cfg.init_control[0]; l.init1 = h->cfg.init_control[1];
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

void free_all(mppi_handle *h)
{
  if (!h) return;
  float *fp[] = {h->d_theta_s, h->d_in_buf[0], h->d_in_buf[1], h->d_scal, h->d_noise, h->d_stage, h->d_costs,
                 h->d_w, h->d_theta, h->d_wpack, h->d_map, h->d_part, h->d_rowpack, h->d_row64pack, h->d_cap};
  for (float *p : fp)
    if (p) (void)hipFree(p);
  if (h->d_invt) (void)hipFree(h->d_invt);
  if (h->d_counter) (void)hipFree(h->d_counter);
  uint32_t *up[] = {h->d_rng[0], h->d_rng[1], h->d_jump, h->d_sub, h->d_one};
  for (uint32_t *p : up)
    if (p) (void)hipFree(p);
  if (h->h_in) (void)hipHostFree(h->h_in);
  if (h->h_res) (void)hipHostFree(h->h_res);
  for (auto &s : h->ev)
    for (auto &e : s.e)
      if (e) (void)hipEventDestroy(e);
  for (float *p : h->d_gen)
    if (p) (void)hipFree(p);
  if (h->ev_gen) (void)hipEventDestroy(h->ev_gen);
  for (auto &e : h->ev_gt)
    if (e) (void)hipEventDestroy(e);
  if (h->ev_s1) (void)hipEventDestroy(h->ev_s1);
  if (h->gstream) (void)hipStreamDestroy(h->gstream);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

}  // namespace

extern "C" {

int mppi_abi_version(void) { return MPPI_ABI_VERSION; }

const char *mppi_strerror(int status)
{
  switch (status) {
    case MPPI_OK: return "ok";
    case MPPI_ERR_INVALID: return "invalid argument";
    case MPPI_ERR_NO_DEVICE: return "no usable gfx950 device";
    case MPPI_ERR_HIP: return "HIP runtime error";
    case MPPI_ERR_STATE: return "call order / missing setup";
    case MPPI_ERR_UNSUPPORTED: return "unsupported configuration";
    default: return "unknown status";
  }
}

const char *mppi_last_error(const mppi_handle *h) { return h ? h->err.c_str() : "null handle"; }

int mppi_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  int ok = 0;
  for (int i = 0; i < n; i++) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, i) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) ok++;
  }
  return ok;
}

int mppi_create(const mppi_config *cfg, mppi_handle **out)
{
  if (!cfg || !out) return MPPI_ERR_INVALID;
  *out = nullptr;
  if (cfg->num_rollouts <= 0 || cfg->num_rollouts % 64 != 0) return MPPI_ERR_INVALID;
  if (cfg->num_timesteps < 2 || cfg->hz <= 0 || cfg->num_iters < 1) return MPPI_ERR_INVALID;
  if (cfg->optimization_stride < 0) return MPPI_ERR_INVALID;
  const bool basis = (cfg->n_layers == 0);  // GeneralizedLinear basis-function dynamics
  if (!basis) {
    if (cfg->n_layers < 2 || cfg->n_layers > MPPI_MAX_LAYERS) return MPPI_ERR_INVALID;
    if (cfg->layers[0] != kNetIn || cfg->layers[cfg->n_layers - 1] != kNetOut) return MPPI_ERR_INVALID;
    for (int i = 0; i < cfg->n_layers; i++)
      if (cfg->layers[i] <= 0 || cfg->layers[i] > 256) return MPPI_ERR_INVALID;
  }
  if ((size_t)cfg->num_rollouts * (size_t)cfg->num_timesteps > (size_t)1 << 28) return MPPI_ERR_INVALID;

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return MPPI_ERR_NO_DEVICE;
  if (cfg->device < 0 || cfg->device >= ndev) return MPPI_ERR_NO_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess) return MPPI_ERR_NO_DEVICE;
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return MPPI_ERR_NO_DEVICE;
  if (hipSetDevice(cfg->device) != hipSuccess) return MPPI_ERR_HIP;
  const int n_cus = prop.multiProcessorCount;

  mppi_handle *h = new (std::nothrow) mppi_handle();
  if (!h) return MPPI_ERR_HIP;
  h->cfg = *cfg;
  h->K = cfg->num_rollouts;
  h->T = cfg->num_timesteps;
  h->k99 = compute_k99(h->K);  // once: the search is O(K) and fill_rollout_args sits between two solves
  h->dt = (float)(1.0 / cfg->hz);  // path_integral_main.cu:100
  h->num_simds = 4 * (n_cus > 0 ? n_cus : 256);
  h->net.n_layers = cfg->n_layers;
  h->net.max_width = 0;
  h->net.num_params = 0;
  for (int i = 0; i < 8; i++) h->net.layers[i] = (i < cfg->n_layers) ? cfg->layers[i] : 0;
  for (int i = 0; i < cfg->n_layers; i++) h->net.max_width = std::max(h->net.max_width, cfg->layers[i]);
  for (int i = 0; i + 1 < cfg->n_layers; i++) h->net.num_params += (cfg->layers[i] + 1) * cfg->layers[i + 1];
  h->basis = basis;
  if (basis) h->net.num_params = 4 * kNumBfs;  // W[4][25], generalized_linear.cu:80
  // MFMA variant: 6 -> H x NHID -> 4
  h->n_hidden = basis ? 0 : cfg->n_layers - 2;
  h->hidden = (h->n_hidden > 0) ? cfg->layers[1] : 0;
  bool uniform = h->n_hidden > 0;
  for (int i = 1; i <= h->n_hidden; i++) uniform = uniform && (cfg->layers[i] == h->hidden);
  h->mfma_ok = uniform && mfma_variant_supported(h->hidden, h->n_hidden);
  h->valu_reg_ok = uniform && valu_reg_supported(h->hidden, h->n_hidden);
  for (int i = 0; i < 2; i++) {
    h->u_lo[i] = cfg->control_min[i];
    h->u_hi[i] = cfg->control_max[i];
  }
  h->U.assign(2 * (size_t)h->T, 0.0f);
  for (int t = 0; t < h->T; t++) {  // resetControls, mppi_controller.cu:448-458
    h->U[2 * t] = cfg->init_control[0];
    h->U[2 * t + 1] = cfg->init_control[1];
  }
  h->hist.assign(4, 0.0f);  // control_hist_, :347
  // noise chunking: enough (k, chunk) threads to cover the chip
  {
    int chunks = std::max(1, (1 << 18) / h->K);
    chunks = std::min(chunks, 64);
    h->noise_L = std::max(1, (h->T + chunks - 1) / chunks);
    h->noise_C = (h->T + h->noise_L - 1) / h->noise_L;
  }
  const size_t KT2 = (size_t)h->K * h->T * 2;
#define CR(call)                                                        \
  do {                                                                  \
    hipError_t e__ = (call);                                            \
    if (e__ != hipSuccess) {                                            \
      fprintf(stderr, "mppi_create: %s: %s\n", #call, hipGetErrorString(e__)); \
      free_all(h);                                                      \
      return MPPI_ERR_HIP;                                              \
    }                                                                   \
  } while (0)
  CR(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  CR(hipStreamCreateWithFlags(&h->gstream, hipStreamNonBlocking));
  CR(hipEventCreateWithFlags(&h->ev_gen, hipEventDisableTiming));
  CR(hipEventCreate(&h->ev_gt[0]));
  CR(hipEventCreate(&h->ev_gt[1]));
  CR(hipEventCreateWithFlags(&h->ev_s1, hipEventDisableTiming));
  h->n_slots = std::max(1, cfg->num_iters);
  CR(hipMalloc(&h->d_in_buf[0], sizeof(float) * (2 * (size_t)h->T + 4)));
  CR(hipMalloc(&h->d_in_buf[1], sizeof(float) * (2 * (size_t)h->T + 4)));
  h->d_in = h->d_in_buf[0];
  CR(hipMalloc(&h->d_scal, sizeof(float) * 4));
  CR(hipMalloc(&h->d_noise, sizeof(float) * KT2 * (size_t)h->n_slots));
  CR(hipMalloc(&h->d_gen[0], sizeof(float) * KT2));
  CR(hipMalloc(&h->d_gen[1], sizeof(float) * KT2));
  h->v_buf = h->d_gen[0];
  h->gen_async = (size_t)h->K * (size_t)h->T >= ((size_t)1 << 20);
  CR(hipMalloc(&h->d_counter, sizeof(unsigned) * (1 + (size_t)h->T)));
  CR(hipMemset(h->d_counter, 0, sizeof(unsigned) * (1 + (size_t)h->T)));
  if (h->K > 4096) CR(hipMalloc(&h->d_part, sizeof(float) * (size_t)h->T * (h->K / 64) * 2));
  CR(hipMalloc(&h->d_stage, sizeof(float) * KT2));
  CR(hipMalloc(&h->d_costs, sizeof(float) * h->K));
  CR(hipMalloc(&h->d_w, sizeof(float) * h->K));
  CR(hipMalloc(&h->d_theta, sizeof(float) * h->net.num_params));
  CR(hipMalloc(&h->d_theta_s, sizeof(float) * h->net.num_params));
  if (h->mfma_ok)
    CR(hipMalloc(&h->d_wpack, sizeof(float) * 64 * (size_t)mfma_pack_floats_per_lane(h->hidden, h->n_hidden)));
  if (h->mfma_ok && row_variant_supported(h->hidden, h->n_hidden)) CR(hipMalloc(&h->d_rowpack, sizeof(float) * (size_t)row_pack_floats()));
  if (h->mfma_ok && row64_variant_supported(h->hidden, h->n_hidden))
    CR(hipMalloc(&h->d_row64pack, sizeof(float) * (size_t)row64_pack_floats(h->n_hidden)));
  CR(hipMalloc(&h->d_rng[0], sizeof(uint32_t) * 6 * h->K));
  CR(hipMalloc(&h->d_rng[1], sizeof(uint32_t) * 6 * h->K));
  CR(hipMalloc(&h->d_jump, sizeof(uint32_t) * 18 * h->noise_C));
  CR(hipMalloc(&h->d_sub, sizeof(uint32_t) * 18 * 32));
  CR(hipMalloc(&h->d_one, sizeof(uint32_t) * 18 * 64));
  CR(hipHostMalloc(&h->h_in, sizeof(float) * (2 * (size_t)h->T + 4), hipHostMallocDefault));
  CR(hipHostMalloc(&h->h_res, sizeof(float) * 4 * ((size_t)h->T + 2), hipHostMallocMapped));
  memset(h->h_res, 0, sizeof(float) * 4 * ((size_t)h->T + 2));
  {
    void *dp = nullptr;
    CR(hipHostGetDevicePointer(&dp, h->h_res, 0));
    h->d_res_map = static_cast<float *>(dp);
  }
  CR(hipMalloc(&h->d_invt, sizeof(double) * (size_t)h->T));
  {
    std::vector<double> invt((size_t)h->T, 0.0);
    for (int t = 1; t < h->T; t++) invt[t] = 1.0 / (double)t;  // correctly rounded reciprocal
    CR(hipMemcpy(h->d_invt, invt.data(), sizeof(double) * (size_t)h->T, hipMemcpyHostToDevice));
  }
  CR(hipMemset(h->d_scal, 0, sizeof(float) * 4));
  h->ev.resize(cfg->num_iters);
  for (auto &s : h->ev)
    for (auto &e : s.e) CR(hipEventCreate(&e));
#undef CR
  int rc = upload_rng_tables(h);
  if (rc == MPPI_OK) rc = seed_device(h, cfg->seed, 0);
  if (rc == MPPI_OK && hipStreamSynchronize(h->stream) != hipSuccess) rc = MPPI_ERR_HIP;
  if (rc != MPPI_OK) {
    fprintf(stderr, "mppi_create: rng setup failed: %s\n", h->err.c_str());
    free_all(h);
    return rc;
  }
  *out = h;
  return MPPI_OK;
}

int mppi_destroy(mppi_handle *h)
{
  if (!h) return MPPI_ERR_INVALID;
  (void)hipSetDevice(h->cfg.device);
  if (h->order_stream && h->order_stream != h->stream) (void)hipStreamSynchronize(h->order_stream);
  if (h->gstream) (void)hipStreamSynchronize(h->gstream);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  free_all(h);
  return MPPI_OK;
}

int mppi_set_bf_params(mppi_handle *h, const float *W, size_t n)
{
  if (!h || !W) return MPPI_ERR_INVALID;
  if (!h->basis) return fail(h, MPPI_ERR_STATE, "handle was created with a network (n_layers != 0)");
  if (n != (size_t)(4 * kNumBfs)) return fail(h, MPPI_ERR_INVALID, "W size != 4 * 25");
  HIPCHK(h, hipSetDevice(h->cfg.device));
  OWN(h);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->theta.assign(W, W + n);
  HIPCHK(h, hipMemcpy(h->d_theta, W, n * sizeof(float), hipMemcpyHostToDevice));
  h->have_nn = true;
  return MPPI_OK;
}

int mppi_set_nn_params(mppi_handle *h, const float *theta, size_t n)
{
  if (!h || !theta) return MPPI_ERR_INVALID;
  if (h->basis) return fail(h, MPPI_ERR_STATE, "handle was created for basis-function dynamics: use mppi_set_bf_params");
  if (n != (size_t)h->net.num_params) return fail(h, MPPI_ERR_INVALID, "theta size != NUM_PARAMS");
  HIPCHK(h, hipSetDevice(h->cfg.device));
  OWN(h);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->theta.assign(theta, theta + n);
  HIPCHK(h, hipMemcpy(h->d_theta, theta, n * sizeof(float), hipMemcpyHostToDevice));
  {
    std::vector<float> ts(h->theta);  // hidden-layer biases pre-scaled for tanh_bias
    size_t off = 0;
    for (int l = 0; l + 1 < h->net.n_layers; l++) {
      const size_t nin = h->net.layers[l], nout = h->net.layers[l + 1];
      if (l + 2 < h->net.n_layers)
        for (size_t j = 0; j < nout; j++) ts[off + nin * nout + j] = ts[off + nin * nout + j] * kTanhScale;
      off += nin * nout + nout;
    }
    HIPCHK(h, hipMemcpy(h->d_theta_s, ts.data(), n * sizeof(float), hipMemcpyHostToDevice));
  }
  h->hnet.init(h->net.layers, h->net.n_layers, h->theta.data());
  if (h->mfma_ok) {
    const std::vector<float> pk = pack_mfma_weights(h->theta, h->hidden, h->n_hidden);
    HIPCHK(h, hipMemcpy(h->d_wpack, pk.data(), pk.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  if (h->d_rowpack) {
    const std::vector<float> pk = pack_row_weights(h->theta);
    HIPCHK(h, hipMemcpy(h->d_rowpack, pk.data(), pk.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  if (h->d_row64pack) {
    const std::vector<float> pk = pack_row64_weights(h->theta, h->n_hidden);
    HIPCHK(h, hipMemcpy(h->d_row64pack, pk.data(), pk.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  h->have_nn = true;
  return MPPI_OK;
}

int mppi_update_model(mppi_handle *h, const int *description, int n_desc, const float *data, size_t n)
{
  if (!h || !description || !data) return MPPI_ERR_INVALID;
  if (h->basis) return fail(h, MPPI_ERR_STATE, "updateModel exists for the network model only");
  // neural_net_model.cu:155-161: a mismatching description leaves the model untouched
  for (int i = 0; i < n_desc; i++)
    if (i >= h->net.n_layers || description[i] != h->net.layers[i])
      return fail(h, MPPI_ERR_INVALID, "description does not match the network structure");
  if (n != (size_t)h->net.num_params) return fail(h, MPPI_ERR_INVALID, "data size != NUM_PARAMS");
  // data = [W1|W2|..|b1|b2|..] -> packed [W1|b1|W2|b2|..]
  std::vector<float> theta(n);
  size_t woff = 0, boff = 0, poff = 0;
  for (int l = 0; l + 1 < h->net.n_layers; l++) boff += (size_t)h->net.layers[l] * h->net.layers[l + 1];
  for (int l = 0; l + 1 < h->net.n_layers; l++) {
    const size_t nw = (size_t)h->net.layers[l] * h->net.layers[l + 1], nb = h->net.layers[l + 1];
    memcpy(&theta[poff], data + woff, nw * sizeof(float));
    memcpy(&theta[poff + nw], data + boff, nb * sizeof(float));
    woff += nw;
    boff += nb;
    poff += nw + nb;
  }
  return mppi_set_nn_params(h, theta.data(), n);
}

int mppi_set_control_limits(mppi_handle *h, const float umin[2], const float umax[2])
{
  if (!h || !umin || !umax) return MPPI_ERR_INVALID;
  for (int i = 0; i < 2; i++) {
    h->u_lo[i] = umin[i];
    h->u_hi[i] = umax[i];
  }
  return MPPI_OK;
}

int mppi_set_costmap(mppi_handle *h, int width, int height, const float *rgba, const float r_c1[3],
                     const float r_c2[3], const float trs[3])
{
  if (!h || !rgba || !r_c1 || !r_c2 || !trs) return MPPI_ERR_INVALID;
  if (width <= 0 || height <= 0 || (size_t)width * (size_t)height > ((size_t)1 << 30))
    return fail(h, MPPI_ERR_INVALID, "bad costmap size");
  HIPCHK(h, hipSetDevice(h->cfg.device));
  OWN(h);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  const size_t n = (size_t)width * height;
  h->map_rgba.assign(rgba, rgba + 4 * n);
  std::vector<float> ch0(n);
  for (size_t i = 0; i < n; i++) ch0[i] = rgba[4 * i];  // only .x is sampled (costs.cu:380-381)
  if (h->d_map) {
    HIPCHK(h, hipFree(h->d_map));
    h->d_map = nullptr;
  }
  HIPCHK(h, hipMalloc(&h->d_map, n * sizeof(float)));
  HIPCHK(h, hipMemcpy(h->d_map, ch0.data(), n * sizeof(float), hipMemcpyHostToDevice));
  h->map_w = width;
  h->map_h = height;
  for (int i = 0; i < 3; i++) {
    h->r_c1[i] = r_c1[i];
    h->r_c2[i] = r_c2[i];
    h->trs[i] = trs[i];
  }
  h->have_map = true;
  return MPPI_OK;
}

int mppi_set_costmap_transform(mppi_handle *h, const float r_c1[3], const float r_c2[3], const float trs[3])
{
  if (!h || !r_c1 || !r_c2 || !trs) return MPPI_ERR_INVALID;
  if (!h->have_map) return fail(h, MPPI_ERR_STATE, "mppi_set_costmap has not been called");
  for (int i = 0; i < 3; i++) {  // kernel arguments of the next launch; nothing on the device changes
    h->r_c1[i] = r_c1[i];
    h->r_c2[i] = r_c2[i];
    h->trs[i] = trs[i];
  }
  return MPPI_OK;
}

int mppi_set_costmap_channel(mppi_handle *h, int channel, const float *data, size_t n)
{
  if (!h || !data) return MPPI_ERR_INVALID;
  if (!h->have_map) return fail(h, MPPI_ERR_STATE, "mppi_set_costmap has not been called");
  if (channel < 0 || channel > 3 || n != (size_t)h->map_w * h->map_h)
    return fail(h, MPPI_ERR_INVALID, "bad channel or size");
  for (size_t i = 0; i < n; i++) h->map_rgba[4 * i + channel] = data[i];
  if (channel == 0) {
    HIPCHK(h, hipSetDevice(h->cfg.device));
    OWN(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(h->d_map, data, n * sizeof(float), hipMemcpyHostToDevice));
  }
  return MPPI_OK;
}

int mppi_set_cost_params(mppi_handle *h, const mppi_cost_params *p)
{
  if (!h || !p) return MPPI_ERR_INVALID;
  h->cost = *p;
  h->have_cost = true;
  return MPPI_OK;
}

int mppi_reset_controls(mppi_handle *h)
{
  if (!h) return MPPI_ERR_INVALID;
  if (h->pending) {
    int rc = mppi_synchronize(h);
    if (rc) return rc;
  }
  for (int t = 0; t < h->T; t++) {
    h->U[2 * t] = h->cfg.init_control[0];
    h->U[2 * t + 1] = h->cfg.init_control[1];
  }
  h->u_dirty = true;
  h->slid_valid = false;
  return MPPI_OK;
}

int mppi_set_control_seq(mppi_handle *h, const float *U, size_t n)
{
  if (!h || !U || n != 2 * (size_t)h->T) return MPPI_ERR_INVALID;
  if (h->pending) {
    int rc = mppi_synchronize(h);
    if (rc) return rc;
  }
  memcpy(h->U.data(), U, n * sizeof(float));
  h->u_dirty = true;
  h->slid_valid = false;
  return MPPI_OK;
}

int mppi_get_control_seq(mppi_handle *h, float *U, size_t n)
{
  if (!h || !U || n != 2 * (size_t)h->T) return MPPI_ERR_INVALID;
  if (h->pending) {
    int rc = mppi_synchronize(h);
    if (rc) return rc;
  }
  memcpy(U, h->U.data(), n * sizeof(float));
  return MPPI_OK;
}

int mppi_set_control_hist(mppi_handle *h, const float hist[4])
{
  if (!h || !hist) return MPPI_ERR_INVALID;
  if (h->pending) {  // the pending solve's host-side smoothing still reads the old history
    int rc = mppi_synchronize(h);
    if (rc) return rc;
  }
  memcpy(h->hist.data(), hist, 4 * sizeof(float));
  h->u_dirty = true;
  h->slid_valid = false;
  return MPPI_OK;
}

int mppi_savitsky_golay(mppi_handle *h)
{
  if (!h) return MPPI_ERR_INVALID;
  if (h->pending) {
    int rc = mppi_synchronize(h);
    if (rc) return rc;
  }
  const std::vector<float> raw(h->U);
  savgol_host(h, raw.data(), 2, 1);
  h->u_dirty = true;  // the device copy follows with the next solve
  h->slid_valid = false;
  return MPPI_OK;
}

int mppi_get_control_hist(mppi_handle *h, float hist[4])
{
  if (!h || !hist) return MPPI_ERR_INVALID;
  memcpy(hist, h->hist.data(), 4 * sizeof(float));
  return MPPI_OK;
}

int mppi_slide_control_seq(mppi_handle *h, int stride)
{
  if (!h || stride < 0 || stride > h->T) return MPPI_ERR_INVALID;
  if (h->pending) {
    int rc = mppi_synchronize(h);
    if (rc) return rc;
  }
  // stride 0 (a control tick in which no new pose arrived, run_control_loop.cuh:208-216 calls
  // slideControlAndStateSeq only for stride >= 0): the reference's loops copy U onto itself, overwrite
  // nothing with init_u and -- taking its stride != 1 branch with t = -2 -- read control_hist_ from
  // before U_; nothing is defined to change, so nothing changes here.
  if (stride == 0) return MPPI_OK;
  // mppi_controller.cu:527-554
  float *U = h->U.data(), *hist = h->hist.data();
  const int T = h->T;
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
    for (int i = 0; i < 2; i++) U[(T - j) * 2 + i] = h->cfg.init_control[i];
  // the same slide on the device copy, so that solve -> slide -> solve never re-uploads U: the
  // last solve's tail kernel already left the copy slid by optimization_stride in the other buffer
  if (!h->u_dirty) {
    if (h->slid_valid && stride == h->cfg.optimization_stride) {
      h->in_cur = 1 - h->in_cur;
      h->d_in = h->d_in_buf[h->in_cur];
    } else {
      HIPCHK(h, hipSetDevice(h->cfg.device));
      HIPCHK(h, launch_slide(h->d_in, T, stride, h->cfg.init_control[0], h->cfg.init_control[1], work_stream(h)));
    }
  }
  h->slid_valid = false;
  return MPPI_OK;
}

int mppi_seed(mppi_handle *h, uint64_t seed, uint64_t offset)
{
  if (!h) return MPPI_ERR_INVALID;
  HIPCHK(h, hipSetDevice(h->cfg.device));
  int rc = mppi_synchronize(h);
  if (rc) return rc;
  OWN(h);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipStreamSynchronize(h->gstream));
  h->prefetch_valid = false;  // draws of the old sequence
  rc = seed_device(h, seed, offset);
  if (rc) return rc;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return MPPI_OK;
}

int mppi_set_noise(mppi_handle *h, const float *eps, size_t n)
{
  if (!h || !eps) return MPPI_ERR_INVALID;
  const size_t slot = (size_t)h->K * h->T * 2;
  if (n != slot * (size_t)h->cfg.num_iters) return fail(h, MPPI_ERR_INVALID, "noise size != num_iters*K*T*2");
  HIPCHK(h, hipSetDevice(h->cfg.device));
  {
    int rc = mppi_synchronize(h);
    if (rc) return rc;
  }
  OWN(h);
  for (int it = 0; it < h->cfg.num_iters; it++) {
    HIPCHK(h, hipMemcpyAsync(h->d_stage, eps + (size_t)it * slot, slot * sizeof(float),
                             hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, launch_kt_to_tk(h->d_stage, h->d_noise + (size_t)it * slot, h->K, h->T, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
  }
  h->explicit_iters = h->cfg.num_iters;
  return MPPI_OK;
}

int mppi_generate_noise(mppi_handle *h, float *eps_out, size_t n)
{
  if (!h || !eps_out) return MPPI_ERR_INVALID;
  const size_t slot_sz = (size_t)h->K * h->T * 2;
  if (n != slot_sz) return fail(h, MPPI_ERR_INVALID, "n != K*T*2");
  HIPCHK(h, hipSetDevice(h->cfg.device));
  int rc = mppi_synchronize(h);
  if (rc) return rc;
  OWN(h);
  float *buf = nullptr;
  rc = acquire_noise(h, &buf);
  if (rc) return rc;
  h->v_buf = buf;
  HIPCHK(h, launch_tk_to_kt(buf, h->d_stage, h->K, h->T, h->stream));
  HIPCHK(h, hipMemcpyAsync(eps_out, h->d_stage, slot_sz * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return MPPI_OK;
}

int mppi_compute_control_async(mppi_handle *h, const float state[MPPI_STATE_DIM])
{
  if (!h) return MPPI_ERR_INVALID;
  HIPCHK(h, hipSetDevice(h->cfg.device));
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
  for (int i = 0; i < n; i++) {
    if (!hs[i]) return MPPI_ERR_INVALID;
    for (int q = 0; q < i; q++)
      if (hs[q] == hs[i]) return fail(hs[i], MPPI_ERR_INVALID, "the same handle twice in one batch");
  }
  // the host-side halves of this tick (nominal replays, DDP passes) are a solve's length away: the helper thread wakes now
  if (n == 2 && g_host_threads.load(std::memory_order_relaxed) >= 2) host_helper().arm();
  // One launch for all instances where the quad form serves them together (every wave of every group still gets a
  // SIMD of its own: the sum of the groups fits the CUs); otherwise every solve on its own handle's stream, as
  // n calls of mppi_compute_control_async would do.
  mppi_handle *h0 = hs[0];
  bool together = n >= 2 && n <= kMaxBatch;
  int waves = 0;
  for (int i = 0; i < n && together; i++) {
    const mppi_handle *h = hs[i];
    // network model: the four-wavefront form; basis-function model: its three-wavefront form (in-kernel generator)
    const bool form_ok = h->basis ? (h0->basis && bf_waves(h) == 3)
                                  : (!h0->basis && use_mfma(h) && use_mfma(h0) &&
                                     (effective_block(h) == 512 || is_row(effective_block(h))) &&
                                     effective_block(h) == effective_block(h0) &&
                                     h->hidden == h0->hidden && h->n_hidden == h0->n_hidden);
    together = form_ok && h->cfg.device == h0->cfg.device && h->cfg.num_iters == h0->cfg.num_iters &&
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
  HIPCHK(h0, hipSetDevice(h0->cfg.device));
  const hipStream_t S = batch_stream(h0->cfg.device);
  if (!S) return fail(h0, MPPI_ERR_HIP, "no batch stream");
  const int iters = h0->cfg.num_iters;
  for (int i = 0; i < n; i++) {
    mppi_handle *h = hs[i];
    int rc = wait_pending(h);  // finish a previous asynchronous solve first
    if (rc) return rc;
    if (h->explicit_iters > 0 && h->explicit_iters != iters)
      return fail(h, MPPI_ERR_STATE, "explicit noise holds a different number of iterations");
    if (h->order_stream != S) {  // first batched solve after work on the handle's own streams: let that finish
      HIPCHK(h, hipStreamSynchronize(h->order_stream ? h->order_stream : h->stream));
      HIPCHK(h, hipStreamSynchronize(h->gstream));
      h->order_stream = S;
    }
    rc = upload_controls_if_dirty(h, S);
    if (rc) return rc;
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
    hipError_t e = h0->basis ? launch_rollout_bf_batch(qb, S)
                   : is_row(effective_block(h0)) ? launch_rollout_row_batch(qb, effective_block(h0) == 901, S)
                                                : launch_rollout_quad_batch(h0->hidden, h0->n_hidden, qb, S);
    if (e == hipSuccess) e = launch_solve_tail_batch(tl, n, S);
    if (e != hipSuccess) return fail(h0, MPPI_ERR_HIP, "batched launch", e);
  }
  for (int i = 0; i < n; i++) {
    hs[i]->explicit_iters = 0;
    hs[i]->pending = true;
    hs[i]->pending_timed = false;
  }
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
  if (!lockstep) {
    const int rc = mppi_nominal_traj(ha, state_a, state_seq_a, control_seq_a);
    return rc ? rc : mppi_nominal_traj(hb, state_b, state_seq_b, control_seq_b);
  }
  for (mppi_handle *h : {ha, hb})
    if (h->pending) {
      const int rc = mppi_synchronize(h);
      if (rc) return rc;
    }
  if (g_host_threads.load(std::memory_order_relaxed) >= 2) {  // one replay per thread (mppi_set_host_threads)
    int rca = MPPI_OK, rcb = MPPI_OK;
    host_helper().run_pair([&] { rcb = mppi_nominal_traj(hb, state_b, state_seq_b, control_seq_b); },
                           [&] { rca = mppi_nominal_traj(ha, state_a, state_seq_a, control_seq_a); });
    return rca ? rca : rcb;
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
    host_helper().run_pair([&] { rcb = mppi_compute_feedback_gains(hb, state_b, target_state_seq_b, target_control_seq_b); },
                           [&] { rca = mppi_compute_feedback_gains(ha, state_a, target_state_seq_a, target_control_seq_a); });
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

int mppi_debug_cost_raster(mppi_handle *h, float x, float y, float heading, int width_m, int height_m,
                           int ppm, float *out, size_t n)
{
  if (!h || !out || width_m <= 0 || height_m <= 0 || ppm <= 0) return MPPI_ERR_INVALID;
  const size_t W = (size_t)width_m * ppm, H = (size_t)height_m * ppm;
  if (W > 8192 || H > 8192 || n != W * H) return fail(h, MPPI_ERR_INVALID, "n != (width_m*ppm) * (height_m*ppm)");
  if (!h->have_map) return fail(h, MPPI_ERR_STATE, "mppi_set_costmap has not been called");
  HIPCHK(h, hipSetDevice(h->cfg.device));
  OWN(h);
  float *d = nullptr;
  HIPCHK(h, hipMalloc(&d, n * sizeof(float)));
  CostArgs c;
  fill_cost_args(h, c);
  hipError_t e = hipMemsetAsync(d, 0, n * sizeof(float), h->stream);
  if (e == hipSuccess) e = launch_debug_cost(c, x, y, heading, width_m, height_m, ppm, d, h->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(out, d, n * sizeof(float), hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  (void)hipFree(d);
  if (e != hipSuccess) return fail(h, MPPI_ERR_HIP, "mppi_debug_cost_raster", e);
  return MPPI_OK;
}

int mppi_enable_stage_timing(mppi_handle *h, int on)
{
  if (!h) return MPPI_ERR_INVALID;
  h->timing = on != 0;
  h->timing_every = on > 1 ? on : 1;  // on = N > 1: sample every Nth solve
  h->timing_count = 0;
  return MPPI_OK;
}

int mppi_reset_stage_times(mppi_handle *h)
{
  if (!h) return MPPI_ERR_INVALID;
  h->acc = mppi_stage_times{};
  return MPPI_OK;
}

int mppi_get_stage_times(mppi_handle *h, mppi_stage_times *out)
{
  if (!h || !out) return MPPI_ERR_INVALID;
  if (h->pending) {
    int rc = mppi_synchronize(h);
    if (rc) return rc;
  }
  *out = h->acc;
  return MPPI_OK;
}

const char *mppi_rollout_variant(const mppi_handle *h)
{
  if (!h) return "";
  if (h->basis) return bf_waves(h) == 3 ? "basis_funcs25_valu_3w" : bf_waves(h) == 2 ? "basis_funcs25_valu_2w" : "basis_funcs25_valu";
  if (!use_mfma(h)) return use_valu_reg(h) ? "valu_reg_lds" : "valu_lds";
  static thread_local char buf[64];
  const int b = effective_block(h);
  if (b == 1040)
    snprintf(buf, sizeof(buf), "mfma16x16x4_h%d_l%d_multi4u%s", h->hidden, h->n_hidden, multi_gen(h) ? "_gen" : "");
  else if (b > 1000)
    snprintf(buf, sizeof(buf), "mfma16x16x4_h%d_l%d_multi%d%s", h->hidden, h->n_hidden, b - 1000,
             multi_gen(h) ? "_gen" : "");
  else if (is_row(b))
    snprintf(buf, sizeof(buf), "valu_row8w%s_h%d_l%d", b == 901 ? "_tree" : "", h->hidden, h->n_hidden);
  else if (is_row64(b))
    snprintf(buf, sizeof(buf), "valu_row64_r%d_tree_h%d_l%d", b - 900, h->hidden, h->n_hidden);
  else
    snprintf(buf, sizeof(buf), "mfma16x16x4_h%d_l%d_%s", h->hidden, h->n_hidden,
             b == 512 ? "quad4w" : b == 800 ? (multi_gen(h) ? "oct8w_gen" : "oct8w") : (b == 256 ? "fused_b256" : "fused_b64"));
  return buf;
}

int mppi_set_rollout_variant(mppi_handle *h, const char *name)
{
  if (!h || !name) return MPPI_ERR_INVALID;
  if (strcmp(name, "auto") == 0) {
    h->variant_pref = 0;
    h->block_threads = 0;
  }
  else if (strcmp(name, "mfma") == 0) {
    if (!h->mfma_ok) return fail(h, MPPI_ERR_UNSUPPORTED, "MFMA variant needs 6-HxN-4 with H in {32,64}, N in {2,4}");
    h->variant_pref = 1;
  } else if (strcmp(name, "valu") == 0) h->variant_pref = 2;
  else if (strcmp(name, "valu_lds") == 0) h->variant_pref = 3;
  else if (strcmp(name, "quad") == 0) h->block_threads = 512;
  else if (strcmp(name, "bf3") == 0) {
    if (!h->basis) return fail(h, MPPI_ERR_UNSUPPORTED, "bf3 is a form of the basis-function model");
    h->block_threads = 768;
  }
  else if (strcmp(name, "row") == 0 || strcmp(name, "row_exact") == 0 || strcmp(name, "row_tree") == 0) {
    if (!h->mfma_ok || !row_variant_supported(h->hidden, h->n_hidden))
      return fail(h, MPPI_ERR_UNSUPPORTED, "row form exists for 6-32x2-4");
    h->block_threads = strcmp(name, "row_tree") == 0 ? 901 : 900;
  }
  else if (strcmp(name, "row64") == 0 || strcmp(name, "row64_r8") == 0 || strcmp(name, "row64_r16") == 0) {
    if (!h->mfma_ok || !row64_variant_supported(h->hidden, h->n_hidden))
      return fail(h, MPPI_ERR_UNSUPPORTED, "row64 form exists for 6-64x2-4 and 6-64x4-4");
    // 8 rollouts per group (one dynamics wave per SIMD) while every such group has a CU of its own, else 16
    const int r = name[5] == 0 ? ((h->K / 8 <= h->num_simds / 4) ? 8 : 16) : (name[7] == '8' ? 8 : 16);
    h->block_threads = 900 + r;
  }
  else if (strcmp(name, "oct") == 0 || strcmp(name, "oct_gen") == 0) {
    if (!h->mfma_ok || !oct_variant_supported(h->hidden, h->n_hidden))
      return fail(h, MPPI_ERR_UNSUPPORTED, "oct form exists for 6-64x2-4 and 6-64x4-4");
    h->block_threads = 800;
    h->multi_standalone_noise = name[3] != 0;
  }
  else if (strcmp(name, "multi4u") == 0 || strcmp(name, "multi4u_gen") == 0) {  // ND = 4, six waves (one cost wave)
    if (h->K % 64 != 0) return fail(h, MPPI_ERR_UNSUPPORTED, "multi form needs K to be a multiple of 16 ND");
    if (!h->mfma_ok || !multi_variant_supported(h->hidden, h->n_hidden))
      return fail(h, MPPI_ERR_UNSUPPORTED, "multi form exists for 6-32x2-4, 6-32x4-4 and 6-64x2-4");
    h->block_threads = 1040;
    h->multi_standalone_noise = name[7] != 0;
  }
  else if (strncmp(name, "multi", 5) == 0) {
    const int nd = name[5] - '0';
    const bool gen = strcmp(name + 6, "_gen") == 0;
    if ((nd != 1 && nd != 2 && nd != 4) || (name[6] != 0 && !gen)) return fail(h, MPPI_ERR_INVALID, "unknown variant");
    if (h->K % (16 * nd) != 0) return fail(h, MPPI_ERR_UNSUPPORTED, "multi form needs K to be a multiple of 16 ND");
    if (!h->mfma_ok || !multi_variant_supported(h->hidden, h->n_hidden))
      return fail(h, MPPI_ERR_UNSUPPORTED, "multi form exists for 6-32x2-4, 6-32x4-4 and 6-64x2-4");
    h->block_threads = 1000 + nd;
    h->multi_standalone_noise = gen;
  }
  else if (strcmp(name, "fused") == 0 || strcmp(name, "block256") == 0) h->block_threads = 256;
  else if (strcmp(name, "block64") == 0) h->block_threads = 64;
  else return fail(h, MPPI_ERR_INVALID, "unknown variant");
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

/* How long a blocking call polls for a solve's result block before it reports MPPI_ERR_HIP (default 30 s). */
int mppi_set_wait_timeout(mppi_handle *h, double seconds)
{
  if (!h || !(seconds > 0.0)) return MPPI_ERR_INVALID;
  h->wait_timeout_s = seconds;
  return MPPI_OK;
}

/* Debug/test entry (not part of the drop-in surface): one wavefront role of the multi-wavefront rollout
 * kernels starts with an exhausted poll budget (it never waits), to prove that a starved wave -- whichever
 * it is -- turns into MPPI_ERR_HIP and not into finite, wrong costs. */
int mppi_debug_inject_handover_fault(mppi_handle *h, int wave, int spin_budget)
{
  if (!h || wave < 0 || wave > 12 || spin_budget < 0) return MPPI_ERR_INVALID;
  h->fault_wave = wave;
  h->spin_budget = spin_budget;
  return MPPI_OK;
}

/* Debug/test entry (not part of the drop-in surface): state derivatives of n (state, control)
 * pairs through the same device functions as the rollout kernel. */
int mppi_debug_dynamics(mppi_handle *h, int n, const float *states, const float *controls, float *ders)
{
  if (!h || n <= 0 || !states || !controls || !ders) return MPPI_ERR_INVALID;
  if (!h->have_nn) return fail(h, MPPI_ERR_STATE, "mppi_set_nn_params has not been called");
  HIPCHK(h, hipSetDevice(h->cfg.device));
  OWN(h);
  float *d_s = nullptr, *d_u = nullptr, *d_o = nullptr;
  HIPCHK(h, hipMalloc(&d_s, sizeof(float) * 7 * n));
  HIPCHK(h, hipMalloc(&d_u, sizeof(float) * 2 * n));
  HIPCHK(h, hipMalloc(&d_o, sizeof(float) * 7 * n));
  hipError_t e = hipMemcpy(d_s, states, sizeof(float) * 7 * n, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_u, controls, sizeof(float) * 2 * n, hipMemcpyHostToDevice);
  if (e == hipSuccess)
    e = h->basis ? launch_dynamics_bf(h->d_theta, d_s, d_u, d_o, n, h->stream)
        : use_mfma(h) ? launch_dynamics_mfma(h->hidden, h->n_hidden, h->d_wpack, d_s, d_u, d_o, n,
                                           h->cfg.negate_yaw_der ? 1 : 0, h->stream)
                    : launch_dynamics_valu(h->net, h->d_theta, d_s, d_u, d_o, n,
                                           h->cfg.negate_yaw_der ? 1 : 0, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (e == hipSuccess) e = hipMemcpy(ders, d_o, sizeof(float) * 7 * n, hipMemcpyDeviceToHost);
  (void)hipFree(d_s);
  (void)hipFree(d_u);
  (void)hipFree(d_o);
  if (e != hipSuccess) return fail(h, MPPI_ERR_HIP, "mppi_debug_dynamics", e);
  return MPPI_OK;
}

}  // extern "C"
