// abi_internal.hpp -- what the files behind include/mppi_hip.h share: the handle, the kernel-form codes and the internal
// functions of one another.  mppi_abi.hip: handle life cycle, setters, getters; abi_forms.hip: which kernel form runs
// (selection table, names); abi_pack.hip: weight images and generator tables; abi_solve.hip: the solve pipeline (noise,
// rollout + tail launches, result polling, batched solves); abi_host.hip: the host-side halves of a tick (nominal replays,
// DDP feedback gains, the helper thread).
#pragma once
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

namespace mppi_abi {

// Kernel forms of the rollout (abi_forms.hip has the table that picks one, and what each is).
enum class Form : int {
  Auto = 0,
  Fused64, Fused256,                // rollout_mfma.hip: one wave per 16 rollouts does everything (workgroups of 1 / 4 waves)
  Quad,                             // rollout_mfma.hip: network split over two waves + cost + control wave per 16 rollouts
  Oct,                              // rollout_oct.hip: 64-wide nets, four dynamics waves (one M tile each) + four riders
  Multi2, Multi4,                   // rollout_multi.hip: ND dynamics waves of 16 rollouts + riders
  Multi4Tree,                       // ... ND = 4 with the output layer as a butterfly over a rollout's four lanes
  Row, RowTree,                     // rollout_row.hip: 6-32-32-4 on the vector ALU; Tree: butterfly output layer
  Row64R16,                         // rollout_row64.hip: 64-wide nets on the vector ALU, 16 rollouts per group
  M44, M44Chain,                    // rollout_m44.hip: 64-wide nets on v_mfma_f32_4x4x1 with A-broadcast; hidden layers as two
                                    // accumulation chains (the automatic form) / Chain: one, the reference's order
  ValuReg, ValuLds,                 // rollout_valu.hip: throughput-style vector kernels (any layer list: ValuLds)
  Bf1, Bf2, Bf3,                    // rollout_bf.hip: basis-function model, waves per 64 rollouts
};
enum class Pref : int { Auto = 0, Mfma, Valu, ValuLds };

struct Events {
  // e[0..3]: markers on the handle's stream before noise / before rollout / after rollout / after tail;
  // e[4], e[5]: begin and end of the rollout kernel's own dispatch (hipExtLaunchKernelGGL, MPPI_LAUNCH_ROLLOUT)
  hipEvent_t e[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
};
}  // namespace mppi_abi
using mppi_abi::Events;


struct mppi_handle {
  using NetDesc = mppi::NetDesc;
  using DdpResult = mppi::DdpResult;
  using HostNetFma = mppi::HostNetFma;
  mppi_config cfg{};
  int K = 0, T = 0, k99 = 0;
  float dt = 0.0f;
  NetDesc net{};
  bool mfma_ok = false;
  int hidden = 0, n_hidden = 0;
  mppi_abi::Pref pref = mppi_abi::Pref::Auto;   // "auto" | "mfma" | "valu" | "valu_lds"
  mppi_abi::Form forced = mppi_abi::Form::Auto;  // a kernel form asked for by name (mppi_set_rollout_variant); Auto: the selection table
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
  hipStream_t batch_s = nullptr;  // the device's batch stream, looked up once (mppi_compute_control_batch)
  unsigned *d_counter = nullptr;  // [1 + T] arrival counters of the tail kernel
  float *d_part = nullptr;        // [T][K/64][2] chain results of the tail kernel when K > 4096
  // K > 4096 (one-launch streaming tail): d_part holds 8-byte {value, epoch} granules; d_gx the granules of
  // the column exchanges; tail_epoch the tag of the last tail launch; tail_poll_ticks the deadline of its waits (100 MHz ticks)
  unsigned long long *d_gx = nullptr;
  // the rollout launch's minimum cost on its way to the tail kernel (mppi_device.hpp: publish_min_cost): kMinCostLines keys, all
  // ones when allocated; the tag of the latest rollout launch that wrote d_costs (next_min_cost_tag); MPPI_MIN_COST=0: not used
  unsigned long long *d_min_cost = nullptr;
  unsigned min_cost_tag = 0;
  bool use_min_cost = true;
  unsigned long long *d_ug = nullptr;  // [T][2] granules of the raw weighted mean: row workgroups -> the smoothing workgroup
  unsigned tail_epoch = 0, tail_poll_ticks = 2000000;  // 20 ms
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
  float *d_m44pack = nullptr;    // 64-wide nets: image of rollout_m44.hip
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
  // chained control ticks (mppi_control_ticks, abi_solve.hip): the gate block of the solve enqueued one tick ahead --
  // kGateReplicas copies of [state[7], gate word], device memory the host stores into through the PCIe BAR where the
  // platform allows it (gate_bar), else host-mapped memory; gate_cpu is the pointer the host writes, d_gate what the kernel reads
  unsigned *gate_cpu = nullptr, *d_gate = nullptr;
  bool gate_bar = false;
  bool chain = true;   // mppi_debug_set_chained_ticks
  bool ahead = false;  // a gated solve is enqueued behind the pending one
  float *ahead_vbuf = nullptr;  // where that solve leaves its applied controls
  bool timed_out = false;        // a wait ran out of time and that solve's device work may still run: recover_timed_out (abi_solve.hip)
  bool no_result = false;        // the last solve was lost (timeout): mppi_get_results refuses until a solve completes
  bool timing = false;
  int timing_every = 1;      // record stage events on every Nth solve only (events add launch gaps)
  unsigned timing_count = 0;
  std::vector<Events> ev;  // one set per iteration
  mppi_stage_times acc{};
  std::string err;
};

namespace mppi_abi {
using namespace mppi;


#define HIPCHK(h, call)                                                  \
  do {                                                                   \
    hipError_t e__ = (call);                                             \
    if (e__ != hipSuccess) return fail((h), MPPI_ERR_HIP, #call, e__);   \
  } while (0)
// The calling thread's current device is `dev` afterwards: hipGetDevice (a thread-local read, 0.07 us) and hipSetDevice only
// when it differs -- a control loop calls the solve entries from one thread whose device never changes.
inline hipError_t ensure_device(int dev)
{
  int cur = -1;
  if (hipGetDevice(&cur) == hipSuccess && cur == dev) return hipSuccess;
  return hipSetDevice(dev);
}
#define OWN(h)                      \
  do {                              \
    int rc__ = own_stream(h);       \
    if (rc__) return rc__;          \
  } while (0)

int compute_k99(int K);
std::vector<float> pack_mfma_weights(const std::vector<float> &theta, int H, int NHID);
std::vector<float> pack_row_weights(const std::vector<float> &theta);
std::vector<float> pack_row64_weights(const std::vector<float> &theta, int NHID);
std::vector<float> pack_m44_weights(const std::vector<float> &theta, int NHID);
int seed_device(mppi_handle *h, uint64_t seed, uint64_t offset);
int upload_rng_tables(mppi_handle *h);
bool use_mfma(const mppi_handle *h);
bool use_valu_reg(const mppi_handle *h);
Form form_of(const mppi_handle *h);
bool form_generator_noise(const mppi_handle *h);
bool has_noise_wave(const mppi_handle *h);
int form_bf_waves(Form f);
int form_multi_nd(Form f);
int form_fused_threads(Form f);
inline bool form_is_row(Form f) { return f == Form::Row || f == Form::RowTree; }
inline bool form_is_row64(Form f) { return f == Form::Row64R16; }
hipStream_t batch_stream(int device);
void fill_cost_args(const mppi_handle *h, CostArgs &c);
void fill_rollout_args(const mppi_handle *h, const float *state, float *noise, RolloutArgs &a);
// a rollout launch that is followed by its tail kernel: the tag its minimum cost travels under (a.min_cost, a.min_cost_tag)
int tag_min_cost(mppi_handle *h, RolloutArgs &a, hipStream_t stream);
int launch_rollout(mppi_handle *h, const RolloutArgs &a);
int check_ready(mppi_handle *h);
int launch_generator(mppi_handle *h, float *dst);
int acquire_noise(mppi_handle *h, float **buf_out);
int prefetch_noise(mppi_handle *h);
int upload_controls_if_dirty(mppi_handle *h, hipStream_t stream);
void savgol_host(mppi_handle *h, const float *src, int stride, int off1);
bool wants_slid_copy(const mppi_handle *h);
TailLaunch tail_launch(const mppi_handle *h, const float *V, bool last);
int fail(mppi_handle *h, int code, const char *what, hipError_t e = hipSuccess);
int recover_timed_out(mppi_handle *h);
bool gen_beside_rollout(const mppi_handle *h);
int own_stream(mppi_handle *h);
void free_all(mppi_handle *h);

// where small follow-up work (upload of U, the slide kernel) goes: behind the handle's latest work, wherever it is
inline hipStream_t work_stream(const mppi_handle *h) { return h->order_stream ? h->order_stream : h->stream; }

// abi_host.hip: the helper thread of the paired host work (mppi_set_host_threads)
extern std::atomic<int> g_host_threads;
void host_helper_arm();

}  // namespace mppi_abi
