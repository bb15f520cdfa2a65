// mppi_kernels.hpp -- launchers implemented in the .hip kernel files.
#pragma once
#include "mppi_device.hpp"

#include <hip/hip_ext.h>

namespace mppi {

// Rollout kernels are launched through this macro.  When the caller has set the two thread-local events
// (mppi_enable_stage_timing: the stage-timing pass of bench.py), the launch goes through
// hipExtLaunchKernelGGL, which stamps them with the begin / end of THIS kernel's dispatch -- the figure
// rocprofv3 --kernel-trace reports -- instead of the time between two marker packets around it (which on this
// runtime is 3-4 us longer).  Otherwise a plain launch.
extern thread_local hipEvent_t tl_kernel_start, tl_kernel_stop;
#define MPPI_LAUNCH_ROLLOUT(kern, grid, block, lds, stream, ...)                                                   \
  do {                                                                                                             \
    if (::mppi::tl_kernel_start != nullptr)                                                                        \
      hipExtLaunchKernelGGL(kern, grid, block, lds, stream, ::mppi::tl_kernel_start, ::mppi::tl_kernel_stop, 0,    \
                            __VA_ARGS__);                                                                          \
    else                                                                                                           \
      hipLaunchKernelGGL(kern, grid, block, lds, stream, __VA_ARGS__);                                             \
  } while (0)

// rollout_mfma.hip
bool mfma_variant_supported(int hidden, int n_hidden);
int mfma_pack_floats_per_lane(int hidden, int n_hidden);
hipError_t launch_rollout_mfma(int hidden, int n_hidden, const RolloutArgs &a, int block_threads,
                               hipStream_t stream);
// several instances in one launch of the quad form (every instance: 6 -> hidden x n_hidden -> 4)
hipError_t launch_rollout_quad_batch(int hidden, int n_hidden, const QuadBatchArgs &b, hipStream_t stream);
hipError_t launch_dynamics_mfma(int hidden, int n_hidden, const float *wpack, const float *states,
                                const float *controls, float *ders, int n, int negate_yaw_der,
                                hipStream_t stream);

// rollout_multi.hip: nd dynamics waves (16 rollouts each) + cost wave + control wave per workgroup, nd in {1, 2, 4}
bool multi_variant_supported(int hidden, int n_hidden);
hipError_t launch_rollout_multi(int hidden, int n_hidden, const RolloutArgs &a, int nd, hipStream_t stream);

// rollout_oct.hip: four dynamics waves (one M tile of a 64-wide net each) + pose, cost, control and noise wave per
// 16 rollouts
bool oct_variant_supported(int hidden, int n_hidden);
hipError_t launch_rollout_oct(int hidden, int n_hidden, const RolloutArgs &a, hipStream_t stream);

// rollout_row.hip: latency form of 6-32-32-4 on the vector ALU -- four dynamics waves (four rollouts each) + pose, cost,
// control and noise wave per 16 rollouts; a.wpack = the weights in register order (pack_row_weights, mppi_abi.hip)
bool row_variant_supported(int hidden, int n_hidden);
int row_pack_floats();
// tree: the output layer as own-activation partials + a DPP butterfly (NOT the reference's summation order; the AUTOMATIC form
// for 6-32-32-4 up to 8192 rollouts: "row_exact" / "mfma" restore the order)
hipError_t launch_rollout_row(int hidden, int n_hidden, const RolloutArgs &a, bool tree, hipStream_t stream);
hipError_t launch_rollout_row_batch(const QuadBatchArgs &b, bool tree, hipStream_t stream);  // grid (groups, instances)

// rollout_row64.hip: latency form of 64-wide nets on the vector ALU -- r / 2 dynamics waves (two rollouts of 32 lanes each) +
// pose, cost, control and noise wave per r = 16 rollouts; hidden layers' weights from LDS, output layer as a butterfly
// (NOT the reference's summation order; never chosen automatically: config 4's vector-ALU A/B arm); a.wpack = pack_row64_weights (mppi_abi.hip)
bool row64_variant_supported(int hidden, int n_hidden);
int row64_pack_floats(int n_hidden);
hipError_t launch_rollout_row64(int hidden, int n_hidden, const RolloutArgs &a, int r, hipStream_t stream);

// rollout_m44.hip: latency form of 64-wide nets on v_mfma_f32_4x4x1 with A-matrix broadcast -- four dynamics waves (four
// rollouts each, all hidden weights in registers) + pose, cost, control and noise wave per 16 rollouts; output layer as a
// butterfly, hidden layers (split) as two accumulation chains -- NOT the reference's summation order, and the AUTOMATIC form
// for 64-wide nets up to 8192 rollouts ("m44_chain": hidden layers in the reference's order; "mfma": every layer);
// a.wpack = pack_m44_weights (mppi_abi.hip)
bool m44_variant_supported(int hidden, int n_hidden);
int m44_pack_floats(int n_hidden);
hipError_t launch_rollout_m44(int hidden, int n_hidden, const RolloutArgs &a, bool split, hipStream_t stream);  // split: two chains per hidden layer

// rollout_valu.hip (generic vector-ALU kernel, any layer list)
struct NetDesc {
  int n_layers;
  int layers[8];
  int max_width;
  int num_params;
};
size_t valu_lds_bytes(const NetDesc &net);
hipError_t launch_rollout_valu(const NetDesc &net, const RolloutArgs &a, hipStream_t stream);
// register/scalar-operand vector-ALU kernel for 6 -> H x NHID -> 4 (theta with pre-scaled hidden biases)
bool valu_reg_supported(int hidden, int n_hidden);
hipError_t launch_rollout_valu_reg(int hidden, int n_hidden, const RolloutArgs &a, hipStream_t stream);
hipError_t launch_dynamics_valu(const NetDesc &net, const float *theta, const float *states,
                                const float *controls, float *ders, int n, int negate_yaw_der,
                                hipStream_t stream);

// rollout_bf.hip (GeneralizedLinear basis-function dynamics, W[4][25] in a.wpack)
hipError_t launch_rollout_bf(const RolloutArgs &a, int waves, hipStream_t stream);  // waves per 64 rollouts: 1, 2, 3
// several instances of the three-wave form in one launch (grid: groups of 64 rollouts x instances)
hipError_t launch_rollout_bf_batch(const QuadBatchArgs &b, hipStream_t stream);
hipError_t launch_dynamics_bf(const float *W, const float *states, const float *controls, float *ders, int n,
                              hipStream_t stream);

// solve_kernels.hip
// everything of one solve iteration after the rollout (solve_tail_kernel up to 4096 rollouts, solve_tail_stream_kernel beyond)
struct TailLaunch {
  const float *costs, *V, *hist;
  float *U, *w, *scal, *res, *slid;
  unsigned *counter;
  int K, T, last_iter, slide_stride;
  unsigned seq;
  float gamma, init0, init1;
  // K > 4096 (solve_tail_stream_kernel): granule buffers (gx: 3 x 64 exchange granules; gpart: [T][K/64][2] chain results,
  // 8 B each, zero when allocated), the tag of this launch's granules (never 0, never the tag of an earlier launch on these
  // buffers), the deadline of every in-launch wait in 100 MHz ticks, and the tests' fault role (0: none)
  unsigned long long *ug = nullptr;  // [T][2] granules of the raw weighted mean (every form)
  int no_device_copy = 0;            // chained ticks: publish only
  float *hist_out = nullptr;         // chained ticks, last solve: where the smoothing workgroup leaves hist[4]
  unsigned long long *gx = nullptr, *gpart = nullptr;
  unsigned epoch = 0, poll_ticks = 0;
  int fault = 0;
  // the rollout launch's minimum cost (mppi_device.hpp: publish_min_cost) and that launch's tag; nullptr: the tail reduces the costs itself
  const unsigned long long *min_cost = nullptr;
  unsigned min_cost_tag = 0;
};
// does a solve of K rollouts run the one-launch streaming tail (in-launch column exchanges; wait_pending then also checks the
// published rows for the NaN a timed-out wait leaves)?
bool tail_is_stream(int K);
constexpr int kTailExchangeGranules = 3 * 64 + 32 * 16 + 32 * 64;  // solve_kernels.hip: 3 x kMaxChunks + kBcastReplicas lines + kBcastReplicas x kMaxChunks
hipError_t launch_solve_tail(const TailLaunch &l, hipStream_t stream);
// the tails of n <= kMaxBatch instances (K <= 4096 each) in one launch
hipError_t launch_solve_tail_batch(const TailLaunch *l, int n, hipStream_t stream);
hipError_t launch_debug_cost(const CostArgs &c, float x, float y, float heading, int width_m, int height_m,
                             int ppm, float *out, hipStream_t stream);
hipError_t launch_slide(float *in, int T, int stride, float init0, float init1, hipStream_t stream);
hipError_t launch_kt_to_tk(const float *src, float *dst, int K, int T, hipStream_t stream);
hipError_t launch_tk_to_kt(const float *src, float *dst, int K, int T, hipStream_t stream);

// noise_mrg32k3a.hip
hipError_t launch_noise(const uint32_t *rng_in, uint32_t *rng_out, const uint32_t *jump, int K, int T,
                        int L, int C, float *eps, hipStream_t stream);
hipError_t launch_noise_init(uint32_t *rng, int K, const uint32_t base[6], const uint32_t *sub,
                             int sub_bits, const uint32_t *one, uint64_t offset, hipStream_t stream);

}  // namespace mppi
