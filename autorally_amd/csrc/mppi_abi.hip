// mppi_abi.hip -- host side of libmppi_hip.so: the C ABI of include/mppi_hip.h.
//
// Owns one HIP stream and all device memory of a solver instance, enqueues one MPPI solve
// (PI/mppi_controller.cu:600-671) as: [H2D U|hist] -> noise -> rollout -> weights -> weighted
// reduction -> Savitzky-Golay -> [D2H scal|U], with ONE stream synchronisation per solve (the
// reference has three plus five blocking parameter uploads).
// There is no CPU fallback anywhere in this file: without a gfx950 device every compute entry
// point returns an error.
#include "abi_internal.hpp"

using namespace mppi;
using namespace mppi_abi;


namespace mppi {
thread_local hipEvent_t tl_kernel_start = nullptr, tl_kernel_stop = nullptr;
}

namespace mppi_abi {

int fail(mppi_handle *h, int code, const char *what, hipError_t e)
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

// Called by every entry point that enqueues on the handle's OWN stream or relies on a synchronise of that stream
// having seen all of the handle's device work: if the handle's last work went to the batch stream, wait for it
// (only on a batch -> single transition: setup calls, result vectors, a stand-alone solve after a batched one).
int own_stream(mppi_handle *h)
{
  if (h->order_stream && h->order_stream != h->stream) HIPCHK(h, hipStreamSynchronize(h->order_stream));
  h->order_stream = h->stream;
  return MPPI_OK;
}

void free_all(mppi_handle *h)
{
  if (!h) return;
  float *fp[] = {h->d_theta_s, h->d_in_buf[0], h->d_in_buf[1], h->d_scal, h->d_noise, h->d_stage, h->d_costs,
                 h->d_w, h->d_theta, h->d_wpack, h->d_map, h->d_part, h->d_rowpack, h->d_row64pack, h->d_m44pack, h->d_cap};
  for (float *p : fp)
    if (p) (void)hipFree(p);
  if (h->d_invt) (void)hipFree(h->d_invt);
  if (h->d_counter) (void)hipFree(h->d_counter);
  if (h->d_gx) (void)hipFree(h->d_gx);
  if (h->d_min_cost) (void)hipFree(h->d_min_cost);
  if (h->d_ug) (void)hipFree(h->d_ug);
  if (h->gate_cpu) { if (h->gate_bar) (void)hipFree(h->gate_cpu); else (void)hipHostFree(h->gate_cpu); }
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

}  // namespace mppi_abi

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
  {
    // the generator's stream at the LOWEST priority: where a rollout kernel on the handle's stream and a generator launch
    // become ready together (chained ticks: both wait for the same tail kernel) the rollout's workgroups are placed first and
    // the generator fills what they leave, not the other way round (MPPI_GSTREAM_PRIO=0: default priority, for the A/B)
    int lo = 0, hi = 0;
    const char *e = getenv("MPPI_GSTREAM_PRIO");
    if ((e == nullptr || atoi(e) != 0) && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && lo != hi)
      CR(hipStreamCreateWithPriority(&h->gstream, hipStreamNonBlocking, lo));
    else
      CR(hipStreamCreateWithFlags(&h->gstream, hipStreamNonBlocking));
  }
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
  CR(hipMalloc(&h->d_ug, sizeof(unsigned long long) * 2 * (size_t)h->T));
  CR(hipMemset(h->d_ug, 0, sizeof(unsigned long long) * 2 * (size_t)h->T));
  CR(hipMalloc(&h->d_counter, sizeof(unsigned) * (1 + (size_t)h->T)));
  CR(hipMemset(h->d_counter, 0, sizeof(unsigned) * (1 + (size_t)h->T)));
  if (h->K > 4096) {
    // chain results of a row spread over several workgroups: 8-byte {value, tag} granules
    // (solve_tail_stream_kernel); zeroed once -- a granule counts only with the tag of the launch that reads it
    const size_t part_bytes = sizeof(unsigned long long) * (size_t)h->T * (h->K / 64) * 2;
    CR(hipMalloc(&h->d_part, part_bytes));
    CR(hipMemset(h->d_part, 0, part_bytes));
    CR(hipMalloc(&h->d_gx, sizeof(unsigned long long) * kTailExchangeGranules));
    CR(hipMemset(h->d_gx, 0, sizeof(unsigned long long) * kTailExchangeGranules));
  }
  CR(hipMalloc(&h->d_min_cost, sizeof(unsigned long long) * kMinCostLines * kMinCostStride));
  CR(hipMemset(h->d_min_cost, 0xFF, sizeof(unsigned long long) * kMinCostLines * kMinCostStride));  // "no launch's key"
  if (const char *e = getenv("MPPI_MIN_COST")) h->use_min_cost = atoi(e) != 0;  // A/B: 0 = every tail kernel reduces the costs itself
  if (const char *e = getenv("MPPI_MIN_COST_TAG")) h->min_cost_tag = (unsigned)strtoul(e, nullptr, 0);  // tests: start near the wrap
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
  if (h->mfma_ok && m44_variant_supported(h->hidden, h->n_hidden))
    CR(hipMalloc(&h->d_m44pack, sizeof(float) * (size_t)m44_pack_floats(h->n_hidden)));
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
  {
    // gate block of the chained control ticks: fine-grained device memory the host can store into (large BAR), else host-mapped
    const size_t gate_bytes = sizeof(float) * gate_block_floats(h->T);  // replicas of [state, gate word], then U[T][2], hist[4]
    void *gp = nullptr;
    if (prop.isLargeBar && hipExtMallocWithFlags(&gp, gate_bytes, hipDeviceMallocFinegrained) == hipSuccess && gp != nullptr) {
      h->gate_cpu = h->d_gate = static_cast<unsigned *>(gp);
      h->gate_bar = true;
      CR(hipMemset(gp, 0, gate_bytes));
    } else {
      (void)hipGetLastError();
      CR(hipHostMalloc(&gp, gate_bytes, hipHostMallocMapped));
      h->gate_cpu = static_cast<unsigned *>(gp);
      memset(gp, 0, gate_bytes);
      void *dp = nullptr;
      CR(hipHostGetDevicePointer(&dp, gp, 0));
      h->d_gate = static_cast<unsigned *>(dp);
    }
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
  if (h->d_m44pack) {
    const std::vector<float> pk = pack_m44_weights(h->theta, h->n_hidden);
    if ((int)pk.size() != m44_pack_floats(h->n_hidden)) return fail(h, MPPI_ERR_INVALID, "m44 image size");
    HIPCHK(h, hipMemcpy(h->d_m44pack, pk.data(), pk.size() * sizeof(float), hipMemcpyHostToDevice));
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

/* Debug/test entry (not part of the drop-in surface): one wavefront role of the multi-wavefront rollout
 * kernels starts with an exhausted poll budget (it never waits), to prove that a starved wave -- whichever
 * it is -- turns into MPPI_ERR_HIP and not into finite, wrong costs. */
int mppi_debug_inject_handover_fault(mppi_handle *h, int wave, int spin_budget)
{
  if (!h || wave < 0 || (wave > 12 && (wave < 32 || wave > 34)) || spin_budget < 0) return MPPI_ERR_INVALID;
  h->fault_wave = wave;
  h->spin_budget = spin_budget;
  return MPPI_OK;
}

/* Test / tooling hook: beta = min cost out of the rollout kernel (many-chunk solves; abi_solve.hip: min_cost_keys) on / off
 * (on < 0: unchanged), and whether the LAST solve's tail kernel took it from there (*from_rollout; may be NULL). */
int mppi_debug_min_cost(mppi_handle *h, int on, int *from_rollout)
{
  if (!h) return MPPI_ERR_INVALID;
  int rc = mppi_synchronize(h);
  if (rc) return rc;
  if (on >= 0) h->use_min_cost = on != 0;
  if (from_rollout) {
    float f = 0.0f;
    if (tail_is_stream(h->K)) HIPCHK(h, hipMemcpy(&f, h->d_scal + 3, sizeof(float), hipMemcpyDeviceToHost));
    *from_rollout = f != 0.0f;
  }
  return MPPI_OK;
}

/* Test / tooling hook: mppi_control_ticks enqueues every solve but the first one tick ahead, gated on the host (on = 1, the
 * default) or launches each solve when its turn comes, as n calls of mppi_compute_control + mppi_slide_control_seq would (0). */
int mppi_debug_set_chained_ticks(mppi_handle *h, int on)
{
  if (!h) return MPPI_ERR_INVALID;
  int rc = mppi_synchronize(h);
  if (rc) return rc;
  h->chain = on != 0;
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
