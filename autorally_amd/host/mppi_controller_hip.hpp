// mppi_controller_hip.hpp -- the reference's controller surface over the libmppi_hip C ABI.
//
// Header-only C++ mirror of the three classes the path's callers touch
// (src/path_integral/path_integral_main.cu:94-122, PI/run_control_loop.cuh:148-300):
//   NeuralNetModel  (PI/neural_net_model.cuh:64-116)  -> mppi_host::NeuralNetModel
//   MPPICosts       (PI/costs.cuh:87-266)             -> mppi_host::MPPICosts
//   MPPIController  (PI/mppi_controller.cuh:52-217)   -> mppi_host::MPPIController
// Same method names and argument meaning.  Differences forced by the environment: Eigen is not
// available, so states are `const float*` / std::array<float,7> instead of Eigen::Matrix<float,7,1>;
// K, BDIM and the layer list are runtime values instead of template constants (the reference must be
// recompiled to change them, path_integral_main.cu:65-78).  DDP feedback gains:
// computeFeedbackGains()/getFeedbackGains() run the library's host DDP (mppi_compute_feedback_gains).
//
// All compute goes through the C ABI; there is no CPU fallback.  The only host arithmetic is what the
// reference also does on the host: computeNominalTraj / model->updateState (mppi_nominal_traj).
#pragma once

#include <array>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/mppi_hip.h"
#include "../csrc/basis_funcs.hpp"  // CarBasisFuncs arithmetic (header-only, shared with the device kernel)
#include "npz.hpp"
#include "param_getter.hpp"

namespace mppi_host {

struct float2_ { float x, y; };

// ------------------------------------------------------------------------------------------
class NeuralNetModel {
 public:
  static const int STATE_DIM = 7, CONTROL_DIM = 2, DYNAMICS_DIM = 4;
  std::vector<int> net_structure_;          // e.g. {6,32,32,4} (template pack in the reference)
  std::vector<float2_> control_rngs_;       // public in the reference (neural_net_model.cuh:76)
  bool negate_yaw_der = true;               // neural_net_model.cuh:84

  // NeuralNetModel(float delta_t, float2* control_rngs = NULL), neural_net_model.cu:38-67
  NeuralNetModel(const std::vector<int> &layers, float delta_t, const float2_ *control_rngs = nullptr)
      : net_structure_(layers), dt_(delta_t)
  {
    control_rngs_.resize(CONTROL_DIM);
    for (int i = 0; i < CONTROL_DIM; i++)
      control_rngs_[i] = control_rngs ? control_rngs[i] : float2_{-FLT_MAX, FLT_MAX};
    int n = 0;
    for (size_t i = 0; i + 1 < layers.size(); i++) n += (layers[i] + 1) * layers[i + 1];
    net_params_.assign((size_t)n, 0.0f);
  }

  int numParams() const { return (int)net_params_.size(); }
  float dt() const { return dt_; }
  const std::vector<float> &packedParams() const { return net_params_; }

  // loadParams(model_path), neural_net_model.cu:73-106: keys dynamics_W{i} (out,in) and
  // dynamics_b{i}, float64 in the shipped files, cast to float, packed [W1|b1|W2|b2|...] (:120-141)
  void loadParams(const std::string &model_path)
  {
    if (!file_exists(model_path)) throw std::runtime_error("Could not load neural net model at path: " + model_path);
    npz_t d = npz_load(model_path);
    size_t off = 0;
    for (size_t i = 1; i < net_structure_.size(); i++) {
      const std::string wn = "dynamics_W" + std::to_string(i), bn = "dynamics_b" + std::to_string(i);
      if (!d.count(wn) || !d.count(bn)) throw std::runtime_error("model file lacks " + wn + "/" + bn);
      const NpyArray &W = d[wn], &b = d[bn];
      const size_t nout = (size_t)net_structure_[i], nin = (size_t)net_structure_[i - 1];
      if (W.num_vals() != nout * nin || b.num_vals() != nout)
        throw std::runtime_error("model file shape does not match the network structure");
      for (size_t j = 0; j < nout * nin; j++) net_params_[off + j] = (float)W.at(j);
      off += nout * nin;
      for (size_t j = 0; j < nout; j++) net_params_[off + j] = (float)b.at(j);
      off += nout;
    }
    version_++;
  }

  // updateModel(description, data), neural_net_model.cu:152-180: data = [W1|W2|..|b1|b2|..]
  void updateModel(const std::vector<int> &description, const std::vector<float> &data)
  {
    for (size_t i = 0; i < description.size(); i++)
      if (i >= net_structure_.size() || description[i] != net_structure_[i]) return;  // invalid: ignored
    if (data.size() != net_params_.size()) return;
    size_t woff = 0, boff = 0, poff = 0;
    for (size_t l = 0; l + 1 < net_structure_.size(); l++) boff += (size_t)net_structure_[l] * net_structure_[l + 1];
    for (size_t l = 0; l + 1 < net_structure_.size(); l++) {
      const size_t nw = (size_t)net_structure_[l] * net_structure_[l + 1], nb = (size_t)net_structure_[l + 1];
      for (size_t j = 0; j < nw; j++) net_params_[poff + j] = data[woff + j];
      for (size_t j = 0; j < nb; j++) net_params_[poff + nw + j] = data[boff + j];
      woff += nw; boff += nb; poff += nw + nb;
    }
    version_++;
  }

  // ---- host twins of the Eigen members (neural_net_model.cu:191-288); Eigen does not exist here, so the
  // public state_der_ / jac_ (neural_net_model.cuh:70,74) are plain arrays, jac_ row-major [7][9] over [x | u]
  float state_der_[STATE_DIM] = {0, 0, 0, 0, 0, 0, 0};
  float jac_[STATE_DIM][STATE_DIM + CONTROL_DIM] = {};

  // enforceConstraints, :266-278
  void enforceConstraints(float * /*state*/, float *control) const
  {
    for (int i = 0; i < CONTROL_DIM; i++) {
      if (control[i] < control_rngs_[i].x) control[i] = control_rngs_[i].x;
      else if (control[i] > control_rngs_[i].y) control[i] = control_rngs_[i].y;
    }
  }
  // computeKinematics, :191-200 (products contracted like the device code, so that the host replay of a
  // trajectory follows the rollouts' arithmetic)
  void computeKinematics(const float *state)
  {
    const float c = cosf(state[2]), s = sinf(state[2]);
    state_der_[0] = fmaf(c, state[4], -(s * state[5]));
    state_der_[1] = fmaf(s, state[4], c * state[5]);
    state_der_[2] = negate_yaw_der ? -state[6] : state[6];
  }
  // computeDynamics, :202-230: forward pass, keeps every layer's weighted input for computeGrad
  void computeDynamics(const float *state, const float *control)
  {
    const size_t L = net_structure_.size() - 1;
    weighted_in_.resize(L);
    std::vector<float> a(maxWidth()), b(maxWidth());
    a[0] = state[3]; a[1] = state[4]; a[2] = state[5]; a[3] = state[6]; a[4] = control[0]; a[5] = control[1];
    size_t off = 0;
    for (size_t l = 0; l < L; l++) {
      const int nin = net_structure_[l], nout = net_structure_[l + 1];
      const float *W = &net_params_[off], *bias = &net_params_[off + (size_t)nin * nout];
      weighted_in_[l].resize((size_t)nout);
      for (int j = 0; j < nout; j++) {
        float tmp = 0.0f;
        for (int k = 0; k < nin; k++) tmp = fmaf(W[j * nin + k], a[k], tmp);
        tmp += bias[j];
        weighted_in_[l][(size_t)j] = tmp;
        b[j] = (l + 1 < L) ? tanhf(tmp) : tmp;  // MPPI_NNET_NONLINEARITY; none on the last layer
      }
      off += (size_t)nin * nout + nout;
      a.swap(b);
    }
    for (int i = 0; i < DYNAMICS_DIM; i++) state_der_[3 + i] = a[i];
  }
  // computeGrad, :233-264: 7 x 9 Jacobian of [kinematics | network] wrt [x | u]; d(yaw rate)/d(s6) is -1
  // whatever negate_yaw_der says (the reference's quirk, :241)
  void computeGrad(const float *state, const float *control)
  {
    for (int i = 0; i < STATE_DIM; i++)
      for (int j = 0; j < STATE_DIM + CONTROL_DIM; j++) jac_[i][j] = 0.0f;
    const float sn = sinf(state[2]), cs = cosf(state[2]);
    jac_[0][2] = -sn * state[4] - cs * state[5]; jac_[0][4] = cs; jac_[0][5] = -sn;
    jac_[1][2] = cs * state[4] - sn * state[5];  jac_[1][4] = sn; jac_[1][5] = cs;
    jac_[2][6] = -1.0f;
    computeDynamics(state, control);  // "First do the forward pass"
    const int L = (int)net_structure_.size() - 1;
    std::vector<size_t> woff((size_t)L);
    size_t off = 0;
    for (int l = 0; l < L; l++) { woff[(size_t)l] = off; off += (size_t)(net_structure_[l] + 1) * net_structure_[l + 1]; }
    // ip_delta_: [neurons of the current layer][4 outputs], starts as the identity at the output
    std::vector<float> d((size_t)maxWidth() * 4, 0.0f), dn((size_t)maxWidth() * 4, 0.0f);
    for (int c = 0; c < DYNAMICS_DIM; c++) d[(size_t)c * 4 + c] = 1.0f;
    for (int l = L - 1; l >= 0; l--) {
      const int nin = net_structure_[l], nout = net_structure_[l + 1];
      const float *W = &net_params_[woff[(size_t)l]];
      for (int i = 0; i < nin; i++)
        for (int c = 0; c < DYNAMICS_DIM; c++) {
          float s = 0.0f;
          for (int k = 0; k < nout; k++) s += W[k * nin + i] * d[(size_t)k * 4 + c];  // W^T ip_delta
          if (l > 0) {  // .* MPPI_NNET_NONLINEARITY_DERIV(weighted_in_[l-1]) = 1 - tanh^2
            const float t = tanhf(weighted_in_[(size_t)l - 1][(size_t)i]);
            s *= (1.0f - t * t);
          }
          dn[(size_t)i * 4 + c] = s;
        }
      d.swap(dn);
    }
    for (int o = 0; o < DYNAMICS_DIM; o++)  // bottomRightCorner(4, 6) += ip_delta_^T
      for (int i = 0; i < DYNAMICS_DIM + CONTROL_DIM; i++) jac_[3 + o][3 + i] += d[(size_t)i * 4 + o];
  }
  // updateState, :280-288: clamp, kinematics, network, Euler step; state_der_ is zero afterwards
  void updateState(float *state, float *control)
  {
    enforceConstraints(state, control);
    computeKinematics(state);
    computeDynamics(state, control);
    for (int i = 0; i < STATE_DIM; i++) {
      state[i] = fmaf(state_der_[i], dt_, state[i]);
      state_der_[i] = 0.0f;
    }
  }
  // The policy objects own no device memory and no stream in this build (the controller's handle does), so
  // these members of the reference's Managed interface (managed.cuh, neural_net_model.cu:108-118,290-294)
  // have nothing to do; they exist so that code written against the reference compiles and runs unchanged.
  void bindToStream(void * /*hipStream_t*/) {}
  void freeCudaMem() {}

  // paramsToDevice(), neural_net_model.cu:120-150 -- pushed into a controller's handle
  void paramsToDevice(mppi_handle *h)
  {
    check(mppi_set_nn_params(h, net_params_.data(), net_params_.size()), h);
    float lo[2] = {control_rngs_[0].x, control_rngs_[1].x}, hi[2] = {control_rngs_[0].y, control_rngs_[1].y};
    check(mppi_set_control_limits(h, lo, hi), h);
  }
  // bumped whenever the network weights change; controllers re-upload when they see a new value
  unsigned version_ = 0;
  void touch() { version_++; }

  static void check(int rc, mppi_handle *h)
  {
    if (rc != MPPI_OK)
      throw std::runtime_error(std::string("libmppi_hip: ") + mppi_strerror(rc) + " (" + (h ? mppi_last_error(h) : "") + ")");
  }

 private:
  size_t maxWidth() const
  {
    int m = 0;
    for (int v : net_structure_) m = v > m ? v : m;
    return (size_t)m;
  }
  float dt_;
  std::vector<float> net_params_;
  std::vector<std::vector<float>> weighted_in_;  // per layer, of the last computeDynamics
};

// ------------------------------------------------------------------------------------------
// GeneralizedLinear<CarBasisFuncs, 7, 2, 25, CarKinematics, 3> (PI/generalized_linear.cuh:52-117):
// the reference's second DYNAMICS_T (path_integral_bf, path_integral_main.cu:70-74).
class GeneralizedLinear {
 public:
  static const int STATE_DIM = 7, CONTROL_DIM = 2, DYNAMICS_DIM = 4, NUM_BFS = 25;
  std::vector<int> net_structure_;          // empty: selects the basis-function kernel (mppi_config.n_layers = 0)
  std::vector<float2_> control_rngs_;       // generalized_linear.cuh:62
  bool negate_yaw_der = true;               // fixed: computeKinematics always negates (generalized_linear.cu:216)

  // GeneralizedLinear(float delta_t, float2* control_rngs = NULL), generalized_linear.cu:63-84
  explicit GeneralizedLinear(float delta_t, const float2_ *control_rngs = nullptr) : dt_(delta_t)
  {
    control_rngs_.resize(CONTROL_DIM);
    for (int i = 0; i < CONTROL_DIM; i++)
      control_rngs_[i] = control_rngs ? control_rngs[i] : float2_{-FLT_MAX, FLT_MAX};
    theta_.assign((size_t)DYNAMICS_DIM * NUM_BFS, 0.0f);
  }
  float dt() const { return dt_; }
  const std::vector<float> &theta() const { return theta_; }

  // setParams(theta), :86-91; row-major [DYNAMICS_DIM][NUM_BFS]
  void setParams(const std::vector<float> &theta)
  {
    if (theta.size() != theta_.size()) throw std::runtime_error("GeneralizedLinear::setParams: need 4 x 25 values");
    theta_ = theta;
    version_++;
  }
  // loadParams(model_path), :93-110: key "W", (4, 25) float64, cast to float
  void loadParams(const std::string &model_path)
  {
    if (!file_exists(model_path)) throw std::runtime_error("Could not load generalized linear model at path: " + model_path);
    npz_t d = npz_load(model_path);
    if (!d.count("W")) throw std::runtime_error("model file lacks W");
    const NpyArray &W = d["W"];
    if (W.num_vals() != theta_.size()) throw std::runtime_error("W must be (4, 25)");
    for (size_t i = 0; i < theta_.size(); i++) theta_[i] = (float)W.at(i);
    version_++;
  }
  // host updateState (:140-167): clamp, kinematics, basis functions, W phi, Euler step
  void updateState(float *state, float *control)
  {
    for (int i = 0; i < CONTROL_DIM; i++) {
      if (control[i] < control_rngs_[i].x) control[i] = control_rngs_[i].x;
      else if (control[i] > control_rngs_[i].y) control[i] = control_rngs_[i].y;
    }
    float sd[STATE_DIM], phi[NUM_BFS];
    const float c = cosf(state[2]), s = sinf(state[2]);
    sd[0] = fmaf(c, state[4], -(s * state[5]));
    sd[1] = fmaf(s, state[4], c * state[5]);
    sd[2] = -state[6];
    mppi::basis_funcs(state, control[0], control[1], phi);
    mppi::basis_dynamics(theta_.data(), phi, sd + 3);
    for (int i = 0; i < STATE_DIM; i++) state[i] = fmaf(sd[i], dt_, state[i]);
  }
  // paramsToDevice(), :112-118
  void paramsToDevice(mppi_handle *h)
  {
    NeuralNetModel::check(mppi_set_bf_params(h, theta_.data(), theta_.size()), h);
    float lo[2] = {control_rngs_[0].x, control_rngs_[1].x}, hi[2] = {control_rngs_[0].y, control_rngs_[1].y};
    NeuralNetModel::check(mppi_set_control_limits(h, lo, hi), h);
  }
  // updateModel(description, data): empty in the reference too (generalized_linear.cuh:88)
  void updateModel(const std::vector<int> &, const std::vector<float> &) {}
  void bindToStream(void * /*hipStream_t*/) {}  // see NeuralNetModel
  void freeCudaMem() {}
  unsigned version_ = 0;
  void touch() { version_++; }

 private:
  float dt_;
  std::vector<float> theta_;
};

// The dynamic_reconfigure message of the reference (cfg/PathIntegralParams.cfg:12-21, its defaults):
// what AutorallyPlant::getDynRcfgParams hands to MPPICosts::updateParams_dcfg.
struct PathIntegralParamsConfig {
  double max_throttle = 0.65, desired_speed = 6.0, speed_coefficient = 4.25, track_coefficient = 200.0,
         max_slip_angle = 1.25, slip_penalty = 10.0, crash_coefficient = 10000.0, track_slop = 0.0,
         steering_coeff = 0.0, throttle_coeff = 0.0;
};

// ------------------------------------------------------------------------------------------
class MPPICosts {
 public:
  // CostParams, costs.cuh:67-85
  struct CostParams {
    float desired_speed, speed_coeff, track_coeff, max_slip_ang, slip_penalty, track_slop, crash_coeff,
        steering_coeff, throttle_coeff, boundary_threshold, discount;
    int num_timesteps, grid_res;
    float r_c1[3], r_c2[3], trs[3];
  };
  CostParams params_{};  // public in the reference
  bool l1_cost_ = false;
  int width_ = 0, height_ = 0;
  std::vector<float> track_costs_;  // float4[H][W]

  // MPPICosts(std::map<std::string,XmlRpcValue>* params), costs.cu:52-66
  explicit MPPICosts(ParamMap *params)
  {
    loadTrackData((std::string)(*params)["map_path"]);
    updateParams(params);
  }
  // MPPICosts(int width, int height), costs.cu:43-50: zero costmap
  MPPICosts(int width, int height) : width_(width), height_(height)
  {
    track_costs_.assign((size_t)4 * width * height, 0.0f);
    const float r1[3] = {1, 0, 0}, r2[3] = {0, 1, 0}, t[3] = {0, 0, 1};
    for (int i = 0; i < 3; i++) { params_.r_c1[i] = r1[i]; params_.r_c2[i] = r2[i]; params_.trs[i] = t[i]; }
  }

  // updateParams, costs.cu:156-173
  void updateParams(ParamMap *params)
  {
    l1_cost_ = (bool)(*params)["l1_cost"];
    params_.desired_speed = (float)(double)(*params)["desired_speed"];
    params_.speed_coeff = (float)(double)(*params)["speed_coefficient"];
    params_.track_coeff = (float)(double)(*params)["track_coefficient"];
    params_.max_slip_ang = (float)(double)(*params)["max_slip_angle"];
    params_.slip_penalty = (float)(double)(*params)["slip_penalty"];
    params_.track_slop = (float)(double)(*params)["track_slop"];
    params_.crash_coeff = (float)(double)(*params)["crash_coeff"];
    params_.steering_coeff = (float)(double)(*params)["steering_coeff"];
    params_.throttle_coeff = (float)(double)(*params)["throttle_coeff"];
    params_.boundary_threshold = (float)(double)(*params)["boundary_threshold"];
    params_.discount = (float)(double)(*params)["discount"];
    params_.num_timesteps = (int)(*params)["num_timesteps"];
    version_++;
  }

  // loadTrackData, costs.cu:190-232: keys xBounds, yBounds, pixelsPerMeter, channel0..3 (float32)
  void loadTrackData(const std::string &map_path)
  {
    if (!file_exists(map_path)) throw std::runtime_error("Could not load costmap at path: " + map_path);
    npz_t d = npz_load(map_path);
    for (const char *k : {"xBounds", "yBounds", "pixelsPerMeter", "channel0", "channel1", "channel2", "channel3"})
      if (!d.count(k)) throw std::runtime_error(std::string("costmap file lacks key ") + k);
    const float x_min = (float)d["xBounds"].at(0), x_max = (float)d["xBounds"].at(1);
    const float y_min = (float)d["yBounds"].at(0), y_max = (float)d["yBounds"].at(1);
    const float ppm = (float)d["pixelsPerMeter"].at(0);
    width_ = int((x_max - x_min) * ppm);
    height_ = int((y_max - y_min) * ppm);
    const size_t n = (size_t)width_ * height_;
    for (int c = 0; c < 4; c++)
      if (d["channel" + std::to_string(c)].num_vals() < n) throw std::runtime_error("costmap channel too short");
    track_costs_.assign(4 * n, 0.0f);
    for (int c = 0; c < 4; c++) {
      const NpyArray &ch = d["channel" + std::to_string(c)];
      for (size_t i = 0; i < n; i++) track_costs_[4 * i + c] = (float)ch.at(i);
    }
    // R and trs, :224-229, stored column-wise by updateTransform :175-188
    params_.r_c1[0] = 1.f / (x_max - x_min); params_.r_c1[1] = 0; params_.r_c1[2] = 0;
    params_.r_c2[0] = 0; params_.r_c2[1] = 1.f / (y_max - y_min); params_.r_c2[2] = 0;
    params_.trs[0] = -x_min / (x_max - x_min); params_.trs[1] = -y_min / (y_max - y_min); params_.trs[2] = 1;
    map_version_++;
  }

  // updateParams_dcfg, costs.cu:75-87: nine cost parameters (not boundary_threshold / discount / l1_cost)
  void updateParams_dcfg(const PathIntegralParamsConfig &config)
  {
    params_.desired_speed = (float)config.desired_speed;
    params_.speed_coeff = (float)config.speed_coefficient;
    params_.track_coeff = (float)config.track_coefficient;
    params_.max_slip_ang = (float)config.max_slip_angle;
    params_.slip_penalty = (float)config.slip_penalty;
    params_.crash_coeff = (float)config.crash_coefficient;
    params_.track_slop = (float)config.track_slop;
    params_.steering_coeff = (float)config.steering_coeff;
    params_.throttle_coeff = (float)config.throttle_coeff;
    version_++;
  }
  // updateTransform(m, trs), costs.cu:175-188: m is the 3x3 homography, row-major here; its first two
  // columns and trs are what coorTransform uses
  void updateTransform(const float m[9], const float trs[3])
  {
    params_.r_c1[0] = m[0]; params_.r_c1[1] = m[3]; params_.r_c1[2] = m[6];
    params_.r_c2[0] = m[1]; params_.r_c2[1] = m[4]; params_.r_c2[2] = m[7];
    for (int i = 0; i < 3; i++) params_.trs[i] = trs[i];  // pushed with the next solve (syncParams compares params_)
  }

  float getDesiredSpeed() const { return params_.desired_speed; }
  void setDesiredSpeed(float v) { params_.desired_speed = v; version_++; }
  // empty bodies in the reference too (costs.cu:297-299)
  void updateCostmap(const std::vector<int> &, const std::vector<float> &) {}
  void updateObstacles(const std::vector<int> &, const std::vector<float> &) {}

  // getCostInfo(), costs.cuh:170 / costs.cu:240-242: an empty body in the reference ("TODO: Return some useful
  // information about the cost"); kept so that code calling it compiles
  void getCostInfo() {}
  // debugDisplayInit() / debugDisplayInit(width_m, height_m, ppm), costs.cuh:186-191, costs.cu:255-269: the window
  // of the debug raster (the reference also allocates its buffers here; the handle allocates per call)
  void debugDisplayInit() { debugDisplayInit(10, 10, 50); }
  void debugDisplayInit(int width_m, int height_m, int ppm)
  {
    debug_img_width_ = width_m;
    debug_img_height_ = height_m;
    debug_img_ppm_ = ppm;
    debugging_ = true;
  }
  // getDebugDisplay(x, y, heading), costs.cu:272-285 -> debugCostKernel (debug_kernels.cuh:39-88), without the
  // cv::Mat: the raster [height_m*ppm][width_m*ppm] of the window debugDisplayInit set (default 10 m x 10 m at
  // 50 px/m, :274-276).  The costs object owns no device state here; the raster runs on the handle of a controller
  // that uses this object (the most recently constructed one still alive), after that handle's pending solve
  // and with the current params_ pushed first.
  std::vector<float> getDebugDisplay(float x, float y, float heading)
  {
    if (!debugging_) debugDisplayInit();
    return getDebugDisplay(x, y, heading, debug_img_width_, debug_img_height_, debug_img_ppm_);
  }
  std::vector<float> getDebugDisplay(float x, float y, float heading, int width_m, int height_m, int ppm)
  {
    if (bound_.empty()) throw std::runtime_error("MPPICosts::getDebugDisplay: no controller uses this costs object");
    mppi_handle *h = bound_.back();
    NeuralNetModel::check(mppi_synchronize(h), h);  // no parameter push under an asynchronous solve
    paramsToDevice(h, false);
    NeuralNetModel::check(mppi_set_costmap_transform(h, params_.r_c1, params_.r_c2, params_.trs), h);
    std::vector<float> img((size_t)width_m * ppm * height_m * ppm);
    NeuralNetModel::check(mppi_debug_cost_raster(h, x, y, heading, width_m, height_m, ppm, img.data(), img.size()), h);
    return img;
  }
  // every controller constructed on this costs object binds its handle; a destroyed one leaves the others bound
  void bindHandle(mppi_handle *h) { bound_.push_back(h); }
  void unbindHandle(mppi_handle *h)
  {
    for (size_t i = 0; i < bound_.size(); i++)
      if (bound_[i] == h) { bound_.erase(bound_.begin() + (long)i); break; }
  }
  size_t boundHandles() const { return bound_.size(); }
  // Managed interface of the reference (managed.cuh): nothing to do here, see NeuralNetModel::bindToStream.
  // paramsToDevice() without a handle asks every controller to push params_ again at its next solve (they do
  // so anyway whenever params_ differs from what they pushed last, like mppi_controller.cu:605).
  void bindToStream(void * /*hipStream_t*/) {}
  void freeCudaMem() {}
  void paramsToDevice() { version_++; }

  // paramsToDevice + costmapToTexture for one controller's handle
  void paramsToDevice(mppi_handle *h, bool with_map)
  {
    if (with_map)
      NeuralNetModel::check(mppi_set_costmap(h, width_, height_, track_costs_.data(), params_.r_c1, params_.r_c2, params_.trs), h);
    mppi_cost_params p;
    p.desired_speed = params_.desired_speed; p.speed_coeff = params_.speed_coeff; p.track_coeff = params_.track_coeff;
    p.max_slip_ang = params_.max_slip_ang; p.slip_penalty = params_.slip_penalty; p.track_slop = params_.track_slop;
    p.crash_coeff = params_.crash_coeff; p.steering_coeff = params_.steering_coeff; p.throttle_coeff = params_.throttle_coeff;
    p.boundary_threshold = params_.boundary_threshold; p.discount = params_.discount; p.l1_cost = l1_cost_ ? 1 : 0;
    NeuralNetModel::check(mppi_set_cost_params(h, &p), h);
  }
  unsigned version_ = 0, map_version_ = 0;

 private:
  std::vector<mppi_handle *> bound_;
  int debug_img_width_ = 10, debug_img_height_ = 10, debug_img_ppm_ = 50;
  bool debugging_ = false;
};

// ------------------------------------------------------------------------------------------
template <class DYNAMICS_T>
class MPPIControllerT {
 public:
  static const int BLOCKSIZE_WRX = 64;
  static const int STATE_DIM = 7, CONTROL_DIM = 2;
  int NUM_ROLLOUTS;
  int numTimesteps_, hz_, optimizationStride_;
  DYNAMICS_T *model_;
  MPPICosts *costs_;

  // MPPIController(model, costs, exploration_var, init_control, hz, num_timesteps,
  //                optimization_stride, gamma, num_iters, stream), mppi_controller.cuh:101-102.
  // `rollouts` replaces the ROLLOUTS template argument; rounded down to a multiple of 64 like
  // NUM_ROLLOUTS (mppi_controller.cuh:58-60).  `device` replaces the cudaStream_t argument: each
  // controller owns its own stream on that device.
  MPPIControllerT(DYNAMICS_T *model, MPPICosts *costs, const float *exploration_var, const float *init_control,
                 int hz, int num_timesteps, int optimization_stride, float gamma, int num_iters, int rollouts,
                 int device = 0, unsigned long long seed = 1234ULL)
      : NUM_ROLLOUTS((rollouts / BLOCKSIZE_WRX) * BLOCKSIZE_WRX), numTimesteps_(num_timesteps), hz_(hz),
        optimizationStride_(optimization_stride), model_(model), costs_(costs)
  {
    mppi_config c{};
    c.device = device;
    c.num_rollouts = NUM_ROLLOUTS;
    c.num_timesteps = num_timesteps;
    c.hz = hz;
    c.optimization_stride = optimization_stride;
    c.gamma = gamma;
    c.num_iters = num_iters;
    c.n_layers = (int)model->net_structure_.size();
    for (int i = 0; i < c.n_layers; i++) c.layers[i] = model->net_structure_[i];
    for (int i = 0; i < 2; i++) {
      c.exploration_std[i] = exploration_var[i];
      c.init_control[i] = init_control[i];
      c.control_min[i] = model->control_rngs_[i].x;
      c.control_max[i] = model->control_rngs_[i].y;
    }
    c.negate_yaw_der = model->negate_yaw_der ? 1 : 0;
    c.seed = seed;  // curandSetPseudoRandomGeneratorSeed(gen_, 1234ULL), mppi_controller.cu:331
    const int rc = mppi_create(&c, &h_);
    if (rc != MPPI_OK) throw std::runtime_error(std::string("mppi_create: ") + mppi_strerror(rc));
    state_solution_.assign((size_t)numTimesteps_ * STATE_DIM, 0.0f);
    control_solution_.assign((size_t)numTimesteps_ * CONTROL_DIM, 0.0f);
    setCudaStream(nullptr);  // mppi_controller.cu:335
    syncParams(true);
    initDDP();               // :359
    costs_->bindHandle(h_);
  }
  ~MPPIControllerT() { deallocateCudaMem(); }
  MPPIControllerT(const MPPIControllerT &) = delete;
  MPPIControllerT &operator=(const MPPIControllerT &) = delete;

  void deallocateCudaMem()
  {
    if (h_) {
      costs_->unbindHandle(h_);
      mppi_destroy(h_);
    }
    h_ = nullptr;
  }
  mppi_handle *handle() { return h_; }
  // Not in the reference: which rollout kernel form serves this controller (mppi_set_rollout_variant).  "auto" (default)
  // lets the library choose; "mfma" keeps the reference's summation order in every layer (the automatic choice for
  // 6-32-32-4 and 64-wide nets at K <= 8192 sums the OUTPUT layer as a butterfly, inside the 1e-4 tolerance: INTEGRATION.md 2)
  void setRolloutVariant(const std::string &name)
  {
    if (mppi_set_rollout_variant(h_, name.c_str()) != MPPI_OK)
      throw std::runtime_error(std::string("setRolloutVariant: ") + mppi_last_error(h_));
  }
  std::string rolloutVariant() const { return mppi_rollout_variant(h_); }

  // setCudaStream(stream), mppi_controller.cuh:109 / mppi_controller.cu:365-370: the handle owns its stream (one
  // per controller, created with it), so there is nothing to rebind; kept for code written against the reference
  void setCudaStream(void * /*hipStream_t*/) {}
  // initDDP(), mppi_controller.cuh:121 / mppi_controller.cu:402-429: the tracking weights of the feedback
  // controller, Q = diag(.5,.5,.25,0,.05,.01,.01), R = diag(10,10), Qf = 0 (:410-417); one DDP iteration, dt = 1/hz
  void initDDP()
  {
    const float Q[STATE_DIM] = {0.5f, 0.5f, 0.25f, 0.0f, 0.05f, 0.01f, 0.01f}, R[CONTROL_DIM] = {10.0f, 10.0f};
    const float Qf[STATE_DIM] = {0, 0, 0, 0, 0, 0, 0};
    ck(mppi_set_ddp_weights(h_, Q, R, Qf));
  }
  // savitskyGolay(), mppi_controller.cuh:134 / mppi_controller.cu:468-499: smooths U_ in place (computeControl
  // already ends with it; public in the reference)
  void savitskyGolay() { ck(mppi_savitsky_golay(h_)); }
  void resetControls() { ck(mppi_reset_controls(h_)); }
  void cutThrottle()  // mppi_controller.cu:460-466: plain writes to the public members; the next solve pushes them
  {
    costs_->params_.desired_speed = 0.0f;
    model_->control_rngs_[1].y = 0.0f;
  }
  // OptimizerResult of ddp/result.h, flattened
  struct FeedbackResult {
    std::vector<float> feedback_gain;      // [T][2][7]
    std::vector<float> feedforward_gain;   // [T][2]
    std::vector<float> state_trajectory;   // [T][7]
    std::vector<float> control_trajectory; // [T][2]
    float total_cost = 0.0f;
  };
  // computeFeedbackGains(state), mppi_controller.cu:431-441: tracks state_solution_/control_solution_
  void computeFeedbackGains(const float *state)
  {
    syncParams(false);  // the host copy of the network / control ranges the Jacobians are taken of
    ck(mppi_compute_feedback_gains(h_, state, state_solution_.data(), control_solution_.data()));
    fetchFeedbackGains();
  }
  void fetchFeedbackGains()
  {
    const size_t T = (size_t)numTimesteps_;
    result_.feedback_gain.resize(T * CONTROL_DIM * STATE_DIM);
    result_.feedforward_gain.resize(T * CONTROL_DIM);
    result_.state_trajectory.resize(T * STATE_DIM);
    result_.control_trajectory.resize(T * CONTROL_DIM);
    ck(mppi_get_feedback_gains(h_, result_.feedback_gain.data(), result_.feedforward_gain.data(),
                               result_.state_trajectory.data(), result_.control_trajectory.data(), &result_.total_cost));
  }
  // computeFeedbackGains(state) of both controllers of a tick (run_control_loop.cuh:220-225) in one call: with
  // mppi_set_host_threads(2) the two DDP passes run side by side; the results are those of the two single calls
  static void computeFeedbackGainsPair(MPPIControllerT *actual, MPPIControllerT *predicted, const float *state)
  {
    actual->syncParams(false);
    predicted->syncParams(false);
    actual->ck(mppi_compute_feedback_gains_pair(actual->h_, state, actual->state_solution_.data(), actual->control_solution_.data(),
                                                predicted->h_, state, predicted->state_solution_.data(),
                                                predicted->control_solution_.data()));
    for (MPPIControllerT *c : {actual, predicted}) c->fetchFeedbackGains();
  }
  const FeedbackResult &getFeedbackGains() const { return result_; }  // :443-446

  void slideControlAndStateSeq(int stride)
  {
    ck(mppi_slide_control_seq(h_, stride));       // slideControlSeq :527-554
    for (int i = 0; i < numTimesteps_ - stride; i++)  // slideStateSeq :558-568
      for (int j = 0; j < STATE_DIM; j++)
        state_solution_[(size_t)i * STATE_DIM + j] = state_solution_[(size_t)(i + stride) * STATE_DIM + j];
  }
  void setState(const float *state)
  {
    for (int i = 0; i < STATE_DIM; i++) state_solution_[i] = state[i];  // (Q9's OOB column index not reproduced)
  }
  void setStateSequence(const std::vector<float> &s) { state_solution_ = s; }
  void setControlSequence(const std::vector<float> &c) { control_solution_ = c; }

  // computeControl(state), mppi_controller.cu:600-675
  void computeControl(const float *state)
  {
    startControl(state);
    finishControl();
  }
  // computeControl(), :588-598: start from the predicted state (first entry of the state sequence)
  void computeControl()
  {
    startControl();
    finishControl();
  }
  // The two halves of computeControl, so that a caller with several controllers (runControlLoop has two
  // that are independent until the arbitration, run_control_loop.cuh:218-270) can put all solves on the
  // GPU before waiting for the first: each controller owns its stream, the kernels overlap.
  void startControl(const float *state)
  {
    syncParams(false);  // costs_->paramsToDevice(); model_->paramsToDevice();  (:605-606), only when changed
    for (int i = 0; i < STATE_DIM; i++) solve_state_[i] = state[i];
    ck(mppi_compute_control_async(h_, solve_state_));
  }
  void startControl()
  {
    float s[STATE_DIM];
    for (int i = 0; i < STATE_DIM; i++) s[i] = state_solution_[i];
    startControl(s);
  }
  // Both solves of runControlLoop's tick (run_control_loop.cuh:218-219) enqueued TOGETHER: where the two
  // controllers' rollouts fit the chip side by side (2 x 1920 rollouts = 240 groups on 256 CUs) they are one
  // launch of the rollout kernel and one of the tail kernel (mppi_compute_control_batch_async); results per
  // controller are bit for bit those of startControl.  finishControl() of each controller collects them.
  static void startControlPair(MPPIControllerT *actual, const float *state, MPPIControllerT *predicted)
  {
    actual->syncParams(false);
    predicted->syncParams(false);
    float st[2 * STATE_DIM];
    for (int i = 0; i < STATE_DIM; i++) {
      st[i] = actual->solve_state_[i] = state[i];
      st[STATE_DIM + i] = predicted->solve_state_[i] = predicted->state_solution_[i];
    }
    mppi_handle *hs[2] = {actual->h_, predicted->h_};
    const int rc = mppi_compute_control_batch_async(hs, st, 2);
    if (rc != MPPI_OK)  // the text sits on whichever handle recorded it
      throw std::runtime_error(std::string("libmppi_hip: ") + mppi_strerror(rc) + " (" + mppi_last_error(actual->h_) +
                               " | " + mppi_last_error(predicted->h_) + ")");
  }
  void finishControl()
  {
    float tc = 0.0f;
    ck(mppi_get_results(h_, nullptr, &tc, nullptr, nullptr));  // waits for the solve
    trajectory_cost_ = tc;
    computeNominalTraj(solve_state_);
  }
  // finishControl() of both controllers of a tick: collects the two solves, then replays the two nominal trajectories
  // in lockstep on the host (mppi_nominal_traj_pair: about the cost of one replay)
  static void finishControlPair(MPPIControllerT *actual, MPPIControllerT *predicted)
  {
    float tc = 0.0f;
    actual->ck(mppi_get_results(actual->h_, nullptr, &tc, nullptr, nullptr));
    actual->trajectory_cost_ = tc;
    predicted->ck(mppi_get_results(predicted->h_, nullptr, &tc, nullptr, nullptr));
    predicted->trajectory_cost_ = tc;
    actual->ck(mppi_nominal_traj_pair(actual->h_, actual->solve_state_, actual->state_solution_.data(), actual->control_solution_.data(),
                                      predicted->h_, predicted->solve_state_, predicted->state_solution_.data(),
                                      predicted->control_solution_.data()));
  }
  void computeNominalTraj(const float *state)  // :501-519
  {
    ck(mppi_nominal_traj(h_, state, state_solution_.data(), control_solution_.data()));
  }
  // shim kept from round 1: the raster on THIS controller's handle; the reference's call site is
  // costs_->getDebugDisplay (run_control_loop.cuh:171), which MPPICosts has itself now
  std::vector<float> getDebugDisplay(float x, float y, float heading, int width_m = 10, int height_m = 10, int ppm = 50)
  {
    syncParams(false);
    std::vector<float> img((size_t)width_m * ppm * height_m * ppm);
    ck(mppi_debug_cost_raster(h_, x, y, heading, width_m, height_m, ppm, img.data(), img.size()));
    return img;
  }
  std::vector<float> getControlSeq() { return control_solution_; }
  std::vector<float> getStateSeq() { return state_solution_; }
  float getComputedTrajectoryCost() { return trajectory_cost_; }
  std::vector<float> getNominalControls()  // U_ after smoothing (what the next solve perturbs)
  {
    std::vector<float> U((size_t)numTimesteps_ * 2);
    ck(mppi_get_control_seq(h_, U.data(), U.size()));
    return U;
  }

 private:
  void ck(int rc) { NeuralNetModel::check(rc, h_); }
  // costs_->paramsToDevice(); model_->paramsToDevice(); of every computeControl (mppi_controller.cu:605-606).
  // The reference uploads unconditionally, so plain writes to the public members (costs_->params_,
  // costs_->l1_cost_, model_->control_rngs_) take effect at the next solve; here the upload happens when the
  // members differ from the copy pushed last (a 100-byte compare per solve) or a version counter moved
  // (network weights, costmap texels: changed through loadParams / updateModel / loadTrackData only).
  void syncParams(bool force)
  {
    if (force || model_seen_ != model_->version_) {
      model_->paramsToDevice(h_);  // weights + control ranges
      model_seen_ = model_->version_;
      for (int i = 0; i < CONTROL_DIM; i++) pushed_rngs_[i] = model_->control_rngs_[i];
    } else if (std::memcmp(pushed_rngs_, model_->control_rngs_.data(), sizeof(pushed_rngs_)) != 0) {
      const float lo[2] = {model_->control_rngs_[0].x, model_->control_rngs_[1].x};
      const float hi[2] = {model_->control_rngs_[0].y, model_->control_rngs_[1].y};
      ck(mppi_set_control_limits(h_, lo, hi));
      for (int i = 0; i < CONTROL_DIM; i++) pushed_rngs_[i] = model_->control_rngs_[i];
    }
    const MPPICosts::CostParams &p = costs_->params_;
    const bool map_new = force || map_seen_ != costs_->map_version_;
    const bool changed = force || cost_seen_ != costs_->version_ || pushed_l1_ != costs_->l1_cost_ ||
                         std::memcmp(&pushed_params_, &p, sizeof(p)) != 0;
    if (map_new || changed) {
      if (!map_new && (std::memcmp(pushed_params_.r_c1, p.r_c1, sizeof(p.r_c1)) || std::memcmp(pushed_params_.r_c2, p.r_c2, sizeof(p.r_c2)) ||
                       std::memcmp(pushed_params_.trs, p.trs, sizeof(p.trs))))
        ck(mppi_set_costmap_transform(h_, p.r_c1, p.r_c2, p.trs));  // updateTransform, costs.cu:175-188
      costs_->paramsToDevice(h_, map_new);
      cost_seen_ = costs_->version_;
      map_seen_ = costs_->map_version_;
      pushed_params_ = p;
      pushed_l1_ = costs_->l1_cost_;
    }
  }
  mppi_handle *h_ = nullptr;
  float trajectory_cost_ = 0.0f;
  float solve_state_[STATE_DIM] = {0, 0, 0, 0, 0, 0, 0};
  std::vector<float> state_solution_, control_solution_;
  FeedbackResult result_;
  unsigned model_seen_ = 0, cost_seen_ = 0, map_seen_ = 0;
  MPPICosts::CostParams pushed_params_{};
  bool pushed_l1_ = false;
  float2_ pushed_rngs_[CONTROL_DIM] = {{0, 0}, {0, 0}};
};

// MPPIController<DynamicsModel, MPPICosts, ...> of the two reference builds (path_integral_main.cu:65-78)
using MPPIController = MPPIControllerT<NeuralNetModel>;
using MPPIControllerBF = MPPIControllerT<GeneralizedLinear>;

}  // namespace mppi_host
