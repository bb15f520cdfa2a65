// host_selftest.cpp -- CPU-only checks of the host layer (no GPU; libmppi_hip is linked because the
// controller classes' headers name its entry points, but no handle is ever created here):
//   npz reader against a numpy-written model file and a round trip of the writer,
//   launch-XML loader against a launch file in the reference's format,
//   the headless plant's feedback law, the live-pose half of runControlLoop with a scripted pose clock and a
//   recording controller, NeuralNetModel's host twins (computeGrad against finite differences).
//   loadTrackData against a costmap file written by the reference's own track_converter.py (optional).
// usage: host_selftest <model.npz> <launch.xml> <tmp_dir> [costmap.npz]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "npz.hpp"
#include "param_getter.hpp"
#include "run_control_loop.hpp"
#include "../csrc/host_net.hpp"

using namespace mppi_host;

// A controller with the surface runControlLoop uses and no GPU behind it: it records what the loop asks for.
struct FakeController {
  struct Gains { std::vector<float> feedback_gain; };
  MPPICosts *costs_;
  NeuralNetModel *model_;
  float cost;
  std::vector<int> slides;
  std::vector<std::array<float, 7>> solved_from;
  std::vector<float> state_seq, control_seq;
  Gains gains;
  int T;
  FakeController(MPPICosts *c, NeuralNetModel *m, int T_, float cost_) : costs_(c), model_(m), cost(cost_), T(T_)
  {
    state_seq.assign((size_t)T * 7, 0.0f);
    control_seq.assign((size_t)T * 2, 0.0f);
    for (int t = 0; t < T; t++) { control_seq[2 * t] = 0.01f * t; control_seq[2 * t + 1] = 0.3f; }
    gains.feedback_gain.assign((size_t)T * 14, 0.0f);
  }
  void setState(const float *s) { for (int i = 0; i < 7; i++) state_seq[i] = s[i]; }
  void resetControls() {}
  void computeFeedbackGains(const float *) {}
  const Gains &getFeedbackGains() const { return gains; }
  void slideControlAndStateSeq(int stride)
  {
    slides.push_back(stride);
    for (int i = 0; i + stride < T; i++)
      for (int j = 0; j < 7; j++) state_seq[i * 7 + j] = state_seq[(i + stride) * 7 + j];
  }
  void startControl(const float *s)
  {
    std::array<float, 7> a;
    for (int i = 0; i < 7; i++) a[i] = s[i];
    solved_from.push_back(a);
    // "nominal trajectory": the state advances 0.1 m in x per step from where the solve started
    for (int t = 0; t < T; t++)
      for (int j = 0; j < 7; j++) state_seq[t * 7 + j] = s[j] + (j == 0 ? 0.1f * t : 0.0f);
  }
  void startControl() { float s[7]; for (int i = 0; i < 7; i++) s[i] = state_seq[i]; startControl(s); }
  static void startControlPair(FakeController *a, const float *s, FakeController *p) { a->startControl(s); p->startControl(); }
  static void finishControlPair(FakeController *a, FakeController *p) { a->finishControl(); p->finishControl(); }
  static void computeFeedbackGainsPair(FakeController *a, FakeController *p, const float *s) { a->computeFeedbackGains(s); p->computeFeedbackGains(s); }
  void finishControl() {}
  float getComputedTrajectoryCost() const { return cost; }
  std::vector<float> getControlSeq() const { return control_seq; }
  std::vector<float> getStateSeq() const { return state_seq; }
  void setStateSequence(const std::vector<float> &s) { state_seq = s; }
  void setControlSequence(const std::vector<float> &c) { control_seq = c; }
};

#define REQUIRE(c)                                                                          \
  do {                                                                                      \
    if (!(c)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; }  \
  } while (0)

int main(int argc, char **argv)
{
  if (argc < 4) return 2;
  // --- model file (numpy.savez: ZIP stored, .npy v1.0, <f8) ---
  npz_t m = npz_load(argv[1]);
  REQUIRE(m.count("dynamics_W1") && m.count("dynamics_b3"));
  REQUIRE(m["dynamics_W1"].shape.size() == 2 && m["dynamics_W1"].shape[0] == 32 && m["dynamics_W1"].shape[1] == 6);
  REQUIRE(m["dynamics_W2"].num_vals() == 1024 && m["dynamics_W3"].shape[0] == 4);
  REQUIRE(m["dynamics_W1"].kind == 'f' && m["dynamics_W1"].word_size == 8);
  size_t n = 0;
  for (int i = 1; i <= 3; i++)
    n += m["dynamics_W" + std::to_string(i)].num_vals() + m["dynamics_b" + std::to_string(i)].num_vals();
  REQUIRE(n == 1412);  // NUM_PARAMS of 6-32-32-4
  printf("W1[0,0]=%.17g b3[3]=%.17g\n", m["dynamics_W1"].at(0), m["dynamics_b3"].at(3));
  // --- writer round trip in the costmap format ---
  const std::string tmp = std::string(argv[3]) + "/selftest_map.npz";
  std::vector<float> ch(12 * 8);
  for (size_t i = 0; i < ch.size(); i++) ch[i] = 0.25f * (float)i;
  const float xb[2] = {-3.0f, 3.0f}, yb[2] = {-2.0f, 2.0f}, ppm[1] = {2.0f};
  NpzWriter w;
  w.add_f32("xBounds", xb, {2});
  w.add_f32("yBounds", yb, {2});
  w.add_f32("pixelsPerMeter", ppm, {1});
  w.add_f32("channel0", ch.data(), {ch.size()});
  w.save(tmp);
  npz_t r = npz_load(tmp);
  REQUIRE(r["channel0"].num_vals() == ch.size() && r["channel0"].word_size == 4);
  for (size_t i = 0; i < ch.size(); i++) REQUIRE(r["channel0"].at(i) == ch[i]);
  REQUIRE(r["xBounds"].at(0) == -3.0 && r["pixelsPerMeter"].at(0) == 2.0);
  // --- launch XML ---
  setenv("AR_MPPI_PARAMS_PATH", "/somewhere/params", 1);
  ParamMap p;
  loadParams(&p, argv[2]);
  REQUIRE((int)p["hz"] == 50 && (int)p["num_timesteps"] == 100 && (int)p["optimization_stride"] == 1);
  REQUIRE(std::fabs((double)p["gamma"] - 0.15) < 1e-12 && (int)p["num_iters"] == 1);
  REQUIRE((bool)p["debug_mode"] == true && (bool)p["l1_cost"] == false && (bool)p["negate_yaw_der"] == true);
  REQUIRE((std::string)p["model_path"] == "/somewhere/params/models/autorally_nnet_09_12_2018.npz");
  REQUIRE((std::string)p["map_path"] == "/somewhere/params/maps/ccrf_costmap_09_29_2017.npz");
  REQUIRE(std::fabs((double)p["max_throttle"] - 0.65) < 1e-12 && std::fabs((double)p["desired_speed"] - 8.0) < 1e-12);
  REQUIRE(p.count("controller_type") == 0);    // params of later <node>s are not read
  REQUIRE(p.count("profiler_max_iter") == 0);  // commented out in the launch file
  bool threw = false;
  try { (void)(int)p["gamma"]; } catch (const std::exception &) { threw = true; }
  REQUIRE(threw);  // typed like XmlRpcValue
  // --- the plant's feedback law (autorally_plant.cpp:217-250) on a 3-step solution ---
  {
    SimPlant plant;
    const std::vector<float> ss = {0, 0, 0, 0, 1, 0, 0, /**/ 1, 0, 0, 0, 1, 0, 0, /**/ 2, 0, 0, 0, 1, 0, 0};
    const std::vector<float> cs = {0.1f, 0.2f, 0.3f, 0.4f, 0.5f, 0.6f};
    std::vector<float> g(3 * 14, 0.0f);
    g[0 * 14 + 1] = -2.0f;      // t=0: steering reacts to the y error
    g[1 * 14 + 1] = -4.0f;      // t=1
    g[1 * 14 + 7 + 4] = 0.5f;   // t=1: throttle reacts to the u_x error
    plant.setSolution(ss, cs, g, ControllerType::ACTUAL_STATE);  // (the 4-argument form: stamp = now, no timing)
    const float x[7] = {0.5f, 0.1f, 0, 0, 0.8f, 0, 0};
    float u[2];
    REQUIRE(!plant.controlAt(0.0, 0.02, x, true, u) && !plant.controlAt(0.04, 0.02, x, true, u));
    REQUIRE(plant.controlAt(0.01, 0.02, x, false, u));                 // halfway between t=0 and t=1
    REQUIRE(std::fabs(u[0] - 0.2f) < 1e-6f && std::fabs(u[1] - 0.3f) < 1e-6f);
    REQUIRE(plant.controlAt(0.01, 0.02, x, true, u));
    REQUIRE(std::fabs(u[0] - (0.2f + -3.0f * 0.1f)) < 1e-6f);           // K_y = -3 at the midpoint
    REQUIRE(std::fabs(u[1] - (0.3f + 0.25f * (0.8f - 1.0f))) < 1e-6f);  // K_ux = 0.25
    const float far[7] = {0, 10.0f, 0, 0, 1, 0, 0};
    REQUIRE(plant.controlAt(0.01, 0.02, far, true, u) && u[0] == -1.0f);  // saturated like pubControl
  }
  // --- runControlLoop's live-pose half (run_control_loop.cuh:140-144,175-181,206-216,304-312) with a scripted
  //     pose clock: the state is refreshed from the plant every tick and the sequences slide by the number of
  //     control periods between the last two pose stamps; without a new pose the last loop time stands ---
  {
    ParamMap lp;
    lp["x_pos"] = ParamValue(1.0); lp["y_pos"] = ParamValue(2.0); lp["heading"] = ParamValue(0.5);
    lp["hz"] = ParamValue(50); lp["optimization_stride"] = ParamValue(1); lp["num_timesteps"] = ParamValue(20);
    lp["debug_mode"] = ParamValue(false); lp["use_only_actual_state_controller"] = ParamValue(false);
    lp["use_only_predicted_state_controller"] = ParamValue(false); lp["profiler_max_iter"] = ParamValue(7);
    MPPICosts costs(1, 1);
    const float2_ rng[2] = {{-0.99f, 0.99f}, {-0.99f, 0.65f}};
    NeuralNetModel model({6, 32, 32, 4}, 0.02f, rng);
    FakeController actual(&costs, &model, 20, 5.0f), predicted(&costs, &model, 20, 4.0f);  // predicted is cheaper: used
    SimPlant plant;
    plant.live = true;
    plant.hz = 50;
    plant.pose_time = 77.0;
    plant.fs.x_pos = -3.0f; plant.fs.y_pos = 4.0f; plant.fs.yaw = 0.25f; plant.fs.u_x = 2.0f;
    plant.pose_script = {0.02, 0.04, 0.0, 0.06, 0.009, 0.02, 0.02};  // 1, 2, none, 3, 0, 1 periods
    plant.drive = [](float *x, float *u) { x[0] += 1.0f + u[0]; };  // one period = one metre (+ the steering applied)
    std::atomic<bool> alive(true);
    const LoopStats ls = runControlLoop(&predicted, &actual, &plant, &lp, &alive, /*sleep_to_rate=*/false);
    REQUIRE(ls.iterations == 7);
    // tick 1: status still 1 -> optimization_stride; then round(dt hz) of the last NEW pose interval
    const std::vector<int> want = {1, 1, 2, 2, 3, 0, 1};
    REQUIRE(ls.strides == want && actual.slides == want && predicted.slides == want);
    // the state every actual-state solve started from is the plant's pose at that tick: the plant was not in
    // debug mode, so the launch pose (1, 2, 0.5) is never used
    REQUIRE(actual.solved_from.size() == 7 && actual.solved_from[0][0] == -3.0f && actual.solved_from[0][4] == 2.0f);
    float x = -3.0f;
    const int drove[7] = {1, 2, 0, 3, 0, 1, 1};
    for (int i = 1; i < 7; i++) {
      for (int t = 0; t < drove[i - 1]; t++) x += 1.0f + predicted.control_seq[2 * t];
      REQUIRE(std::fabs(actual.solved_from[i][0] - x) < 1e-5f);
    }
    // the predicted-state controller starts from the slid head of its own state sequence, not from the pose
    REQUIRE(std::fabs(predicted.solved_from[2][0] - (predicted.solved_from[1][0] + 0.1f * 2)) < 1e-5f);
    // solutions carry the stamp of the pose they were computed for and the running mean of the loop time
    REQUIRE(plant.n_solutions == 7 && plant.last_used == ControllerType::PREDICTED_STATE);
    REQUIRE(std::fabs(plant.solution_ts - (77.0 + 0.02 + 0.04 + 0.06 + 0.009 + 0.02)) < 1e-9);
    REQUIRE(ls.avg_loop_ms > 15.0 && ls.avg_loop_ms < 45.0);
  }
  // --- NeuralNetModel's host twins: computeGrad (neural_net_model.cu:233-264) against central differences of
  //     computeKinematics + computeDynamics, the -1 quirk, updateState = clamp + f + Euler, state_der_ zeroed ---
  {
    const float2_ rng[2] = {{-0.99f, 0.99f}, {-0.99f, 0.65f}};
    NeuralNetModel model({6, 32, 32, 4}, 0.02f, rng);
    model.loadParams(argv[1]);
    model.negate_yaw_der = false;  // the quirk: jac_(2,6) stays -1
    float x[7] = {1.0f, -2.0f, 0.7f, 0.05f, 4.0f, 0.3f, -0.4f}, u[2] = {0.1f, 0.4f};
    model.computeGrad(x, u);
    float J[7][9];
    for (int i = 0; i < 7; i++) for (int j = 0; j < 9; j++) J[i][j] = model.jac_[i][j];
    REQUIRE(J[2][6] == -1.0f && J[0][4] == cosf(0.7f) && J[1][4] == sinf(0.7f) && J[3][0] == 0.0f);
    double worst = 0.0, scale = 0.0;
    for (int j = 0; j < 9; j++) {
      float zp[9], zm[9], fp[7], fm[7];
      for (int i = 0; i < 7; i++) zp[i] = zm[i] = x[i];
      zp[7] = zm[7] = u[0]; zp[8] = zm[8] = u[1];
      const float h = 3e-3f;
      zp[j] += h; zm[j] -= h;
      model.computeKinematics(zp); model.computeDynamics(zp, zp + 7);
      for (int i = 0; i < 7; i++) fp[i] = model.state_der_[i];
      model.computeKinematics(zm); model.computeDynamics(zm, zm + 7);
      for (int i = 0; i < 7; i++) fm[i] = model.state_der_[i];
      for (int i = 0; i < 7; i++) {
        if (i == 2 && j == 6) continue;  // the quirk: f has +1 here with negate_yaw_der == false
        const double fd = ((double)fp[i] - fm[i]) / (2.0 * h);
        worst = std::fmax(worst, std::fabs(fd - J[i][j]));
        scale = std::fmax(scale, std::fabs((double)J[i][j]));
      }
    }
    printf("computeGrad vs central differences: worst %.3g of scale %.3g\n", worst, scale);
    REQUIRE(scale > 1.0 && worst < 3e-3 * scale);
    float xs[7], us[2] = {2.0f, -3.0f};  // both outside the control ranges
    for (int i = 0; i < 7; i++) xs[i] = x[i];
    model.updateState(xs, us);
    REQUIRE(us[0] == 0.99f && us[1] == -0.99f);
    for (int i = 0; i < 7; i++) REQUIRE(model.state_der_[i] == 0.0f);
    model.computeKinematics(x); model.computeDynamics(x, us);
    for (int i = 0; i < 7; i++) REQUIRE(xs[i] == fmaf(model.state_der_[i], 0.02f, x[i]));
  }
  // --- HostNetFma (csrc/host_net.hpp: the AVX2 replay of computeNominalTraj, tanhf_vec.hpp inside) against the
  //     scalar statement of the same arithmetic (NeuralNetModel::computeDynamics: k-ascending fmaf, + bias,
  //     libm tanhf): 2000 steps of a driven trajectory, every output bit for bit ---
  {
    const float2_ rng[2] = {{-0.99f, 0.99f}, {-0.99f, 0.65f}};
    NeuralNetModel model({6, 32, 32, 4}, 0.02f, rng);
    model.loadParams(argv[1]);
    mppi::HostNetFma net;
    const int layers[4] = {6, 32, 32, 4};
    net.init(layers, 4, model.packedParams().data());
    REQUIRE(net.vec_tanh);  // tanhf8 equals this machine's libm (tanhf_vec_selfcheck)
    float x[7] = {0.0f, 0.0f, 0.1f, 0.02f, 3.0f, 0.2f, -0.1f};
    for (int t = 0; t < 2000; t++) {
      float u[2] = {0.6f * sinf(0.013f * t), 0.4f + 0.3f * cosf(0.007f * t)};
      const float in6[6] = {x[3], x[4], x[5], x[6], u[0], u[1]};
      float out[4];
      net.forward(in6, out);
      model.computeDynamics(x, u);
      for (int i = 0; i < 4; i++) REQUIRE(memcmp(&out[i], &model.state_der_[3 + i], 4) == 0);
      model.computeKinematics(x);
      for (int i = 0; i < 7; i++) x[i] = fmaf(model.state_der_[i], 0.02f, x[i]);
    }
    REQUIRE(std::fabs(x[4]) > 0.5f && std::isfinite(x[0]));
    // two replays in lockstep (host_net_forward2) = two single replays, bit for bit
    mppi::HostNetFma net2;
    net2.init(layers, 4, model.packedParams().data());
    float xa[6] = {0.02f, 3.0f, 0.2f, -0.1f, 0.3f, 0.4f}, xb[6] = {-0.05f, 6.0f, -0.4f, 0.3f, -0.7f, 0.1f};
    for (int t = 0; t < 500; t++) {
      float oa[4], ob[4], ra[4], rb[4];
      mppi::host_net_forward2(net, net2, xa, xb, oa, ob);
      net.forward(xa, ra);
      net2.forward(xb, rb);
      REQUIRE(memcmp(oa, ra, 16) == 0 && memcmp(ob, rb, 16) == 0);
      for (int i = 0; i < 4; i++) { xa[i] = fmaf(oa[i], 0.02f, xa[i]); xb[i] = fmaf(ob[i], 0.02f, xb[i]); }
    }
  }
  // --- MPPICosts: the non-caller public names (costs.cuh:170,186-191) and the list of bound controller handles ---
  {
    MPPICosts c(4, 4);
    c.getCostInfo();          // empty in the reference too (costs.cu:240-242)
    c.debugDisplayInit();     // 10 m x 10 m at 50 px/m (costs.cu:255-258)
    c.debugDisplayInit(6, 4, 20);
    bool threw = false;
    try { c.getDebugDisplay(0.0f, 0.0f, 0.0f); } catch (const std::runtime_error &) { threw = true; }
    REQUIRE(threw);  // no controller uses this costs object yet
    mppi_handle *h1 = reinterpret_cast<mppi_handle *>(0x10), *h2 = reinterpret_cast<mppi_handle *>(0x20);
    c.bindHandle(h1);  // actual-state controller
    c.bindHandle(h2);  // predicted-state controller
    REQUIRE(c.boundHandles() == 2);
    c.unbindHandle(h2);  // destroying the last-constructed controller leaves the other one bound
    REQUIRE(c.boundHandles() == 1);
    c.unbindHandle(h2);
    REQUIRE(c.boundHandles() == 1);
    c.unbindHandle(h1);
    REQUIRE(c.boundHandles() == 0);
  }
  // --- MPPICosts::loadTrackData (costs.cu:190-232) on a file produced by the reference's writer ---
  if (argc > 4) {
    MPPICosts c(1, 1);
    c.loadTrackData(argv[4]);
    double sum = 0.0;
    for (size_t i = 0; i < (size_t)c.width_ * c.height_; i++) sum += c.track_costs_[4 * i];
    printf("costmap W=%d H=%d r_c1=[%.9g %.9g %.9g] r_c2=[%.9g %.9g %.9g] trs=[%.9g %.9g %.9g] ch0[0]=%.9g ch0[last]=%.9g sum0=%.9g ch1max=%.9g\n",
           c.width_, c.height_, c.params_.r_c1[0], c.params_.r_c1[1], c.params_.r_c1[2], c.params_.r_c2[0],
           c.params_.r_c2[1], c.params_.r_c2[2], c.params_.trs[0], c.params_.trs[1], c.params_.trs[2],
           c.track_costs_[0], c.track_costs_[4 * ((size_t)c.width_ * c.height_ - 1)], sum, c.track_costs_[1]);
    // all four channels of the float4 texture (costs.cu:207-222): per-channel sums and one texel
    double cs[4] = {0, 0, 0, 0};
    for (size_t i = 0; i < (size_t)c.width_ * c.height_; i++)
      for (int ch = 0; ch < 4; ch++) cs[ch] += c.track_costs_[4 * i + ch];
    const size_t mid = (size_t)c.width_ * (c.height_ / 2) + c.width_ / 3;
    printf("costmap4 sums=[%.9g %.9g %.9g %.9g] texel%zu=[%.9g %.9g %.9g %.9g]\n", cs[0], cs[1], cs[2], cs[3], mid,
           c.track_costs_[4 * mid], c.track_costs_[4 * mid + 1], c.track_costs_[4 * mid + 2], c.track_costs_[4 * mid + 3]);
  }
  printf("host selftest OK (%zu params)\n", p.size());
  return 0;
}
