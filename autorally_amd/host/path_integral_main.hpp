// path_integral_main.hpp -- the body of the reference's path_integral_main.cu:80-153, shared by the two
// ROS-free binaries (path_integral_nn: NeuralNetModel, path_integral_bf: GeneralizedLinear; the
// reference selects the model with USE_NEURAL_NETWORK_MODEL__ / USE_BASIS_FUNC_MODEL__, :65-78).
// Reads the SAME launch XML (same keys, same $(env AR_MPPI_PARAMS_PATH) expansion) and the same
// cnpy-format .npz model / costmap files, builds costs + model + two controllers and runs the control
// loop in the reference's debug_mode (self-simulation, PI/run_control_loop.cuh:296-302).  ROS does not
// exist in this image, so there is no plant I/O; `profiler_max_iter` (or --max-iter) ends the run like
// the reference's profiler mode.
#pragma once

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>

#include "run_control_loop.hpp"

namespace mppi_host {

inline std::vector<int> parse_layers(const std::string &s)
{
  std::vector<int> v;
  size_t i = 0;
  while (i < s.size()) {
    size_t j = s.find('-', i);
    if (j == std::string::npos) j = s.size();
    v.push_back(std::atoi(s.substr(i, j - i).c_str()));
    i = j + 1;
  }
  return v;
}

// make_model(params, layers, control_constraints) -> std::unique_ptr<DYNAMICS_T>
template <class DYNAMICS_T, class MAKE_MODEL>
int path_integral_main(int argc, char **argv, int default_rollouts, MAKE_MODEL make_model)
{
  if (argc < 2) {
    fprintf(stderr, "usage: %s <launch.xml> [--rollouts K] [--layers 6-32-32-4] [--max-iter N] [--no-sleep] [--host-threads 1|2] [--rollout-variant auto|mfma|...] "
                    "[--device D] [--trace file] [--set key=value]\n", argv[0]);
    return 2;
  }
  int rollouts = default_rollouts;  // MPPI_NUM_ROLLOUTS__, path_integral_main.cu:66,71
  std::vector<int> layers = {6, 32, 32, 4};  // NeuralNetModel<7,2,3,6,32,32,4>, :69
  int max_iter = -1, device = 0;
  bool sleep_to_rate = true;
  const char *trace_path = nullptr;
  bool have_dcfg = false, debug_image = false;  // stand-ins for the plant's dynamic_reconfigure / debug window
  double dcfg_speed = 0.0;
  std::vector<std::string> overrides;
  // --pose-script "0.02,0.04,0,0.06": a scripted pose source instead of the debug-mode self-simulation -- the
  // time between the pose stamps of successive ticks (SimPlant); the live-pose half of runControlLoop runs
  const char *pose_script = nullptr;
  // --poke-desired-speed V / --poke-max-throttle V: at tick 5 a PLAIN WRITE to the public members
  // costs.params_.desired_speed / model.control_rngs_[1].y (no setter, no version bump), as code written against
  // the reference does (e.g. cutThrottle, mppi_controller.cu:460-466); the next solve must see it
  double poke_speed = -1.0, poke_throttle = -1.0;
  // --host-threads 1|2: mppi_set_host_threads -- 2 (default here): the two controllers' nominal replays and DDP passes of
  // a tick side by side, on the optimizer thread and one helper (the reference runs them one after the other)
  int host_threads = 2;
  // --rollout-variant NAME: mppi_set_rollout_variant on both controllers ("mfma": the reference's summation order in every layer)
  const char *rollout_variant = nullptr;
  for (int i = 2; i < argc; i++) {
    if (!strcmp(argv[i], "--rollouts") && i + 1 < argc) rollouts = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--layers") && i + 1 < argc) layers = parse_layers(argv[++i]);
    else if (!strcmp(argv[i], "--max-iter") && i + 1 < argc) max_iter = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--trace") && i + 1 < argc) trace_path = argv[++i];
    else if (!strcmp(argv[i], "--set") && i + 1 < argc) overrides.push_back(argv[++i]);
    else if (!strcmp(argv[i], "--no-sleep")) sleep_to_rate = false;
    else if (!strcmp(argv[i], "--host-threads") && i + 1 < argc) host_threads = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--rollout-variant") && i + 1 < argc) rollout_variant = argv[++i];
    else if (!strcmp(argv[i], "--dcfg-desired-speed") && i + 1 < argc) { dcfg_speed = atof(argv[++i]); have_dcfg = true; }
    else if (!strcmp(argv[i], "--debug-image")) debug_image = true;
    else if (!strcmp(argv[i], "--pose-script") && i + 1 < argc) pose_script = argv[++i];
    else if (!strcmp(argv[i], "--poke-desired-speed") && i + 1 < argc) poke_speed = atof(argv[++i]);
    else if (!strcmp(argv[i], "--poke-max-throttle") && i + 1 < argc) poke_throttle = atof(argv[++i]);
    else { fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
  }
  try {
    if (mppi_set_host_threads(host_threads) != MPPI_OK) throw std::runtime_error("--host-threads must be 1 or 2");
    ParamMap params;
    loadParams(&params, argv[1]);
    for (const std::string &kv : overrides) {  // e.g. --set x_pos=0.0 (double), typed like the existing key
      const size_t eq = kv.find('=');
      if (eq == std::string::npos) throw std::runtime_error("--set needs key=value");
      const std::string k = kv.substr(0, eq), v = kv.substr(eq + 1);
      const ParamValue::Type t = params.count(k) ? params[k].getType() : ParamValue::TypeString;
      if (t == ParamValue::TypeInt) params[k] = ParamValue(std::stoi(v));
      else if (t == ParamValue::TypeDouble) params[k] = ParamValue(std::stod(v));
      else if (t == ParamValue::TypeBoolean) params[k] = ParamValue(v == "true");
      else params[k] = ParamValue(v);
    }
    if (max_iter >= 0) params["profiler_max_iter"] = ParamValue(max_iter);
    params["debug_mode"] = ParamValue(pose_script == nullptr);  // headless: self-simulation unless poses are scripted

    MPPICosts costs(&params);
    const float2_ control_constraints[2] = {{-.99f, .99f}, {-.99f, (float)(double)params["max_throttle"]}};
    std::unique_ptr<DYNAMICS_T> model_p = make_model(params, layers, control_constraints);
    DYNAMICS_T &model = *model_p;

    float exploration_std[2] = {(float)(double)params["steering_std"], (float)(double)params["throttle_std"]};
    float init_u[2] = {(float)(double)params["init_steering"], (float)(double)params["init_throttle"]};
    const int hz = (int)params["hz"], T = (int)params["num_timesteps"], stride = (int)params["optimization_stride"];
    const float gamma = (float)(double)params["gamma"];
    const int num_iters = (int)params["num_iters"];

    // both controllers seed their generator with 1234 (mppi_controller.cu:331): identical streams
    MPPIControllerT<DYNAMICS_T> actual(&model, &costs, exploration_std, init_u, hz, T, stride, gamma, num_iters, rollouts, device);
    MPPIControllerT<DYNAMICS_T> predicted(&model, &costs, exploration_std, init_u, hz, T, stride, gamma, num_iters, rollouts, device);
    if (rollout_variant) {
      actual.setRolloutVariant(rollout_variant);
      predicted.setRolloutVariant(rollout_variant);
    }
    SimPlant robot;
    if (have_dcfg) {  // one dynamic_reconfigure message waiting at the first tick (cfg defaults + the given speed)
      robot.new_dcfg = true;
      robot.dcfg.desired_speed = dcfg_speed;
    }
    robot.want_debug_image = debug_image;
    if (poke_speed >= 0.0 || poke_throttle >= 0.0)
      robot.on_tick = [&costs, &model, poke_speed, poke_throttle](int tick) {
        if (tick != 5) return;
        if (poke_speed >= 0.0) costs.params_.desired_speed = (float)poke_speed;
        if (poke_throttle >= 0.0) model.control_rngs_[1].y = (float)poke_throttle;
      };
    if (pose_script) {
      sleep_to_rate = false;  // the scripted clock only advances when a solution is handed over: nothing to wait for
      robot.live = true;
      robot.hz = hz;
      robot.pose_time = 1000.0;  // any stamp
      robot.fs.x_pos = (float)(double)params["x_pos"];
      robot.fs.y_pos = (float)(double)params["y_pos"];
      robot.fs.yaw = (float)(double)params["heading"];
      for (const char *c = pose_script; *c;) {
        char *e = nullptr;
        robot.pose_script.push_back(strtod(c, &e));
        c = (*e == ',') ? e + 1 : e;
        if (e == c && *e) break;
      }
      robot.drive = [&model](float *x, float *u) { model.updateState(x, u); };
    }
    std::atomic<bool> is_alive(true);
    FILE *trace = trace_path ? fopen(trace_path, "w") : nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    const LoopStats st = runControlLoop(&predicted, &actual, &robot, &params, &is_alive, sleep_to_rate, trace);
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (trace) fclose(trace);
    // the gains handed to the plant with the last solution: first-step row sums as a fingerprint
    double g0 = 0.0, g1 = 0.0;
    if (robot.last_feedback_gains.size() >= 14)
      for (int i = 0; i < 7; i++) { g0 += robot.last_feedback_gains[i]; g1 += robot.last_feedback_gains[7 + i]; }
    double img_sum = 0.0;
    for (float v : robot.debug_image) img_sum += v;
    std::string strides;
    for (int v : st.strides) strides += (strides.empty() ? "" : " ") + std::to_string(v);
    printf("{\"iterations\": %d, \"rollouts\": %d, \"timesteps\": %d, \"avg_tick_ms\": %.4f, \"tick_parts_ms\": {\"slide_updates\": %.4f, "
           "\"solves_and_replays\": %.4f, \"gains\": %.4f, \"arbitration_handover\": %.4f}, \"avg_sleep_ms\": %.4f, "
           "\"wall_s\": %.4f, \"actual_state_used\": %d, \"final_state\": [%.6f, %.6f, %.6f, %.6f, %.6f, %.6f, %.6f], "
           "\"feedback_gain_row_sums_t0\": [%.6f, %.6f], \"desired_speed\": %.4f, \"debug_image_pixels\": %zu, "
           "\"debug_image_sum\": %.4f, \"avg_loop_ms\": %.4f, \"strides\": \"%s\", \"plant_state\": [%.6f, %.6f, %.6f, %.6f, %.6f, %.6f, %.6f]}\n",
           st.iterations, actual.NUM_ROLLOUTS, T, st.avg_tick_ms, st.avg_pre_ms, st.avg_solve_ms, st.avg_gains_ms, st.avg_post_ms,
           st.avg_sleep_ms, wall, robot.n_actual,
           st.final_state[0], st.final_state[1], st.final_state[2], st.final_state[3], st.final_state[4],
           st.final_state[5], st.final_state[6], g0, g1, costs.params_.desired_speed, robot.debug_image.size(), img_sum,
           st.avg_loop_ms, strides.c_str(), robot.fs.x_pos, robot.fs.y_pos, robot.fs.yaw, robot.fs.roll, robot.fs.u_x,
           robot.fs.u_y, robot.fs.yaw_mder);
  } catch (const std::exception &e) {
    fprintf(stderr, "path_integral_nn: %s\n", e.what());
    return 1;
  }
  return 0;
}

}  // namespace mppi_host
