// run_control_loop.hpp -- ROS-free runControlLoop (PI/run_control_loop.cuh:84-321).
//
// Same structure as the reference loop: two controllers (actual-state / predicted-state), slide by
// the stride, computeControl(state) / computeControl(), arbitration by getComputedTrajectoryCost(),
// feedback gains of the chosen controller (use_feedback_gains), hand the chosen solution to the plant,
// debug-mode self-simulation, profiler_max_iter stop.
// The plant is a template parameter with the subset of AutorallyPlant the loop uses; SimPlant is the
// headless stand-in: without a pose source (status 1 => fixed stride, exactly the reference's debug_mode)
// or with a scripted pose clock (status 0 => the live-pose half of the loop: pose refresh every tick,
// stride from the pose stamps, pose-gated sleep).
#pragma once

#include <array>
#include <atomic>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdio>
#include <functional>
#include <thread>

#include "mppi_controller_hip.hpp"

namespace mppi_host {

enum class ControllerType { NONE, ACTUAL_STATE, PREDICTED_STATE };

struct SimPlant {
  struct FullState { float x_pos = 0, y_pos = 0, yaw = 0, roll = 0, u_x = 0, u_y = 0, yaw_mder = 0; };
  FullState fs;
  std::vector<float> last_state_seq, last_control_seq, last_feedback_gains;  // gains [T][2][7]
  ControllerType last_used = ControllerType::NONE;
  int n_solutions = 0, n_actual = 0;
  double avgLoop = 0, avgTick = 0, avgSleep = 0;
  double solution_ts = 0, solution_loop_speed = 0;
  FullState getState() const { return fs; }
  int n_ticks = 0;
  std::function<void(int)> on_tick;  // test hook: called at the top of every tick with its 1-based number
  void setTimingInfo(double a, double b, double c)
  {
    avgLoop = a; avgTick = b; avgSleep = c;
    n_ticks++;
    if (on_tick) on_tick(n_ticks);
  }

  // ---- pose source.  The headless plant has no state estimator: by default no pose ever arrives (status 1,
  // the reference's debug mode).  A SCRIPTED pose clock stands in for one in tests and dry runs: pose_script[i]
  // is the time (seconds) between the pose stamp tick i sees and the one tick i+1 sees.  When the loop hands
  // over the solution of tick i (setSolution), the plant "drives" it for round(dt hz) control periods through
  // `drive` (the model's updateState), publishes the resulting state and advances its pose stamp by dt.
  // 0 leaves the stamp alone (no new pose: the loop keeps its last optimisation-loop time).
  bool live = false;                 // pose estimates are arriving: checkStatus() == 0 (autorally_plant.cpp:443-459)
  double pose_time = 0.0;            // stamp of the last pose, seconds
  int hz = 50;
  std::vector<double> pose_script;
  size_t script_pos = 0;
  std::function<void(float *, float *)> drive;  // (state[7], control[2]) -> state advanced by one period
  std::vector<int> strides_driven;   // control periods driven after each tick (what the loop should slide by next)
  double getLastPoseTime() const { return pose_time; }  // AutorallyPlant::getLastPoseTime, :437-441
  int checkStatus() const { return live ? 0 : 1; }

  // setSolution(traj, controls, gains, ts, loop_speed, controller_type_used), autorally_plant.cpp:107-126
  void setSolution(const std::vector<float> &ss, const std::vector<float> &cs, const std::vector<float> &gains,
                   double ts, double loop_speed, ControllerType used)
  {
    last_state_seq = ss;
    last_control_seq = cs;
    last_feedback_gains = gains;
    last_used = used;
    solution_ts = ts;
    solution_loop_speed = loop_speed;
    n_solutions++;
    if (used == ControllerType::ACTUAL_STATE) n_actual++;
    if (live && script_pos < pose_script.size()) {
      const double dt = pose_script[script_pos++];
      const int n = (int)std::lround(dt * hz);
      strides_driven.push_back(dt > 0 ? n : -1);
      if (dt > 0) {
        float x[7] = {fs.x_pos, fs.y_pos, fs.yaw, fs.roll, fs.u_x, fs.u_y, fs.yaw_mder};
        for (int t = 0; t < n && drive; t++) {
          float u[2] = {cs[2 * (size_t)t], cs[2 * (size_t)t + 1]};
          drive(x, u);
        }
        fs.x_pos = x[0]; fs.y_pos = x[1]; fs.yaw = x[2]; fs.roll = x[3]; fs.u_x = x[4]; fs.u_y = x[5]; fs.yaw_mder = x[6];
        pose_time += dt;
      }
    }
  }
  void setSolution(const std::vector<float> &ss, const std::vector<float> &cs, const std::vector<float> &gains,
                   ControllerType used)
  {
    setSolution(ss, cs, gains, pose_time, 0.0, used);
  }
  // Live updates the plant relays from ROS topics / dynamic_reconfigure (autorally_plant.cpp:262-310).  The
  // headless plant has none unless a test or a caller injects them here.
  bool new_dcfg = false, new_model = false, want_debug_image = false;
  PathIntegralParamsConfig dcfg;
  std::vector<int> model_description;
  std::vector<float> model_data;  // [W1|W2|..|b1|b2|..] as AutorallyPlant::getModel delivers it
  std::vector<float> debug_image;
  bool hasNewDynRcfg() const { return new_dcfg; }
  PathIntegralParamsConfig getDynRcfgParams() { new_dcfg = false; return dcfg; }
  bool hasNewObstacles() const { return false; }
  void getObstacles(std::vector<int> &, std::vector<float> &) {}
  bool hasNewCostmap() const { return false; }
  void getCostmap(std::vector<int> &, std::vector<float> &) {}
  bool hasNewModel() const { return new_model; }
  void getModel(std::vector<int> &description, std::vector<float> &data) { description = model_description; data = model_data; new_model = false; }
  void setDebugImage(const std::vector<float> &img) { debug_image = img; }
  // pubControl's feedback law (autorally_plant.cpp:217-250) at a time `since` seconds after the
  // solution, for a measured state: u = u_ff(t) + K(t) (x - x_des(t)), all linearly interpolated.
  // false when outside the solution's horizon (the plant then publishes nothing).
  bool controlAt(double since, double dt, const float x[7], bool use_feedback, float u_out[2]) const
  {
    const int T = (int)(last_control_seq.size() / 2);
    if (T < 2 || !(since > 0) || !(since < (T - 1) * dt)) return false;
    const int lo = (int)(since / dt), hi = lo + 1;
    const double a = (since - lo * dt) / dt;
    double u[2];
    for (int j = 0; j < 2; j++) u[j] = (1 - a) * last_control_seq[2 * lo + j] + a * last_control_seq[2 * hi + j];
    if (use_feedback && last_feedback_gains.size() == (size_t)T * 14) {
      float du[2] = {0.0f, 0.0f};
      for (int j = 0; j < 2; j++)
        for (int i = 0; i < 7; i++) {
          const float des = (float)((1 - a) * last_state_seq[7 * lo + i] + a * last_state_seq[7 * hi + i]);
          const float k = (float)((1 - a) * last_feedback_gains[(lo * 2 + j) * 7 + i] + a * last_feedback_gains[(hi * 2 + j) * 7 + i]);
          du[j] += k * (x[i] - des);
        }
      if (!std::isnan(du[0]) && !std::isnan(du[1])) {  // :243-246
        u[0] += du[0];
        u[1] += du[1];
      }
    }
    for (int j = 0; j < 2; j++) u_out[j] = (float)std::fmax(-1.0, std::fmin(1.0, u[j]));  // :252-253
    return true;
  }
};

struct LoopStats {
  int iterations = 0;
  double avg_tick_ms = 0, avg_sleep_ms = 0, avg_loop_ms = 0;
  // where a tick goes (ms, means): slide + live updates | both solves + both nominal replays | feedback gains | arbitration,
  // hand-over to the plant, debug-mode state update
  double avg_pre_ms = 0, avg_solve_ms = 0, avg_gains_ms = 0, avg_post_ms = 0;
  std::array<float, 7> final_state{};
  std::vector<int> strides;  // the stride every tick slid by (-1: no slide)
};

// runControlLoop (PI/run_control_loop.cuh:84-321).  PLANT_T is the subset of AutorallyPlant the loop uses:
//   getState(), getLastPoseTime() [seconds], checkStatus(), setTimingInfo, setSolution(traj, controls, gains,
//   ts, loop_speed, used), hasNew{DynRcfg,Obstacles,Costmap,Model} + getters, setDebugImage, want_debug_image.
// Both halves of the reference loop are here: with status != 0 (no pose estimates: debug mode, or the car
// stopped) the sequences slide by optimization_stride and, in debug mode, the loop simulates the car itself;
// with status == 0 the newest pose is pulled every tick, the sequences slide by the number of control periods
// between the last two pose stamps (:208-211) and the tick ends only once the next pose is almost due (:308).
// sleep_to_rate = false (tests, profiling) skips every sleep and wait.
template <class CONTROLLER_T, class PLANT_T>
LoopStats runControlLoop(CONTROLLER_T *predicted_state_controller, CONTROLLER_T *actual_state_controller,
                         PLANT_T *robot, ParamMap *params, std::atomic<bool> *is_alive, bool sleep_to_rate = true,
                         FILE *trace = nullptr)
{
  const float x_pos = (float)(double)(*params)["x_pos"];
  const float y_pos = (float)(double)(*params)["y_pos"];
  const float heading = (float)(double)(*params)["heading"];
  const int hz = (int)(*params)["hz"];
  const int optimization_stride = (int)(*params)["optimization_stride"];
  const int num_timesteps = (int)(*params)["num_timesteps"];
  const bool debug_mode = (bool)(*params)["debug_mode"];
  const bool only_actual = (bool)(*params)["use_only_actual_state_controller"];
  const bool only_predicted = (bool)(*params)["use_only_predicted_state_controller"];
  const int max_iter = params->count("profiler_max_iter") ? (int)(*params)["profiler_max_iter"] : INT_MAX;
  const bool use_feedback_gains = params->count("use_feedback_gains") ? (bool)(*params)["use_feedback_gains"] : false;  // :100

  float state[7] = {x_pos, y_pos, heading, 0, 0, 0, 0};
  std::vector<float> controlSolution, stateSolution, feedback_gain;
  LoopStats st;
  int num_iter = 0, status = 1;
  double avgOptimizationLoopTime = 0, avgTick = 0, avgSleep = 0;  // ms (:131-133)
  double last_pose_update = robot->getLastPoseTime();                  // :134
  double optimizationLoopTime = optimization_stride / (1.0 * hz);     // :135, seconds
  const std::chrono::duration<double, std::milli> period((int)(optimization_stride * 1000.0 / hz));  // :138

  if (!debug_mode && sleep_to_rate) {  // :140-144: wait until a pose estimate has arrived
    while (last_pose_update == robot->getLastPoseTime() && is_alive->load())
      std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
  {
    const typename PLANT_T::FullState fs = robot->getState();  // :146-147 (all zeros from a plant without poses)
    const float s[7] = {fs.x_pos, fs.y_pos, fs.yaw, fs.roll, fs.u_x, fs.u_y, fs.yaw_mder};
    if (!debug_mode)
      for (int i = 0; i < 7; i++) state[i] = s[i];
    // debug mode: the reference overwrites the launch file's start pose with the plant's full_state_ here,
    // which nothing has written at that point without a pose source (autorally_plant.cpp:75-83 sets yaw_mder
    // only: the other members are uninitialised).  Not reproduced: the launch file's x_pos / y_pos / heading stand.
  }
  actual_state_controller->setState(state);
  predicted_state_controller->setState(state);
  actual_state_controller->resetControls();
  actual_state_controller->computeFeedbackGains(state);  // :151-154 (gains around the initial sequence)
  predicted_state_controller->resetControls();
  predicted_state_controller->computeFeedbackGains(state);

  while (is_alive->load() && num_iter < max_iter) {
    const auto loop_start = std::chrono::steady_clock::now();
    robot->setTimingInfo(avgOptimizationLoopTime, avgTick, avgSleep);
    num_iter++;
    if (debug_mode && robot->want_debug_image) {  // :162-174, the raster around the predicted state (display left to the plant)
      const std::vector<float> seq = predicted_state_controller->getStateSeq();
      robot->setDebugImage(predicted_state_controller->costs_->getDebugDisplay(seq[0], seq[1], seq[2]));
    }
    // Update the state estimate, :175-181
    if (last_pose_update != robot->getLastPoseTime()) {
      optimizationLoopTime = robot->getLastPoseTime() - last_pose_update;
      last_pose_update = robot->getLastPoseTime();
      const typename PLANT_T::FullState fs = robot->getState();
      const float s[7] = {fs.x_pos, fs.y_pos, fs.yaw, fs.roll, fs.u_x, fs.u_y, fs.yaw_mder};
      for (int i = 0; i < 7; i++) state[i] = s[i];
    }
    // live updates relayed by the plant, :182-204
    if (robot->hasNewDynRcfg()) {
      const PathIntegralParamsConfig c = robot->getDynRcfgParams();
      actual_state_controller->costs_->updateParams_dcfg(c);
      predicted_state_controller->costs_->updateParams_dcfg(c);
    }
    if (robot->hasNewObstacles()) {
      std::vector<int> d; std::vector<float> v;
      robot->getObstacles(d, v);
      actual_state_controller->costs_->updateObstacles(d, v);
      predicted_state_controller->costs_->updateObstacles(d, v);
    }
    if (robot->hasNewCostmap()) {
      std::vector<int> d; std::vector<float> v;
      robot->getCostmap(d, v);
      actual_state_controller->costs_->updateCostmap(d, v);
      predicted_state_controller->costs_->updateCostmap(d, v);
    }
    if (robot->hasNewModel()) {
      std::vector<int> d; std::vector<float> v;
      robot->getModel(d, v);
      actual_state_controller->model_->updateModel(d, v);
      predicted_state_controller->model_->updateModel(d, v);
    }
    // how many controls have been published since we were last here, :206-216
    int stride = (int)std::round(optimizationLoopTime * hz);
    if (status != 0) stride = optimization_stride;
    if (stride >= 0 && stride < num_timesteps) {
      actual_state_controller->slideControlAndStateSeq(stride);
      predicted_state_controller->slideControlAndStateSeq(stride);
      st.strides.push_back(stride);
    } else {
      st.strides.push_back(-1);
    }
    // computeControl(state) / computeControl() (:218-219); the two solves are independent, so both are
    // put on the GPU together -- one launch for the rollouts of both controllers -- before either is waited for
    const auto t_solve0 = std::chrono::steady_clock::now();
    CONTROLLER_T::startControlPair(actual_state_controller, state, predicted_state_controller);
    CONTROLLER_T::finishControlPair(actual_state_controller, predicted_state_controller);
    const auto t_solve1 = std::chrono::steady_clock::now();
    if (use_feedback_gains)  // :220-225: both controllers, from the measured state
      CONTROLLER_T::computeFeedbackGainsPair(actual_state_controller, predicted_state_controller, state);
    const auto t_gains1 = std::chrono::steady_clock::now();
    feedback_gain = predicted_state_controller->getFeedbackGains().feedback_gain;  // :229

    ControllerType to_use = ControllerType::NONE;
    if (only_actual && !only_predicted) to_use = ControllerType::ACTUAL_STATE;
    else if (!only_actual && only_predicted) to_use = ControllerType::PREDICTED_STATE;
    ControllerType used = ControllerType::NONE;
    if (to_use == ControllerType::ACTUAL_STATE ||
        (to_use == ControllerType::NONE &&
         actual_state_controller->getComputedTrajectoryCost() < predicted_state_controller->getComputedTrajectoryCost())) {
      controlSolution = actual_state_controller->getControlSeq();
      stateSolution = actual_state_controller->getStateSeq();
      feedback_gain = actual_state_controller->getFeedbackGains().feedback_gain;  // :254
      if (to_use == ControllerType::NONE) {  // :255-258
        predicted_state_controller->setStateSequence(stateSolution);
        predicted_state_controller->setControlSequence(controlSolution);
      }
      used = ControllerType::ACTUAL_STATE;
    } else {
      controlSolution = predicted_state_controller->getControlSeq();
      stateSolution = predicted_state_controller->getStateSeq();
      feedback_gain = predicted_state_controller->getFeedbackGains().feedback_gain;  // :267
      used = ControllerType::PREDICTED_STATE;
    }
    robot->setSolution(stateSolution, controlSolution, feedback_gain, last_pose_update, avgOptimizationLoopTime, used);  // :286-287
    status = robot->checkStatus();
    if (status != 0 && debug_mode) {
      // :296-302 -- both controllers share ONE model object, so the reference advances `state`
      // twice per executed control (model_->updateState is called through each controller)
      for (int t = 0; t < optimization_stride; t++) {
        float u[2] = {controlSolution[2 * t], controlSolution[2 * t + 1]};
        actual_state_controller->model_->updateState(state, u);
        predicted_state_controller->model_->updateState(state, u);
      }
    }
    if (trace)
      fprintf(trace, "%d %s %.6f %.6f | %.5f %.5f %.5f %.5f %.5f | %.5f %.5f | %d\n", num_iter,
              used == ControllerType::ACTUAL_STATE ? "actual" : "predicted",
              actual_state_controller->getComputedTrajectoryCost(),
              predicted_state_controller->getComputedTrajectoryCost(), state[0], state[1], state[2], state[4],
              state[5], controlSolution[0], controlSolution[1], st.strides.back());
    // sleep for any leftover time, and -- with a pose source -- until the next pose is almost due, :304-312
    std::chrono::duration<double, std::milli> fp_ms = std::chrono::steady_clock::now() - loop_start;
    const double tick = fp_ms.count();
    {
      const auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
      };
      const double pre = ms(loop_start, t_solve0), solve = ms(t_solve0, t_solve1), gains = ms(t_solve1, t_gains1);
      st.avg_pre_ms += (pre - st.avg_pre_ms) / num_iter;
      st.avg_solve_ms += (solve - st.avg_solve_ms) / num_iter;
      st.avg_gains_ms += (gains - st.avg_gains_ms) / num_iter;
      st.avg_post_ms += ((tick - pre - solve - gains) - st.avg_post_ms) / num_iter;
    }
    while (sleep_to_rate && is_alive->load() &&
           (fp_ms < period || ((robot->getLastPoseTime() - last_pose_update) < (1.0 / hz - 0.0025) && status == 0))) {
      std::this_thread::sleep_for(std::chrono::microseconds(50));
      fp_ms = std::chrono::steady_clock::now() - loop_start;
    }
    const double sleep = fp_ms.count() - tick;
    // :315-318
    avgOptimizationLoopTime = (num_iter - 1.0) / num_iter * avgOptimizationLoopTime + 1000.0 * optimizationLoopTime / num_iter;
    avgTick = (num_iter - 1.0) / num_iter * avgTick + tick / num_iter;
    avgSleep = (num_iter - 1.0) / num_iter * avgSleep + sleep / num_iter;
  }
  st.iterations = num_iter;
  st.avg_tick_ms = avgTick;
  st.avg_sleep_ms = avgSleep;
  st.avg_loop_ms = avgOptimizationLoopTime;
  for (int i = 0; i < 7; i++) st.final_state[i] = state[i];
  return st;
}

}  // namespace mppi_host
