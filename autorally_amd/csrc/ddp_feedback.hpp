// ddp_feedback.hpp -- feedback gains around the MPPI solution.
//
// Reference: MPPIController::computeFeedbackGains (PI/mppi_controller.cu:431-441) -> DDP::run
// (ddp/ddp.h:49-157) with ModelWrapperDDP (ddp/ddp_model_wrapper.h:37-80), TrackingCostDDP /
// TrackingTerminalCost (ddp/ddp_tracking_costs.h:7-117) and the analytic network Jacobian
// NeuralNetModel::computeGrad (PI/neural_net_model.cu:233-264).  Like the reference this is host
// code: one forward rollout, T Jacobians, T-1 sequential 7x7 / 2x7 Riccati steps, one forward pass.
#pragma once

#include <vector>

namespace mppi {

constexpr int kDdpS = 7, kDdpC = 2, kDdpSC = kDdpS + kDdpC;

struct DdpNet {
  int n_layers;         // layer sizes incl. input and output; 0: basis-function model, theta = W[4][25]
  const int *layers;
  const float *theta;   // packed [W1|b1|W2|b2|...], W row-major [out][in] (neural_net_model.cu:120-141)
  int max_width;
};

struct DdpProblem {
  int T;                // horizon H = numTimesteps_
  float dt;             // (float)(1.0/hz), mppi_controller.cu:408
  float u_lo[kDdpC], u_hi[kDdpC];
  float Q[kDdpS], R[kDdpC], Qf[kDdpS];  // diagonals, mppi_controller.cu:410-417
  int negate_yaw_der;   // kinematics only; computeGrad hard-codes d(yaw rate)/d(s6) = -1 (neural_net_model.cu:241)
};

struct DdpResult {
  std::vector<float> feedback;     // [T][2][7]  Lk_ (the last one stays zero)
  std::vector<float> feedforward;  // [T][2]     lk_
  std::vector<float> x;            // [T][7]     state trajectory after the forward pass
  std::vector<float> u;            // [T][2]     control trajectory after the forward pass (last column zero)
  std::vector<float> cost;         // [T]        per-step cost of the accepted forward pass
  float total_cost = 0.0f;
  int iterations = 0;
};

// 0 on success; 1 when a 2x2 control Hessian could not be factorised (the reference exits with -3)
int ddp_feedback_gains(const DdpNet &net, const DdpProblem &p, const float *x0, const float *target_x,
                       const float *target_u, DdpResult &out);

}  // namespace mppi
