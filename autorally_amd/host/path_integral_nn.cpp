// path_integral_nn.cpp -- ROS-free equivalent of the reference binary `path_integral_nn`
// (src/path_integral/path_integral_main.cu with USE_NEURAL_NETWORK_MODEL__, :65-69): see
// path_integral_main.hpp.
//
// usage: path_integral_nn <launch.xml> [--rollouts K] [--layers 6-32-32-4] [--max-iter N]
//                         [--no-sleep] [--device D] [--trace file] [--set key=value ...]
#include "path_integral_main.hpp"

using namespace mppi_host;

int main(int argc, char **argv)
{
  // MPPI_NUM_ROLLOUTS__ = 1920, NeuralNetModel<7,2,3,6,32,32,4> (path_integral_main.cu:66-69)
  return path_integral_main<NeuralNetModel>(
      argc, argv, 1920, [](ParamMap &params, const std::vector<int> &layers, const float2_ *control_constraints) {
        std::unique_ptr<NeuralNetModel> m(new NeuralNetModel(layers, (float)(1.0 / (int)params["hz"]), control_constraints));
        m->loadParams((std::string)params["model_path"]);
        if (params.count("negate_yaw_der")) m->negate_yaw_der = (bool)params["negate_yaw_der"];
        return m;
      });
}
