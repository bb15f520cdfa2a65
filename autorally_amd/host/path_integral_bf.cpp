// path_integral_bf.cpp -- ROS-free equivalent of the reference binary `path_integral_bf`
// (src/path_integral/path_integral_main.cu with USE_BASIS_FUNC_MODEL__, :70-74): the same loop with
// GeneralizedLinear<CarBasisFuncs,7,2,25,CarKinematics,3> dynamics; see path_integral_main.hpp.
//
// usage: path_integral_bf <launch.xml> [--rollouts K] [--max-iter N] [--no-sleep] [--device D]
//                         [--trace file] [--set key=value ...]
#include "path_integral_main.hpp"

using namespace mppi_host;

int main(int argc, char **argv)
{
  // MPPI_NUM_ROLLOUTS__ = 2560 (path_integral_main.cu:71)
  return path_integral_main<GeneralizedLinear>(
      argc, argv, 2560, [](ParamMap &params, const std::vector<int> &, const float2_ *control_constraints) {
        std::unique_ptr<GeneralizedLinear> m(new GeneralizedLinear((float)(1.0 / (int)params["hz"]), control_constraints));
        m->loadParams((std::string)params["model_path"]);
        return m;
      });
}
