// host_selftest.cpp -- CPU-only checks of the host layer (no GPU, no libmppi_hip):
//   npz reader against a numpy-written model file and a round trip of the writer,
//   launch-XML loader against a launch file in the reference's format,
//   the headless plant's feedback law.
//   loadTrackData against a costmap file written by the reference's own track_converter.py (optional).
// usage: host_selftest <model.npz> <launch.xml> <tmp_dir> [costmap.npz]
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "npz.hpp"
#include "param_getter.hpp"
#include "run_control_loop.hpp"  // SimPlant only: nothing of libmppi_hip is referenced

using namespace mppi_host;

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
    plant.setSolution(ss, cs, g, ControllerType::ACTUAL_STATE);
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
