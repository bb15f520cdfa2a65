// mfma4x4_ub.hip -- a 64 -> 64 layer as 64 DEPENDENT v_mfma_f32_4x4x1_16b_f32, one k per instruction, for the latency form of
// 64-wide nets (rollout_row64.hip).
//
// v_mfma_f32_4x4x1 multiplies, in each of 16 blocks, a 4x1 A by a 1x4 B into a 4x4 D (+= C).  With the A-matrix broadcast
// controls (CBSZ = 4, ABID = b') ALL 16 blocks take block b''s four A values.  So:
//   A = a VGPR whose lanes 4b'+i hold the activation a_i[k] of rollouts i = 0..3           (rows of D = rollouts)
//   B = a VGPR whose lane n holds W[n][k]                                                    (columns of D = neurons 4b+j)
//   D[i] (VGPR i), lane n = z_i[n]: one instruction is the step k of the k-ascending chain of neural_net_model.cu:379-394 for
//   4 rollouts x 64 neurons, the weights of a layer are 64 VGPRs (a lane holds ITS neuron's row: no copy per rollout), and
//   the broadcast of a_i[k] costs no instruction.
// Between layers the activations go from D's layout (VGPR = rollout, lane = neuron) to A's (lane-in-quad = rollout, VGPR =
// neuron-in-quad): a 4x4 transpose inside every quad (quad_perm moves + selects), once per layer.
// Part 1 probes the operand layouts (which A / B lane feeds D[v][lane], with and without the broadcast);
// part 2 times the layer (+ tanh + transpose), `iters` layers per launch, and checks the bits against the fmaf chain.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ub/mfma4x4_ub.hip -o mfma4x4_ub && ./mfma4x4_ub
#include "../../autorally_amd/csrc/mppi_device.hpp"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
using namespace mppi;

typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <int CBSZ, int ABID>
__global__ void k_probe(float *out, int mode)
{
  const int lane = threadIdx.x;
  const float a = (mode == 0) ? (float)(lane + 1) : 1.0f;
  const float b = (mode == 0) ? 1.0f : (float)(lane + 1);
  f32x4_t d = {0, 0, 0, 0};
  d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d, CBSZ, ABID, 0);
  for (int v = 0; v < 4; v++) out[v * 64 + lane] = d[v];
}

constexpr int H = 64;

template <int Q>
__device__ __forceinline__ float qp(float v)
{
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), Q, 0xF, 0xF, false));
}
// 4x4 transpose inside every quad: in[r] lane 4b+j  ->  out[s] lane 4b+i = in[i] lane 4b+s
__device__ __forceinline__ void quad_transpose(const float (&in)[4], float (&out)[4], int li)
{
  // stage 1: exchange the off-diagonal 2x2 blocks (registers r <-> r^2, lanes ^2)
  const bool hi = (li & 2) != 0;
  float t[4];
  {
    const float x0 = qp<0x4E>(hi ? in[0] : in[2]);  // quad_perm [2,3,0,1]: the partner's register of the other half
    const float x1 = qp<0x4E>(hi ? in[1] : in[3]);
    t[0] = hi ? x0 : in[0];
    t[2] = hi ? in[2] : x0;
    t[1] = hi ? x1 : in[1];
    t[3] = hi ? in[3] : x1;
  }
  // stage 2: inside each 2x2 block (registers r <-> r^1, lanes ^1)
  const bool od = (li & 1) != 0;
  {
    const float y0 = qp<0xB1>(od ? t[0] : t[1]);  // quad_perm [1,0,3,2]
    const float y1 = qp<0xB1>(od ? t[2] : t[3]);
    out[0] = od ? y0 : t[0];
    out[1] = od ? t[1] : y0;
    out[2] = od ? y1 : t[2];
    out[3] = od ? t[3] : y1;
  }
}

template <int K>
__device__ __forceinline__ void mstep(f32x4_t &d, const float (&T)[4], const float *w)
{
  d = __builtin_amdgcn_mfma_f32_4x4x1f32(T[K & 3], w[K], d, 4, K >> 2, 0);
}
template <int K0>
__device__ __forceinline__ void msteps16(f32x4_t &d, const float (&T)[4], const float *w)
{
#define S4(K) mstep<K>(d, T, w); mstep<K + 1>(d, T, w); mstep<K + 2>(d, T, w); mstep<K + 3>(d, T, w);
  S4(K0) S4(K0 + 4) S4(K0 + 8) S4(K0 + 12)
#undef S4
}

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_layer(const float *W, const float *B, float *out, unsigned long long *cyc, int iters)
{
  const int lane = threadIdx.x & 63, li = lane & 3;
  float w[H];
#pragma unroll
  for (int k = 0; k < H; k++) w[k] = W[lane * H + k];
#pragma unroll
  for (int k = 0; k < H; k++) asm volatile("" : "+v"(w[k]));
  const float bs = B[lane] * kTanhScale;
  // activations in D's layout: act[r] lane n = a_r[n]
  float act[4];
#pragma unroll
  for (int r = 0; r < 4; r++) act[r] = (0.01f * (float)lane - 0.3f) * (1.0f - 0.4f * (float)r);
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
    float T[4];
    quad_transpose(act, T, li);
    f32x4_t d = {0, 0, 0, 0};
    msteps16<0>(d, T, w);
    msteps16<16>(d, T, w);
    msteps16<32>(d, T, w);
    msteps16<48>(d, T, w);
    const f32x2 a01 = tanh_bias2(f32x2{d[0], d[1]}, f32x2{bs, bs});
    const f32x2 a23 = tanh_bias2(f32x2{d[2], d[3]}, f32x2{bs, bs});
    act[0] = a01.x; act[1] = a01.y; act[2] = a23.x; act[3] = a23.y;
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[blockIdx.x * WAVES + (threadIdx.x >> 6)] = c1 - c0;
  if (blockIdx.x == 0 && threadIdx.x < 64)
    for (int r = 0; r < 4; r++) out[r * 64 + lane] = act[r];
}

static float tanh_dev(float z, float b)
{
  const float y = fmaf(z, kTanhScale, b * kTanhScale);
  const float e = exp2f(y);
  return fmaf(-2.0f, 1.0f / (e + 1.0f), 1.0f);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <class F>
static int probe(const char *name, F launch, float *d_out)
{
  std::vector<float> o(256);
  for (int mode = 0; mode < 2; mode++) {
    launch(mode);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(o.data(), d_out, 256 * 4, hipMemcpyDeviceToHost));
    printf("%s  D[v][lane] <- %s lane:", name, mode == 0 ? "A" : "B");
    for (int v = 0; v < 4; v++) {
      printf("  v%d:", v);
      for (int l = 0; l < 12; l++) printf(" %d", (int)o[v * 64 + l] - 1);
      printf(" .. %d", (int)o[v * 64 + 63] - 1);
    }
    printf("\n");
  }
  return 0;
}

int main()
{
  float *d_out;
  CK(hipMalloc(&d_out, 256 * 4));
  if (probe("cbsz 0        ", [&](int m) { hipLaunchKernelGGL((k_probe<0, 0>), dim3(1), dim3(64), 0, 0, d_out, m); }, d_out)) return 1;
  if (probe("cbsz 4 abid 0 ", [&](int m) { hipLaunchKernelGGL((k_probe<4, 0>), dim3(1), dim3(64), 0, 0, d_out, m); }, d_out)) return 1;
  if (probe("cbsz 4 abid 5 ", [&](int m) { hipLaunchKernelGGL((k_probe<4, 5>), dim3(1), dim3(64), 0, 0, d_out, m); }, d_out)) return 1;

  const int iters = 2000, blocks = 256;
  std::vector<float> W(H * H), B(H);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
  for (auto &x : W) x = rnd() * 0.5f;
  for (auto &x : B) x = rnd() * 0.2f;
  std::vector<float> ref(256);
  for (int r = 0; r < 4; r++) {
    float a[H], n[H];
    for (int j = 0; j < H; j++) a[j] = (0.01f * (float)j - 0.3f) * (1.0f - 0.4f * (float)r);
    for (int it = 0; it < iters; it++) {
      for (int j = 0; j < H; j++) {
        float z = 0.0f;
        for (int k = 0; k < H; k++) z = fmaf(W[j * H + k], a[k], z);
        n[j] = tanh_dev(z, B[j]);
      }
      memcpy(a, n, sizeof(a));
    }
    for (int j = 0; j < H; j++) ref[r * 64 + j] = a[j];
  }
  float *dW, *dB;
  unsigned long long *d_cyc;
  CK(hipMalloc(&dW, W.size() * 4)); CK(hipMalloc(&dB, B.size() * 4));
  CK(hipMalloc(&d_cyc, (size_t)blocks * 8 * 8));
  CK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
  for (int waves : {4, 8}) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto launch = [&]() {
      if (waves == 4) hipLaunchKernelGGL((k_layer<4>), dim3(blocks), dim3(256), 0, 0, dW, dB, d_out, d_cyc, iters);
      else hipLaunchKernelGGL((k_layer<8>), dim3(blocks), dim3(512), 0, 0, dW, dB, d_out, d_cyc, iters);
    };
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    launch();
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<float> o(256);
    std::vector<unsigned long long> c((size_t)blocks * waves);
    CK(hipMemcpy(o.data(), d_out, 256 * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(c.data(), d_cyc, c.size() * 8, hipMemcpyDeviceToHost));
    double mx = 0, worst = 0;
    for (int i = 0; i < 256; i++) worst = fmax(worst, fabs((double)o[i] - (double)ref[i]));
    for (auto v : c) mx = fmax(mx, (double)v);
    printf("64x64 layer as 64 v_mfma_f32_4x4x1 (4 rollouts / wave) + tanh + quad transpose, %d wave(s) / SIMD: %8.1f ns / layer by events, "
           "%7.1f s_memtime ticks / layer   max |out - host| %.2e\n", waves / 4, 1e6 * ms / iters, mx / iters, worst);
  }
  return 0;
}
