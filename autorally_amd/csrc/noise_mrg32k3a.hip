// noise_mrg32k3a.hip -- control-noise generation for gfx950.
//
// Replaces curandGenerateNormal(gen_, du_d_, K*T*2, 0, 1) (PI/mppi_controller.cu:330-331, 612).
// cuRAND's XORWOW bitstream cannot be reproduced outside cuRAND, so this build defines its own
// generator (DESIGN.md "noise spec"): MRG32k3a (L'Ecuyer 1999), rollout k owns the k-th
// subsequence of 2^76 draws (cuRAND's MRG32k3a spacing), two draws per timestep are turned into
// (eps[k][t][0], eps[k][t][1]) by a Box-Muller transform written only in IEEE basic operations,
// so a CPU statement of the same spec (the test oracle) matches bit for bit.
//
// Parallelisation: thread = (rollout k, chunk c of L timesteps); the chunk's start state is the
// rollout's current state advanced by 2*L*c draws with a host-computed 3x3 jump matrix; the last
// chunk writes the state after 2T draws to the ping-pong state buffer.  Output is written
// time-major [T][K][2] (coalesced 8-B per lane, consecutive k).
#include "noise_device.hpp"

namespace mppi {

// rng: [6][K] current per-rollout states (SoA). jump: [C][18] = A1^(2Lc) | A2^(2Lc).
__global__ __launch_bounds__(256) void noise_kernel(const uint32_t *__restrict__ rng_in,
                                                    uint32_t *__restrict__ rng_out,
                                                    const uint32_t *__restrict__ jump, int K, int T,
                                                    int L, int C, float2 *__restrict__ eps)
{
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= K * C) return;
  const int c = tid / K;  // wave-uniform (K % 64 == 0)
  const int k = tid - c * K;
  Mrg g;
  g.s10 = rng_in[k]; g.s11 = rng_in[K + k]; g.s12 = rng_in[2 * K + k];
  g.s20 = rng_in[3 * K + k]; g.s21 = rng_in[4 * K + k]; g.s22 = rng_in[5 * K + k];
  if (c > 0) {
    const uint32_t *J = jump + (size_t)c * 18;
    mat3_vec_m1(J, g.s10, g.s11, g.s12);
    mat3_vec_m2(J + 9, g.s20, g.s21, g.s22);
  }
  const int t0 = c * L, t1 = min(T, t0 + L);
  for (int t = t0; t < t1; t++) {
    eps[(size_t)t * K + k] = noise_pair(g);
  }
  if (c == C - 1) {
    rng_out[k] = g.s10; rng_out[K + k] = g.s11; rng_out[2 * K + k] = g.s12;
    rng_out[3 * K + k] = g.s20; rng_out[4 * K + k] = g.s21; rng_out[5 * K + k] = g.s22;
  }
}

// Per-rollout start states: base advanced by k*2^76 (binary expansion of k over sub[b] =
// A^(2^76 * 2^b)) and by `offset` draws (binary expansion over one[b] = A^(2^b)).
__global__ void noise_init_kernel(uint32_t *__restrict__ rng, int K, Mrg base,
                                  const uint32_t *__restrict__ sub, int sub_bits,
                                  const uint32_t *__restrict__ one, uint64_t offset)
{
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  Mrg g = base;
  for (int b = 0; b < sub_bits; b++)
    if ((k >> b) & 1) {
      mat3_vec_m1(sub + b * 18, g.s10, g.s11, g.s12);
      mat3_vec_m2(sub + b * 18 + 9, g.s20, g.s21, g.s22);
    }
  for (int b = 0; b < 64; b++)
    if ((offset >> b) & 1ULL) {
      mat3_vec_m1(one + b * 18, g.s10, g.s11, g.s12);
      mat3_vec_m2(one + b * 18 + 9, g.s20, g.s21, g.s22);
    }
  rng[k] = g.s10; rng[K + k] = g.s11; rng[2 * K + k] = g.s12;
  rng[3 * K + k] = g.s20; rng[4 * K + k] = g.s21; rng[5 * K + k] = g.s22;
}

hipError_t launch_noise(const uint32_t *rng_in, uint32_t *rng_out, const uint32_t *jump, int K, int T,
                        int L, int C, float *eps, hipStream_t stream)
{
  const int total = K * C;
  hipLaunchKernelGGL(noise_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, rng_in, rng_out,
                     jump, K, T, L, C, reinterpret_cast<float2 *>(eps));
  return hipGetLastError();
}

hipError_t launch_noise_init(uint32_t *rng, int K, const uint32_t base[6], const uint32_t *sub,
                             int sub_bits, const uint32_t *one, uint64_t offset, hipStream_t stream)
{
  Mrg b{base[0], base[1], base[2], base[3], base[4], base[5]};
  hipLaunchKernelGGL(noise_init_kernel, dim3((K + 255) / 256), dim3(256), 0, stream, rng, K, b, sub,
                     sub_bits, one, offset);
  return hipGetLastError();
}

}  // namespace mppi
