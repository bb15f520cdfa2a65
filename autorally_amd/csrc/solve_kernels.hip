// solve_kernels.hip -- everything of computeControl (PI/mppi_controller.cu:600-675) that is not
// the rollout: baseline/normExp/normaliser, weightedReductionKernel, savitskyGolay, plus the
// [K][T][2] <-> [T][K][2] layout transposes used at the ABI boundary.
#include "mppi_device.hpp"

namespace mppi {

// ---------------------------------------------------------------------------------------------
// weights: beta = min_k J_k ; w_k = expf(-gamma (J_k - beta)) ; eta = sum w ; traj = sum w^2/eta
// Reference: host loops + normExpKernel, mppi_controller.cu:627-652, 193-203 (two D2H round
// trips and two stream syncs there; one single-workgroup kernel here, results stay in HBM).
// Sums are pairwise (LDS tree) instead of the host's sequential fp32 loop: same value to ~1e-7.
// scal[0]=beta scal[1]=eta scal[2]=trajectory_cost
// ---------------------------------------------------------------------------------------------
constexpr int kWeightsThreads = 1024;

__device__ __forceinline__ float wave_min(float v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__global__ __launch_bounds__(kWeightsThreads) void weights_kernel(const float *__restrict__ costs,
                                                                   int K, float gamma,
                                                                   float *__restrict__ w,
                                                                   float *__restrict__ wn,
                                                                   float *__restrict__ scal)
{
  __shared__ float red[kWeightsThreads / 64];
  __shared__ float bcast;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  float m = INFINITY;
  for (int k = tid; k < K; k += kWeightsThreads) m = fminf(m, costs[k]);
  m = wave_min(m);
  if (lane == 0) red[wv] = m;
  __syncthreads();
  if (wv == 0) {
    float v = (lane < kWeightsThreads / 64) ? red[lane] : INFINITY;
    v = wave_min(v);
    if (lane == 0) bcast = v;
  }
  __syncthreads();
  const float beta = bcast;
  __syncthreads();
  float part = 0.0f;
  for (int k = tid; k < K; k += kWeightsThreads) {
    const float cost2go = costs[k] - beta;
    const float e = expf(-gamma * cost2go);  // normExpKernel :201
    w[k] = e;
    part += e;
  }
  part = wave_sum(part);
  if (lane == 0) red[wv] = part;
  __syncthreads();
  if (wv == 0) {
    float v = (lane < kWeightsThreads / 64) ? red[lane] : 0.0f;
    v = wave_sum(v);
    if (lane == 0) bcast = v;
  }
  __syncthreads();
  const float eta = bcast;
  __syncthreads();
  float tc = 0.0f;
  for (int k = tid; k < K; k += kWeightsThreads) {
    const float e = w[k];  // written by this same thread above
    tc += e * e / eta;     // :651 (Q8)
    wn[k] = e / eta;       // the per-use divide of weightedReductionKernel :244, hoisted
  }
  tc = wave_sum(tc);
  if (lane == 0) red[wv] = tc;
  __syncthreads();
  if (wv == 0) {
    float v = (lane < kWeightsThreads / 64) ? red[lane] : 0.0f;
    v = wave_sum(v);
    if (lane == 0) {
      scal[0] = beta;
      scal[1] = eta;
      scal[2] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// weighted reduction: Unew[t][j] = sum_m ( sum_{k=64m..64m+63} wn[k] * V[t][k][j] )
// Reference: weightedReductionKernel, mppi_controller.cu:219-267 -- block t, thread m walks 64
// rollouts in order with an fma per step, thread 0 adds the partials in order.  Same summation
// order here, so the result is bit-identical to the reference order given the same wn.
// One workgroup per timestep; the row V[t][*][*] (K*2 contiguous floats in the time-major
// buffer) is staged through LDS in coalesced 16-B loads, then each lane runs one (m, j) chain.
// ---------------------------------------------------------------------------------------------
constexpr int kRedThreads = 256;
constexpr int kRedChunk = 4096;  // rollouts staged per pass: 32 KiB of LDS (+ pad)

__global__ __launch_bounds__(kRedThreads) void weighted_reduction_kernel(
    const float *__restrict__ wn, const float *__restrict__ V, int K, float *__restrict__ Unew)
{
  // chunk rows padded by 2 floats per 64-rollout group: lanes of one wave (m varies) hit
  // distinct banks when they walk their chains in lock step.
  __shared__ __attribute__((aligned(16))) float tile[kRedChunk * 2 + (kRedChunk / 64) * 2];
  __shared__ float wtile[kRedChunk];
  extern __shared__ float partial[];  // [K/64][2]
  const int t = blockIdx.x, tid = threadIdx.x;
  const float *row = V + (size_t)t * K * 2;
  const int groups = K / 64;
  for (int base = 0; base < K; base += kRedChunk) {
    const int n = min(kRedChunk, K - base);  // multiple of 64
    // stage: n*2 floats, float4 per lane
    const float4 *src = reinterpret_cast<const float4 *>(row + (size_t)base * 2);
    for (int q = tid; q < n / 2; q += kRedThreads) {
      const float4 v = src[q];  // rollouts base+2q, base+2q+1
      const int kk = 2 * q;
      const int o = kk * 2 + (kk >> 6) * 2;
      tile[o + 0] = v.x; tile[o + 1] = v.y; tile[o + 2] = v.z; tile[o + 3] = v.w;
    }
    for (int q = tid; q < n; q += kRedThreads) wtile[q] = wn[base + q];
    __syncthreads();
    // chains: c = (m_local, j)
    for (int c = tid; c < (n / 64) * 2; c += kRedThreads) {
      const int ml = c >> 1, j = c & 1;
      const float *p = tile + ml * 130 + j;
      const float *wp = wtile + ml * 64;
      float acc = 0.0f;
#pragma unroll 8
      for (int i = 0; i < 64; i++) acc = fmaf(wp[i], p[2 * i], acc);  // u_system += weight*u :246
      partial[(base / 64 + ml) * 2 + j] = acc;
    }
    __syncthreads();
  }
  if (tid < 2) {
    float u = 0.0f;
    for (int m = 0; m < groups; m++) u += partial[m * 2 + tid];  // :256-260
    Unew[t * 2 + tid] = u;
  }
}

// ---------------------------------------------------------------------------------------------
// savitskyGolay, mppi_controller.cu:468-499. In place on U[T][2]; hist[4]; one workgroup.
// res[0..2] <- scal (beta, eta, traj cost) and res[4 ..] <- smoothed U so that ONE D2H copy
// returns everything the host needs.
// ---------------------------------------------------------------------------------------------
__global__ void savgol_kernel(float *__restrict__ U, const float *__restrict__ hist, int T,
                              const float *__restrict__ scal, float *__restrict__ res, int smooth)
{
  extern __shared__ float X[];  // [(T+4)][2]
  const int tid = threadIdx.x;
  for (int i = tid; i < (T + 4) * 2; i += blockDim.x) {
    const int r = i >> 1, j = i & 1;
    float v;
    if (r < 2) v = hist[2 * r + j];
    else if (r < T + 2) v = U[2 * (r - 2) + j];
    else v = U[2 * (T - 1) + j];
    X[i] = v;
  }
  __syncthreads();
  const float f0 = -3.0f / 35.0f, f1 = 12.0f / 35.0f, f2 = 17.0f / 35.0f;
  for (int i = tid; i < T * 2; i += blockDim.x) {
    float out;
    if (smooth) {
      float acc = f0 * X[i];
      float p = f1 * X[i + 2];
      acc = acc + p;
      p = f2 * X[i + 4];
      acc = acc + p;
      p = f1 * X[i + 6];
      acc = acc + p;
      p = f0 * X[i + 8];
      acc = acc + p;
      out = acc;
    } else {
      out = X[i + 4];
    }
    U[i] = out;
    res[4 + i] = out;
  }
  if (tid < 3) res[tid] = scal[tid];
}

// ---------------------------------------------------------------------------------------------
// layout transposes at the ABI boundary: reference layout [K][T][2] (index 2T*k + 2t + j,
// mppi_controller.cu:133) <-> internal time-major [T][K][2].
// ---------------------------------------------------------------------------------------------
__global__ void kt_to_tk_kernel(const float2 *__restrict__ src, float2 *__restrict__ dst, int K, int T)
{
  __shared__ float2 tile[32][33];
  const int k0 = blockIdx.x * 32, t0 = blockIdx.y * 32;
  const int tx = threadIdx.x, ty = threadIdx.y;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int k = k0 + r, t = t0 + tx;
    if (k < K && t < T) tile[r][tx] = src[(size_t)k * T + t];
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int t = t0 + r, k = k0 + tx;
    if (k < K && t < T) dst[(size_t)t * K + k] = tile[tx][r];
  }
}

__global__ void tk_to_kt_kernel(const float2 *__restrict__ src, float2 *__restrict__ dst, int K, int T)
{
  __shared__ float2 tile[32][33];
  const int k0 = blockIdx.x * 32, t0 = blockIdx.y * 32;
  const int tx = threadIdx.x, ty = threadIdx.y;
  for (int r = ty; r < 32; r += 8) {
    const int t = t0 + r, k = k0 + tx;
    if (k < K && t < T) tile[r][tx] = src[(size_t)t * K + k];
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int k = k0 + r, t = t0 + tx;
    if (k < K && t < T) dst[(size_t)k * T + t] = tile[tx][r];
  }
}

// ---- launchers ----
hipError_t launch_weights(const float *costs, int K, float gamma, float *w, float *wn, float *scal,
                          hipStream_t stream)
{
  hipLaunchKernelGGL(weights_kernel, dim3(1), dim3(kWeightsThreads), 0, stream, costs, K, gamma, w,
                     wn, scal);
  return hipGetLastError();
}

hipError_t launch_weighted_reduction(const float *wn, const float *V, int K, int T, float *Unew,
                                     hipStream_t stream)
{
  const size_t dyn = (size_t)(K / 64) * 2 * sizeof(float);
  hipLaunchKernelGGL(weighted_reduction_kernel, dim3(T), dim3(kRedThreads), dyn, stream, wn, V, K,
                     Unew);
  return hipGetLastError();
}

hipError_t launch_savgol(float *U, const float *hist, int T, const float *scal, float *res, int smooth,
                         hipStream_t stream)
{
  const size_t dyn = (size_t)(T + 4) * 2 * sizeof(float);
  hipLaunchKernelGGL(savgol_kernel, dim3(1), dim3(256), dyn, stream, U, hist, T, scal, res, smooth);
  return hipGetLastError();
}

hipError_t launch_kt_to_tk(const float *src, float *dst, int K, int T, hipStream_t stream)
{
  dim3 grid((K + 31) / 32, (T + 31) / 32), block(32, 8);
  hipLaunchKernelGGL(kt_to_tk_kernel, grid, block, 0, stream, reinterpret_cast<const float2 *>(src),
                     reinterpret_cast<float2 *>(dst), K, T);
  return hipGetLastError();
}

hipError_t launch_tk_to_kt(const float *src, float *dst, int K, int T, hipStream_t stream)
{
  dim3 grid((K + 31) / 32, (T + 31) / 32), block(32, 8);
  hipLaunchKernelGGL(tk_to_kt_kernel, grid, block, 0, stream, reinterpret_cast<const float2 *>(src),
                     reinterpret_cast<float2 *>(dst), K, T);
  return hipGetLastError();
}

}  // namespace mppi
