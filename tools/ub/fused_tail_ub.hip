// fused_tail_ub.hip -- what would ONE launch per solve cost at K <= 4096?  
// The fused form has to replace the kernel boundary between the rollout kernel and the tail kernel by an in-launch
// hand-over: all G = K/16 rollout workgroups (one per CU) store their costs, arrive at a device-scope counter, and
// every workgroup then loads ALL K costs (beta = min, eta = sum of exps need them) before its row of the weighted
// reduction.  This benchmark measures exactly that seam, both ways, on the same synthetic phases:
//   phase A: every workgroup spins ~SPIN_US (stand-in for the T-step recurrence), then 16 lanes store 16 "costs";
//   phase B: every workgroup loads the K costs (four 16-B loads per thread in flight), min + sum (block reductions),
//            one result per workgroup -- the front of solve_tail_kernel.
//   V0  two kernels (A, then B) on one stream: the seam is a kernel boundary (what the product does)
//   V1  one kernel, ONE agent-scope counter: sc1 stores -> s_waitcnt vmcnt(0) -> barrier -> lane 0 atomic add ->
//       sc1-load poll with s_sleep -> barrier -> sc1 loads of the costs (MI355X guide, hand-off table row 3)
//   V2  one kernel, counter sharded per XCD (s_getreg XCC_ID): 32 arrivals per shard, the last of a shard adds to
//       the top counter, everybody polls the top counter
// Reported: host-timed microseconds per iteration (hipEvents around N back-to-back iterations) for each variant and
// V1 - V0, V2 - V0: what the in-launch seam costs MORE than the boundary it replaces.  Results are checked.
//   hipcc --offload-arch=gfx950 -O3 tools/ub/fused_tail_ub.hip -o fused_tail_ub && ./fused_tail_ub
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

constexpr int G = 256, K = 4096, THREADS = 256;

__device__ __forceinline__ void spin_us(float us)
{
  const unsigned long long n = (unsigned long long)(us * 100.0f);  // wall_clock64: constant 100 MHz counter
  const unsigned long long w0 = wall_clock64();
  while (wall_clock64() - w0 < n) __builtin_amdgcn_s_sleep(4);
}

__device__ __forceinline__ void phase_a(float *costs, int iter, float us)
{
  spin_us(us);
  if (threadIdx.x < 16) {
    const int k = blockIdx.x * 16 + threadIdx.x;
    __hip_atomic_store(&costs[k], 1000.0f + (float)((k * 2654435761u + iter * 40503u) % 8191u) * 0.125f, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);  // sc1, write-through
  }
}

__device__ __forceinline__ float wave_min(float v)
{
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_sum(float v)
{
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

template <bool SC1>
__device__ __forceinline__ void phase_b(const float *costs, float *result)
{
  __shared__ float red[4];
  __shared__ float bc;
  const int tid = threadIdx.x;
  float4 cv[4];
  const float4 *c4 = reinterpret_cast<const float4 *>(costs);
  if (SC1) {  // four 16-B sc1 loads in flight, one wait: a single asm block, so that no use can slip in front of the wait
    const float4 *p0 = c4 + tid, *p1 = p0 + THREADS, *p2 = p1 + THREADS, *p3 = p2 + THREADS;
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\t"
                 "global_load_dwordx4 %2, %6, off sc1\n\tglobal_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(cv[0]), "=&v"(cv[1]), "=&v"(cv[2]), "=&v"(cv[3])
                 : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
                 : "memory");
  } else {
#pragma unroll
    for (int i = 0; i < 4; i++) cv[i] = c4[i * THREADS + tid];
  }
  float m = INFINITY;
#pragma unroll
  for (int i = 0; i < 4; i++) m = fminf(fminf(m, fminf(cv[i].x, cv[i].y)), fminf(cv[i].z, cv[i].w));
  m = wave_min(m);
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  if (tid == 0) bc = fminf(fminf(red[0], red[1]), fminf(red[2], red[3]));
  __syncthreads();
  const float beta = bc;
  __syncthreads();
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 4; i++)
    s += (expf(-0.15f * (cv[i].x - beta)) + expf(-0.15f * (cv[i].y - beta))) + (expf(-0.15f * (cv[i].z - beta)) + expf(-0.15f * (cv[i].w - beta)));
  s = wave_sum(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) result[blockIdx.x] = beta + ((red[0] + red[1]) + (red[2] + red[3]));
}

__global__ __launch_bounds__(THREADS) void k_a(float *costs, int iter, float us) { phase_a(costs, iter, us); }
__global__ __launch_bounds__(THREADS) void k_b(const float *costs, float *result) { phase_b<false>(costs, result); }

// V1: one counter.  epoch = iter + 1: the counter is monotonic, never reset.
__global__ __launch_bounds__(THREADS) void k_fused1(float *costs, float *result, unsigned *counter, int iter, float us, int *timeout)
{
  phase_a(costs, iter, us);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned want = (unsigned)G * (unsigned)(iter + 1);
    int budget = 1 << 22;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want && --budget > 0) __builtin_amdgcn_s_sleep(1);
    if (budget <= 0) *timeout = 1;
  }
  __syncthreads();
  phase_b<true>(costs, result);
}

// V2: per-XCD shards + top counter
__global__ __launch_bounds__(THREADS) void k_fused2(float *costs, float *result, unsigned *shards /*[8][32] words apart*/, unsigned *top,
                                                    unsigned *census /*[8]*/, int iter, float us, int *timeout)
{
  phase_a(costs, iter, us);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    xcc &= 7u;
    const unsigned per = census[xcc];  // workgroups of this grid on this XCD (measured by a census launch)
    const unsigned t = __hip_atomic_fetch_add(&shards[xcc * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == per * (unsigned)(iter + 1) - 1u) __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned want = 8u * (unsigned)(iter + 1);
    int budget = 1 << 22;
    while (__hip_atomic_load(top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want && --budget > 0) __builtin_amdgcn_s_sleep(1);
    if (budget <= 0) *timeout = 1;
  }
  __syncthreads();
  phase_b<true>(costs, result);
}

__global__ void k_census(unsigned *census)
{
  if (threadIdx.x == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    atomicAdd(&census[xcc & 7u], 1u);
  }
}

#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e__)); return 1; } } while (0)

int main()
{
  float *costs, *result; unsigned *ctr; int *timeout;
  CK(hipMalloc(&costs, K * 4)); CK(hipMalloc(&result, G * 4)); CK(hipMalloc(&ctr, 4096)); CK(hipMalloc(&timeout, 4));
  CK(hipMemset(timeout, 0, 4));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  unsigned census_h[8];
  CK(hipMemset(ctr, 0, 4096));
  unsigned *census = ctr + 512;
  hipLaunchKernelGGL(k_census, dim3(G), dim3(THREADS), 0, s, census);
  CK(hipStreamSynchronize(s));
  CK(hipMemcpy(census_h, census, 32, hipMemcpyDeviceToHost));
  printf("census (workgroups of a %d-workgroup grid per XCD):", G);
  bool even = true;
  for (int i = 0; i < 8; i++) { printf(" %u", census_h[i]); even = even && census_h[i] > 0; }
  printf("\n");
  const int N = 300;
  std::vector<float> ref(G), got(G);
  for (float us : {5.0f, 20.0f, 60.0f}) {
    double t[3] = {0, 0, 0};
    for (int v = 0; v < 3; v++) {
      if (v == 2 && !even) { t[v] = -1; continue; }
      CK(hipMemsetAsync(ctr, 0, 2048, s));
      std::vector<double> reps;
      for (int rep = 0; rep < 5; rep++) {
        // monotonic counters: iteration numbers continue over the repeats
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < N; i++) {
          const int it = rep * N + i;
          if (v == 0) { hipLaunchKernelGGL(k_a, dim3(G), dim3(THREADS), 0, s, costs, it, us); hipLaunchKernelGGL(k_b, dim3(G), dim3(THREADS), 0, s, costs, result); }
          else if (v == 1) hipLaunchKernelGGL(k_fused1, dim3(G), dim3(THREADS), 0, s, costs, result, ctr, it, us, timeout);
          else hipLaunchKernelGGL(k_fused2, dim3(G), dim3(THREADS), 0, s, costs, result, ctr + 16, ctr + 400, census, it, us, timeout);
        }
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        reps.push_back(1e3 * ms / N);
      }
      std::sort(reps.begin(), reps.end());
      t[v] = reps[reps.size() / 2];
      CK(hipMemcpy((v == 0 ? ref : got).data(), result, G * 4, hipMemcpyDeviceToHost));
      if (v > 0)
        for (int i = 0; i < G; i++)
          if (got[i] != ref[i] || got[i] != got[0]) { printf("V%d: result %d differs (%g vs %g)\n", v, i, got[i], ref[i]); return 1; }
    }
    int to = 0; CK(hipMemcpy(&to, timeout, 4, hipMemcpyDeviceToHost));
    printf("phase A %.0f us: V0 two kernels %.2f us/iter | V1 one counter %.2f (%+.2f) | V2 XCD shards %.2f (%+.2f)%s\n", us, t[0], t[1],
           t[1] - t[0], t[2], t[2] - t[0], to ? "  TIMEOUT" : "");
  }
  return 0;
}
