// valu_issue_ub.hip -- how fast does ONE wavefront (and two on a SIMD) issue the instructions the row form is made of?
// Every case is a loop body of 32 (or 48 / 64) vector instructions, repeated; cycles per instruction from clock64.
//   0: 32 independent v_pk_fma_f32 (8 accumulators)            1: 32 independent v_fma_f32
//   2: 32 v_mov_b32_dpp row_newbcast (independent)              3: 16 v_mov_b64_dpp row_newbcast
//   4: 32 x (v_mov_b32_dpp, independent v_pk_fma_f32)           5: 16 x (v_mov_b64_dpp, 2 independent v_pk_fma_f32)
//   6: one DEPENDENT chain of 32 v_pk_fma_f32                   7: two interleaved dependent chains of 16 v_pk_fma_f32
//   8: four interleaved dependent chains of 8 v_pk_fma_f32
//   9: 32 x (v_mov_b32_dpp, v_pk_fma_f32 of ONE chain) = the product's layer   10: 16 x (v_mov_b64_dpp, 2 v_pk_fma_f32 of two chains)
// (inline assembly throughout: the compiler merges, drops or reorders the plain form)
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ub/valu_issue_ub.hip -o valu_issue_ub
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int Q>
__device__ __forceinline__ float bc32(float a)
{
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(a), 0x150 + (Q & 15), 0xF, 0xF, false));
}
template <int Q>
__device__ __forceinline__ f32x2 bc64(f32x2 a)
{
  return __builtin_bit_cast(f32x2, (long long)__builtin_amdgcn_mov_dpp(__builtin_bit_cast(long long, a), 0x150 + (Q & 15), 0xF, 0xF, false));
}
#define SB __builtin_amdgcn_sched_barrier(0)

template <int CASE>
__global__ __launch_bounds__(1024) void ub(const float *src, float *out, long long *ticks, int iters)
{
  const int lane = threadIdx.x & 63;
  f32x2 w[8], acc[8], m[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    w[i] = f32x2{src[lane + i], src[lane + 8 + i]};
    acc[i] = f32x2{0.001f * i, 0.002f * i};
    m[i] = f32x2{src[lane + 16 + i], src[lane + 24 + i]};
  }
  float sacc[8];
#pragma unroll
  for (int i = 0; i < 8; i++) sacc[i] = acc[i].x;
  const long long c0 = clock64();
#define PKFMA(A, W, M) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(A) : "v"(W), "v"(M))
#define SFMA(A, W, M) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(A) : "v"(W), "v"(M))
#define MOV32(D, S, Q) asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:" #Q " row_mask:0xf bank_mask:0xf" : "=v"(D) : "v"(S))
#define MOV64(D, S, Q) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:" #Q " row_mask:0xf bank_mask:0xf" : "=v"(D) : "v"(S))
  for (int it = 0; it < iters; it++) {
    if constexpr (CASE == 0) {
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int i = 0; i < 8; i++) PKFMA(acc[i], w[i], m[i]);
    } else if constexpr (CASE == 1) {
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int i = 0; i < 8; i++) SFMA(sacc[i], w[i].x, m[i].x);
    } else if constexpr (CASE == 2) {
#pragma unroll
      for (int r = 0; r < 2; r++)
#pragma unroll
        for (int i = 0; i < 8; i++) { MOV32(m[i].x, w[i].x, 3); MOV32(m[i].y, w[i].y, 5); }
    } else if constexpr (CASE == 3) {
#pragma unroll
      for (int r = 0; r < 2; r++)
#pragma unroll
        for (int i = 0; i < 8; i++) MOV64(m[i], w[i], 3);
    } else if constexpr (CASE == 4) {
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int i = 0; i < 8; i++) { MOV32(m[i].x, w[i].x, 3); PKFMA(acc[(i + 3) & 7], w[(i + 3) & 7], m[(i + 3) & 7]); }
    } else if constexpr (CASE == 5) {
#pragma unroll
      for (int r = 0; r < 2; r++)
#pragma unroll
        for (int i = 0; i < 8; i++) { MOV64(m[i], w[i], 3); PKFMA(acc[(i + 3) & 7], w[(i + 3) & 7], m[(i + 3) & 7]); PKFMA(acc[(i + 5) & 7], w[(i + 5) & 7], m[(i + 5) & 7]); }
    } else if constexpr (CASE == 6) {
#pragma unroll
      for (int r = 0; r < 32; r++) PKFMA(acc[0], w[r & 7], m[r & 7]);
    } else if constexpr (CASE == 7) {
#pragma unroll
      for (int r = 0; r < 32; r++) PKFMA(acc[r & 1], w[r & 7], m[r & 7]);
    } else if constexpr (CASE == 8) {
#pragma unroll
      for (int r = 0; r < 32; r++) PKFMA(acc[r & 3], w[r & 7], m[r & 7]);
    } else if constexpr (CASE == 9) {  // the product's chain: mov for k+1, dependent pk_fma of k
#pragma unroll
      for (int r = 0; r < 32; r++) { MOV32(m[(r + 1) & 7].x, w[(r + 1) & 7].x, 3); PKFMA(acc[0], w[r & 7], m[r & 7]); }
    } else {  // two chains, one 64-bit move per pair
#pragma unroll
      for (int r = 0; r < 16; r++) { MOV64(m[(r + 1) & 7], w[(r + 1) & 7], 3); PKFMA(acc[0], w[r & 7], m[r & 7]); PKFMA(acc[1], w[(r + 4) & 7], m[r & 7]); }
    }
  }
  const long long c1 = clock64();
  if (lane == 0) {
    ticks[threadIdx.x >> 6] = c1 - c0;
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    ticks[16 + (threadIdx.x >> 6)] = (hw >> 4) & 3;  // SIMD_ID
  }
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 8; i++) s += acc[i].x + acc[i].y + m[i].x + m[i].y + sacc[i];
  out[threadIdx.x] = s;
}

template <int CASE>
static void run(const float *dsrc, float *dout, long long *dt, int ninst, const char *what)
{
  const int iters = 2000;
  for (int waves = 4; waves <= 16; waves *= 2) {  // one wave per SIMD / two / four
    for (int rep = 0; rep < 3; rep++) {
      hipLaunchKernelGGL(ub<CASE>, dim3(1), dim3(64 * waves), 0, 0, dsrc, dout, dt, iters);
      hipDeviceSynchronize();
    }
    long long ht[32];
    hipMemcpy(ht, dt, sizeof(ht), hipMemcpyDeviceToHost);
    long long lo = ht[0], hi = ht[0];
    int per_simd[4] = {0, 0, 0, 0};
    for (int i = 0; i < waves; i++) { lo = ht[i] < lo ? ht[i] : lo; hi = ht[i] > hi ? ht[i] : hi; per_simd[ht[16 + i] & 3]++; }
    printf("case %d (%s), waves on SIMD 0..3 = %d %d %d %d: %.2f .. %.2f cycles per instruction and wave (%d per iteration, %.1f cycles)\n", CASE, what,
           per_simd[0], per_simd[1], per_simd[2], per_simd[3], (double)lo / iters / ninst, (double)hi / iters / ninst, ninst, (double)hi / iters);
  }
}

int main()
{
  float hs[128];
  for (int i = 0; i < 128; i++) hs[i] = 0.001f * (float)(i + 1);
  float *dsrc, *dout; long long *dt;
  hipMalloc(&dsrc, sizeof(hs)); hipMalloc(&dout, 1024 * 4); hipMalloc(&dt, 32 * 8);
  hipMemcpy(dsrc, hs, sizeof(hs), hipMemcpyHostToDevice);
  run<0>(dsrc, dout, dt, 32, "32 independent v_pk_fma_f32");
  run<1>(dsrc, dout, dt, 32, "32 independent v_fma_f32");
  run<2>(dsrc, dout, dt, 32, "32 v_mov_b32_dpp");
  run<3>(dsrc, dout, dt, 16, "16 v_mov_b64_dpp");
  run<4>(dsrc, dout, dt, 64, "32 x (v_mov_b32_dpp, v_pk_fma_f32)");
  run<5>(dsrc, dout, dt, 48, "16 x (v_mov_b64_dpp, 2 v_pk_fma_f32)");
  run<6>(dsrc, dout, dt, 32, "one dependent chain of 32 v_pk_fma_f32");
  run<7>(dsrc, dout, dt, 32, "two interleaved dependent chains");
  run<8>(dsrc, dout, dt, 32, "four interleaved dependent chains");
  run<9>(dsrc, dout, dt, 64, "32 x (v_mov_b32_dpp, DEPENDENT v_pk_fma_f32): the product's layer");
  run<10>(dsrc, dout, dt, 48, "16 x (v_mov_b64_dpp, 2 v_pk_fma_f32 of two chains)");
  return 0;
}
