// launch_api_ub.hip -- host time of one kernel launch call (the call returns; the GPU runs an empty kernel), for the ways HIP
// offers to launch: hipLaunchKernelGGL (what the library uses), hipModuleLaunchKernel through hipGetFuncBySymbol with the
// argument block passed as one buffer, and both with a 272-byte / 24-byte argument block.
// hipcc --offload-arch=gfx950 -O3 tools/ub/launch_api_ub.hip -o launch_api_ub
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { float v[68]; };
struct Small { float *p; int a, b; float c, d; };
__global__ void kbig(const Big b, float *out) { if (b.v[0] == 12345.0f) out[0] = b.v[1]; }
__global__ void ksmall(const Small s) { if (s.c == 12345.0f) s.p[0] = s.d; }
template <class F>
static double time_calls(F f, int n, hipStream_t st)
{
  for (int i = 0; i < 200; i++) { f(); }
  hipStreamSynchronize(st);
  double tot = 0.0;
  for (int i = 0; i < n; i++) {
    const auto t0 = std::chrono::steady_clock::now();
    f();
    tot += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    hipStreamSynchronize(st);  // every call onto an idle queue, as in the solve loop
  }
  return tot / n;
}
int main()
{
  hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  float *d; hipMalloc(&d, 64);
  Big b{}; Small s{d, 1, 2, 0.0f, 1.0f};
  hipFunction_t fb = nullptr, fs = nullptr;
  const hipError_t e1 = hipGetFuncBySymbol(&fb, (const void *)kbig), e2 = hipGetFuncBySymbol(&fs, (const void *)ksmall);
  printf("hipGetFuncBySymbol: %d %d\n", (int)e1, (int)e2);
  struct { Big b; float *out; } abig{b, d};
  size_t sz_big = sizeof(abig), sz_small = sizeof(s);
  void *cfg_big[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &abig, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz_big, HIP_LAUNCH_PARAM_END};
  void *cfg_small[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &s, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz_small, HIP_LAUNCH_PARAM_END};
  const int n = 3000;
  printf("hipLaunchKernelGGL, 280-B arguments:      %.2f us per call\n", time_calls([&] { hipLaunchKernelGGL(kbig, dim3(256), dim3(512), 0, st, b, d); }, n, st));
  printf("hipLaunchKernelGGL, 24-B arguments:       %.2f us per call\n", time_calls([&] { hipLaunchKernelGGL(ksmall, dim3(256), dim3(512), 0, st, s); }, n, st));
  if (e1 == hipSuccess) {
    printf("hipModuleLaunchKernel, 280-B arguments:   %.2f us per call\n", time_calls([&] { hipModuleLaunchKernel(fb, 256, 1, 1, 512, 1, 1, 0, st, nullptr, cfg_big); }, n, st));
    printf("hipModuleLaunchKernel, 24-B arguments:    %.2f us per call\n", time_calls([&] { hipModuleLaunchKernel(fs, 256, 1, 1, 512, 1, 1, 0, st, nullptr, cfg_small); }, n, st));
  }
  // two launches back to back (rollout + tail), the second while the first may still run
  printf("two hipLaunchKernelGGL back to back:      %.2f us per pair\n", time_calls([&] { hipLaunchKernelGGL(kbig, dim3(256), dim3(512), 0, st, b, d); hipLaunchKernelGGL(ksmall, dim3(101), dim3(512), 0, st, s); }, n, st));
  return 0;
}
