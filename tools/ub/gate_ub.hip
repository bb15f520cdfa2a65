// gate_ub.hip -- VERDICT round 4, item 4: can launch + dispatch (and the prologue) of the NEXT solve's rollout kernel be taken off
// the step by enqueueing it ahead, gated on a word the host sets once it has seen this solve's result?
//
// Stand-ins with the headline's shape: a "rollout" of 256 workgroups x 512 threads that lasts ROLL us (a spin on the 100 MHz
// real-time counter), a "tail" of 101 workgroups that lasts TAIL us, publishes the step's sequence number to host-mapped
// memory ("rows published") and then spends POST us more ("the last arriver smooths the device copy") before it sets a device
// flag.  A step = the host sees the published number, does its 0.3 us of smoothing, starts the next step.  Forms:
//   0  today: launch rollout, launch tail, poll.
//   1  same stream, one step ahead: rollout n+1 (gated) and tail n+1 are enqueued before the host polls for n; the gate is a
//      word in HOST-MAPPED memory that workgroup 0 polls and relays to a device word; the other workgroups poll the relay.
//   2  two streams, one step ahead: the gated rollout n+1 becomes resident while tail n runs (rollouts on stream A, tails on
//      stream B, an event per step each way); it waits for the host's gate AND the tail's device flag ("U is ready").
//   3, 4  as 1, 2 with the gate word in DEVICE memory the host writes through the PCIe BAR (if the platform allows it).
// Every spin has a deadline (20 ms): a gate that never opens ends the kernel, it cannot hang the queue.
// hipcc --offload-arch=gfx950 -O3 tools/ub/gate_ub.hip -o gate_ub
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <csetjmp>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned long long now() { return __builtin_amdgcn_s_memrealtime(); }
__device__ __forceinline__ void spin_ticks(unsigned long long t)
{
  const unsigned long long t0 = now();
  while (now() - t0 < t) __builtin_amdgcn_s_sleep(2);
}
__device__ __forceinline__ unsigned ld_sys(const unsigned *p)  // host-mapped or peer-written word
{
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned ld_dev(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

constexpr unsigned long long kDeadline = 2000000ull;  // 20 ms

__global__ __launch_bounds__(512) void rollout_kernel(unsigned long long ticks, unsigned *sink)
{
  spin_ticks(ticks);
  if (sink && threadIdx.x == 0 && blockIdx.x == 0 && ticks == 12345) sink[0] = 1;
}

// gate: the word the host sets to `seq`; relay: device word workgroup 0 copies it to (nullptr: every workgroup polls `gate`
// itself -- the gate is device memory); ready / ready_seq: the tail's device flag (nullptr: none); ok[0] = 1 if a wait ran out
__global__ __launch_bounds__(512) void gated_rollout_kernel(unsigned long long ticks, const unsigned *gate, unsigned seq, unsigned *relay,
                                                            const unsigned *ready, unsigned ready_seq, unsigned *expired)
{
  __shared__ int go;
  if (threadIdx.x == 0) {
    const unsigned long long t0 = now();
    bool ok = false;
    if (relay && blockIdx.x == 0) {
      for (;;) {
        if (ld_sys(gate) == seq) { ok = true; break; }
        if (now() - t0 > kDeadline) break;
        __builtin_amdgcn_s_sleep(1);
      }
      __hip_atomic_store(relay, ok ? seq : 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      const unsigned *w = relay ? relay : gate;
      for (;;) {
        const unsigned v = relay ? ld_dev(w) : ld_sys(w);
        if (v == seq) { ok = true; break; }
        if (v == 0xFFFFFFFFu || now() - t0 > kDeadline) break;
        __builtin_amdgcn_s_sleep(1);
      }
    }
    if (ok && ready) {
      ok = false;
      for (;;) {
        if (ld_dev(ready) == ready_seq) { ok = true; break; }
        if (now() - t0 > kDeadline) break;
        __builtin_amdgcn_s_sleep(1);
      }
    }
    go = ok ? 1 : 0;
    if (!ok) expired[0] = 1;
  }
  __syncthreads();
  if (!go) return;
  spin_ticks(ticks);
}

__global__ __launch_bounds__(512) void tail_kernel(unsigned long long ticks, unsigned long long post, unsigned *publish, unsigned seq, unsigned *ready)
{
  spin_ticks(ticks);
  if (blockIdx.x == 0) {
    if (threadIdx.x == 0) __hip_atomic_store(publish, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    spin_ticks(post);
    if (threadIdx.x == 0) __hip_atomic_store(ready, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

static sigjmp_buf g_jmp;
static void on_segv(int) { siglongjmp(g_jmp, 1); }

int main(int argc, char **argv)
{
  const double roll_us = argc > 1 ? atof(argv[1]) : 34.0, tail_us = argc > 2 ? atof(argv[2]) : 3.5, post_us = argc > 3 ? atof(argv[3]) : 2.5;
  const int steps = argc > 4 ? atoi(argv[4]) : 3000;
  const unsigned long long RT = (unsigned long long)(roll_us * 100), TT = (unsigned long long)(tail_us * 100), PT = (unsigned long long)(post_us * 100);
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  unsigned *h_pub, *d_pub, *h_gate, *d_gate_map, *d_relay, *d_ready, *d_expired, *d_bar = nullptr;
  CK(hipHostMalloc(&h_pub, 64, hipHostMallocMapped));
  CK(hipHostMalloc(&h_gate, 64, hipHostMallocMapped));
  CK(hipHostGetDevicePointer((void **)&d_pub, h_pub, 0));
  CK(hipHostGetDevicePointer((void **)&d_gate_map, h_gate, 0));
  CK(hipMalloc(&d_relay, 256)); CK(hipMalloc(&d_ready, 256)); CK(hipMalloc(&d_expired, 256));
  CK(hipMemset(d_relay, 0, 256)); CK(hipMemset(d_ready, 0, 256)); CK(hipMemset(d_expired, 0, 256));
  memset(h_pub, 0, 64); memset(h_gate, 0, 64);
  // can the host store into device memory (fine-grained allocation, PCIe BAR)?
  bool bar = false;
  if (hipExtMallocWithFlags((void **)&d_bar, 4096, hipDeviceMallocFinegrained) == hipSuccess) {
    CK(hipMemset(d_bar, 0, 4096)); CK(hipDeviceSynchronize());
    struct sigaction sa_new, sa_old_segv, sa_old_bus;
    memset(&sa_new, 0, sizeof(sa_new)); sa_new.sa_handler = on_segv;
    sigaction(SIGSEGV, &sa_new, &sa_old_segv); sigaction(SIGBUS, &sa_new, &sa_old_bus);
    if (sigsetjmp(g_jmp, 1) == 0) {
      ((volatile unsigned *)d_bar)[1] = 77u;
      std::atomic_thread_fence(std::memory_order_seq_cst);
      unsigned back = 0;
      CK(hipMemcpy(&back, d_bar + 1, 4, hipMemcpyDeviceToHost));
      bar = back == 77u;
      printf("host store into fine-grained device memory: %s (read back %u)\n", bar ? "works" : "not visible", back);
    } else printf("host store into fine-grained device memory: faults\n");
    sigaction(SIGSEGV, &sa_old_segv, nullptr); sigaction(SIGBUS, &sa_old_bus, nullptr);
  } else printf("hipExtMallocWithFlags(hipDeviceMallocFinegrained) failed\n");
  (void)hipGetLastError();
  hipEvent_t ev_a[4], ev_b[4];
  for (int i = 0; i < 4; i++) { CK(hipEventCreateWithFlags(&ev_a[i], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ev_b[i], hipEventDisableTiming)); }
  const volatile unsigned *pub = h_pub;
  unsigned seq = 0;
  auto wait_pub = [&](unsigned s) {
    const auto t0 = std::chrono::steady_clock::now();
    unsigned long spins = 0;
    while (__atomic_load_n(pub, __ATOMIC_ACQUIRE) != s) {
      __builtin_ia32_pause();
      if ((++spins & 0xFFFF) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2.0) { printf("host wait timed out at seq %u\n", s); exit(2); }
    }
  };
  auto host_work = [&] {  // poll -> smoothed: 0.3 us
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < 0.3) { }
  };
  for (int form = 0; form <= 4; form++) {
    if (form >= 3 && !bar) { printf("form %d: skipped (no host-writable device memory)\n", form); continue; }
    volatile unsigned *gate_h = form >= 3 ? (volatile unsigned *)d_bar : (volatile unsigned *)h_gate;
    const unsigned *gate_d = form >= 3 ? d_bar : d_gate_map;
    unsigned *relay = form >= 3 ? nullptr : d_relay;
    const bool two = (form == 2 || form == 4);
    CK(hipDeviceSynchronize());
    auto enqueue_plain = [&](unsigned s) {
      hipLaunchKernelGGL(rollout_kernel, dim3(256), dim3(512), 0, sa, RT, (unsigned *)nullptr);
      hipLaunchKernelGGL(tail_kernel, dim3(101), dim3(512), 0, sa, TT, PT, d_pub, s, d_ready);
    };
    auto enqueue_gated = [&](unsigned s) {  // solve s, to start when the host has set the gate to s (and, two streams, tail s-1 has set ready)
      if (!two) {
        hipLaunchKernelGGL(gated_rollout_kernel, dim3(256), dim3(512), 0, sa, RT, gate_d, s, relay, (const unsigned *)nullptr, 0u, d_expired);
        hipLaunchKernelGGL(tail_kernel, dim3(101), dim3(512), 0, sa, TT, PT, d_pub, s, d_ready);
      } else {
        // rollouts on stream A (in order), tails on stream B (in order); tail s waits for rollout s; rollout s is resident
        // beside tail s-1 and waits for its ready flag
        hipLaunchKernelGGL(gated_rollout_kernel, dim3(256), dim3(512), 0, sa, RT, gate_d, s, relay, (const unsigned *)d_ready, s - 1, d_expired);
        CK(hipEventRecord(ev_a[s & 3], sa));
        CK(hipStreamWaitEvent(sb, ev_a[s & 3], 0));
        hipLaunchKernelGGL(tail_kernel, dim3(101), dim3(512), 0, sb, TT, PT, d_pub, s, d_ready);
      }
    };
    const int warm = 300;
    double el = 0.0;
    if (form == 0) {
      for (int i = 0; i < warm + steps; i++) {
        if (i == warm) el = -std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
        ++seq; enqueue_plain(seq); wait_pub(seq); host_work();
      }
    } else {
      // the first solve: gate opened at once; in the two-stream form its rollout waits for ready == seq - 1: set it
      ++seq;
      if (two) { CK(hipMemcpy(d_ready, &(const unsigned &)(seq - 1), 4, hipMemcpyHostToDevice)); }
      enqueue_gated(seq);
      *gate_h = seq;
      for (int i = 0; i < warm + steps; i++) {
        if (i == warm) el = -std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
        const unsigned cur = seq;
        const bool more = i + 1 < warm + steps;
        if (more) enqueue_gated(cur + 1);  // one step ahead, before the host looks at this step's result
        wait_pub(cur); host_work();
        if (more) { ++seq; *gate_h = seq; std::atomic_thread_fence(std::memory_order_seq_cst); }
      }
    }
    el += std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    CK(hipDeviceSynchronize());
    unsigned expired = 0;
    CK(hipMemcpy(&expired, d_expired, 4, hipMemcpyDeviceToHost));
    const char *names[] = {"0 today: launch, launch, poll", "1 same stream, one step ahead, host-mapped gate + relay", "2 two streams, resident beside the tail, host-mapped gate + relay",
                           "3 same stream, gate in device memory (host store through the BAR)", "4 two streams, gate in device memory (host store through the BAR)"};
    printf("form %-70s %.2f us per step (rollout %.1f + tail %.1f us of work; %d steps)%s\n", names[form], 1e6 * el / steps, roll_us, tail_us, steps, expired ? "  [a wait RAN OUT]" : "");
  }
  return 0;
}
