// solve_kernels.hip -- everything of computeControl (PI/mppi_controller.cu:600-675) that is not
// the rollout: baseline/normExp/normaliser, weightedReductionKernel, savitskyGolay, plus the
// [K][T][2] <-> [T][K][2] layout transposes used at the ABI boundary.
#include "mppi_device.hpp"
#include "mppi_kernels.hpp"


namespace mppi {


// ---------------------------------------------------------------------------------------------
// solve_tail_kernel: everything of one solve iteration after the rollout, in ONE launch.
//
//   beta = min_k J_k ; w_k = expf(-gamma (J_k - beta)) ; eta = sum_k w_k ; traj = sum_k w_k^2/eta
//       reference: host loops + normExpKernel, mppi_controller.cu:627-652, 193-203 (two D2H round
//       trips and two stream syncs there)
//   Unew[t][j] = sum_m ( sum_{k=64m..64m+63} (w_k/eta) * V[t][k][j] )
//       reference: weightedReductionKernel, mppi_controller.cu:219-267 -- block t, thread m walks 64
//       rollouts in order with an fma per step, thread 0 adds the partials in order.  Same order
//       here, so U is bit-identical to the reference order given the same weights.
//   U = SavitzkyGolay([hist | Unew | pad])   (last iteration only), mppi_controller.cu:468-499
//
// K <= 4096 (solve_tail_kernel; beyond: solve_tail_stream_kernel below):
// Grid = T workgroups (one per timestep) + one that publishes the weights and scalars.  Every workgroup recomputes beta and
// eta from the K costs (16 KB at K=4096, L2 resident; same code in every workgroup => the same bits), stages its row
// V[t][*][*] (contiguous bytes of the time-major buffer) through LDS with 16-B loads and runs the (m, j) chains.  On the last
// iteration every row workgroup writes
// its raw weighted mean straight into host-mapped memory (16-B entries carrying the solve's sequence
// number; the extra workgroup does the same for beta, eta and the trajectory cost): the host needs no D2H
// copy and no stream synchronise, it polls the T+2 entries and applies the 5-tap smoothing itself.
// The workgroup that finishes last (agent-scope arrival counter; the rows come to it as {value, seq} granules,
// MI355X guide G16) smooths the DEVICE copy of the sequence -- the one the next solve perturbs; same
// operations as the host, bit-identical -- and leaves its stride-slid copy for slideControlSeq; none of
// that is on the host's critical path, and inside chained control ticks (abi_solve.hip) none of it runs.
// Sums over k are pairwise (LDS tree) instead of the host's sequential loop: same value to ~1e-7.
// ---------------------------------------------------------------------------------------------
// Diagnostic build only (-DMPPI_TAIL_STAMPS, tools/tail_stamps.py): s_memrealtime stamps (100 MHz, comparable across CUs) of
// row workgroup T/2 -- where the time between the tail kernel's first instruction and the publication of a row goes.  The
// product build has no stamp instruction.
#ifdef MPPI_TAIL_STAMPS
__device__ unsigned long long g_tail_stamps[16];
#define TSTAMP(i)                                                                                   \
  do {                                                                                              \
    if (block == T / 2 && threadIdx.x == 0) {                                                       \
      unsigned long long t__;                                                                       \
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");               \
      g_tail_stamps[i] = t__;                                                                       \
    }                                                                                               \
  } while (0)
#else
#define TSTAMP(i) do { } while (0)
#endif
// Eight waves per workgroup: a lone wave issues a vector instruction every ~5.4 cycles whatever the SIMD could take
// (tools/ub/valu_issue_ub.hip), so the 4 096 exps and divisions of a workgroup run on two waves per SIMD in half the time of
// one -- a row is published 3.49 us after the first instruction instead of 3.98 (256 threads), 1 024 threads measure the same
// (profiles/r04_v_tail_threads_ab.txt)
#ifndef MPPI_TAIL_THREADS
#define MPPI_TAIL_THREADS 512
#endif
constexpr int kTailThreads = MPPI_TAIL_THREADS;
constexpr int kRedChunk = 4096;  // rollouts staged per pass: 32 KiB of LDS (+ pad)

// wave_reduce: mppi_device.hpp
__device__ __forceinline__ float wave_min(float v) { return wave_reduce<true>(v); }
__device__ __forceinline__ float wave_sum(float v) { return wave_reduce<false>(v); }
// Workgroup-wide (kTailThreads): ONE barrier -- every wave leaves its result in red4[wave], every thread combines the four
// in the same order.  red4 must not be reused by a later reduction of the same workgroup (no barrier behind the reads).
template <bool MIN>
__device__ __forceinline__ float block_reduce(float v, float *red4)
{
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  v = wave_reduce<MIN>(v);
  if (lane == 0) red4[wv] = v;
  __syncthreads();
  float r = red4[0];
#pragma unroll
  for (int i = 1; i < kTailThreads / 64; i++) r = MIN ? fminf(r, red4[i]) : r + red4[i];
  return r;
}

// One result entry (16 B) straight into host-mapped memory: [v0, seq, v1, seq], one uncached
// system-scope store.  Both 8-byte halves carry the solve's sequence number, so a reader that finds
// it in words 1 and 3 has both values even if the write reached memory as two 8-byte pieces in
// either order.
__device__ __forceinline__ void publish_entry(float *res, int entry, float v0, float v1, unsigned seq)
{
  const f32x4 v = {v0, __uint_as_float(seq), v1, __uint_as_float(seq)};
  float *p = res + 4 * (size_t)entry;
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

// One value handed to another workgroup of the same launch: an 8-byte {value, tag} granule, ONE sc1 store (MI355X guide G16,
// form R2: the data is the flag -- no fence, no flag word; the reader polls the granule with sc1 loads until the tag is this
// launch's).
__device__ __forceinline__ unsigned long long make_granule(unsigned epoch, float v)
{
  return ((unsigned long long)epoch << 32) | (unsigned long long)__float_as_uint(v);
}
__device__ __forceinline__ void store_granule(unsigned long long *g, unsigned epoch, float v)
{
  __hip_atomic_store(g, make_granule(epoch, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // global_store_dwordx2 sc1
}

struct TailArgs {
  const float *costs;   // [K]
  const float *V;       // [T][K][2] applied controls of this iteration
  float *U;             // [T][2] in/out: receives Unew, smoothed in place on the last iteration
  const float *hist;    // [4]
  float *w;             // [K] exp weights (for mppi_get_results)
  float *scal;          // [4] device scratch: beta, eta, trajectory cost (workgroup 0 -> last workgroup); streaming tail: [3] = 1 when
                        // beta came out of the rollout kernel
  float *res;           // host-mapped result block, T+2 entries of 16 B: rows [u0, seq, u1, seq], then
                        // [beta, seq, eta, seq] and [trajectory cost, seq, 0, seq]
  int no_device_copy;   // inside chained ticks: the rows and scalars are published, nothing else -- no arrival, no smoothing of the
                        // device copy, no slid copy (the next solve takes U from the host through its gate block)
  float *hist_out;      // optional: the smoothing workgroup copies hist[4] here (the last solve of a chain read it from the gate block)
  unsigned long long *ug;  // [T][2] the raw weighted mean of the last iteration as {value, seq} granules: from the row workgroups to the
                        // workgroup that arrives last and smooths the device copy
  unsigned *counter;    // [0]: the arrival counter of the row-closing workgroups (+ the extra one), zero on entry, reset by the last arriver
  int K, T;
  float gamma;
  float *slid;          // optional [2T + 4]: receives [U | hist] slid by slide_stride (or nullptr)
  int slide_stride;
  float init0, init1;
  int last_iter;        // smooth + publish results
  unsigned seq;         // sequence number published in res[3] once everything else is visible
  const unsigned long long *min_cost;  // solve_tail_stream_kernel: beta as the rollout kernel left it (mppi_device.hpp: load_min_cost), or nullptr
  unsigned min_cost_tag;
};

// The end of every tail kernel: the workgroups that closed a row (and, K <= 4096, the extra one) meet at the arrival counter;
// the last one smooths the DEVICE copy of the sequence and leaves its stride-slid copy.  The raw rows reach it as {value, seq}
// granules (G16, form R2); the counter only says WHO smooths (each workgroup drains its granule stores, s_waitcnt vmcnt(0),
// before one lane bumps it).  Inside chained control ticks (a.no_device_copy) nothing of this runs: the kernel ends with the
// publication.
__device__ __forceinline__ void tail_arrive_and_smooth(const TailArgs &a, const int K, const int T, const unsigned n_arrivers, int &is_last,
                                                       float *dyn)
{
  const int tid = threadIdx.x;
#ifdef MPPI_DIAG_TAIL_NOARRIVE  // diagnostic build: what do the arrival counter and the last workgroup's smoothing cost?
  return;
#endif
  if (a.no_device_copy) return;
  __syncthreads();  // is_last may still be read from the row hand-off above
  if (tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned ticket = __hip_atomic_fetch_add(a.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    is_last = (ticket == n_arrivers - 1u) ? 1 : 0;
  }
  __syncthreads();
  if (!is_last) return;
  if (tid == 0) __hip_atomic_store(a.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // next launch
  if (!a.last_iter) return;       // more iterations follow: U stays the raw weighted mean
  // The rows arrive as {value, seq} granules (round 5; until then: plain words behind an agent-scope acquire, 1.7 us of every
  // tail kernel -- nothing while the next launch came 4 us later anyway, the whole gap to the next rollout in the chained
  // ticks).  Every row workgroup stored its granules and drained them before its ticket, and this workgroup's ticket came
  // last: the tags are normally all there at the first look; a granule that is not yet visible is simply read again.
  float *X = dyn + (K / 64) * 2;  // [(T+4)][2]
  float *Y = X + (T + 4) * 2;     // [T][2] smoothed sequence
  {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = tid; i < (T + 4) * 2; i += kTailThreads) {
      const int r = i >> 1, j = i & 1;
      float v;
      if (r < 2) {
        v = a.hist[2 * r + j];
      } else {
        const unsigned long long *g = a.ug + 2 * (r < T + 2 ? r - 2 : T - 1) + j;
        for (;;) {
          const unsigned long long x = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          v = __uint_as_float((unsigned)x);
          if ((unsigned)(x >> 32) == a.seq) break;
          if (__builtin_amdgcn_s_memrealtime() - t0 > 1000000ull) { v = __builtin_nanf(""); break; }  // 10 ms: the next solve fails loudly
          __builtin_amdgcn_s_sleep(1);
        }
      }
      X[i] = v;
    }
  }
  __syncthreads();
  {
    const float f0 = -3.0f / 35.0f, f1 = 12.0f / 35.0f, f2 = 17.0f / 35.0f;
    for (int i = tid; i < T * 2; i += kTailThreads) {
      float acc = f0 * X[i];
      float p = f1 * X[i + 2];
      acc = acc + p;
      p = f2 * X[i + 4];
      acc = acc + p;
      p = f1 * X[i + 6];
      acc = acc + p;
      p = f0 * X[i + 8];
      acc = acc + p;
      a.U[i] = acc;  // the device copy the next solve perturbs (the host computes the same values itself)
      Y[i] = acc;    // and in LDS for the slid copy below
    }
    if (a.hist_out != nullptr && tid < 4) a.hist_out[tid] = X[tid];
  }
  __syncthreads();

  // Leave a copy of [U | hist]
  // slid by the controller's optimization stride (slideControlSeq, mppi_controller.cu:527-554) in
  // the other buffer, so that the control loop's slide -> solve costs no kernel and no upload.
  if (a.slid != nullptr) {
    const int st = a.slide_stride;
    // (i & 1 == tid & 1: the value is picked HERE, between two scalar registers -- written as `j ? a.init1 : a.init0` inside
    // the loop's conditional the compiler made it a load from a two-element private array, i.e. two scratch stores in every
    // workgroup's prologue and flat loads in this loop)
    const float init_j = (tid & 1) ? a.init1 : a.init0;
    // hist rows sit in X[0..3] (the smoothing's left padding)
    for (int i = tid; i < 2 * T; i += kTailThreads) {
      const int r = i >> 1, j = i & 1;
      a.slid[i] = (r < T - st) ? Y[(r + st) * 2 + j] : init_j;
    }
    if (tid < 4) {
      float hv;
      if (st == 1) hv = (tid < 2) ? X[tid + 2] : Y[tid - 2];
      else hv = Y[(st - 2) + tid];  // flat-index quirk (Q15)
      a.slid[2 * T + tid] = hv;
    }
  }
}

// body of solve_tail_kernel for workgroup `block` of the instance `a` (the batched kernel passes blockIdx.x minus
// the instance's first workgroup)
// One chunk per row (K <= kRedChunk = 4096 rollouts: the latency path): every workgroup computes beta and eta from all K costs itself
// (same code, same bits).  Rows of more chunks are solve_tail_stream_kernel's (until the end of round 5 a "wide" two-chunk form of
// this body served 4096 < K <= 8192, its two workgroups per row meeting at an arrival counter: K = 8192 0.0651 -> 0.0640 ms per step
// with the streaming kernel in its place, profiles/r05_aa_*).
// `V`, `costs`, `K`, `T`: the values of a.V, a.costs, a.K, a.T as the kernel received them -- in the single-instance kernels
// leading scalar parameters.  (Preloading them into scalar registers, -amdgpu-kernarg-preload-count, was measured: the
// workgroup's first loads go out 0.1 us earlier and the step is unchanged; the same for the row rollout kernel -- its first
// controls 0.28 us earlier, the STEP 0.3 us longer: the command processor reads the segment before it launches the first
// wave.  profiles/r04_v_kernarg_preload.txt.  Not used.)
__device__ __forceinline__ void solve_tail_body(const TailArgs &a, const int block, const float *V, const float *costs_p, const int K,
                                                const int T)
{
  // The chunk of the row in LDS, de-interleaved: plane j holds V[t][k][j]; a 64-rollout group takes kGS = 68 floats (64 + 4:
  // 16-B reads stay aligned and the groups of neighbouring lanes start in different banks), the normalised weights likewise --
  // a chain reads its 64 values and 64 weights as 2 x 16 ds_read_b128 (interleaved planes and 4-B reads: 128 LDS
  // instructions per chain, 0.74 us of the 5.6 a row workgroup took from its first instruction to the publication)
  constexpr int kGS = 68, kPlane = (kRedChunk / 64) * kGS;
  __shared__ __attribute__((aligned(16))) float tile[2][kPlane];
  __shared__ __attribute__((aligned(16))) float wtile[kPlane];
  __shared__ float redm[kTailThreads / 64], reds[kTailThreads / 64], redt[kTailThreads / 64];  // one per reduction (block_reduce)
  __shared__ int is_last;
  extern __shared__ __attribute__((aligned(16))) float dyn[];  // partial[2][K/64], then X[(T+4)*2] for the smoothing
  float *partial = dyn;
  const int tid = threadIdx.x;
  const bool extra = (block == T);  // publishes w[], beta, eta, trajectory cost
  const int t = extra ? 0 : block;
  constexpr int base = 0;
  const int n = K;  // rollouts of the row (multiple of 64, at most kRedChunk)
  TSTAMP(0);  // first instructions

  // The chunk of row t is requested NOW, before anything else, so that its HBM latency overlaps the
  // latency of the cost vector and the weight arithmetic below (the row was written by the rollout
  // kernel on other XCDs: it comes from HBM / Infinity Cache, not from this L2).
  const float *row = V + ((size_t)t * K + base) * 2;
  constexpr int kPre = kRedChunk / 2 / kTailThreads;  // float4 per thread in a full chunk
  float4 pre[kPre];
  {
    const float4 *src0 = reinterpret_cast<const float4 *>(row);
#pragma unroll
    for (int i = 0; i < kPre; i++) {
      const int q = tid + i * kTailThreads;
      pre[i] = (q < n / 2) ? src0[q] : make_float4(0, 0, 0, 0);
    }
  }
  // ---- weights: beta, eta (every workgroup), w[] and the trajectory cost (the extra workgroup) ----
  // Each exp of the workgroup's own chunk is evaluated once and kept in LDS.
  float eta;
  {
    // The K <= kRedChunk costs, 16 per thread, are requested with four 16-B loads that are in flight
    // together (and together with the row above): one memory round trip for the min and the exp pass.
    constexpr int kCostV = kRedChunk / 4 / kTailThreads;
    const float4 *c4 = reinterpret_cast<const float4 *>(costs_p);
    const int K4 = K / 4;
    float4 cv[kCostV];
#pragma unroll
    for (int i = 0; i < kCostV; i++) {
      const int q = i * kTailThreads + tid;
      cv[i] = (q < K4) ? c4[q] : make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
    }
    float m = INFINITY;
    TSTAMP(1);  // loads requested
#pragma unroll
    for (int i = 0; i < kCostV; i++) m = fminf(fminf(m, fminf(cv[i].x, cv[i].y)), fminf(cv[i].z, cv[i].w));
    TSTAMP(2);  // costs arrived, thread minimum
    const float beta = block_reduce<true>(m, redm);
    TSTAMP(3);  // beta
    // the exps stay in this thread's registers: they are normalised and staged once eta is known (the chunk's), and the
    // extra workgroup publishes all of them
    float part = 0.0f;
#pragma unroll
    for (int i = 0; i < kCostV; i++) {
      const int q = i * kTailThreads + tid;
      if (q < K4) {
        const float e0 = expf(-a.gamma * (cv[i].x - beta));  // normExpKernel :201
        const float e1 = expf(-a.gamma * (cv[i].y - beta));
        const float e2 = expf(-a.gamma * (cv[i].z - beta));
        const float e3 = expf(-a.gamma * (cv[i].w - beta));
        cv[i] = make_float4(e0, e1, e2, e3);
        part += (e0 + e1) + (e2 + e3);
      }
    }
    TSTAMP(4);  // exps
    eta = block_reduce<false>(part, reds);
    TSTAMP(5);  // eta
    if (extra) {
      float tc = 0.0f;
#pragma unroll
      for (int i = 0; i < kCostV; i++) {
        const int q = i * kTailThreads + tid;
        if (q < K4) {
          reinterpret_cast<float4 *>(a.w)[q] = cv[i];
          tc += cv[i].x * cv[i].x / eta;  // :651 (Q8)
          tc += cv[i].y * cv[i].y / eta;
          tc += cv[i].z * cv[i].z / eta;
          tc += cv[i].w * cv[i].w / eta;
        }
      }
      const float traj = block_reduce<false>(tc, redt);
      if (tid == 0) {
        __hip_atomic_store(&a.scal[0], beta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&a.scal[1], eta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&a.scal[2], traj, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a.last_iter) {
          publish_entry(a.res, T, beta, eta, a.seq);
          publish_entry(a.res, T + 1, traj, 0.0f, a.seq);
        }
      }
    } else {
      // weight = w/normalizer (:244) of this workgroup's chunk, straight from the registers into LDS
#pragma unroll
      for (int i = 0; i < kCostV; i++) {
        const int q = i * kTailThreads + tid;
        const int kl = 4 * q - base;  // k .. k+3 lie in one 64-rollout group
        if (q < K4 && kl >= 0 && kl < n)
          *reinterpret_cast<float4 *>(&wtile[(kl >> 6) * kGS + (kl & 63)]) =
              make_float4(cv[i].x / eta, cv[i].y / eta, cv[i].z / eta, cv[i].w / eta);
      }
    }
  }

  // ---- weighted reduction of the chunk ----
  const int G = K / 64;  // 64-rollout groups of a row
  if (!extra) {
#pragma unroll
    for (int i = 0; i < kPre; i++) {
      const int q = tid + i * kTailThreads;
      if (q < n / 2) {
        const int kk = 2 * q;  // rollouts base+2q, base+2q+1
        const int o = (kk >> 6) * kGS + (kk & 63);
        *reinterpret_cast<float2 *>(&tile[0][o]) = make_float2(pre[i].x, pre[i].z);
        *reinterpret_cast<float2 *>(&tile[1][o]) = make_float2(pre[i].y, pre[i].w);
      }
    }
    TSTAMP(6);  // row staged (its loads arrived)
    __syncthreads();
    TSTAMP(7);  // barrier
    for (int c = tid; c < (n / 64) * 2; c += kTailThreads) {
      const int ml = c >> 1, j = c & 1;
      const float4 *p4 = reinterpret_cast<const float4 *>(&tile[j][ml * kGS]);
      const float4 *w4 = reinterpret_cast<const float4 *>(&wtile[ml * kGS]);
      float acc = 0.0f;
#pragma unroll
      for (int i = 0; i < 16; i++) {  // u_system += weight*u :246, the 64 rollouts of the group in order
        const float4 wv = w4[i], pv = p4[i];
        acc = fmaf(wv.x, pv.x, acc);
        acc = fmaf(wv.y, pv.y, acc);
        acc = fmaf(wv.z, pv.z, acc);
        acc = fmaf(wv.w, pv.w, acc);
      }
      partial[j * G + ml] = acc;
    }
    TSTAMP(8);  // chains
    __syncthreads();
  }
  float u = 0.0f;
  if (tid < 2 && !extra) {
    // thread j adds the partials of control j in order (:256-260); fetched 32 at a time as 16-B reads where the plane is aligned
    const float *pj = partial + tid * G;
    int mm = 0;
    if ((G & 3) == 0) {
      for (; mm + 32 <= G; mm += 32) {
        float4 v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = *reinterpret_cast<const float4 *>(pj + mm + 4 * i);
#pragma unroll
        for (int i = 0; i < 8; i++) { u += v[i].x; u += v[i].y; u += v[i].z; u += v[i].w; }
      }
    }
    for (; mm < G; mm++) u += pj[mm];
    // last iteration: a granule for the workgroup that smooths the device copy; else the raw mean, for the next iteration's rollout
    if (a.last_iter) store_granule(a.ug + t * 2 + tid, a.seq, u);
    else a.U[t * 2 + tid] = u;
  }
  if (tid < 64) {
    // The host gets row t NOW (last iteration): it smooths the sequence itself (5 taps per value) as
    // soon as all T rows and the scalars carry this solve's sequence number, so nothing below -- the
    // arrival counter, the device-side smoothing for the next solve -- is on its critical path.
    const float u1 = __shfl(u, 1);  // all lanes of wave 0 active
    TSTAMP(9);  // partials added
    if (tid == 0 && !extra && a.last_iter) publish_entry(a.res, t, u, u1, a.seq);
    TSTAMP(10);  // row published (store issued)
  }
  // ---- arrival: the last workgroup smooths the device copy (tail_arrive_and_smooth) ----
  tail_arrive_and_smooth(a, K, T, (unsigned)(T + 1), is_last, dyn);  // T rows + the extra workgroup
}

// ---------------------------------------------------------------------------------------------
// solve_tail_stream_kernel (K > kRedChunk, round 5): the whole tail stage of a many-chunk solve in ONE launch.
//
// Before (round 4: weights_kernel + solve_tail_kernel<PRE>): beta, the exps, eta and the trajectory cost of all K rollouts in
// ONE 1 024-thread workgroup (8.7 us at K = 16 384, 23.6 us at 65 536), a kernel boundary, then T*C row workgroups whose
// chain results met at a per-row arrival counter (drain, barrier, atomic, agent acquire, re-read), two workgroups per CU:
// 21.5 us at K = 16 384 / T = 100, 28 us for config 4 -- a chain of latencies.  Now 14.5 / 17 us (profiles/r05_a_*).
//
// Grid: C weights workgroups (one per chunk of kRedChunk rollouts; they stream no row, so their polls do not queue behind
// their own loads), then the row workgroups (t, c) -- row t, chunk c -- dealt so that all chunks of a row run on ONE XCD.
//  * beta = min_k costs[k] comes out of the ROLLOUT kernel: its cost waves leave it behind as a tagged atomic minimum
//    (mppi_device.hpp: publish_min_cost) and every workgroup here reads the eight keys with its first loads -- no hand-over
//    for beta inside this launch (-0.6 us of the step at K = 16 384, -1.2 us at config 4; profiles/r05_s_*).  Where the keys are
//    not this launch's (a rollout form that does not publish, no finite cost, mppi_debug_min_cost off) the first version's way
//    runs: weights workgroup c takes the minimum of ALL costs where one load batch covers them (K <= 16 384), else the chunk
//    minima are exchanged among the C weights workgroups (exact, order-free), and {beta} goes out in kBcastReplicas replica
//    lines, of which a row workgroup polls ONE.
//  * Weights workgroup c: w_k = expf(-gamma (J_k - beta)) of its chunk; the chunk's sum goes into column c of EVERY one of the
//    kBcastReplicas sum replicas (32 stores of one wave), and a workgroup -- weights and row workgroups alike -- collects the C
//    columns of ONE replica and adds them in chunk order: eta is ONE hand-over away from the chunk sums (the first version
//    exchanged the sums among the weights workgroups and sent {eta} out in replica lines: two; -0.5 us of the step at K <= 16 384,
//    profiles/r05_x_*), and the fixed order makes it the same bits in every workgroup and every run (the reference's host loop
//    is sequential over k, :641-652 -- the pairwise / chunked order differs from it by ~1e-7 relative, as the tree of the old
//    weights pass did); w[] and the chunk's share of the trajectory cost sum w^2/eta (:651, Q8); workgroup 0 adds the shares
//    in chunk order and publishes beta, eta and the trajectory cost.
//  * Row workgroup (t, c) requests the keys, its chunk of the costs and its piece of V[t] at once, evaluates the exps of its
//    chunk while the weights workgroups reduce theirs, collects the C chunk sums of ONE sum replica (its block index picks it:
//    ~grid / kBcastReplicas pollers each) for eta, and keeps the chunk's weights w_k / eta (:244: a division per rollout, then
//    the fma) to itself.  The (m, j) chains of 64 rollouts and the in-order sum of their results (:246, :256-260) are the
//    reference's, untouched: the chain results of chunks 0 .. C-2 travel as granules, and the workgroup of the row's LAST chunk
//    -- started right behind the others -- polls them, adds all K/64 results in order and publishes the row.  The T
//    row-closing workgroups then meet at the arrival counter for the device-side smoothing, as in the one-chunk form.
// Every value that crosses workgroups is ONE 8-byte {value, epoch} granule (MI355X guide G16, form R2: the data is the flag;
// sc1 store, sc1 load poll, no fence, no counter to reset).  What the first versions of this kernel taught (stamps in
// profiles/r05_a_*): (1) with EVERY row workgroup publishing its column's granule and polling all C -- 400 stores and 400
// pollers on the same few lines -- an exchange took 6 us: same-line requests are served one after another, so the exchanges
// are among C workgroups and the fan-out goes through replica lines; (2) dealt by plain index, chunk c of every row ran on
// XCD c % 8, and with more workgroups than slots the XCDs drift apart: a row-closing workgroup was stamped starting 7 us
// BEFORE chunk 0 of its row and holding its slot for 9 us of polling; (3) the padded chunk image of solve_tail_body allows two
// workgroups per CU, the swizzled one three.
// Waits are for workgroups that start earlier on the same XCD (a row's earlier chunks) or first of all (the weights
// workgroups).  Every poll is bounded by a deadline on the 100 MHz real-time counter (poll_ticks); a workgroup that gives up
// stores NaN in place of what it waited for, NaN reaches eta or a published row, and the host reports MPPI_ERR_HIP -- never a
// hang, never finite wrong controls (mppi_debug_inject_handover_fault, roles 32-34).  The tag is a kernel argument: a captured
// graph would replay it -- this launch is not capturable as it stands.
// ---------------------------------------------------------------------------------------------
#ifdef MPPI_TAIL_STAMPS
__device__ unsigned long long g_stream_stamps[2][16];  // [0]: the row-closing workgroup of row T/2, [1]: workgroup (T/2, 0)
#define SSTAMP(i)                                                                                   \
  do {                                                                                              \
    if (t == T / 2 && (closer || c == 0) && threadIdx.x == 0) {                                     \
      unsigned long long t__;                                                                       \
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");               \
      g_stream_stamps[closer ? 0 : 1][i] = t__;                                                     \
    }                                                                                               \
  } while (0)
#else
#define SSTAMP(i) do { } while (0)
#endif
#ifndef MPPI_STREAM_SLEEP
#define MPPI_STREAM_SLEEP 1
#endif
constexpr int kMaxChunks = 64;  // one wave polls a column exchange: K <= 64 * kRedChunk = 262 144
constexpr int kBcastReplicas = 32;  // copies of what the weights workgroups hand to everybody: {beta} lines (128 B; only where the rollout
                                    // kernel left no beta) and the sum replicas (kMaxChunks granules each)
constexpr int kGxSumReplicas = 3 * kMaxChunks + kBcastReplicas * 16;  // [kBcastReplicas][kMaxChunks]: every chunk's sum in every replica
static_assert(kTailExchangeGranules == kGxSumReplicas + kBcastReplicas * kMaxChunks, "mppi_kernels.hpp");
static_assert(2 * kBcastReplicas <= kTailThreads, "one lane per replica granule");
struct StreamTailArgs {
  TailArgs a;                // a.counter[0] is the arrival counter of the T row-closing workgroups
  unsigned long long *gx;    // [3][kMaxChunks] exchange granules: chunk minima (no published beta), unused, chunk shares of the trajectory
                             // cost; then [kBcastReplicas][16]: {beta} lines (no published beta); then [kBcastReplicas][kMaxChunks]: the
                             // chunk sums, every chunk's in every replica
  unsigned long long *gpart; // [T][K/64][2] chain-result granules
  unsigned epoch;            // tag of this launch's granules (never 0, differs from every earlier launch on these buffers)
  unsigned poll_ticks;       // deadline of every wait, in ticks of s_memrealtime (100 MHz)
  int fault;                 // tests only: 32 = weights workgroup 0 never publishes its chunk sum, 33 = chunk 0 of every row never
                             // publishes its chain results, 34 = no weights workgroup publishes its chunk sum (nor {beta})
};


// One granule of a replica line, polled by wave 0 (lane 0 loads): its value into *out (LDS); the caller's barrier follows.
// NaN when the wait ran out of time.
__device__ __forceinline__ void poll_replica(const unsigned long long *g, const unsigned epoch, const unsigned long long t0,
                                             const unsigned poll_ticks, float *out)
{
  if (threadIdx.x < 64) {
    float v = 0.0f;
    bool ok = threadIdx.x != 0;
    for (;;) {
      if (!ok) {
        const unsigned long long x = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // sc1 load
        if ((unsigned)(x >> 32) == epoch) { v = __uint_as_float((unsigned)x); ok = true; }
      }
      if (__all(ok)) break;
      if (__builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long)poll_ticks) {
        if (!ok) v = __builtin_nanf("");
        break;
      }
      __builtin_amdgcn_s_sleep(MPPI_STREAM_SLEEP);
    }
    if (threadIdx.x == 0) *out = v;
  }
}

// Column exchange: column c's value `mine` (the same bits in every workgroup of the column) out, all C values in (xv[0 .. C-1],
// LDS).  Wave 0 polls, lane i column i; the caller's barrier follows.  A lane that runs out of time leaves NaN.
__device__ __forceinline__ void column_exchange(unsigned long long *g, const int C, const int c, const float mine, const unsigned epoch,
                                                const bool publish, const unsigned long long t0, const unsigned poll_ticks, float *xv)
{
  const int tid = threadIdx.x;
  if (tid < 64) {
    if (tid == 0 && publish) store_granule(g + c, epoch, mine);
    float v = mine;
    bool ok = (tid >= C) || (tid == c && publish);  // the own value needs no round trip
    for (;;) {
      if (!ok) {
        const unsigned long long x = __hip_atomic_load(g + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // sc1 load
        if ((unsigned)(x >> 32) == epoch) { v = __uint_as_float((unsigned)x); ok = true; }
      }
      if (__all(ok)) break;
      if (__builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long)poll_ticks) {
        if (!ok) v = __builtin_nanf("");
        break;
      }
      __builtin_amdgcn_s_sleep(MPPI_STREAM_SLEEP);
    }
    if (tid < C) xv[tid] = v;
  }
}

// LDS image of a chunk, 48 KB with no padding (three workgroups per CU; the padded image of solve_tail_body: two): plane j
// of the piece of V and the weights, a 64-rollout group = 16 slots of 16 B; slot s of group g sits at s ^ ((2g + j) & 15)
// (V) / s ^ (g & 7) (weights).  A wave's chain read i then takes, in each of ds_read_b128's 16-lane groups, 16 different
// slots (V: lane = 2 ml + j, and the lanes of a group differ in lane & 15) / 8 different slots read by two lanes each.
__device__ __forceinline__ int img_v(int g, int j, int e) { return g * 64 + ((((e >> 2) ^ (2 * g + j)) & 15) << 2) + (e & 3); }
__device__ __forceinline__ int img_w(int g, int e) { return g * 64 + ((((e >> 2) ^ (g & 7)) & 15) << 2) + (e & 3); }

// The weights workgroups of solve_tail_stream_kernel (blocks 0 .. C-1 of the grid, one per chunk; they stream no row, so
// their polls do not queue behind 48 KB of their own loads): beta, the chunk's exps, the chunk's sum into the sum replicas, eta,
// w[] and the trajectory cost.
__device__ __forceinline__ void stream_weights_body(const StreamTailArgs &sa, const float *costs_p, const int K, const int T, const int c,
                                                    float *redm, float *reds, float *redt, float *xmin, float *xsum, float *xtc)
{
  const TailArgs &a = sa.a;
  const int tid = threadIdx.x;
  const int C = (K + kRedChunk - 1) / kRedChunk;
  const int base = c * kRedChunk;
  const int n = min(kRedChunk, K - base);
  const unsigned epoch = sa.epoch;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long *const bcast = sa.gx + 3 * kMaxChunks;
  constexpr int kCostV = kRedChunk / 4 / kTailThreads;
  constexpr int kAllV = 8;  // float4 per thread that cover all costs of K <= 16 384
  const bool full_min = K <= kAllV * 4 * kTailThreads;
  float beta_pub;
  const bool have_beta = load_min_cost(a.min_cost, a.min_cost_tag, beta_pub);  // the same answer in every workgroup of the launch
  float4 cv[kCostV];
  {
    const float4 *c4 = reinterpret_cast<const float4 *>(costs_p + base);
#pragma unroll
    for (int i = 0; i < kCostV; i++) {
      const int q = i * kTailThreads + tid;
      cv[i] = (q < n / 4) ? c4[q] : make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
    }
  }
  // ---- beta: as the rollout kernel left it (the row workgroups read it there themselves: no hand-over); else the minimum of
  // ALL costs where one load batch covers them (no exchange), else the chunk minima exchanged ----
  float m = INFINITY, beta;
  if (have_beta) {
    beta = beta_pub;
  } else if (full_min) {
    float4 call[kAllV];
    const float4 *c4 = reinterpret_cast<const float4 *>(costs_p);
#pragma unroll
    for (int i = 0; i < kAllV; i++) {
      const int q = i * kTailThreads + tid;
      call[i] = (q < K / 4) ? c4[q] : make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
    }
#pragma unroll
    for (int i = 0; i < kAllV; i++) m = fminf(fminf(m, fminf(call[i].x, call[i].y)), fminf(call[i].z, call[i].w));
    beta = block_reduce<true>(m, redm);
  } else {
#pragma unroll
    for (int i = 0; i < kCostV; i++) m = fminf(fminf(m, fminf(cv[i].x, cv[i].y)), fminf(cv[i].z, cv[i].w));
    const float cmin = block_reduce<true>(m, redm);
    column_exchange(sa.gx, C, c, cmin, epoch, true, t0, sa.poll_ticks, xmin);
    __syncthreads();
    beta = xmin[0];
    for (int i = 1; i < C; i++) {
      const float x = xmin[i];
      beta = (x != x) ? x : fminf(beta, x);  // a wait that ran out of time left NaN: keep it
    }
  }
  // (no published beta) {beta} goes out in this workgroup's share of the replica lines at once: the row workgroups evaluate their
  // exps while the chunk sums are reduced
  if (!have_beta && tid < kBcastReplicas && (tid % C) == c && sa.fault != 34) store_granule(bcast + (size_t)tid * 16, epoch, beta);
  // ---- w_k, eta ----
  float part = 0.0f;
#pragma unroll
  for (int i = 0; i < kCostV; i++) {
    const int q = i * kTailThreads + tid;
    if (q < n / 4) {
      const float e0 = expf(-a.gamma * (cv[i].x - beta));  // normExpKernel :201
      const float e1 = expf(-a.gamma * (cv[i].y - beta));
      const float e2 = expf(-a.gamma * (cv[i].z - beta));
      const float e3 = expf(-a.gamma * (cv[i].w - beta));
      cv[i] = make_float4(e0, e1, e2, e3);
      part += (e0 + e1) + (e2 + e3);
    }
  }
  const float csum = block_reduce<false>(part, reds);
  // the chunk's sum into column c of EVERY replica (32 stores of one wave); a workgroup -- this one too -- collects the C
  // columns of ONE replica and adds them in chunk order: eta is one hand-over away from the chunk sums
  if (tid < kBcastReplicas && !(sa.fault == 32 && c == 0) && sa.fault != 34)
    store_granule(sa.gx + kGxSumReplicas + (size_t)tid * kMaxChunks + c, epoch, csum);
  column_exchange(sa.gx + kGxSumReplicas + (size_t)(c % kBcastReplicas) * kMaxChunks, C, -1, 0.0f, epoch, false, t0, sa.poll_ticks, xsum);
  __syncthreads();
  float eta = xsum[0];
  for (int i = 1; i < C; i++) eta += xsum[i];  // chunk order
  // ---- w[] and the chunk's share of the trajectory cost sum w^2/eta (:651, Q8); workgroup 0 adds the shares in chunk order ----
  float tc = 0.0f;
#pragma unroll
  for (int i = 0; i < kCostV; i++) {
    const int q = i * kTailThreads + tid;
    if (q < n / 4) {
      reinterpret_cast<float4 *>(a.w + base)[q] = cv[i];
      tc += (cv[i].x * cv[i].x / eta + cv[i].y * cv[i].y / eta) + (cv[i].z * cv[i].z / eta + cv[i].w * cv[i].w / eta);
    }
  }
  const float ctc = block_reduce<false>(tc, redt);
  if (c != 0) {
    if (tid == 0) store_granule(sa.gx + 2 * kMaxChunks + c, epoch, ctc);
    return;
  }
  column_exchange(sa.gx + 2 * kMaxChunks, C, 0, ctc, epoch, true, t0, sa.poll_ticks, xtc);
  __syncthreads();
  if (tid == 0) {
    float traj = xtc[0];
    for (int i = 1; i < C; i++) traj += xtc[i];  // chunk order
    a.scal[0] = beta; a.scal[1] = eta; a.scal[2] = traj;
    a.scal[3] = have_beta ? 1.0f : 0.0f;  // mppi_debug_min_cost
    if (a.last_iter) {
      publish_entry(a.res, T, beta, eta, a.seq);
      publish_entry(a.res, T + 1, traj, 0.0f, a.seq);
    }
  }
}

__global__ __launch_bounds__(kTailThreads, 6) void solve_tail_stream_kernel(const float *V, const float *costs_p, const int K, const int T,
                                                                            const StreamTailArgs sa)
{
  const TailArgs &a = sa.a;
  __shared__ __attribute__((aligned(16))) float img[3 * kRedChunk];  // V plane 0 | V plane 1 | weights; later partial | X | Y
  float *const tile0 = img, *const tile1 = img + kRedChunk, *const wtile = img + 2 * kRedChunk;
  __shared__ float redm[kTailThreads / 64], reds[kTailThreads / 64], redt[kTailThreads / 64];
  __shared__ float xmin[kMaxChunks], xsum[kMaxChunks], xtc[kMaxChunks];
  __shared__ float bc[2];
  __shared__ int is_last;
  const int tid = threadIdx.x;
  const int C = (K + kRedChunk - 1) / kRedChunk;
  const int block = (int)blockIdx.x;
  // Blocks 0 .. C-1: the weights workgroups.  From block Cpad = C rounded up to 8 on: the row workgroups, dealt so that ALL
  // chunks of a row run on ONE XCD (blocks b and b + 8 share an XCD and every XCD starts its blocks in index order): a row's
  // last chunk then starts right behind the row's other chunks.  Dealt by plain index, chunk c of every row ran on XCD c % 8;
  // with more workgroups than slots the XCDs drift apart, and a row-closing workgroup was stamped starting 7 us BEFORE chunk 0
  // of its row and holding its slot for 9 us of polling.
  if (block < C) {
    stream_weights_body(sa, costs_p, K, T, block, redm, reds, redt, xmin, xsum, xtc);
    return;
  }
  const int Cpad = (C + 7) & ~7;
  if (block < Cpad) return;
  const int bx = (block - Cpad) & 7, bi = (block - Cpad) >> 3;
  const int t = (bi / C) * 8 + bx, c = bi % C;
  if (t >= T) return;
  const int base = c * kRedChunk;
  const int n = min(kRedChunk, K - base);  // multiple of 64
  const int G = K / 64;
  const bool closer = (c == C - 1);        // adds the row's K/64 chain results and publishes the row
  const unsigned epoch = sa.epoch;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  SSTAMP(0);
  const unsigned long long *const bcast = sa.gx + 3 * kMaxChunks + (size_t)(block % kBcastReplicas) * 16;  // {beta, epoch}

  // beta as the rollout kernel left it (no hand-over inside this launch), the chunk of the costs, then the piece of row t:
  // requested together, before anything else; a wave's loads return in issue order -- what the weights wait for first
  float beta_pub;
  const bool have_beta = load_min_cost(a.min_cost, a.min_cost_tag, beta_pub);  // the same answer in every workgroup of the launch
  constexpr int kCostV = kRedChunk / 4 / kTailThreads;
  float4 cv[kCostV];
  {
    const float4 *c4 = reinterpret_cast<const float4 *>(costs_p + base);
#pragma unroll
    for (int i = 0; i < kCostV; i++) {
      const int q = i * kTailThreads + tid;
      cv[i] = (q < n / 4) ? c4[q] : make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
    }
  }
  const float *row = V + ((size_t)t * K + base) * 2;
  constexpr int kPre = kRedChunk / 2 / kTailThreads;
  float4 pre[kPre];
  {
    const float4 *src0 = reinterpret_cast<const float4 *>(row);
#pragma unroll
    for (int i = 0; i < kPre; i++) {
      const int q = tid + i * kTailThreads;
      pre[i] = (q < n / 2) ? src0[q] : make_float4(0, 0, 0, 0);
    }
  }
  SSTAMP(1);  // loads requested
  // ---- beta (where the rollout form left none: from the replica line this block's index picks, ~grid / kBcastReplicas pollers
  // per line), the chunk's exps, eta ----
  SSTAMP(2);
  float beta = beta_pub;
  if (!have_beta) {
    poll_replica(bcast, epoch, t0, sa.poll_ticks, &bc[0]);
    __syncthreads();
    beta = bc[0];
  }
  SSTAMP(3);
#pragma unroll
  for (int i = 0; i < kCostV; i++) {
    const int q = i * kTailThreads + tid;
    if (q < n / 4) {
      const float e0 = expf(-a.gamma * (cv[i].x - beta));  // normExpKernel :201
      const float e1 = expf(-a.gamma * (cv[i].y - beta));
      const float e2 = expf(-a.gamma * (cv[i].z - beta));
      const float e3 = expf(-a.gamma * (cv[i].w - beta));
      cv[i] = make_float4(e0, e1, e2, e3);
    }
  }
  SSTAMP(4);
  column_exchange(sa.gx + kGxSumReplicas + (size_t)(block % kBcastReplicas) * kMaxChunks, C, -1, 0.0f, epoch, false, t0, sa.poll_ticks, xsum);
  __syncthreads();
  float eta = xsum[0];
  for (int i = 1; i < C; i++) eta += xsum[i];  // chunk order: the bits of the weights workgroups' eta
  SSTAMP(5);
  // ---- weight = w/normalizer (:244), the piece of the row: into LDS ----
#pragma unroll
  for (int i = 0; i < kCostV; i++) {
    const int q = i * kTailThreads + tid;
    if (q < n / 4) {
      const int k = 4 * q;
      *reinterpret_cast<float4 *>(&wtile[img_w(k >> 6, k & 63)]) =
          make_float4(cv[i].x / eta, cv[i].y / eta, cv[i].z / eta, cv[i].w / eta);
    }
  }
#pragma unroll
  for (int i = 0; i < kPre; i++) {
    const int q = tid + i * kTailThreads;
    if (q < n / 2) {
      const int kk = 2 * q;  // rollouts base + kk, base + kk + 1
      *reinterpret_cast<float2 *>(&tile0[img_v(kk >> 6, 0, kk & 63)]) = make_float2(pre[i].x, pre[i].z);
      *reinterpret_cast<float2 *>(&tile1[img_v(kk >> 6, 1, kk & 63)]) = make_float2(pre[i].y, pre[i].w);
    }
  }
  SSTAMP(6);  // weights and row staged
  __syncthreads();
  SSTAMP(7);
  // ---- the (m, j) chains of this chunk: one per thread (at most 128 of them) ----
  float acc = 0.0f;
  const int ml = tid >> 1, j = tid & 1;
  const int mg = base / 64 + ml;
  const bool has_chain = tid < (n / 64) * 2;
  if (has_chain) {
    const float4 *p4 = reinterpret_cast<const float4 *>((j ? tile1 : tile0) + ml * 64);
    const float4 *w4 = reinterpret_cast<const float4 *>(wtile + ml * 64);
    const int sv = (2 * ml + j) & 15, sw = ml & 7;
#pragma unroll
    for (int i = 0; i < 16; i++) {  // u_system += weight*u :246, the 64 rollouts of the group in order
      const float4 wv = w4[i ^ sw], pv = p4[i ^ sv];
      acc = fmaf(wv.x, pv.x, acc);
      acc = fmaf(wv.y, pv.y, acc);
      acc = fmaf(wv.z, pv.z, acc);
      acc = fmaf(wv.w, pv.w, acc);
    }
    if (!closer && !(sa.fault == 33 && c == 0)) store_granule(sa.gpart + ((size_t)t * G + mg) * 2 + j, epoch, acc);
  }
  SSTAMP(8);  // chains
  if (!closer) return;  // a granule is data and flag in one store: nothing to drain, no counter to bump

  // ---- the row's last chunk: all K/64 chain results of the row into LDS (over the image: every chain has read it) ----
  float *partial = img;  // [2][G]
  __syncthreads();
  if (has_chain) partial[j * G + mg] = acc;
  {
    const int ng = (C - 1) * (kRedChunk / 64) * 2;
    const unsigned long long *gp = sa.gpart + (size_t)t * G * 2;
    for (int i0 = 0; i0 < ng; i0 += 4 * kTailThreads) {
      unsigned long long x[4];
      bool ok[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { ok[u] = (i0 + u * kTailThreads + tid) >= ng; x[u] = 0; }
      for (;;) {
        bool all = true;
#pragma unroll
        for (int u = 0; u < 4; u++)
          if (!ok[u]) x[u] = __hip_atomic_load(gp + i0 + u * kTailThreads + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int u = 0; u < 4; u++) {
          if (!ok[u]) ok[u] = ((unsigned)(x[u] >> 32) == epoch);
          all = all && ok[u];
        }
        if (__all(all)) break;
        if (__builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long)sa.poll_ticks) {
#pragma unroll
          for (int u = 0; u < 4; u++)
            if (!ok[u]) x[u] = (unsigned long long)__float_as_uint(__builtin_nanf(""));
          break;
        }
        __builtin_amdgcn_s_sleep(MPPI_STREAM_SLEEP);
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int i = i0 + u * kTailThreads + tid;
        if (i < ng) partial[(i & 1) * G + (i >> 1)] = __uint_as_float((unsigned)x[u]);
      }
    }
  }
  SSTAMP(9);  // the other chunks' chain results collected
  __syncthreads();
  float u = 0.0f;
  if (tid < 2) {
    // thread j adds the partials of control j in order (:256-260): K/64 dependent adds, the next 32 partials on their way
    // from LDS while these 32 are added
    const float *pj = partial + tid * G;
    int mm = 0;
    if ((G & 3) == 0 && G >= 32) {
      float4 v[8], w[8];
#pragma unroll
      for (int i = 0; i < 8; i++) v[i] = *reinterpret_cast<const float4 *>(pj + 4 * i);
      for (; mm + 32 <= G; mm += 32) {
        const int nx = (mm + 64 <= G) ? mm + 32 : mm;  // the last full batch re-reads itself
#pragma unroll
        for (int i = 0; i < 8; i++) w[i] = *reinterpret_cast<const float4 *>(pj + nx + 4 * i);
#pragma unroll
        for (int i = 0; i < 8; i++) { u += v[i].x; u += v[i].y; u += v[i].z; u += v[i].w; }
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = w[i];
      }
    }
    for (; mm < G; mm++) u += pj[mm];
    if (a.last_iter) store_granule(a.ug + t * 2 + tid, a.seq, u);  // for the row-closing workgroup that arrives last
    else a.U[t * 2 + tid] = u;
  }
  if (tid < 64) {
    const float u1 = __shfl(u, 1);
    if (tid == 0 && a.last_iter) publish_entry(a.res, t, u, u1, a.seq);
    SSTAMP(10);  // row published (store issued)
  }
  tail_arrive_and_smooth(a, K, T, (unsigned)T, is_last, img);  // its X | Y follow the row's K/64 x 2 partials in img
}

__global__ __launch_bounds__(kTailThreads) void solve_tail_kernel(const float *V, const float *costs, const int K, const int T,
                                                                  const TailArgs a)
{
  solve_tail_body(a, (int)blockIdx.x, V, costs, K, T);
}

// The tails of several instances (K <= kRedChunk each: T + 1 workgroups per instance) in one launch, behind
// rollout_quad_batch_kernel: grid (T + 1 of the longest instance, instances), workgroup (x, y) is row x of instance y.
template <int NB>
struct TailBatchArgs {
  TailArgs inst[NB];
};
template <int NB>
__global__ __launch_bounds__(kTailThreads) void solve_tail_batch_kernel(const TailBatchArgs<NB> b)
{
  // grid (T + 1 of the longest instance, instances): the instance from the workgroup's own index, its block at a compile-time
  // position of the argument segment (MPPI_BATCH_DISPATCH, mppi_device.hpp); NB = 2 for the two controllers of a tick
#define MPPI_TAIL_BODY(A)                                                   \
  do {                                                                      \
    if ((int)blockIdx.x > (A).T) return;                                    \
    solve_tail_body((A), (int)blockIdx.x, (A).V, (A).costs, (A).K, (A).T);    \
  } while (0)
  MPPI_BATCH_DISPATCH(NB, b, MPPI_TAIL_BODY);
#undef MPPI_TAIL_BODY
}

// slideControlSeq (mppi_controller.cu:527-554) on the device copy of [U(2T) | hist(4)], so that a
// solve -> slide -> solve loop never re-uploads the sequence.  One workgroup; mirrors the host code.
__global__ void slide_kernel(float *__restrict__ in, int T, int stride, float init0, float init1)
{
  extern __shared__ float buf[];  // old U
  float *U = in, *hist = in + 2 * T;
  for (int i = threadIdx.x; i < 2 * T; i += blockDim.x) buf[i] = U[i];
  float hold = 0.0f;
  if (threadIdx.x < 4) hold = hist[threadIdx.x];
  const float hold2 = __shfl(hold, (threadIdx.x + 2) & 63);  // all lanes active: lane i gets hist[i + 2]
  __syncthreads();
  if (threadIdx.x < 4) {
    const int i = threadIdx.x;
    float hv;
    if (stride == 1) hv = (i < 2) ? hold2 : buf[i - 2];
    else hv = buf[(stride - 2) + i];  // flat-index quirk (Q15)
    hist[i] = hv;
  }
  for (int i = threadIdx.x; i < 2 * T; i += blockDim.x) {
    const int r = i >> 1, j = i & 1;
    float v;
    if (r < T - stride) v = buf[(r + stride) * 2 + j];
    else v = j ? init1 : init0;
    U[i] = v;
  }
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

// debugCostKernel (PI/debug_kernels.cuh:39-88): raster of the costmap around (x, y) with a car marker,
// for MPPICosts::getDebugDisplay (costs.cu:272-285; the OpenCV display itself is out of scope).
__global__ __launch_bounds__(256) void debug_cost_kernel(CostArgs c, float x, float y, float heading, int width_m,
                                                        int height_m, int ppm, float *__restrict__ out)
{
  const int x_idx = blockIdx.x * 16 + (threadIdx.x & 15);
  const int y_idx = blockIdx.y * 16 + (threadIdx.x >> 4);
  const int W = width_m * ppm, H = height_m * ppm;
  // int / (1.0*ppm), -= width_m/2.0: single double operations on values exact in fp32 == the fp32 operations
  float x_pos = (float)x_idx / (float)ppm;
  float y_pos = (float)y_idx / (float)ppm;
  x_pos -= (float)width_m * 0.5f;
  y_pos -= (float)height_m * 0.5f;
  x_pos += x;
  y_pos += y;
  float cost = c.map[c.affine ? texel_index<true>(c, x_pos, y_pos) : texel_index<false>(c, x_pos, y_pos)];
  if (x_idx < W && (H - y_idx) < H) {  // i.e. y_idx > 0, as in the reference
    float sh, ch;
    sincos_fast(heading, sh, ch);
    const float dx = x_pos - x, dy = y_pos - y;
    const float xt = fmaf(ch, dx, sh * dy);
    const float yt = fmaf(-sh, dx, ch * dy);
    const float dist = (float)(0.25 * (double)fabsf(xt) + (double)fabsf(yt));
    if ((double)dist < .15 && xt > 0.0f) cost = ((double)dist < .1 && (double)xt > 0.05) ? 1.0f : 0.0f;
    const int idx = (H - (y_idx + 1)) * W + x_idx;
    if (idx > 0 && idx < W * H) out[idx] = cost;
  }
}

hipError_t launch_debug_cost(const CostArgs &c, float x, float y, float heading, int width_m, int height_m,
                             int ppm, float *out, hipStream_t stream)
{
  const dim3 grid((width_m * ppm - 1) / 16 + 1, (height_m * ppm - 1) / 16 + 1);
  hipLaunchKernelGGL(debug_cost_kernel, grid, dim3(256), 0, stream, c, x, y, heading, width_m, height_m, ppm, out);
  return hipGetLastError();
}

// ---- launchers ----
static size_t tail_dyn_bytes(int K, int T)
{
  return ((size_t)(K / 64) * 2 + (size_t)(T + 4) * 2 + (size_t)T * 2) * sizeof(float);
}

static TailArgs fill_tail(const TailLaunch &l)
{
  TailArgs a;
  a.slid = l.slid; a.slide_stride = l.slide_stride; a.init0 = l.init0; a.init1 = l.init1;
  a.costs = l.costs; a.V = l.V; a.U = l.U; a.hist = l.hist; a.w = l.w; a.scal = l.scal; a.res = l.res; a.ug = l.ug;
  a.no_device_copy = l.no_device_copy; a.hist_out = l.hist_out; a.min_cost = l.min_cost; a.min_cost_tag = l.min_cost_tag;
  a.counter = l.counter;
  a.K = l.K; a.T = l.T; a.gamma = l.gamma; a.last_iter = l.last_iter; a.seq = l.seq;
  return a;
}

bool tail_is_stream(int K) { return K > kRedChunk; }

hipError_t launch_solve_tail(const TailLaunch &l, hipStream_t stream)
{
  const TailArgs a = fill_tail(l);
  const int K = l.K, T = l.T;
  const int C = (K + kRedChunk - 1) / kRedChunk;
  if (tail_is_stream(K)) {
    if (C > kMaxChunks || !l.gx || !l.gpart || l.epoch == 0 || l.poll_ticks == 0) return hipErrorInvalidValue;
    StreamTailArgs sa;
    sa.a = a;
    sa.gx = l.gx; sa.gpart = l.gpart; sa.epoch = l.epoch; sa.poll_ticks = l.poll_ticks; sa.fault = l.fault;
    // the row's K/64 x 2 chain results and the smoothing's X | Y live in the 48 KB chunk image once the chains have read it
    if ((size_t)(K / 64) * 2 + (size_t)(T + 4) * 2 + (size_t)T * 2 > (size_t)3 * kRedChunk) return hipErrorInvalidValue;
    // C weights workgroups, padded to a multiple of 8, then 8 x ceil(T / 8) x C row workgroups (all chunks of a row on one XCD)
    const int grid = ((C + 7) & ~7) + 8 * ((T + 7) / 8) * C;
    hipLaunchKernelGGL(solve_tail_stream_kernel, dim3(grid), dim3(kTailThreads), 0, stream, a.V, a.costs, K, T, sa);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(solve_tail_kernel, dim3(T + 1), dim3(kTailThreads), tail_dyn_bytes(K, T), stream, a.V, a.costs, K, T, a);
  return hipGetLastError();
}

template <int NB>
static void launch_tail_batch_nb(const TailLaunch *l, int n, int tmax, size_t dyn, hipStream_t stream)
{
  TailBatchArgs<NB> b;
  for (int i = 0; i < NB; i++) b.inst[i] = fill_tail(l[i < n ? i : 0]);
  hipLaunchKernelGGL(solve_tail_batch_kernel<NB>, dim3(tmax + 1, n), dim3(kTailThreads), dyn, stream, b);
}
hipError_t launch_solve_tail_batch(const TailLaunch *l, int n, hipStream_t stream)
{
  if (n < 1 || n > kMaxBatch) return hipErrorInvalidValue;
  size_t dyn = 0;
  int tmax = 0;
  for (int i = 0; i < n; i++) {
    if (l[i].K > kRedChunk) return hipErrorInvalidValue;  // one workgroup per row only
    dyn = tail_dyn_bytes(l[i].K, l[i].T) > dyn ? tail_dyn_bytes(l[i].K, l[i].T) : dyn;
    tmax = l[i].T > tmax ? l[i].T : tmax;
  }
  if (n <= 2) launch_tail_batch_nb<2>(l, n, tmax, dyn, stream);
  else launch_tail_batch_nb<4>(l, n, tmax, dyn, stream);
  return hipGetLastError();
}

hipError_t launch_slide(float *in, int T, int stride, float init0, float init1, hipStream_t stream)
{
  hipLaunchKernelGGL(slide_kernel, dim3(1), dim3(256), (size_t)2 * T * sizeof(float), stream, in, T, stride,
                     init0, init1);
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

#ifdef MPPI_TAIL_STAMPS
extern "C" int mppi_debug_read_tail_stamps(unsigned long long *out)
{
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mppi::g_tail_stamps), sizeof(unsigned long long) * 16);
}
extern "C" int mppi_debug_read_stream_stamps(unsigned long long *out)  // [2][16]
{
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mppi::g_stream_stamps), sizeof(unsigned long long) * 32);
}
#endif
