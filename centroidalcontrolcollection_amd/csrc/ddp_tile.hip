// ddp_tile.hip -- the TILE build of the DDP planners (csrc/ddp_tile.h): CCC::DdpCentroidal / CCC::DdpSingleRigidBody with
// either regularisation, one instance per wavefront (64-thread workgroup), every matrix distributed over the 64 lanes, the
// backward step in structured form (no M x M object; DESIGN.md sections 7.2-7.4).  Register budget: no spills before
// occupancy -- two wavefronts per SIMD at 16 and 32 ridges per step, one at 64 (CCC_TILE_WAVES* below; the compiler's
// register / scratch / LDS figures of every instantiation are pinned in tests/test_kernel_resources.py); LDS 7.4-11.2 KB
// per wavefront whatever the ridge stride.  One resident set of workgroups pulls instances from a work queue, longest
// remaining first, in bit-identical slices of iterations (csrc/ddp_batch.h DdpSched).
// What ccc_ddp_plan_batch_device runs by default (csrc/ddp.hip); its arithmetic is the tile specification of
// oracle/ddp_tile.c, reproduced bit for bit (tests/test_ddp_gpu.py, tests/test_ddp_tile_emu.py).
// Replaces the same reference code as csrc/ddp.hip.
#include "ddp_tile.h"

namespace ccc_amd
{
// doubles of workspace per instance: trajectories [kSlots][N+1][S] and [kSlots][N][M], gains [N][M] and [N][M][S]
size_t ddp_tile_ws_doubles(int N, int S, int M)
{
  return (size_t)ddp_tile::kSlots * ((size_t)(N + 1) * S + (size_t)N * M) + (size_t)N * M + (size_t)N * M * S;
}

// wavefronts per SIMD the register budget is set for (measured, round 4: the kernel is bound by instruction issue, not by
// latency -- at 16 ridges two wavefronts per SIMD without spills (256 VGPRs) beat three (168, 240-490 B of scratch) and
// four (128, 430-660 B): config 3 50.9 k / 50.4 k / 46.6 k solves/s, config 5 shape 152 k / 139 k / 128 k)
#ifndef CCC_TILE_WAVES
#  define CCC_TILE_WAVES 2
#endif
#ifndef CCC_TILE_WAVES2
#  define CCC_TILE_WAVES2 2
#endif
#ifndef CCC_TILE_WAVES4
#  define CCC_TILE_WAVES4 1
#endif
__global__ void ddp_sched_reset_kernel(int * counters, size_t words, int * slot, size_t fill)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if(i < words)
  {
    if(i < 2 || i >= 8) counters[i] = 0; // (words 2 .. 7 carry the verdict on the history from launch to launch: DdpSched::trust)
  }
  else if(i - words < fill)
    slot[i - words] = -1;
}

// device-scope loads / stores of the scheduling words (other wavefronts, possibly on another XCD, write them)
__device__ __forceinline__ int sched_load(const int * p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned sched_load(const unsigned * p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// bucket of an estimate of `ticks` (100 MHz) of remaining work: four per octave from 2^8 ticks up -- monotone in the
// estimate, which is all the order needs
__device__ __forceinline__ int sched_bucket(long long ticks)
{
  const float f = (float)(ticks > 1 ? ticks : 1);
  const int key = (int)(__float_as_uint(f) >> 21) - ((127 + 8) << 2);
  return key < 0 ? 0 : (key >= kDdpSchedBuckets ? kDdpSchedBuckets - 1 : key);
}

// Fresh tickets -> instances, longest previous busy time first (csrc/ddp_batch.h DdpSched::prev): a counting sort over the
// same four-per-octave buckets, one workgroup; also the decision whether this launch follows the history at all
// (DdpSched::trust).  The order inside a bucket is whatever the atomics make it: the answers do not depend on it.
__global__ __launch_bounds__(1024) void ddp_order_kernel(const float * prev, long n, int * order, int * trust, int check)
{
  __shared__ unsigned hist[kDdpSchedBuckets], base[kDdpSchedBuckets];
  const int t = threadIdx.x;
  if(t < kDdpSchedBuckets) hist[t] = 0u;
  __syncthreads();
  for(long i = t; i < n; i += 1024) atomicAdd(&hist[sched_bucket((long long)prev[i])], 1u);
  __syncthreads();
  if(t == 0)
  {
    // Follow the history unless the previous launch, which could check it, found it wrong (check = 0: nothing to go by
    // yet).  What the order needs from the history is the TAIL: an instance that was among the longest tenth is long
    // again.  The previous launch counted, of the instances its own history put in that tenth (trust[1]), how many came in
    // within two buckets (a factor 1.4) below the tenth's threshold or above (trust[2]); a repeated batch scores ~1, an
    // unrelated one the tail's share of the distribution (~0.2).  (The bulk says nothing either way: most solves of a batch
    // lie within a factor two of its median whatever the batch, and the busy time of a median solve moves by a bucket with
    // the company it keeps on its SIMD.)
    trust[0] = (!check || 2 * trust[2] >= trust[1]) ? 1 : 0;
    trust[1] = 0;
    trust[2] = 0;
    unsigned acc = 0u;
    int tail = kDdpSchedBuckets - 1;
    for(int k = kDdpSchedBuckets - 1; k >= 0; k--) // descending: the highest bucket gets the first tickets
    {
      base[k] = acc;
      acc += hist[k];
      if(10L * acc <= n) tail = k; // (the lowest bucket with at most a tenth of the instances at or above it)
    }
    trust[3] = tail;
  }
  __syncthreads();
  for(long i = t; i < n; i += 1024)
  {
    const unsigned at = atomicAdd(&base[sched_bucket((long long)prev[i])], 1u);
    order[at] = (int)i;
  }
}

template<int S, int NB>
__global__ __launch_bounds__(64, (NB == 1 ? CCC_TILE_WAVES : (NB == 2 ? CCC_TILE_WAVES2 : CCC_TILE_WAVES4))) void ddp_tile_kernel(
    ddp_common::Params P, DdpBatch B, double * ws, size_t ws_stride, long n, DdpSched Sc)
{
  __shared__ ddp_tile::Mem<S, NB> mem;
  constexpr int M = 16 * NB;
  const int N = P.N;
  const int lane = threadIdx.x;
  const size_t sx_stride = (size_t)(N + 1) * S;
  // Work queue: the grid is one resident set of workgroups (one workspace per resident workgroup).  Each
  // takes the next FRESH instance from a ticket counter; when those are handed out, the suspended ones, slowest first.
  bool fresh_left = true;
  // longest-processing-time-first from the previous call's busy times (csrc/ddp_batch.h), when they are there and held
  const bool follow = Sc.use_history != 0 && __hip_atomic_load(Sc.trust, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
  for(;;)
  {
    long b = -1;
    bool resumed = false;
    if(fresh_left)
    {
      unsigned t = 0;
      if(lane == 0) t = atomicAdd(Sc.ticket, 1u);
      t = (unsigned)__builtin_amdgcn_readfirstlane((int)t);
      if((long)t < n)
        b = follow ? (long)Sc.order[t] : (long)t;
      else
        fresh_left = false;
    }
    if(b < 0 && Sc.slice > 0)
    {
      for(;;)
      {
        // lane k looks at bucket k; the highest non-empty one is taken from (compare-and-swap on its head: a head never
        // passes its tail)
        const int hd = sched_load(Sc.head + lane), tl = sched_load(Sc.tail + lane);
        const unsigned long long ne = __ballot(hd < tl);
        if(ne != 0ull)
        {
          const int k = 63 - __builtin_clzll(ne);
          const int hk = __builtin_amdgcn_readlane(hd, k);
          int ok = 0;
          if(lane == 0) ok = atomicCAS(Sc.head + k, hk, hk + 1) == hk ? 1 : 0;
          ok = __builtin_amdgcn_readfirstlane(ok);
          if(!ok) continue;
          int id;
          for(;;) // (the entry was reserved before the tail moved; its writer is a few instructions behind at most)
          {
            id = lane == 0 ? sched_load(Sc.slot + (size_t)k * Sc.cap + hk % Sc.cap) : 0;
            id = __builtin_amdgcn_readfirstlane(id);
            if(id >= 0) break;
            __builtin_amdgcn_s_sleep(8);
          }
          // (the lists are rings: an instance is in one list at a time, so at most n <= cap entries are outstanding and
          //  the entry is free again for the push that comes round to it)
          if(lane == 0) __hip_atomic_store(Sc.slot + (size_t)k * Sc.cap + hk % Sc.cap, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          b = id;
          resumed = true;
          break;
        }
        unsigned f = lane == 0 ? sched_load(Sc.finished) : 0u;
        f = (unsigned)__builtin_amdgcn_readfirstlane((int)f);
        if((long)f >= n) break; // every instance is complete
        __builtin_amdgcn_s_sleep(64); // (an instance still in its first slice may yet be suspended)
      }
    }
    if(b < 0) break;
    ddp_tile::Instance I;
    I.phase_dim = B.phase_dim + b * P.P;
    I.phase_vertex = B.phase_vertex + b * P.P * M * 3;
    I.phase_ridge = B.phase_ridge + b * P.P * M * 3;
    I.step_phase = B.step_phase + b * N;
    I.ref_pos = B.ref_pos + b * (N + 1) * 3;
    I.ref_ori = B.ref_ori ? B.ref_ori + b * (N + 1) * 3 : nullptr;
    I.inertia = B.inertia ? B.inertia + b * 9 : nullptr;
    I.x0 = B.x0 + b * S;
    I.u_init = B.u_init ? B.u_init + b * N * M : nullptr;
    double * w = ws + (size_t)blockIdx.x * ws_stride;
    I.xbuf = w;
    w += (size_t)ddp_tile::kSlots * (N + 1) * S;
    I.ubuf = w;
    w += (size_t)ddp_tile::kSlots * N * M;
    I.ks = w;
    w += (size_t)N * M;
    I.Ks = w;
    I.u_out = B.u_out + b * N * M;
    I.x_out = B.x_out ? B.x_out + b * (N + 1) * S : nullptr;
    I.out_iters = B.iters ? B.iters + b : nullptr;
    I.out_status = B.status ? B.status + b : nullptr;
    I.out_cost = B.cost ? B.cost + b : nullptr;
    ddp_tile::Solver<S, NB> solver(P, I, mem);
    if(resumed)
    {
      __threadfence(); // (acquire: the state below was written by the wavefront that suspended the instance)
      solver.resume(Sc.save_x + (size_t)b * sx_stride, Sc.save_s + (size_t)b * 8);
    }
    else
      solver.begin();
    // busy ticks of the instance so far travel in save_s[4] (the history of the next call; with -DCCC_TILE_TIMING also the
    // first start in save_s[5], scripts/ddp_sched_probe.py)
    const long long tt0 = (long long)wall_clock64();
    double * const tsv = Sc.slice > 0 ? Sc.save_s + (size_t)b * 8 : nullptr;
    if(!resumed && Sc.slice > 0 && lane == 0)
    {
      tsv[4] = 0.0;
#if defined(CCC_TILE_TIMING)
      tsv[5] = (double)tt0;
#endif
    }
#if defined(CCC_TILE_TIMING)
    solver.timing_busy = (Sc.slice > 0 && resumed) ? tsv[4] : 0.0;
    solver.timing_first = (Sc.slice > 0 && resumed) ? tsv[5] : (double)tt0;
    solver.timing_slice0 = tt0;
#endif
    for(int budget = Sc.slice > 0 ? (resumed ? Sc.slice_next : Sc.slice) : -1;; budget = Sc.slice_next)
    {
      const long long t0 = (long long)wall_clock64();
      const int it0 = solver.iters_done;
      if(solver.iterate(budget))
      {
        solver.finish();
        __syncthreads();
        if(Sc.slice > 0 && lane == 0)
        {
          const float busy = (float)(tsv[4] + (double)((long long)wall_clock64() - tt0));
          if(Sc.use_history != 0)
          {
            const int tail = Sc.trust[3];
            if(sched_bucket((long long)Sc.prev[b]) >= tail)
            {
              atomicAdd(Sc.trust + 1, 1);
              if(sched_bucket((long long)busy) >= tail - 2) atomicAdd(Sc.trust + 2, 1);
            }
          }
          Sc.prev[b] = busy;
          atomicAdd(Sc.finished, 1u);
        }
        break;
      }
      // what is left of this solve, as far as one can tell: the pace of the slice x the iterations it may still take
      const long long per = ((long long)wall_clock64() - t0) / (solver.iters_done > it0 ? solver.iters_done - it0 : 1);
      const int k = sched_bucket(per * (P.max_iter - solver.iters_done));
      // longest remaining first: the instance steps aside for fresh ones (nobody knows yet how long those are) and for
      // suspended ones that look about as long or longer; when only shorter ones wait, it keeps its wavefront
      const unsigned tk = lane == 0 ? sched_load(Sc.ticket) : 0u;
      const bool fresh_waiting = (long)(unsigned)__builtin_amdgcn_readfirstlane((int)tk) < n;
      const unsigned long long ne = __ballot(sched_load(Sc.head + lane) < sched_load(Sc.tail + lane));
      const int lowest = k > 0 ? k - 1 : 0;
      if(!fresh_waiting && (ne >> lowest) == 0ull) continue;
      if(follow) continue; // (list scheduling: the order was settled at the hand-out, every solve runs to completion)
      solver.suspend(Sc.save_x + (size_t)b * sx_stride, Sc.save_s + (size_t)b * 8);
      if(lane == 0) tsv[4] = tsv[4] + (double)((long long)wall_clock64() - tt0);
      __threadfence(); // (release: the state is out before the entry is)
      if(lane == 0)
      {
        const int e = atomicAdd(Sc.tail + k, 1);
        __hip_atomic_store(Sc.slot + (size_t)k * Sc.cap + e % Sc.cap, (int)b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      break;
    }
    __syncthreads();
  }
}

// workgroups of one resident set: wavefronts per SIMD (launch bounds above) x 4 SIMDs x CUs
int ddp_tile_grid(long n, int M, int num_cu)
{
  const int per_cu = 4 * (M == 16 ? CCC_TILE_WAVES : (M == 32 ? CCC_TILE_WAVES2 : CCC_TILE_WAVES4));
  const long resident = (long)per_cu * (num_cu > 0 ? num_cu : 256);
  return (int)(n < resident ? n : resident);
}

// layout behind a DdpSched: [ticket, finished, trust x 4, pad .. 64 words][head 64][tail 64][slot 64 x cap][save_s cap x 8]
// [save_x cap x (N+1) S][prev cap][order cap]
static size_t sched_off_slot() { return (size_t)(64 + 2 * kDdpSchedBuckets) * 4; }
static size_t sched_off_s(long cap) { return (sched_off_slot() + (size_t)kDdpSchedBuckets * (size_t)cap * 4 + 255) / 256 * 256; }
static size_t sched_off_x(long cap) { return sched_off_s(cap) + (size_t)cap * 8 * 8; }
static size_t sched_off_prev(long cap, int N, int S) { return sched_off_x(cap) + (size_t)cap * (size_t)(N + 1) * S * 8; }
size_t ddp_sched_bytes(long cap, int N, int S) { return sched_off_prev(cap, N, S) + (size_t)cap * 8 + 256; }
DdpSched ddp_sched_carve(void * mem, long cap, int N, int S)
{
  char * base = static_cast<char *>(mem);
  DdpSched sc;
  sc.ticket = reinterpret_cast<unsigned *>(base);
  sc.finished = sc.ticket + 1;
  sc.head = reinterpret_cast<int *>(base) + 64;
  sc.tail = sc.head + kDdpSchedBuckets;
  sc.slot = reinterpret_cast<int *>(base + sched_off_slot());
  sc.save_s = reinterpret_cast<double *>(base + sched_off_s(cap));
  sc.save_x = reinterpret_cast<double *>(base + sched_off_x(cap));
  sc.prev = reinterpret_cast<float *>(base + sched_off_prev(cap, N, S));
  sc.order = reinterpret_cast<int *>(sc.prev + cap);
  sc.trust = reinterpret_cast<int *>(base) + 2; // (words 2 .. 5 of the header: NOT reset with the counters)
  sc.cap = cap;
  sc.slice = 0;
  sc.slice_next = 0;
  sc.use_history = 0;
  return sc;
}

hipError_t launch_ddp_tile(const ddp_common::Params & P, const DdpBatch & B, double * ws, const DdpSched & sched, int grid, long n,
                           int S, int M, hipStream_t stream)
{
  const size_t stride = ddp_tile_ws_doubles(P.N, S, M);
  // counters and list ends to zero; with slicing, the entries of the lists to -1 ("reserved, not written").  (A kernel of
  // the library's own, not hipMemsetAsync: see zero_words in csrc/common.h.)
  {
    const size_t words = sched_off_slot() / 4, fill = sched.slice > 0 ? (size_t)kDdpSchedBuckets * (size_t)sched.cap : 0;
    const size_t total = words + fill;
    hipLaunchKernelGGL(ddp_sched_reset_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                       reinterpret_cast<int *>(sched.ticket), words, sched.slot, fill);
  }
  if(sched.use_history)
    hipLaunchKernelGGL(ddp_order_kernel, dim3(1), dim3(1024), 0, stream, sched.prev, n, sched.order, sched.trust,
                       sched.use_history > 1 ? 1 : 0);
#define CCC_TILE_LAUNCH(S_, NB_) hipLaunchKernelGGL((ddp_tile_kernel<S_, NB_>), dim3(grid), dim3(64), 0, stream, P, B, ws, stride, n, sched)
  if(S == 9 && M == 16) CCC_TILE_LAUNCH(9, 1);
  else if(S == 12 && M == 16) CCC_TILE_LAUNCH(12, 1);
  else if(S == 9 && M == 32) CCC_TILE_LAUNCH(9, 2);
  else if(S == 12 && M == 32) CCC_TILE_LAUNCH(12, 2);
  else if(S == 9 && M == 64) CCC_TILE_LAUNCH(9, 4);
  else if(S == 12 && M == 64) CCC_TILE_LAUNCH(12, 4);
  else return hipErrorInvalidValue;
#undef CCC_TILE_LAUNCH
  return hipGetLastError();
}
} // namespace ccc_amd
