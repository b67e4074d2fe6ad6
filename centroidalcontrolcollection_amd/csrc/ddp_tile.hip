// ddp_tile.hip -- the TILE build of the DDP planners (csrc/ddp_tile.h): CCC::DdpCentroidal / CCC::DdpSingleRigidBody with
// either regularisation, one instance per wavefront (64-thread workgroup), every matrix distributed over the 64 lanes, the
// backward step in structured form (no M x M object; DESIGN.md sections 7.2-7.4).  Register budget: no spills before
// occupancy -- two wavefronts per SIMD at 16 and 32 ridges per step, one at 64 (CCC_TILE_WAVES* below; the compiler's
// register / scratch / LDS figures of every instantiation are pinned in tests/test_kernel_resources.py); LDS 10.7-15.4 KB
// per wavefront (pinned with the register figures).  One resident set of workgroups pulls instances from a work queue, longest
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
// status of an instance the launch did not complete (include/ccc_amd.h CCC_DDP_STATUS_ABORTED): given to every instance at
// launch, overwritten by the exit code when its solve finishes
constexpr int kDdpStatusAborted = -2;
__global__ void ddp_sched_reset_kernel(int * counters, size_t words, unsigned long long * slot, size_t fill, int * status, size_t n)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if(i < words)
  {
    if(i < 2 || i >= 6) counters[i] = 0; // (words 2 .. 5 carry the verdict on the history from launch to launch: DdpSched::trust)
  }
  else if(i - words < fill)
    slot[i - words] = 0ull; // (free)
  if(status && i < n) status[i] = kDdpStatusAborted;
}

// device-scope loads / stores of the scheduling words (other wavefronts, possibly on another XCD, write them)
__device__ __forceinline__ int sched_load(const int * p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned sched_load(const unsigned * p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long sched_load(const unsigned long long * p)
{
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One look of a waiting wavefront at the launch's state (DdpSched: bounded waits).  True = leave: another wait gave up
// already, or this one has looked spin_limit times in a row without seeing an instance finish or a slice end -- then it
// raises `abort` itself, for the other wavefronts and (page-locked word) for the host.  `watch` is the wavefront's
// watch (two words of LDS).  The budget counts LOOKS, not time (a look of the long wait is an s_sleep 64 and two
// loads, ~4 us: csrc/ddp.hip converts): reading the clock here (s_memrealtime in the wait loops) cost every build of the
// kernel 200 B of scratch per lane -- the words of DdpSched by value for the same reason.
__device__ __forceinline__ bool sched_wait_gives_up(int * abort, int * abort_host, const unsigned * finished, const unsigned * beat,
                                                    unsigned spin_limit, int lane, unsigned * watch)
{
  // (lane 0 alone, its watch in LDS -- watch[0]: progress last seen, 0xffffffff starts the watch; watch[1]: looks since:
  //  carried in scalar registers by the whole wavefront, the same statements cost every build 200 B of scratch per lane)
  int r = 0;
  if(lane == 0)
  {
    if(sched_load(abort) != 0)
      r = 1;
    else
    {
      const unsigned pr = (sched_load(finished) + sched_load(beat)) & 0x7fffffffu;
      if(pr != watch[0])
      {
        watch[0] = pr;
        watch[1] = 0u;
      }
      else if(spin_limit != 0u && ++watch[1] > spin_limit)
      {
        __hip_atomic_store(abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if(abort_host) __hip_atomic_store(abort_host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        r = 1;
      }
    }
  }
  return __builtin_amdgcn_readfirstlane(r) != 0;
}

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

#define CCC_TILE_BOUNDS(NB) __launch_bounds__(64, (NB == 1 ? CCC_TILE_WAVES : (NB == 2 ? CCC_TILE_WAVES2 : CCC_TILE_WAVES4)))
template<int S, int NB>
__global__ CCC_TILE_BOUNDS(NB) void ddp_tile_kernel(ddp_common::Params P, DdpBatch B, double * ws, size_t ws_stride, long n, DdpSched Sc)
{
#define CCC_TILE_IPP false
#include "ddp_tile_body.inc"
#undef CCC_TILE_IPP
}
// one inertia matrix per contact phase (ccc_ddp_params_t::inertia_per_phase; the single-rigid-body model): kernels of their
// own, so that the one-matrix-per-instance kernels keep their names and their register allocation
template<int NB>
__global__ CCC_TILE_BOUNDS(NB) void ddp_tile_ipp_kernel(ddp_common::Params P, DdpBatch B, double * ws, size_t ws_stride, long n, DdpSched Sc)
{
  constexpr int S = 12;
#define CCC_TILE_IPP true
#include "ddp_tile_body.inc"
#undef CCC_TILE_IPP
}
#undef CCC_TILE_BOUNDS

// workgroups per CU of one resident set: wavefronts per SIMD (launch bounds above) x 4 SIMDs, capped by what the runtime
// grants the kernel on the current device (registers, LDS) -- the scheduler's waits do not NEED the whole grid resident
// (csrc/ddp_batch.h), but workgroups beyond the resident set would only queue behind it
int ddp_tile_blocks_per_cu(int S, int M, bool ipp)
{
  const int want = 4 * (M == 16 ? CCC_TILE_WAVES : (M == 32 ? CCC_TILE_WAVES2 : CCC_TILE_WAVES4));
  int got = 0;
  hipError_t e = hipErrorInvalidValue;
#define CCC_TILE_OCC(K) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&got, K, 64, 0)
  if(S == 12 && ipp)
  {
    if(M == 16) CCC_TILE_OCC(ddp_tile_ipp_kernel<1>);
    else if(M == 32) CCC_TILE_OCC(ddp_tile_ipp_kernel<2>);
    else if(M == 64) CCC_TILE_OCC(ddp_tile_ipp_kernel<4>);
  }
  else if(S == 9 && M == 16) CCC_TILE_OCC((ddp_tile_kernel<9, 1>));
  else if(S == 12 && M == 16) CCC_TILE_OCC((ddp_tile_kernel<12, 1>));
  else if(S == 9 && M == 32) CCC_TILE_OCC((ddp_tile_kernel<9, 2>));
  else if(S == 12 && M == 32) CCC_TILE_OCC((ddp_tile_kernel<12, 2>));
  else if(S == 9 && M == 64) CCC_TILE_OCC((ddp_tile_kernel<9, 4>));
  else if(S == 12 && M == 64) CCC_TILE_OCC((ddp_tile_kernel<12, 4>));
#undef CCC_TILE_OCC
  if(e != hipSuccess || got <= 0) return 0;
  return got < want ? got : want;
}
int ddp_tile_grid(long n, int per_cu, int num_cu)
{
  const long resident = (long)per_cu * (num_cu > 0 ? num_cu : 256);
  return (int)(n < resident ? n : resident);
}

// layout behind a DdpSched: [ticket, finished, trust x 4, beat, abort, pad .. 64 words][head 64][tail 64][slot 64 x cap][save_s cap x 8]
// [save_x cap x (N+1) S][prev cap][order cap]
static size_t sched_off_slot() { return (size_t)(64 + 2 * kDdpSchedBuckets) * 4; }
static size_t sched_off_s(long cap) { return (sched_off_slot() + (size_t)kDdpSchedBuckets * (size_t)cap * 8 + 255) / 256 * 256; }
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
  sc.slot = reinterpret_cast<unsigned long long *>(base + sched_off_slot());
  sc.save_s = reinterpret_cast<double *>(base + sched_off_s(cap));
  sc.save_x = reinterpret_cast<double *>(base + sched_off_x(cap));
  sc.prev = reinterpret_cast<float *>(base + sched_off_prev(cap, N, S));
  sc.order = reinterpret_cast<int *>(sc.prev + cap);
  sc.trust = reinterpret_cast<int *>(base) + 2; // (words 2 .. 5 of the header: NOT reset with the counters)
  sc.beat = sc.ticket + 6;
  sc.abort = reinterpret_cast<int *>(base) + 7;
  sc.abort_host = nullptr;
  sc.spin_limit = 0u;
  sc.test_drop = -1;
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
    const size_t total = words + fill > (size_t)n ? words + fill : (size_t)n;
    hipLaunchKernelGGL(ddp_sched_reset_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                       reinterpret_cast<int *>(sched.ticket), words, sched.slot, fill, B.status, (size_t)n);
  }
  if(sched.use_history)
    hipLaunchKernelGGL(ddp_order_kernel, dim3(1), dim3(1024), 0, stream, sched.prev, n, sched.order, sched.trust,
                       sched.use_history > 1 ? 1 : 0);
#define CCC_TILE_LAUNCH(S_, NB_) hipLaunchKernelGGL((ddp_tile_kernel<S_, NB_>), dim3(grid), dim3(64), 0, stream, P, B, ws, stride, n, sched)
#define CCC_TILE_LAUNCH_IPP(NB_) hipLaunchKernelGGL((ddp_tile_ipp_kernel<NB_>), dim3(grid), dim3(64), 0, stream, P, B, ws, stride, n, sched)
  if(S == 12 && P.inertia_per_phase)
  {
    if(M == 16) CCC_TILE_LAUNCH_IPP(1);
    else if(M == 32) CCC_TILE_LAUNCH_IPP(2);
    else if(M == 64) CCC_TILE_LAUNCH_IPP(4);
    else return hipErrorInvalidValue;
  }
  else if(S == 9 && M == 16) CCC_TILE_LAUNCH(9, 1);
  else if(S == 12 && M == 16) CCC_TILE_LAUNCH(12, 1);
  else if(S == 9 && M == 32) CCC_TILE_LAUNCH(9, 2);
  else if(S == 12 && M == 32) CCC_TILE_LAUNCH(12, 2);
  else if(S == 9 && M == 64) CCC_TILE_LAUNCH(9, 4);
  else if(S == 12 && M == 64) CCC_TILE_LAUNCH(12, 4);
  else return hipErrorInvalidValue;
#undef CCC_TILE_LAUNCH
#undef CCC_TILE_LAUNCH_IPP
  return hipGetLastError();
}
} // namespace ccc_amd
