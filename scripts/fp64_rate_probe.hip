// Development probe: issue rate of v_fma_f64 on one SIMD of gfx950 -- cycles per wave-instruction with 1, 2, 4 wavefronts
// per SIMD and 1 .. 8 independent dependency chains per wavefront (s_memtime around an unrolled loop).
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 scripts/fp64_rate_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template<int CH>
__global__ __launch_bounds__(64) void probe(double * out, long long * clk, int iters, double a, double b)
{
  double x[CH];
#pragma unroll
  for(int c = 0; c < CH; ++c) x[c] = threadIdx.x + c;
  const long long t0 = __builtin_readcyclecounter();
  for(int i = 0; i < iters; ++i)
  {
#pragma unroll
    for(int r = 0; r < 16; ++r)
    {
#pragma unroll
      for(int c = 0; c < CH; ++c) x[c] = __builtin_fma(x[c], a, b);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  double s = 0;
#pragma unroll
  for(int c = 0; c < CH; ++c) s += x[c];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if(threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template<int CH>
void run(int waves_per_simd)
{
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int blocks = p.multiProcessorCount * 4 * waves_per_simd, iters = 2000;
  double * out;
  long long * clk;
  (void)hipMalloc(&out, sizeof(double) * blocks * 64);
  (void)hipMalloc(&clk, sizeof(long long) * blocks);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<CH>, dim3(blocks), dim3(64), 0, 0, out, clk, 10, 0.999, 1e-3);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(probe<CH>, dim3(blocks), dim3(64), 0, 0, out, clk, iters, 0.999, 1e-3);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(blocks);
  (void)hipMemcpy(h.data(), clk, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
  double avg = 0;
  for(long long v : h) avg += (double)v;
  avg /= blocks;
  const double insts = (double)iters * 16 * CH;
  std::printf("chains %d, waves/SIMD %d: %.2f shader clocks per fma per wavefront; kernel %.3f ms -> %.1f TFLOP/s fp64\n", CH,
              waves_per_simd, avg / insts, ms, 2.0 * insts * 64 * blocks / (ms * 1e-3) / 1e12);
  (void)hipFree(out);
  (void)hipFree(clk);
}

int main()
{
  for(int w : {1, 2, 4})
  {
    run<1>(w);
    run<2>(w);
    run<4>(w);
    run<8>(w);
  }
  return 0;
}
