// xy_mfma_probe.hip -- MEASUREMENT AID (not part of the library): the one GEMM-shaped stage BASELINE's north star names for
// CCC::LinearMpcXY -- the blocks  H[s][s'] = B_s' P_{s,s'} B_s'  (16 x 6 . 6 x 6 . 6 x 16) of the condensed Hessian
// B^'WB^ of src/LinearMpcXY.cpp:141-144 -- formed two ways on gfx950, to put a number on what v_mfma_f64_16x16x4_f64 would
// buy:
//   valu   lane (g, c) holds rows 4g .. 4g+3 of column c of the 16 x 16 block: 4 x 6 v_fma_f64 per block
//   mfma   the same block as two v_mfma_f64_16x16x4_f64 (K = 6 padded to 8), same output layout (4 doubles per lane)
// Both read the same operands (T = P B_s' is formed per block by 6-term fma chains on the VALU: 6 x 16 outputs, one per
// lane and a half) and accumulate the blocks of an instance into one 16 x 16 sum, so that the probe is bound by the
// contraction, not by writing 27 GB of Hessians.  Prints blocks/s and fp64 TFLOP/s (useful flop: 2 x 16 x 16 x 6 per block).
//   hipcc --offload-arch=gfx950 -O3 scripts/xy_mfma_probe.hip -o scratch/xy_mfma_probe && scratch/xy_mfma_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

using d4 = __attribute__((ext_vector_type(4))) double;

// one wavefront per instance; N stages, the N(N+1)/2 stage pairs (s <= s') of the block upper triangle
template<bool MFMA>
__global__ __launch_bounds__(64) void probe(const double * __restrict__ Bv, const double * __restrict__ Pm, double * __restrict__ out,
                                            int N, long n)
{
  const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
  __shared__ double T[8 * 16];  // T = P B_s' (6 x 16), rows 6, 7 zero (the K padding of the MFMA path)
  __shared__ double Bl[20 * 8 * 16]; // the instance's ridge vectors, [stage][k (6, padded to 8 with zeros)][ridge]
  for(long b = blockIdx.x; b < n; b += gridDim.x)
  {
    const double * Bi = Bv + b * N * 96; // [stage][6][16]
    const double * Pi = Pm + b * N * 36; // [stage][6][6]  (P_{s,s'} stands in as P_{s'}: the probe times the contraction)
    __syncthreads();
    for(int e = lane; e < N * 128; e += 64)
    {
      const int st = e >> 7, k = (e >> 4) & 7, r = e & 15;
      Bl[e] = k < 6 ? Bi[st * 96 + k * 16 + r] : 0.0;
    }
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    for(int sp = 0; sp < N; sp++)
    {
      // T[j][c] = sum_l P[j][l] B_s'[l][c]  (j = g and g + 4 per lane: rows 0 .. 7, rows 6, 7 are zero)
      __syncthreads();
      for(int jj = 0; jj < 2; jj++)
      {
        const int j = g + 4 * jj;
        double t = 0.0;
        if(j < 6)
          for(int l = 0; l < 6; l++) t = fma(Pi[sp * 36 + j * 6 + l], Bl[sp * 128 + l * 16 + c], t);
        T[j * 16 + c] = t;
      }
      __syncthreads();
      const double t0 = T[g * 16 + c], t1 = T[(g + 4) * 16 + c];
      for(int s = 0; s <= sp; s++)
      {
        if(MFMA)
        {
          // A = B_s' (16 x K): lane (g, c) supplies A[i = c][k = g] of each K chunk; B = T (K x 16): lane supplies T[k = g][j = c]
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Bl[s * 128 + g * 16 + c], t0, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Bl[s * 128 + (g + 4) * 16 + c], t1, acc, 0, 0, 0);
        }
        else
        {
          // rows 4g .. 4g+3 of column c: H[r][c] = sum_k B_s[k][r] T[k][c]   (operands from LDS: B_s rows as b128 pairs)
          for(int k = 0; k < 6; k++)
          {
            const double tk = T[k * 16 + c];
            for(int q = 0; q < 4; q++) acc[q] = fma(Bl[s * 128 + k * 16 + 4 * g + q], tk, acc[q]);
          }
        }
      }
    }
    // (output layout of v_mfma_f64_16x16x4_f64, measured on gfx950: register q of lane (g, c) is D[4 q + g][c])
    for(int q = 0; q < 4; q++) out[b * 256 + (MFMA ? 4 * q + g : 4 * g + q) * 16 + c] = acc[q];
  }
}

int main(int argc, char ** argv)
{
  const long n = argc > 1 ? atol(argv[1]) : 65536;
  const int N = 20, reps = 5;
  std::vector<double> hB((size_t)n * N * 96), hP((size_t)n * N * 36);
  unsigned long long sd = 88172645463325252ull;
  auto rnd = [&]() { sd ^= sd << 13; sd ^= sd >> 7; sd ^= sd << 17; return (double)(sd % 20001) / 10000.0 - 1.0; };
  for(auto & v : hB) v = rnd();
  for(auto & v : hP) v = rnd();
  double *dB, *dP, *dO[2];
  hipMalloc(&dB, hB.size() * 8);
  hipMalloc(&dP, hP.size() * 8);
  hipMalloc(&dO[0], (size_t)n * 256 * 8);
  hipMalloc(&dO[1], (size_t)n * 256 * 8);
  hipMemcpy(dB, hB.data(), hB.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dP, hP.data(), hP.size() * 8, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int grid = 256 * 32;
  const double blocks = (double)n * N * (N + 1) / 2, flop = blocks * 2.0 * 16 * 16 * 6;
  double ms[2];
  for(int m = 0; m < 2; m++)
  {
    for(int r = 0; r < reps + 1; r++)
    {
      if(r == 1) hipEventRecord(e0, 0);
      if(m == 0)
        hipLaunchKernelGGL(probe<false>, dim3(grid), dim3(64), 0, 0, dB, dP, dO[0], N, n);
      else
        hipLaunchKernelGGL(probe<true>, dim3(grid), dim3(64), 0, 0, dB, dP, dO[1], N, n);
    }
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float t;
    hipEventElapsedTime(&t, e0, e1);
    ms[m] = t / reps;
  }
  std::vector<double> o0((size_t)n * 256), o1((size_t)n * 256);
  hipMemcpy(o0.data(), dO[0], o0.size() * 8, hipMemcpyDeviceToHost);
  hipMemcpy(o1.data(), dO[1], o1.size() * 8, hipMemcpyDeviceToHost);
  double md = 0, mx = 0;
  for(size_t i = 0; i < o0.size(); i++)
  {
    md = fmax(md, fabs(o0[i] - o1[i]));
    mx = fmax(mx, fabs(o0[i]));
  }
  printf("{\"probe\": \"LinearMpcXY block-Hessian contraction B_s' P B_s' (16x6 . 6x6 . 6x16), %ld instances x %d stage pairs\", "
         "\"valu_ms\": %.4f, \"mfma_ms\": %.4f, \"valu_tflops\": %.3f, \"mfma_tflops\": %.3f, \"mfma_over_valu_time\": %.3f, "
         "\"max_abs_diff\": %.3g, \"max_abs\": %.3g}\n",
         n, N * (N + 1) / 2, ms[0], ms[1], flop / ms[0] * 1e-9, flop / ms[1] * 1e-9, ms[1] / ms[0], md, mx);
  return 0;
}
