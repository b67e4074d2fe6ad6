#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
template<int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void swap16(double v, double & a, double & b)
{
  unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  a = __hiloint2double((int)rh[0], (int)rl[0]);
  b = __hiloint2double((int)rh[1], (int)rl[1]);
}
__device__ __forceinline__ void swap32(double v, double & a, double & b)
{
  unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  a = __hiloint2double((int)rh[0], (int)rl[0]);
  b = __hiloint2double((int)rh[1], (int)rl[1]);
}
template<int LG>
__device__ __forceinline__ double group_max(double v)
{
  v = fmax(v, dpp_f64<0xB1>(v));  // quad_perm [1,0,3,2]
  v = fmax(v, dpp_f64<0x4E>(v));  // quad_perm [2,3,0,1]
  v = fmax(v, dpp_f64<0x141>(v)); // row_half_mirror
  v = fmax(v, dpp_f64<0x140>(v)); // row_mirror
  double a, b;
  swap16(v, a, b);
  v = fmax(a, b);
  if(LG == 64)
  {
    swap32(v, a, b);
    v = fmax(a, b);
  }
  return v;
}
__global__ void k(const double * in, double * out32, double * out64, double* dbgA, double* dbgB)
{
  int l = threadIdx.x;
  double v = in[l];
  out32[l] = group_max<32>(v);
  out64[l] = group_max<64>(v);
  double a,b; swap16((double)l, a, b); dbgA[l]=a; dbgB[l]=b;
}
int main()
{
  std::vector<double> h(64), o32(64), o64(64), A(64), B(64);
  for(int t = 0; t < 3; t++)
  {
    for(int i = 0; i < 64; i++) h[i] = std::sin(i * 1.7 + t) * 10 - (t == 2 ? 100 : 0);
    double *d, *d32, *d64, *dA, *dB;
    hipMalloc(&d, 512); hipMalloc(&d32, 512); hipMalloc(&d64, 512); hipMalloc(&dA, 512); hipMalloc(&dB, 512);
    hipMemcpy(d, h.data(), 512, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, d32, d64, dA, dB);
    hipMemcpy(o32.data(), d32, 512, hipMemcpyDeviceToHost);
    hipMemcpy(o64.data(), d64, 512, hipMemcpyDeviceToHost);
    hipMemcpy(A.data(), dA, 512, hipMemcpyDeviceToHost);
    hipMemcpy(B.data(), dB, 512, hipMemcpyDeviceToHost);
    double m0 = -1e300, m1 = -1e300;
    for(int i = 0; i < 32; i++) { m0 = fmax(m0, h[i]); m1 = fmax(m1, h[32 + i]); }
    int bad = 0;
    for(int i = 0; i < 64; i++) { if(o32[i] != (i < 32 ? m0 : m1)) bad++; if(o64[i] != fmax(m0, m1)) bad++; }
    printf("trial %d bad=%d (m0=%g m1=%g got %g %g %g)\n", t, bad, m0, m1, o32[0], o32[63], o64[5]);
    if(t==0){ printf("swap16 a:"); for(int i=0;i<64;i+=8) printf(" %g",A[i]); printf("\nswap16 b:"); for(int i=0;i<64;i+=8) printf(" %g",B[i]); printf("\n"); }
  }
  return 0;
}
