// sharded_linear_mpc_zmp.cpp -- a C++ host reaching every GPU of the node through the C-ABI alone (include/ccc_amd.h,
// "One node, several GPUs"): 65536 LinearMpcZmp instances in host memory, contiguous shards over all visible MI355X,
// planned ZMPs back in host memory; checked against the one-device entry point.
//   g++ -std=c++17 -Iinclude examples/sharded_linear_mpc_zmp.cpp -Lcentroidalcontrolcollection_amd/lib -lccc_amd
#include <ccc_amd.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char ** argv)
{
  const int64_t n = argc > 1 ? std::atoll(argv[1]) : 65536;
  const int N = 32;
  const int D = ccc_device_count();
  if(D <= 0)
  {
    std::fprintf(stderr, "no gfx950 device: %s\n", ccc_last_error_string());
    return 2;
  }
  std::vector<int> devices(D);
  for(int d = 0; d < D; d++) devices[d] = d;
  // a standing robot with random-ish CoM offsets, limits of a double support phase
  std::vector<double> x0(n * 6), zlim(n * 4 * N), zmp(n * 2), zmp1(n * 2);
  std::vector<int32_t> status(n * 2);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) * (1.0 / 16777216.0) - 0.5; };
  for(int64_t k = 0; k < n; k++)
  {
    for(int a = 0; a < 2; a++)
    {
      x0[k * 6 + a * 3 + 0] = 0.04 * rnd();
      x0[k * 6 + a * 3 + 1] = 0.3 * rnd();
      x0[k * 6 + a * 3 + 2] = 0.2 * rnd();
      for(int i = 0; i < N; i++)
      {
        zlim[((k * 2 + a) * 2 + 0) * N + i] = a == 0 ? -0.05 : -0.125;
        zlim[((k * 2 + a) * 2 + 1) * N + i] = a == 0 ? 0.05 : 0.125;
      }
    }
  }
  ccc_zmp_sharded_t * sh = nullptr;
  if(ccc_zmp_sharded_create(1.0, 2.0, 0.0625, devices.data(), D, &sh) != CCC_OK)
  {
    std::fprintf(stderr, "create: %s\n", ccc_last_error_string());
    return 1;
  }
  if(ccc_zmp_sharded_plan_batch(sh, n, x0.data(), zlim.data(), 0.005, zmp.data(), status.data()) != CCC_OK)
  {
    std::fprintf(stderr, "plan: %s\n", ccc_last_error_string());
    return 1;
  }
  ccc_zmp_t * one = nullptr;
  if(ccc_zmp_create(1.0, 2.0, 0.0625, 0, &one) != CCC_OK
     || ccc_zmp_plan_batch(one, n, x0.data(), zlim.data(), 0.005, zmp1.data(), nullptr, nullptr) != CCC_OK)
  {
    std::fprintf(stderr, "single: %s\n", ccc_last_error_string());
    return 1;
  }
  int64_t differ = 0, unsolved = 0;
  for(int64_t k = 0; k < 2 * n; k++)
  {
    differ += zmp[k] != zmp1[k];
    unsolved += CCC_STATUS_CODE(status[k]) != CCC_STATUS_SOLVED;
  }
  std::printf("devices=%d instances=%lld differ=%lld unsolved=%lld\n", D, (long long)n, (long long)differ,
              (long long)unsolved);
  ccc_zmp_destroy(one);
  ccc_zmp_sharded_destroy(sh);
  return (differ == 0 && unsolved == 0) ? 0 : 1;
}
