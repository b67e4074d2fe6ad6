// plan_once_linear_mpc_z.cpp -- planOnce() / planOnceBatch() of CCC::LinearMpcZ through the drop-in header
// (host C++ -> header shim -> C-ABI -> HIP kernel) on the contact / reference schedule of
// /root/reference/tests/src/TestLinearMpcZ.cpp:26-27.  Prints the planned forces so that tests/test_z_gpu.py can compare
// them with the Python mirror (same kernel, same inputs).
#include <CCC/LinearMpcZ.h>

#include <cstdio>

int main()
{
  try
  {
    CCC::LinearMpcZ mpc(100.0, 0.05, 40);
    std::function<bool(double)> contact = [](double t) { return !((5.0 < t && t < 5.25) || (6.0 < t && t < 6.5)); };
    std::function<double(double)> ref = [](double t) { return t < 8.5 ? 1.0 : 0.8; };
    const double times[4] = {0.0, 4.4, 5.1, 7.9};
    const CCC::LinearMpcZ::InitialParam ip(1.05, -0.4);
    for(double t : times) std::printf("t=%.2f force= %.17g\n", t, mpc.planOnce(contact, ref, ip, t));
    const auto all = mpc.planOnceBatch({contact, contact, contact, contact}, {ref, ref, ref, ref}, {ip, ip, ip, ip},
                                       {times[0], times[1], times[2], times[3]});
    for(size_t k = 0; k < all.size(); k++) std::printf("batch[%zu] force= %.17g\n", k, all[k]);
    return 0;
  }
  catch(const std::exception & e)
  {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
}
