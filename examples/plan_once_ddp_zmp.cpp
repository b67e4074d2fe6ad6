// plan_once_ddp_zmp.cpp -- planOnce() / planOnceBatch() of CCC::DdpZmp through the drop-in header (host C++ -> header
// shim -> C-ABI -> HIP kernel), set up like /root/reference/tests/src/TestDdpZmp.cpp:17-51,78-92 (2 s horizon @ 20 ms,
// max_iter = 3, warm start = (CoM xy, m g)) with a reference ZMP that steps from 0 to (0.2, 0.1) at t = 2 s.  Prints the
// planned data so that tests/test_ddpzmp_gpu.py can compare them with the Python mirror (same kernel, same inputs).
#include <CCC/DdpZmp.h>

#include <cstdio>

int main()
{
  try
  {
    const double mass = 100.0, dt = 0.02;
    const int N = 100;
    CCC::DdpZmp ddp(mass, dt, N);
    ddp.ddp_solver_->config().max_iter = 3;
    std::function<CCC::DdpZmp::RefData(double)> ref = [](double t)
    {
      CCC::DdpZmp::RefData rd;
      const double s = t < 2.0 ? 0.0 : (t < 2.5 ? (t - 2.0) / 0.5 : 1.0);
      rd.zmp = CCC::Vector3d(0.2 * s, 0.1 * s, 0.0);
      rd.com_z = 1.0;
      return rd;
    };
    CCC::DdpZmp::InitialParam ip;
    ip.pos = CCC::Vector3d(0.01, -0.02, 1.0);
    ip.vel = CCC::Vector3d(0.05, 0.0, 0.0);
    ip.u_list.assign(N, CCC::DdpZmp::InputDimVector(ip.pos[0], ip.pos[1], mass * 9.80665));
    const double times[3] = {0.0, 1.2, 1.9};
    for(double t : times)
    {
      const auto pd = ddp.planOnce(ref, ip, t);
      std::printf("t=%.2f zmp= %.17g %.17g force_z= %.17g iter= %d\n", t, pd.zmp[0], pd.zmp[1], pd.force_z,
                  ddp.ddp_solver_->traceDataList().back().iter);
    }
    const auto all = ddp.planOnceBatch({ref, ref, ref}, {ip, ip, ip}, {times[0], times[1], times[2]});
    for(size_t k = 0; k < all.size(); k++)
      std::printf("batch[%zu] zmp= %.17g %.17g force_z= %.17g\n", k, all[k].zmp[0], all[k].zmp[1], all[k].force_z);
    return 0;
  }
  catch(const std::exception & e)
  {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
}
