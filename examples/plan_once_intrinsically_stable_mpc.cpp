// plan_once_intrinsically_stable_mpc.cpp -- planOnce() / planOnceBatch() of CCC::IntrinsicallyStableMpc through the
// drop-in header (host C++ -> header shim -> C-ABI -> HIP kernel).  Constant double-support limits and reference, a
// capture point off-centre: prints the planned ZMPs so that tests/test_ism_gpu.py can compare them with the Python mirror.
#include <CCC/IntrinsicallyStableMpc.h>

#include <cstdio>

int main()
{
  try
  {
    CCC::IntrinsicallyStableMpc mpc(1.0, 2.0, 0.02);
    auto ref = [](double t) {
      CCC::IntrinsicallyStableMpc::RefData rd;
      const double x = t < 1.0 ? 0.0 : 0.2;
      rd.zmp = CCC::Vector2d(x, 0.0);
      rd.zmp_limits[0] = CCC::Vector2d(x - 0.05, -0.125);
      rd.zmp_limits[1] = CCC::Vector2d(x + 0.05, 0.125);
      return rd;
    };
    CCC::IntrinsicallyStableMpc::InitialParam ip;
    ip.capture_point = CCC::Vector2d(0.03, -0.06);
    ip.planned_zmp = CCC::Vector2d(0.01, 0.0);
    const double times[3] = {0.0, 0.5, 0.9};
    for(double t : times)
    {
      const CCC::Vector2d z = mpc.planOnce(ref, ip, t, 0.005);
      std::printf("t=%.2f zmp= %.17g %.17g\n", t, z.x(), z.y());
    }
    std::vector<std::function<CCC::IntrinsicallyStableMpc::RefData(double)>> rf(3, ref);
    std::vector<CCC::IntrinsicallyStableMpc::InitialParam> ips(3, ip);
    const auto all = mpc.planOnceBatch(rf, ips, {times[0], times[1], times[2]}, 0.005);
    for(size_t k = 0; k < all.size(); k++) std::printf("batch[%zu] zmp= %.17g %.17g\n", k, all[k].x(), all[k].y());
    std::printf("horizon_steps=%d\n", mpc.horizonSteps());
    // the one-dimensional class of the reference (IntrinsicallyStableMpc.h:17-120) with an explicit QP solver type, as
    // reference code writes it: the x axis of the plan above
    CCC::IntrinsicallyStableMpc1d mpc1d(1.0, 2.0, 0.02, QpSolverCollection::QpSolverType::QLD);
    auto ref1d = [&](double t) {
      CCC::IntrinsicallyStableMpc1d::RefData rd;
      const auto r2 = ref(t);
      rd.zmp = r2.zmp.x();
      rd.zmp_limits = {r2.zmp_limits[0].x(), r2.zmp_limits[1].x()};
      return rd;
    };
    CCC::IntrinsicallyStableMpc1d::InitialParam ip1d;
    ip1d.capture_point = ip.capture_point.x();
    ip1d.planned_zmp = ip.planned_zmp.x();
    std::printf("1d zmp= %.17g status=%d\n", mpc1d.planOnce(ref1d, ip1d, 0.0, 0.005), (int)CCC_STATUS_CODE(mpc1d.lastStatus()));
    return 0;
  }
  catch(const std::exception & e)
  {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
}
