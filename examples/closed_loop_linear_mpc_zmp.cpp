// closed_loop_linear_mpc_zmp.cpp -- the reference's TestLinearMpcZmp scenario written against the drop-in
// header include/CCC/LinearMpcZmp.h (host C++ -> C-ABI -> HIP kernels).
//
// Restates /root/reference/tests/src/TestLinearMpcZmp.cpp:15-126 with a closed-form footstep timeline
// (equivalent to FootstepManager::zmpLimits, tests/src/FootstepManager.h:147-254) and the exact-ZOH LIPM
// simulator of tests/src/SimModels.h:11-41,76-137.  Exit code 0 iff the reference's property assertions hold.
//   g++ -std=c++17 -Iinclude examples/closed_loop_linear_mpc_zmp.cpp -Lcentroidalcontrolcollection_amd/lib -lccc_amd
#include <CCC/LinearMpcZmp.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace
{
constexpr double g = 9.80665;

struct Step
{
  int foot; // 0 left, 1 right
  double x, y, swing_start, swing_end;
};

// stance at time t (the two +1e-6 of FootstepManager.h:245,360 are added by the caller)
CCC::LinearMpcZmp::RefData limitsAt(const std::vector<Step> & steps, double t)
{
  double pos[2][2] = {{0.0, 0.1}, {0.0, -0.1}};
  bool ground[2] = {true, true};
  for(const auto & s : steps)
  {
    if(t >= s.swing_end)
    {
      pos[s.foot][0] = s.x;
      pos[s.foot][1] = s.y;
    }
    else if(t >= s.swing_start)
    {
      ground[s.foot] = false;
    }
  }
  CCC::LinearMpcZmp::RefData ref;
  double lo[2] = {1e30, 1e30}, hi[2] = {-1e30, -1e30};
  for(int f = 0; f < 2; f++)
    if(ground[f])
      for(int a = 0; a < 2; a++)
      {
        lo[a] = std::min(lo[a], pos[f][a]);
        hi[a] = std::max(hi[a], pos[f][a]);
      }
  ref.zmp_limits[0] = CCC::Vector2d(lo[0] - 0.05, lo[1] - 0.025);
  ref.zmp_limits[1] = CCC::Vector2d(hi[0] + 0.05, hi[1] + 0.025);
  return ref;
}
} // namespace

int main(int argc, char ** argv)
{
  const double horizon_duration = 2.0;
  const double horizon_dt = argc > 1 ? std::atof(argv[1]) : 0.02;
  const double sim_dt = 0.005, com_height = 1.0, end_time = 10.0;
  const double disturb_time[2] = {4.5, 8.5};
  try
  {
    CCC::LinearMpcZmp mpc(com_height, horizon_duration, horizon_dt);
    std::vector<Step> steps;
    const double td = 0.2, sd = 0.8;
    const double fx[6] = {0.2, 0.4, 0.6, 0.8, 0.6, 0.6};
    for(int k = 0; k < 6; k++)
      steps.push_back({k % 2, fx[k], k % 2 == 0 ? 0.1 : -0.1, 2.0 + k + 0.5 * td, 2.0 + k + 0.5 * td + sd});
    auto ref_func = [&](double t) { return limitsAt(steps, t + 2e-6); };

    const double w = std::sqrt(g / com_height), ch = std::cosh(w * sim_dt), sh = std::sinh(w * sim_dt);
    double sx[2] = {0, 0}, sy[2] = {0, 0}; // (pos, vel) per axis
    CCC::Vector2d planned(0, 0);
    int violations = 0, cycles = 0;
    double t = 0;
    while(t < end_time)
    {
      CCC::LinearMpcZmp::InitialParam ip;
      ip.pos = CCC::Vector2d(sx[0], sy[0]);
      ip.vel = CCC::Vector2d(sx[1], sy[1]);
      ip.acc = CCC::Vector2d(g / com_height * (sx[0] - planned.x()), g / com_height * (sy[0] - planned.y()));
      planned = mpc.planOnce(ref_func, ip, t, sim_dt);
      const auto lim = limitsAt(steps, t + 1e-6);
      if(planned.x() < lim.zmp_limits[0].x() || planned.y() < lim.zmp_limits[0].y()
         || planned.x() > lim.zmp_limits[1].x() || planned.y() > lim.zmp_limits[1].y())
        violations++;
      cycles++;
      t += sim_dt;
      const double nx0 = ch * sx[0] + sh / w * sx[1] + (1 - ch) * planned.x();
      const double nx1 = w * sh * sx[0] + ch * sx[1] - w * sh * planned.x();
      const double ny0 = ch * sy[0] + sh / w * sy[1] + (1 - ch) * planned.y();
      const double ny1 = w * sh * sy[0] + ch * sy[1] - w * sh * planned.y();
      sx[0] = nx0, sx[1] = nx1, sy[0] = ny0, sy[1] = ny1;
      for(double dtm : disturb_time)
        if(dtm <= t && t < dtm + sim_dt)
        {
          sx[1] += 0.05; // SimModels.h:125-129: impulse.x() on both axes
          sy[1] += 0.05;
          break;
        }
    }
    const auto lim = limitsAt(steps, t + 1e-6);
    const bool com_ok = sx[0] >= lim.zmp_limits[0].x() && sx[0] <= lim.zmp_limits[1].x()
                        && sy[0] >= lim.zmp_limits[0].y() && sy[0] <= lim.zmp_limits[1].y();
    std::printf("horizon_steps=%d cycles=%d violations=%d final_com=%.9f %.9f com_inside=%d\n", mpc.horizonSteps(),
                cycles, violations, sx[0], sy[0], (int)com_ok);
    // CCC::LinearMpcZmp1d (LinearMpcZmp.h:20-90) with an explicit QP solver type: the x axis of a 2-d plan
    CCC::LinearMpcZmp1d mpc1d(com_height, horizon_duration, horizon_dt, QpSolverCollection::QpSolverType::QLD);
    auto ref1d = [&](double tt) {
      CCC::LinearMpcZmp1d::RefData rd;
      const auto r2 = ref_func(tt);
      rd.zmp_limits = {r2.zmp_limits[0].x(), r2.zmp_limits[1].x()};
      return rd;
    };
    CCC::LinearMpcZmp::InitialParam ip2;
    ip2.pos = CCC::Vector2d(0.01, -0.02);
    ip2.vel = CCC::Vector2d(0.03, 0.01);
    const CCC::Vector2d z2 = mpc.planOnce(ref_func, ip2, 2.3, sim_dt);
    const double z1 = mpc1d.planOnce(ref1d, CCC::Vector3d(0.01, 0.03, 0.0), 2.3, sim_dt);
    std::printf("1d_matches_2d=%d\n", (int)(z1 == z2.x()));
    return (violations == 0 && com_ok && z1 == z2.x()) ? 0 : 1;
  }
  catch(const std::exception & e)
  {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 2;
  }
}
