// plan_once_linear_mpc_xy.cpp -- planOnce() and planOnceBatch() of CCC::LinearMpcXY through the drop-in header
// (host C++ -> header shim -> C-ABI -> HIP kernel) on the contact / reference schedule of
// /root/reference/tests/src/TestLinearMpcXY.cpp:29-80.  Prints the planned force scales of the first horizon step so
// that tests/test_xy_gpu.py can compare them with the Python mirror (same kernel, same inputs).
#include <CCC/LinearMpcXY.h>

#include <cstdio>

int main()
{
  try
  {
    const double dt = 0.1, mass = 100.0, g = 9.80665;
    const int N = 15;
    CCC::LinearMpcXY mpc(mass, dt, N);
    auto rect = [](double t, double * ref) {
      double r[4];
      if(t < 3.0) { r[0] = 0.9; r[1] = -0.15; r[2] = 1.1; r[3] = 0.15; ref[0] = 1.0; ref[1] = 0.0; }
      else if(t < 4.0) { r[0] = 0.9; r[1] = 0.05; r[2] = 1.1; r[3] = 0.15; ref[0] = 1.0; ref[1] = 0.1; }
      else if(t < 5.0) { r[0] = 1.15; r[1] = -0.15; r[2] = 1.35; r[3] = -0.05; ref[0] = 1.25; ref[1] = -0.1; }
      else if(t < 6.0) { r[0] = 1.4; r[1] = 0.05; r[2] = 1.6; r[3] = 0.15; ref[0] = 1.5; ref[1] = 0.1; }
      else { r[0] = 1.4; r[1] = -0.15; r[2] = 1.6; r[3] = 0.15; ref[0] = 1.5; ref[1] = 0.0; }
      return CCC::makeContactFromRect({CCC::Vector2d(r[0], r[1]), CCC::Vector2d(r[2], r[3])});
    };
    auto motion = [&](double t) {
      CCC::LinearMpcXY::MotionParam mp;
      double ref[2];
      mp.com_z = 1.0;
      mp.total_force_z = mass * g;
      mp.contact_list.push_back(rect(t, ref));
      return mp;
    };
    auto refdata = [&](double t) {
      CCC::LinearMpcXY::RefData rd;
      double ref[2];
      rect(t, ref);
      rd.pos = CCC::Vector2d(ref[0], ref[1]);
      return rd;
    };
    CCC::LinearMpcXY::InitialParam ip;
    ip.pos = CCC::Vector2d(1.01, -0.02);
    ip.vel = CCC::Vector2d(0.05, 0.0);
    const double times[3] = {0.0, 2.45, 4.3};
    for(double t : times)
    {
      CCC::VectorXd u = mpc.planOnce(motion, refdata, ip, t);
      std::printf("t=%.2f dim=%d u0=", t, u.size());
      for(int r = 0; r < u.size(); r++) std::printf(" %.17g", u[r]);
      std::printf("\n");
    }
    std::vector<std::function<CCC::LinearMpcXY::MotionParam(double)>> mf(3, motion);
    std::vector<std::function<CCC::LinearMpcXY::RefData(double)>> rf(3, refdata);
    std::vector<CCC::LinearMpcXY::InitialParam> ips(3, ip);
    auto all = mpc.planOnceBatch(mf, rf, ips, {times[0], times[1], times[2]});
    for(size_t k = 0; k < all.size(); k++) std::printf("batch[%zu] dim=%d u0[0]=%.17g\n", k, all[k].size(), all[k][0]);
    {
      // Walking with two separate foot contacts in double support (a two-element contact_list = 32 ridges,
      // src/LinearMpcXY.cpp:69-82) over 30 steps: beyond the 20 x 16 tables, routed to the 32-slot handle by the shim.
      const int WN = 30;
      CCC::LinearMpcXY wmpc(mass, dt, WN);
      auto foot = [](double x, double y) {
        return CCC::makeContactFromRect({CCC::Vector2d(x - 0.1, y - 0.05), CCC::Vector2d(x + 0.1, y + 0.05)});
      };
      auto wmotion = [&](double t) {
        CCC::LinearMpcXY::MotionParam mp;
        mp.com_z = 1.0;
        mp.total_force_z = mass * g;
        const int ph = static_cast<int>((t + 1e-9) / 0.5); // 0.5 s phases: DS, left, DS, right, DS, ...
        const double xl = 1.0 + 0.4 * ((ph + 1) / 4), xr = 1.2 + 0.4 * ((ph + 3) / 4 - 1 >= 0 ? (ph + 3) / 4 - 1 : 0);
        if(ph % 2 == 0)
          mp.contact_list = {foot(xl, 0.1), foot(xr, -0.1)};
        else if(ph % 4 == 1)
          mp.contact_list = {foot(xl, 0.1)};
        else
          mp.contact_list = {foot(xr, -0.1)};
        return mp;
      };
      auto wref = [&](double t) {
        CCC::LinearMpcXY::RefData rd;
        rd.pos = CCC::Vector2d(1.1 + 0.2 * t, 0.0);
        return rd;
      };
      CCC::LinearMpcXY::InitialParam wip;
      wip.pos = CCC::Vector2d(1.1, 0.01);
      CCC::VectorXd u = wmpc.planOnce(wmotion, wref, wip, 0.0);
      std::printf("walking dim=%d status=%d u0=", u.size(), wmpc.lastStatuses()[0] & 0xff);
      for(int r = 0; r < u.size(); r++) std::printf(" %.17g", u[r]);
      std::printf("\n");
    }
    {
      // Multi-contact: both feet and the right hand on a wall at x = 1.45 (a three-element contact_list = 48 ridges,
      // src/LinearMpcXY.cpp:69-82 walks the whole list) for the first second: routed to the 64-slot handle by the shim.
      const int MN = 20;
      CCC::LinearMpcXY mmpc(mass, dt, MN);
      const auto lf = CCC::makeContactFromRect({CCC::Vector2d(0.9, 0.05), CCC::Vector2d(1.1, 0.15)});
      const auto rf = CCC::makeContactFromRect({CCC::Vector2d(0.9, -0.15), CCC::Vector2d(1.1, -0.05)});
      auto hand = std::make_shared<CCC::Contact>(*CCC::makeContactFromRect({CCC::Vector2d(-0.05, -0.05), CCC::Vector2d(0.05, 0.05)}));
      for(auto & vr : hand->vertexWithRidgeList_)
      {
        const CCC::Vector3d v = vr.vertex;
        vr.vertex = CCC::Vector3d(1.45 - v[2], -0.2 + v[0], 1.0 - v[1]); // pose: normal -x, tangent +y
        for(auto & rd : vr.ridgeList)
        {
          const CCC::Vector3d r = rd;
          rd = CCC::Vector3d(-r[2], r[0], -r[1]);
        }
      }
      auto mmotion = [&](double t) {
        CCC::LinearMpcXY::MotionParam mp;
        mp.com_z = 0.9;
        mp.total_force_z = mass * g;
        if(t + 1e-9 < 1.0)
          mp.contact_list = {lf, rf, hand};
        else
          mp.contact_list = {lf, rf};
        return mp;
      };
      auto mref = [&](double t) {
        CCC::LinearMpcXY::RefData rd;
        rd.pos = CCC::Vector2d(1.0 + 0.02 * t, 0.0);
        return rd;
      };
      CCC::LinearMpcXY::InitialParam mip;
      mip.pos = CCC::Vector2d(1.01, -0.01);
      CCC::VectorXd u = mmpc.planOnce(mmotion, mref, mip, 0.0);
      std::printf("multicontact dim=%d status=%d u0=", u.size(), mmpc.lastStatuses()[0] & 0xff);
      for(int r = 0; r < u.size(); r++) std::printf(" %.17g", u[r]);
      std::printf("\n");
    }
    return 0;
  }
  catch(const std::exception & e)
  {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
}
