// plan_once_ddp.cpp -- one planOnce() of CCC::DdpCentroidal and CCC::DdpSingleRigidBody through the drop-in
// headers (host C++ -> header shim -> C-ABI -> HIP kernel), on the contact / reference schedule of
// /root/reference/tests/src/TestDdpCentroidal.cpp:35-80.  Prints the planned force scales of the first horizon
// step so that tests/test_ddp_gpu.py can compare them with the Python mirror (same kernels, same inputs).
#include <CCC/DdpCentroidal.h>
#include <CCC/DdpSingleRigidBody.h>

#include <cstdio>

int main()
{
  try
  {
    const double dt = 0.03, mass = 100.0;
    const int N = 100;
    auto c0 = CCC::makeContactFromRect({CCC::Vector2d(-0.1, -0.1), CCC::Vector2d(0.1, 0.1)});
    auto c2 = CCC::makeContactFromRect({CCC::Vector2d(0.4, -0.1), CCC::Vector2d(0.6, 0.1)});
    {
      CCC::DdpCentroidal::WeightParam w;
      w.running_pos = CCC::Vector3d(1.0, 1.0, 10.0);
      w.terminal_pos = CCC::Vector3d(1.0, 1.0, 10.0);
      CCC::DdpCentroidal ddp(mass, dt, N, w);
      ddp.ddp_solver_->config().max_iter = 20;
      if(ddp.ddp_solver_->config().horizon_steps != N || !ddp.ddp_solver_->config().with_input_constraint) return 3;
      auto motion = [&](double t) {
        t += 1e-6;
        CCC::DdpCentroidal::MotionParam mp;
        if(t < 1.4)
          mp.contact_list.push_back(c0);
        else if(t >= 1.6)
          mp.contact_list.push_back(c2);
        return mp;
      };
      auto ref = [](double t) {
        t += 1e-6;
        CCC::DdpCentroidal::RefData r;
        r.pos = t < 1.4 ? CCC::Vector3d(0.0, 0.0, 1.0) : (t < 1.6 ? CCC::Vector3d(0.25, 0.0, 1.2) : CCC::Vector3d(0.5, 0.0, 1.0));
        return r;
      };
      CCC::DdpCentroidal::InitialParam ip;
      ip.pos = CCC::Vector3d(0.01, -0.02, 1.0);
      ip.vel = CCC::Vector3d(0.05, 0.0, 0.0);
      CCC::VectorXd u = ddp.planOnce(motion, ref, ip, 0.0);
      std::printf("centroidal iter=%d dim=%d inputDim(1.5)=%d u0=", ddp.ddp_solver_->traceDataList().back().iter, u.size(),
                  ddp.ddp_problem_->inputDim(1.5));
      for(int r = 0; r < u.size(); r++) std::printf(" %.17g", u[r]);
      std::printf("\n");
      // warm start, one iteration (TestDdpCentroidal.cpp:102-116)
      ip.u_list = ddp.ddp_solver_->controlData().u_list;
      ddp.ddp_solver_->config().max_iter = 1;
      u = ddp.planOnce(motion, ref, ip, 0.0);
      std::printf("centroidal_warm iter=%d replaced=%d u0[0]=%.17g\n", ddp.ddp_solver_->traceDataList().back().iter,
                  (int)ddp.ddp_solver_->traceDataList().back().warm_start_replaced, u[0]);
      // a warm start that rolls out worse than zero inputs (3 x the plan): the warm-start guard (on by default, NOT a
      // nmpc_ddp option) replaces it and says so; with the guard off the recalled nmpc_ddp semantics -- u_list handed to
      // solve() as is, src/DdpCentroidal.cpp:229-233 -- apply
      for(auto & ui : ip.u_list)
        for(int r = 0; r < ui.size(); r++) ui[r] = 3.0 * ui[r];
      u = ddp.planOnce(motion, ref, ip, 0.0);
      std::printf("centroidal_bad_warm iter=%d replaced=%d u0[0]=%.17g\n", ddp.ddp_solver_->traceDataList().back().iter,
                  (int)ddp.ddp_solver_->traceDataList().back().warm_start_replaced, u[0]);
      ddp.ddp_solver_->config().warm_start_guard = 0;
      u = ddp.planOnce(motion, ref, ip, 0.0);
      std::printf("centroidal_bad_warm_unguarded iter=%d replaced=%d u0[0]=%.17g\n",
                  ddp.ddp_solver_->traceDataList().back().iter,
                  (int)ddp.ddp_solver_->traceDataList().back().warm_start_replaced, u[0]);
    }
    {
      CCC::DdpSingleRigidBody::WeightParam w;
      w.running_pos = CCC::Vector3d(1.0, 1.0, 10.0);
      w.terminal_pos = CCC::Vector3d(1.0, 1.0, 10.0);
      w.running_ori = CCC::Vector3d::Constant(0.5);
      w.terminal_ori = CCC::Vector3d::Constant(0.5);
      CCC::DdpSingleRigidBody ddp(mass, dt, N, w);
      ddp.ddp_solver_->config().max_iter = 20;
      auto motion = [&](double t) {
        t += 1e-6;
        CCC::DdpSingleRigidBody::MotionParam mp;
        if(t < 1.4)
          mp.contact_list.push_back(c0);
        else if(t >= 1.6)
          mp.contact_list.push_back(c2);
        mp.inertia_mat(0, 0) = 40.0;
        mp.inertia_mat(1, 1) = 20.0;
        mp.inertia_mat(2, 2) = 10.0;
        return mp;
      };
      auto ref = [](double t) {
        t += 1e-6;
        CCC::DdpSingleRigidBody::RefData r;
        r.pos = t < 1.4 ? CCC::Vector3d(0.0, 0.0, 1.0) : (t < 1.6 ? CCC::Vector3d(0.25, 0.0, 1.2) : CCC::Vector3d(0.5, 0.0, 1.0));
        return r;
      };
      CCC::DdpSingleRigidBody::InitialParam ip;
      ip.pos = CCC::Vector3d(0.01, -0.02, 1.0);
      ip.ori = CCC::Vector3d(0.02, -0.01, 0.03);
      CCC::VectorXd u = ddp.planOnce(motion, ref, ip, 0.0);
      std::printf("srb iter=%d dim=%d u0=", ddp.ddp_solver_->traceDataList().back().iter, u.size());
      for(int r = 0; r < u.size(); r++) std::printf(" %.17g", u[r]);
      std::printf("\n");
      // MotionParam::inertia_mat that changes over the horizon: the reference reads motion_param_func_(t).inertia_mat at every
      // step (src/DdpSingleRigidBody.cpp:56-57,120-123), so does the shim (one contact phase per distinct MotionParam)
      auto motion_varying = [&](double t) {
        CCC::DdpSingleRigidBody::MotionParam mp = motion(t);
        const double s = t / 3.0;
        mp.inertia_mat(0, 0) += 5.0 * s;
        mp.inertia_mat(0, 1) = mp.inertia_mat(1, 0) = 1.0 * s;
        mp.inertia_mat(1, 1) += -3.0 * s;
        mp.inertia_mat(1, 2) = mp.inertia_mat(2, 1) = 0.5 * s;
        mp.inertia_mat(2, 2) += 2.0 * s;
        return mp;
      };
      u = ddp.planOnce(motion_varying, ref, ip, 0.0);
      std::printf("srb_varying_inertia iter=%d dim=%d u0=", ddp.ddp_solver_->traceDataList().back().iter, u.size());
      for(int r = 0; r < u.size(); r++) std::printf(" %.17g", u[r]);
      std::printf("\n");
      // force_scale_limits_ is a live public member (include/CCC/DdpSingleRigidBody.h, read by the lambda of
      // src/DdpSingleRigidBody.cpp:272-280 at every solve): assigned after construction, it bounds the next plan
      ddp.force_scale_limits_ = {2.0, 60.0};
      u = ddp.planOnce(motion, ref, ip, 0.0);
      double lo = 1e300, hi = -1e300;
      for(const auto & ui : ddp.ddp_solver_->controlData().u_list)
        for(int r = 0; r < ui.size(); r++)
        {
          lo = ui[r] < lo ? ui[r] : lo;
          hi = ui[r] > hi ? ui[r] : hi;
        }
      std::printf("srb_limits iter=%d min=%.17g max=%.17g u0=", ddp.ddp_solver_->traceDataList().back().iter, lo, hi);
      for(int r = 0; r < u.size(); r++) std::printf(" %.17g", u[r]);
      std::printf("\n");
    }
    {
      // Walking with double support: a two-element contact_list (32 ridges, src/DdpCentroidal.cpp:49-60) and six
      // distinct contact lists inside the horizon; the shim routes the call to a handle with the ridge stride the sampled
      // contact lists need (here 32) behind the same planOnce().
      const double wdt = 0.05;
      const int WN = 40;
      CCC::DdpCentroidal::WeightParam w;
      w.running_pos = CCC::Vector3d(1.0, 1.0, 10.0);
      w.terminal_pos = CCC::Vector3d(1.0, 1.0, 10.0);
      CCC::DdpCentroidal ddp(mass, wdt, WN, w);
      ddp.ddp_solver_->config().max_iter = 30;
      auto foot = [](double x, double y) {
        return CCC::makeContactFromRect({CCC::Vector2d(x - 0.1, y - 0.05), CCC::Vector2d(x + 0.1, y + 0.05)});
      };
      // left foot at y = +0.1, right foot at y = -0.1; phases of 0.4 s: DS, right swings, DS, left swings, DS, flight, DS
      const std::vector<std::shared_ptr<CCC::Contact>> L = {foot(0.0, 0.1), foot(0.3, 0.1), foot(0.6, 0.1)};
      const std::vector<std::shared_ptr<CCC::Contact>> R = {foot(0.0, -0.1), foot(0.15, -0.1), foot(0.45, -0.1)};
      auto motion = [&](double t) {
        t += 1e-6;
        CCC::DdpCentroidal::MotionParam mp;
        const int ph = static_cast<int>(t / 0.3);
        switch(ph)
        {
          case 0: mp.contact_list = {L[0], R[0]}; break;
          case 1: mp.contact_list = {L[0]}; break;
          case 2: mp.contact_list = {L[0], R[1]}; break;
          case 3: mp.contact_list = {R[1]}; break;
          case 4: mp.contact_list = {L[1], R[1]}; break;
          case 5: mp.contact_list = {L[1]}; break;
          default: mp.contact_list = {L[1], R[2]}; break;
        }
        return mp;
      };
      auto ref = [](double t) {
        CCC::DdpCentroidal::RefData r;
        r.pos = CCC::Vector3d(0.15 * t, 0.0, 1.0);
        return r;
      };
      CCC::DdpCentroidal::InitialParam ip;
      ip.pos = CCC::Vector3d(0.0, 0.01, 1.0);
      CCC::VectorXd u = ddp.planOnce(motion, ref, ip, 0.0);
      std::printf("walking iter=%d dim=%d u0=", ddp.ddp_solver_->traceDataList().back().iter, u.size());
      for(int r = 0; r < u.size(); r++) std::printf(" %.17g", u[r]);
      std::printf("\n");
    }
    {
      // Multi-contact: both feet and, for the first 0.6 s, the right hand on a wall at x = 0.45 (a three-element
      // contact_list = 48 ridges; src/DdpCentroidal.cpp:49-60 takes any contact_list).  The hand contact is the
      // rectangle contact mapped by the pose (normal -x, tangent +y): world = centre + (-z, x, -y) of the local frame.
      const double mdt = 0.05;
      const int MN = 24;
      CCC::DdpCentroidal::WeightParam w;
      w.running_pos = CCC::Vector3d(1.0, 1.0, 10.0);
      w.terminal_pos = CCC::Vector3d(1.0, 1.0, 10.0);
      CCC::DdpCentroidal ddp(mass, mdt, MN, w);
      ddp.ddp_solver_->config().max_iter = 25;
      const auto lf = CCC::makeContactFromRect({CCC::Vector2d(-0.1, 0.05), CCC::Vector2d(0.1, 0.15)});
      const auto rf = CCC::makeContactFromRect({CCC::Vector2d(-0.1, -0.15), CCC::Vector2d(0.1, -0.05)});
      auto hand = std::make_shared<CCC::Contact>(*CCC::makeContactFromRect({CCC::Vector2d(-0.05, -0.05), CCC::Vector2d(0.05, 0.05)}));
      for(auto & vr : hand->vertexWithRidgeList_)
      {
        const CCC::Vector3d v = vr.vertex;
        vr.vertex = CCC::Vector3d(0.45 - v[2], -0.2 + v[0], 1.0 - v[1]);
        for(auto & rd : vr.ridgeList)
        {
          const CCC::Vector3d r = rd;
          rd = CCC::Vector3d(-r[2], r[0], -r[1]);
        }
      }
      auto motion = [&](double t) {
        CCC::DdpCentroidal::MotionParam mp;
        if(t + 1e-6 < 0.6)
          mp.contact_list = {lf, rf, hand};
        else
          mp.contact_list = {lf, rf};
        return mp;
      };
      auto ref = [](double t) {
        CCC::DdpCentroidal::RefData r;
        r.pos = CCC::Vector3d(0.05 * t, 0.0, 0.9);
        return r;
      };
      CCC::DdpCentroidal::InitialParam ip;
      ip.pos = CCC::Vector3d(0.01, -0.01, 0.9);
      CCC::VectorXd u = ddp.planOnce(motion, ref, ip, 0.0);
      std::printf("multicontact iter=%d dim=%d u0=", ddp.ddp_solver_->traceDataList().back().iter, u.size());
      for(int r = 0; r < u.size(); r++) std::printf(" %.17g", u[r]);
      std::printf("\n");
    }
    return 0;
  }
  catch(const std::exception & e)
  {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 2;
  }
}
