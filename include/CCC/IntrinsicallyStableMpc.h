/* IntrinsicallyStableMpc.h -- drop-in header shim: the class surface of
 * /root/reference/include/CCC/IntrinsicallyStableMpc.h:127-193 over the MI355X C-ABI (include/ccc_amd.h).
 * Same namespace, class, nested RefData / InitialParam / WeightParam, constructor and planOnce() signature; the
 * QpSolverCollection::QpSolverType argument is accepted and ignored (the QP is solved by this library's own kernels); a QP
 * that is not solved -- e.g. an uncatchable capture point -- is reported on stderr and through lastStatuses()
 * (ShimCommon.h).  IntrinsicallyStableMpc1d (IntrinsicallyStableMpc.h:17-120) is provided over the same kernel.  Eigen is used when installed, otherwise the stand-ins of EigenLite.h.
 * planOnceBatch() is new: n independent planOnce() problems in one launch.
 */
#pragma once

#include <array>
#include <cstdint>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../ccc_amd.h"

#include "EigenLite.h"
#include "ShimCommon.h"

namespace CCC
{
/** \brief Intrinsically stable MPC for the ZMP (Scianca 2016), batched on MI355X.  Mirrors CCC::IntrinsicallyStableMpc, /root/reference/include/CCC/IntrinsicallyStableMpc.h:127-193. */
class IntrinsicallyStableMpc
{
public:
  /** \brief Reference data (IntrinsicallyStableMpc.h:133-141). */
  struct RefData
  {
    Vector2d zmp = Vector2d::Zero();          //!< ZMP [m]
    std::array<Vector2d, 2> zmp_limits;        //!< Min/max limits of ZMP [m]
  };

  /** \brief Initial parameter (IntrinsicallyStableMpc.h:144-153). */
  struct InitialParam
  {
    Vector2d capture_point = Vector2d::Zero(); //!< Capture point [m]
    Vector2d planned_zmp = Vector2d::Zero();   //!< Current ZMP planned in previous step [m]
  };

  /** \brief Weight parameter (IntrinsicallyStableMpc.h:42-55, same defaults). */
  struct WeightParam
  {
    double zmp;     //!< ZMP weight
    double zmp_vel; //!< ZMP velocity weight
    WeightParam(double _zmp = 1.0, double _zmp_vel = 1e-3) : zmp(_zmp), zmp_vel(_zmp_vel) {}
  };

public:
  /** \brief Constructor (IntrinsicallyStableMpc.h:162-169).
      \param com_height height of robot CoM [m]
      \param horizon_duration horizon duration [sec]
      \param horizon_dt discretization timestep in horizon [sec]
      \param qp_solver_type ignored (kept for source compatibility)
      \param weight_param objective weight parameter
      \param device HIP device ordinal (new) */
  IntrinsicallyStableMpc(double com_height,
                         double horizon_duration,
                         double horizon_dt,
                         QpSolverCollection::QpSolverType qp_solver_type = QpSolverCollection::QpSolverType::Any,
                         const WeightParam & weight_param = WeightParam(),
                         int device = 0)
  {
    (void)qp_solver_type;
    ccc_ism_t * h = nullptr;
    check(ccc_ism_create(com_height, horizon_duration, horizon_dt, weight_param.zmp, weight_param.zmp_vel, device, &h));
    handle_.reset(h, ccc_ism_destroy);
    horizon_dt_ = horizon_dt;
    horizon_steps_ = ccc_ism_horizon_steps(h);
  }

  /** \brief Plan one step (IntrinsicallyStableMpc.h:178-181, src/IntrinsicallyStableMpc.cpp:106-139).
      \param ref_data_func function of reference data
      \param initial_param initial parameter
      \param current_time current time (i.e., start time of horizon) [sec]
      \param control_dt control timestep used to calculate ZMP (if omitted, horizon_dt is used)
      \returns planned ZMP */
  Vector2d planOnce(const std::function<RefData(double)> & ref_data_func,
                    const InitialParam & initial_param,
                    double current_time,
                    double control_dt = -1)
  {
    const size_t N = static_cast<size_t>(horizon_steps_);
    std::vector<double> init(4), ref(6 * N);
    pack(initial_param, init.data());
    sample(ref_data_func, current_time, ref.data());
    double zmp[2];
    last_status_.assign(2, 0);
    check(ccc_ism_plan_batch(handle_.get(), 1, init.data(), ref.data(), control_dt, zmp, nullptr, last_status_.data()));
    shim::reportStatus("IntrinsicallyStableMpc", last_status_);
    return Vector2d(zmp[0], zmp[1]);
  }

  /** \brief Plan n independent instances in one launch (new). */
  std::vector<Vector2d> planOnceBatch(const std::vector<std::function<RefData(double)>> & ref_data_funcs,
                                      const std::vector<InitialParam> & initial_params,
                                      const std::vector<double> & current_times,
                                      double control_dt = -1)
  {
    const size_t n = ref_data_funcs.size(), N = static_cast<size_t>(horizon_steps_);
    if(initial_params.size() != n || current_times.size() != n)
    {
      throw std::runtime_error("[IntrinsicallyStableMpc::planOnceBatch] argument sizes differ");
    }
    std::vector<double> init(4 * n), ref(6 * N * n), zmp(2 * n);
    for(size_t k = 0; k < n; k++)
    {
      pack(initial_params[k], init.data() + 4 * k);
      sample(ref_data_funcs[k], current_times[k], ref.data() + 6 * N * k);
    }
    last_status_.assign(2 * n, 0);
    check(ccc_ism_plan_batch(handle_.get(), static_cast<int64_t>(n), init.data(), ref.data(), control_dt, zmp.data(),
                             nullptr, last_status_.data()));
    shim::reportStatus("IntrinsicallyStableMpc", last_status_);
    std::vector<Vector2d> out(n);
    for(size_t k = 0; k < n; k++) out[k] = Vector2d(zmp[2 * k], zmp[2 * k + 1]);
    return out;
  }

  /** \brief Number of steps in horizon (IntrinsicallyStableMpc1d::horizon_steps_). */
  int horizonSteps() const
  {
    return horizon_steps_;
  }

  /** \brief The C-ABI handle, for the flat-array entry points of ccc_amd.h. */
  ccc_ism_t * handle() const
  {
    return handle_.get();
  }

  /** \brief Solver status of the last call, [instance][axis]: (pivots << 8) | CCC_STATUS_* (new). */
  const std::vector<int32_t> & lastStatuses() const
  {
    return last_status_;
  }

protected:
  static void check(int rc)
  {
    if(rc != CCC_OK)
    {
      throw std::runtime_error(std::string("[IntrinsicallyStableMpc] ") + ccc_last_error_string());
    }
  }

  static void pack(const InitialParam & ip, double * init)
  {
    // src/IntrinsicallyStableMpc.cpp:127-137: per axis (capture_point, planned_zmp)
    init[0] = ip.capture_point.x();
    init[1] = ip.planned_zmp.x();
    init[2] = ip.capture_point.y();
    init[3] = ip.planned_zmp.y();
  }

  void sample(const std::function<RefData(double)> & ref_data_func, double current_time, double * ref) const
  {
    // src/IntrinsicallyStableMpc.cpp:112-124 -> layout [axis][zmp | min | max][N] of ccc_amd.h
    const int N = horizon_steps_;
    for(int i = 0; i < N; i++)
    {
      const RefData rd = ref_data_func(current_time + i * horizon_dt_);
      ref[(0 * 3 + 0) * N + i] = rd.zmp.x();
      ref[(1 * 3 + 0) * N + i] = rd.zmp.y();
      for(int j = 0; j < 2; j++)
      {
        ref[(0 * 3 + 1 + j) * N + i] = rd.zmp_limits[j].x();
        ref[(1 * 3 + 1 + j) * N + i] = rd.zmp_limits[j].y();
      }
    }
  }

protected:
  std::shared_ptr<ccc_ism_t> handle_;
  double horizon_dt_ = 0;
  int horizon_steps_ = -1;
  std::vector<int32_t> last_status_;
};

/** \brief Intrinsically stable MPC for the one-dimensional motion.
    Mirrors CCC::IntrinsicallyStableMpc1d, /root/reference/include/CCC/IntrinsicallyStableMpc.h:17-120 (planOnce =
    src/IntrinsicallyStableMpc.cpp:47-61 + procOnce :63-104).  The 1-d problem rides on the x axis of the C-ABI's
    two-axis instance; the y axis is given the same problem. */
class IntrinsicallyStableMpc1d
{
public:
  /** \brief Reference data (IntrinsicallyStableMpc.h:23-30). */
  struct RefData
  {
    double zmp = 0;                    //!< ZMP [m]
    std::array<double, 2> zmp_limits;  //!< Min/max limits of ZMP [m]
  };

  /** \brief Initial parameter (IntrinsicallyStableMpc.h:33-40). */
  struct InitialParam
  {
    double capture_point = 0; //!< Capture point [m]
    double planned_zmp = 0;   //!< Current ZMP planned in previous step [m]
  };

  using WeightParam = IntrinsicallyStableMpc::WeightParam;

public:
  /** \brief Constructor (IntrinsicallyStableMpc.h:65-69). */
  IntrinsicallyStableMpc1d(double com_height,
                           double horizon_duration,
                           double horizon_dt,
                           QpSolverCollection::QpSolverType qp_solver_type = QpSolverCollection::QpSolverType::Any,
                           const WeightParam & weight_param = WeightParam(),
                           int device = 0)
  : weight_param_(weight_param)
  {
    (void)qp_solver_type;
    ccc_ism_t * h = nullptr;
    if(ccc_ism_create(com_height, horizon_duration, horizon_dt, weight_param.zmp, weight_param.zmp_vel, device, &h)
       != CCC_OK)
    {
      throw std::runtime_error(std::string("[IntrinsicallyStableMpc1d] ") + ccc_last_error_string());
    }
    handle_.reset(h, ccc_ism_destroy);
    horizon_dt_ = horizon_dt;
    horizon_steps_ = ccc_ism_horizon_steps(h);
  }

  /** \brief Plan one step (IntrinsicallyStableMpc.h:78-81).
      \returns planned ZMP */
  double planOnce(const std::function<RefData(double)> & ref_data_func,
                  const InitialParam & initial_param,
                  double current_time,
                  double control_dt = -1)
  {
    const size_t N = static_cast<size_t>(horizon_steps_);
    std::vector<double> init(4), ref(6 * N);
    for(int a = 0; a < 2; a++)
    {
      init[2 * a] = initial_param.capture_point;
      init[2 * a + 1] = initial_param.planned_zmp;
    }
    for(size_t i = 0; i < N; i++)
    {
      const RefData rd = ref_data_func(current_time + static_cast<double>(i) * horizon_dt_);
      for(size_t a = 0; a < 2; a++)
      {
        ref[(a * 3 + 0) * N + i] = rd.zmp;
        ref[(a * 3 + 1) * N + i] = rd.zmp_limits[0];
        ref[(a * 3 + 2) * N + i] = rd.zmp_limits[1];
      }
    }
    double zmp[2];
    last_status_.assign(2, 0);
    if(ccc_ism_plan_batch(handle_.get(), 1, init.data(), ref.data(), control_dt, zmp, nullptr, last_status_.data())
       != CCC_OK)
    {
      throw std::runtime_error(std::string("[IntrinsicallyStableMpc1d] ") + ccc_last_error_string());
    }
    last_status_.resize(1);
    shim::reportStatus("IntrinsicallyStableMpc1d", last_status_);
    return zmp[0];
  }

  /** \brief Solver status of the last call: (pivots << 8) | CCC_STATUS_* (new). */
  int32_t lastStatus() const
  {
    return last_status_.empty() ? 0 : last_status_[0];
  }

protected:
  WeightParam weight_param_;
  std::shared_ptr<ccc_ism_t> handle_;
  double horizon_dt_ = 0;
  int horizon_steps_ = -1;
  std::vector<int32_t> last_status_;
};
} // namespace CCC
