/* ShimCommon.h -- pieces the QP-class header shims share: the QpSolverCollection::QpSolverType argument of the
 * reference constructors (/root/reference/include/CCC/LinearMpcZmp.h:45-48, LinearMpcXY.h:211-215, ...) and the
 * reporting of solver failures.
 *
 * The reference hands the QP to QpSolverCollection::QpSolver::solve and never looks at a status: the back-end (QLD,
 * ...) prints its own diagnostic and whatever vector it holds is returned.  The shims keep that contract -- planOnce
 * returns, it does not throw -- but they DO fetch the kernel's status: a plan that is not CCC_STATUS_SOLVED is reported
 * once per call on stderr and through lastStatus() / lastStatuses(), so an infeasible or unfinished QP is never
 * silently taken for a plan.
 */
#pragma once

#include <cstdint>
#include <cstdio>
#include <vector>

#include "../ccc_amd.h"

#if __has_include(<qp_solver_collection/QpSolverCollection.h>)
#  include <qp_solver_collection/QpSolverCollection.h>
#else
namespace QpSolverCollection
{
/** Stand-in for QpSolverCollection::QpSolverType (isri-aist/QpSolverCollection, not in the MI355X build image), so that
    `CCC::LinearMpcZmp mpc(h, T, dt, QpSolverCollection::QpSolverType::QLD)` compiles unchanged.  The value is ignored:
    every QP is solved by this library's own exact active-set kernels on the GPU. */
enum class QpSolverType
{
  Any = -2,
  Uninitialized = -1,
  QLD = 0,
  QuadProg,
  JRLQP,
  qpOASES,
  OSQP,
  NASOQ,
  HPIPM,
  PROXQP,
  QPMAD,
  LSSOL
};
} // namespace QpSolverCollection
#endif

namespace CCC
{
namespace shim
{
/** Count the entries of `status` whose low byte is not CCC_STATUS_SOLVED and say so on stderr (once per call). */
inline int reportStatus(const char * cls, const std::vector<int32_t> & status)
{
  int bad = 0, first = -1;
  for(size_t i = 0; i < status.size(); i++)
  {
    if(CCC_STATUS_CODE(status[i]) != CCC_STATUS_SOLVED)
    {
      if(first < 0) first = static_cast<int>(i);
      bad++;
    }
  }
  if(bad > 0)
  {
    const int code = CCC_STATUS_CODE(status[static_cast<size_t>(first)]);
    std::fprintf(stderr, "[%s] QP not solved for %d of %zu problem(s) (first: #%d, %s); the returned plan is the "
                         "solver's last iterate.\n",
                 cls, bad, status.size(), first,
                 code == CCC_STATUS_INFEASIBLE ? "infeasible" : (code == CCC_STATUS_MAX_ITER ? "iteration limit" : "?"));
  }
  return bad;
}
} // namespace shim
} // namespace CCC
