/*
 * ccc_amd.h -- C-ABI of the MI355X-native batched centroidal-MPC planOnce() path.
 *
 * The reference (isri-aist/CentroidalControlCollection) has no FFI/plugin layer: its boundary is the
 * C++ class API of its include/CCC headers.  This C-ABI is what a binding for the
 * batched path binds to; the include/CCC headers in THIS repo are shims that keep the reference's
 * class/planOnce() surface and forward to these entry points.  Every entry point cites the
 * reference interface (file:line under /root/reference) it replaces.
 *
 * Conventions: plain C types, caller-owned buffers, int return codes (0 = CCC_OK), no exceptions
 * across the ABI, one handle may be used from one host thread at a time.  "_device" entry points
 * take pointers to memory resident on the handle's GPU and enqueue asynchronously on the given
 * hipStream_t (passed as void*; NULL = the default stream); the others take host pointers and
 * return when the results are in the host buffers.
 */
#ifndef CCC_AMD_H
#define CCC_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* return codes */
#define CCC_OK 0
#define CCC_ERR_INVALID_ARGUMENT 1 /* reference: std::runtime_error on bad dimensions/arguments */
#define CCC_ERR_UNSUPPORTED 2      /* configuration outside what the kernels are built for */
#define CCC_ERR_HIP 3              /* a HIP runtime call failed (ccc_last_error_string has the text) */
#define CCC_ERR_NO_DEVICE 4        /* no gfx950 device visible: the product has NO CPU fallback */

/* per-instance solver status written to the status arrays */
#define CCC_STATUS_SOLVED 0
#define CCC_STATUS_INFEASIBLE 1 /* some zmin > zmax: the reference's QP back-end would report failure */
#define CCC_STATUS_MAX_ITER 2
#define CCC_STATUS_CODE(s) ((s)&0xff)
#define CCC_STATUS_PIVOTS(s) ((s) >> 8)

const char * ccc_last_error_string(void);
/* version of this ABI (bumped on incompatible change) */
int ccc_abi_version(void);

/* =========================================================================================
 * CCC::LinearMpcZmp      /root/reference/include/CCC/LinearMpcZmp.h:97-165
 * ========================================================================================= */
typedef struct ccc_zmp ccc_zmp_t;

/* Replaces CCC::LinearMpcZmp::LinearMpcZmp(com_height, horizon_duration, horizon_dt, qp_solver_type)
 * (include/CCC/LinearMpcZmp.h:134-142) and the LinearMpcZmp1d constructor it forwards to
 * (src/LinearMpcZmp.cpp:9-28): builds the jerk-input CoM-ZMP model, its ZOH discretisation, the
 * output-condensed sequence matrices and the batch-constant QP data, and uploads them to `device`.
 * qp_solver_type has no equivalent: the QP is solved by this library's own exact active-set kernel. */
int ccc_zmp_create(double com_height, double horizon_duration, double horizon_dt, int device, ccc_zmp_t ** out);
void ccc_zmp_destroy(ccc_zmp_t * h);
/* horizon_steps_ = ceil(horizon_duration / horizon_dt)      src/LinearMpcZmp.cpp:13 */
int ccc_zmp_horizon_steps(const ccc_zmp_t * h);
/* host copies of seq_ext_->A_seq_ (N x 3) and seq_ext_->B_seq_ (N x N), row-major
 * (include/CCC/InvariantSequentialExtension.h:191-194); either pointer may be NULL */
int ccc_zmp_get_seq(const ccc_zmp_t * h, double * A_seq, double * B_seq);

/* Replaces n calls of CCC::LinearMpcZmp::planOnce(ref_data_func, initial_param, current_time, control_dt)
 * (include/CCC/LinearMpcZmp.h:151-154, src/LinearMpcZmp.cpp:83-112) with the callbacks already
 * sampled at current_time + i*horizon_dt, i = 0..N-1 (src/LinearMpcZmp.cpp:86-98).
 *
 *   x0      [n][2][3]     per axis (x then y): CoM pos, vel, acc      (InitialParam, LinearMpcZmp.h:113-125)
 *   zlim    [n][2][2][N]  per axis: the N lower limits, then the N upper limits (RefData::zmp_limits, :105-110)
 *   control_dt            < 0 means horizon_dt                         (src/LinearMpcZmp.cpp:72-75)
 *   zmp     [n][2]        planned ZMP (the return value of planOnce)
 *   jerk    [n][2][N]     optional (NULL to skip): the whole planned CoM-jerk sequence, i.e. the QP
 *                         solution of src/LinearMpcZmp.cpp:69 of which the reference keeps only [0]
 *   status  [n][2]        optional, per axis: (pivots << 8) | CCC_STATUS_*  -- low byte = solver status,
 *                         upper bits = active-set pivots spent on that axis' QP
 *
 * All pointers are DEVICE pointers; the call is asynchronous on `stream`. */
int ccc_zmp_plan_batch_device(ccc_zmp_t * h, int64_t n, const double * x0, const double * zlim, double control_dt,
                              double * zmp, double * jerk, int32_t * status, void * stream);

/* Same with HOST pointers: stages through pinned buffers owned by the handle, runs the device entry
 * point, copies back and synchronises. */
int ccc_zmp_plan_batch(ccc_zmp_t * h, int64_t n, const double * x0, const double * zlim, double control_dt,
                       double * zmp, double * jerk, int32_t * status);

#ifdef __cplusplus
}
#endif
#endif
