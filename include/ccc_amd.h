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
 * A handle owns device scratch (work-queue counters, workspaces): launches of ONE handle must not overlap on the
 * device -- enqueue them on one stream (or order the streams); use one handle per concurrent stream.
 * hipGraph: every "_device" call may be captured (hipStreamBeginCapture on the stream it is given) once the handle has
 * run one call of at least that batch size eagerly -- workspaces are allocated, synchronously, when a batch size is
 * first seen; a call enqueues only kernels and counter resets (and, LinearMpcXY, an event fork/join with a stream the
 * handle owns).
 * A captured graph holds the addresses of the handle's workspaces: it stays valid until a LATER call with a larger batch
 * makes a workspace grow (the old one is freed) -- capture at the largest batch the handle will see.  LinearMpcZmp handles
 * retire what a larger batch outgrows instead of freeing it (scheduling buffers, the state-space kernel's
 * stage records and hand-over list): graphs of theirs stay valid for the life of the handle.
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
/* version of this ABI (bumped on incompatible change): CCC_ABI_VERSION is what this header describes; a caller compiled
 * against another version passes structs of another size (ccc_ddp_config_t grew by warm_start_guard in the library
 * that answered 3) -- compare before the first call:  if(ccc_abi_version() != CCC_ABI_VERSION) refuse.
 *   4 (round 5): ccc_ddp_config_t::warm_start_guard counted in; the DDP status word carries
 *     CCC_DDP_STATUS_WARM_REPLACED_BIT; ccc_ddp_effective_precision; stats[5] of ccc_ddp_closed_loop_device. */
#define CCC_ABI_VERSION 5
int ccc_abi_version(void);
/* number of visible HIP devices that are gfx950 parts (0 when none: every create call then fails with
 * CCC_ERR_NO_DEVICE); device ordinals are HIP's */
int ccc_device_count(void);

/* =========================================================================================
 * CCC::LinearMpcZmp      /root/reference/include/CCC/LinearMpcZmp.h:97-165
 * ========================================================================================= */
typedef struct ccc_zmp ccc_zmp_t;

/* Replaces CCC::LinearMpcZmp::LinearMpcZmp(com_height, horizon_duration, horizon_dt, qp_solver_type)
 * (include/CCC/LinearMpcZmp.h:134-142) and the LinearMpcZmp1d constructor it forwards to
 * (src/LinearMpcZmp.cpp:9-28): builds the jerk-input CoM-ZMP model, its ZOH discretisation, the
 * output-condensed sequence matrices and the batch-constant QP data, and uploads them to `device`.
 * qp_solver_type has no equivalent: the QP is solved by this library's own exact active-set kernel.
 * horizon_steps = ceil(horizon_duration / horizon_dt) must be <= 512 (CCC_ERR_UNSUPPORTED beyond). */
int ccc_zmp_create(double com_height, double horizon_duration, double horizon_dt, int device, ccc_zmp_t ** out);
void ccc_zmp_destroy(ccc_zmp_t * h);
/* horizon_steps_ = ceil(horizon_duration / horizon_dt)      src/LinearMpcZmp.cpp:13 */
int ccc_zmp_horizon_steps(const ccc_zmp_t * h);
/* host copies of seq_ext_->A_seq_ (N x 3) and seq_ext_->B_seq_ (N x N), row-major
 * (include/CCC/InvariantSequentialExtension.h:191-194); either pointer may be NULL */
int ccc_zmp_get_seq(const ccc_zmp_t * h, double * A_seq, double * B_seq);
/* Name of the kernel the handle's last plan call launched ("zmp_plan_kernel_dyn<32,2>", "zmp_plan_kernel<32,2>",
 * "zmp_plan_sym_kernel", "zmp_plan_block_kernel", "none" before the first call): what a profile of that call lists.
 * New -- no reference counterpart; for measurement code (bench.py's roofline object). */
const char * ccc_zmp_last_kernel(const ccc_zmp_t * h);
/* What the schedule of the handle's last plan call (N <= 32) came from: "last call's pivot counts" (a handle that sees its
 * batch again, round 5), "predicted pivot counts" (round 6: no usable history -- one pass over the inputs predicts every
 * QP's pivot trips from the rows its unconstrained optimum violates, csrc/zmp.hip zmp_predict_kernel) or "none".  The
 * answers never depend on the schedule.  New (ABI 5); for measurement code and tests. */
const char * ccc_zmp_last_schedule(const ccc_zmp_t * h);

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
 * All pointers are DEVICE pointers; the call is asynchronous on `stream`.
 *
 * Scheduling (N <= 32; round 5): a handle remembers the pivots every QP of its last call took; a call with the same n runs
 * the QPs longest first, QPs of like counts sharing a wavefront (closed-loop callers hand over nearly the same batch cycle
 * after cycle); when the counts of consecutive calls stop agreeing -- unrelated batches of one size -- the handle goes back
 * to the plain dispatch until they agree again.  The answers never depend on it -- bit-identical to a handle created with
 * CCC_ZMP_HISTORY=0 in the environment.  A handle holds per-call state (this order, the work-queue tickets): use one handle per stream at a time. */
int ccc_zmp_plan_batch_device(ccc_zmp_t * h, int64_t n, const double * x0, const double * zlim, double control_dt,
                              double * zmp, double * jerk, int32_t * status, void * stream);

/* Same with HOST pointers, synchronous.  Page-locked caller buffers (hipHostMalloc / hipHostRegister / torch
 * pin_memory) are used in place for N <= 32 (the kernel reads and writes them over PCIe, no copy: 1.39 ms per 65 536
 * instances); pageable ones are staged through pinned buffers owned by the handle, chunk by chunk beside the kernel. */
int ccc_zmp_plan_batch(ccc_zmp_t * h, int64_t n, const double * x0, const double * zlim, double control_dt,
                       double * zmp, double * jerk, int32_t * status);

/* --------------------------------------------------------------------------------------------
 * The steps either side of LinearMpcZmp::planOnce, on the device.
 * A footstep timeline per instance replaces the reference test's FootstepManager
 * (/root/reference/tests/src/FootstepManager.h:130-254): feet start at foot0; footstep j moves foot foot_id[j]
 * (0 = left, 1 = right) to foot_pos[j]; the foot is off the ground during [swing_start[j], swing_end[j]) and sits at the
 * new position from swing_end[j] on; ZMP limits = support region of the feet on the ground -+ half the foot size.
 *   foot0 [n][2][2], foot_pos [n][K][2], foot_id [n][K] i32, swing_start / swing_end [n][K], foot_size [2] HOST
 *   (NULL: (0.1, 0.05), FootstepManager.h:464).  All other pointers DEVICE.
 * -------------------------------------------------------------------------------------------- */
/* com_height / horizon_dt the handle was created with (either may be NULL). */
int ccc_zmp_get_model(const ccc_zmp_t * h, double * com_height, double * horizon_dt);

/* zlim [n][2][2][N] <- what N calls of FootstepManager::makeLinearMpcZmpRefData(t + i*horizon_dt) return
 * (src/LinearMpcZmp.cpp:86-98, FootstepManager.h:356-365 incl. both +1e-6), t = t_eval[k] (DEVICE array) or t_common
 * when t_eval is NULL.  Bit-identical to the host fixture (fixtures.zmp_limits_timeline). */
int ccc_zmp_sample_limits_device(ccc_zmp_t * h, int64_t n, int K, const double * foot0, const double * foot_pos,
                                 const int32_t * foot_id, const double * swing_start, const double * swing_end,
                                 const double * foot_size, const double * t_eval, double t_common, double * zlim,
                                 void * stream);

/* `cycles` control cycles of /root/reference/tests/src/TestLinearMpcZmp.cpp:55-102 for n instances at once, without
 * leaving the device: sample the limits at t, planOnce(control_dt = sim_dt), ComZmpSim2d::update(planned_zmp)
 * (SimModels.h:76-137, exact ZOH), t += sim_dt, add disturb_impulse to both axes' velocity when a disturb_times[d]
 * falls into [t, t + sim_dt) (SimModels.h:125-129 adds impulse.x() to both axes; disturb_times is a HOST array).
 *   com_state    [n][2 axes][2]  in/out  (position, velocity)
 *   planned_zmp  [n][2]          in/out  (in: the ZMP of the previous cycle, used for the first InitialParam, :52,69)
 *   work_x0 [n][2][3], work_zlim [n][2][2][N]   workspaces
 *   violations   [n] i32  optional, incremented for every cycle whose planned ZMP is outside zmpLimits(t) (:86-87)
 *   traj_com / traj_zmp  [cycles][n][2]  optional: CoM position before, planned ZMP of, every cycle
 *   t_end        HOST, optional: the time after the last cycle (accumulated like the reference's `t += sim_dt`)
 * Asynchronous on `stream`. */
int ccc_zmp_closed_loop_device(ccc_zmp_t * h, int64_t n, int K, const double * foot0, const double * foot_pos,
                               const int32_t * foot_id, const double * swing_start, const double * swing_end,
                               const double * foot_size, double * com_state, double * planned_zmp, double t0,
                               double sim_dt, int cycles, int n_disturb, const double * disturb_times,
                               double disturb_impulse, double * work_x0, double * work_zlim, int32_t * violations,
                               double * traj_com, double * traj_zmp, double * t_end, void * stream);

/* =========================================================================================
 * CCC::DdpCentroidal          /root/reference/include/CCC/DdpCentroidal.h:13-366
 * CCC::DdpSingleRigidBody     /root/reference/include/CCC/DdpSingleRigidBody.h
 * ========================================================================================= */
typedef struct ccc_ddp ccc_ddp_t;

#define CCC_DDP_CENTROIDAL 0        /* 9 states  [c, P, L]         src/DdpCentroidal.cpp:32-64 */
#define CCC_DDP_SINGLE_RIGID_BODY 1 /* 12 states [c, alpha, v, w]  src/DdpSingleRigidBody.cpp:52-91 */
#define CCC_DDP_MAX_RIDGES 16       /* default ridge stride: one 4-vertex surface contact per step (the fast kernel) */
#define CCC_DDP_MAX_RIDGES_WIDE 32  /* params.max_ridges = 32: two surface contacts per step (double support) */
#define CCC_DDP_MAX_RIDGES_MULTI 64 /* params.max_ridges = 64: up to four surface contacts per step (feet + hands) */

/* Constructor arguments of DdpCentroidal(mass, horizon_dt, horizon_steps, weight_param)
 * (include/CCC/DdpCentroidal.h:342) / DdpSingleRigidBody (include/CCC/DdpSingleRigidBody.h:379-394), with the
 * WeightParam flattened per state entry:
 *   centroidal: w_run = [running_pos(3), running_linear_momentum(3), running_angular_momentum(3)]
 *   SRB:        w_run = [running_pos(3), running_ori(3), running_linear_vel(3), running_angular_vel(3)]
 * (w_term likewise), w_force = running_force, force_scale_limits = force_scale_limits_ (DdpCentroidal.h:364). */
typedef struct
{
  int model;
  double mass;
  double horizon_dt;
  int horizon_steps;
  double w_run[12], w_term[12], w_force;
  double force_scale_limits[2];
  int max_phases; /* P: contact phases per instance (distinct contact lists inside one horizon); any value >= 1, up to
                   * one phase per horizon step. */
  int max_ridges; /* M: ridge stride of phase_vertex / phase_ridge / u_init / u_out: 0 or 16 = CCC_DDP_MAX_RIDGES, 32 =
                   * CCC_DDP_MAX_RIDGES_WIDE (a step with two 4-vertex surface contacts, src/DdpCentroidal.cpp:49-60
                   * over a two-element contact_list), 64 = CCC_DDP_MAX_RIDGES_MULTI (up to four: feet and hands).  A
                   * problem gives the same answer at every stride that holds it; smaller strides run faster.  Other
                   * values: CCC_ERR_UNSUPPORTED. */
  int inertia_per_phase; /* (ABI 5; single-rigid-body model) 0: `inertia` of ccc_ddp_plan_batch* is [n][3][3], one matrix
                   * per instance for the whole horizon.  != 0: `inertia` is [n][P][3][3], MotionParam::inertia_mat of the
                   * steps in contact phase p: the reference reads motion_param_func_(t).inertia_mat at EVERY step
                   * (src/DdpSingleRigidBody.cpp:56-57 in stateEq, :120-123 in calcStateEqDeriv), so a "contact phase" is
                   * then a distinct MotionParam -- contact list AND inertia matrix; a horizon whose inertia varies from
                   * step to step has one phase per step (max_phases = horizon_steps).  ccc_ddp_set_inertia_per_phase
                   * changes it after construction. */
} ccc_ddp_params_t;

/* ddp_solver_->config() (nmpc_ddp::DDPSolver::Configuration, external).  ccc_ddp_default_config
 * fills the nmpc_ddp defaults with the overrides of src/DdpCentroidal.cpp:197-201 (with_input_constraint = true,
 * initial_lambda = 1e-6, lambda_min = 1e-8, lambda_thre = 1e-7). */
typedef struct
{
  int max_iter;
  double initial_lambda, initial_dlambda, lambda_factor, lambda_min, lambda_max;
  double k_rel_norm_thre, lambda_thre, cost_update_ratio_thre, cost_update_thre;
  double alpha_list[11];
  int reg_type; /* 1: Quu_F + lambda I (default; iLQG regType 1), 2: Vxx + lambda I */
  int precision; /* 64 (default) or 32.  BASELINE configs[4] asks for "fp32 with fp64 tolerance check": both values run the
                  * SAME fp64 kernel (trivially inside any fp64 tolerance).  Rounds 2-3 shipped a build with single-precision
                  * storage of the backward pass; it ran at a third of the fp64 tile kernel's rate and was removed in
                  * round 4, and a solver in single-precision ARITHMETIC does not converge on this problem (the reference's
                  * thresholds -- box-QP gradient 1e-8, cost_update_thre 1e-7, force weight 1e-6 -- sit below fp32
                  * resolution).  Not a nmpc_ddp option. */
  int warm_start_guard; /* 1 (default): a warm start u_init whose open-loop rollout from x0 costs more than the rollout
                  * of zero inputs -- the start planOnce() itself uses when InitialParam::u_list is empty,
                  * src/DdpCentroidal.cpp:221-229 -- or is not finite, is replaced by zero inputs.  It fires on 1 to 7 of
                  * the 601 control cycles of TestDdpSingleRigidBody.cpp:106-153 (one iteration per cycle on an unshifted
                  * warm start) and is what makes that loop pass under perturbations.  0: the
                  * recalled nmpc_ddp behaviour.  Not a nmpc_ddp option; ignored by CCC::DdpZmp.
                  * On a call where it fires the result is, by design, NOT what the reference computes from the same
                  * u_list (src/DdpSingleRigidBody.cpp:299-303 hands initial_param.u_list to solve() as is); every such
                  * instance is flagged in its status word (CCC_DDP_STATUS_WARM_REPLACED below), in
                  * TraceData::warm_start_replaced of the header shims and in stats[5] of ccc_ddp_closed_loop_device.
                  * Unflagged instances are the recalled algorithm, bit for bit. */
} ccc_ddp_config_t;

/* The per-instance status word of the DDP planners (`status` of ccc_ddp_plan_batch*): the exit code -- 0 max_iter
 * reached, 1 gradient small, 2 cost change small, -1 regularisation exceeded lambda_max -- exactly as before ABI 4 on
 * every instance whose warm start was kept (and always with warm_start_guard = 0 or u_init = NULL); on an instance whose
 * warm start the guard replaced by zero inputs: CCC_DDP_STATUS_WARM_REPLACED_BIT | (exit code & 0xff), a value >= 0x100. */
#define CCC_DDP_STATUS_WARM_REPLACED_BIT 0x100
/* (ABI 5) the status every instance carries from the launch until its solve completes: what is left in `status` for the
 * instances a launch did NOT complete because a wait of its scheduler gave up (ccc_ddp_last_call_aborted) */
#define CCC_DDP_STATUS_ABORTED (-2)
#define CCC_DDP_STATUS_EXIT(s) ((int)(signed char)((s)&0xff))
#define CCC_DDP_STATUS_WARM_REPLACED(s) ((s) >= 0 && ((s)&CCC_DDP_STATUS_WARM_REPLACED_BIT) != 0)

void ccc_ddp_default_config(ccc_ddp_config_t * cfg);
int ccc_ddp_create(const ccc_ddp_params_t * params, int device, ccc_ddp_t ** out);
void ccc_ddp_destroy(ccc_ddp_t * h);
/* ddp_solver_->config() = cfg   (e.g. max_iter = 1 after the first control cycle, TestDdpCentroidal.cpp:116) */
int ccc_ddp_set_config(ccc_ddp_t * h, const ccc_ddp_config_t * cfg);
/* force_scale_limits_ = {lo, hi}   (ABI 5).  The reference keeps force_scale_limits_ as a public data member
 * (include/CCC/DdpCentroidal.h:364, include/CCC/DdpSingleRigidBody.h) that the input-limits lambda reads at EVERY solve
 * (src/DdpCentroidal.cpp:202-210, src/DdpSingleRigidBody.cpp:272-280): assigning to it after construction is how a caller
 * changes the limits.  The header shims and the Python mirrors push the member's current value through this entry before
 * every planOnce(); params.force_scale_limits is only the value the handle starts with.  lo <= hi, both finite or
 * +-infinity; applies from the next plan call on. */
int ccc_ddp_set_limits(ccc_ddp_t * h, double lo, double hi);
/* the layout of `inertia` (ccc_ddp_params_t::inertia_per_phase) from the next plan call on   (ABI 5) */
int ccc_ddp_set_inertia_per_phase(ccc_ddp_t * h, int per_phase);
/* (ABI 5) Bounded waits.  The DDP kernel is one resident set of workgroups that pull instances from a work queue; each of
 * its waits (for a queue entry's writer, for the instances still being solved elsewhere) watches the launch's progress
 * and gives up after 10 s without any (CCC_DDP_SPIN_BUDGET_MS in the environment at create time; 0 = never): the kernel
 * then EXITS, the instances it did not complete keep status CCC_DDP_STATUS_ABORTED, ccc_ddp_plan_batch returns
 * CCC_ERR_HIP, and after a ccc_ddp_plan_batch_device call this entry answers 1 once the launch has completed (0 otherwise;
 * -1: NULL handle).  Progress does not depend on the whole grid being resident; the budget catches a wavefront lost with
 * an instance.  The resident set itself is checked against the runtime's occupancy figure in ccc_ddp_create. */
int ccc_ddp_last_call_aborted(const ccc_ddp_t * h);
int ccc_ddp_state_dim(const ccc_ddp_t * h);
/* the constructor arguments / the current configuration / the device of a handle */
int ccc_ddp_get_params(const ccc_ddp_t * h, ccc_ddp_params_t * params);
int ccc_ddp_get_config(const ccc_ddp_t * h, ccc_ddp_config_t * cfg);
int ccc_ddp_get_device(const ccc_ddp_t * h, int * device);
/* Which frozen arithmetic a handle computes in (nmpc_ddp forms its sums with Eigen, whose order is not pinned): always 1
 * since round 4 = the tile arithmetic of oracle/ddp_tile.c (trees, fma chains, the structured backward step), at every
 * ridge stride and for both regularisations; bit-for-bit parity tests run the oracle with this value.  (0 = the dense
 * left-to-right arithmetic of oracle/ddp.c: the oracle's independent cross-check, no kernel.)  New. */
int ccc_ddp_arithmetic(const ccc_ddp_t * h);
/* The precision a handle's kernel computes in, whatever ccc_ddp_config_t::precision asked for: 64 (a `precision = 32`
 * request is accepted and runs the fp64 kernel, see above; this call makes the substitution explicit).  New. */
int ccc_ddp_effective_precision(const ccc_ddp_t * h);

/* Replaces n calls of DdpCentroidal::planOnce / DdpSingleRigidBody::planOnce(motion_param_func, ref_data_func,
 * initial_param, current_time) (src/DdpCentroidal.cpp:213-237, src/DdpSingleRigidBody.cpp:283-307) including the
 * external ddp_solver_->solve (:229,:233), with the callbacks already sampled at current_time + i*dt and the
 * contact lists flattened in contact -> vertex -> ridge order (src/DdpCentroidal.cpp:49-60):
 *
 *   phase_dim    [n][P]            i32  ridges of contact phase p (0 = no contact)        inputDim(t)
 *   phase_vertex [n][P][M][3]      f64  vertex of ridge r         (Contact::vertexWithRidgeList_[..].vertex); M = max_ridges
 *   phase_ridge  [n][P][M][3]      f64  ridge direction r         (..ridgeList[..])
 *   step_phase   [n][N]            i32  contact phase of horizon step i
 *   ref_pos      [n][N+1][3]       f64  RefData::pos at step i (i = N: terminal cost)
 *   ref_ori      [n][N+1][3]       f64  RefData::ori            (SRB only, else NULL)
 *   inertia      [n][3][3]         f64  MotionParam::inertia_mat (SRB only, else NULL): one matrix per instance, or, with
 *                [n][P][3][3]           params.inertia_per_phase, one per contact phase (the step's: src/DdpSingleRigidBody.cpp:56-57,120-123)
 *   x0           [n][S]            f64  InitialParam::toState()  (S = 9: [pos, mass*vel, angular_momentum];
 *                                       S = 12: [pos, ori, linear_vel, angular_vel])
 *   u_init       [n][N][M]         f64  InitialParam::u_list (warm start) or NULL (zeros, src/DdpCentroidal.cpp:221-229)
 *   u_out        [n][N][M]         f64  controlData().u_list; planOnce returns u_out[k][0][0 : phase_dim of step 0]
 *   x_out        [n][N+1][S]       f64  optional: controlData().x_list
 *   iters        [n]               i32  optional: traceDataList().back().iter
 *   status       [n]               i32  optional: the status word above -- CCC_DDP_STATUS_EXIT(s): 0 max_iter reached,
 *                                       1 gradient small, 2 cost change small, -1 regularisation exceeded lambda_max;
 *                                       CCC_DDP_STATUS_WARM_REPLACED(s): the warm-start guard dropped u_init
 *   cost         [n]               f64  optional: final cost
 * All DEVICE pointers, asynchronous on `stream`. */
int ccc_ddp_plan_batch_device(ccc_ddp_t * h, int64_t n, const int32_t * phase_dim, const double * phase_vertex,
                              const double * phase_ridge, const int32_t * step_phase, const double * ref_pos,
                              const double * ref_ori, const double * inertia, const double * x0,
                              const double * u_init, double * u_out, double * x_out, int32_t * iters,
                              int32_t * status, double * cost, void * stream);
/* Same with HOST pointers (H2D, solve, D2H, synchronise). */
int ccc_ddp_plan_batch(ccc_ddp_t * h, int64_t n, const int32_t * phase_dim, const double * phase_vertex,
                       const double * phase_ridge, const int32_t * step_phase, const double * ref_pos,
                       const double * ref_ori, const double * inertia, const double * x0, const double * u_init,
                       double * u_out, double * x_out, int32_t * iters, int32_t * status, double * cost);

/* =========================================================================================
 * CCC::LinearMpcXY          /root/reference/include/CCC/LinearMpcXY.h:27-257
 * ========================================================================================= */
typedef struct ccc_xy ccc_xy_t;

#define CCC_XY_MAX_STEPS 20       /* horizon steps both XY kernels take (BASELINE config 4: N = 20) */
#define CCC_XY_MAX_STEPS_WIDE 256 /* ... beyond that, up to here, the stage-recursion kernel alone (workspace: 64 KB x N / 20 per instance) */
#define CCC_XY_MAX_RIDGES 16      /* default ridge slots per step (one 4-vertex surface contact) */
#define CCC_XY_MAX_RIDGES_WIDE 32 /* params.max_ridges = 32: two surface contacts per step (double support) */
#define CCC_XY_MAX_RIDGES_MULTI 64 /* params.max_ridges = 64: up to four surface contacts per step (feet + hands) */

/* Constructor arguments of LinearMpcXY(mass, horizon_dt, horizon_steps, weight_param, qp_solver_type)
 * (include/CCC/LinearMpcXY.h:211-215, src/LinearMpcXY.cpp:85-94); WeightParam (:104-142) flattened:
 * w_lmi = linear_momentum_integral, w_lm = linear_momentum, w_am = angular_momentum, w_force = force.
 * force_range_ = (3, 3 m g) is fixed as in src/LinearMpcXY.cpp:91. */
typedef struct
{
  double mass;
  double horizon_dt;
  int horizon_steps;
  double w_lmi[2], w_lm[2], w_am[2], w_force;
  int max_ridges; /* M: ridge slots per horizon step: 0 or 16 = CCC_XY_MAX_RIDGES (one 4-vertex surface contact), 32 =
                   * CCC_XY_MAX_RIDGES_WIDE (two: double support; src/LinearMpcXY.cpp:69-82 walks the whole contact_list),
                   * 64 = CCC_XY_MAX_RIDGES_MULTI (up to four: feet and hands).  Other values: CCC_ERR_UNSUPPORTED. */
} ccc_xy_params_t;

int ccc_xy_create(const ccc_xy_params_t * params, int device, ccc_xy_t ** out);
void ccc_xy_destroy(ccc_xy_t * h);
int ccc_xy_get_params(const ccc_xy_t * h, ccc_xy_params_t * params, int * device);

/* Replaces n calls of LinearMpcXY::planOnce(motion_param_func, ref_data_func, initial_param, current_time)
 * (include/CCC/LinearMpcXY.h:224-227, src/LinearMpcXY.cpp:96-182: per-step models :59-83 with their ZOH
 * discretisation, VariantSequentialExtension, QP coefficients, the external QP solve :181), callbacks sampled at
 * current_time + i*dt, contact lists flattened in contact -> vertex -> ridge order (:69-82):
 *
 *   dim            [n][N]          i32  ridges of step i (0: no contact -> no variables, no equality row, :126-133)
 *   vertex, ridge  [n][N][M][3]    f64  per-ridge vertex / ridge direction of step i; M = max_ridges
 *   com_z          [n][N]          f64  MotionParam::com_z
 *   total_force_z  [n][N]          f64  MotionParam::total_force_z
 *   ref_out        [n][N][6]       f64  RefData::toOutput(mass) = [m px, m vx, m py, m vy, Lx, Ly]   (:33-38)
 *   x0             [n][6]          f64  InitialParam::toState(mass)                                  (:26-31)
 *   u0             [n][M]          f64  planned force scales of step 0 (first dim[.][0] entries = the return value)
 *   lambda_all     [n][N][M]       f64  optional: every QP variable, per step
 *   status         [n]             i32  optional: CCC_STATUS_*
 * All DEVICE pointers, asynchronous on `stream`.
 * Scheduling (round 5): a call with the n of the handle's last call takes the instances in the order of the sweeps that call
 * spent on each (the lanes of a wavefront then stop together); the answers do not depend on it (CCC_XY_HISTORY=0 in the
 * environment when the handle is created: never).  One handle per stream at a time. */
int ccc_xy_plan_batch_device(ccc_xy_t * h, int64_t n, const int32_t * dim, const double * vertex, const double * ridge,
                             const double * com_z, const double * total_force_z, const double * ref_out,
                             const double * x0, double * u0, double * lambda_all, int32_t * status, void * stream);
/* Same with HOST pointers. */
int ccc_xy_plan_batch(ccc_xy_t * h, int64_t n, const int32_t * dim, const double * vertex, const double * ridge,
                      const double * com_z, const double * total_force_z, const double * ref_out, const double * x0,
                      double * u0, double * lambda_all, int32_t * status);

/* ============================================================================================
 * CCC::IntrinsicallyStableMpc   (/root/reference/include/CCC/IntrinsicallyStableMpc.h:127-193)
 * the first widening of the hot path (same range-QP kernel family as LinearMpcZmp).
 * ============================================================================================ */
typedef struct ccc_ism ccc_ism_t;

/* Replaces IntrinsicallyStableMpc::IntrinsicallyStableMpc(com_height, horizon_duration, horizon_dt, qp_solver_type,
 * weight_param) (IntrinsicallyStableMpc.h:162-169) -> IntrinsicallyStableMpc1d constructor
 * (src/IntrinsicallyStableMpc.cpp:8-45).  WeightParam{zmp = 1, zmp_vel = 1e-3} (IntrinsicallyStableMpc.h:42-55).
 * horizon_steps = ceil(horizon_duration / horizon_dt) must be <= 191 (up to 127: the tridiagonal projected-Newton kernel
 * with the packed LDS tableau as fallback; 128..191: the tableau kernel alone). */
int ccc_ism_create(double com_height, double horizon_duration, double horizon_dt, double w_zmp, double w_zmp_vel,
                   int device, ccc_ism_t ** out);
void ccc_ism_destroy(ccc_ism_t * h);
int ccc_ism_horizon_steps(const ccc_ism_t * h);

/* Replaces n calls of IntrinsicallyStableMpc::planOnce(ref_data_func, initial_param, current_time, control_dt)
 * (IntrinsicallyStableMpc.h:178-181, src/IntrinsicallyStableMpc.cpp:106-139 incl. procOnce :63-104 and the QP solve
 * :93), ref_data_func sampled at current_time + i*horizon_dt (:112-124):
 *
 *   init    [n][2 axes][2]     f64  (capture_point, planned_zmp) per axis           (InitialParam, .h:144-153)
 *   ref     [n][2 axes][3][N]  f64  rows: RefData::zmp, zmp_limits[0], zmp_limits[1] of the axis
 *   zmp     [n][2]             f64  planned ZMP (the return value)
 *   vel     [n][2][N]          f64  optional: the planned ZMP-velocity sequence (the QP solution)
 *   status  [n][2]             i32  optional, per axis: (pivots << 8) | CCC_STATUS_*
 * control_dt < 0 means horizon_dt (:96-99).  All DEVICE pointers, asynchronous on `stream`. */
int ccc_ism_plan_batch_device(ccc_ism_t * h, int64_t n, const double * init, const double * ref, double control_dt,
                              double * zmp, double * vel, int32_t * status, void * stream);
/* Same with HOST pointers. */
int ccc_ism_plan_batch(ccc_ism_t * h, int64_t n, const double * init, const double * ref, double control_dt,
                       double * zmp, double * vel, int32_t * status);

/* ============================================================================================
 * CCC::LinearMpcZ   (/root/reference/include/CCC/LinearMpcZ.h:14-179)
 * the vertical companion of LinearMpcXY (it plans the total_force_z LinearMpcXY consumes).
 * ============================================================================================ */
typedef struct ccc_z ccc_z_t;

/* Replaces LinearMpcZ::LinearMpcZ(mass, horizon_dt, horizon_steps, weight_param, qp_solver_type)
 * (LinearMpcZ.h:127-131, src/LinearMpcZ.cpp:31-46).  WeightParam{pos = 1, force = 1e-7} (LinearMpcZ.h:34-47);
 * force_range_ = (10, 10 m g) (:37).  horizon_steps <= CCC_Z_MAX_STEPS_WIDE (up to 64 steps both kernels apply, beyond
 * the streaming projected-Newton kernel alone). */
#define CCC_Z_MAX_STEPS_WIDE 256
int ccc_z_create(double mass, double horizon_dt, int horizon_steps, double w_pos, double w_force, int device,
                 ccc_z_t ** out);
void ccc_z_destroy(ccc_z_t * h);

/* Replaces n calls of LinearMpcZ::planOnce(contact_func, ref_pos_func, initial_param, current_time)
 * (LinearMpcZ.h:140-143, src/LinearMpcZ.cpp:48-94 incl. the QP solve :93), callbacks sampled at current_time + i*dt:
 *
 *   contact    [n][N]  i32  contact_func(t_i) (0 / non-zero); step 0 without contact -> planned force 0 (:54-57)
 *   ref_pos    [n][N]  f64  ref_pos_func(t_i)
 *   x0         [n][2]  f64  InitialParam = (CoM height, vertical velocity)                    (LinearMpcZ.h:31)
 *   force      [n]     f64  planned vertical force of step 0 (the return value)
 *   force_all  [n][N]  f64  optional: planned force of every step (0 at steps without contact)
 *   status     [n]     i32  optional: (pivots << 8) | CCC_STATUS_*
 * All DEVICE pointers, asynchronous on `stream`. */
int ccc_z_plan_batch_device(ccc_z_t * h, int64_t n, const int32_t * contact, const double * ref_pos, const double * x0,
                            double * force, double * force_all, int32_t * status, void * stream);
/* Same with HOST pointers. */
int ccc_z_plan_batch(ccc_z_t * h, int64_t n, const int32_t * contact, const double * ref_pos, const double * x0,
                     double * force, double * force_all, int32_t * status);

/* ---------------------------------------------------------------------------------------------------------------
 * CCC::DdpZmp -- csrc/ddpzmp.hip, one instance per lane, trajectories and gains streamed
 * through an HBM workspace laid out [step][field][instance].
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct ccc_ddpzmp ccc_ddpzmp_t;

/* nmpc_ddp::DDPSolver::Configuration defaults (no input constraint; CCC::DdpZmp overrides nothing,
 * include/CCC/DdpZmp.h:277-282): max_iter 500, initial_lambda 1e-4, lambda_min 1e-6, lambda_thre 1e-5, ... */
void ccc_ddpzmp_default_config(ccc_ddp_config_t * cfg);
/* Replaces CCC::DdpZmp::DdpZmp(mass, horizon_dt, horizon_steps, weight_param) (include/CCC/DdpZmp.h:277-282).
 * weights [6] = WeightParam (running_com_pos_z, running_zmp, running_force_z, terminal_com_pos_xy, terminal_com_pos_z,
 * terminal_com_vel), NULL = the defaults of include/CCC/DdpZmp.h:72-77. */
int ccc_ddpzmp_create(double mass, double horizon_dt, int horizon_steps, const double * weights, int device,
                      ccc_ddpzmp_t ** out);
void ccc_ddpzmp_destroy(ccc_ddpzmp_t * h);
/* ddp_solver_->config() = cfg   (e.g. max_iter = 3, tests/src/TestDdpZmp.cpp:29) */
int ccc_ddpzmp_set_config(ccc_ddpzmp_t * h, const ccc_ddp_config_t * cfg);
/* bytes of device workspace a batch of n instances takes (allocated by the library, grown to the largest batch seen) */
int64_t ccc_ddpzmp_workspace_bytes(const ccc_ddpzmp_t * h, int64_t n);

/* Replaces n calls of CCC::DdpZmp::planOnce(ref_data_func, initial_param, current_time) (src/DdpZmp.cpp:156-174)
 * including the external ddp_solver_->solve (:162-169), with ref_data_func already sampled at current_time + i*dt:
 *
 *   ref     [n][N+1][4]  f64  RefData (zmp x, y, z, com_z) at step i (i = N: terminal cost)   include/CCC/DdpZmp.h:19-28
 *   x0      [n][6]       f64  InitialParam::toState() = [pos_x, vel_x, pos_y, vel_y, pos_z, vel_z]   src/DdpZmp.cpp:149-154
 *   u_init  [n][N][3]    f64  InitialParam::u_list (warm start) or NULL (zeros, src/DdpZmp.cpp:161-166)
 *   u_out   [n][N][3]    f64  controlData().u_list; PlannedData::zmp = u_out[.][0][0:2], force_z = u_out[.][0][2] (:171-172)
 *   x_out   [n][N+1][6]  f64  controlData().x_list, or NULL
 *   iters   [n]          i32  iterations executed (traceDataList().back().iter), or NULL
 *   status  [n]          i32  0 max_iter reached, 1 gradient small, 2 cost change small, -1 regularisation exhausted; or NULL
 *   cost    [n]          f64  final cost, or NULL
 * _device: device pointers + a HIP stream (asynchronous); the other entry stages host arrays through the device. */
int ccc_ddpzmp_plan_batch_device(ccc_ddpzmp_t * h, int64_t n, const double * ref, const double * x0,
                                 const double * u_init, double * u_out, double * x_out, int32_t * iters,
                                 int32_t * status, double * cost, void * stream);
int ccc_ddpzmp_plan_batch(ccc_ddpzmp_t * h, int64_t n, const double * ref, const double * x0, const double * u_init,
                          double * u_out, double * x_out, int32_t * iters, int32_t * status, double * cost);

/* The control loop of tests/src/TestDdpZmp.cpp:70-125 for n instances in ONE launch (plan ->
 * simulate -> plan ... on the device): per cycle the RefData of the horizon is sampled from the instance's reference-ZMP
 * polyline (FootstepManager::refZmp, tests/src/FootstepManager.h:228-237), the planner runs warm-started with its
 * previous input sequence (:88-91; first cycle (CoM xy, m g), :84-86), ComZmpSim3d (tests/src/SimModels.h:140-222)
 * advances by sim_dt, and the kicks of :118-125 are added.  All arrays are instance-fastest ("SoA"):
 *   knot_t    [K][n]      f64  knot times of the polyline (ascending; it is constant before the first / after the last)
 *   knot_zmp  [K][2][n]   f64  reference ZMP at the knots
 *   com_height            RefData::com_z (RefData::zmp z = 0)
 *   state     [6][n]      f64  in / out: simulator state [cx, vx, cy, vy, cz, vz]
 *   t0, sim_dt, cycles         time of the first cycle, simulator step, number of cycles
 *   disturb_times [n_disturb <= 8]  HOST: at the first t with times[d] <= t < times[d] + sim_dt (t after the step) the
 *                             impulse per mass `disturb_impulse` is added to BOTH horizontal velocities (reference quirk)
 *   stats     [4][n]      f64  max |planned zmp - ref zmp| over the cycles, max |cz - com_height|, |planned zmp - ref
 *                              zmp| of the last cycle, DDP iterations in total; or NULL
 *   log       [cycles][3][n] f64  planned (zmp x, zmp y, f_z) of every cycle; or NULL
 * Asynchronous on `stream`. */
int ccc_ddpzmp_closed_loop_device(ccc_ddpzmp_t * h, int64_t n, int K, const double * knot_t, const double * knot_zmp,
                                  double com_height, double * state, double t0, double sim_dt, int cycles, int n_disturb,
                                  const double * disturb_times, double disturb_impulse, double * stats, double * log,
                                  void * stream);

/* ---------------------------------------------------------------------------------------------------------------
 * The step after planOnce of the force-scale planners -- csrc/wrench.hip.
 * Replaces n calls of ForceColl::calcTotalWrench(contact_list, force_scales, moment_origin) (external dependency; call
 * sites tests/src/TestLinearMpcXY.cpp:119-120, TestDdpCentroidal.cpp:125-126, TestDdpSingleRigidBody.cpp:141-142) on
 * the flattened contact lists the planners take:
 *   dim     [n]                  i32  ridges of the instance's current contact list
 *   vertex  [n][max_ridges][3]   f64  vertex of ridge r          ridge [n][max_ridges][3]  ridge direction r
 *   scales  [n][scale_stride]    f64  planned force scales (first dim entries used; e.g. u0 of ccc_xy_plan_batch_device
 *                                     with scale_stride = 16, or step 0 of ccc_ddp_plan_batch_device's u with N * 16)
 *   origin  [n][3]               f64  moment origin (the CoM position in the reference tests)
 *   wrench  [n][6]               f64  [moment; force]  (sva::ForceVecd::vector() order)
 * Device pointers + a HIP stream (asynchronous).
 * ------------------------------------------------------------------------------------------------------------- */
int ccc_total_wrench_device(int64_t n, int max_ridges, const int32_t * dim, const double * vertex, const double * ridge,
                            const double * scales, int scale_stride, const double * origin, double * wrench,
                            void * stream);

/* =========================================================================================
 * The closed-loop tests of the force-scale planners on the device -- csrc/centroidal_loop.hip:
 *   plan -> ForceColl::calcTotalWrench about the CoM -> CentroidalSim::update (+ addDisturb) -> plan ...
 * for n instances, replacing the host loops of tests/src/TestDdpCentroidal.cpp:96-150,
 * TestDdpSingleRigidBody.cpp:103-170 and TestLinearMpcXY.cpp:98-132 around tests/src/SimModels.h:233-340.
 * The tests' motion_param_func / ref_data_func are piecewise constant in time; here they are a per-instance CONTACT
 * TIMELINE (all DEVICE pointers):
 *   seg_end      [n][K]        f64  end time of segment k (the last segment never ends; its entry is ignored)
 *   seg_contact  [n][K]        i32  contact-table entry in force during the segment
 *   seg_ref      [n][K][6]     f64  RefData of the segment: CoM position (3), base orientation ZYX (3; SRB only)
 *   contact_dim  [n][C]        i32  ridges of contact-table entry c (0 = flight)
 *   contact_vertex / contact_ridge  [n][C][M][3]   f64  flattened contact -> vertex -> ridge; M = the planner handle's
 *                                                       max_ridges (16, 32 or 64: one, two, up to four surface contacts per entry)
 *   time_eps                        added to every sampling time (the DDP tests add 1e-6, TestDdpCentroidal.cpp:39)
 * A sample at time t takes the first segment with t < seg_end.
 * ========================================================================================= */
typedef struct
{
  int K, C;
  const double * seg_end;
  const int32_t * seg_contact;
  const double * seg_ref;
  const int32_t * contact_dim;
  const double * contact_vertex;
  const double * contact_ridge;
  double time_eps;
} ccc_contact_timeline_t;

/* DdpCentroidal / DdpSingleRigidBody closed loop (C of the timeline = max_phases of the handle: the contact table IS
 * the planner's phase table).  Per cycle: RefData and contact phases sampled at t + i dt, InitialParam from the
 * simulator state (SRB: orientation reversed into Z, Y, X, TestDdpSingleRigidBody.cpp:110-115), warm start = the previous
 * input sequence UNSHIFTED with the steps whose input dimension changed zeroed (:118-127), planOnce with max_iter =
 * first_max_iter in the first cycle and warm_max_iter afterwards (:125), t += sim_dt, CentroidalSim::update with the
 * total wrench of u_list[0] about the CoM, the linear kick disturb_lin when disturb_time <= t < disturb_time + sim_dt.
 *   disturb_times [n_disturb <= 8], disturb_lin [3]  HOST arrays (read before the first cycle is enqueued): one
 *                              linear kick vector, applied at every disturb_times[d]; every other pointer is a DEVICE pointer
 *   inertia_diag [n][3]   f64  moment of inertia of the simulator (the SRB planner takes diag(inertia_diag))
 *   sim_state    [n][18]  f64  in/out: pos, ori (X, Y, Z), vel, ang_vel, linear momentum, angular momentum
 *   stats        [n][8]   f64  optional: max over the cycles, taken where the test asserts (before the update), of
 *                              |pos - ref|, |ori - ref_ori| (unreversed vectors, as the test), |vel|, |ang_vel|, |ang_mom|;
 *                              [5] = the number of cycles whose warm start the warm-start guard replaced
 *                              (CCC_DDP_STATUS_WARM_REPLACED; always 0 with ccc_ddp_config_t::warm_start_guard = 0 and in
 *                              the LinearMpcXY loop); [6..7] = 0
 *   log          [cycles][n][9] f64 optional: pos, planned force, planned moment about the CoM of every cycle
 *   t_end                      optional (host): the time after the last cycle
 * Synchronous (returns when the last cycle is done; its workspaces live and die inside the call). */
int ccc_ddp_closed_loop_device(ccc_ddp_t * h, int64_t n, const ccc_contact_timeline_t * timeline,
                               const double * inertia_diag, double * sim_state, double t0, double sim_dt, int cycles,
                               int first_max_iter, int warm_max_iter, int n_disturb, const double * disturb_times,
                               const double * disturb_lin, double * stats, double * log, double * t_end, void * stream);
/* LinearMpcXY closed loop (TestLinearMpcXY.cpp:98-132): MotionParam{com_z, total_force_z = mass g, contact of the
 * segment}, RefData::pos = the first two entries of seg_ref (the third is the height the statistics compare with). */
int ccc_xy_closed_loop_device(ccc_xy_t * h, int64_t n, const ccc_contact_timeline_t * timeline, double com_z,
                              const double * inertia_diag, double * sim_state, double t0, double sim_dt, int cycles,
                              double * stats, double * log, double * t_end, void * stream);

/* =========================================================================================
 * One node, several GPUs behind the C-ABI (north_star: "the batch dimension shards trivially across the
 * 8 GPUs of one node with an RCCL all-gather of the planned CoM/ZMP outputs over xGMI").  The reference has no
 * counterpart: one CCC::LinearMpcZmp object plans one instance on one CPU thread (src/LinearMpcZmp.cpp:83-112).
 * A sharded handle owns one ccc_zmp_t per listed device; instances are split into contiguous, balanced shards
 * (ccc_shard_bounds: the first n % D shards get one instance more), constants are replicated.
 * ========================================================================================= */
typedef struct ccc_zmp_sharded ccc_zmp_sharded_t;

/* [begin, end) of shard `shard` of n instances over num_shards (the partition every sharded entry point uses) */
int ccc_shard_bounds(int64_t n, int num_shards, int shard, int64_t * begin, int64_t * end);

/* devices [num_devices]: HIP ordinals, each at most once */
int ccc_zmp_sharded_create(double com_height, double horizon_duration, double horizon_dt, const int * devices,
                           int num_devices, ccc_zmp_sharded_t ** out);
void ccc_zmp_sharded_destroy(ccc_zmp_sharded_t * h);
int ccc_zmp_sharded_num_devices(const ccc_zmp_sharded_t * h);

/* Host arrays in, host arrays out, as ccc_zmp_plan_batch (same layouts): shard r is planned on devices[r], one host
 * thread per device calls ccc_zmp_plan_batch on its shard; pinned caller buffers are read and written by the
 * kernels in place.  No collective: the caller's arrays ARE the gathered result. */
int ccc_zmp_sharded_plan_batch(ccc_zmp_sharded_t * h, int64_t n, const double * x0, const double * zlim,
                               double control_dt, double * zmp, int32_t * status);

/* Device-resident shards + RCCL all-gather: on devices[r], x0[r] / zlim[r] hold n_per_device instances (layouts of
 * ccc_zmp_plan_batch_device), zmp_all[r] is a [num_devices * n_per_device][2] array ON devices[r].  Every device plans
 * its shard into its own slot of zmp_all[r]; one in-place ncclAllGather per device (grouped, each on that device's
 * stream, RCCL over xGMI) then leaves the planned ZMPs of ALL shards on EVERY device.  status[r] ([n_per_device][2] on
 * devices[r]) may be NULL.  Synchronous (returns when every device is done).  RCCL is loaded at first use
 * (librccl.so); CCC_ERR_UNSUPPORTED if it is not there. */
int ccc_zmp_sharded_plan_batch_device(ccc_zmp_sharded_t * h, int64_t n_per_device, const double * const * x0,
                                      const double * const * zlim, double control_dt, double * const * zmp_all,
                                      int32_t * const * status);
/* The same with the ORDERING made explicit: caller_streams[r] (a hipStream_t, or NULL = the legacy default stream of
 * devices[r]) is the stream on which the caller produced x0[r] / zlim[r] and last touched zmp_all[r]; the handle's
 * stream of that device waits for it (event record + hipStreamWaitEvent) before the plan kernels start, so inputs
 * written by asynchronous kernels are complete and output buffers still being filled are not overwritten early.
 * ccc_zmp_sharded_plan_batch_device is this call with caller_streams = NULL (every device's default stream). */
int ccc_zmp_sharded_plan_batch_device_ordered(ccc_zmp_sharded_t * h, int64_t n_per_device, const double * const * x0,
                                              const double * const * zlim, double control_dt,
                                              double * const * zmp_all, int32_t * const * status,
                                              void * const * caller_streams);

/* The same for the classes BASELINE's configs put on eight GPUs ("LinearMpcXY ... sharded 8xMI355X over xGMI",
 * "DdpSingleRigidBody ... 8xMI355X"): one planner handle per device, device-resident shards of n_per_device instances in
 * the layouts of ccc_xy_plan_batch_device / ccc_ddp_plan_batch_device (every argument is a list of num_devices DEVICE
 * pointers, entry r on devices[r]), each device plans its shard on the handle's stream of that device (ordered behind
 * caller_streams[r] as above) and one grouped in-place ncclAllGather leaves the planned FIRST-STEP force scales of all
 * shards on every device:
 *   u0_all[r]   [num_devices * n_per_device][M]  on devices[r]; slot r is written by the plan, the rest by the all-gather
 * LinearMpcXY: status[r] [n_per_device] optional.  DDP: u_out[r] [n_per_device][N][M] (the whole planned sequence of the
 * shard stays on its device: it is the next call's warm start), iters / status / cost lists optional, ref_ori / inertia
 * lists NULL for DdpCentroidal, u_init list or entries NULL for a cold start.  Synchronous. */
typedef struct ccc_xy_sharded ccc_xy_sharded_t;
int ccc_xy_sharded_create(const ccc_xy_params_t * params, const int * devices, int num_devices, ccc_xy_sharded_t ** out);
void ccc_xy_sharded_destroy(ccc_xy_sharded_t * h);
int ccc_xy_sharded_num_devices(const ccc_xy_sharded_t * h);
int ccc_xy_sharded_plan_batch_device(ccc_xy_sharded_t * h, int64_t n_per_device, const int32_t * const * dim,
                                     const double * const * vertex, const double * const * ridge,
                                     const double * const * com_z, const double * const * total_force_z,
                                     const double * const * ref_out, const double * const * x0, double * const * u0_all,
                                     int32_t * const * status, void * const * caller_streams);

typedef struct ccc_ddp_sharded ccc_ddp_sharded_t;
/* config may be NULL (ccc_ddp_default_config) */
int ccc_ddp_sharded_create(const ccc_ddp_params_t * params, const ccc_ddp_config_t * config, const int * devices,
                           int num_devices, ccc_ddp_sharded_t ** out);
void ccc_ddp_sharded_destroy(ccc_ddp_sharded_t * h);
int ccc_ddp_sharded_num_devices(const ccc_ddp_sharded_t * h);
int ccc_ddp_sharded_set_config(ccc_ddp_sharded_t * h, const ccc_ddp_config_t * config);
/* ccc_ddp_set_limits on every device's planner (force_scale_limits_ is a live member of the reference classes)   (ABI 5) */
int ccc_ddp_sharded_set_limits(ccc_ddp_sharded_t * h, double lo, double hi);
int ccc_ddp_sharded_plan_batch_device(ccc_ddp_sharded_t * h, int64_t n_per_device, const int32_t * const * phase_dim,
                                      const double * const * phase_vertex, const double * const * phase_ridge,
                                      const int32_t * const * step_phase, const double * const * ref_pos,
                                      const double * const * ref_ori, const double * const * inertia,
                                      const double * const * x0, const double * const * u_init, double * const * u_out,
                                      double * const * u0_all, int32_t * const * iters, int32_t * const * status,
                                      double * const * cost, void * const * caller_streams);

#ifdef __cplusplus
}
#endif
#endif
