/*
 * miqp_gpu.h - C ABI of libmiqp_gpu.so, the MI355X-native replacement of the reference's
 * CPLEX/OPL solve path (L1+L2+L3 of SURVEY.md section 1).
 *
 * Each entry point names the reference interface it replaces (paths relative to the planner-miqp
 * checkout).  The Eigen-typed class with the reference's own name and methods is the header-only
 * adapter include/cplex_wrapper.hpp, which forwards to these functions.
 *
 * Threading: one caller thread per solver handle (src/cplex_wrapper.hpp:61-275 has the same contract: one IloEnv
 * per wrapper).  Handles are independent objects; the device buffers behind them are kept per HIP device and a solve
 * holds that device's lock, so threads with different handles may call concurrently (same device: the solves run one
 * after the other; different devices: in parallel).  No GPU context is touched before the first solve.
 */
#ifndef MIQP_GPU_H
#define MIQP_GPU_H

#include "miqp_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct miqp_solver miqp_solver_t;

typedef struct miqp_solver_opts {
  int precision;            /* CplexWrapper ctor `precision` (cplex_wrapper.hpp:81-115); inputs are rounded to
                               precision-2 decimals like ModelInputDataSource (hpp:88); <= 0 -> 12 */
  int device;               /* HIP device ordinal, -1: the calling thread's current device */
  int nodes_per_round;      /* B&B nodes solved per instance and round (0: default = 32768 / batch size, within 16..16384) */
  int max_open_nodes;       /* per-instance open-list capacity (0: default = 2^27 / batch size, within 32768..2^20, and at
                               most an eighth of the free device memory) */
  double gap_override;      /* < 0: use ModelParameters.relative_mip_gap_tolerance */
  int verbose;
} miqp_solver_opts;

/* CplexWrapper::CplexWrapper(...) / ~CplexWrapper()            src/cplex_wrapper.hpp:81-158 */
miqp_solver_t* miqp_solver_create(const miqp_solver_opts* opts);
void miqp_solver_destroy(miqp_solver_t* s);

/* CplexWrapper::resetParameters(shared_ptr<ModelParameters>)   src/cplex_wrapper.cpp:661-664
 * + ModelInputDataSource::read (rounding, edge tuples)         src/model_input_data_source.cpp:180-275
 * The arrays are copied; returns 0 on success, <0 on invalid sizes. */
int miqp_solver_set_params(miqp_solver_t* s, const miqp_model_params_c* p);

/* CplexWrapper::setParameterDatFileAbsolute + DATFILE source   src/cplex_wrapper.cpp:30-42
 * Reads the OPL .dat subset of the fixtures (no rounding, as OPL would). */
int miqp_solver_load_dat(miqp_solver_t* s, const char* path);

/* CplexWrapper::overrideSolverSettingsDataSource               src/cplex_wrapper.cpp:893-896 */
int miqp_solver_override_settings(miqp_solver_t* s, double max_solution_time, double relative_mip_gap_tolerance);

/* CplexWrapper::addRecedingHorizonWarmstart / setLastSolutionWarmstart   src/cplex_wrapper.cpp:459-483
 * + initializeWarmstart / addMIPStart / readMIPStarts                    src/cplex_wrapper.cpp:124-138, 494-639
 * The record is copied.  Two starts can be registered at once, as the reference does with
 * BOTH_WARMSTART_STRATEGIES: type MIQP_WARMSTART_LAST_SOLUTION fills the last-solution slot, every other type the
 * receding-horizon slot; MIQP_WARMSTART_NONE (or start == NULL) clears both.  Each start is tried as an initial
 * incumbent (CPLEX MIPStartSolveMIP: the binaries of the start are fixed, the continuous QP is solved on the device; an
 * infeasible start, or one whose seven sizes differ from the instance, is ignored).  Starts stay registered across
 * miqp_solver_set_params.  Returns 0, <0 when the record is refused. */
int miqp_solver_set_warmstart(miqp_solver_t* s, const miqp_raw_results_c* start, int warmstart_type);

/* CplexWrapper::callCplex(timestamp)                           src/cplex_wrapper.cpp:65-249
 * Returns OptimizationStatus (MIQP_STATUS_*); never throws.  Fails with MIQP_STATUS_FAILED_SEG_FAULT when
 * no HIP device / kernel image is available - there is no CPU fallback. */
int miqp_solver_solve(miqp_solver_t* s, double timestamp);

/* Batch of independent instances (receding-horizon steps / scenario seeds) solved concurrently on one
 * device; statuses[k] receives the OptimizationStatus of solver k.  All instances must share
 * NumCars, NumSteps, nr_regions, nr_environments, nr_obstacles, max_lines_obstacles and opts.device.
 * Returns 0 on success.  (MiqpPlanner issues one callCplex at a time, src/miqp_planner.cpp:731; the batch entry
 * is what a scenario-parallel caller binds.) */
int miqp_solver_solve_batch(miqp_solver_t* const* solvers, int n, int* statuses);

/* The same call as a QUEUE drained with at most `inflight` instances in flight on the device (<= 0 or >= n: all at once =
 * miqp_solver_solve_batch): an instance that is proven - or has used up its own max_solution_time, counted from its
 * admission - hands its slot (open lists in HBM) to the next instance of the queue at the following branch-and-bound
 * round, so the device never idles behind the hardest instances of a batch.  SolutionProperties.time of an instance is the
 * time from its admission to its proof.  (No reference counterpart: MiqpPlanner issues one callCplex at a time,
 * src/miqp_planner.cpp:731; this is the entry a scenario-parallel caller binds to keep one GPU saturated.)
 * The device context (node pool, lists) is kept between calls of the same shape - slots, model dimensions - whose queue is no
 * longer than the per-instance arrays of the context hold (64 x inflight at least); miqp_solver_last_setup says whether a call
 * had to build it.  A single solve (miqp_solver_solve) returns the same result bit for bit when repeated, as CPLEX's default
 * deterministic mode does; the instances of a queue influence each other's shares of a round, so their node counts move by a
 * few nodes between runs (every reported bound and solution is valid either way). */
int miqp_solver_solve_stream(miqp_solver_t* const* solvers, int n, int inflight, int* statuses);

/* The same batch sharded over the first `gpus` HIP devices of this process (<= 0: all visible): instance b runs on
 * device b mod gpus with one host thread per device and no exchange between the shards;
 * opts.device of every handle is set to the device it ran on. */
int miqp_solver_solve_batch_multi(miqp_solver_t* const* solvers, int n, int gpus, int* statuses);

/* ---- one hard instance split over the ranks of a job (the reference walks its alternative
 * start configurations one after the other on one CPU, src/miqp_planner.cpp:694-749) ----
 * Every rank loads the SAME instance into its own handle and calls miqp_solver_solve_split with its rank.  The alternatives
 * of one or two car/car disjunctions partition the branch-and-bound tree over the ranks; once per round the ranks run
 * `exchange(user, 0, words, 4, 0)` = in-place all-reduce(min) over `count` unsigned 64-bit words (incumbent | owner rank,
 * lower bound, done, time-up: the best incumbent prunes everywhere, every rank takes the same stop decision), and at the
 * end `exchange(user, 1, buf, nbytes, root)` = broadcast of the owner's solution.  The callback returns 0 on success.
 * Returns the OptimizationStatus, identical on every rank, and every rank holds the full result. */
typedef int (*miqp_exchange_fn)(void* user, int op, void* buf, int count, int root);
int miqp_solver_solve_split(miqp_solver_t* s, double timestamp, int world, int rank, miqp_exchange_fn exchange, void* user);

/* Built-in transport of that exchange: RCCL (librccl is loaded at run time) over xGMI, one communicator per process.
 * Rank 0 creates the 128-byte id (miqp_comm_unique_id) and hands it to the others by any means (e.g. the launcher's
 * store); every rank then calls miqp_comm_init with the device it solves on.  miqp_solver_solve_split_rccl = solve_split
 * with ncclAllReduce(ncclUint64, ncclMin) / ncclBroadcast on the solver's stream. */
int miqp_comm_unique_id(char* out128);
int miqp_comm_init(int world, int rank, const char* id128, int device);
int miqp_comm_finalize(void);
int miqp_solver_solve_split_rccl(miqp_solver_t* s, double timestamp);
/* checks a transport against the contract above (min over unsigned words incl. values with the top bit set, broadcast from
 * every root); exchange == NULL: the RCCL communicator of miqp_comm_init.  0 when it conforms. */
int miqp_comm_selftest(miqp_exchange_fn exchange, void* user, int world, int rank);

/* the roots rank `rank` of `world` starts from in a tree split of the loaded instance (no device needed): writes up to `cap`
 * entries of (record index, alternative) pairs flattened as root_of_pair[k], index[k], value[k]; returns the number of pairs,
 * *nroots_out = the number of roots of this rank, *ncombos_out = the size of the partition */
int miqp_solver_split_roots(const miqp_solver_t* s, int world, int rank, int* root_of_pair, int* index, int* value, int cap, int* nroots_out, int* ncombos_out);

/* cplex.getNrows / getNbinVars / getNcols - getNbinVars / getNNZs of the model OPL would generate for the loaded
 * instance (collectCplexStatistics, src/cplex_wrapper.cpp:679-690): out[0..3] = rows, binary columns, continuous
 * columns, non-zeros.  Needs no device. */
int miqp_solver_raw_sizes(const miqp_solver_t* s, int* out4);
/* Diagnostic (no reference counterpart, no device needed): the response tables behind the bound lifting of the branch and
   bound - for every (car, axis, step) the 4 x 4 matrix Zu H^-1 Zu' of the chain's (position, velocity,
   acceleration, jerk) at that step under the objective's Hessian; out[((car * 2 + axis) * N + step) * 16 + 4 a + b].
   Returns the number of doubles written, < 0 on error (-3: cap too small). */
int miqp_solver_lift_tables(const miqp_solver_t* s, double* out, int cap);

/* CplexWrapper::getRawResults()                                src/cplex_wrapper.hpp:204, cpp:311-448
 * Fills the caller-allocated record (sizes must match the instance). */
int miqp_solver_get_results(const miqp_solver_t* s, miqp_raw_results_c* out);

/* collectRawResults for a whole batch                           src/cplex_wrapper.cpp:186 (collectRawResults() runs inside callCplex, before it returns)
 * Builds the RawResults record of every handle that holds a solution, on `threads` host threads (0: up to 32 - more contend for the allocator), and
 * keeps it inside the handle; miqp_solver_get_results then only copies.  The reference produces the record inside callCplex;
 * the batch entry points leave it to this call so that a service can overlap it - bench.py calls it inside its timed region.
 * Returns the number of records built, < 0 on error. */
int miqp_solver_materialize_results(miqp_solver_t* const* solvers, int n, int threads);

/* CplexWrapper::getSolutionProperties()                        src/cplex_wrapper.hpp:206-208, cpp:672-690 */
int miqp_solver_get_properties(const miqp_solver_t* s, miqp_solution_properties_c* out);

/* sizes of the loaded instance: out[0..5] = NumCars, NumSteps, nr_regions, nr_environments,
 * nr_obstacles, max_lines_obstacles                            (collectRawResults, cpp:314-327) */
int miqp_solver_get_dims(const miqp_solver_t* s, int* out6);

/* cplex.exportModel(".lp") debug output                        src/cplex_wrapper.cpp:150-154 */
int miqp_solver_export_lp(const miqp_solver_t* s, const char* path);

/* opl.printExternalData(): the loaded parameters as OPL external data (.dat syntax, re-readable by
 * miqp_solver_load_dat and by OPL)                              src/cplex_wrapper.cpp:141-149
 * (test_hardcoded_data_versus_datfile, test/cplex_wrapper_test.cc:474-505) */
int miqp_solver_write_dat(const miqp_solver_t* s, const char* path);

/* opl.printSolution(): every decision variable of the last solution as `name = [...]` blocks
 * (layout of cplexmodel/modelRun.txt)                           src/cplex_wrapper.cpp:212-219 */
int miqp_solver_write_solution(const miqp_solver_t* s, const char* path);

/* cplex.writeMIPStarts(): the last solution as a CPLEX MIP start (.mst XML, variable names of the LP export)
 *                                                               src/cplex_wrapper.cpp:206-209, 221-228 */
int miqp_solver_write_mst(const miqp_solver_t* s, const char* path);

/* cplex.readMIPStarts(): loads an .mst written by miqp_solver_write_mst (or by CPLEX for the exported .lp) as the
 * MIP start of the next solve; returns 0 when the start was accepted  src/cplex_wrapper.cpp:128-138 */
int miqp_solver_read_mst(miqp_solver_t* s, const char* path);

/* continuous QP with the binaries of `fixed` asserted, solved by the device interior-point kernel
 * (used by the parity tests to compare the QP machinery with the CPU oracle and with K3);
 * returns 0 when feasible */
int miqp_solver_solve_fixed(miqp_solver_t* s, const miqp_raw_results_c* fixed, miqp_raw_results_c* out,
                            double* objective, int* iterations);

/* timing of the last solve / batch in seconds, measured with HIP events on the solver stream:
 * out[0] = whole solve, out[1] = interior-point kernel total, out[2] = number of IPM kernel launches,
 * out[3] = node relaxations solved, out[4] = IPM iterations (summed over nodes), out[5] = rows x iterations */
int miqp_solver_last_timing(const miqp_solver_t* s, double* out6);

/* the dual active-set launches of the last solve / batch (two cars: the node relaxations of a round; the reference's counterpart is CPLEX's dual
 * simplex re-solve of a child node inside cplex.solve(), src/cplex_wrapper.cpp:158-185): out[0] = node relaxations they solved,
 * out[1] = their steps (rows added + rows dropped; they are part of out[4] of miqp_solver_last_timing and of NrIterations),
 * out[2] = nodes they could not finish (returned unsolved, solved by the interior point a round later), out[3] = rows dropped,
 * out[4] = sum over the nodes of the active rows at the end, out[5] = sum of the rows taken over from the parents' active sets,
 * out[6] = seconds of the STANDARD active-set launches alone (HIP events on the solver stream: from the start of a round's launch group
 * to the end of that kernel; the other three launches of the group run beside it on their own streams), out[7] = number of those launches */
int miqp_solver_last_active_set(const miqp_solver_t* s, double* out8);

/* host set-up of the last solve / batch / stream call this handle took part in: out[0] = seconds from the entry of the call to
 * the first round, out[1] = of which building the device context (pools, lists: reused by a call of the same shape with no more
 * instances than its per-instance arrays hold), out[2] = 1 when the context was built or rebuilt by that call, else 0 */
int miqp_solver_last_setup(const miqp_solver_t* s, double* out3);

/* out[0] = when the last batch / stream call admitted this instance to a slot, in seconds after the call's first branch-and-bound round started
 * (0 for a single solve and for the instances in flight from the start).  Together with SolutionProperties.time (admission to proof) it places
 * every instance of a drained queue on the call's time axis - bench.py derives the drain rate of the queue while it still had a backlog from it. */
int miqp_solver_last_admission(const miqp_solver_t* s, double* out1);

/* why the last solve of this handle did not run or did not finish, as text ("" when there is nothing to say; the pointer is valid
 * until the next call on the handle).  The reference logs such conditions with LOG(ERROR) inside callCplex
 * (src/cplex_wrapper.cpp:97-109, 162-180); here the status code says WHAT (the four OptimizationStatus values), this says WHY:
 * malformed parameters, the queue abandoned before the instance was admitted, an instance retired because it made no progress
 * for 64 branch-and-bound rounds.  Such an instance reports FAILED_SEG_FAULT without an incumbent (never FAILED_TIMEOUT) and SUCCESS with
 * one; its props.status stays the CPLEX code of an unfinished solve (107 with, 108 without an incumbent - CPLEX has no code for "stalled"):
 * what tells it from an instance that really used up max_solution_time is this text and, without an incumbent, the status code. */
const char* miqp_solver_last_error(const miqp_solver_t* s);

/* ---- planner core: the host logic directly above the solve ---- */

/* ParameterPreparer::CalculateFractionParameters             common/parameter/parameter_preparer.cpp:37-52; out[R*4] */
int miqp_fraction_parameters(int nr_regions, float max_velocity_fitting, double* out);
/* FittingPolynomialParameters::GetPOLY_*()                    common/parameter/fitting_polynomial_parameters.hpp:28-168
 * out[6][R*3], row-major [region][3], order SINT_UB, SINT_LB, COSS_UB, COSS_LB, KAPPA_AX_MAX, KAPPA_AX_MIN; returns -2 for a
 * (nr_regions, max, min velocity) combination outside the seven the reference ships (it throws std::invalid_argument) */
int miqp_fitting_polynomial_parameters(int nr_regions, float max_velocity_fitting, float min_velocity_fitting, double* out);
/* ParameterPreparer::CalculateMeanAngleVector                 parameter_preparer.cpp:96-113; out[R] */
int miqp_mean_angles(const double* fraction_parameters, int nr_regions, double* out);
/* CalculateAccLimitsPerCar / CalculateJerkLimitsPerCar (RotateLimitVectors)   parameter_preparer.cpp:54-94, 115-143
 * acc: (acc_min, acc_max, -acc_lat, +acc_lat); jerk: (-jerk_max, jerk_max, -jerk_lat, +jerk_lat); four arrays of R */
int miqp_limits_per_region(const double* fraction_parameters, int nr_regions, float long_min, float long_max, float lat_min, float lat_max,
                           double* min_x, double* max_x, double* min_y, double* max_y);
/* CalculateRegionIdx                                          common/parameter/regions.cpp:16-33; returns the count */
int miqp_calculate_region_idx(const double* fraction_parameters, int nr_regions, float vx, float vy, int* out);
/* ReserveNeighborRegions on one row of nr_regions flags       regions.cpp:75-112 */
int miqp_reserve_neighbor_regions(int* row, int nr_regions, int expansions);
/* CalculatePossibleRegions                                    regions.cpp:114-127; flags[R] */
int miqp_calculate_possible_regions(const double* fraction_parameters, int nr_regions, const double* theta_ref, int n, int* flags);
/* MiqpPlanner::CalculateWarmstart: last solution shifted by one step (with its quirks)   src/miqp_planner.cpp:787-1051 */
/* ReferenceTrajectoryGenerator::GenerateTrajectory on a polyline (common/reference/reference_trajectory_generator.cpp:51-148;
   bark's spline smoothing of the centre line replaced by the polyline itself: identical on straight reference lines).
   ref_xy: n_ref points (x, y); state5 = (time, x, y, theta, v); out[num_points][5] in the same order. */
int miqp_reference_trajectory(const double* ref_xy, int n_ref, const double* state5, double dt, int num_points, double line_interp_inc, double vel_desired,
                              double delta_s_desired, double acc_lat_max, int vel_curve_dep, double* out);
/* MiqpPlanner::UpdateCar for one car (src/miqp_planner.cpp:284-390): reference rows, possible regions, weights.
   settings12 = nr_regions, nr_steps, nr_neighbouring_possible_regions, additionalStepsForReferenceLongerHorizon, ts,
   refLineInterpInc, straight lateral acceleration limit, lambda, positionWeight, velocityWeight, acclerationWeight, jerkWeight;
   initial_state6 = (x, vx, ax, y, vy, ay); ref4N = x_ref, y_ref, vx_ref, vy_ref rows; weights8 = POS_X, VEL_X, ACC_X, POS_Y,
   VEL_Y, ACC_Y, JERK_X, JERK_Y.  0 ok, 1 region expansion failed (the reference logs and goes on), < 0 invalid arguments. */
int miqp_update_car(const double* settings12, const double* fraction_parameters, const double* initial_state6, const double* ref_xy, int n_ref, double desired_velocity,
                    double delta_s_desired, double timestep, int track_reference_positions, int is_ego, int num_cars, double* ref4N, int* possible_region, double* weights8);
int miqp_calculate_warmstart(const miqp_raw_results_c* last, miqp_raw_results_c* out, double ts, double minimum_region_change_speed);
/* MiqpPlanner::Plan, region-combination retry loop            src/miqp_planner.cpp:634-645, 692-766
 * initial_region[C] / possible_region[C*R] are the caller's arrays (p is re-pointed to them) and are updated in place as
 * the reference updates its ModelParameters; returns 1 when a combination solved (Plan() == true),
 * *status_out = last OptimizationStatus */
int miqp_plan(miqp_solver_t* s, miqp_model_params_c* p, int* initial_region, int* possible_region, const miqp_raw_results_c* warmstart,
              int warmstart_type, double timestamp, int* status_out);

/* ---- environment and obstacles of MiqpPlanner, on convex counter-clockwise pieces given as vertex arrays (the convexification of a
 * bark map polygon, common/map/convexified_map.cpp, is out of scope).  pieces_xy: x0, y0, x1, y1, ... of all pieces, piece_off[n + 1]
 * vertex offsets ---- */
/* the initial-pose check of MiqpPlanner::Plan (src/miqp_planner.cpp:654-685): -1 when the rear and the front axle point of every car
 * lie within a piece of p's environment, else 2 * car + (1: front point, 0: rear point); -2 on invalid arguments.  miqp_plan runs it. */
int miqp_initial_pose_check(const miqp_model_params_c* p);
/* MiqpPlanner::ResetEnvironment (src/miqp_planner.cpp:490-537): selected[e] = 1 for the pieces that a reference trajectory (x, y
 * polylines, traj_off[n_traj + 1] point offsets) touches; returns their number */
int miqp_select_environment(const double* pieces_xy, const int* piece_off, int n_pieces, const double* traj_xy, const int* traj_off, int n_traj, int* selected);
/* MiqpPlanner::ObstacleIntersectsEnvironment (src/miqp_planner.cpp:1248-1306, without the region of interest): 1 when the obstacle
 * (n_steps x 4 vertices) intersects a piece - checked at step 0 only when it is static; an empty environment admits every obstacle */
int miqp_obstacle_intersects_environment(const double* pieces_xy, const int* piece_off, int n_pieces, const double* obstacle_xy, int n_steps, int is_static);
/* MiqpPlanner::GetBarkTrajectory (src/miqp_planner.cpp:1132-1170) on plain arrays: out_rows5 receives up to N rows (time, x, y, theta, v)
 * - bark's StateDefinition order - of car `car`; the trajectory is cut off at the first step whose |vx| and |vy| are both <= min_speed
 * (IsVxVyValid :1184-1187; the planner uses 0.7, :53-54).  Returns the number of rows, -1 on invalid arguments */
int miqp_bark_trajectory(const miqp_raw_results_c* results, int car, double start_time, double ts, double min_speed, double* out_rows5);
/* MiqpPlanner::UpdateObstaclesROI (src/miqp_planner.cpp:1308-1335): the region of interest around the ego car as 4 vertices (x, y pairs:
 * front upper, front lower, rear lower, rear upper), computed exactly as the reference writes it */
int miqp_obstacles_roi(double x, double y, double theta, double behind_distance, double front_distance, double side_distance, double* roi_xy);
/* MiqpPlanner::ObstacleIntersectsEnvironment with its region-of-interest filter (src/miqp_planner.cpp:1278-1288; settings
 * obstacle_roi_filter, src/miqp_planner_settings.h:74-77): as above, but a step at which the obstacle does not intersect roi_xy (4 vertices;
 * NULL = no filter) is skipped - a static obstacle is then irrelevant (0), a moving one is checked at its next step */
int miqp_obstacle_intersects_environment_roi(const double* pieces_xy, const int* piece_off, int n_pieces, const double* obstacle_xy, int n_steps, int is_static, const double* roi_xy);
/* MiqpPlanner::EnvironmentWarmstart (src/miqp_planner.cpp:1053-1115): the five environment arrays of `last` ([C][n_old][N]) re-indexed
 * into `out` ([C][n_new][N]) by piece id; new pieces and the last step start as 1 */
int miqp_environment_warmstart(const miqp_raw_results_c* last, miqp_raw_results_c* out, const int* ids_old, int n_old, const int* ids_new, int n_new);

const char* miqp_gpu_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MIQP_GPU_H */
