/*
 * miqp_types.h - plain-C mirrors of the reference's data contract for the MIQP solve path.
 *
 * Replaces (reference paths relative to the planner-miqp checkout):
 *   struct ModelParameters   src/miqp_planner_data.hpp:99-185   -> miqp_model_params_c
 *   struct RawResults        src/miqp_planner_data.hpp:46-97    -> miqp_raw_results_c
 *   struct SolutionProperties src/cplex_wrapper.hpp:41-52       -> miqp_solution_properties_c
 *   enum OptimizationStatus  src/cplex_wrapper.hpp:54-59        -> MIQP_STATUS_*
 *   enum MiqpPlannerWarmstartType src/miqp_planner_settings.h:13-18 -> MIQP_WARMSTART_*
 *
 * Layout rule: every array is flat, row-major in the logical index order of the reference
 * member (Eigen members are column-major; the C++ adapter include/cplex_wrapper.hpp converts).
 *   x_ref[c*N+i], min_acc_x[c*R+j], possible_region[c*R+j],
 *   fraction_parameters[j*4+k], POLY_*[j*3+k],
 *   obstacle vertices [((o*N+i)*L+k)*2+{0,1}]  (index order [obstacle][time], parameters.mod:114),
 *   environment polygons: ragged, env_offsets[E+1] vertex offsets into env_vertices[2*total].
 * Polygons are counter-clockwise vertex lists (common/geometry/geometry.cpp:126-139); the edge
 * tuples <k,x1,y1,x2,y2> of the OPL model are built with wrap-around of the last vertex exactly as
 * ModelInputDataSource::addLineSet does (src/model_input_data_source.cpp:167-178).
 */
#ifndef MIQP_TYPES_H
#define MIQP_TYPES_H

#ifdef __cplusplus
extern "C" {
#endif

enum {
  MIQP_STATUS_SUCCESS = 0,
  MIQP_STATUS_FAILED_NO_SOLUT = 1,
  MIQP_STATUS_FAILED_SEG_FAULT = 2,
  MIQP_STATUS_FAILED_TIMEOUT = 3
};

enum {
  MIQP_WARMSTART_NONE = 0,
  MIQP_WARMSTART_RECEDING_HORIZON = 1,
  MIQP_WARMSTART_LAST_SOLUTION = 2,
  MIQP_WARMSTART_BOTH = 3
};

/* CPLEX status integers reported in SolutionProperties.status (src/cplex_wrapper.cpp:672-677) */
enum {
  MIQP_CPX_STAT_OPTIMAL = 101,
  MIQP_CPX_STAT_OPTIMAL_TOL = 102,
  MIQP_CPX_STAT_INFEASIBLE = 103,
  MIQP_CPX_STAT_TIME_LIM_FEAS = 107,
  MIQP_CPX_STAT_TIME_LIM_INFEAS = 108
};

typedef struct miqp_model_params_c {
  /* solver parameters (cplexmodel/cplexmodel.mod:8-21); the CPLEX-specific knobs are accepted and ignored */
  double max_solution_time;
  double relative_mip_gap_tolerance;
  int mipdisplay, mipemphasis;
  double relobjdif;
  int cutpass, probe, repairtries, rinsheur, varsel, mircuts, parallelmode;
  /* sizes */
  int NumSteps;            /* N */
  int nr_regions;          /* R */
  int NumCars;             /* C */
  int nr_obstacles;        /* O */
  int max_lines_obstacles; /* L, every obstacle polygon has exactly L vertices */
  int nr_environments;     /* E */
  /* scalars */
  double ts;
  double min_vel_x_y, max_vel_x_y;
  double total_min_acc, total_max_acc, total_min_jerk, total_max_jerk;
  double maximum_slack;
  double WEIGHTS_SLACK, WEIGHTS_SLACK_OBSTACLE;
  double minimum_region_change_speed;
  /* [N] */
  const double* agent_safety_distance;
  const double* agent_safety_distance_slack;
  /* [C] */
  const double* WEIGHTS_POS_X; const double* WEIGHTS_VEL_X; const double* WEIGHTS_ACC_X;
  const double* WEIGHTS_POS_Y; const double* WEIGHTS_VEL_Y; const double* WEIGHTS_ACC_Y;
  const double* WEIGHTS_JERK_X; const double* WEIGHTS_JERK_Y;
  const double* WheelBase; const double* CollisionRadius;
  /* [C*6]  x,vx,ax,y,vy,ay (InitialStateIndices, miqp_planner_data.hpp:34-42) */
  const double* IntitialState;
  /* [C*N] */
  const double* x_ref; const double* vx_ref; const double* y_ref; const double* vy_ref;
  /* [C*R] */
  const double* min_acc_x; const double* max_acc_x; const double* min_acc_y; const double* max_acc_y;
  const double* min_jerk_x; const double* max_jerk_x; const double* min_jerk_y; const double* max_jerk_y;
  const int* initial_region;  /* [C], 1-based (miqp_planner.cpp:696-700) */
  const int* possible_region; /* [C*R] 0/1 */
  /* obstacles */
  const double* obstacle_vertices; /* [O*N*L*2] */
  const int* obstacle_is_soft;     /* [O] */
  /* environments */
  const int* env_offsets;          /* [E+1] */
  const double* env_vertices;      /* [2*env_offsets[E]] */
  /* [R*4], [R*3] */
  const double* fraction_parameters;
  const double* POLY_SINT_UB; const double* POLY_SINT_LB;
  const double* POLY_COSS_UB; const double* POLY_COSS_LB;
  const double* POLY_KAPPA_AX_MAX; const double* POLY_KAPPA_AX_MIN;
} miqp_model_params_c;

/* Caller-allocated result record; index order identical to RawResults (row-major here).
 * K = NumCars-1.  Entries the model does not define keep the reference's fill value 9999999
 * (src/cplex_wrapper.cpp:259).  The continuous slacks are truncated to int exactly as the
 * reference does (miqp_planner_data.hpp:84-88); the untruncated values are in the *_real arrays
 * (optional, may be NULL). */
typedef struct miqp_raw_results_c {
  int N, NrEnvironments, NrRegions, NrObstacles, MaxLinesObstacles, NrCarToCarCollisions, NrCars;
  double* u_x; double* u_y; double* pos_x; double* vel_x; double* acc_x; double* pos_y; double* vel_y; double* acc_y;
  double* pos_x_front_UB; double* pos_x_front_LB; double* pos_y_front_UB; double* pos_y_front_LB; /* each [C*N] */
  int* notWithinEnvironmentRear; int* notWithinEnvironmentFrontUbUb; int* notWithinEnvironmentFrontLbUb;
  int* notWithinEnvironmentFrontUbLb; int* notWithinEnvironmentFrontLbLb;  /* each [C*E*N] */
  int* active_region;                                                       /* [C*N*R] */
  int* region_change_not_allowed_x_positive; int* region_change_not_allowed_y_positive;
  int* region_change_not_allowed_x_negative; int* region_change_not_allowed_y_negative;
  int* region_change_not_allowed_combined;                                  /* each [C*N] */
  int* deltacc;                 /* [C*O*N*L] */
  int* deltacc_front;           /* [C*O*N*L*4] */
  int* car2car_collision;       /* [K*K*N*16] */
  int* slackvars;               /* [K*K*N*4] */
  int* slackvarsObstacle;       /* [C*O*N] */
  int* slackvarsObstacle_front; /* [C*O*N*4] */
  double* slackvars_real;       /* [K*K*N*4] or NULL */
} miqp_raw_results_c;

typedef struct miqp_solution_properties_c {
  int status;        /* CPLEX-style status integer, MIQP_CPX_STAT_* */
  double gap;        /* |best - inc| / (1e-10 + |inc|), NaN on failure */
  double objective;  /* NaN on failure */
  double time;       /* seconds spent in the solve only (src/cplex_wrapper.cpp:158-185) */
  int NrConstraints, NrBinaryVariables, NrFloatVariables, NonZeroCoefficients; /* of the raw OPL model */
  int NrIterations;  /* interior-point iterations summed over all nodes */
  int NrSolutionPool; /* number of incumbents found */
  /* extras (not in the reference struct) */
  double best_bound;
  long long nodes;
} miqp_solution_properties_c;

#ifdef __cplusplus
}
#endif
#endif /* MIQP_TYPES_H */
