// TEST STAND-IN (never shipped) of the two enums include/cplex_wrapper.hpp takes from the reference's
// src/miqp_planner_settings.h:13-26 (names and values only).
#ifndef MIQP_PLANNER_SETTINGS_HEADER
#define MIQP_PLANNER_SETTINGS_HEADER
enum MiqpPlannerWarmstartType { NO_WARMSTART = 0, RECEDING_HORIZON_WARMSTART = 1, LAST_SOLUTION_WARMSTART = 2, BOTH_WARMSTART_STRATEGIES = 3 };
enum MiqpPlannerParallelMode { DETERMINISTIC = 0, AUTO = 1, OPPORTUNISTIC = -1 };
#endif
