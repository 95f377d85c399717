// Test program (never shipped): compiles include/cplex_wrapper.hpp against the stand-in Eigen / data-contract headers of
// this directory and drives the adapter the way MiqpPlanner does (resetParameters / callCplex / getRawResults /
// getSolutionProperties, src/miqp_planner.cpp:722-759).
//   adapter_check dat <file.dat>        DATFILE source
//   adapter_check cpp <params.txt>      CPPINPUTS source; params.txt: lines "name rows cols v...", polygons as "env n x y ..."
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>

#include "cplex_wrapper.hpp"

using namespace miqp::planner;
using miqp::planner::cplex::CplexWrapper;

static Eigen::MatrixXd mat(const std::vector<double>& v, int r, int c) { Eigen::MatrixXd m(r, c); for (int i = 0; i < r; ++i) for (int j = 0; j < c; ++j) m(i, j) = v[(size_t)i * c + j]; return m; }

int main(int argc, char** argv) {
  if (argc < 3) return 64;
  CplexWrapper spare;                       // default construction must not touch files or the GPU (behavior_miqp_agent.cpp:49)
  CplexWrapper copy(spare);                 // BARK clones agents freely (copy constructor)
  (void)copy;
  std::shared_ptr<RawResults> res; cplex::SolutionProperties pr{}; int st = -1;
  if (!std::strcmp(argv[1], "dat")) {
    CplexWrapper cw("cplexmodel.mod", CplexWrapper::DATFILE, 12);
    cw.setParameterDatFileAbsolute(argv[2]);
    st = cw.callCplex(); pr = cw.getSolutionProperties(); res = cw.getRawResults();
    std::printf("status %d objective %.17g gap %.17g nnz %d\n", st, pr.objective, pr.gap, pr.NonZeroCoefficients);
  } else {
    std::ifstream f(argv[2]); std::string line; std::map<std::string, std::vector<double>> V; std::map<std::string, std::pair<int, int>> S;
    auto P = std::make_shared<ModelParameters>();
    while (std::getline(f, line)) {
      std::istringstream is(line); std::string nm; int r, c; is >> nm >> r >> c; std::vector<double> v((size_t)r * c); for (auto& x : v) is >> x;
      if (nm == "env") P->MultiEnvironmentConvexPolygon.push_back(mat(v, r, c));
      else if (nm == "obs_new") P->ObstacleConvexPolygon.emplace_back();
      else if (nm == "obs") P->ObstacleConvexPolygon.back().push_back(mat(v, r, c));
      else { V[nm] = v; S[nm] = {r, c}; }
    }
    auto s = [&](const char* n) { return V.at(n)[0]; };
    auto m = [&](const char* n) { return mat(V.at(n), S.at(n).first, S.at(n).second); };
    auto mi = [&](const char* n) { Eigen::MatrixXi o(S.at(n).first, S.at(n).second); for (int i = 0; i < o.rows(); ++i) for (int j = 0; j < o.cols(); ++j) o(i, j) = (int)V.at(n)[(size_t)i * o.cols() + j]; return o; };
    P->max_solution_time = (float)s("max_solution_time"); P->relative_mip_gap_tolerance = (float)s("relative_mip_gap_tolerance");
    P->NumSteps = (int)s("NumSteps"); P->ts = (float)s("ts"); P->nr_regions = (int)s("nr_regions"); P->NumCars = (int)s("NumCars");
    P->min_vel_x_y = (float)s("min_vel_x_y"); P->max_vel_x_y = (float)s("max_vel_x_y"); P->total_min_acc = (float)s("total_min_acc"); P->total_max_acc = (float)s("total_max_acc");
    P->total_min_jerk = (float)s("total_min_jerk"); P->total_max_jerk = (float)s("total_max_jerk"); P->maximum_slack = (float)s("maximum_slack");
    P->WEIGHTS_SLACK = (float)s("WEIGHTS_SLACK"); P->WEIGHTS_SLACK_OBSTACLE = (float)s("WEIGHTS_SLACK_OBSTACLE"); P->minimum_region_change_speed = (float)s("minimum_region_change_speed");
    P->agent_safety_distance = m("agent_safety_distance"); P->agent_safety_distance_slack = m("agent_safety_distance_slack");
    P->WEIGHTS_POS_X = m("WEIGHTS_POS_X"); P->WEIGHTS_VEL_X = m("WEIGHTS_VEL_X"); P->WEIGHTS_ACC_X = m("WEIGHTS_ACC_X"); P->WEIGHTS_POS_Y = m("WEIGHTS_POS_Y");
    P->WEIGHTS_VEL_Y = m("WEIGHTS_VEL_Y"); P->WEIGHTS_ACC_Y = m("WEIGHTS_ACC_Y"); P->WEIGHTS_JERK_X = m("WEIGHTS_JERK_X"); P->WEIGHTS_JERK_Y = m("WEIGHTS_JERK_Y");
    P->WheelBase = m("WheelBase"); P->CollisionRadius = m("CollisionRadius"); P->IntitialState = m("IntitialState");
    P->x_ref = m("x_ref"); P->vx_ref = m("vx_ref"); P->y_ref = m("y_ref"); P->vy_ref = m("vy_ref");
    P->acc_limit_params = {m("min_acc_x"), m("max_acc_x"), m("min_acc_y"), m("max_acc_y")};
    P->jerk_limit_params = {m("min_jerk_x"), m("max_jerk_x"), m("min_jerk_y"), m("max_jerk_y")};
    P->initial_region = mi("initial_region"); P->possible_region = mi("possible_region");
    P->nr_obstacles = (int)s("nr_obstacles"); P->max_lines_obstacles = (int)s("max_lines_obstacles"); P->nr_environments = (int)s("nr_environments");
    for (int o = 0; o < P->nr_obstacles; ++o) P->obstacle_is_soft.push_back((int)V.at("obstacle_is_soft")[o]);
    P->fraction_parameters = m("fraction_parameters");
    P->poly_curvature_params = {m("POLY_KAPPA_AX_MAX"), m("POLY_KAPPA_AX_MIN")};
    P->poly_orientation_params = {m("POLY_SINT_UB"), m("POLY_SINT_LB"), m("POLY_COSS_UB"), m("POLY_COSS_LB")};
    CplexWrapper cw("cplexmodel.mod", CplexWrapper::CPPINPUTS, 12);
    cw.resetParameters(P);
    st = cw.callCplex(); pr = cw.getSolutionProperties(); res = cw.getRawResults();
    std::printf("status %d objective %.17g gap %.17g nnz %d\n", st, pr.objective, pr.gap, pr.NonZeroCoefficients);
    {
      // the copy of a configured wrapper (MiqpPlanner's copy constructor, src/miqp_planner.cpp:153-171) carries the configuration
      // but no parameters: it cannot solve before resetParameters, and solves the same instance after it
      cw.setBranchingPriorityValueExtent(3, 7); cw.setUseBranchingPriorities(true);
      CplexWrapper cp(cw);
      int stc0 = cp.callCplex();
      cp.resetParameters(P);
      int stc1 = cp.callCplex();
      std::printf("copy before %d after %d objective %.17g tmpfile_equal %d\n", stc0, stc1, cp.getSolutionProperties().objective,
                  (int)(cp.getTmpWarmstartFile() == cw.getTmpWarmstartFile()));
    }
    if (st == cplex::SUCCESS) {
      // the solution fed back as receding-horizon start must be accepted (not worse)
      cw.addRecedingHorizonWarmstart(std::make_shared<RawResults>(*res));
      int st2 = cw.callCplex();
      std::printf("warm status %d objective %.17g\n", st2, cw.getSolutionProperties().objective);
    }
  }
  if (st == cplex::SUCCESS && res) {
    std::printf("pos_x");
    for (int c = 0; c < res->NrCars; ++c) for (int i = 0; i < res->N; ++i) std::printf(" %.17g", res->pos_x({c, i}));
    std::printf("\nregion");
    for (int c = 0; c < res->NrCars; ++c) for (int i = 0; i < res->N; ++i) { int a = -1; for (int j = 0; j < res->NrRegions; ++j) if (res->active_region({c, i, j}) == 1) a = j; std::printf(" %d", a); }
    std::printf("\n");
  }
  return 0;
}
