// cplex_wrapper.hpp - drop-in replacement of src/cplex_wrapper.{hpp,cpp} + src/model_input_data_source.{hpp,cpp}
// of bark-simulator/planner-miqp: same namespace, class name and public methods (src/cplex_wrapper.hpp:61-275),
// implemented over the C ABI of libmiqp_gpu.so (include/miqp_gpu.h).  Header-only; it needs Eigen and the
// reference's own miqp_planner_data.hpp, so it is compiled only inside a reference checkout.
// It does NOT include ilopl/iloopl.h - miqp_planner.hpp:14 pulls this header into every caller.
#ifndef CPLEX_WRAPPER_HEADER
#define CPLEX_WRAPPER_HEADER

#include <array>
#include <cmath>
#include <cstdio>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include <Eigen/Dense>
#include <unsupported/Eigen/CXX11/Tensor>

#include "miqp_gpu.h"
#include "miqp_planner_data.hpp"
#include "miqp_planner_settings.h"

namespace miqp {
namespace planner {
namespace cplex {

class NotImplementedException : public std::logic_error {  // src/cplex_wrapper.hpp:35-39
 public:
  NotImplementedException() : std::logic_error{"Function not yet implemented."} {}
};

struct SolutionProperties {  // src/cplex_wrapper.hpp:41-52
  int status;
  double gap;
  double objective;
  double time;
  int NrConstraints;
  int NrBinaryVariables;
  int NrFloatVariables;
  int NonZeroCoefficients;
  int NrIterations;
  int NrSolutionPool;
};

enum OptimizationStatus { SUCCESS = 0, FAILED_NO_SOLUT = 1, FAILED_SEG_FAULT = 2, FAILED_TIMEOUT = 3 };

class CplexWrapper {
 public:
  enum ParameterSource { DATFILE = 0, CPPINPUTS = 1, MIXED = 2 };
  typedef MiqpPlannerWarmstartType WarmstartType;
  typedef MiqpPlannerParallelMode ParallelMode;

  CplexWrapper(std::string modpath, std::string modfile, ParameterSource parameterSource, const int precision)
      : modfile_(modpath + modfile), precision_(precision), roundingDecimals_(precision - 2), parameterSource_(parameterSource),
        rawResults_(std::make_shared<RawResults>()), warmstartValues_(std::make_shared<RawResults>()) {}
  CplexWrapper(std::string modfile, ParameterSource parameterSource, const int precision)
      : CplexWrapper("cplexmodel/", modfile, parameterSource, precision) {}
  CplexWrapper(std::string modfile, const int precision) : CplexWrapper(modfile, ParameterSource::DATFILE, precision) {}
  // cheap and side-effect free: BehaviorMiqpAgent default-constructs a spare wrapper per agent
  // (behavior_miqp_agent.cpp:49); the device is touched on the first callCplex only
  CplexWrapper() : CplexWrapper("", ParameterSource::CPPINPUTS, 12) {}
  // the reference's copy (src/cplex_wrapper.hpp:116-151) takes over the configuration - file names, precision, parameter
  // source, debug output settings, warm start type, the solution properties, the .mst path, the branching priority settings and
  // the output buffering flag - and starts with a fresh environment: no parameters (ds_ is new; MiqpPlanner's copy constructor
  // calls resetParameters right after, src/miqp_planner.cpp:170), no results, no MIP start values, SOS switch at its default.
  // Its data source is built with cp2.precision_ where the constructors use precision - 2 (:129 against :88): a copy rounds
  // its inputs to `precision` decimals.
  CplexWrapper(const CplexWrapper& o)
      : tmpWarmstartFile_(o.tmpWarmstartFile_), modfile_(o.modfile_), datfile_(o.datfile_), precision_(o.precision_),
        roundingDecimals_(o.precision_), parameterSource_(o.parameterSource_),
        rawResults_(std::make_shared<RawResults>()), print_debug_outputs_(o.print_debug_outputs_), doWarmstart_(o.doWarmstart_),
        warmstartValues_(std::make_shared<RawResults>()), solutionProperties_(o.solutionProperties_),
        debugOutputFilePath_(o.debugOutputFilePath_), debugOutputFilePrefix_(o.debugOutputFilePrefix_),
        debugOutputParameterFilePath_(o.debugOutputParameterFilePath_),
        useBranchingPriorities_(o.useBranchingPriorities_), prioValue_(o.prioValue_), prioExtent_(o.prioExtent_),
        bufferCplexOutputsToStream_(o.bufferCplexOutputsToStream_) {}
  CplexWrapper& operator=(const CplexWrapper& rhs) {  // src/cplex_wrapper.hpp:153-156
    debugOutputParameterFilePath_ = rhs.debugOutputParameterFilePath_;
    return *this;
  }
  ~CplexWrapper() { if (h_) miqp_solver_destroy(h_); }

  void addRecedingHorizonWarmstart(std::shared_ptr<RawResults> warmstart, WarmstartType wt = RECEDING_HORIZON_WARMSTART) {
    doWarmstart_ = wt != RECEDING_HORIZON_WARMSTART ? BOTH_WARMSTART_STRATEGIES : wt;
    warmstartValues_ = warmstart;
  }
  void setLastSolutionWarmstart(WarmstartType wt = LAST_SOLUTION_WARMSTART) {
    doWarmstart_ = wt != LAST_SOLUTION_WARMSTART ? BOTH_WARMSTART_STRATEGIES : wt;
  }
  void deleteLastSolutionWarmstartFile() { lastSolution_.reset(); std::remove(tmpWarmstartFile_.c_str()); }  // cplex_wrapper.cpp:482-488
  void setParameterDatFileRelative(const char* datfile) { datfile_ = std::string("cplexmodel/") + datfile; }
  void setParameterDatFileAbsolute(const char* datfile) { datfile_ = datfile; }
  void resetParameters(std::shared_ptr<ModelParameters> parameters) { parameters_ = parameters; }
  void overrideSolverSettingsDataSource(std::shared_ptr<ModelParameters> p) {
    if (!parameters_) return;
    parameters_->max_solution_time = p->max_solution_time; parameters_->relative_mip_gap_tolerance = p->relative_mip_gap_tolerance;
    parameters_->mipdisplay = p->mipdisplay; parameters_->mipemphasis = p->mipemphasis; parameters_->relobjdif = p->relobjdif;
    parameters_->cutpass = p->cutpass; parameters_->probe = p->probe; parameters_->repairtries = p->repairtries;
    parameters_->rinsheur = p->rinsheur; parameters_->varsel = p->varsel; parameters_->mircuts = p->mircuts;
    parameters_->parallelmode = p->parallelmode;
  }

  OptimizationStatus callCplex(const double timestemp = 0.0) {
    try {
      if (!h_) { miqp_solver_opts o{}; o.precision = roundingDecimals_ + 2; o.device = -1; o.gap_override = -1.0; h_ = miqp_solver_create(&o); }   // (the C ABI rounds to precision - 2)
      if (!h_) return FAILED_SEG_FAULT;
      if (parameterSource_ == DATFILE || (parameterSource_ == MIXED && !parameters_)) {   // MIXED: the C++ inputs when given, else the file
        if (miqp_solver_load_dat(h_, datfile_.c_str()) != 0) return FAILED_SEG_FAULT;
      } else { if (!parameters_ || !pushParameters()) return FAILED_SEG_FAULT; }
      // MIP starts (cplex_wrapper.cpp:121-138): with BOTH_WARMSTART_STRATEGIES the reference applies the receding-horizon
      // start AND reads the .mst of the last solution
      miqp_solver_set_warmstart(h_, nullptr, MIQP_WARMSTART_NONE);
      if ((doWarmstart_ == RECEDING_HORIZON_WARMSTART || doWarmstart_ == BOTH_WARMSTART_STRATEGIES) && warmstartValues_) pushWarmstart(*warmstartValues_);
      if (doWarmstart_ == LAST_SOLUTION_WARMSTART || doWarmstart_ == BOTH_WARMSTART_STRATEGIES)
        miqp_solver_read_mst(h_, tmpWarmstartFile_.c_str());   // readMIPStarts when the file exists (cplex_wrapper.cpp:128-138)
      const std::string base = debugOutputFilePath_ + "/" + debugOutputFilePrefix_, stamp = stampOf(timestemp);
      if (print_debug_outputs_) {  // cplex_wrapper.cpp:141-155
        debugOutputParameterFilePath_ = base + "parameters_" + stamp + ".txt";
        miqp_solver_write_dat(h_, debugOutputParameterFilePath_.c_str());
        miqp_solver_export_lp(h_, (base + "lpexport_" + stamp + ".lp").c_str());
      }
      int st = miqp_solver_solve(h_, timestemp);
      miqp_solution_properties_c p{}; miqp_solver_get_properties(h_, &p);
      solutionProperties_ = {p.status, p.gap, p.objective, p.time, p.NrConstraints, p.NrBinaryVariables, p.NrFloatVariables,
                             p.NonZeroCoefficients, p.NrIterations, p.NrSolutionPool};
      if (st != SUCCESS) { solutionProperties_.objective = std::nan(""); solutionProperties_.gap = std::nan(""); }   // cplex_wrapper.cpp:231-248
      if (st == SUCCESS) {
        pullResults(); lastSolution_ = std::make_shared<RawResults>(*rawResults_);
        const bool last = doWarmstart_ == LAST_SOLUTION_WARMSTART || doWarmstart_ == BOTH_WARMSTART_STRATEGIES;
        if (last) miqp_solver_write_mst(h_, tmpWarmstartFile_.c_str());                     // cplex_wrapper.cpp:206-209
        if (print_debug_outputs_) {                                                           // cplex_wrapper.cpp:212-229
          miqp_solver_write_solution(h_, (base + "solution_" + stamp + ".txt").c_str());
          if (last) miqp_solver_write_mst(h_, (base + "warmstartsolution_" + stamp + ".mst").c_str());
        }
      }
      return static_cast<OptimizationStatus>(st);
    } catch (...) {
      return FAILED_SEG_FAULT;  // src/cplex_wrapper.cpp:97-109,162-180: every exception maps to this code
    }
  }

  std::shared_ptr<RawResults> getRawResults() const { return rawResults_; }  // aliasing pointer, overwritten by the next solve
  SolutionProperties getSolutionProperties() const { return solutionProperties_; }
  void setDebugOutputPrint(bool v) { print_debug_outputs_ = v; }
  void setDebugOutputFilePath(std::string in) { debugOutputFilePath_ = in; }
  void setDebugOutputFilePrefix(std::string in) { debugOutputFilePrefix_ = in; }
  std::string getDebugOutputParameterFilePath() const { return debugOutputParameterFilePath_; }
  void setSpecialOrderedSets(bool in) { useSpecialOrderedSets_ = in; }       // search hint of CPLEX: no effect on the result
  void setUseBranchingPriorities(bool in) { useBranchingPriorities_ = in; }  // idem
  std::string getTmpWarmstartFile() { return tmpWarmstartFile_; }
  void setBranchingPriorityValueExtent(int value, int extent) { prioValue_ = value; prioExtent_ = extent; }
  void setBufferCplexOutputsToStream(bool in) { bufferCplexOutputsToStream_ = in; }   // (there is no CPLEX log to buffer; the flag is kept and copied)

 private:
  static std::string stampOf(double t) { char b[64]; std::snprintf(b, sizeof(b), "%.15g", t); return b; }  // << setprecision(15)
  std::string tmpWarmstartFile_ = "/tmp/warmstart_debug_res.mst";  // src/cplex_wrapper.hpp:104
  template <class V> static std::vector<double> vec(const V& v) { return std::vector<double>(v.data(), v.data() + v.size()); }
  static std::vector<double> rowMajor(const Eigen::MatrixXd& m) {
    std::vector<double> o((size_t)m.rows() * m.cols());
    for (int r = 0; r < m.rows(); ++r) for (int c = 0; c < m.cols(); ++c) o[(size_t)r * m.cols() + c] = m(r, c);
    return o;
  }
  bool pushParameters() {
    const ModelParameters& P = *parameters_;
    miqp_model_params_c c{};
    c.max_solution_time = P.max_solution_time; c.relative_mip_gap_tolerance = P.relative_mip_gap_tolerance;
    c.mipdisplay = P.mipdisplay; c.mipemphasis = P.mipemphasis; c.relobjdif = P.relobjdif; c.cutpass = P.cutpass; c.probe = P.probe;
    c.repairtries = P.repairtries; c.rinsheur = P.rinsheur; c.varsel = P.varsel; c.mircuts = P.mircuts; c.parallelmode = P.parallelmode;
    c.NumSteps = P.NumSteps; c.nr_regions = P.nr_regions; c.NumCars = P.NumCars; c.nr_obstacles = P.nr_obstacles;
    c.max_lines_obstacles = P.max_lines_obstacles; c.nr_environments = P.nr_environments;
    c.ts = P.ts; c.min_vel_x_y = P.min_vel_x_y; c.max_vel_x_y = P.max_vel_x_y; c.total_min_acc = P.total_min_acc; c.total_max_acc = P.total_max_acc;
    c.total_min_jerk = P.total_min_jerk; c.total_max_jerk = P.total_max_jerk; c.maximum_slack = P.maximum_slack;
    c.WEIGHTS_SLACK = P.WEIGHTS_SLACK; c.WEIGHTS_SLACK_OBSTACLE = P.WEIGHTS_SLACK_OBSTACLE;
    c.minimum_region_change_speed = P.minimum_region_change_speed;
    std::vector<std::vector<double>> keep; keep.reserve(64);
    auto D = [&](std::vector<double> v) { keep.push_back(std::move(v)); return keep.back().data(); };
    c.agent_safety_distance = D(vec(P.agent_safety_distance)); c.agent_safety_distance_slack = D(vec(P.agent_safety_distance_slack));
    c.WEIGHTS_POS_X = D(vec(P.WEIGHTS_POS_X)); c.WEIGHTS_VEL_X = D(vec(P.WEIGHTS_VEL_X)); c.WEIGHTS_ACC_X = D(vec(P.WEIGHTS_ACC_X));
    c.WEIGHTS_POS_Y = D(vec(P.WEIGHTS_POS_Y)); c.WEIGHTS_VEL_Y = D(vec(P.WEIGHTS_VEL_Y)); c.WEIGHTS_ACC_Y = D(vec(P.WEIGHTS_ACC_Y));
    c.WEIGHTS_JERK_X = D(vec(P.WEIGHTS_JERK_X)); c.WEIGHTS_JERK_Y = D(vec(P.WEIGHTS_JERK_Y));
    c.WheelBase = D(vec(P.WheelBase)); c.CollisionRadius = D(vec(P.CollisionRadius)); c.IntitialState = D(rowMajor(P.IntitialState));
    c.x_ref = D(rowMajor(P.x_ref)); c.vx_ref = D(rowMajor(P.vx_ref)); c.y_ref = D(rowMajor(P.y_ref)); c.vy_ref = D(rowMajor(P.vy_ref));
    c.min_acc_x = D(rowMajor(P.acc_limit_params.min_x)); c.max_acc_x = D(rowMajor(P.acc_limit_params.max_x));
    c.min_acc_y = D(rowMajor(P.acc_limit_params.min_y)); c.max_acc_y = D(rowMajor(P.acc_limit_params.max_y));
    c.min_jerk_x = D(rowMajor(P.jerk_limit_params.min_x)); c.max_jerk_x = D(rowMajor(P.jerk_limit_params.max_x));
    c.min_jerk_y = D(rowMajor(P.jerk_limit_params.min_y)); c.max_jerk_y = D(rowMajor(P.jerk_limit_params.max_y));
    std::vector<int> ir(P.initial_region.data(), P.initial_region.data() + P.initial_region.size());
    std::vector<int> pr((size_t)P.possible_region.rows() * P.possible_region.cols());
    for (int r = 0; r < P.possible_region.rows(); ++r) for (int q = 0; q < P.possible_region.cols(); ++q) pr[(size_t)r * P.possible_region.cols() + q] = P.possible_region(r, q);
    c.initial_region = ir.data(); c.possible_region = pr.data();
    std::vector<double> ov; std::vector<int> eo{0}; std::vector<double> ev;
    for (auto& o : P.ObstacleConvexPolygon) for (auto& t : o) { auto m = rowMajor(t); ov.insert(ov.end(), m.begin(), m.end()); }  // [obstacle][time][L][2]
    for (auto& e : P.MultiEnvironmentConvexPolygon) { auto m = rowMajor(e); ev.insert(ev.end(), m.begin(), m.end()); eo.push_back(eo.back() + (int)e.rows()); }
    std::vector<int> soft(P.obstacle_is_soft.begin(), P.obstacle_is_soft.end()); soft.push_back(0); ov.push_back(0); ev.push_back(0);
    c.obstacle_vertices = ov.data(); c.obstacle_is_soft = soft.data(); c.env_offsets = eo.data(); c.env_vertices = ev.data();
    c.fraction_parameters = D(rowMajor(P.fraction_parameters));
    c.POLY_SINT_UB = D(rowMajor(P.poly_orientation_params.POLY_SINT_UB)); c.POLY_SINT_LB = D(rowMajor(P.poly_orientation_params.POLY_SINT_LB));
    c.POLY_COSS_UB = D(rowMajor(P.poly_orientation_params.POLY_COSS_UB)); c.POLY_COSS_LB = D(rowMajor(P.poly_orientation_params.POLY_COSS_LB));
    c.POLY_KAPPA_AX_MAX = D(rowMajor(P.poly_curvature_params.POLY_KAPPA_AX_MAX)); c.POLY_KAPPA_AX_MIN = D(rowMajor(P.poly_curvature_params.POLY_KAPPA_AX_MIN));
    return miqp_solver_set_params(h_, &c) == 0;
  }

  // RawResults (column-major Eigen tensors) <-> row-major POD record
  struct Pod {
    miqp_raw_results_c c{};
    std::vector<std::vector<double>> d; std::vector<std::vector<int>> i;
    Pod(int C, int N, int R, int E, int O, int L) {
      int K = C - 1; c.N = N; c.NrEnvironments = E; c.NrRegions = R; c.NrObstacles = O; c.MaxLinesObstacles = L; c.NrCarToCarCollisions = K; c.NrCars = C;
      d.assign(13, std::vector<double>()); i.assign(17, std::vector<int>());
      double** dp[12] = {&c.u_x, &c.u_y, &c.pos_x, &c.vel_x, &c.acc_x, &c.pos_y, &c.vel_y, &c.acc_y, &c.pos_x_front_UB, &c.pos_x_front_LB, &c.pos_y_front_UB, &c.pos_y_front_LB};
      for (int k = 0; k < 12; ++k) { d[k].assign((size_t)C * N + 1, 9999999.0); *dp[k] = d[k].data(); }
      d[12].assign((size_t)K * K * N * 4 + 1, 0.0); c.slackvars_real = d[12].data();
      int** ip[17] = {&c.notWithinEnvironmentRear, &c.notWithinEnvironmentFrontUbUb, &c.notWithinEnvironmentFrontLbUb, &c.notWithinEnvironmentFrontUbLb,
                      &c.notWithinEnvironmentFrontLbLb, &c.active_region, &c.region_change_not_allowed_x_positive, &c.region_change_not_allowed_y_positive,
                      &c.region_change_not_allowed_x_negative, &c.region_change_not_allowed_y_negative, &c.region_change_not_allowed_combined, &c.deltacc,
                      &c.deltacc_front, &c.car2car_collision, &c.slackvars, &c.slackvarsObstacle, &c.slackvarsObstacle_front};
      size_t sz[17] = {(size_t)C * E * N, (size_t)C * E * N, (size_t)C * E * N, (size_t)C * E * N, (size_t)C * E * N, (size_t)C * N * R, (size_t)C * N, (size_t)C * N,
                       (size_t)C * N, (size_t)C * N, (size_t)C * N, (size_t)C * O * N * L, (size_t)C * O * N * L * 4, (size_t)K * K * N * 16, (size_t)K * K * N * 4,
                       (size_t)C * O * N, (size_t)C * O * N * 4};
      for (int k = 0; k < 17; ++k) { i[k].assign(sz[k] + 1, 9999999); *ip[k] = i[k].data(); }
    }
  };
  template <class T, int R_> static void toTensor(const T* src, Eigen::Tensor<T, R_>& t, std::array<Eigen::Index, R_> dims) {
    t.resize(dims);
    std::array<Eigen::Index, R_> idx{}; size_t n = 1; for (auto v : dims) n *= (size_t)v;
    for (size_t lin = 0; lin < n; ++lin) {  // row-major linear index -> multi index
      size_t rem = lin; for (int k = R_ - 1; k >= 0; --k) { idx[k] = rem % dims[k]; rem /= dims[k]; }
      t(idx) = src[lin];
    }
  }
  template <class T, int R_> static void fromTensor(const Eigen::Tensor<T, R_>& t, T* dst) {
    auto dims = t.dimensions(); std::array<Eigen::Index, R_> idx{}; size_t n = 1; for (int k = 0; k < R_; ++k) n *= (size_t)dims[k];
    for (size_t lin = 0; lin < n; ++lin) { size_t rem = lin; for (int k = R_ - 1; k >= 0; --k) { idx[k] = rem % dims[k]; rem /= dims[k]; } dst[lin] = t(idx); }
  }
  void pullResults() {
    int dm[6]; miqp_solver_get_dims(h_, dm);
    const int C = dm[0], N = dm[1], R = dm[2], E = dm[3], O = dm[4], L = dm[5], K = C - 1;
    Pod p(C, N, R, E, O, L);
    miqp_solver_get_results(h_, &p.c);
    RawResults& r = *rawResults_;
    r.N = N; r.NrEnvironments = E; r.NrRegions = R; r.NrObstacles = O; r.MaxLinesObstacles = L; r.NrCarToCarCollisions = K; r.NrCars = C;
    Eigen::Tensor<double, 2>* d2[12] = {&r.u_x, &r.u_y, &r.pos_x, &r.vel_x, &r.acc_x, &r.pos_y, &r.vel_y, &r.acc_y, &r.pos_x_front_UB, &r.pos_x_front_LB, &r.pos_y_front_UB, &r.pos_y_front_LB};
    for (int k = 0; k < 12; ++k) toTensor<double, 2>(p.d[k].data(), *d2[k], {C, N});
    Eigen::Tensor<int, 3>* e3[5] = {&r.notWithinEnvironmentRear, &r.notWithinEnvironmentFrontUbUb, &r.notWithinEnvironmentFrontLbUb, &r.notWithinEnvironmentFrontUbLb, &r.notWithinEnvironmentFrontLbLb};
    for (int k = 0; k < 5; ++k) toTensor<int, 3>(p.i[k].data(), *e3[k], {C, E, N});
    toTensor<int, 3>(p.i[5].data(), r.active_region, {C, N, R});
    Eigen::Tensor<int, 2>* r2[5] = {&r.region_change_not_allowed_x_positive, &r.region_change_not_allowed_y_positive, &r.region_change_not_allowed_x_negative,
                                    &r.region_change_not_allowed_y_negative, &r.region_change_not_allowed_combined};
    for (int k = 0; k < 5; ++k) toTensor<int, 2>(p.i[6 + k].data(), *r2[k], {C, N});
    toTensor<int, 4>(p.i[11].data(), r.deltacc, {C, O, N, L});
    toTensor<int, 5>(p.i[12].data(), r.deltacc_front, {C, O, N, L, 4});
    toTensor<int, 4>(p.i[13].data(), r.car2car_collision, {K, K, N, 16});
    toTensor<int, 4>(p.i[14].data(), r.slackvars, {K, K, N, 4});
    toTensor<int, 3>(p.i[15].data(), r.slackvarsObstacle, {C, O, N});
    toTensor<int, 4>(p.i[16].data(), r.slackvarsObstacle_front, {C, O, N, 4});
  }
  void pushWarmstart(const RawResults& w) {
    if (w.N <= 0 || w.NrCars <= 0) return;
    Pod p(w.NrCars, w.N, w.NrRegions, w.NrEnvironments, w.NrObstacles, w.MaxLinesObstacles);
    // every variable of the start travels (initializeWarmstart flattens all 27 arrays, src/cplex_wrapper.cpp:494-639)
    const Eigen::Tensor<double, 2>* d2[12] = {&w.u_x, &w.u_y, &w.pos_x, &w.vel_x, &w.acc_x, &w.pos_y, &w.vel_y, &w.acc_y, &w.pos_x_front_UB, &w.pos_x_front_LB, &w.pos_y_front_UB, &w.pos_y_front_LB};
    for (int k = 0; k < 12; ++k) fromTensor<double, 2>(*d2[k], p.d[k].data());
    const Eigen::Tensor<int, 3>* e3[5] = {&w.notWithinEnvironmentRear, &w.notWithinEnvironmentFrontUbUb, &w.notWithinEnvironmentFrontLbUb, &w.notWithinEnvironmentFrontUbLb, &w.notWithinEnvironmentFrontLbLb};
    for (int k = 0; k < 5; ++k) fromTensor<int, 3>(*e3[k], p.i[k].data());
    fromTensor<int, 3>(w.active_region, p.i[5].data());
    const Eigen::Tensor<int, 2>* r2[5] = {&w.region_change_not_allowed_x_positive, &w.region_change_not_allowed_y_positive, &w.region_change_not_allowed_x_negative,
                                          &w.region_change_not_allowed_y_negative, &w.region_change_not_allowed_combined};
    for (int k = 0; k < 5; ++k) fromTensor<int, 2>(*r2[k], p.i[6 + k].data());
    fromTensor<int, 4>(w.deltacc, p.i[11].data()); fromTensor<int, 5>(w.deltacc_front, p.i[12].data());
    fromTensor<int, 4>(w.car2car_collision, p.i[13].data()); fromTensor<int, 4>(w.slackvars, p.i[14].data());
    fromTensor<int, 3>(w.slackvarsObstacle, p.i[15].data()); fromTensor<int, 4>(w.slackvarsObstacle_front, p.i[16].data());
    miqp_solver_set_warmstart(h_, &p.c, MIQP_WARMSTART_RECEDING_HORIZON);
  }

  std::string modfile_, datfile_;
  int precision_;
  int roundingDecimals_;   // decimals the inputs are rounded to: precision - 2, precision for a copy (see the copy constructor)
  ParameterSource parameterSource_;
  std::shared_ptr<RawResults> rawResults_;
  bool print_debug_outputs_ = false;
  WarmstartType doWarmstart_ = NO_WARMSTART;
  std::shared_ptr<RawResults> warmstartValues_, lastSolution_;
  SolutionProperties solutionProperties_{};
  std::string debugOutputFilePath_, debugOutputFilePrefix_, debugOutputParameterFilePath_;
  bool useSpecialOrderedSets_ = false, useBranchingPriorities_ = false;
  int prioValue_ = 1, prioExtent_ = 1;
  bool bufferCplexOutputsToStream_ = false;
  std::shared_ptr<ModelParameters> parameters_;
  miqp_solver_t* h_ = nullptr;  // created lazily
};

}  // namespace cplex
}  // namespace planner
}  // namespace miqp

#endif  // CPLEX_WRAPPER_HEADER
