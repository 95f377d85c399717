// wire_formats.hpp - the debug / interchange files of the reference's CplexWrapper, written without OPL or CPLEX:
//   * OPL external data (.dat syntax; `opl.printExternalData`, src/cplex_wrapper.cpp:141-149) - what
//     test_hardcoded_data_versus_datfile (test/cplex_wrapper_test.cc:474-505) writes and reads back,
//   * OPL-style solution print (`opl.printSolution`, cplex_wrapper.cpp:212-219; same layout as cplexmodel/modelRun.txt),
//   * CPLEX MIP start XML (.mst; `writeMIPStarts` / `readMIPStarts`, cplex_wrapper.cpp:128-138, 206-229).
// Variable names follow the LP export (base#i#j..., 1-based), so an .mst written here loads into CPLEX next to the
// .lp written by lp_export.hpp.
#pragma once
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "host_inst.hpp"

namespace miqp {

// ---------------------------------------------------------------- OPL .dat
inline void dat_vec(FILE* f, const char* name, const double* v, int n) {
  std::fprintf(f, "%s=[", name);
  for (int k = 0; k < n; ++k) std::fprintf(f, "%s%.17g", k ? " " : "", v[k]);
  std::fprintf(f, "];\n");
}
inline void dat_mat(FILE* f, const char* name, const std::vector<double>& v, int rows, int cols) {
  std::fprintf(f, "%s=[", name);
  for (int r = 0; r < rows; ++r) {
    std::fprintf(f, "[");
    for (int k = 0; k < cols; ++k) std::fprintf(f, "%s%.17g", k ? " " : "", v[(size_t)r * cols + k]);
    std::fprintf(f, "]\n");
  }
  std::fprintf(f, "];\n");
}

// every name that parameters.mod declares as external data, in the order of ModelInputDataSource::read
// (src/model_input_data_source.cpp:180-275); solver steering constants are written with the values the reference uses
inline bool write_dat(const HostInst& I, FILE* f) {
  const int N = I.N, C = I.C, R = I.R;
  std::fprintf(f, "/* OPL data written by libmiqp_gpu (same names as cplexmodel/parameters.mod) */\n");
  std::fprintf(f, "NumSteps=%d;\nnr_environments=%d;\nnr_regions=%d;\nnr_obstacles=%d;\nmax_lines_obstacles=%d;\nNumCars=%d;\n", N, I.E, R, I.O, I.L, C);
  std::fprintf(f, "max_solution_time=%.17g;\nrelative_mip_gap_tolerance=%.17g;\n", I.tilim, I.gap);
  std::fprintf(f, "mipdisplay=2;\nmipemphasis=0;\nrelobjdif=0;\ncutpass=0;\nprobe=0;\nrepairtries=0;\nrinsheur=0;\nvarsel=0;\nmircuts=0;\nparallelmode=0;\n");
  std::fprintf(f, "ts=%.17g;\ntotal_min_acc=%.17g;\ntotal_max_acc=%.17g;\ntotal_min_jerk=%.17g;\ntotal_max_jerk=%.17g;\nmin_vel_x_y=%.17g;\nmax_vel_x_y=%.17g;\n",
               I.ts, I.amin, I.amax, I.jmin, I.jmax, I.vmin, I.vmax);
  dat_vec(f, "agent_safety_distance", I.safety.data(), N);
  dat_vec(f, "agent_safety_distance_slack", I.safety_slack.data(), N);
  std::fprintf(f, "maximum_slack=%.17g;\n", I.max_slack);
  const char* wn[8] = {"WEIGHTS_POS_X", "WEIGHTS_VEL_X", "WEIGHTS_ACC_X", "WEIGHTS_POS_Y", "WEIGHTS_VEL_Y", "WEIGHTS_ACC_Y", "WEIGHTS_JERK_X", "WEIGHTS_JERK_Y"};
  for (int k = 0; k < 8; ++k) { std::vector<double> w(C); for (int c = 0; c < C; ++c) w[c] = I.W[c * 8 + k]; dat_vec(f, wn[k], w.data(), C); }
  std::fprintf(f, "WEIGHTS_SLACK=%.17g;\nWEIGHTS_SLACK_OBSTACLE=%.17g;\n", I.w_slack, I.w_slack_obs);
  dat_vec(f, "WheelBase", I.wb.data(), C); dat_vec(f, "CollisionRadius", I.rad.data(), C);
  dat_mat(f, "IntitialState", I.x0, C, 6);
  const char* rn[4] = {"x_ref", "vx_ref", "y_ref", "vy_ref"}; const int ri[4] = {0, 1, 3, 4};
  for (int k = 0; k < 4; ++k) {
    std::vector<double> t((size_t)C * N);
    for (int c = 0; c < C; ++c) for (int i = 0; i < N; ++i) t[(size_t)c * N + i] = I.ref[((size_t)c * N + i) * 6 + ri[k]];
    dat_mat(f, rn[k], t, C, N);
  }
  const char* an[4] = {"min_acc_x", "max_acc_x", "min_acc_y", "max_acc_y"}; const char* jn[4] = {"min_jerk_x", "max_jerk_x", "min_jerk_y", "max_jerk_y"};
  for (int k = 0; k < 4; ++k) {
    std::vector<double> ta((size_t)C * R), tj((size_t)C * R);
    for (int q = 0; q < C * R; ++q) { ta[q] = I.acc_lim[(size_t)q * 4 + k]; tj[q] = I.jerk_lim[(size_t)q * 4 + k]; }
    dat_mat(f, an[k], ta, C, R); dat_mat(f, jn[k], tj, C, R);
  }
  std::fprintf(f, "minimum_region_change_speed=%.17g;\n", I.vm);
  { std::vector<double> t(C); for (int c = 0; c < C; ++c) t[c] = I.init_region[c]; dat_vec(f, "initial_region", t.data(), C); }
  { std::vector<double> t((size_t)C * R); for (int q = 0; q < C * R; ++q) t[q] = I.possible[q]; dat_mat(f, "possible_region", t, C, R); }
  dat_mat(f, "fraction_parameters", I.frac, R, 4);
  const char* pn[6] = {"POLY_SINT_UB", "POLY_SINT_LB", "POLY_COSS_UB", "POLY_COSS_LB", "POLY_KAPPA_AX_MAX", "POLY_KAPPA_AX_MIN"};
  for (int t = 0; t < 6; ++t) dat_mat(f, pn[t], I.poly[t], R, 3);
  // obstacles: [obstacle][time]{<k x1 y1 x2 y2>} (parameters.mod:114)
  std::fprintf(f, "ObstacleConvexPolygon=[");
  for (int o = 0; o < I.O; ++o) {
    std::fprintf(f, "%s[", o ? "," : "");
    for (int i = 0; i < N; ++i) {
      std::fprintf(f, "%s{", i ? "," : "");
      for (int k = 0; k < I.L; ++k) { const double* e = &I.obs_edges[((size_t)(o * N + i) * I.L + k) * 4]; std::fprintf(f, "<%d,%.17g,%.17g,%.17g,%.17g>\n", k + 1, e[0], e[1], e[2], e[3]); }
      std::fprintf(f, "}");
    }
    std::fprintf(f, "]");
  }
  std::fprintf(f, "];\n");
  { std::vector<double> t(I.O > 0 ? I.O : 0); for (int o = 0; o < I.O; ++o) t[o] = I.obs_soft[o]; dat_vec(f, "obstacle_is_soft", t.data(), I.O); }
  std::fprintf(f, "MultiEnvironmentConvexPolygon=[");
  for (int e = 0; e < I.E; ++e) {
    std::fprintf(f, "%s{", e ? "," : "");
    for (int k = I.env_off[e]; k < I.env_off[e + 1]; ++k) { const double* ed = &I.env_edges[(size_t)k * 4]; std::fprintf(f, "<%d,%.17g,%.17g,%.17g,%.17g>\n", k - I.env_off[e] + 1, ed[0], ed[1], ed[2], ed[3]); }
    std::fprintf(f, "}");
  }
  std::fprintf(f, "];\n");
  return !std::ferror(f);
}

// ---------------------------------------------------------------- RawResults <-> named arrays
struct ResultField { const char* name; bool is_int; const void* ptr; std::vector<int> shape; };

inline std::vector<ResultField> result_fields(const miqp_raw_results_c& r) {
  const int C = r.NrCars, N = r.N, R = r.NrRegions, E = r.NrEnvironments, O = r.NrObstacles, L = r.MaxLinesObstacles, K = r.NrCarToCarCollisions;
  std::vector<ResultField> F;
  auto d = [&](const char* n, const double* p, std::vector<int> s) { F.push_back({n, false, p, std::move(s)}); };
  auto i = [&](const char* n, const int* p, std::vector<int> s) { F.push_back({n, true, p, std::move(s)}); };
  d("u_x", r.u_x, {C, N}); d("u_y", r.u_y, {C, N}); d("pos_x", r.pos_x, {C, N}); d("vel_x", r.vel_x, {C, N}); d("acc_x", r.acc_x, {C, N});
  d("pos_y", r.pos_y, {C, N}); d("vel_y", r.vel_y, {C, N}); d("acc_y", r.acc_y, {C, N});
  d("pos_x_front_UB", r.pos_x_front_UB, {C, N}); d("pos_x_front_LB", r.pos_x_front_LB, {C, N});
  d("pos_y_front_UB", r.pos_y_front_UB, {C, N}); d("pos_y_front_LB", r.pos_y_front_LB, {C, N});
  i("notWithinEnvironmentRear", r.notWithinEnvironmentRear, {C, E, N}); i("notWithinEnvironmentFrontUbUb", r.notWithinEnvironmentFrontUbUb, {C, E, N});
  i("notWithinEnvironmentFrontLbUb", r.notWithinEnvironmentFrontLbUb, {C, E, N}); i("notWithinEnvironmentFrontUbLb", r.notWithinEnvironmentFrontUbLb, {C, E, N});
  i("notWithinEnvironmentFrontLbLb", r.notWithinEnvironmentFrontLbLb, {C, E, N});
  i("active_region", r.active_region, {C, N, R});
  i("region_change_not_allowed_x_positive", r.region_change_not_allowed_x_positive, {C, N}); i("region_change_not_allowed_y_positive", r.region_change_not_allowed_y_positive, {C, N});
  i("region_change_not_allowed_x_negative", r.region_change_not_allowed_x_negative, {C, N}); i("region_change_not_allowed_y_negative", r.region_change_not_allowed_y_negative, {C, N});
  i("region_change_not_allowed_combined", r.region_change_not_allowed_combined, {C, N});
  i("deltacc", r.deltacc, {C, O, N, L}); i("deltacc_front", r.deltacc_front, {C, O, N, L, 4});
  i("car2car_collision", r.car2car_collision, {K, K, N, 16});
  if (r.slackvars_real) d("slackvars", r.slackvars_real, {K, K, N, 4}); else i("slackvars", r.slackvars, {K, K, N, 4});
  i("slackvarsObstacle", r.slackvarsObstacle, {C, O, N}); i("slackvarsObstacle_front", r.slackvarsObstacle_front, {C, O, N, 4});
  return F;
}

inline size_t field_size(const ResultField& f) { size_t n = 1; for (int s : f.shape) n *= (size_t)(s > 0 ? s : 0); return n; }

// `name = [[..] [..]];` blocks in the layout of OPL's printSolution (cplexmodel/modelRun.txt)
inline bool write_solution(const miqp_raw_results_c& r, double objective, FILE* f) {
  std::fprintf(f, "// solution (optimal) with objective %.17g\n", objective);
  for (auto& fd : result_fields(r)) {
    const size_t n = field_size(fd);
    std::fprintf(f, "%s = ", fd.name);
    if (n == 0 || !fd.ptr) { std::fprintf(f, "[];\n"); continue; }
    std::vector<size_t> stride(fd.shape.size(), 1);
    for (int k = (int)fd.shape.size() - 2; k >= 0; --k) stride[k] = stride[k + 1] * (size_t)fd.shape[k + 1];
    for (size_t q = 0; q < n; ++q) {
      for (size_t k = 0; k < fd.shape.size(); ++k) if (q % (stride[k] * (size_t)fd.shape[k]) == 0) std::fprintf(f, "[");
      if (fd.is_int) std::fprintf(f, "%d", ((const int*)fd.ptr)[q]); else std::fprintf(f, "%.17g", ((const double*)fd.ptr)[q]);
      bool closed = false;
      for (int k = (int)fd.shape.size() - 1; k >= 0; --k) if ((q + 1) % (stride[k] * (size_t)fd.shape[k]) == 0) { std::fprintf(f, "]"); closed = true; }
      std::fprintf(f, closed ? "\n" : " ");
    }
    std::fprintf(f, ";\n");
  }
  return !std::ferror(f);
}

// ---------------------------------------------------------------- CPLEX .mst
inline std::string var_name(const ResultField& fd, size_t q) {
  std::string s = fd.name;
  std::vector<int> idx(fd.shape.size());
  for (int k = (int)fd.shape.size() - 1; k >= 0; --k) { idx[k] = (int)(q % (size_t)fd.shape[k]); q /= (size_t)fd.shape[k]; }
  for (int v : idx) { s += "#"; s += std::to_string(v + 1); }
  return s;
}

inline bool write_mst(const miqp_raw_results_c& r, double objective, FILE* f) {
  std::fprintf(f, "<?xml version = \"1.0\" encoding=\"UTF-8\" standalone=\"yes\"?>\n<CPLEXSolutions version=\"1.2\">\n <CPLEXSolution version=\"1.2\">\n");
  std::fprintf(f, "  <header\n    problemName=\"planner-miqp\"\n    solutionName=\"m1\"\n    solutionIndex=\"0\"\n    objectiveValue=\"%.17g\"\n    MIPStartEffortLevel=\"4\"\n    writeLevel=\"2\"/>\n  <variables>\n", objective);
  long index = 0;
  for (auto& fd : result_fields(r)) {
    const size_t n = field_size(fd);
    if (!fd.ptr) continue;
    for (size_t q = 0; q < n; ++q, ++index) {
      if (fd.is_int) std::fprintf(f, "   <variable name=\"%s\" index=\"%ld\" value=\"%d\"/>\n", var_name(fd, q).c_str(), index, ((const int*)fd.ptr)[q]);
      else std::fprintf(f, "   <variable name=\"%s\" index=\"%ld\" value=\"%.17g\"/>\n", var_name(fd, q).c_str(), index, ((const double*)fd.ptr)[q]);
    }
  }
  std::fprintf(f, "  </variables>\n </CPLEXSolution>\n</CPLEXSolutions>\n");
  return !std::ferror(f);
}

// fills the arrays of `r` (dims and pointers preset by the caller) from <variable name=".." value=".."/> entries;
// unknown names are ignored, returns the number of entries stored (-1: cannot open)
inline long read_mst(const char* path, miqp_raw_results_c& r) {
  FILE* f = std::fopen(path, "rb");
  if (!f) return -1;
  std::string text; char buf[65536]; size_t n;
  while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0) text.append(buf, n);
  std::fclose(f);
  auto fields = result_fields(r);
  long stored = 0; size_t pos = 0;
  for (;;) {
    size_t a = text.find("<variable", pos); if (a == std::string::npos) break;
    size_t e = text.find("/>", a); if (e == std::string::npos) break;
    pos = e + 2;
    size_t n0 = text.find("name=\"", a); size_t v0 = text.find("value=\"", a);
    if (n0 == std::string::npos || v0 == std::string::npos || n0 > e || v0 > e) continue;
    n0 += 6; v0 += 7;
    std::string name = text.substr(n0, text.find('"', n0) - n0);
    double val = std::strtod(text.c_str() + v0, nullptr);
    size_t h = name.find('#'); std::string base = name.substr(0, h);
    for (auto& fd : fields) {
      if (base != fd.name || !fd.ptr) continue;
      std::vector<int> idx; size_t p2 = h;
      while (p2 != std::string::npos) { idx.push_back(std::atoi(name.c_str() + p2 + 1) - 1); p2 = name.find('#', p2 + 1); }
      if (idx.size() != fd.shape.size()) break;
      size_t q = 0; bool ok = true;
      for (size_t k = 0; k < idx.size(); ++k) { if (idx[k] < 0 || idx[k] >= fd.shape[k]) { ok = false; break; } q = q * (size_t)fd.shape[k] + (size_t)idx[k]; }
      if (!ok) break;
      if (fd.is_int) ((int*)fd.ptr)[q] = (int)std::lround(val); else ((double*)fd.ptr)[q] = val;
      stored++;
      break;
    }
  }
  return stored;
}

}  // namespace miqp
