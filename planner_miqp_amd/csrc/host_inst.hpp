// host_inst.hpp - host-side instance of the MIQP solve path and its compilation into device tables.
//
// Product code (no dependency on oracle/).  Restates, for the device solver:
//   ModelInputDataSource::read / addLineSet       src/model_input_data_source.cpp:167-275 (rounding, edges)
//   cplexmodel/parameters.mod:24-32               big-M constants
//   cplexmodel/initial_conditions.mod:30-48       step-1 jerk box
//   cplexmodel/model_region_constraints.mod:43-114 region block (sector, front polynomials, boxes, curvature)
//   cplexmodel/minimum_speed_constraints.mod:9-49 low-speed freeze -> non-slow half-planes per sector
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/miqp_types.h"

// Environment switches of the library.  KNOB_P: read by the shipped library - the list, with defaults, is INTEGRATION.md section "Environment
// switches" (tests/test_abi_cpu.py compares the two).  KNOB_T: an experiment or a tuning sweep of an earlier round; live only in a tuning build
// (-DMIQP_TUNING=1, tools/build_variants.sh), in the product build the name is not even in the binary and the documented default applies.
#define KNOB_P(name) std::getenv(name)
#ifdef MIQP_TUNING
#define KNOB_T(name) std::getenv(name)
#else
#define KNOB_T(name) ((const char*)nullptr)
#endif

extern char** environ;

namespace miqp {

// A MIQP_* variable that the shipped library does not read (a switch of a tuning build, a typo) is named once on stderr instead of being a
// silent no-op (advisor finding of round 5).  The names the product build reads (KNOB_P) and the two of the Python side:
inline void warn_ignored_switches() {
#ifndef MIQP_TUNING
  static bool done = false;
  if (done) return;
  done = true;
  static const char* const known[] = {"MIQP_SEQ_KINDS", "MIQP_LNS", "MIQP_PUMP", "MIQP_CUT_GATE", "MIQP_NPR", "MIQP_OPEN_CAP", "MIQP_FAR_CAP", "MIQP_LANES", "MIQP_AS",
                                      "MIQP_STATS", "MIQP_DEBUG_SYNC", "MIQP_TRACE", "MIQP_ROUND_LOG", "MIQP_GPU_LIB", "MIQP_BENCH_DUMP_SEEDS"};
  for (char** e = environ; e && *e; ++e) {
    if (std::strncmp(*e, "MIQP_", 5) != 0) continue;
    const char* eq = std::strchr(*e, '='); const size_t len = eq ? (size_t)(eq - *e) : std::strlen(*e);
    bool ok = false;
    for (const char* k : known) if (std::strlen(k) == len && std::strncmp(k, *e, len) == 0) { ok = true; break; }
    if (!ok) std::fprintf(stderr, "[miqp_gpu] environment variable %.*s is ignored: the shipped library reads only the switches of INTEGRATION.md section 5 (experiment switches exist in a tuning build, tools/build_variants.sh)\n", (int)len, *e);
  }
#endif
}

constexpr int MAXC = 4;          // cars supported by the device kernels (stage vector of at most 32 entries)
constexpr int REGSZ = 32;        // doubles per (car, possible region) table entry
constexpr int HULLM_MAXBITS = 8;   // region sets of up to 2^8 members are tabulated (d_hullm); beyond that the static hull of the step is used
constexpr int HULLSZ = 16;       // doubles per (car, step) of the hull of the region alternatives: acc box 4, jerk box 4, two velocity rows (g_vx, g_vy, rhs) 6, valid flag, pad
constexpr double BIGM_JERK = 10.0, BIGM_ACC = 10.0;

struct HostInst {
  int N = 0, C = 0, R = 0, E = 0, O = 0, L = 0;
  double ts = 0, vmin = 0, vmax = 0, amin = 0, amax = 0, jmin = 0, jmax = 0, max_slack = 0, w_slack = 0,
         w_slack_obs = 0, vm = 0, gap = 0.1, tilim = 10;
  std::vector<double> safety, safety_slack, W /*C*8*/, wb, rad, x0 /*C*6*/, ref /*C*N*6*/, acc_lim, jerk_lim /*C*R*4*/,
      frac /*R*4*/, poly[6] /*R*3: SINT_UB SINT_LB COSS_UB COSS_LB KMAX KMIN*/, env_edges /*4 per edge*/, obs_edges;
  std::vector<int> init_region, possible, env_off, obs_soft;
  int NP() const { return C * (C - 1) / 2; }
};

inline double round_dec(double v, int dec) {
  if (dec < 0) return v;
  double s = std::pow(10.0, dec);
  return std::round(v * s) / s;
}

inline bool inst_from_params(const miqp_model_params_c* p, int dec, HostInst& I, std::string& err) {
  I = HostInst();
  I.N = p->NumSteps; I.C = p->NumCars; I.R = p->nr_regions; I.E = p->nr_environments; I.O = p->nr_obstacles;
  I.L = p->max_lines_obstacles;
  if (I.C < 1 || I.N < 2 || I.R < 1 || I.E < 0 || I.O < 0 || I.L < 0 || (I.O > 0 && I.L < 1)) { err = "invalid sizes"; return false; }
  {  // every array the sizes announce must be there
    const void* need[] = {p->agent_safety_distance, p->agent_safety_distance_slack, p->WEIGHTS_POS_X, p->WEIGHTS_VEL_X, p->WEIGHTS_ACC_X, p->WEIGHTS_POS_Y,
                          p->WEIGHTS_VEL_Y, p->WEIGHTS_ACC_Y, p->WEIGHTS_JERK_X, p->WEIGHTS_JERK_Y, p->WheelBase, p->CollisionRadius, p->IntitialState, p->x_ref,
                          p->vx_ref, p->y_ref, p->vy_ref, p->min_acc_x, p->max_acc_x, p->min_acc_y, p->max_acc_y, p->min_jerk_x, p->max_jerk_x, p->min_jerk_y,
                          p->max_jerk_y, p->initial_region, p->possible_region, p->fraction_parameters, p->POLY_SINT_UB, p->POLY_SINT_LB, p->POLY_COSS_UB,
                          p->POLY_COSS_LB, p->POLY_KAPPA_AX_MAX, p->POLY_KAPPA_AX_MIN};
    for (const void* q : need) if (!q) { err = "null array in ModelParameters"; return false; }
    if (I.E > 0 && (!p->env_offsets || !p->env_vertices)) { err = "null environment arrays"; return false; }
    if (I.O > 0 && (!p->obstacle_vertices || !p->obstacle_is_soft)) { err = "null obstacle arrays"; return false; }
    for (int e = 0; e < I.E; ++e) if (p->env_offsets[e + 1] - p->env_offsets[e] < 1 || p->env_offsets[e] < 0) { err = "invalid env_offsets"; return false; }
    for (int c = 0; c < I.C; ++c) {
      if (p->initial_region[c] < 1 || p->initial_region[c] > I.R) { err = "initial_region outside 1..nr_regions"; return false; }
      int np = 0; for (int j = 0; j < I.R; ++j) np += p->possible_region[c * I.R + j] == 1;
      if (np < 1) { err = "car without a possible region"; return false; }
    }
  }
  auto r = [&](double v) { return round_dec(v, dec); };
  I.ts = r(p->ts); I.vmin = r(p->min_vel_x_y); I.vmax = r(p->max_vel_x_y); I.amin = r(p->total_min_acc);
  I.amax = r(p->total_max_acc); I.jmin = r(p->total_min_jerk); I.jmax = r(p->total_max_jerk);
  I.max_slack = r(p->maximum_slack); I.w_slack = r(p->WEIGHTS_SLACK); I.w_slack_obs = r(p->WEIGHTS_SLACK_OBSTACLE);
  I.vm = r(p->minimum_region_change_speed); I.gap = r(p->relative_mip_gap_tolerance); I.tilim = r(p->max_solution_time);
  int N = I.N, C = I.C, R = I.R;
  I.safety.resize(N); I.safety_slack.resize(N);
  for (int i = 0; i < N; ++i) { I.safety[i] = r(p->agent_safety_distance[i]); I.safety_slack[i] = r(p->agent_safety_distance_slack[i]); }
  const double* Ws[8] = {p->WEIGHTS_POS_X, p->WEIGHTS_VEL_X, p->WEIGHTS_ACC_X, p->WEIGHTS_POS_Y,
                         p->WEIGHTS_VEL_Y, p->WEIGHTS_ACC_Y, p->WEIGHTS_JERK_X, p->WEIGHTS_JERK_Y};
  I.W.assign(C * 8, 0); I.wb.resize(C); I.rad.resize(C); I.x0.resize(C * 6); I.ref.assign((size_t)C * N * 6, 0);
  I.acc_lim.resize((size_t)C * R * 4); I.jerk_lim.resize((size_t)C * R * 4); I.init_region.resize(C); I.possible.resize(C * R);
  for (int c = 0; c < C; ++c) {
    for (int k = 0; k < 8; ++k) I.W[c * 8 + k] = r(Ws[k][c]);
    I.wb[c] = r(p->WheelBase[c]); I.rad[c] = r(p->CollisionRadius[c]);
    for (int k = 0; k < 6; ++k) I.x0[c * 6 + k] = r(p->IntitialState[c * 6 + k]);
    for (int i = 0; i < N; ++i) {
      double* f = &I.ref[((size_t)c * N + i) * 6];
      f[0] = r(p->x_ref[c * N + i]); f[1] = r(p->vx_ref[c * N + i]); f[3] = r(p->y_ref[c * N + i]); f[4] = r(p->vy_ref[c * N + i]);
    }
    for (int j = 0; j < R; ++j) {
      double* a = &I.acc_lim[((size_t)c * R + j) * 4]; double* jl = &I.jerk_lim[((size_t)c * R + j) * 4];
      a[0] = r(p->min_acc_x[c * R + j]); a[1] = r(p->max_acc_x[c * R + j]); a[2] = r(p->min_acc_y[c * R + j]); a[3] = r(p->max_acc_y[c * R + j]);
      jl[0] = r(p->min_jerk_x[c * R + j]); jl[1] = r(p->max_jerk_x[c * R + j]); jl[2] = r(p->min_jerk_y[c * R + j]); jl[3] = r(p->max_jerk_y[c * R + j]);
      I.possible[c * R + j] = p->possible_region[c * R + j];
    }
    I.init_region[c] = p->initial_region[c];
  }
  const double* Ps[6] = {p->POLY_SINT_UB, p->POLY_SINT_LB, p->POLY_COSS_UB, p->POLY_COSS_LB, p->POLY_KAPPA_AX_MAX, p->POLY_KAPPA_AX_MIN};
  I.frac.resize(R * 4);
  for (int k = 0; k < R * 4; ++k) I.frac[k] = r(p->fraction_parameters[k]);
  for (int t = 0; t < 6; ++t) { I.poly[t].resize(R * 3); for (int k = 0; k < R * 3; ++k) I.poly[t][k] = r(Ps[t][k]); }
  I.env_off.assign(I.E + 1, 0);
  for (int e = 0; e < I.E; ++e) {
    int a = p->env_offsets[e], b = p->env_offsets[e + 1], n = b - a;
    I.env_off[e] = a; I.env_off[e + 1] = b;
    for (int k = 0; k < n; ++k) {  // addLineSet: consecutive vertices, last edge wraps around
      int k2 = (k + 1) % n;
      I.env_edges.push_back(r(p->env_vertices[2 * (a + k)])); I.env_edges.push_back(r(p->env_vertices[2 * (a + k) + 1]));
      I.env_edges.push_back(r(p->env_vertices[2 * (a + k2)])); I.env_edges.push_back(r(p->env_vertices[2 * (a + k2) + 1]));
    }
  }
  I.obs_soft.resize(I.O);
  I.obs_edges.resize((size_t)I.O * N * I.L * 4);
  for (int o = 0; o < I.O; ++o) {
    I.obs_soft[o] = p->obstacle_is_soft[o];
    for (int i = 0; i < N; ++i)
      for (int k = 0; k < I.L; ++k) {
        int k2 = (k + 1) % I.L;
        const double* v = p->obstacle_vertices + ((size_t)(o * N + i) * I.L) * 2;
        double* ed = &I.obs_edges[((size_t)(o * N + i) * I.L + k) * 4];
        ed[0] = r(v[2 * k]); ed[1] = r(v[2 * k + 1]); ed[2] = r(v[2 * k2]); ed[3] = r(v[2 * k2 + 1]);
      }
  }
  return true;
}

// ---------------------------------------------------------------- OPL .dat subset
struct DatValue {
  bool is_num = false; double num = 0; char open = 0; std::vector<DatValue> items;
  void flatten(std::vector<double>& out) const { if (is_num) out.push_back(num); else for (auto& v : items) v.flatten(out); }
  int row_width() const { if (is_num) return 1; if (!items.empty() && !items[0].is_num) return items[0].row_width(); return (int)items.size(); }
};

class DatReader {
 public:
  explicit DatReader(const std::string& text) : s_(text) {}
  bool parse(std::vector<std::pair<std::string, DatValue>>& out) {
    for (;;) {
      ws(); if (pos_ >= s_.size()) return true;
      size_t st = pos_;
      while (pos_ < s_.size() && (isalnum((unsigned char)s_[pos_]) || s_[pos_] == '_')) pos_++;
      if (pos_ == st) return false;
      std::string name = s_.substr(st, pos_ - st);
      ws(); if (pos_ >= s_.size() || s_[pos_] != '=') return false; pos_++;
      DatValue v; if (!value(v)) return false;
      ws(); if (pos_ >= s_.size() || s_[pos_] != ';') return false; pos_++;
      out.emplace_back(name, std::move(v));
    }
  }
 private:
  void ws() {
    for (;;) {
      while (pos_ < s_.size() && (isspace((unsigned char)s_[pos_]) || s_[pos_] == ',')) pos_++;
      if (pos_ + 1 < s_.size() && s_[pos_] == '/' && s_[pos_ + 1] == '*') { size_t e = s_.find("*/", pos_ + 2); pos_ = e == std::string::npos ? s_.size() : e + 2; }
      else if (pos_ + 1 < s_.size() && s_[pos_] == '/' && s_[pos_ + 1] == '/') { while (pos_ < s_.size() && s_[pos_] != '\n') pos_++; }
      else return;
    }
  }
  bool value(DatValue& v) {
    ws(); if (pos_ >= s_.size()) return false;
    char ch = s_[pos_];
    if (ch == '[' || ch == '{' || ch == '<') {
      char close = ch == '[' ? ']' : (ch == '{' ? '}' : '>');
      v.open = ch; pos_++;
      for (;;) {
        ws(); if (pos_ >= s_.size()) return false;
        if (s_[pos_] == close) { pos_++; return true; }
        DatValue it; if (!value(it)) return false; v.items.push_back(std::move(it));
      }
    }
    char* end = nullptr; v.is_num = true; v.num = std::strtod(s_.c_str() + pos_, &end);
    if (end == s_.c_str() + pos_) return false;
    pos_ = (size_t)(end - s_.c_str());
    return true;
  }
  const std::string& s_; size_t pos_ = 0;
};

inline bool inst_from_dat(const char* path, HostInst& I, std::string& err) {
  FILE* f = std::fopen(path, "rb");
  if (!f) { err = std::string("cannot open ") + path; return false; }
  std::string text; char buf[65536]; size_t n;
  while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0) text.append(buf, n);
  std::fclose(f);
  std::vector<std::pair<std::string, DatValue>> ents;
  DatReader rd(text);
  if (!rd.parse(ents)) { err = "syntax error in .dat"; return false; }
  auto find = [&](const char* nm) -> const DatValue* { for (auto& e : ents) if (e.first == nm) return &e.second; return nullptr; };
  bool miss = false;
  auto num = [&](const char* nm) { auto* v = find(nm); if (!v || !v->is_num) { miss = true; err = std::string("missing ") + nm; return 0.0; } return v->num; };
  // table narrower than the header says is zero filled (cplexmodel.dat: 16-wide tables under nr_regions = 32)
  auto table = [&](const char* nm, int rows, int width) {
    std::vector<double> out((size_t)rows * width, 0.0);
    auto* v = find(nm); if (!v) { miss = true; err = std::string("missing ") + nm; return out; }
    std::vector<double> flat; v->flatten(flat);
    int w = v->row_width(); if (w <= 0) w = 1;
    int rs = (int)flat.size() / w;
    for (int r = 0; r < rows && r < rs; ++r) for (int k = 0; k < width && k < w; ++k) out[(size_t)r * width + k] = flat[(size_t)r * w + k];
    return out;
  };
  I = HostInst();
  I.N = (int)num("NumSteps"); I.C = (int)num("NumCars"); I.R = (int)num("nr_regions"); I.E = (int)num("nr_environments");
  I.O = (int)num("nr_obstacles"); I.L = (int)num("max_lines_obstacles");
  if (miss || I.C < 1 || I.N < 2) { if (err.empty()) err = "invalid sizes"; return false; }
  int N = I.N, C = I.C, R = I.R;
  I.ts = num("ts"); I.vmin = num("min_vel_x_y"); I.vmax = num("max_vel_x_y"); I.amin = num("total_min_acc"); I.amax = num("total_max_acc");
  I.jmin = num("total_min_jerk"); I.jmax = num("total_max_jerk"); I.max_slack = num("maximum_slack"); I.w_slack = num("WEIGHTS_SLACK");
  I.w_slack_obs = num("WEIGHTS_SLACK_OBSTACLE"); I.vm = num("minimum_region_change_speed"); I.gap = num("relative_mip_gap_tolerance");
  I.tilim = num("max_solution_time");
  I.safety = table("agent_safety_distance", 1, N); I.safety_slack = table("agent_safety_distance_slack", 1, N);
  const char* wn[8] = {"WEIGHTS_POS_X", "WEIGHTS_VEL_X", "WEIGHTS_ACC_X", "WEIGHTS_POS_Y", "WEIGHTS_VEL_Y", "WEIGHTS_ACC_Y", "WEIGHTS_JERK_X", "WEIGHTS_JERK_Y"};
  I.W.assign(C * 8, 0);
  for (int k = 0; k < 8; ++k) { auto t = table(wn[k], 1, C); for (int c = 0; c < C; ++c) I.W[c * 8 + k] = t[c]; }
  I.wb = table("WheelBase", 1, C); I.rad = table("CollisionRadius", 1, C); I.x0 = table("IntitialState", C, 6);
  I.ref.assign((size_t)C * N * 6, 0);
  const char* rn[4] = {"x_ref", "vx_ref", "y_ref", "vy_ref"}; const int ri[4] = {0, 1, 3, 4};
  for (int k = 0; k < 4; ++k) { auto t = table(rn[k], C, N); for (int c = 0; c < C; ++c) for (int i = 0; i < N; ++i) I.ref[((size_t)c * N + i) * 6 + ri[k]] = t[(size_t)c * N + i]; }
  const char* an[4] = {"min_acc_x", "max_acc_x", "min_acc_y", "max_acc_y"}; const char* jn[4] = {"min_jerk_x", "max_jerk_x", "min_jerk_y", "max_jerk_y"};
  I.acc_lim.assign((size_t)C * R * 4, 0); I.jerk_lim.assign((size_t)C * R * 4, 0);
  for (int k = 0; k < 4; ++k) {
    auto ta = table(an[k], C, R); auto tj = table(jn[k], C, R);
    for (int q = 0; q < C * R; ++q) { I.acc_lim[(size_t)q * 4 + k] = ta[q]; I.jerk_lim[(size_t)q * 4 + k] = tj[q]; }
  }
  { auto t = table("initial_region", 1, C); I.init_region.resize(C); for (int c = 0; c < C; ++c) I.init_region[c] = (int)t[c]; }
  { auto t = table("possible_region", C, R); I.possible.resize(C * R); for (int q = 0; q < C * R; ++q) I.possible[q] = (int)t[q]; }
  I.frac = table("fraction_parameters", R, 4);
  const char* pn[6] = {"POLY_SINT_UB", "POLY_SINT_LB", "POLY_COSS_UB", "POLY_COSS_LB", "POLY_KAPPA_AX_MAX", "POLY_KAPPA_AX_MIN"};
  for (int t = 0; t < 6; ++t) I.poly[t] = table(pn[t], R, 3);
  const DatValue* env = find("MultiEnvironmentConvexPolygon"); const DatValue* obs = find("ObstacleConvexPolygon");
  I.env_off.assign(I.E + 1, 0);
  if (I.E > 0 && (!env || (int)env->items.size() < I.E)) { err = "MultiEnvironmentConvexPolygon too short"; return false; }
  for (int e = 0; e < I.E; ++e) {
    I.env_off[e] = (int)I.env_edges.size() / 4;
    for (auto& t : env->items[e].items) for (int q = 1; q <= 4; ++q) I.env_edges.push_back(q < (int)t.items.size() ? t.items[q].num : 0.0);
  }
  I.env_off[I.E] = (int)I.env_edges.size() / 4;
  I.obs_edges.assign((size_t)I.O * N * I.L * 4, 0.0); I.obs_soft.assign(I.O, 0);
  if (I.O > 0) {
    if (!obs || (int)obs->items.size() < I.O) { err = "ObstacleConvexPolygon too short"; return false; }
    for (int o = 0; o < I.O; ++o)
      for (int i = 0; i < N && i < (int)obs->items[o].items.size(); ++i) {
        auto& set = obs->items[o].items[i];
        for (int k = 0; k < I.L && k < (int)set.items.size(); ++k)
          for (int q = 1; q <= 4 && q < (int)set.items[k].items.size(); ++q)
            I.obs_edges[((size_t)(o * N + i) * I.L + k) * 4 + q - 1] = set.items[k].items[q].num;
      }
    auto t = table("obstacle_is_soft", 1, I.O); for (int o = 0; o < I.O; ++o) I.obs_soft[o] = (int)t[o];
  }
  if (miss) return false;
  return true;
}

// ---------------------------------------------------------------- device table layout (shared by a batch)
struct Layout {
  int C, N, R, P, E, EL, O, L, NP;   // P = max possible regions per car, EL = max edges per environment piece
  int nx, nu, nz, SC, NSLOT, ROWCAP;
  // double offsets
  int d_x0, d_wd, d_ref, d_glob, d_u0box, d_misc, d_dsep, d_ssl, d_smax, d_reg, d_env, d_obs, d_theta, d_lift, d_hull, d_hullm, d_fpk, d_astab, dstride;
  int PT;   // bits of the region-set index of d_hullm: min(P, HULLM_MAXBITS)
  int relax_front_off;   // 1 (default): car/car rows on front points wait for the region of their car; 0: experiment, see make_layout
  // int offsets
  int i_nposs, i_regj, i_nhs, i_hs, i_envn, i_obssoft, i_initj, i_dom, i_allow, i_boxskip, i_c2callow, i_rallow, istride;
  // fix record (bytes)
  int f_reg, f_env, f_obs, f_c2c, f_c2n, f_rmask, fixlen;   // f_c2n: per (pair, step, group) bit mask of excluded car/car alternatives
  // f_rmask: per (car, step) two bytes = bit set of the possible regions the node still allows there (0xFFFF in a root: every
  // region; the kernels intersect it with the static reachability set i_rallow).  eval_kernel clears the bits of regions whose
  // cheapest alternative, priced by the bound lifting from the node's dual solution, cannot beat the incumbent any more; the
  // children inherit the set, and the relaxation of an undecided step uses the hull of the boxes of the regions LEFT (d_hullm).
};

inline Layout make_layout(int C, int N, int R, int P, int E, int EL, int O, int L) {
  Layout Y; std::memset(&Y, 0, sizeof(Y));
  Y.C = C; Y.N = N; Y.R = R; Y.P = P; Y.E = E; Y.EL = EL; Y.O = O; Y.L = L; Y.NP = C * (C - 1) / 2;
  Y.nx = 6 * C; Y.nu = 2 * C; Y.nz = 8 * C;
  // (OFF by default; MIQP_RELAX_FRONT=1 switches the relaxed rows on.  Measured on the bench instances: deciding a car/car group
  // on a front point BEFORE the region of its car - with the front-point offset bounded over the region set - doubles the nodes,
  // 23 M against 10.9 M on a 2048-instance queue: the exact row is still violated afterwards, the region is branched anyway)
  { const char* e = KNOB_T("MIQP_RELAX_FRONT"); Y.relax_front_off = (e && std::atoi(e) == 1) ? 0 : 1; }
  Y.SC = 16 + 5 * EL + 5 * O; Y.NSLOT = C * Y.SC + Y.NP * 24; Y.ROWCAP = N * Y.NSLOT;
  int o = 0;
  Y.d_x0 = o; o += C * 6; Y.d_wd = o; o += Y.nz; Y.d_ref = o; o += N * Y.nz; Y.d_glob = o; o += 8; Y.d_u0box = o; o += C * 4;
  Y.d_misc = o; o += 4; Y.d_dsep = o; o += Y.NP * N; Y.d_ssl = o; o += N; Y.d_smax = o; o += N; Y.d_reg = o; o += C * P * REGSZ;
  Y.d_env = o; o += E * EL * 3; Y.d_obs = o; o += O * N * L * 3; Y.d_theta = o; o += C * 4; Y.d_lift = o; o += C * 2 * N * 16; Y.d_hull = o; o += C * N * HULLSZ;
  Y.PT = P < HULLM_MAXBITS ? P : HULLM_MAXBITS; Y.d_hullm = o; o += C * (1 << Y.PT) * 8; Y.d_fpk = o; o += C * (1 << Y.PT) * 8; o = (o + 1) & ~1; Y.d_astab = o; o += 18 * C * N; Y.dstride = (o + 7) & ~7;
  o = 0;
  Y.i_nposs = o; o += C; Y.i_regj = o; o += C * P; Y.i_nhs = o; o += C * P; Y.i_hs = o; o += C * P * 4; Y.i_envn = o; o += E;
  Y.i_obssoft = o; o += O; Y.i_initj = o; o += C; Y.i_dom = o; o += C * P * 4; Y.i_allow = o; o += C * N * 2; Y.i_boxskip = o; o += C * N; Y.i_c2callow = o; o += Y.NP * N; Y.i_rallow = o; o += C * N; Y.istride = (o + 3) & ~3;
  Y.f_reg = 0; Y.f_env = Y.f_reg + C * N; Y.f_obs = Y.f_env + C * N * 5; Y.f_c2c = Y.f_obs + C * O * N * 5;
  Y.f_c2n = Y.f_c2c + Y.NP * N * 4;
  Y.f_rmask = Y.f_c2n + Y.NP * N * 4;
  Y.fixlen = (Y.f_rmask + C * N * 2 + 15) & ~15;
  return Y;
}

// non-slow half-planes of a sector: sector \ slow-square = U_h sector ^ {sign*v_axis >= vm}, dominated ones dropped
inline int nonslow_halfplanes(const double* F, int hs[2][2]) {
  double t1 = std::atan2(F[1], F[0]), t2 = std::atan2(F[3], F[2]);
  if (t2 < t1) t2 += 2 * M_PI;
  const int cand[4][2] = {{0, 1}, {0, -1}, {1, 1}, {1, -1}};
  double vals[4][65]; bool ok[4];
  for (int q = 0; q < 4; ++q) {
    ok[q] = false;
    for (int s = 0; s < 65; ++s) {
      double th = t1 + (t2 - t1) * s / 64.0;
      vals[q][s] = cand[q][0] == 0 ? cand[q][1] * std::cos(th) : cand[q][1] * std::sin(th);
      if (vals[q][s] > 1e-12) ok[q] = true;
    }
  }
  int n = 0;
  for (int a = 0; a < 4; ++a) {
    if (!ok[a]) continue;
    bool dom = false;
    for (int b = 0; b < 4 && !dom; ++b) {
      if (a == b || !ok[b]) continue;
      bool all_ge = true, any_gt = false;
      for (int s = 0; s < 65; ++s) {
        if (vals[a][s] <= 1e-12) continue;
        if (vals[b][s] < vals[a][s] - 1e-12) all_ge = false;
        if (vals[b][s] > vals[a][s] + 1e-12) any_gt = true;
      }
      if (all_ge && (any_gt || b < a)) dom = true;
    }
    if (!dom && n < 2) { hs[n][0] = cand[a][0]; hs[n][1] = cand[a][1]; n++; }
  }
  return n;
}

inline int max_possible(const HostInst& I) {
  int P = 1;
  for (int c = 0; c < I.C; ++c) { int n = 0; for (int j = 0; j < I.R; ++j) n += I.possible[c * I.R + j] == 1; P = std::max(P, n); }
  return P;
}
inline int max_env_edges(const HostInst& I) {
  int m = 0; for (int e = 0; e < I.E; ++e) m = std::max(m, I.env_off[e + 1] - I.env_off[e]); return m;
}

// min of f(v) = a0 + a1*vx + a2*vy over the polygon {rows[k][0]*vx + rows[k][1]*vy <= rows[k][2]} (vertex enumeration);
// +inf when the polygon is empty
inline double min_affine_over_polygon(const std::vector<std::array<double, 3>>& rows, double a0, double a1, double a2) {
  double best = 1e300; const double tol = 1e-9;
  for (size_t a = 0; a < rows.size(); ++a)
    for (size_t b = a + 1; b < rows.size(); ++b) {
      double det = rows[a][0] * rows[b][1] - rows[a][1] * rows[b][0];
      if (std::fabs(det) < 1e-14) continue;
      double vx = (rows[a][2] * rows[b][1] - rows[a][1] * rows[b][2]) / det;
      double vy = (rows[a][0] * rows[b][2] - rows[a][2] * rows[b][0]) / det;
      bool ok = true;
      for (auto& r : rows) if (r[0] * vx + r[1] * vy > r[2] + tol * (1.0 + std::fabs(r[2]))) { ok = false; break; }
      if (ok) best = std::min(best, a0 + a1 * vx + a2 * vy);
    }
  return best;
}

// Response tables of the objective for the bound lifting of eval_kernel.  With the multipliers of a solved node fixed, its
// Lagrangian is D + 1/2 (z - z*)' H (z - z*) on the trajectories of the dynamics (H: the objective's Hessian 2 W per
// stage, nothing else); a child that adds the row g.z <= r, violated by v at z*, therefore costs at least
// D + v^2 / (2 g Sigma g'), Sigma = Zu H_u^-1 Zu' the response of the stage's (p, v, a, u) to the inputs under H.
// The chains (car, axis) are independent triple integrators with their own weights: one 4 x 4 block per (car, axis, stage).
inline void lift_tables(const HostInst& I, const Layout& Y, double* D) {
  const int N = Y.N, M = N - 1;   // inputs u_0 .. u_{N-2} (the last stage has no input)
  const double ts = I.ts;
  std::vector<double> R((size_t)N * 3), H((size_t)M * M), Zu((size_t)4 * M), Yv((size_t)4 * M);
  { double r[3] = {ts * ts * ts / 6.0, 0.5 * ts * ts, ts};   // A^k B
    for (int k = 0; k < N; ++k) { R[k * 3] = r[0]; R[k * 3 + 1] = r[1]; R[k * 3 + 2] = r[2]; r[0] += ts * r[1] + 0.5 * ts * ts * r[2]; r[1] += ts * r[2]; } }
  for (int c = 0; c < Y.C; ++c)
    for (int ax = 0; ax < 2; ++ax) {
      double* out = D + Y.d_lift + (size_t)((c * 2 + ax) * N) * 16;
      const double w[4] = {I.W[c * 8 + 3 * ax], I.W[c * 8 + 3 * ax + 1], I.W[c * 8 + 3 * ax + 2], I.W[c * 8 + 6 + ax]};
      std::fill(H.begin(), H.end(), 0.0);
      for (int i = 1; i < N; ++i)
        for (int j = 0; j < i && j < M; ++j)
          for (int k = 0; k <= j; ++k) {
            const double* a = &R[(i - 1 - j) * 3]; const double* b = &R[(i - 1 - k) * 3];
            H[(size_t)j * M + k] += 2.0 * (w[0] * a[0] * b[0] + w[1] * a[1] * b[1] + w[2] * a[2] * b[2]);
          }
      for (int j = 0; j < M; ++j) H[(size_t)j * M + j] += 2.0 * w[3];
      bool ok = true;   // Cholesky, lower triangle in place
      for (int j = 0; j < M && ok; ++j) {
        double d = H[(size_t)j * M + j];
        for (int k = 0; k < j; ++k) d -= H[(size_t)j * M + k] * H[(size_t)j * M + k];
        if (!(d > 1e-12)) { ok = false; break; }
        d = std::sqrt(d); H[(size_t)j * M + j] = d;
        for (int i = j + 1; i < M; ++i) { double v = H[(size_t)i * M + j]; for (int k = 0; k < j; ++k) v -= H[(size_t)i * M + k] * H[(size_t)j * M + k]; H[(size_t)i * M + j] = v / d; }
      }
      for (int i = 0; i < N; ++i) {
        if (!ok) { for (int q = 0; q < 16; ++q) out[i * 16 + q] = (q % 5 == 0) ? 1e300 : 0.0; continue; }   // no curvature: no lifting
        std::fill(Zu.begin(), Zu.end(), 0.0);
        for (int j = 0; j < i && j < M; ++j) for (int q = 0; q < 3; ++q) Zu[(size_t)q * M + j] = R[(i - 1 - j) * 3 + q];
        if (i < M) Zu[(size_t)3 * M + i] = 1.0;
        for (int q = 0; q < 4; ++q)   // Yv = L^-1 Zu'
          for (int j = 0; j < M; ++j) { double v = Zu[(size_t)q * M + j]; for (int k = 0; k < j; ++k) v -= H[(size_t)j * M + k] * Yv[(size_t)q * M + k]; Yv[(size_t)q * M + j] = v / H[(size_t)j * M + j]; }
        for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) { double v = 0; for (int j = 0; j < M; ++j) v += Yv[(size_t)a * M + j] * Yv[(size_t)b * M + j]; out[i * 16 + a * 4 + b] = v; }
      }
    }
}

// Tables of the active-set node solver (as_onchip.hip): the Hessian of every node QP of an instance is the objective's alone, so the regulator of
// the unconstrained problem - per chain (car, axis) a scalar-input triple integrator with weights (Wp, Wv, Wa | Wj) - is an instance constant:
//   KS   [N][2C][4]  feedback gain K (p, v, a) and 1 / S_uu per stage and chain (3 x 3 Riccati recursion, S_uu = 2 Wj + B' P B),
//   kref [N][2C]     feed-forward of the objective's linear term -2 W r,
//   zunc [N][8C]     the unconstrained optimum from x_0, columns chain-contiguous (4 ch + {p, v, a, u}).
// Computed HERE, once, so that every launch of every kernel variant reads the same bits (computed on the device by whichever wavefront came first,
// two template instantiations could publish last-bit-different values and repeated solves differed by one step).
inline void as_tables(const HostInst& I, const Layout& Y, double* D) {
  const int N = Y.N, NCH = 2 * Y.C;
  double* KS = D + Y.d_astab; double* kref = KS + (size_t)N * NCH * 4; double* zunc = kref + (size_t)N * NCH;
  const double h1 = I.ts, h2 = 0.5 * I.ts * I.ts, h3 = I.ts * I.ts * I.ts / 6.0;
  for (int ch = 0; ch < NCH; ++ch) {
    const int c = ch >> 1, s = ch & 1;
    const double q0 = 2.0 * I.W[c * 8 + 3 * s], q1 = 2.0 * I.W[c * 8 + 3 * s + 1], q2 = 2.0 * I.W[c * 8 + 3 * s + 2], rr = 2.0 * I.W[c * 8 + 6 + s];
    double p00 = q0, p01 = 0.0, p02 = 0.0, p11 = q1, p12 = 0.0, p22 = q2;
    for (int i = N - 2; i >= 0; --i) {
      const double t01 = p00 * h1 + p01, t02 = p00 * h2 + p01 * h1 + p02, t11 = p01 * h1 + p11, t12 = p01 * h2 + p11 * h1 + p12, t22 = p02 * h2 + p12 * h1 + p22;
      const double b0 = p00 * h3 + p01 * h2 + p02 * h1, b1 = p01 * h3 + p11 * h2 + p12 * h1, b2 = p02 * h3 + p12 * h2 + p22 * h1;
      const double suu = rr + h3 * b0 + h2 * b1 + h1 * b2;
      const double x0 = b0, x1 = h1 * b0 + b1, x2 = h2 * b0 + h1 * b1 + b2;
      const double s00 = q0 + p00, s01 = t01, s02 = t02, s11 = q1 + h1 * t01 + t11, s12 = h1 * t02 + t12, s22 = q2 + h2 * t02 + h1 * t12 + t22;
      const double is = 1.0 / std::max(suu, 1e-300);
      const double k0 = x0 * is, k1 = x1 * is, k2 = x2 * is;
      double* ks = KS + ((size_t)i * NCH + ch) * 4; ks[0] = k0; ks[1] = k1; ks[2] = k2; ks[3] = is;
      p00 = s00 - k0 * x0; p01 = s01 - k0 * x1; p02 = s02 - k0 * x2; p11 = s11 - k1 * x1; p12 = s12 - k1 * x2; p22 = s22 - k2 * x2;
    }
    // feed-forward of v = -2 W r (no reference on the jerk), then the roll-out from x_0
    const double w[3] = {I.W[c * 8 + 3 * s], I.W[c * 8 + 3 * s + 1], I.W[c * 8 + 3 * s + 2]};
    auto vref = [&](int i, int k) { return -2.0 * w[k] * I.ref[((size_t)c * N + i) * 6 + 3 * s + k]; };
    double p0 = vref(N - 1, 0), p1 = vref(N - 1, 1), p2 = vref(N - 1, 2);
    for (int i = N - 2; i >= 0; --i) {
      const double* ks = KS + ((size_t)i * NCH + ch) * 4;
      const double su = h3 * p0 + h2 * p1 + h1 * p2;
      const double sx0 = vref(i, 0) + p0, sx1 = vref(i, 1) + h1 * p0 + p1, sx2 = vref(i, 2) + h2 * p0 + h1 * p1 + p2;
      kref[(size_t)i * NCH + ch] = su * ks[3];
      p0 = sx0 - ks[0] * su; p1 = sx1 - ks[1] * su; p2 = sx2 - ks[2] * su;
    }
    kref[(size_t)(N - 1) * NCH + ch] = 0.0;
    double x0 = I.x0[c * 6 + 3 * s], x1 = I.x0[c * 6 + 3 * s + 1], x2 = I.x0[c * 6 + 3 * s + 2];
    for (int i = 0; i < N; ++i) {
      double u = 0.0;
      if (i < N - 1) { const double* ks = KS + ((size_t)i * NCH + ch) * 4; u = -kref[(size_t)i * NCH + ch] - (ks[0] * x0 + ks[1] * x1 + ks[2] * x2); }
      double* z = zunc + (size_t)i * 4 * NCH + 4 * ch; z[0] = x0; z[1] = x1; z[2] = x2; z[3] = u;
      const double n0 = x0 + h1 * x1 + h2 * x2 + h3 * u, n1 = x1 + h1 * x2 + h2 * u, n2 = x2 + h1 * u;
      x0 = n0; x1 = n1; x2 = n2;
    }
  }
}

// fills one instance's block of the device tables
inline void compile_instance(const HostInst& I, const Layout& Y, double* D, int* T) {
  std::fill(D, D + Y.dstride, 0.0); std::fill(T, T + Y.istride, 0);
  int C = I.C, N = I.N, R = I.R;
  for (int k = 0; k < C * 6; ++k) D[Y.d_x0 + k] = I.x0[k];
  for (int c = 0; c < C; ++c) {
    for (int k = 0; k < 6; ++k) D[Y.d_wd + 6 * c + k] = I.W[c * 8 + k];
    D[Y.d_wd + 6 * C + 2 * c] = I.W[c * 8 + 6]; D[Y.d_wd + 6 * C + 2 * c + 1] = I.W[c * 8 + 7];
    for (int i = 0; i < N; ++i) for (int k = 0; k < 6; ++k) D[Y.d_ref + i * Y.nz + 6 * c + k] = I.ref[((size_t)c * N + i) * 6 + k];
    double th = std::atan2(I.x0[c * 6 + 4], I.x0[c * 6 + 1]);
    D[Y.d_theta + c * 4 + 0] = I.x0[c * 6 + 0] + std::cos(th) * I.wb[c]; D[Y.d_theta + c * 4 + 1] = I.x0[c * 6 + 3] + std::sin(th) * I.wb[c];
  }
  lift_tables(I, Y, D);
  as_tables(I, Y, D);
  double* G = D + Y.d_glob; G[0] = I.vmin; G[1] = I.vmax; G[2] = I.amin; G[3] = I.amax; G[4] = I.jmin; G[5] = I.jmax; G[6] = I.vm; G[7] = I.ts;
  for (int c = 0; c < C; ++c) {  // initial_conditions.mod:30-48
    int j0 = I.init_region[c] - 1;
    for (int s = 0; s < 2; ++s) {
      double hi = I.jmax, lo = I.jmin;
      for (int j = 0; j < R; ++j) {
        double M = j == j0 ? 0.0 : BIGM_JERK; const double* jl = &I.jerk_lim[((size_t)c * R + j) * 4];
        hi = std::min(hi, jl[2 * s + 1] + M); lo = std::max(lo, jl[2 * s] - M);
      }
      D[Y.d_u0box + c * 4 + 2 * s] = lo; D[Y.d_u0box + c * 4 + 2 * s + 1] = hi;
    }
  }
  D[Y.d_misc + 0] = I.w_slack; D[Y.d_misc + 1] = I.w_slack_obs;
  { int p = 0;
    for (int a = 0; a < C; ++a) for (int b = a + 1; b < C; ++b) { for (int i = 0; i < N; ++i) D[Y.d_dsep + p * N + i] = I.rad[a] + I.rad[b] + I.safety[i]; p++; } }
  for (int i = 0; i < N; ++i) { D[Y.d_ssl + i] = I.safety_slack[i]; D[Y.d_smax + i] = std::max(0.0, std::min(I.safety_slack[i], I.max_slack)); }
  for (int c = 0; c < C; ++c) {
    std::vector<int> pl; for (int j = 0; j < R; ++j) if (I.possible[c * R + j] == 1) pl.push_back(j);
    T[Y.i_nposs + c] = (int)pl.size(); T[Y.i_initj + c] = I.init_region[c] - 1;
    for (int q = 0; q < (int)pl.size(); ++q) {
      int j = pl[q]; T[Y.i_regj + c * Y.P + q] = j;
      double* g = D + Y.d_reg + (c * Y.P + q) * REGSZ; const double* F = &I.frac[j * 4];
      double n1 = std::hypot(F[0], F[1]), n3 = std::hypot(F[2], F[3]);
      g[0] = F[1] / n1; g[1] = -F[0] / n1;   //  F2*vx - F1*vy <= 0
      g[2] = -F[3] / n3; g[3] = F[2] / n3;   // -F4*vx + F3*vy <= 0
      g[4] = (F[1] + F[3]) / (F[0] + F[2]);
      for (int k = 0; k < 3; ++k) { g[5 + k] = I.poly[4][j * 3 + k]; g[8 + k] = I.poly[5][j * 3 + k]; }
      for (int s = 0; s < 2; ++s) {
        double hi = I.amax, lo = I.amin, jh = I.jmax, jlo = I.jmin;
        for (int jj : pl) {
          double Ma = jj == j ? 0.0 : BIGM_ACC, Mj = jj == j ? 0.0 : BIGM_JERK;
          const double* al = &I.acc_lim[((size_t)c * R + jj) * 4]; const double* jl = &I.jerk_lim[((size_t)c * R + jj) * 4];
          hi = std::min(hi, al[2 * s + 1] + Ma); lo = std::max(lo, al[2 * s] - Ma);
          jh = std::min(jh, jl[2 * s + 1] + Mj); jlo = std::max(jlo, jl[2 * s] - Mj);
        }
        g[11 + 2 * s] = lo; g[12 + 2 * s] = hi; g[15 + 2 * s] = jlo; g[16 + 2 * s] = jh;
      }
      const int order[4] = {2, 3, 0, 1};  // COSS_UB, COSS_LB, SINT_UB, SINT_LB
      for (int t = 0; t < 4; ++t) for (int k = 0; k < 3; ++k) g[19 + 3 * t + k] = I.wb[c] * I.poly[order[t]][j * 3 + k];
      int hs[2][2] = {{0, 1}, {0, 1}}; int nh = nonslow_halfplanes(F, hs);
      T[Y.i_nhs + c * Y.P + q] = nh;
      for (int h = 0; h < 2; ++h) { T[Y.i_hs + ((c * Y.P + q) * 2 + h) * 2] = hs[h][0]; T[Y.i_hs + ((c * Y.P + q) * 2 + h) * 2 + 1] = hs[h][1]; }
      // corner dominance (exact presolve): pos_x_front_UB >= pos_x_front_LB and pos_y_front_UB >= pos_y_front_LB hold on the
      // whole velocity set of the alternative (sector + half-plane or slow square, global bounds, reachable |vy|) ->
      // of the rows of one polygon edge on the four front corners only the worst corner's row can be active
      {
        double vyb = 0.0; for (int cc = 0; cc < C; ++cc) vyb = std::max(vyb, std::fabs(I.x0[cc * 6 + 4]));
        vyb += std::max(std::fabs(I.amax), std::fabs(I.amin)) * I.ts * N + 1.0;
        for (int h = 0; h < 4; ++h) {
          int flag = 0;
          if (h < 3 && h >= nh) { T[Y.i_dom + (c * Y.P + q) * 4 + h] = 0; continue; }
          std::vector<std::array<double, 3>> rows;
          rows.push_back({-1.0, 0.0, -I.vmin}); rows.push_back({1.0, 0.0, I.vmax}); rows.push_back({0.0, -1.0, std::min(-I.vmin, vyb)});
          rows.push_back({0.0, 1.0, vyb}); rows.push_back({0.0, -1.0, vyb});
          if (h == 3) { rows.push_back({1, 0, I.vm}); rows.push_back({-1, 0, I.vm}); rows.push_back({0, 1, I.vm}); rows.push_back({0, -1, I.vm}); }
          else {
            rows.push_back({g[0], g[1], 0.0}); rows.push_back({g[2], g[3], 0.0});
            double sg = hs[h][1]; if (hs[h][0] == 0) rows.push_back({-sg, 0.0, -I.vm}); else rows.push_back({0.0, -sg, -I.vm});
          }
          const double m = 1e-9;
          double mx = min_affine_over_polygon(rows, g[19] - g[22], g[20] - g[23], g[21] - g[24]);
          double my = min_affine_over_polygon(rows, g[25] - g[28], g[26] - g[29], g[27] - g[30]);
          if (mx >= -m) flag |= 1;
          if (my >= -m) flag |= 2;
          T[Y.i_dom + (c * Y.P + q) * 4 + h] = flag;
        }
      }
    }
  }
  // Reachability presolve (exact): with every acceleration in [total_min_acc, total_max_acc] (A3 rows) the velocity of
  // step i lies in v0 + i*ts*[amin', amax'] (v_{k+1} = v_k + ts*(a_k + a_{k+1})/2).  A region alternative whose velocity
  // set (sector and half-plane, or the slow square) misses that box can never hold -> it is never branched on.
  for (int c = 0; c < C; ++c) {
    const double alo = std::min(I.amin, std::min(I.x0[c * 6 + 2], I.x0[c * 6 + 5])), ahi = std::max(I.amax, std::max(I.x0[c * 6 + 2], I.x0[c * 6 + 5]));
    const int np = T[Y.i_nposs + c];
    for (int i = 0; i < N; ++i) {
      unsigned long long mask = 0ull;
      double t = i * I.ts;
      double bx0 = std::max(I.vmin, I.x0[c * 6 + 1] + t * alo), bx1 = std::min(I.vmax, I.x0[c * 6 + 1] + t * ahi);
      double by0 = std::max(I.vmin, I.x0[c * 6 + 4] + t * alo), by1 = I.x0[c * 6 + 4] + t * ahi;
      const double pad = 1e-7;
      for (int q = 0; q < np && i >= 1; ++q) {
        const double* g = D + Y.d_reg + (c * Y.P + q) * REGSZ;
        for (int h = 0; h < 4; ++h) {
          if (h < 3 && h >= T[Y.i_nhs + c * Y.P + q]) continue;
          std::vector<std::array<double, 3>> rows;
          rows.push_back({-1.0, 0.0, -bx0 + pad}); rows.push_back({1.0, 0.0, bx1 + pad});
          rows.push_back({0.0, -1.0, -by0 + pad}); rows.push_back({0.0, 1.0, by1 + pad});
          if (h == 3) { rows.push_back({1, 0, I.vm + pad}); rows.push_back({-1, 0, I.vm + pad}); rows.push_back({0, 1, I.vm + pad}); rows.push_back({0, -1, I.vm + pad}); }
          else {
            rows.push_back({g[0], g[1], pad}); rows.push_back({g[2], g[3], pad});
            const int* hs = T + Y.i_hs + ((c * Y.P + q) * 2 + h) * 2;
            if (hs[0] == 0) rows.push_back({-(double)hs[1], 0.0, -I.vm + pad}); else rows.push_back({0.0, -(double)hs[1], -I.vm + pad});
          }
          if (min_affine_over_polygon(rows, 0.0, 0.0, 0.0) < 1e299) mask |= 1ull << (q * 4 + h);
        }
      }
      T[Y.i_allow + (c * N + i) * 2] = (int)(mask & 0xFFFFFFFFull); T[Y.i_allow + (c * N + i) * 2 + 1] = (int)(mask >> 32);
    }
  }
  // Hull of the region disjunction (exact, valid inequalities of the MIQP): exactly one possible region is active at every step
  // (model_region_constraints.mod:43-117, sum_j active_region = 1), so while the region of (car, step) is undecided
  //   * acceleration and jerk lie in the smallest box around the boxes of the regions that can still hold there (instead of
  //     total_min/max_acc/jerk, the extreme over ALL regions of the table: 4.3 against 2.4 m/s^2 forward on the bench instances -
  //     the relaxation of an undecided step used to accelerate twice as hard as any region allows), and
  //   * the velocity lies in the cone spanned by their sectors (when that cone is convex), widened by the slow square when a
  //     slow alternative is possible (|vx|, |vy| <= v_m leaves every sector row relaxed: minimum_speed_constraints.mod:9-49).
  // "Can still hold": some alternative of the region passes the reachability presolve above.
  for (int c = 0; c < C; ++c) {
    const int np = T[Y.i_nposs + c];
    for (int i = 0; i < N; ++i) {
      double* H = D + Y.d_hull + (size_t)(c * N + i) * HULLSZ;
      H[0] = I.amin; H[1] = I.amax; H[2] = I.amin; H[3] = I.amax; H[4] = I.jmin; H[5] = I.jmax; H[6] = I.jmin; H[7] = I.jmax;
      for (int k = 8; k < HULLSZ; ++k) H[k] = 0.0;
      if (i < 1) continue;
      const unsigned long long mask = ((unsigned long long)(unsigned int)T[Y.i_allow + (c * N + i) * 2 + 1] << 32) | (unsigned int)T[Y.i_allow + (c * N + i) * 2];
      double lo[8]; bool any = false, any_slow = false;
      std::vector<std::array<double, 2>> cones; std::vector<int> cone_q;
      for (int q = 0; q < np; ++q) {
        const unsigned m4 = (unsigned)((mask >> (q * 4)) & 15ull);
        if (!m4) continue;
        const double* g = D + Y.d_reg + (c * Y.P + q) * REGSZ;
        const double b[8] = {g[11], g[12], g[13], g[14], g[15], g[16], g[17], g[18]};
        for (int k = 0; k < 8; ++k) lo[k] = !any ? b[k] : ((k & 1) ? std::max(lo[k], b[k]) : std::min(lo[k], b[k]));
        any = true;
        if (m4 & 8u) any_slow = true;
        if (m4 & 7u) {
          const double* F = &I.frac[T[Y.i_regj + c * Y.P + q] * 4];
          double t1 = std::atan2(F[1], F[0]), t2 = std::atan2(F[3], F[2]);
          if (t1 < 0) t1 += 2 * M_PI;
          while (t2 < t1) t2 += 2 * M_PI;
          cones.push_back({t1, t2}); cone_q.push_back(q);
        }
      }
      if (!any) continue;   // (no alternative is reachable: the instance is infeasible and the search will say so)
      // never wider than the global rows that stay in force anyway
      H[0] = std::max(H[0], lo[0]); H[1] = std::min(H[1], lo[1]); H[2] = std::max(H[2], lo[2]); H[3] = std::min(H[3], lo[3]);
      H[4] = std::max(H[4], lo[4]); H[5] = std::min(H[5], lo[5]); H[6] = std::max(H[6], lo[6]); H[7] = std::min(H[7], lo[7]);
      if (cones.empty()) continue;
      // The cone rows are OFF by default (MIQP_HULL_CONE=1 switches them on): measured on the bench instances they do not change
      // the node counts (the velocity of an undecided step almost never leaves the cone: 7 % of the region branchings were
      // flagged by a sector row, and those are settled by the branching itself), but two more general rows per undecided
      // (car, step) push two thirds of the nodes over the on-chip kernel's 128 general rows into the memory-backed kernel
      // (+25 % interior point time).  The boxes are box rows of the kernels: free.
      { static const bool cone = KNOB_T("MIQP_HULL_CONE") != nullptr && std::atoi(KNOB_T("MIQP_HULL_CONE")) != 0; if (!cone) continue; }
      // the cone around all sectors = complement of the widest angular gap between them
      std::vector<int> ord(cones.size()); for (size_t k = 0; k < ord.size(); ++k) ord[k] = (int)k;
      std::sort(ord.begin(), ord.end(), [&](int a, int b) { return cones[a][0] < cones[b][0]; });
      double best_gap = -1.0; int first = 0;   // the hull starts at the sector that follows the widest gap
      double reach = cones[ord[0]][1];          // running end of the covered arc
      for (size_t k = 0; k < ord.size(); ++k) {
        const size_t nx = (k + 1) % ord.size();
        double end_k = cones[ord[k]][1]; if (k == 0) reach = end_k; else reach = std::max(reach, end_k);
        double start_n = cones[ord[nx]][0] + (nx == 0 ? 2 * M_PI : 0.0);
        const double gap = start_n - reach;
        if (gap > best_gap) { best_gap = gap; first = (int)nx; }
      }
      const int qa = cone_q[ord[first]];
      // last sector of the hull: the one whose end is the largest when walking from `first` around the circle
      double a0 = cones[ord[first]][0], amax_end = -1e300; int qb = qa;
      for (size_t k = 0; k < ord.size(); ++k) {
        double st = cones[ord[k]][0], en = cones[ord[k]][1];
        while (st < a0 - 1e-12) { st += 2 * M_PI; en += 2 * M_PI; }
        if (en > amax_end) { amax_end = en; qb = cone_q[ord[k]]; }
      }
      if (!(amax_end - a0 < M_PI - 1e-6)) continue;   // not a convex cone: no rows
      const double* ga = D + Y.d_reg + (c * Y.P + qa) * REGSZ; const double* gb = D + Y.d_reg + (c * Y.P + qb) * REGSZ;
      const double pad = 1e-9;
      H[8] = ga[0]; H[9] = ga[1]; H[10] = pad + (any_slow ? I.vm * (std::fabs(ga[0]) + std::fabs(ga[1])) : 0.0);
      H[11] = gb[2]; H[12] = gb[3]; H[13] = pad + (any_slow ? I.vm * (std::fabs(gb[2]) + std::fabs(gb[3])) : 0.0);
      H[14] = 1.0;
    }
  }
  // Region sets: i_rallow[c][i] = possible regions with a reachable alternative at step i (the static part of a node's region
  // set); d_hullm[c][set] = smallest box around the acceleration / jerk boxes of the regions of `set` (bit q = possible region
  // q; the empty set and sets beyond the table get the global box)
  for (int c = 0; c < C; ++c) {
    const int np = T[Y.i_nposs + c];
    for (int i = 0; i < N; ++i) {
      const unsigned long long mask = ((unsigned long long)(unsigned int)T[Y.i_allow + (c * N + i) * 2 + 1] << 32) | (unsigned int)T[Y.i_allow + (c * N + i) * 2];
      int rm = 0; for (int q = 0; q < np; ++q) if ((mask >> (q * 4)) & 15ull) rm |= 1 << q;
      T[Y.i_rallow + c * N + i] = rm;
    }
    for (int m = 0; m < (1 << Y.PT); ++m) {
      double* H = D + Y.d_hullm + (size_t)(c * (1 << Y.PT) + m) * 8;
      const double glob[8] = {I.amin, I.amax, I.amin, I.amax, I.jmin, I.jmax, I.jmin, I.jmax};
      bool any = false; double b[8];
      for (int q = 0; q < np && q < Y.PT; ++q) {
        if (!((m >> q) & 1)) continue;
        const double* g = D + Y.d_reg + (c * Y.P + q) * REGSZ;
        for (int k = 0; k < 8; ++k) b[k] = !any ? g[11 + k] : ((k & 1) ? std::max(b[k], g[11 + k]) : std::min(b[k], g[11 + k]));
        any = true;
      }
      for (int k = 0; k < 8; ++k) H[k] = !any ? glob[k] : ((k & 1) ? std::min(glob[k], b[k]) : std::max(glob[k], b[k]));
    }
    // d_fpk[c][set] = range of the front-point offsets over the regions of `set`: the front axle point of car c is
    //   (pos_x + o_x, pos_y + o_y),  o = wheel base x (p1 + p2 vx + p3 vy) of the active region's fitted polynomial
    // (model_region_constraints.mod:56-69; upper / lower variants).  While the region of a step is undecided a row on a front
    // point is kept in the relaxation with the offset replaced by its bound over the regions still possible (minimum where the
    // offset enters with a positive coefficient, maximum otherwise) - implied by the row of whichever region turns out active.
    // Per region the bound is taken over its velocity set: the sector inside the velocity box, and the slow square (a slow
    // alternative keeps the region's polynomial at any velocity with |vx|, |vy| <= v_m).  Entries: x upper-variant (min, max), x
    // lower-variant (min, max), y upper-variant (min, max), y lower-variant (min, max).
    {
      double vyb = 0.0; for (int cc = 0; cc < C; ++cc) vyb = std::max(vyb, std::fabs(I.x0[cc * 6 + 4]));
      vyb += std::max(std::fabs(I.amax), std::fabs(I.amin)) * I.ts * N + 1.0;
      std::vector<std::array<double, 8>> per(np);
      for (int q = 0; q < np; ++q) {
        const double* g = D + Y.d_reg + (c * Y.P + q) * REGSZ;
        std::vector<std::array<double, 3>> sector, slow;
        sector.push_back({-1.0, 0.0, -I.vmin}); sector.push_back({1.0, 0.0, I.vmax}); sector.push_back({0.0, 1.0, vyb}); sector.push_back({0.0, -1.0, vyb});
        sector.push_back({g[0], g[1], 1e-9}); sector.push_back({g[2], g[3], 1e-9});
        slow.push_back({1, 0, I.vm}); slow.push_back({-1, 0, I.vm}); slow.push_back({0, 1, I.vm}); slow.push_back({0, -1, I.vm});
        for (int t = 0; t < 4; ++t) {   // g[19 + 3 t ..]: x upper, x lower, y upper, y lower (already scaled by the wheel base)
          const double a0 = g[19 + 3 * t], a1 = g[20 + 3 * t], a2 = g[21 + 3 * t];
          double lo = std::min(min_affine_over_polygon(sector, a0, a1, a2), min_affine_over_polygon(slow, a0, a1, a2));
          double hi = -std::min(min_affine_over_polygon(sector, -a0, -a1, -a2), min_affine_over_polygon(slow, -a0, -a1, -a2));
          if (!(lo < 1e299)) lo = -1e6;   // (empty sector polygon: no information)
          if (!(hi > -1e299)) hi = 1e6;
          per[q][2 * t] = lo - 1e-9; per[q][2 * t + 1] = hi + 1e-9;
        }
      }
      for (int m = 0; m < (1 << Y.PT); ++m) {
        double* K = D + Y.d_fpk + (size_t)(c * (1 << Y.PT) + m) * 8;
        bool any = false;
        for (int q = 0; q < np && q < Y.PT; ++q) {
          if (!((m >> q) & 1)) continue;
          for (int t = 0; t < 4; ++t) { K[2 * t] = !any ? per[q][2 * t] : std::min(K[2 * t], per[q][2 * t]); K[2 * t + 1] = !any ? per[q][2 * t + 1] : std::max(K[2 * t + 1], per[q][2 * t + 1]); }
          any = true;
        }
        if (!any) for (int t = 0; t < 4; ++t) { K[2 * t] = -1e6; K[2 * t + 1] = 1e6; }
      }
    }
  }
  // Box presolve (exact): interval propagation of (a, v) per axis from the initial state with the jerk and acceleration
  // bounds that hold for every region (a+ = a + ts u, v+ = v + ts (a + a+)/2).  A velocity / acceleration bound row of
  // step i that the reachable interval already satisfies is implied by the rows of the earlier steps and is not
  // generated (bit rr of i_boxskip: rows 0..6 of decode_row).
  std::vector<double> preach((size_t)C * N * 4, 0.0);   // reachable position box per car and step: x lo, x hi, y lo, y hi
  double diam = 0.0;   // L1 diameter of the reachable set over all stage variables of the horizon (what a stationarity residual can add to a bound: kernels, batch_bound)
  for (int c = 0; c < C; ++c) {
    const int np = T[Y.i_nposs + c];
    double abox[4] = {I.amin, I.amax, I.amin, I.amax};     // lo_x, hi_x, lo_y, hi_y that every alternative respects
    double atest[4] = {I.amin, I.amax, I.amin, I.amax};    // tightest right-hand sides a row can carry
    for (int q = 0; q < np; ++q) { const double* g = D + Y.d_reg + (c * Y.P + q) * REGSZ;
      atest[0] = std::max(atest[0], g[11]); atest[1] = std::min(atest[1], g[12]); atest[2] = std::max(atest[2], g[13]); atest[3] = std::min(atest[3], g[14]); }
    double alo[2] = {I.x0[c * 6 + 2], I.x0[c * 6 + 5]}, ahi[2] = {alo[0], alo[1]};
    double vlo[2] = {I.x0[c * 6 + 1], I.x0[c * 6 + 4]}, vhi[2] = {vlo[0], vlo[1]};
    const double jlo = std::min(I.jmin, std::min(D[Y.d_u0box + c * 4], D[Y.d_u0box + c * 4 + 2])), jhi = std::max(I.jmax, std::max(D[Y.d_u0box + c * 4 + 1], D[Y.d_u0box + c * 4 + 3]));
    const double pad = 1e-9;
    double plo[2] = {I.x0[c * 6 + 0], I.x0[c * 6 + 3]}, phi[2] = {plo[0], plo[1]};
    for (int k = 0; k < 2; ++k) { preach[((size_t)c * N) * 4 + 2 * k] = plo[k]; preach[((size_t)c * N) * 4 + 2 * k + 1] = phi[k]; }
    for (int i = 1; i < N; ++i) {
      int skip = 0;
      for (int ax = 0; ax < 2; ++ax) {
        // x+ = x + ts v + ts^2/2 a + ts^3/6 u with the intervals of step i-1 (monotone in every term)
        plo[ax] += I.ts * vlo[ax] + 0.5 * I.ts * I.ts * alo[ax] + I.ts * I.ts * I.ts / 6.0 * jlo;
        phi[ax] += I.ts * vhi[ax] + 0.5 * I.ts * I.ts * ahi[ax] + I.ts * I.ts * I.ts / 6.0 * jhi;
        preach[((size_t)c * N + i) * 4 + 2 * ax] = plo[ax]; preach[((size_t)c * N + i) * 4 + 2 * ax + 1] = phi[ax];
        double nlo = alo[ax] + I.ts * jlo, nhi = ahi[ax] + I.ts * jhi;           // a_i before its own box
        double wlo = vlo[ax] + 0.5 * I.ts * (alo[ax] + nlo), whi = vhi[ax] + 0.5 * I.ts * (ahi[ax] + nhi);
        // acceleration rows of step i: rr 3 (ax <= hi), 4 (ax >= lo), 5 / 6 for y
        if (nhi <= atest[2 * ax + 1] - pad) skip |= 1 << (ax == 0 ? 3 : 5);
        if (nlo >= atest[2 * ax] + pad) skip |= 1 << (ax == 0 ? 4 : 6);
        nlo = std::max(nlo, abox[2 * ax]); nhi = std::min(nhi, abox[2 * ax + 1]);
        wlo = vlo[ax] + 0.5 * I.ts * (alo[ax] + nlo); whi = vhi[ax] + 0.5 * I.ts * (ahi[ax] + nhi);
        // velocity rows: rr 0 (vx >= vmin), 1 (vy >= vmin), 2 (vx <= vmax)
        if (wlo >= I.vmin + pad) skip |= 1 << (ax == 0 ? 0 : 1);
        if (ax == 0 && whi <= I.vmax - pad) skip |= 1 << 2;
        alo[ax] = nlo; ahi[ax] = nhi;
        vlo[ax] = std::max(wlo, I.vmin); vhi[ax] = ax == 0 ? std::min(whi, I.vmax) : whi;
        diam += (phi[ax] - plo[ax]) + (vhi[ax] - vlo[ax]) + (ahi[ax] - alo[ax]) + (jhi - jlo);
      }
      T[Y.i_boxskip + c * N + i] = skip;
    }
  }
  // (twice the diameter: |z' - z|_1 is taken between the interior point ITERATE z and a point z' of the box, and the iterate may lie outside the box
  // by the elastic slack of its velocity / acceleration / jerk rows - those are penalised, not hard - while its distance to the box never exceeds
  // the box's own size once the penalty term is what the cutoff test carries anyway; the inputs of step 0 are counted through jhi - jlo of step 1)
  D[Y.d_misc + 2] = std::max(1.0, 2.0 * diam);
  // Car/car alternatives that no reachable pair of positions can satisfy (exact): alternative a of group g at step i
  // asks  coord(A) - coord(B) <= -(separation) for a rear or front point of each car; front points lie within one wheel
  // base of the rear point.  Bit (4 g + a) of i_c2callow[pair][step] is set when the alternative is possible.
  { int p = 0;
    for (int c1 = 0; c1 < C; ++c1) for (int c2 = c1 + 1; c2 < C; ++c2, ++p)
      for (int i = 0; i < N; ++i) {
        int mask = 0;
        for (int g = 0; g < 4; ++g) for (int a = 0; a < 4; ++a) {
          const bool isx = a < 2, lo = (a == 0 || a == 2), soft = (g == 0 || g == 3);
          int ca, cb; bool fa, fb;   // cars and "is a front point" of the two sides, as in decode_row
          if (g == 0) { ca = lo ? c1 : c2; cb = lo ? c2 : c1; fa = fb = false; }
          else if (g == 1) { if (lo) { ca = c1; fa = false; cb = c2; fb = true; } else { ca = c2; fa = true; cb = c1; fb = false; } }
          else if (g == 2) { if (lo) { ca = c2; fa = false; cb = c1; fb = true; } else { ca = c1; fa = true; cb = c2; fb = false; } }
          else { if (lo) { ca = c2; cb = c1; } else { ca = c1; cb = c2; } fa = fb = true; }
          const double sep = I.rad[c1] + I.rad[c2] + I.safety[i] + (soft ? I.safety_slack[i] - std::max(0.0, std::min(I.safety_slack[i], I.max_slack)) : 0.0);
          const int ax = isx ? 0 : 1;
          const double amin_ = preach[((size_t)ca * N + i) * 4 + 2 * ax] - (fa ? I.wb[ca] : 0.0);
          const double bmax_ = preach[((size_t)cb * N + i) * 4 + 2 * ax + 1] + (fb ? I.wb[cb] : 0.0);
          if (amin_ - bmax_ <= -sep + 1e-6) mask |= 1 << (4 * g + a);
        }
        T[Y.i_c2callow + p * N + i] = mask;
      } }
  for (int e = 0; e < I.E; ++e) {
    int n = I.env_off[e + 1] - I.env_off[e]; T[Y.i_envn + e] = n;
    for (int k = 0; k < n; ++k) {  // inside: cross >= 0  <=>  dy*X - dx*Y <= dy*x1 - dx*y1
      const double* ed = &I.env_edges[(size_t)(I.env_off[e] + k) * 4];
      double dx = ed[2] - ed[0], dy = ed[3] - ed[1], al = dy, be = -dx, nr = std::hypot(al, be);
      double* g = D + Y.d_env + (e * Y.EL + k) * 3;
      if (nr > 0) { g[0] = al / nr; g[1] = be / nr; g[2] = (al * ed[0] + be * ed[1]) / nr; }
    }
  }
  for (int o = 0; o < I.O; ++o) {
    T[Y.i_obssoft + o] = I.obs_soft[o];
    for (int i = 0; i < N; ++i) for (int k = 0; k < I.L; ++k) {  // separated: cross <= 0
      const double* ed = &I.obs_edges[((size_t)(o * N + i) * I.L + k) * 4];
      double dx = ed[2] - ed[0], dy = ed[3] - ed[1], al = -dy, be = dx, nr = std::hypot(al, be);
      double* g = D + Y.d_obs + ((o * N + i) * I.L + k) * 3;
      if (nr > 0) { g[0] = al / nr; g[1] = be / nr; g[2] = (al * ed[0] + be * ed[1]) / nr; }
    }
  }
}

}  // namespace miqp
