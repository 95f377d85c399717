/* inst.c - instance construction for the CPU oracle (test infrastructure).
 *
 * orc_from_params follows ModelInputDataSource::read (src/model_input_data_source.cpp:180-275):
 *   every float is rounded to `round_decimals` decimals (RoundWithPrecision, hpp:74-79; the wrapper
 *   passes precision-2 = 10, src/cplex_wrapper.hpp:88) and polygons become edge tuples with
 *   wrap-around (addLineSet, cpp:167-178).
 * orc_from_dat reads the OPL .dat subset of the fixtures: no rounding.
 */
#include "oracle.h"

#include <ctype.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static double rnd(double v, int dec) {
  if (dec < 0) return v;
  double s = pow(10.0, dec);
  return round(v * s) / s;
}

static double* dalloc(int n) { return (double*)calloc((size_t)(n > 0 ? n : 1), sizeof(double)); }
static int* ialloc(int n) { return (int*)calloc((size_t)(n > 0 ? n : 1), sizeof(int)); }

static oinst* inst_alloc(int N, int C, int R, int E, int O, int L, int env_edges_total) {
  oinst* I = (oinst*)calloc(1, sizeof(oinst));
  I->N = N; I->C = C; I->R = R; I->E = E; I->O = O; I->L = L; I->K = C - 1; I->NP = C * (C - 1) / 2;
  I->safety = dalloc(N); I->safety_slack = dalloc(N);
  I->W = dalloc(C * 8); I->wb = dalloc(C); I->rad = dalloc(C); I->x0 = dalloc(C * 6);
  I->ref = dalloc(C * N * 6);
  I->acc_lim = dalloc(C * R * 4); I->jerk_lim = dalloc(C * R * 4);
  I->init_region = ialloc(C); I->possible = ialloc(C * R);
  I->frac = dalloc(R * 4);
  for (int k = 0; k < 6; ++k) I->poly[k] = dalloc(R * 3);
  I->env_off = ialloc(E + 1); I->env_edges = dalloc(4 * env_edges_total);
  I->obs_edges = dalloc(O * N * L * 4); I->obs_soft = ialloc(O);
  return I;
}

void orc_free(oinst* I) {
  if (!I) return;
  free(I->safety); free(I->safety_slack); free(I->W); free(I->wb); free(I->rad); free(I->x0); free(I->ref);
  free(I->acc_lim); free(I->jerk_lim); free(I->init_region); free(I->possible); free(I->frac);
  for (int k = 0; k < 6; ++k) free(I->poly[k]);
  free(I->env_off); free(I->env_edges); free(I->obs_edges); free(I->obs_soft);
  free(I);
}

oinst* orc_from_params(const miqp_model_params_c* p, int dec) {
  int N = p->NumSteps, C = p->NumCars, R = p->nr_regions, E = p->nr_environments, O = p->nr_obstacles,
      L = p->max_lines_obstacles;
  if (C < 1 || C > ORC_MAXC || N < 2) return NULL;
  int tot = E > 0 ? p->env_offsets[E] : 0;
  oinst* I = inst_alloc(N, C, R, E, O, L, tot);
  I->ts = rnd(p->ts, dec);
  I->vmin = rnd(p->min_vel_x_y, dec); I->vmax = rnd(p->max_vel_x_y, dec);
  I->amin = rnd(p->total_min_acc, dec); I->amax = rnd(p->total_max_acc, dec);
  I->jmin = rnd(p->total_min_jerk, dec); I->jmax = rnd(p->total_max_jerk, dec);
  I->max_slack = rnd(p->maximum_slack, dec);
  I->w_slack = rnd(p->WEIGHTS_SLACK, dec); I->w_slack_obs = rnd(p->WEIGHTS_SLACK_OBSTACLE, dec);
  I->vm = rnd(p->minimum_region_change_speed, dec);
  I->gap = rnd(p->relative_mip_gap_tolerance, dec); I->tilim = rnd(p->max_solution_time, dec);
  for (int i = 0; i < N; ++i) {
    I->safety[i] = rnd(p->agent_safety_distance[i], dec);
    I->safety_slack[i] = rnd(p->agent_safety_distance_slack[i], dec);
  }
  const double* Wsrc[8] = {p->WEIGHTS_POS_X, p->WEIGHTS_VEL_X, p->WEIGHTS_ACC_X, p->WEIGHTS_POS_Y,
                           p->WEIGHTS_VEL_Y, p->WEIGHTS_ACC_Y, p->WEIGHTS_JERK_X, p->WEIGHTS_JERK_Y};
  for (int c = 0; c < C; ++c) {
    for (int k = 0; k < 8; ++k) I->W[c * 8 + k] = rnd(Wsrc[k][c], dec);
    I->wb[c] = rnd(p->WheelBase[c], dec); I->rad[c] = rnd(p->CollisionRadius[c], dec);
    for (int k = 0; k < 6; ++k) I->x0[c * 6 + k] = rnd(p->IntitialState[c * 6 + k], dec);
    for (int i = 0; i < N; ++i) {
      double* r = I->ref + (c * N + i) * 6;
      r[0] = rnd(p->x_ref[c * N + i], dec); r[1] = rnd(p->vx_ref[c * N + i], dec);
      r[3] = rnd(p->y_ref[c * N + i], dec); r[4] = rnd(p->vy_ref[c * N + i], dec);
    }
    for (int j = 0; j < R; ++j) {
      double* a = I->acc_lim + (c * R + j) * 4; double* jl = I->jerk_lim + (c * R + j) * 4;
      a[0] = rnd(p->min_acc_x[c * R + j], dec); a[1] = rnd(p->max_acc_x[c * R + j], dec);
      a[2] = rnd(p->min_acc_y[c * R + j], dec); a[3] = rnd(p->max_acc_y[c * R + j], dec);
      jl[0] = rnd(p->min_jerk_x[c * R + j], dec); jl[1] = rnd(p->max_jerk_x[c * R + j], dec);
      jl[2] = rnd(p->min_jerk_y[c * R + j], dec); jl[3] = rnd(p->max_jerk_y[c * R + j], dec);
      I->possible[c * R + j] = p->possible_region[c * R + j];
    }
    I->init_region[c] = p->initial_region[c];
  }
  const double* Psrc[6] = {p->POLY_SINT_UB, p->POLY_SINT_LB, p->POLY_COSS_UB, p->POLY_COSS_LB,
                           p->POLY_KAPPA_AX_MAX, p->POLY_KAPPA_AX_MIN};
  for (int j = 0; j < R; ++j) {
    for (int k = 0; k < 4; ++k) I->frac[j * 4 + k] = rnd(p->fraction_parameters[j * 4 + k], dec);
    for (int t = 0; t < 6; ++t)
      for (int k = 0; k < 3; ++k) I->poly[t][j * 3 + k] = rnd(Psrc[t][j * 3 + k], dec);
  }
  for (int e = 0; e < E; ++e) {
    int a = p->env_offsets[e], b = p->env_offsets[e + 1], n = b - a;
    I->env_off[e] = a; I->env_off[e + 1] = b;
    for (int k = 0; k < n; ++k) {
      int k2 = (k + 1) % n;
      double* ed = I->env_edges + 4 * (a + k);
      ed[0] = rnd(p->env_vertices[2 * (a + k)], dec); ed[1] = rnd(p->env_vertices[2 * (a + k) + 1], dec);
      ed[2] = rnd(p->env_vertices[2 * (a + k2)], dec); ed[3] = rnd(p->env_vertices[2 * (a + k2) + 1], dec);
    }
  }
  for (int o = 0; o < O; ++o) {
    I->obs_soft[o] = p->obstacle_is_soft[o];
    for (int i = 0; i < N; ++i)
      for (int k = 0; k < L; ++k) {
        int k2 = (k + 1) % L;
        const double* v = p->obstacle_vertices + ((size_t)(o * N + i) * L) * 2;
        double* ed = I->obs_edges + ((size_t)(o * N + i) * L + k) * 4;
        ed[0] = rnd(v[2 * k], dec); ed[1] = rnd(v[2 * k + 1], dec);
        ed[2] = rnd(v[2 * k2], dec); ed[3] = rnd(v[2 * k2 + 1], dec);
      }
  }
  return I;
}

/* ------------------------------------------------------------------ */
/*  OPL .dat subset reader                                             */
/* ------------------------------------------------------------------ */
typedef struct dval {
  int kind; /* 0 number, 1 list '[', 2 set '{', 3 tuple '<' */
  double num;
  struct dval** items;
  int n, cap;
} dval;

typedef struct { const char* s; size_t pos, len; char* err; int errlen; int failed; } dparser;

static void dfree(dval* v) {
  if (!v) return;
  for (int i = 0; i < v->n; ++i) dfree(v->items[i]);
  free(v->items); free(v);
}

static void skip_ws(dparser* P) {
  for (;;) {
    while (P->pos < P->len && (isspace((unsigned char)P->s[P->pos]) || P->s[P->pos] == ',')) P->pos++;
    if (P->pos + 1 < P->len && P->s[P->pos] == '/' && P->s[P->pos + 1] == '*') {
      P->pos += 2;
      while (P->pos + 1 < P->len && !(P->s[P->pos] == '*' && P->s[P->pos + 1] == '/')) P->pos++;
      P->pos += 2;
    } else if (P->pos + 1 < P->len && P->s[P->pos] == '/' && P->s[P->pos + 1] == '/') {
      while (P->pos < P->len && P->s[P->pos] != '\n') P->pos++;
    } else
      return;
  }
}

static dval* parse_value(dparser* P) {
  skip_ws(P);
  if (P->pos >= P->len) { P->failed = 1; return NULL; }
  char ch = P->s[P->pos];
  dval* v = (dval*)calloc(1, sizeof(dval));
  if (ch == '[' || ch == '{' || ch == '<') {
    char close = ch == '[' ? ']' : (ch == '{' ? '}' : '>');
    v->kind = ch == '[' ? 1 : (ch == '{' ? 2 : 3);
    P->pos++;
    for (;;) {
      skip_ws(P);
      if (P->pos >= P->len) { P->failed = 1; break; }
      if (P->s[P->pos] == close) { P->pos++; break; }
      dval* it = parse_value(P);
      if (!it || P->failed) { dfree(it); P->failed = 1; break; }
      if (v->n == v->cap) { v->cap = v->cap ? 2 * v->cap : 8; v->items = (dval**)realloc(v->items, sizeof(dval*) * v->cap); }
      v->items[v->n++] = it;
    }
    return v;
  }
  char* end = NULL;
  v->kind = 0;
  v->num = strtod(P->s + P->pos, &end);
  if (end == P->s + P->pos) { P->failed = 1; free(v); return NULL; }
  P->pos = (size_t)(end - P->s);
  return v;
}

typedef struct { char name[64]; dval* v; } dentry;

static dval* dfind(dentry* ents, int n, const char* name) {
  for (int i = 0; i < n; ++i)
    if (!strcmp(ents[i].name, name)) return ents[i].v;
  return NULL;
}

static double dnum(dentry* e, int n, const char* name, int* miss) {
  dval* v = dfind(e, n, name);
  if (!v || v->kind != 0) { (*miss)++; return 0; }
  return v->num;
}

/* flatten nested lists of numbers into out (up to max) */
static int dflat(const dval* v, double* out, int max, int pos) {
  if (!v) return pos;
  if (v->kind == 0) { if (pos < max) out[pos] = v->num; return pos + 1; }
  for (int i = 0; i < v->n; ++i) pos = dflat(v->items[i], out, max, pos);
  return pos;
}

static int drow_width(const dval* v) { /* width of the innermost rows of a 2-D list */
  if (!v || v->kind == 0) return 1;
  if (v->n > 0 && v->items[0]->kind != 0) return drow_width(v->items[0]);
  return v->n;
}

static void fill_table(dentry* ents, int ne, const char* name, double* dst, int rows, int width, int* miss) {
  /* tolerate tables narrower than the header says (cplexmodel.dat: 16-wide tables, nr_regions = 32) */
  dval* v = dfind(ents, ne, name);
  if (!v) { (*miss)++; return; }
  int w = drow_width(v);
  int total = dflat(v, NULL, 0, 0);
  double* tmp = dalloc(total);
  dflat(v, tmp, total, 0);
  if (w <= 0) w = 1;
  int r_src = total / w;
  for (int r = 0; r < rows && r < r_src; ++r)
    for (int k = 0; k < width && k < w; ++k) dst[r * width + k] = tmp[r * w + k];
  free(tmp);
}

oinst* orc_from_dat(const char* path, char* err, int errlen) {
  FILE* f = fopen(path, "rb");
  if (!f) { if (err) snprintf(err, errlen, "cannot open %s", path); return NULL; }
  fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
  char* buf = (char*)malloc(sz + 1);
  if (fread(buf, 1, sz, f) != (size_t)sz) { fclose(f); free(buf); return NULL; }
  buf[sz] = 0; fclose(f);
  dparser P = {buf, 0, (size_t)sz, err, errlen, 0};
  dentry* ents = NULL; int ne = 0, cap = 0;
  for (;;) {
    skip_ws(&P);
    if (P.pos >= P.len) break;
    size_t st = P.pos;
    while (P.pos < P.len && (isalnum((unsigned char)buf[P.pos]) || buf[P.pos] == '_')) P.pos++;
    if (P.pos == st) { P.failed = 1; break; }
    if (ne == cap) { cap = cap ? 2 * cap : 96; ents = (dentry*)realloc(ents, sizeof(dentry) * cap); }
    size_t nl = P.pos - st; if (nl > 63) nl = 63;
    memcpy(ents[ne].name, buf + st, nl); ents[ne].name[nl] = 0; ents[ne].v = NULL;
    skip_ws(&P);
    if (P.pos >= P.len || buf[P.pos] != '=') { P.failed = 1; break; }
    P.pos++;
    ents[ne].v = parse_value(&P);
    ne++;
    if (P.failed) break;
    skip_ws(&P);
    if (P.pos >= P.len || buf[P.pos] != ';') { P.failed = 1; break; }
    P.pos++;
  }
  oinst* I = NULL;
  int miss = 0;
  if (!P.failed) {
    int N = (int)dnum(ents, ne, "NumSteps", &miss), C = (int)dnum(ents, ne, "NumCars", &miss);
    int R = (int)dnum(ents, ne, "nr_regions", &miss), E = (int)dnum(ents, ne, "nr_environments", &miss);
    int O = (int)dnum(ents, ne, "nr_obstacles", &miss), L = (int)dnum(ents, ne, "max_lines_obstacles", &miss);
    dval* env = dfind(ents, ne, "MultiEnvironmentConvexPolygon");
    dval* obs = dfind(ents, ne, "ObstacleConvexPolygon");
    int tot = 0;
    if (env) for (int e = 0; e < env->n && e < E; ++e) tot += env->items[e]->n;
    if (!miss && C >= 1 && C <= ORC_MAXC && N >= 2) {
      I = inst_alloc(N, C, R, E, O, L, tot);
      I->ts = dnum(ents, ne, "ts", &miss);
      I->vmin = dnum(ents, ne, "min_vel_x_y", &miss); I->vmax = dnum(ents, ne, "max_vel_x_y", &miss);
      I->amin = dnum(ents, ne, "total_min_acc", &miss); I->amax = dnum(ents, ne, "total_max_acc", &miss);
      I->jmin = dnum(ents, ne, "total_min_jerk", &miss); I->jmax = dnum(ents, ne, "total_max_jerk", &miss);
      I->max_slack = dnum(ents, ne, "maximum_slack", &miss);
      I->w_slack = dnum(ents, ne, "WEIGHTS_SLACK", &miss);
      I->w_slack_obs = dnum(ents, ne, "WEIGHTS_SLACK_OBSTACLE", &miss);
      I->vm = dnum(ents, ne, "minimum_region_change_speed", &miss);
      I->gap = dnum(ents, ne, "relative_mip_gap_tolerance", &miss);
      I->tilim = dnum(ents, ne, "max_solution_time", &miss);
      fill_table(ents, ne, "agent_safety_distance", I->safety, 1, N, &miss);
      fill_table(ents, ne, "agent_safety_distance_slack", I->safety_slack, 1, N, &miss);
      const char* wn[8] = {"WEIGHTS_POS_X", "WEIGHTS_VEL_X", "WEIGHTS_ACC_X", "WEIGHTS_POS_Y",
                           "WEIGHTS_VEL_Y", "WEIGHTS_ACC_Y", "WEIGHTS_JERK_X", "WEIGHTS_JERK_Y"};
      double* tmp = dalloc(C * (N > R ? N : R) + 8);
      for (int k = 0; k < 8; ++k) {
        memset(tmp, 0, sizeof(double) * C);
        fill_table(ents, ne, wn[k], tmp, 1, C, &miss);
        for (int c = 0; c < C; ++c) I->W[c * 8 + k] = tmp[c];
      }
      fill_table(ents, ne, "WheelBase", I->wb, 1, C, &miss);
      fill_table(ents, ne, "CollisionRadius", I->rad, 1, C, &miss);
      fill_table(ents, ne, "IntitialState", I->x0, C, 6, &miss);
      const char* rn[4] = {"x_ref", "vx_ref", "y_ref", "vy_ref"};
      const int ri[4] = {0, 1, 3, 4};
      for (int k = 0; k < 4; ++k) {
        memset(tmp, 0, sizeof(double) * C * N);
        fill_table(ents, ne, rn[k], tmp, C, N, &miss);
        for (int c = 0; c < C; ++c)
          for (int i = 0; i < N; ++i) I->ref[(c * N + i) * 6 + ri[k]] = tmp[c * N + i];
      }
      const char* an[4] = {"min_acc_x", "max_acc_x", "min_acc_y", "max_acc_y"};
      const char* jn[4] = {"min_jerk_x", "max_jerk_x", "min_jerk_y", "max_jerk_y"};
      for (int k = 0; k < 4; ++k) {
        memset(tmp, 0, sizeof(double) * C * R);
        fill_table(ents, ne, an[k], tmp, C, R, &miss);
        for (int c = 0; c < C; ++c) for (int j = 0; j < R; ++j) I->acc_lim[(c * R + j) * 4 + k] = tmp[c * R + j];
        memset(tmp, 0, sizeof(double) * C * R);
        fill_table(ents, ne, jn[k], tmp, C, R, &miss);
        for (int c = 0; c < C; ++c) for (int j = 0; j < R; ++j) I->jerk_lim[(c * R + j) * 4 + k] = tmp[c * R + j];
      }
      memset(tmp, 0, sizeof(double) * C);
      fill_table(ents, ne, "initial_region", tmp, 1, C, &miss);
      for (int c = 0; c < C; ++c) I->init_region[c] = (int)tmp[c];
      memset(tmp, 0, sizeof(double) * C * R);
      fill_table(ents, ne, "possible_region", tmp, C, R, &miss);
      for (int k = 0; k < C * R; ++k) I->possible[k] = (int)tmp[k];
      free(tmp);
      fill_table(ents, ne, "fraction_parameters", I->frac, R, 4, &miss);
      const char* pn[6] = {"POLY_SINT_UB", "POLY_SINT_LB", "POLY_COSS_UB", "POLY_COSS_LB", "POLY_KAPPA_AX_MAX",
                           "POLY_KAPPA_AX_MIN"};
      for (int t = 0; t < 6; ++t) fill_table(ents, ne, pn[t], I->poly[t], R, 3, &miss);
      int off = 0;
      for (int e = 0; e < E; ++e) {
        I->env_off[e] = off;
        dval* set = env->items[e];
        for (int k = 0; k < set->n; ++k) {
          dval* t = set->items[k];
          for (int q = 0; q < 4 && q + 1 < t->n; ++q) I->env_edges[4 * (off + k) + q] = t->items[q + 1]->num;
        }
        off += set->n;
      }
      I->env_off[E] = off;
      for (int o = 0; o < O; ++o) {
        dval* per_t = obs->items[o];
        for (int i = 0; i < N && i < per_t->n; ++i) {
          dval* set = per_t->items[i];
          for (int k = 0; k < L && k < set->n; ++k) {
            dval* t = set->items[k];
            for (int q = 0; q < 4 && q + 1 < t->n; ++q)
              I->obs_edges[((size_t)(o * N + i) * L + k) * 4 + q] = t->items[q + 1]->num;
          }
        }
      }
      if (O > 0) {
        double* ts_ = dalloc(O);
        fill_table(ents, ne, "obstacle_is_soft", ts_, 1, O, &miss);
        for (int o = 0; o < O; ++o) I->obs_soft[o] = (int)ts_[o];
        free(ts_);
      }
    }
  }
  if ((P.failed || miss || !I) && err) snprintf(err, errlen, "parse error in %s (pos %zu, %d missing)", path, P.pos, miss);
  if (miss && I) { orc_free(I); I = NULL; }
  for (int i = 0; i < ne; ++i) dfree(ents[i].v);
  free(ents); free(buf);
  return I;
}

/* ------------------------------------------------------------------ */
static int* ifill(int n) {
  int* p = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  for (int i = 0; i < n; ++i) p[i] = 9999999; /* src/cplex_wrapper.cpp:259 */
  return p;
}
static double* dfill(int n) {
  double* p = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
  for (int i = 0; i < n; ++i) p[i] = 9999999.0;
  return p;
}

miqp_raw_results_c* orc_results_alloc(int C, int N, int R, int E, int O, int L) {
  miqp_raw_results_c* r = (miqp_raw_results_c*)calloc(1, sizeof(*r));
  int K = C - 1;
  r->N = N; r->NrEnvironments = E; r->NrRegions = R; r->NrObstacles = O; r->MaxLinesObstacles = L;
  r->NrCarToCarCollisions = K; r->NrCars = C;
  double** dd[12] = {&r->u_x, &r->u_y, &r->pos_x, &r->vel_x, &r->acc_x, &r->pos_y, &r->vel_y, &r->acc_y,
                     &r->pos_x_front_UB, &r->pos_x_front_LB, &r->pos_y_front_UB, &r->pos_y_front_LB};
  for (int k = 0; k < 12; ++k) *dd[k] = dfill(C * N);
  r->notWithinEnvironmentRear = ifill(C * E * N); r->notWithinEnvironmentFrontUbUb = ifill(C * E * N);
  r->notWithinEnvironmentFrontLbUb = ifill(C * E * N); r->notWithinEnvironmentFrontUbLb = ifill(C * E * N);
  r->notWithinEnvironmentFrontLbLb = ifill(C * E * N);
  r->active_region = ifill(C * N * R);
  r->region_change_not_allowed_x_positive = ifill(C * N); r->region_change_not_allowed_y_positive = ifill(C * N);
  r->region_change_not_allowed_x_negative = ifill(C * N); r->region_change_not_allowed_y_negative = ifill(C * N);
  r->region_change_not_allowed_combined = ifill(C * N);
  r->deltacc = ifill(C * O * N * L); r->deltacc_front = ifill(C * O * N * L * 4);
  r->car2car_collision = ifill(K * K * N * 16); r->slackvars = ifill(K * K * N * 4);
  r->slackvarsObstacle = ifill(C * O * N); r->slackvarsObstacle_front = ifill(C * O * N * 4);
  r->slackvars_real = dfill(K * K * N * 4);
  return r;
}

void orc_results_free(miqp_raw_results_c* r) {
  if (!r) return;
  free(r->u_x); free(r->u_y); free(r->pos_x); free(r->vel_x); free(r->acc_x); free(r->pos_y); free(r->vel_y);
  free(r->acc_y); free(r->pos_x_front_UB); free(r->pos_x_front_LB); free(r->pos_y_front_UB); free(r->pos_y_front_LB);
  free(r->notWithinEnvironmentRear); free(r->notWithinEnvironmentFrontUbUb); free(r->notWithinEnvironmentFrontLbUb);
  free(r->notWithinEnvironmentFrontUbLb); free(r->notWithinEnvironmentFrontLbLb); free(r->active_region);
  free(r->region_change_not_allowed_x_positive); free(r->region_change_not_allowed_y_positive);
  free(r->region_change_not_allowed_x_negative); free(r->region_change_not_allowed_y_negative);
  free(r->region_change_not_allowed_combined); free(r->deltacc); free(r->deltacc_front); free(r->car2car_collision);
  free(r->slackvars); free(r->slackvarsObstacle); free(r->slackvarsObstacle_front); free(r->slackvars_real);
  free(r);
}
