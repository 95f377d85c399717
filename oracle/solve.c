/* solve.c - disjunctive form of the model, stage-banded interior point QP and branch & bound
 * (CPU oracle, TEST INFRASTRUCTURE).
 *
 * Disjunctive reading of the cplexmodel .mod files:
 *   region   (c,i>=2): exactly one active_region[c,i,j] (model_region_constraints.mod:43-114) combined with
 *                      the low-speed freeze of minimum_speed_constraints.mod:9-49:
 *                      alternative (j, h)   = sector j, outside the slow square through half-plane h,
 *                      alternative (j,slow) = |vx|,|vy| <= v_m, region frozen to that of step i-1.
 *   env      (c,i,point): point lies in >= 1 convex piece (obstacle_environment_constraints.mod:6-33)
 *   obstacle (c,o,i,point): >= 1 edge separates the point (:52-96); a soft obstacle may be ignored at cost
 *   c2c      (pair,i,group): >= 1 of 4 separations (agent_collision_constraints.mod:35-71)
 * A node fixes a subset of the disjunctions; its relaxation keeps only the rows of the fixed
 * alternatives (dropping the rows of undecided binaries is a valid relaxation of the big-M rows).
 */
#include "oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define RHO_EL 1.0e5
#define FEAS_TOL 1.0e-6
#define QP_TOL 1.0e-8
#define QP_TOL_FINAL 1.0e-13 /* polish of the returned incumbent */
#define QP_MAXIT 80
#define QP_SIGMA 0.1
#define QP_T0 1.0e-3
#define QP_SIGMA_LO 0.1
#define QP_SIGMA_HI 0.1

enum { SX = 0, SVX, SAX, SY, SVY, SAY };
enum { PT_R = 0, PT_U = 1, PT_L = 2 };
/* env corner order (mod:15-28): Rear, UbUb, LbUb, UbLb, LbLb given as (x type, y type) */
static const int ENV_PT[5][2] = {{PT_R, PT_R}, {PT_U, PT_U}, {PT_L, PT_U}, {PT_U, PT_L}, {PT_L, PT_L}};
/* obstacle corner order (mod:64-68): rear, (LB,LB), (UB,LB), (LB,UB), (UB,UB) */
static const int OBS_PT[5][2] = {{PT_R, PT_R}, {PT_L, PT_L}, {PT_U, PT_L}, {PT_L, PT_U}, {PT_U, PT_U}};

typedef struct { int stage; double a; double rhs; double co[ORC_NZMAX]; } orow;

typedef struct {
  orow* r; int n, cap;
} rowvec;

static orow* rv_push(rowvec* v) {
  if (v->n == v->cap) { v->cap = v->cap ? 2 * v->cap : 256; v->r = (orow*)realloc(v->r, sizeof(orow) * v->cap); }
  orow* r = &v->r[v->n++];
  memset(r, 0, sizeof(*r));
  return r;
}

/* ------------------------------------------------------------------ model context */
typedef struct {
  const oinst* I;
  int nx, nu, nz;
  int nposs[ORC_MAXC];
  int* plist[ORC_MAXC];     /* possible regions (0-based) per car */
  int* nhs;                 /* [R] number of non-slow half-planes of sector j */
  int (*hs)[2][2];          /* [R][h][axis,sign] (at most 2 kept) */
  /* fix record layout */
  int off_reg, off_env, off_obs, off_c2c, fixlen;
} dmodel;

#define FIX_REG(M, f, c, i) ((f)[(M)->off_reg + (c) * (M)->I->N + (i)])
#define FIX_ENV(M, f, c, i, p) ((f)[(M)->off_env + ((c) * (M)->I->N + (i)) * 5 + (p)])
#define FIX_OBS(M, f, c, o, i, p) ((f)[(M)->off_obs + (((c) * (M)->I->O + (o)) * (M)->I->N + (i)) * 5 + (p)])
#define FIX_C2C(M, f, pr, i, g) ((f)[(M)->off_c2c + ((pr) * (M)->I->N + (i)) * 4 + (g)])
/* region alternative code: pidx*4 + h, h in 0..2 half-plane index, 3 = slow */
#define ALT_J(M, c, code) ((M)->plist[c][(code) >> 2])
#define ALT_H(code) ((code) & 3)

static void nonslow_halfplanes(dmodel* M, int j) {
  const double* F = M->I->frac + j * 4;
  double t1 = atan2(F[1], F[0]), t2 = atan2(F[3], F[2]);
  if (t2 < t1) t2 += 2 * M_PI;
  const int cand[4][2] = {{0, 1}, {0, -1}, {1, 1}, {1, -1}};
  double vals[4][65]; int ok[4];
  for (int q = 0; q < 4; ++q) {
    ok[q] = 0;
    for (int s = 0; s < 65; ++s) {
      double th = t1 + (t2 - t1) * s / 64.0;
      double v = cand[q][0] == 0 ? cand[q][1] * cos(th) : cand[q][1] * sin(th);
      vals[q][s] = v;
      if (v > 1e-12) ok[q] = 1;
    }
  }
  int n = 0;
  for (int a = 0; a < 4; ++a) {
    if (!ok[a]) continue;
    int dominated = 0;
    for (int b = 0; b < 4 && !dominated; ++b) {
      if (a == b || !ok[b]) continue;
      int all_ge = 1, any_gt = 0;
      for (int s = 0; s < 65; ++s) {
        if (vals[a][s] <= 1e-12) continue;
        if (vals[b][s] < vals[a][s] - 1e-12) all_ge = 0;
        if (vals[b][s] > vals[a][s] + 1e-12) any_gt = 1;
      }
      if (all_ge && (any_gt || b < a)) dominated = 1;
    }
    if (!dominated && n < 2) { M->hs[j][n][0] = cand[a][0]; M->hs[j][n][1] = cand[a][1]; n++; }
  }
  M->nhs[j] = n;
}

static dmodel* dm_new(const oinst* I) {
  dmodel* M = (dmodel*)calloc(1, sizeof(dmodel));
  M->I = I; M->nx = 6 * I->C; M->nu = 2 * I->C; M->nz = 8 * I->C;
  for (int c = 0; c < I->C; ++c) {
    M->plist[c] = (int*)malloc(sizeof(int) * (I->R + 1));
    M->nposs[c] = 0;
    for (int j = 0; j < I->R; ++j)
      if (I->possible[c * I->R + j] == 1) M->plist[c][M->nposs[c]++] = j;
  }
  M->nhs = (int*)calloc(I->R + 1, sizeof(int));
  M->hs = (int(*)[2][2])calloc(I->R + 1, sizeof(int[2][2]));
  for (int j = 0; j < I->R; ++j) nonslow_halfplanes(M, j);
  M->off_reg = 0;
  M->off_env = M->off_reg + I->C * I->N;
  M->off_obs = M->off_env + I->C * I->N * 5;
  M->off_c2c = M->off_obs + I->C * I->O * I->N * 5;
  M->fixlen = M->off_c2c + I->NP * I->N * 4;
  return M;
}

static void dm_free(dmodel* M) {
  for (int c = 0; c < M->I->C; ++c) free(M->plist[c]);
  free(M->nhs); free(M->hs); free(M);
}

static void pair_cars(const oinst* I, int p, int* c1, int* c2) {
  int k = 0;
  for (int a = 0; a < I->C; ++a)
    for (int b = a + 1; b < I->C; ++b) { if (k == p) { *c1 = a; *c2 = b; return; } k++; }
  *c1 = 0; *c2 = 1;
}

/* ------------------------------------------------------------------ row builders */
/* affine point coordinate: value = co . z_i + k ; which = 0 x / 1 y */
static void point_affine(const dmodel* M, int c, int i, int j, int type, int which, double* co, double* k) {
  const oinst* I = M->I;
  memset(co, 0, sizeof(double) * M->nz);
  if (i == 0) {
    double th = atan2(I->x0[c * 6 + SVY], I->x0[c * 6 + SVX]);
    double base = I->x0[c * 6 + (which ? SY : SX)];
    *k = base + (type != PT_R ? (which ? sin(th) : cos(th)) * I->wb[c] : 0.0);
    return;
  }
  co[6 * c + (which ? SY : SX)] = 1.0; *k = 0.0;
  if (type != PT_R) {
    int t = which ? (type == PT_U ? OP_SINT_UB : OP_SINT_LB) : (type == PT_U ? OP_COSS_UB : OP_COSS_LB);
    const double* p = I->poly[t] + j * 3;
    *k = I->wb[c] * p[0]; co[6 * c + SVX] += I->wb[c] * p[1]; co[6 * c + SVY] += I->wb[c] * p[2];
  }
}

/* alpha*X + beta*Y (of car c) added into row */
static void add_point(const dmodel* M, orow* r, int c, int i, int j, int tx, int ty, double al, double be) {
  double co[ORC_NZMAX], k;
  if (al != 0.0) {
    point_affine(M, c, i, j, tx, 0, co, &k);
    for (int q = 0; q < M->nz; ++q) r->co[q] += al * co[q];
    r->rhs -= al * k;
  }
  if (be != 0.0) {
    point_affine(M, c, i, j, ty, 1, co, &k);
    for (int q = 0; q < M->nz; ++q) r->co[q] += be * co[q];
    r->rhs -= be * k;
  }
}

static void row_scale(const dmodel* M, orow* r, double nrm) {
  if (nrm <= 0) return;
  for (int q = 0; q < M->nz; ++q) r->co[q] /= nrm;
  r->rhs /= nrm;
  if (r->a > 0) r->a *= nrm * nrm;
}

static void bound_row(rowvec* v, int stage, int idx, double sign, double rhs) {
  orow* r = rv_push(v); r->stage = stage; r->co[idx] = sign; r->rhs = rhs;
}

static void global_rows(const dmodel* M, rowvec* v, int c, int i) {
  const oinst* I = M->I; int o = 6 * c, u = 6 * I->C + 2 * c;
  if (i >= 1) { /* A3, model_region_constraints.mod:22-39 */
    bound_row(v, i, o + SVX, -1, -I->vmin); bound_row(v, i, o + SVY, -1, -I->vmin); bound_row(v, i, o + SVX, 1, I->vmax);
    bound_row(v, i, o + SAX, 1, I->amax); bound_row(v, i, o + SAX, -1, -I->amin);
    bound_row(v, i, o + SAY, 1, I->amax); bound_row(v, i, o + SAY, -1, -I->amin);
  }
  if (i <= I->N - 2) {
    for (int s = 0; s < 2; ++s) {
      double hi = I->jmax, lo = I->jmin;
      if (i == 0) { /* A1 jerk box of the initial region + residual rows of all others, initial_conditions.mod:30-48 */
        int j0 = I->init_region[c] - 1;
        for (int j = 0; j < I->R; ++j) {
          double Mj = j == j0 ? 0.0 : 10.0;
          const double* jl = I->jerk_lim + (c * I->R + j) * 4;
          if (jl[2 * s + 1] + Mj < hi) hi = jl[2 * s + 1] + Mj;
          if (jl[2 * s] - Mj > lo) lo = jl[2 * s] - Mj;
        }
      }
      bound_row(v, i, u + s, 1, hi); bound_row(v, i, u + s, -1, -lo);
    }
  }
}

static void region_rows(const dmodel* M, rowvec* v, int c, int i, int code) {
  const oinst* I = M->I; int j = ALT_J(M, c, code), h = ALT_H(code);
  int o = 6 * c, u = 6 * I->C + 2 * c;
  const double* F = I->frac + j * 4;
  if (h != 3) {
    double n1 = hypot(F[0], F[1]), n3 = hypot(F[2], F[3]);
    orow* r = rv_push(v); r->stage = i; r->co[o + SVY] = -F[0] / n1; r->co[o + SVX] = F[1] / n1; r->rhs = 0;
    r = rv_push(v); r->stage = i; r->co[o + SVY] = F[2] / n3; r->co[o + SVX] = -F[3] / n3; r->rhs = 0;
    int ax = M->hs[j][h][0], sg = M->hs[j][h][1];
    bound_row(v, i, o + (ax == 0 ? SVX : SVY), -sg, -I->vm);
    double rho = (F[1] + F[3]) / (F[0] + F[2]);
    const double* kx = I->poly[OP_KMAX] + j * 3; const double* kn = I->poly[OP_KMIN] + j * 3;
    r = rv_push(v); r->stage = i; r->co[o + SAY] = 1; r->co[o + SAX] = -rho; r->co[o + SVX] = -kx[1]; r->co[o + SVY] = -kx[2];
    r->rhs = kx[0];
    r = rv_push(v); r->stage = i; r->co[o + SAY] = -1; r->co[o + SAX] = rho; r->co[o + SVX] = kn[1]; r->co[o + SVY] = kn[2];
    r->rhs = -kn[0];
  } else {
    bound_row(v, i, o + SVX, 1, I->vm); bound_row(v, i, o + SVX, -1, I->vm);
    bound_row(v, i, o + SVY, 1, I->vm); bound_row(v, i, o + SVY, -1, I->vm);
  }
  for (int s = 0; s < 2; ++s) { /* acc box incl. residual big-M rows of the other possible regions */
    double hi = 1e300, lo = -1e300;
    for (int q = 0; q < M->nposs[c]; ++q) {
      int jj = M->plist[c][q]; double Mj = jj == j ? 0.0 : 10.0;
      const double* al = I->acc_lim + (c * I->R + jj) * 4;
      if (al[2 * s + 1] + Mj < hi) hi = al[2 * s + 1] + Mj;
      if (al[2 * s] - Mj > lo) lo = al[2 * s] - Mj;
    }
    bound_row(v, i, o + (s ? SAY : SAX), 1, hi); bound_row(v, i, o + (s ? SAY : SAX), -1, -lo);
  }
  if (i <= I->N - 2)
    for (int s = 0; s < 2; ++s) {
      double hi = 1e300, lo = -1e300;
      for (int q = 0; q < M->nposs[c]; ++q) {
        int jj = M->plist[c][q]; double Mj = jj == j ? 0.0 : 10.0;
        const double* jl = I->jerk_lim + (c * I->R + jj) * 4;
        if (jl[2 * s + 1] + Mj < hi) hi = jl[2 * s + 1] + Mj;
        if (jl[2 * s] - Mj > lo) lo = jl[2 * s] - Mj;
      }
      bound_row(v, i, u + s, 1, hi); bound_row(v, i, u + s, -1, -lo);
    }
}

/* edge row: inside (sense=+1: cross >= 0, environment) or separated (sense=-1: cross <= 0, obstacle) */
static void edge_row(const dmodel* M, rowvec* v, const double* e, int sense, int c, int i, int j, int tx, int ty) {
  double dx = e[2] - e[0], dy = e[3] - e[1];
  double al = sense > 0 ? dy : -dy, be = sense > 0 ? -dx : dx;
  orow* r = rv_push(v); r->stage = i; r->rhs = al * e[0] + be * e[1];
  add_point(M, r, c, i, j, tx, ty, al, be);
  row_scale(M, r, hypot(al, be));
}

static void env_rows(const dmodel* M, rowvec* v, int c, int i, int pt, int e, int j) {
  const oinst* I = M->I;
  for (int k = I->env_off[e]; k < I->env_off[e + 1]; ++k)
    edge_row(M, v, I->env_edges + 4 * k, 1, c, i, j, ENV_PT[pt][0], ENV_PT[pt][1]);
}

static void obs_row(const dmodel* M, rowvec* v, int c, int o, int i, int pt, int k, int j) {
  const oinst* I = M->I;
  edge_row(M, v, I->obs_edges + ((size_t)(o * I->N + i) * I->L + k) * 4, -1, c, i, j, OBS_PT[pt][0], OBS_PT[pt][1]);
}

/* c2c rows of (pair, step, group, alt).  Appends 1 row (hard) or the limit row followed by the zero-slack
 * quadratic-soft row; returns the number of rows appended (the LAST one is the zero-slack form). */
static int c2c_rows(const dmodel* M, rowvec* v, int p, int i, int grp, int alt, int j1, int j2) {
  const oinst* I = M->I; int c1, c2; pair_cars(I, p, &c1, &c2);
  double D = I->rad[c1] + I->rad[c2] + I->safety[i], S = I->safety_slack[i];
  int isx = alt < 2, lo = (alt == 0 || alt == 2);
  /* A - B <= -sep, A/B = (car, region, xtype, ytype) */
  int ca, ja, ta, cb, jb, tb, soft;
  if (grp == 0) { soft = 1; ta = tb = PT_R; if (lo) { ca = c1; ja = j1; cb = c2; jb = j2; } else { ca = c2; ja = j2; cb = c1; jb = j1; } }
  else if (grp == 1) { soft = 0; if (lo) { ca = c1; ja = j1; ta = PT_R; cb = c2; jb = j2; tb = PT_L; } else { ca = c2; ja = j2; ta = PT_U; cb = c1; jb = j1; tb = PT_R; } }
  else if (grp == 2) { soft = 0; if (lo) { ca = c2; ja = j2; ta = PT_R; cb = c1; jb = j1; tb = PT_L; } else { ca = c1; ja = j1; ta = PT_U; cb = c2; jb = j2; tb = PT_R; } }
  else { soft = 1; if (lo) { ca = c2; ja = j2; ta = PT_U; cb = c1; jb = j1; tb = PT_L; } else { ca = c1; ja = j1; ta = PT_U; cb = c2; jb = j2; tb = PT_L; } }
  double al = isx ? 1.0 : 0.0, be = isx ? 0.0 : 1.0;
  int n = 0;
  if (!soft) {
    orow* r = rv_push(v); r->stage = i; r->rhs = -D;
    add_point(M, r, ca, i, ja, ta, ta, al, be); add_point(M, r, cb, i, jb, tb, tb, -al, -be);
    return 1;
  }
  double smax = S < I->max_slack ? S : I->max_slack;
  if (smax < 0) smax = 0;
  orow* r = rv_push(v); r->stage = i; r->rhs = -(D + S) + smax; n++;
  add_point(M, r, ca, i, ja, ta, ta, al, be); add_point(M, r, cb, i, jb, tb, tb, -al, -be);
  if (smax > 0 && I->w_slack > 0) {
    r = rv_push(v); r->stage = i; r->rhs = -(D + S); r->a = 2.0 * I->w_slack; n++;
    add_point(M, r, ca, i, ja, ta, ta, al, be); add_point(M, r, cb, i, jb, tb, tb, -al, -be);
  } else if (smax <= 0) {
    /* single hard row already has rhs -(D+S) */
  }
  return n;
}

static double row_val(const dmodel* M, const orow* r, const double* Z) {
  const double* z = Z + (size_t)r->stage * M->nz;
  double s = -r->rhs;
  for (int q = 0; q < M->nz; ++q) s += r->co[q] * z[q];
  return s;
}

/* ------------------------------------------------------------------ QP: stage-banded primal-dual IPM */
typedef struct { double* Z; double obj; double viol; int it; int ok; double slack_cost; double* lam; double dgap; } qpres;

static void build_AB(const oinst* I, double* A, double* B, int nx, int nu) {
  double ts = I->ts;
  memset(A, 0, sizeof(double) * nx * nx); memset(B, 0, sizeof(double) * nx * nu);
  for (int c = 0; c < I->C; ++c)
    for (int ax = 0; ax < 2; ++ax) {
      int o = 6 * c + 3 * ax;
      A[(o + 0) * nx + o + 0] = 1; A[(o + 0) * nx + o + 1] = ts; A[(o + 0) * nx + o + 2] = ts * ts / 2;
      A[(o + 1) * nx + o + 1] = 1; A[(o + 1) * nx + o + 2] = ts; A[(o + 2) * nx + o + 2] = 1;
      B[(o + 0) * nu + 2 * c + ax] = ts * ts * ts / 6; B[(o + 1) * nu + 2 * c + ax] = ts * ts / 2; B[(o + 2) * nu + 2 * c + ax] = ts;
    }
}

/* rows must be usable in any order; they are bucketed by stage here */
static int qp_solve_tol(const dmodel* M, const orow* rows, int m, qpres* out, double qp_tol) {
  const oinst* I = M->I;
  const int N = I->N, nx = M->nx, nu = M->nu, nz = M->nz;
  double* A = (double*)malloc(sizeof(double) * nx * nx); double* B = (double*)malloc(sizeof(double) * nx * nu);
  build_AB(I, A, B, nx, nu);
  double* Wd = (double*)calloc(nz, sizeof(double));
  for (int c = 0; c < I->C; ++c) {
    for (int k = 0; k < 6; ++k) Wd[6 * c + k] = I->W[c * 8 + k];
    Wd[6 * I->C + 2 * c] = I->W[c * 8 + 6]; Wd[6 * I->C + 2 * c + 1] = I->W[c * 8 + 7];
  }
  double* Z = (double*)calloc((size_t)N * nz, sizeof(double));
  double* Rf = (double*)calloc((size_t)N * nz, sizeof(double));
  for (int i = 0; i < N; ++i)
    for (int c = 0; c < I->C; ++c)
      for (int k = 0; k < 6; ++k) Rf[i * nz + 6 * c + k] = I->ref[(c * N + i) * 6 + k];
  for (int c = 0; c < I->C; ++c)
    for (int k = 0; k < 6; ++k) Z[6 * c + k] = I->x0[c * 6 + k];
  for (int i = 0; i + 1 < N; ++i)
    for (int r = 0; r < nx; ++r) {
      double s = 0;
      for (int q = 0; q < nx; ++q) s += A[r * nx + q] * Z[i * nz + q];
      Z[(i + 1) * nz + r] = s;
    }
  /* bucket rows by stage */
  int* start = (int*)calloc(N + 2, sizeof(int)); int* order = (int*)malloc(sizeof(int) * (m + 1));
  for (int k = 0; k < m; ++k) start[rows[k].stage + 1]++;
  for (int i = 0; i < N; ++i) start[i + 1] += start[i];
  { int* pos = (int*)malloc(sizeof(int) * (N + 1)); memcpy(pos, start, sizeof(int) * (N + 1));
    for (int k = 0; k < m; ++k) order[pos[rows[k].stage]++] = k;
    free(pos); }
  double* s = (double*)malloc(sizeof(double) * (m + 1)); double* lam = (double*)malloc(sizeof(double) * (m + 1));
  double* ds = (double*)malloc(sizeof(double) * (m + 1)); double* dlam = (double*)malloc(sizeof(double) * (m + 1));
  double* w = (double*)malloc(sizeof(double) * (m + 1)); double* kap = (double*)malloc(sizeof(double) * (m + 1));
  double* gd = (double*)malloc(sizeof(double) * (m + 1));
  double* tt = (double*)calloc(m + 1, sizeof(double)); double* dtt = (double*)calloc(m + 1, sizeof(double));
  int mel = 0;
  for (int k = 0; k < m; ++k) {
    double c = -row_val(M, &rows[k], Z);
    if (rows[k].a == 0.0) {
      double t0 = getenv("ORC_T0") ? atof(getenv("ORC_T0")) : QP_T0;
      lam[k] = 1.0; mel++;
      /* elastic slack starts small so that t*mu (mu ~ rho) is of the order of s*lambda */
      if (c > t0) { tt[k] = t0; s[k] = c + t0; } else { s[k] = 100 * t0; tt[k] = s[k] - c; }
    }
    else { double a = rows[k].a; lam[k] = fmax(1.0, -2 * c * a + 1.0); s[k] = c + lam[k] / a; }
  }
  double* Kg = (double*)calloc((size_t)N * nu * nx, sizeof(double)); double* kg = (double*)calloc((size_t)N * nu, sizeof(double));
  double* dZ = (double*)calloc((size_t)N * nz, sizeof(double));
  double* Phi = (double*)malloc(sizeof(double) * nz * nz); double* rr = (double*)malloc(sizeof(double) * nz);
  double* P = (double*)malloc(sizeof(double) * nx * nx); double* pv = (double*)malloc(sizeof(double) * nx);
  double* T = (double*)malloc(sizeof(double) * nx * nz); double* S = (double*)malloc(sizeof(double) * nz * nz);
  double* sv = (double*)malloc(sizeof(double) * nz); double* Lc = (double*)malloc(sizeof(double) * nu * nu);
  double* Pn = (double*)malloc(sizeof(double) * nx * nx); double* pn = (double*)malloc(sizeof(double) * nx);
  int it = 0, ok = 0; double resid_fac = 1.0, R0 = 0.0, alpha_prev = 0.0, dgap = 0.0;
  for (it = 1; it <= QP_MAXIT; ++it) {
    double comp = 0, obj = 0;
    for (int i = 0; i < N; ++i)
      for (int q = 0; q < nz; ++q) { double d = Z[i * nz + q] - Rf[i * nz + q]; obj += Wd[q] * d * d; }
    for (int k = 0; k < m; ++k) {
      double c = -row_val(M, &rows[k], Z);
      comp += s[k] * lam[k];
      if (rows[k].a == 0.0) { double t = tt[k], mu = RHO_EL - lam[k]; comp += t * mu; (void)c; }
    }
    dgap = comp; /* total complementarity = primal value - dual value of the iterate */
    comp /= (m + mel > 0 ? m + mel : 1);
    if (comp < qp_tol * fmax(1.0, fabs(obj)) && resid_fac * R0 < 1e-7) { ok = 1; break; }
    /* centering: aggressive after a (nearly) full step, conservative after a blocked one */
    double sigma = getenv("ORC_SIGMA_LO") ? (alpha_prev >= 0.9 ? atof(getenv("ORC_SIGMA_LO")) : atof(getenv("ORC_SIGMA_HI"))) : (alpha_prev >= 0.9 ? QP_SIGMA_LO : QP_SIGMA_HI);
    double tau = sigma * comp;
    /* backward sweep */
    memset(P, 0, sizeof(double) * nx * nx); memset(pv, 0, sizeof(double) * nx);
    double rmax = 0;
    for (int i = N - 1; i >= 0; --i) {
      memset(Phi, 0, sizeof(double) * nz * nz);
      for (int q = 0; q < nz; ++q) { Phi[q * nz + q] = 2 * Wd[q]; rr[q] = 2 * Wd[q] * (Z[i * nz + q] - Rf[i * nz + q]); }
      for (int kk = start[i]; kk < start[i + 1]; ++kk) {
        int k = order[kk]; const orow* r = &rows[k];
        double c = -row_val(M, r, Z), zz, r2mu = 0;
        if (r->a == 0.0) { double t = tt[k], mu = RHO_EL - lam[k]; zz = t / mu; r2mu = (tau - t * mu) / mu; (void)c; }
        else zz = 1.0 / r->a;
        double D = s[k] / lam[k] + zz;
        w[k] = 1.0 / D;
        kap[k] = ((tau - s[k] * lam[k]) / lam[k] - r2mu) / D;
        double f = lam[k] + kap[k];
        for (int a_ = 0; a_ < nz; ++a_) {
          double ca = r->co[a_];
          if (ca == 0.0) continue;
          rr[a_] += f * ca;
          for (int b_ = 0; b_ < nz; ++b_) Phi[a_ * nz + b_] += w[k] * ca * r->co[b_];
        }
      }
      if (i == N - 1) { /* u_{N-1} = 0 fixed (initial_conditions.mod:25-26) */
        for (int a_ = 0; a_ < nx; ++a_) { pv[a_] = rr[a_]; for (int b_ = 0; b_ < nx; ++b_) P[a_ * nx + b_] = Phi[a_ * nz + b_]; }
        if (it == 1) for (int a_ = 0; a_ < nx; ++a_) rmax = fmax(rmax, fabs(rr[a_]));
        continue;
      }
      /* T = P [A B] (nx x nz) ; S = [A B]' T + Phi ; sv = rr + [A B]' p */
      for (int a_ = 0; a_ < nx; ++a_)
        for (int b_ = 0; b_ < nz; ++b_) {
          double acc = 0;
          for (int q = 0; q < nx; ++q) acc += P[a_ * nx + q] * (b_ < nx ? A[q * nx + b_] : B[q * nu + (b_ - nx)]);
          T[a_ * nz + b_] = acc;
        }
      for (int a_ = 0; a_ < nz; ++a_) {
        double accv = rr[a_];
        for (int q = 0; q < nx; ++q) accv += (a_ < nx ? A[q * nx + a_] : B[q * nu + (a_ - nx)]) * pv[q];
        sv[a_] = accv;
        for (int b_ = 0; b_ < nz; ++b_) {
          double acc = Phi[a_ * nz + b_];
          for (int q = 0; q < nx; ++q) acc += (a_ < nx ? A[q * nx + a_] : B[q * nu + (a_ - nx)]) * T[q * nz + b_];
          S[a_ * nz + b_] = acc;
        }
      }
      if (it == 1) for (int a_ = 0; a_ < nz; ++a_) rmax = fmax(rmax, fabs(rr[a_]));
      /* Cholesky of Suu */
      for (int a_ = 0; a_ < nu; ++a_)
        for (int b_ = 0; b_ <= a_; ++b_) {
          double acc = S[(nx + a_) * nz + nx + b_];
          for (int q = 0; q < b_; ++q) acc -= Lc[a_ * nu + q] * Lc[b_ * nu + q];
          if (a_ == b_) Lc[a_ * nu + a_] = sqrt(acc > 1e-300 ? acc : 1e-300);
          else Lc[a_ * nu + b_] = acc / Lc[b_ * nu + b_];
        }
      /* K = Suu^-1 Sux (nu x nx), k = Suu^-1 su */
      double* Ki = Kg + (size_t)i * nu * nx; double* ki = kg + (size_t)i * nu;
      for (int col = 0; col <= nx; ++col) {
        double y[2 * ORC_MAXC];
        for (int a_ = 0; a_ < nu; ++a_) {
          double acc = col < nx ? S[(nx + a_) * nz + col] : sv[nx + a_];
          for (int q = 0; q < a_; ++q) acc -= Lc[a_ * nu + q] * y[q];
          y[a_] = acc / Lc[a_ * nu + a_];
        }
        for (int a_ = nu - 1; a_ >= 0; --a_) {
          double acc = y[a_];
          for (int q = a_ + 1; q < nu; ++q) acc -= Lc[q * nu + a_] * y[q];
          y[a_] = acc / Lc[a_ * nu + a_];
        }
        for (int a_ = 0; a_ < nu; ++a_) { if (col < nx) Ki[a_ * nx + col] = y[a_]; else ki[a_] = y[a_]; }
      }
      for (int a_ = 0; a_ < nx; ++a_) {
        double accv = sv[a_];
        for (int q = 0; q < nu; ++q) accv -= S[a_ * nz + nx + q] * ki[q];
        pn[a_] = accv;
        for (int b_ = 0; b_ < nx; ++b_) {
          double acc = S[a_ * nz + b_];
          for (int q = 0; q < nu; ++q) acc -= S[a_ * nz + nx + q] * Ki[q * nx + b_];
          Pn[a_ * nx + b_] = acc;
        }
      }
      for (int a_ = 0; a_ < nx; ++a_) { pv[a_] = pn[a_]; for (int b_ = 0; b_ < nx; ++b_) P[a_ * nx + b_] = 0.5 * (Pn[a_ * nx + b_] + Pn[b_ * nx + a_]); }
    }
    if (it == 1) R0 = rmax;
    /* forward sweep */
    memset(dZ, 0, sizeof(double) * N * nz);
    for (int i = 0; i + 1 < N; ++i) {
      const double* Ki = Kg + (size_t)i * nu * nx; const double* ki = kg + (size_t)i * nu;
      for (int a_ = 0; a_ < nu; ++a_) {
        double acc = -ki[a_];
        for (int q = 0; q < nx; ++q) acc -= Ki[a_ * nx + q] * dZ[i * nz + q];
        dZ[i * nz + nx + a_] = acc;
      }
      for (int a_ = 0; a_ < nx; ++a_) {
        double acc = 0;
        for (int q = 0; q < nx; ++q) acc += A[a_ * nx + q] * dZ[i * nz + q];
        for (int q = 0; q < nu; ++q) acc += B[a_ * nu + q] * dZ[i * nz + nx + q];
        dZ[(i + 1) * nz + a_] = acc;
      }
    }
    /* step length */
    double amax = 1e300;
    for (int k = 0; k < m; ++k) {
      const orow* r = &rows[k];
      double g = 0; const double* dz = dZ + (size_t)r->stage * nz;
      for (int q = 0; q < nz; ++q) g += r->co[q] * dz[q];
      gd[k] = g;
      dlam[k] = w[k] * g + kap[k];
      ds[k] = ((tau - s[k] * lam[k]) - s[k] * dlam[k]) / lam[k];
      if (ds[k] < 0) amax = fmin(amax, -s[k] / ds[k]);
      if (dlam[k] < 0) amax = fmin(amax, -lam[k] / dlam[k]);
      if (r->a == 0.0) {
        double t = tt[k], mu = RHO_EL - lam[k], dmu = -dlam[k], dt = ((tau - t * mu) - t * dmu) / mu;
        dtt[k] = dt;
        if (dt < 0) amax = fmin(amax, -t / dt);
        if (dmu < 0) amax = fmin(amax, -mu / dmu);
      }
    }
    double alpha = fmin(1.0, 0.995 * amax);
    if (getenv("ORC_QP_TRACE")) fprintf(stderr, "   it %d comp %.3e obj %.6f alpha %.4f resid %.2e\n", it, comp, obj, alpha, resid_fac * R0);
    for (int q = 0; q < N * nz; ++q) Z[q] += alpha * dZ[q];
    for (int k = 0; k < m; ++k) { s[k] += alpha * ds[k]; lam[k] += alpha * dlam[k]; tt[k] += alpha * dtt[k]; }
    resid_fac *= (1.0 - alpha); alpha_prev = alpha;
    if (alpha < 1e-12) break;
  }
  double viol = 0, slack_cost = 0, obj = 0;
  for (int i = 0; i < N; ++i)
    for (int q = 0; q < nz; ++q) { double d = Z[i * nz + q] - Rf[i * nz + q]; obj += Wd[q] * d * d; }
  for (int k = 0; k < m; ++k) {
    double c = -row_val(M, &rows[k], Z);
    if (rows[k].a == 0.0) { if (-c > viol) viol = -c; }
    else { double t = lam[k] / rows[k].a; slack_cost += 0.5 * rows[k].a * t * t; }
  }
  out->Z = Z; out->obj = obj + slack_cost; out->viol = viol; out->it = it > QP_MAXIT ? QP_MAXIT : it; out->ok = ok;
  out->slack_cost = slack_cost; out->lam = lam; out->dgap = dgap;
  free(A); free(B); free(Wd); free(Rf); free(start); free(order); free(s); free(ds); free(dlam); free(w); free(kap);
  free(gd); free(tt); free(dtt); free(Kg); free(kg); free(dZ); free(Phi); free(rr); free(P); free(pv); free(T); free(S); free(sv); free(Lc);
  free(Pn); free(pn);
  return ok;
}

static __thread double g_node_tol = QP_TOL; /* node relaxations: accurate to a small fraction of the requested MIP gap */
static int qp_solve(const dmodel* M, const orow* rows, int m, qpres* out) { return qp_solve_tol(M, rows, m, out, g_node_tol); }

/* ------------------------------------------------------------------ node relaxation rows */
static void node_rows(const dmodel* M, const signed char* fix, rowvec* v) {
  const oinst* I = M->I; int C = I->C, N = I->N;
  v->n = 0;
  for (int c = 0; c < C; ++c)
    for (int i = 0; i < N; ++i) {
      global_rows(M, v, c, i);
      if (i >= 1 && FIX_REG(M, fix, c, i) >= 0) region_rows(M, v, c, i, FIX_REG(M, fix, c, i));
    }
  for (int c = 0; c < C; ++c)
    for (int i = 1; i < N; ++i) {
      int code = FIX_REG(M, fix, c, i); int j = code >= 0 ? ALT_J(M, c, code) : -1;
      if (I->E > 0)
        for (int pt = 0; pt < 5; ++pt) {
          int e = I->E == 1 ? 0 : FIX_ENV(M, fix, c, i, pt);
          if (e < 0 || (pt > 0 && j < 0)) continue;
          env_rows(M, v, c, i, pt, e, j < 0 ? 0 : j);
        }
      for (int o = 0; o < I->O; ++o)
        for (int pt = 0; pt < 5; ++pt) {
          int k = FIX_OBS(M, fix, c, o, i, pt);
          if (k < 0 || k >= I->L || (pt > 0 && j < 0)) continue;
          obs_row(M, v, c, o, i, pt, k, j < 0 ? 0 : j);
        }
    }
  for (int p = 0; p < I->NP; ++p) {
    int c1, c2; pair_cars(I, p, &c1, &c2);
    for (int i = 1; i < N; ++i) {
      int k1 = FIX_REG(M, fix, c1, i), k2 = FIX_REG(M, fix, c2, i);
      int j1 = k1 >= 0 ? ALT_J(M, c1, k1) : -1, j2 = k2 >= 0 ? ALT_J(M, c2, k2) : -1;
      for (int g = 0; g < 4; ++g) {
        int alt = FIX_C2C(M, fix, p, i, g);
        if (alt < 0) continue;
        int need1 = (g == 2 || g == 3), need2 = (g == 1 || g == 3);
        if ((need1 && j1 < 0) || (need2 && j2 < 0)) continue;
        c2c_rows(M, v, p, i, g, alt, j1 < 0 ? 0 : j1, j2 < 0 ? 0 : j2);
      }
    }
  }
}

static double const_cost(const dmodel* M, const signed char* fix) {
  const oinst* I = M->I; double k = 0;
  for (int c = 0; c < I->C; ++c)
    for (int o = 0; o < I->O; ++o)
      for (int i = 0; i < I->N; ++i)
        for (int p = 0; p < 5; ++p)
          if (FIX_OBS(M, fix, c, o, i, p) >= I->L) k += I->w_slack_obs;
  return k;
}

/* ------------------------------------------------------------------ completion */
typedef struct { int step, kind; int key[5]; int nalts; int alts[64]; double mag; } violation;
static __thread double g_cur_viol = 0.0; /* violation of the least violated alternative of the disjunction being reported */

static int region_cands(const dmodel* M, const signed char* fix, int c, int i, int* out) {
  const oinst* I = M->I;
  int prevj = -2; /* -2 unknown */
  if (i == 1) prevj = I->init_region[c] - 1;
  else if (FIX_REG(M, fix, c, i - 1) >= 0) prevj = ALT_J(M, c, FIX_REG(M, fix, c, i - 1));
  int nxt = (i + 1 < I->N) ? FIX_REG(M, fix, c, i + 1) : -1;
  int n = 0;
  for (int q = 0; q < M->nposs[c]; ++q) {
    int j = M->plist[c][q];
    for (int h = 0; h < 4; ++h) {
      if (h < 3 && h >= M->nhs[j]) continue;
      if (h == 3 && prevj != -2 && prevj != j) continue;
      if (nxt >= 0 && ALT_H(nxt) == 3 && ALT_J(M, c, nxt) != j) continue;
      out[n++] = q * 4 + h;
    }
  }
  return n;
}

static double alt_viol_region(const dmodel* M, rowvec* tmp, int c, int i, int code, const double* Z) {
  tmp->n = 0; region_rows(M, tmp, c, i, code);
  double v = 0;
  for (int k = 0; k < tmp->n; ++k) { double x = row_val(M, &tmp->r[k], Z); if (x > v) v = x; }
  return v;
}

static void sort_alts(int n, int* alts, double* vals) {
  for (int a = 1; a < n; ++a) {
    int ia = alts[a]; double va = vals[a]; int b = a - 1;
    while (b >= 0 && vals[b] > va) { alts[b + 1] = alts[b]; vals[b + 1] = vals[b]; b--; }
    alts[b + 1] = ia; vals[b + 1] = va;
  }
}

/* branching order: 3 (default) = car/car disjunctions before obstacle, environment and region ones, the most violated
   first within a kind, and the earliest violated step until an incumbent exists (the order of the device solver);
   0 earliest step, 1 latest step, 2 most violated (experiments: ORC_BRANCH) */
static int g_branch_rule = 3;
static __thread int g_rule_now = 0;
static void viol_consider(violation* best, int step, int kind, const int* key, int nalts, const int* alts) {
  const int g_branch_rule = g_rule_now;
  if (g_branch_rule == 3) { if (best->step >= 0 && (best->kind > kind || (best->kind == kind && best->mag >= g_cur_viol))) return; }
  else if (g_branch_rule == 0) { if (best->step >= 0 && (best->step < step || (best->step == step && best->kind <= kind))) return; }
  else if (g_branch_rule == 1) { if (best->step >= 0 && (best->step > step || (best->step == step && best->kind <= kind))) return; }
  else { if (best->step >= 0 && best->mag >= g_cur_viol) return; }
  best->mag = g_cur_viol;
  best->step = step; best->kind = kind; memcpy(best->key, key, sizeof(int) * 5);
  best->nalts = nalts > 64 ? 64 : nalts; memcpy(best->alts, alts, sizeof(int) * best->nalts);
}

static void viol_region(const dmodel* M, const signed char* fix, violation* best, int c, int i) {
  int cands[64]; int n = region_cands(M, fix, c, i, cands);
  int key[5] = {'r', c, i, 0, 0};
  viol_consider(best, i, 0, key, n, cands);
}

/* returns 1 when integer feasible; comp receives the completed fix record; otherwise *best is the
 * disjunction to branch on (earliest step, region first) */
static int complete(const dmodel* M, const signed char* fix, const double* Z, signed char* comp, violation* best,
                    double tol) {
  const oinst* I = M->I; int C = I->C, N = I->N;
  rowvec tmp = {0, 0, 0};
  memcpy(comp, fix, M->fixlen);
  best->step = -1; best->nalts = 0;
  for (int c = 0; c < C; ++c) {
    int prevj = I->init_region[c] - 1;
    for (int i = 1; i < N; ++i) {
      if (FIX_REG(M, fix, c, i) >= 0) { prevj = ALT_J(M, c, FIX_REG(M, fix, c, i)); continue; }
      int cands[64]; int n = region_cands(M, fix, c, i, cands);
      /* canonical choice (ties at sector borders are common): the first non-slow alternative, in (region, half-plane)
       * order, whose rows hold within tol; else the slow alternative if it holds; else the least violated one */
      int bi = -1; double bv = 1e300; int first_ok = -1, slow_ok = -1;
      for (int q = 0; q < n; ++q) {
        if (ALT_H(cands[q]) == 3 && ALT_J(M, c, cands[q]) != prevj) continue;
        double v = alt_viol_region(M, &tmp, c, i, cands[q], Z);
        if (v < bv) { bv = v; bi = cands[q]; }
        if (v <= tol) { if (ALT_H(cands[q]) != 3) { if (first_ok < 0) first_ok = cands[q]; } else if (slow_ok < 0) slow_ok = cands[q]; }
      }
      if (first_ok >= 0) { bi = first_ok; bv = 0.0; } else if (slow_ok >= 0) { bi = slow_ok; bv = 0.0; }
      if (bi < 0 || bv > tol) {
        g_cur_viol = bv < 1e299 ? bv : 1e3;
        viol_region(M, fix, best, c, i);
        if (bi < 0) bi = n > 0 ? cands[0] : 0;
      }
      FIX_REG(M, comp, c, i) = (signed char)bi;
      prevj = ALT_J(M, c, bi);
    }
  }
  for (int c = 0; c < C; ++c)
    for (int i = 1; i < N; ++i) {
      int j = ALT_J(M, c, FIX_REG(M, comp, c, i));
      int runfixed = FIX_REG(M, fix, c, i) < 0;
      if (I->E >= 1)
        for (int pt = 0; pt < 5; ++pt) {
          if (I->E == 1) {
            if (pt > 0 && runfixed) {
              tmp.n = 0; env_rows(M, &tmp, c, i, pt, 0, j);
              double v = 0; for (int k = 0; k < tmp.n; ++k) v = fmax(v, row_val(M, &tmp.r[k], Z));
              if (v > tol) { g_cur_viol = v; viol_region(M, fix, best, c, i); }
            }
            FIX_ENV(M, comp, c, i, pt) = 0;
            continue;
          }
          int fx = FIX_ENV(M, fix, c, i, pt);
          if (fx >= 0 && !(pt > 0 && runfixed)) continue;
          int alts[64]; double vals[64] = {0}; int ne = I->E > 64 ? 64 : I->E;
          for (int e = 0; e < ne; ++e) {
            tmp.n = 0; env_rows(M, &tmp, c, i, pt, e, j);
            double v = -1e300; for (int k = 0; k < tmp.n; ++k) v = fmax(v, row_val(M, &tmp.r[k], Z));
            alts[e] = e; vals[e] = v;
          }
          int okk;
          if (fx >= 0) okk = vals[fx] <= tol;
          else { sort_alts(ne, alts, vals); okk = vals[0] <= tol; FIX_ENV(M, comp, c, i, pt) = (signed char)alts[0]; }
          if (!okk) {
            g_cur_viol = fx >= 0 ? vals[fx] : vals[0];
            if (pt > 0 && runfixed) viol_region(M, fix, best, c, i);
            else { int key[5] = {'e', c, i, pt, 0}; viol_consider(best, i, 1, key, ne, alts); }
          }
        }
      for (int o = 0; o < I->O; ++o)
        for (int pt = 0; pt < 5; ++pt) {
          int fx = FIX_OBS(M, fix, c, o, i, pt);
          if (fx >= 0 && (fx >= I->L || !(pt > 0 && runfixed))) continue;
          int alts[64]; double vals[64];
          for (int k = 0; k < I->L; ++k) {
            tmp.n = 0; obs_row(M, &tmp, c, o, i, pt, k, j);
            alts[k] = k; vals[k] = row_val(M, &tmp.r[0], Z);
          }
          int okk;
          if (fx >= 0) okk = vals[fx] <= tol;
          else { sort_alts(I->L, alts, vals); okk = vals[0] <= tol; FIX_OBS(M, comp, c, o, i, pt) = (signed char)alts[0]; }
          if (!okk) {
            g_cur_viol = fx >= 0 ? vals[fx] : vals[0];
            if (pt > 0 && runfixed) viol_region(M, fix, best, c, i);
            else {
              int na = I->L; if (I->obs_soft[o]) alts[na++] = I->L;
              int key[5] = {'o', c, o, i, pt}; viol_consider(best, i, 2, key, na, alts);
            }
          }
        }
    }
  for (int p = 0; p < I->NP; ++p) {
    int c1, c2; pair_cars(I, p, &c1, &c2);
    for (int i = 1; i < N; ++i) {
      int j1 = ALT_J(M, c1, FIX_REG(M, comp, c1, i)), j2 = ALT_J(M, c2, FIX_REG(M, comp, c2, i));
      for (int g = 0; g < 4; ++g) {
        int need1 = (g == 2 || g == 3), need2 = (g == 1 || g == 3);
        int unf = -1;
        if (need1 && FIX_REG(M, fix, c1, i) < 0) unf = c1;
        else if (need2 && FIX_REG(M, fix, c2, i) < 0) unf = c2;
        int fx = FIX_C2C(M, fix, p, i, g);
        if (fx >= 0 && unf < 0) continue;
        int alts[4]; double vals[4];
        for (int a = 0; a < 4; ++a) {
          tmp.n = 0; c2c_rows(M, &tmp, p, i, g, a, j1, j2);
          alts[a] = a; vals[a] = row_val(M, &tmp.r[tmp.n - 1], Z); /* zero-slack form */
        }
        int okk;
        if (fx >= 0) okk = vals[fx] <= tol;
        else { sort_alts(4, alts, vals); okk = vals[0] <= tol; FIX_C2C(M, comp, p, i, g) = (signed char)alts[0]; }
        if (!okk) {
          g_cur_viol = fx >= 0 ? vals[fx] : vals[0];
          if (unf >= 0) viol_region(M, fix, best, unf, i);
          else { int key[5] = {'a', p, i, g, 0}; viol_consider(best, i, 3, key, 4, alts); }
        }
      }
    }
  }
  free(tmp.r);
  return best->step < 0;
}

/* ------------------------------------------------------------------ presolve of step 1 (all constants) */
static int step0_check(const dmodel* M, double* const_obj) {
  const oinst* I = M->I; double tol = FEAS_TOL; rowvec tmp = {0, 0, 0};
  double Z0[ORC_NZMAX]; memset(Z0, 0, sizeof(Z0));
  for (int c = 0; c < I->C; ++c) for (int k = 0; k < 6; ++k) Z0[6 * c + k] = I->x0[c * 6 + k];
  *const_obj = 0;
  int ok = 1;
  for (int c = 0; c < I->C && ok; ++c) {
    if (I->E >= 1)
      for (int pt = 0; pt < 5 && ok; ++pt) {
        int any = 0;
        for (int e = 0; e < I->E && !any; ++e) {
          tmp.n = 0; env_rows(M, &tmp, c, 0, pt, e, 0);
          double v = -1e300; for (int k = 0; k < tmp.n; ++k) v = fmax(v, row_val(M, &tmp.r[k], Z0));
          if (v <= tol) any = 1;
        }
        if (!any) ok = 0;
      }
    for (int o = 0; o < I->O && ok; ++o)
      for (int pt = 0; pt < 5 && ok; ++pt) {
        int any = 0;
        for (int k = 0; k < I->L && !any; ++k) { tmp.n = 0; obs_row(M, &tmp, c, o, 0, pt, k, 0); if (row_val(M, &tmp.r[0], Z0) <= tol) any = 1; }
        if (!any) { if (I->obs_soft[o]) *const_obj += I->w_slack_obs; else ok = 0; }
      }
  }
  for (int p = 0; p < I->NP && ok; ++p)
    for (int g = 0; g < 4 && ok; ++g) {
      double bestc = 1e300;
      for (int a = 0; a < 4; ++a) {
        tmp.n = 0; int n = c2c_rows(M, &tmp, p, 0, g, a, 0, 0);
        double vlim = row_val(M, &tmp.r[0], Z0);
        if (vlim > tol) continue;
        double need = n > 1 ? fmax(0.0, row_val(M, &tmp.r[n - 1], Z0)) : 0.0;
        double cst = I->w_slack * need * need;
        if (cst < bestc) bestc = cst;
      }
      if (bestc > 1e299) ok = 0; else *const_obj += bestc;
    }
  free(tmp.r);
  return ok;
}

/* ------------------------------------------------------------------ results */
static void fill_results(const dmodel* M, const signed char* comp, const double* Z, miqp_raw_results_c* r) {
  const oinst* I = M->I; int C = I->C, N = I->N, R = I->R, E = I->E, O = I->O, L = I->L, K = I->K, nz = M->nz;
  double tol = 10 * FEAS_TOL; rowvec tmp = {0, 0, 0};
  double Zs0[ORC_NZMAX];
  for (int c = 0; c < C; ++c)
    for (int i = 0; i < N; ++i) {
      const double* z = Z + (size_t)i * nz; int q = c * N + i;
      r->pos_x[q] = z[6 * c + SX]; r->vel_x[q] = z[6 * c + SVX]; r->acc_x[q] = z[6 * c + SAX];
      r->pos_y[q] = z[6 * c + SY]; r->vel_y[q] = z[6 * c + SVY]; r->acc_y[q] = z[6 * c + SAY];
      r->u_x[q] = z[6 * C + 2 * c]; r->u_y[q] = z[6 * C + 2 * c + 1];
      int code = i >= 1 ? FIX_REG(M, comp, c, i) : -1;
      int j = i >= 1 ? ALT_J(M, c, code) : I->init_region[c] - 1;
      double co[ORC_NZMAX], k;
      double* dst[4] = {r->pos_x_front_UB, r->pos_x_front_LB, r->pos_y_front_UB, r->pos_y_front_LB};
      const int ty[4] = {PT_U, PT_L, PT_U, PT_L}; const int wh[4] = {0, 0, 1, 1};
      for (int t = 0; t < 4; ++t) {
        point_affine(M, c, i, j, ty[t], wh[t], co, &k);
        double v = k; for (int a = 0; a < nz; ++a) v += co[a] * z[a];
        dst[t][q] = v;
      }
      for (int jj = 0; jj < R; ++jj) r->active_region[(c * N + i) * R + jj] = jj == j ? 1 : 0;
      int xp = 0, yp = 0, xn = 0, yn = 0, cb = 0;
      if (i >= 1) {
        double vx = z[6 * c + SVX], vy = z[6 * c + SVY];
        xp = vx <= I->vm; xn = vx >= -I->vm; yp = vy <= I->vm; yn = vy >= -I->vm;
        int h = ALT_H(code);
        if (h == 3) { xp = xn = yp = yn = cb = 1; }
        else {
          int ax = M->hs[j][h][0], sg = M->hs[j][h][1];
          if (ax == 0 && sg > 0) xp = 0;
          if (ax == 0 && sg < 0) xn = 0;
          if (ax == 1 && sg > 0) yp = 0;
          if (ax == 1 && sg < 0) yn = 0;
          cb = 0;
        }
      }
      r->region_change_not_allowed_x_positive[q] = xp; r->region_change_not_allowed_y_positive[q] = yp;
      r->region_change_not_allowed_x_negative[q] = xn; r->region_change_not_allowed_y_negative[q] = yn;
      r->region_change_not_allowed_combined[q] = cb;
      /* environment: canonical form = 0 for every piece that contains the point */
      int* nw[5] = {r->notWithinEnvironmentRear, r->notWithinEnvironmentFrontUbUb, r->notWithinEnvironmentFrontLbUb,
                    r->notWithinEnvironmentFrontUbLb, r->notWithinEnvironmentFrontLbLb};
      const double* zz = z; (void)Zs0;
      for (int pt = 0; pt < 5; ++pt)
        for (int e = 0; e < E; ++e) {
          tmp.n = 0; env_rows(M, &tmp, c, i, pt, e, j);
          double v = -1e300;
          for (int kk = 0; kk < tmp.n; ++kk) { orow rr = tmp.r[kk]; rr.stage = 0; v = fmax(v, row_val(M, &rr, zz)); }
          nw[pt][(c * E + e) * N + i] = v <= tol ? 0 : 1;
        }
      for (int o = 0; o < O; ++o) {
        for (int pt = 0; pt < 5; ++pt) {
          int ignored = i >= 1 && FIX_OBS(M, comp, c, o, i, pt) >= L;
          for (int kk = 0; kk < L; ++kk) {
            tmp.n = 0; obs_row(M, &tmp, c, o, i, pt, kk, j);
            orow rr = tmp.r[0]; rr.stage = 0;
            int d = (row_val(M, &rr, zz) <= tol && !ignored) ? 0 : 1;
            if (pt == 0) r->deltacc[((c * O + o) * N + i) * L + kk] = d;
            else r->deltacc_front[(((c * O + o) * N + i) * L + kk) * 4 + pt - 1] = d;
          }
          if (pt == 0) r->slackvarsObstacle[(c * O + o) * N + i] = ignored ? 1 : 0;
          else r->slackvarsObstacle_front[((c * O + o) * N + i) * 4 + pt - 1] = ignored ? 1 : 0;
        }
      }
    }
  for (int a = 0; a < K * K * N * 16; ++a) r->car2car_collision[a] = 0;
  for (int a = 0; a < K * K * N * 4; ++a) { r->slackvars[a] = 0; if (r->slackvars_real) r->slackvars_real[a] = 0; }
  for (int p = 0; p < I->NP; ++p) {
    int c1, c2; pair_cars(I, p, &c1, &c2);
    for (int i = 0; i < N; ++i) {
      int j1 = i >= 1 ? ALT_J(M, c1, FIX_REG(M, comp, c1, i)) : 0, j2 = i >= 1 ? ALT_J(M, c2, FIX_REG(M, comp, c2, i)) : 0;
      const double* zz = Z + (size_t)i * nz;
      for (int g = 0; g < 4; ++g) {
        double sl[2] = {0, 0}; /* slack of the x rows, of the y rows */
        int chosen = i >= 1 ? FIX_C2C(M, comp, p, i, g) : -1;
        double vals[4];
        for (int a = 0; a < 4; ++a) {
          tmp.n = 0; c2c_rows(M, &tmp, p, i, g, a, j1, j2);
          orow rr = tmp.r[tmp.n - 1]; rr.stage = 0;
          double zloc[ORC_NZMAX]; memcpy(zloc, zz, sizeof(double) * nz);
          vals[a] = row_val(M, &rr, zloc);
        }
        if (chosen < 0) { chosen = 0; for (int a = 1; a < 4; ++a) if (vals[a] < vals[chosen]) chosen = a; }
        if ((g == 0 || g == 3) && vals[chosen] > 0) sl[chosen < 2 ? 0 : 1] = vals[chosen];
        for (int a = 0; a < 4; ++a) {
          double need = vals[a] - ((g == 0 || g == 3) ? sl[a < 2 ? 0 : 1] : 0.0);
          r->car2car_collision[((c1 * K + (c2 - 1)) * N + i) * 16 + 4 * g + a] = (a == chosen || need <= tol) ? 0 : 1;
        }
        if (g == 0 || g == 3)
          for (int q = 0; q < 2; ++q) {
            int idx = ((c1 * K + (c2 - 1)) * N + i) * 4 + (g == 0 ? 0 : 2) + q;
            r->slackvars[idx] = (int)sl[q];
            if (r->slackvars_real) r->slackvars_real[idx] = sl[q];
          }
      }
    }
  }
  free(tmp.r);
}

/* ------------------------------------------------------------------ B&B */
typedef struct { double bound; long long seq; signed char* fix; int depth; } bnode;
typedef struct { bnode* a; int n, cap; } heap_t;

/* open list as a plain array: node selection is best-bound, interleaved with dives (deepest node, ties by
 * bound) while no incumbent exists and on every 4th selection afterwards (standard B&B practice; the search
 * order does not change what is proven, only how fast incumbents appear) */
static void heap_push(heap_t* h, bnode nd) {
  if (h->n == h->cap) { h->cap = h->cap ? 2 * h->cap : 1024; h->a = (bnode*)realloc(h->a, sizeof(bnode) * h->cap); }
  h->a[h->n++] = nd;
}
static double heap_min_bound(const heap_t* h) {
  double b = INFINITY;
  for (int k = 0; k < h->n; ++k) if (h->a[k].bound < b) b = h->a[k].bound;
  return b;
}
static bnode heap_pop(heap_t* h, int dive) {
  int best = 0;
  for (int k = 1; k < h->n; ++k) {
    const bnode* x = &h->a[k]; const bnode* y = &h->a[best];
    int better;
    if (dive) better = x->depth > y->depth || (x->depth == y->depth && (x->bound < y->bound || (x->bound == y->bound && x->seq < y->seq)));
    else better = x->bound < y->bound || (x->bound == y->bound && x->seq < y->seq);
    if (better) best = k;
  }
  bnode top = h->a[best]; h->a[best] = h->a[--h->n];
  return top;
}

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int orc_solve(const oinst* I, const orc_opts* o, miqp_raw_results_c* res, miqp_solution_properties_c* props) {
  double t0 = now_s();
  if (getenv("ORC_BRANCH")) g_branch_rule = atoi(getenv("ORC_BRANCH"));
  dmodel* M = dm_new(I);
  double gap = (o && o->gap >= 0) ? o->gap : I->gap;
  double tlim = (o && o->time_limit > 0) ? o->time_limit : I->tilim;
  g_node_tol = fmin(QP_TOL, fmax(1e-12, 1e-4 * gap));
  long long max_nodes = (o && o->max_nodes > 0) ? o->max_nodes : 2000000;
  int verbose = o ? o->verbose : 0;
  orc_sizes sz = orc_raw_sizes(I);
  props->NrConstraints = sz.rows; props->NrBinaryVariables = sz.bin; props->NrFloatVariables = sz.cont;
  props->NonZeroCoefficients = sz.nnz; props->NrIterations = 0; props->NrSolutionPool = 0; props->nodes = 0;
  double inc = INFINITY, best_bound = -INFINITY, cobj0 = 0;
  signed char* inc_fix = (signed char*)malloc(M->fixlen); double* incZ = (double*)malloc(sizeof(double) * I->N * M->nz);
  signed char* comp = (signed char*)malloc(M->fixlen);
  int timed_out = 0;
  heap_t H = {0, 0, 0}; long long seq = 0;
  rowvec rows = {0, 0, 0};
  if (step0_check(M, &cobj0)) {
    bnode root; root.bound = -INFINITY; root.seq = seq++; root.depth = 0; root.fix = (signed char*)malloc(M->fixlen);
    memset(root.fix, -1, M->fixlen);
    heap_push(&H, root);
  }
  long long pops = 0;
  double dropped_min = INFINITY; /* smallest bound of a node that was dropped because it lies within the gap */
  while (H.n > 0) {
    double lb = heap_min_bound(&H);
    if (inc < INFINITY && (inc - lb) <= gap * (1e-10 + fabs(inc))) { best_bound = lb; break; }
    if (props->nodes >= max_nodes || now_s() - t0 > tlim) { timed_out = 1; best_bound = lb; break; }
    bnode nd = heap_pop(&H, !(inc < INFINITY) || (pops++ % 4) == 3);
    if (inc < INFINITY && (inc - nd.bound) <= gap * (1e-10 + fabs(inc))) { if (nd.bound < dropped_min) dropped_min = nd.bound; free(nd.fix); continue; }
    props->nodes++;
    node_rows(M, nd.fix, &rows);
    qpres q; qp_solve(M, rows.r, rows.n, &q);
    props->NrIterations += q.it;
    double obj = q.obj + const_cost(M, nd.fix) + cobj0;
    /* what the node proves is the dual value of its relaxation: the primal value of the interior point iterate minus
       the remaining complementarity */
    double objlb = obj - fmax(0.0, q.dgap);
    if (q.viol > FEAS_TOL || !q.ok) {
      if (verbose > 1) fprintf(stderr, "  node %lld infeasible viol %.2e ok %d it %d\n", props->nodes, q.viol, q.ok, q.it);
    } else if (!(inc < INFINITY) || objlb < inc - 1e-12 * fabs(inc)) {
      violation vb;
      g_rule_now = (g_branch_rule == 3 && !(inc < INFINITY)) ? 0 : g_branch_rule;
      if (complete(M, nd.fix, q.Z, comp, &vb, FEAS_TOL)) {
        if (!(inc < INFINITY) || obj < inc) { inc = obj; memcpy(inc_fix, comp, M->fixlen); memcpy(incZ, q.Z, sizeof(double) * I->N * M->nz); }
        props->NrSolutionPool++;
        if (verbose) fprintf(stderr, "  node %lld incumbent %.8f depth %d open %d\n", props->nodes, obj, nd.depth, H.n);
      } else {
        if (verbose) { static long hist[4][64]; static long hist_cnt = 0; int kd = vb.key[0]=='r'?0:(vb.key[0]=='e'?1:(vb.key[0]=='o'?2:3)); int stp = kd==2?vb.key[3]:vb.key[2]; hist[kd][stp]++; if (++hist_cnt % 1000 == 0) { for (int k=0;k<4;++k){ fprintf(stderr,"kind %d:",k); for(int q=0;q<I->N;++q) fprintf(stderr," %ld",hist[k][q]); fprintf(stderr,"\n"); } } }
        for (int a = 0; a < vb.nalts; ++a) {
          bnode ch; ch.bound = objlb; ch.seq = seq++; ch.depth = nd.depth + 1; ch.fix = (signed char*)malloc(M->fixlen);
          memcpy(ch.fix, nd.fix, M->fixlen);
          switch (vb.key[0]) {
            case 'r': FIX_REG(M, ch.fix, vb.key[1], vb.key[2]) = (signed char)vb.alts[a]; break;
            case 'e': FIX_ENV(M, ch.fix, vb.key[1], vb.key[2], vb.key[3]) = (signed char)vb.alts[a]; break;
            case 'o': FIX_OBS(M, ch.fix, vb.key[1], vb.key[2], vb.key[3], vb.key[4]) = (signed char)vb.alts[a]; break;
            default: FIX_C2C(M, ch.fix, vb.key[1], vb.key[2], vb.key[3]) = (signed char)vb.alts[a]; break;
          }
          heap_push(&H, ch);
        }
      }
    }
    free(q.Z); free(q.lam); free(nd.fix);
  }
  if (!timed_out && H.n == 0 && best_bound == -INFINITY) best_bound = inc; /* tree exhausted */
  if (best_bound > dropped_min) best_bound = dropped_min;
  if (inc < INFINITY && best_bound > inc) best_bound = inc;
  for (int k = 0; k < H.n; ++k) free(H.a[k].fix);
  free(H.a); free(rows.r);
  int status;
  props->time = now_s() - t0;
  props->best_bound = best_bound;
  if (inc < INFINITY) {
    status = MIQP_STATUS_SUCCESS;
    props->objective = inc; props->gap = fabs(best_bound - inc) / (1e-10 + fabs(inc));
    props->status = timed_out ? MIQP_CPX_STAT_TIME_LIM_FEAS : (props->gap <= 1e-9 ? MIQP_CPX_STAT_OPTIMAL : MIQP_CPX_STAT_OPTIMAL_TOL);
    { /* polish: re-solve the incumbent's QP (all disjunctions fixed as completed) to a tight tolerance */
      rowvec pr = {0, 0, 0}; node_rows(M, inc_fix, &pr);
      qpres q; qp_solve_tol(M, pr.r, pr.n, &q, QP_TOL_FINAL);
      props->NrIterations += q.it;
      if (q.viol <= FEAS_TOL && (q.ok || q.it >= 10)) {
        memcpy(incZ, q.Z, sizeof(double) * I->N * M->nz);
        props->objective = q.obj + const_cost(M, inc_fix) + cobj0;
        if (best_bound > props->objective) best_bound = props->objective;
        props->best_bound = best_bound;
        props->gap = fabs(best_bound - props->objective) / (1e-10 + fabs(props->objective));
      }
      free(q.Z); free(q.lam); free(pr.r);
    }
    if (res) fill_results(M, inc_fix, incZ, res);
  } else {
    status = timed_out ? MIQP_STATUS_FAILED_TIMEOUT : MIQP_STATUS_FAILED_NO_SOLUT;
    props->objective = NAN; props->gap = NAN;
    props->status = timed_out ? MIQP_CPX_STAT_TIME_LIM_INFEAS : MIQP_CPX_STAT_INFEASIBLE;
  }
  free(inc_fix); free(incZ); free(comp); dm_free(M);
  return status;
}

/* ------------------------------------------------------------------ fixed-binary QP (pins the QP machinery on K3) */
int orc_solve_fixed(const oinst* I, const miqp_raw_results_c* f, miqp_raw_results_c* res, double* objective, int* iters) {
  dmodel* M = dm_new(I); int C = I->C, N = I->N, R = I->R, E = I->E, O = I->O, L = I->L, K = I->K;
  rowvec rows = {0, 0, 0};
  signed char* comp = (signed char*)malloc(M->fixlen); memset(comp, -1, M->fixlen);
  for (int c = 0; c < C; ++c)
    for (int i = 0; i < N; ++i) {
      global_rows(M, &rows, c, i);
      if (i == 0) continue;
      int j = -1;
      for (int jj = 0; jj < R; ++jj) if (f->active_region[(c * N + i) * R + jj] == 1) j = jj;
      int pidx = -1;
      for (int q = 0; q < M->nposs[c]; ++q) if (M->plist[c][q] == j) pidx = q;
      if (pidx < 0) { free(rows.r); free(comp); dm_free(M); return -1; }
      int slow = f->region_change_not_allowed_combined[c * N + i] == 1;
      int h = 3;
      if (!slow) { /* the half-plane implied by the rc binaries */
        h = 0;
        for (int q = 0; q < M->nhs[j]; ++q) {
          int ax = M->hs[j][q][0], sg = M->hs[j][q][1];
          int b = ax == 0 ? (sg > 0 ? f->region_change_not_allowed_x_positive[c * N + i] : f->region_change_not_allowed_x_negative[c * N + i])
                          : (sg > 0 ? f->region_change_not_allowed_y_positive[c * N + i] : f->region_change_not_allowed_y_negative[c * N + i]);
          if (b == 0) { h = q; break; }
        }
      }
      int code = pidx * 4 + h;
      FIX_REG(M, comp, c, i) = (signed char)code;
      region_rows(M, &rows, c, i, code);
      const int* nw[5] = {f->notWithinEnvironmentRear, f->notWithinEnvironmentFrontUbUb, f->notWithinEnvironmentFrontLbUb,
                          f->notWithinEnvironmentFrontUbLb, f->notWithinEnvironmentFrontLbLb};
      for (int pt = 0; pt < 5; ++pt)
        for (int e = 0; e < E; ++e)
          if (nw[pt][(c * E + e) * N + i] == 0) { env_rows(M, &rows, c, i, pt, e, j); FIX_ENV(M, comp, c, i, pt) = (signed char)e; }
      for (int o = 0; o < O; ++o)
        for (int pt = 0; pt < 5; ++pt)
          for (int k = 0; k < L; ++k) {
            int d = pt == 0 ? f->deltacc[((c * O + o) * N + i) * L + k] : f->deltacc_front[(((c * O + o) * N + i) * L + k) * 4 + pt - 1];
            if (d == 0) { obs_row(M, &rows, c, o, i, pt, k, j); FIX_OBS(M, comp, c, o, i, pt) = (signed char)k; }
          }
    }
  for (int p = 0; p < I->NP; ++p) {
    int c1, c2; pair_cars(I, p, &c1, &c2);
    for (int i = 1; i < N; ++i) {
      int j1 = ALT_J(M, c1, FIX_REG(M, comp, c1, i)), j2 = ALT_J(M, c2, FIX_REG(M, comp, c2, i));
      for (int g = 0; g < 4; ++g)
        for (int a = 0; a < 4; ++a)
          if (f->car2car_collision[((c1 * K + (c2 - 1)) * N + i) * 16 + 4 * g + a] == 0) {
            c2c_rows(M, &rows, p, i, g, a, j1, j2); FIX_C2C(M, comp, p, i, g) = (signed char)a;
          }
    }
  }
  qpres q; qp_solve_tol(M, rows.r, rows.n, &q, QP_TOL_FINAL);
  if (objective) *objective = q.obj;
  if (iters) *iters = q.it;
  int rc = (q.ok && q.viol <= FEAS_TOL) ? 0 : 1;
  if (res && rc == 0) {
    /* fill_results needs a complete record: take unset leaf choices from completion */
    signed char* c2 = (signed char*)malloc(M->fixlen); violation vb;
    complete(M, comp, q.Z, c2, &vb, FEAS_TOL);
    for (int k = 0; k < M->fixlen; ++k) if (comp[k] < 0) comp[k] = c2[k];
    fill_results(M, comp, q.Z, res);
    free(c2);
  }
  free(q.Z); free(q.lam); free(rows.r); free(comp); dm_free(M);
  return rc;
}
