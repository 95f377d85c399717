/* raw_model.c - row-by-row restatement of the OPL model (TEST INFRASTRUCTURE).
 *
 * Every `subject to` statement of the cplexmodel .mod files is emitted as one sparse row over the raw
 * decision variables of cplexmodel/decision_variables.mod:10-53 - duplicates kept, `==` fixings
 * are rows, structural zero coefficients dropped - so that the row/column/non-zero counts can be
 * compared with the reference's own known answer (test/cplex_wrapper_test.cc:866-871:
 * 12361 rows, 1240 binaries, 340 continuous, 29834 non-zeros) and complete solution vectors of the
 * reference (K3, K5) can be checked for feasibility.
 */
#include "oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* big-M constants, cplexmodel/parameters.mod:24-32 */
#define M_JERK 10.0
#define M_VELFRAC 1000.0
#define M_POSPOLY 100.0
#define M_ACC 10.0
#define M_KAPPA 1000.0
#define M_VEL 100.0
#define M_ENV 10000.0
#define M_OBS 10000.0
#define M_AGENTS 1000.0

typedef struct {
  const oinst* I;
  /* variable block offsets */
  int o_cont[12]; /* u_x u_y pos_x vel_x acc_x pos_y vel_y acc_y pxfU pxfL pyfU pyfL : [c*N+i] */
  int o_sO, o_sOf, o_s;
  int o_nw[5]; /* Rear UbUb LbUb UbLb LbLb : [(c*E+e)*N+i] */
  int o_ar;    /* [(c*N+i)*R+j] */
  int o_rc[5]; /* xp yp xn yn comb : [c*N+i] */
  int o_d, o_df, o_cc;
  int ncont, nvar;
} vmap;

enum { V_UX = 0, V_UY, V_PX, V_VX, V_AX, V_PY, V_VY, V_AY, V_XFU, V_XFL, V_YFU, V_YFL };

static void vmap_init(vmap* m, const oinst* I) {
  int C = I->C, N = I->N, R = I->R, E = I->E, O = I->O, L = I->L, K = I->K;
  int o = 0;
  m->I = I;
  for (int k = 0; k < 12; ++k) { m->o_cont[k] = o; o += C * N; }
  m->o_sO = o; o += C * O * N;
  m->o_sOf = o; o += C * O * N * 4;
  m->o_s = o; o += K * K * N * 4;
  m->ncont = o;
  for (int k = 0; k < 5; ++k) { m->o_nw[k] = o; o += C * E * N; }
  m->o_ar = o; o += C * N * R;
  for (int k = 0; k < 5; ++k) { m->o_rc[k] = o; o += C * N; }
  m->o_d = o; o += C * O * N * L;
  m->o_df = o; o += C * O * N * L * 4;
  m->o_cc = o; o += K * K * N * 16;
  m->nvar = o;
}

#define VC(m, k, c, i) ((m)->o_cont[k] + (c) * (m)->I->N + (i))
#define VAR_AR(m, c, i, j) ((m)->o_ar + ((c) * (m)->I->N + (i)) * (m)->I->R + (j))
#define VAR_RC(m, k, c, i) ((m)->o_rc[k] + (c) * (m)->I->N + (i))
#define VAR_NW(m, k, c, e, i) ((m)->o_nw[k] + ((c) * (m)->I->E + (e)) * (m)->I->N + (i))
#define VAR_D(m, c, o, i, k) ((m)->o_d + (((c) * (m)->I->O + (o)) * (m)->I->N + (i)) * (m)->I->L + (k))
#define VAR_DF(m, c, o, i, k, q) ((m)->o_df + ((((c) * (m)->I->O + (o)) * (m)->I->N + (i)) * (m)->I->L + (k)) * 4 + (q))
#define VAR_SO(m, c, o, i) ((m)->o_sO + ((c) * (m)->I->O + (o)) * (m)->I->N + (i))
#define VAR_SOF(m, c, o, i, q) ((m)->o_sOf + (((c) * (m)->I->O + (o)) * (m)->I->N + (i)) * 4 + (q))
#define VAR_CC(m, a, b, i, q) ((m)->o_cc + (((a) * (m)->I->K + (b)) * (m)->I->N + (i)) * 16 + (q))
#define VAR_S(m, a, b, i, q) ((m)->o_s + (((a) * (m)->I->K + (b)) * (m)->I->N + (i)) * 4 + (q))

typedef void (*row_cb)(void* ctx, int n, const int* var, const double* coef, int sense, double rhs, const char* tag);

typedef struct { int n; int var[24]; double coef[24]; double k; } lin; /* sum coef*var + k */

static void lin_clear(lin* l) { l->n = 0; l->k = 0.0; }
static void lin_add(lin* l, int v, double c) {
  for (int q = 0; q < l->n; ++q)
    if (l->var[q] == v) { l->coef[q] += c; return; }
  l->var[l->n] = v; l->coef[l->n] = c; l->n++;
}
/* emit lhs (sense) rhs_lin : moves everything to the left, constants to the right */
static void emit(row_cb cb, void* ctx, const lin* lhs, int sense, const lin* rhs, const char* tag) {
  lin t; lin_clear(&t);
  for (int q = 0; q < lhs->n; ++q) lin_add(&t, lhs->var[q], lhs->coef[q]);
  for (int q = 0; q < rhs->n; ++q) lin_add(&t, rhs->var[q], -rhs->coef[q]);
  int var[24]; double coef[24]; int n = 0;
  for (int q = 0; q < t.n; ++q)
    if (t.coef[q] != 0.0) { var[n] = t.var[q]; coef[n] = t.coef[q]; n++; }
  cb(ctx, n, var, coef, sense, rhs->k - lhs->k, tag);
}

/* cross product term of obstacle_environment_constraints.mod:
 * (x2-x1)*(Y - y1) - (X - x1)*(y2-y1) as linear form in (vx: X variable, vy: Y variable) */
static void cross_lin(lin* l, const double* e, int vX, int vY) {
  lin_clear(l);
  double dx = e[2] - e[0], dy = e[3] - e[1];
  lin_add(l, vY, dx); lin_add(l, vX, -dy);
  l->k = -dx * e[1] + e[0] * dy;
}

static void enumerate_rows(const oinst* I, const vmap* m, row_cb cb, void* ctx) {
  int C = I->C, N = I->N, R = I->R, E = I->E, O = I->O, L = I->L, K = I->K;
  double ts = I->ts;
  lin a, b;
  /* ---- A1 initial_conditions.mod:11-61 ---- */
  for (int c = 0; c < C; ++c) {
    const int sv[6] = {V_PX, V_VX, V_AX, V_PY, V_VY, V_AY};
    for (int k = 0; k < 6; ++k) {
      lin_clear(&a); lin_add(&a, VC(m, sv[k], c, 0), 1); lin_clear(&b); b.k = I->x0[c * 6 + k];
      emit(cb, ctx, &a, 0, &b, "A1.state");
    }
    double th = atan2(I->x0[c * 6 + 4], I->x0[c * 6 + 1]);
    double fx = I->x0[c * 6 + 0] + cos(th) * I->wb[c], fy = I->x0[c * 6 + 3] + sin(th) * I->wb[c];
    const int fv[4] = {V_XFU, V_XFL, V_YFU, V_YFL};
    for (int k = 0; k < 4; ++k) {
      lin_clear(&a); lin_add(&a, VC(m, fv[k], c, 0), 1); lin_clear(&b); b.k = k < 2 ? fx : fy;
      emit(cb, ctx, &a, 0, &b, "A1.front");
    }
    for (int k = 0; k < 2; ++k) {
      lin_clear(&a); lin_add(&a, VC(m, k ? V_UY : V_UX, c, N - 1), 1); lin_clear(&b);
      emit(cb, ctx, &a, 0, &b, "A1.uN");
    }
  }
  for (int j = 0; j < R; ++j)
    for (int c = 0; c < C; ++c) {
      lin_clear(&a); lin_add(&a, VAR_AR(m, c, 0, j), 1); lin_clear(&b); b.k = (j + 1 == I->init_region[c]) ? 1 : 0;
      emit(cb, ctx, &a, 0, &b, "A1.region");
      const double* jl = I->jerk_lim + (c * R + j) * 4;
      for (int ax = 0; ax < 2; ++ax) {
        int u = VC(m, ax ? V_UY : V_UX, c, 0);
        lin_clear(&a); lin_add(&a, u, 1);
        lin_clear(&b); b.k = jl[2 * ax + 1] + M_JERK; lin_add(&b, VAR_AR(m, c, 0, j), -M_JERK);
        emit(cb, ctx, &a, -1, &b, "A1.jerk_ub");
        lin_clear(&b); b.k = jl[2 * ax] - M_JERK; lin_add(&b, VAR_AR(m, c, 0, j), M_JERK);
        emit(cb, ctx, &a, 1, &b, "A1.jerk_lb");
      }
    }
  for (int c = 0; c < C; ++c)
    for (int k = 0; k < 5; ++k) {
      lin_clear(&a); lin_add(&a, VAR_RC(m, k, c, 0), 1); lin_clear(&b);
      emit(cb, ctx, &a, 0, &b, "A1.rc");
    }
  /* ---- A2 dynamics model_region_constraints.mod:11-19 ---- */
  for (int i = 1; i < N; ++i)
    for (int c = 0; c < C; ++c)
      for (int ax = 0; ax < 2; ++ax) {
        int P_ = ax ? V_PY : V_PX, V_ = ax ? V_VY : V_VX, A_ = ax ? V_AY : V_AX, U_ = ax ? V_UY : V_UX;
        lin_clear(&a); lin_add(&a, VC(m, P_, c, i), 1);
        lin_clear(&b); lin_add(&b, VC(m, P_, c, i - 1), 1); lin_add(&b, VC(m, V_, c, i - 1), ts);
        lin_add(&b, VC(m, A_, c, i - 1), 0.5 * ts * ts); lin_add(&b, VC(m, U_, c, i - 1), ts * ts * ts / 6.0);
        emit(cb, ctx, &a, 0, &b, "A2.pos");
        lin_clear(&a); lin_add(&a, VC(m, V_, c, i), 1);
        lin_clear(&b); lin_add(&b, VC(m, V_, c, i - 1), 1); lin_add(&b, VC(m, A_, c, i - 1), ts);
        lin_add(&b, VC(m, U_, c, i - 1), 0.5 * ts * ts);
        emit(cb, ctx, &a, 0, &b, "A2.vel");
        lin_clear(&a); lin_add(&a, VC(m, A_, c, i), 1);
        lin_clear(&b); lin_add(&b, VC(m, A_, c, i - 1), 1); lin_add(&b, VC(m, U_, c, i - 1), ts);
        emit(cb, ctx, &a, 0, &b, "A2.acc");
      }
  /* ---- A3 global limits :22-39 (max_vel row on vel_x twice, none on vel_y: reference typo kept) ---- */
  for (int i = 0; i < N; ++i)
    for (int c = 0; c < C; ++c) {
      struct { int v; int sense; double k; } g[12] = {
          {VC(m, V_VX, c, i), 1, I->vmin}, {VC(m, V_VY, c, i), 1, I->vmin}, {VC(m, V_VX, c, i), -1, I->vmax},
          {VC(m, V_VX, c, i), -1, I->vmax}, {VC(m, V_AX, c, i), -1, I->amax}, {VC(m, V_AX, c, i), 1, I->amin},
          {VC(m, V_AY, c, i), -1, I->amax}, {VC(m, V_AY, c, i), 1, I->amin}, {VC(m, V_UX, c, i), -1, I->jmax},
          {VC(m, V_UX, c, i), 1, I->jmin}, {VC(m, V_UY, c, i), -1, I->jmax}, {VC(m, V_UY, c, i), 1, I->jmin}};
      for (int k = 0; k < 12; ++k) {
        lin_clear(&a); lin_add(&a, g[k].v, 1); lin_clear(&b); b.k = g[k].k;
        emit(cb, ctx, &a, g[k].sense, &b, "A3");
      }
    }
  /* ---- A4 region block :43-117 ---- */
  for (int i = 1; i < N; ++i)
    for (int c = 0; c < C; ++c) {
      int vx = VC(m, V_VX, c, i), vy = VC(m, V_VY, c, i), axv = VC(m, V_AX, c, i), ayv = VC(m, V_AY, c, i);
      int rcc = VAR_RC(m, 4, c, i);
      for (int j = 0; j < R; ++j) {
        int ar = VAR_AR(m, c, i, j);
        if (I->possible[c * R + j] == 1) {
          const double* F = I->frac + j * 4;
          lin_clear(&a); lin_add(&a, vy, F[0]);
          lin_clear(&b); lin_add(&b, vx, F[1]); b.k = -M_VELFRAC; lin_add(&b, ar, M_VELFRAC); lin_add(&b, rcc, -M_VELFRAC);
          emit(cb, ctx, &a, 1, &b, "A4.frac_lo");
          lin_clear(&a); lin_add(&a, vy, F[2]);
          lin_clear(&b); lin_add(&b, vx, F[3]); b.k = M_VELFRAC; lin_add(&b, ar, -M_VELFRAC); lin_add(&b, rcc, M_VELFRAC);
          emit(cb, ctx, &a, -1, &b, "A4.frac_hi");
          const int fv[4] = {V_XFU, V_XFL, V_YFU, V_YFL};
          const int pv[4] = {V_PX, V_PX, V_PY, V_PY};
          const int pt[4] = {OP_COSS_UB, OP_COSS_LB, OP_SINT_UB, OP_SINT_LB};
          for (int k = 0; k < 4; ++k) {
            const double* p = I->poly[pt[k]] + j * 3;
            double wb = I->wb[c];
            lin_clear(&b); lin_add(&b, VC(m, fv[k], c, i), 1); lin_add(&b, VC(m, pv[k], c, i), -1);
            b.k = -wb * p[0]; lin_add(&b, vx, -wb * p[1]); lin_add(&b, vy, -wb * p[2]);
            lin_clear(&a); a.k = -M_POSPOLY; lin_add(&a, ar, M_POSPOLY);
            emit(cb, ctx, &a, -1, &b, "A4.front_lo");
            lin_clear(&a); a.k = M_POSPOLY; lin_add(&a, ar, -M_POSPOLY);
            emit(cb, ctx, &a, 1, &b, "A4.front_hi");
          }
          const double* jl = I->jerk_lim + (c * R + j) * 4;
          const double* al = I->acc_lim + (c * R + j) * 4;
          for (int ax = 0; ax < 2; ++ax) {
            int u = VC(m, ax ? V_UY : V_UX, c, i);
            lin_clear(&a); lin_add(&a, u, 1);
            lin_clear(&b); b.k = jl[2 * ax + 1] + M_JERK; lin_add(&b, ar, -M_JERK);
            emit(cb, ctx, &a, -1, &b, "A4.jerk_ub");
            lin_clear(&b); b.k = jl[2 * ax] - M_JERK; lin_add(&b, ar, M_JERK);
            emit(cb, ctx, &a, 1, &b, "A4.jerk_lb");
          }
          for (int ax = 0; ax < 2; ++ax) {
            int av = ax ? ayv : axv;
            lin_clear(&a); lin_add(&a, av, 1);
            lin_clear(&b); b.k = al[2 * ax + 1] + M_ACC; lin_add(&b, ar, -M_ACC);
            emit(cb, ctx, &a, -1, &b, "A4.acc_ub");
            lin_clear(&b); b.k = al[2 * ax] - M_ACC; lin_add(&b, ar, M_ACC);
            emit(cb, ctx, &a, 1, &b, "A4.acc_lb");
          }
          double rho = (F[1] + F[3]) / (F[0] + F[2]);
          const double* kx = I->poly[OP_KMAX] + j * 3; const double* kn = I->poly[OP_KMIN] + j * 3;
          lin_clear(&a); lin_add(&a, ayv, 1);
          lin_clear(&b); b.k = kx[0] + M_KAPPA; lin_add(&b, vx, kx[1]); lin_add(&b, vy, kx[2]); lin_add(&b, axv, rho);
          lin_add(&b, ar, -M_KAPPA); lin_add(&b, rcc, M_KAPPA);
          emit(cb, ctx, &a, -1, &b, "A4.kappa_max");
          lin_clear(&b); b.k = kn[0] - M_KAPPA; lin_add(&b, vx, kn[1]); lin_add(&b, vy, kn[2]); lin_add(&b, axv, rho);
          lin_add(&b, ar, M_KAPPA); lin_add(&b, rcc, -M_KAPPA);
          emit(cb, ctx, &a, 1, &b, "A4.kappa_min");
        } else {
          lin_clear(&a); lin_add(&a, ar, 1); lin_clear(&b);
          emit(cb, ctx, &a, 0, &b, "A4.impossible");
        }
      }
      lin_clear(&a);
      for (int j = 0; j < R; ++j) lin_add(&a, VAR_AR(m, c, i, j), 1);
      lin_clear(&b); b.k = 1;
      /* emit() holds at most 24 terms; the sum row can be wider -> call the callback directly */
      {
        int* var = (int*)malloc(sizeof(int) * R); double* coef = (double*)malloc(sizeof(double) * R);
        for (int j = 0; j < R; ++j) { var[j] = VAR_AR(m, c, i, j); coef[j] = 1.0; }
        cb(ctx, R, var, coef, 0, 1.0, "A4.sum");
        free(var); free(coef);
      }
    }
  /* ---- A5 minimum_speed_constraints.mod:9-49 (replicated for every j) ---- */
  for (int i = 1; i < N; ++i)
    for (int c = 0; c < C; ++c) {
      int vx = VC(m, V_VX, c, i), vy = VC(m, V_VY, c, i);
      int xp = VAR_RC(m, 0, c, i), yp = VAR_RC(m, 1, c, i), xn = VAR_RC(m, 2, c, i), yn = VAR_RC(m, 3, c, i),
          cb_ = VAR_RC(m, 4, c, i);
      for (int j = 0; j < R; ++j) {
        for (int ax = 0; ax < 2; ++ax) {
          int v = ax ? vy : vx, p = ax ? yp : xp, n_ = ax ? yn : xn;
          lin_clear(&a); lin_add(&a, v, 1); a.k = -I->vm; lin_clear(&b); lin_add(&b, p, -M_VEL);
          emit(cb, ctx, &a, 1, &b, "A5.pos_lo");
          lin_clear(&b); b.k = M_VEL; lin_add(&b, p, -M_VEL);
          emit(cb, ctx, &a, -1, &b, "A5.pos_hi");
          lin_clear(&a); lin_add(&a, v, -1); a.k = -I->vm; lin_clear(&b); b.k = M_VEL; lin_add(&b, n_, -M_VEL);
          emit(cb, ctx, &a, -1, &b, "A5.neg_hi");
          lin_clear(&b); lin_add(&b, n_, -M_VEL);
          emit(cb, ctx, &a, 1, &b, "A5.neg_lo");
        }
        lin_clear(&a); lin_add(&a, VAR_AR(m, c, i, j), 1); lin_add(&a, VAR_AR(m, c, i - 1, j), -1);
        lin_clear(&b); b.k = 1; lin_add(&b, cb_, -1);
        emit(cb, ctx, &a, -1, &b, "A5.freeze_hi");
        lin_clear(&b); b.k = -1; lin_add(&b, cb_, 1);
        emit(cb, ctx, &a, 1, &b, "A5.freeze_lo");
        const int others[4] = {xp, yp, xn, yn};
        for (int k = 0; k < 4; ++k) {
          lin_clear(&a); lin_add(&a, cb_, 1); lin_clear(&b); lin_add(&b, others[k], 1);
          emit(cb, ctx, &a, -1, &b, "A5.and_ub");
        }
        lin_clear(&a); lin_add(&a, cb_, 1);
        lin_clear(&b); b.k = -3; for (int k = 0; k < 4; ++k) lin_add(&b, others[k], 1);
        emit(cb, ctx, &a, 1, &b, "A5.and_lb");
      }
    }
  /* ---- A6 environment obstacle_environment_constraints.mod:6-47 ---- */
  if (E > 0) {
    for (int i = 0; i < N; ++i)
      for (int c = 0; c < C; ++c) {
        const int PXs[5] = {V_PX, V_XFU, V_XFL, V_XFU, V_XFL};
        const int PYs[5] = {V_PY, V_YFU, V_YFU, V_YFL, V_YFL};
        for (int e = 0; e < E; ++e)
          for (int k = I->env_off[e]; k < I->env_off[e + 1]; ++k)
            for (int p = 0; p < 5; ++p) {
              cross_lin(&a, I->env_edges + 4 * k, VC(m, PXs[p], c, i), VC(m, PYs[p], c, i));
              lin_clear(&b); lin_add(&b, VAR_NW(m, p, c, e, i), -M_ENV);
              emit(cb, ctx, &a, 1, &b, "A6.edge");
            }
        for (int p = 0; p < 5; ++p) {
          int* var = (int*)malloc(sizeof(int) * E); double* coef = (double*)malloc(sizeof(double) * E);
          for (int e = 0; e < E; ++e) { var[e] = VAR_NW(m, p, c, e, i); coef[e] = 1.0; }
          cb(ctx, E, var, coef, -1, (double)(E - 1), "A6.card");
          free(var); free(coef);
        }
      }
  }
  /* ---- A7 obstacles :52-109 ---- */
  if (O > 0) {
    for (int i = 0; i < N; ++i)
      for (int c = 0; c < C; ++c)
        for (int o = 0; o < O; ++o) {
          /* deltacc: rear ; front 1..4 = (LB,LB) (UB,LB) (LB,UB) (UB,UB) as (x,y) */
          const int PXs[5] = {V_PX, V_XFL, V_XFU, V_XFL, V_XFU};
          const int PYs[5] = {V_PY, V_YFL, V_YFL, V_YFU, V_YFU};
          for (int k = 0; k < L; ++k)
            for (int p = 0; p < 5; ++p) {
              cross_lin(&a, I->obs_edges + ((size_t)(o * N + i) * L + k) * 4, VC(m, PXs[p], c, i), VC(m, PYs[p], c, i));
              lin_clear(&b);
              lin_add(&b, p == 0 ? VAR_D(m, c, o, i, k) : VAR_DF(m, c, o, i, k, p - 1), M_OBS);
              emit(cb, ctx, &a, -1, &b, "A7.edge");
            }
          for (int p = 0; p < 5; ++p) {
            lin_clear(&a);
            for (int k = 0; k < L; ++k) lin_add(&a, p == 0 ? VAR_D(m, c, o, i, k) : VAR_DF(m, c, o, i, k, p - 1), 1);
            if (I->obs_soft[o] == 1) lin_add(&a, p == 0 ? VAR_SO(m, c, o, i) : VAR_SOF(m, c, o, i, p - 1), -1);
            lin_clear(&b); b.k = L - 1;
            emit(cb, ctx, &a, -1, &b, "A7.card");
          }
        }
  }
  /* ---- A8 agent_collision_constraints.mod:10-73 ---- */
  if (C > 1) {
    for (int i = 0; i < N; ++i)
      for (int c1 = 1; c1 < K; ++c1)
        for (int c2 = 0; c2 < c1; ++c2) {
          for (int s = 0; s < 4; ++s) {
            lin_clear(&a); lin_add(&a, VAR_S(m, c1, c2, i, s), 1); lin_clear(&b);
            emit(cb, ctx, &a, 0, &b, "A8.zero_s");
          }
          for (int s = 0; s < 16; ++s) {
            lin_clear(&a); lin_add(&a, VAR_CC(m, c1, c2, i, s), 1); lin_clear(&b);
            emit(cb, ctx, &a, 0, &b, "A8.zero_cc");
          }
        }
    for (int i = 0; i < N; ++i)
      for (int c1 = 0; c1 < C - 1; ++c1)
        for (int c2 = c1 + 1; c2 < C; ++c2) {
          double D = I->rad[c1] + I->rad[c2] + I->safety[i], S = I->safety_slack[i];
          int q2 = c2 - 1;
#define CCV(s) VAR_CC(m, c1, q2, i, (s))
#define SLV(s) VAR_S(m, c1, q2, i, (s))
          /* generic helper rows: lhsvar <= rhsvar - (D [+S - slack]) + M*b   (sense -1)
           *                      lhsvar >= rhsvar + (D [+S - slack]) - M*b   (sense +1) */
          struct { int lv, rv, sense, b, sl; } rows[16] = {
              {VC(m, V_PX, c1, i), VC(m, V_PX, c2, i), -1, 0, 0},  {VC(m, V_PX, c1, i), VC(m, V_PX, c2, i), 1, 1, 0},
              {VC(m, V_PY, c1, i), VC(m, V_PY, c2, i), -1, 2, 1},  {VC(m, V_PY, c1, i), VC(m, V_PY, c2, i), 1, 3, 1},
              {VC(m, V_PX, c1, i), VC(m, V_XFL, c2, i), -1, 4, -1}, {VC(m, V_PX, c1, i), VC(m, V_XFU, c2, i), 1, 5, -1},
              {VC(m, V_PY, c1, i), VC(m, V_YFL, c2, i), -1, 6, -1}, {VC(m, V_PY, c1, i), VC(m, V_YFU, c2, i), 1, 7, -1},
              {VC(m, V_PX, c2, i), VC(m, V_XFL, c1, i), -1, 8, -1}, {VC(m, V_PX, c2, i), VC(m, V_XFU, c1, i), 1, 9, -1},
              {VC(m, V_PY, c2, i), VC(m, V_YFL, c1, i), -1, 10, -1}, {VC(m, V_PY, c2, i), VC(m, V_YFU, c1, i), 1, 11, -1},
              /* front/front :58-61  0 <= xfL1 - (..) - xfU2 + M b13 ; 0 >= xfU1 + (..) - xfL2 - M b14 */
              {VC(m, V_XFU, c2, i), VC(m, V_XFL, c1, i), -1, 12, 2}, {VC(m, V_XFL, c2, i), VC(m, V_XFU, c1, i), 1, 13, 2},
              {VC(m, V_YFU, c2, i), VC(m, V_YFL, c1, i), -1, 14, 3}, {VC(m, V_YFL, c2, i), VC(m, V_YFU, c1, i), 1, 15, 3}};
          for (int g = 0; g < 4; ++g) {
            for (int q = 0; q < 4; ++q) {
              int r = 4 * g + q;
              lin_clear(&a); lin_add(&a, rows[r].lv, 1);
              lin_clear(&b); lin_add(&b, rows[r].rv, 1);
              double sg = rows[r].sense < 0 ? -1.0 : 1.0;
              b.k = sg * (D + (rows[r].sl >= 0 ? S : 0.0));
              if (rows[r].sl >= 0) lin_add(&b, SLV(rows[r].sl), -sg);
              lin_add(&b, CCV(rows[r].b), -sg * M_AGENTS);
              emit(cb, ctx, &a, rows[r].sense, &b, "A8.sep");
            }
            lin_clear(&a); a.k = 3; lin_clear(&b);
            for (int q = 0; q < 4; ++q) lin_add(&b, CCV(4 * g + q), 1);
            emit(cb, ctx, &a, 1, &b, "A8.card");
            if (g == 0 || g == 3) {
              for (int q = 0; q < 2; ++q) {
                lin_clear(&a); lin_add(&a, SLV((g == 0 ? 0 : 2) + q), 1); lin_clear(&b); b.k = S;
                emit(cb, ctx, &a, -1, &b, "A8.slack_ub");
              }
            }
          }
#undef CCV
#undef SLV
        }
  }
}

/* ------------------------------------------------------------------ sizes */
typedef struct { int rows, nnz; unsigned char* seen; } count_ctx;
static void count_cb(void* ctx, int n, const int* var, const double* coef, int sense, double rhs, const char* tag) {
  count_ctx* c = (count_ctx*)ctx; (void)coef; (void)sense; (void)rhs; (void)tag;
  c->rows++; c->nnz += n;
  for (int q = 0; q < n; ++q) c->seen[var[q]] = 1;
}

orc_sizes orc_raw_sizes(const oinst* I) {
  vmap m; vmap_init(&m, I);
  count_ctx c = {0, 0, (unsigned char*)calloc((size_t)m.nvar + 1, 1)};
  enumerate_rows(I, &m, count_cb, &c);
  /* variables that only appear in the objective are extracted too (objective_function.mod:7-19) */
  for (int v = m.o_sO; v < m.ncont; ++v) c.seen[v] = 1;
  for (int k = 0; k < 8; ++k)
    for (int q = 0; q < I->C * I->N; ++q) c.seen[m.o_cont[k] + q] = 1;
  orc_sizes s = {c.rows, 0, 0, c.nnz};
  for (int v = 0; v < m.nvar; ++v)
    if (c.seen[v]) { if (v < m.ncont) s.cont++; else s.bin++; }
  free(c.seen);
  return s;
}

/* ------------------------------------------------------------------ evaluation */
typedef struct { const double* x; double worst; char tag[64]; int row; int worst_row; } eval_ctx;
static void eval_cb(void* ctx, int n, const int* var, const double* coef, int sense, double rhs, const char* tag) {
  eval_ctx* e = (eval_ctx*)ctx;
  double lhs = 0;
  for (int q = 0; q < n; ++q) lhs += coef[q] * e->x[var[q]];
  double v = sense < 0 ? lhs - rhs : (sense > 0 ? rhs - lhs : fabs(lhs - rhs));
  if (v > e->worst) { e->worst = v; snprintf(e->tag, sizeof(e->tag), "%s", tag); e->worst_row = e->row; }
  e->row++;
}

double orc_raw_eval(const oinst* I, const miqp_raw_results_c* r, const double* slack_real, double* objective,
                    char* worst, int worstlen) {
  vmap m; vmap_init(&m, I);
  int C = I->C, N = I->N, R = I->R, E = I->E, O = I->O, L = I->L, K = I->K;
  double* x = (double*)calloc((size_t)m.nvar + 1, sizeof(double));
  const double* cont[12] = {r->u_x, r->u_y, r->pos_x, r->vel_x, r->acc_x, r->pos_y, r->vel_y, r->acc_y,
                            r->pos_x_front_UB, r->pos_x_front_LB, r->pos_y_front_UB, r->pos_y_front_LB};
  for (int k = 0; k < 12; ++k)
    for (int q = 0; q < C * N; ++q) x[m.o_cont[k] + q] = cont[k][q];
  const int* nw[5] = {r->notWithinEnvironmentRear, r->notWithinEnvironmentFrontUbUb, r->notWithinEnvironmentFrontLbUb,
                      r->notWithinEnvironmentFrontUbLb, r->notWithinEnvironmentFrontLbLb};
  for (int k = 0; k < 5; ++k)
    for (int q = 0; q < C * E * N; ++q) x[m.o_nw[k] + q] = nw[k][q];
  for (int q = 0; q < C * N * R; ++q) x[m.o_ar + q] = r->active_region[q];
  const int* rc[5] = {r->region_change_not_allowed_x_positive, r->region_change_not_allowed_y_positive,
                      r->region_change_not_allowed_x_negative, r->region_change_not_allowed_y_negative,
                      r->region_change_not_allowed_combined};
  for (int k = 0; k < 5; ++k)
    for (int q = 0; q < C * N; ++q) x[m.o_rc[k] + q] = rc[k][q];
  for (int q = 0; q < C * O * N * L; ++q) x[m.o_d + q] = r->deltacc[q];
  for (int q = 0; q < C * O * N * L * 4; ++q) x[m.o_df + q] = r->deltacc_front[q];
  for (int q = 0; q < K * K * N * 16; ++q) x[m.o_cc + q] = r->car2car_collision[q];
  for (int q = 0; q < K * K * N * 4; ++q) x[m.o_s + q] = slack_real ? slack_real[q] : (double)r->slackvars[q];
  for (int q = 0; q < C * O * N; ++q) x[m.o_sO + q] = r->slackvarsObstacle[q];
  for (int q = 0; q < C * O * N * 4; ++q) x[m.o_sOf + q] = r->slackvarsObstacle_front[q];
  eval_ctx e = {x, 0.0, "", 0, -1};
  enumerate_rows(I, &m, eval_cb, &e);
  if (objective) {
    double obj = 0;
    for (int i = 0; i < N; ++i)
      for (int c = 0; c < C; ++c) {
        const double* W = I->W + c * 8; const double* rf = I->ref + (c * N + i) * 6;
        int q = c * N + i;
        double dx = r->pos_x[q] - rf[0], dvx = r->vel_x[q] - rf[1], dy = r->pos_y[q] - rf[3], dvy = r->vel_y[q] - rf[4];
        obj += W[0] * dx * dx + W[1] * dvx * dvx + W[2] * r->acc_x[q] * r->acc_x[q] + W[3] * dy * dy + W[4] * dvy * dvy +
               W[5] * r->acc_y[q] * r->acc_y[q] + W[6] * r->u_x[q] * r->u_x[q] + W[7] * r->u_y[q] * r->u_y[q];
      }
    for (int q = 0; q < C * O * N; ++q) obj += I->w_slack_obs * x[m.o_sO + q] * x[m.o_sO + q];
    for (int q = 0; q < C * O * N * 4; ++q) obj += I->w_slack_obs * x[m.o_sOf + q] * x[m.o_sOf + q];
    for (int q = 0; q < K * K * N * 4; ++q) obj += I->w_slack * x[m.o_s + q] * x[m.o_s + q];
    *objective = obj;
  }
  if (worst) snprintf(worst, worstlen, "%s#%d", e.tag, e.worst_row);
  free(x);
  return e.worst;
}
