// miqp_gpu.hip - host side of libmiqp_gpu.so: C ABI (include/miqp_gpu.h), device buffer management and the
// round loop  select -> interior point -> evaluate/branch  of the on-device branch and bound.
//
// There is no CPU solve path in this library: without a HIP device the solve entry points return
// MIQP_STATUS_FAILED_SEG_FAULT (the reference's code for "the solver could not run", cplex_wrapper.cpp:97-109).
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <atomic>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/miqp_gpu.h"
#include "kernels.hip"
#include "wire_formats.hpp"
#include "planner_core.hpp"
#include "lp_export.hpp"

// A round of two cars runs FOUR launches side by side on four streams (launch_ipm_batch), and the process has its null stream: with the HIP runtime's
// default of four hardware queues two of those streams share one, and the launch behind on the shared queue - the memory-backed kernel - started only
// when the larger active-set launch in front of it had ended, alone, for 1.7 ms of a 12 ms round (rocprofv3 trace of the driver's command,
// profiles/r06_*round_gaps*).  The runtime reads GPU_MAX_HW_QUEUES once, at its first call: the library asks for eight when it is loaded, unless the
// variable is set already; a process that initialised HIP before loading the library keeps its four queues (and that serial tail) - INTEGRATION.md 5.
__attribute__((constructor)) static void miqp_gpu_hw_queues() { setenv("GPU_MAX_HW_QUEUES", "8", 0); }

using namespace miqp;

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "[miqp_gpu] %s failed: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); return false; } } while (0)

// caller-independent RawResults record (deep copy target; MIP starts are kept in this form until the solve compiles them)
struct OwnedResults {
  miqp_raw_results_c r{}; std::vector<std::vector<double>> d; std::vector<std::vector<int>> i;
  OwnedResults(int C, int N, int R, int E, int O, int L) {
    const int K = C - 1;
    r.N = N; r.NrEnvironments = E; r.NrRegions = R; r.NrObstacles = O; r.MaxLinesObstacles = L; r.NrCarToCarCollisions = K; r.NrCars = C;
    auto D = [&](size_t n) { d.emplace_back(n ? n : 1, 9999999.0); return d.back().data(); };
    auto J = [&](size_t n) { i.emplace_back(n ? n : 1, 9999999); return i.back().data(); };
    d.reserve(16); i.reserve(24);
    const size_t cn = (size_t)C * N;
    r.u_x = D(cn); r.u_y = D(cn); r.pos_x = D(cn); r.vel_x = D(cn); r.acc_x = D(cn); r.pos_y = D(cn); r.vel_y = D(cn); r.acc_y = D(cn);
    r.pos_x_front_UB = D(cn); r.pos_x_front_LB = D(cn); r.pos_y_front_UB = D(cn); r.pos_y_front_LB = D(cn);
    r.notWithinEnvironmentRear = J(cn * E); r.notWithinEnvironmentFrontUbUb = J(cn * E); r.notWithinEnvironmentFrontLbUb = J(cn * E);
    r.notWithinEnvironmentFrontUbLb = J(cn * E); r.notWithinEnvironmentFrontLbLb = J(cn * E); r.active_region = J(cn * R);
    r.region_change_not_allowed_x_positive = J(cn); r.region_change_not_allowed_y_positive = J(cn); r.region_change_not_allowed_x_negative = J(cn);
    r.region_change_not_allowed_y_negative = J(cn); r.region_change_not_allowed_combined = J(cn);
    r.deltacc = J(cn * O * L); r.deltacc_front = J(cn * O * L * 4); r.car2car_collision = J((size_t)K * K * N * 16); r.slackvars = J((size_t)K * K * N * 4);
    r.slackvarsObstacle = J(cn * O); r.slackvarsObstacle_front = J(cn * O * 4); r.slackvars_real = D((size_t)K * K * N * 4);
  }
  explicit OwnedResults(const HostInst& I) : OwnedResults(I.C, I.N, I.R, I.E, I.O, I.L) {}
  // deep copy of a caller's record of the same seven sizes (null arrays keep the fill value)
  void copy_from(const miqp_raw_results_c& f) {
    const double* const sd[13] = {f.u_x, f.u_y, f.pos_x, f.vel_x, f.acc_x, f.pos_y, f.vel_y, f.acc_y, f.pos_x_front_UB, f.pos_x_front_LB, f.pos_y_front_UB, f.pos_y_front_LB, f.slackvars_real};
    for (int k = 0; k < 13; ++k) if (sd[k]) std::copy(sd[k], sd[k] + d[k].size(), d[k].begin());
    const int* const si[17] = {f.notWithinEnvironmentRear, f.notWithinEnvironmentFrontUbUb, f.notWithinEnvironmentFrontLbUb, f.notWithinEnvironmentFrontUbLb, f.notWithinEnvironmentFrontLbLb,
                               f.active_region, f.region_change_not_allowed_x_positive, f.region_change_not_allowed_y_positive, f.region_change_not_allowed_x_negative,
                               f.region_change_not_allowed_y_negative, f.region_change_not_allowed_combined, f.deltacc, f.deltacc_front, f.car2car_collision, f.slackvars,
                               f.slackvarsObstacle, f.slackvarsObstacle_front};
    for (int k = 0; k < 17; ++k) if (si[k]) std::copy(si[k], si[k] + i[k].size(), i[k].begin());
  }
  // ... and the other way: this record into a caller's record of the same seven sizes (null arrays are skipped)
  void copy_to(miqp_raw_results_c& f) const {
    double* const sd[13] = {f.u_x, f.u_y, f.pos_x, f.vel_x, f.acc_x, f.pos_y, f.vel_y, f.acc_y, f.pos_x_front_UB, f.pos_x_front_LB, f.pos_y_front_UB, f.pos_y_front_LB, f.slackvars_real};
    for (int k = 0; k < 13; ++k) if (sd[k]) std::copy(d[k].begin(), d[k].end(), sd[k]);
    int* const si[17] = {f.notWithinEnvironmentRear, f.notWithinEnvironmentFrontUbUb, f.notWithinEnvironmentFrontLbUb, f.notWithinEnvironmentFrontUbLb, f.notWithinEnvironmentFrontLbLb,
                         f.active_region, f.region_change_not_allowed_x_positive, f.region_change_not_allowed_y_positive, f.region_change_not_allowed_x_negative,
                         f.region_change_not_allowed_y_negative, f.region_change_not_allowed_combined, f.deltacc, f.deltacc_front, f.car2car_collision, f.slackvars,
                         f.slackvarsObstacle, f.slackvarsObstacle_front};
    for (int k = 0; k < 17; ++k) if (si[k]) std::copy(i[k].begin(), i[k].end(), si[k]);
  }
};

// the seven sizes of a RawResults record against the loaded instance (a record of another shape is never indexed)
static bool dims_match(const miqp_raw_results_c& r, const HostInst& I) {
  return r.N == I.N && r.NrCars == I.C && r.NrRegions == I.R && r.NrEnvironments == I.E && r.NrObstacles == I.O &&
         (I.O == 0 || r.MaxLinesObstacles == I.L) && r.NrCarToCarCollisions == I.C - 1;
}

struct miqp_solver {
  miqp_solver_opts opts{};
  HostInst inst; bool has_inst = false;
  // last solution
  int status = MIQP_STATUS_FAILED_NO_SOLUT;
  miqp_solution_properties_c props{};
  std::vector<double> Z; std::vector<signed char> comp; bool has_sol = false;
  Layout lay{};
  double timing[6] = {0, 0, 0, 0, 0, 0};
  double as_timing[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // active-set launches of the last call (miqp_solver_last_active_set)
  double setup[3] = {0, 0, 0};   // host set-up of the last call: seconds, of which the device context, 1 when the context was (re)built
  double admit_s = 0.0;          // when the last batch / stream call admitted this instance, in seconds after the first round of that call started
  // MIP starts (each tried as an additional root: binaries fixed, QP solved, accepted as incumbent when feasible).
  // Slot 0: receding-horizon start (addRecedingHorizonWarmstart), slot 1: last-solution start (.mst file); with
  // BOTH_WARMSTART_STRATEGIES the reference applies both (src/cplex_wrapper.cpp:124-138)
  std::unique_ptr<OwnedResults> ws[2];
  // result record of the last solve, built by miqp_solver_materialize_results (host threads, inside a batch call's timing) and
  // handed out by miqp_solver_get_results; dropped whenever the solution or the instance changes
  std::unique_ptr<OwnedResults> rescache;
  std::string err;
};

// host copies of the corner tables (kernels.hip keeps them in constant memory)
static const int ENV_PT_H[5][2] = {{PT_R, PT_R}, {PT_U, PT_U}, {PT_L, PT_U}, {PT_U, PT_L}, {PT_L, PT_L}};
static const int OBS_PT_H[5][2] = {{PT_R, PT_R}, {PT_L, PT_L}, {PT_U, PT_L}, {PT_L, PT_U}, {PT_U, PT_U}};

namespace {

// ---------------------------------------------------------------- raw model sizes
// rows / binary / continuous columns / non-zeros of the model OPL generates from cplexmodel/*.mod for this instance
// (cplex.getNrows / getNbinVars / getNcols - bin / getNNZs, src/cplex_wrapper.cpp:679-690).  Counted per statement of
// the .mod files: every `==` fixing is a row, duplicated rows stay, structural zero coefficients are dropped.
void raw_sizes(const HostInst& I, int& rows, int& bin, int& cont, int& nnz_out) {
  long C = I.C, N = I.N, R = I.R, E = I.E, O = I.O, L = I.L, K = I.C - 1;
  bin = (int)(C * N * (5 * E + R + 5 + 5 * O * L) + K * K * N * 16);
  cont = (int)(C * N * (12 + 5 * O) + K * K * N * 4);
  auto nz = [](double v) { return v != 0.0 ? 1L : 0L; };
  long a4 = 0, nnz = 0;
  nnz += C * (17 + 9 * R);                      // A1 initial_conditions.mod:11-61
  nnz += 24 * (N - 1) * C;                      // A2 model_region_constraints.mod:11-19
  nnz += 12 * N * C;                            // A3 :22-39
  for (int c = 0; c < C; ++c) {                 // A4 :43-117
    long P = 0, per = R;                        // the sum row carries R entries
    for (int j = 0; j < R; ++j) {
      if (I.possible[c * R + j] != 1) { per += 1; continue; }
      P++;
      const double* F = &I.frac[j * 4];
      per += nz(F[0]) + nz(F[1]) + 2 + nz(F[2]) + nz(F[3]) + 2;
      const int pt[4] = {2, 3, 0, 1};
      for (int k = 0; k < 4; ++k) { const double* p = &I.poly[pt[k]][j * 3]; per += 2 * (3 + nz(I.wb[c] * p[1]) + nz(I.wb[c] * p[2])); }
      per += 16;                                // jerk and acceleration boxes: 8 rows of 2
      double rho = (F[1] + F[3]) / (F[0] + F[2]);
      const double* kx = &I.poly[4][j * 3]; const double* kn = &I.poly[5][j * 3];
      per += 3 + nz(kx[1]) + nz(kx[2]) + nz(rho) + 3 + nz(kn[1]) + nz(kn[2]) + nz(rho);
    }
    a4 += 20 * P + (R - P) + 1;
    nnz += (N - 1) * per;
  }
  nnz += 35 * R * (N - 1) * C;                  // A5 minimum_speed_constraints.mod:9-49
  long env_edges = E > 0 ? I.env_off[E] : 0;
  long r = C * (17 + 5 * R) + 6 * (N - 1) * C + 12 * N * C + (N - 1) * a4 + 15 * R * (N - 1) * C;
  if (E > 0) {                                  // A6 obstacle_environment_constraints.mod:6-47
    r += N * C * (5 * env_edges + 5);
    long per = 5 * E;
    for (long k = 0; k < env_edges; ++k) { const double* e = &I.env_edges[(size_t)k * 4]; per += 5 * (1 + nz(e[2] - e[0]) + nz(e[3] - e[1])); }
    nnz += N * C * per;
  }
  if (O > 0) {                                  // A7 :52-109
    r += N * C * O * (5 * L + 5);
    for (int o = 0; o < O; ++o) for (int i = 0; i < N; ++i) {
      long per = 5 * (L + (I.obs_soft[o] == 1 ? 1 : 0));
      for (int k = 0; k < L; ++k) { const double* e = &I.obs_edges[((size_t)(o * N + i) * L + k) * 4]; per += 5 * (1 + nz(e[2] - e[0]) + nz(e[3] - e[1])); }
      nnz += C * per;
    }
  }
  if (C > 1) {                                  // A8 agent_collision_constraints.mod:10-73
    long tri = 0; for (long c1 = 2; c1 <= K; ++c1) tri += c1 - 1;
    r += N * 20 * tri + 24 * N * C * (C - 1) / 2;
    nnz += N * 20 * tri + 76 * N * C * (C - 1) / 2;
  }
  rows = (int)r; nnz_out = (int)nnz;
}

// ---------------------------------------------------------------- host geometry helpers (results, step-1 presolve)
struct HostGeo {
  const HostInst& I; const Layout& Y; const double* D; const int* T;
  void point(int c, int i, int q /*possible idx*/, const double* z, int tx, int ty, double& X, double& Yc) const {
    if (i == 0) {
      X = I.x0[c * 6 + 0]; Yc = I.x0[c * 6 + 3];
      if (tx != PT_R) X = D[Y.d_theta + c * 4 + 0];
      if (ty != PT_R) Yc = D[Y.d_theta + c * 4 + 1];
      return;
    }
    const double* rt = D + Y.d_reg + (c * Y.P + q) * REGSZ;
    double vx = z[6 * c + 1], vy = z[6 * c + 4];
    X = z[6 * c + 0]; Yc = z[6 * c + 3];
    if (tx != PT_R) { const double* p = rt + 19 + (tx == PT_U ? 0 : 3); X += p[0] + p[1] * vx + p[2] * vy; }
    if (ty != PT_R) { const double* p = rt + 25 + (ty == PT_U ? 0 : 3); Yc += p[0] + p[1] * vx + p[2] * vy; }
  }
  double env_viol(int e, double X, double Yc) const {
    double v = -1e300; for (int k = 0; k < T[Y.i_envn + e]; ++k) { const double* ed = D + Y.d_env + (e * Y.EL + k) * 3; v = std::max(v, ed[0] * X + ed[1] * Yc - ed[2]); }
    return v;
  }
  double obs_viol(int o, int i, int k, double X, double Yc) const { const double* ed = D + Y.d_obs + ((o * Y.N + i) * Y.L + k) * 3; return ed[0] * X + ed[1] * Yc - ed[2]; }
  // zero-slack violation of (pair, step, group, alt): lhs + sep
  double c2c_viol(int p, int c1, int c2, int i, int grp, int alt, int q1, int q2, const double* z) const {
    double Dsep = D[Y.d_dsep + p * Y.N + i], S = D[Y.d_ssl + i];
    bool isx = alt < 2, lo = (alt == 0 || alt == 2), soft = (grp == 0 || grp == 3);
    int ca, cb, ta, tb;
    if (grp == 0) { ta = tb = PT_R; ca = lo ? c1 : c2; cb = lo ? c2 : c1; }
    else if (grp == 1) { if (lo) { ca = c1; ta = PT_R; cb = c2; tb = PT_L; } else { ca = c2; ta = PT_U; cb = c1; tb = PT_R; } }
    else if (grp == 2) { if (lo) { ca = c2; ta = PT_R; cb = c1; tb = PT_L; } else { ca = c1; ta = PT_U; cb = c2; tb = PT_R; } }
    else { if (lo) { ca = c2; ta = PT_U; cb = c1; tb = PT_L; } else { ca = c1; ta = PT_U; cb = c2; tb = PT_L; } }
    double XA, YA, XB, YB;
    point(ca, i, ca == c1 ? q1 : q2, z, ta, ta, XA, YA); point(cb, i, cb == c1 ? q1 : q2, z, tb, tb, XB, YB);
    return (isx ? XA - XB : YA - YB) + (soft ? Dsep + S : Dsep);
  }
};

// step 1 of the horizon is fixed by initial_conditions.mod:11-24 -> every row of it is a constant
bool step0_check(const HostGeo& G, double& cobj) {
  const HostInst& I = G.I; const Layout& Y = G.Y; double tol = FEAS_TOL; cobj = 0;
  for (int c = 0; c < I.C; ++c) {
    if (I.E >= 1)
      for (int pt = 0; pt < 5; ++pt) {
        double X, Yc; G.point(c, 0, 0, nullptr, ENV_PT_H[pt][0], ENV_PT_H[pt][1], X, Yc);
        bool any = false; for (int e = 0; e < I.E && !any; ++e) any = G.env_viol(e, X, Yc) <= tol;
        if (!any) return false;
      }
    for (int o = 0; o < I.O; ++o)
      for (int pt = 0; pt < 5; ++pt) {
        double X, Yc; G.point(c, 0, 0, nullptr, OBS_PT_H[pt][0], OBS_PT_H[pt][1], X, Yc);
        bool any = false; for (int k = 0; k < I.L && !any; ++k) any = G.obs_viol(o, 0, k, X, Yc) <= tol;
        if (!any) { if (I.obs_soft[o]) cobj += I.w_slack_obs; else return false; }
      }
  }
  int p = 0;
  for (int c1 = 0; c1 < I.C; ++c1)
    for (int c2 = c1 + 1; c2 < I.C; ++c2, ++p)
      for (int g = 0; g < 4; ++g) {
        double best = 1e300; double smax = G.D[Y.d_smax + 0];
        for (int a = 0; a < 4; ++a) {
          double v = G.c2c_viol(p, c1, c2, 0, g, a, 0, 0, nullptr);
          bool soft = (g == 0 || g == 3);
          if (v - (soft ? smax : 0.0) > tol) continue;
          double need = soft ? std::max(0.0, v) : 0.0;
          best = std::min(best, I.w_slack * need * need);
        }
        if (best > 1e299) return false;
        cobj += best;
      }
  return true;
}

}  // namespace


// ================================================================================================
//  device context (buffers are cached between calls with the same shape)
// ================================================================================================
namespace {

struct DevCtx {
  std::mutex mu;   // held for the whole solve: one solve at a time per device, different devices run concurrently
  bool ready = false; int device = -1;
  hipStream_t stream = nullptr;
  Layout Y{}; int n_inst = 0, n_slots = 0, open_cap = 0, far_cap = 0, batch_cap = 0, batch_alloc = 0, pool_cap = 0, npr = 0, ipm_grid_max = 1024;
  int n_inst_cap = 0;      // instances the per-instance arrays hold: a call with fewer reuses the context as it is
  int open_cap_req = 0;    // near-list capacity the caller asked for (open_cap: what the free memory allowed when the context was built)
  int mem_div = 1;         // contexts that share the device (lanes of one stream call): each sizes its pools for its share of the free memory
  int lane = 0;
  int* d_pairs = nullptr;   // admissions of one round: (slot, instance) pairs
  // concurrent launch of the memory-backed kernel on the rounding probes of a batch (second stream, its own work counter and
  // per-block buffers): see launch_ipm_batch
  hipStream_t stream2 = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_sfork = nullptr, ev_sjoin = nullptr;
  hipStream_t stream3 = nullptr, stream4 = nullptr; hipEvent_t ev_join3 = nullptr, ev_join4 = nullptr; double* kgain3 = nullptr;   // the larger active-set launch beside the interior point chain of stream2
  hipEvent_t ev_mid = nullptr;   // MIQP_LAUNCH_TRACE
  int* work_counter2 = nullptr; double* rowstate2 = nullptr; double* rowcache2 = nullptr; double* kgain2 = nullptr; int probe_grid = 0;
  int oc_grid = 0;   // resident wavefronts of the on-chip interior point kernel (0: the shape does not qualify)
  int kg_blocks = 0; // blocks the gain buffer holds
  int* ctr = nullptr; // two parity sets of 8 counters for the launches of a round (batch count, work counters, hand-over counts): a round uses one set, roll_kernel zeroes the other
  bool concurrent_big = true;   // MIQP_CONCURRENT_BIG=0: the sequential chain standard -> larger -> memory-backed
  int ocb_grid = 0;  // resident wavefronts of its larger variant (OC_GCAP_BIG general rows, one wavefront per SIMD; 0: not in use)
  bool as_on = false, as_cap = false; unsigned long long* as_stats = nullptr;
  int* cls_list = nullptr; int cls_n[3] = {-1, -1, -1};   // the round's class lists and (read back with the batch count) their lengths; -1: not known, the launches scan the batch
  unsigned short* as_batch_A = nullptr; unsigned short* as_pool_A = nullptr;   // (kept here: a call with MIQP_AS=0 runs with the DevBuf pointers nulled)   // dual active-set launch in front of the standard interior point launch (two cars; MIQP_AS=0: off)
  DevBuf B{};
  std::vector<void*> allocs;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::vector<hipEvent_t> ipm_ev;  // pairs
  std::vector<hipEvent_t> std_ev;  // per round: the end of the standard launch on the solver stream
  template <class Tp> bool alloc(Tp** p, size_t n) {
    void* q = nullptr;
    if (hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(Tp)) != hipSuccess) { std::fprintf(stderr, "[miqp_gpu] hipMalloc of %zu bytes failed\n", n * sizeof(Tp)); return false; }
    allocs.push_back(q); *p = (Tp*)q; return true;
  }
  void release() {
    for (void* q : allocs) (void)hipFree(q);
    allocs.clear();
    for (auto e : ipm_ev) (void)hipEventDestroy(e);
    ipm_ev.clear();
    for (auto e : std_ev) (void)hipEventDestroy(e);
    std_ev.clear();
    ready = false;
  }
};

// one context per HIP device (the reference has one IloEnv per wrapper; here the device buffers are shared by the
// handles that solve on the same device and serialised by the context's mutex)
std::mutex g_ctx_table_mu;
std::map<int, std::unique_ptr<DevCtx>> g_ctx_table;

// resolves `device` (-1: the current device) and returns its context; nullptr without a HIP device
DevCtx* ctx_for_device(int device, int lane = 0) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { std::fprintf(stderr, "[miqp_gpu] no HIP device: the solver has no CPU path\n"); return nullptr; }
  if (device < 0) { if (hipGetDevice(&device) != hipSuccess) device = 0; }
  if (device >= ndev) { std::fprintf(stderr, "[miqp_gpu] device %d requested, %d visible\n", device, ndev); return nullptr; }
  std::lock_guard<std::mutex> lk(g_ctx_table_mu);
  auto& slot = g_ctx_table[device * 16 + (lane & 15)];   // lane 0: the context of the device; lanes 1..: the further contexts of a stream call that shares the device
  if (!slot) { slot.reset(new DevCtx()); slot->device = device; slot->lane = lane; }
  return slot.get();
}

size_t ipm_lds_bytes(const Layout& Y);

bool same_layout(const Layout& a, const Layout& b) { return std::memcmp(&a, &b, sizeof(Layout)) == 0; }

// caller holds X.mu; X.device is the resolved device ordinal
// n_inst instances in the call (the queue), n_slots of them in flight at a time (list storage per slot), roots_per_inst root records each
// `clamp_open`: the near-list capacity may be cut to what an eighth of the free memory holds (40 B per entry)
// A ready context of the same shape is reused as it is when the call has no more instances than its per-instance arrays hold
// (built for 64 queues' worth at least, within 4 GB): releasing and reallocating the pools - most of the device memory - costs
// seconds, and a service that drains one queue after the other must not pay them per call (measured: 4.3 s of a 17 s bench stream).
std::mutex g_ctx_build_mu;   // contexts are built one at a time: each sizes its pools from the memory that is free at that moment
bool ctx_prepare(DevCtx& X, const Layout& Y, int n_inst, int n_slots, int open_cap, int npr, int roots_per_inst, bool clamp_open = false, int mem_div = 1) {
  int batch_cap = n_slots * npr;
  HIP_OK(hipSetDevice(X.device));   // the calling thread's current device (threads of solve_batch_multi each set their own)
  if (X.ready && same_layout(X.Y, Y) && n_inst <= X.n_inst_cap && X.n_slots == n_slots && X.open_cap_req == open_cap && X.npr == npr && X.B.root_stride == roots_per_inst && X.mem_div == mem_div) {
    X.n_inst = n_inst; X.B.n_inst = n_inst;
    return true;
  }
  std::lock_guard<std::mutex> build_lock(g_ctx_build_mu);
  if (X.ready || !X.allocs.empty()) X.release();
  if (!X.stream) HIP_OK(hipStreamCreate(&X.stream));
  if (!X.ev0) { HIP_OK(hipEventCreate(&X.ev0)); HIP_OK(hipEventCreate(&X.ev1)); }
  X.open_cap_req = open_cap; X.mem_div = mem_div < 1 ? 1 : mem_div;
  if (clamp_open) { size_t fb = 0, tb = 0; if (hipMemGetInfo(&fb, &tb) == hipSuccess) { const size_t lim = fb / (size_t)X.mem_div / 8 / 40 / (size_t)n_slots; if ((size_t)open_cap > lim) open_cap = (int)std::max<size_t>(4096, lim); } }
  {   // capacity of the per-instance arrays (tables, incumbent records: dstride * 8 + istride * 4 + fix record + solution per instance)
    const size_t per = (size_t)Y.dstride * 8 + (size_t)Y.istride * 4 + (size_t)Y.fixlen * 2 + (size_t)Y.N * Y.nz * 16 + 256;
    size_t cap = std::min<size_t>((size_t)64 * (size_t)n_slots, ((size_t)4 << 30) / per);
    X.n_inst_cap = (int)std::max<size_t>((size_t)n_inst, cap);
  }
  const int n_call = n_inst;
  n_inst = X.n_inst_cap;   // everything below is sized for the capacity
  const int batch_alloc = std::max(batch_cap, n_inst);   // the final polish solves one node per instance in one launch
  X.Y = Y; X.n_inst = n_call; X.n_slots = n_slots; X.open_cap = open_cap; X.npr = npr; X.batch_cap = batch_cap; X.batch_alloc = batch_alloc;
  { hipDeviceProp_t pr; int dv = 0; (void)hipGetDevice(&dv); int cus = 256; if (hipGetDeviceProperties(&pr, dv) == hipSuccess) cus = pr.multiProcessorCount;
    // resident workgroups of the memory-backed kernel per CU: what its LDS admits, and its wavefronts per SIMD (three and four cars: workgroups of four wavefronts)
    size_t l = ipm_lds_bytes(Y); int per = (int)std::max<size_t>(1, std::min<size_t>(Y.C <= 2 ? 4 * MIQP_IPM_WPE : (4 * MIQP_WIDE_WPE) / (MIQP_WIDE_NT / 64), (160 * 1024) / std::max<size_t>(l + 8, 1))); X.ipm_grid_max = cus * per;
    // on-chip kernel: up to two cars, horizon within its register slots; 2 wavefronts per SIMD, as many as its LDS admits
    X.oc_grid = 0;
    const OcLds ol = oc_lds_layout(Y.N, Y.fixlen);
    const size_t bitmap_b = (size_t)((Y.N * Y.NSLOT + 63) / 64) * 10 + 16;   // decode bitmap + prefix inside the scratch region
    if (Y.C <= 2 && Y.N <= 2 * OC_NSL && bitmap_b + 1024 <= (size_t)(ol.r - ol.u) && !KNOB_T("MIQP_IPM_V1")) {
      size_t lo = (size_t)ol.total + 16;
      int perc = (int)std::min<size_t>(8, (160 * 1024) / lo);
      if (KNOB_T("MIQP_OC_WAVES")) perc = std::max(1, std::min(perc, std::atoi(KNOB_T("MIQP_OC_WAVES"))));   // (experiment: resident wavefronts of the on-chip kernel per CU)
      if (perc >= 1) X.oc_grid = cus * perc;
      X.ocb_grid = 0; X.concurrent_big = !(KNOB_T("MIQP_CONCURRENT_BIG") && std::atoi(KNOB_T("MIQP_CONCURRENT_BIG")) == 0);
      if (X.oc_grid > 0 && !(KNOB_T("MIQP_OC_BIG") && std::atoi(KNOB_T("MIQP_OC_BIG")) == 0)) {
        const size_t lb = (size_t)oc_lds_layout(Y.N, Y.fixlen, OC_GCAP_BIG).total + 16;
        const int pb = (int)std::min<size_t>(4, (160 * 1024) / lb);
        if (pb >= 1) X.ocb_grid = cus * pb;
      }
    }
    if (X.oc_grid > X.ipm_grid_max) X.ipm_grid_max = X.oc_grid;   // the per-block buffers are sized for the larger grid
    // Oversubscribed launches of the on-chip kernels (MIQP_OC_OVERSUB = F): F times the resident wavefronts, each working through 1 / F of the
    // nodes - a wavefront slot comes free F times as often, so the short kernels of ANOTHER lane of the same device (MIQP_LANES: its
    // evaluation and selection) are dispatched between them instead of waiting for the whole launch.  Only the gain buffer grows.
    X.kg_blocks = X.ipm_grid_max;
    if (const char* e = KNOB_T("MIQP_OC_OVERSUB")) { const int f = std::max(1, std::min(64, std::atoi(e)));
      X.oc_grid = std::min(batch_alloc, X.oc_grid * f); X.ocb_grid = std::min(batch_alloc, X.ocb_grid * f); X.kg_blocks = std::max(X.kg_blocks, std::max(X.oc_grid, X.ocb_grid)); }
  }
  // node pool: live nodes are bounded by the open lists plus one round of children; processed records are recycled
  size_t free_b = 0, total_b = 0; if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = (size_t)16 << 30;
  free_b /= (size_t)X.mem_div;
  // far tier of the open lists (16 B per entry): up to 2^24 entries per instance within a tenth of the free memory
  // (256 instances: 4 M entries each, 16 GB); MIQP_FAR_CAP overrides, 0 switches the tier off
  { size_t fc = std::min<size_t>((size_t)1 << 24, free_b / 10 / 16 / (size_t)n_slots);
    if (const char* e = KNOB_P("MIQP_FAR_CAP")) fc = (size_t)std::max(0LL, std::atoll(e));
    X.far_cap = fc < 4096 ? 0 : (int)fc; }
  size_t want = (size_t)n_slots * ((size_t)open_cap + (size_t)X.far_cap + (size_t)npr * 64 + 64) + (size_t)n_inst * roots_per_inst;
  const bool ws_on = !(KNOB_T("MIQP_WARM") && std::atoi(KNOB_T("MIQP_WARM")) == 0);   // warm start of the node relaxations (MIQP_WARM=0: cold)
  size_t maxrec = std::min<size_t>((size_t)64 << 30, free_b / 4) / (size_t)Y.fixlen;   // node records: up to 64 GB of the 288 GB, at most a quarter of what is free
  X.pool_cap = (int)std::min<size_t>(std::min(want, maxrec), (size_t)0x7FFFFFF0);
  DevBuf& B = X.B; std::memset(&B, 0, sizeof(B));
  B.qp_tol = QP_TOL; B.use_cutoff = 1;
  B.cut_gate = KNOB_P("MIQP_CUT_GATE") ? std::atof(KNOB_P("MIQP_CUT_GATE")) : 1.0e-5;
  B.opt2 = KNOB_T("MIQP_OPT2") ? std::atoi(KNOB_T("MIQP_OPT2")) : (8 << 4);   // rounding probe at nodes with at most 8 violated sites (eval_kernel)
  B.seq_kinds = KNOB_P("MIQP_SEQ_KINDS") ? (int)std::strtoul(KNOB_P("MIQP_SEQ_KINDS"), nullptr, 0) : (5 << 8);   // plain K-way children, branching order 5 (see eval_kernel)
  B.Y = Y; B.pool_cap = X.pool_cap; B.open_cap = open_cap; B.batch_cap = batch_cap; B.nodes_per_round = npr; B.n_inst = n_call; B.n_slots = n_slots; B.root_stride = roots_per_inst;
  double* dd; int* ii;
  if (!X.alloc(&dd, (size_t)n_inst * Y.dstride)) return false; B.inst_d = dd;
  if (!X.alloc(&ii, (size_t)n_inst * Y.istride)) return false; B.inst_i = ii;
  if (!X.alloc(&B.pool_fix, (size_t)X.pool_cap * Y.fixlen)) return false;
  // the parents' solutions for the warm starts: for the first z_cap records (recycled records keep the live set at low
  // indices) - 16384 per instance in flight, at least 2 M, within an eighth of the free memory (42 GB at 1024 in flight, 5 GB
  // for a single solve: a large allocation costs seconds, and a context is rebuilt whenever the shape of the call changes)
  B.z_cap = 0;
  if (ws_on) {
    const size_t zb = (size_t)Y.N * Y.nz * 8;
    size_t zc = std::max<size_t>((size_t)2 << 20, (size_t)n_slots * 16384);
    zc = std::min(zc, std::min<size_t>((size_t)X.pool_cap, free_b / 8 / zb));
    if (!X.alloc(&B.pool_Z, zc * (size_t)Y.N * Y.nz)) return false;
    B.z_cap = (int)zc;
  }
  B.ws_on = ws_on ? 1 : 0;
  B.ws_mu = KNOB_T("MIQP_WS_MU") ? std::atof(KNOB_T("MIQP_WS_MU")) : 1.0;
  B.ws_delta = KNOB_T("MIQP_WS_DELTA") ? std::atof(KNOB_T("MIQP_WS_DELTA")) : 1.0e-3;
  if (!X.alloc(&B.pool_count, 1)) return false;
  if (!X.alloc(&B.free_q, (size_t)X.pool_cap)) return false;
  if (!X.alloc(&B.free_head, 1)) return false;
  if (!X.alloc(&B.free_tail, 1)) return false;
  if (!X.alloc(&B.free_limit, 1)) return false;
  if (!X.alloc(&B.open_bound, (size_t)2 * n_slots * open_cap)) return false;
  if (!X.alloc(&B.open_node, (size_t)2 * n_slots * open_cap)) return false;
  if (!X.alloc(&B.open_depth, (size_t)2 * n_slots * open_cap)) return false;
  if (!X.alloc(&B.open_key, (size_t)n_slots * open_cap)) return false;
  if (!X.alloc(&B.inst_slot, n_inst)) return false;
  if (!X.alloc(&B.slot_inst, n_slots)) return false;
  if (!X.alloc(&B.inst_kill, n_inst)) return false;
  if (!X.alloc(&B.slot_demand, n_slots)) return false;
  if (!X.alloc(&B.slot_take, n_slots)) return false;
  { int* rc; int* rn; int* rd; if (!X.alloc(&rc, n_inst)) return false; if (!X.alloc(&rn, (size_t)n_inst * roots_per_inst)) return false; if (!X.alloc(&rd, (size_t)n_inst * roots_per_inst)) return false;
    B.root_cnt = rc; B.root_node = rn; B.root_depth = rd; }
  if (!X.alloc(&X.d_pairs, (size_t)2 * n_slots)) return false;
  if (!X.alloc(&B.open_count, n_inst)) return false;
  B.far_cap = X.far_cap;
  if (X.far_cap > 0) {
    if (!X.alloc(&B.far_bound, (size_t)n_slots * X.far_cap)) return false;
    if (!X.alloc(&B.far_node, (size_t)n_slots * X.far_cap)) return false;
    if (!X.alloc(&B.far_depth, (size_t)n_slots * X.far_cap)) return false;
  }
  if (!X.alloc(&B.far_count, n_inst)) return false;
  if (!X.alloc(&B.far_minkey, n_inst)) return false;
  if (!X.alloc(&B.near_thr, n_inst)) return false;
  if (!X.alloc(&B.inst_mode, n_inst)) return false;
  if (!X.alloc(&B.inc_key, n_inst)) return false;
  if (!X.alloc(&B.inc_seen, n_inst)) return false;
  if (!X.alloc(&B.inc_obj, n_inst)) return false;
  if (!X.alloc(&B.inc_ext, n_inst)) return false;
  if (!X.alloc(&B.inc_fix, (size_t)n_inst * Y.fixlen)) return false;
  if (!X.alloc(&B.inc_Z, (size_t)n_inst * Y.N * Y.nz)) return false;
  if (Y.C == 2) { if (!X.alloc(&B.inc_A, (size_t)n_inst * 64) || !X.alloc(&B.inc_Mtag, (size_t)n_inst)) return false; } else { B.inc_A = nullptr; B.inc_Mtag = nullptr; }
  if (!X.alloc(&B.lower_bound, n_inst)) return false;
  if (!X.alloc(&B.inst_done, n_inst)) return false;
  if (!X.alloc(&B.inst_flags, n_inst)) return false;
  if (!X.alloc(&B.inst_gap, n_inst)) return false;
  if (!X.alloc(&B.inst_const, n_inst)) return false;
  if (!X.alloc(&B.inst_nodes, n_inst)) return false;
  if (!X.alloc(&B.inst_iters, n_inst)) return false;
  if (!X.alloc(&B.inst_ninc, n_inst)) return false;
  if (!X.alloc(&B.inst_lns, n_inst)) return false;
  if (!X.alloc(&B.inst_lns_obj, n_inst)) return false;
  if (!X.alloc(&B.batch_count, 1)) return false;
  if (!X.alloc(&B.batch_node, batch_alloc)) return false;
  if (!X.alloc(&B.batch_candkey, batch_alloc)) return false;
  HIP_OK(hipMemset(B.batch_candkey, 0xFF, (size_t)batch_alloc * 8));
  if (!X.alloc(&B.batch_candinst, batch_alloc)) return false;
  HIP_OK(hipMemset(B.batch_candinst, 0xFF, (size_t)batch_alloc * 4));
  if (!X.alloc(&B.batch_inst, batch_alloc)) return false;
  if (!X.alloc(&B.batch_large, batch_alloc)) return false;
  HIP_OK(hipMemset(B.batch_large, 0, (size_t)batch_alloc));
  if (!X.alloc(&B.batch_depth, batch_alloc)) return false;
  if (!X.alloc(&B.batch_Z, (size_t)batch_alloc * Y.N * Y.nz)) return false;
  if (!X.alloc(&B.batch_obj, batch_alloc)) return false;
  if (!X.alloc(&B.batch_viol, batch_alloc)) return false;
  if (!X.alloc(&B.batch_ok, batch_alloc)) return false;
  if (!X.alloc(&B.batch_it, batch_alloc)) return false;
  if (!X.alloc(&B.batch_bound, batch_alloc)) return false;
  if (!X.alloc(&B.batch_comp, (size_t)batch_alloc * Y.fixlen)) return false;
  if (!X.alloc(&B.rowstate, (size_t)std::min(batch_alloc, X.ipm_grid_max) * NFIELD * Y.ROWCAP)) return false;
  if (!X.alloc(&B.rowcache, (size_t)std::min(batch_alloc, X.ipm_grid_max) * NCACHE * Y.ROWCAP)) return false;
  if (!X.alloc(&B.kgain, (size_t)std::min(batch_alloc, X.kg_blocks) * std::max(Y.N * Y.nu * (Y.nx + 2), oc_gain_doubles(Y.N)))) return false;
  if (!X.alloc(&B.work_counter, 1)) return false;
  if (!X.alloc(&X.ctr, 2 * CTR_SET)) return false;
  HIP_OK(hipMemset(X.ctr, 0, 2 * CTR_SET * 4));
  X.as_cap = Y.C == 2 && X.oc_grid > 0;   // the shape has the active-set launches (whether a call uses them: MIQP_AS, read per call)
  X.as_on = X.as_cap;
  if (X.as_cap && !X.alloc(&X.cls_list, (size_t)3 * batch_cap)) return false;   // the class lists of a round (large_class 1 / 2 / 3)
  if (!X.alloc(&X.as_stats, 32)) return false;
  HIP_OK(hipMemset(X.as_stats, 0, 256));
  B.as_stats = X.as_stats; B.as_chunk = KNOB_T("MIQP_AS_CHUNK") ? std::atoi(KNOB_T("MIQP_AS_CHUNK")) : 1; B.as_quota = KNOB_T("MIQP_AS_QUOTA") ? std::atoi(KNOB_T("MIQP_AS_QUOTA")) : 0;   // (runs of 2 / 4 / 8 / 16: the standard launch 4.7 -> 5.2 / 6.6 / 7.9 / 10.3 ms - a wavefront solves ~14 nodes per launch, longer runs only lengthen its tail)
  B.batch_A = nullptr; B.pool_A = nullptr; X.as_batch_A = nullptr; X.as_pool_A = nullptr; B.ring_M = nullptr; B.ring_head = nullptr; B.ring_doubles = 0; B.ring_margin = 0; B.batch_Mtag = nullptr; B.pool_Mtag = nullptr;
  if (X.as_on && B.z_cap > 0 && Y.N * Y.NSLOT + 1024 < 65535) {   // the parents' active sets for the children's starts (128 B per record)
    if (!X.alloc(&B.batch_A, (size_t)batch_alloc * 64)) return false;
    if (!X.alloc(&B.pool_A, (size_t)B.z_cap * 64)) return false;
    // (pool_A / pool_Mtag need no initial value: a record's entries are written when the record is created - eval_kernel, lns_kernel - and roots start cold)
    HIP_OK(hipMemset(B.batch_A, 0xFF, (size_t)batch_alloc * 128));
    X.as_batch_A = B.batch_A; X.as_pool_A = B.pool_A;
    // ... and the ring their M travels through (4 KB per node on average, at most 12.8): a quarter of the free memory, at most 96 GB
    size_t rd = std::min<size_t>(std::min<size_t>((size_t)12 << 30, free_b / 4 / 8), std::max<size_t>((size_t)1 << 30, (size_t)n_slots * ((size_t)8 << 20)));   // doubles: 64 MB per instance in flight, at least 8 GB (a single solve: hundreds of its rounds; a large allocation costs seconds when the context is built)
    const size_t margin = (size_t)batch_alloc * AS_MSTR + ((size_t)1 << 20);   // what one round's launches can allocate, and more
    if (rd >= 2 * margin) {
      if (!X.alloc(&B.ring_M, rd)) return false;
      if (!X.alloc(&B.ring_head, 1)) return false;
      if (!X.alloc(&B.batch_Mtag, batch_alloc)) return false;
      if (!X.alloc(&B.pool_Mtag, (size_t)B.z_cap)) return false;
      { const unsigned long long h0 = 1024ull; HIP_OK(hipMemcpy(B.ring_head, &h0, 8, hipMemcpyHostToDevice)); }
      HIP_OK(hipMemset(B.batch_Mtag, 0, (size_t)batch_alloc * 8));
      B.ring_doubles = (unsigned long long)rd; B.ring_margin = (unsigned long long)margin;
    }
  }
  // buffers of the concurrent probe launch (two cars and fewer, on-chip kernel in use): 1024 resident blocks
  X.probe_grid = 0;
  if (X.oc_grid > 0 && !(KNOB_T("MIQP_PROBE_OVERLAP") && std::atoi(KNOB_T("MIQP_PROBE_OVERLAP")) == 0)) {
    X.probe_grid = std::min(batch_alloc, 1536);
    if (!X.alloc(&X.work_counter2, 1)) return false;
    if (!X.alloc(&X.rowstate2, (size_t)X.probe_grid * NFIELD * Y.ROWCAP)) return false;
    if (!X.alloc(&X.rowcache2, (size_t)X.probe_grid * NCACHE * Y.ROWCAP)) return false;
    if (!X.alloc(&X.kgain2, (size_t)X.probe_grid * std::max(Y.N * Y.nu * (Y.nx + 2), oc_gain_doubles(Y.N)))) return false;
    if (!X.stream2) { HIP_OK(hipStreamCreate(&X.stream2)); HIP_OK(hipEventCreate(&X.ev_fork)); HIP_OK(hipEventCreate(&X.ev_join)); HIP_OK(hipEventCreate(&X.ev_sfork)); HIP_OK(hipEventCreate(&X.ev_sjoin)); }
    if (!X.stream3) { HIP_OK(hipStreamCreate(&X.stream3)); HIP_OK(hipEventCreate(&X.ev_join3)); HIP_OK(hipStreamCreate(&X.stream4)); HIP_OK(hipEventCreate(&X.ev_join4)); }
    if (!X.alloc(&X.kgain3, (size_t)X.probe_grid * std::max(Y.N * Y.nu * (Y.nx + 2), oc_gain_doubles(Y.N)))) return false;
  }
  if (!X.alloc(&B.ovf_count, 1)) return false;
  if (!X.alloc(&B.ovf_list, batch_alloc)) return false;
  { if (!X.alloc(&B.pool_big, (size_t)X.pool_cap)) return false; HIP_OK(hipMemset(B.pool_big, 0, (size_t)X.pool_cap)); }   // (size mark of the on-chip variants; its high nibble counts the re-roundings of a probe, for every kernel)
  if (!X.alloc(&B.ovf2_count, 1)) return false;
  if (!X.alloc(&B.ovf2_list, batch_alloc)) return false;
  HIP_OK(hipMemset(B.ovf2_count, 0, 4));
  HIP_OK(hipMemset(B.ovf_count, 0, 4));
  if (KNOB_P("MIQP_STATS")) {
    if (!X.alloc(&B.stats, 256)) return false; HIP_OK(hipMemset(B.stats, 0, 256 * 8));
    if (!X.alloc(&B.pool_origin, (size_t)X.pool_cap)) return false; HIP_OK(hipMemset(B.pool_origin, 0, (size_t)X.pool_cap));
  }
  if (KNOB_T("MIQP_NOCUT")) B.use_cutoff = 0;   // diagnostic: every node relaxation runs to convergence (tells infeasible children from expensive ones)
  if (!X.alloc(&B.prof, 160 + 5 * 4 * 4096)) return false;
  (void)hipMemset(B.prof, 0, (160 + 5 * 4 * 4096) * 8);
  if (!X.alloc(&B.active_insts, 1)) return false;
  if (!X.alloc(&B.stat_rowiters, 1)) return false;
#ifdef MIQP_PROFILE
  if (!B.pool_origin) { if (!X.alloc(&B.pool_origin, (size_t)X.pool_cap)) return false; HIP_OK(hipMemset(B.pool_origin, 0, (size_t)X.pool_cap)); }
#endif
  X.ready = true;
  return true;
}

constexpr int IPM_NT = 64;  // one wavefront per node in the interior point kernel ...
constexpr int IPM_NT_WIDE = MIQP_WIDE_NT;   // ... two (or four) for three and four cars (ipm_kernel: the four MFMA tiles of the 2 x 2-tiled stage algebra are shared out over the wavefronts)
constexpr int ipm_nt(int C) { return C >= 3 ? IPM_NT_WIDE : IPM_NT; }
size_t ipm_lds_bytes(const Layout& Y) {
  int NZ = Y.nz, N = Y.N;
  size_t d = (size_t)N * NZ + (size_t)ipm_scratch_doubles(N, Y.C) + NZ + 8 + 32 + 2 * ((N + 6) / 2 + 1);
  return d * 8 + (size_t)Y.fixlen + 16;
}
bool multi_row_lifting_on() { const char* e = KNOB_P("MIQP_SEQ_KINDS"); return e && (std::strtoul(e, nullptr, 0) & 0x80000000ul) != 0ul; }
size_t eval_lds_bytes(const Layout& Y) {
  size_t d = (size_t)Y.N * Y.nz + (size_t)eval_shared_doubles(Y.C, Y.N, Y.P) + 2 * (size_t)Y.C * Y.N;   // (the slow-alternative table and the lifting's dense rows share one region)
  return d * 8 + (size_t)(2 * Y.C * Y.N + 64) * 4 + 64 * 8 + 3 * (size_t)Y.fixlen + 8 + 64
         + (multi_row_lifting_on() ? (size_t)(LIFT_ROWS * (2 * Y.nz + 3) + Y.nz) * 8 : 0) + 64   // rows of the multi-row lifting (an experiment: MIQP_SEQ_KINDS bit 31)
         + (size_t)Y.fixlen + 16;                                  // ploose
}

// `zero` false: the counters of the launch are the round's parity set, zeroed by roll_kernel one round ahead (no memset in the stream)
template <int C> void launch_ipm(const DevBuf& B, int nblocks, size_t lds, hipStream_t st, bool zero = true) { if (zero) (void)hipMemsetAsync(B.work_counter, 0, 4, st); hipLaunchKernelGGL((ipm_kernel<C, ipm_nt(C)>), dim3(nblocks), dim3(ipm_nt(C)), lds, st, B); }
template <int C> void launch_ipm_oc(const DevBuf& B, int nblocks, size_t lds, hipStream_t st, bool zero = true) {
  if (zero) { (void)hipMemsetAsync(B.work_counter, 0, 4, st); (void)hipMemsetAsync(B.ovf_count, 0, 4, st); }
  hipLaunchKernelGGL((ipm_onchip_kernel<C, OC_NSL>), dim3(nblocks), dim3(64), lds, st, B);
}
// the larger variant: works through the list of the standard one (ovf_mode 1) or through the rounding probes of the batch (ovf_mode 2, on its
// own stream with its own work counter); the caller has zeroed ovf2_count
template <int C> void launch_ipm_oc_big(const DevBuf& B, int nblocks, size_t lds, hipStream_t st, bool zero = true) {
  if (zero) (void)hipMemsetAsync(B.work_counter, 0, 4, st);
  hipLaunchKernelGGL((ipm_onchip_kernel<C, OC_NSL, 0, OC_GCAP_BIG>), dim3(nblocks), dim3(64), lds, st, B);
}
template <int C> void launch_eval(const DevBuf& B, int nblocks, size_t lds, hipStream_t st) { hipLaunchKernelGGL(eval_kernel<C>, dim3(nblocks), dim3(64), lds, st, B); }

void launch_ipm_c(int C, const DevBuf& B, int nblocks, size_t lds, hipStream_t st, bool zero = true) {
  switch (C) { case 1: launch_ipm<1>(B, nblocks, lds, st, zero); break; case 2: launch_ipm<2>(B, nblocks, lds, st, zero); break;
               case 3: launch_ipm<3>(B, nblocks, lds, st, zero); break; default: launch_ipm<4>(B, nblocks, lds, st, zero); }
}
// interior point solves of the first `bc` batch entries: the on-chip kernel where the shape qualifies, followed by the
// memory-backed kernel on the nodes it handed over (their count stays on the device: no host round trip); else the
// memory-backed kernel on everything
// The rounding probes of the batch (every disjunction fixed: >= 480 general rows, always beyond the on-chip capacity) are known
// before the launch - their depth word says so - and go to a launch of the memory-backed kernel on a SECOND stream that runs
// beside the on-chip kernel: that kernel waits on memory for most of its cycles (58 % in s_waitcnt), the on-chip kernel is bound
// by VALU issue, and the probes no longer cost a generation of their own behind it.  `overlap` false (the polish, solve_fixed): the
// serial order of round 2.
// `par` >= 0 (the rounds of a solve): the counters of this launch group are set `par` of X.ctr, zeroed one round ahead by roll_kernel
// LDS of a larger launch's block beside the standard active-set launch: padded to TWO blocks of that launch (rounded to the allocation granule), as
// long as four of them still fit a CU.  A larger block leaves a hole behind when it ends; the unpadded 34 KB took one standard block of 19.6 KB and
// wasted the rest - no CU ever held more than seven of its eight standard blocks again (tools/wave_dump.py, tools/resident_lab.hip)
inline size_t oc_big_lds_beside_as(const Layout& Y) {
  const size_t lb = (size_t)oc_lds_layout(Y.N, Y.fixlen, OC_GCAP_BIG).total, la = (size_t)oc_lds_layout(Y.N, Y.fixlen, OC_GCAP, true).total;
  const size_t st_ = 8 + 256;   // static LDS of the active-set kernels: the hand-out word and the wavefront's statistics
  const size_t two = 2 * ((la + st_ + 1279) / 1280 * 1280) - st_;
  return (two >= lb && 4 * (two + st_) <= 160 * 1024) ? two : lb;
}
void launch_ipm_batch(DevCtx& X, const DevBuf& B, int bc, hipStream_t st, bool overlap = false, int par = -1, hipEvent_t ev_std_end = nullptr) {
  const Layout& Y = X.Y;
  const size_t l_ipm = ipm_lds_bytes(Y);
  if (X.oc_grid > 0) {
    const size_t l_oc = (size_t)oc_lds_layout(Y.N, Y.fixlen).total;
    const size_t l_ocb = (size_t)oc_lds_layout(Y.N, Y.fixlen, OC_GCAP_BIG).total;
    const size_t l_ocb_as = Y.C == 2 && !(KNOB_T("MIQP_BIG_PAD") && std::atoi(KNOB_T("MIQP_BIG_PAD")) == 0) ? oc_big_lds_beside_as(Y) : l_ocb;
    const bool big = X.ocb_grid > 0;
    const bool ov = overlap && X.probe_grid > 0 && X.stream2 && bc <= 4096;   // (a full batch keeps the device busy on its own: measured no gain there, 5.4 against 5.1 s on a 2048-instance queue; single solves: median 6.0 instead of 7.0 ms)
    DevBuf Bc = B;
    const bool pc = par >= 0 && X.ctr && big && overlap && X.probe_grid > 0 && X.stream2 && X.concurrent_big;
    if (big && !pc) (void)hipMemsetAsync(B.ovf2_count, 0, 4, st);
    if (big && overlap && X.probe_grid > 0 && X.stream2 && X.concurrent_big) {
      // Every round: the larger variant works beside the standard one, on its own stream, through the nodes known to be large
      // (rounding probes, marked records); behind it, on that stream, the memory-backed kernel takes what even it cannot hold.
      // Both kernels hand their nodes out dynamically, so the wavefronts of the standard launch that find no room at first start
      // as the large nodes finish: the tail of the large nodes (40+ iterations) hides behind the standard launch instead of
      // being a launch of its own.  A node the standard kernel finds too large at its decode is marked and returned unsolved
      // (bounce): no second launch behind the standard one.
      (void)hipEventRecord(X.ev_fork, st); (void)hipStreamWaitEvent(X.stream2, X.ev_fork, 0);
      int* const cs = pc ? X.ctr + CTR_SET * par : nullptr;   // [0] batch count, [1] standard launch, [2] its hand-over list, [3] the larger variant's list, [4] the larger variant, [5] the memory-backed launch behind it
      if (pc) { Bc.work_counter = cs + 1; Bc.ovf_count = cs + 2; Bc.ovf2_count = cs + 3; }
      DevBuf Bp = Bc; Bp.ovf_mode = 2; Bp.work_counter = pc ? cs + 4 : X.work_counter2; Bp.rowstate = X.rowstate2; Bp.rowcache = X.rowcache2; Bp.kgain = X.kgain2;
      static const int big_grid_cap = KNOB_T("MIQP_BIG_GRID") ? std::atoi(KNOB_T("MIQP_BIG_GRID")) : 1 << 30;
      const int gb = std::min(std::min(bc, big_grid_cap), std::min(X.probe_grid, X.ocb_grid));
      const bool as2 = X.as_on && pc && Y.C == 2 && X.stream3;
      const bool lists = as2 && X.stream4 && B.cls_list && X.cls_n[0] >= 0;
      // With the class lists the larger launches get the SHARE of the device their work is of the round's, not all of it: their wavefronts take a SIMD
      // each (450 / 424 registers) and are enqueued first - a full grid of them held every SIMD until the larger active-set launch was through
      // (5 of 16 ms, tools/wave_dump.py), the standard launch started behind them and never got its holes back.  Weights: SIMD time of a node of the
      // class in units of a standard node's (half a SIMD for ~0.19 ms)
      static const double w1_ = KNOB_T("MIQP_BIG_W1") ? std::atof(KNOB_T("MIQP_BIG_W1")) : 6.0, w2_ = KNOB_T("MIQP_BIG_W2") ? std::atof(KNOB_T("MIQP_BIG_W2")) : 28.0;   // (driver's stream, launch group / standard launch: 9.10 / 8.41 ms at 5 / 20, 8.76 / 8.50 at 6 / 28, 8.87 / 8.53 at 8 / 36)
      int g1s = gb, g2s = gb;
      if (lists && w1_ > 0) {
        const double n1 = X.cls_n[0], n2 = X.cls_n[1], n0 = std::max(0, bc - X.cls_n[0] - X.cls_n[1] - X.cls_n[2]);
        const double den = w1_ * n1 + w2_ * n2 + n0 + 1.0;
        g1s = std::max(32, (int)(gb * w1_ * n1 / den + 0.5)); g2s = std::max(32, (int)(gb * w2_ * n2 / den + 0.5));
      }
      if (as2) {
        // the large nodes of the round that the active-set method takes (large_class 1): its larger block, on a third stream beside the interior
        // point chain, which keeps the rest of them
        (void)hipStreamWaitEvent(X.stream3, X.ev_fork, 0);
        DevBuf Bq = Bp; Bq.work_counter = cs + 6;
        if (lists) Bq.cls_take = 1;
        const int g1 = lists ? std::min(std::min(gb, g1s), X.cls_n[0]) : gb;   // (with the class lists: as many workgroups as the class has nodes, none when it is empty)
        if (g1 > 0) hipLaunchKernelGGL((as_onchip_kernel<2, OC_NSL, OC_GCAP_BIG>), dim3(g1), dim3(64), l_ocb_as, X.stream3, Bq);
        (void)hipEventRecord(X.ev_join3, X.stream3);
        Bp.as_split = 1;
      }
      if (lists) { Bp.cls_take = 2; const int g2 = std::min(std::min(gb, g2s), X.cls_n[1]); if (g2 > 0) launch_ipm_oc_big<2>(Bp, g2, l_ocb_as, X.stream2, false); Bp.cls_take = 0; }
      else
      if (Y.C == 1) launch_ipm_oc_big<1>(Bp, gb, l_ocb, X.stream2, !pc); else launch_ipm_oc_big<2>(Bp, gb, l_ocb, X.stream2, !pc);
      DevBuf Bm = Bp; Bm.ovf_mode = 1; Bm.ovf_count = Bc.ovf2_count; Bm.ovf_list = B.ovf2_list; if (pc) Bm.work_counter = cs + 5;
      if (as2 && X.stream4) {   // the records known to exceed the larger block: the memory-backed kernel beside the three others, on a fourth stream (gain buffer of its own)
        (void)hipStreamWaitEvent(X.stream4, X.ev_fork, 0);
        Bm.ovf_mode = 3; Bm.kgain = X.kgain3;
        if (lists) { Bm.cls_take = 3; const int g3 = std::min(X.probe_grid, X.cls_n[2]); if (g3 > 0) launch_ipm_c(Y.C, Bm, g3, l_ipm, X.stream4, false); }
        else
        launch_ipm_c(Y.C, Bm, std::min(bc, X.probe_grid), l_ipm, X.stream4, !pc);
        (void)hipEventRecord(X.ev_join4, X.stream4);
      } else
      launch_ipm_c(Y.C, Bm, std::min(bc, X.probe_grid), l_ipm, X.stream2, !pc);
      (void)hipEventRecord(X.ev_join, X.stream2);
      Bc.skip_probes = 1; Bc.bounce = 1;
      if (X.as_on && pc && Y.C == 2) {
        // the ordinary nodes of the round: dual active-set solves (as_onchip.hip) in place of the standard interior point launch; a node that
        // launch cannot finish comes back marked, like a node that is too large, and the larger interior point variant takes it next round
        DevBuf Ba = Bc; Ba.work_counter = cs + 1;
        const size_t l_as = (size_t)oc_lds_layout(Y.N, Y.fixlen, OC_GCAP, true).total;
        const int ch_ = std::max(1, Ba.as_chunk);
        const int ga = Ba.as_quota > 0 ? std::max(1, (bc + Ba.as_quota * ch_ - 1) / (Ba.as_quota * ch_)) : std::min(bc, X.oc_grid);
        hipLaunchKernelGGL((as_onchip_kernel<2, OC_NSL>), dim3(ga), dim3(64), l_as, st, Ba);
      } else if (Y.C == 1) launch_ipm_oc<1>(Bc, std::min(bc, X.oc_grid), l_oc, st, !pc); else launch_ipm_oc<2>(Bc, std::min(bc, X.oc_grid), l_oc, st, !pc);
      if (X.ev_mid) (void)hipEventRecord(X.ev_mid, st);
      if (ev_std_end) (void)hipEventRecord(ev_std_end, st);   // (where the standard launch of the round ends: the dominant kernel's own time)
      (void)hipStreamWaitEvent(st, X.ev_join, 0);
      if (as2) (void)hipStreamWaitEvent(st, X.ev_join3, 0);
      if (as2 && X.stream4) (void)hipStreamWaitEvent(st, X.ev_join4, 0);
      return;
    }
    if (ov) {
      (void)hipEventRecord(X.ev_fork, st); (void)hipStreamWaitEvent(X.stream2, X.ev_fork, 0);
      DevBuf Bp = B; Bp.ovf_mode = 2; Bp.work_counter = X.work_counter2; Bp.rowstate = X.rowstate2; Bp.rowcache = X.rowcache2; Bp.kgain = X.kgain2;
      if (big) { if (Y.C == 1) launch_ipm_oc_big<1>(Bp, std::min(bc, std::min(X.probe_grid, X.ocb_grid)), l_ocb, X.stream2); else launch_ipm_oc_big<2>(Bp, std::min(bc, std::min(X.probe_grid, X.ocb_grid)), l_ocb, X.stream2); }
      else launch_ipm_c(Y.C, Bp, std::min(bc, X.probe_grid), l_ipm, X.stream2);
      (void)hipEventRecord(X.ev_join, X.stream2);
      Bc.skip_probes = 1;
    }
    if (Y.C == 1) launch_ipm_oc<1>(Bc, std::min(bc, X.oc_grid), l_oc, st); else launch_ipm_oc<2>(Bc, std::min(bc, X.oc_grid), l_oc, st);
    if (X.ev_mid) (void)hipEventRecord(X.ev_mid, st);   // (diagnostic, MIQP_LAUNCH_TRACE: where the on-chip kernel ends)
    if (ov) (void)hipStreamWaitEvent(st, X.ev_join, 0);   // (the launches below and the evaluation need the probes' results)
    DevBuf Bo = B; Bo.ovf_mode = 1;
    if (big) {
      // the nodes the standard variant could not hold go to the larger one (up to OC_GCAP_BIG general rows: the rounding probes
      // without their slack front-point rows, the other large nodes); what even that one cannot hold - the polish of an
      // incumbent, every row of it - to the memory-backed kernel
      if (Y.C == 1) launch_ipm_oc_big<1>(Bo, std::min(bc, X.ocb_grid), l_ocb, st); else launch_ipm_oc_big<2>(Bo, std::min(bc, X.ocb_grid), l_ocb, st);
      Bo.ovf_count = B.ovf2_count; Bo.ovf_list = B.ovf2_list;
    }
    launch_ipm_c(Y.C, Bo, std::min(bc, X.ipm_grid_max), l_ipm, st);   // blocks without a node read the count and leave
  } else {
    launch_ipm_c(Y.C, B, std::min(bc, X.ipm_grid_max), l_ipm, st);
  }
}

void launch_eval_c(int C, const DevBuf& B, int nblocks, size_t lds, hipStream_t st) {
  switch (C) { case 1: launch_eval<1>(B, nblocks, lds, st); break; case 2: launch_eval<2>(B, nblocks, lds, st); break;
               case 3: launch_eval<3>(B, nblocks, lds, st); break; default: launch_eval<4>(B, nblocks, lds, st); }
}

template <int C> bool set_kernel_lds_oc(size_t lds, size_t lds_big) {
  HIP_OK(hipFuncSetAttribute((const void*)ipm_onchip_kernel<C, OC_NSL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  HIP_OK(hipFuncSetAttribute((const void*)ipm_onchip_kernel<C, OC_NSL, 0, OC_GCAP_BIG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_big));
  if constexpr (C == 2) { HIP_OK(hipFuncSetAttribute((const void*)as_onchip_kernel<2, OC_NSL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIP_OK(hipFuncSetAttribute((const void*)as_onchip_kernel<2, OC_NSL, OC_GCAP_BIG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_big)); }
  return true;
}
template <int C> bool set_kernel_lds_c(size_t ipm_lds, size_t eval_lds) {
  HIP_OK(hipFuncSetAttribute((const void*)ipm_kernel<C, ipm_nt(C)>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ipm_lds));
  HIP_OK(hipFuncSetAttribute((const void*)eval_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)eval_lds));
  return true;
}
bool set_kernel_lds(const Layout& Y, size_t ipm_lds, size_t eval_lds) {
  if (Y.C <= 2 && Y.N <= 2 * OC_NSL) {
    const size_t l = (size_t)oc_lds_layout(Y.N, Y.fixlen).total, lb = std::max((size_t)oc_lds_layout(Y.N, Y.fixlen, OC_GCAP_BIG).total, Y.C == 2 ? oc_big_lds_beside_as(Y) : (size_t)0);
    if (l <= 160 * 1024 && !(Y.C == 1 ? set_kernel_lds_oc<1>(l, lb) : set_kernel_lds_oc<2>(l, lb))) return false;
  }
  switch (Y.C) { case 1: return set_kernel_lds_c<1>(ipm_lds, eval_lds); case 2: return set_kernel_lds_c<2>(ipm_lds, eval_lds);
                 case 3: return set_kernel_lds_c<3>(ipm_lds, eval_lds); default: return set_kernel_lds_c<4>(ipm_lds, eval_lds); }
}

// ---------------------------------------------------------------- results (collectRawResults, cplex_wrapper.cpp:311-448)
// canonical binaries: 0 wherever the asserted side holds for the returned continuous point
void fill_results(const HostInst& I, const Layout& Y, const double* D, const int* T, const signed char* comp, const double* Z,
                  miqp_raw_results_c* r) {
  HostGeo G{I, Y, D, T};
  const int C = I.C, N = I.N, R = I.R, E = I.E, O = I.O, L = I.L, K = I.C - 1, nz = Y.nz;
  const double tol = 10 * FEAS_TOL;
  r->N = N; r->NrEnvironments = E; r->NrRegions = R; r->NrObstacles = O; r->MaxLinesObstacles = L; r->NrCarToCarCollisions = K; r->NrCars = C;
  for (int c = 0; c < C; ++c)
    for (int i = 0; i < N; ++i) {
      const double* z = Z + (size_t)i * nz; int qi = c * N + i;
      r->pos_x[qi] = z[6 * c]; r->vel_x[qi] = z[6 * c + 1]; r->acc_x[qi] = z[6 * c + 2];
      r->pos_y[qi] = z[6 * c + 3]; r->vel_y[qi] = z[6 * c + 4]; r->acc_y[qi] = z[6 * c + 5];
      r->u_x[qi] = z[6 * C + 2 * c]; r->u_y[qi] = z[6 * C + 2 * c + 1];
      int code = i >= 1 ? (int)comp[Y.f_reg + c * N + i] : -1;
      int q = i >= 1 ? (code >> 2) : 0;
      int j = i >= 1 ? T[Y.i_regj + c * Y.P + q] : I.init_region[c] - 1;
      double X, Yc;
      G.point(c, i, q, z, PT_U, PT_U, X, Yc); r->pos_x_front_UB[qi] = X; r->pos_y_front_UB[qi] = Yc;
      G.point(c, i, q, z, PT_L, PT_L, X, Yc); r->pos_x_front_LB[qi] = X; r->pos_y_front_LB[qi] = Yc;
      for (int jj = 0; jj < R; ++jj) r->active_region[(c * N + i) * R + jj] = jj == j ? 1 : 0;
      int xp = 0, yp = 0, xn = 0, yn = 0, cb = 0;
      if (i >= 1) {
        double vx = z[6 * c + 1], vy = z[6 * c + 4];
        xp = vx <= I.vm; xn = vx >= -I.vm; yp = vy <= I.vm; yn = vy >= -I.vm;
        int h = code & 3;
        if (h == 3) { xp = xn = yp = yn = cb = 1; }
        else {
          const int* hs = T + Y.i_hs + ((c * Y.P + q) * 2 + h) * 2;
          if (hs[0] == 0 && hs[1] > 0) xp = 0;
          if (hs[0] == 0 && hs[1] < 0) xn = 0;
          if (hs[0] == 1 && hs[1] > 0) yp = 0;
          if (hs[0] == 1 && hs[1] < 0) yn = 0;
        }
      }
      r->region_change_not_allowed_x_positive[qi] = xp; r->region_change_not_allowed_y_positive[qi] = yp;
      r->region_change_not_allowed_x_negative[qi] = xn; r->region_change_not_allowed_y_negative[qi] = yn;
      r->region_change_not_allowed_combined[qi] = cb;
      int* nw[5] = {r->notWithinEnvironmentRear, r->notWithinEnvironmentFrontUbUb, r->notWithinEnvironmentFrontLbUb,
                    r->notWithinEnvironmentFrontUbLb, r->notWithinEnvironmentFrontLbLb};
      for (int pt = 0; pt < 5; ++pt) {
        G.point(c, i, q, z, ENV_PT_H[pt][0], ENV_PT_H[pt][1], X, Yc);
        for (int e = 0; e < E; ++e) nw[pt][(c * E + e) * N + i] = G.env_viol(e, X, Yc) <= tol ? 0 : 1;
      }
      for (int o = 0; o < O; ++o)
        for (int pt = 0; pt < 5; ++pt) {
          bool ignored = i >= 1 && comp[Y.f_obs + ((c * O + o) * N + i) * 5 + pt] >= L;
          G.point(c, i, q, z, OBS_PT_H[pt][0], OBS_PT_H[pt][1], X, Yc);
          for (int k = 0; k < L; ++k) {
            int d = (G.obs_viol(o, i, k, X, Yc) <= tol && !ignored) ? 0 : 1;
            if (pt == 0) r->deltacc[((c * O + o) * N + i) * L + k] = d; else r->deltacc_front[(((c * O + o) * N + i) * L + k) * 4 + pt - 1] = d;
          }
          if (pt == 0) r->slackvarsObstacle[(c * O + o) * N + i] = ignored ? 1 : 0; else r->slackvarsObstacle_front[((c * O + o) * N + i) * 4 + pt - 1] = ignored ? 1 : 0;
        }
    }
  for (int a = 0; a < K * K * N * 16; ++a) r->car2car_collision[a] = 0;
  for (int a = 0; a < K * K * N * 4; ++a) { r->slackvars[a] = 0; if (r->slackvars_real) r->slackvars_real[a] = 0; }
  int p = 0;
  for (int c1 = 0; c1 < C; ++c1)
    for (int c2 = c1 + 1; c2 < C; ++c2, ++p)
      for (int i = 0; i < N; ++i) {
        const double* z = Z + (size_t)i * nz;
        int q1 = i >= 1 ? (comp[Y.f_reg + c1 * N + i] >> 2) : 0, q2 = i >= 1 ? (comp[Y.f_reg + c2 * N + i] >> 2) : 0;
        for (int g = 0; g < 4; ++g) {
          double vals[4]; for (int a = 0; a < 4; ++a) vals[a] = G.c2c_viol(p, c1, c2, i, g, a, q1, q2, z);
          int chosen = i >= 1 ? (int)comp[Y.f_c2c + (p * N + i) * 4 + g] : -1;
          if (chosen < 0) { chosen = 0; for (int a = 1; a < 4; ++a) if (vals[a] < vals[chosen]) chosen = a; }
          double sl[2] = {0, 0};
          if ((g == 0 || g == 3) && vals[chosen] > 0) sl[chosen < 2 ? 0 : 1] = vals[chosen];
          for (int a = 0; a < 4; ++a) {
            double need = vals[a] - ((g == 0 || g == 3) ? sl[a < 2 ? 0 : 1] : 0.0);
            r->car2car_collision[((c1 * K + (c2 - 1)) * N + i) * 16 + 4 * g + a] = (a == chosen || need <= tol) ? 0 : 1;
          }
          if (g == 0 || g == 3)
            for (int s = 0; s < 2; ++s) {
              int idx = ((c1 * K + (c2 - 1)) * N + i) * 4 + (g == 0 ? 0 : 2) + s;
              r->slackvars[idx] = (int)sl[s];  // truncation to int as in RawResults (miqp_planner_data.hpp:84-88)
              if (r->slackvars_real) r->slackvars_real[idx] = sl[s];
            }
        }
      }
}

// fix record from a complete assignment (warm start / solve_fixed): first asserted alternative of each disjunction
bool fix_from_results(const HostInst& I, const Layout& Y, const int* T, const miqp_raw_results_c* f, std::vector<signed char>& fix) {
  const int C = I.C, N = I.N, R = I.R, E = I.E, O = I.O, L = I.L, K = I.C - 1;
  fix.assign(Y.fixlen, -1);
  for (int c = 0; c < C; ++c)
    for (int i = 1; i < N; ++i) {
      int j = -1; for (int jj = 0; jj < R; ++jj) if (f->active_region[(c * N + i) * R + jj] == 1) j = jj;
      int q = -1; for (int k = 0; k < T[Y.i_nposs + c]; ++k) if (T[Y.i_regj + c * Y.P + k] == j) q = k;
      // a step without a (possible) active region stays undecided: the start is then a partial assignment that the
      // search completes (CPLEX repairs / completes partial MIP starts; the shifted receding-horizon start has an
      // all-zero last region row, src/miqp_planner.cpp:982)
      if (q >= 0) {
        int h = 3;
        if (f->region_change_not_allowed_combined[c * N + i] != 1) {
          h = 0;
          for (int k = 0; k < T[Y.i_nhs + c * Y.P + q]; ++k) {
            const int* hs = T + Y.i_hs + ((c * Y.P + q) * 2 + k) * 2;
            int b = hs[0] == 0 ? (hs[1] > 0 ? f->region_change_not_allowed_x_positive[c * N + i] : f->region_change_not_allowed_x_negative[c * N + i])
                               : (hs[1] > 0 ? f->region_change_not_allowed_y_positive[c * N + i] : f->region_change_not_allowed_y_negative[c * N + i]);
            if (b == 0) { h = k; break; }
          }
        }
        fix[Y.f_reg + c * N + i] = (signed char)(q * 4 + h);
      }
      const int* nw[5] = {f->notWithinEnvironmentRear, f->notWithinEnvironmentFrontUbUb, f->notWithinEnvironmentFrontLbUb,
                          f->notWithinEnvironmentFrontUbLb, f->notWithinEnvironmentFrontLbLb};
      for (int pt = 0; pt < 5; ++pt) {
        for (int e = 0; e < E; ++e) if (nw[pt][(c * E + e) * N + i] == 0) { fix[Y.f_env + (c * N + i) * 5 + pt] = (signed char)e; break; }
        for (int o = 0; o < O; ++o) {
          int so = pt == 0 ? f->slackvarsObstacle[(c * O + o) * N + i] : f->slackvarsObstacle_front[((c * O + o) * N + i) * 4 + pt - 1];
          if (so >= 1 && I.obs_soft[o]) { fix[Y.f_obs + ((c * O + o) * N + i) * 5 + pt] = (signed char)L; continue; }
          for (int k = 0; k < L; ++k) {
            int d = pt == 0 ? f->deltacc[((c * O + o) * N + i) * L + k] : f->deltacc_front[(((c * O + o) * N + i) * L + k) * 4 + pt - 1];
            if (d == 0) { fix[Y.f_obs + ((c * O + o) * N + i) * 5 + pt] = (signed char)k; break; }
          }
        }
      }
    }
  int p = 0;
  for (int c1 = 0; c1 < C; ++c1)
    for (int c2 = c1 + 1; c2 < C; ++c2, ++p)
      for (int i = 1; i < N; ++i)
        for (int g = 0; g < 4; ++g)
          for (int a = 0; a < 4; ++a)
            if (f->car2car_collision[((c1 * K + (c2 - 1)) * N + i) * 16 + 4 * g + a] == 0) { fix[Y.f_c2c + (p * N + i) * 4 + g] = (signed char)a; break; }
  return true;
}

struct BatchShape { Layout Y; bool ok; std::string err; };

BatchShape batch_layout(miqp_solver_t* const* S, int n) {
  BatchShape bs; bs.ok = false;
  const HostInst& I0 = S[0]->inst;
  int P = 1, EL = 0;
  for (int k = 0; k < n; ++k) {
    if (!S[k] || !S[k]->has_inst) { bs.err = "solver without parameters"; return bs; }
    const HostInst& I = S[k]->inst;
    if (I.C != I0.C || I.N != I0.N || I.O != I0.O || I.L != I0.L || I.E != I0.E || I.R != I0.R) { bs.err = "instances of one batch must share NumCars, NumSteps, nr_regions, nr_environments, nr_obstacles, max_lines_obstacles"; return bs; }
    P = std::max(P, max_possible(I)); EL = std::max(EL, max_env_edges(I));
  }
  if (I0.C > MAXC) { bs.err = "NumCars > 4 is not supported by the device kernels"; return bs; }
  if (P > 15) { bs.err = "more than 15 possible regions per car"; return bs; }
  // one branching creates a child per alternative of the chosen disjunction, at most 63 per node (eval_kernel): a shape
  // with more alternatives is refused instead of searched incompletely
  if (I0.E > 62 || I0.L + 1 > 62) { bs.err = "more than 62 environment pieces / 61 obstacle edges"; return bs; }
  bs.Y = make_layout(I0.C, I0.N, I0.R, P, I0.E, EL, I0.O, I0.L); bs.ok = true;
  return bs;
}

unsigned long long host_d2key(double v) { unsigned long long u; std::memcpy(&u, &v, 8); return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull); }
double host_key2d(unsigned long long k) { unsigned long long u = (k & 0x8000000000000000ull) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k; double v; std::memcpy(&v, &u, 8); return v; }

double wall_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ---------------------------------------------------------------- the batch solve
// Tree split of ONE instance over the ranks of a job.  The alternatives of one or two car/car
// disjunctions partition the tree (every rank starts from the roots it owns); once per round the ranks exchange
// {incumbent | owner rank, lower bound, done, time-up} by an all-reduce(min) of four 64-bit words - the best incumbent prunes
// on every rank - and at the end the owner broadcasts the solution.
struct SplitCtx { int world, rank; miqp_exchange_fn fn; void* user; };
constexpr int SPLIT_MAXROOTS = 64;

// root records of the split: the cartesian product of the allowed alternatives of the rear/rear group of every car pair at
// the last step, then at the middle step, until there are at least 4 combinations per rank (at most SPLIT_MAXROOTS)
void split_roots(const Layout& Y, const int* T, std::vector<std::vector<std::pair<int, int>>>& combos, int world) {
  combos.assign(1, {});
  if (Y.C < 2) return;   // a single car has no car/car disjunction: rank 0 keeps the whole tree
  const int steps[2] = {Y.N - 1, (Y.N - 1) / 2};
  for (int si = 0; si < 2; ++si) {
    const int i = steps[si];
    if (i < 1 || (si == 1 && i == steps[0])) continue;
    for (int p = 0; p < Y.NP; ++p) {
      if ((int)combos.size() >= 4 * world) return;
      const int am0 = T[Y.i_c2callow + p * Y.N + i] & 15, am = am0 ? am0 : 15;
      int na = 0; for (int a = 0; a < 4; ++a) na += (am >> a) & 1;
      if (na < 2 || (int)combos.size() * na > SPLIT_MAXROOTS) continue;
      std::vector<std::vector<std::pair<int, int>>> next;
      for (auto& c : combos) for (int a = 0; a < 4; ++a) if ((am >> a) & 1) { auto d = c; d.emplace_back(Y.f_c2c + (p * Y.N + i) * 4 + 0, a); next.push_back(std::move(d)); }
      combos.swap(next);
    }
  }
}

// `inflight`: instances solved concurrently (<= 0 or >= n: all of them).  With fewer than n the call is a queue drained by
// streaming admission: an instance that is proven (or has used up its own max_solution_time, counted from its admission)
// hands its slot to the next one at the following round.
// `lane` / `lanes`: this call is one of `lanes` concurrent calls that share the device, each with its own context (miqp_solver_solve_stream)
bool solve_batch_impl(miqp_solver_t* const* S, int n, int* statuses, const SplitCtx* split = nullptr, int inflight = 0, int lane = 0, int lanes = 1) {
  // every way out of this function that is not the result loop at its end (a failed HIP call, a failed exchange) leaves "the solver
  // could not run" behind: a caller's zero-filled status array would otherwise read as SUCCESS for instances that were never solved
  for (int k = 0; k < n; ++k) { statuses[k] = MIQP_STATUS_FAILED_SEG_FAULT; if (S[k]) { S[k]->status = MIQP_STATUS_FAILED_SEG_FAULT; S[k]->has_sol = false; S[k]->rescache.reset(); } }
  if (split && n != 1) return false;
  BatchShape bs = batch_layout(S, n);
  if (!bs.ok) { for (int k = 0; k < n; ++k) { if (S[k]) S[k]->err = bs.err; statuses[k] = MIQP_STATUS_FAILED_SEG_FAULT; } std::fprintf(stderr, "[miqp_gpu] %s\n", bs.err.c_str()); return false; }
  const Layout& Y = bs.Y;
  const miqp_solver_opts& O0 = S[0]->opts;
  auto fail_all = [&](const char* why) { if (why) std::fprintf(stderr, "[miqp_gpu] %s\n", why); for (int k = 0; k < n; ++k) statuses[k] = MIQP_STATUS_FAILED_SEG_FAULT; return false; };
  for (int k = 1; k < n; ++k) if (S[k]->opts.device != O0.device) return fail_all("instances of one batch must name the same device (use miqp_solver_solve_batch_multi to span devices)");
  const double t_enter = wall_s();
  DevCtx* Xp = ctx_for_device(O0.device, lane);
  if (!Xp) return fail_all(nullptr);
  warn_ignored_switches();
  DevCtx& X = *Xp;
  std::lock_guard<std::mutex> ctx_lock(X.mu);
  if (hipSetDevice(X.device) != hipSuccess) return fail_all("hipSetDevice failed");
  const int NS = (split || inflight <= 0 || inflight >= n) ? n : inflight;   // slots = instances in flight
  // nodes of a round, all instances together: 32768 for three and four cars (and one), 98304 for two - since the node relaxations of two cars are active-set
  // solves (round 6) a round of 32768 lasts 5-6 ms of which 1.1 ms are the serial kernels and the tail of the launch; the driver's stream at 32 k / 49 k / 65 k /
  // 97 k / 131 k / 164 k nodes per round: 1783 / 1849 / 1864 / 1911 / 1856 / 1826 solves/s (time to a proof p50 0.043 -> 0.106 s, p99 7.4 -> 4.3 s at 97 k)
  const int round_nodes = Y.C == 2 ? 98304 : 32768;
  int npr = O0.nodes_per_round > 0 ? O0.nodes_per_round : std::max(16, std::min(16384, round_nodes / (NS * lanes)));   // (the lanes of a call keep the round width of the undivided call)
  // three and four cars: a node relaxation costs ~15 x that of two cars (memory-backed kernel, stage vector 24 / 32), a round of 32768 nodes
  // lasts 0.13 s and an instance that shares the device gets 80 rounds in its 10 s - fewer than the levels of its first dive.  Rounds of
  // 5120 nodes (0.035 s) give the tree its depth back at a quarter less node throughput: cfg5, 16 in flight, 7 -> 11 of 16 proven in 10 s.
  // Since the re-rounding finds the first incumbents within a few rounds the depth matters less and the throughput more: 8192 nodes
  // (3072 / 5120 / 8192 / 10240 / 12288 / 16384: 14 / 14 / 15 / 14 / 14 / 14 of 16, the open gaps smallest at 8192 - 12288; profiles/r04b_heuristics_ab.txt)
  if (O0.nodes_per_round <= 0 && Y.C >= 3 && NS > 1) npr = std::max(16, std::min(npr, 8192 / NS));
  // ... and a single solve of three or four cars takes 2048 nodes per round (16384 / 8192 / 4096 / 2048: the sixteen cfg5 seeds in 35.1 / 34.3 / 29.5 / 25.3 s, 15 proven each time;
  // seed 15 8.5 -> 4.5 s): its rounds are as long as their slowest node whatever their width (~25 ms), so what a narrower round gives up is node throughput it could not use for the proof anyway
  if (O0.nodes_per_round <= 0 && Y.C >= 3 && NS == 1) npr = std::min(npr, 2048);
  // ... and so does a single solve of one or two cars (round 5): 2048 nodes are one per resident wavefront of the standard on-chip launch (256 CUs x 8) - a
  // wider round lasts longer (several nodes per wavefront) and solves nodes that the incumbents of a narrower round would have pruned.  Measured on seeds 0-95
  // (tools/single_latency.py, profiles/r05_single_latency.txt), 16384 -> 2048 nodes per round: p99 58 -> 47 ms at gap 0.1 (seed 62: 55 k -> 26 k node
  // relaxations), 92 -> 75 ms at 0.01; p50 / p90 unchanged (5 / 18 ms); 1024: 51 / 76 ms (too narrow: more rounds), 4096: 51 / 86, 8192: 58 / 90
  if (O0.nodes_per_round <= 0 && Y.C <= 2 && NS == 1) npr = std::min(npr, 2048);
  if (O0.nodes_per_round <= 0 && KNOB_P("MIQP_NPR")) npr = std::max(1, std::atoi(KNOB_P("MIQP_NPR")));  // tuning knob
  // A single solve starts narrow (above) and WIDENS its rounds once it is bound-limited: the incumbent has not moved for 32 rounds and the
  // near list offers eight rounds' worth of eligible nodes - then the tree needs node throughput, not fresher incumbents (cfg5 seed 11: 1.61 / 2.17 / 2.92 M relaxations in 10 s at
  // 1024 / 2048 / 4096 nodes per round).  The batch arrays are sized for the widest round, select_kernel caps at width_cap
  const bool adaptive_width = NS == 1 && !split && O0.nodes_per_round <= 0 && !KNOB_P("MIQP_NPR");
  const int width0 = npr, width_max = Y.C >= 3 ? 4096 : 16384;   // (three and four cars: 4096 - cfg5 seed 11 relaxes 2.57 / 3.21 / 3.09 / 2.96 M nodes in its 10 s at 2048 / 4096 / 8192 / 16384, its gap 2.1 / 1.8 / 3.4 / 45 %)
  if (adaptive_width) npr = width_max;
  int open_cap = O0.max_open_nodes > 0 ? O0.max_open_nodes : (KNOB_P("MIQP_OPEN_CAP") ? std::atoi(KNOB_P("MIQP_OPEN_CAP")) : std::max(1 << 17, std::min(1 << 20, (1 << 28) / NS)));   // near lists: 1 M entries per instance up to n = 256 (10 GB of list entries), 262144 at n = 1024; records are shared
  if (open_cap < 64) open_cap = 64;
  if (split && open_cap < SPLIT_MAXROOTS + 4 + 64) open_cap = SPLIT_MAXROOTS + 4 + 64;   // the root records of a tree split are the head of the list
  // (list entries - 40 B per open node - must fit an eighth of the free device memory: ctx_prepare cuts the capacity when it builds the context)
  if ((size_t)NS * npr >= ((size_t)1 << 20)) npr = (int)((((size_t)1 << 20) - 1) / NS);
  const int MAXR = split ? SPLIT_MAXROOTS + 4 : 5;   // root records per instance: the root (or this rank's roots of a tree split), the MIP starts and their repair roots
  size_t l_ipm = ipm_lds_bytes(Y), l_eval = eval_lds_bytes(Y);
  bool ctx_built = false;
  {
    // the fallible part of the set-up (device buffers, kernel attributes).  In a tree split the ranks agree on its outcome with
    // one exchange before the first round: a rank that failed alone would otherwise leave its peers waiting in their all-reduce
    ctx_built = !(X.ready && same_layout(X.Y, Y) && n <= X.n_inst_cap && X.n_slots == NS && X.open_cap_req == open_cap && X.npr == npr && X.B.root_stride == MAXR && X.mem_div == lanes);
    bool setup_ok = ctx_prepare(X, Y, n, NS, open_cap, npr, MAXR, O0.max_open_nodes <= 0, lanes);
    open_cap = X.open_cap;
    if (setup_ok && (l_ipm > 160 * 1024 || l_eval > 160 * 1024)) { std::fprintf(stderr, "[miqp_gpu] instance too large for LDS (%zu bytes)\n", l_ipm); setup_ok = false; }
    if (setup_ok && !set_kernel_lds(Y, l_ipm, l_eval)) setup_ok = false;
    if (split) {
      unsigned long long w = setup_ok ? 1ull : 0ull;
      if (split->fn(split->user, 0, &w, 1, 0) != 0) return fail_all("set-up exchange failed");
      if (setup_ok && w == 0ull) return fail_all("another rank of the tree split failed its set-up");
    }
    if (!setup_ok) return fail_all(nullptr);
  }
  const double t_ctx = wall_s() - t_enter;
  DevBuf& B = X.B;
  // ---- host tables, step-1 presolve
  // (not value-initialised: compile_instance writes every word of an instance's slice, and zeroing 1.4 GB for the 51 200 instances of the driver's stream
  // on the calling thread came before the threads below could start)
  const size_t nD_ = (size_t)n * Y.dstride, nT_ = (size_t)n * Y.istride;
  std::unique_ptr<double[]> hD_own(new double[nD_]); std::unique_ptr<int[]> hT_own(new int[nT_]);
  double* const hD = hD_own.get(); int* const hT = hT_own.get();
  std::vector<double> h_const(n, 0.0), h_gap(n), h_tlim(n);
  std::vector<int> h_done(n, 0);
  std::vector<signed char> roots; roots.reserve((size_t)MAXR * n * Y.fixlen);
  std::vector<int> on((size_t)n * MAXR, 0), od((size_t)n * MAXR, 0), oc(n, 0);   // root records of every instance: the head of its open list at admission
  int nrec = 0;
  auto add_root = [&](int k, const std::vector<signed char>& fx, int depth_word) { roots.insert(roots.end(), fx.begin(), fx.end()); on[(size_t)k * MAXR + oc[k]] = nrec++; od[(size_t)k * MAXR + oc[k]] = depth_word; oc[k]++; };
  int active = 0;
  // per instance (independent, spread over host threads): tables, step-0 check, the fix records of its roots
  std::vector<std::vector<std::vector<signed char>>> inst_roots(n);
  std::vector<std::vector<int>> inst_root_depth(n);
  std::vector<char> h_feas0(n, 0);
  auto prepare_one = [&](int k) {
    miqp_solver* s = S[k]; s->lay = Y; s->has_sol = false; s->rescache.reset();
    compile_instance(s->inst, Y, &hD[(size_t)k * Y.dstride], &hT[(size_t)k * Y.istride]);
    HostGeo G{s->inst, Y, &hD[(size_t)k * Y.dstride], &hT[(size_t)k * Y.istride]};
    double cobj = 0; bool feas0 = step0_check(G, cobj);
    h_const[k] = cobj; h_gap[k] = s->opts.gap_override >= 0 ? s->opts.gap_override : s->inst.gap; h_tlim[k] = s->inst.tilim;
    h_feas0[k] = feas0 ? 1 : 0;
    auto& R = inst_roots[k];
    if (feas0 && !split) R.emplace_back(Y.fixlen, (signed char)-1);
    else if (feas0) {
      std::vector<std::vector<std::pair<int, int>>> combos; split_roots(Y, &hT[(size_t)k * Y.istride], combos, split->world);
      for (size_t q = 0; q < combos.size(); ++q) {
        if ((int)(q % (size_t)split->world) != split->rank) continue;
        std::vector<signed char> fx(Y.fixlen, (signed char)-1);
        for (auto& d : combos[q]) fx[d.first] = (signed char)d.second;
        R.push_back(std::move(fx));
      }
    }
    // MIP starts (initializeWarmstart + addMIPStart / readMIPStarts, src/cplex_wrapper.cpp:124-138, 494-639): the binaries
    // of each start become an additional root whose QP the first round solves; a feasible one is the first incumbent
    for (int w = 0; w < 2 && feas0; ++w) {
      if (!s->ws[w] || !dims_match(s->ws[w]->r, s->inst)) continue;
      std::vector<signed char> fx;
      if (!fix_from_results(s->inst, Y, &hT[(size_t)k * Y.istride], &s->ws[w]->r, fx)) continue;
      // repair root of the start (CPLEX: repairtries): its region binaries alone; the search completes the rest by rounding
      std::vector<signed char> rx(Y.fixlen, (signed char)-1);
      bool any_reg = false;
      for (int q = Y.f_reg; q < Y.f_reg + Y.C * Y.N; ++q) { rx[q] = fx[q]; any_reg = any_reg || fx[q] >= 0; }
      if ((int)R.size() < MAXR) R.push_back(std::move(fx));
      if (any_reg && (int)R.size() < MAXR) { R.push_back(std::move(rx)); inst_root_depth[k].resize(R.size(), 0); inst_root_depth[k].back() = REPAIR_ROOT; }
    }
    inst_root_depth[k].resize(R.size(), 0);
    int rows, bin, cont, nnz; raw_sizes(s->inst, rows, bin, cont, nnz);
    s->props = miqp_solution_properties_c{}; s->props.NrConstraints = rows; s->props.NrBinaryVariables = bin; s->props.NrFloatVariables = cont;
    s->props.NonZeroCoefficients = nnz;
  };
  { static const int prep_cap = KNOB_T("MIQP_PREP_THREADS") ? std::atoi(KNOB_T("MIQP_PREP_THREADS")) : 16;   // (the 51 200 instances of the driver's stream on a 2 x 64-core host: 0.38 / 0.44 / 0.57 / 0.76 s at 16 / 32 / 64 / 128 threads - allocator and first-touch contention, not arithmetic)
    const int nth = std::max(1, std::min<int>({n / 8, (int)std::thread::hardware_concurrency(), prep_cap}));
    if (nth <= 1) for (int k = 0; k < n; ++k) prepare_one(k);
    else {
      std::atomic<int> next{0}; std::vector<std::thread> th;
      for (int t = 0; t < nth; ++t) th.emplace_back([&] { for (int k = next.fetch_add(1); k < n; k = next.fetch_add(1)) prepare_one(k); });
      for (auto& t : th) t.join();
    } }
  for (int k = 0; k < n; ++k) {
    if (!h_feas0[k]) h_done[k] = 1;
    for (size_t q = 0; q < inst_roots[k].size(); ++q) add_root(k, inst_roots[k][q], inst_root_depth[k][q]);
    if (oc[k] > 0) active++; else h_done[k] = 1;   // (a rank of a tree split may own no root)
  }
  const double t_tables = wall_s() - t_enter - t_ctx;
  { double gmin = 1.0; for (int k = 0; k < n; ++k) gmin = std::min(gmin, h_gap[k]);
    const double ftol = KNOB_T("MIQP_QPTOL_F") ? std::atof(KNOB_T("MIQP_QPTOL_F")) : 1e-4;   // (tuning knob)
    const double tolcap = KNOB_T("MIQP_QPTOL") ? std::atof(KNOB_T("MIQP_QPTOL")) : QP_TOL;
    B.qp_tol = std::min(tolcap, std::max(1e-12, ftol * gmin)); }  // node relaxations: accurate to a small fraction of the MIP gap
  hipStream_t st = X.stream;
  HIP_OK(hipMemcpyAsync((void*)B.inst_d, hD, nD_ * 8, hipMemcpyHostToDevice, st));
  HIP_OK(hipMemcpyAsync((void*)B.inst_i, hT, nT_ * 4, hipMemcpyHostToDevice, st));
  HIP_OK(hipMemcpyAsync(B.pool_fix, roots.data(), roots.size(), hipMemcpyHostToDevice, st));
  int pool0 = nrec;
  if (B.pool_big && nrec > 0) HIP_OK(hipMemsetAsync(B.pool_big, 0, (size_t)nrec, st));   // the root records start unmarked (children are marked or cleared when they are written)
  HIP_OK(hipMemcpyAsync(B.pool_count, &pool0, 4, hipMemcpyHostToDevice, st));
  HIP_OK(hipMemsetAsync(B.free_head, 0, 4, st)); HIP_OK(hipMemsetAsync(B.free_tail, 0, 4, st)); HIP_OK(hipMemsetAsync(B.free_limit, 0, 4, st));
  HIP_OK(hipMemcpyAsync((void*)B.root_node, on.data(), on.size() * 4, hipMemcpyHostToDevice, st));
  HIP_OK(hipMemcpyAsync((void*)B.root_depth, od.data(), od.size() * 4, hipMemcpyHostToDevice, st));
  HIP_OK(hipMemcpyAsync((void*)B.root_cnt, oc.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
  HIP_OK(hipMemsetAsync(B.open_count, 0, (size_t)n * 4, st));
  HIP_OK(hipMemsetAsync(B.inst_slot, 0xFF, (size_t)n * 4, st)); HIP_OK(hipMemsetAsync(B.slot_inst, 0xFF, (size_t)NS * 4, st));
  HIP_OK(hipMemsetAsync(B.inst_kill, 0, (size_t)n * 4, st));
  HIP_OK(hipMemsetAsync(B.slot_demand, 0, (size_t)NS * 4, st));
  { std::vector<int> t0_(NS, npr); HIP_OK(hipMemcpyAsync(B.slot_take, t0_.data(), (size_t)NS * 4, hipMemcpyHostToDevice, st)); HIP_OK(hipStreamSynchronize(st)); }
  B.base_take = std::max(1, std::min(npr, KNOB_T("MIQP_BASE_TAKE") ? std::atoi(KNOB_T("MIQP_BASE_TAKE")) : 8));
  // Batch shares (share_kernel).  A QUEUE (more instances than slots: admissions go on while the old instances run against their limits): half
  // of the batch is shared evenly as a floor, the rest goes by admission order up to 256 nodes per instance - an instance that waits for its
  // turn no longer crawls at 8 nodes per round (its tree then costs 3-4 x the nodes, tools/crowd_probe.py), and narrow shares cost the fewest
  // nodes in all; measured on the 12-step bench stream: 1010 -> 1160 solves/s, 99.57 -> 99.39 % proven, p95 6.9 -> 4.5 s (the whole curve:
  // profiles/r04_share_policy.txt).  ONE batch with every instance in flight from the start (cfg4's 256, cfg5's 16): all deadlines are the same and
  // serving a few instances to their end frees the device for the others - admission order up to 1024 each, no floor (cfg5: 11 of 16 proven against 5).
  const bool queue_mode = NS < n;
  B.share_cap = std::max(1, KNOB_T("MIQP_SHARE_CAP") ? std::atoi(KNOB_T("MIQP_SHARE_CAP")) : (queue_mode ? 256 : 1024));
  B.floor_pct = std::max(0, std::min(100, KNOB_T("MIQP_FLOOR_PCT") ? std::atoi(KNOB_T("MIQP_FLOOR_PCT")) : (queue_mode ? 50 : 0)));
  B.lns_narrow = std::max(0, KNOB_T("MIQP_LNS_NARROW") ? std::atoi(KNOB_T("MIQP_LNS_NARROW")) : 512);   // width of a round that carries local-search leaves (0: as wide as any)
  B.defer_cap = KNOB_T("MIQP_DEFER") ? std::atoi(KNOB_T("MIQP_DEFER")) : (Y.C >= 3 ? 24 : 0);   // (see ipm_kernel.  Measured on cfg5 at 0 / 16 / 20 / 24: seed 8 0.96 / 0.85 / 0.78 / 0.77 s, seed 14 0.75 / 0.67 / 0.65 / 0.66 s, seed 15 2.47 / 2.62 / 2.39 / 2.42 s, node relaxations in seed 11's 10 s 2.22 / 2.25 / 2.43 / 2.49 M, sixteen in flight 3.33 / 3.71 / 3.42 / 3.49 M.  Two cars (both on-chip variants have the same exit): OFF - the bench on 8 steps at 0 / 16 / 18 / 20 / 24 / 28: 1160 / 1174 / 1176-1182 / 1172 / 1156 / 1160 solves/s (the standard launch 10.0 -> 9.2 ms but 6 % more nodes per instance), single solves p99 46 -> 44 ms; and cfg3 seed 1913 - the pinned hard instance of test_local_search_changes_the_order_not_the_answer - 135 k -> 327 k nodes: a result that arrives a round late reorders the local search's chains)
  B.probe_itcap0 = KNOB_T("MIQP_PROBE_ITCAP0") ? std::atoi(KNOB_T("MIQP_PROBE_ITCAP0")) : (Y.C >= 3 ? 0 : 40);   // (the cap while the instance has no incumbent)
  B.pump_max = std::max(0, std::min(15, KNOB_P("MIQP_PUMP") ? std::atoi(KNOB_P("MIQP_PUMP")) : 6));   // re-rounding of infeasible rounding probes (eval_kernel)
  B.pump_inc = KNOB_T("MIQP_PUMP_INC") ? std::atoi(KNOB_T("MIQP_PUMP_INC")) : (Y.C >= 3 ? 1 : 0);   // re-rounding also with an incumbent (probes whose OBJECTIVE is below it): three and four cars - cfg5 seed 15 proven in 8 s, the gaps of the two seeds left at 10 s with 16 in flight 0.38 / 0.32 -> 0.09 / 0.03; two cars: the probes it lets converge are the critical path of a round (single-solve p99 62 -> 72 ms, queue -1 %)
  B.young_nodes = std::max(0, KNOB_T("MIQP_YOUNG_NODES") ? std::atoi(KNOB_T("MIQP_YOUNG_NODES")) : 0);
  B.probe_room = KNOB_T("MIQP_PROBE_ROOM") ? std::atof(KNOB_T("MIQP_PROBE_ROOM")) : 0.0;
  B.live_inc = KNOB_T("MIQP_LIVE_INC") ? std::atoi(KNOB_T("MIQP_LIVE_INC")) : 0;
  B.probe_every = KNOB_T("MIQP_PROBE_EVERY") ? std::atoi(KNOB_T("MIQP_PROBE_EVERY")) : 1;
  B.probe_itcap = KNOB_T("MIQP_PROBE_ITCAP") ? std::atoi(KNOB_T("MIQP_PROBE_ITCAP")) : (Y.C >= 3 ? 0 : 24);   // (three and four cars: no cap - EVERY probe of the phase without incumbent hit it there, at 25 iterations: infeasible roundings take 36-48, and it is their least-violation solution that the re-rounding needs)   // (0: never; solved probes take 9-24 iterations; 40 until round 4: the heuristic nodes - probes, local-search leaves - are the critical path of a single solve's round: p99 99 -> 80 ms at 24)
  B.probe_margin = KNOB_T("MIQP_PROBE_MARGIN") ? std::atof(KNOB_T("MIQP_PROBE_MARGIN")) : 0.25;   // (0: every disjunction of a probe fixed, as in round 2)
  B.det_ties = KNOB_T("MIQP_DET_TIES") ? std::atoi(KNOB_T("MIQP_DET_TIES")) : 1;
  B.window_pct = std::max(1, std::min(100, KNOB_T("MIQP_WINDOW") ? std::atoi(KNOB_T("MIQP_WINDOW")) : 100));
  HIP_OK(hipMemsetAsync(B.far_count, 0, (size_t)n * 4, st)); HIP_OK(hipMemsetAsync(B.far_minkey, 0xFF, (size_t)n * 8, st));
  HIP_OK(hipMemsetAsync(B.inst_mode, 0, (size_t)n * 4, st));

  HIP_OK(hipMemsetAsync(B.inc_key, 0xFF, (size_t)n * 8, st));
  HIP_OK(hipMemsetAsync(B.inc_seen, 0xFF, (size_t)n * 8, st));
  HIP_OK(hipMemsetAsync(B.inc_fix, 0xFF, (size_t)n * Y.fixlen, st));   // no incumbent yet: every disjunction undecided
  if (B.inc_Mtag) HIP_OK(hipMemsetAsync(B.inc_Mtag, 0, (size_t)n * 8, st));
  B.lns_warm = !(KNOB_T("MIQP_LNS_WARM") && std::atoi(KNOB_T("MIQP_LNS_WARM")) == 0);
  { std::vector<double> big(n, 1e300); HIP_OK(hipMemcpyAsync(B.inc_obj, big.data(), n * 8, hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(B.inc_ext, big.data(), n * 8, hipMemcpyHostToDevice, st)); HIP_OK(hipMemcpyAsync(B.near_thr, big.data(), n * 8, hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(B.lower_bound, big.data(), n * 8, hipMemcpyHostToDevice, st)); HIP_OK(hipStreamSynchronize(st)); }
  HIP_OK(hipMemcpyAsync(B.inst_done, h_done.data(), n * 4, hipMemcpyHostToDevice, st));
  HIP_OK(hipMemsetAsync(B.inst_flags, 0, (size_t)n * 4, st));
  HIP_OK(hipMemcpyAsync(B.inst_gap, h_gap.data(), n * 8, hipMemcpyHostToDevice, st));
  HIP_OK(hipMemcpyAsync(B.inst_const, h_const.data(), n * 8, hipMemcpyHostToDevice, st));
  HIP_OK(hipMemsetAsync(B.inst_nodes, 0, (size_t)n * 8, st));
  HIP_OK(hipMemsetAsync(B.inst_iters, 0, (size_t)n * 8, st));
  HIP_OK(hipMemsetAsync(B.inst_ninc, 0, (size_t)n * 4, st));
  HIP_OK(hipMemsetAsync(B.as_stats, 0, 256, st));
  HIP_OK(hipMemsetAsync(B.inst_lns, 0, (size_t)n * 4, st));
  { std::vector<double> big_(n, 1e300); HIP_OK(hipMemcpyAsync(B.inst_lns_obj, big_.data(), (size_t)n * 8, hipMemcpyHostToDevice, st)); HIP_OK(hipStreamSynchronize(st)); }
  B.lns_step = KNOB_T("MIQP_LNS_STEP") ? std::atof(KNOB_T("MIQP_LNS_STEP")) : 0.0;
  // (read per call, not only when the device context is built: a context is reused by every later call of the same shape)
  B.cut_gate = KNOB_P("MIQP_CUT_GATE") ? std::atof(KNOB_P("MIQP_CUT_GATE")) : 1.0e-5;
  B.seq_kinds = KNOB_P("MIQP_SEQ_KINDS") ? (int)std::strtoul(KNOB_P("MIQP_SEQ_KINDS"), nullptr, 0) : (5 << 8);
  // a single solve sends its re-roundable rounding probes to the active-set launch first (the interior point - up to 40 iterations of 70 us on the
  // critical path of its round - only sees the ones that turn out infeasible, a round later): seeds 0-95 p50 5.0 -> 4.0 ms at gap 0.1, 5.7 -> 5.0 at
  // 0.01, p90 / p99 unchanged; a queue keeps the direct route (1898 against 1868 solves/s on the driver's stream)
  B.as_probe_first = KNOB_T("MIQP_AS_PROBE_FIRST") ? std::atoi(KNOB_T("MIQP_AS_PROBE_FIRST")) : (NS == 1 ? 1 : 0);
  { const char* e = KNOB_P("MIQP_AS"); X.as_on = X.as_cap && !(e && std::atoi(e) == 0);   // (per call, like the other search switches: a context is reused by later calls of the same shape)
    B.batch_A = X.as_on ? X.as_batch_A : nullptr; B.pool_A = X.as_on ? X.as_pool_A : nullptr; }
  B.lns_mode = KNOB_P("MIQP_LNS") ? std::atoi(KNOB_P("MIQP_LNS")) : 45;
#ifndef MIQP_TUNING
  B.lns_mode &= 127;   // (bits 7-9: the neighbourhood sub-problems of round 4 - measured without effect, DESIGN.md 2b - exist in a tuning build only)
#endif
  B.lns_min_nodes = KNOB_T("MIQP_LNS_MIN") ? std::atoi(KNOB_T("MIQP_LNS_MIN")) : (NS == 1 ? 500 : 2000);   // (a single solve: sooner - 90 % quantile of seeds 0-95 24 -> 19 ms; a queue at 500: 1 % slower)
  HIP_OK(hipMemsetAsync(B.active_insts, 0, 4, st));   // admit_kernel counts the instances in as they enter
  HIP_OK(hipMemsetAsync(B.stat_rowiters, 0, 8, st));
  HIP_OK(hipStreamSynchronize(st));

  // ---- rounds
  double t0 = wall_s();
  const double t_setup = t0 - t_enter;
  double tlim = 0; for (int k = 0; k < n; ++k) tlim = std::max(tlim, h_tlim[k]);
  HIP_OK(hipEventRecord(X.ev0, st));
  std::vector<int> h_done_now(h_done); std::vector<double> h_tdone(n, -1.0);
  size_t nev = 0; int rounds = 0, empty_rounds = 0; long long launched_nodes = 0;
  double sp_inc = 1e300, sp_lb = -1e300; int sp_owner = 0; bool sp_timeup = false, sp_finished = false;   // state of a tree split
  // ---- streaming admission (host side): which instance holds which slot, when it entered, who is next
  std::vector<int> h_slot_inst(NS, -1), h_kill(n, 0), h_pairs; std::vector<double> t_admit(n, 0.0);
  std::vector<char> h_stalled(n, 0);   // retired because it made no progress (not because its time was up)
  int next_q = 0, in_flight = 0;
  bool abandoned = false, stuck_once = false;   // the round loop was left with instances still queued or in flight (reported, never silent)
  // frees the slots of proven / retired instances and fills them from the queue; `sel`: the list buffer the next select reads
  auto admit = [&](double now, int sel) -> bool {
    h_pairs.clear();
    for (int sl = 0; sl < NS; ++sl) {
      const int k = h_slot_inst[sl];
      if (k >= 0 && !h_done_now[k]) continue;   // busy
      if (k >= 0) { in_flight--; h_slot_inst[sl] = -1; }
      while (next_q < n && h_done[next_q]) next_q++;   // instances that never start (infeasible first step, no root on this rank)
      if (next_q < n) { h_pairs.push_back(sl); h_pairs.push_back(next_q); h_slot_inst[sl] = next_q; t_admit[next_q] = now; next_q++; in_flight++; }
      else if (k >= 0) { h_pairs.push_back(sl); h_pairs.push_back(-1); }
    }
    if (!h_pairs.empty()) {
      const int np_ = (int)h_pairs.size() / 2;
      HIP_OK(hipMemcpyAsync(X.d_pairs, h_pairs.data(), h_pairs.size() * 4, hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(admit_kernel, dim3((np_ + 63) / 64), dim3(64), 0, st, B, (const int*)X.d_pairs, np_, sel);
      // (no synchronisation: the copy from pageable memory is staged before the call returns, the device buffer is protected by the stream order)
    }
    return true;
  };
  if (!admit(0.0, 0)) return fail_all(nullptr);
  // (Enqueueing the launches of a round before the host has read the selection's counters - "pipelined rounds" - was measured twice and
  // removed in round 5: no gain on queues (round 3: the time between the interior point launches is the selection and evaluation kernels, not
  // the host) and none on single solves (round 5: p99 58.3 -> 59.6 ms, profiles/r05_single_latency.txt))
  int prev_bc = 0;
  static const bool round_log = KNOB_P("MIQP_ROUND_LOG") != nullptr;   // diagnostic: the batch sizes of the rounds, printed after the solve (no extra synchronisation)
  std::vector<int> round_bc;
  // the counters of a round's launches (batch count, work counters, hand-over counts) come in two parity sets: a round uses one, its
  // roll_kernel zeroes the other for the round after - six 4-byte memsets per round less in the stream (0.65 ms of the 1.4 ms round of a single solve)
  const bool use_par = X.ctr && X.oc_grid > 0 && X.ocb_grid > 0 && X.concurrent_big && X.stream2 && X.probe_grid > 0 && !KNOB_T("MIQP_MEMSETS");
  if (use_par) HIP_OK(hipMemsetAsync(X.ctr, 0, 2 * CTR_SET * 4, st));
  const bool use_cls = use_par && X.as_on && X.cls_list && !(KNOB_T("MIQP_CLS_LISTS") && std::atoi(KNOB_T("MIQP_CLS_LISTS")) == 0);
  B.cls_list = use_cls ? X.cls_list : nullptr; B.cls_take = 0; X.cls_n[0] = X.cls_n[1] = X.cls_n[2] = -1;
  int width_now = adaptive_width ? width0 : 0, width_since = 0, demand_now = 0; unsigned long long width_key = ~0ull, kinc_now = ~0ull; double lb_now = -1e300;
  for (;;) {
    const int par = rounds & 1;
    B.width_cap = width_now;
    if (use_par) { B.batch_count = X.ctr + CTR_SET * par; B.cls_count = B.batch_count + 8; } else HIP_OK(hipMemsetAsync(B.batch_count, 0, 4, st));
    B.open_sel = rounds & 1;
    B.prev_bc = prev_bc;
    hipLaunchKernelGGL(select_kernel, dim3(NS), dim3(SEL_THREADS), 0, st, B, rounds);
    // (share_kernel - one workgroup, 0.23 ms: the shares of the NEXT round from this selection's demands - beside lns_kernel, 0.22 ms, on the second stream:
    // neither reads what the other writes)
    const bool share_aside = B.lns_mode > 0 && NS >= 64 && X.stream2 && X.ev_sfork && !KNOB_P("MIQP_DEBUG_SYNC");
    if (share_aside) { (void)hipEventRecord(X.ev_sfork, st); (void)hipStreamWaitEvent(X.stream2, X.ev_sfork, 0); hipLaunchKernelGGL(share_kernel, dim3(1), dim3(1024), 0, X.stream2, B); (void)hipEventRecord(X.ev_sjoin, X.stream2); }
    if (B.lns_mode > 0) hipLaunchKernelGGL(lns_kernel, dim3(NS), dim3(64), 0, st, B);   // the neighbours of new incumbents join this round's batch
    if (KNOB_P("MIQP_DEBUG_SYNC")) { hipError_t e_ = hipStreamSynchronize(st); std::fprintf(stderr, "[dbg] round %d select: %s\n", rounds, hipGetErrorString(e_)); }
    hipLaunchKernelGGL(roll_kernel, dim3(1), dim3(CTR_SET), 0, st, B, use_par ? X.ctr + CTR_SET * (par ^ 1) : (int*)nullptr);
    if (share_aside) (void)hipStreamWaitEvent(st, X.ev_sjoin, 0); else hipLaunchKernelGGL(share_kernel, dim3(1), dim3(1024), 0, st, B);
    int bc = 0; int hctr[CTR_SET] = {0};
    HIP_OK(hipMemcpyAsync(use_cls ? hctr : &bc, B.batch_count, use_cls ? CTR_SET * 4 : 4, hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(h_done_now.data(), B.inst_done, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    if (adaptive_width) { HIP_OK(hipMemcpyAsync(&kinc_now, B.inc_key, 8, hipMemcpyDeviceToHost, st)); HIP_OK(hipMemcpyAsync(&demand_now, B.slot_demand, 4, hipMemcpyDeviceToHost, st)); HIP_OK(hipMemcpyAsync(&lb_now, B.lower_bound, 8, hipMemcpyDeviceToHost, st)); }
    HIP_OK(hipStreamSynchronize(st));
    if (use_cls) { bc = hctr[0]; for (int q = 0; q < 3; ++q) X.cls_n[q] = std::min(hctr[8 + q], X.batch_cap); }
    if (adaptive_width) {
      if (kinc_now != width_key) { width_key = kinc_now; width_since = rounds; }
      else if (kinc_now < 0xFFF0000000000000ull && rounds - width_since >= 32 && demand_now >= 8 * width_now && width_now < width_max) {
        // ... and the incumbent is already close to the bound (within 5 %): with a poor incumbent a wide round solves what a better one would have pruned
        // (cfg5 seed 11 widened on a stale incumbent alone ended at a gap of 45 % instead of 2 %)
        const double io_ = host_key2d(kinc_now & ~0xFFFFFull);
        if (lb_now > -1e299 && io_ - lb_now <= 0.05 * std::fabs(io_)) { width_now *= 2; width_since = rounds; }
      }
    }
    const double tnow = wall_s() - t0;
    for (int sl = 0; sl < NS; ++sl) { const int k = h_slot_inst[sl]; if (k >= 0 && h_done_now[k] && h_tdone[k] < 0) h_tdone[k] = tnow - t_admit[k]; }   // time from admission to proof
    if (split) {   // once per round: the ranks agree on incumbent, bound and whether to go on (identical decisions everywhere)
      unsigned long long kinc = ~0ull; double lbl = 1e300;
      HIP_OK(hipMemcpy(&kinc, B.inc_key, 8, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(&lbl, B.lower_bound, 8, hipMemcpyDeviceToHost));
      const bool local_done = bc <= 0;
      unsigned long long w[4];
      w[0] = kinc >= 0xFFF0000000000000ull ? ~0ull : ((kinc & ~0xFFFFFull) | (unsigned long long)split->rank);
      w[1] = host_d2key(local_done && kinc >= 0xFFF0000000000000ull ? 1e300 : (local_done ? std::min(lbl, 1e300) : lbl));
      w[2] = local_done ? ~0ull : 0ull;
      w[3] = (wall_s() - t0 > tlim) ? 0ull : ~0ull;
      if (split->fn(split->user, 0, w, 4, 0) != 0) return fail_all("incumbent exchange failed");
      sp_inc = w[0] >= 0xFFF0000000000000ull ? 1e300 : host_key2d(w[0] & ~0xFFFFFull); sp_owner = (int)(w[0] & 0xFFFFFull);
      sp_lb = host_key2d(w[1]); sp_timeup = w[3] == 0ull;
      HIP_OK(hipMemcpyAsync(B.inc_ext, &sp_inc, 8, hipMemcpyHostToDevice, st));
      const bool all_done = w[2] == ~0ull;
      const bool gap_ok = sp_inc < 1e299 && (sp_inc - sp_lb) <= h_gap[0] * (1e-10 + std::fabs(sp_inc));
      if (all_done || sp_timeup || gap_ok) { sp_finished = all_done || gap_ok; break; }
      if (local_done) { rounds++; continue; }   // nothing left here: keep taking part in the exchange
    } else {
      // time limit per instance, counted from its admission: the instance is retired by the next select_kernel (its records
      // return to the pool) and reports TIME_LIM_* below
      bool any_kill = false;
      for (int sl = 0; sl < NS; ++sl) { const int k = h_slot_inst[sl]; if (k >= 0 && !h_done_now[k] && !h_kill[k] && tnow - t_admit[k] > h_tlim[k]) { h_kill[k] = 1; any_kill = true; } }
      if (any_kill) HIP_OK(hipMemcpyAsync(B.inst_kill, h_kill.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
      if (!admit(tnow, 1 - (rounds & 1))) return fail_all(nullptr);
      if (in_flight == 0 && next_q >= n) break;   // the queue is drained
      if (tnow > tlim * (double)((n + NS - 1) / NS + 1) + 30.0) { abandoned = true; break; }   // (safety net: no instance can outlive its limit by more than a round)
      if (bc <= 0) {   // a round may come up empty while a list tier is being reorganised, or right after admissions
        // 64 empty rounds in a row: the instances in flight are stuck (none of them is done, none has an open node to offer).  They are
        // retired like instances at their time limit - the slots go to the rest of the queue, which is NOT abandoned; a second such
        // streak with nothing admitted in between ends the call with an error
        if (++empty_rounds > 64) {
          if (stuck_once) { for (int sl = 0; sl < NS; ++sl) { const int k = h_slot_inst[sl]; if (k >= 0 && !h_done_now[k]) h_stalled[k] = 1; } abandoned = true; break; }   // (the instances in flight did not run out of time either)
          stuck_once = true; empty_rounds = 0;
          for (int sl = 0; sl < NS; ++sl) { const int k = h_slot_inst[sl]; if (k >= 0 && !h_done_now[k]) { h_kill[k] = 1; h_stalled[k] = 1; } }
          HIP_OK(hipMemcpyAsync(B.inst_kill, h_kill.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
          std::fprintf(stderr, "[miqp_gpu] %d instances in flight made no progress for 64 rounds: retired, the queue goes on\n", in_flight);
        }
        rounds++; prev_bc = 0; continue;
      }
      stuck_once = false;
      empty_rounds = 0;
    }
    if (bc > X.batch_cap) bc = X.batch_cap;
    if (KNOB_P("MIQP_TRACE") && bc > 0 && bc <= 256) {   // diagnostic: what the selection picked (list bound, depth word), the incumbent it pruned with, its mode
      std::vector<double> sb(bc); std::vector<int> sd(bc); double io_ = 0; int md_ = 0;
      HIP_OK(hipMemcpy(sb.data(), B.batch_bound, (size_t)bc * 8, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(sd.data(), B.batch_depth, (size_t)bc * 4, hipMemcpyDeviceToHost));
      HIP_OK(hipMemcpy(&io_, B.inc_obj, 8, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(&md_, B.inst_mode, 4, hipMemcpyDeviceToHost));
      std::vector<int> ord(bc); for (int k = 0; k < bc; ++k) ord[k] = k;
      std::sort(ord.begin(), ord.end(), [&](int a, int b) { return sd[a] != sd[b] ? sd[a] > sd[b] : sb[a] < sb[b]; });
      std::fprintf(stderr, "[sel] r%d mode %d incumbent %a:", rounds, md_, io_);
      for (int k : ord) std::fprintf(stderr, " %d.%d/%a", sd[k] >> 6, sd[k] & 63, sb[k]);
      std::fprintf(stderr, "\n");
    }
    if (X.ipm_ev.size() < nev + 2) { hipEvent_t a, b; HIP_OK(hipEventCreate(&a)); HIP_OK(hipEventCreate(&b)); X.ipm_ev.push_back(a); X.ipm_ev.push_back(b); }
    static const bool launch_trace = KNOB_T("MIQP_LAUNCH_TRACE") != nullptr;
    if (launch_trace && !X.ev_mid) HIP_OK(hipEventCreate(&X.ev_mid));
    if (X.std_ev.size() < nev / 2 + 1) { hipEvent_t a_; HIP_OK(hipEventCreate(&a_)); X.std_ev.push_back(a_); }
    HIP_OK(hipEventRecord(X.ipm_ev[nev], st));
    launch_ipm_batch(X, B, bc, st, true, use_par ? par : -1, X.std_ev[nev / 2]);
    HIP_OK(hipEventRecord(X.ipm_ev[nev + 1], st));
    if (launch_trace) {   // diagnostic: the two interior point launches of the round apart, and what the memory-backed one had to solve
      HIP_OK(hipStreamSynchronize(st));
      float m1 = 0, m2 = 0; HIP_OK(hipEventElapsedTime(&m1, X.ipm_ev[nev], X.ev_mid)); HIP_OK(hipEventElapsedTime(&m2, X.ev_mid, X.ipm_ev[nev + 1]));
      int oc = 0; HIP_OK(hipMemcpy(&oc, B.ovf_count, 4, hipMemcpyDeviceToHost));
      std::vector<int> ol(std::max(oc, 1)), hit(bc), hdw(bc), hok(bc);
      if (oc > 0) HIP_OK(hipMemcpy(ol.data(), B.ovf_list, (size_t)oc * 4, hipMemcpyDeviceToHost));
      HIP_OK(hipMemcpy(hit.data(), B.batch_it, (size_t)bc * 4, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(hdw.data(), B.batch_depth, (size_t)bc * 4, hipMemcpyDeviceToHost));
      HIP_OK(hipMemcpy(hok.data(), B.batch_ok, (size_t)bc * 4, hipMemcpyDeviceToHost));
      int np = 0, mxp = 0, mxo = 0, mxc = 0; long long sp = 0, so = 0, sc = 0;
      for (int q = 0; q < oc; ++q) { const int k = ol[q]; if ((hdw[k] & 63) == 63 && (hdw[k] >> 6) >= 1) { np++; sp += hit[k]; mxp = std::max(mxp, hit[k]); } else { so += hit[k]; mxo = std::max(mxo, hit[k]); } }
      std::vector<char> isov(bc, 0); for (int q = 0; q < oc; ++q) isov[ol[q]] = 1;
      for (int k = 0; k < bc; ++k) if (!isov[k]) { sc += hit[k]; mxc = std::max(mxc, hit[k]); }
      // (with the concurrent launches of the large nodes - the default - the first time is the standard launch incl. its wait for room, the
      // second the wait for the second stream after it, and the node split below is empty: nothing is handed on behind the standard launch;
      // MIQP_CONCURRENT_BIG=0 MIQP_OC_BIG=0 gives the two launches of the first half of round 3 apart)
      std::fprintf(stderr, "[launch] round %d nodes %d: on-chip %.2f ms (%d nodes, mean it %.1f, max %d); behind it %.2f ms: %d probes (mean it %.1f, max %d), %d others (mean it %.1f, max %d)\n",
                   rounds, bc, m1, bc - oc, (double)sc / std::max(1, bc - oc), mxc, m2, np, (double)sp / std::max(1, np), mxp, oc - np, (double)so / std::max(1, oc - np), mxo);
    }
    nev += 2;
    static const int replay_round = KNOB_T("MIQP_REPLAY_ROUND") ? std::atoi(KNOB_T("MIQP_REPLAY_ROUND")) : 0;
    if (bc >= X.batch_cap / 2 && rounds >= replay_round) {   // MIQP_REPLAY=k (diagnostic): the first batch that is at least half full is solved k more times under a timer - the kernels
      static int replay = KNOB_T("MIQP_REPLAY") ? std::atoi(KNOB_T("MIQP_REPLAY")) : 0;   // only read and write batch slots
      if (replay > 0) {
        hipEvent_t e0, e1; HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
        HIP_OK(hipEventRecord(e0, st));
        for (int r = 0; r < replay; ++r) launch_ipm_batch(X, B, bc, st);
        HIP_OK(hipEventRecord(e1, st)); HIP_OK(hipStreamSynchronize(st));
        float ms = 0; HIP_OK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<int> its(bc); HIP_OK(hipMemcpy(its.data(), B.batch_it, (size_t)bc * 4, hipMemcpyDeviceToHost));
        long long tot = 0; for (int v : its) tot += v;
        std::fprintf(stderr, "[miqp_gpu replay] %d nodes, %lld node-iterations: %.3f ms per pass, %.1f ns per node-iteration\n", bc, tot, ms / replay, 1e6 * ms / replay / (double)tot);
#ifdef MIQP_ABLATE
        if (X.oc_grid > 0 && Y.C == 2) {   // cost map of the on-chip kernel: the same batch, 15 iterations per node, parts switched off
          const size_t l_oc = (size_t)oc_lds_layout(Y.N, Y.fixlen).total;
          auto run = [&](auto kern, int mask) {
            HIP_OK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l_oc));
            auto once = [&] { (void)hipMemsetAsync(B.work_counter, 0, 4, st); (void)hipMemsetAsync(B.ovf_count, 0, 4, st); hipLaunchKernelGGL(kern, dim3(std::min(bc, X.oc_grid)), dim3(64), l_oc, st, B); };
            once(); HIP_OK(hipEventRecord(e0, st)); for (int r = 0; r < 3; ++r) once(); HIP_OK(hipEventRecord(e1, st)); HIP_OK(hipStreamSynchronize(st));
            float m2 = 0; HIP_OK(hipEventElapsedTime(&m2, e0, e1));
            std::fprintf(stderr, "[miqp_gpu ablate] mask %4d: %.3f ms per pass of %d nodes x 15 iterations = %.1f ns per node-iteration\n", mask, m2 / 3, bc, 1e6 * m2 / 3 / (bc * 15.0));
            return true;
          };
          run(ipm_onchip_kernel<2, OC_NSL, 1>, 1); run(ipm_onchip_kernel<2, OC_NSL, 3>, 3); run(ipm_onchip_kernel<2, OC_NSL, 5>, 5); run(ipm_onchip_kernel<2, OC_NSL, 9>, 9);
          run(ipm_onchip_kernel<2, OC_NSL, 17>, 17); run(ipm_onchip_kernel<2, OC_NSL, 33>, 33); run(ipm_onchip_kernel<2, OC_NSL, 65>, 65); run(ipm_onchip_kernel<2, OC_NSL, 129>, 129);
          run(ipm_onchip_kernel<2, OC_NSL, 257>, 257); run(ipm_onchip_kernel<2, OC_NSL, 513>, 513); run(ipm_onchip_kernel<2, OC_NSL, 1023>, 1023);
          run(ipm_onchip_kernel<2, OC_NSL, 1025>, 1025);   // the MFMA form of P [A B], [A B]' T on the model's column order
          launch_ipm_batch(X, B, bc, st);   // the replays clobbered the batch results: solve the real batch again
        }
#endif
        replay = 0;
      }
    }
    if (KNOB_P("MIQP_DEBUG_SYNC")) { hipError_t e_ = hipStreamSynchronize(st); std::fprintf(stderr, "[dbg] round %d ipm (%d nodes): %s\n", rounds, bc, hipGetErrorString(e_)); }
    if (KNOB_P("MIQP_TRACE")) {   // diagnostic: the solved batch of the round in an order that does not depend on the batch slots, for diffing two runs
      HIP_OK(hipStreamSynchronize(st));
      if (X.stream2) HIP_OK(hipStreamSynchronize(X.stream2));
      std::vector<int> hd(bc), hi(bc), hk(bc); std::vector<double> ho(bc), hb(bc), hv(bc);
      HIP_OK(hipMemcpy(hd.data(), B.batch_depth, bc * 4, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(hi.data(), B.batch_it, bc * 4, hipMemcpyDeviceToHost));
      HIP_OK(hipMemcpy(hk.data(), B.batch_ok, bc * 4, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(ho.data(), B.batch_obj, bc * 8, hipMemcpyDeviceToHost));
      HIP_OK(hipMemcpy(hb.data(), B.batch_bound, bc * 8, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(hv.data(), B.batch_viol, bc * 8, hipMemcpyDeviceToHost));
      std::vector<int> ord(bc); for (int k = 0; k < bc; ++k) ord[k] = k;
      std::sort(ord.begin(), ord.end(), [&](int a, int b) { return hd[a] != hd[b] ? hd[a] < hd[b] : (ho[a] != ho[b] ? ho[a] < ho[b] : hi[a] < hi[b]); });
      { int oc_ = 0, fc_ = 0, tk_ = 0, dm_ = 0; double nt_ = 0;
        HIP_OK(hipMemcpy(&oc_, B.open_count, 4, hipMemcpyDeviceToHost)); if (B.far_cap > 0) HIP_OK(hipMemcpy(&fc_, B.far_count, 4, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(&nt_, B.near_thr, 8, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(&tk_, B.slot_take, 4, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(&dm_, B.slot_demand, 4, hipMemcpyDeviceToHost));
        std::fprintf(stderr, "[lists] r%d batch %d: near list after selection %d, far tier %d, near_thr %g, share of the next round %d, demand %d\n", rounds, bc, oc_, fc_, nt_, tk_, dm_); }
      for (int k : ord) std::fprintf(stderr, "[trace] r%d depth %d.%d ok %d it %d obj %a bound %a viol %.3e\n", rounds, hd[k] >> 6, hd[k] & 63, hk[k], hi[k], ho[k], hb[k], hv[k]);
    }
    { DevBuf Be = B; Be.open_sel = 1 - (rounds & 1); launch_eval_c(Y.C, Be, bc, l_eval, st); }
    if (KNOB_P("MIQP_DEBUG_SYNC")) { hipError_t e_ = hipStreamSynchronize(st); std::fprintf(stderr, "[dbg] round %d eval: %s\n", rounds, hipGetErrorString(e_)); }
    launched_nodes += bc; rounds++; prev_bc = bc;
    if (round_log) round_bc.push_back(bc);
    if (O0.verbose > 1) std::fprintf(stderr, "[miqp_gpu] round %d: %d nodes\n", rounds, bc);
    if (O0.verbose == 1 && rounds % 25 == 0) {  // progress of the first instance
      double lb0 = 0, io0 = 0; int oc0 = 0; unsigned long long k0 = 0;
      HIP_OK(hipMemcpyAsync(&lb0, B.lower_bound, 8, hipMemcpyDeviceToHost, st)); HIP_OK(hipMemcpyAsync(&io0, B.inc_obj, 8, hipMemcpyDeviceToHost, st));
      HIP_OK(hipMemcpyAsync(&oc0, B.open_count, 4, hipMemcpyDeviceToHost, st)); HIP_OK(hipMemcpyAsync(&k0, B.inc_key, 8, hipMemcpyDeviceToHost, st));
      int fc0 = 0; HIP_OK(hipMemcpyAsync(&fc0, B.far_count, 4, hipMemcpyDeviceToHost, st));
      HIP_OK(hipStreamSynchronize(st));
      std::fprintf(stderr, "[miqp_gpu] t %.2f s round %d nodes %lld: instance 0 bound %.4f incumbent %.4f open %d + %d\n", wall_s() - t0, rounds, launched_nodes, lb0, k0 >= 0xFFF0000000000000ull ? INFINITY : io0 + h_const[0], oc0, fc0);
    }
  }
  if (const char* dp = KNOB_T("MIQP_DUMP_OPEN")) {   // diagnostic: open list of instance 0 (bound, depth, fix record of the 400 lowest)
    int oc0 = 0; HIP_OK(hipMemcpy(&oc0, B.open_count, 4, hipMemcpyDeviceToHost)); oc0 = std::min(oc0, open_cap);
    const size_t src = ((size_t)(rounds & 1) * NS + 0) * open_cap;
    std::vector<double> hb(oc0); std::vector<int> hn(oc0), hd(oc0);
    if (oc0 > 0) { HIP_OK(hipMemcpy(hb.data(), B.open_bound + src, (size_t)oc0 * 8, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(hn.data(), B.open_node + src, (size_t)oc0 * 4, hipMemcpyDeviceToHost));
      HIP_OK(hipMemcpy(hd.data(), B.open_depth + src, (size_t)oc0 * 4, hipMemcpyDeviceToHost)); }
    std::vector<int> ord(oc0); for (int k = 0; k < oc0; ++k) ord[k] = k;
    std::sort(ord.begin(), ord.end(), [&](int x, int y) { return hb[x] < hb[y]; });
    if (FILE* f = std::fopen(dp, "w")) {
      std::fprintf(f, "%d %d %d %d %d %d %d\n", oc0, Y.fixlen, Y.f_reg, Y.f_env, Y.f_obs, Y.f_c2c, Y.N);
      std::vector<signed char> rec(Y.fixlen);
      for (int q = 0; q < std::min(oc0, 400); ++q) {
        int k = ord[q]; HIP_OK(hipMemcpy(rec.data(), B.pool_fix + (size_t)hn[k] * Y.fixlen, Y.fixlen, hipMemcpyDeviceToHost));
        std::fprintf(f, "%.9g %d", hb[k] + h_const[0], hd[k] >> 6); for (int x = 0; x < Y.fixlen; ++x) std::fprintf(f, " %d", (int)rec[x]); std::fprintf(f, "\n");
      }
      std::fclose(f);
    }
  }
  // ---- polish: the incumbent of every instance is re-solved (all disjunctions fixed as completed) to a tight
  //      tolerance; its objective and states are what the caller receives
  std::vector<double> h_pobj(n, 0.0), h_pviol(n, 1.0); std::vector<int> h_pok(n, 0), h_pit(n, 0);
  {
    HIP_OK(hipMemcpyAsync(B.pool_fix, B.inc_fix, (size_t)n * Y.fixlen, hipMemcpyDeviceToDevice, st));
    std::vector<int> ids(n); for (int k = 0; k < n; ++k) ids[k] = k;
    HIP_OK(hipMemcpyAsync(B.batch_node, ids.data(), n * 4, hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(B.batch_inst, ids.data(), n * 4, hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(B.batch_count, &n, 4, hipMemcpyHostToDevice, st));
    HIP_OK(hipStreamSynchronize(st));
    DevBuf Bp = B; Bp.qp_tol = QP_TOL_FINAL; Bp.use_cutoff = 0; Bp.batch_cap = X.batch_alloc;
    // the polish starts at the incumbent's own solution (integer feasible: every row of the completed record holds there), centred
    // at a small complementarity - a third of the iterations of a cold solve to 1e-13 (the last thing a single solve waits for)
    Bp.ws_on = (B.ws_on && !KNOB_T("MIQP_POLISH_COLD")) ? 2 : 0; Bp.ws_mu = KNOB_T("MIQP_POLISH_MU") ? std::atof(KNOB_T("MIQP_POLISH_MU")) : 1.0e-2; Bp.ws_delta = KNOB_T("MIQP_POLISH_DELTA") ? std::atof(KNOB_T("MIQP_POLISH_DELTA")) : 1.0e-4;
    int nb = std::min(n, X.batch_alloc);
    launch_ipm_batch(X, Bp, nb, st);
    HIP_OK(hipMemcpyAsync(h_pobj.data(), B.batch_obj, nb * 8, hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(h_pviol.data(), B.batch_viol, nb * 8, hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(h_pok.data(), B.batch_ok, nb * 4, hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(h_pit.data(), B.batch_it, nb * 4, hipMemcpyDeviceToHost, st));
  }
  HIP_OK(hipEventRecord(X.ev1, st));
  HIP_OK(hipStreamSynchronize(st));
  HIP_OK(hipGetLastError());
  float ms_all = 0; HIP_OK(hipEventElapsedTime(&ms_all, X.ev0, X.ev1));
#ifdef MIQP_PROFILE
  { unsigned long long pf[64]; HIP_OK(hipMemcpy(pf, B.prof + 64, 64 * 8, hipMemcpyDeviceToHost));
    const char* nm[10] = {"decode", "rowpass", "bw.phi", "bw.TS+p", "bw.readlane+LDL", "bw.Ksolve", "bw.update", "forward", "step", "update"};
    double tot = 0; for (int q = 0; q < 10; ++q) tot += (double)pf[q];
    { unsigned long long po[16]; HIP_OK(hipMemcpy(po, B.prof, 16 * 8, hipMemcpyDeviceToHost));
      const char* no[9] = {"build", "bw.rows/assemble", "bw.mfma", "bw.TS", "bw.cholK", "bw.P", "forward", "step", "update"};
      double to = 0; for (int q = 0; q < 9; ++q) to += (double)po[q];
      if (po[10]) { std::fprintf(stderr, "[miqp_gpu profile] memory-backed kernel nodes %llu iters %llu cycles/node-iter %.0f :", po[10], po[9], to / std::max(1ull, po[9]));
        for (int q = 0; q < 9; ++q) std::fprintf(stderr, " %s %.1f%% (%.0f)", no[q], 100.0 * po[q] / to, (double)po[q] / std::max(1ull, po[9]));
        std::fprintf(stderr, "; inside bw.TS, the stage-Hessian chain: weights + staging %.0f, single-entry rows %.0f, MFMA loop %.0f, reductions + diagonal %.0f", (double)po[12] / std::max(1ull, po[9]), (double)po[13] / std::max(1ull, po[9]), (double)po[14] / std::max(1ull, po[9]), (double)po[15] / std::max(1ull, po[9]));
        std::fprintf(stderr, "\n"); } }
    std::fprintf(stderr, "[miqp_gpu profile] on-chip nodes %llu iters %llu cycles/node-iter %.0f :", pf[11], pf[10], tot / std::max(1ull, pf[10]));
    for (int q = 0; q < 10; ++q) std::fprintf(stderr, " %s %.1f%% (%.0f)", nm[q], 100.0 * pf[q] / tot, (double)pf[q] / std::max(1ull, pf[10]));
    std::fprintf(stderr, "\n");
    { unsigned long long pa[14]; HIP_OK(hipMemcpy(pa, B.prof + 80, 14 * 8, hipMemcpyDeviceToHost));
      const char* na[9] = {"decode", "gains + first iterate", "scan", "response of the row", "q", "directions + ratio test + M update", "iterate refresh", "results", "warm start"};
      double ta = 0; for (int q = 0; q < 9; ++q) ta += (double)pa[q];
      if (pa[10]) { std::fprintf(stderr, "[miqp_gpu profile] active-set kernel nodes %llu steps %llu cycles/node %.0f :", pa[10], pa[11], ta / (double)pa[10]);
        for (int q = 0; q < 9; ++q) std::fprintf(stderr, " %s %.1f%% (%.0f)", na[q], 100.0 * pa[q] / ta, (double)pa[q] / (double)pa[10]);
        std::fprintf(stderr, "; inside the decode: bound classes %.0f, general-row pass %.0f", (double)pa[12] / (double)pa[10], (double)pa[13] / (double)pa[10]);
        std::fprintf(stderr, "\n");
        unsigned long long pw[3], pl[6]; HIP_OK(hipMemcpy(pw, B.prof + 94, 3 * 8, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(pl, B.prof + 120, 6 * 8, hipMemcpyDeviceToHost));
        if (pw[2] && pw[1] && pl[5]) std::fprintf(stderr, "[miqp_gpu profile] standard active-set launches: %llu wavefronts resident %.0f shader cycles = %.1f us each (shader clock %.0f MHz); per launch (%llu folded): span %.3f ms, from the first wavefront out of work to the last end %.3f ms, wavefronts %.0f, resident wavefront time / (span x wavefronts) %.3f\n",
          pw[2], (double)pw[0] / pw[2], (double)pw[1] / pw[2] / 100.0, 100.0 * (double)pw[0] / (double)pw[1], pl[5], (double)pl[3] / pl[5] / 1e5, (double)pl[4] / pl[5] / 1e5, (double)pw[2] / pl[5], ((double)pw[1]) / ((double)pl[3] * ((double)pw[2] / pl[5])));
        unsigned long long ph[16]; HIP_OK(hipMemcpy(ph, B.prof + 128, 16 * 8, hipMemcpyDeviceToHost));
        if (pw[2]) { std::fprintf(stderr, "[miqp_gpu profile] standard active-set wavefronts by their start after the launch's first (< 0.05 / 0.2 / 0.5 / 1 / 2 / 4 / 8 ms / later), %% :");
          for (int q = 0; q < 8; ++q) std::fprintf(stderr, " %.1f", 100.0 * ph[q] / pw[2]);
          std::fprintf(stderr, "; by the nodes they solved (0 / <= 4 / <= 16 / <= 32 / <= 64 / more), %% :");
          for (int q = 8; q < 14; ++q) std::fprintf(stderr, " %.1f", 100.0 * ph[q] / pw[2]);
          std::fprintf(stderr, "\n"); }
        if (const char* wd = KNOB_T("MIQP_WAVE_DUMP")) {
          std::vector<unsigned long long> w(4 * 4 * 4096); HIP_OK(hipMemcpy(w.data(), B.prof + 160, w.size() * 8, hipMemcpyDeviceToHost));
          if (FILE* f = std::fopen(wd, "w")) { for (int q = 0; q < 4 * 4096; ++q) if (w[4 * q]) std::fprintf(f, "%d %llu %llu %u %u %llu %d\n", q & 4095, w[4 * q], w[4 * q + 1], (unsigned)(w[4 * q + 2] >> 32), (unsigned)w[4 * q + 2], w[4 * q + 3], q >> 12); std::fclose(f); } } } }
    { unsigned long long pe[8]; HIP_OK(hipMemcpy(pe, B.prof + 100, 8 * 8, hipMemcpyDeviceToHost));
      const char* ne[6] = {"load", "regions", "leaf disjunctions", "branching", "lifting + reservation", "records"};
      double te = 0; for (int q = 0; q < 6; ++q) te += (double)pe[q];
      if (pe[6]) { std::fprintf(stderr, "[miqp_gpu profile] eval_kernel, %llu branched nodes, cycles/node %.0f :", pe[6], te / (double)pe[6]);
        for (int q = 0; q < 6; ++q) std::fprintf(stderr, " %s %.1f%% (%.0f)", ne[q], 100.0 * pe[q] / te, (double)pe[q] / (double)pe[6]);
        std::fprintf(stderr, "\n"); } }
    { unsigned long long ps[10]; HIP_OK(hipMemcpy(ps, B.prof + 110, 10 * 8, hipMemcpyDeviceToHost));
      const char* ns[8] = {"incumbent copy / kill", "setup", "pass 1 (prune, keys)", "far refill", "bound reduce + spill select", "focus + window + share", "radix select + ties", "pass 3 (emit, compact)"};
      double ts = 0; for (int q = 0; q < 8; ++q) ts += (double)ps[q];
      if (ps[8]) { std::fprintf(stderr, "[miqp_gpu profile] select_kernel, %llu workgroups that reached the end (thread 0's clock), mean list %.0f entries, cycles each %.0f :", ps[8], (double)ps[9] / (double)ps[8], ts / (double)ps[8]);
        for (int q = 0; q < 8; ++q) std::fprintf(stderr, " %s %.1f%% (%.0f)", ns[q], 100.0 * ps[q] / ts, (double)ps[q] / (double)ps[8]);
        std::fprintf(stderr, "\n"); } }
    HIP_OK(hipMemset(B.prof, 0, 160 * 8)); }
#endif
  if (B.stats) {
    unsigned long long hs[256]; HIP_OK(hipMemcpy(hs, B.stats, sizeof(hs), hipMemcpyDeviceToHost)); HIP_OK(hipMemset(B.stats, 0, sizeof(hs)));
    const double nn_ = (double)std::max(1ull, hs[0]);
    std::fprintf(stderr, "[miqp_gpu stats] on-chip nodes %llu (general rows %.1f, coefficients %.1f, box keys %.1f, iterations %.1f per node), handed over %llu; general rows / 32 histogram:", hs[0], hs[1] / nn_, hs[4] / nn_, hs[2] / nn_, hs[5] / nn_, hs[3]);
    for (int q = 0; q < 16; ++q) std::fprintf(stderr, " %llu", hs[8 + q]);
    std::fprintf(stderr, "\n");
    std::fprintf(stderr, "[miqp_gpu stats] node outcomes: infeasible %llu (%.1f it), cut off %llu (%.1f it), not converged %llu (%.1f it), solved %llu (%.1f it) of which: bound >= incumbent %llu, within gap %llu, integer feasible %llu, branched %llu (%.2f children; by kind region/env/obstacle/car-car: %llu x %.1f, %llu x %.1f, %llu x %.1f, %llu x %.1f)\n",
                 hs[32], hs[36] / (double)std::max(1ull, hs[32]), hs[33], hs[37] / (double)std::max(1ull, hs[33]), hs[34], hs[38] / (double)std::max(1ull, hs[34]), hs[35], hs[39] / (double)std::max(1ull, hs[35]),
                 hs[40], hs[41], hs[42], hs[43], hs[44] / (double)std::max(1ull, hs[43]), hs[48], hs[52] / (double)std::max(1ull, hs[48]), hs[49], hs[53] / (double)std::max(1ull, hs[49]),
                 hs[50], hs[54] / (double)std::max(1ull, hs[50]), hs[51], hs[55] / (double)std::max(1ull, hs[51]));
    std::fprintf(stderr, "[miqp_gpu stats] region sets: tightened at %llu (car, step) sites, %llu nodes closed because a step had no region left; children not created because of their lifted bound %llu (multi-row lift larger than the single-row one: %llu), car/car sets tightened at %llu groups\n", hs[58], hs[59], hs[56], hs[57], hs[61]);
    std::fprintf(stderr, "[miqp_gpu stats] region branchings flagged by: own rows %llu (worst class acc box %llu, jerk box %llu, sector %llu, half-plane %llu, curvature %llu, slow square %llu), environment front rows %llu, obstacle front rows %llu, car/car front rows %llu; by step:",
                 hs[64], hs[70], hs[71], hs[72], hs[73], hs[74], hs[75], hs[65], hs[66], hs[67]);
    for (int q = 0; q < 32 && q < Y.N; ++q) std::fprintf(stderr, " %llu", hs[160 + q]);
    std::fprintf(stderr, "\n[miqp_gpu stats] node outcomes by origin (processed: infeasible / cut off / not converged / solved):");
    const char* on_[16] = {"root|reg-ref", "reg-adjacent", "reg-other", "reg-slow", "env-ref", "env-other", "-", "-", "obs-ref", "obs-other", "-", "-", "c2c-ref", "c2c-other", "-", "probe"};
    for (int q = 0; q < 16; ++q) if (hs[80 + q]) std::fprintf(stderr, " %s %llu: %llu / %llu / %llu / %llu;", on_[q], hs[80 + q], hs[96 + q], hs[112 + q], hs[128 + q], hs[144 + q]);
    std::fprintf(stderr, "\n[miqp_gpu stats] infeasible rounding probes re-rounded: %llu", hs[62]);
    std::fprintf(stderr, "\n[miqp_gpu stats] handed-over nodes with >= 480 general rows: %llu, mean %.0f, most %llu general rows", hs[7], (double)hs[6] / std::max(1ull, hs[7]), hs[5]);
    std::fprintf(stderr, "\n[miqp_gpu stats] rounding probes by iterations / 3 (0-2, 3-5, ..., 45+):");
    { const char* oc_n[4] = {"infeasible", "cut off", "not converged", "solved"};
      for (int o = 0; o < 4; ++o) { std::fprintf(stderr, " %s", oc_n[o]); for (int q = 0; q < 16; ++q) std::fprintf(stderr, " %llu", hs[192 + 16 * o + q]); std::fprintf(stderr, ";"); } }
    std::fprintf(stderr, "\n");
  }
  if (round_log) { std::fprintf(stderr, "[rounds]"); for (int v : round_bc) std::fprintf(stderr, " %d", v); std::fprintf(stderr, "\n"); }
  double ms_ipm = 0;
  for (size_t e = 0; e + 1 < nev; e += 2) { float ms = 0; HIP_OK(hipEventElapsedTime(&ms, X.ipm_ev[e], X.ipm_ev[e + 1])); ms_ipm += ms; }
  double ms_std = 0; const bool std_timed = use_par;   // (the event is recorded in the concurrent launch path only)
  if (std_timed) for (size_t e = 0; e + 1 < nev; e += 2) { float ms = 0; if (hipEventElapsedTime(&ms, X.ipm_ev[e], X.std_ev[e / 2]) == hipSuccess) ms_std += ms; }
  double t_solve = wall_s() - t0;

  // ---- results
  std::vector<double> h_inc(n), h_lb(n); std::vector<int> h_flags(n), h_ninc(n), h_oc(n), h_dn(n); std::vector<long long> h_nodes(n), h_iters(n);
  std::vector<unsigned long long> h_key(n);
  unsigned long long rowiters = 0;
  HIP_OK(hipMemcpy(h_inc.data(), B.inc_obj, n * 8, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(h_key.data(), B.inc_key, n * 8, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(h_lb.data(), B.lower_bound, n * 8, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(h_flags.data(), B.inst_flags, n * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(h_ninc.data(), B.inst_ninc, n * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(h_oc.data(), B.open_count, n * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(h_dn.data(), B.inst_done, n * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(h_nodes.data(), B.inst_nodes, n * 8, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(h_iters.data(), B.inst_iters, n * 8, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(&rowiters, B.stat_rowiters, 8, hipMemcpyDeviceToHost));
  unsigned long long h_as[32] = {0};
  HIP_OK(hipMemcpy(h_as, B.as_stats, 256, hipMemcpyDeviceToHost));
  if (B.stats && h_as[0] + h_as[2] > 0)
    std::fprintf(stderr, "[miqp_gpu stats] active-set launch: %llu nodes (%.1f steps, %.1f drops, %.1f rows from the parent's active set, %.1f active rows at the end per node; %llu infeasible, %llu cut off; %llu started from the parent's M, %llu fell back to a cold start), %llu handed to the interior point (no free slot %llu, step cap %llu, curvature %llu, down-date pivot %llu, rows off their equalities %llu, other %llu); M rebuilt because: the parent left none %llu, the ring had come round %llu, another row count %llu; in the larger block %llu nodes (%.1f steps), of them leaves of the local search %llu (%.1f steps), rounding probes %llu; by their general rows (<= 64 / 96 / 128 / 192 / more) %llu / %llu / %llu / %llu / %llu\n",
                 h_as[0], h_as[1] / (double)std::max(1ull, h_as[0]), h_as[3] / (double)std::max(1ull, h_as[0]), h_as[7] / (double)std::max(1ull, h_as[0]), h_as[6] / (double)std::max(1ull, h_as[0]), h_as[4], h_as[5], h_as[8], h_as[9], h_as[2], h_as[11], h_as[12], h_as[13], h_as[14], h_as[15], h_as[10], h_as[17], h_as[18], h_as[19], h_as[20], h_as[21] / (double)std::max(1ull, h_as[20]), h_as[22], h_as[23] / (double)std::max(1ull, h_as[22]), h_as[29], h_as[24], h_as[25], h_as[26], h_as[27], h_as[28]);
  std::vector<signed char> h_fix((size_t)n * Y.fixlen); std::vector<double> h_Z((size_t)n * Y.N * Y.nz);
  HIP_OK(hipMemcpy(h_fix.data(), B.inc_fix, h_fix.size(), hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(h_Z.data(), B.inc_Z, h_Z.size() * 8, hipMemcpyDeviceToHost));
  std::vector<double> h_pZ((size_t)n * Y.N * Y.nz);
  HIP_OK(hipMemcpy(h_pZ.data(), B.batch_Z, h_pZ.size() * 8, hipMemcpyDeviceToHost));
  for (int k = 0; k < n; ++k) {
    if (!(h_inc[k] < 1e299)) continue;
    if (h_pviol[k] <= FEAS_TOL && (h_pok[k] || h_pit[k] >= 10)) {
      int nign = 0; for (int q = 0; q < Y.C * Y.O * Y.N * 5; ++q) nign += h_fix[(size_t)k * Y.fixlen + Y.f_obs + q] >= Y.L;
      h_inc[k] = h_pobj[k] + h_const[k] + nign * S[k]->inst.w_slack_obs;
      std::copy(h_pZ.begin() + (size_t)k * Y.N * Y.nz, h_pZ.begin() + (size_t)(k + 1) * Y.N * Y.nz, h_Z.begin() + (size_t)k * Y.N * Y.nz);
    }
  }
  if (split) {
    // the owner of the best incumbent hands its polished solution to every rank: {objective, states, completed record}
    std::vector<unsigned char> buf(8 + h_Z.size() * 8 + h_fix.size());
    unsigned long long w[2] = {h_inc[0] < 1e299 ? ((host_d2key(h_inc[0]) & ~0xFFFFFull) | (unsigned long long)split->rank) : ~0ull, ~0ull};
    if (split->fn(split->user, 0, w, 2, 0) != 0) return fail_all("incumbent exchange failed");
    if (w[0] < 0xFFF0000000000000ull) {
      const int owner = (int)(w[0] & 0xFFFFFull);
      if (owner == split->rank) { std::memcpy(buf.data(), &h_inc[0], 8); std::memcpy(buf.data() + 8, h_Z.data(), h_Z.size() * 8); std::memcpy(buf.data() + 8 + h_Z.size() * 8, h_fix.data(), h_fix.size()); }
      if (split->fn(split->user, 1, buf.data(), (int)buf.size(), owner) != 0) return fail_all("solution broadcast failed");
      std::memcpy(&h_inc[0], buf.data(), 8); std::memcpy(h_Z.data(), buf.data() + 8, h_Z.size() * 8); std::memcpy(h_fix.data(), buf.data() + 8 + h_Z.size() * 8, h_fix.size());
      h_lb[0] = std::min(sp_lb, h_inc[0]);
    }
    h_flags[0] = sp_finished ? 0 : 1; h_oc[0] = 0; h_dn[0] = 1;   // the verdict of the whole job, the same on every rank
  }
  long long tot_iters = 0; for (int k = 0; k < n; ++k) tot_iters += h_iters[k];
  for (int k = 0; k < n; ++k) {
    miqp_solver* s = S[k];
    bool have = h_inc[k] < 1e299;
    bool unfinished = (h_flags[k] & 3) || h_oc[k] > 0 || !h_dn[k];   // bit 0: a list or the record pool overflowed, bit 1: retired at its time limit
    if (!split && k >= next_q && !h_done[k]) {   // never admitted (the round loop was abandoned): the solver did not run on it - not a time-limit verdict
      s->status = MIQP_STATUS_FAILED_SEG_FAULT; s->has_sol = false; s->props.objective = NAN; s->props.gap = NAN; s->props.best_bound = NAN; s->props.status = 0; s->props.time = 0.0;
      s->err = "the queue was abandoned before this instance was admitted";
      statuses[k] = s->status; continue;
    }
    s->props.time = h_tdone[k] >= 0 ? h_tdone[k] : std::max(0.0, t_solve - t_admit[k]);   // from the instance's admission to its proof (or to the end of the call)
    s->admit_s = t_admit[k];
    s->props.NrIterations = (int)std::min<long long>(h_iters[k], 2147483647LL); s->props.nodes = h_nodes[k];
    s->props.NrSolutionPool = h_ninc[k];
    s->timing[0] = ms_all * 1e-3; s->timing[1] = ms_ipm * 1e-3; s->timing[2] = (double)(nev / 2); s->timing[3] = (double)launched_nodes;
    s->timing[4] = (double)tot_iters; s->timing[5] = (double)rowiters;
    s->as_timing[0] = (double)h_as[0]; s->as_timing[1] = (double)h_as[1]; s->as_timing[2] = (double)h_as[2]; s->as_timing[3] = (double)h_as[3];
    s->as_timing[4] = (double)h_as[6]; s->as_timing[5] = (double)h_as[7]; s->as_timing[6] = ms_std * 1e-3; s->as_timing[7] = std_timed ? (double)(nev / 2) : 0.0;
    s->setup[0] = t_setup; s->setup[1] = t_ctx; s->setup[2] = ctx_built ? 1.0 : 0.0;
    s->err.clear();
    if (h_stalled[k] && unfinished) s->err = "retired without a proof: no progress for 64 branch-and-bound rounds (not a time-limit verdict)";
    if (have) {
      s->status = MIQP_STATUS_SUCCESS; s->has_sol = true;
      s->props.objective = h_inc[k];
      double lb = std::min(h_lb[k], h_inc[k]);
      // nodes within the gap of the incumbent are dropped without being refined: what is proven is the smaller of the open
      // list's best bound and incumbent - gap*|incumbent| (CPLEX reports the bound of its remaining nodes the same way)
      lb = std::min(lb, h_inc[k] - h_gap[k] * std::fabs(h_inc[k]));
      s->props.best_bound = lb; s->props.gap = std::fabs(lb - h_inc[k]) / (1e-10 + std::fabs(h_inc[k]));
      s->props.status = unfinished ? MIQP_CPX_STAT_TIME_LIM_FEAS : (s->props.gap <= 1e-9 ? MIQP_CPX_STAT_OPTIMAL : MIQP_CPX_STAT_OPTIMAL_TOL);
      s->Z.assign(h_Z.begin() + (size_t)k * Y.N * Y.nz, h_Z.begin() + (size_t)(k + 1) * Y.N * Y.nz);
      s->comp.assign(h_fix.begin() + (size_t)k * Y.fixlen, h_fix.begin() + (size_t)(k + 1) * Y.fixlen);
    } else {
      s->status = unfinished ? (h_stalled[k] ? MIQP_STATUS_FAILED_SEG_FAULT : MIQP_STATUS_FAILED_TIMEOUT) : MIQP_STATUS_FAILED_NO_SOLUT;   // (a stalled instance did not run out of time: the solver could not go on)
      s->props.objective = NAN; s->props.gap = NAN; s->props.best_bound = h_lb[k];
      s->props.status = unfinished ? MIQP_CPX_STAT_TIME_LIM_INFEAS : MIQP_CPX_STAT_INFEASIBLE;
    }
    statuses[k] = s->status;
  }
  if (abandoned) std::fprintf(stderr, "[miqp_gpu] the round loop was abandoned with %d of %d instances never admitted: they report FAILED_SEG_FAULT, the call fails\n", n - next_q, n);
  if (KNOB_P("MIQP_STATS")) std::fprintf(stderr, "[miqp_gpu stats] reachable-set diameter (L1) of instance 0: %.1f\n", hD[Y.d_misc + 2]);
  if (KNOB_P("MIQP_STATS")) std::fprintf(stderr, "[miqp_gpu stats] host: setup %.3f s (device context %.3f, instance tables and presolve %.3f, upload %.3f), rounds %.3f s (%d), results %.3f s\n", t_setup, t_ctx, t_tables, t_setup - t_ctx - t_tables, t_solve, rounds, wall_s() - t0 - t_solve);
  return !abandoned;
}

}  // namespace

// ================================================================================================
//  C ABI
// ================================================================================================
extern "C" {

const char* miqp_gpu_version(void) { return "miqp_gpu 0.1 (gfx950)"; }

miqp_solver_t* miqp_solver_create(const miqp_solver_opts* opts) {
  miqp_solver* s = new miqp_solver();
  if (opts) s->opts = *opts; else { s->opts.precision = 12; s->opts.device = -1; s->opts.gap_override = -1; }
  if (s->opts.precision <= 0) s->opts.precision = 12;
  return s;
}

void miqp_solver_destroy(miqp_solver_t* s) { delete s; }

int miqp_solver_set_params(miqp_solver_t* s, const miqp_model_params_c* p) {
  if (!s || !p) return -1;
  s->has_inst = inst_from_params(p, s->opts.precision - 2, s->inst, s->err);
  s->rescache.reset();
  s->has_sol = false;   // MIP starts stay registered (the reference keeps them in the wrapper across resetParameters)
  if (!s->has_inst) std::fprintf(stderr, "[miqp_gpu] %s\n", s->err.c_str());
  return s->has_inst ? 0 : -2;
}

int miqp_solver_load_dat(miqp_solver_t* s, const char* path) {
  if (!s || !path) return -1;
  s->has_inst = inst_from_dat(path, s->inst, s->err);
  s->has_sol = false; s->rescache.reset();
  if (!s->has_inst) std::fprintf(stderr, "[miqp_gpu] %s\n", s->err.c_str());
  return s->has_inst ? 0 : -2;
}

int miqp_solver_override_settings(miqp_solver_t* s, double max_solution_time, double relative_mip_gap_tolerance) {
  if (!s || !s->has_inst) return -1;
  s->inst.tilim = max_solution_time; s->inst.gap = relative_mip_gap_tolerance;
  return 0;
}

int miqp_solver_get_dims(const miqp_solver_t* s, int* o) {
  if (!s || !s->has_inst) return -1;
  o[0] = s->inst.C; o[1] = s->inst.N; o[2] = s->inst.R; o[3] = s->inst.E; o[4] = s->inst.O; o[5] = s->inst.L;
  return 0;
}

int miqp_solver_set_warmstart(miqp_solver_t* s, const miqp_raw_results_c* start, int warmstart_type) {
  if (!s) return -1;
  if (warmstart_type == MIQP_WARMSTART_NONE || !start) { s->ws[0].reset(); s->ws[1].reset(); return 0; }
  // slot 1: the last-solution start (readMIPStarts of the .mst file); slot 0: the receding-horizon start.  With
  // MIQP_WARMSTART_BOTH the caller registers each of the two with its own call (type RECEDING_HORIZON / LAST_SOLUTION)
  const int slot = warmstart_type == MIQP_WARMSTART_LAST_SOLUTION ? 1 : 0;
  if (start->N < 2 || start->NrCars < 1 || start->NrRegions < 1 || start->NrEnvironments < 0 || start->NrObstacles < 0 || start->MaxLinesObstacles < 0 ||
      start->NrCarToCarCollisions != start->NrCars - 1) return -2;
  if (s->has_inst && !dims_match(*start, s->inst)) { s->ws[slot].reset(); return -3; }   // a start of another shape is ignored
  s->ws[slot].reset(new OwnedResults(start->NrCars, start->N, start->NrRegions, start->NrEnvironments, start->NrObstacles, start->MaxLinesObstacles));
  s->ws[slot]->copy_from(*start);
  return 0;
}

int miqp_solver_solve_batch(miqp_solver_t* const* solvers, int n, int* statuses) {
  if (!solvers || n < 1 || !statuses) return -1;
  return solve_batch_impl(solvers, n, statuses) ? 0 : -2;
}

// A long queue is drained by LANES: the instances are dealt round robin to `lanes` sub-queues, each with its share of the slots,
// its own device context (pools, streams) and its own host thread, all on the same device.  The rounds of the lanes are not
// synchronised with each other, so the phases of a round that leave most of the device idle - the memory-backed launch on
// the few large nodes (it lasts as long as its slowest node), selection, the host's look at the counters between two
// rounds - run beside the on-chip launch of another lane.  Measured on one MI355X (tools/lanes_sweep.sh, the bench queue):
// 633 solves/s with one lane, 660 with two, 649 with three, 571 with four - the on-chip kernel fills the register files, so
// what hides is little, and the per-launch times of concurrent lanes no longer say what a kernel costs.  OFF by default
// (MIQP_LANES=2 switches it on); lanes are used when every lane keeps at least 128 instances in flight and the queue is
// longer than the slots.
int miqp_solver_solve_stream(miqp_solver_t* const* solvers, int n, int inflight, int* statuses) {
  if (!solvers || n < 1 || !statuses) return -1;
  int lanes = KNOB_P("MIQP_LANES") ? std::atoi(KNOB_P("MIQP_LANES")) : 1;
  if (lanes > 8) lanes = 8;
  while (lanes > 1 && (inflight <= 0 || inflight >= n || inflight / lanes < 128)) lanes--;
  if (lanes <= 1) return solve_batch_impl(solvers, n, statuses, nullptr, inflight) ? 0 : -2;
  for (int b = 0; b < n; ++b) { if (!solvers[b]) return -1; if (solvers[b]->opts.device != solvers[0]->opts.device) lanes = 1; }
  if (lanes <= 1) return solve_batch_impl(solvers, n, statuses, nullptr, inflight) ? 0 : -2;
  std::vector<std::vector<miqp_solver_t*>> sub(lanes); std::vector<std::vector<int>> st(lanes);
  for (int b = 0; b < n; ++b) sub[b % lanes].push_back(solvers[b]);
  std::vector<char> ok(lanes, 0);
  std::vector<std::thread> th;
  for (int l = 0; l < lanes; ++l) {
    st[l].assign(sub[l].size(), MIQP_STATUS_FAILED_SEG_FAULT);
    const int infl = (inflight + lanes - 1 - l) / lanes;   // the slots are dealt like the instances
    th.emplace_back([&, l, infl] { ok[l] = solve_batch_impl(sub[l].data(), (int)sub[l].size(), st[l].data(), nullptr, infl, l, lanes) ? 1 : 0; });
  }
  for (auto& t : th) t.join();
  int rc = 0;
  for (int b = 0; b < n; ++b) statuses[b] = st[b % lanes][b / lanes];
  for (int l = 0; l < lanes; ++l) if (!ok[l]) rc = -2;
  return rc;
}

int miqp_solver_solve_batch_multi(miqp_solver_t* const* solvers, int n, int gpus, int* statuses) {
  if (!solvers || n < 1 || !statuses) return -1;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { std::fprintf(stderr, "[miqp_gpu] no HIP device: the solver has no CPU path\n"); for (int k = 0; k < n; ++k) statuses[k] = MIQP_STATUS_FAILED_SEG_FAULT; return -2; }
  if (gpus <= 0 || gpus > ndev) gpus = ndev;
  if (gpus > n) gpus = n;
  // instance b -> device b mod G; no data-path exchange between the shards
  std::vector<std::vector<miqp_solver_t*>> shard(gpus); std::vector<std::vector<int>> st(gpus);
  for (int b = 0; b < n; ++b) { if (!solvers[b]) return -1; solvers[b]->opts.device = b % gpus; shard[b % gpus].push_back(solvers[b]); }
  std::vector<char> ok(gpus, 0);
  std::vector<std::thread> th;
  for (int g = 0; g < gpus; ++g) {
    st[g].assign(shard[g].size(), MIQP_STATUS_FAILED_SEG_FAULT);
    th.emplace_back([&, g] { ok[g] = solve_batch_impl(shard[g].data(), (int)shard[g].size(), st[g].data()) ? 1 : 0; });
  }
  for (auto& t : th) t.join();
  int rc = 0;
  for (int b = 0; b < n; ++b) statuses[b] = st[b % gpus][b / gpus];
  for (int g = 0; g < gpus; ++g) if (!ok[g]) rc = -2;
  return rc;
}

int miqp_solver_solve_split(miqp_solver_t* s, double timestamp, int world, int rank, miqp_exchange_fn exchange, void* user) {
  (void)timestamp;
  if (!s || !s->has_inst || world < 1 || rank < 0 || rank >= world || !exchange) return MIQP_STATUS_FAILED_SEG_FAULT;
  SplitCtx sc{world, rank, exchange, user};
  miqp_solver_t* one[1] = {s}; int st = MIQP_STATUS_FAILED_SEG_FAULT;
  if (!solve_batch_impl(one, 1, &st, &sc)) return MIQP_STATUS_FAILED_SEG_FAULT;
  return st;
}

int miqp_solver_split_roots(const miqp_solver_t* s, int world, int rank, int* root_of_pair, int* index, int* value, int cap, int* nroots_out, int* ncombos_out) {
  if (!s || !s->has_inst || world < 1 || rank < 0 || rank >= world) return -1;
  miqp_solver_t* one[1] = {const_cast<miqp_solver_t*>(s)};
  BatchShape bs = batch_layout(one, 1);
  if (!bs.ok) return -2;
  std::vector<double> D(bs.Y.dstride); std::vector<int> T(bs.Y.istride);
  compile_instance(s->inst, bs.Y, D.data(), T.data());
  std::vector<std::vector<std::pair<int, int>>> combos; split_roots(bs.Y, T.data(), combos, world);
  int np = 0, nr = 0;
  for (size_t q = 0; q < combos.size(); ++q) {
    if ((int)(q % (size_t)world) != rank) continue;
    for (auto& d : combos[q]) { if (np < cap) { if (root_of_pair) root_of_pair[np] = nr; if (index) index[np] = d.first; if (value) value[np] = d.second; } np++; }
    nr++;
  }
  if (nroots_out) *nroots_out = nr;
  if (ncombos_out) *ncombos_out = (int)combos.size();
  return np;
}

// ---------------------------------------------------------------- RCCL transport of the exchange (librccl loaded at run time)
struct Id128 { char b[128]; };   // ncclUniqueId is passed by value
namespace {
struct Rccl {
  void* lib = nullptr; void* comm = nullptr; int world = 0, rank = 0, device = -1; void* dbuf = nullptr; size_t dcap = 0; hipStream_t stream = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, Id128, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
};
Rccl g_rccl; std::mutex g_rccl_mu;
bool rccl_load() {
  if (g_rccl.lib) return true;
  for (const char* nm : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { g_rccl.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL); if (g_rccl.lib) break; }
  if (!g_rccl.lib) { std::fprintf(stderr, "[miqp_gpu] librccl not found: %s\n", dlerror()); return false; }
  g_rccl.GetUniqueId = (int (*)(void*))dlsym(g_rccl.lib, "ncclGetUniqueId");
  g_rccl.CommInitRank = (int (*)(void**, int, Id128, int))dlsym(g_rccl.lib, "ncclCommInitRank");
  g_rccl.CommDestroy = (int (*)(void*))dlsym(g_rccl.lib, "ncclCommDestroy");
  g_rccl.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(g_rccl.lib, "ncclAllReduce");
  g_rccl.Broadcast = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(g_rccl.lib, "ncclBroadcast");
  return g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllReduce && g_rccl.Broadcast;
}
// miqp_exchange_fn over the communicator: tiny host buffers staged through one device buffer
int rccl_exchange(void*, int op, void* buf, int count, int root) {
  Rccl& R = g_rccl;
  if (!R.comm) return -1;
  const size_t bytes = op == 0 ? (size_t)count * 8 : (size_t)count;
  if (hipSetDevice(R.device) != hipSuccess) return -2;
  if (bytes > R.dcap) { if (R.dbuf) (void)hipFree(R.dbuf); R.dcap = std::max<size_t>(bytes, 1 << 16); if (hipMalloc(&R.dbuf, R.dcap) != hipSuccess) { R.dbuf = nullptr; R.dcap = 0; return -2; } }
  if (hipMemcpyAsync(R.dbuf, buf, bytes, hipMemcpyHostToDevice, R.stream) != hipSuccess) return -2;
  const int rc = op == 0 ? R.AllReduce(R.dbuf, R.dbuf, (size_t)count, /*ncclUint64*/ 5, /*ncclMin*/ 3, R.comm, R.stream)
                         : R.Broadcast(R.dbuf, R.dbuf, bytes, /*ncclUint8*/ 1, root, R.comm, R.stream);
  if (rc != 0) return -3;
  if (hipMemcpyAsync(buf, R.dbuf, bytes, hipMemcpyDeviceToHost, R.stream) != hipSuccess) return -2;
  return hipStreamSynchronize(R.stream) == hipSuccess ? 0 : -2;
}
}  // namespace

int miqp_comm_unique_id(char* out128) {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (!out128 || !rccl_load()) return -1;
  return g_rccl.GetUniqueId(out128) == 0 ? 0 : -2;
}
int miqp_comm_init(int world, int rank, const char* id128, int device) {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (!id128 || world < 1 || rank < 0 || rank >= world || !rccl_load()) return -1;
  if (g_rccl.comm) return -4;   // one communicator per process
  int ndev = 0; if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { std::fprintf(stderr, "[miqp_gpu] no HIP device\n"); return -2; }
  if (device < 0) { if (hipGetDevice(&device) != hipSuccess) device = 0; }
  if (hipSetDevice(device) != hipSuccess) return -2;
  Id128 id; std::memcpy(id.b, id128, 128);
  if (g_rccl.CommInitRank(&g_rccl.comm, world, id, rank) != 0) { g_rccl.comm = nullptr; return -3; }
  if (hipStreamCreate(&g_rccl.stream) != hipSuccess) return -2;
  g_rccl.world = world; g_rccl.rank = rank; g_rccl.device = device;
  return 0;
}
int miqp_comm_finalize(void) {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (g_rccl.comm) { g_rccl.CommDestroy(g_rccl.comm); g_rccl.comm = nullptr; }
  if (g_rccl.dbuf) { (void)hipFree(g_rccl.dbuf); g_rccl.dbuf = nullptr; g_rccl.dcap = 0; }
  if (g_rccl.stream) { (void)hipStreamDestroy(g_rccl.stream); g_rccl.stream = nullptr; }
  return 0;
}
int miqp_comm_selftest(miqp_exchange_fn exchange, void* user, int world, int rank) {
  if (!exchange) { if (!g_rccl.comm) return -1; exchange = rccl_exchange; world = g_rccl.world; rank = g_rccl.rank; }
  // all-reduce(min) of unsigned words: one word that is smallest on the last rank, one on rank 0, one with the top bit set
  unsigned long long w[3] = {1000ull - (unsigned long long)rank, 5ull + (unsigned long long)rank, 0x8000000000000000ull + (unsigned long long)(world - rank)};
  if (exchange(user, 0, w, 3, 0) != 0) return -2;
  if (w[0] != 1000ull - (unsigned long long)(world - 1) || w[1] != 5ull || w[2] != 0x8000000000000001ull) return -3;
  for (int root = 0; root < world; ++root) {
    unsigned char b[37]; for (int k = 0; k < 37; ++k) b[k] = (unsigned char)(rank == root ? (7 * k + root) & 255 : 0xEE);
    if (exchange(user, 1, b, 37, root) != 0) return -4;
    for (int k = 0; k < 37; ++k) if (b[k] != (unsigned char)((7 * k + root) & 255)) return -5;
  }
  return 0;
}
int miqp_solver_solve_split_rccl(miqp_solver_t* s, double timestamp) {
  if (!g_rccl.comm || !s) return MIQP_STATUS_FAILED_SEG_FAULT;
  s->opts.device = g_rccl.device;
  return miqp_solver_solve_split(s, timestamp, g_rccl.world, g_rccl.rank, rccl_exchange, nullptr);
}

int miqp_solver_raw_sizes(const miqp_solver_t* s, int* out4) {
  if (!s || !s->has_inst || !out4) return -1;
  raw_sizes(s->inst, out4[0], out4[1], out4[2], out4[3]);
  return 0;
}

int miqp_solver_lift_tables(const miqp_solver_t* s, double* out, int cap) {
  if (!s || !s->has_inst || !out) return -1;
  miqp_solver_t* one[1] = {const_cast<miqp_solver_t*>(s)};
  BatchShape bs = batch_layout(one, 1);
  if (!bs.ok) return -2;
  const int n = bs.Y.C * 2 * bs.Y.N * 16;
  if (cap < n) return -3;
  std::vector<double> D(bs.Y.dstride, 0.0);
  lift_tables(s->inst, bs.Y, D.data());
  std::copy(D.begin() + bs.Y.d_lift, D.begin() + bs.Y.d_lift + n, out);
  return n;
}

int miqp_solver_solve(miqp_solver_t* s, double timestamp) {
  (void)timestamp;
  if (!s || !s->has_inst) return MIQP_STATUS_FAILED_SEG_FAULT;
  miqp_solver_t* one[1] = {s}; int st = MIQP_STATUS_FAILED_SEG_FAULT;
  if (!solve_batch_impl(one, 1, &st)) return MIQP_STATUS_FAILED_SEG_FAULT;
  return st;
}

int miqp_solver_last_admission(const miqp_solver_t* s, double* out1) { if (!s || !out1) return -1; out1[0] = s->admit_s; return 0; }

const char* miqp_solver_last_error(const miqp_solver_t* s) { return s ? s->err.c_str() : "null handle"; }

int miqp_solver_get_results(const miqp_solver_t* s, miqp_raw_results_c* out) {
  if (!s || !out || !s->has_sol) return -1;
  if (!dims_match(*out, s->inst)) return -2;   // the caller's record must be sized for this instance
  if (s->rescache) { out->N = s->rescache->r.N; out->NrEnvironments = s->rescache->r.NrEnvironments; out->NrRegions = s->rescache->r.NrRegions; out->NrObstacles = s->rescache->r.NrObstacles;
                     out->MaxLinesObstacles = s->rescache->r.MaxLinesObstacles; out->NrCarToCarCollisions = s->rescache->r.NrCarToCarCollisions; out->NrCars = s->rescache->r.NrCars;
                     s->rescache->copy_to(*out); return 0; }
  std::vector<double> D(s->lay.dstride); std::vector<int> T(s->lay.istride);
  compile_instance(s->inst, s->lay, D.data(), T.data());
  fill_results(s->inst, s->lay, D.data(), T.data(), s->comp.data(), s->Z.data(), out);
  return 0;
}

int miqp_solver_materialize_results(miqp_solver_t* const* solvers, int n, int threads) {
  if (!solvers || n < 0) return -1;
  int nth = threads > 0 ? threads : std::min((int)std::thread::hardware_concurrency(), 32);   // (20 480 records on a 2 x 64-core host: 0.29 / 0.16 / 0.13 / 0.15 / 0.25 s at 8 / 16 / 32 / 64 / 256 threads)
  nth = std::max(1, std::min(nth, std::max(1, n / 4)));
  std::atomic<int> next{0}, made{0};
  auto work = [&] {
    std::vector<double> D; std::vector<int> T;   // (one pair of table buffers per thread, not per record)
    for (int k = next.fetch_add(1); k < n; k = next.fetch_add(1)) {
      miqp_solver* s = solvers[k];
      if (!s || !s->has_sol || s->rescache) continue;
      std::unique_ptr<OwnedResults> R(new OwnedResults(s->inst));
      D.resize(s->lay.dstride); T.resize(s->lay.istride);
      compile_instance(s->inst, s->lay, D.data(), T.data());
      fill_results(s->inst, s->lay, D.data(), T.data(), s->comp.data(), s->Z.data(), &R->r);
      s->rescache = std::move(R); made.fetch_add(1);
    }
  };
  if (nth <= 1) work();
  else { std::vector<std::thread> th; for (int t = 0; t < nth; ++t) th.emplace_back(work); for (auto& t : th) t.join(); }
  return made.load();
}

int miqp_solver_get_properties(const miqp_solver_t* s, miqp_solution_properties_c* out) {
  if (!s || !out) return -1;
  *out = s->props;
  return 0;
}

int miqp_solver_last_timing(const miqp_solver_t* s, double* out6) {
  if (!s || !out6) return -1;
  for (int k = 0; k < 6; ++k) out6[k] = s->timing[k];
  return 0;
}

int miqp_solver_last_active_set(const miqp_solver_t* s, double* out8) {
  if (!s || !out8) return -1;
  for (int k = 0; k < 8; ++k) out8[k] = s->as_timing[k];
  return 0;
}

int miqp_solver_last_setup(const miqp_solver_t* s, double* out3) {
  if (!s || !out3) return -1;
  for (int k = 0; k < 3; ++k) out3[k] = s->setup[k];
  return 0;
}

int miqp_solver_export_lp(const miqp_solver_t* s, const char* path) {
  if (!s || !s->has_inst || !path) return -1;
  return miqp::export_lp(s->inst, path);
}

// ---------------------------------------------------------------- planner core (rows f1 / f2 of SURVEY.md section 8)
int miqp_fraction_parameters(int nr_regions, float max_velocity_fitting, double* out) {
  if (nr_regions < 1 || !out) return -1;
  miqp::fraction_parameters(nr_regions, max_velocity_fitting, out); return 0;
}
int miqp_fitting_polynomial_parameters(int nr_regions, float max_velocity_fitting, float min_velocity_fitting, double* out) {
  if (nr_regions < 1 || !out) return -1;
  return miqp::fitting_polynomial_parameters(nr_regions, max_velocity_fitting, min_velocity_fitting, out) ? 0 : -2;
}
int miqp_mean_angles(const double* fraction_parameters, int nr_regions, double* out) {
  if (!fraction_parameters || nr_regions < 1 || !out) return -1;
  miqp::mean_angles(fraction_parameters, nr_regions, out); return 0;
}
int miqp_limits_per_region(const double* fraction_parameters, int nr_regions, float long_min, float long_max, float lat_min, float lat_max,
                           double* min_x, double* max_x, double* min_y, double* max_y) {
  if (!fraction_parameters || nr_regions < 1 || !min_x || !max_x || !min_y || !max_y) return -1;
  miqp::limits_per_region(fraction_parameters, nr_regions, long_min, long_max, lat_min, lat_max, min_x, max_x, min_y, max_y); return 0;
}
int miqp_calculate_region_idx(const double* fraction_parameters, int nr_regions, float vx, float vy, int* out) {
  if (!fraction_parameters || nr_regions < 1 || !out) return -1;
  return miqp::calculate_region_idx(fraction_parameters, nr_regions, vx, vy, out);
}
int miqp_reserve_neighbor_regions(int* row, int nr_regions, int expansions) {
  if (!row || nr_regions < 1) return -1;
  return miqp::reserve_neighbor_regions(row, nr_regions, expansions) ? 1 : 0;
}
int miqp_calculate_possible_regions(const double* fraction_parameters, int nr_regions, const double* theta_ref, int n, int* flags) {
  if (!fraction_parameters || nr_regions < 1 || !theta_ref || !flags) return -1;
  miqp::calculate_possible_regions(fraction_parameters, nr_regions, theta_ref, n, flags); return 0;
}
int miqp_reference_trajectory(const double* ref_xy, int n_ref, const double* state5, double dt, int num_points, double line_interp_inc, double vel_desired,
                              double delta_s_desired, double acc_lat_max, int vel_curve_dep, double* out) {
  if (!ref_xy || n_ref < 2 || !state5 || num_points < 1 || !out || !(dt > 0) || delta_s_desired < 0) return -1;
  miqp::PolyLine line; line.build(ref_xy, n_ref);
  if (line.x.size() < 2) return -2;
  miqp::reference_trajectory(line, state5, dt, num_points, line_interp_inc, vel_desired, delta_s_desired, acc_lat_max, vel_curve_dep != 0, out);
  return 0;
}
int miqp_update_car(const double* settings12, const double* fraction_parameters, const double* initial_state6, const double* ref_xy, int n_ref, double desired_velocity,
                    double delta_s_desired, double timestep, int track_reference_positions, int is_ego, int num_cars, double* ref4N, int* possible_region, double* weights8) {
  if (!settings12 || !fraction_parameters || !initial_state6 || !ref_xy || n_ref < 2 || !ref4N || !possible_region || !weights8 || num_cars < 1) return -1;
  miqp::CarUpdateSettings S{(int)settings12[0], (int)settings12[1], (int)settings12[2], (int)settings12[3], settings12[4], settings12[5], settings12[6], settings12[7],
                            settings12[8], settings12[9], settings12[10], settings12[11]};
  if (S.nr_regions < 1 || S.nr_steps < 2 || !(S.ts > 0)) return -1;
  return miqp::update_car(S, fraction_parameters, initial_state6, ref_xy, n_ref, desired_velocity, delta_s_desired, timestep, track_reference_positions != 0, is_ego != 0,
                          num_cars, ref4N, possible_region, weights8) ? 0 : 1;
}
int miqp_calculate_warmstart(const miqp_raw_results_c* last, miqp_raw_results_c* out, double ts, double minimum_region_change_speed) {
  if (!last || !out || last->N != out->N || last->NrCars != out->NrCars || last->NrRegions != out->NrRegions || last->N < 2) return -1;
  miqp::calculate_warmstart(*last, *out, ts, minimum_region_change_speed); return 0;
}

int miqp_initial_pose_check(const miqp_model_params_c* p) {
  if (!p || p->NumCars < 1 || !p->IntitialState || !p->WheelBase || (p->nr_environments > 0 && (!p->env_offsets || !p->env_vertices))) return -2;
  return miqp::initial_pose_check(*p);
}
int miqp_select_environment(const double* pieces_xy, const int* piece_off, int n_pieces, const double* traj_xy, const int* traj_off, int n_traj, int* selected) {
  if (n_pieces < 0 || n_traj < 0 || (n_pieces > 0 && (!pieces_xy || !piece_off || !selected)) || (n_traj > 0 && (!traj_xy || !traj_off))) return -1;
  int cnt = 0;
  for (int e = 0; e < n_pieces; ++e) {
    bool hit = false;
    for (int t = 0; t < n_traj && !hit; ++t) hit = miqp::polyline_hits_convex(traj_xy + 2 * traj_off[t], traj_off[t + 1] - traj_off[t], pieces_xy + 2 * piece_off[e], piece_off[e + 1] - piece_off[e]);
    selected[e] = hit ? 1 : 0; cnt += hit ? 1 : 0;
  }
  return cnt;
}
int miqp_obstacle_intersects_environment(const double* pieces_xy, const int* piece_off, int n_pieces, const double* obstacle_xy, int n_steps, int is_static) {
  if (n_pieces < 0 || n_steps < 1 || !obstacle_xy || (n_pieces > 0 && (!pieces_xy || !piece_off))) return -1;
  return miqp::obstacle_intersects_environment(pieces_xy, piece_off, n_pieces, obstacle_xy, n_steps, is_static != 0) ? 1 : 0;
}
int miqp_bark_trajectory(const miqp_raw_results_c* results, int car, double start_time, double ts, double min_speed, double* out_rows5) {
  if (!results || !out_rows5 || car < 0 || car >= results->NrCars || results->N < 0 || !results->vel_x || !results->vel_y || !results->pos_x || !results->pos_y) return -1;
  return miqp::bark_trajectory(*results, car, start_time, ts, min_speed, out_rows5);
}
int miqp_obstacles_roi(double x, double y, double theta, double behind_distance, double front_distance, double side_distance, double* roi_xy) {
  if (!roi_xy) return -1;
  miqp::obstacles_roi(x, y, theta, behind_distance, front_distance, side_distance, roi_xy);
  return 0;
}
int miqp_obstacle_intersects_environment_roi(const double* pieces_xy, const int* piece_off, int n_pieces, const double* obstacle_xy, int n_steps, int is_static, const double* roi_xy) {
  if (n_pieces < 0 || n_steps < 1 || !obstacle_xy || (n_pieces > 0 && (!pieces_xy || !piece_off))) return -1;
  return miqp::obstacle_intersects_environment(pieces_xy, piece_off, n_pieces, obstacle_xy, n_steps, is_static != 0, roi_xy) ? 1 : 0;
}
int miqp_environment_warmstart(const miqp_raw_results_c* last, miqp_raw_results_c* out, const int* ids_old, int n_old, const int* ids_new, int n_new) {
  if (!last || !out || n_old < 0 || n_new < 0 || last->NrEnvironments != n_old || out->NrEnvironments != n_new || last->N != out->N || last->NrCars != out->NrCars) return -1;
  const int* const in[5] = {last->notWithinEnvironmentRear, last->notWithinEnvironmentFrontUbUb, last->notWithinEnvironmentFrontLbUb, last->notWithinEnvironmentFrontUbLb, last->notWithinEnvironmentFrontLbLb};
  int* const o[5] = {out->notWithinEnvironmentRear, out->notWithinEnvironmentFrontUbUb, out->notWithinEnvironmentFrontLbUb, out->notWithinEnvironmentFrontUbLb, out->notWithinEnvironmentFrontLbLb};
  miqp::environment_warmstart(in, o, last->NrCars, last->N, ids_old, n_old, ids_new, n_new);
  return 0;
}

// MiqpPlanner::Plan, the part between the environment update and the trajectory read-out (src/miqp_planner.cpp:634-645,
// 692-766): initial regions of every car from its initial velocity, all combinations, one solve per combination until
// one succeeds; the start region is made possible for the attempt and rolled back when the attempt fails.
int miqp_plan(miqp_solver_t* s, miqp_model_params_c* p, int* initial_region, int* possible_region, const miqp_raw_results_c* warmstart, int warmstart_type,
              double timestamp, int* status_out) {
  if (!s || !p || !initial_region || !possible_region || p->NumCars < 1 || p->nr_regions < 1) return 0;
  p->initial_region = initial_region; p->possible_region = possible_region;
  const int C = p->NumCars, R = p->nr_regions;
  // with an environment: every car's rear and front axle point must lie within one of its pieces, else Plan fails before any
  // solve (src/miqp_planner.cpp:654-685)
  if (p->nr_environments > 0 && p->env_offsets && p->env_vertices && p->WheelBase) {
    const int bad = miqp::initial_pose_check(*p);
    if (bad >= 0) { std::fprintf(stderr, "[miqp_gpu] Initial pose %s collides for car idx = %d\n", (bad & 1) ? "front" : "rear", bad >> 1); if (status_out) *status_out = MIQP_STATUS_FAILED_NO_SOLUT; return 0; }
  }
  std::vector<std::vector<int>> per_car(C), combos;
  std::vector<int> idx(R);
  for (int c = 0; c < C; ++c) {
    int m = miqp::calculate_region_idx(p->fraction_parameters, R, (float)p->IntitialState[c * 6 + 1], (float)p->IntitialState[c * 6 + 4], idx.data());
    per_car[c].assign(idx.begin(), idx.begin() + m);
    if (m < 1) return 0;
  }
  miqp::region_combinations(per_car, combos);
  int status = MIQP_STATUS_FAILED_NO_SOLUT;
  std::vector<char> rollback(C, 0);
  for (auto& comb : combos) {
    for (int c = 0; c < C; ++c) {
      initial_region[c] = comb[c] + 1;   // OPL is 1-based
      if (possible_region[c * R + comb[c]] == 0) { possible_region[c * R + comb[c]] = 1; rollback[c] = 1; } else rollback[c] = 0;
    }
    if (miqp_solver_set_params(s, p) != 0) { status = MIQP_STATUS_FAILED_SEG_FAULT; break; }
    if ((warmstart_type == MIQP_WARMSTART_RECEDING_HORIZON || warmstart_type == MIQP_WARMSTART_BOTH) && warmstart)
      miqp_solver_set_warmstart(s, warmstart, warmstart_type);
    status = miqp_solver_solve(s, timestamp);
    if (status == MIQP_STATUS_SUCCESS) break;
    if (status == MIQP_STATUS_FAILED_SEG_FAULT || status == MIQP_STATUS_FAILED_TIMEOUT) break;
    for (int c = 0; c < C; ++c) if (rollback[c]) possible_region[c * R + comb[c]] = 0;
  }
  if (status_out) *status_out = status;
  return status == MIQP_STATUS_SUCCESS ? 1 : 0;
}


int miqp_solver_write_dat(const miqp_solver_t* s, const char* path) {
  if (!s || !s->has_inst || !path) return -1;
  FILE* f = std::fopen(path, "w"); if (!f) return -2;
  bool ok = miqp::write_dat(s->inst, f);
  return (std::fclose(f) == 0 && ok) ? 0 : -3;
}

int miqp_solver_write_solution(const miqp_solver_t* s, const char* path) {
  if (!s || !s->has_sol || !path) return -1;
  OwnedResults R(s->inst);
  if (miqp_solver_get_results(s, &R.r) != 0) return -2;
  FILE* f = std::fopen(path, "w"); if (!f) return -2;
  bool ok = miqp::write_solution(R.r, s->props.objective, f);
  return (std::fclose(f) == 0 && ok) ? 0 : -3;
}

int miqp_solver_write_mst(const miqp_solver_t* s, const char* path) {
  if (!s || !s->has_sol || !path) return -1;
  OwnedResults R(s->inst);
  if (miqp_solver_get_results(s, &R.r) != 0) return -2;
  FILE* f = std::fopen(path, "w"); if (!f) return -2;
  bool ok = miqp::write_mst(R.r, s->props.objective, f);
  return (std::fclose(f) == 0 && ok) ? 0 : -3;
}

int miqp_solver_read_mst(miqp_solver_t* s, const char* path) {
  if (!s || !s->has_inst || !path) return -1;
  OwnedResults R(s->inst);
  if (miqp::read_mst(path, R.r) <= 0) return -2;
  return miqp_solver_set_warmstart(s, &R.r, MIQP_WARMSTART_LAST_SOLUTION);
}

int miqp_solver_solve_fixed(miqp_solver_t* s, const miqp_raw_results_c* fixed, miqp_raw_results_c* out, double* objective, int* iterations) {
  if (!s || !s->has_inst || !fixed) return -1;
  miqp_solver_t* one[1] = {s};
  BatchShape bs = batch_layout(one, 1);
  if (!bs.ok) return -2;
  const Layout& Y = bs.Y;
  if (!dims_match(*fixed, s->inst) || (out && !dims_match(*out, s->inst))) return -2;
  DevCtx* Xp = ctx_for_device(s->opts.device);
  if (!Xp) return -3;
  DevCtx& X = *Xp;
  std::lock_guard<std::mutex> ctx_lock(X.mu);
  if (!ctx_prepare(X, Y, 1, 1, 64, 16, 3)) return -3;
  if (!set_kernel_lds(Y, ipm_lds_bytes(Y), eval_lds_bytes(Y))) return -3;
  std::vector<double> D(Y.dstride); std::vector<int> T(Y.istride);
  compile_instance(s->inst, Y, D.data(), T.data());
  std::vector<signed char> fix;
  if (!fix_from_results(s->inst, Y, T.data(), fixed, fix)) return -4;
  DevBuf& B = X.B; hipStream_t st = X.stream;
  int one_i = 1, zero = 0;
  if (hipMemcpyAsync((void*)B.inst_d, D.data(), D.size() * 8, hipMemcpyHostToDevice, st) != hipSuccess) return -3;
  (void)hipMemcpyAsync((void*)B.inst_i, T.data(), T.size() * 4, hipMemcpyHostToDevice, st);
  (void)hipMemcpyAsync(B.pool_fix, fix.data(), Y.fixlen, hipMemcpyHostToDevice, st);
  (void)hipMemcpyAsync(B.batch_count, &one_i, 4, hipMemcpyHostToDevice, st);
  (void)hipMemcpyAsync(B.batch_node, &zero, 4, hipMemcpyHostToDevice, st);
  (void)hipMemcpyAsync(B.batch_inst, &zero, 4, hipMemcpyHostToDevice, st);
  (void)hipMemsetAsync(B.inst_nodes, 0, 8, st); (void)hipMemsetAsync(B.inst_iters, 0, 8, st); (void)hipMemsetAsync(B.stat_rowiters, 0, 8, st);
  { DevBuf Bp = B; Bp.qp_tol = QP_TOL_FINAL; Bp.use_cutoff = 0; Bp.ws_on = 0;
    launch_ipm_batch(X, Bp, 1, st); }
  std::vector<double> Z((size_t)Y.N * Y.nz); double obj = 0, viol = 0; int ok = 0, it = 0;
  (void)hipMemcpyAsync(Z.data(), B.batch_Z, Z.size() * 8, hipMemcpyDeviceToHost, st);
  (void)hipMemcpyAsync(&obj, B.batch_obj, 8, hipMemcpyDeviceToHost, st);
  (void)hipMemcpyAsync(&viol, B.batch_viol, 8, hipMemcpyDeviceToHost, st);
  (void)hipMemcpyAsync(&ok, B.batch_ok, 4, hipMemcpyDeviceToHost, st);
  (void)hipMemcpyAsync(&it, B.batch_it, 4, hipMemcpyDeviceToHost, st);
  if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) return -3;
  HostGeo G{s->inst, Y, D.data(), T.data()}; double cobj = 0; (void)step0_check(G, cobj);
  if (objective) *objective = obj + cobj;
  if (iterations) *iterations = it;
  if (!ok || viol > FEAS_TOL) return 1;
  if (out) {
    // undecided leaf disjunctions: canonical completion is done by fill_results (it evaluates every side)
    for (auto& b : fix) if (b < 0) b = 0;
    fill_results(s->inst, Y, D.data(), T.data(), fix.data(), Z.data(), out);
  }
  return 0;
}

}  // extern "C"
