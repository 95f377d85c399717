// kernels.hip - CDNA4 (gfx950) device code of the MIQP solver.
//
//   select_kernel     one workgroup per instance: prune the near list against the incumbent, pick the `take` best
//                     bounds by an 8-bit radix select (dives: deepest first), spill to / refill from the far tier of
//                     the open list in HBM, lower bound over both tiers, termination.
//   ipm_onchip_kernel<C>  (ipm_onchip.hip, C <= 2, N <= 20) one 64-lane wavefront per B&B node, the whole working set in
//                     registers and LDS: box rows keyed by (stage, column, side) in registers, general rows compacted in
//                     LDS, Riccati recursion in the MFMA tile layout with shift-form products and Gauss-Jordan elimination.
//   ipm_kernel<C>     the general, memory-backed interior point kernel: nodes the on-chip kernel hands over (more general
//                     rows than its LDS holds), and everything for C = 3, 4 (dense LDS stage matrices, 2x2-tiled MFMA).
//   eval_kernel<C>    one wavefront per node: canonical completion of the undecided disjunctions, incumbent update
//                     (64-bit atomicMin), choice of the branching disjunction (wave reduction), bound lifting of the
//                     children from the node's dual solution, child emission into the two list tiers.
//   roll_kernel       publishes the node records freed in this round to the next one.
//
// The model being solved is the disjunctive form of cplexmodel/*.mod described in host_inst.hpp / DESIGN.md.
#include <hip/hip_runtime.h>

#include "host_inst.hpp"

#ifndef MIQP_RHO
#define MIQP_RHO 1.0e5
#endif

#ifndef MIQP_IPM_WPE
#define MIQP_IPM_WPE 3   // wavefronts per SIMD the interior point kernel is register-allocated for (168 VGPRs; measured best of 2, 3, 4)
#endif

// timing ablations (diagnostic build -DMIQP_ABLATE: the host replays one full batch with parts of the kernel switched
// off by the runtime mask DevBuf::abl and a fixed iteration count; the results of such launches are discarded)
#ifdef MIQP_ABLATE
#define MIQP_ABL (B.abl)
#else
#define MIQP_ABL 0
#endif
namespace miqp {

constexpr double RHO_EL = MIQP_RHO;     // exact-penalty weight of the elastic rows
constexpr double FEAS_TOL = 1.0e-6;
constexpr double QP_TOL = 1.0e-6;      // cap of the node relaxations' tolerance (1e-4 x gap below it): with ~1000 complementarity pairs the bound of a node is then loose by <= 0.1 % of the objective (1e-8: 8 % more iterations for nothing); the returned incumbent is polished to QP_TOL_FINAL
#ifndef MIQP_T0
#define MIQP_T0 1.0e-3
#endif
#ifndef MIQP_LAM0
#define MIQP_LAM0 2000.0   // initial multiplier of every elastic row: s*lambda and t*(rho - lambda) start at comparable size (swept 0.3 .. 3e4)
#endif
#ifndef MIQP_S0
#define MIQP_S0 100.0
#endif
#ifndef MIQP_STEPFRAC
#define MIQP_STEPFRAC 0.995
#endif
#ifndef MIQP_SIGMA0
#define MIQP_SIGMA0 0.3
#endif
constexpr double QP_T0 = MIQP_T0;       // initial elastic slack: t*mu (mu ~ rho) starts at the order of s*lambda
constexpr double QP_TOL_FINAL = 1.0e-13;  // polish of the returned incumbent / solve_fixed
constexpr double QP_SIGMA = MIQP_SIGMA0;       // centering parameter of the first iteration
#ifndef MIQP_SIGMA_LO
#define MIQP_SIGMA_LO 0.02
#endif
#ifndef MIQP_SIGMA_HI
#define MIQP_SIGMA_HI 0.5
#endif
constexpr double QP_SIGMA_LO = MIQP_SIGMA_LO, QP_SIGMA_HI = MIQP_SIGMA_HI;
constexpr int QP_MAXIT = 80;
constexpr int NFIELD = 4;            // per-row state: s, lambda, t, g.dz
constexpr int NCACHE = 9;            // per-row cache: rhs, aq (-1: inactive), packed columns, 6 coefficients
constexpr int GROWS = 48;            // rows of a stage staged through LDS per chunk of the stage-Hessian contraction
constexpr int GSTR = 17;             // LDS stride of a dense row (16 columns of the MFMA tile + 1: conflict-free)
// LDS doubles behind Z of the interior point kernel: [dZ | row chunk | scaled residuals]; the decode phase reuses the
// region as one dense scratch row per lane
// (more than two cars: the dense stage matrices S, P, T, the vectors and the gains of one stage live there instead)
__host__ __device__ inline int ipm_stage_doubles(int C) {
  int NX = 6 * C, NU = 2 * C, NZ = 8 * C;
  return C <= 2 ? GROWS * (GSTR + 1) : NZ * (NZ + 1) + NX * (NX + 1) + NX * (NZ + 1) + NZ + NX + NU * (NX + 1);   // (rows padded by one: the column accesses of S, P, T stay off one LDS bank)
}
// eval_kernel: doubles of the LDS region shared by the slow-alternative table (floats) and the dense rows of the lifting
// (the lifting decodes one dense row per lane, LIFT_LANES lanes at a time: with all 64 the rows were 8.7 KB of the 18 KB that keep eval_kernel at 8 wavefronts per CU)
constexpr int LIFT_LANES = 32;
__host__ __device__ inline int eval_shared_doubles(int C, int N, int P) {
  const int a = (C * N * P + 1) / 2, b = LIFT_LANES * (8 * C + 1);
  return a > b ? a : b;
}
__host__ __device__ inline int ipm_scratch_doubles(int N, int C) {
  int NZ = 8 * C, a = N * NZ + ipm_stage_doubles(C), b = 64 * (C <= 2 ? GSTR : NZ + 1);
  return a > b ? a : b;
}
typedef double d4_t __attribute__((ext_vector_type(4)));

enum { PT_R = 0, PT_U = 1, PT_L = 2 };
__device__ __constant__ int ENV_PT_D[5][2] = {{PT_R, PT_R}, {PT_U, PT_U}, {PT_L, PT_U}, {PT_U, PT_L}, {PT_L, PT_L}};
__device__ __constant__ int OBS_PT_D[5][2] = {{PT_R, PT_R}, {PT_L, PT_L}, {PT_U, PT_L}, {PT_L, PT_U}, {PT_U, PT_U}};

struct DevBuf {
  Layout Y;
  const double* inst_d; const int* inst_i;
  // node pool
  signed char* pool_fix;         // [pool_cap][fixlen]
  int* pool_count; int pool_cap;
  // recycled node records: queue of free slots; pops are limited to entries pushed before the current round
  int* free_q; unsigned int* free_head; unsigned int* free_tail; unsigned int* free_limit;
  // open lists per instance
  // open lists: two buffers per instance (select reads buffer `open_sel`, writes the survivors to the other one,
  // eval appends children to that other one); unsorted, selection by radix select on the 64-bit key
  double* open_bound; int* open_node; int* open_depth; int* open_count; int open_cap; int open_sel;
  unsigned long long* open_key;
  // second tier of the open list ("far"): single-buffered, append-only between refills.  The two-buffer list above holds the
  // nodes with the lowest bounds (everything <= near_thr), which is all that best-bound selection has to scan and copy every
  // round; children with a bound above near_thr are appended here and come back, lowest bounds first, when the near list
  // runs low (select_kernel).  Sized from the 288 GB of HBM: best-bound order survives frontiers of millions of nodes.
  double* far_bound; int* far_node; int* far_depth; int* far_count; int far_cap;
  unsigned long long* far_minkey;   // orderable(min bound) over the far entries (atomicMin on insert, recomputed by a refill)
  double* near_thr;                 // children with bound > near_thr go to the far tier (1e300: tier unused so far)
  int* inst_mode;                   // 1: the round dives (depth first): children stay in the near list whatever their bound
  // per instance state
  unsigned long long* inc_key;   // orderable(objective), the low 20 bits replaced by a hash of the completed record (what wins a tie must not depend on batch slots)
  unsigned long long* batch_candkey;   // per batch slot: the key of the incumbent candidate evaluated there this round, ~0 otherwise
  int* batch_candinst;           // ... and the instance it belongs to (batch_inst is being rewritten by other workgroups of select_kernel while one looks its winner up)
  int prev_bc;                   // batch slots of the round before (select_kernel looks the winner of the incumbent key up among them)
  unsigned long long* inc_seen;  // key whose solution has been copied to inc_fix / inc_Z
  double* inc_obj;               // objective of the stored incumbent
  double* inc_ext;               // upper bound from outside (tree split over ranks: the best incumbent of the other ranks), 1e300 otherwise
  signed char* inc_fix; double* inc_Z;
  unsigned short* inc_A; unsigned long long* inc_Mtag;   // [n_inst][64], [n_inst]: active set and ring tag of M of the solve that found the incumbent (active-set launches; tag 0: none) - the start of the local search's leaves
  double* lower_bound; int* inst_done; int* inst_flags; double* inst_gap; double* inst_const;
  long long* inst_nodes; long long* inst_iters; int* inst_ninc;
  // batch of the current round
  int* batch_count; int* batch_node; int* batch_inst; int* batch_depth; int batch_cap;
  double* batch_Z; double* batch_obj; double* batch_viol; int* batch_ok; int* batch_it; double* batch_bound;
  signed char* batch_comp;       // completed fix record of feasible nodes
  double* rowstate;              // [grid][NFIELD][ROWCAP]
  double* rowcache;              // [grid][NCACHE][ROWCAP] decoded sparse rows of the node being solved
  double* kgain;                 // [grid][N][NU][NX+2] feedback gains K and feed-forward k of the Riccati sweep
  int* active_insts;             // number of instances not yet done
  int nodes_per_round; int n_inst;   // n_inst: instances of the call (the queue); every per-instance array has that many entries
  // Streaming admission: the list storage (near buffers, keys, far tier) exists per SLOT, n_slots of them = the instances in
  // flight; an instance occupies a slot from its admission to its proof (or its time limit) and the slot is then handed to
  // the next instance of the queue.  Without streaming n_slots == n_inst and slot == instance.
  int n_slots; int* inst_slot; int* slot_inst;   // -1: not in flight / empty
  int* inst_kill;                // set by the host when an instance has used up its time limit: select_kernel retires it
  // Share of a round's batch per instance: select_kernel publishes how many nodes every slot could use (its live near list),
  // share_kernel turns the demands into the shares of the next round.  Every instance is granted a small base share (all make
  // progress; the easy ones - a handful of open nodes - finish within a few rounds).  The rest of the batch goes to the
  // instances IN ORDER OF ADMISSION, each up to min(demand, share_cap): the hard instances of this workload need 10^5..10^6
  // node relaxations whatever the width of their rounds (measured: +10..30 % nodes at 4096 per round against 64, +50..150 % at
  // 16384), so an instance that gets ~1000 nodes per round is done in a few hundred rounds, while hundreds of hard instances
  // sharing the batch evenly all run into their max_solution_time (which counts from the admission) and their work is lost.
  // Earliest deadline first finishes what it starts; the cap (1024: narrow rounds waste the fewest nodes - every round prunes
  // with the incumbents of the one before) keeps a pathological instance to 3 % of the device.
  int* slot_demand; int* slot_take; int share_cap; int base_take; int floor_pct; int young_nodes; int pump_max; int pump_inc; int window_pct; double probe_room;
  double probe_margin;           // > 0: the rounding probe leaves front-point environment / obstacle disjunctions undecided whose completed alternative holds with this much room
  int probe_itcap0;              // the same while the instance has no incumbent
  int defer_cap;                 // three and four cars: a node (not a probe, not a root) still unconverged after this many iterations is put back on its list with its iterate as its warm start and finished in a later round - ONCE (pool_big bit 3); 0: never
  int probe_itcap;               // iterations after which an unconverged rounding probe is abandoned (0: never)
  int probe_every;               // rounding probes are eligible every probe_every-th round (1: always)
  int det_ties;                  // 1: ties of the node selection are broken by the nodes' own low key bits and sibling preference (reproducible), 0: by arrival
  int live_inc;                  // 1: node evaluation prunes with the incumbent as other nodes of the same round update it (order dependent); 0: with the incumbent of the start of the round (reproducible)
  const int* root_cnt; const int* root_node; const int* root_depth; int root_stride;   // root records of every instance (uploaded once; admit_kernel writes them into the slot's list)
  double qp_tol;
  int use_cutoff;                // 0: solve every node to convergence (polish of the incumbent, solve_fixed)
  double cut_gate;               // the early cutoff is tested once the stationarity residual is below cut_gate x (1 + |objective|) (what is left of it enters the test with the instance's diameter: rigorous at any value)
  int seq_kinds;                 // bit k set: first-deviation (time family) branching for disjunction kind k, else single step
  int abl;                       // ablation mask of the diagnostic build (0 otherwise)
  int opt2;                      // MIQP_OPT2: bit 0 rounding probe at every branched node; bits 4.. = K: probe at nodes where at most K lanes of the completion saw a violated disjunction (default 8; 0 = only until the first incumbent); bits 2..3 = s: only every 4^s-th such node (by a hash of its record number); bits 8.. = largest violation, in units of 0.05, a probed node may show; bit 1: dives prefer the sibling with the smallest lifted bound; bit 16: probes leave the front-point environment / obstacle disjunctions undecided (all measured: no gain); bits 20 / 21: ties between disjunctions of equal branching priority go to the earliest / latest step whatever the car (default: the earliest step of the first car; latest: 2.4 x the work)
  int* work_counter;             // next node of the batch to be solved (reset before every ipm launch)
  unsigned long long* prof;      // [40] cycle counters of the phases of ipm_kernel (diagnostic build -DMIQP_PROFILE only)
  unsigned long long* stat_rowiters;
  int* ovf_count; int* ovf_list;  // nodes the on-chip interior point kernel handed to the memory-backed one (more general rows than its LDS holds)
  int* ovf2_count; int* ovf2_list;   // nodes the larger variant of the on-chip kernel handed on in its turn
  unsigned char* batch_large;    // per batch slot, written by select_kernel: the node goes to the concurrent launch of the larger on-chip variant (rounding probe, or its record is
                                 // marked).  A snapshot taken BEFORE the two launches fork: both read only this byte, so a node is never eligible for both in one round
                                 // (a record the standard kernel marks at its decode is large from the NEXT round on)
  unsigned char* pool_big;       // per record: 1 = more general rows than the standard on-chip kernel holds (found by that kernel at its decode, inherited by the children)
  int bounce;                    // 1: the standard on-chip kernel does not hand such a node on but marks it (pool_big) and returns it unsolved (batch_ok 5): eval_kernel
                                 // puts it back on its list, and from the next round on the larger variant takes it in its concurrent launch
  int ovf_mode;                  // 1: ipm_kernel works through ovf_list instead of the whole batch; 2: through the batch, rounding probes only; 3: through the batch, the marked records with pool_big bit 1
  int skip_probes;               // on-chip kernel: the rounding probes of the batch (depth word: sibling preference 63) are solved by a concurrent launch of ipm_kernel (ovf_mode 2)
  unsigned long long* stats;     // [32] diagnostic counters of the on-chip kernel (MIQP_STATS=1), else null
  signed char* pool_origin;      // diagnostic build: 2*kind + (deviating child) of the branching that created a node record
  // Warm start of the node relaxations: every child record carries the primal solution Z of its parent (a trajectory that
  // satisfies the dynamics and every row of the parent; only the rows the branching adds are violated).  The interior point
  // starts there with every row centred at complementarity ws_mu - slack s = max(residual, ws_delta), elastic slack t = s -
  // residual, multiplier ws_mu / s - instead of at the free rollout with the same multiplier on every row.
  double* pool_Z;                // [z_cap][N * nz] in the model's column order (null: cold starts only); records beyond z_cap start cold
  double ws_mu, ws_delta; int ws_on; int z_cap;
  // Local search around a new incumbent (lns_kernel): select_kernel raises inst_lns when it adopts a new incumbent; the neighbours of
  // its region sequences join the batch of the same round
  double* inst_lns_obj; double lns_step;   // the local search runs again only when the incumbent has improved by lns_step (relative) since its last run: a hill climb in steps of 1e-5 re-evaluates the whole neighbourhood for nothing
  int* inst_lns; int lns_mode; int lns_min_nodes; int lns_narrow;   // (an instance gets its local search once it has cost lns_min_nodes node relaxations: the easy ones are done before)
  // Dual active-set launch (as_onchip.hip) in place of the standard interior point launch of a round: it solves the ordinary nodes; one it cannot
  // finish is marked (pool_big bit 2: this record only, not inherited) and returned unsolved like a node that is too large (batch_ok 5), so that
  // the larger interior point variant takes it in its concurrent launch of the next round
  unsigned short* batch_A;       // [batch_cap][64] final active set of a node the active-set launch solved: box rows by their key (stage * 2 + side) * 16 + column, general rows as 1024 + (stage * NSLOT + slot); 0xFFFF: empty
  unsigned short* pool_A;        // [z_cap][64] the parent's, per child record (eval_kernel copies it like pool_Z): the child's first active set
  // M = the inverse of the active rows' Schur complement, packed triangle over the slots in rank order (n (n + 1) / 2 doubles), goes from a node to
  // its children through a RING of doubles: the solve of the node allocates its triangle at the head (a 64-bit count of doubles that only grows) and
  // writes it; the tag (start count << 8 | rows) travels with the children's records; nobody frees anything - a triangle is whole as long as the
  // head has not come within ring_margin of a full turn past its start, which a child checks against the head before it loads
  double* ring_M; unsigned long long* ring_head; unsigned long long ring_doubles, ring_margin;
  unsigned long long* batch_Mtag;   // [batch_cap]
  unsigned long long* pool_Mtag;    // [z_cap]
  int width_cap;                 // > 0: nodes a single solve takes per round at most (the host widens it when the solve is bound-limited)
  int as_probe_first;            // 1: a rounding probe that may be re-rounded goes to the active-set launch first and to the interior point only when it turns out infeasible (a round later)
  int as_chunk;                  // consecutive batch slots a wavefront of the active-set launches takes at a time
  int* cls_list;                 // [3][batch_cap] batch slots of large_class 1 / 2 / 3, appended by select_kernel / lns_kernel as they write batch_large: the larger launches of a round
                                 // take their nodes from these lists instead of scanning the whole batch through their hand-out counter (98 k same-address atomics = 1.1 ms each, tools/atomic_lab.hip)
  int* cls_count;                // [3] their lengths (a set of the round's counters)
  int cls_take;                  // per launch: 0 the whole batch (with the filters), c: the list of class c
  int lns_warm;                  // 1: the leaves of the local search start from the incumbent's active set (inc_A, inc_Mtag)
  int as_quota;                  // hand-outs after which a wavefront of the standard active-set launch leaves (0: it stays until the batch is handed out); the grid is sized to match
  int as_split;                  // 1: the larger interior point variant takes only the nodes of large_class 2 (class 1: the larger active-set launch on the third stream, class 3: the memory-backed launch on the fourth)
  unsigned long long* as_stats;  // [8] nodes, steps (rows added + dropped), handed to the interior point, rows dropped, infeasible, cut off, sum of the final active set sizes
};

__device__ inline unsigned long long d2key(double v) {
  unsigned long long u = (unsigned long long)__double_as_longlong(v);
  return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}
__device__ inline double key2d(unsigned long long k) {
  unsigned long long u = (k & 0x8000000000000000ull) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
  return __longlong_as_double((long long)u);
}

// objective stored in an incumbent key (low 20 bits = batch slot); 1e300 when no incumbent exists
__device__ inline double inc_from_key(unsigned long long key) {
  if (key >= 0xFFF0000000000000ull) return 1e300;
  return key2d(key & ~0xFFFFFull);
}

// cross-lane movement without the LDS crossbar: DPP within a row of 16 lanes, v_permlane16_swap / v_permlane32_swap
// (gfx950) between the rows - a full-rate VALU instruction each instead of a ds_bpermute round trip
template <int CTRL> __device__ inline double dpp_mov(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
// (value of the even row of the pair, value of the odd row) for the row pairs (0,1), (2,3): lane l and lane l ^ 16
__device__ inline void rows16(double x, double& even, double& odd) {
  const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
  const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  even = __hiloint2double((int)b[0], (int)a[0]); odd = __hiloint2double((int)b[1], (int)a[1]);
}
// (value of the lower half, value of the upper half): lane l and lane l ^ 32
__device__ inline void halves32(double x, double& lower, double& upper) {
  const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  lower = __hiloint2double((int)b[0], (int)a[0]); upper = __hiloint2double((int)b[1], (int)a[1]);
}
__device__ inline double sum_xor16(double v) { double a, b; rows16(v, a, b); return a + b; }      // v + v(lane ^ 16), in every lane
__device__ inline double sum_xor32(double v) { double a, b; halves32(v, a, b); return a + b; }     // v + v(lane ^ 32)
__device__ inline double wave_min(double v) {
  v = fmin(v, dpp_mov<0xB1>(v)); v = fmin(v, dpp_mov<0x4E>(v)); v = fmin(v, dpp_mov<0x141>(v)); v = fmin(v, dpp_mov<0x140>(v));
  double a, b; rows16(v, a, b); v = fmin(a, b); halves32(v, a, b); return fmin(a, b);
}
__device__ inline double wave_max(double v) {
  v = fmax(v, dpp_mov<0xB1>(v)); v = fmax(v, dpp_mov<0x4E>(v)); v = fmax(v, dpp_mov<0x141>(v)); v = fmax(v, dpp_mov<0x140>(v));
  double a, b; rows16(v, a, b); v = fmax(a, b); halves32(v, a, b); return fmax(a, b);
}
__device__ inline double wave_sum(double v) {   // quad, quad pairs, half rows, rows, row pairs, halves
  v += dpp_mov<0xB1>(v); v += dpp_mov<0x4E>(v); v += dpp_mov<0x141>(v); v += dpp_mov<0x140>(v);
  v = sum_xor16(v); return sum_xor32(v);
}

// alpha*X + beta*Y of car c added to a dense row over the stage vector
// rt == nullptr (the region of the step is undecided): the offset of a front point is replaced by its bound over the regions
// still possible, K = d_fpk entry of the car's region set (host_inst.hpp) - the minimum where the offset enters the row
// "g.z <= rhs" with a positive coefficient, the maximum otherwise: a row implied by the exact row of whichever region is active
__device__ inline void add_point(double* g, double& rhs, int c, const double* rt, int tx, int ty, double al, double be, const double* K = nullptr) {
  g[6 * c + 0] += al;
  if (tx != PT_R) {
    if (rt) { const double* p = rt + 19 + (tx == PT_U ? 0 : 3); rhs -= al * p[0]; g[6 * c + 1] += al * p[1]; g[6 * c + 4] += al * p[2]; }
    else if (al != 0.0) rhs -= al * K[(tx == PT_U ? 0 : 2) + (al > 0.0 ? 0 : 1)];
  }
  g[6 * c + 3] += be;
  if (ty != PT_R) {
    if (rt) { const double* p = rt + 25 + (ty == PT_U ? 0 : 3); rhs -= be * p[0]; g[6 * c + 1] += be * p[1]; g[6 * c + 4] += be * p[2]; }
    else if (be != 0.0) rhs -= be * K[(ty == PT_U ? 4 : 6) + (be > 0.0 ? 0 : 1)];
  }
}


// corner index of a front point with x/y types (tx, ty) in the environment / obstacle corner orders
__device__ inline int env_corner(int tx, int ty) { return tx == PT_U ? (ty == PT_U ? 1 : 3) : (ty == PT_U ? 2 : 4); }
__device__ inline int obs_corner(int tx, int ty) { return tx == PT_L ? (ty == PT_L ? 1 : 3) : (ty == PT_L ? 2 : 4); }

// the row "al*X + be*Y <= gamma" on front corner (tx, ty) is implied by the same row on the worst corner (tx*, ty*)
// when that corner carries the same row and UB >= LB is proven for every coordinate in which the corners differ
__device__ inline bool corner_dominated(int tx, int ty, double al, double be, int domflag, int& txs, int& tys) {
  txs = al > 0.0 ? PT_U : (al < 0.0 ? PT_L : tx);
  tys = be > 0.0 ? PT_U : (be < 0.0 ? PT_L : ty);
  if (txs == tx && tys == ty) return false;
  if (txs != tx && !(domflag & 1)) return false;
  if (tys != ty && !(domflag & 2)) return false;
  return true;
}

// cars (c1 < c2) of pair p in the order c1 = 0: c2 = 1..C-1, c1 = 1: ...
__device__ inline void pair_cars(int p, int C, int& c1, int& c2) {
  c1 = 0; int rem = p;
  while (rem >= C - 1 - c1) { rem -= C - 1 - c1; ++c1; }
  c2 = c1 + 1 + rem;
}

struct RowOut { double rhs; double aq; bool active; };
// depth word of a rounding probe: tree depth >= 1, low bits 63 (eval_kernel: the children of a node carry 62 - their preference)
// stage of a byte of the fix record (region codes, environment / obstacle / car-car disjunctions; -1: the masks behind them)
__device__ inline int fix_stage(const Layout& Y, int k) {
  const int N = Y.N;
  if (k < Y.f_env) return (k - Y.f_reg) % N;
  if (k < Y.f_obs) return ((k - Y.f_env) / 5) % N;
  if (k < Y.f_c2c) return ((k - Y.f_obs) / 5) % N;
  if (k < Y.f_c2n) return ((k - Y.f_c2c) / 4) % N;
  return -1;
}
__device__ inline bool is_probe_word(int dw) { return (dw & 63) == 63 && (dw >> 6) >= 1; }
// The nodes of a round go to one of four concurrent launches, and which one is decided ONCE, by select_kernel / lns_kernel when they write the batch
// (batch_large: nothing a launch of the round writes - the marks on the records, the incumbents - can then change the split while the launches
// run beside each other; a decision read live from pool_big made the iteration counts of repeated solves differ):
//   0 the ordinary nodes: the standard launch;
//   1 the large nodes - rounding probes, local-search leaves, marked records - that the active-set method takes: its larger block;
//   2 ... that the interior point keeps: records the active-set method failed on (pool_big bit 2), and the rounding probes of an instance whose
//     infeasible probes are re-rounded (eval_kernel: no incumbent yet, or pump_inc) - the re-rounding starts from the least-violation point that
//     only the elastic interior point delivers;
//   3 records known to exceed even the larger on-chip block (pool_big bit 1): the memory-backed kernel.
// (Without the active-set launches - one, three, four cars, MIQP_AS=0 - only zero / non-zero matters.)
#ifdef MIQP_PROFILE
// one launch from the middle of a stream, wavefront by wavefront (MIQP_WAVE_DUMP, tools/wave_dump.py): start, end, where, nodes; kind 0 the standard
// active-set launch (written there), 1 its larger block, 2 the larger interior point variant, 3 the memory-backed kernel
struct WaveDump {
  const DevBuf& B; int kind; unsigned long long r0; int nodes;
  __device__ WaveDump(const DevBuf& b, int k) : B(b), kind(k), r0(__builtin_amdgcn_s_memrealtime()), nodes(0) {}
  __device__ ~WaveDump() {
    if (threadIdx.x == 0 && B.prof[125] == 300 && blockIdx.x < 4096) {
      unsigned int hw, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      unsigned long long* const w = B.prof + 160 + 4 * (kind * 4096 + blockIdx.x);
      w[0] = r0; w[1] = __builtin_amdgcn_s_memrealtime(); w[2] = ((unsigned long long)xcc << 32) | hw; w[3] = (unsigned long long)nodes;
    }
  }
};
#define WAVE_DUMP(kind) WaveDump wd_(B, kind)
#define WAVE_DUMP_NODE() (++wd_.nodes)
#else
#define WAVE_DUMP(kind)
#define WAVE_DUMP_NODE()
#endif
constexpr int CTR_SET = 16;   // counters of a round: [0] batch count, [1] .. [6] the launches' hand-out and hand-over counters, [8] .. [10] lengths of the class lists
__device__ inline unsigned char large_class(const DevBuf& B, int rec, int dw, bool has_inc) {
  const unsigned int pb = B.pool_big ? (unsigned int)B.pool_big[rec] : 0u;
  const bool probe = is_probe_word(dw);
  if (!probe && !(pb & 0xF7u)) return 0;
  if (pb & 2u) return 3;
  if (pb & 4u) return 2;
  if (probe && B.pump_max > 0 && (!has_inc || B.pump_inc) && !B.as_probe_first) return 2;
  return 1;
}

// region set of (car, step) in a node: the bits of its fix record that the static reachability presolve allows
__device__ inline int region_set(const Layout& Y, const int* T, const signed char* fix, int c, int i) {
  const int m = (int)(unsigned char)fix[Y.f_rmask + 2 * (c * Y.N + i)] | ((int)(unsigned char)fix[Y.f_rmask + 2 * (c * Y.N + i) + 1] << 8);
  return m & T[Y.i_rallow + c * Y.N + i];
}
// acceleration / jerk box of an undecided step: the hull over the regions the node still allows there (host_inst.hpp: d_hullm),
// or the static hull of the step when the car has more possible regions than the table indexes
__device__ inline const double* region_hull(const Layout& Y, const double* D, const int* T, const signed char* fix, int c, int i) {
  if (T[Y.i_nposs + c] > Y.PT) return D + Y.d_hull + (c * Y.N + i) * HULLSZ;
  return D + Y.d_hullm + (c * (1 << Y.PT) + region_set(Y, T, fix, c, i)) * 8;
}
// front-point offset ranges of (car, step) over its region set (nullptr: the car has more possible regions than the table indexes)
__device__ inline const double* region_fpk(const Layout& Y, const double* D, const int* T, const signed char* fix, int c, int i) {
  if (T[Y.i_nposs + c] > Y.PT) return nullptr;
  return D + Y.d_fpk + (c * (1 << Y.PT) + region_set(Y, T, fix, c, i)) * 8;
}

// rows of the relaxation of a node (only alternatives that are fixed); g is this lane's LDS row.  BUILD = false only
// decides whether slot (i, slot) carries a row (same tests, nothing written): the kernel first collects the active
// slots of a node and then builds 64 rows at a time with every lane busy.
template <int C, bool BUILD = true>
__device__ inline RowOut decode_row(const Layout& Y, const double* D, const int* T, const signed char* fix, int i, int slot, double* g) {
  constexpr int NZ = 8 * C;
  RowOut r; r.rhs = 0; r.aq = 0; r.active = false;
  if (BUILD) {
#pragma unroll
    for (int q = 0; q < NZ; ++q) g[q] = 0.0;
  }
  const int N = Y.N;
  const double* G = D + Y.d_glob;
  if (slot < C * Y.SC) {
    int c = slot / Y.SC, rr = slot - c * Y.SC;
    int code = i >= 1 ? (int)fix[Y.f_reg + c * N + i] : -1;
    const double* rt = code >= 0 ? D + Y.d_reg + (c * Y.P + (code >> 2)) * REGSZ : nullptr;
    const double* Hc = D + Y.d_hull + (c * N + i) * HULLSZ;   // undecided region: the cone rows of the alternatives that can still hold (host_inst.hpp; off by default)
    if (rr < 7) {
      if (i < 1 || ((T[Y.i_boxskip + c * N + i] >> rr) & 1)) return r;   // implied by the earlier steps (host box presolve)
      r.active = true;
      if (!BUILD) return r;
      switch (rr) {
        case 0: g[6 * c + 1] = -1; r.rhs = -G[0]; break;
        case 1: g[6 * c + 4] = -1; r.rhs = -G[0]; break;
        case 2: g[6 * c + 1] = 1; r.rhs = G[1]; break;
        case 3: g[6 * c + 2] = 1; r.rhs = rt ? rt[12] : region_hull(Y, D, T, fix, c, i)[1]; break;
        case 4: g[6 * c + 2] = -1; r.rhs = -(rt ? rt[11] : region_hull(Y, D, T, fix, c, i)[0]); break;
        case 5: g[6 * c + 5] = 1; r.rhs = rt ? rt[14] : region_hull(Y, D, T, fix, c, i)[3]; break;
        default: g[6 * c + 5] = -1; r.rhs = -(rt ? rt[13] : region_hull(Y, D, T, fix, c, i)[2]); break;
      }
      return r;
    }
    if (rr < 11) {
      if (i > N - 2) return r;
      int s = (rr - 7) >> 1, up = ((rr - 7) & 1) == 0;
      double lo, hi;
      if (i == 0) { lo = D[Y.d_u0box + c * 4 + 2 * s]; hi = D[Y.d_u0box + c * 4 + 2 * s + 1]; }
      else if (rt) { lo = rt[15 + 2 * s]; hi = rt[16 + 2 * s]; }
      else if (!BUILD) { lo = hi = 0.0; }
      else { const double* Hm = region_hull(Y, D, T, fix, c, i); lo = Hm[4 + 2 * s]; hi = Hm[5 + 2 * s]; }
      r.active = true;
      if (!BUILD) return r;
      g[6 * C + 2 * c + s] = up ? 1.0 : -1.0; r.rhs = up ? hi : -lo;
      return r;
    }
    if (rr < 16) {
      if (i < 1) return r;
      if (code < 0) {   // undecided region: the velocity stays inside the cone around the sectors that can still hold
        const int k = rr - 11;
        if (k > 1 || Hc[14] == 0.0) return r;
        r.active = true; r.rhs = Hc[10 + 3 * k];
        if (!BUILD) return r;
        g[6 * c + 1] = Hc[8 + 3 * k]; g[6 * c + 4] = Hc[9 + 3 * k];
        return r;
      }
      int h = code & 3, k = rr - 11;
      if (h == 3) {
        if (k > 3) return r;
        r.active = true; r.rhs = G[6];
        if (!BUILD) return r;
        g[6 * c + (k < 2 ? 1 : 4)] = (k & 1) ? -1.0 : 1.0;
        return r;
      }
      r.active = true;
      if (!BUILD) return r;
      switch (k) {
        case 0: g[6 * c + 1] = rt[0]; g[6 * c + 4] = rt[1]; break;
        case 1: g[6 * c + 1] = rt[2]; g[6 * c + 4] = rt[3]; break;
        case 2: {
          const int* hs = T + Y.i_hs + ((c * Y.P + (code >> 2)) * 2 + h) * 2;
          g[6 * c + (hs[0] == 0 ? 1 : 4)] = -(double)hs[1]; r.rhs = -G[6];
        } break;
        case 3: g[6 * c + 5] = 1; g[6 * c + 2] = -rt[4]; g[6 * c + 1] = -rt[6]; g[6 * c + 4] = -rt[7]; r.rhs = rt[5]; break;
        default: g[6 * c + 5] = -1; g[6 * c + 2] = rt[4]; g[6 * c + 1] = rt[9]; g[6 * c + 4] = rt[10]; r.rhs = -rt[8]; break;
      }
      return r;
    }
    int q = rr - 16;
    if (q < 5 * Y.EL) {
      if (i < 1 || Y.E < 1) return r;
      int pt = q / Y.EL, k = q - pt * Y.EL;
      int e = Y.E == 1 ? 0 : (int)fix[Y.f_env + (c * N + i) * 5 + pt];
      if (e < 0 || k >= T[Y.i_envn + e] || (pt > 0 && code < 0)) return r;
      const double* ed = D + Y.d_env + (e * Y.EL + k) * 3;
      if (pt > 0) {  // exact presolve: rows on dominated corners of the front box are implied
        int txs, tys;
        if (corner_dominated(ENV_PT_D[pt][0], ENV_PT_D[pt][1], ed[0], ed[1], T[Y.i_dom + (c * Y.P + (code >> 2)) * 4 + (code & 3)], txs, tys)) {
          int ps = env_corner(txs, tys);
          int es = Y.E == 1 ? 0 : (int)fix[Y.f_env + (c * N + i) * 5 + ps];
          if (es == e) return r;
        }
      }
      r.active = true; r.rhs = ed[2];
      if (!BUILD) return r;
      add_point(g, r.rhs, c, rt, ENV_PT_D[pt][0], ENV_PT_D[pt][1], ed[0], ed[1]);
      return r;
    }
    q -= 5 * Y.EL;
    int o = q / 5, pt = q - o * 5;
    if (i < 1) return r;
    int kk = (int)fix[Y.f_obs + ((c * Y.O + o) * N + i) * 5 + pt];
    if (kk < 0 || kk >= Y.L || (pt > 0 && code < 0)) return r;
    const double* ed = D + Y.d_obs + ((o * N + i) * Y.L + kk) * 3;
    if (pt > 0) {
      int txs, tys;
      if (corner_dominated(OBS_PT_D[pt][0], OBS_PT_D[pt][1], ed[0], ed[1], T[Y.i_dom + (c * Y.P + (code >> 2)) * 4 + (code & 3)], txs, tys)) {
        int ps = obs_corner(txs, tys);
        if ((int)fix[Y.f_obs + ((c * Y.O + o) * N + i) * 5 + ps] == kk) return r;
      }
    }
    r.active = true; r.rhs = ed[2];
    if (!BUILD) return r;
    add_point(g, r.rhs, c, rt, OBS_PT_D[pt][0], OBS_PT_D[pt][1], ed[0], ed[1]);
    return r;
  }
  if (C < 2 || i < 1) return r;
  int q = slot - C * Y.SC;
  // car/car rows.  Slots [0, 8 NP): the fixed alternative of every group (hard cap row and,
  // for the soft groups, the quadratic-soft row).  Slots [8 NP, 24 NP): exclusion rows - alternative a of the group is
  // asserted NOT to hold at zero slack (first-deviation children exclude the alternatives of their earlier siblings:
  // a trajectory that satisfies one of those for free is covered, at no greater cost, by that sibling).
  const bool excl = q >= Y.NP * 8;
  int p, grp, which, alt;
  if (!excl) { p = q >> 3; grp = (q & 7) >> 1; which = q & 1; alt = (int)fix[Y.f_c2c + (p * N + i) * 4 + grp]; if (alt < 0) return r; }
  else {
    int q2 = q - Y.NP * 8; p = q2 >> 4; grp = (q2 >> 2) & 3; alt = q2 & 3; which = 1;
    int m = (int)fix[Y.f_c2n + (p * N + i) * 4 + grp];
    if (m <= 0 || !((m >> alt) & 1)) return r;
  }
  int c1, c2; pair_cars(p, C, c1, c2);
  int code1 = (int)fix[Y.f_reg + c1 * N + i], code2 = (int)fix[Y.f_reg + c2 * N + i];
  bool need1 = grp >= 2, need2 = (grp == 1 || grp == 3);
  // a front point of a car whose region is undecided: the row stays in the relaxation with the offset bounded over the car's
  // region set (add_point); the exclusion rows (they bound from the other side) wait for the region
  const double* K1 = nullptr; const double* K2 = nullptr;
  if (need1 && code1 < 0) { if (excl || Y.relax_front_off) return r; K1 = region_fpk(Y, D, T, fix, c1, i); if (!K1) return r; }
  if (need2 && code2 < 0) { if (excl || Y.relax_front_off) return r; K2 = region_fpk(Y, D, T, fix, c2, i); if (!K2) return r; }
  const double* rt1 = code1 >= 0 ? D + Y.d_reg + (c1 * Y.P + (code1 >> 2)) * REGSZ : nullptr;
  const double* rt2 = code2 >= 0 ? D + Y.d_reg + (c2 * Y.P + (code2 >> 2)) * REGSZ : nullptr;
  double Dsep = D[Y.d_dsep + p * N + i], S = D[Y.d_ssl + i], smax = D[Y.d_smax + i], wsl = D[Y.d_misc + 0];
  bool soft = (grp == 0 || grp == 3);
  if (!excl) {
    if (!soft && which == 1) return r;
    if (soft && which == 1 && !(smax > 0 && wsl > 0)) return r;
  }
  bool isx = alt < 2, lo = (alt == 0 || alt == 2);
  int ca, cb, ta, tb;
  if (grp == 0) { ta = tb = PT_R; ca = lo ? c1 : c2; cb = lo ? c2 : c1; }
  else if (grp == 1) { if (lo) { ca = c1; ta = PT_R; cb = c2; tb = PT_L; } else { ca = c2; ta = PT_U; cb = c1; tb = PT_R; } }
  else if (grp == 2) { if (lo) { ca = c2; ta = PT_R; cb = c1; tb = PT_L; } else { ca = c1; ta = PT_U; cb = c2; tb = PT_R; } }
  else { if (lo) { ca = c2; ta = PT_U; cb = c1; tb = PT_L; } else { ca = c1; ta = PT_U; cb = c2; tb = PT_L; } }
  double al = isx ? 1.0 : 0.0, be = isx ? 0.0 : 1.0;
  r.active = true;
  if (!BUILD) return r;
  r.rhs = soft ? (which == 0 ? -(Dsep + S) + smax : -(Dsep + S)) : -Dsep;
  r.aq = (!excl && soft && which == 1) ? 2.0 * wsl : 0.0;
  add_point(g, r.rhs, ca, ca == c1 ? rt1 : rt2, ta, ta, al, be, ca == c1 ? K1 : K2);
  add_point(g, r.rhs, cb, cb == c1 ? rt1 : rt2, tb, tb, -al, -be, cb == c1 ? K1 : K2);
  if (excl) {   // g.z >= rhs
#pragma unroll
    for (int k = 0; k < NZ; ++k) g[k] = -g[k];
    r.rhs = -r.rhs;
  }
  return r;
}

// entry (q, b) of [A B] of the triple integrator chains (model_region_constraints.mod:11-19)
template <int C>
__device__ inline double ab_entry(int q, int b, double ts) {
  constexpr int NX = 6 * C;
  int ch = q / 3, k = q - 3 * ch;
  if (b < NX) {
    int chb = b / 3, kb = b - 3 * chb;
    if (chb != ch || kb < k) return 0.0;
    int d = kb - k;
    return d == 0 ? 1.0 : (d == 1 ? ts : 0.5 * ts * ts);
  }
  if (b - NX != ch) return 0.0;
  return k == 0 ? ts * ts * ts / 6.0 : (k == 1 ? 0.5 * ts * ts : ts);
}

// ------------------------------------------------------------------------------------------------
struct RowRegs { double aq, col, s, lam, t, v[6]; };
__device__ inline RowRegs load_row(const double* rc_aq, const double* rc_col, const double* rc_v, const double* rs_s, const double* rs_l,
                                   const double* rs_t, int rowcap, int idx) {
  RowRegs R;
  R.aq = rc_aq[idx]; R.col = rc_col[idx]; R.s = rs_s[idx]; R.lam = rs_l[idx]; R.t = rs_t[idx];
#pragma unroll
  for (int k = 0; k < 6; ++k) R.v[k] = rc_v[(size_t)k * rowcap + idx];
  return R;
}

// Workgroup barrier for data exchanged through LDS only: __syncthreads() also waits for every global load and store of the wavefront
// (vmcnt(0): the fence in front of s_barrier covers global memory) - inside the backward sweep of three and four cars that would be the write
// latency of the gains and the read latency of the prefetched rows of the next stage, once per barrier.
__device__ inline void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// reductions over the one wavefront that solves a node (the xor butterfly leaves the result in every lane)
// reductions over the workgroup of a node: one wavefront (two cars and fewer), or four (three and four cars: `red` holds the partial results
// of the wavefronts; the barrier in front keeps the result of the previous reduction readable until every thread has it)
template <int NT> __device__ inline double block_sum(double v, double* red) {
  v = wave_sum(v);
  if constexpr (NT > 64) { __syncthreads(); if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v; __syncthreads(); v = 0.0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) v += red[w]; }
  return v;
}
template <int NT> __device__ inline double block_min(double v, double* red) {
  v = wave_min(v);
  if constexpr (NT > 64) { __syncthreads(); if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v; __syncthreads(); v = red[0];
#pragma unroll
    for (int w = 1; w < NT / 64; ++w) v = fmin(v, red[w]); }
  return v;
}
template <int NT> __device__ inline double block_max(double v, double* red) {
  v = wave_max(v);
  if constexpr (NT > 64) { __syncthreads(); if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v; __syncthreads(); v = red[0];
#pragma unroll
    for (int w = 1; w < NT / 64; ++w) v = fmax(v, red[w]); }
  return v;
}

#ifdef MIQP_PROFILE
#define PROF_T(var) long long var = clock64()
#define PROF_ACC(k, t0, t1) pr_[k] += (unsigned long long)((t1) - (t0))
#else
#define PROF_T(var)
#define PROF_ACC(k, t0, t1)
#endif

__device__ inline double readlane_d(double v, int lane) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// 1/x from v_rcp_f64 and one Newton step (relative error ~1e-13; the interior point directions tolerate far more, and
// every place that needs the same quantity twice recomputes it with the same instructions)
__device__ inline double frcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  return fma(r, fma(-x, r, 1.0), r);
}
// 1/sqrt(x) from v_rsq_f64 and one Newton step
__device__ inline double frsq(double x) {
  double r = __builtin_amdgcn_rsq(x);
  return fma(0.5 * r, fma(-x * r, r, 1.0), r);
}

// initial state (s, lambda, t) of an elastic row with residual c = rhs - g.z at the starting point (s - t = c at every iterate).
// Cold (free rollout): the slack follows the residual, the same multiplier everywhere.  Warm (the parent's solution): the row is
// put on the central path at mu0 - an inactive row (large residual) gets a small multiplier, an active or violated one a slack
// of delta and the matching multiplier (capped well inside (0, rho)).
__device__ inline void init_elastic(double c, bool warm, double mu0, double delta, double& s, double& lam, double& t) {
  if (!warm) {
    if (c > QP_T0) { t = QP_T0; s = c + QP_T0; } else { s = MIQP_S0 * QP_T0; t = s - c; }
    lam = MIQP_LAM0;
    return;
  }
  s = c > delta ? c : delta;   // (a larger floor for the rows the branching violates, and the parent's multipliers as the start, were tried in round 4: DESIGN.md 3.2 - no fewer iterations)
  lam = mu0 / s; if (lam > 0.5 * RHO_EL) lam = 0.5 * RHO_EL;
  double tt = mu0 / (RHO_EL - lam);          // central value of the elastic slack
  if (s - c > tt) tt = s - c;                // ... or what the violated row needs
  t = tt; s = c + tt;                        // (s - t = c exactly)
}

// ------------------------------------------------------------------------------------------------
//  Newton step of one row (elastic: aq == 0, quadratic-soft: aq > 0) given g.dz
__device__ inline void row_step(double s, double lam, double t, double aq, double gd, double tau, double& ds, double& dl, double& dt) {
  // three reciprocals per row: 1/lambda, 1/mu (or 1/aq), 1/D
  const double il = frcp(lam);
  const double r1 = tau - s * lam;
  double zz, r2m = 0.0, im = 0.0, r2 = 0.0;
  if (aq == 0.0) { const double mu = RHO_EL - lam; im = frcp(mu); zz = t * im; r2 = tau - t * mu; r2m = r2 * im; }
  else zz = frcp(aq);
  const double w = frcp(s * il + zz);
  dl = (gd + r1 * il - r2m) * w;
  ds = (r1 - s * dl) * il;
  dt = aq == 0.0 ? (r2 + t * dl) * im : 0.0;
}

// ------------------------------------------------------------------------------------------------
//  interior point kernel: one wavefront per node.  The rows of the node are decoded once, compacted per stage
//  (only active rows are stored) and kept as sparse rows (<= 6 non-zeros) in a block-indexed cache.
#ifndef MIQP_WIDE_WPE
#define MIQP_WIDE_WPE 2   // wavefronts per SIMD the multi-wavefront kernel of three and four cars is register-allocated for
#endif
#ifndef MIQP_WIDE_NT
#define MIQP_WIDE_NT 128   // threads per node of the three / four car kernel: 64, 128 or 256
#endif
template <int C, int NT>
__global__ void __launch_bounds__(NT, (C <= 2 ? MIQP_IPM_WPE : (NT > 64 ? MIQP_WIDE_WPE : 1))) ipm_kernel(DevBuf B) {
  static_assert(C <= 4, "stage vectors of at most 32 entries");
  constexpr bool WIDE = C > 2;   // more than two cars: the stage does not fit one 16x16 MFMA tile -> dense LDS stage algebra
  // Three and four cars (round 5): one WORKGROUP OF TWO WAVEFRONTS per node (MIQP_WIDE_NT = 128; 256 = four wavefronts also builds).  The stage
  // vector of 24 / 32 entries makes every stage matrix a 2 x 2 tiling of 16 x 16 MFMA tiles: wavefront w owns the tiles t = 2 ti + tj with
  // t % NW == w of the three products of a Riccati step, the row loops (assembly, step length, update, decode test) run over all NT lanes, and
  // four nodes' workgroups still share a CU (the dense stage matrices in LDS are per node, not per wavefront: 37 KB at four cars) - 2 wavefronts
  // per SIMD instead of 1.  Measured on cfg5 (one MI355X, same box): node relaxations/s of a single solve 124 k -> 149 k, of sixteen solves in
  // flight 153 k -> 217 k; four wavefronts per node (two nodes per CU at 256 VGPRs) 124 k / 166 k.  NT = 64 remains the form of one and two cars.
  static_assert(NT == 64 || (WIDE && (NT == 128 || NT == 256)), "one wavefront per node, or two / four for the 2 x 2-tiled stage algebra");
  constexpr int NW = NT / 64;
  constexpr int NX = 6 * C, NU = 2 * C, NZ = 8 * C, GS = WIDE ? NZ + 1 : GSTR;
  constexpr int KB = (NX + 3) / 4;         // k blocks of the products with [A B]
  constexpr int RU = NX / 4, GU0 = NX % 4;  // register / first lane group holding the input rows NX..NZ-1  // rows are zero padded to the 16 columns of the MFMA tile
  const Layout& Y = B.Y;
  const int tid = threadIdx.x;
  WAVE_DUMP(3);
  const int* const clist = B.cls_take ? B.cls_list + (size_t)(B.cls_take - 1) * B.batch_cap : nullptr;   // (the list of this launch's class: nothing to scan)
  const int nbatch = clist ? (B.cls_count[B.cls_take - 1] < B.batch_cap ? B.cls_count[B.cls_take - 1] : B.batch_cap) : B.ovf_mode == 1 ? *B.ovf_count : (*B.batch_count < B.batch_cap ? *B.batch_count : B.batch_cap);   // select may over-count when the batch is full
  const int N = Y.N, NSLOT = Y.NSLOT;
  extern __shared__ double lds[];
  double* Z = lds;                       // [N][NZ]
  double* dZ = Z + N * NZ;               // [N][NZ]
  double* Gh = dZ + N * NZ;              // [GROWS][GS] scaled rows of the stage being contracted
  double* fs = Gh + GROWS * GS;          // [GROWS]
  double* Wd = Z + N * NZ + ipm_scratch_doubles(N, C);  // [NZ]
  double* red = Wd + NZ;                 // [8]
  double* dgq = red + 8;                 // [2][16] diagonal / gradient contributions of the single-entry rows of a stage
  int* sstart = (int*)(dgq + 32);        // [N+2] first multi-entry row of every stage (rows grow up from index 0)
  int* sst = sstart + ((N + 4) & ~1);    // [N+2] single-entry rows before every stage (they grow down from ROWCAP-1)
  signed char* fix = (signed char*)(sst + ((N + 4) & ~1));  // [fixlen]
  double* dscr = dZ;                     // decode phase: one dense scratch row per lane
  constexpr int KSTR = NX + 2;           // gain row: K[q][0..NX-1], k[q], pad
  double* KG = B.kgain + (size_t)blockIdx.x * N * NU * KSTR;
  __shared__ int sh_node;
  // per-row data of the node being solved, indexed by resident block (stays L2 / Infinity-Cache resident)
  double* RS = B.rowstate + (size_t)blockIdx.x * NFIELD * Y.ROWCAP;
  double* rs_s = RS, *rs_l = RS + Y.ROWCAP, *rs_t = RS + 2 * Y.ROWCAP, *rs_g = RS + 3 * Y.ROWCAP;
  double* RC = B.rowcache + (size_t)blockIdx.x * NCACHE * Y.ROWCAP;
  double* rc_rhs = RC, *rc_aq = RC + Y.ROWCAP, *rc_col = RC + 2 * Y.ROWCAP, *rc_v = RC + 3 * Y.ROWCAP;
  for (;;) {
  // dynamic distribution: the next unsolved node of the batch (solve times vary 4x between nodes)
  __syncthreads();
  if (tid == 0) sh_node = atomicAdd(B.work_counter, 1);
  __syncthreads();
  if (sh_node >= nbatch) break;
  const int node = __builtin_amdgcn_readfirstlane(clist ? clist[sh_node] : B.ovf_mode == 1 ? B.ovf_list[sh_node] : sh_node);   // wave-uniform: addressed from SGPRs
  if (B.ovf_mode == 2 && !is_probe_word(B.batch_depth[node])) continue;   // (this launch takes the rounding probes only)
  if (B.ovf_mode == 3 && B.batch_large[node] != 3) continue;   // (... the records known to exceed the larger on-chip block: large_class 3)
  WAVE_DUMP_NODE();
  const int inst = __builtin_amdgcn_readfirstlane(B.batch_inst[node]);
  const double* D = B.inst_d + (size_t)inst * Y.dstride;
  const int* T = B.inst_i + (size_t)inst * Y.istride;
  const double ts = D[Y.d_glob + 7];
  const double zdiam = D[Y.d_misc + 2];   // L1 diameter of the instance's reachable set (host box presolve): the weight of a stationarity residual in a bound
  {
    const signed char* src = B.pool_fix + (size_t)B.batch_node[node] * Y.fixlen;
    for (int k = tid; k < Y.fixlen; k += NT) fix[k] = src[k];
    for (int k = tid; k < NZ; k += NT) Wd[k] = D[Y.d_wd + k];
    for (int k = tid; k < N * NZ; k += NT) Z[k] = 0.0;
  }
  // roots start cold; ws_on == 2 (the polish of the incumbents): from the incumbent's own solution
  const bool warm = B.ws_on == 2 || (B.ws_on && B.pool_Z && (B.batch_depth[node] >> 6) >= 1 && B.batch_node[node] < B.z_cap);
  __syncthreads();
  if (warm) {
    const double* zp = B.ws_on == 2 ? B.inc_Z + (size_t)inst * N * NZ : B.pool_Z + (size_t)B.batch_node[node] * N * NZ;
    for (int k = tid; k < N * NZ; k += NT) Z[k] = zp[k];
    __syncthreads();
  } else {
  if (tid < NX) Z[tid] = D[Y.d_x0 + tid];
  __syncthreads();
  for (int i = 0; i + 1 < N; ++i) {  // free rollout (u = 0)
    if (tid < NX) {
      double acc = 0;
      for (int q = 3 * (tid / 3); q < 3 * (tid / 3) + 3; ++q) acc += ab_entry<C>(tid, q, ts) * Z[i * NZ + q];
      Z[(i + 1) * NZ + tid] = acc;
    }
    __syncthreads();
  }
  }
  const double* Rf = D + Y.d_ref;
  // cutoff: a node whose dual bound already exceeds what can still improve the incumbent by more than the gap is
  // abandoned (weak duality on the penalised QP; also catches infeasible nodes, whose penalty term is huge)
  double cutoff = 1e300;
  {
    const double inc0 = fmin(inc_from_key(*(volatile unsigned long long*)&B.inc_key[inst]), B.inc_ext[inst]);
    if (B.use_cutoff && inc0 < 1e300) cutoff = inc0 - (is_probe_word(B.batch_depth[node]) ? 0.0 : B.inst_gap[inst]) * (1e-10 + fabs(inc0)) - B.inst_const[inst];   // (a heuristic leaf is cut off at the incumbent itself: as_onchip.hip)
  }
  double tsum = 0.0;   // sum of the elastic slacks of the current iterate
  double abr[KB];  // [A B] as MFMA operand: lane (g, c) holds AB[4kb + g][c]
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) abr[kb] = (4 * kb + (tid >> 4) < NX && (tid & 15) < NZ) ? ab_entry<C>(4 * kb + (tid >> 4), tid & 15, ts) : 0.0;
#ifdef MIQP_PROFILE
  unsigned long long pr_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  int first_proxy = 0;
#endif
  PROF_T(tb0);

  // ---- decode every row of the node once; active rows are compacted per stage; initial row state
  // Rows with one non-zero and no quadratic slack (the boxes: most rows of a node) are kept apart from the general
  // rows: they touch only the diagonal of the stage Hessian and need one coefficient.
  // Two passes: (A) every (stage, slot) pair is tested for carrying a row (cheap: the tests of decode_row without
  // building anything) and the active pairs are compacted, in (stage, slot) order, into a per-block list; (B) the rows
  // are built 64 at a time with every lane busy (about one slot in five is active).
  double csum = 0.0; int cnt = 0; int base = 0, sbase = 0;
  int* cand = (int*)rs_g;   // the g.dz plane is free until the first step-length loop
  const unsigned long long lt = (1ull << (tid & 63)) - 1ull;   // the lanes of this wavefront below this one
  int ncand = 0;
  __shared__ int sh_wcnt[4], sh_nrows[2];
  for (int p0 = 0; p0 < N * NSLOT; p0 += NT) {
    const int pcode = p0 + tid;
    bool act = false;
    if (pcode < N * NSLOT) { const int i = pcode / NSLOT; act = decode_row<C, false>(Y, D, T, fix, i, pcode - i * NSLOT, nullptr).active; }
    const unsigned long long m = __ballot(act);
    if constexpr (NW == 1) {
      if (act) cand[ncand + __popcll(m & lt)] = pcode;
      ncand += __popcll(m);
    } else {   // four wavefronts: the list keeps its (stage, slot) order - wavefront w writes behind the active pairs of the wavefronts before it
      if ((tid & 63) == 0) sh_wcnt[tid >> 6] = __popcll(m);
      __syncthreads();
      int off = 0, tot = 0;
#pragma unroll
      for (int w = 0; w < NW; ++w) { if (w < (tid >> 6)) off += sh_wcnt[w]; tot += sh_wcnt[w]; }
      if (act) cand[ncand + off + __popcll(m & lt)] = pcode;
      ncand += tot;
      __syncthreads();
    }
  }
  for (int k = tid; k <= N; k += NT) { sstart[k] = 0; sst[k] = 0; }
  __syncthreads();
  // (pass B is the work of the first wavefront whatever NT: a dense scratch row per lane is 64 rows of LDS, and the pass is 7 % of a node's time)
  for (int c0 = 0; c0 < ncand && tid < 64; c0 += 64) {
    double* g = dscr + tid * GS;
    RowOut r; r.active = false; r.rhs = 0; r.aq = 0;
    int i = 0;
    if (c0 + tid < ncand) { const int pcode = cand[c0 + tid]; i = pcode / NSLOT; r = decode_row<C, true>(Y, D, T, fix, i, pcode - i * NSLOT, g); }
    unsigned long long cols = 0ull; int nn = 0;
    double c = r.rhs;
    if (r.active) {
      for (int q = 0; q < NZ; ++q) {
        double v = g[q];
        if (v != 0.0 && nn < 6) { cols |= (unsigned long long)q << (8 * nn); nn++; c -= v * Z[i * NZ + q]; }
      }
    }
    const bool sgl = r.active && nn == 1 && r.aq == 0.0, mul = r.active && !sgl;
    const unsigned long long maskM = __ballot(mul), maskS = __ballot(sgl);
    if (r.active) {
      const int idx = mul ? base + __popcll(maskM & lt) : Y.ROWCAP - 1 - (sbase + __popcll(maskS & lt));
      for (int k = 0; k < nn; ++k) rc_v[(size_t)k * Y.ROWCAP + idx] = g[(cols >> (8 * k)) & 255];
      cols |= ((unsigned long long)i << 48) | ((unsigned long long)nn << 56);
      double s, lam = MIQP_LAM0, t;
      if (r.aq == 0.0) {
        init_elastic(c, warm, B.ws_mu, B.ws_delta, s, lam, t);
        csum += s * lam + t * (RHO_EL - lam); cnt += 2; tsum += t;
      } else { lam = fmax(1.0, -2.0 * c * r.aq + 1.0); s = c + lam / r.aq; t = 0.0; csum += s * lam; cnt += 1; }
      rc_rhs[idx] = r.rhs; rc_col[idx] = __longlong_as_double((long long)cols);
      if (mul) rc_aq[idx] = r.aq;
      rs_s[idx] = s; rs_l[idx] = lam; rs_t[idx] = t;
      atomicAdd(mul ? &sstart[i + 1] : &sst[i + 1], 1);   // rows per stage, shifted by one for the prefix sums below
    }
    base += __popcll(maskM); sbase += __popcll(maskS);
  }
  if (tid == 0) { sh_nrows[0] = base; sh_nrows[1] = sbase; }
  __syncthreads();
  if (tid == 0) for (int i = 1; i <= N; ++i) { sstart[i] += sstart[i - 1]; sst[i] += sst[i - 1]; }   // first row of every stage
  __syncthreads();
  const int NM = sh_nrows[0], NS = sh_nrows[1], NROWS = NM + NS;
  // row r of the node (general rows first): its index in the row arrays
  auto ridx = [&](int r) { return r < NM ? r : Y.ROWCAP - 1 - (r - NM); };
  PROF_T(tb1); PROF_ACC(0, tb0, tb1);
  double comp = block_sum<NT>(csum, red);
  tsum = block_sum<NT>(tsum, red);
  int ncomp = (int)block_sum<NT>((double)cnt, red);
  if (ncomp < 1) ncomp = 1;
  comp /= ncomp;

  int it = 0, ok = 0;
  double resid_fac = 1.0, R0 = 0.0, obj = 0.0;
  double sigma = QP_SIGMA;   // centering parameter of the next iteration
  unsigned long long rowiters = 0;
  for (it = 1; it <= QP_MAXIT; ++it) {
    {
      double o = 0.0;
      for (int k = tid; k < N * NZ; k += NT) { double d = Z[k] - Rf[k]; o += Wd[k % NZ] * d * d; }
      obj = block_sum<NT>(o, red);
    }
    if (MIQP_ABL) { if (it > (((MIQP_ABL) & 512) ? 0 : 20)) { ok = 1; break; } }
    else if (comp < B.qp_tol * fmax(1.0, fabs(obj)) && resid_fac * R0 < 1e-7) { ok = 1; break; }
    // A rounding probe is a heuristic (it lies inside the first child, the children stay exhaustive without it): one that has
    // not converged after probe_itcap iterations - nearly always an infeasible rounding, 27 iterations to prove - is abandoned.
    // The launch of this kernel lasts as long as its slowest node.
    const bool pump_probe = B.pump_inc && B.pump_max > 0 && is_probe_word(B.batch_depth[node]);   // (re-rounding with an incumbent, pump_inc: a probe is then only cut off by its objective, not by the penalty of its violated rows - an infeasible one converges to its least-violation point and is re-rounded by eval_kernel)
    { const int pcap = (cutoff < 1e299 && (!pump_probe || (B.pump_inc & 2))) ? B.probe_itcap : B.probe_itcap0;   // (without an incumbent the probes are given longer: the re-rounding needs the converged solution of an infeasible one; pump_inc bit 1: the cap of the phase with an incumbent stays)
      if (pcap > 0 && it > pcap && B.ws_on != 2 && is_probe_word(B.batch_depth[node])) { ok = 2; break; } }
    // A launch lasts as long as its slowest node, and a single solve of three or four cars waits for it with most of the device idle
    // (infeasible nodes take 36 iterations, the average 14): a node that is still running after defer_cap iterations is put back on its
    // list (batch_ok 5: eval_kernel re-queues it with the bound it was selected with) with its iterate as its warm start, and runs to the
    // end the next time it is selected.
    {
      if (B.defer_cap > 0 && it > B.defer_cap && B.ws_on == 1 && B.pool_Z && (B.batch_depth[node] >> 6) >= 1 && B.batch_node[node] < B.z_cap
          && !is_probe_word(B.batch_depth[node]) && !(B.pool_big[B.batch_node[node]] & 8)) { ok = 5; break; }
    }
    // dual bound of the penalised problem: primal value - total complementarity (valid once the iterate is dual feasible)
#ifdef MIQP_PROFILE
    if (it > 1 && first_proxy == 0 && obj + RHO_EL * tsum - (double)ncomp * comp > cutoff + 1e-9 * fabs(cutoff)) first_proxy = it;
#endif
    if (!(MIQP_ABL) && it > 1 && resid_fac * R0 < B.cut_gate * (1.0 + fabs(obj)) && obj + (pump_probe ? 0.0 : RHO_EL * tsum) - (double)ncomp * comp - resid_fac * R0 * zdiam > cutoff + 1e-9 * fabs(cutoff)) { ok = 2; break; }   // (dual value minus the allowance for the stationarity residual still left, see batch_bound)
    const double tau = sigma * comp;
    // ================= backward sweep: Riccati recursion, the whole stage algebra stays in the registers of the wave.
    // Matrices live in the D layout of v_mfma_f64_16x16x4_f64 (lane l: g = l>>4, c = l&15, register r <-> M[g+4r][c]).
    // For a symmetric M that register file is directly the A operand of M*X (A[i=c][k=4kb+g] = M[4kb+g][c]) and the
    // B operand of X*M, so  T = P [A B],  S = Phi + [A B]' T  and the rank-NU update  P = S - Sxu K  are MFMA chains
    // without any LDS traffic; the NU x NU block is factored redundantly by every lane from v_readlane broadcasts.
    double rmax = 0.0;
    if constexpr (!WIDE) {
    const int lg = tid >> 4, lc = tid & 15;
    d4_t Pd = {0.0, 0.0, 0.0, 0.0};
    double pcol = 0.0;  // p[c], replicated over the four lane groups
    double rfn = lc < NZ ? Rf[(N - 1) * NZ + lc] : 0.0;   // reference of the stage whose Phi is built next (prefetched)
    RowRegs pre; pre.aq = 0.0; pre.col = 0.0; pre.s = 1.0; pre.lam = 1.0; pre.t = 1.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) pre.v[k] = 0.0;
    { int rp = sstart[N - 1] + tid; if (!((MIQP_ABL) & 1024) && tid < GROWS && rp < sstart[N]) pre = load_row(rc_aq, rc_col, rc_v, rs_s, rs_l, rs_t, Y.ROWCAP, rp); }

    // single-entry rows of the stage whose Phi is built next: coefficient, column word, s, lambda, t
    double sv_ = 0.0, sc_ = 0.0, ss_ = 1.0, sl_ = 1.0, st_ = 1.0;
    auto load_single = [&](int q) { const int idx = Y.ROWCAP - 1 - q; sv_ = rc_v[idx]; sc_ = rc_col[idx]; ss_ = rs_s[idx]; sl_ = rs_l[idx]; st_ = rs_t[idx]; };
    { int q = sst[N - 1] + tid; if (!((MIQP_ABL) & 1024) && q < sst[N]) load_single(q); }

    // Phi_j = 2W + diag(single-entry rows) + Gh' Gh, rr_j = 2W(z - ref) + (single-entry rows) + Gh' fs.  Single-entry
    // rows add w v^2 / v (lambda + kappa) to one diagonal / gradient entry (LDS f64 atomics, one lane per row).
    // The scaled general rows of stage j pass through LDS in chunks of GROWS
    // rows (one lane per row; the first chunk comes from the prefetch registers, and the rows of stage j-1 are
    // requested as soon as they are free); 4 rows per MFMA with A operand = B operand.
    auto phi_chain = [&](int j, d4_t& acc, double& rrc) {
      const int rb = sstart[j], nrj = sstart[j + 1] - rb;
      acc = d4_t{0.0, 0.0, 0.0, 0.0};
      double racc = 0.0;
      if (tid < 32) dgq[tid] = 0.0;
      PROF_T(tq0);
      for (int c0 = 0; c0 == 0 || c0 < nrj; c0 += GROWS) {
        const int nr = nrj - c0 < GROWS ? nrj - c0 : GROWS;
        const int nsl4 = (nr + 3) & ~3;
        if (tid < (((MIQP_ABL) & 128) ? 0 : nsl4)) {
          double* g = Gh + tid * GS;
          double fsv = 0.0;
#pragma unroll
          for (int q = 0; q < 16; ++q) g[q] = 0.0;
          if (tid < nr) {
            RowRegs R = c0 == 0 ? pre : load_row(rc_aq, rc_col, rc_v, rs_s, rs_l, rs_t, Y.ROWCAP, rb + c0 + tid);
            unsigned long long cols = (unsigned long long)__double_as_longlong(R.col);
            int nn = (int)(cols >> 56);
            double sw;
            if ((MIQP_ABL) & 2) { sw = R.s + R.lam + R.t; fsv = sw; }
            else {
            const double s = R.s, lam = R.lam, il = frcp(lam);
            double zz, r2mu = 0.0;
            if (R.aq == 0.0) { const double t = R.t, mu = RHO_EL - lam, im = frcp(mu); zz = t * im; r2mu = (tau - t * mu) * im; }
            else zz = frcp(R.aq);
            const double w = frcp(s * il + zz);
            const double kap = ((tau - s * lam) * il - r2mu) * w;
            const double isw = frsq(w); sw = w * isw;   // sqrt(w) and 1/sqrt(w) from one reciprocal square root
            fsv = (lam + kap) * isw;
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) if (k < nn) g[(cols >> (8 * k)) & 255] = sw * R.v[k];
          }
          fs[tid] = fsv;
        }
        if (c0 == 0 && j > 0) { int rp = sstart[j - 1] + tid; if (!((MIQP_ABL) & 1024) && tid < GROWS && rp < rb) pre = load_row(rc_aq, rc_col, rc_v, rs_s, rs_l, rs_t, Y.ROWCAP, rp); }
        __syncthreads();
        PROF_T(tq1); PROF_ACC(12, tq0, tq1);
        if (c0 == 0) {
          const int sb = sst[j], nsj = sst[j + 1] - sb;
          for (int q0 = 0; q0 < nsj; q0 += NT) {
            if (q0 > 0 && q0 + tid < nsj) load_single(sb + q0 + tid);
            if (q0 + tid < nsj) {
              const double s = ss_, lam = sl_, t = st_, il = frcp(lam), mu = RHO_EL - lam, im = frcp(mu);
              const double w = frcp(s * il + t * im);
              const double lk = lam + ((tau - s * lam) * il - (tau - t * mu) * im) * w;
              const int col = (int)((unsigned long long)__double_as_longlong(sc_) & 255ull);
              atomicAdd(&dgq[col], w * sv_ * sv_); atomicAdd(&dgq[16 + col], sv_ * lk);
            }
          }
          if (j > 0) { int q = sst[j - 1] + tid; if (!((MIQP_ABL) & 1024) && q < sb) load_single(q); }
        }
        PROF_T(tq2); PROF_ACC(13, tq1, tq2);
        for (int kb = 0; kb < (((MIQP_ABL) & 4) ? 0 : nsl4); kb += 16) {  // operands of up to 4 MFMAs are fetched before the dependent chain
          double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, f0 = 0.0, f1 = 0.0, f2 = 0.0, f3 = 0.0;
          a0 = Gh[(kb + lg) * GS + lc]; f0 = fs[kb + lg];
          if (kb + 4 < nsl4) { a1 = Gh[(kb + 4 + lg) * GS + lc]; f1 = fs[kb + 4 + lg]; }
          if (kb + 8 < nsl4) { a2 = Gh[(kb + 8 + lg) * GS + lc]; f2 = fs[kb + 8 + lg]; }
          if (kb + 12 < nsl4) { a3 = Gh[(kb + 12 + lg) * GS + lc]; f3 = fs[kb + 12 + lg]; }
          racc += a0 * f0 + a1 * f1 + a2 * f2 + a3 * f3;
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a0, acc, 0, 0, 0);
          if (kb + 4 < nsl4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, a1, acc, 0, 0, 0);
          if (kb + 8 < nsl4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, a2, acc, 0, 0, 0);
          if (kb + 12 < nsl4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, a3, acc, 0, 0, 0);
        }
        if (c0 + GROWS < nrj) __syncthreads();   // the next chunk overwrites the staging rows
        PROF_T(tq3); PROF_ACC(14, tq2, tq3);
      }
      PROF_T(tq4);
      racc += __shfl_xor(racc, 16); racc += __shfl_xor(racc, 32);
      __syncthreads();   // the single-entry contributions are complete
      { const double dd = lc < NZ ? 2.0 * Wd[lc] + dgq[lc] : 0.0;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) if (lg + 4 * rg == lc) acc[rg] += dd; }
      rrc = lc < NZ ? racc + dgq[16 + lc] + 2.0 * Wd[lc] * (Z[j * NZ + lc] - rfn) : 0.0;  // rr[c], replicated over groups
      if (j > 0 && lc < NZ) rfn = Rf[(j - 1) * NZ + lc];
      if (it == 1) rmax = fmax(rmax, fabs(rrc));
      PROF_T(tq5); PROF_ACC(15, tq4, tq5);
    };

    // Software pipeline over the stages: while the VALU chain of the elimination of stage i runs, the matrix core works
    // on Phi_{i-1}; the row block of stage i-1 is built under the T/S chains of stage i.
    d4_t accA; double rrA;
    PROF_T(tp0);
    phi_chain(N - 1, accA, rrA);
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {  // u_{N-1} = 0 (initial_conditions.mod:25-26): P = Phi_xx (+ 2W), p = rr_x
      double v = accA[rg];
      Pd[rg] = (lg + 4 * rg < NX && lc < NX) ? v : 0.0;
    }
    pcol = lc < NX ? rrA : 0.0;
    if (N >= 2) phi_chain(N - 2, accA, rrA);
    PROF_T(tp1); PROF_ACC(1, tp0, tp1);
    for (int i = N - 2; i >= 0; --i) {
      PROF_T(ts2);
      d4_t acc = accA; const double rrc = rrA;
      // T = P [A B]
      d4_t accT = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kb = 0; kb < (((MIQP_ABL) & 64) ? 0 : KB); ++kb) accT = __builtin_amdgcn_mfma_f64_16x16x4f64(Pd[kb], abr[kb], accT, 0, 0, 0);
      // S = Phi + [A B]' T   (acc becomes S)
#pragma unroll
      for (int kb = 0; kb < (((MIQP_ABL) & 64) ? 0 : KB); ++kb) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(abr[kb], accT[kb], acc, 0, 0, 0);
      // sv = rr + [A B]' p
      double part = 0.0;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) part += abr[kb] * __shfl(pcol, 4 * kb + lg);
      part += __shfl_xor(part, 16); part += __shfl_xor(part, 32);
      const double svc = rrc + part;
      // next stage: row block under the T/S chains, then its Phi chain is queued behind them
      d4_t accB = {0.0, 0.0, 0.0, 0.0}; double rrB = 0.0;
      if (i > 0) phi_chain(i - 1, accB, rrB);
      PROF_T(ts3); PROF_ACC(3, ts2, ts3);
      if ((MIQP_ABL) & 2048) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) Pd[rg] = (lg + 4 * rg < NX && lc < NX) ? acc[rg] : 0.0;
        pcol = lc < NX ? svc : 0.0; accA = accB; rrA = rrB; continue;
      }
      // Suu and su as wave-uniform values; LDL' in registers (every lane, redundantly)
      const double own = acc[RU];  // row NX + (g - GU0) of S for the lane groups that hold the u rows
      double Lm[NU][NU], su[NU], dinv[NU], dvec[NU];
#pragma unroll
      for (int q = 0; q < NU; ++q) {
        su[q] = readlane_d(svc, NX + q);
#pragma unroll
        for (int q2 = 0; q2 <= q; ++q2) Lm[q][q2] = readlane_d(own, (GU0 + q) * 16 + NX + q2);
      }
      // Suu = L D L' (unit lower L, no square roots); dinv = 1 / D
      if ((MIQP_ABL) & 1) {
#pragma unroll
        for (int a = 0; a < NU; ++a) { dvec[a] = 1.0; dinv[a] = 1.0; }
      } else {
#pragma unroll
      for (int a = 0; a < NU; ++a) {
#pragma unroll
        for (int b = 0; b <= a; ++b) {
          double v = Lm[a][b];
#pragma unroll
          for (int q = 0; q < b; ++q) v -= Lm[a][q] * Lm[b][q] * dvec[q];
          if (a == b) { dvec[a] = fmax(v, 1e-300); dinv[a] = frcp(dvec[a]); } else Lm[a][b] = v * dinv[b];
        }
      }
      }
      // column c of Sux: S[NX+q][c] sits in lane (GU0+q, c); fetched from the three other lane groups
      const double o1 = __shfl_xor(own, 16), o2 = __shfl_xor(own, 32), o3 = __shfl_xor(own, 48);
      double col[NU], xk[NU], kk[NU];
#pragma unroll
      for (int q = 0; q < NU; ++q) { int m = lg ^ (GU0 + q); col[q] = m == 0 ? own : (m == 1 ? o1 : (m == 2 ? o2 : o3)); }
      // K[:, c] = Suu^-1 Sux[:, c],  k = Suu^-1 su   (forward with unit L, scale by 1/D, backward with unit L')
      if ((MIQP_ABL) & 1) {
#pragma unroll
        for (int a = 0; a < NU; ++a) { xk[a] = col[a] * 1e-3; kk[a] = su[a] * 1e-3; }
      } else {
#pragma unroll
      for (int a = 0; a < NU; ++a) {
        double v = col[a], w2 = su[a];
#pragma unroll
        for (int q = 0; q < a; ++q) { v -= Lm[a][q] * xk[q]; w2 -= Lm[a][q] * kk[q]; }
        xk[a] = v; kk[a] = w2;
      }
#pragma unroll
      for (int a = 0; a < NU; ++a) { xk[a] *= dinv[a]; kk[a] *= dinv[a]; }
#pragma unroll
      for (int a = NU - 1; a >= 0; --a) {
        double v = xk[a], w2 = kk[a];
#pragma unroll
        for (int q = a + 1; q < NU; ++q) { v -= Lm[q][a] * xk[q]; w2 -= Lm[q][a] * kk[q]; }
        xk[a] = v; kk[a] = w2;
      }
      }
      PROF_T(ts4); PROF_ACC(4, ts3, ts4);
      // P = S - Sxu K : one MFMA (k = lane group), A[i=c][k=g] = -S[NX+q][c], B[k=g][j=c] = K[q][c]
      {
        const bool urow = lg >= GU0 && lg < GU0 + NU;
        double bop = 0.0, kop = 0.0;
#pragma unroll
        for (int q = 0; q < NU; ++q) if (lg - GU0 == q) { bop = xk[q]; kop = kk[q]; }
        if (urow && lc <= NX) KG[(i * NU + (lg - GU0)) * KSTR + lc] = lc < NX ? bop : kop;   // gains of stage i (L2 resident)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(urow ? -own : 0.0, urow ? bop : 0.0, acc, 0, 0, 0);
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) Pd[rg] = (lg + 4 * rg < NX && lc < NX) ? acc[rg] : 0.0;
        double pn = svc;
#pragma unroll
        for (int q = 0; q < NU; ++q) pn -= col[q] * kk[q];
        pcol = lc < NX ? pn : 0.0;
      }
      accA = accB; rrA = rrB;
      PROF_T(ts5); PROF_ACC(5, ts4, ts5);
    }
    } else {
    // ================= backward sweep for more than two cars: dense stage algebra in LDS.  [A B] is block sparse (three
    // entries per chain column), so T = P [A B] and S = Phi + [A B]' T cost three terms per element; the NU x NU block is
    // factored redundantly by every lane, lane c solves column c of K, P = Sxx - Sxu' K.
    constexpr int SS = NZ + 1, PS = NX + 1;   // row strides of S / T and of P (odd: rows and columns both spread over the LDS banks)
    double* Sm = dZ + N * NZ;        // [NZ][SS]
    double* Pm = Sm + NZ * SS;       // [NX][PS]
    double* Tm = Pm + NX * PS;       // [NX][SS]
    double* svv = Tm + NX * SS;      // [NZ]
    double* pv = svv + NZ;           // [NX]
    double* Km = pv + NX;            // [NU][NX+1]
    double* spv = dgq;               // [NZ] [A B]' p of the stage (dgq is the two-car form's)
    const double h1 = ts, h2 = 0.5 * ts * ts, h3 = ts * ts * ts / 6.0;
    const int wg = (tid & 63) >> 4, wc = tid & 15, wv = tid >> 6;   // lane group / column inside an MFMA tile, wavefront of the workgroup
    // the first row of every lane for the stage assembled next is requested a stage ahead (the rows of a node stream from HBM: 13 planes of
    // ROWCAP doubles per resident node are far more than the L2 holds): its latency passes under the T / S / K / P phases of the stage before
    RowRegs pre;
    auto stage_row = [&](int j, int r0) { const int nmj = sstart[j + 1] - sstart[j]; return r0 >= nmj ? Y.ROWCAP - 1 - (sst[j] + r0 - nmj) : sstart[j] + r0; };
    auto stage_rows = [&](int j) { return (sstart[j + 1] - sstart[j]) + (sst[j + 1] - sst[j]); };
    if (tid < stage_rows(N - 1)) pre = load_row(rc_aq, rc_col, rc_v, rs_s, rs_l, rs_t, Y.ROWCAP, stage_row(N - 1, tid));
    double rfn = tid < NZ ? Rf[(N - 1) * NZ + tid] : 0.0;   // reference of the stage assembled next (requested a stage ahead like the rows)
    // The compiler's s_waitcnt placement follows registers, not paths: a global load whose result is consumed on some paths only (the unused
    // coefficients of a short row) stays "pending" in its model, and every later reuse of that register waits for ALL younger loads - the
    // prefetched rows - to land: the prefetch was waited for at the top of the next phase.  An explicit vmcnt(0) where the loads of a
    // phase have certainly landed (a real s_waitcnt the pass accounts for) keeps its model clean.
    constexpr int VMCNT0 = 0x0F70;   // s_waitcnt vmcnt(0) (expcnt 7, lgkmcnt 15: not waited for)
    for (int j = N - 1; j >= 0; --j) {
      // Order of a stage: S = 2W + [A B]' P [A B] is STORED first (no zeroing pass, no read-modify-write), the rows of the stage are then added
      // on top with LDS atomics: five workgroup barriers per stage (eight when the rows came first).
      PROF_T(tw0);
      if (j == N - 1) {
        for (int k = tid; k < NZ * NZ; k += NT) { const int a = k / NZ, c = k - a * NZ; Sm[a * SS + c] = c == a ? 2.0 * Wd[a] : 0.0; }
        if (tid < NZ) spv[tid] = 0.0;
      } else {
        // T = P [A B] and S = 2W + [A B]' T on the VALU: [A B] has 1 / 2 / 3 / 3 entries in the columns (position, velocity, acceleration, input)
        // of a chain, so a (row, chain) pair of T is 6 fma on three entries of P and a (chain, column) pair of S the same on three entries of
        // T - 13.5 + 18 fma per lane and stage at four cars.  (Rounds 3-5 ran these as 48 dense v_mfma_f64_16x16x4_f64 per stage: 64 cycles of
        // the matrix pipe each on gfx950, 8 x the flops the sparsity needs.)
        for (int pr = tid; pr < NX * NU; pr += NT) {
          const int i = pr / NU, ch = pr - i * NU, q0 = 3 * ch;
          const double p0 = Pm[i * PS + q0], p1 = Pm[i * PS + q0 + 1], p2 = Pm[i * PS + q0 + 2];
          double* tr = Tm + i * SS;
          tr[q0] = p0; tr[q0 + 1] = fma(h1, p0, p1); tr[q0 + 2] = fma(h2, p0, fma(h1, p1, p2)); tr[NX + ch] = fma(h3, p0, fma(h2, p1, h1 * p2));
        }
        lds_barrier();
        for (int pr = tid; pr < NU * NZ; pr += NT) {
          const int ch = pr / NZ, c = pr - ch * NZ, q0 = 3 * ch;
          const double t0 = Tm[q0 * SS + c], t1 = Tm[(q0 + 1) * SS + c], t2 = Tm[(q0 + 2) * SS + c];
          const double dg = 2.0 * Wd[c];   // (the diagonal entry of the four rows of this pair, where the column is theirs)
          Sm[q0 * SS + c] = t0 + (c == q0 ? dg : 0.0); Sm[(q0 + 1) * SS + c] = fma(h1, t0, t1) + (c == q0 + 1 ? dg : 0.0);
          Sm[(q0 + 2) * SS + c] = fma(h2, t0, fma(h1, t1, t2)) + (c == q0 + 2 ? dg : 0.0);
          Sm[(NX + ch) * SS + c] = fma(h3, t0, fma(h2, t1, h1 * t2)) + (c == NX + ch ? dg : 0.0);
        }
        if (tid < NZ) {   // [A B]' p, kept apart from the rows' part of the gradient (the first iteration measures the latter: R0)
          const int a = tid; double v;
          if (a < NX) { const int ch = a / 3, ka = a - 3 * ch; v = pv[a]; if (ka >= 1) v += h1 * pv[a - 1]; if (ka >= 2) v += h2 * pv[a - 2]; }
          else { const int ch = a - NX; v = h3 * pv[3 * ch] + h2 * pv[3 * ch + 1] + h1 * pv[3 * ch + 2]; }
          spv[a] = v;
        }
      }
      __builtin_amdgcn_s_waitcnt(VMCNT0);   // (the rows and the reference requested a stage ago)
      if (tid < NZ) svv[tid] = 2.0 * Wd[tid] * (Z[j * NZ + tid] - rfn);
      lds_barrier();
      PROF_T(tw1); PROF_ACC(3, tw0, tw1);
      const int nmj = sstart[j + 1] - sstart[j], nsj = sst[j + 1] - sst[j];
      for (int r0 = tid; r0 < nmj + nsj; r0 += NT) {
        const bool sgl = r0 >= nmj;
        const int r = sgl ? Y.ROWCAP - 1 - (sst[j] + r0 - nmj) : sstart[j] + r0;
        RowRegs R = pre;
        if (r0 != tid) { R = load_row(rc_aq, rc_col, rc_v, rs_s, rs_l, rs_t, Y.ROWCAP, r); __builtin_amdgcn_s_waitcnt(VMCNT0); }   // (a stage of more than NT rows)
        if (sgl) R.aq = 0.0;
        unsigned long long cols = (unsigned long long)__double_as_longlong(R.col);
        const int nn = (int)(cols >> 56);
        const double s = R.s, lam = R.lam, il = frcp(lam);
        double zz, r2mu = 0.0;
        if (R.aq == 0.0) { const double t = R.t, mu = RHO_EL - lam, im = frcp(mu); zz = t * im; r2mu = (tau - t * mu) * im; }
        else zz = frcp(R.aq);
        const double w = frcp(s * il + zz);
        const double lk = lam + ((tau - s * lam) * il - r2mu) * w;
        // (only the lower triangle of S is kept: the columns of a row are stored in ascending order, and every reader of S below takes
        // S[max(i, j)][min(i, j)])
        if (sgl) {
          const int c = (int)(cols & 255);
          atomicAdd(&svv[c], R.v[0] * lk); atomicAdd(&Sm[c * SS + c], w * R.v[0] * R.v[0]);
        } else {
#pragma unroll
          for (int a = 0; a < 6; ++a) {
            if (a < nn) {
              const int ca_ = (int)((cols >> (8 * a)) & 255);
              const double wa = w * R.v[a];
              atomicAdd(&svv[ca_], R.v[a] * lk);
#pragma unroll
              for (int b = 0; b <= a; ++b) atomicAdd(&Sm[ca_ * SS + (int)((cols >> (8 * b)) & 255)], wa * R.v[b]);
            }
          }
        }
      }
      if (j > 0 && tid < stage_rows(j - 1)) pre = load_row(rc_aq, rc_col, rc_v, rs_s, rs_l, rs_t, Y.ROWCAP, stage_row(j - 1, tid));
      if (j > 0 && tid < NZ) rfn = Rf[(j - 1) * NZ + tid];
      lds_barrier();
      PROF_T(tw2); PROF_ACC(1, tw1, tw2);
      if (it == 1 && tid < NZ) rmax = fmax(rmax, fabs(svv[tid]));
      if (j == N - 1) {   // u_{N-1} = 0 (initial_conditions.mod:25-26): P = Phi_xx, p = rr_x
        for (int k = tid; k < NX * NX; k += NT) { const int i = k / NX, c = k - i * NX; Pm[i * PS + c] = Sm[(i > c ? i : c) * SS + (i > c ? c : i)]; }
        if (tid < NX) pv[tid] = svv[tid];
        lds_barrier();
        continue;
      }
      // Suu = L D L' in the registers of every lane (Lw[a][q] = L[a][q] d[q] is kept beside L: one fma per term of the elimination)
      double Lm[NU][NU], Lw[NU][NU], dinv[NU];
#pragma unroll
      for (int a = 0; a < NU; ++a) {
#pragma unroll
        for (int b = 0; b <= a; ++b) {
          double v = Sm[(NX + a) * SS + NX + b];
#pragma unroll
          for (int q = 0; q < b; ++q) v -= Lw[a][q] * Lm[b][q];
          if (a == b) dinv[a] = frcp(fmax(v, 1e-300)); else { Lw[a][b] = v; Lm[a][b] = v * dinv[b]; }
        }
      }
      // lane c < NX: K[:, c] = Suu^-1 Sux[:, c]; lane NX: k = Suu^-1 su
      if (tid <= NX) {
        double xk[NU];
#pragma unroll
        for (int a = 0; a < NU; ++a) {
          double v = tid < NX ? Sm[(NX + a) * SS + tid] : svv[NX + a] + spv[NX + a];
#pragma unroll
          for (int q = 0; q < a; ++q) v -= Lm[a][q] * xk[q];
          xk[a] = v;
        }
#pragma unroll
        for (int a = 0; a < NU; ++a) xk[a] *= dinv[a];
#pragma unroll
        for (int a = NU - 1; a >= 0; --a) {
          double v = xk[a];
#pragma unroll
          for (int q = a + 1; q < NU; ++q) v -= Lm[q][a] * xk[q];
          xk[a] = v;
        }
#pragma unroll
        for (int a = 0; a < NU; ++a) { Km[a * (NX + 1) + tid] = xk[a]; KG[(j * NU + a) * KSTR + tid] = xk[a]; }
      }
      lds_barrier();
      PROF_T(tw3); PROF_ACC(4, tw2, tw3);
      // P = Sxx - Sxu' K as an MFMA rank-NU update of the tiles of S:  A[i][k = q] = -S[NX + q][i],  B[k = q][j] = K[q][j]
      // (a 2 x 2 tiling of 16 x 16 tiles: tile t = 2 ti + tj = rows 16 ti .., columns 16 tj .. belongs to wavefront t % NW; lane (g, c) feeds
      // A[i = c][k = g] / B[k = g][j = c] and owns D[g + 4 r][c])
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if ((t % NW) != wv) continue;   // (wavefront-uniform)
        const int ti = t >> 1, tj = t & 1;
        const bool cok = 16 * tj + wc < NX;
        d4_t pp;
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int i = 16 * ti + wg + 4 * r, jc = 16 * tj + wc; pp[r] = (i < NX && cok) ? Sm[(i > jc ? i : jc) * SS + (i > jc ? jc : i)] : 0.0; }
#pragma unroll
        for (int qb = 0; qb < (NU + 3) / 4; ++qb) {
          const int q = 4 * qb + wg;
          const bool qv = q < NU;
          const double a = (qv && 16 * ti + wc < NZ) ? -Sm[(NX + q) * SS + 16 * ti + wc] : 0.0;
          const double b = (qv && cok) ? Km[q * (NX + 1) + 16 * tj + wc] : 0.0;
          pp = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, pp, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int i = 16 * ti + wg + 4 * r; if (i < NX && cok) Pm[i * PS + 16 * tj + wc] = pp[r]; }
      }
      if (tid < NX) {
        double v = svv[tid] + spv[tid];
#pragma unroll
        for (int q = 0; q < NU; ++q) v -= Sm[(NX + q) * SS + tid] * Km[q * (NX + 1) + NX];
        pv[tid] = v;
      }
      lds_barrier();
      PROF_T(tw4); PROF_ACC(5, tw3, tw4);
    }
    }
    if (it == 1) R0 = block_max<NT>(rmax, red);
    __syncthreads();
    PROF_T(tf0);
    // ================= forward sweep (one wavefront: LDS traffic in program order, no workgroup barrier needed)
    if (tid < NZ) dZ[tid] = 0.0;
    // [A B] row of lane tid < NX: x+[tid] = sum_m ca[m] x[q0 + m] + cb u[tid / 3]
    double ca[3], cb;
    { const int k3 = tid % 3;
#pragma unroll
      for (int m = 0; m < 3; ++m) { int d = m - k3; ca[m] = d < 0 ? 0.0 : (d == 0 ? 1.0 : (d == 1 ? ts : 0.5 * ts * ts)); }
      cb = k3 == 0 ? ts * ts * ts / 6.0 : (k3 == 1 ? 0.5 * ts * ts : ts); }
    // (three and four cars: lane 4 ch + m of the first wavefront holds the coefficients of entry m < 3 of chain ch, and the chain's state)
    double fa[3] = {0.0, 0.0, 0.0}, fb = 0.0, fx0 = 0.0, fx1 = 0.0, fx2 = 0.0;
    if constexpr (WIDE) { const int k3 = tid & 3;
#pragma unroll
      for (int m = 0; m < 3; ++m) { int d = m - k3; fa[m] = (k3 == 3 || d < 0) ? 0.0 : (d == 0 ? 1.0 : (d == 1 ? ts : 0.5 * ts * ts)); }
      fb = k3 == 0 ? ts * ts * ts / 6.0 : (k3 == 1 ? 0.5 * ts * ts : (k3 == 2 ? ts : 0.0)); }
    // the gains come back from L2 in bulk (the row staging area is free now): SPL stages per load, one wait each
    double* KL = dZ + N * NZ;   // the stage area (row staging / dense stage matrices) is free during the forward sweep
    constexpr int KSZ = NU * KSTR;
    constexpr int SPL = (WIDE ? NZ * NZ + NX * NX + NX * NZ : GROWS * (GSTR + 1)) / KSZ;
    for (int s0 = 0; s0 + 1 < (((MIQP_ABL) & 8) ? 0 : N); s0 += SPL) {
      __syncthreads();
      { const int n = ((N - 1 - s0) < SPL ? (N - 1 - s0) : SPL) * KSZ;
        for (int k = tid; k < n; k += NT) KL[k] = KG[s0 * KSZ + k]; }
      __syncthreads();
      const int s1 = s0 + SPL < N - 1 ? s0 + SPL : N - 1;
      for (int i = s0; i < s1; ++i) {
        if constexpr (WIDE) {
          // four lanes per chain: each sums a quarter of u = -k - K x, two quad_perm butterflies give all four the input, lanes 0-2 advance
          // one entry of the chain's state each (the state of the own chain travels in registers through quad_perm broadcasts; the other
          // chains' entries come back from LDS): a dependent chain of NX / 4 + 2 + 4 operations per stage instead of NX + 4
          if (tid < 4 * NU) {
            const double* kv = KL + (i - s0) * KSZ + (tid >> 2) * KSTR;
            const double* xi = dZ + i * NZ;
            const int fpt = tid & 3;
            double part = fpt == 0 ? -kv[NX] : 0.0;
#pragma unroll
            for (int q = 0; q < NX; q += 4) if (q + fpt < NX) part -= kv[q + fpt] * xi[q + fpt];
            part += dpp_mov<0xB1>(part); part += dpp_mov<0x4E>(part);   // quad_perm [1,0,3,2], [2,3,0,1]
            const double xn = fa[0] * fx0 + fa[1] * fx1 + fa[2] * fx2 + fb * part;
            if (fpt == 3) dZ[i * NZ + NX + (tid >> 2)] = part; else dZ[(i + 1) * NZ + 3 * (tid >> 2) + fpt] = xn;
            fx0 = dpp_mov<0x00>(xn); fx1 = dpp_mov<0x55>(xn); fx2 = dpp_mov<0xAA>(xn);   // quad_perm broadcasts of lanes 0, 1, 2
          }
        } else
        // one phase per stage: the three lanes of a chain compute their input redundantly (u = -k - K x) and advance
        // their own state with it, so the stage needs a single LDS round trip and wave barrier
        if (tid < NX) {
          const int ch = tid / 3, q0 = 3 * ch;
          const double* kv = KL + (i - s0) * KSZ + ch * KSTR;
          double u = -kv[NX];
#pragma unroll
          for (int q = 0; q < NX; ++q) u -= kv[q] * dZ[i * NZ + q];
          if (tid == q0) dZ[i * NZ + NX + ch] = u;
          dZ[(i + 1) * NZ + tid] = ca[0] * dZ[i * NZ + q0] + ca[1] * dZ[i * NZ + q0 + 1] + ca[2] * dZ[i * NZ + q0 + 2] + cb * u;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
    if (tid < NU) dZ[(N - 1) * NZ + NX + tid] = 0.0;
    __syncthreads();
    PROF_T(tf1); PROF_ACC(6, tf0, tf1);
    // ================= step length: ratio test over all rows; only g.dz is stored per row
    double rinv = 0.0, a0 = 0.0, a1 = 0.0, a2 = 0.0;
#pragma unroll 4
    for (int r = tid; r < (((MIQP_ABL) & 16) ? 0 : NROWS); r += NT) {
      const int idx = ridx(r);
      double aq = r < NM ? rc_aq[idx] : 0.0;
      unsigned long long cols = (unsigned long long)__double_as_longlong(rc_col[idx]);
      int nn = (int)(cols >> 56), i = (int)((cols >> 48) & 255);
      double gd = 0.0;
      for (int k = 0; k < nn; ++k) gd += rc_v[(size_t)k * Y.ROWCAP + idx] * dZ[i * NZ + ((cols >> (8 * k)) & 255)];
      double s = rs_s[idx], lam = rs_l[idx], t = rs_t[idx], ds, dl, dt;
      row_step(s, lam, t, aq, gd, tau, ds, dl, dt);
      // ratio test in reciprocal form: alpha_max = 1 / max_k(-dv_k / v_k)
      rinv = fmax(rinv, fmax(-ds * __builtin_amdgcn_rcp(s), -dl * __builtin_amdgcn_rcp(lam)));
      a0 += s * lam; a1 += s * dl + lam * ds; a2 += ds * dl;
      if (aq == 0.0) {
        double mu = RHO_EL - lam, dmu = -dl;
        rinv = fmax(rinv, fmax(-dt * __builtin_amdgcn_rcp(t), -dmu * __builtin_amdgcn_rcp(mu)));
        a0 += t * mu; a1 += t * dmu + mu * dt; a2 += dt * dmu;
      }
      rs_g[idx] = gd;
    }
    rowiters += (unsigned long long)NROWS;
    rinv = block_max<NT>(rinv, red);
    const double amax = rinv > 1e-300 ? 1.0 / rinv : 1e300;   // raw v_rcp_f64 in the ratios: far inside the 0.5 % margin below
    a0 = block_sum<NT>(a0, red); a1 = block_sum<NT>(a1, red); a2 = block_sum<NT>(a2, red);
    double alpha = fmin(1.0, MIQP_STEPFRAC * amax);
    comp = (a0 + alpha * a1 + alpha * alpha * a2) / ncomp;
    PROF_T(tf2); PROF_ACC(7, tf1, tf2);
    // ================= update (the step of every row is recomputed from its stored g.dz)
    double tnew = 0.0;
    for (int k = tid; k < N * NZ; k += NT) Z[k] += alpha * dZ[k];
#pragma unroll 4
    for (int r = tid; r < (((MIQP_ABL) & 32) ? 0 : NROWS); r += NT) {
      const int idx = ridx(r);
      double s = rs_s[idx], lam = rs_l[idx], t = rs_t[idx], ds, dl, dt;
      row_step(s, lam, t, r < NM ? rc_aq[idx] : 0.0, rs_g[idx], tau, ds, dl, dt);
      rs_s[idx] = s + alpha * ds; rs_l[idx] = lam + alpha * dl; rs_t[idx] = t + alpha * dt;
      tnew += t + alpha * dt;   // quadratic-soft rows carry t = 0
    }
    tsum = block_sum<NT>(tnew, red);
    resid_fac *= (1.0 - alpha);
    // centering of the next iteration follows the step just taken: a long step asks for little centering, a short one
    // for more (sigma = 1 - alpha within [0.02, 0.5]; measured -12 % iterations against the fixed sigma = 0.1)
    sigma = fmin(QP_SIGMA_HI, fmax(QP_SIGMA_LO, 1.0 - alpha));
    __syncthreads();
    PROF_T(tf3); PROF_ACC(8, tf2, tf3);
    if (!(MIQP_ABL) && alpha < 1e-12) break;
  }
  if (ok == 5) {   // deferred: the iterate becomes the record's warm start
    double* zp = B.pool_Z + (size_t)B.batch_node[node] * N * NZ;
    for (int k = tid; k < N * NZ; k += NT) zp[k] = Z[k];
    if (tid == 0) {
      B.batch_ok[node] = 5; B.pool_big[B.batch_node[node]] |= 8; B.batch_it[node] = it - 1;
      atomicAdd((unsigned long long*)&B.inst_iters[inst], (unsigned long long)(it - 1));
      atomicAdd(B.stat_rowiters, rowiters);
    }
    continue;
  }
  // ---- final measures: worst elastic violation, slack cost
  double viol = 0.0, scost = 0.0;
  for (int r = tid; r < NROWS; r += NT) {
    const int idx = ridx(r);
    double aq = r < NM ? rc_aq[idx] : 0.0;
    unsigned long long cols = (unsigned long long)__double_as_longlong(rc_col[idx]);
    int nn = (int)(cols >> 56), i = (int)((cols >> 48) & 255);
    double c = rc_rhs[idx];
    for (int k = 0; k < nn; ++k) c -= rc_v[(size_t)k * Y.ROWCAP + idx] * Z[i * NZ + ((cols >> (8 * k)) & 255)];
    if (aq == 0.0) viol = fmax(viol, -c);
    else { double t = rs_l[idx] / aq; scost += 0.5 * aq * t * t; }
  }
  viol = block_max<NT>(viol, red); scost = block_sum<NT>(scost, red);
  {
    double o = 0.0;
    for (int k = tid; k < N * NZ; k += NT) { double d = Z[k] - Rf[k]; o += Wd[k % NZ] * d * d; }
    obj = block_sum<NT>(o, red) + scost;
  }
  double* Zo = B.batch_Z + (size_t)node * N * NZ;
  for (int k = tid; k < N * NZ; k += NT) Zo[k] = Z[k];
  if (tid == 0) {
    B.batch_obj[node] = obj; B.batch_viol[node] = viol; B.batch_ok[node] = ok;
    // primal - dual value of the final iterate: total complementarity, plus what a stationarity residual r of the iterate can
    // add to the Lagrangian over the trajectories a child may reach (|r|_inf x |z' - z|_1, the latter bounded by the L1 diameter of the
    // reachable set of the instance, host_inst.hpp: d_misc + 2) - the dual value and the bound lifting built on it stay rigorous at loose tolerances
    B.batch_bound[node] = (double)ncomp * comp + resid_fac * R0 * zdiam;
    B.batch_it[node] = it > QP_MAXIT ? QP_MAXIT : it;
    atomicAdd((unsigned long long*)&B.inst_iters[inst], (unsigned long long)(it > QP_MAXIT ? QP_MAXIT : it));
    atomicAdd((unsigned long long*)&B.inst_nodes[inst], 1ull);
    atomicAdd(B.stat_rowiters, rowiters);
#ifdef MIQP_PROFILE
    for (int q = 0; q < 9; ++q) atomicAdd(&B.prof[q], pr_[q]);
    for (int q = 12; q < 16; ++q) atomicAdd(&B.prof[q], pr_[q]);   // (parts of the stage-Hessian chain, inside phase 3)
    atomicAdd(&B.prof[9], (unsigned long long)(it > QP_MAXIT ? QP_MAXIT : it)); atomicAdd(&B.prof[10], 1ull); atomicAdd(&B.prof[11], (unsigned long long)NROWS);
    { int hb = (it > QP_MAXIT ? QP_MAXIT : it) / 10; if (hb > 8) hb = 8; atomicAdd(&B.prof[16 + hb], 1ull); if (viol > FEAS_TOL) atomicAdd(&B.prof[25 + hb], 1ull); }
    if (ok == 2) { atomicAdd(&B.prof[36], (unsigned long long)first_proxy); atomicAdd(&B.prof[37], (unsigned long long)it); atomicAdd(&B.prof[38], 1ull); }
    { int og = (int)B.pool_origin[B.batch_node[node]] & 7; atomicAdd(&B.prof[40 + og], 1ull); if (viol > FEAS_TOL) atomicAdd(&B.prof[48 + og], 1ull); if (ok == 2) atomicAdd(&B.prof[56 + og], 1ull); }
#endif
  }
  }  // node loop
}

}  // namespace miqp
#include "ipm_onchip.hip"
#include "as_onchip.hip"
namespace miqp {

// ------------------------------------------------------------------------------------------------
//  eval kernel helpers: violation of alternatives evaluated directly from the stage state
struct CarState { double px, vx, ax, py, vy, ay, ux, uy; };

// rt == nullptr: the bound of the point over the car's region set (K = d_fpk entry; lower: the smallest coordinate, else the largest)
__device__ inline void point_xy(const CarState& s, const double* rt, int tx, int ty, double& X, double& Y, const double* K = nullptr, bool lower = true) {
  X = s.px; Y = s.py;
  if (tx != PT_R) { if (rt) { const double* p = rt + 19 + (tx == PT_U ? 0 : 3); X += p[0] + p[1] * s.vx + p[2] * s.vy; } else X += K[(tx == PT_U ? 0 : 2) + (lower ? 0 : 1)]; }
  if (ty != PT_R) { if (rt) { const double* p = rt + 25 + (ty == PT_U ? 0 : 3); Y += p[0] + p[1] * s.vx + p[2] * s.vy; } else Y += K[(ty == PT_U ? 4 : 6) + (lower ? 0 : 1)]; }
}

__device__ inline double box_viol(const CarState& s, const double* rt, bool with_jerk) {
  double v = fmax(fmax(s.ax - rt[12], rt[11] - s.ax), fmax(s.ay - rt[14], rt[13] - s.ay));
  if (with_jerk) v = fmax(v, fmax(fmax(s.ux - rt[16], rt[15] - s.ux), fmax(s.uy - rt[18], rt[17] - s.uy)));
  return v;
}

__device__ inline double region_alt_viol(const Layout& Y, const double* D, const int* T, int c, int q, int h, const CarState& s,
                                         bool with_jerk) {
  const double* rt = D + Y.d_reg + (c * Y.P + q) * REGSZ;
  double vm = D[Y.d_glob + 6];
  double v = box_viol(s, rt, with_jerk);
  if (h == 3) return fmax(v, fmax(fabs(s.vx) - vm, fabs(s.vy) - vm));
  v = fmax(v, rt[0] * s.vx + rt[1] * s.vy);
  v = fmax(v, rt[2] * s.vx + rt[3] * s.vy);
  const int* hs = T + Y.i_hs + ((c * Y.P + q) * 2 + h) * 2;
  v = fmax(v, vm - hs[1] * (hs[0] == 0 ? s.vx : s.vy));
  v = fmax(v, s.ay - rt[4] * s.ax - rt[6] * s.vx - rt[7] * s.vy - rt[5]);
  v = fmax(v, -s.ay + rt[4] * s.ax + rt[9] * s.vx + rt[10] * s.vy + rt[8]);
  return v;
}

// region_alt_viol in two parts: what every half-plane alternative h of region q shares (the boxes; the sector and curvature rows), read from the
// region's table once, and what depends on h.  max is exact: region_viol_from(.., h, ..) == region_alt_viol(.., q, h, ..) bit for bit.
__device__ inline void region_viol_parts(const Layout& Y, const double* D, int c, int q, const CarState& s, bool with_jerk, double& bx, double& fast) {
  const double* rt = D + Y.d_reg + (c * Y.P + q) * REGSZ;
  bx = box_viol(s, rt, with_jerk);
  fast = fmax(fmax(rt[0] * s.vx + rt[1] * s.vy, rt[2] * s.vx + rt[3] * s.vy),
              fmax(s.ay - rt[4] * s.ax - rt[6] * s.vx - rt[7] * s.vy - rt[5], -s.ay + rt[4] * s.ax + rt[9] * s.vx + rt[10] * s.vy + rt[8]));
}
__device__ inline double region_viol_from(const Layout& Y, const double* D, const int* T, int c, int q, int h, const CarState& s, double bx, double fast) {
  const double vm = D[Y.d_glob + 6];
  if (h == 3) return fmax(bx, fmax(fabs(s.vx) - vm, fabs(s.vy) - vm));
  const int* hs = T + Y.i_hs + ((c * Y.P + q) * 2 + h) * 2;
  return fmax(fmax(bx, fast), vm - hs[1] * (hs[0] == 0 ? s.vx : s.vy));
}

__device__ inline double env_alt_viol(const Layout& Y, const double* D, const int* T, int e, double X, double Yc) {
  double v = -1e300; int n = T[Y.i_envn + e];
  for (int k = 0; k < n; ++k) { const double* ed = D + Y.d_env + (e * Y.EL + k) * 3; v = fmax(v, ed[0] * X + ed[1] * Yc - ed[2]); }
  return v;
}

// zero-slack violation of c2c alternative; (s1, rt1) / (s2, rt2) are the two cars of pair p
// (rt1 / rt2 == nullptr with K1 / K2: the violation of the RELAXED row of a car whose region is undecided, see add_point)
__device__ inline double c2c_alt_viol(const Layout& Y, const double* D, int p, int i, int grp, int alt, const CarState& s1, const double* rt1,
                                      const CarState& s2, const double* rt2, const double* K1 = nullptr, const double* K2 = nullptr) {
  double Dsep = D[Y.d_dsep + p * Y.N + i], S = D[Y.d_ssl + i];
  bool isx = alt < 2, lo = (alt == 0 || alt == 2);
  bool soft = (grp == 0 || grp == 3);
  int a1, ta, tb;  // A is car a1 (1 or 2), B the other
  if (grp == 0) { ta = tb = PT_R; a1 = lo ? 1 : 2; }
  else if (grp == 1) { if (lo) { a1 = 1; ta = PT_R; tb = PT_L; } else { a1 = 2; ta = PT_U; tb = PT_R; } }
  else if (grp == 2) { if (lo) { a1 = 2; ta = PT_R; tb = PT_L; } else { a1 = 1; ta = PT_U; tb = PT_R; } }
  else { if (lo) { a1 = 2; ta = PT_U; tb = PT_L; } else { a1 = 1; ta = PT_U; tb = PT_L; } }
  double XA, YA, XB, YB;
  if (a1 == 1) { point_xy(s1, rt1, ta, ta, XA, YA, K1, true); point_xy(s2, rt2, tb, tb, XB, YB, K2, false); }
  else { point_xy(s2, rt2, ta, ta, XA, YA, K2, true); point_xy(s1, rt1, tb, tb, XB, YB, K1, false); }
  double lhs = isx ? XA - XB : YA - YB;
  return lhs + (soft ? Dsep + S : Dsep);
}

// --- scores for the choice of the branching disjunction: lower bound on what an alternative costs, from the diagonal of the
// response tables (the exact lift with the cross terms is computed for the children of the chosen disjunction only)
struct LiftDiag { double p, v, a, u; };
__device__ inline void lift_diag(const Layout& Y, const double* D, int c, int i, LiftDiag& X, LiftDiag& Yd) {
  const double* S = D + Y.d_lift + ((c * 2) * Y.N + i) * 16;
  X.p = S[0]; X.v = S[5]; X.a = S[10]; X.u = S[15];
  S += Y.N * 16;
  Yd.p = S[0]; Yd.v = S[5]; Yd.a = S[10]; Yd.u = S[15];
}
__device__ inline double lift1(double v, double gam) { return (v > 0.0 && gam > 1e-300 && gam < 1e200) ? 0.5 * v * v / gam : 0.0; }

__device__ inline double region_alt_lift(const Layout& Y, const double* D, const int* T, int c, int q, int h, const CarState& s, bool with_jerk,
                                         const LiftDiag& X, const LiftDiag& Yd) {
  const double* rt = D + Y.d_reg + (c * Y.P + q) * REGSZ;
  const double vm = D[Y.d_glob + 6];
  double l = fmax(lift1(fmax(s.ax - rt[12], rt[11] - s.ax), X.a), lift1(fmax(s.ay - rt[14], rt[13] - s.ay), Yd.a));
  if (with_jerk) l = fmax(l, fmax(lift1(fmax(s.ux - rt[16], rt[15] - s.ux), X.u), lift1(fmax(s.uy - rt[18], rt[17] - s.uy), Yd.u)));
  if (h == 3) return fmax(l, fmax(lift1(fabs(s.vx) - vm, X.v), lift1(fabs(s.vy) - vm, Yd.v)));
  l = fmax(l, lift1(rt[0] * s.vx + rt[1] * s.vy, rt[0] * rt[0] * X.v + rt[1] * rt[1] * Yd.v));
  l = fmax(l, lift1(rt[2] * s.vx + rt[3] * s.vy, rt[2] * rt[2] * X.v + rt[3] * rt[3] * Yd.v));
  const int* hs = T + Y.i_hs + ((c * Y.P + q) * 2 + h) * 2;
  l = fmax(l, lift1(vm - hs[1] * (hs[0] == 0 ? s.vx : s.vy), hs[0] == 0 ? X.v : Yd.v));
  l = fmax(l, lift1(s.ay - rt[4] * s.ax - rt[6] * s.vx - rt[7] * s.vy - rt[5], Yd.a + rt[4] * rt[4] * X.a + rt[6] * rt[6] * X.v + rt[7] * rt[7] * Yd.v));
  l = fmax(l, lift1(-s.ay + rt[4] * s.ax + rt[9] * s.vx + rt[10] * s.vy + rt[8], Yd.a + rt[4] * rt[4] * X.a + rt[9] * rt[9] * X.v + rt[10] * rt[10] * Yd.v));
  return l;
}

constexpr double LIFT_MARGIN = 0.5;   // a row of the child joins the multi-row lifting when it is violated or within this slack of active
constexpr int LIFT_ROWS = 24;     // rows per child that enter the multi-row bound lifting (eval_kernel)
constexpr int REPAIR_ROOT = 62;   // depth word of a MIP-start repair root (tree depth 0, sibling preference 62: no node of the tree carries it)
// Lower bound on what it costs a node to take region alternative (q, h) of (car, step): the largest single-row lift over the
// rows of the alternative, with the EXACT response g Sigma g' of every row (the chains of a car are independent, a row touches
// at most velocity and acceleration of a chain) - unlike region_alt_lift, whose diagonal form is only a branching score.
// node set of the alternatives of a car/car group, kept in the (negative) undecided value of its fix byte: -1 = all four,
// -(16 + set) = the alternatives of `set` (bit a) - eval_kernel removes alternatives that cannot pay off below a node
__device__ inline int c2c_set(int fx) { return fx >= 0 ? 0 : (fx == -1 ? 15 : ((-fx - 16) & 15)); }
__device__ inline signed char c2c_set_byte(int set) { return (signed char)(set == 15 ? -1 : -(16 + set)); }
// response g Sigma g' of the separation row of car/car alternative (group, alt) at a step: the row is coord(A) - coord(B) <= ...,
// a rear point contributes the position of its chain, a front point position + p1 vx + p2 vy (chains of one car are independent,
// the two cars are independent)
struct PosBlk { double pp, pv, vv; };
__device__ inline void pos_blocks(const Layout& Y, const double* D, int c, int i, PosBlk& X, PosBlk& Yd) {
  const double* S = D + Y.d_lift + ((c * 2) * Y.N + i) * 16;
  X.pp = S[0]; X.pv = S[1]; X.vv = S[5];
  S += Y.N * 16;
  Yd.pp = S[0]; Yd.pv = S[1]; Yd.vv = S[5];
}
__device__ inline double point_gamma(const PosBlk& X, const PosBlk& Yd, const double* rt, int t, bool isx) {
  if (t == PT_R || !rt) return isx ? X.pp : Yd.pp;   // (no region table: the relaxed row of an undecided region carries the position only)
  const double* p = rt + (isx ? 19 : 25) + (t == PT_U ? 0 : 3);   // coordinate + p[0] + p[1] vx + p[2] vy
  return isx ? X.pp + 2.0 * p[1] * X.pv + p[1] * p[1] * X.vv + p[2] * p[2] * Yd.vv
             : Yd.pp + 2.0 * p[2] * Yd.pv + p[2] * p[2] * Yd.vv + p[1] * p[1] * X.vv;
}
__device__ inline double c2c_alt_gamma(int grp, int alt, const PosBlk& X1, const PosBlk& Y1, const double* rt1, const PosBlk& X2, const PosBlk& Y2, const double* rt2) {
  const bool isx = alt < 2, lo = (alt == 0 || alt == 2);
  int a1, ta, tb;   // as in c2c_alt_viol: A is car a1 (1 or 2), B the other
  if (grp == 0) { ta = tb = PT_R; a1 = lo ? 1 : 2; }
  else if (grp == 1) { if (lo) { a1 = 1; ta = PT_R; tb = PT_L; } else { a1 = 2; ta = PT_U; tb = PT_R; } }
  else if (grp == 2) { if (lo) { a1 = 2; ta = PT_R; tb = PT_L; } else { a1 = 1; ta = PT_U; tb = PT_R; } }
  else { if (lo) { a1 = 2; ta = PT_U; tb = PT_L; } else { a1 = 1; ta = PT_U; tb = PT_L; } }
  return a1 == 1 ? point_gamma(X1, Y1, rt1, ta, isx) + point_gamma(X2, Y2, rt2, tb, isx)
                 : point_gamma(X2, Y2, rt2, ta, isx) + point_gamma(X1, Y1, rt1, tb, isx);
}

struct LiftBlk { double vv, va, aa, uu; };
__device__ inline void lift_blocks(const Layout& Y, const double* D, int c, int i, LiftBlk& X, LiftBlk& Yd) {
  const double* S = D + Y.d_lift + ((c * 2) * Y.N + i) * 16;
  X.vv = S[5]; X.va = S[6]; X.aa = S[10]; X.uu = S[15];
  S += Y.N * 16;
  Yd.vv = S[5]; Yd.va = S[6]; Yd.aa = S[10]; Yd.uu = S[15];
}
__device__ inline double region_alt_lift_exact(const Layout& Y, const double* D, const int* T, int c, int q, int h, const CarState& s, bool with_jerk,
                                               const LiftBlk& X, const LiftBlk& Yd) {
  const double* rt = D + Y.d_reg + (c * Y.P + q) * REGSZ;
  const double vm = D[Y.d_glob + 6];
  double l = fmax(lift1(fmax(s.ax - rt[12], rt[11] - s.ax), X.aa), lift1(fmax(s.ay - rt[14], rt[13] - s.ay), Yd.aa));
  if (with_jerk) l = fmax(l, fmax(lift1(fmax(s.ux - rt[16], rt[15] - s.ux), X.uu), lift1(fmax(s.uy - rt[18], rt[17] - s.uy), Yd.uu)));
  if (h == 3) return fmax(l, fmax(lift1(fabs(s.vx) - vm, X.vv), lift1(fabs(s.vy) - vm, Yd.vv)));
  l = fmax(l, lift1(rt[0] * s.vx + rt[1] * s.vy, rt[0] * rt[0] * X.vv + rt[1] * rt[1] * Yd.vv));
  l = fmax(l, lift1(rt[2] * s.vx + rt[3] * s.vy, rt[2] * rt[2] * X.vv + rt[3] * rt[3] * Yd.vv));
  const int* hs = T + Y.i_hs + ((c * Y.P + q) * 2 + h) * 2;
  l = fmax(l, lift1(vm - hs[1] * (hs[0] == 0 ? s.vx : s.vy), hs[0] == 0 ? X.vv : Yd.vv));
  // curvature rows: +-(ay - rho ax - k2 vx - k3 vy) <= ...: chain x carries (v: -+k2, a: -+rho), chain y (v: -+k3, a: +-1)
  { const double rho = rt[4], k2 = rt[6], k3 = rt[7];
    const double gam = k2 * k2 * X.vv + 2.0 * k2 * rho * X.va + rho * rho * X.aa + k3 * k3 * Yd.vv - 2.0 * k3 * Yd.va + Yd.aa;
    l = fmax(l, lift1(s.ay - rho * s.ax - k2 * s.vx - k3 * s.vy - rt[5], gam)); }
  { const double rho = rt[4], k2 = rt[9], k3 = rt[10];
    const double gam = k2 * k2 * X.vv + 2.0 * k2 * rho * X.va + rho * rho * X.aa + k3 * k3 * Yd.vv - 2.0 * k3 * Yd.va + Yd.aa;
    l = fmax(l, lift1(-s.ay + rho * s.ax + k2 * s.vx + k3 * s.vy + rt[8], gam)); }
  return l;
}

// region_alt_lift_exact in the same two parts (bit for bit the same value)
__device__ inline void region_lift_parts(const Layout& Y, const double* D, int c, int q, const CarState& s, bool with_jerk, const LiftBlk& X, const LiftBlk& Yd, double& l0, double& lfast) {
  const double* rt = D + Y.d_reg + (c * Y.P + q) * REGSZ;
  l0 = fmax(lift1(fmax(s.ax - rt[12], rt[11] - s.ax), X.aa), lift1(fmax(s.ay - rt[14], rt[13] - s.ay), Yd.aa));
  if (with_jerk) l0 = fmax(l0, fmax(lift1(fmax(s.ux - rt[16], rt[15] - s.ux), X.uu), lift1(fmax(s.uy - rt[18], rt[17] - s.uy), Yd.uu)));
  lfast = fmax(lift1(rt[0] * s.vx + rt[1] * s.vy, rt[0] * rt[0] * X.vv + rt[1] * rt[1] * Yd.vv), lift1(rt[2] * s.vx + rt[3] * s.vy, rt[2] * rt[2] * X.vv + rt[3] * rt[3] * Yd.vv));
  { const double rho = rt[4], k2 = rt[6], k3 = rt[7];
    const double gam = k2 * k2 * X.vv + 2.0 * k2 * rho * X.va + rho * rho * X.aa + k3 * k3 * Yd.vv - 2.0 * k3 * Yd.va + Yd.aa;
    lfast = fmax(lfast, lift1(s.ay - rho * s.ax - k2 * s.vx - k3 * s.vy - rt[5], gam)); }
  { const double rho = rt[4], k2 = rt[9], k3 = rt[10];
    const double gam = k2 * k2 * X.vv + 2.0 * k2 * rho * X.va + rho * rho * X.aa + k3 * k3 * Yd.vv - 2.0 * k3 * Yd.va + Yd.aa;
    lfast = fmax(lfast, lift1(-s.ay + rho * s.ax + k2 * s.vx + k3 * s.vy + rt[8], gam)); }
}
__device__ inline double region_lift_from(const Layout& Y, const double* D, const int* T, int c, int q, int h, const CarState& s, const LiftBlk& X, const LiftBlk& Yd, double l0, double lfast) {
  const double vm = D[Y.d_glob + 6];
  if (h == 3) return fmax(l0, fmax(lift1(fabs(s.vx) - vm, X.vv), lift1(fabs(s.vy) - vm, Yd.vv)));
  const int* hs = T + Y.i_hs + ((c * Y.P + q) * 2 + h) * 2;
  return fmax(fmax(l0, lfast), lift1(vm - hs[1] * (hs[0] == 0 ? s.vx : s.vy), hs[0] == 0 ? X.vv : Yd.vv));
}

struct BranchDesc { int prio; int kind; int c; int o; int i; int pt; int cause; };  // kind: 0 region 1 env 2 obs 3 c2c; cause (diagnostic): what flagged a region disjunction - 0 its own rows, 1 / 2 / 3 an environment / obstacle / car-car row on a front point of a car whose region is undecided

#ifndef MIQP_EVAL_WPE
#define MIQP_EVAL_WPE 0   // wavefronts per SIMD eval_kernel is register-allocated for (0: the compiler's choice - 141 VGPRs, 3 per SIMD; measured against 2 and 4, tools/eval_wpe.sh)
#endif
template <int C>
#if MIQP_EVAL_WPE > 0
__global__ void __launch_bounds__(64, MIQP_EVAL_WPE) eval_kernel(DevBuf B) {
#else
__global__ void __launch_bounds__(64) eval_kernel(DevBuf B) {
#endif
  constexpr int NZ = 8 * C;
  const Layout& Y = B.Y;
  const int node = blockIdx.x, lane = threadIdx.x;
  if (node >= *B.batch_count || node >= B.batch_cap) return;
  if (lane == 0) B.batch_candkey[node] = ~0ull;
#define FREE_NODE() do { if (lane == 0) { unsigned int q_ = atomicAdd(B.free_tail, 1u); B.free_q[q_ % (unsigned int)B.pool_cap] = B.batch_node[node]; } } while (0)
  const int inst = B.batch_inst[node];
  const int slot = B.inst_slot[inst];                // where the instance's open lists live
  const double* D = B.inst_d + (size_t)inst * Y.dstride;
  const int* T = B.inst_i + (size_t)inst * Y.istride;
  const int N = Y.N, P = Y.P;
  extern __shared__ double lds[];
  double* Z = lds;                                   // [N][NZ]
  // (one region for two tenants that are never alive together: the slow-alternative table of phase R - read for the last time by the freeze
  // resolution right behind it - and the dense rows of the lifting.  Apart they were 10 + 9 KB of a two-car workgroup's 28 KB (5 workgroups per
  // CU of a kernel that waits on memory for 60 % of its cycles; 8 now) and 61 + 17 KB at four cars x 64 regions (ONE workgroup per CU; 3 now).)
  float* slowv = (float*)(Z + N * NZ);               // [C*N][P] slow-alternative violation per possible region (float: compared with tolerances and with each other only - a violation within one float ulp of the tolerance may be labelled differently from the double path; the label is a canonical choice among alternatives that hold, never a feasibility verdict)
  double* gsc = Z + N * NZ;                          // [LIFT_LANES][NZ + 1] one dense row per lane (lifting)
  double* fastv = Z + N * NZ + eval_shared_doubles(C, N, P);   // [C*N] best fast alternative violation
  double* rlift = fastv + C * N;                     // [C*N] smallest lift over the region alternatives (branching score)
  int* fastc = (int*)(rlift + C * N);                // [C*N] its code
  int* vflag = fastc + C * N;                        // [C*N] region violated
  int* altbuf = vflag + C * N;                       // [64]
  double* altval = (double*)(altbuf + 64);           // [64]
  signed char* fix = (signed char*)(altval + 64);    // [fixlen]
  signed char* comp = fix + Y.fixlen;                // [fixlen]
  signed char* cfix = comp + Y.fixlen;               // [fixlen] fix record of the child whose rows are being lifted
  // multi-row lifting: up to LIFT_ROWS rows of a child at the branching stage - coefficients g, response y = Sigma g, violation
  // v at the node's solution, diagonal g Sigma g', multiplier; and the running vector w = sum_r lambda_r y_r
  double* mr_g = (double*)(cfix + ((Y.fixlen + 7) & ~7));   // [LIFT_ROWS][NZ]
  double* mr_y = mr_g + LIFT_ROWS * NZ;              // [LIFT_ROWS][NZ]
  double* mr_v = mr_y + LIFT_ROWS * NZ;              // [LIFT_ROWS]
  double* mr_d = mr_v + LIFT_ROWS;                   // [LIFT_ROWS]
  double* mr_l = mr_d + LIFT_ROWS;                   // [LIFT_ROWS]
  double* mr_w = mr_l + LIFT_ROWS;                   // [NZ]
  // (the multi-row buffers exist only when that experiment is switched on - 6.8 KB for two cars, the difference between 6 and 8 workgroups per CU of a kernel that waits on memory for 60 % of its cycles)
  signed char* ploose = (B.seq_kinds & 0x80000000u) ? (signed char*)(mr_w + NZ) : (signed char*)mr_g;   // [fixlen] 1: the completed alternative of this undecided front-point disjunction holds with room to spare (probe_margin)
  __shared__ BranchDesc chosen;
  __shared__ int sh_base[4];
  __shared__ int slots[64];
  __shared__ int ck[64], ca[64], fam[4];
  __shared__ int k_ck[64], k_ca[64], k_neg[64], k_ord[64], k_pos[64];
  __shared__ double clift[64], k_bnd[64];

  const double* Zi = B.batch_Z + (size_t)node * N * NZ;
  const signed char* src = B.pool_fix + (size_t)B.batch_node[node] * Y.fixlen;
  PROF_T(te0);
  for (int k = lane; k < N * NZ; k += 64) Z[k] = Zi[k];
  for (int k8 = lane; k8 < Y.fixlen / 8; k8 += 64) { const unsigned long long w8 = ((const unsigned long long*)src)[k8]; ((unsigned long long*)fix)[k8] = w8; ((unsigned long long*)comp)[k8] = w8; ((unsigned long long*)ploose)[k8] = 0ull; }   // (fixlen is a multiple of 16, the records and the LDS arrays 8-byte aligned)
  __syncthreads();
  PROF_T(te1);
  const double viol = B.batch_viol[node];
  const int okq = B.batch_ok[node];
  if (okq == 5) {   // returned unsolved by the standard on-chip kernel (too large for it, now marked): back on the list with the bound it came with
    if (lane == 0) {
      const int pos = atomicAdd(&B.open_count[inst], 1);
      if (pos < B.open_cap) { const size_t oi = ((size_t)B.open_sel * B.n_slots + slot) * B.open_cap + pos; B.open_bound[oi] = B.batch_bound[node]; B.open_node[oi] = B.batch_node[node]; B.open_depth[oi] = B.batch_depth[node]; }
      else { atomicOr(&B.inst_flags[inst], 1); unsigned int q_ = atomicAdd(B.free_tail, 1u); B.free_q[q_ % (unsigned int)B.pool_cap] = B.batch_node[node]; }
    }
    return;
  }
  const unsigned char big_parent = B.pool_big ? B.pool_big[B.batch_node[node]] : 0;
  if (B.stats && lane == 0) {   // diagnostic (MIQP_STATS): outcome of the node and the iterations it took
    const int oc_ = okq == 2 ? 1 : (viol > FEAS_TOL ? 0 : (okq != 1 ? 2 : 3));   // infeasible, cut off, not converged, solved
    atomicAdd(&B.stats[32 + oc_], 1ull); atomicAdd(&B.stats[36 + oc_], (unsigned long long)B.batch_it[node]);
    if (B.pool_origin) { const int og_ = (int)B.pool_origin[B.batch_node[node]] & 15; atomicAdd(&B.stats[80 + og_], 1ull); atomicAdd(&B.stats[96 + 16 * oc_ + og_], 1ull); if (og_ == 15) { const int ib_ = B.batch_it[node] / 3; atomicAdd(&B.stats[192 + 16 * oc_ + (ib_ > 15 ? 15 : ib_)], 1ull); } }   // the same by the branching that created the node
  }
  // Re-rounding (the step of a feasibility pump): a rounding probe whose relaxation is infeasible - the alternatives its parent's solution
  // completed to cannot hold together - has, through its elastic rows, a solution of least violation.  While the instance has no incumbent,
  // that solution is completed afresh (EVERY disjunction undecided again) and the completion becomes the next probe: its one child, as for a
  // repair root.  At most pump_max generations (the high nibble of the record's size mark counts them).
  bool pumped = false;
  if (viol > FEAS_TOL && okq == 1 && B.pump_max > 0 && B.pool_big && is_probe_word(B.batch_depth[node]) && (int)(big_parent >> 4) < B.pump_max
      && (!(fmin(B.live_inc ? inc_from_key(*(volatile unsigned long long*)&B.inc_key[inst]) : B.inc_obj[inst], B.inc_ext[inst]) < 1e300)
          || (B.pump_inc && B.batch_obj[node] + B.inst_const[inst] < fmin(B.inc_obj[inst], B.inc_ext[inst])))) pumped = true;
  if (!pumped && (viol > FEAS_TOL || okq != 1)) { FREE_NODE(); return; }  // infeasible relaxation, or abandoned at the incumbent cutoff
  if (pumped) {
    for (int k = lane; k < Y.f_rmask + C * N * 2 && k < Y.fixlen; k += 64) { fix[k] = (signed char)-1; comp[k] = (signed char)-1; }
    __syncthreads();
    if (B.stats && lane == 0) atomicAdd(&B.stats[62], 1ull);
  }
  // soft obstacles that this node ignores cost WEIGHTS_SLACK_OBSTACLE each (obstacle_environment_constraints.mod:85-91)
  int nign = 0;
  for (int k = lane; k < C * Y.O * N * 5; k += 64) nign += fix[Y.f_obs + k] >= Y.L ? 1 : 0;
  nign = (int)wave_sum((double)nign);
  const double obj = B.batch_obj[node] + B.inst_const[inst] + nign * D[Y.d_misc + 1];   // primal value: what an incumbent costs
  // what the node proves: the dual value of its relaxation (the primal value of an interior point iterate lies above the
  // optimum of the relaxation by the remaining complementarity)
  const double objlb = pumped ? B.lower_bound[inst] : obj - fmax(0.0, B.batch_bound[node]);   // (a re-rounded probe proves nothing: its child carries the instance's bound)
  // own incumbent or the one another rank of a tree split found.  The incumbent of the START of the round (select_kernel
  // copied it): nodes of one round are evaluated in no fixed order, and pruning with an incumbent another node of the same
  // round has just found would make the tree - and with it the returned solution within the gap - differ from run to run.
  const double inc_now = fmin(B.live_inc ? inc_from_key(*(volatile unsigned long long*)&B.inc_key[inst]) : B.inc_obj[inst], B.inc_ext[inst]);
  if (inc_now < 1e300 && !(objlb < inc_now - 1e-12 * fabs(inc_now))) { if (B.stats && lane == 0) atomicAdd(&B.stats[40], 1ull); FREE_NODE(); return; }  // bound not better than the incumbent
  if (B.stats && lane == 0 && inc_now < 1e300 && (inc_now - objlb) <= B.inst_gap[inst] * (1e-10 + fabs(inc_now))) atomicAdd(&B.stats[41], 1ull);   // (children will be pruned by the gap)
  const double tol = FEAS_TOL;
  const int NCI = C * (N - 1);
  const double gapi_ = B.inst_gap[inst];
  bool dead_lane = false;
  // ---------------- phase R: region alternatives per (c, i)
  for (int L0 = 0; L0 < NCI; L0 += 64) {
    int L = L0 + lane;
    if (L < NCI) {
      int c = L / (N - 1), i = 1 + L % (N - 1);
      const double* z = Z + i * NZ;
      CarState s = {z[6 * c], z[6 * c + 1], z[6 * c + 2], z[6 * c + 3], z[6 * c + 4], z[6 * c + 5], z[6 * C + 2 * c], z[6 * C + 2 * c + 1]};
      bool wj = i <= N - 2;
      int np = T[Y.i_nposs + c];
      int nxt = (i + 1 < N) ? (int)fix[Y.f_reg + c * N + i + 1] : -1;
      int nxtq = (nxt >= 0 && (nxt & 3) == 3) ? (nxt >> 2) : -1;  // next step frozen to this region
      // canonical choice (ties at sector borders are common): the first non-slow alternative, in (region, half-plane)
      // order, whose rows hold within tol; else the slow alternative if it holds; else the least violated one
      double bv = 1e300; int bc = -1; bool found = false;
      const bool want_score = ((B.seq_kinds >> 8) & 15) >= 10;
      LiftDiag LX = {0.0, 0.0, 0.0, 0.0}, LYd = {0.0, 0.0, 0.0, 0.0}; if (want_score) lift_diag(Y, D, c, i, LX, LYd);   // (the score of the branching orders 10 / 11 only)
      double ml = 1e300;
      const unsigned long long allow = ((unsigned long long)(unsigned int)T[Y.i_allow + (c * N + i) * 2 + 1] << 32) | (unsigned int)T[Y.i_allow + (c * N + i) * 2];
      int rset = region_set(Y, T, fix, c, i);
      // Region set tightening (bound propagation from the node's dual solution): a region of an undecided step whose every
      // alternative costs more than the incumbent allows can never hold in an improving descendant of this node - it leaves the
      // set, for the children and everything below them.  The smaller set tightens the hull boxes of the step in their
      // relaxations (region_hull) and shortens their region branchings.
      if (fix[Y.f_reg + c * N + i] < 0 && inc_now < 1e300 && !(B.seq_kinds & 0x8000)) {
        LiftBlk BX, BY; lift_blocks(Y, D, c, i, BX, BY);
        const double room = inc_now - gapi_ * (1e-10 + fabs(inc_now)) - objlb;   // what a descendant may cost on top of this node's dual value
        int nleft = 0, last_code = -1;
        for (int q = 0; q < np; ++q) {
          if (!((rset >> q) & 1)) continue;
          if (nxtq >= 0 && nxtq != q) { rset &= ~(1 << q); continue; }   // the next step is frozen to another region (slow alternative fixed there)
          bool any = false;
          const int nhq = T[Y.i_nhs + c * P + q];
          double bxq, fq, l0q, lfq;   // (the region's table is read once for its up to four alternatives)
          region_viol_parts(Y, D, c, q, s, wj, bxq, fq); region_lift_parts(Y, D, c, q, s, wj, BX, BY, l0q, lfq);
          for (int h = 0; h < 4; ++h) {
            if ((h < 3 && h >= nhq) || !((allow >> (q * 4 + h)) & 1ull)) continue;
            if (region_lift_from(Y, D, T, c, q, h, s, BX, BY, l0q, lfq) * (1.0 - 1e-6) >= room && region_viol_from(Y, D, T, c, q, h, s, bxq, fq) > tol) continue;   // this alternative cannot pay off below this node (one that holds at the node's solution always stays)
            any = true; nleft++; last_code = q * 4 + h;
          }
          if (!any) rset &= ~(1 << q);
        }
        (void)nleft; (void)last_code;   // (deciding the last alternative for the children was tried: no gain, and it can label a tie differently from the canonical completion)
        if (rset != region_set(Y, T, fix, c, i)) {
          const int idx = Y.f_rmask + 2 * (c * N + i);
          fix[idx] = comp[idx] = (signed char)(rset & 255); fix[idx + 1] = comp[idx + 1] = (signed char)((rset >> 8) & 255);
          if (B.stats) atomicAdd(&B.stats[58], 1ull);
        }
      }
      if (rset == 0 && fix[Y.f_reg + c * N + i] < 0) dead_lane = true;   // no region left at this undecided step: nothing below this node can improve the incumbent
      // (the completion below looks at every statically possible alternative, whatever the node's set: it labels the solution
      // of THIS node, and the labels stay the canonical ones - first alternative in order that holds)
      for (int q = 0; q < np; ++q) {
        if (nxtq >= 0 && nxtq != q) { slowv[(c * N + i) * P + q] = 3.0e38f; continue; }
        double bxq, fq; region_viol_parts(Y, D, c, q, s, wj, bxq, fq);   // (once per region, not once per alternative)
        slowv[(c * N + i) * P + q] = !((allow >> (q * 4 + 3)) & 1ull) ? 3.0e38f : (float)fmin(region_viol_from(Y, D, T, c, q, 3, s, bxq, fq), 3.0e38);
        int nh = T[Y.i_nhs + c * P + q];
        for (int h = 0; h < nh; ++h) {
          if (!((allow >> (q * 4 + h)) & 1ull)) continue;  // unreachable velocity set (host presolve)
          double v = region_viol_from(Y, D, T, c, q, h, s, bxq, fq);
          if (want_score) ml = fmin(ml, region_alt_lift(Y, D, T, c, q, h, s, wj, LX, LYd));
          if (!found && v <= tol) { found = true; bv = 0.0; bc = q * 4 + h; }
          if (!found && v < bv) { bv = v; bc = q * 4 + h; }
        }
      }
      fastv[c * N + i] = bv; fastc[c * N + i] = bc; vflag[c * N + i] = 0; rlift[c * N + i] = ml < 1e300 ? ml : 0.0;
    }
  }
  __syncthreads();
  PROF_T(te2);
  if (__ballot(dead_lane)) { if (B.stats && lane == 0) atomicAdd(&B.stats[59], 1ull); FREE_NODE(); return; }
  if (lane < C) {  // sequential resolution of the freeze (slow => same region as the previous step)
    int c = lane; int prevj = T[Y.i_initj + c];
    int np = T[Y.i_nposs + c];
    for (int i = 1; i < N; ++i) {
      int code = (int)fix[Y.f_reg + c * N + i];
      if (code < 0) {
        double bv = fastv[c * N + i]; int bc = fastc[c * N + i];
        for (int q = 0; q < np; ++q)
          if (T[Y.i_regj + c * P + q] == prevj) {
            double v = (double)slowv[(c * N + i) * P + q];
            if (bv > tol && (v <= tol || v < bv)) { bv = v <= tol ? 0.0 : v; bc = q * 4 + 3; }
          }
        if (bc < 0 || bv > tol) { vflag[c * N + i] = 1; if (bc < 0) bc = 0; }
        comp[Y.f_reg + c * N + i] = (signed char)bc;
        code = bc;
      }
      prevj = T[Y.i_regj + c * P + (code >> 2)];
    }
  }
  __syncthreads();
  // ---------------- phase L: leaf disjunctions; every lane keeps its most urgent violated disjunction
  BranchDesc mine; mine.prio = 0x7FFFFFFF; mine.kind = 0; mine.c = 0; mine.o = 0; mine.i = 0; mine.pt = 0; mine.cause = 0;
  // branching order.  Until an incumbent exists: earliest violated step first (a dive then fixes the horizon front to
  // back and reaches a feasible leaf fast).  Afterwards the order selected by bits 8..11 of seq_kinds; default 5 =
  // car/car disjunctions before obstacle, environment and region ones, the most violated first within a kind (measured:
  // 250 instead of 184 of 256 headline instances reach 1 % in 10 s; car/car decisions fix the homotopy class, and the
  // many region alternatives are only enumerated inside a class).
  const int prio_mode = inc_now < 1e300 ? ((B.seq_kinds >> 8) & 15) : 0;
  double myvmax = 0.0;   // largest violation this lane saw (the rounding probe is only worth its QP at nearly integral nodes)
  unsigned long long vsites = 0ull; int cur_site = 0;
  // ties between disjunctions of equal priority go, as when a lane was a site, to the lower lane of then, and there to the one considered first
  unsigned int mine_tie = 0xFFFFFFFFu, cur_tie = 0u;   // the (car, step) / (pair, step) sites with a violated disjunction, one bit per site as when a lane was a site (bal_viol below counts them)
  auto consider = [&](int step, int kind, int c, int o, int pt, double vv, double sc = 0.0, int cause = 0) {
    myvmax = fmax(myvmax, vv); vsites |= 1ull << (cur_site & 63);
    int major = step * 4 + kind;
    // score modes: sc = the smallest lift over the alternatives of the disjunction, i.e. what the bound gains at least on
    // every child (a surrogate of strong branching); 10: largest score first, 11: the same inside the kind order of mode 5
    const int sci = 65535 - (int)fmin(65535.0, sc * 16.0);
    if (prio_mode == 10) major = sci;
    else if (prio_mode == 11) major = (3 - kind) * 65536 + sci;
    else
    if (prio_mode == 1) major = (3 - kind) * 32 + step;
    else if (prio_mode == 2) major = kind * 32 + step;
    else if (prio_mode == 3) major = 100000 - (int)(fmin(vv, 99.0) * 1000.0);
    else if (prio_mode == 4) major = (31 - step) * 4 + kind;
    else if (prio_mode == 5) major = (3 - kind) * 32768 + (30000 - (int)(fmin(vv, 29.0) * 1000.0));   // kind-major, most violated first
    else if (prio_mode == 6) major = (3 - kind) * 32 + (31 - step);                                   // kind-major, latest step first
    else if (prio_mode == 8) major = kind == 3 ? (30000 - (int)(fmin(vv, 29.0) * 1000.0)) : 32768 * (4 - kind) + step;   // car/car most violated, the rest earliest step
    else if (prio_mode == 9) major = (kind == 3 ? 0 : 1 + kind) * 32768 + (30000 - (int)(fmin(vv, 29.0) * 1000.0));  // car/car, region, env, obstacle; most violated
    else if (prio_mode == 7) major = (kind == 3 ? 0 : 1 + kind) * 32 + step;                          // car/car, then region, env, obstacle
    int prio = (major << 12) | ((c & 7) << 9) | ((o & 31) << 4) | (pt & 15);
    if (prio < mine.prio || (prio == mine.prio && cur_tie < mine_tie)) { mine_tie = cur_tie; mine.prio = prio; mine.kind = kind; mine.c = c; mine.o = o; mine.i = step; mine.pt = pt; mine.cause = cause; }
  };
  // one lane per (car, step, POINT) - a lane per (car, step) working through the five points kept 38 of the 64 lanes busy for five passes
  for (int L0 = 0; L0 < NCI * 5; L0 += 64) {
    int L = L0 + lane;
    if (L < NCI * 5) {
      const int pt0 = L % 5, ci_ = L / 5; cur_site = ci_;
      int c = ci_ / (N - 1), i = 1 + ci_ % (N - 1);
      const double* z = Z + i * NZ;
      CarState s = {z[6 * c], z[6 * c + 1], z[6 * c + 2], z[6 * c + 3], z[6 * c + 4], z[6 * c + 5], z[6 * C + 2 * c], z[6 * C + 2 * c + 1]};
      const unsigned int tie0_ = (B.opt2 & 0x300000) ? ((unsigned int)((B.opt2 & 0x200000) ? 63 - i : i) << 24) | ((unsigned int)c << 20) : ((unsigned int)(ci_ & 63) << 20) | ((unsigned int)(ci_ >> 6) << 12);   // (experiment, MIQP_OPT2 bits 20 / 21: ties to the earliest / latest step whatever the car)
      cur_tie = tie0_;
      if (pt0 == 0 && vflag[c * N + i]) consider(i, 0, c, 0, 0, fastv[c * N + i], rlift[c * N + i]);
      LiftDiag LX = {0.0, 0.0, 0.0, 0.0}, LYd = {0.0, 0.0, 0.0, 0.0}; if (prio_mode >= 10 || Y.O > 0) lift_diag(Y, D, c, i, LX, LYd);   // (scores: branching orders 10 / 11, and the obstacles' lifts)
      const double rsc = rlift[c * N + i];
      int code = (int)comp[Y.f_reg + c * N + i];
      const double* rt = D + Y.d_reg + (c * P + (code >> 2)) * REGSZ;
      bool runfixed = fix[Y.f_reg + c * N + i] < 0;
      if (Y.E >= 1)
        for (int pt = pt0; pt < pt0 + 1; ++pt) {
          cur_tie = tie0_ | (unsigned int)(1 + pt);
          double X, Yc; point_xy(s, rt, ENV_PT_D[pt][0], ENV_PT_D[pt][1], X, Yc);
          if (Y.E == 1) {
            if (pt > 0 && runfixed) { double v = env_alt_viol(Y, D, T, 0, X, Yc); if (v > tol) consider(i, 0, c, 0, 0, v, rsc, 1); }
            comp[Y.f_env + (c * N + i) * 5 + pt] = 0;
            continue;
          }
          int fx = (int)fix[Y.f_env + (c * N + i) * 5 + pt];
          if (fx >= 0 && !(pt > 0 && runfixed)) continue;
          bool okk; double bv = 1e300, sc = 1e300;
          if (fx >= 0) { bv = env_alt_viol(Y, D, T, fx, X, Yc); okk = bv <= tol; }
          else {
            int be = 0;
            for (int e = 0; e < Y.E; ++e) {
              double v = env_alt_viol(Y, D, T, e, X, Yc); if (v < bv) { bv = v; be = e; }
              if (prio_mode >= 10) {   // lift of piece e: its most expensive edge
                double le = 0.0; const int ne = T[Y.i_envn + e];
                for (int k = 0; k < ne; ++k) { const double* ed = D + Y.d_env + (e * Y.EL + k) * 3; le = fmax(le, lift1(ed[0] * X + ed[1] * Yc - ed[2], ed[0] * ed[0] * LX.p + ed[1] * ed[1] * LYd.p)); }
                sc = fmin(sc, le);
              }
            }
            okk = bv <= tol; comp[Y.f_env + (c * N + i) * 5 + pt] = (signed char)be;
            if ((pt > 0 || (B.opt2 & 0x20000)) && B.probe_margin > 0.0 && bv <= -B.probe_margin) ploose[Y.f_env + (c * N + i) * 5 + pt] = 1;   // (experiment, MIQP_OPT2 bit 17: the rear point as well)
          }
          if (!okk) { if (pt > 0 && runfixed) consider(i, 0, c, 0, 0, bv, rsc, 1); else consider(i, 1, c, 0, pt, bv, sc < 1e300 ? sc : 0.0); }
        }
      for (int o = 0; o < Y.O; ++o)
        for (int pt = pt0; pt < pt0 + 1; ++pt) {
          cur_tie = tie0_ | (unsigned int)(8 + o * 5 + pt);
          int fx = (int)fix[Y.f_obs + ((c * Y.O + o) * N + i) * 5 + pt];
          if (fx >= 0 && (fx >= Y.L || !(pt > 0 && runfixed))) continue;
          double X, Yc; point_xy(s, rt, OBS_PT_D[pt][0], OBS_PT_D[pt][1], X, Yc);
          bool okk; double bv = 1e300, sc = T[Y.i_obssoft + o] ? D[Y.d_misc + 1] : 1e300;   // (a soft obstacle can be ignored at its price)
          if (fx >= 0) { const double* ed = D + Y.d_obs + ((o * N + i) * Y.L + fx) * 3; bv = ed[0] * X + ed[1] * Yc - ed[2]; okk = bv <= tol; }
          else {
            int bk = 0;
            for (int k = 0; k < Y.L; ++k) {
              const double* ed = D + Y.d_obs + ((o * N + i) * Y.L + k) * 3; double v = ed[0] * X + ed[1] * Yc - ed[2]; if (v < bv) { bv = v; bk = k; }
              sc = fmin(sc, lift1(v, ed[0] * ed[0] * LX.p + ed[1] * ed[1] * LYd.p));
            }
            okk = bv <= tol; comp[Y.f_obs + ((c * Y.O + o) * N + i) * 5 + pt] = (signed char)bk;
            if (pt > 0 && B.probe_margin > 0.0 && bv <= -B.probe_margin) ploose[Y.f_obs + ((c * Y.O + o) * N + i) * 5 + pt] = 1;
          }
          if (!okk) { if (pt > 0 && runfixed) consider(i, 0, c, 0, 0, bv, rsc, 2); else consider(i, 2, c, o, pt, bv, sc < 1e300 ? sc : 0.0); }
        }
    }
  }
  if (C >= 2) {
    const int NPI = Y.NP * (N - 1);
    // one lane per (pair, step, GROUP): 19 lanes with four groups each were the longest stretch of the kernel
    for (int L0 = 0; L0 < NPI * 4; L0 += 64) {
      int L = L0 + lane;
      if (L < NPI * 4) {
        const int g0 = L & 3, pi_ = L >> 2; cur_site = pi_;
        int p = pi_ / (N - 1), i = 1 + pi_ % (N - 1); const double* z = Z + i * NZ;
        int c1, c2; pair_cars(p, C, c1, c2);
        CarState s1 = {z[6 * c1], z[6 * c1 + 1], z[6 * c1 + 2], z[6 * c1 + 3], z[6 * c1 + 4], z[6 * c1 + 5], z[6 * C + 2 * c1], z[6 * C + 2 * c1 + 1]};
        CarState s2 = {z[6 * c2], z[6 * c2 + 1], z[6 * c2 + 2], z[6 * c2 + 3], z[6 * c2 + 4], z[6 * c2 + 5], z[6 * C + 2 * c2], z[6 * C + 2 * c2 + 1]};
        int code1 = (int)comp[Y.f_reg + c1 * N + i], code2 = (int)comp[Y.f_reg + c2 * N + i];
        const double* rt1 = D + Y.d_reg + (c1 * P + (code1 >> 2)) * REGSZ; const double* rt2 = D + Y.d_reg + (c2 * P + (code2 >> 2)) * REGSZ;
        double gx12 = 0.0, gy12 = 0.0;
        if (prio_mode >= 10) { LiftDiag A1, B1, A2, B2; lift_diag(Y, D, c1, i, A1, B1); lift_diag(Y, D, c2, i, A2, B2); gx12 = A1.p + A2.p; gy12 = B1.p + B2.p; }   // (the score of the branching orders 10 / 11 only)
        PosBlk PX1, PY1, PX2, PY2; bool have_pos = false;   // (the blocks of the set tightening: fetched when a group of this (pair, step) needs them)
        for (int g = g0; g < g0 + 1; ++g) {
          cur_tie = (B.opt2 & 0x300000) ? ((unsigned int)((B.opt2 & 0x200000) ? 63 - i : i) << 24) | ((unsigned int)p << 20) | (1u << 19) | (unsigned int)g : ((unsigned int)(pi_ & 63) << 20) | (1u << 19) | ((unsigned int)(pi_ >> 6) << 12) | (unsigned int)g;
          bool need1 = g >= 2, need2 = (g == 1 || g == 3);
          int unf = -1;
          if (need1 && fix[Y.f_reg + c1 * N + i] < 0) unf = c1; else if (need2 && fix[Y.f_reg + c2 * N + i] < 0) unf = c2;
          int fx = (int)fix[Y.f_c2c + (p * N + i) * 4 + g];
          if (fx >= 0 && unf < 0) continue;
          bool okk; double bv = 1e300, sc = 1e300;
          if (fx >= 0) { bv = c2c_alt_viol(Y, D, p, i, g, fx, s1, rt1, s2, rt2); okk = bv <= tol; }
          else {
            int ba = 0;
            int am = (T[Y.i_c2callow + p * N + i] >> (4 * g)) & 15;   // alternatives some reachable positions can satisfy (host presolve)
            if (!am) am = 15;
            const int amc = am;
            am &= c2c_set(fx);                                         // ... that the ancestors of this node have not ruled out
            // set tightening (as for the regions): an alternative whose separation row, priced by the bound lifting, costs more
            // than the incumbent allows leaves the group's set for the whole subtree (no child is created for it).  Only for
            // groups whose rows are known (front points need the region of their car).
            // rows as the relaxation of a child would carry them: exact where the region of the car is decided, with the
            // front-point offset bounded over the region set where it is not (add_point)
            const double* rr1 = fix[Y.f_reg + c1 * N + i] >= 0 ? rt1 : nullptr; const double* rr2 = fix[Y.f_reg + c2 * N + i] >= 0 ? rt2 : nullptr;
            const double* K1 = rr1 ? nullptr : region_fpk(Y, D, T, fix, c1, i); const double* K2 = rr2 ? nullptr : region_fpk(Y, D, T, fix, c2, i);
            const bool relax_ok = !Y.relax_front_off && (rr1 || K1) && (rr2 || K2);
            if ((unf < 0 || relax_ok) && inc_now < 1e300 && !(B.seq_kinds & 0x8000)) {
              const double room = inc_now - gapi_ * (1e-10 + fabs(inc_now)) - objlb;
              const bool softg = (g == 0 || g == 3); const double smx = softg ? D[Y.d_smax + i] : 0.0;
              const int am0 = am;
              if (!have_pos) { pos_blocks(Y, D, c1, i, PX1, PY1); pos_blocks(Y, D, c2, i, PX2, PY2); have_pos = true; }
              for (int a = 0; a < 4; ++a) {
                if (!((am >> a) & 1)) continue;
                const double hv = c2c_alt_viol(Y, D, p, i, g, a, s1, rr1, s2, rr2, K1, K2) - smx;   // violation of the hard row of the alternative
                if (hv > tol && lift1(hv, c2c_alt_gamma(g, a, PX1, PY1, rr1, PX2, PY2, rr2)) * (1.0 - 1e-6) >= room) am &= ~(1 << a);
              }
              if (am != am0) { fix[Y.f_c2c + (p * N + i) * 4 + g] = c2c_set_byte(am); if (B.stats) atomicAdd(&B.stats[61], 1ull); }
              if (am == 0) dead_lane = true;
            }
            for (int a = 0; a < 4; ++a) {
              if (!((amc >> a) & 1)) continue;   // (the completion labels this node's own solution: every statically possible alternative)
              double v = c2c_alt_viol(Y, D, p, i, g, a, s1, rt1, s2, rt2); if (v < bv) { bv = v; ba = a; }
              if (prio_mode >= 10) sc = fmin(sc, lift1(v, a < 2 ? gx12 : gy12));
            }
            okk = bv <= tol; comp[Y.f_c2c + (p * N + i) * 4 + g] = (signed char)ba;
          }
          if (!okk) {
            // an undecided group on a front point of a car without region: when the relaxed rows of all its alternatives cut the
            // node's solution off, the group itself is branched (its children carry those rows: progress without the 3 - 5
            // children of a region decision); when a relaxed row holds and only the exact one fails, the region comes first
            bool branch_group = unf < 0;
            if (unf >= 0 && fx < 0 && !Y.relax_front_off) {
              const double* rr1 = fix[Y.f_reg + c1 * N + i] >= 0 ? rt1 : nullptr; const double* rr2 = fix[Y.f_reg + c2 * N + i] >= 0 ? rt2 : nullptr;
              const double* K1 = rr1 ? nullptr : region_fpk(Y, D, T, fix, c1, i); const double* K2 = rr2 ? nullptr : region_fpk(Y, D, T, fix, c2, i);
              if ((rr1 || K1) && (rr2 || K2)) {
                int amr = (T[Y.i_c2callow + p * N + i] >> (4 * g)) & 15; if (!amr) amr = 15; amr &= c2c_set((int)fix[Y.f_c2c + (p * N + i) * 4 + g]);
                double best_rel = 1e300;   // zero-slack violation of the relaxed rows (a soft group's child pays for it through its quadratic-soft row)
                for (int a = 0; a < 4; ++a) if ((amr >> a) & 1) best_rel = fmin(best_rel, c2c_alt_viol(Y, D, p, i, g, a, s1, rr1, s2, rr2, K1, K2));
                branch_group = best_rel > tol;
              }
            }
            if (!branch_group) consider(i, 0, unf, 0, 0, bv, rlift[unf * N + i], 3); else consider(i, 3, p, g, 0, bv, sc < 1e300 ? sc : 0.0);
          }
        }
      }
    }
  }
  if (__ballot(dead_lane)) { if (B.stats && lane == 0) atomicAdd(&B.stats[59], 1ull); FREE_NODE(); return; }   // a car/car group without an alternative left
  // wave-wide most urgent disjunction
  int best = mine.prio;
  for (int o = 32; o > 0; o >>= 1) best = min(best, __shfl_xor(best, o));
  __syncthreads();
  PROF_T(te3);
  if (best == 0x7FFFFFFF && !pumped) {   // (the solution of a re-rounded probe violates rows of its own: only its completion, solved as the next probe, can be an incumbent)
    if (B.stats && lane == 0) atomicAdd(&B.stats[42], 1ull);
    // integer feasible: candidate incumbent.  The winner of the 64-bit atomicMin owns the low 20 bits (batch slot).
    signed char* dst = B.batch_comp + (size_t)node * Y.fixlen;
    for (int k = lane; k < Y.fixlen; k += 64) dst[k] = comp[k];
    int hsh = 0;
    for (int k = lane; k < Y.fixlen; k += 64) hsh += ((int)comp[k] + 3) * (2 * k + 1);
    for (int o = 32; o > 0; o >>= 1) hsh += __shfl_xor(hsh, o);
    if (lane == 0) {
      // ties in the resolved 44 bits of the objective go to the smaller hash of the completed record, not to the smaller batch slot
      // (slots are handed out in arrival order: the same solve would keep one or the other solution from run to run)
      unsigned long long key = (d2key(obj) & ~0xFFFFFull) | (unsigned long long)((unsigned int)hsh & 0xFFFFFu);
      B.batch_obj[node] = obj;
      B.batch_candkey[node] = key; B.batch_candinst[node] = inst;
      if (B.ring_M && B.batch_Mtag && !(B.batch_large && B.batch_large[node] <= 1)) B.batch_Mtag[node] = 0ull;   // (an interior point node: the slot's active set is an earlier round's - select_kernel copies the candidate's to the incumbent)
      atomicMin(&B.inc_key[inst], key);
      atomicAdd(&B.inst_ninc[inst], 1);
    }
    FREE_NODE();
    return;
  }
  unsigned long long bal_viol = vsites;   // sites that saw a violated disjunction (the lanes of the time when a lane was a site)
  for (int o = 32; o > 0; o >>= 1) bal_viol |= ((unsigned long long)(unsigned int)__shfl_xor((int)(bal_viol >> 32), o) << 32) | (unsigned int)__shfl_xor((int)bal_viol, o);
  const double vmax_all = wave_max(myvmax);
  unsigned int best_tie = mine.prio == best ? mine_tie : 0xFFFFFFFFu;
  for (int o = 32; o > 0; o >>= 1) best_tie = min(best_tie, (unsigned int)__shfl_xor((int)best_tie, o));
  unsigned long long bal = __ballot(mine.prio == best && (best == 0x7FFFFFFF || mine_tie == best_tie));
  int winner = __ffsll((long long)bal) - 1;
  if (lane == winner) { chosen = mine; if (best == 0x7FFFFFFF) chosen.i = 1; }   // (no violated disjunction: a re-rounded probe - any valid step serves, it has no children but the probe)
  __syncthreads();
  // ---------------- first-deviation branching over the time family of the chosen disjunction.
  // The same kind of decision exists at every step j (family: same car / obstacle / point / group).  With the completed
  // choices ref_j of the still undecided steps as reference sequence, the children are
  //     child_inf      : every undecided step j of the window takes ref_j
  //     child (k, b)   : undecided steps j < k take ref_j, step k takes b != ref_k, later steps stay undecided
  // which is an exhaustive and disjoint partition of all assignments on the window (decisions persist over time in this
  // model, so child_inf usually carries the whole homotopy class and the deviating children are pruned quickly).
  int nalt = 0;
  if (lane == 0) {
    BranchDesc d = chosen; const int i = d.i;
    if (B.stats && d.kind == 0) {   // diagnostic: what flagged the region disjunction, and for its own rows the class of the worst one
      atomicAdd(&B.stats[64 + (d.cause & 3)], 1ull);
      atomicAdd(&B.stats[160 + (i < 31 ? i : 31)], 1ull);
      if (d.cause == 0) {
        const int code = (int)comp[Y.f_reg + d.c * N + i], q = code >> 2, h = code & 3;
        const double* z = Z + i * NZ; const double* rt = D + Y.d_reg + (d.c * P + q) * REGSZ; const double vm = D[Y.d_glob + 6];
        const CarState s = {z[6 * d.c], z[6 * d.c + 1], z[6 * d.c + 2], z[6 * d.c + 3], z[6 * d.c + 4], z[6 * d.c + 5], z[6 * C + 2 * d.c], z[6 * C + 2 * d.c + 1]};
        double cv[6];
        cv[0] = fmax(fmax(s.ax - rt[12], rt[11] - s.ax), fmax(s.ay - rt[14], rt[13] - s.ay));
        cv[1] = i <= N - 2 ? fmax(fmax(s.ux - rt[16], rt[15] - s.ux), fmax(s.uy - rt[18], rt[17] - s.uy)) : -1e300;
        if (h == 3) { cv[2] = cv[3] = cv[4] = -1e300; cv[5] = fmax(fabs(s.vx) - vm, fabs(s.vy) - vm); }
        else {
          const int* hs = T + Y.i_hs + ((d.c * P + q) * 2 + h) * 2;
          cv[2] = fmax(rt[0] * s.vx + rt[1] * s.vy, rt[2] * s.vx + rt[3] * s.vy);
          cv[3] = vm - hs[1] * (hs[0] == 0 ? s.vx : s.vy);
          cv[4] = fmax(s.ay - rt[4] * s.ax - rt[6] * s.vx - rt[7] * s.vy - rt[5], -s.ay + rt[4] * s.ax + rt[9] * s.vx + rt[10] * s.vy + rt[8]);
          cv[5] = -1e300;
        }
        int bk = 0; for (int k = 1; k < 6; ++k) if (cv[k] > cv[bk]) bk = k;
        atomicAdd(&B.stats[70 + bk], 1ull);
      }
    }
    int base, stride;
    if (d.kind == 0) { base = Y.f_reg + d.c * N; stride = 1; }
    else if (d.kind == 1) { base = Y.f_env + (d.c * N) * 5 + d.pt; stride = 5; }
    else if (d.kind == 2) { base = Y.f_obs + ((d.c * Y.O + d.o) * N) * 5 + d.pt; stride = 5; }
    else { base = Y.f_c2c + (d.c * N) * 4 + d.o; stride = 4; }
    // alternatives of step j other than ref_j; returns count, writes into tmp
    auto alts_of = [&](int j, int* tmp) -> int {
      int refv = (int)comp[base + j * stride]; int n = 0;
      if (d.kind == 0) {
        int c = d.c; int np = T[Y.i_nposs + c];
        int pv_ = j == 1 ? -1 : (int)comp[base + (j - 1) * stride];          // previous step: decided or reference value
        int prevj = j == 1 ? T[Y.i_initj + c] : T[Y.i_regj + c * P + (pv_ >> 2)];
        int nxt = (j + 1 < N) ? (int)fix[base + (j + 1) * stride] : -1;
        int nxtq = (nxt >= 0 && (nxt & 3) == 3) ? (nxt >> 2) : -1;
        const unsigned long long allow_b = ((unsigned long long)(unsigned int)T[Y.i_allow + (c * N + j) * 2 + 1] << 32) | (unsigned int)T[Y.i_allow + (c * N + j) * 2];
        const int rset_j = region_set(Y, T, fix, c, j);
        for (int q = 0; q < np; ++q) {
          if (!((rset_j >> q) & 1)) continue;   // the node has ruled this region out at step j
          if (nxtq >= 0 && nxtq != q) continue;
          int nh = T[Y.i_nhs + c * P + q];
          for (int h = 0; h < 4; ++h) {
            if (h < 3 && h >= nh) continue;
            if (!((allow_b >> (q * 4 + h)) & 1ull)) continue;
            if (h == 3 && prevj != T[Y.i_regj + c * P + q]) continue;
            if (q * 4 + h == refv) continue;
            if (n < 63) tmp[n++] = q * 4 + h;
          }
        }
      } else if (d.kind == 1) { for (int e = 0; e < Y.E && n < 63; ++e) if (e != refv) tmp[n++] = e; }
      else if (d.kind == 2) { int na = Y.L + (T[Y.i_obssoft + d.o] ? 1 : 0); for (int k = 0; k < na && n < 63; ++k) if (k != refv) tmp[n++] = k; }
      else { int am = (T[Y.i_c2callow + d.c * N + j] >> (4 * d.o)) & 15; if (!am) am = 15; am &= c2c_set((int)fix[base + j * stride]);
             for (int a2 = 0; a2 < 4; ++a2) if (a2 != refv && ((am >> a2) & 1)) tmp[n++] = a2; }
      return n;
    };
    // window of undecided steps: all of them when the children fit, else the steps from the violated one onwards
    int tmp[64];   // alternatives of one disjunction: the host admits at most 62 (batch_layout), region alternatives are <= 15 x 3
    int jlo = 1, jhi = N - 1, total = 1, n_i = -1;
    // (plain K-way branching on step i is the default for every kind: its alternatives are enumerated ONCE - the count over all undecided steps,
    // which only the first-deviation family needs, cost N - 1 enumerations per node for nothing)
    if (!((B.seq_kinds >> d.kind) & 1)) { jlo = i; jhi = i; n_i = alts_of(i, tmp); total = 1 + n_i; }
    else { for (int j = 1; j < N; ++j) if (fix[base + j * stride] < 0) total += alts_of(j, tmp); }
    if (((B.seq_kinds >> d.kind) & 1) && (total > 63 || ((B.seq_kinds >> 20) & 15))) {
      // window: from the violated step onwards (steps before it hold in the relaxation and stay undecided), at most
      // `win` steps when that experiment switch is set
      const int win = (B.seq_kinds >> 20) & 15;
      jlo = i; total = 1; jhi = i - 1;
      for (int j = i; j < N; ++j) {
        if (win && j >= i + win) break;
        if (fix[base + j * stride] >= 0) { jhi = j; continue; }
        int n = alts_of(j, tmp);
        if (total + n > 63) break;
        total += n; jhi = j;
      }
    }
    // MIP-start repair (CPLEX repairtries, src/cplex_wrapper.cpp:494-639 with cplexmodel.mod:8-21): the root that carries only the
    // REGION binaries of a start (depth word REPAIR_ROOT) is not part of the tree - the tree's own root covers it.  When the
    // completion of its relaxation is not integer feasible it gets one child, the rounding probe (every undecided disjunction
    // fixed to its completed value: feasible -> incumbent), and nothing else.
    const bool repair_root = B.batch_depth[node] == REPAIR_ROOT || pumped;
    if (repair_root) nalt = 0;
    else {
    ck[0] = N; ca[0] = 0; nalt = 1;                       // child_inf
    for (int j = jlo; j <= jhi; ++j) {
      if (fix[base + j * stride] >= 0) continue;
      int n = (j == i && n_i >= 0) ? n_i : alts_of(j, tmp);   // (plain branching: tmp still holds the alternatives of step i)
      for (int q = 0; q < n && nalt < 63; ++q) { ck[nalt] = j; ca[nalt] = tmp[q]; nalt++; }
    }
    }
    // rounding probe (until the instance has an incumbent, and afterwards at nearly integral nodes - at most K (car | pair, step)
    // sites with a violated disjunction, K = 8: 256 instances 9.1 -> 6.5 s, 90 % quantile of the finish time 1.5 -> 0.75 s):
    // one extra child with EVERY undecided disjunction fixed to its
    // completed value.  It lies inside the first child, so the children stay exhaustive; its relaxation is the exact cost
    // of the rounding and, when feasible, the first incumbent two rounds after the root instead of one dive level per round
    // (a probe at a node whose dual value is within probe_room gaps of the incumbent can only improve the incumbent by that little: not worth its QP - one of the large ones)
    // which nodes get a sampled probe: by the bits of the node's own objective (its record id is handed out in no fixed order
    // and would make the tree differ from run to run)
    const unsigned int probe_hash = (unsigned int)((unsigned long long)__double_as_longlong(obj) >> 6);
    const bool probe_room_ok = !(inc_now < 1e300) || (inc_now - objlb) > B.probe_room * B.inst_gap[inst] * (1e-10 + fabs(inc_now));
    if (probe_room_ok && (repair_root || !(inc_now < 1e300) || (B.opt2 & 1) || (((B.opt2 >> 4) & 15) && __popcll(bal_viol) <= ((B.opt2 >> 4) & 15) && ((probe_hash * 2654435761u >> 16) & ((1u << (2 * ((B.opt2 >> 2) & 3))) - 1u)) == 0u && (!(B.opt2 >> 8) || vmax_all <= 0.05 * (double)(B.opt2 >> 8)))) && nalt < 63 && (B.seq_kinds & 0x4000000) == 0) { ck[nalt] = -2; ca[nalt] = 0; nalt++; }
    fam[0] = base; fam[1] = stride; fam[2] = jlo; fam[3] = jhi;
    sh_base[2] = nalt;
  }
  __syncthreads();
  PROF_T(te4);
  nalt = sh_base[2];
  const int base = fam[0], stride = fam[1], jlo = fam[2], jhi = fam[3];
  // ---------------- bound lifting.  With the node's multipliers fixed its Lagrangian grows like 1/2 (z - z*)' H (z - z*)
  // around the solution z* (H: the objective's Hessian on the trajectories of the dynamics - host tables d_lift); a child
  // whose fix record activates the hard row g.z <= r, violated by v at z*, therefore costs at least
  //     (dual value of the node) + v^2 / (2 g Sigma g').
  // Every row the child's record activates at the branching step is decoded exactly as the interior point kernels will see
  // it; the largest single-row lift is the child's bound.  Children that cannot beat the incumbent any more are not created.
  const bool plain = jlo == jhi && !(B.seq_kinds & 0x20000);   // (experiment switch: lifting off)
  if (plain) {
    const int i = chosen.i;
    for (int k = lane; k < Y.fixlen; k += 64) cfix[k] = fix[k];
    __syncthreads();
    const double* LT = D + Y.d_lift;
    double* gl = gsc + (lane & (LIFT_LANES - 1)) * (NZ + 1);   // (lanes >= LIFT_LANES sit the row passes out)
    for (int a = 0; a < nalt; ++a) {
      const int kk = ck[a];
      if (kk == -2) { if (lane == 0) clift[a] = (a > 0 && ck[0] != -2) ? clift[0] : 0.0; continue; }   // the probe lies inside the first child: it costs at least what that one costs
      int negidx = -1, negm = 0;
      if (chosen.kind == 3 && kk >= 0 && kk < N && (B.seq_kinds & 0x10000) == 0) {
        negidx = Y.f_c2n + (chosen.c * N + kk) * 4 + chosen.o;
        negm = 1 << (int)comp[base + kk * stride];
        for (int a2 = 1; a2 < a; ++a2) if (ck[a2] == kk) negm |= 1 << ca[a2];
      }
      if (lane == 0) { cfix[base + i * stride] = kk == i ? (signed char)ca[a] : comp[base + i * stride]; if (negidx >= 0) cfix[negidx] = (signed char)negm; }
      __syncthreads();
      double lift = 0.0;
      // slots whose row can depend on the changed byte: a region code touches every slot of its car and the car/car slots,
      // the other kinds only their own rows
      int s0 = 0, s1 = Y.NSLOT, s2 = 0, s3 = 0;   // [s0, s1) and [s2, s3)
      if (chosen.kind == 0) { s0 = chosen.c * Y.SC; s1 = s0 + Y.SC; s2 = C * Y.SC; s3 = Y.NSLOT; }
      else if (chosen.kind == 1) { s0 = chosen.c * Y.SC + 16 + chosen.pt * Y.EL; s1 = s0 + Y.EL; }
      else if (chosen.kind == 2) { s0 = chosen.c * Y.SC + 16 + 5 * Y.EL + chosen.o * 5 + chosen.pt; s1 = s0 + 1; }
      else { s0 = C * Y.SC + chosen.c * 8 + chosen.o * 2; s1 = s0 + 2; s2 = C * Y.SC + 8 * Y.NP + chosen.c * 16 + chosen.o * 4; s3 = s2 + 4; }
      const int n01 = s1 - s0, ntot = n01 + (s3 - s2);
      // Single-row lift of every violated row, and the rows for the multi-row lift: with the node's multipliers fixed the child
      // costs at least  max_{lambda >= 0}  lambda' v - 1/2 lambda' (G Sigma G') lambda  over ANY set of its rows (weak duality of
      // min { 1/2 dz' H dz : G dz <= -v } on the trajectories; every lambda >= 0 gives a valid bound, so a few coordinate-ascent
      // sweeps suffice).  Rows of the child at the branching stage that are violated or close to active are collected - rows the
      // node already satisfies matter too: the step that repairs one row runs into the others (sector against half-plane,
      // curvature against the acceleration box).
      const bool multi = (B.seq_kinds & 0x80000000u) != 0u;   // OFF by default (bit 31 of MIQP_SEQ_KINDS switches it on): measured without effect on the node counts (17.2 M against 17.4 M nodes on a 1024-instance queue; tools/lift_ab.sh) - what a child costs beyond its single-row lift comes from the rows of the OTHER stages
      int ncand = 0;
      for (int e0 = 0; e0 < ntot; e0 += LIFT_LANES) {
        const int e = e0 + lane;
        bool cand = false; double v = 0.0, gam = 0.0;
        if (lane < LIFT_LANES && e < ntot) {
          const int sl = e < n01 ? s0 + e : s2 + (e - n01);
          const RowOut r = decode_row<C, true>(Y, D, T, cfix, i, sl, gl);
          if (r.active && r.aq == 0.0) {
            v = -r.rhs;
#pragma unroll
            for (int q = 0; q < NZ; ++q) v += gl[q] * Z[i * NZ + q];
            if (v > -LIFT_MARGIN) {
              for (int c = 0; c < C; ++c)
                for (int ax = 0; ax < 2; ++ax) {
                  const double g4[4] = {gl[6 * c + 3 * ax], gl[6 * c + 3 * ax + 1], gl[6 * c + 3 * ax + 2], gl[6 * C + 2 * c + ax]};
                  if (g4[0] == 0.0 && g4[1] == 0.0 && g4[2] == 0.0 && g4[3] == 0.0) continue;
                  const double* S4 = LT + ((c * 2 + ax) * N + i) * 16;
#pragma unroll
                  for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int q = 0; q < 4; ++q) gam += g4[p] * S4[p * 4 + q] * g4[q];
                }
              if (gam > 1e-300 && gam < 1e200) { cand = true; if (v > 1e-7) lift = fmax(lift, 0.5 * v * v / gam); }
            }
          }
        }
        if (multi) {   // the candidates of this pass join the list (violated rows and nearly active ones alike; first come, first served)
          const unsigned long long mk = __ballot(cand);
          const int pos = ncand + __popcll(mk & ((1ull << lane) - 1ull));
          if (cand && pos < LIFT_ROWS) {
#pragma unroll
            for (int q = 0; q < NZ; ++q) mr_g[pos * NZ + q] = gl[q];
            for (int c = 0; c < C; ++c)
              for (int ax = 0; ax < 2; ++ax) {
                const int ix[4] = {6 * c + 3 * ax, 6 * c + 3 * ax + 1, 6 * c + 3 * ax + 2, 6 * C + 2 * c + ax};
                const double* S4 = LT + ((c * 2 + ax) * N + i) * 16;
#pragma unroll
                for (int p = 0; p < 4; ++p) { double y = 0.0;
#pragma unroll
                  for (int q = 0; q < 4; ++q) y += S4[p * 4 + q] * gl[ix[q]];
                  mr_y[pos * NZ + ix[p]] = y; }
              }
            mr_v[pos] = v; mr_d[pos] = gam; mr_l[pos] = 0.0;
          }
          ncand += __popcll(mk);
        }
      }
      lift = wave_max(lift);
      if (ncand > LIFT_ROWS) ncand = LIFT_ROWS;
      if (multi && ncand >= 2) {
        if (lane < NZ) mr_w[lane] = 0.0;
        __syncthreads();
        for (int sweep = 0; sweep < 3; ++sweep)
          for (int k = 0; k < ncand; ++k) {
            double t = lane < NZ ? mr_g[k * NZ + lane] * mr_w[lane] : 0.0;
            t = wave_sum(t);                                        // g_k . w = sum_s M_ks lambda_s
            const double lk = mr_l[k], dk = mr_d[k];
            double ln = (mr_v[k] - (t - dk * lk)) / dk; if (!(ln > 0.0)) ln = 0.0;
            __syncthreads();
            if (lane < NZ) mr_w[lane] += (ln - lk) * mr_y[k * NZ + lane];
            if (lane == 0) mr_l[k] = ln;
            __syncthreads();
          }
        // value of the dual at the final multipliers: lambda' v - 1/2 (sum lambda g) . w
        double acc = 0.0;
        if (lane < ncand) {
          double gw = 0.0;
          for (int q = 0; q < NZ; ++q) gw += mr_g[lane * NZ + q] * mr_w[q];
          acc = mr_l[lane] * (mr_v[lane] - 0.5 * gw);
        }
        acc = wave_sum(acc);
        if (acc > lift) { lift = acc; if (B.stats && lane == 0) atomicAdd(&B.stats[57], 1ull); }
        __syncthreads();
      }
      if (lane == 0) { clift[a] = lift * (1.0 - 1e-6); cfix[base + i * stride] = fix[base + i * stride]; if (negidx >= 0) cfix[negidx] = fix[negidx]; }
      __syncthreads();
    }
  } else if (lane < 64) clift[lane] = 0.0;
  __syncthreads();
  if (lane == 0) {
    // children that survive, each with its own bound and list tier
    const double cst = B.inst_const[inst], gapi = B.inst_gap[inst];
    const double thr = B.far_cap > 0 && !B.inst_mode[inst] ? B.near_thr[inst] : 1e300;
    int nk = 0, nfar = 0, nnear = 0; double fmin_ = 1e300;
    for (int a = 0; a < nalt; ++a) {
      const double cb = objlb + clift[a];
      if (inc_now < 1e300 && (inc_now - cb) <= gapi * (1e-10 + fabs(inc_now))) { if (B.stats) atomicAdd(&B.stats[56], 1ull); continue; }
      int negm = 0;
      if (chosen.kind == 3 && ck[a] >= 0 && ck[a] < N && (B.seq_kinds & 0x10000) == 0) {
        negm = 1 << (int)comp[base + ck[a] * stride];
        for (int a2 = 1; a2 < a; ++a2) if (ck[a2] == ck[a]) negm |= 1 << ca[a2];
      }
      const bool tf = (cb - cst) > thr && B.batch_depth[node] != REPAIR_ROOT;   // (the rounding probe of a repair / skeleton root carries that root's LOW bound: it stays in the near list, where the next best-bound round takes it - parked in the far tier it held the instance's bound down until the next refill)
      k_ck[nk] = ck[a]; k_ca[nk] = ca[a]; k_neg[nk] = negm; k_ord[nk] = a; k_bnd[nk] = cb - cst;
      k_pos[nk] = tf ? (0x40000000 | nfar++) : nnear++;
      if (tf) fmin_ = fmin(fmin_, cb - cst);
      nk++;
    }
    int fbase = 0, obase = 0, ob2 = 0, extra2 = 0; bool okalloc = true;
    if (nnear > 0) {
      obase = atomicAdd(&B.open_count[inst], nnear);
      if (obase + nnear > B.open_cap) {   // near list full: its reserved slots below the capacity become dead entries, the children wait in the far tier
        for (int q = obase; q < obase + nnear && q < B.open_cap; ++q) { size_t oi = ((size_t)B.open_sel * B.n_slots + slot) * B.open_cap + q; B.open_bound[oi] = 1e300; B.open_node[oi] = -1; B.open_depth[oi] = 0; }
        if (B.far_cap > 0) { for (int q = 0; q < nk; ++q) if (!(k_pos[q] & 0x40000000)) { k_pos[q] = 0x40000000 | nfar++; fmin_ = fmin(fmin_, k_bnd[q]); } }
        else okalloc = false;
        nnear = 0; obase = 0;
      }
    }
    const int nnear_kept = nnear;
    if (nfar > 0) {
      fbase = atomicAdd(&B.far_count[inst], nfar);
      if (fbase + nfar <= B.far_cap) atomicMin(&B.far_minkey[inst], d2key(fmin_));
      else {   // tier full: the reserved slots below the capacity become dead entries, the children go to the near list
        for (int q = fbase; q < fbase + nfar && q < B.far_cap; ++q) { B.far_bound[(size_t)slot * B.far_cap + q] = 1e300; B.far_node[(size_t)slot * B.far_cap + q] = -1; }
        int extra = 0;
        for (int q = 0; q < nk; ++q) if (k_pos[q] & 0x40000000) k_pos[q] = 0x20000000 | extra++;   // second near reservation behind the first
        nfar = 0;
        ob2 = atomicAdd(&B.open_count[inst], extra); extra2 = extra;
        if (ob2 + extra > B.open_cap) {
          okalloc = false;
          for (int q = ob2; q < ob2 + extra && q < B.open_cap; ++q) { size_t oi = ((size_t)B.open_sel * B.n_slots + slot) * B.open_cap + q; B.open_bound[oi] = 1e300; B.open_node[oi] = -1; B.open_depth[oi] = 0; }
        }
        sh_base[3] = ob2;
      }
    }
    if (okalloc && nk > 0) {
      unsigned int h = atomicAdd(B.free_head, (unsigned int)nk);
      if ((int)(*B.free_limit - h) >= nk) {          // recycled records
        for (int q = 0; q < nk; ++q) slots[q] = B.free_q[(h + q) % (unsigned int)B.pool_cap];
      } else {                                          // fresh records
        int nb = atomicAdd(B.pool_count, nk);
        if (nb + nk > B.pool_cap) okalloc = false;
        for (int q = 0; q < nk; ++q) slots[q] = nb + q;
      }
    }
    if (!okalloc) {   // list or record pool exhausted: the instance is flagged incomplete; the reserved list slots become dead entries
      atomicOr(&B.inst_flags[inst], 1);
      for (int q = fbase; q < fbase + nfar; ++q) { B.far_bound[(size_t)slot * B.far_cap + q] = 1e300; B.far_node[(size_t)slot * B.far_cap + q] = -1; }
      for (int q = obase; q < obase + nnear_kept && q < B.open_cap; ++q) { size_t oi = ((size_t)B.open_sel * B.n_slots + slot) * B.open_cap + q; B.open_bound[oi] = 1e300; B.open_node[oi] = -1; B.open_depth[oi] = 0; }
      for (int q = ob2; q < ob2 + extra2 && q < B.open_cap; ++q) { size_t oi = ((size_t)B.open_sel * B.n_slots + slot) * B.open_cap + q; B.open_bound[oi] = 1e300; B.open_node[oi] = -1; B.open_depth[oi] = 0; }   // the second near reservation (far tier overflow)
      nk = 0;
    }
    if (B.stats && nk > 0) { atomicAdd(&B.stats[43], 1ull); atomicAdd(&B.stats[44], (unsigned long long)nk); atomicAdd(&B.stats[48 + chosen.kind], 1ull); atomicAdd(&B.stats[52 + chosen.kind], (unsigned long long)nk); }
    sh_base[0] = fbase; sh_base[1] = obase; sh_base[2] = nk;
  }
  __syncthreads();
  PROF_T(te5);
  const int nk = sh_base[2];
  if (nk <= 0) { FREE_NODE(); return; }
  {
    for (int q = 0; q < nk; ++q) {
      signed char* dst = B.pool_fix + (size_t)slots[q] * Y.fixlen;
      const int kk = k_ck[q]; const signed char av = (signed char)k_ca[q];
      // exclusive children (car/car): the deviating child excludes, at zero slack, the reference alternative of step kk
      // and the alternatives of its earlier siblings at that step
      int negidx = -1; const int negm = k_neg[q];
      if (chosen.kind == 3 && kk >= 0 && kk < N && (B.seq_kinds & 0x10000) == 0) negidx = Y.f_c2n + (chosen.c * N + kk) * 4 + chosen.o;
      if (kk != -2) {
        // an ordinary child differs from its parent in the bytes of the branching family (one byte with the default plain branching) and the
        // exclusion mask: the record is assembled in LDS - the parent's bytes copied 8 at a time, the few bytes patched - and stored 8 bytes per
        // lane (the byte loop below, with a division per byte, was most of this phase; it remains the path of the rounding probe)
        for (int k8 = lane; k8 < Y.fixlen / 8; k8 += 64) ((unsigned long long*)cfix)[k8] = ((const unsigned long long*)fix)[k8];
        __syncthreads();
        for (int j = jlo + lane; j <= jhi; j += 64) { const int k = base + j * stride; if (fix[k] < 0) { if (j < kk) cfix[k] = comp[k]; else if (j == kk) cfix[k] = av; } }
        __syncthreads();
        if (lane == 0 && negidx >= 0) cfix[negidx] = (signed char)negm;
        __syncthreads();
        for (int k8 = lane; k8 < Y.fixlen / 8; k8 += 64) ((unsigned long long*)dst)[k8] = ((const unsigned long long*)cfix)[k8];
        __syncthreads();   // (cfix is reused by the next child)
      } else
      for (int k = lane; k < Y.fixlen; k += 64) {
        signed char v = fix[k];
        int rel = k - base;
        if (v < 0 && rel >= 0 && rel % stride == 0) {
          int j = rel / stride;
          if (j >= jlo && j <= jhi) { if (j < kk) v = comp[k]; else if (j == kk) v = av; }
        }
        if (k == negidx) v = (signed char)negm;
        if (kk == -2) {   // rounding probe
          v = (fix[k] < 0 && k < Y.f_c2n) ? comp[k] : fix[k];
          // (experiment, MIQP_OPT2 bit 16: the probe leaves the front-point environment / obstacle disjunctions undecided - a
          // smaller relaxation; it is then an ordinary node that the completion may still find integer feasible)
          if ((B.opt2 & 0x10000) && fix[k] < 0 && k >= Y.f_env && k < Y.f_c2c && (k - Y.f_env) % 5 != 0) v = fix[k];
          // Front-point environment / obstacle disjunctions whose completed alternative holds with probe_margin to spare at this
          // node's solution stay undecided in the probe: their rows - most of a probe's ~570 general rows - would be inactive
          // anyway.  The probe is then an ordinary node: its own completion checks every disjunction at ITS solution, so it is
          // only an incumbent when all of them hold (and it is branched like any node when one does not).
          if (B.probe_margin > 0.0 && fix[k] < 0 && ploose[k]) v = fix[k];
        }
        dst[k] = v;
      }
      if (B.pool_origin && lane == 0) {   // diagnostic: 4 kind + class of the child (0 the reference alternative; region: 1 an adjacent sector, 2 another, 3 slow); 15 the rounding probe
        int cls = k_ord[q] > 0 ? 1 : 0;
        if (chosen.kind == 0 && cls) { const int rq = (int)comp[base + chosen.i * stride] >> 2, cq = k_ca[q] >> 2; const int dq = T[Y.i_regj + chosen.c * P + cq] - T[Y.i_regj + chosen.c * P + rq]; const int ad = abs(dq) % Y.R;
          cls = (k_ca[q] & 3) == 3 ? 3 : ((ad == 1 || ad == Y.R - 1) ? 1 : 2); }
        B.pool_origin[slots[q]] = (signed char)(kk == -2 ? 15 : 4 * chosen.kind + cls);
      }
    }
    if (B.pool_big && lane < nk) B.pool_big[slots[lane]] = pumped ? (unsigned char)((((big_parent >> 4) + 1) << 4) | 1) : (unsigned char)(big_parent & ~12);   // (bit 3, "deferred once", and bit 2, "the active-set launch could not finish it", are the parent's own)   // (a child has the rows of its parent and more)
    if (B.pool_Z) {   // the children start their relaxation from this node's solution (see DevBuf::pool_Z)
      for (int q = 0; q < nk; ++q) { if (slots[q] >= B.z_cap) continue; double* zd = B.pool_Z + (size_t)slots[q] * N * NZ; for (int k = lane; k < N * NZ; k += 64) zd[k] = Z[k]; }
    }
    if (B.pool_A) {   // ... and, where the active-set launch solved this node, from its active set (the nodes of the larger interior point variant have none)
      const bool as_node = B.batch_A && B.batch_large && B.batch_large[node] <= 1;   // (large_class 0 / 1: one of the two active-set launches solved this node and wrote its set - or 0xFFFF; the batch slots of the interior point's nodes hold what an earlier round left there)
      const unsigned short av_ = as_node ? B.batch_A[(size_t)node * 64 + lane] : (unsigned short)0xFFFFu;
      for (int q = 0; q < nk; ++q) { if (slots[q] >= B.z_cap) continue; B.pool_A[(size_t)slots[q] * 64 + lane] = av_; }
      if (B.ring_M && lane < nk && slots[lane] < B.z_cap) B.pool_Mtag[slots[lane]] = as_node ? B.batch_Mtag[node] : 0ull;
    }
    if (lane < nk) {
      // depth word: (tree depth << 6) | preference among siblings (child_inf first) - used by the dive ordering
      int pd = B.batch_depth[node] >> 6;
      int pref = k_ord[lane];
      if (B.opt2 & 2) {   // experiment: dives prefer the sibling with the smallest lifted bound instead of the sibling order
        pref = 0;
        for (int q = 0; q < nk; ++q) if (k_bnd[q] < k_bnd[lane] || (k_bnd[q] == k_bnd[lane] && q < lane)) pref++;
      }
      // the probe is dived into first; 63 in the low bits is ITS mark (is_probe_word), the siblings count down from 62 - so a probe
      // and the first child it lies in, which carry the same bound, never have the same key and preference (their tie in a
      // best-bound round went to whichever lane came first: the one source of run-to-run differences of single solves that
      // was left, 4 % of the repetitions of some instances)
      const int dw = k_ck[lane] == -2 ? (((pd + 2) << 6) | 63) : (((pd + 1) << 6) | (62 - (pref < 62 ? pref : 62)));
      if (k_pos[lane] & 0x40000000) {
        size_t oi = (size_t)slot * B.far_cap + sh_base[0] + (k_pos[lane] & 0x3FFFFFFF);
        B.far_bound[oi] = k_bnd[lane]; B.far_node[oi] = slots[lane]; B.far_depth[oi] = dw;
      } else {
        const int np_ = (k_pos[lane] & 0x20000000) ? sh_base[3] + (k_pos[lane] & 0x1FFFFFFF) : sh_base[1] + k_pos[lane];
        size_t oi = ((size_t)B.open_sel * B.n_slots + slot) * B.open_cap + np_;
        B.open_bound[oi] = k_bnd[lane]; B.open_node[oi] = slots[lane]; B.open_depth[oi] = dw;
      }
    }
  }
  FREE_NODE();
#ifdef MIQP_PROFILE
  { PROF_T(te6);   // (branched nodes only: the early exits - pruned, infeasible, leaves - are not counted)
    if (lane == 0) { atomicAdd(&B.prof[100], (unsigned long long)(te1 - te0)); atomicAdd(&B.prof[101], (unsigned long long)(te2 - te1)); atomicAdd(&B.prof[102], (unsigned long long)(te3 - te2));
                     atomicAdd(&B.prof[103], (unsigned long long)(te4 - te3)); atomicAdd(&B.prof[104], (unsigned long long)(te5 - te4)); atomicAdd(&B.prof[105], (unsigned long long)(te6 - te5)); atomicAdd(&B.prof[106], 1ull); } }
#endif
#undef FREE_NODE
}

// ------------------------------------------------------------------------------------------------
//  select: per instance prune + sort + pick
constexpr int SEL_THREADS = 1024;  // one workgroup per instance scans its open list (up to 2^20 entries)
constexpr int SEL_LOWSHIFT = 24;   // the radix select resolves the top 40 key bits (28 mantissa bits); lower bits are ties

// One workgroup per instance.  Reads the open list from buffer `B.open_sel`, prunes it against the incumbent,
// selects the `take` smallest keys by an MSB-first 8-bit radix select (no full sort, lists live in HBM/L2),
// emits them into the round's batch and writes the survivors to the other buffer.
// position of a lane's item in a list that the whole workgroup appends to: one atomic per wavefront
__device__ inline int wave_append(int* counter, bool pred) {
  const unsigned long long mk = __ballot(pred);
  const int lane = threadIdx.x & 63;
  int base = 0;
  if (lane == (int)(__ffsll((long long)mk) - 1)) base = atomicAdd(counter, __popcll(mk));
  base = __shfl(base, mk ? (int)(__ffsll((long long)mk) - 1) : 0);
  return base + __popcll(mk & ((1ull << lane) - 1ull));
}

__global__ void __launch_bounds__(SEL_THREADS) select_kernel(DevBuf B, int round) {
  const Layout& Y = B.Y;
  const int slot = blockIdx.x, tid = threadIdx.x;
#ifdef MIQP_PROFILE   // fold the last standard active-set launch's span and tail (as_onchip.hip) and reset its per-launch words
  if (slot == 0 && tid == 0) {
    const unsigned long long a = B.prof[120], b = B.prof[121], c = B.prof[122];
    if (b != 0 && a != ~0ull && a != 0) { B.prof[123] += b - a; B.prof[124] += b - c; B.prof[125] += 1; }
    B.prof[120] = ~0ull; B.prof[121] = 0; B.prof[122] = ~0ull;
  }
#endif
  const int inst = B.slot_inst[slot];
  if (inst < 0 || B.inst_done[inst] || B.inst_kill[inst]) { if (tid == 0) B.slot_demand[slot] = 0; }
  if (inst < 0) return;   // empty slot
  const int cap = B.open_cap;
  __shared__ int sh_take, sh_base, sh_m, sh_keep, sh_pick, sh_ties, sh_w, sh_mv, sh_all, sh_elig;
  __shared__ double sh_inc;
  __shared__ unsigned int hist[256], dhist[256], sh_wsum[4];
  __shared__ int sh_bin;
  __shared__ int sh_dkeep, sh_tiecnt, sh_ndef;
  __shared__ unsigned long long sh_thr2;
  __shared__ unsigned long long sh_prefix, sh_thr, sh_fmin;
  __shared__ double red[SEL_THREADS];
  if (B.inst_done[inst]) return;
  const size_t src = ((size_t)B.open_sel * B.n_slots + slot) * cap, dst = ((size_t)(1 - B.open_sel) * B.n_slots + slot) * cap;
  unsigned long long* keys = B.open_key + (size_t)slot * cap;
  PROF_T(ts0);
  // ---- incumbent bookkeeping: copy the solution of the atomicMin winner of the last round
  if (tid == 0) {
    unsigned long long key = B.inc_key[inst];
    sh_inc = inc_from_key(key);
    sh_take = -1;
    sh_thr = key != B.inc_seen[inst] ? key : ~0ull;   // a new incumbent key: its solution is looked up below
    sh_fmin = ~0ull;
    sh_m = 0; sh_keep = 0; sh_pick = 0; sh_ties = 0; sh_ndef = 0;
  }
  __syncthreads();
  if (sh_thr != ~0ull) {
    // the batch slot whose candidate carries the winning key (several with the same key: the same objective to 44 bits and the
    // same record hash - the smallest full objective, then the smallest slot)
    const unsigned long long key = sh_thr;
    const int pbc = B.prev_bc < B.batch_cap ? B.prev_bc : B.batch_cap;
    for (int k = tid; k < pbc; k += SEL_THREADS) if (B.batch_candkey[k] == key && B.batch_candinst[k] == inst) atomicMin(&sh_fmin, d2key(B.batch_obj[k]));   // (this instance's slots only: two instances of a queue can reach the same key)
    __syncthreads();
    const unsigned long long bo = sh_fmin;
    if (tid == 0) sh_dkeep = 0x7FFFFFFF;
    __syncthreads();
    if (bo != ~0ull) for (int k = tid; k < pbc; k += SEL_THREADS) if (B.batch_candkey[k] == key && B.batch_candinst[k] == inst && d2key(B.batch_obj[k]) == bo) atomicMin(&sh_dkeep, k);
    __syncthreads();
    if (tid == 0 && sh_dkeep != 0x7FFFFFFF) { sh_take = sh_dkeep; B.inc_seen[inst] = key; }
  }
  __syncthreads();
  if (sh_take >= 0) {
    const int bslot = sh_take;   // batch slot of the node that won the atomicMin
    const signed char* cf = B.batch_comp + (size_t)bslot * Y.fixlen; signed char* df = B.inc_fix + (size_t)inst * Y.fixlen;
    for (int k = tid; k < Y.fixlen; k += SEL_THREADS) df[k] = cf[k];
    const double* zs = B.batch_Z + (size_t)bslot * Y.N * Y.nz; double* zd = B.inc_Z + (size_t)inst * Y.N * Y.nz;
    for (int k = tid; k < Y.N * Y.nz; k += SEL_THREADS) zd[k] = zs[k];
    if (B.inc_A && B.batch_A && B.batch_Mtag) {   // the active set the incumbent was found with, for the leaves of the local search around it
      if (tid < 64) B.inc_A[(size_t)inst * 64 + tid] = B.batch_A[(size_t)bslot * 64 + tid];
      if (tid == 0) B.inc_Mtag[inst] = B.batch_Mtag[bslot];
    }
    if (tid == 0) { B.inc_obj[inst] = B.batch_obj[bslot]; if (B.inst_lns) B.inst_lns[inst] |= 1; }   // (bit 1, "skeleton roots tried", stays)
  }
  __syncthreads();
  if (B.inst_kill[inst]) {
    // retired by the host (time limit of the instance): its node records go back to the pool, the lists are emptied, the
    // instance reports what it has (the incumbent copied above, the bound of the last round) as unfinished and its slot is free for the next one of the queue
    int nn = B.open_count[inst]; if (nn > cap) nn = cap;
    for (int k = tid; k < nn; k += SEL_THREADS) { const int nd = B.open_node[src + k]; if (nd >= 0) { unsigned int q_ = atomicAdd(B.free_tail, 1u); B.free_q[q_ % (unsigned int)B.pool_cap] = nd; } }
    int fn = B.far_cap > 0 ? B.far_count[inst] : 0; if (fn > B.far_cap) fn = B.far_cap;
    for (int k = tid; k < fn; k += SEL_THREADS) { const int nd = B.far_node[(size_t)slot * (size_t)B.far_cap + k]; if (nd >= 0) { unsigned int q_ = atomicAdd(B.free_tail, 1u); B.free_q[q_ % (unsigned int)B.pool_cap] = nd; } }
    __syncthreads();
    if (tid == 0) {
      B.open_count[inst] = 0; if (B.far_cap > 0) B.far_count[inst] = 0;
      atomicOr(&B.inst_flags[inst], 2); B.inst_done[inst] = 1; atomicSub(B.active_insts, 1);
    }
    return;
  }
  PROF_T(ts1);
  const double inc = fmin(sh_inc < 1e300 ? B.inc_obj[inst] : 1e300, B.inc_ext[inst]);   // what prunes: the best incumbent known (own or, in a tree split, another rank's)
  const double cst = B.inst_const[inst];
  const double gap = B.inst_gap[inst];
  int n = B.open_count[inst]; if (n > cap) n = cap;
  auto prunable = [&](double b) { return inc < 1e300 && (inc - (b + cst)) <= gap * (1e-10 + fabs(inc)); };   // cannot improve the incumbent by more than the gap
  // MSB-first 8-bit radix select on the resolved key bits: leaves in sh_prefix the smallest T with count(key <= T) >= need
  // (sh_all = 1 instead when fewer than `need` keys exist) and returns how many keys equal to T are still needed
  auto radix = [&](auto keyf, int cnt, int need) -> int {
    if (tid == 0) { sh_prefix = 0ull; sh_all = 0; }
    __syncthreads();
    for (int shift = 56; shift >= SEL_LOWSHIFT; shift -= 8) {
      if (tid < 256) hist[tid] = 0u;
      __syncthreads();
      const unsigned long long prefix = sh_prefix;
      const unsigned long long himask = shift == 56 ? 0ull : (~0ull << (shift + 8));
      for (int k = tid; k < cnt; k += SEL_THREADS) {
        unsigned long long key = keyf(k);
        if (key != ~0ull && (key & himask) == prefix) atomicAdd(&hist[(key >> shift) & 255ull], 1u);
      }
      __syncthreads();
      // the first bin whose inclusive prefix count reaches `need`: a scan over the 256 bins by four wavefronts (one thread walking the bins paid an
      // LDS round trip per bin: 61 % of this kernel's cycles, measured)
      if (tid == 0) sh_bin = 256;
      unsigned int hv = 0, hx = 0;
      if (tid < 256) {
        hv = hist[tid]; hx = hv;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned int y = __shfl_up(hx, o); if ((tid & 63) >= o) hx += y; }
        if ((tid & 63) == 63) sh_wsum[tid >> 6] = hx;
      }
      __syncthreads();
      if (tid < 256) {
#pragma unroll
        for (int w = 0; w < 3; ++w) if (w < (tid >> 6)) hx += sh_wsum[w];   // inclusive prefix over all bins up to this one
        if (hx >= (unsigned int)need) atomicMin(&sh_bin, tid);
        if (tid == 255) sh_wsum[3] = hx;   // (the total, for the case that no bin reaches `need`)
      }
      __syncthreads();
      {
        const int bin0 = sh_bin;
        if (tid == (bin0 < 256 ? bin0 : 255)) {   // the owner of the chosen bin
          const int bin = bin0 < 256 ? bin0 : 255;
          const unsigned int cum = bin0 < 256 ? hx - hv : sh_wsum[3];   // keys in the bins before it (none reaches `need`: all of them, as the walk counted)
          if (bin0 >= 256 && shift == 56) sh_all = 1;
          sh_prefix = prefix | ((unsigned long long)bin << shift);
          sh_pick = need - (int)cum;   // how many of the keys with this prefix are still needed
          sh_tiecnt = (int)hv;         // (after the last pass: how many keys share the resolved bits of the threshold)
        }
      }
      __syncthreads();
      need = sh_pick;
      if (sh_all) break;
    }
    __syncthreads();
    return need;
  };
  const unsigned long long lowmask = (1ull << SEL_LOWSHIFT) - 1ull;
  // near list after a spill / refill, and the length that triggers a spill: a few batches' worth is enough (every pass of this
  // kernel is linear in it), refills are cheap
  int keep = cap / 4; { const int k2 = 2 * B.batch_cap > 65536 ? 2 * B.batch_cap : 65536; if (keep > k2) keep = k2; }
  const int spill_at = 2 * keep;
  const size_t fb = (size_t)slot * (size_t)B.far_cap;
  int fc = B.far_cap > 0 ? B.far_count[inst] : 0; if (fc > B.far_cap) fc = B.far_cap;
  // node selection: best bound, interleaved with dives (deepest first) while no incumbent exists, on every 4th
  // round afterwards and whenever the list is full (without a far tier: half full) or the record pool is nearly exhausted (a depth-first
  // frontier stays small); the order changes how fast incumbents appear, not what is proven
  const int dive_every = (B.seq_kinds >> 18) & 3;   // experiment switch: 0 every 4th round, 1 never, 2 every 8th, 3 every 2nd
  const bool periodic = dive_every == 0 ? (round & 3) == 3 : (dive_every == 1 ? false : (dive_every == 2 ? (round & 7) == 7 : (round & 1) == 1));
  bool pool_tight;
  { const int pc = *B.pool_count; const long long live = (long long)(pc < B.pool_cap ? pc : B.pool_cap) - (long long)(int)(*B.free_tail - *B.free_head);
    pool_tight = live > (long long)B.pool_cap / 10 * 9; }
  // (with a far tier a long near list is spilled, not dived; only when that tier fills up too does the search go depth first)
  const bool lists_full = B.far_cap > 0 ? fc > B.far_cap / 4 * 3 : n > cap / 2;
  const bool dive = !(inc < 1e300) || periodic || lists_full || pool_tight;
  auto order_key = [&](double b, int dp) -> unsigned long long {
    // best bound; experiment switch (bits 24..25 of seq_kinds): deeper nodes first among nearly equal bounds
    const int dbias = (B.seq_kinds >> 24) & 3;
    const double bb = dbias ? b - (dbias == 1 ? 1e-4 : (dbias == 2 ? 1e-3 : 1e-2)) * fabs(b) * (double)(dp >> 6) : b;
    // a dive: the depth word in the top 16 bits (deepest, most preferred sibling first), below it the top 48 bits of the bound's
    // key - NOT -dp * 1e9 + b in one double: at 1e12 the bound was resolved to 1e-4 only, and cousins with nearly equal bounds
    // tied on all 64 bits (the tie then went to whichever lane came first: the same solve took one of two paths from run to run)
    if (dive) { const unsigned long long kd = ((unsigned long long)(0xFFFF - (dp < 0xFFFF ? dp : 0xFFFF)) << 48) | (d2key(fmax(b, -1e300)) >> 16); return kd == ~0ull ? ~0ull - 1 : kd; }
    const double kv = fmax(bb, -1e300);
    const unsigned long long key = d2key(kv); return key == ~0ull ? ~0ull - 1 : key;
  };
  PROF_T(ts2);
  // ---- pass 1: prune, keys, lower bound, population per tree depth
  for (int k = tid; k < 256; k += SEL_THREADS) dhist[k] = 0u;
  __syncthreads();
  double lb = 1e300; int mloc = 0, ndef = 0;
  // Rounding probes are the large nodes (every region fixed: >= 480 general rows) that the on-chip kernel hands to the
  // memory-backed one, whose launch lasts as long as its slowest node however few there are.  With many instances in flight
  // they are only eligible every probe_every-th round: the same launch then takes the probes of several rounds.
  const bool big_round = B.probe_every <= 1 || (round % B.probe_every) == B.probe_every - 1;
  for (int k = tid; k < n; k += SEL_THREADS) {
    double b = B.open_bound[src + k]; int nd = B.open_node[src + k]; int dp = B.open_depth[src + k];
    unsigned long long key = ~0ull;
    if (nd < 0) { keys[k] = key; continue; }   // dead entry (a reservation that could not be filled)
    if (prunable(b)) {
      unsigned int q_ = atomicAdd(B.free_tail, 1u); B.free_q[q_ % (unsigned int)B.pool_cap] = nd;
    } else {
      lb = fmin(lb, b);
      key = order_key(b, dp);
      if (!big_round && is_probe_word(dp)) { key = ~0ull - 1; ndef++; }   // waits for its round behind every other key
      atomicAdd(&dhist[(dp >> 6) > 255 ? 255 : (dp >> 6)], 1u);
      mloc++;
    }
    keys[k] = key;
  }
  if (mloc) atomicAdd(&sh_m, mloc);
  if (ndef) atomicAdd(&sh_ndef, ndef);
  __syncthreads();
  int m = sh_m;
  PROF_T(ts3);
  // ---- far tier: when the near list cannot fill a batch any more, the far entries with the lowest bounds come back
  // (threshold by radix select on their bounds), the ones the incumbent prunes are dropped and the rest is compacted in
  // place, chunk by chunk (a chunk is read completely before anything is written at or below it)
  // (skipped while fewer than 64 entries are free behind the list: pass 3 compacts the list, the next round refills)
  if (fc > 0 && m < keep / 4 && cap - n >= 64) {   // (keep / 4 still fills the widest share of a batch: open_cap >= 8 x batch_cap, miqp_gpu.hip)
    auto fkey = [&](int k) -> unsigned long long {
      const double b = B.far_bound[fb + k];
      if (B.far_node[fb + k] < 0 || prunable(b)) return ~0ull;
      const unsigned long long key = d2key(b); return key == ~0ull ? ~0ull - 1 : key;
    };
    // a long tier is not radix-selected itself (five passes of one workgroup over millions of entries) but through a strided
    // sample of at most `keep` entries parked behind the near list's keys: the threshold only has to be about right
    // (the sample never leaves the key array: at most cap - n entries, a longer stride when the space behind the list is short)
    int stride = fc > 4 * keep ? (fc + keep - 1) / keep : 1;
    if (stride > 1 && (fc + stride - 1) / stride > cap - n) stride = (fc + (cap - n) - 1) / (cap - n);
    if (stride > 1) {
      const int ns = (fc + stride - 1) / stride;
      for (int j = tid; j < ns; j += SEL_THREADS) keys[n + j] = fkey(j * stride);
      __syncthreads();
      int need_s = (keep - m) / stride; if (need_s < 1) need_s = 1;
      radix([&](int k) { return keys[n + k]; }, ns, need_s);
    } else radix(fkey, fc, keep - m);
    const unsigned long long thr = sh_all ? ~0ull : sh_prefix;
    if (tid == 0) { sh_w = 0; sh_mv = 0; sh_fmin = ~0ull; }
    __syncthreads();
    int room = cap - n; if (room > keep - m) room = keep - m; if (room < 0) room = 0;
    constexpr int PER = 4;
    for (int c0 = 0; c0 < fc; c0 += SEL_THREADS * PER) {
      double eb[PER]; int en[PER], ed[PER], act[PER];   // -1 nothing, 0 drop, 1 stay, 2 move to the near list
#pragma unroll
      for (int j = 0; j < PER; ++j) {
        const int k = c0 + j * SEL_THREADS + tid;
        act[j] = -1;
        if (k < fc) {
          eb[j] = B.far_bound[fb + k]; en[j] = B.far_node[fb + k]; ed[j] = B.far_depth[fb + k];
          if (en[j] < 0) act[j] = -1;
          else if (prunable(eb[j])) act[j] = 0;
          else { unsigned long long key = d2key(eb[j]); if (key == ~0ull) key = ~0ull - 1; act[j] = ((key & ~lowmask) <= thr) ? 2 : 1; }
        }
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < PER; ++j) {
        if (act[j] == 0) { unsigned int q_ = atomicAdd(B.free_tail, 1u); B.free_q[q_ % (unsigned int)B.pool_cap] = en[j]; }
        int t = wave_append(&sh_mv, act[j] == 2);
        if (act[j] == 2) {
          if (t < room) {
            B.open_bound[src + n + t] = eb[j]; B.open_node[src + n + t] = en[j]; B.open_depth[src + n + t] = ed[j]; keys[n + t] = order_key(eb[j], ed[j]);
            lb = fmin(lb, eb[j]); atomicAdd(&dhist[(ed[j] >> 6) > 255 ? 255 : (ed[j] >> 6)], 1u);
          } else act[j] = 1;
        }
        int pos = wave_append(&sh_w, act[j] == 1);
        if (act[j] == 1) { B.far_bound[fb + pos] = eb[j]; B.far_node[fb + pos] = en[j]; B.far_depth[fb + pos] = ed[j]; atomicMin(&sh_fmin, d2key(eb[j])); }
      }
      __syncthreads();
    }
    const int moved = sh_mv < room ? sh_mv : room;
    n += moved; m += moved;
    fc = sh_w;
    __syncthreads();
    if (tid == 0) {
      B.far_count[inst] = fc; B.far_minkey[inst] = sh_fmin;
      B.near_thr[inst] = (fc > 0 && thr != ~0ull) ? key2d(thr | lowmask) : 1e300;   // what stays behind lies above the threshold
    }
  }
  PROF_T(ts4);
  lb = wave_min(lb);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = lb;
  __syncthreads();
#pragma unroll
  for (int w = 0; w < SEL_THREADS / 64; ++w) lb = fmin(lb, red[w]);
  if (fc > 0) lb = fmin(lb, key2d(B.far_minkey[inst]));   // the instance's bound covers both tiers
  // ---- spill: a best-bound round that finds the near list long keeps the `keep` lowest keys there and moves the rest to the
  // far tier (pass 3); from then on children above the threshold are appended to the far tier directly (eval_kernel).
  // The deepest levels (up to keep / 2 nodes) stay whatever their bound: they are the stack of the periodic dives.
  const bool spill = B.far_cap > 0 && m > spill_at && fc < B.far_cap;   // (in a dive round the keys are depths: the deepest nodes stay)
  unsigned long long thr_spill = ~0ull;
  if (spill) {
    radix([&](int k) { return keys[k]; }, n, keep);
    thr_spill = sh_all ? ~0ull : sh_prefix;
    if (tid == 0) { unsigned int cum = 0; int d = 255; for (; d >= 0; --d) { if (cum + dhist[d] > (unsigned int)(keep / 2)) break; cum += dhist[d]; } sh_dkeep = d + 1; }
    __syncthreads();
  }
  const int dkeep = sh_dkeep;
  // Endgame focus: when only a few instances are still open, the batch is not split evenly but geometrically by gap rank
  // (1/2 to the instance closest to its proof, 1/4 to the next, ...): finishing one instance after the other brings more
  // of them below the gap before the time limit than advancing all of them at the same pace.
  PROF_T(ts5);
  if (tid == 0) sh_pick = 0;   // reused as rank counter until the selection below resets it
  __syncthreads();
  const int act_now = *B.active_insts;
  const int focus_at = 16 << ((B.seq_kinds >> 29) & 3);   // experiment switch: 16 (default), 32, 64, ...
  const bool focus = act_now <= focus_at && inc < 1e300 && (B.seq_kinds & 0x8000000) == 0;
  if (focus) {
    const double gme = (inc - (lb + cst)) / (1e-10 + fabs(inc));
    int cntl = 0;
    for (int sj = tid; sj < B.n_slots; sj += SEL_THREADS) {
      const int j = B.slot_inst[sj];
      if (j < 0 || j == inst || B.inst_done[j]) continue;
      const double ij = B.inc_obj[j];
      const double gj = ij < 1e299 ? (ij - B.lower_bound[j]) / (1e-10 + fabs(ij)) : 1e300;
      if (gj < gme || (gj == gme && j < inst)) cntl++;
    }
    if (cntl) atomicAdd(&sh_pick, cntl);
  }
  __syncthreads();
  const int rank = sh_pick;
  __syncthreads();
  // Bound window of a best-bound round: only nodes in the lower part of [lower bound, incumbent - gap] are eligible.  The
  // proof has to process every node below (final incumbent) - gap anyway; nodes above that are only ever touched because the
  // incumbent of the moment is not the final one - a wide round reaches far up the list, a later round would find them pruned.
  // Deferring the upper part costs nothing but width, and the width goes to other instances (share_kernel).
  int m_elig = m - sh_ndef;   // (probes that wait for their round are not offered)
  if (!dive && inc < 1e300 && B.window_pct < 100 && m > 0) {
    const double top = inc - gap * (1e-10 + fabs(inc)) - cst;   // bounds are stored without the instance's constant
    const double thrw = lb + 0.01 * (double)B.window_pct * fmax(0.0, top - lb);
    const unsigned long long kw = d2key(thrw);
    if (tid == 0) sh_elig = 0;
    __syncthreads();
    int ce = 0;
    for (int k = tid; k < n; k += SEL_THREADS) { const unsigned long long key = keys[k]; if (key != ~0ull && key <= kw) ce++; }
    if (ce) atomicAdd(&sh_elig, ce);
    __syncthreads();
    m_elig = sh_elig < 1 ? 1 : sh_elig;
    if (m_elig > m - sh_ndef) m_elig = m - sh_ndef;
  }
  if (tid == 0) {
    sh_pick = 0;
    int act = act_now; if (act < 1) act = 1;
    int w;
    if (B.seq_kinds & 0x40000000) {   // (experiment switch, bit 30: the equal split and the endgame focus of round 2)
      w = (focus && !(B.seq_kinds & 0x10000000)) ? (B.batch_cap >> (rank + 1 < 30 ? rank + 1 : 30)) : B.batch_cap / act;
      if (w < B.nodes_per_round) w = B.nodes_per_round;
    } else { w = B.slot_take[slot]; if (w < B.base_take && !(B.young_nodes > 0 && B.inst_nodes[inst] >= (long long)B.young_nodes)) w = B.base_take; }   // (an older instance may be told to wait: share 0)
    if (B.width_cap > 0 && w > B.width_cap) w = B.width_cap;
    if (!(inc < 1e300) && w > 512) w = 512;   // no incumbent yet: a narrow dive (a wide one degenerates into breadth first)
    // a round that carries the neighbours of a new incumbent is kept narrow: the local search is a chain - a better neighbour, ITS neighbours the
    // round after - and what the tree would solve beside it in a wide round is mostly what the next incumbent of the chain prunes
    if (B.lns_narrow > 0 && B.inst_lns && (B.inst_lns[inst] & 1) && inc < 1e300 && B.inst_nodes[inst] >= (long long)B.lns_min_nodes && w > B.lns_narrow) w = B.lns_narrow;
    B.slot_demand[slot] = inc < 1e300 ? m_elig : (m < 512 ? m : 512);   // what this instance could use next round
    int take = m_elig < w ? m_elig : w;
    const int maxch = B.far_cap > 0 ? 8 : 64;     // children of one node: at most 63 (eval_kernel); with a far tier behind the list an overflow is absorbed there
    int room = (cap - m) / maxch; if (room < 1) room = 1;
    if (B.far_cap > 0 && room < 1024) room = 1024;   // (children that find the near list full go to the far tier)
    if (take > room) take = room;
    int base = take > 0 ? atomicAdd(B.batch_count, take) : 0;
    if (base + take > B.batch_cap) take = B.batch_cap > base ? B.batch_cap - base : 0;
    sh_take = take; sh_base = base;
    double lbt = ((n > 0 || fc > 0) && lb < 1e300 ? lb + cst : inc);
    if (inc < 1e300 && lbt > inc) lbt = inc;
    B.lower_bound[inst] = lbt;
    if (m == 0 && fc == 0) { B.inst_done[inst] = 1; atomicSub(B.active_insts, 1); }
    B.inst_mode[inst] = dive ? 1 : 0;
    sh_thr = ~0ull; sh_w = fc; sh_fmin = ~0ull;
  }
  __syncthreads();
  PROF_T(ts6);
  const int take = sh_take, base = sh_base;
  // ---- radix select: smallest key value T such that count(key <= T) >= take
  // Keys that agree in the resolved bits are ties.  Which of them are taken is decided by a second select on what the node
  // itself carries - the unresolved low bits of its key and its sibling preference - not by the order the lanes arrive in:
  // the list order is not reproducible (children are appended as their parents' evaluations finish), and a pick by arrival
  // would make the tree, and with it which solution within the gap is returned, differ from run to run.
  auto second_key = [&](unsigned long long key, int dp) -> unsigned long long { return (((key & lowmask) << 6) | (unsigned long long)(63 - (dp & 63))) << 34; };
  if (take > 0 && take < m) {
    int need = radix([&](int k) { return keys[k]; }, n, take);
    const unsigned long long thr1 = sh_prefix; const int tiecnt = sh_tiecnt;
    __syncthreads();
    unsigned long long thr2 = ~0ull;
    if (B.det_ties && need < tiecnt) {
      if (tiecnt <= SEL_THREADS) {
        // the usual case, a handful of siblings: the ties are collected (one pass over the keys) and ranked in LDS
        unsigned long long* tie = (unsigned long long*)red;
        if (tid == 0) sh_elig = 0;
        __syncthreads();
        for (int k = tid; k < n; k += SEL_THREADS) {
          const unsigned long long key = keys[k];
          if (key != ~0ull && (key & ~lowmask) == thr1) { const int t = atomicAdd(&sh_elig, 1); if (t < SEL_THREADS) tie[t] = second_key(key, B.open_depth[src + k]); }
        }
        __syncthreads();
        const int tc = sh_elig < SEL_THREADS ? sh_elig : SEL_THREADS;
        if (tid < tc) {
          const unsigned long long v = tie[tid]; int lt = 0, le = 0;
          for (int j = 0; j < tc; ++j) { const unsigned long long u = tie[j]; lt += u < v ? 1 : 0; le += u <= v ? 1 : 0; }
          if (lt < need && need <= le) { sh_thr2 = v; sh_ties = need - lt; }   // (equal values write the same)
        }
        __syncthreads();
        thr2 = sh_thr2; need = sh_ties;
        __syncthreads();
      } else {
        need = radix([&](int k) -> unsigned long long { const unsigned long long key = keys[k]; if (key == ~0ull || (key & ~lowmask) != thr1) return ~0ull; return second_key(key, B.open_depth[src + k]); }, n, need);
        thr2 = sh_prefix;
        __syncthreads();
      }
    }
    if (tid == 0) { sh_thr = thr1; sh_thr2 = thr2; sh_ties = need; sh_pick = 0; }   // threshold on the resolved (high) bits
    __syncthreads();
  } else if (tid == 0) { sh_thr = take >= m ? ~0ull : 0ull;   /* ~0: above every masked key */ sh_thr2 = ~0ull; sh_ties = 0x7FFFFFFF; sh_pick = 0; }
  __syncthreads();
  PROF_T(ts7);
  // ---- pass 3: emit the selected nodes, keep the rest (near list, or far tier when the round spills)
  const unsigned long long thr = sh_thr, thr2 = sh_thr2;
  for (int k0 = 0; k0 < n; k0 += SEL_THREADS) {
    const int k = k0 + tid;
    unsigned long long key = k < n ? keys[k] : ~0ull;
    const bool live = key != ~0ull;
    double b = 0.0; int nd = 0, dp = 0;
    bool pick = false;
    if (live) {
      b = B.open_bound[src + k]; nd = B.open_node[src + k]; dp = B.open_depth[src + k];
      if (take > 0) {
        const unsigned long long kh = key & (~0ull << SEL_LOWSHIFT);
        if (kh < thr) pick = true;
        else if (kh == thr) {
          const unsigned long long s2 = thr2 != ~0ull ? second_key(key, dp) : 0ull;
          if (thr2 != ~0ull && s2 < thr2) pick = true;
          else if (thr2 == ~0ull || s2 == thr2) { int t = atomicAdd(&sh_ties, -1); pick = t > 0; }
        }
      }
      if (pick) {
        int pos = atomicAdd(&sh_pick, 1);
        if (pos < take) { B.batch_node[base + pos] = nd; B.batch_inst[base + pos] = inst; B.batch_bound[base + pos] = b; B.batch_depth[base + pos] = dp;
                          if (B.batch_large) { const int lc_ = large_class(B, nd, dp, inc < 1e300); B.batch_large[base + pos] = (unsigned char)lc_;   // (the launch of the round that solves the node)
                            if (lc_ && B.cls_list) { const int q_ = atomicAdd(&B.cls_count[lc_ - 1], 1); if (q_ < B.batch_cap) B.cls_list[(size_t)(lc_ - 1) * B.batch_cap + q_] = base + pos; } } }
        else pick = false;
      }
    }
    bool tofar = live && !pick && spill && (key & ~lowmask) > thr_spill && (dp >> 6) < dkeep;
    int fpos = wave_append(&sh_w, tofar);
    if (tofar) {
      if (fpos < B.far_cap) { B.far_bound[fb + fpos] = b; B.far_node[fb + fpos] = nd; B.far_depth[fb + fpos] = dp; atomicMin(&sh_fmin, d2key(b)); }
      else tofar = false;
    }
    const bool stay = live && !pick && !tofar;
    int pos = wave_append(&sh_keep, stay);
    if (stay) { B.open_bound[dst + pos] = b; B.open_node[dst + pos] = nd; B.open_depth[dst + pos] = dp; }
  }
  __syncthreads();
  if (tid == 0) {
    B.open_count[inst] = sh_keep;
    if (spill) {
      B.far_count[inst] = sh_w < B.far_cap ? sh_w : B.far_cap;
      if (sh_fmin != ~0ull) atomicMin(&B.far_minkey[inst], sh_fmin);
      // after a spill by depth the near list is no longer "everything below a bound": children go to the far tier until the next refill sets the threshold again
      B.near_thr[inst] = dive ? -1e300 : (thr_spill != ~0ull ? key2d(thr_spill | lowmask) : 1e300);
    }
  }
#ifdef MIQP_PROFILE
  { PROF_T(ts8);
    if (tid == 0) { const long long tt[9] = {ts0, ts1, ts2, ts3, ts4, ts5, ts6, ts7, ts8};
      for (int q = 0; q < 8; ++q) atomicAdd(&B.prof[110 + q], (unsigned long long)(tt[q + 1] - tt[q]));
      atomicAdd(&B.prof[118], 1ull); atomicAdd(&B.prof[119], (unsigned long long)n); } }
#endif
}

// admission of queued instances into free slots (one thread per admission): the instance that held the slot leaves it, the
// new one's root records become the head of the slot's near list in the buffer the next select_kernel reads
__global__ void admit_kernel(DevBuf B, const int* pairs, int n, int sel) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int slot = pairs[2 * k], inst = pairs[2 * k + 1];
  const int old = B.slot_inst[slot];
  if (old >= 0) B.inst_slot[old] = -1;
  B.slot_inst[slot] = inst;
  if (inst < 0) return;   // the slot is only vacated
  B.inst_slot[inst] = slot;
  const int nr = B.root_cnt[inst];
  const size_t base = ((size_t)sel * B.n_slots + slot) * B.open_cap;
  for (int r = 0; r < nr; ++r) { B.open_bound[base + r] = -1e300; B.open_node[base + r] = B.root_node[(size_t)inst * B.root_stride + r]; B.open_depth[base + r] = B.root_depth[(size_t)inst * B.root_stride + r]; }
  B.open_count[inst] = nr;
  atomicAdd(B.active_insts, 1);
}

// shares of the next round's batch (one workgroup; see DevBuf::slot_take)
__global__ void __launch_bounds__(1024) share_kernel(DevBuf B) {
  __shared__ long long red[1024];
  __shared__ int sh_lo, sh_hi;
  const int tid = threadIdx.x, NSL = B.n_slots;
  const int WCAP = NSL == 1 ? (B.batch_cap > 8192 ? B.batch_cap : 8192) : 8192;   // most nodes one instance takes per round (a single solve: its widest round, see width_cap)
  auto bsum = [&](long long v) -> long long {   // (wavefront sums by shuffles, sixteen partials through LDS: two barriers instead of twelve per sum - this kernel is ~ 15 of them)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();   // (the partials of the previous sum have been read)
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    long long t = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += red[w];
    return t;
  };
  const long long target = (long long)B.batch_cap - (long long)B.batch_cap / 16;   // (demands are one round old: a little head room)
  // The base share is a FLOOR that adapts to the load: the largest F in [base_take, share_cap] with sum_k min(demand_k, F) <= floor_pct % of
  // the batch.  An instance that waits for its turn in the admission order must not crawl at a handful of nodes per round meanwhile - at 8 -
  // 16 nodes per round the dives reach their leaves late, the local search (it starts after lns_min_nodes nodes) later still, and the wide
  // rounds the instance gets afterwards wade through what a timely incumbent would have pruned: measured, two hard instances at the END of
  // the admission order of 1280 cost 0.89 M and 1.20 M nodes against 0.24 M and 0.30 M at its head (tools/crowd_probe.py).
  int base = B.base_take;
  if (B.floor_pct > 0) {
    if (tid == 0) { sh_lo = B.base_take; sh_hi = B.share_cap; }
    __syncthreads();
    const long long budget = target * B.floor_pct / 100;
    for (int itn = 0; itn < 12; ++itn) {
      const int lo = sh_lo, hi = sh_hi;
      if (lo >= hi) break;
      const int mid = lo + (hi - lo + 1) / 2;
      long long a = 0;
      for (int k = tid; k < NSL; k += 1024) { const int d = B.slot_demand[k]; a += d < mid ? d : mid; }
      const long long tot = bsum(a);
      if (tid == 0) { if (tot <= budget) sh_lo = mid; else sh_hi = mid - 1; }
      __syncthreads();
    }
    base = sh_lo;
    __syncthreads();
  }
  // ... and with young_nodes > 0 only instances that have cost fewer node relaxations than that have a base share at all (the easy ones finish on
  // it within a few rounds); an older instance gets its share in the pass by admission order or WAITS - it keeps its tree as it is instead of
  // growing it a handful of nodes at a time
  auto base_of = [&](int k) -> int {
    const int d = B.slot_demand[k], inst = B.slot_inst[k];
    if (B.young_nodes > 0 && inst >= 0 && B.inst_nodes[inst] >= (long long)B.young_nodes) return 0;
    return d < base ? d : base;
  };
  long long ab = 0;
  for (int k = tid; k < NSL; k += 1024) ab += base_of(k);
  const long long SB = bsum(ab);
  const long long rest = target - SB;
  const long long cap = B.share_cap;
  auto more_of = [&](int k) -> long long {   // what slot k may take beyond its base share in the pass by admission order
    const int d = B.slot_demand[k]; const long long c = d < cap ? d : cap;
    const int b = base_of(k);
    return c > b ? c - b : 0;
  };
  // pass by admission order (earliest deadline first), every instance up to the cap
  // (every slot sums over all slots: the instance of a slot and what it may take are staged in LDS once - read from global memory inside the
  // double loop this pass was 0.1 ms of every round at 1280 slots)
  constexpr int SH_SLOTS = 2048;
  __shared__ int s_inst[SH_SLOTS]; __shared__ long long s_more[SH_SLOTS];
  const bool staged = NSL <= SH_SLOTS;
  if (staged) { for (int k = tid; k < NSL; k += 1024) { s_inst[k] = B.slot_inst[k]; s_more[k] = more_of(k); } }
  __syncthreads();
  long long at = 0;
  for (int k = tid; k < NSL; k += 1024) {
    const int inst = staged ? s_inst[k] : B.slot_inst[k];
    int take = base_of(k);
    const long long more = staged ? s_more[k] : more_of(k);
    if (inst >= 0 && more > 0 && rest > 0) {
      long long before = 0;   // what the instances admitted earlier take from the rest
      if (staged) { for (int j = 0; j < NSL; ++j) { const int ij = s_inst[j]; if (ij >= 0 && ij < inst) before += s_more[j]; } }
      else { for (int j = 0; j < NSL; ++j) { const int ij = B.slot_inst[j]; if (ij >= 0 && ij < inst) before += more_of(j); } }
      long long ex = rest - before; if (ex > more) ex = more;
      if (ex > 0) take += (int)ex;
    }
    B.slot_take[k] = take; at += take;
  }
  // capacity still free (few hard instances in flight): the same extra allowance for every instance that can use it, so that the
  // batch is full - an idle device is worse than the nodes a wider round wastes - up to WCAP nodes per instance
  const long long left = target - bsum(at);
  if (left <= 0) return;
  if (tid == 0) { sh_lo = 0; sh_hi = WCAP; }
  __syncthreads();
  for (int itn = 0; itn < 16; ++itn) {
    const int lo = sh_lo, hi = sh_hi;
    if (lo >= hi) break;
    const int mid = lo + (hi - lo + 1) / 2;
    long long a = 0;
    for (int k = tid; k < NSL; k += 1024) { long long r = (long long)B.slot_demand[k] - B.slot_take[k]; if (r > WCAP - B.slot_take[k]) r = WCAP - B.slot_take[k]; a += r < mid ? (r > 0 ? r : 0) : mid; }
    const long long tot = bsum(a);
    if (tid == 0) { if (tot <= left) sh_lo = mid; else sh_hi = mid - 1; }
    __syncthreads();
  }
  const int L = sh_lo;
  for (int k = tid; k < NSL; k += 1024) {
    long long r = (long long)B.slot_demand[k] - B.slot_take[k]; if (r > WCAP - B.slot_take[k]) r = WCAP - B.slot_take[k];
    if (r > 0) B.slot_take[k] += (int)(r < L ? r : L);
  }
}

// Local search around a new incumbent - the device's answer to CPLEX's polishing heuristics (rinsheur, cplexmodel/cplexmodel.mod:8-21).
// What separates the incumbents a search finds early from the optimum of the hard instances of this workload is rarely the
// skeleton (who merges behind whom) but WHEN each car changes its region: the acceleration a car may use depends on the sector its
// velocity lies in, and a change one step earlier or later is worth several per cent of the objective (measured: tools/skeleton.py).
// The tree finds those one at a time, 10^5 nodes apart.  Here every new incumbent spawns its neighbours - each change of region of
// a car moved one or two steps earlier or later, a short stay in a region removed - as LEAVES: the incumbent's completed record
// with the changed region codes, every other disjunction as the incumbent has it.  They join the batch of the same round; a
// feasible one that is better becomes the incumbent through the ordinary evaluation (which checks every disjunction at the leaf's
// own solution) and spawns its own neighbours next round.  Heuristic nodes duplicate parts of the tree, they never replace it.
constexpr int LNS_MAX = 512;
__global__ void __launch_bounds__(64) lns_kernel(DevBuf B) {
  const Layout& Y = B.Y;
  const int slot = blockIdx.x, lane = threadIdx.x;
  const int inst = B.slot_inst[slot];
  if (inst < 0) return;
  const int flags = B.inst_lns[inst];   // bit 0: a new incumbent waits for its neighbours; bit 1: the skeleton roots of the instance have been tried
  const bool want_skel = (B.lns_mode & 32) && !(flags & 2) && Y.NP > 0;
  if (!(flags & 1) && !want_skel) return;
  if (B.inst_nodes[inst] < (long long)B.lns_min_nodes) return;   // (the flags stay: the heuristics start when the instance proves hard)
  if ((flags & 1) && !want_skel && B.lns_step > 0.0) {   // (the flag stays: a later, larger improvement runs the search)
    const double io = B.inc_obj[inst] + B.inst_const[inst], last = B.inst_lns_obj[inst];
    if (last < 1e299 && io > last - B.lns_step * fabs(last)) return;
  }
  __shared__ int nb_c[LNS_MAX], nb_i[LNS_MAX], nb_n[LNS_MAX], nb_code[LNS_MAX];   // neighbour: first byte of the record, stride, number of entries, the value they take
  __shared__ int nb_c2[LNS_MAX], nb_n2[LNS_MAX], nb_code2[LNS_MAX];               // ... and a second stretch of the same stride (0 entries: none)
  __shared__ int nb_rec[LNS_MAX];                                                  // the node record of every neighbour
  __shared__ int sh_n, sh_base, sh_rec, sh_skel;
  const int N = Y.N, C = Y.C;
  const signed char* inc = B.inc_fix + (size_t)inst * Y.fixlen;
  if (lane == 0) {
    int n = 0;
    auto push = [&](int first, int stride, int cnt, int val) { if (n < LNS_MAX) { nb_c[n] = first; nb_i[n] = stride; nb_n[n] = cnt; nb_code[n] = val; nb_n2[n] = 0; n++; } };
    const bool alive = !B.inst_done[inst] && !B.inst_kill[inst];
    sh_skel = 0;
    if (alive && want_skel) {
      // Skeleton roots (once per hard instance): who drives beside, behind or ahead of whom is decided by the rear / rear car/car
      // disjunction over the horizon, and the first dive of the tree commits to ONE such skeleton - the others are met again 10^5
      // nodes later in best-bound order (measured: tools/skeleton.py).  For every pair of cars every sequence
      // "alternative a throughout" and "a1 before step k, a2 from k on" (k = N/4, N/2, 3N/4) that the reachability presolve allows
      // becomes a root of the MIP-start repair kind: its relaxation is solved, its completion rounded (one probe child), and a
      // feasible rounding is an incumbent whose neighbourhood the local search then walks.  Never part of the tree.
      sh_skel = 1;
      const int* T = B.inst_i + (size_t)inst * Y.istride;
      for (int p = 0; p < Y.NP; ++p) {
        const int first = Y.f_c2c + p * N * 4;   // group 0 (rear / rear) of step i at first + 4 i
        auto allowed = [&](int a, int i0, int i1) { for (int i = i0; i < i1; ++i) if (!((T[Y.i_c2callow + p * N + i] >> a) & 1)) return false; return true; };
        for (int a = 0; a < 4; ++a) if (allowed(a, 1, N)) push(first + 4, 4, N - 1, a);
        for (int kq = 1; kq <= 3; ++kq) {
          const int k = (kq * N) / 4;
          if (k < 2 || k >= N - 1) continue;
          for (int a1 = 0; a1 < 4; ++a1) for (int a2 = 0; a2 < 4; ++a2) {
            if (a1 == a2 || !allowed(a1, 1, k) || !allowed(a2, k, N) || n >= LNS_MAX) continue;
            push(first + 4, 4, k - 1, a1); nb_c2[n - 1] = first + 4 * k; nb_n2[n - 1] = N - k; nb_code2[n - 1] = a2;
          }
        }
      }
      // three and four cars: also every ORDER of the cars along the road over the last quarter of the horizon (C! roots: the rear / rear
      // alternative "the one behind the other in x" of every pair, wherever the presolve allows it) - the pairwise sequences above fix one
      // pair at a time, an overtake that reorders three cars is not among them
      if (C >= 3 && (B.lns_mode & 64)) {
        int nperm = 1; for (int c = 2; c <= C; ++c) nperm *= c;
        for (int pi = 0; pi < nperm && n < LNS_MAX; ++pi) { nb_c[n] = -1; nb_i[n] = 4; nb_n[n] = 0; nb_code[n] = pi; nb_n2[n] = 0; n++; }
      }
      B.inst_lns[inst] = flags | 2;   // (a pending incumbent keeps its bit: its neighbours follow next round)
    } else if (alive) {
      B.inst_lns[inst] = flags & ~1; B.inst_lns_obj[inst] = B.inc_obj[inst] + B.inst_const[inst];
      // a sequence over the steps (the region codes of a car: stride 1; the alternatives of a car/car group: stride 4): every change
      // between two steps moved one / two steps later and earlier, every short run between two changes given to its neighbours
      auto moves = [&](int first, int stride, int mode) {
        const signed char* q = inc + first;
        auto at = [&](int i) { return (int)q[i * stride]; };
        int run0 = 1;
        for (int i = 2; i <= N; ++i) {
          const bool brk = i == N || at(i) != at(i - 1);
          if (!brk) continue;
          if (i < N && at(i) >= 0 && at(i - 1) >= 0) {
            push(first + i * stride, stride, 1, at(i - 1));
            if ((mode & 2) && i + 1 < N) push(first + i * stride, stride, 2, at(i - 1));
            push(first + (i - 1) * stride, stride, 1, at(i));
            if ((mode & 2) && i - 2 >= 1) push(first + (i - 2) * stride, stride, 2, at(i));
          }
          const int len = i - run0;   // the run [run0, i)
          if ((mode & 4) && run0 > 1 && i < N && len >= 3 && len <= 6 && at(run0 - 1) >= 0 && at(i) >= 0) {
            push(first + run0 * stride, stride, len, at(run0 - 1));
            if (at(i) != at(run0 - 1)) push(first + run0 * stride, stride, len, at(i));
          }
          run0 = i;
        }
      };
      const int mode = B.lns_mode;
      for (int c = 0; c < C; ++c) moves(Y.f_reg + c * N, 1, mode | 2);
      if (mode & 8) for (int p = 0; p < Y.NP; ++p) for (int g = 0; g < 4; ++g) moves(Y.f_c2c + p * N * 4 + g, 4, mode & ~2);
      if (mode & 16) {   // one (car, step) moved to a neighbouring sector (same kind of alternative): the excursions the moves above cannot make
        const int* T = B.inst_i + (size_t)inst * Y.istride;
        for (int c = 0; c < C; ++c) {
          const int np = T[Y.i_nposs + c];
          for (int i = 1; i < N; ++i) {
            const int code = (int)inc[Y.f_reg + c * N + i];
            if (code < 0 || (code & 3) == 3) continue;
            const int j = T[Y.i_regj + c * Y.P + (code >> 2)];
            for (int q = 0; q < np && q < Y.P; ++q) {
              const int dj = (T[Y.i_regj + c * Y.P + q] - j + Y.R) % Y.R;
              if (dj != 1 && dj != Y.R - 1) continue;
              if ((i > 1 && (int)inc[Y.f_reg + c * N + i - 1] >> 2 == q) || (i + 1 < N && (int)inc[Y.f_reg + c * N + i + 1] >> 2 == q)) continue;   // (a shift: made above)
              const int h = (code & 3) < T[Y.i_nhs + c * Y.P + q] ? (code & 3) : 0;
              push(Y.f_reg + c * N + i, 1, 1, (q << 2) | h);
            }
          }
        }
      }
    } else B.inst_lns[inst] = flags & ~1;
    int base = 0, rec = 0;
    if (n > 0) {
      // node records FIRST: recycled ones (as eval_kernel takes them), fresh ones otherwise; the evaluation of the round frees them again.  Batch
      // slots are reserved only for records that exist - a reserved slot that is never written would keep the node of the round before, and the
      // interior point and evaluation kernels would process a record that has been freed or reused since (advisor finding, round 4)
      const unsigned int h = atomicAdd(B.free_head, (unsigned int)n);
      if ((int)(*B.free_limit - h) >= n) { for (int q = 0; q < n; ++q) nb_rec[q] = B.free_q[(h + q) % (unsigned int)B.pool_cap]; }
      else {
        rec = atomicAdd(B.pool_count, n);
        if (rec + n > B.pool_cap) { atomicSub(B.pool_count, n); n = 0; }   // the pool is exhausted: no neighbours this time (heuristic nodes - the tree is not affected), the count is put back
        for (int q = 0; q < n; ++q) nb_rec[q] = rec + q;
      }
      if (n > 0) {
        base = atomicAdd(B.batch_count, n);
        const int fit = base + n > B.batch_cap ? (base < B.batch_cap ? B.batch_cap - base : 0) : n;
        for (int q = fit; q < n; ++q) { const unsigned int t_ = atomicAdd(B.free_tail, 1u); B.free_q[t_ % (unsigned int)B.pool_cap] = nb_rec[q]; }   // records without a batch slot go back to the pool
        n = fit;
      }
    }
    sh_n = n; sh_base = base; sh_rec = rec;
  }
  __syncthreads();
  const int n = sh_n;
  if (n <= 0) return;
  const double lbq = B.lower_bound[inst] - B.inst_const[inst];
  const bool skel = sh_skel != 0;
  for (int q = 0; q < n; ++q) {
    const int rec = nb_rec[q], bs = sh_base + q;
    signed char* dst = B.pool_fix + (size_t)rec * Y.fixlen;
    const int k0 = nb_c[q], st_ = nb_i[q], k1 = k0 + nb_n[q] * st_;
    if (skel) {   // a root record (everything undecided) with the sequence of one pair's rear / rear group, or with an order of all cars
      const int k2 = nb_c2[q], k3 = k2 + nb_n2[q] * st_;
      int rank[4] = {0, 0, 0, 0};
      if (k0 < 0) {   // the nb_code-th permutation of the cars (factorial number system): rank[c] = place of car c, 0 = rearmost
        int code = nb_code[q], used = 0;
        for (int place = 0; place < C; ++place) {
          const int f = C - place; int pick = code % f; code /= f;
          for (int c = 0; c < C; ++c) if (!((used >> c) & 1)) { if (pick == 0) { rank[c] = place; used |= 1 << c; break; } pick--; }
        }
      }
      const int* Tq = B.inst_i + (size_t)inst * Y.istride;
      for (int k = lane; k < Y.fixlen; k += 64) {
        signed char v = (signed char)-1;
        if (k0 >= 0) {
          if (k >= k0 && k < k1 && (k - k0) % st_ == 0) v = (signed char)nb_code[q];
          if (nb_n2[q] > 0 && k >= k2 && k < k3 && (k - k2) % st_ == 0) v = (signed char)nb_code2[q];
        } else if (k >= Y.f_c2c && k < Y.f_c2c + Y.NP * N * 4 && ((k - Y.f_c2c) & 3) == 0) {
          const int e = (k - Y.f_c2c) >> 2, pp = e / N, i = e - pp * N;
          if (i >= N - N / 4) {
            int c1, c2; pair_cars(pp, C, c1, c2);
            const int alt = rank[c1] < rank[c2] ? 0 : 1;   // alternative 0: c1 behind c2 in x, 1: c2 behind c1 (decode_row)
            if ((Tq[Y.i_c2callow + pp * N + i] >> alt) & 1) v = (signed char)alt;
          }
        }
        dst[k] = v;
      }
      if (lane == 0) {
        if (B.pool_big) B.pool_big[rec] = 0;
        if (B.pool_origin) B.pool_origin[rec] = 13;
        B.batch_node[bs] = rec; B.batch_inst[bs] = inst; B.batch_bound[bs] = lbq; B.batch_depth[bs] = REPAIR_ROOT;
        if (B.batch_large) B.batch_large[bs] = 0;
      }
      continue;
    }
    for (int k = lane; k < Y.fixlen; k += 64) {
      signed char v = inc[k];
      if (k >= k0 && k < k1 && (k - k0) % st_ == 0) v = (signed char)nb_code[q];
      if (k >= Y.f_env && k < Y.f_c2c && (k - Y.f_env) % 5 != 0) v = (signed char)-1;   // front-point environment / obstacle disjunctions: undecided (their rows are most of a leaf's rows; the evaluation checks them at the leaf's solution)
      if (k >= Y.f_c2n && k < Y.f_rmask) v = (signed char)-1;          // no exclusion rows
      if (k >= Y.f_rmask && k < Y.f_rmask + C * N * 2) v = (signed char)-1;   // every region allowed (0xFF 0xFF)
      dst[k] = v;
    }
    if (B.pool_Z && rec < B.z_cap) {
      const double* zs = B.inc_Z + (size_t)inst * N * Y.nz; double* zd = B.pool_Z + (size_t)rec * N * Y.nz;
      for (int k = lane; k < N * Y.nz; k += 64) zd[k] = zs[k];
    }
    if (B.pool_A && rec < B.z_cap) {
      // The leaf starts from the active set and the M of the solve that found the incumbent (the active-set launches; a recycled record has no parent):
      // it differs from the incumbent in the stages of the moved stretch - the rows of those stages are handed on as rows the leaf no longer has (an
      // identity no decode produces: the start removes them from M), the rest by identity; the start checks that M fits them.  Without it a leaf is a
      // cold solve of |A| + 4 steps, and the leaves of a round - 5 % of its nodes - held a third of the device for two thirds of the round
      unsigned short av = (unsigned short)0xFFFFu; unsigned long long tg = 0ull;
      if (B.inc_A && B.ring_M && B.lns_warm) {
        tg = B.inc_Mtag[inst];
        if (tg != 0ull) {
          unsigned int chg = 0u;
          for (int k = k0; k < k1; k += st_) { const int sg = fix_stage(Y, k); if (sg >= 0 && sg < 32) chg |= 1u << sg; }
          av = B.inc_A[(size_t)inst * 64 + lane];
          if (av != 0xFFFFu) { const int sg = av < 1024u ? (int)(av >> 5) : ((int)av - 1024) / Y.NSLOT; if (sg < 32 && ((chg >> sg) & 1u)) av = (unsigned short)0xFFFEu; }
        }
      }
      B.pool_A[(size_t)rec * 64 + lane] = av; if (B.ring_M && lane == 0) B.pool_Mtag[rec] = tg;
    }
    if (lane == 0) {
      if (B.pool_big) B.pool_big[rec] = 1;
      if (B.pool_origin) B.pool_origin[rec] = 14;
      B.batch_node[bs] = rec; B.batch_inst[bs] = inst; B.batch_bound[bs] = lbq; B.batch_depth[bs] = (1 << 6) | 63;   // (the probe mark: a heuristic node - one that has not converged after probe_itcap iterations is abandoned)
      if (B.batch_large) { const int lc_ = large_class(B, rec, (1 << 6) | 63, true); B.batch_large[bs] = (unsigned char)lc_;
        if (lc_ && B.cls_list) { const int q_ = atomicAdd(&B.cls_count[lc_ - 1], 1); if (q_ < B.batch_cap) B.cls_list[(size_t)(lc_ - 1) * B.batch_cap + q_] = bs; } }
    }
  }
}

// makes the records freed so far available to the next eval launch
// ... and zeroes the counter set of the NEXT round (`zero8`: eight ints, null = the host uses memsets)
__global__ void roll_kernel(DevBuf B, int* zero8) {
  if (zero8 && threadIdx.x < CTR_SET) zero8[threadIdx.x] = 0;
  if (threadIdx.x != 0) return;
  unsigned int t = *B.free_tail, h = *B.free_head, l = *B.free_limit;
  if ((int)(h - l) > 0) h = l;   // pops that overshot the limit took fresh records instead
  *B.free_head = h; *B.free_limit = t;
}

}  // namespace miqp
