// ipm_onchip.hip - interior point kernel with the whole per-node working set on chip (included by kernels.hip).
//
// Same mathematics as ipm_kernel (stage-banded primal-dual interior point, Riccati recursion in MFMA registers, one
// 64-lane wavefront per B&B node); what changes is where the data lives and how the rows are walked:
//   * box rows (one coefficient +-1: velocity / acceleration / jerk bounds, slow square, half-planes, axis-aligned
//     polygon edges on the rear point - the bulk of a node's rows) are KEYED by (stage, column, side): lane (g, c) of the
//     wavefront owns column c, side g & 1 and the stages of parity g >> 1, slot k <-> stage 2k + parity.  Their state
//     (s, lambda, t) and right-hand side stay in registers for the whole solve (no index, no coefficient, no memory
//     traffic); several rows on one key collapse to the tightest right-hand side (the same feasible set).
//   * general rows (2..6 coefficients: sector, curvature, front-point and car/car rows) are compacted per stage into LDS
//     (a 16-nibble column map, packed coefficients, right-hand side); their state lives in the registers of lane
//     row % 64.  Once per iteration every row computes its scaling in one pass with all lanes busy; the stage loop only
//     gathers MFMA operands through the column map.
//   * the contributions of the box rows to the stage Hessians / gradients are computed for all stages in one pass and
//     parked in LDS (the region later reused for the step), so the Riccati loop has no row loop at all.
//   * only the Riccati gains leave the CU (10 KB per node, L2 resident) for the forward sweep.
//   * two cars: the columns of the stage vector are ordered chain-major (oc_pcol), which turns P [A B] into DPP column
//     shifts and [A B]' T into sums over a lane's registers; the input block is eliminated by Gauss-Jordan on the four
//     input rows of the tile (row_newbcast / permlane swaps), one MFMA forms the Schur complement and the next vector.
// Nodes with more general rows than the on-chip capacity are queued for ipm_kernel (the general, memory-backed kernel).
#ifndef MIQP_PHI_BRANCH
#define MIQP_PHI_BRANCH 1   // 1: the second group of four general rows of a stage only when the stage has more than four (a branch in the sweep)
#endif
namespace miqp {

constexpr int OC_GCAP = 128;      // general rows kept on chip (2 register slots per lane)
constexpr int OC_GSLOTS = OC_GCAP / 64;
constexpr int OC_GCOEF = 432;     // their packed coefficients (with N = 20 and a 480-byte fix record the block stays within 160 KB / 8: 2 wavefronts per SIMD)
constexpr int OC_SCR = 32;        // rows decoded per round through the dense scratch rows
constexpr int OC_SSTR = 17;       // stride of a scratch row (conflict free)
constexpr int OC_KL0 = 6;         // stages whose gains stay in LDS
#ifndef MIQP_OC_PF
#define MIQP_OC_PF 4
#endif
constexpr int OC_PF = MIQP_OC_PF;          // prefetch distance (stages) of the gains that come back from L2
#ifndef MIQP_OC_GRP
#define MIQP_OC_GRP 5   // (1: +9 %, 2: the value of rounds 2-3, 4 / 5 / 10: -2.1 / -3.0 / -0.9 % per pass of the replayed batch, profiles/r04_kernel_ab_replay.txt)
#endif
constexpr int OC_GRP = MIQP_OC_GRP;         // box slots whose chains are interleaved in the row passes
constexpr int OC_NSL = 10;        // box-row slots per lane: horizons of up to 2 * OC_NSL steps

struct OcLds { int z, u, r, gmeta, gcoef, grhs, wd, sstart, cand, fix, total; };   // byte offsets
// capacity of the larger variant of the kernel (the nodes the standard one hands on: rounding probes and the other nodes with up
// to OC_GCAP_BIG general rows): 5 register slots per lane, one wavefront per SIMD, 4 blocks of ~34 KB per CU
constexpr int OC_GCAP_BIG = 320;
__host__ __device__ constexpr int oc_gcoef_of(int gcap) { return gcap == 128 ? 432 : gcap * 7 / 2; }
// (as_std: the block of the standard active-set launch - its second region holds one stage vector and the decode scratch, its third the box keys
// alone: 768 B less at N = 20, so that eight blocks leave 6 KB of a CU's 160 KB free instead of none - with none, the holes the larger launches'
// 34 KB blocks leave behind kept every CU at six or seven blocks, tools/wave_dump.py)
__host__ __device__ inline OcLds oc_lds_layout(int N, int fixlen, int OC_GCAP = miqp::OC_GCAP, bool as_std = false) {
  const int OC_GCOEF = oc_gcoef_of(OC_GCAP);
  OcLds L; int o = 0;
  L.z = o; o += N * 16 * 8;
  L.u = o; { int a = as_std ? N * 16 * 8 : N * 32 * 8, b = OC_SCR * OC_SSTR * 8; o += a > b ? a : b; }    // D | Gd  /  dZ | gains  /  decode scratch
  L.r = o; { int a = N * 32 * 8, b = as_std ? 0 : OC_GCAP * 16 + OC_KL0 * 64 * 8; o += a > b ? a : b; }   // box right-hand side keys (decode) / (sqrt(w), f) of the general rows + gains of the first stages
  L.gmeta = o; o += OC_GCAP * 16;
  L.gcoef = o; o += OC_GCOEF * 8;
  L.grhs = o; o += OC_GCAP * 8;
  L.wd = o; o += 16 * 8;
  L.sstart = o; o += ((N + 2) * 4 + 7) & ~7;
  L.cand = o; o += (OC_GCAP + 64) * 2;
  L.fix = o; o += (fixlen + 15) & ~15;
  L.total = (o + 15) & ~15;
  return L;
}

#ifdef MIQP_PROFILE
#define OCP_T(var) const long long var = clock64()
#define OCP_ACC(k, t0, t1) ocp_[k] += (unsigned long long)((t1) - (t0))
#elif defined(MIQP_ISA_MARKS)   // tools/isa_report.py: phase boundaries as comments in the assembly
#define OCP_T(var) asm volatile("; OCMARK " #var)
#define OCP_ACC(k, t0, t1)
#else
#define OCP_T(var)
#define OCP_ACC(k, t0, t1)
#endif
__host__ __device__ inline int oc_gain_doubles(int N) { return N * 64; }
#define OC_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

// box row of (stage, slot), if that slot carries a row with a single coefficient +-1: column, sign, right-hand side
// (the cases of decode_row whose row touches one stage variable)
template <int C>
__device__ inline bool box_of_slot(const Layout& Y, const double* D, const int* T, const signed char* fix, int i, int slot, int& col, double& sgn, double& rhs) {
  if (slot >= C * Y.SC) return false;
  const int N = Y.N;
  const double* G = D + Y.d_glob;
  const int c = slot / Y.SC, rr = slot - c * Y.SC;
  const int code = i >= 1 ? (int)fix[Y.f_reg + c * N + i] : -1;
  const double* rt = code >= 0 ? D + Y.d_reg + (c * Y.P + (code >> 2)) * REGSZ : nullptr;
  if (rr < 7) {
    const double* Hc = (rt || rr < 3) ? D : region_hull(Y, D, T, fix, c, i);   // (only read when the region is undecided)
    switch (rr) {
      case 0: col = 6 * c + 1; sgn = -1; rhs = -G[0]; break;
      case 1: col = 6 * c + 4; sgn = -1; rhs = -G[0]; break;
      case 2: col = 6 * c + 1; sgn = 1; rhs = G[1]; break;
      case 3: col = 6 * c + 2; sgn = 1; rhs = rt ? rt[12] : Hc[1]; break;
      case 4: col = 6 * c + 2; sgn = -1; rhs = -(rt ? rt[11] : Hc[0]); break;
      case 5: col = 6 * c + 5; sgn = 1; rhs = rt ? rt[14] : Hc[3]; break;
      default: col = 6 * c + 5; sgn = -1; rhs = -(rt ? rt[13] : Hc[2]); break;
    }
    return true;
  }
  if (rr < 11) {
    const int s = (rr - 7) >> 1; const bool up = ((rr - 7) & 1) == 0;
    double lo, hi;
    if (i == 0) { lo = D[Y.d_u0box + c * 4 + 2 * s]; hi = D[Y.d_u0box + c * 4 + 2 * s + 1]; }
    else if (rt) { lo = rt[15 + 2 * s]; hi = rt[16 + 2 * s]; }
    else { const double* Hc = region_hull(Y, D, T, fix, c, i); lo = Hc[4 + 2 * s]; hi = Hc[5 + 2 * s]; }
    col = 6 * C + 2 * c + s; sgn = up ? 1.0 : -1.0; rhs = up ? hi : -lo;
    return true;
  }
  if (rr < 16) {
    if (code < 0) return false;   // the hull rows of an undecided region are general rows
    const int h = code & 3, k = rr - 11;
    if (h == 3) { col = 6 * c + (k < 2 ? 1 : 4); sgn = (k & 1) ? -1.0 : 1.0; rhs = G[6]; return true; }
    if (k == 2) {
      const int* hs = T + Y.i_hs + ((c * Y.P + (code >> 2)) * 2 + h) * 2;
      col = 6 * c + (hs[0] == 0 ? 1 : 4); sgn = -(double)hs[1]; rhs = -G[6];
      return true;
    }
    return false;
  }
  int q = rr - 16;
  const double* ed;
  if (q < 5 * Y.EL) {
    const int pt = q / Y.EL, k = q - pt * Y.EL;
    if (pt != 0) return false;
    const int e = Y.E == 1 ? 0 : (int)fix[Y.f_env + (c * N + i) * 5];
    ed = D + Y.d_env + (e * Y.EL + k) * 3;
  } else {
    q -= 5 * Y.EL;
    const int o = q / 5, pt = q - o * 5;
    if (pt != 0) return false;
    const int kk = (int)fix[Y.f_obs + ((c * Y.O + o) * N + i) * 5];
    ed = D + Y.d_obs + ((o * N + i) * Y.L + kk) * 3;
  }
  if (ed[1] == 0.0 && fabs(ed[0]) == 1.0) { col = 6 * c; sgn = ed[0]; rhs = ed[2]; return true; }
  if (ed[0] == 0.0 && fabs(ed[1]) == 1.0) { col = 6 * c + 3; sgn = ed[1]; rhs = ed[2]; return true; }
  return false;
}

// Column order of the stage vector inside the on-chip kernel.  Two cars: CHAIN-MAJOR - the four triple integrator chains
// (car, axis) side by side per derivative, column 4 k + chain for (position, velocity, acceleration) = k = 0, 1, 2 and
// 12 + chain for the jerk input - so that [A B] becomes "identity blocks h^m / m! on the m-th block diagonal":
//   P [A B]      is a sum of the tile shifted right by 4 m columns  (DPP row_shr within the 16 lanes of a row),
//   [A B]' T     is a sum over the registers of one lane             (row 4 k + chain <-> lane group chain, register k),
// i.e. the two products of the Riccati step need no MFMA and no LDS.  One car keeps the model's order and the MFMA form.
template <int C, bool CM> __device__ inline int oc_pcol(int l) { return (CM && l < 6 * C) ? 2 * C * (l % 3) + l / 3 : l; }   // model column -> kernel column
template <int C, bool CM> __device__ inline int oc_lcol(int p) { return (CM && p < 6 * C) ? 3 * (p % (2 * C)) + p / (2 * C) : p; }   // and back
// value of the lane CTRL columns to the left / right inside the row of 16 lanes (row_shr : 0x110 + n, row_shl : 0x100 + n), 0 outside
template <int CTRL> __device__ inline double dpp_shift0(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

// lane index recomputed from the execution mask (64-thread workgroups: lane = thread).  The passes of the iteration loop take
// their lane-dependent addresses from this instead of from values defined before the loop: those would have to stay in
// registers across the Riccati sweep, where none are spare, and come back from scratch memory at every use.
__device__ inline int fresh_lane() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

// cheap necessary condition for slot (i, slot) to carry a row, from the fix record alone (the first tests of decode_row; no
// table loads): the decode walks all N x NSLOT slots with this and runs the full test only on the few hundred survivors
template <int C>
__device__ inline bool slot_maybe(const Layout& Y, const signed char* fix, int i, int slot) {
  const int N = Y.N;
  if (slot < C * Y.SC) {
    const int c = slot / Y.SC, rr = slot - c * Y.SC;
    if (rr < 11) return true;
    if (i < 1) return false;
    const int code = (int)fix[Y.f_reg + c * N + i];
    if (rr < 16) return code >= 0 ? ((code & 3) != 3 || rr - 11 <= 3) : rr - 11 <= 1;
    int q = rr - 16;
    if (q < 5 * Y.EL) {
      if (Y.E < 1) return false;
      const int pt = q / Y.EL;
      const int e = Y.E == 1 ? 0 : (int)fix[Y.f_env + (c * N + i) * 5 + pt];
      return e >= 0 && (pt == 0 || code >= 0);
    }
    q -= 5 * Y.EL;
    const int o = q / 5, pt = q - o * 5;
    const int kk = (int)fix[Y.f_obs + ((c * Y.O + o) * N + i) * 5 + pt];
    return kk >= 0 && kk < Y.L && (pt == 0 || code >= 0);
  }
  if (C < 2 || i < 1) return false;
  const int q = slot - C * Y.SC;
  if (q < Y.NP * 8) { const int p = q >> 3, grp = (q & 7) >> 1; return (int)fix[Y.f_c2c + (p * N + i) * 4 + grp] >= 0; }
  const int q2 = q - Y.NP * 8, p = q2 >> 4, grp = (q2 >> 2) & 3, alt = q2 & 3;
  const int m = (int)fix[Y.f_c2n + (p * N + i) * 4 + grp];
  return m > 0 && ((m >> alt) & 1);
}

// lambda + kappa and w of one row for the Newton system of the next iteration (see row_step)
__device__ inline void row_weight(double s, double lam, double t, bool soft, double aq, double tau, double& w, double& lk) {
  const double il = frcp(lam);
  double zz, r2mu = 0.0;
  if (!soft) { const double mu = RHO_EL - lam, im = frcp(mu); zz = t * im; r2mu = (tau - t * mu) * im; }
  else zz = frcp(aq);
  w = frcp(s * il + zz);
  lk = lam + ((tau - s * lam) * il - r2mu) * w;
}

// ABL != 0 (diagnostic build -DMIQP_ABLATE, replayed on a batch the real kernel has solved): 15 iterations per node without
// convergence tests and with the parts named by the mask switched off - only the run time of such an instance is read
template <int C, int NSL, int ABL = 0, int GCAP = miqp::OC_GCAP>
__global__ void __launch_bounds__(64, (GCAP > 128 ? 1 : 2)) ipm_onchip_kernel(DevBuf B) {
  constexpr int OC_ABL = ABL;
  constexpr int OC_GCAP = GCAP, OC_GSLOTS = GCAP / 64, OC_GCOEF = oc_gcoef_of(GCAP);   // (shadow the capacities of the standard variant)
  constexpr bool BIG = GCAP > 128;   // the larger variant works through the list of the standard one and hands what it cannot hold to the memory-backed kernel
  static_assert(GCAP % 64 == 0 && GCAP <= 512, "general rows: whole register slots per lane, flags of up to 8 slots");
  constexpr bool CM = C == 2 && !(ABL & 1024);   // chain-major columns, shift form of the Riccati products (mask 1024 of the diagnostic build: the MFMA form)
  static_assert(C <= 2, "one 16 x 16 tile per stage");
  constexpr int NX = 6 * C, NU = 2 * C, NZ = 8 * C;
  constexpr int KB = (NX + 3) / 4;
  constexpr int RU = NX / 4, GU0 = NX % 4;
  const Layout& Y = B.Y;
  const int tid = threadIdx.x, lg = tid >> 4, lc = tid & 15;
  const int par = lg >> 1, side = lg & 1;                   // box rows of this lane: column lc, side, stages 2k + par
  const double bsgn = side ? -1.0 : 1.0;
  WAVE_DUMP(2);
  const int* const clist = BIG && B.cls_take ? B.cls_list + (size_t)(B.cls_take - 1) * B.batch_cap : nullptr;   // (the list of this launch's class: nothing to scan)
  const int nbatch = clist ? (B.cls_count[B.cls_take - 1] < B.batch_cap ? B.cls_count[B.cls_take - 1] : B.batch_cap) : BIG && B.ovf_mode == 1 ? *B.ovf_count : (*B.batch_count < B.batch_cap ? *B.batch_count : B.batch_cap);
  const int N = Y.N, NSLOT = Y.NSLOT;
  extern __shared__ double lds[];
  char* const L0 = (char*)lds;
  const OcLds LL = oc_lds_layout(N, Y.fixlen, GCAP);
  double* const Z = (double*)(L0 + LL.z);                   // [N][16]
  double* const Dg = (double*)(L0 + LL.u);                  // [N][16] diagonal contributions of the box rows
  double* const Gd = Dg + N * 16;                           // [N][16] their gradient contributions
  double* const dZ = Dg;                                    // [N][16] step (alive from the forward sweep to the row update)
  double* const scr = Dg;                                   // decode: dense scratch rows
  unsigned long long* const bkey = (unsigned long long*)(L0 + LL.r);   // decode: orderable key of the tightest right-hand side per (stage, side, column)
  double* const gswfs = (double*)(L0 + LL.r);               // [OC_GCAP][2] sqrt(w) and f of the general rows
#ifndef MIQP_KL0_PACK
#define MIQP_KL0_PACK 1   // chain-major form: the row scalings are one double per row (the gradient goes straight into Gd), the other half of the table holds the gains of two (five) more stages
#endif
  constexpr int GS = (CM && MIQP_KL0_PACK) ? 1 : 2;          // doubles per general row in gswfs
  constexpr int KL0N = OC_KL0 + (GS == 1 ? OC_GCAP / 64 : 0);   // stages whose gains stay in LDS
  double* const KL0 = gswfs + GS * OC_GCAP;                  // [KL0N][64] gains of the first stages (computed last, used first): they stay on chip
  uint4* const gmeta = (uint4*)(L0 + LL.gmeta);             // x,y: column map (nibble c = 1 + index of the coefficient of column c), z: coefficient offset | nn << 16 | stage << 20 | soft << 31, w: columns (4 bits each)
  double* const gcoef = (double*)(L0 + LL.gcoef);
  double* const grhs = (double*)(L0 + LL.grhs);
  double* const Wd = (double*)(L0 + LL.wd);
  int* const sstart = (int*)(L0 + LL.sstart);
  unsigned short* const cand = (unsigned short*)(L0 + LL.cand);
  signed char* const fix = (signed char*)(L0 + LL.fix);
  double* const KG = B.kgain + (size_t)blockIdx.x * oc_gain_doubles(N);   // [N][64]: K[q][c] of a stage at q * 16 + c, feed-forward k[q] at c = NX
  const double* KGs;   // the same base as a scalar value: loads take it from SGPRs plus the lane offset
  { const unsigned long long a = (unsigned long long)KG;
    KGs = (const double*)(((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) | (unsigned int)__builtin_amdgcn_readfirstlane((int)a)); }
  const bool gown = lg < NU && lc <= NX;   // this lane owns a gain entry
  __shared__ int sh_node;
  const unsigned long long lt = (1ull << tid) - 1ull;

  for (;;) {
    __syncthreads();
    if (tid == 0) sh_node = atomicAdd(B.work_counter, 1);
    __syncthreads();
    if (__builtin_amdgcn_readfirstlane(sh_node) >= nbatch) break;
    const int node = __builtin_amdgcn_readfirstlane(clist ? clist[sh_node] : BIG && B.ovf_mode == 1 ? B.ovf_list[sh_node] : sh_node);   // wave-uniform by construction: said so, everything derived from it
                                                                  // (instance tables, references) is then addressed from SGPRs
    // the concurrent launch of the larger variant takes the nodes known to be large before the round: the rounding probes (their
    // depth word says so) and the records marked by an earlier decode or inherited from a marked parent
    // (the split is select_kernel's snapshot batch_large: the mark the standard kernel sets when it returns a node is for the next round)
    if ((BIG && B.ovf_mode == 2) || (!BIG && B.skip_probes)) {
      const bool marked = B.batch_large ? B.batch_large[node] != 0 : is_probe_word(B.batch_depth[node]);
      if (BIG ? !marked : marked) continue;   // the other launch of the round solves it
      if (BIG && B.as_split && B.batch_large[node] != 2) continue;   // ... or the larger active-set launch (class 1), or the memory-backed launch on its own stream (class 3): large_class
    }
    WAVE_DUMP_NODE();
    const int inst = __builtin_amdgcn_readfirstlane(B.batch_inst[node]);
    const double* D = B.inst_d + (size_t)inst * Y.dstride;
    const int* T = B.inst_i + (size_t)inst * Y.istride;
    const double ts = D[Y.d_glob + 7];
    const double aqs = 2.0 * D[Y.d_misc + 0];               // quadratic weight of the soft car/car rows
    {
      const signed char* src = B.pool_fix + (size_t)B.batch_node[node] * Y.fixlen;
      for (int k = tid; k < Y.fixlen; k += 64) fix[k] = src[k];
      if (tid < 16) Wd[tid] = tid < NZ ? D[Y.d_wd + oc_lcol<C, CM>(tid)] : 0.0;
      for (int k = tid; k < N * 16; k += 64) Z[k] = 0.0;
      for (int k = tid; k < N * 32; k += 64) bkey[k] = ~0ull;
      for (int k = tid; k <= N + 1; k += 64) sstart[k] = 0;
    }
    const bool warm = B.ws_on == 2 || (B.ws_on && B.pool_Z && (B.batch_depth[node] >> 6) >= 1 && B.batch_node[node] < B.z_cap);   // roots start cold; 2: the polish starts from the incumbent's solution
    __syncthreads();
    if (warm) {   // the parent's solution, into the kernel's column order
      const double* zp = B.ws_on == 2 ? B.inc_Z + (size_t)inst * N * NZ : B.pool_Z + (size_t)B.batch_node[node] * N * NZ;
      for (int k = tid; k < N * 16; k += 64) { const int q = k & 15; if (q < NZ) Z[k] = zp[(k >> 4) * NZ + oc_lcol<C, CM>(q)]; }
      OC_WAVE_SYNC();
    } else {
    if (tid < NX) Z[tid] = D[Y.d_x0 + oc_lcol<C, CM>(tid)];
    OC_WAVE_SYNC();
    for (int i = 0; i + 1 < N; ++i) {  // free rollout (u = 0)
      if (tid < NX) {
        double acc = 0;
        if (CM) { const int k = tid / (2 * C); acc = Z[i * 16 + tid]; if (k < 2) acc += ts * Z[i * 16 + tid + 2 * C]; if (k < 1) acc += 0.5 * ts * ts * Z[i * 16 + tid + 4 * C]; }
        else for (int q = 3 * (tid / 3); q < 3 * (tid / 3) + 3; ++q) acc += ab_entry<C>(tid, q, ts) * Z[i * 16 + q];
        Z[(i + 1) * 16 + tid] = acc;
      }
      OC_WAVE_SYNC();
    }
    }
    const double* Rf = D + Y.d_ref;
    double cutoff = 1e300;
    {
      const double inc0 = fmin(inc_from_key(*(volatile unsigned long long*)&B.inc_key[inst]), B.inc_ext[inst]);
      if (B.use_cutoff && inc0 < 1e300) cutoff = inc0 - (is_probe_word(B.batch_depth[node]) ? 0.0 : B.inst_gap[inst]) * (1e-10 + fabs(inc0)) - B.inst_const[inst];   // (a heuristic leaf is cut off at the incumbent itself: as_onchip.hip)
    }
#ifdef MIQP_PROFILE
    unsigned long long ocp_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    OCP_T(tp_d0);
    double abr[KB];   // [A B] as MFMA operand (one car; two cars use the shift form, see oc_pcol)
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) abr[kb] = (!CM && 4 * kb + lg < NX && lc < NZ) ? ab_entry<C>(4 * kb + lg, lc, ts) : 0.0;

    // ---- decode, pass A: which (stage, slot) pairs carry a row; box rows go straight to their key, general rows are marked in a
    // bitmap over (stage, slot).  The pairs are walked class by class (velocity / acceleration bounds, jerk bounds, region
    // rows, rear-point edges, front-point edges, obstacles, car/car, car/car exclusions), so that the lanes of one pass take
    // the same branch of the decoder and their table loads go out together; the bitmap restores the (stage, slot) order.
    int ngen = 0;
    {
      unsigned long long* const bmp = (unsigned long long*)scr;          // [nw] one bit per (stage, slot)
      const int nw = (N * NSLOT + 63) >> 6;
      unsigned short* const pre = (unsigned short*)(bmp + nw);           // [nw] general rows before every word
      for (int k = tid; k < nw; k += 64) bmp[k] = 0ull;
      OC_WAVE_SYNC();
      const int cls_off[8] = {0, 7, 11, 16, 16 + Y.EL, 16 + 5 * Y.EL, C * Y.SC, C * Y.SC + 8 * Y.NP};
      const int cls_cnt[8] = {7, 4, 5, Y.EL, 4 * Y.EL, 5 * Y.O, 8 * Y.NP, 16 * Y.NP};
      auto take = [&](int i, int slot) {   // full test of one (stage, slot); box rows to their key, general rows to the bitmap
        if (decode_row<C, false>(Y, D, T, fix, i, slot, nullptr).active) {
          int col; double sg, rh;
          if (box_of_slot<C>(Y, D, T, fix, i, slot, col, sg, rh)) atomicMin(&bkey[(i * 2 + (sg < 0.0 ? 1 : 0)) * 16 + oc_pcol<C, CM>(col)], d2key(rh));
          else { const int pcode = i * NSLOT + slot; atomicOr(&bmp[pcode >> 6], 1ull << (pcode & 63)); }
        }
      };
      // the candidates of the sparse classes (everything but the state / input bounds) are collected first with the cheap test
      unsigned short* const plist = pre + ((nw + 1 + 3) & ~3);
      const int LCAP = (int)(((char*)(L0 + LL.r) - (char*)plist) / 2) - 64;
      int nlist = 0;
      auto flush = [&]() {
        OC_WAVE_SYNC();
        for (int j0 = 0; j0 < nlist; j0 += 64) if (j0 + tid < nlist) { const int pc = plist[j0 + tid]; const int i = pc / NSLOT; take(i, pc - i * NSLOT); }
        OC_WAVE_SYNC();
        nlist = 0;
      };
#pragma unroll 1
      for (int cl = 0; cl < 8; ++cl) {
        const int cnt = cls_cnt[cl], off = cls_off[cl];
        const bool percar = cl < 6;
        const int per = percar ? C * cnt : cnt, total = N * per;
        for (int e0 = 0; e0 < total; e0 += 64) {
          const int e = e0 + tid;
          int i = 0, slot = 0; bool in = e < total;
          if (in) { i = e / per; const int rem = e - i * per; slot = percar ? (rem / cnt) * Y.SC + off + rem % cnt : off + rem; }
          if (cl < 2) { if (in) take(i, slot); continue; }
          const bool cnd = in && slot_maybe<C>(Y, fix, i, slot);
          const unsigned long long mk = __ballot(cnd);
          if (cnd) plist[nlist + __popcll(mk & lt)] = (unsigned short)(i * NSLOT + slot);
          nlist += __popcll(mk);
          if (nlist > LCAP) flush();
        }
      }
      flush();
      if (tid == 0) { int a = 0; for (int k = 0; k < nw; ++k) { pre[k] = (unsigned short)(a < 65535 ? a : 65535); a += __popcll(bmp[k]); } pre[nw] = (unsigned short)(a < 65535 ? a : 65535); }
      OC_WAVE_SYNC();
      ngen = pre[nw];
      if (ngen <= OC_GCAP)
        for (int k = tid; k < nw; k += 64) {
          unsigned long long bits = bmp[k]; int pos = pre[k];
          while (bits) { const int b = __ffsll((long long)bits) - 1; cand[pos++] = (unsigned short)(k * 64 + b); bits &= bits - 1ull; }
        }
      OC_WAVE_SYNC();
    }
    // ---- pass B: the general rows, OC_SCR at a time through dense scratch rows; packed into LDS in (stage, slot) order
    bool overflow = ngen > OC_GCAP;
    int ncoef = 0;
    for (int c0 = 0; c0 < ngen && !overflow; c0 += OC_SCR) {
      double* g = scr + (tid & (OC_SCR - 1)) * OC_SSTR;
      RowOut r; r.active = false; r.rhs = 0; r.aq = 0;
      int i = 0, nn = 0;
      const bool mine = tid < OC_SCR && c0 + tid < ngen;
      int slot_ = 0;
      if (mine) { const int pcode = cand[c0 + tid]; i = pcode / NSLOT; slot_ = pcode - i * NSLOT; r = decode_row<C, true>(Y, D, T, fix, i, slot_, g); }
      unsigned long long map = 0ull; unsigned int cols = 0u;
      double v6[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) v6[k] = 0.0;
      if (mine) {
        for (int q = 0; q < NZ; ++q) {
          const double v = g[q];
          if (v != 0.0 && nn < 6) {
#pragma unroll
            for (int k = 0; k < 6; ++k) if (k == nn) v6[k] = v;
            const int pq = oc_pcol<C, CM>(q);
            map |= (unsigned long long)(nn + 1) << (4 * pq); cols |= (unsigned int)pq << (4 * nn); nn++;
          }
        }
      }
      const bool keep = mine && nn > 0;   // a row without coefficients constrains nothing
      // prefix sums over the lanes: row index and coefficient offset
      const unsigned long long mk = __ballot(keep);
      const unsigned long long b0 = __ballot(keep && (nn & 1)), b1 = __ballot(keep && (nn & 2)), b2 = __ballot(keep && (nn & 4));
      const int tot = __popcll(b0) + 2 * __popcll(b1) + 4 * __popcll(b2);
      if (ncoef + tot > OC_GCOEF) { overflow = true; break; }
      if (keep) {
        const int idx = sstart[N + 1] + __popcll(mk & lt);   // sstart[N + 1]: rows so far (lane-uniform value read before the update below)
        const int off = ncoef + __popcll(b0 & lt) + 2 * __popcll(b1 & lt) + 4 * __popcll(b2 & lt);
#pragma unroll
        for (int k = 0; k < 6; ++k) if (k < nn) gcoef[off + k] = v6[k];
        uint4 m4; m4.x = (unsigned int)map; m4.y = (unsigned int)(map >> 32);
        m4.z = (unsigned int)off | ((unsigned int)nn << 16) | ((unsigned int)i << 20) | (r.aq > 0.0 ? 0x80000000u : 0u); m4.w = cols;
        gmeta[idx] = m4; grhs[idx] = r.rhs;
        atomicAdd(&sstart[i + 1], 1);
      }
      OC_WAVE_SYNC();
      if (tid == 0) sstart[N + 1] += __popcll(mk);
      ncoef += tot;
      OC_WAVE_SYNC();
    }
    if (overflow && !BIG && B.bounce) {   // found too large here: marked and returned unsolved; the concurrent launch of the larger variant takes it next round
      if (tid == 0) { B.batch_ok[node] = 5; B.pool_big[B.batch_node[node]] |= 1; if (B.stats) { atomicAdd(&B.stats[3], 1ull); atomicAdd(&B.stats[8 + (ngen >= 512 ? 15 : ngen / 32)], 1ull); } }
      continue;
    }
    if (overflow && BIG && B.as_split) {   // (four concurrent launches: no hand-over list) marked and returned unsolved; the memory-backed launch of the next round takes the record
      if (tid == 0) { B.batch_ok[node] = 5; B.pool_big[B.batch_node[node]] |= 2; }
      continue;
    }
    if (overflow) {   // more general rows than fit on chip: the memory-backed kernel takes the node
      if (tid == 0) { const int q = atomicAdd(BIG ? B.ovf2_count : B.ovf_count, 1); (BIG ? B.ovf2_list : B.ovf_list)[q] = node; if (B.stats && !BIG) { atomicAdd(&B.stats[3], 1ull); atomicAdd(&B.stats[8 + (ngen >= 512 ? 15 : ngen / 32)], 1ull); if (ngen >= 480) { atomicMax(&B.stats[5], (unsigned long long)ngen); atomicAdd(&B.stats[6], (unsigned long long)ngen); atomicAdd(&B.stats[7], 1ull); } } }
      continue;
    }
    const int NM = sstart[N + 1];
    OC_WAVE_SYNC();
    if (tid == 0) { int a = 0; for (int i = 0; i <= N; ++i) { const int n = sstart[i]; a += n; sstart[i] = a; } }   // sstart[i]: first general row of stage i (counts were stored at i + 1)
    OC_WAVE_SYNC();

    // ---- initial row state (the interior point starts at the free rollout, or at the parent's solution: DevBuf::pool_Z)
    double bs[NSL], bl[NSL], bt[NSL];
    unsigned int bact = 0u;
    double csum = 0.0, tsum = 0.0; int cnt = 0;
#pragma unroll
    for (int k = 0; k < NSL; ++k) {
      const int i = 2 * k + par;
      bs[k] = 1.0; bl[k] = 1.0; bt[k] = 1.0;
      if (i < N && lc < NZ) {
        const unsigned long long key = bkey[(i * 2 + side) * 16 + lc];
        if (key != ~0ull) {
          const double rh = key2d(key), c = rh - bsgn * Z[i * 16 + lc];
          double s, t, l0;
          init_elastic(c, warm, B.ws_mu, B.ws_delta, s, l0, t);
          bs[k] = s; bl[k] = l0; bt[k] = t; bact |= 1u << k;
          csum += s * l0 + t * (RHO_EL - l0); cnt += 2; tsum += t;
        }
      }
    }
    double gs_[OC_GSLOTS], gl_[OC_GSLOTS], gt_[OC_GSLOTS], ggd[OC_GSLOTS];
    unsigned int gflag = 0u;   // bit q: the row of slot q is a quadratic-soft row; bit 8 + q: slot q carries a row
#pragma unroll
    for (int q = 0; q < OC_GSLOTS; ++q) {
      const int r = q * 64 + tid;
      gs_[q] = 1.0; gl_[q] = 1.0; gt_[q] = 1.0; ggd[q] = 0.0;
      if (r < NM) {
        const uint4 m4 = gmeta[r];
        gflag |= (256u << q) | ((m4.z & 0x80000000u) ? (1u << q) : 0u);
        const int off = (int)(m4.z & 0xFFFFu), nn = (int)((m4.z >> 16) & 7u), i = (int)((m4.z >> 20) & 0x7FFu);
        double c = grhs[r];
#pragma unroll
        for (int k = 0; k < 6; ++k) if (k < nn) c -= gcoef[off + k] * Z[i * 16 + ((m4.w >> (4 * k)) & 15u)];
        if (!(m4.z & 0x80000000u)) {
          double s, t, l0;
          init_elastic(c, warm, B.ws_mu, B.ws_delta, s, l0, t);
          gs_[q] = s; gl_[q] = l0; gt_[q] = t;
          csum += s * l0 + t * (RHO_EL - l0); cnt += 2; tsum += t;
        } else {
          const double lam = fmax(1.0, -2.0 * c * aqs + 1.0), s = c + lam / aqs;
          gs_[q] = s; gl_[q] = lam; gt_[q] = 1.0; csum += s * lam; cnt += 1;   // t of a soft row is never used
        }
      }
    }
    OC_WAVE_SYNC();   // the keys are consumed: their region becomes (sqrt(w), f)
    OCP_T(tp_d1); OCP_ACC(0, tp_d0, tp_d1);
    double comp = wave_sum(csum);
    tsum = wave_sum(tsum);
    int ncomp = (int)wave_sum((double)cnt);
    if (ncomp < 1) ncomp = 1;
    comp /= ncomp;
    int nbox = __popc(bact); nbox = (int)wave_sum((double)nbox);
    const int NROWS = NM + nbox;

    int it = 0, ok = 0;
    double resid_fac = 1.0, R0 = 0.0, obj = 0.0;
    double sigma = QP_SIGMA;
    unsigned long long rowiters = 0;
    {
      double o = 0.0;
      for (int k = tid; k < N * 16; k += 64) { const int q = k & 15; if (q < NZ) { const double d = Z[k] - Rf[(k >> 4) * NZ + oc_lcol<C, CM>(q)]; o += Wd[q] * d * d; } }
      obj = wave_sum(o);
    }
    for (it = 1; it <= QP_MAXIT; ++it) {
      if (OC_ABL) { if (it > 15) { ok = 1; break; } }
      else {
      if (comp < B.qp_tol * fmax(1.0, fabs(obj)) && resid_fac * R0 < 1e-7) { ok = 1; break; }
      // a rounding probe is a heuristic (it lies inside the first child: the children stay exhaustive without it).  One that has not
      // converged after probe_itcap iterations is abandoned: 1 % of the probes that leave slack front-point disjunctions undecided
      // ran to the iteration limit (80) and held the launch of the larger variant up
      const bool pump_probe = BIG && B.pump_inc && B.pump_max > 0 && is_probe_word(B.batch_depth[node]);
      if (BIG) { const int pcap = (cutoff < 1e299 && (!pump_probe || (B.pump_inc & 2))) ? B.probe_itcap : B.probe_itcap0; if (pcap > 0 && it > pcap && B.ws_on != 2 && is_probe_word(B.batch_depth[node])) { ok = 2; break; } }
      // (deferral, see ipm_kernel: a node still unconverged after defer_cap iterations goes back on its list with its iterate as warm start, once)
      if (B.defer_cap > 0 && it > B.defer_cap && B.ws_on == 1 && B.pool_Z && (B.batch_depth[node] >> 6) >= 1 && B.batch_node[node] < B.z_cap
          && !is_probe_word(B.batch_depth[node]) && !(B.pool_big[B.batch_node[node]] & 8)) { ok = 5; break; }
      if (it > 1 && resid_fac * R0 < B.cut_gate * (1.0 + fabs(obj)) && obj + (pump_probe ? 0.0 : RHO_EL * tsum) - (double)ncomp * comp - resid_fac * R0 * D[Y.d_misc + 2] > cutoff + 1e-9 * fabs(cutoff)) { ok = 2; break; }   // (dual value minus the allowance for the stationarity residual still left, see batch_bound)
      }
      const double tau = sigma * comp;   // the common centring target of every complementarity pair
      OCP_T(tp_r0);
      // ================= row pass 1: weights of every row for this iteration
      // box rows -> diagonal and gradient contribution per (stage, column): the two sides of a column sit in lanes l, l ^ 16
      // (no branches around the slots: the ten independent chains interleave; unused slots carry a benign state and are masked)
      const int l1 = fresh_lane();
      constexpr int NZW1 = (2 * NSL * 16 + 63) / 64;
      double rf1[CM ? NZW1 : 1];   // chain-major form: the objective's share of the gradient joins Gd at the end of the pass
      if constexpr (CM) {
#pragma unroll
        for (int j = 0; j < NZW1; ++j) { const int k = l1 + 64 * j, q = k & 15; rf1[j] = (k < N * 16 && q < NZ) ? Rf[(k >> 4) * NZ + oc_lcol<C, CM>(q)] : 0.0; }
      }
      double* const dgp = Dg + (l1 >> 5) * 16 + (l1 & 15); double* const gdp = dgp + N * 16;
      const bool side0 = ((l1 >> 4) & 1) == 0; const double sg1 = side0 ? 1.0 : -1.0; const int par1 = l1 >> 5;
#pragma unroll
      for (int k0 = 0; k0 < (((OC_ABL) & 2) ? 0 : NSL); k0 += OC_GRP) {
        if (2 * k0 < N) {   // wave-uniform; a basic block per group keeps the live ranges short
#pragma unroll
          for (int k = k0; k < k0 + OC_GRP && k < NSL; ++k) {
            const int i = 2 * k + par1;
            const bool act = (bact >> k) & 1u;
            double w, lk; row_weight(bs[k], bl[k], bt[k], false, 0.0, tau, w, lk);
            w = act ? w : 0.0; lk = act ? lk * sg1 : 0.0;
            w = sum_xor16(w); lk = sum_xor16(lk);
            if (side0 && i < N) { dgp[k * 32] = w; gdp[k * 32] = lk; }   // row 2 k + par, column lc: one base, the slot as immediate offset
          }
        }
      }
      if constexpr (CM) OC_WAVE_SYNC();   // the box rows' gradient entries are in place before the general rows add theirs
#pragma unroll
      for (int q = 0; q < (((OC_ABL) & 4) ? 0 : OC_GSLOTS); ++q) {
        const int r = q * 64 + l1;
        const bool soft = ((gflag >> q) & 1u) != 0u;
        double w, lk; row_weight(gs_[q], gl_[q], gt_[q], soft, aqs, tau, w, lk);
        const double isw = frsq(w);
        if (r < NM) {
          gswfs[GS * r] = w * isw;
          if constexpr (CM) {   // gradient of the row straight into the stage's gradient vector (LDS atomics; the sweep then reads one array)
            const uint4 m4 = gmeta[r];
            const int off = (int)(m4.z & 0xFFFFu), nn = (int)((m4.z >> 16) & 7u), i = (int)((m4.z >> 20) & 0x7FFu);
#pragma unroll
            for (int k = 0; k < 6; ++k) if (k < nn) atomicAdd(&Gd[i * 16 + ((m4.w >> (4 * k)) & 15u)], gcoef[off + k] * lk);
          } else gswfs[2 * r + 1] = lk * isw;
        }
      }
      OC_WAVE_SYNC();
      double rmax = 0.0;
      if constexpr (CM) {
#pragma unroll
        for (int j = 0; j < NZW1; ++j) {
          const int k = l1 + 64 * j, q = k & 15;
          if (k < N * 16 && q < NZ) { const double v = Gd[k] + 2.0 * Wd[q] * (Z[k] - rf1[j]); Gd[k] = v; rmax = fmax(rmax, fabs(v)); }
        }
        OC_WAVE_SYNC();
      }
      OCP_T(tp_r1); OCP_ACC(1, tp_r0, tp_r1);
      // ================= backward sweep (Riccati recursion in MFMA registers, see ipm_kernel)
      if constexpr (CM) {
        // Chain-major form.  Gd holds the complete gradient rr of every stage; the vector p of the recursion lives in ROW layout
        // (lane group = chain, register = derivative: the layout of the tile's rows), so that [A B]' p is a sum over the lane's
        // registers.  The input block is eliminated by Gauss-Jordan on the four input rows of the tile (one register): row q is
        // broadcast over the lane groups, normalised by its pivot (row_newbcast picks the pivot / multiplier column inside the
        // rows), and subtracted from the others; after pivot 0 its column is spent (a unit vector) and carries the vector s_u,
        // so the rows end as [K | k] in the layout the forward sweep reads, and ONE MFMA forms the Schur complement of the
        // tile and, in column 12, the next p.
        const double h1 = ts, h2 = 0.5 * ts * ts, h3 = ts * ts * ts / 6.0;
        d4_t Pd = {0.0, 0.0, 0.0, 0.0};
        double pr[3];
        auto phiM = [&](int j, d4_t& acc) {
          const int rb = sstart[j], re = sstart[j + 1];
          acc = d4_t{0.0, 0.0, 0.0, 0.0};
          auto kblock = [&](int r0) {
            const int r = r0 + lg;
            double a = 0.0;
            if (r < re) {
              const uint4 m4 = gmeta[r];
              const unsigned int nib = lc < 8 ? (m4.x >> (4 * lc)) & 15u : (m4.y >> (4 * (lc - 8))) & 15u;
              if (nib) a = gcoef[(m4.z & 0xFFFFu) + nib - 1u] * gswfs[GS * r];
            }
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc, 0, 0, 0);
          };
#if MIQP_PHI_BRANCH
          if (!((OC_ABL) & 8)) { kblock(rb); if (rb + 4 < re) { kblock(rb + 4);
          for (int r0 = rb + 8; r0 < re; r0 += 4) kblock(r0); } }
#else
          if (!((OC_ABL) & 8)) { kblock(rb); kblock(rb + 4);
          for (int r0 = rb + 8; r0 < re; r0 += 4) kblock(r0); }
#endif
          const int lp = fresh_lane(), lcp = lp & 15, lgp = lp >> 4;
          const double dd = lcp < NZ ? 2.0 * Wd[lcp] + Dg[j * 16 + lcp] : 0.0;
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) if (lgp + 4 * rg == lcp) acc[rg] += dd;
        };
        auto bcast_group = [&](double v, int q) {   // the value of lane group q in every group
          // (as ONE MFMA with a selector operand this was 4 % slower in round 4: the elimination is a dependent chain, the 64-cycle latency costs more than the 13 issue slots it frees)
          double t = lg == q ? v : 0.0;
          t = sum_xor16(t); return sum_xor32(t);
        };
        d4_t accA;
        phiM(N - 1, accA);
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) Pd[rg] = (rg < 3 && lc < NX) ? accA[rg] : 0.0;   // u_{N-1} = 0: P = Phi_xx, p = rr_x
        { const int lq = fresh_lane();
#pragma unroll
          for (int k = 0; k < 3; ++k) pr[k] = Gd[(N - 1) * 16 + (lq >> 4) + 4 * k]; }
        if (N >= 2) phiM(N - 2, accA);
        for (int i = N - 2; i >= 0; --i) {
          d4_t acc = accA;
          OCP_T(tp_s0);
          if (i > 0) phiM(i - 1, accA);
          OCP_T(tp_s1); OCP_ACC(2, tp_s0, tp_s1);
          double sv[4];
          { const int lq = fresh_lane(); const double* gr = Gd + i * 16 + (lq >> 4);
            sv[0] = gr[0]; sv[1] = gr[4]; sv[2] = gr[8]; sv[3] = gr[12]; }
          if (!((OC_ABL) & 16)) {
            d4_t Tt;
#pragma unroll
            for (int r = 0; r < 3; ++r) Tt[r] = Pd[r] + h1 * dpp_shift0<0x114>(Pd[r]) + h2 * dpp_shift0<0x118>(Pd[r]) + h3 * dpp_shift0<0x11C>(Pd[r]);
            acc[0] += Tt[0];
            acc[1] += Tt[1] + h1 * Tt[0];
            acc[2] += Tt[2] + h1 * Tt[1] + h2 * Tt[0];
            acc[3] += h1 * Tt[2] + h2 * Tt[1] + h3 * Tt[0];
          }
          if (!((OC_ABL) & 32)) {
            sv[0] += pr[0];
            sv[1] += pr[1] + h1 * pr[0];
            sv[2] += pr[2] + h1 * pr[1] + h2 * pr[0];
            sv[3] += h1 * pr[2] + h2 * pr[1] + h3 * pr[0];
          }
          const double own = acc[3];
          OCP_T(tp_s2); OCP_ACC(3, tp_s1, tp_s2);
          double W3 = own;
          if (!((OC_ABL) & 64)) {
            {   // pivot 0, the vector still beside the rows
              const double d = dpp_mov<0x15C>(W3);
              const double rowq = bcast_group(W3, 0), s0 = bcast_group(sv[3], 0);
              const double inv = frcp(fmax(dpp_mov<0x15C>(rowq), 1e-300));
              const double rown = rowq * inv, s0n = s0 * inv;
              W3 = lg == 0 ? rown : fma(-d, rown, W3);
              sv[3] = lg == 0 ? s0n : fma(-d, s0n, sv[3]);
              W3 = lc == 12 ? sv[3] : W3;
            }
#define OC_PIVOT(Q, CTRL) { const double d = dpp_mov<CTRL>(W3); const double rowq = bcast_group(W3, Q); \
              const double inv = frcp(fmax(dpp_mov<CTRL>(rowq), 1e-300)); const double rown = rowq * inv; \
              W3 = lg == Q ? rown : fma(-d, rown, W3); }
            OC_PIVOT(1, 0x15D) OC_PIVOT(2, 0x15E) OC_PIVOT(3, 0x15F)
#undef OC_PIVOT
          }
          OCP_T(tp_s4); OCP_ACC(5, tp_s2, tp_s4);
          if (lc <= NX) { if (i < KL0N) KL0[i * 64 + lg * 16 + lc] = W3; else KG[i * 64 + lg * 16 + lc] = W3; }
#pragma unroll
          for (int k = 0; k < 3; ++k) acc[k] = lc == 12 ? sv[k] : acc[k];
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-own, lc <= NX ? W3 : 0.0, acc, 0, 0, 0);
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) Pd[rg] = (rg < 3 && lc < NX) ? acc[rg] : 0.0;
#pragma unroll
          for (int k = 0; k < 3; ++k) pr[k] = dpp_mov<0x15C>(acc[k]);
          OCP_T(tp_s5); OCP_ACC(6, tp_s4, tp_s5);
        }
      } else {
        d4_t Pd = {0.0, 0.0, 0.0, 0.0};
        double pcol = 0.0;
        double rfn = lc < NZ ? Rf[(N - 1) * NZ + oc_lcol<C, CM>(lc)] : 0.0;
        // Phi_j = 2W + diag(box rows) + Gh' Gh, rr_j = 2W(z - ref) + (box rows) + Gh' f : the general rows of stage j enter four
        // at a time, lane (g, c) picks the coefficient of column c of row 4 kb + g through the row's column map
        auto phi = [&](int j, d4_t& acc, double& rrc) {
          const int rb = sstart[j], re = sstart[j + 1];
          acc = d4_t{0.0, 0.0, 0.0, 0.0};
          double racc = 0.0;
          auto kblock = [&](int r0) {
            const int r = r0 + lg;
            double a = 0.0, f = 0.0;
            if (r < re) {
              const uint4 m4 = gmeta[r];
              const unsigned int nib = lc < 8 ? (m4.x >> (4 * lc)) & 15u : (m4.y >> (4 * (lc - 8))) & 15u;
              if (nib) { a = gcoef[(m4.z & 0xFFFFu) + nib - 1u] * gswfs[2 * r]; f = gswfs[2 * r + 1]; }
            }
            racc += a * f;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc, 0, 0, 0);
          };
          // the first eight rows (most stages have fewer) as straight-line code that the scheduler can spread over the
          // elimination of the previous stage; the rest in a loop
          if (!((OC_ABL) & 8)) { kblock(rb); kblock(rb + 4);
          for (int r0 = rb + 8; r0 < re; r0 += 4) kblock(r0); }
          racc = sum_xor16(racc); racc = sum_xor32(racc);
          const int lp = fresh_lane(), lcp = lp & 15, lgp = lp >> 4;
          {
            const double dd = lcp < NZ ? 2.0 * Wd[lcp] + Dg[j * 16 + lcp] : 0.0;
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) if (lgp + 4 * rg == lcp) acc[rg] += dd;
          }
          rrc = lcp < NZ ? racc + Gd[j * 16 + lcp] + 2.0 * Wd[lcp] * (Z[j * 16 + lcp] - rfn) : 0.0;
          if (j > 0 && lcp < NZ) rfn = Rf[(j - 1) * NZ + oc_lcol<C, CM>(lcp)];
          if (it == 1) rmax = fmax(rmax, fabs(rrc));
        };
        d4_t accA; double rrA;
        phi(N - 1, accA, rrA);
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) Pd[rg] = (lg + 4 * rg < NX && lc < NX) ? accA[rg] : 0.0;   // u_{N-1} = 0: P = Phi_xx, p = rr_x
        pcol = lc < NX ? rrA : 0.0;
        if (N >= 2) phi(N - 2, accA, rrA);
        for (int i = N - 2; i >= 0; --i) {
          d4_t acc = accA; const double rrc = rrA;
          OCP_T(tp_s0);
          if (i > 0) phi(i - 1, accA, rrA);   // independent of this stage's elimination: its LDS round trips hide behind the MFMA chains below
          OCP_T(tp_s1); OCP_ACC(2, tp_s0, tp_s1);
          double part = 0.0;
          if constexpr (CM) {
            // T = P [A B] (columns shifted by 4 m, weight h^m / m!), S += [A B]' T (registers of the lane), part = [A B]' p
            const double h1 = ts, h2 = 0.5 * ts * ts, h3 = ts * ts * ts / 6.0;
            if (!((OC_ABL) & 16)) {
              d4_t Tt;
#pragma unroll
              for (int r = 0; r < 3; ++r) Tt[r] = Pd[r] + h1 * dpp_shift0<0x114>(Pd[r]) + h2 * dpp_shift0<0x118>(Pd[r]) + h3 * dpp_shift0<0x11C>(Pd[r]);
              acc[0] += Tt[0];
              acc[1] += Tt[1] + h1 * Tt[0];
              acc[2] += Tt[2] + h1 * Tt[1] + h2 * Tt[0];
              acc[3] += h1 * Tt[2] + h2 * Tt[1] + h3 * Tt[0];
            }
            if (!((OC_ABL) & 32)) part = pcol + h1 * dpp_shift0<0x114>(pcol) + h2 * dpp_shift0<0x118>(pcol) + h3 * dpp_shift0<0x11C>(pcol);
          } else {
            d4_t accT = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kb = 0; kb < (((OC_ABL) & 16) ? 0 : KB); ++kb) accT = __builtin_amdgcn_mfma_f64_16x16x4f64(Pd[kb], abr[kb], accT, 0, 0, 0);
#pragma unroll
            for (int kb = 0; kb < (((OC_ABL) & 16) ? 0 : KB); ++kb) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(abr[kb], accT[kb], acc, 0, 0, 0);
#pragma unroll
            for (int kb = 0; kb < (((OC_ABL) & 32) ? 0 : KB); ++kb) part += abr[kb] * __shfl(pcol, 4 * kb + lg);
            if (!((OC_ABL) & 32)) { part = sum_xor16(part); part = sum_xor32(part); }
          }
          const double svc = rrc + part;
          const double own = acc[RU];
          OCP_T(tp_s2); OCP_ACC(3, tp_s1, tp_s2);
          double Lm[NU][NU], su[NU], dinv[NU], dvec[NU];
#pragma unroll
          for (int q = 0; q < NU; ++q) {
            su[q] = readlane_d(svc, NX + q);
#pragma unroll
            for (int q2 = 0; q2 <= q; ++q2) Lm[q][q2] = readlane_d(own, (GU0 + q) * 16 + NX + q2);
          }
          if ((OC_ABL) & 64) {
#pragma unroll
            for (int a = 0; a < NU; ++a) { dvec[a] = 1.0; dinv[a] = 1.0; }
          } else
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
          OCP_T(tp_s3); OCP_ACC(4, tp_s2, tp_s3);
          double o1, o2, o3;   // own of lane ^ 16, ^ 32, ^ 48
          { double ev, od, lo_, hi_; rows16(own, ev, od); o1 = (lg & 1) ? ev : od; halves32(own, lo_, hi_); o2 = (lg & 2) ? lo_ : hi_;
            halves32(o1, lo_, hi_); o3 = (lg & 2) ? lo_ : hi_; }
          double col[NU], xk[NU], kk[NU];
#pragma unroll
          for (int q = 0; q < NU; ++q) { int m = lg ^ (GU0 + q); col[q] = m == 0 ? own : (m == 1 ? o1 : (m == 2 ? o2 : o3)); }
#pragma unroll
          for (int a = 0; a < NU; ++a) {
            double v = col[a], w2 = su[a];
#pragma unroll
            for (int q = 0; q < (((OC_ABL) & 64) ? 0 : a); ++q) { v -= Lm[a][q] * xk[q]; w2 -= Lm[a][q] * kk[q]; }
            xk[a] = v; kk[a] = w2;
          }
#pragma unroll
          for (int a = 0; a < NU; ++a) { xk[a] *= dinv[a]; kk[a] *= dinv[a]; }
#pragma unroll
          for (int a = NU - 1; a >= 0; --a) {
            double v = xk[a], w2 = kk[a];
#pragma unroll
            for (int q = (((OC_ABL) & 64) ? NU : a + 1); q < NU; ++q) { v -= Lm[q][a] * xk[q]; w2 -= Lm[q][a] * kk[q]; }
            xk[a] = v; kk[a] = w2;
          }
          OCP_T(tp_s4); OCP_ACC(5, tp_s3, tp_s4);
          {
            const bool urow = lg >= GU0 && lg < GU0 + NU;
            double bop = 0.0, kop = 0.0;
#pragma unroll
            for (int q = 0; q < NU; ++q) if (lg - GU0 == q) { bop = xk[q]; kop = kk[q]; }
            if (urow && lc <= NX) { const double kv_ = lc < NX ? bop : kop; if (i < KL0N) KL0[i * 64 + (lg - GU0) * 16 + lc] = kv_; else KG[i * 64 + (lg - GU0) * 16 + lc] = kv_; }
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(urow ? -own : 0.0, urow ? bop : 0.0, acc, 0, 0, 0);
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) Pd[rg] = (lg + 4 * rg < NX && lc < NX) ? acc[rg] : 0.0;
            double pn = svc;
#pragma unroll
            for (int q = 0; q < NU; ++q) pn -= col[q] * kk[q];
            pcol = lc < NX ? pn : 0.0;
          }
          OCP_T(tp_s5); OCP_ACC(6, tp_s4, tp_s5);
        }
      }
      OCP_T(tp_f0);
      if (it == 1) R0 = wave_max(rmax);
      __syncthreads();   // the gains of all stages are stored; D | Gd are consumed: the region becomes dZ | gains
      // ================= forward sweep.  Lane (g, c), g < NU, c <= NX owns the gain K[g][c] (c = NX: feed-forward) of every stage:
      // u[g] = -k[g] - sum_c K[g][c] dx[c] is one product per lane and a sum over the 16 lanes of the row (DPP, no LDS); the
      // gains of the first OC_KL0 stages never left the CU (LDS), the others come back from L2 OC_PF stages ahead of their use
      {
        // no branch inside the stage loop (the memory counters stay exact): the 2 NSL - 1 stages always run, the ones behind
        // the horizon on clamped addresses, and write rows of the step buffer that nobody reads
        const int lf = fresh_lane(), lcf = lf & 15, lgf = lf >> 4;
        const bool gownf = lgf < NU && lcf <= NX;
        auto kload = [&](int i) { const int ic = i < N - 2 ? i : N - 2; return KGs[ic * 64 + lf]; };
        double kring[OC_PF];
#pragma unroll
        for (int i = 0; i < OC_PF; ++i) kring[i] = kload(KL0N + i);
        if (lf < 16) dZ[lf] = 0.0;
        // [A B] row of this lane, rebuilt from a value of this iteration so that it is not kept in registers across the backward sweep
        const double tsl = fma(0.0, tau, ts);
        double ca[3], cb;
        { const int k3 = CM ? (lf / (2 * C)) % 3 : lf % 3;   // derivative this lane's state column carries
#pragma unroll
          for (int m = 0; m < 3; ++m) { int d = m - k3; ca[m] = d < 0 ? 0.0 : (d == 0 ? 1.0 : (d == 1 ? tsl : 0.5 * tsl * tsl)); }
          cb = k3 == 0 ? tsl * tsl * tsl / 6.0 : (k3 == 1 ? 0.5 * tsl * tsl : tsl); }
        const int chs = lf < NX ? (CM ? lf % (2 * C) : lf / 3) : 0, q0 = CM ? chs : 3 * chs, qs = CM ? 2 * C : 1;   // the chain's (p, v, a) sit at q0 + m qs
        OC_WAVE_SYNC();
#pragma unroll
        for (int i = 0; i < (((OC_ABL) & 128) ? 0 : 2 * NSL - 1); ++i) {
          double kq;
          if (i < KL0N) kq = KL0[i * 64 + lf];
          else { kq = kring[(i - KL0N) % OC_PF]; kring[(i - KL0N) % OC_PF] = kload(i + OC_PF); }
          const double xq = lcf < NX ? dZ[i * 16 + lcf] : 1.0;
          double pu = gownf ? -kq * xq : 0.0;   // (a select: the entries nobody owns were never written)
          pu += dpp_mov<0xB1>(pu); pu += dpp_mov<0x4E>(pu); pu += dpp_mov<0x141>(pu); pu += dpp_mov<0x140>(pu);   // sum over the row
          if (lcf == 0 && lgf < NU) dZ[i * 16 + NX + lgf] = pu;
          OC_WAVE_SYNC();
          {
            const double* x = dZ + i * 16;
            const double xn = ca[0] * x[q0] + ca[1] * x[q0 + qs] + ca[2] * x[q0 + 2 * qs] + cb * x[NX + chs];
            if (lf < NX) dZ[(i + 1) * 16 + lf] = xn;
          }
          OC_WAVE_SYNC();
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (tid < NU) dZ[(N - 1) * 16 + NX + tid] = 0.0;
      OC_WAVE_SYNC();
      OCP_T(tp_f1); OCP_ACC(7, tp_f0, tp_f1);
      // ================= step length: ratio test over all rows
      double rinv = 0.0, a0 = 0.0, a1 = 0.0, a2 = 0.0;
      const int l2 = fresh_lane();
      const double* const dzp = dZ + (l2 >> 5) * 16 + (l2 & 15);   // step of the box rows of this lane: row 2 k + par at offset 32 k
      const double sg2 = ((l2 >> 4) & 1) ? -1.0 : 1.0;
#pragma unroll
      for (int k0 = 0; k0 < (((OC_ABL) & 256) ? 0 : NSL); k0 += OC_GRP) {
        if (2 * k0 < N) {
#pragma unroll
          for (int k = k0; k < k0 + OC_GRP && k < NSL; ++k) {
            const bool act = (bact >> k) & 1u;
            const double s = bs[k], lam = bl[k], t = bt[k], gd = act ? sg2 * dzp[k * 32] : 0.0;
            double ds, dl, dt; row_step(s, lam, t, 0.0, gd, tau, ds, dl, dt);
            const double mu = RHO_EL - lam;
            const double rr_ = fmax(fmax(-ds * __builtin_amdgcn_rcp(s), -dl * __builtin_amdgcn_rcp(lam)), fmax(-dt * __builtin_amdgcn_rcp(t), dl * __builtin_amdgcn_rcp(mu)));
            rinv = fmax(rinv, act ? rr_ : 0.0);
            const double m_ = act ? 1.0 : 0.0;
            a0 += m_ * (s * lam + t * mu); a1 += m_ * (s * dl + lam * ds - t * dl + mu * dt); a2 += m_ * (ds * dl - dt * dl);
          }
        }
      }
#pragma unroll
      for (int q = 0; q < OC_GSLOTS; ++q) {
        const int r = q * 64 + l2;
        const bool used = ((gflag >> (8 + q)) & 1u) != 0u, soft = ((gflag >> q) & 1u) != 0u;
        double gd = 0.0;
        if (used) {
          const uint4 m4 = gmeta[r];
          const int off = (int)(m4.z & 0xFFFFu), nn = (int)((m4.z >> 16) & 7u), i = (int)((m4.z >> 20) & 0x7FFu);
#pragma unroll
          for (int k = 0; k < 6; ++k) if (k < nn) gd += gcoef[off + k] * dZ[i * 16 + ((m4.w >> (4 * k)) & 15u)];
        }
        ggd[q] = gd;
        const double s = gs_[q], lam = gl_[q], t = gt_[q];
        double ds, dl, dt; row_step(s, lam, t, soft ? aqs : 0.0, gd, tau, ds, dl, dt);
        const double mu = RHO_EL - lam;
        double rr_ = fmax(-ds * __builtin_amdgcn_rcp(s), -dl * __builtin_amdgcn_rcp(lam));
        const double r2_ = fmax(-dt * __builtin_amdgcn_rcp(t), dl * __builtin_amdgcn_rcp(mu));
        rr_ = soft ? rr_ : fmax(rr_, r2_);
        rinv = fmax(rinv, used ? rr_ : 0.0);
        const double m_ = used ? 1.0 : 0.0, e_ = (used && !soft) ? 1.0 : 0.0;
        a0 += m_ * (s * lam) + e_ * (t * mu); a1 += m_ * (s * dl + lam * ds) + e_ * (mu * dt - t * dl); a2 += m_ * (ds * dl) - e_ * (dt * dl);
      }
      rowiters += (unsigned long long)NROWS;
      rinv = wave_max(rinv);
      const double amax = rinv > 1e-300 ? 1.0 / rinv : 1e300;
      a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2);
      const double alpha = fmin(1.0, MIQP_STEPFRAC * amax);
      comp = (a0 + alpha * a1 + alpha * alpha * a2) / ncomp;
      OCP_T(tp_f2); OCP_ACC(8, tp_f1, tp_f2);
      // ================= update
      double tnew = 0.0;
      const int l3 = fresh_lane();
      // objective of the next iterate: its reference values are requested here and meet the updated Z at the end of the pass
      constexpr int NZW = (2 * NSL * 16 + 63) / 64;
      double rfv[NZW];
#pragma unroll
      for (int j = 0; j < NZW; ++j) { const int k = l3 + 64 * j, q = k & 15; rfv[j] = (k < N * 16 && q < NZ) ? Rf[(k >> 4) * NZ + oc_lcol<C, CM>(q)] : 0.0; }
#pragma unroll
      for (int j = 0; j < NZW; ++j) { const int k = l3 + 64 * j; if (k < N * 16) Z[k] += alpha * dZ[k]; }
      const double* const dzq = dZ + (l3 >> 5) * 16 + (l3 & 15);
      const double sg3 = ((l3 >> 4) & 1) ? -1.0 : 1.0;
#pragma unroll
      for (int k0 = 0; k0 < (((OC_ABL) & 512) ? 0 : NSL); k0 += OC_GRP) {
        if (2 * k0 < N) {
#pragma unroll
          for (int k = k0; k < k0 + OC_GRP && k < NSL; ++k) {
            const bool act = (bact >> k) & 1u;
            const double s = bs[k], lam = bl[k], t = bt[k], gd = act ? sg3 * dzq[k * 32] : 0.0;
            double ds, dl, dt; row_step(s, lam, t, 0.0, gd, tau, ds, dl, dt);
            const double al = act ? alpha : 0.0;
            bs[k] = s + al * ds; bl[k] = lam + al * dl; bt[k] = t + al * dt;
            tnew += act ? bt[k] : 0.0;
          }
        }
      }
#pragma unroll
      for (int q = 0; q < OC_GSLOTS; ++q) {
        const bool used = ((gflag >> (8 + q)) & 1u) != 0u, soft = ((gflag >> q) & 1u) != 0u;
        const double s = gs_[q], lam = gl_[q], t = gt_[q];
        double ds, dl, dt; row_step(s, lam, t, soft ? aqs : 0.0, ggd[q], tau, ds, dl, dt);
        const double al = used ? alpha : 0.0;
        gs_[q] = s + al * ds; gl_[q] = lam + al * dl; gt_[q] = t + al * dt;
        tnew += (used && !soft) ? gt_[q] : 0.0;
      }
      tsum = wave_sum(tnew);
      {
        const int l4 = fresh_lane();
        double o = 0.0;
#pragma unroll
        for (int j = 0; j < NZW; ++j) { const int k = l4 + 64 * j, q = k & 15; if (k < N * 16 && q < NZ) { const double d = Z[k] - rfv[j]; o += Wd[q] * d * d; } }
        obj = wave_sum(o);
      }
      resid_fac *= (1.0 - alpha);
      sigma = fmin(QP_SIGMA_HI, fmax(QP_SIGMA_LO, 1.0 - alpha));
      OC_WAVE_SYNC();
      OCP_T(tp_f3); OCP_ACC(9, tp_f2, tp_f3);
      if (!(OC_ABL) && alpha < 1e-12) break;
    }
    if (ok == 5) {   // deferred: the iterate becomes the record's warm start (the layout eval_kernel writes: [N][NZ], logical columns)
      double* zp = B.pool_Z + (size_t)B.batch_node[node] * N * NZ;
      for (int k = tid; k < N * 16; k += 64) { const int q = k & 15; if (q < NZ) zp[(k >> 4) * NZ + oc_lcol<C, CM>(q)] = Z[k]; }
      if (tid == 0) {
        B.batch_ok[node] = 5; B.pool_big[B.batch_node[node]] |= 8; B.batch_it[node] = it - 1;
        atomicAdd((unsigned long long*)&B.inst_iters[inst], (unsigned long long)(it - 1));
        atomicAdd(B.stat_rowiters, rowiters);
      }
      continue;
    }
    // ---- final measures: worst elastic violation, slack cost.  Every elastic row keeps g.z + s - t = rhs along the whole
    // iteration (feasible start, ds - dt = -g.dz), so its residual rhs - g.z is s - t
    double viol = 0.0, scost = 0.0;
#pragma unroll
    for (int k = 0; k < NSL; ++k)
      if ((bact >> k) & 1u) viol = fmax(viol, bt[k] - bs[k]);
#pragma unroll
    for (int q = 0; q < OC_GSLOTS; ++q) {
      const int r = q * 64 + tid;
      if (r < NM) {
        if (!((gflag >> q) & 1u)) viol = fmax(viol, gt_[q] - gs_[q]);
        else { const double t = gl_[q] / aqs; scost += 0.5 * aqs * t * t; }
      }
    }
    viol = wave_max(viol); scost = wave_sum(scost);
    obj += scost;   // (obj is the quadratic objective of the final Z: recomputed by every update, and a node leaves the loop right after one or at its top)
    double* Zo = B.batch_Z + (size_t)node * N * NZ;
    for (int k = tid; k < N * 16; k += 64) { const int q = k & 15; if (q < NZ) Zo[(k >> 4) * NZ + oc_lcol<C, CM>(q)] = Z[k]; }
    if (tid == 0) {
      const int itc = it > QP_MAXIT ? QP_MAXIT : it;
      B.batch_obj[node] = obj; B.batch_viol[node] = viol; B.batch_ok[node] = ok;
      B.batch_bound[node] = (double)ncomp * comp + resid_fac * R0 * D[Y.d_misc + 2];   // (see ipm_kernel: complementarity + stationarity residual allowance)
      B.batch_it[node] = itc;
      atomicAdd((unsigned long long*)&B.inst_iters[inst], (unsigned long long)itc);
      atomicAdd((unsigned long long*)&B.inst_nodes[inst], 1ull);
      atomicAdd(B.stat_rowiters, rowiters);
#ifdef MIQP_PROFILE
      for (int q = 0; q < 10; ++q) atomicAdd(&B.prof[64 + q], ocp_[q]);
      atomicAdd(&B.prof[74], (unsigned long long)itc); atomicAdd(&B.prof[75], 1ull);
#endif
      if (B.stats) { atomicAdd(&B.stats[0], 1ull); atomicAdd(&B.stats[1], (unsigned long long)NM); atomicAdd(&B.stats[2], (unsigned long long)nbox); atomicAdd(&B.stats[4], (unsigned long long)ncoef);
                     atomicAdd(&B.stats[5], (unsigned long long)itc); atomicAdd(&B.stats[8 + NM / 32], 1ull); }
    }
  }
}

}  // namespace miqp
