// as_onchip.hip - dual active-set solve of a node relaxation, whole working set on chip (included by kernels.hip behind ipm_onchip.hip).
//
// What it replaces: the interior point of ipm_onchip_kernel<2, ...> for the ordinary nodes of a round (everything but the rounding probes, the
// local-search leaves and the marked large records, which stay with the larger interior point variant: their infeasible relaxations are
// re-rounded from the least-violation point only an elastic interior point delivers).  Region of the reference this stands for:
// cplex.solve()'s node re-solves, src/cplex_wrapper.cpp:158-185 (CPLEX re-solves a child by dual simplex pivots from the parent's basis).
//
// The node QP  min sum_i (z_i - r_i)' W (z_i - r_i)  s.t. dynamics, rows g.z <= h  has a Hessian that belongs to the OBJECTIVE ALONE: the same
// for every node of an instance, diagonal per stage, and the dynamics are 2C independent triple integrator chains.  So
//   * H^-1 (on the trajectories of the dynamics) applied to any stage-wise vector is one backward + one forward substitution per chain with the
//     CONSTANT gains of the unconstrained regulator (3 x 3 Riccati recursion per chain, once per node; no factorisation per iteration, no MFMA:
//     there is no dense contraction left - the work is O(N) per chain),
//   * Goldfarb-Idnani's dual method needs, per added row p: w = H^-1 g_p (one substitution), q = G_A w, the step directions from the inverse
//     M = (G_A H^-1 G_A')^-1 of the active rows' Schur complement (kept explicitly, one row per lane in registers, updated by symmetric rank-one
//     terms on add / drop), a ratio test over the multipliers, and the new iterate z = z_unc - H^-1 G' lambda (a second substitution).
// Every iterate is dual feasible: its dual value is a valid lower bound, rises monotonically and is tested against the incumbent cutoff after
// every step; an infeasible node shows as a dependent row without a blocking multiplier (or a multiplier beyond the exact penalty rho of the
// interior point's elastic rows).  Started cold from the unconstrained optimum the method takes |A| + 3..4 steps (tools/active_set_lab.py,
// profiles/r06_active_set_lab.txt: the parent's active set re-added row by row costs the same as the most-violated order finds it).
// A node the method cannot finish (64 active rows, step cap, loss of precision) is marked and returned unsolved; the larger interior point
// variant takes it in its concurrent launch of the next round (the same path as a node with more rows than the LDS block holds).
//
// Layout: the LDS block of ipm_onchip_kernel (oc_lds_layout: same decode, same capacities, 8 wavefronts per CU); the region of the decode
// scratch becomes the substitution vector V | the chain gains, the region of the box keys the feed-forward terms.
namespace miqp {

#ifdef MIQP_PROFILE
#define ASP_T(var) const long long var = clock64()
#define ASP_ACC(k, t0, t1) asp_[k] += (unsigned long long)((t1) - (t0))
#else
#define ASP_T(var)
#define ASP_ACC(k, t0, t1)
#endif
constexpr int AS_MAXSTEP = 220;        // adds + drops after which a node goes to the interior point instead
constexpr double AS_VTOL = 1.0e-8;     // a row is violated above this (rows are normalised: metres, m/s, ...)
constexpr double AS_DEP = 1.0e-8;
constexpr int AS_MT = 64;               // active sets of up to this many rows hand their M to the children (56 until the last day of round 6: the children of the 57 - 64 row nodes rebuilt theirs with a substitution per row)
constexpr int AS_MSTR = AS_MT * (AS_MT + 1) / 2;   // doubles per slot of the ring (packed triangle)      // curvature g P g' below this share of g H^-1 g': the row depends on the active ones

// Column order inside this kernel: CHAIN-CONTIGUOUS - (position, velocity, acceleration, jerk) of chain ch = 2 car + axis at 4 ch .. 4 ch + 3 (the
// transpose of ipm_onchip_kernel's chain-major 4 k + ch; an involution), so that the lane of a chain reads and writes its stage entries as two
// 16-byte LDS accesses
__device__ inline int as_col(int pq) { return ((pq & 3) << 2) | (pq >> 2); }

// GCAP = OC_GCAP: the ordinary nodes of a round (8 wavefronts per CU).  GCAP = OC_GCAP_BIG: the nodes known to be large before the round - rounding
// probes, local-search leaves, marked records - beside it on a third stream (4 wavefronts per CU), except the ones large_class() leaves to the
// interior point chain on the second stream (records the method failed on before or that exceed even this block, probes that may be re-rounded)
template <int C, int NSL, int GCAP = miqp::OC_GCAP>
__global__ void __launch_bounds__(64, (GCAP > 128 ? 1 : 2)) as_onchip_kernel(DevBuf B) {
  static_assert(C == 2, "chain-major columns of two cars");
  constexpr bool CM = true;
  constexpr bool BIG = GCAP > 128;
  constexpr int OC_GCAP = GCAP, OC_GSLOTS = OC_GCAP / 64, OC_GCOEF = oc_gcoef_of(OC_GCAP);
  constexpr int NX = 6 * C, NZ = 8 * C, NCH = 2 * C;
  const Layout& Y = B.Y;
  const int tid = threadIdx.x, lg = tid >> 4, lc = tid & 15;
  const int par = lg >> 1, side = lg & 1;
  const double bsgn = side ? -1.0 : 1.0;
  WAVE_DUMP(BIG ? 1 : 4);
  const int* const clist = BIG && B.cls_take ? B.cls_list + (size_t)(B.cls_take - 1) * B.batch_cap : nullptr;   // (the larger launch: the list of its class, nothing to scan)
  const int nbatch = clist ? (B.cls_count[B.cls_take - 1] < B.batch_cap ? B.cls_count[B.cls_take - 1] : B.batch_cap) : (*B.batch_count < B.batch_cap ? *B.batch_count : B.batch_cap);
  const int N = Y.N, NSLOT = Y.NSLOT;
  extern __shared__ double lds[];
  char* const L0 = (char*)lds;
  const OcLds LL = oc_lds_layout(N, Y.fixlen, OC_GCAP, !BIG);
  double* const Z = (double*)(L0 + LL.z);                   // [N][16] iterate
  double* const scr = (double*)(L0 + LL.u);                 // decode: dense scratch rows
  double* const V = (double*)(L0 + LL.u);                   // [N][16] right-hand side / result of a substitution
  unsigned long long* const bkey = (unsigned long long*)(L0 + LL.r);
  // (the regions of the decode scratch and of the box keys, contiguous, after the decode: V at the bottom, the gains and feed-forward terms at the
  // top, and between them room that - with V - stages the parent's M on its way from memory into the registers)
  const int SPAN = LL.gmeta - LL.u;                         // bytes of the two regions
  double* const kref = (double*)(L0 + LL.u + SPAN - N * 32);   // [N][4] feed-forward of the objective's linear term
  double* const kff = kref - N * 4;                         // [N][4] feed-forward of the current substitution
  double* const KS = kff - N * 16;                          // [N][4 chains][4]: K (p, v, a) and 1 / S_uu of the unconstrained regulator
  const int STG_CAP = (SPAN - N * 192) / 8;                 // doubles from V up to the gains
  uint4* const gmeta = (uint4*)(L0 + LL.gmeta);
  double* const gcoef = (double*)(L0 + LL.gcoef);
  double* const grhs = (double*)(L0 + LL.grhs);
  double* const Wd = (double*)(L0 + LL.wd);
  int* const sstart = (int*)(L0 + LL.sstart);
  unsigned short* const cand = (unsigned short*)(L0 + LL.cand);
  signed char* const fix = (signed char*)(L0 + LL.fix);
  __shared__ int sh_node;
  // the counters of the launches' statistics, per wavefront, added to the device's once at its end: ten adds per node to adjacent words were
  // 8.5 ms of the memory system's atomic unit per round of 87 k nodes (9.8 ns each, one after the other: tools/atomic_lab.hip) - as long as the launch itself
  __shared__ unsigned long long sh_stat[32];
  if (tid < 32) sh_stat[tid] = 0ull;
  const unsigned long long lt = (1ull << tid) - 1ull;

  // nodes are handed out in runs of `chunk` consecutive batch slots (a batch lists an instance's nodes next to each other: the nodes of a run
  // read the same instance tables - 28 KB, most of the decode's loads - from this CU's L1 instead of from L2)
  const int chunk = B.as_chunk > 0 ? B.as_chunk : 1;
  int run_next = 0, run_end = 0;
  const int quota = BIG ? 0 : B.as_quota; int handed = 0;
#ifdef MIQP_PROFILE   // residence of this wavefront: shader cycles and the 100 MHz clock; per launch the first start, the first wavefront out of work, the last end
  const unsigned long long wv_c0 = __builtin_amdgcn_s_memtime(), wv_r0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long wv_first = 0; int wv_nodes = 0;
  if (!BIG && tid == 0) wv_first = atomicMin(&B.prof[120], wv_r0);
#endif
  for (;;) {
    __syncthreads();
    if (run_next >= run_end) {
      if (quota > 0 && handed >= quota) break;
      ++handed;
      if (tid == 0) sh_node = atomicAdd(B.work_counter, chunk);
      __syncthreads();
      run_next = __builtin_amdgcn_readfirstlane(sh_node); run_end = run_next + chunk;
      if (run_next >= nbatch) break;
    }
    const int slot_ = run_next++;
    if (slot_ >= nbatch) { run_next = run_end; continue; }
    const int node = clist ? __builtin_amdgcn_readfirstlane(clist[slot_]) : slot_;
    {   // the split of the round between the launches: select_kernel's snapshot (large_class)
      const int cls = B.batch_large ? (int)B.batch_large[node] : (is_probe_word(B.batch_depth[node]) ? 1 : 0);
      if (cls != (BIG ? 1 : 0)) continue;
    }
    WAVE_DUMP_NODE();
    const int inst = __builtin_amdgcn_readfirstlane(B.batch_inst[node]);
    const double* D = B.inst_d + (size_t)inst * Y.dstride;
    const int* T = B.inst_i + (size_t)inst * Y.istride;
    const double ts = D[Y.d_glob + 7];
    const double aqs = 2.0 * D[Y.d_misc + 0];
    const double iaq = aqs > 0.0 ? 1.0 / aqs : 0.0;
    {
      const signed char* src = B.pool_fix + (size_t)B.batch_node[node] * Y.fixlen;
      for (int k = tid; k < Y.fixlen; k += 64) fix[k] = src[k];
      if (tid < 16) Wd[tid] = tid < NZ ? D[Y.d_wd + oc_lcol<C, CM>(as_col(tid))] : 0.0;
      for (int k = tid; k < N * 32; k += 64) bkey[k] = ~0ull;
      for (int k = tid; k <= N + 1; k += 64) sstart[k] = 0;
    }
    __syncthreads();
    const double* Rf = D + Y.d_ref;
    double cutoff = 1e300;
    {
      const double inc0 = fmin(inc_from_key(*(volatile unsigned long long*)&B.inc_key[inst]), B.inc_ext[inst]);
      // (a heuristic leaf - rounding probe, neighbour of the local search - is cut off at the incumbent itself, not a gap below it: a leaf that is
      // better by less than the gap IS the next incumbent, and the local search climbs in such steps.  The interior point never met this: it tests
      // the cutoff only once it is nearly stationary, and a leaf started from the incumbent's solution has converged by then)
      const double gap_ = is_probe_word(B.batch_depth[node]) ? 0.0 : B.inst_gap[inst];
      if (B.use_cutoff && inc0 < 1e300) cutoff = inc0 - gap_ * (1e-10 + fabs(inc0)) - B.inst_const[inst];
    }

#ifdef MIQP_PROFILE
    unsigned long long asp_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    ASP_T(ta0);
    // ---- decode (the two passes of ipm_onchip_kernel, unchanged: box rows to their (stage, side, column) key, general rows packed per stage)
    int ngen = 0;
    {
      unsigned long long* const bmp = (unsigned long long*)scr;
      const int nw = (N * NSLOT + 63) >> 6;
      unsigned short* const pre = (unsigned short*)(bmp + nw);
      for (int k = tid; k < nw; k += 64) bmp[k] = 0ull;
      OC_WAVE_SYNC();
      const int cls_off[8] = {0, 7, 11, 16, 16 + Y.EL, 16 + 5 * Y.EL, C * Y.SC, C * Y.SC + 8 * Y.NP};
      const int cls_cnt[8] = {7, 4, 5, Y.EL, 4 * Y.EL, 5 * Y.O, 8 * Y.NP, 16 * Y.NP};
      auto take = [&](int i, int slot) {
        if (decode_row<C, false>(Y, D, T, fix, i, slot, nullptr).active) {
          int col; double sg, rh;
          if (box_of_slot<C>(Y, D, T, fix, i, slot, col, sg, rh)) atomicMin(&bkey[(i * 2 + (sg < 0.0 ? 1 : 0)) * 16 + as_col(oc_pcol<C, CM>(col))], d2key(rh));
          else { const int pcode = i * NSLOT + slot; atomicOr(&bmp[pcode >> 6], 1ull << (pcode & 63)); }
        }
      };
      unsigned short* const plist = pre + ((nw + 1 + 3) & ~3);
      const int LCAP = (int)(((char*)(L0 + LL.r) - (char*)plist) / 2) - 64;
      int nlist = 0;
      auto flush = [&]() {
        OC_WAVE_SYNC();
        for (int j0 = 0; j0 < nlist; j0 += 64) if (j0 + tid < nlist) { const int pc = plist[j0 + tid]; const int i = pc / NSLOT; take(i, pc - i * NSLOT); }
        OC_WAVE_SYNC();
        nlist = 0;
      };
      // The state / input bounds (slots 0..10 of every car and stage: decode_row's cases rr < 11 with box_of_slot's right-hand sides) - three
      // quarters of a node's rows - one LANE PER (car, stage): its skip word, its region code and the eight box values (the region's table or the
      // hull of the region set) are loaded once for the eleven rows, instead of eleven walks through decode_row and box_of_slot in eight passes
      for (int e0 = 0; e0 < C * N; e0 += 64) {
        const int e = e0 + tid;
        if (e < C * N) {
          const int c = e / N, i = e - c * N;
          const double* G = D + Y.d_glob;
          const int code = i >= 1 ? (int)fix[Y.f_reg + c * N + i] : -1;
          const unsigned int skip = i >= 1 ? (unsigned int)T[Y.i_boxskip + c * N + i] : 0x7Fu;   // (stage 0: no state rows)
          const double* rt = code >= 0 ? D + Y.d_reg + (c * Y.P + (code >> 2)) * REGSZ : nullptr;
          const double* Hc = (!rt && i >= 1) ? region_hull(Y, D, T, fix, c, i) : nullptr;
          auto put = [&](int col, bool neg, double rh) { atomicMin(&bkey[(i * 2 + (neg ? 1 : 0)) * 16 + as_col(oc_pcol<C, CM>(col))], d2key(rh)); };
          if (i >= 1) {
            const double a_lo_x = rt ? rt[11] : Hc[0], a_hi_x = rt ? rt[12] : Hc[1], a_lo_y = rt ? rt[13] : Hc[2], a_hi_y = rt ? rt[14] : Hc[3];
            if (!(skip & 1u)) put(6 * c + 1, true, -G[0]);
            if (!(skip & 2u)) put(6 * c + 4, true, -G[0]);
            if (!(skip & 4u)) put(6 * c + 1, false, G[1]);
            if (!(skip & 8u)) put(6 * c + 2, false, a_hi_x);
            if (!(skip & 16u)) put(6 * c + 2, true, -a_lo_x);
            if (!(skip & 32u)) put(6 * c + 5, false, a_hi_y);
            if (!(skip & 64u)) put(6 * c + 5, true, -a_lo_y);
          }
          if (i <= N - 2) {
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) {
              double lo, hi;
              if (i == 0) { lo = D[Y.d_u0box + c * 4 + 2 * s_]; hi = D[Y.d_u0box + c * 4 + 2 * s_ + 1]; }
              else if (rt) { lo = rt[15 + 2 * s_]; hi = rt[16 + 2 * s_]; }
              else { lo = Hc[4 + 2 * s_]; hi = Hc[5 + 2 * s_]; }
              put(6 * C + 2 * c + s_, false, hi); put(6 * C + 2 * c + s_, true, -lo);
            }
          }
        }
      }
#pragma unroll 1
      for (int cl = 0; cl < 8; ++cl) {
#ifdef MIQP_PROFILE
        if (cl == 2) { const long long tq_ = clock64(); asp_[9] += (unsigned long long)(tq_ - ta0); }
#endif
        const int cnt = cls_cnt[cl], off = cls_off[cl];
        const bool percar = cl < 6;
        const int per = percar ? C * cnt : cnt, total = N * per;
        for (int e0 = 0; e0 < total; e0 += 64) {
          const int e = e0 + tid;
          int i = 0, slot = 0; bool in = e < total;
          if (in) { i = e / per; const int rem = e - i * per; slot = percar ? (rem / cnt) * Y.SC + off + rem % cnt : off + rem; }
          if (cl < 2) continue;   // (the velocity / acceleration / jerk bounds: decoded per (car, stage) below)
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
#ifdef MIQP_PROFILE
    const long long tq1_ = clock64();
#endif
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
            const int pq = as_col(oc_pcol<C, CM>(q));
            map |= (unsigned long long)(nn + 1) << (4 * pq); cols |= (unsigned int)pq << (4 * nn); nn++;
          }
        }
      }
      const bool keep = mine && nn > 0;
      const unsigned long long mk = __ballot(keep);
      const unsigned long long b0 = __ballot(keep && (nn & 1)), b1 = __ballot(keep && (nn & 2)), b2 = __ballot(keep && (nn & 4));
      const int tot = __popcll(b0) + 2 * __popcll(b1) + 4 * __popcll(b2);
      if (ncoef + tot > OC_GCOEF) { overflow = true; break; }
      if (keep) {
        const int idx = sstart[N + 1] + __popcll(mk & lt);
        const int off = ncoef + __popcll(b0 & lt) + 2 * __popcll(b1 & lt) + 4 * __popcll(b2 & lt);
#pragma unroll
        for (int k = 0; k < 6; ++k) if (k < nn) gcoef[off + k] = v6[k];
        uint4 m4; m4.x = (unsigned int)map; m4.y = (unsigned int)(map >> 32);
        m4.z = (unsigned int)off | ((unsigned int)nn << 16) | ((unsigned int)i << 20) | (r.aq > 0.0 ? 0x80000000u : 0u); m4.w = cols;
        gmeta[idx] = m4; grhs[idx] = r.rhs;
        cand[idx] = (unsigned short)(i * NSLOT + slot_);   // identity of the packed row (idx <= its position in the candidate list: in-place compaction)
      }
      OC_WAVE_SYNC();
      if (tid == 0) sstart[N + 1] += __popcll(mk);
      ncoef += tot;
      OC_WAVE_SYNC();
    }
    if (overflow) {   // marked and returned unsolved: next round the larger block takes the record - or, from there, the interior point chain
      if (B.batch_A) B.batch_A[(size_t)node * 64 + tid] = 0xFFFFu;
      if (tid == 0) {
        if (B.ring_M) B.batch_Mtag[node] = 0ull;
        B.batch_ok[node] = 5; B.pool_big[B.batch_node[node]] |= (BIG ? 2 : 1); if (B.stats) { atomicAdd(&B.stats[3], 1ull); atomicAdd(&B.stats[8 + (ngen >= 512 ? 15 : ngen / 32)], 1ull); }
      }
      continue;
    }
    const int NM = sstart[N + 1];
    OC_WAVE_SYNC();

    ASP_T(ta1); ASP_ACC(0, ta0, ta1);
#ifdef MIQP_PROFILE
    asp_[11] += (unsigned long long)(ta1 - tq1_);
#endif
    // ---- the box rows of this lane (column lc, side, stages 2 k + par) into registers
    double brhs[NSL];
    unsigned int bact = 0u, binA = 0u, ginA = 0u;
#pragma unroll
    for (int k = 0; k < NSL; ++k) {
      const int i = 2 * k + par;
      brhs[k] = 1e300;
      if (i < N && lc < NZ) {
        const unsigned long long key = bkey[(i * 2 + side) * 16 + lc];
        if (key != ~0ull) { brhs[k] = key2d(key); bact |= 1u << k; }
      }
    }
    OC_WAVE_SYNC();   // the keys are consumed: their region becomes the feed-forward terms

    // ---- constant gains of the unconstrained regulator, per chain (lane = chain): P' = Q + A' P A - K S_xu', K = S_xu / S_uu
    // (gains, the objective's feed-forward and the unconstrained optimum are the same for every node of an instance: host tables, host_inst.hpp::as_tables)
    const double h1 = ts, h2 = 0.5 * ts * ts, h3 = ts * ts * ts / 6.0;
    const double* const tb = D + Y.d_astab;
    for (int k = tid; k < N * 16; k += 64) KS[k] = tb[k];
    for (int k = tid; k < N * 4; k += 64) kref[k] = tb[N * 16 + k];
    OC_WAVE_SYNC();
    // one substitution: V holds a stage-wise vector v (zero above stage itop) on entry; `out` receives, for the stages up to iend, the minimiser of
    // 1/2 z' H z + v' z over the trajectories of the dynamics from x_0 = 0 (fromx0 false: a response, -H^-1 v), or the minimiser of the
    // objective + v' z from the true x_0 (fromx0 true: kref carries the objective's own linear term).  One lane per chain; the loads of a
    // stage are requested one stage ahead of their use
    typedef double d2_t __attribute__((ext_vector_type(2)));
    auto subst = [&](const bool fromx0, const int itop, const int iend, double* const out) {
      for (int k = tid; k < (N - 1) * 4; k += 64) if ((k >> 2) > itop) kff[k] = fromx0 ? kref[k] : 0.0;
      if (tid < NCH) {
        const int ch = tid;
        double p0 = 0.0, p1 = 0.0, p2 = 0.0;
        int i0 = itop;
        if (itop >= N - 1) { const d2_t a = *(const d2_t*)(V + (N - 1) * 16 + 4 * ch); p0 = a[0]; p1 = a[1]; p2 = V[(N - 1) * 16 + 4 * ch + 2]; i0 = N - 2; }
        if (i0 >= 0) {
          d2_t va = *(const d2_t*)(V + i0 * 16 + 4 * ch), vb = *(const d2_t*)(V + i0 * 16 + 4 * ch + 2);
          d2_t ka = *(const d2_t*)(KS + (i0 * 4 + ch) * 4), kb = *(const d2_t*)(KS + (i0 * 4 + ch) * 4 + 2);
          double kr = fromx0 ? kref[i0 * 4 + ch] : 0.0;
          for (int i = i0; i >= 0; --i) {
            const int j = i > 0 ? i - 1 : 0;
            const d2_t nva = *(const d2_t*)(V + j * 16 + 4 * ch), nvb = *(const d2_t*)(V + j * 16 + 4 * ch + 2);
            const d2_t nka = *(const d2_t*)(KS + (j * 4 + ch) * 4), nkb = *(const d2_t*)(KS + (j * 4 + ch) * 4 + 2);
            const double nkr = fromx0 ? kref[j * 4 + ch] : 0.0;
            const double su = vb[1] + h3 * p0 + h2 * p1 + h1 * p2;
            const double sx0 = va[0] + p0, sx1 = va[1] + h1 * p0 + p1, sx2 = vb[0] + h2 * p0 + h1 * p1 + p2;
            kff[i * 4 + ch] = fma(su, kb[1], kr);
            p0 = sx0 - ka[0] * su; p1 = sx1 - ka[1] * su; p2 = sx2 - kb[0] * su;
            va = nva; vb = nvb; ka = nka; kb = nkb; kr = nkr;
          }
        }
      }
      OC_WAVE_SYNC();
      if (tid < NCH) {
        const int ch = tid;
        double x0 = 0.0, x1 = 0.0, x2 = 0.0;
        if (fromx0) { const int c = ch >> 1, s = ch & 1; x0 = D[Y.d_x0 + 6 * c + 3 * s]; x1 = D[Y.d_x0 + 6 * c + 3 * s + 1]; x2 = D[Y.d_x0 + 6 * c + 3 * s + 2]; }
        const int ie = iend < N - 1 ? iend : N - 2;       // last stage with an input
        d2_t ka = *(const d2_t*)(KS + ch * 4), kb = *(const d2_t*)(KS + ch * 4 + 2);
        double kf = kff[ch];
        for (int i = 0; i <= ie; ++i) {
          const int j = i < N - 2 ? i + 1 : N - 2;
          const d2_t nka = *(const d2_t*)(KS + (j * 4 + ch) * 4), nkb = *(const d2_t*)(KS + (j * 4 + ch) * 4 + 2);
          const double nkf = kff[j * 4 + ch];
          const double u = -kf - (ka[0] * x0 + ka[1] * x1 + kb[0] * x2);
          d2_t oa, ob; oa[0] = x0; oa[1] = x1; ob[0] = x2; ob[1] = u;
          *(d2_t*)(out + i * 16 + 4 * ch) = oa; *(d2_t*)(out + i * 16 + 4 * ch + 2) = ob;
          const double n0 = x0 + h1 * x1 + h2 * x2 + h3 * u, n1 = x1 + h1 * x2 + h2 * u, n2 = x2 + h1 * u;
          x0 = n0; x1 = n1; x2 = n2;
          ka = nka; kb = nkb; kf = nkf;
        }
        if (iend >= N - 1 || ie == N - 2) { d2_t oa, ob; oa[0] = x0; oa[1] = x1; ob[0] = x2; ob[1] = 0.0; *(d2_t*)(out + (ie + 1) * 16 + 4 * ch) = oa; *(d2_t*)(out + (ie + 1) * 16 + 4 * ch + 2) = ob; }
      }
      OC_WAVE_SYNC();
    };
    // value of a row at a stage-wise vector: box rows id = (stage * 2 + side) * 16 + column, general rows id = 1024 + index
    auto row_dot = [&](int id, const double* vec) -> double {
      if (id < 1024) return (((id >> 4) & 1) ? -1.0 : 1.0) * vec[(id >> 5) * 16 + (id & 15)];
      const uint4 m4 = gmeta[id - 1024];
      const int off = (int)(m4.z & 0xFFFFu), nn = (int)((m4.z >> 16) & 7u), i = (int)((m4.z >> 20) & 0x7FFu);
      double a = 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k) if (k < nn) a += gcoef[off + k] * vec[i * 16 + ((m4.w >> (4 * k)) & 15u)];
      return a;
    };
    auto row_soft = [&](int id) -> bool { return id >= 1024 && (gmeta[id - 1024].z & 0x80000000u) != 0u; };
    auto zeroV = [&]() { for (int k = tid; k < N * 16; k += 64) V[k] = 0.0; OC_WAVE_SYNC(); };
    auto objective = [&]() -> double {
      double o = 0.0;
      for (int k = tid; k < N * 16; k += 64) { const int q = k & 15; if (q < NZ) { const double d = Z[k] - Rf[(k >> 4) * NZ + oc_lcol<C, CM>(as_col(q))]; o += Wd[q] * d * d; } }
      return wave_sum(o);
    };

    // ---- the active set: slot a lives in lane a (its row, its multiplier, row a of M)
    int arow = -1, astage = -1; double alam = 0.0, arhs = 0.0;   // (arhs: the row's right-hand side, at hand in its slot lane)
    double M[64];
#pragma unroll
    for (int b = 0; b < 64; ++b) M[b] = 0.0;
    unsigned long long used = 0ull;
    // symmetric rank-one term M += alpha u u' over the columns of `mask` (u_b = 0 elsewhere)
    auto rank1 = [&](const double u, const double alpha, const unsigned long long mask) {
#pragma unroll
      for (int g8 = 0; g8 < 8; ++g8) {
        if ((mask >> (8 * g8)) & 0xFFull) {
#pragma unroll
          for (int j = 0; j < 8; ++j) { const int b = 8 * g8 + j; M[b] = fma(u * readlane_d(u, b), alpha, M[b]); }
        }
      }
    };
    // iterate of the current multipliers: z = argmin objective + lambda' G z
    auto refresh = [&]() {
      zeroV();
      // lambda' G, stage by stage, in FIXED POINT (2^-38, 64-bit integer LDS atomics): several rows can meet in one entry, the order in which the
      // LDS serves colliding atomic adds is not ours to fix, and a floating point sum that depends on it can flip a tie of the ratio test (repeated
      // solves of one instance differed by one step in 5 of 8 runs) - integer sums do not depend on the order.  What the rounding (4e-12 absolute)
      // changes is the linear term the iterate minimises, far inside the allowance of the bound; a multiplier large enough to overflow the sum
      // (rho x a coefficient of ~20 is 2e6 of the 3e7 that fit) is beyond the exact penalty, where the node counts as infeasible anyway
      const double FX = 274877906944.0, IFX = 1.0 / 274877906944.0;   // 2^38
      if (arow >= 0 && alam != 0.0) {
        if (arow < 1024) atomicAdd((unsigned long long*)&V[(arow >> 5) * 16 + (arow & 15)], (unsigned long long)(long long)__double2ll_rn((((arow >> 4) & 1) ? -alam : alam) * FX));
        else {
          const uint4 m4 = gmeta[arow - 1024];
          const int off = (int)(m4.z & 0xFFFFu), nn = (int)((m4.z >> 16) & 7u), i = (int)((m4.z >> 20) & 0x7FFu);
#pragma unroll
          for (int k = 0; k < 6; ++k) if (k < nn) atomicAdd((unsigned long long*)&V[i * 16 + ((m4.w >> (4 * k)) & 15u)], (unsigned long long)(long long)__double2ll_rn(gcoef[off + k] * alam * FX));
        }
      }
      OC_WAVE_SYNC();
      for (int k = tid; k < N * 16; k += 64) V[k] = (double)(long long)__double_as_longlong(V[k]) * IFX;
      const int amax = (int)wave_max((double)astage);
      OC_WAVE_SYNC();
      subst(true, amax, N - 1, Z);
    };
    for (int k = tid; k < N * 16; k += 64) Z[k] = tb[N * 20 + k];   // lambda = 0: the unconstrained optimum
    OC_WAVE_SYNC();
    ASP_T(ta2); ASP_ACC(1, ta1, ta2);
    int steps = 0, ndrop = 0, ok = 1, nwarm = 0, nfast = 0, ncold = 0;
    bool infeas = false, fail = false;
    int why = 0;   // (diagnostic) what stopped a node the method could not finish: 1 no free slot, 2 step cap, 3 curvature / S_pp, 4 pivot of a down-date, 5 rows off their equalities at the end, 6 start
    // ---- warm start: the rows that were active at the parent's optimum (their identities travel with the record: box rows by their key, general
    // rows by their (stage, slot) code) are taken as the first active set where this node still has them, in the parent's slots.  M = S_A0^-1
    // comes from the parent as well (packed triangle over its slots in rank order, copied per child by eval_kernel; a row this node no longer
    // has leaves by the usual rank-one down-date) - or, where the record carries none, is rebuilt: S_A0 = G_A0 H^-1 G_A0' column by column (one
    // response each) and inverted in the registers by symmetric sweeps.  Multipliers of the equality-constrained problem on A0 that come out
    // negative leave one at a time (the rows the branching made slack), and Goldfarb-Idnani goes on from that dual feasible point.  Right-hand
    // sides are this node's own; that the inherited M still fits the rows is CHECKED (the active rows must hold with equality at the first
    // iterate), else the node starts cold.
    if (B.pool_A && (B.batch_depth[node] >> 6) >= 1 && B.batch_node[node] < B.z_cap) {
      const int rec = B.batch_node[node];
      const unsigned int pa = B.pool_A[(size_t)rec * 64 + tid];
      int cid = -1;
      int ownl = 0;
      if (pa < 1024u) ownl = (int)((((pa >> 5) & 1u) * 2u + ((pa >> 4) & 1u)) << 4 | (pa & 15u));
      const unsigned int ob = (unsigned int)__shfl((int)bact, ownl);
      if (pa < 1024u) { if ((ob >> (pa >> 6)) & 1u) cid = (int)pa; }
      else if (pa != 0xFFFFu) {
        const int pc = (int)pa - 1024;
        int lo = 0, hi = NM;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if ((int)cand[mid] < pc) lo = mid + 1; else hi = mid; }
        if (lo < NM && (int)cand[lo] == pc) cid = 1024 + lo;
      }
      const unsigned long long u0p = __ballot(pa != 0xFFFFu);   // the parent's slots
      const unsigned long long u0 = __ballot(cid >= 0);        // ... that this node still has
      if (u0) {
        arow = cid;
        astage = cid < 0 ? -1 : (cid < 1024 ? (cid >> 5) : (int)((gmeta[cid - 1024].z >> 20) & 0x7FFu));
        // the parent's M lives in a ring of slots that its solve wrote and nobody frees: the record carries (slot, tag), and the slot still holds
        // that M as long as the tag in the ring is the record's (a ring of millions of slots against ~15 k writes per round: seconds of history;
        // a node selected later than that rebuilds M below)
        int nmp = 0; unsigned long long moff = 0ull;
        if (B.ring_M) {
          const unsigned long long tg = B.pool_Mtag[rec];
          const unsigned long long c_ = tg >> 8;
          // (still whole: nothing allocated since can have come round to it - the head only grows, by less than the margin during one launch)
          if (tg != 0ull && *(volatile unsigned long long*)B.ring_head - c_ < B.ring_doubles - B.ring_margin) { nmp = (int)(tg & 255ull); moff = c_ % B.ring_doubles; }
        }
        nmp = __builtin_amdgcn_readfirstlane(nmp);
        if (B.as_stats && tid == 0 && !(nmp > 0 && nmp == __popcll(u0p))) sh_stat[16 + (B.ring_M == nullptr ? 0 : (B.pool_Mtag[rec] == 0ull ? 1 : (nmp == 0 ? 2 : 3)))] += 1ull;   // (diagnostic) why M is rebuilt: no ring, the parent left none, the ring has come round, another row count
        if (nmp > 0 && nmp == __popcll(u0p)) {
          // the parent's M: entry (a, b) of the packed triangle at tri(max rank) + min rank.  The triangle comes in with contiguous loads, all in
          // flight at once, into LDS; the lanes pick their rows there (loads of single entries, one wait each, cost 4 x the rest of a node in a
          // long stream, where the records of a queue spread over tens of GB)
          const double* const rm = B.ring_M; const unsigned long long rd_ = B.ring_doubles;
          const int ra = __popcll(u0p & lt);
          const bool mine = (u0p >> tid) & 1ull;
          const int len = nmp * (nmp + 1) / 2;
          for (int c0 = 0; c0 < len; c0 += STG_CAP) {   // (the staging room holds 800 doubles: a triangle of more than 39 rows comes in two or three parts)
            const int c1 = c0 + STG_CAP < len ? c0 + STG_CAP : len;
            OC_WAVE_SYNC();
            for (int k = c0 + tid; k < c1; k += 64) { unsigned long long q_ = moff + (unsigned long long)k; if (q_ >= rd_) q_ -= rd_; V[k - c0] = rm[q_]; }
            OC_WAVE_SYNC();
#pragma unroll
            for (int b8 = 0; b8 < 64; ++b8) {
              if ((u0p >> b8) & 1ull) {
                const int rb = __popcll(u0p & ((1ull << b8) - 1ull));
                const int hi_ = ra > rb ? ra : rb, lo_ = ra > rb ? rb : ra;
                const int ix = hi_ * (hi_ + 1) / 2 + lo_;
                if (mine && ix >= c0 && ix < c1) M[b8] = V[ix - c0];
              }
            }
          }
          OC_WAVE_SYNC();
          used = u0p;
          for (unsigned long long um = u0p & ~u0; um; um &= um - 1ull) {   // rows this node no longer has
            const int kd = __builtin_amdgcn_readfirstlane(__ffsll((long long)um) - 1);
            double m = 0.0;
#pragma unroll
            for (int b = 0; b < 64; ++b) m = (b == kd) ? M[b] : m;
            const double mkk = readlane_d(m, kd);
            if (mkk > 0.0) rank1(m, -1.0 / mkk, used);
#pragma unroll
            for (int b = 0; b < 64; ++b) if (b == kd || tid == kd) M[b] = 0.0;
            used &= ~(1ull << kd);
          }
          nfast = 1;
        } else {
          used = u0;
          const int amax0 = (int)wave_max((double)astage);
          // S column by column
          double sdiag = 1.0;
          for (unsigned long long um = u0; um; um &= um - 1ull) {
            const int a = __builtin_amdgcn_readfirstlane(__ffsll((long long)um) - 1);
            const int ida = __builtin_amdgcn_readlane(arow, a);
            zeroV();
            if (ida < 1024) { if (tid == 0) V[(ida >> 5) * 16 + (ida & 15)] = ((ida >> 4) & 1) ? -1.0 : 1.0; }
            else {
              const uint4 m4 = gmeta[ida - 1024];
              const int off = (int)(m4.z & 0xFFFFu), nn = (int)((m4.z >> 16) & 7u), i = (int)((m4.z >> 20) & 0x7FFu);
              if (tid < nn) V[i * 16 + ((m4.w >> (4 * tid)) & 15u)] = gcoef[off + tid];
            }
            const int sta = __builtin_amdgcn_readlane(astage, a);
            OC_WAVE_SYNC();
            subst(false, sta, amax0, V);
            double sv = arow >= 0 ? -row_dot(arow, V) : 0.0;
            if (tid == a && row_soft(ida)) sv += iaq;
            if (tid == a) sdiag = sv;
#pragma unroll
            for (int b = 0; b < 64; ++b) M[b] = (b == a) ? sv : M[b];
          }
          // in-place inverse by symmetric sweeps (S is positive definite on independent rows; a pivot that has gone flat - a dependent
          // row - takes its row out of the start)
          for (unsigned long long um = u0; um; um &= um - 1ull) {
            const int k = __builtin_amdgcn_readfirstlane(__ffsll((long long)um) - 1);
            double m = 0.0;
#pragma unroll
            for (int b = 0; b < 64; ++b) m = (b == k) ? M[b] : m;
            const double d = readlane_d(m, k);
            if (!(d > AS_DEP * readlane_d(sdiag, k))) {   // dependent on the rows swept so far (the pivot against the row's own g H^-1 g'): out of the start
#pragma unroll
              for (int b = 0; b < 64; ++b) if (b == k || tid == k) M[b] = 0.0;
              if (tid == k) { arow = -1; astage = -1; }
              used &= ~(1ull << k);
              continue;
            }
            const double invd = 1.0 / d;
#pragma unroll
            for (int g8 = 0; g8 < 8; ++g8) {
              if ((u0 >> (8 * g8)) & 0xFFull) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { const int b = 8 * g8 + j; const double mb = readlane_d(m, b); M[b] = (tid == k) ? M[b] * invd : fma(m * mb, -invd, M[b]); }
              }
            }
#pragma unroll
            for (int b = 0; b < 64; ++b) M[b] = (b == k) ? ((tid == k) ? -invd : m * invd) : M[b];
          }
#pragma unroll
          for (int b = 0; b < 64; ++b) M[b] = (arow < 0) ? 0.0 : -M[b];
        }
        // multipliers of the equality-constrained problem on the start; negative ones leave one at a time
        {   // right-hand sides into the slot lanes (a box row's: in the registers of its owner lane)
          int ol = 0, ok_ = -1;
          if (arow >= 0 && arow < 1024) { ol = (((arow >> 5) & 1) * 2 + ((arow >> 4) & 1)) << 4 | (arow & 15); ok_ = arow >> 6; }
          double rsel = arow >= 1024 ? grhs[arow - 1024] : 0.0;
#pragma unroll
          for (int k = 0; k < NSL; ++k) {   // every lane offers its slot-k right-hand side; the slot lanes pick the one they need
            const double rk = __shfl(brhs[k], ol);
            if (ok_ == k) rsel = rk;
          }
          arhs = rsel;
        }
        double cv = arow >= 0 ? row_dot(arow, Z) - arhs : 0.0;
        for (int guard = 0; guard < 64; ++guard) {
          double l0 = 0.0;
#pragma unroll
          for (int g8 = 0; g8 < 8; ++g8) {
            if ((used >> (8 * g8)) & 0xFFull) {
#pragma unroll
              for (int j = 0; j < 8; ++j) { const int b = 8 * g8 + j; l0 = fma(M[b], readlane_d(cv, b), l0); }
            }
          }
          if (arow < 0) l0 = 0.0;
          alam = l0;
          const double lmin = wave_min(arow >= 0 ? l0 : 1e300);
          if (!(lmin < -1e-10)) break;
          const int kd = __builtin_amdgcn_readfirstlane(__ffsll((long long)__ballot(arow >= 0 && l0 == lmin)) - 1);
          double m = 0.0;
#pragma unroll
          for (int b = 0; b < 64; ++b) m = (b == kd) ? M[b] : m;
          const double mkk = readlane_d(m, kd);
          if (!(mkk > 0.0)) { fail = true; break; }
          rank1(m, -1.0 / mkk, used);
#pragma unroll
          for (int b = 0; b < 64; ++b) if (b == kd || tid == kd) M[b] = 0.0;
          if (tid == kd) { arow = -1; astage = -1; alam = 0.0; cv = 0.0; }
          used &= ~(1ull << kd);
          ndrop++;
        }
        if (alam < 0.0) alam = 0.0;
        nwarm = __popcll(used);
        refresh();
        // the start is a point of the method only if its rows hold with equality (an inherited M that no longer fits the rows, or one that
        // has lost its precision over the generations, shows here): else the node starts cold
        double resid = arow >= 0 ? fabs(row_dot(arow, Z) - arhs - ((arow >= 1024 && row_soft(arow)) ? alam * iaq : 0.0)) : 0.0;
        resid = wave_max(resid);
        if (!(resid <= 1e-7) || fail) {   // cold
          fail = false;
          arow = -1; astage = -1; alam = 0.0; arhs = 0.0; used = 0ull; nwarm = 0; nfast = 0; ncold = 1;
#pragma unroll
          for (int b = 0; b < 64; ++b) M[b] = 0.0;
          refresh();
        } else {
          // the owner lanes learn which of their rows are active (one LDS word per owner lane, in the spent region of the fix record)
          unsigned int* const inab = (unsigned int*)fix;
          inab[tid] = 0u;
          OC_WAVE_SYNC();
          if (arow >= 0) {
            if (arow < 1024) atomicOr(&inab[(((arow >> 5) & 1) * 2 + ((arow >> 4) & 1)) << 4 | (arow & 15)], 1u << (arow >> 6));
            else { const int rr_ = arow - 1024; atomicOr(&inab[rr_ & 63], 1u << (16 + (rr_ >> 6))); }
          }
          OC_WAVE_SYNC();
          { const unsigned int w_ = inab[tid]; binA = w_ & 0xFFFFu; ginA = w_ >> 16; }
        }
      }
    }
    ASP_T(ta3); ASP_ACC(8, ta2, ta3);
    // dual value of the current multipliers: at a point where the active rows hold with equality it is the primal value of the iterate
    double Dval = objective() + wave_sum((arow >= 1024 && row_soft(arow)) ? 0.5 * alam * alam * iaq : 0.0);

    for (; !fail;) {
      // ---- most violated row at the iterate (rows of the active set hold with equality)
      ASP_T(tb0);
      double bv = -1e300, brh = 0.0; int bid = -1;
#pragma unroll
      for (int k = 0; k < NSL; ++k) {
        const int i = 2 * k + par;
        if (((bact >> k) & 1u) && !((binA >> k) & 1u)) { const double v = bsgn * Z[i * 16 + lc] - brhs[k]; if (v > bv) { bv = v; brh = brhs[k]; bid = (i * 2 + side) * 16 + lc; } }
      }
#pragma unroll
      for (int q = 0; q < OC_GSLOTS; ++q) {
        const int r = q * 64 + tid;
        if (r < NM && !((ginA >> q) & 1u)) { const double rh_ = grhs[r]; const double v = row_dot(1024 + r, Z) - rh_; if (v > bv) { bv = v; brh = rh_; bid = 1024 + r; } }
      }
      const double vmax = wave_max(bv);
      if (!(vmax > AS_VTOL)) break;                                   // optimal
      if (steps >= AS_MAXSTEP || used == ~0ull || !(vmax < 1e290)) { fail = true; why = used == ~0ull ? 1 : 2; break; }
      const int src = __builtin_amdgcn_readfirstlane(__ffsll((long long)__ballot(bv == vmax)) - 1);
      const int pid = __builtin_amdgcn_readlane(bid, src);
      const bool psoft = row_soft(pid);
      const double prhs = readlane_d(brh, src);
      // ---- w = H^-1 g_p (V), q = G_A w, S_pp
      ASP_T(tb1); ASP_ACC(2, tb0, tb1);
      zeroV();
      if (pid < 1024) { if (tid == 0) V[(pid >> 5) * 16 + (pid & 15)] = ((pid >> 4) & 1) ? -1.0 : 1.0; }
      else {
        const uint4 m4 = gmeta[pid - 1024];
        const int off = (int)(m4.z & 0xFFFFu), nn = (int)((m4.z >> 16) & 7u), i = (int)((m4.z >> 20) & 0x7FFu);
        if (tid < nn) V[i * 16 + ((m4.w >> (4 * tid)) & 15u)] = gcoef[off + tid];
      }
      const int pstage = pid < 1024 ? (pid >> 5) : (int)((gmeta[pid - 1024].z >> 20) & 0x7FFu);
      const int amax_ = (int)wave_max((double)astage);
      OC_WAVE_SYNC();
      subst(false, pstage, amax_ > pstage ? amax_ : pstage, V);      // V = -H^-1 g_p (up to the last stage that carries an active row)
      ASP_T(tb2); ASP_ACC(3, tb1, tb2);
      double q = arow >= 0 ? -row_dot(arow, V) : 0.0;
      const double Spp = -row_dot(pid, V) + (psoft ? iaq : 0.0);
      double vp = vmax, lp = 0.0;
      steps++;
      ASP_T(tb3); ASP_ACC(4, tb2, tb3);
      for (;;) {
        // r = M q, curvature of the new row against the active ones, ratio test on the multipliers
        double r = 0.0;
#pragma unroll
        for (int g8 = 0; g8 < 8; ++g8) {
          if ((used >> (8 * g8)) & 0xFFull) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const int b = 8 * g8 + j; r = fma(M[b], readlane_d(q, b), r); }
          }
        }
        if (arow < 0) r = 0.0;
        const double curv = Spp - wave_sum(q * r);
        const double ratio = (arow >= 0 && r > 1e-13) ? alam / r : 1e300;
        const double td = wave_min(ratio);
        const int kd = td < 1e299 ? __builtin_amdgcn_readfirstlane(__ffsll((long long)__ballot(ratio == td)) - 1) : -1;
        if (!(curv == curv) || !(Spp > 0.0)) { fail = true; why = 3; break; }
        const bool dep = !(curv > AS_DEP * Spp);
        double t; bool full;
        if (dep) { if (kd < 0) { infeas = true; break; } t = td; full = false; }
        else { const double tp = vp / curv; if (tp <= td) { t = tp; full = true; } else { t = td; full = false; } }
        if (arow >= 0) { alam = fma(-t, r, alam); if (alam < 0.0) alam = 0.0; }
        lp += t;
        Dval += t * (vp - 0.5 * t * (dep ? 0.0 : curv));
        if (lp > RHO_EL) { infeas = true; break; }                    // (the exact penalty of the interior point's elastic rows: beyond rho the row gives way)
        if (full) {
          const int s = __builtin_amdgcn_readfirstlane(__ffsll((long long)~used) - 1);
          double u = arow >= 0 ? r : 0.0;
          if (tid == s) u = -1.0;
          const unsigned long long nmask = used | (1ull << s);
          rank1(u, 1.0 / curv, nmask);
          if (tid == s) { arow = pid; alam = lp; astage = pstage; arhs = prhs; }
          used = nmask;
          if (pid < 1024) { if (tid == ((((pid >> 5) & 1) * 2 + ((pid >> 4) & 1)) << 4 | (pid & 15))) binA |= 1u << (pid >> 6); }
          else { const int rr_ = pid - 1024; if (tid == (rr_ & 63)) ginA |= 1u << (rr_ >> 6); }
          break;
        }
        // partial step: the blocking row leaves the active set
        vp -= t * (dep ? 0.0 : curv);
        const int idk = __builtin_amdgcn_readlane(arow, kd);
        if (idk < 1024) { if (tid == ((((idk >> 5) & 1) * 2 + ((idk >> 4) & 1)) << 4 | (idk & 15))) binA &= ~(1u << (idk >> 6)); }
        else { const int rr_ = idk - 1024; if (tid == (rr_ & 63)) ginA &= ~(1u << (rr_ >> 6)); }
        double m = 0.0;
#pragma unroll
        for (int b = 0; b < 64; ++b) m = (b == kd) ? M[b] : m;
        const double mkk = readlane_d(m, kd);
        if (!(mkk > 0.0)) { fail = true; why = 4; break; }
        rank1(m, -1.0 / mkk, used);
#pragma unroll
        for (int b = 0; b < 64; ++b) if (b == kd || tid == kd) M[b] = 0.0;
        if (tid == kd) { arow = -1; astage = -1; alam = 0.0; q = 0.0; }
        used &= ~(1ull << kd);
        ndrop++; steps++;
        if (steps >= AS_MAXSTEP) { fail = true; why = 2; break; }
      }
      ASP_T(tb4); ASP_ACC(5, tb3, tb4);
      if (infeas || fail) break;
      refresh();
      ASP_T(tb5); ASP_ACC(6, tb4, tb5);
      if (Dval - 1e-9 * fabs(Dval) > cutoff + 1e-9 * fabs(cutoff)) { ok = 2; break; }
    }
    ASP_T(tc0);
    // ---- results.  obj: primal value of the iterate (objective + cost of the soft rows' slack lambda / a); viol: worst row at the iterate;
    // bound allowance: obj - L(z, lambda) (the Lagrangian at the iterate is a lower bound of the relaxation for any lambda >= 0)
    double viol = 0.0, scost = 0.0, lres = 0.0;
    if (!fail && !infeas && ok == 1) {
#pragma unroll
      for (int k = 0; k < NSL; ++k) { const int i = 2 * k + par; if ((bact >> k) & 1u) viol = fmax(viol, bsgn * Z[i * 16 + lc] - brhs[k]); }
#pragma unroll
      for (int q = 0; q < OC_GSLOTS; ++q) { const int r = q * 64 + tid; if (r < NM && !(gmeta[r].z & 0x80000000u)) viol = fmax(viol, row_dot(1024 + r, Z) - grhs[r]); }
      if (arow >= 0) {
        const bool sf = arow >= 1024 && row_soft(arow);
        const double res = row_dot(arow, Z) - arhs - (sf ? alam * iaq : 0.0);
        if (sf) { scost = 0.5 * alam * alam * iaq; viol = fmax(viol, res); }
        lres = alam * res;
      }
      viol = wave_max(viol); scost = wave_sum(scost);
      lres = fabs(wave_sum(lres));
    } else if (infeas) viol = 1.0;
    if (!fail && !infeas && ok == 1 && !(viol <= 5.0e-7)) { fail = true; why = 5; }   // (the rows of the active set drifted off their equalities: not a result)
    // (as_probe_first) a rounding probe that turns out infeasible while the re-rounding applies: the interior point solves it again, for its least-violation point
    if (BIG && infeas && B.as_probe_first && B.pump_max > 0 && is_probe_word(B.batch_depth[node]) && (cutoff > 1e299 || B.pump_inc)) fail = true;
    if (fail) {
      // marked and returned unsolved: the interior point chain takes the record next round (large_class 2)
      if (B.batch_A) B.batch_A[(size_t)node * 64 + tid] = 0xFFFFu;
      if (B.ring_M && tid == 0) B.batch_Mtag[node] = 0ull;
      if (tid == 0) { B.batch_ok[node] = 5; B.pool_big[B.batch_node[node]] |= 4; if (B.as_stats) { sh_stat[2] += 1ull; sh_stat[10 + (why < 6 ? why : 5)] += 1ull; } }
      continue;
    }
    const double obj = (infeas || ok == 2) ? Dval : objective() + scost;
    double* Zo = B.batch_Z + (size_t)node * N * NZ;
    for (int k = tid; k < N * 16; k += 64) { const int q = k & 15; if (q < NZ) Zo[(k >> 4) * NZ + oc_lcol<C, CM>(as_col(q))] = Z[k]; }
    if (B.batch_A) {   // the final active set, by row identity, for the children's starts
      unsigned short enc = 0xFFFFu;
      if (!fail && !infeas && ok == 1 && arow >= 0) enc = arow < 1024 ? (unsigned short)arow : (unsigned short)(1024 + (int)cand[arow - 1024]);
      B.batch_A[(size_t)node * 64 + tid] = enc;
    }
    if (B.ring_M) {   // ... and its M, packed over the slots in rank order, into the next slot of the ring (staged in LDS - the iterate, the vector and the gains are spent - so that the stores are contiguous); the children inherit (slot, tag)
      const int n_ = __popcll(used);
      unsigned long long tg_ = 0ull;
      OC_WAVE_SYNC();
      if (!fail && !infeas && ok == 1 && n_ >= 1 && n_ <= AS_MT && n_ * (n_ + 1) * 4 <= LL.total) {   // (... and the packed triangle fits the LDS block it is staged in)
        double* const stg = (double*)L0;
        const int ra = __popcll(used & lt);
#pragma unroll
        for (int b8 = 0; b8 < 64; ++b8) {
          if ((used >> b8) & 1ull) {
            const int rb = __popcll(used & ((1ull << b8) - 1ull));
            if (arow >= 0 && ra >= rb) stg[ra * (ra + 1) / 2 + rb] = M[b8];
          }
        }
        const int len = n_ * (n_ + 1) / 2;
        unsigned long long cnt_ = 0ull;
        if (tid == 0) cnt_ = atomicAdd(B.ring_head, (unsigned long long)len);
        cnt_ = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(cnt_ >> 32)) << 32) | (unsigned int)__builtin_amdgcn_readfirstlane((int)cnt_);
        tg_ = (cnt_ << 8) | (unsigned long long)n_;
        OC_WAVE_SYNC();
        const unsigned long long o_ = cnt_ % B.ring_doubles, rd_ = B.ring_doubles;
        for (int k = tid; k < len; k += 64) { unsigned long long q_ = o_ + (unsigned long long)k; if (q_ >= rd_) q_ -= rd_; B.ring_M[q_] = stg[k]; }
      }
      if (tid == 0) B.batch_Mtag[node] = tg_;
    }
    if (tid == 0) {
      B.batch_obj[node] = obj; B.batch_viol[node] = viol; B.batch_ok[node] = ok;
      B.batch_bound[node] = lres + 1e-9 * (1.0 + fabs(obj));
      B.batch_it[node] = steps;
      atomicAdd((unsigned long long*)&B.inst_nodes[inst], 1ull);
      atomicAdd((unsigned long long*)&B.inst_iters[inst], (unsigned long long)steps);
#ifdef MIQP_PROFILE
      ++wv_nodes;
      { const long long tc1 = clock64(); asp_[7] += (unsigned long long)(tc1 - tc0);
        for (int q_ = 0; q_ < 9; ++q_) atomicAdd(&B.prof[80 + q_], asp_[q_]);
        atomicAdd(&B.prof[92], asp_[9]); atomicAdd(&B.prof[93], asp_[11]);
        atomicAdd(&B.prof[90], 1ull); atomicAdd(&B.prof[91], (unsigned long long)steps); }
#endif
      if (B.as_stats) {
        sh_stat[0] += 1ull; sh_stat[1] += (unsigned long long)steps; sh_stat[3] += (unsigned long long)ndrop;
        sh_stat[4] += infeas ? 1ull : 0ull; sh_stat[5] += ok == 2 ? 1ull : 0ull; sh_stat[6] += (unsigned long long)__popcll(used); sh_stat[7] += (unsigned long long)nwarm; sh_stat[8] += (unsigned long long)nfast; sh_stat[9] += (unsigned long long)ncold;
        if (BIG) { const int og_ = B.pool_origin ? (int)B.pool_origin[B.batch_node[node]] : 0; sh_stat[20] += 1ull; sh_stat[21] += (unsigned long long)steps; if (og_ == 14) { sh_stat[22] += 1ull; sh_stat[23] += (unsigned long long)steps; }
          sh_stat[24 + (ngen <= 64 ? 0 : ngen <= 96 ? 1 : ngen <= 128 ? 2 : ngen <= 192 ? 3 : 4)] += 1ull; if (is_probe_word(B.batch_depth[node]) && og_ != 14) sh_stat[29] += 1ull; }   // (the larger block: its nodes and steps, the local search's leaves among them)
      }
    }
  }
  __syncthreads();
  if (B.as_stats && tid < 32 && sh_stat[tid] != 0ull) atomicAdd(&B.as_stats[tid], sh_stat[tid]);
#ifdef MIQP_PROFILE
  if (!BIG && tid == 0) {
    const unsigned long long wv_c1 = __builtin_amdgcn_s_memtime(), wv_r1 = __builtin_amdgcn_s_memrealtime();
    atomicAdd(&B.prof[94], wv_c1 - wv_c0); atomicAdd(&B.prof[95], wv_r1 - wv_r0); atomicAdd(&B.prof[96], 1ull);
    atomicMin(&B.prof[122], wv_r1); atomicMax(&B.prof[121], wv_r1);
    // start of this wavefront after the launch's first (100 MHz ticks): < 0.05, 0.2, 0.5, 1, 2, 4, 8 ms, later; nodes it solved: 0, <= 4, <= 16, <= 32, <= 64, more
    const unsigned long long dl = wv_r0 > wv_first ? wv_r0 - wv_first : 0ull;
    const int hb = dl < 5000 ? 0 : dl < 20000 ? 1 : dl < 50000 ? 2 : dl < 100000 ? 3 : dl < 200000 ? 4 : dl < 400000 ? 5 : dl < 800000 ? 6 : 7;
    atomicAdd(&B.prof[128 + hb], 1ull);
    const int nb = wv_nodes == 0 ? 0 : wv_nodes <= 4 ? 1 : wv_nodes <= 16 ? 2 : wv_nodes <= 32 ? 3 : wv_nodes <= 64 ? 4 : 5;
    atomicAdd(&B.prof[136 + nb], 1ull);
    if (B.prof[125] == 300 && blockIdx.x < 4096) {   // one launch from the middle of the stream, wavefront by wavefront (MIQP_WAVE_DUMP)
      unsigned int hw, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      unsigned long long* const w = B.prof + 160 + 4 * blockIdx.x;
      w[0] = wv_r0; w[1] = wv_r1; w[2] = ((unsigned long long)xcc << 32) | hw; w[3] = (unsigned long long)wv_nodes;
    }
  }
#endif
}

}  // namespace miqp
