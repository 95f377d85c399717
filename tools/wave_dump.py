"""reads the MIQP_WAVE_DUMP file of the profile build (one round's launches: block, start, end [100 MHz ticks], XCC id, HW_ID, nodes solved, kind:
0 the standard active-set launch, 1 its larger block, 2 the larger interior point variant, 3 the memory-backed kernel): when the wavefronts
started, where, and how many nodes each solved.  python tools/wave_dump.py file"""
import sys, collections
rows = [tuple(int(x) for x in l.split()) for l in open(sys.argv[1]) if l.strip()]
rows = [r if len(r) > 6 else r + (0,) for r in rows]
NAMES = {0: "standard active-set", 1: "larger active-set", 2: "larger interior point", 3: "memory-backed"}
t0 = min(r[1] for r in rows); t1 = max(r[2] for r in rows)
def cu_of(r): hw = r[4]; return (r[3], (hw >> 13) & 7, (hw >> 8) & 15)   # gfx9 HW_ID: se_id [15:13], sh_id [12], cu_id [11:8], simd_id [5:4], wave_id [3:0]
def simd_of(r): return cu_of(r) + ((r[4] >> 4) & 3,)
for k in sorted(set(r[6] for r in rows)):
    rk = [r for r in rows if r[6] == k]
    nd = sorted(r[5] for r in rk)
    print("%-22s %5d wavefronts, first start %.3f ms, last end %.3f ms; nodes %d (per wavefront median %d, max %d); wavefronts without a node %d; resident time mean %.3f ms" % (
        NAMES.get(k, k), len(rk), (min(r[1] for r in rk) - t0) / 1e5, (max(r[2] for r in rk) - t0) / 1e5, sum(nd), nd[len(nd) // 2], nd[-1], sum(1 for x in nd if x == 0), sum(r[2] - r[1] for r in rk) / len(rk) / 1e5))
print("resident wavefronts over the round (ms after the first start): standard / larger active-set / larger interior point / memory-backed; SIMDs without a standard wavefront; CUs by their standard wavefronts")
for q in range(25):
    t = t0 + (t1 - t0) * q // 24
    res = [r for r in rows if r[1] <= t < r[2]]
    by = collections.Counter(r[6] for r in res)
    std = [r for r in res if r[6] == 0]
    percu = collections.Counter(cu_of(r) for r in std)
    hist = collections.Counter(percu.values())
    print("  t = %6.3f: %4d / %4d / %4d / %4d; SIMDs with a standard wavefront %4d; CUs with 8 / 7 / 6 / 5 / fewer: %d / %d / %d / %d / %d" % ((t - t0) / 1e5, by[0], by[1], by[2], by[3], len(set(simd_of(r) for r in std)), hist[8], hist[7], hist[6], hist[5], 256 - hist[8] - hist[7] - hist[6] - hist[5]))
