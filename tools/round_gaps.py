"""where a B&B round's wall time goes, from a rocprofv3 --kernel-trace CSV: per kernel the busy time, and the idle time of the device
between consecutive kernels (no kernel running).  python tools/round_gaps.py <dir with *_kernel_trace.csv>"""
import csv, glob, os, sys, collections
d = sys.argv[1]
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
# the steady state: between the first and the last launch of the standard on-chip kernel
ANCHOR = "as_onchip_kernel<2, 10, 128>" if any("as_onchip_kernel<2, 10, 128>" in r[2] for r in rows) else "ipm_onchip_kernel<2, 10, 0, 128>"   # the standard launch of a round
oc = [r for r in rows if ANCHOR in r[2]]
lo, hi = oc[len(oc) // 10][0], oc[-len(oc) // 10][1]
sel = [r for r in rows if r[0] >= lo and r[1] <= hi]
busy = collections.Counter(); cnt = collections.Counter()
for a, b, n in sel:
    k = n.split("(")[0][:60]; busy[k] += b - a; cnt[k] += 1
# union of busy intervals -> idle time
ev = sorted((a, b) for a, b, n in sel)
idle = 0; cur = ev[0][1]; gaps = []
for a, b in ev[1:]:
    if a > cur: idle += a - cur; gaps.append(a - cur)
    cur = max(cur, b)
span = hi - lo
nr = len([r for r in sel if ANCHOR in r[2]])
print("steady span %.3f s, %d rounds, %.3f ms per round; device idle (no kernel running) %.1f %% = %.3f ms per round" % (span / 1e9, nr, span / 1e6 / nr, 100.0 * idle / span, idle / 1e6 / nr))
for k, v in busy.most_common(12):
    print("  %-60s %7.3f ms per round (%5.1f %% of the span, %d launches)" % (k, v / 1e6 / nr, 100.0 * v / span, cnt[k]))
# what follows the end of the standard launch of a round until the next one starts
nxt = []
for i in range(len(oc) - 1):
    if oc[i][0] >= lo and oc[i + 1][1] <= hi: nxt.append(oc[i + 1][0] - oc[i][1])
if nxt:
    nxt.sort(); print("end of the standard launch -> start of the next: median %.3f ms, mean %.3f ms" % (nxt[len(nxt) // 2] / 1e6, sum(nxt) / len(nxt) / 1e6))
# the timeline of three rounds from the middle of the steady state: start and end of every kernel, ms after the round's select_kernel started
if len(sys.argv) > 2 and sys.argv[2] == "timeline":
    mid = [r for r in sel if "select_kernel" in r[2]]
    for s in mid[len(mid) // 2: len(mid) // 2 + 3]:
        nx = [r for r in mid if r[0] > s[0]]
        end = nx[0][0] if nx else hi
        print("round at %.3f s:" % ((s[0] - t0) / 1e9))
        for a, b, n in sel:
            if s[0] <= a < end: print("    %-50s start %7.3f  end %7.3f  (%.3f ms)" % (n.split("(")[0][:50], (a - s[0]) / 1e6, (b - s[0]) / 1e6, (b - a) / 1e6))
