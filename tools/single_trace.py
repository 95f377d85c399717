"""single solves of a few slow cfg3 seeds, for a kernel trace (GPU only):  rocprofv3 --kernel-trace ... -- python3 tools/single_trace.py [seeds...]
   python tools/single_trace.py --analyse <trace dir>: per kernel the time per round, and the device's idle time inside the solves"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 2 and sys.argv[1] == "--analyse":
    import csv, glob, collections
    f = sorted(glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:56]) for r in csv.DictReader(open(f)))
    busy = collections.Counter(); cnt = collections.Counter(); idle = 0; idle_n = 0; cur = rows[0][1]; span = 0; seg0 = rows[0][0]
    for a, b, n in rows:
        busy[n] += b - a; cnt[n] += 1
        if a > cur:
            if a - cur < 1_000_000: idle += a - cur; idle_n += 1      # (a gap of more than 1 ms is between two solves)
            else: span += cur - seg0; seg0 = a
        cur = max(cur, b)
    span += cur - seg0
    nr = max(1, cnt[[k for k in cnt if k.startswith("miqp::select_kernel")][0]])
    print("solve span %.1f ms, %d rounds, %.3f ms per round; idle inside the solves %.3f ms per round (%d gaps)" % (span / 1e6, nr, span / 1e6 / nr, idle / 1e6 / nr, idle_n))
    for k, v in busy.most_common(14): print("  %-58s %7.3f ms per round, %5.2f launches per round, %7.1f us each" % (k, v / 1e6 / nr, cnt[k] / nr, v / 1e3 / cnt[k]))
    sys.exit(0)
import planner_miqp_amd as P
from planner_miqp_amd import synthetic
seeds = [int(a) for a in sys.argv[1:]] or [0, 11, 53, 87, 85]
w = P.CplexWrapper()
for s in seeds:
    w.resetParameters(synthetic.generate("cfg3", s, gap=0.1, max_time=10.0))
    t = time.time(); st = w.callCplex(); dt = time.time() - t; pr = w.getSolutionProperties()
    print("seed %d: %.1f ms, %d nodes, %d rounds, status %d" % (s, 1e3 * dt, pr.nodes, w.lastTiming()["ipm_launches"], pr.status))
