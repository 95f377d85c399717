"""seed(s) of cfg3 under MIQP_LNS / MIQP_AS combinations: nodes, time, objective (python tools/lns_ab.py 1913 662)"""
import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import planner_miqp_amd as P
    from planner_miqp_amd import synthetic
    for seed in map(int, sys.argv[2:]):
        w = P.CplexWrapper(); w.resetParameters(synthetic.generate("cfg3", seed, gap=1e-2, max_time=60)); t = time.time(); st = w.callCplex(); dt = time.time() - t
        pr = w.getSolutionProperties(); tm = w.lastTiming()
        print("  seed %d status %d/%d obj %.6f nodes %d time %.3f s pool %d as_nodes %d" % (seed, int(st), pr.status, pr.objective, pr.nodes, dt, pr.NrSolutionPool, tm["as_nodes"]), flush=True)
else:
    for lns in ("0", "45"):
        for a in ("0", "1"):
            print("MIQP_LNS=%s MIQP_AS=%s" % (lns, a), flush=True)
            subprocess.call([sys.executable, __file__, "child"] + sys.argv[1:], env=dict(os.environ, MIQP_LNS=lns, MIQP_AS=a))
