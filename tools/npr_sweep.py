"""single hard instances (cfg3 seeds 118, 179, 207, 243) alone on the GPU for several round widths (diagnostic)"""
import os, sys, time, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import planner_miqp_amd as P
    from planner_miqp_amd import synthetic
    out = []
    for seed in (118, 179, 207, 243):
        w = P.CplexWrapper(); w.resetParameters(synthetic.generate("cfg3", seed, gap=0.01, max_time=12.0))
        t = time.time(); st = w.callCplex(); dt = time.time() - t
        pr = w.getSolutionProperties()
        out.append((seed, int(st), pr.status, round(pr.gap, 5), round(dt, 2), int(pr.nodes), round(pr.objective, 3)))
    print(json.dumps(out))
else:
    for npr in (1024, 2048, 4096, 8192, 16384, 32768):
        env = dict(os.environ, MIQP_NPR=str(npr))
        r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        print(npr, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:])
