"""state a solve leaves behind: instance A, then instance B twice in one process with MIQP_TRACE=1; the per-round traces of the two
B solves are compared (the first follows A, the second follows B itself).  python tools/repeat_trace.py seedA seedB gap"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("RT_CHILD"):
    sys.path.insert(0, ROOT)
    import planner_miqp_amd as P
    from planner_miqp_amd import synthetic
    a, b, gap = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
    w = P.CplexWrapper()
    for tag, seed in (("A", a), ("B1", b), ("B2", b)):
        sys.stderr.write("[marker] %s\n" % tag); sys.stderr.flush()
        w.resetParameters(synthetic.generate("cfg3", seed, gap=gap, max_time=10.0)); w.callCplex()
        pr = w.getSolutionProperties()
        sys.stderr.write("[result] %s %s %d\n" % (tag, float(pr.objective).hex(), pr.nodes)); sys.stderr.flush()
else:
    env = dict(os.environ, RT_CHILD="1", MIQP_TRACE="1")
    out = subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, capture_output=True, text=True)
    parts, cur = {}, None
    for line in out.stderr.splitlines():
        if line.startswith("[marker]"): cur = line.split()[1]; parts[cur] = []
        elif cur and (line.startswith("[trace]") or line.startswith("[result]")): parts[cur].append(line)
    b1, b2 = parts.get("B1", []), parts.get("B2", [])
    print(len(b1), len(b2), b1[-1] if b1 else None, b2[-1] if b2 else None)
    for k, (x, y) in enumerate(zip(b1, b2)):
        if x != y:
            print("first difference at line", k); print(" B1:", x); print(" B2:", y)
            for j in range(max(0, k - 3), k): print(" (same)", b1[j])
            break
