"""Solve one instance repeatedly in one process and print objective / gap / nodes as hex floats: any difference between
lines is run-to-run nondeterminism (the debug-output round trip of test_gpu_parity needs equal results)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import planner_miqp_amd as P

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    gap = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-3
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.path.abspath(os.path.join(here, "..", "tests", "golden", "ref_data", "cplexmodel_testcase.dat"))
    seen = {}
    for k in range(n):
        cw = P.CplexWrapper("cplexmodel.mod", P.ParameterSource.DATFILE, 12, gap_override=gap)
        cw.setParameterDatFileAbsolute(path)
        st = int(cw.callCplex())
        s = cw.getSolutionProperties()
        key = (float(s.objective).hex(), float(s.gap).hex(), int(s.nodes), float(s.best_bound).hex(), int(s.NrIterations))
        seen[key] = seen.get(key, 0) + 1
    for key, c in seen.items():
        print(c, key)
    print("distinct", len(seen))

if __name__ == "__main__":
    main()
