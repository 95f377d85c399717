"""sequence of node counts of one cfg3 instance solved n times in a row (after one solve of another seed): python tools/repeat_seq.py seed n gap"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import planner_miqp_amd as P
from planner_miqp_amd import synthetic
seed, n, gap = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
w = P.CplexWrapper()
w.resetParameters(synthetic.generate("cfg3", seed - 1, gap=gap, max_time=10.0)); w.callCplex()
seq = []
for k in range(n):
    w.resetParameters(synthetic.generate("cfg3", seed, gap=gap, max_time=10.0)); w.callCplex()
    seq.append(int(w.getSolutionProperties().nodes))
print(seq)
