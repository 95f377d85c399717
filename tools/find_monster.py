import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import planner_miqp_amd as P
from planner_miqp_amd import synthetic
ws=[]
for s in range(2560, 2560+2048):
    w=P.CplexWrapper(); w.resetParameters(synthetic.generate("cfg3", s, gap=0.01, max_time=60)); ws.append(w)
P.solve_batch(ws, inflight=256)
r=sorted(((w.getSolutionProperties().time, 2560+k, int(w.getSolutionProperties().nodes)) for k,w in enumerate(ws)), reverse=True)[:5]
print("slowest in the crowd:", r)
for t,s,n in r[:2]:
    w=P.CplexWrapper(); w.resetParameters(synthetic.generate("cfg3", s, gap=0.01, max_time=60)); t0=time.time(); w.callCplex(); pr=w.getSolutionProperties()
    print("alone: seed %d status %d nodes %d time %.2f obj %.2f" % (s, pr.status, pr.nodes, time.time()-t0, pr.objective))
