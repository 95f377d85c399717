"""solve hard instances one at a time and dump what the offline cut analysis needs: objective, bound, nodes, the region
sequence and the car/car alternatives of the incumbent, the trajectory.  python tools/dump_hard.py cfg out.json seed [seed ...]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import planner_miqp_amd as P
from planner_miqp_amd import synthetic
cfg, out = sys.argv[1], sys.argv[2]
res = {}
for s in sys.argv[3:]:
    p = synthetic.generate(cfg, int(s), gap=float(os.environ.get("GAP", "0.01")), max_time=float(os.environ.get("TL", "40")))
    w = P.CplexWrapper(); w.resetParameters(p)
    t = time.time(); st = w.callCplex(); dt = time.time() - t
    pr = w.getSolutionProperties()
    print("%s seed %s: status %d obj %.4f gap %.4f nodes %d iters %d time %.2f" % (cfg, s, pr.status, pr.objective, pr.gap, pr.nodes, pr.NrIterations, dt), flush=True)
    if int(st) != 0:
        continue
    r = w.getRawResults()
    res[s] = dict(obj=pr.objective, gap=pr.gap, nodes=int(pr.nodes), time=dt, status=int(pr.status),
                  region=np.argmax(r.active_region, axis=2).tolist(),
                  c2c=np.asarray(r.car2car_collision).tolist(),
                  env=[np.asarray(getattr(r, n)).tolist() for n in ("notWithinEnvironmentRear", "notWithinEnvironmentFrontUbUb", "notWithinEnvironmentFrontLbUb", "notWithinEnvironmentFrontUbLb", "notWithinEnvironmentFrontLbLb")],
                  rc=[np.asarray(getattr(r, n)).tolist() for n in ("region_change_not_allowed_x_positive", "region_change_not_allowed_y_positive", "region_change_not_allowed_x_negative", "region_change_not_allowed_y_negative", "region_change_not_allowed_combined")],
                  vx=r.vel_x.tolist(), vy=r.vel_y.tolist(), ax=r.acc_x.tolist(), ay=r.acc_y.tolist(), ux=r.u_x.tolist(), uy=r.u_y.tolist(), px=r.pos_x.tolist(), py=r.pos_y.tolist())
json.dump(res, open(out, "w"))
