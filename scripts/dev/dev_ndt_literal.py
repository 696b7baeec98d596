import sys, numpy as np
sys.path.insert(0,'/root/repo')
from libwave_amd import capi
from libwave_amd.pcd import load_pcd_xyz
from oracle import oracle_py as O
scan = load_pcd_xyz('/root/repo/tests/golden/testscan.pcd')
P=np.eye(4); P[0,3]=0.2
tgt = O.transform_cloud_d(scan, P)
c = capi.Context(0)
c.set_source(scan); c.set_target(tgt)
for mi in (1, 3, 35, 100):
    r = c.ndt_align(res=0.3, step_size=3.0, max_iter=mi, t_eps=1e-8, skip_line_search=1)
    print("HIP literal max_iter", mi, "iters", r["iterations"], "conv", r["converged"], "diff %.4f" % np.linalg.norm(r["T"]-P))
