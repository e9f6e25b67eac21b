"""Hard inputs (make_independent_pairs, dense, K = 10): the match certificates on / off, and how many clouds end a call with theirs switched off.
usage (MI355X): PYTHONPATH=. python scripts/hard_certs.py [K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_independent_pairs
B, n = 256, 16384
K = int(sys.argv[1]) if len(sys.argv) > 1 else 10
S, Tg = make_independent_pairs(B, n, n, seed=3, dtype=torch.float32, ragged=False)
S, Tg = S.cuda(), Tg.cuda()
T0 = torch.eye(4, device="cuda").repeat(B, 1, 1)
for reuse in (True, False, True):
    icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=1e-12); icp.const_iter = True
    icp.reuse_matches = reuse
    ts, offs = [], []
    for i in range(9):
        s_, t_ = S.detach().requires_grad_(True), Tg.detach().requires_grad_(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        o = icp.icp(s_, t_, T0, trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=3)
        o["T"].sum().backward()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        offs.append(int(icp.knn_stats["certs_off"].sum()) if "certs_off" in icp.knn_stats else -1)
    print("reuse_matches=%-5s  ms per call by call: %s   clouds with certificates off at the end (-1: call without certificates): %s" % (reuse, ["%.1f" % t for t in ts], offs))
