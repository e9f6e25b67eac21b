"""A/B on one box: query re-order schedule of the sweep (ICP._tuning["sweep_resort"]), whole call fwd+bwd at the headline shape."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs
B, n = 256, 16384
src, tgt = make_pairs(B, n, n, seed=3)
src, tgt = src.cuda(), tgt.cuda()
T0 = torch.eye(4, device="cuda").repeat(B, 1, 1)
def bench(resort, K, cert_from=None):
    icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=1e-12); icp.const_iter = True; icp._tuning["sweep_resort"] = resort; icp._tuning["cert_from"] = cert_from
    def call():
        s, t = src.detach().requires_grad_(True), tgt.detach().requires_grad_(True)
        icp.icp(s, t, T0, trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0})["T"].sum().backward()
    for _ in range(4): call()
    ts = []
    for _ in range(9):
        torch.cuda.synchronize(); t0 = time.perf_counter(); call(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[4] * 1e3
for rnd in range(2):
    for K in (10, 20):
        print("K=%d " % K + "  ".join("%s/%s: %.3f ms" % (r, c, bench(r, K, c)) for r, c in (((0, 1, 2, 3), None), ((0, 1, 2), 2), ((0, 1, 2), 3), ((0, 1), 1), ((0, 1), 2), ((0, 2), 2), ((0, 1, 3), 3))), flush=True)
