"""Host time from the entry of ICP.icp() to its first kernel launches (a call that starts on an idle GPU pays it in full)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dicp_amd.ICP import ICP
from dicp_amd import _ops
from dicp_amd.synthetic import make_pairs
B, n, K = 256, 16384, 10
src, tgt = make_pairs(B, n, n, seed=3); src, tgt = src.cuda(), tgt.cuda()
T0 = torch.eye(4, device="cuda").repeat(B, 1, 1)
icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=1e-12); icp.const_iter = True
kw = dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=3)
def full():
    s, t = src.detach().requires_grad_(True), tgt.detach().requires_grad_(True)
    icp.icp(s, t, T0, **kw)["T"].sum().backward()
for _ in range(5): full()
torch.cuda.synchronize()
def timed(f, reps=20):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e6
s, t = src.detach().requires_grad_(True), tgt.detach().requires_grad_(True)
print("_batch (host, no sync)           %7.1f us" % timed(lambda: icp._batch(s, t, T0, None)))
print("prebuild_search (host, no sync)  %7.1f us" % timed(lambda: _ops.prebuild_search(s, t, 0, True)))
print("icp() host return                %7.1f us" % timed(lambda: icp.icp(s, t, T0, **kw)))
import cProfile, pstats
pr = cProfile.Profile(); torch.cuda.synchronize(); pr.enable(); icp._batch(s, t, T0, None); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
