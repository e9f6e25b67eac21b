"""Tolerance mode at the benchmark shape: early stop, sync cadence, iterations used, time per call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs
B, n = 256, 16384
src, tgt = make_pairs(B, n, n, seed=3); src, tgt = src.cuda(), tgt.cuda()
T0 = torch.eye(4, device="cuda").repeat(B, 1, 1)
for tol, sync in ((1e-4, None), (1e-4, 1), (1e-6, None), (1e-9, None)):
    icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=50, tolerance=tol)
    icp.sync_every = sync
    def call():
        s, t = src.detach().requires_grad_(True), tgt.detach().requires_grad_(True)
        out = icp.icp(s, t, T0, trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0})
        out["T"].sum().backward()
        return out
    for _ in range(2): out = call()
    torch.cuda.synchronize(); t0 = time.perf_counter(); out = call(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    its = out["stats"]["iterations"]
    print("tol %.0e sync_every %s: K executed %d, per-cloud iterations min/mean/max %d/%.1f/%d, converged %d/%d, %.2f ms/call (%.3f ms/iteration)"
          % (tol, sync, out["deltas"].shape[1], int(its.min()), float(its.mean()), int(its.max()), int(out["stats"]["converged"].sum()), B, dt * 1e3, dt * 1e3 / out["deltas"].shape[1]))
