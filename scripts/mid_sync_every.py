"""Tolerance mode (the reference's default) at mid sizes: how many iterations between the host's all-converged checks?  fwd + bwd, back-to-back calls, interleaved rounds."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs

kw = dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=3)
for N, n in ((32, 4096), (64, 8192), (128, 16384), (256, 16384)):
    src, tgt = make_pairs(N, n, n, seed=3, dtype=torch.float32)
    S, Tg = src.cuda().requires_grad_(True), tgt.cuda()
    Ti = torch.eye(4).repeat(N, 1, 1).cuda()
    objs = {}
    for every in (None, 1, 2, 3, 4):
        icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=50, tolerance=1e-4)
        icp.const_iter, icp.sync_every = False, every
        for _ in range(10):
            out = icp.icp(S, Tg, Ti, **kw)
            out["T"].sum().backward()
        objs[every] = (icp, out["deltas"].shape[1])
    torch.cuda.synchronize()
    times = {e: [] for e in objs}
    reps = 40 if N * n < 1e6 else 15
    for rnd in range(5):
        for e, (icp, _) in objs.items():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                icp.icp(S, Tg, Ti, **kw)["T"].sum().backward()
            torch.cuda.synchronize()
            times[e].append((time.perf_counter() - t0) / reps * 1e3)
    for e, v in times.items():
        v.sort()
        print("%4d x %5d  sync_every %-5s iterations returned %d   median %.3f ms per call" % (N, n, e, objs[e][1], v[2]), flush=True)
