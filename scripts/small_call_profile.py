"""Small calls for a kernel trace: the reference's bundled 65-point pair (float64, as its tests pass it) and a batch of 64 clouds x 256 points (float32); fwd + bwd."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
scan, mp = torch.from_numpy(np.load(os.path.join(G, "points_scan.npy"))), torch.from_numpy(np.load(os.path.join(G, "points_map.npy")))
cases = []
for name, dtype, const in (("65-point pair f64 tolerance 1e-10 (the reference's test)", torch.float64, False), ("65-point pair f64 K=20", torch.float64, True), ("65-point pair f32 K=20", torch.float32, True)):
    icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=100 if not const else 20, tolerance=1e-10)
    icp.const_iter = const
    S, Tg, Ti = scan.to(dtype).cuda(), mp.to(dtype).cuda(), torch.eye(4, dtype=dtype).cuda()
    cases.append((name, icp, S, Tg, Ti))
src, tgt = make_pairs(64, 256, 256, seed=5, dtype=torch.float32)
icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=20, tolerance=1e-12)
icp.const_iter = True
cases.append(("64 x 256 f32 K=20", icp, src.cuda(), tgt.cuda(), torch.eye(4).repeat(64, 1, 1).cuda()))
for name, icp, S, Tg, Ti in cases:
    def call():
        s = S.detach().requires_grad_(True)
        out = icp.icp(s, Tg, Ti, trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=3)
        out["T"].sum().backward()
        return out
    for _ in range(10):
        out = call()
    torch.cuda.synchronize()
    ts = []
    for _ in range(30):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        call()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    print("%-62s %d iterations   median %.3f ms per call" % (name, out["deltas"].shape[1], ts[15] * 1e3), flush=True)
