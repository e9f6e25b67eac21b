"""configs[1]-shaped eager calls (32 x 4096, K = 10, fwd + bwd): how often should the sweep re-order its queries at this size?  Interleaved rounds, back-to-back calls."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs, make_scene_pairs

N, n, K = (int(v) for v in (sys.argv[1:4] + ["32", "4096", "10"][len(sys.argv) - 1:]))
kw = dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=3)
for kind, make in (("random", make_pairs), ("scene", make_scene_pairs)):
    src, tgt = make(N, n, n, seed=3, dtype=torch.float32)
    S, Tg = src.cuda().requires_grad_(True), tgt.cuda()
    Ti = torch.eye(4).repeat(N, 1, 1).cuda()
    objs = {}
    for resort in ((0, 1, 2, 3), (0, 1, 2), (0, 1), (0,), (0, 2)):
        icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=1e-12)
        icp.const_iter = True
        icp._tuning["sweep_resort"] = resort
        for _ in range(20):
            icp.icp(S, Tg, Ti, **kw)["T"].sum().backward()
        objs[resort] = icp
    torch.cuda.synchronize()
    times = {r: [] for r in objs}
    for rnd in range(5):
        for r, icp in objs.items():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(100):
                icp.icp(S, Tg, Ti, **kw)["T"].sum().backward()
            torch.cuda.synchronize()
            times[r].append((time.perf_counter() - t0) / 100 * 1e3)
    for r, v in times.items():
        v.sort()
        print("%-7s re-ordering before iterations %-14s median %.3f ms per call  (min %.3f)" % (kind, r, v[2], v[0]), flush=True)
