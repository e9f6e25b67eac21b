"""Host and wall time of one eager configs[1] call (32 x 4096 x 4096, K = 10, forward + backward) on the one-call path and on ICPLoop."""
import sys
import time

import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs

N, n, K = (int(v) for v in (sys.argv[1:4] + ["32", "4096", "10"][len(sys.argv) - 1:]))
src, tgt = make_pairs(N, n, n, seed=3, dtype=torch.float32)
S, Tg = src.cuda().requires_grad_(True), tgt.cuda()
Ti = torch.eye(4).repeat(N, 1, 1).cuda().requires_grad_(True)
kw = dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=3)
for one in (True, False, True, False):
    icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=1e-12)
    icp.const_iter = True
    icp._tuning["one_call"] = one
    for _ in range(20):
        icp.icp(S, Tg, Ti, **kw)["T"].sum().backward()
    torch.cuda.synchronize()
    reps = 200
    t0 = time.perf_counter()
    host_f = host_b = 0.0
    for _ in range(reps):
        a = time.perf_counter()
        loss = icp.icp(S, Tg, Ti, **kw)["T"].sum()
        b = time.perf_counter()
        loss.backward()
        c = time.perf_counter()
        host_f += b - a
        host_b += c - b
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    print("%-9s wall %.3f ms/call   host: forward %.3f  backward %.3f ms" % ("one_call" if one else "ICPLoop", wall * 1e3, host_f / reps * 1e3, host_b / reps * 1e3), flush=True)
