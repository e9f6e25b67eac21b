import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_independent_pairs
B, n, K = 256, 16384, 10
S, Tg = make_independent_pairs(B, n, n, seed=3, dtype=torch.float32, ragged=False)
S, Tg = S.cuda(), Tg.cuda()
T0 = torch.eye(4, device="cuda").repeat(B, 1, 1)
for det in (False, True, False, True):
    icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=1e-12); icp.const_iter = True
    icp.deterministic = det
    ts = []
    for i in range(8):
        s_, t_ = S.detach().requires_grad_(True), Tg.detach().requires_grad_(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        o = icp.icp(s_, t_, T0, trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=3)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        o["T"].sum().backward()
        torch.cuda.synchronize(); ts.append(((t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3))
    print("deterministic=%-5s forward / backward ms by call: %s" % (det, ["%.1f/%.1f" % t for t in ts[3:]]))
