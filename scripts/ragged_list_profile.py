import cProfile, pstats, sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_independent_pairs
B, n = 256, 16384
S, Tg = make_independent_pairs(B, n, n, seed=3, dtype=torch.float32, ragged=True)
S, Tg = [x.cuda() for x in S], [x.cuda() for x in Tg]
T0 = [torch.eye(4, device="cuda")] * B
icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=10, tolerance=1e-12); icp.const_iter = True
def call():
    s_ = [x.detach().requires_grad_(True) for x in S]; t_ = [x.detach().requires_grad_(True) for x in Tg]
    o = icp.icp(s_, t_, T0, trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=3)
    t1 = time.perf_counter()
    o["T"].sum().backward()
    torch.cuda.synchronize()
    return t1
for _ in range(3): call()
t0 = time.perf_counter(); t1 = call(); t2 = time.perf_counter()
print("call %.2f ms: icp() returned after %.2f ms, backward + sync %.2f ms" % ((t2 - t0) * 1e3, (t1 - t0) * 1e3, (t2 - t1) * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(5): call()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
