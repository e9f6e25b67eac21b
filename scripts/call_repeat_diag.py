import os, sys, time, gc
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs
B, n, K = 256, 16384, 10
src, tgt = make_pairs(B, n, n, seed=3)
src, tgt = src.cuda(), tgt.cuda()
T0 = torch.eye(4, device="cuda").repeat(B, 1, 1)
icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=2, tolerance=1e-12)
icp.const_iter = True
kw = dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=3)
def call():
    s, t = src.detach().requires_grad_(True), tgt.detach().requires_grad_(True)
    out = icp.icp(s, t, T0, **kw)
    out["T"].sum().backward()
    return out
call()
icp.max_iterations = K
gc.collect(); gc.disable()
if os.environ.get("EVENTS") == "1":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    log = bench.EventLog(); log.handles(K); icp._tuning["timing_events"] = log
for rep in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter(); o = call(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("call %d: host-return %.2f ms, done %.2f ms" % (rep, (t1 - t0) * 1e3, (t2 - t0) * 1e3), flush=True)
