"""Host time of one small call (the reference's own 65-point test pair size): where the Python goes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs
B, n, K = 1, 65, 10
src, tgt = make_pairs(B, n, n, seed=3); src, tgt = src.cuda(), tgt.cuda()
T0 = torch.eye(4, device="cuda").repeat(B, 1, 1)
icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=1e-12); icp.const_iter = True
kw = dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0})
def call():
    s, t = src.detach().requires_grad_(True), tgt.detach().requires_grad_(True)
    icp.icp(s, t, T0, **kw)["T"].sum().backward()
def fwd():
    icp.icp(src, tgt, T0, **kw)
for _ in range(20): call()
torch.cuda.synchronize()
for name, f in (("fwd+bwd", call), ("fwd only", fwd)):
    t0 = time.perf_counter()
    for _ in range(200): f()
    torch.cuda.synchronize()
    print("%s: %.1f us per call" % (name, (time.perf_counter() - t0) / 200 * 1e6))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(100): call()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
