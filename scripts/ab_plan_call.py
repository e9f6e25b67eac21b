"""A/B of the constant-iteration forward: one dicp_icp_forward_plan call against one dicp_icp_forward call per segment (B=256 x 16384, fwd + bwd)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs
B, n = 256, 16384
src, tgt = make_pairs(B, n, n, seed=3); src, tgt = src.cuda(), tgt.cuda()
T0 = torch.eye(4, device="cuda").repeat(B, 1, 1)
def bench(K, **kw):
    icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=1e-12); icp.const_iter = True
    for k, v in kw.items(): setattr(icp, k, v)
    def call():
        s, t = src.detach().requires_grad_(True), tgt.detach().requires_grad_(True)
        o = icp.icp(s, t, T0, trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}); o["T"].sum().backward(); return o
    for _ in range(4): o = call()
    ts = []
    for _ in range(9):
        torch.cuda.synchronize(); t0 = time.perf_counter(); o = call(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[4] * 1e3, o["T"]
for rnd in range(2):
    for K in (10, 20):
        a, Ta = bench(K, plan_call=True)
        b, Tb = bench(K, plan_call=False)
        print("K=%d plan call %.3f ms | per-segment calls %.3f ms  (same T: %s)" % (K, a, b, torch.equal(Ta, Tb)), flush=True)
