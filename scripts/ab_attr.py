"""A/B of one boolean ICP attribute on one box (B=256 x 16384 pt2pl huber, fwd + bwd, median of 9 calls after 4 warm-ups).  usage: python scripts/ab_attr.py <attribute> [K ...]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs
attr = sys.argv[1]
Ks = [int(v) for v in sys.argv[2:]] or [10, 20]
B, n = 256, 16384
src, tgt = make_pairs(B, n, n, seed=3); src, tgt = src.cuda(), tgt.cuda()
T0 = torch.eye(4, device="cuda").repeat(B, 1, 1)
def bench(K, value):
    icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=1e-12); icp.const_iter = True
    assert hasattr(icp, attr), attr
    setattr(icp, attr, value)
    def call():
        s, t = src.detach().requires_grad_(True), tgt.detach().requires_grad_(True)
        o = icp.icp(s, t, T0, trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}); o["T"].sum().backward(); return o
    for _ in range(4): o = call()
    ts = []
    for _ in range(9):
        torch.cuda.synchronize(); t0 = time.perf_counter(); o = call(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[4] * 1e3, o["T"]
for rnd in range(2):
    for K in Ks:
        a, Ta = bench(K, True)
        b, Tb = bench(K, False)
        print("K=%d %s=True %.3f ms | False %.3f ms  (same T: %s)" % (K, attr, a, b, torch.equal(Ta, Tb)), flush=True)
