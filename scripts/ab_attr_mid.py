"""A/B of one boolean ICP attribute at a mid-size shape (host-bound): forward + backward, median of 60 calls, interleaved.  usage: python scripts/ab_attr_mid.py <attribute> [B n icp_type K]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs
attr = sys.argv[1]
B, n, typ, K = (int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])) if len(sys.argv) > 5 else (32, 4096, "pt2pl", 10)
src, tgt = make_pairs(B, n, n, seed=3); src, tgt = src.cuda(), tgt.cuda()
if typ == "pt2pt":
    tgt = tgt[:, :, :3].contiguous()
T0 = torch.eye(4, device="cuda").repeat(B, 1, 1)
kw = dict(trim_dist=5.0) if typ == "pt2pt" else dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0})
icps = {}
for v in (True, False):
    icp = ICP(icp_type=typ, differentiable=True, max_iterations=K, tolerance=1e-12); icp.const_iter = True
    setattr(icp, attr, v); icps[v] = icp
def call(icp):
    s, t = src.detach().requires_grad_(True), tgt.detach().requires_grad_(True)
    icp.icp(s, t, T0, **kw)["T"].sum().backward()
for v in (True, False):
    for _ in range(10): call(icps[v])
ts = {True: [], False: []}
for rep in range(60):
    for v in (True, False):
        torch.cuda.synchronize(); t0 = time.perf_counter(); call(icps[v]); torch.cuda.synchronize(); ts[v].append(time.perf_counter() - t0)
med = lambda x: sorted(x)[len(x) // 2] * 1e3
print("B=%d n=%d %s K=%d: %s=True %.3f ms | False %.3f ms per forward + backward (median of 60, interleaved)" % (B, n, typ, K, attr, med(ts[True]), med(ts[False])))
