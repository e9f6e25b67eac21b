"""BASELINE configs[3] on a 64-cloud slice (65536-point clouds, point-to-plane + huber + trim, K = 5, fwd + bwd): ms per call, median of 7 (bench.py's value_c4.sweep leg alone).
usage: [DICP_F16_ADAPTIVE=0] PYTHONPATH=. python scripts/c4_slice_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs
B, n, K = int(os.environ.get("B", 64)), 65536, 5
src, tgt = make_pairs(B, n, n, seed=4, dtype=torch.float32)
src, tgt = src.cuda(), tgt.cuda()
T0 = torch.eye(4, device="cuda").repeat(B, 1, 1)
icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=1e-12); icp.const_iter = True
def call():
    s, t = src.detach().requires_grad_(True), tgt.detach().requires_grad_(True)
    icp.icp(s, t, T0, trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0})["T"].sum().backward()
for _ in range(4):
    call()
torch.cuda.synchronize()
ts = []
for _ in range(7):
    t0 = time.perf_counter(); call(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print("adaptive=%s  B=%d x %d  K=%d: %.3f ms per call (%.3f per iteration), pairs %.4f" % (os.environ.get("DICP_F16_ADAPTIVE", "1"), B, n, K, sorted(ts)[3] * 1e3, sorted(ts)[3] * 1e3 / K,
      float(icp.knn_stats["knn_pairs"].sum().item()) / K / (float(B) * n * n)))
