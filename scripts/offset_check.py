"""Speed of the sweep path on clouds far from the origin: the prune margin grows with 0.5|x|^2 (rounding of the expanded
score), so a map-frame cloud has wider slabs than the same cloud centred.  Exactness is covered by the tests."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs
B, n, K = 64, 16384, 10
src0, tgt0 = make_pairs(B, n, n, seed=3)
for off in (0.0, 100.0, 1000.0, 10000.0):
    shift = torch.tensor([off, -0.5 * off, 0.25 * off])
    src = (src0 + shift).cuda()
    tgt = tgt0.clone(); tgt[:, :, :3] += shift; tgt = tgt.cuda()
    T0 = torch.eye(4, device="cuda").repeat(B, 1, 1)
    icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=1e-12); icp.const_iter = True
    def call():
        s, t = src.detach().requires_grad_(True), tgt.detach().requires_grad_(True)
        out = icp.icp(s, t, T0, trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}); out["T"].sum().backward(); return out
    for _ in range(3): call()
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); call(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    frac = float(icp.knn_stats["knn_pairs"].sum()) / (float(B) * n * n * K)
    print("offset %8.0f m: %.3f ms/iteration  pairs scored %.2f %%" % (off, sorted(ts)[2] * 1e3 / K, 100 * frac), flush=True)
