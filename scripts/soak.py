"""Stability soak: many differentiable ICP calls of mixed shapes/configs; memory must stay flat, results finite."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs

torch.manual_seed(0)
cases = [(32, 4096, "pt2pt", 0), (8, 16384, "pt2pl", 0), (3, 777, "pt2pl", 1), (64, 2048, "pt2pl", 3), (2, 30000, "pt2pt", 0)]
data = {}
for B, n, typ, knn in cases:
    s, t = make_pairs(B, n, n + 13, seed=B, dtype=torch.float32)
    data[(B, n)] = (s.cuda(), (t if typ == "pt2pl" else t[:, :, :3].contiguous()).cuda())
t0 = time.time()
peak0 = None
for it in range(120):
    B, n, typ, knn = cases[it % len(cases)]
    s, t = data[(B, n)]
    s, t = s.detach().requires_grad_(True), t.detach().requires_grad_(True)
    icp = ICP(icp_type=typ, differentiable=(it % 3 != 0), max_iterations=8 + it % 5, tolerance=1e-6 if it % 2 else 1e-12)
    icp.const_iter = bool(it % 4 == 0)
    icp.knn_variant = knn
    icp.sync_every = 1 + it % 3
    out = icp.icp(s, t, torch.eye(4, device="cuda").repeat(B, 1, 1), trim_dist=5.0,
                  loss_fn=None if it % 7 == 0 else {"name": "huber" if it % 2 else "cauchy", "metric": 1.0}, dim=3 if it % 5 else 2)
    (out["T"].sum() + out["pc"].mean()).backward()
    assert torch.isfinite(out["T"]).all() and torch.isfinite(s.grad).all() and torch.isfinite(t.grad).all(), it
    if it == 19:
        torch.cuda.synchronize(); peak0 = torch.cuda.max_memory_allocated(); torch.cuda.reset_peak_memory_stats()
torch.cuda.synchronize()
peak1 = torch.cuda.max_memory_allocated()
print("120 calls in %.1f s; peak memory first 20 calls %.2f GB, last 100 calls %.2f GB, allocated now %.2f GB" %
      (time.time() - t0, peak0 / 2**30, peak1 / 2**30, torch.cuda.memory_allocated() / 2**30))
assert peak1 <= peak0 * 1.25 + (1 << 28), "memory grows across calls"
print("soak ok")
