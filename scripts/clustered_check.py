"""Speed on uneven clouds: 90 % of the points in a small blob, 10 % spread over the whole range (and one far outlier),
against uniform clouds of the same size.  Exactness is covered by the tests; this is about the equal-width buckets."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dicp_amd import _lib, _ops
from dicp_amd.ICP import ICP
B, n, K = 64, 16384, 10
g = torch.Generator().manual_seed(0)
def clouds(kind):
    if kind == "uniform":
        pts = (torch.rand((B, n, 3), generator=g) - 0.5) * 20
    else:
        blob = torch.randn((B, n, 3), generator=g) * 0.3
        wide = (torch.rand((B, n, 3), generator=g) - 0.5) * 20
        pick = torch.rand((B, n, 1), generator=g) < 0.9
        pts = torch.where(pick, blob, wide)
        if kind == "blob+outlier":
            pts[:, 0, 0] = 500.0
    nrm = torch.nn.functional.normalize(torch.randn((B, n, 3), generator=g), dim=2)
    tgt = torch.cat((pts, nrm), dim=2)
    src = pts[:, torch.randperm(n, generator=g)] + 0.01 * torch.randn((B, n, 3), generator=g) - torch.tensor([0.1, 0.05, -0.02])
    return src.cuda().contiguous(), tgt.cuda().contiguous()
for kind in ("uniform", "blob", "blob+outlier"):
    src, tgt = clouds(kind)
    T0 = torch.eye(4, device="cuda").repeat(B, 1, 1)
    for knn in (0, 1):
        icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=1e-12); icp.const_iter = True
        icp.knn_variant = _lib.KNN_SWEEP if knn == 0 else _lib.KNN_VALU
        def call():
            s, t = src.detach().requires_grad_(True), tgt.detach().requires_grad_(True)
            out = icp.icp(s, t, T0, trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}); out["T"].sum().backward(); return out
        for _ in range(3): call()
        ts = []
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter(); out = call(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        dt = sorted(ts)[2]
        frac = float(icp.knn_stats["knn_pairs"].sum()) / (float(B) * n * n * K) if knn == 0 else 1.0
        print("%-13s %-6s %.2f ms/call  %.3f ms/iteration  pairs scored %.2f %%" % (kind, "sweep" if knn == 0 else "brute", dt * 1e3, dt * 1e3 / K, 100 * frac))
