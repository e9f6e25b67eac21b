import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import test_gpu_fuzz as F
from dicp_amd import _lib
from dicp_amd.ICP import ICP
for seed, kind in [(1, "dups"), (1, "far"), (2, "plain"), (2, "plane"), (2, "dups")]:
    rng = np.random.default_rng(1000 + seed * 7 + len(kind))
    N = int(rng.integers(1, 5)); n = int(rng.choice([1, 7, 64, 65, 300, 1500, 5000])); m = int(rng.choice([1, 5, 64, 129, 700, 2500, 7000]))
    dtype = torch.float64 if rng.random() < 0.4 else torch.float32
    icp_type = "pt2pl" if rng.random() < 0.6 else "pt2pt"; diff = bool(rng.random() < 0.7)
    loss = [None, {"name": "huber", "metric": 0.5}, {"name": "cauchy", "metric": 1.0}][int(rng.integers(3))]
    trim = None if rng.random() < 0.3 else 3.0; dim = 2 if rng.random() < 0.2 else 3; K = int(rng.integers(1, 7))
    print("case", seed, kind, "N", N, "n", n, "m", m, dtype, icp_type, "diff", diff, loss, "trim", trim, "dim", dim, "K", K)
    src, tgt = F.cloud_pair(rng, N, n, m, dtype, kind)
    wgt = torch.rand((N, n), generator=torch.Generator().manual_seed(seed), dtype=torch.float64).to(dtype) * 0.5 + 0.5
    outs = []
    for variant in (_lib.KNN_VALU, _lib.KNN_SWEEP):
        s, t, w = src.cuda().requires_grad_(True), tgt.cuda().requires_grad_(True), wgt.cuda().requires_grad_(True)
        T0 = torch.eye(4, dtype=dtype, device="cuda").repeat(N, 1, 1).requires_grad_(True)
        icp = ICP(icp_type=icp_type, differentiable=diff, max_iterations=K, tolerance=1e-12); icp.const_iter = True; icp.knn_variant = variant
        out = icp.icp(s, t, T0, weight=w, trim_dist=trim, loss_fn=loss, dim=dim)
        (out["T"][:, :3].sum() + 0.1 * out["pc"].sum()).backward()
        outs.append((out, s.grad, t.grad, w.grad, T0.grad))
    a, b = outs
    for key in ("T", "deltas", "weights", "costs"):
        x, y = a[0][key].detach().double(), b[0][key].detach().double()
        print("   %-8s max|a| %.3e  max diff %.3e" % (key, float(x.abs().max()), float((x - y).abs().max())))
    for k, nm in ((1, "gsrc"), (2, "gtgt"), (3, "gw"), (4, "gT0")):
        x, y = a[k].double(), b[k].double()
        print("   %-8s max|a| %.3e  max diff %.3e" % (nm, float(x.abs().max()), float((x - y).abs().max())))
