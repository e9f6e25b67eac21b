import sys, os, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from dicp_amd import _lib
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs
for (B, n) in ((3, 100), (3, 5000)):
    src, tgt = make_pairs(B, n, n, seed=3)
    src[1, 7, 0] = float("nan"); tgt[2, 5, 1] = float("inf")
    src, tgt = src.cuda(), tgt.cuda()
    T0 = torch.eye(4, device="cuda").repeat(B, 1, 1)
    for knn in (_lib.KNN_VALU, _lib.KNN_SWEEP):
        icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=4, tolerance=1e-12); icp.const_iter = True; icp.knn_variant = knn
        s, t = src.detach().requires_grad_(True), tgt.detach().requires_grad_(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = icp.icp(s, t, T0, trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}); out["T"][0].sum().backward()
        torch.cuda.synchronize()
        print("B=%d n=%d knn=%d: %.1f ms; cloud 0 finite: T %s grad %s; cloud 1 T finite %s; cloud 2 T finite %s" % (B, n, knn, (time.perf_counter() - t0) * 1e3,
              bool(torch.isfinite(out["T"][0]).all()), bool(torch.isfinite(s.grad[0]).all()), bool(torch.isfinite(out["T"][1]).all()), bool(torch.isfinite(out["T"][2]).all())))
