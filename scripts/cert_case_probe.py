import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from dicp_amd import _lib
from dicp_amd.ICP import ICP
import test_gpu_configs as TG
case = sys.argv[1]
dtype = torch.float32 if len(sys.argv) < 3 else getattr(torch, sys.argv[2])
src, tgt, K = TG._cert_case(case, dtype)
N = src.shape[0]
icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=1e-12); icp.const_iter = True
icp.knn_variant = _lib.KNN_SWEEP
out = icp.icp(src.cuda().requires_grad_(True), tgt.cuda(), torch.eye(4, dtype=dtype).cuda().repeat(N, 1, 1), **TG.KW)
torch.cuda.synchronize()
c = icp.knn_stats["searched_again"]
print("units", c[:, :64].sum(1).tolist(), "single", c[:, 64:].sum(1).tolist())
q = icp.knn_stats["budgets"]
print("budgets: -1:", int((q == -1).sum()), "inf:", int(torch.isinf(q).sum()), "nan:", int(torch.isnan(q).sum()), "median", float(q.median()), "of", q.numel())
print("deltas norm per iteration (cloud 0):", out["deltas"][0].norm(dim=1).tolist())
print("mean weight per iteration:", out["weights"].mean(dim=(0, 2, 3)).tolist())
import time
def fwd_ms(reuse):
    i2 = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=1e-12); i2.const_iter = True
    i2.knn_variant = _lib.KNN_SWEEP; i2.reuse_matches = reuse
    s, t, T0 = src.cuda(), tgt.cuda(), torch.eye(4, dtype=dtype).cuda().repeat(N, 1, 1)
    ts = []
    for _ in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter(); i2.icp(s, t, T0, **TG.KW); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[2] * 1e3
print("forward call, %d iterations: %.3f ms with certificates, %.3f ms without" % (K, fwd_ms(True), fwd_ms(False)))
