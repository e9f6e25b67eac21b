"""Match certificates at the benchmark shape: units searched per iteration (of N * ceil(n/128))."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs
B, n, K = int(sys.argv[2]) if len(sys.argv) > 2 else 64, 16384, int(sys.argv[1]) if len(sys.argv) > 1 else 10
src, tgt = make_pairs(B, n, n, seed=3)
src, tgt = src.cuda(), tgt.cuda()
icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=1e-12); icp.const_iter = True
out = icp.icp(src.requires_grad_(True), tgt, torch.eye(4).cuda().repeat(B, 1, 1), trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0})
torch.cuda.synchronize()
units = B * ((n + 127) // 128)
cnt = icp.knn_stats["searched_again"]
print("searched again per iteration, of %d units / %d slots (0, 0 = full search):" % (units, B * n))
print("  whole units :", cnt[:, :64].sum(1).tolist())
print("  single slots:", cnt[:, 64:].sum(1).tolist())
# rows scored by the single-slot searches: pairs(K) - pairs(K-1) at steady state = one iteration's searches
def pairs_of(k):
    i2 = ICP(icp_type="pt2pl", differentiable=True, max_iterations=k, tolerance=1e-12); i2.const_iter = True
    i2.icp(src.detach().requires_grad_(True), tgt, torch.eye(4).cuda().repeat(B, 1, 1), trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0})
    torch.cuda.synchronize()
    return float(i2.knn_stats["knn_pairs"].sum()), i2.knn_stats["searched_again"][k - 1, 64:].sum().item()
pa, _ = pairs_of(K - 1)
pb, ns = pairs_of(K)
print("last iteration: %d single-slot searches scored %.0f rows, %.1f rows per slot (of %d)" % (ns, pb - pa, (pb - pa) / max(ns, 1), n))
