"""Which queries of a structured scene get no match certificate, and why (budgets as the last iteration left them)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_scene_pairs
B, n, K = 32, 16384, 8
src, tgt = make_scene_pairs(B, n, n, seed=3); src, tgt = src.cuda(), tgt.cuda()
T0 = torch.eye(4, device="cuda").repeat(B, 1, 1)
icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=1e-12); icp.const_iter = True
out = icp.icp(src, tgt, T0, trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0})
torch.cuda.synchronize()
q = icp.knn_stats["budgets"]                      # (N,n) by query
again = icp.knn_stats["searched_again"]
print("units / queries searched again per iteration:", again[:K, :64].sum(1).tolist(), again[:K, 64:].sum(1).tolist())
pc = out["pc"].detach()                           # transformed source
bad = q <= 0
print("queries without a certificate: %.2f %%" % (100 * bad.float().mean().item()))
b = 0
P, Y = pc[b].double(), tgt[b, :, :3].double()
d = torch.cdist(P, Y)                             # exact distances of cloud 0
v, ix = torch.topk(d, 3, dim=1, largest=False)
cls = lambda p: torch.where(p[:, 0].abs() > 9.9, 1, torch.where(p[:, 1].abs() > 9.9, 2, torch.where(p[:, 2].abs() < 0.05, 0, 3)))   # 0 ground, 1 x-wall, 2 y-wall, 3 clutter
c = cls(pc[b])
for k, name in enumerate(["ground", "wall perpendicular to x", "wall perpendicular to y", "clutter"]):
    sel = c == k
    print("%-26s %5d queries, %5.1f %% without certificate;  d1 median %.4f  d2 median %.4f  d2-d1 median %.4f  |  of the uncertified: d1 %.4f d2 %.4f" % (
        name, int(sel.sum()), 100 * bad[b][sel].float().mean().item(), v[sel, 0].median(), v[sel, 1].median(), (v[sel, 1] - v[sel, 0]).median(),
        v[sel & bad[b], 0].median() if bool((sel & bad[b]).any()) else float("nan"), v[sel & bad[b], 1].median() if bool((sel & bad[b]).any()) else float("nan")))
h1 = 0.5 * v[:, 0] ** 2; h2 = 0.5 * v[:, 1] ** 2; hx = 0.5 * (P ** 2).sum(1)
u = 6e-8
A = (h2 - h1) - (65 * u * h1 + 91 * u * hx) - (24 * u * h1 + 36 * u * hx) - (24 * u * h2 + 36 * u * hx)
print("by the certificate's own arithmetic (exact distances): A <= 0 for %.2f %% of cloud 0's queries; uncertified in the run: %.2f %%" % (100 * (A <= 0).float().mean().item(), 100 * bad[b].float().mean().item()))
print("quantiles of A / (d1 + d2) [m of allowed motion]:", [float("%.2e" % x) for x in torch.quantile((A / (v[:, 0] + v[:, 1])), torch.tensor([0.01, 0.05, 0.25, 0.5], dtype=torch.float64, device="cuda")).tolist()])
print("budget quantiles of the certified:", [float("%.2e" % x) for x in torch.quantile(q[b][~bad[b]].double(), torch.tensor([0.01, 0.05, 0.25, 0.5], dtype=torch.float64, device="cuda")).tolist()])
