"""The headline call (256 x 16384 pt2pl + Huber, fwd + bwd of T.sum()) as ONE captured hipGraph (dicp_amd.graphed.graphed_icp_step) against eager calls back to back:
what the host's ~1 ms per eager call is worth when the host is the slow side.  usage: python scripts/graphed_headline.py [K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dicp_amd.ICP import ICP
from dicp_amd.graphed import graphed_icp_step
from dicp_amd.synthetic import make_pairs
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B, n = 256, 16384
src, tgt = make_pairs(B, n, n, seed=3); src, tgt = src.cuda(), tgt.cuda()
T0 = torch.eye(4, device="cuda").repeat(B, 1, 1)
kw = dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0})
icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=1e-12); icp.const_iter = True
s, t = src.detach().requires_grad_(True), tgt.detach().requires_grad_(True)


def eager():
    a, b = src.detach().requires_grad_(True), tgt.detach().requires_grad_(True)
    o = icp.icp(a, b, T0, **kw); o["T"].sum().backward(); return o, a.grad, b.grad


for _ in range(6):
    o_e, gs_e, gt_e = eager()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30):
    eager()
torch.cuda.synchronize(); e_ms = (time.perf_counter() - t0) / 30 * 1e3
step = graphed_icp_step(icp, lambda o: o["T"].sum(), s, t, T0, num_warmup_iters=6, **kw)
for _ in range(3):
    out, grads = step(s, t, T0)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30):
    out, grads = step(s, t, T0)
torch.cuda.synchronize(); g_ms = (time.perf_counter() - t0) / 30 * 1e3
try:
    step.check_errors()
except Exception as e:
    w = icp.knn_stats.get("bwd_tail_error")
    print("check_errors raised:", type(e).__name__, "word 0x%08x" % (int(w.item()) & 0xffffffff if w is not None else 0), "tail_from", icp.knn_stats.get("bwd_tail_from"),
          "finite grads:", bool(torch.isfinite(grads["source"]).all()), bool(torch.isfinite(grads["target"]).all()), "live", icp.knn_stats.get("bwd_live"))
same = torch.equal(out["T"], o_e["T"])
dg = float((grads["source"] - gs_e).abs().max() / gs_e.abs().max()), float((grads["target"] - gt_e).abs().max() / gt_e.abs().max())
print("K=%d: eager %.3f ms per call (%.0f cloud-it/s); captured step %.3f ms per replay (%.0f cloud-it/s); T identical: %s; gradients differ by %.1e / %.1e of their size" % (
    K, e_ms, B * K / e_ms * 1e3, g_ms, B * K / g_ms * 1e3, same, dg[0], dg[1]))
