"""graphed_icp against the eager call: same results, time per forward + backward.  usage: python scripts/graphed_timing.py [B n icp_type K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dicp_amd.ICP import ICP
from dicp_amd.graphed import graphed_icp
from dicp_amd.synthetic import make_pairs
B, n, typ, K = (int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])) if len(sys.argv) > 4 else (32, 4096, "pt2pt", 10)
src, tgt = make_pairs(B, n, n, seed=3)
if typ == "pt2pt":
    tgt = tgt[:, :, :3].contiguous()
src, tgt = src.cuda(), tgt.cuda()
T0 = torch.eye(4, device="cuda").repeat(B, 1, 1)
icp = ICP(icp_type=typ, differentiable=True, max_iterations=K, tolerance=1e-12); icp.const_iter = True
kw = dict(trim_dist=5.0) if typ == "pt2pt" else dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0})
def eager():
    s, t = src.detach().requires_grad_(True), tgt.detach().requires_grad_(True)
    out = icp.icp(s, t, T0, **kw); out["T"].sum().backward(); return out["T"], s.grad, t.grad
g = graphed_icp(icp, src.detach().requires_grad_(True), tgt.detach().requires_grad_(True), T0, **kw)
def graphed():
    s, t = src.detach().requires_grad_(True), tgt.detach().requires_grad_(True)
    out = g(s, t, T0); out["T"].sum().backward(); return out["T"], s.grad, t.grad
from dicp_amd.graphed import graphed_icp_step
st = graphed_icp_step(icp, lambda out: out["T"].sum(), src.detach().requires_grad_(True), tgt.detach().requires_grad_(True), T0, **kw)
def stepped():
    out, grads = st(src, tgt, T0); return out["T"], grads["source"], grads["target"]
def timed(fn, reps=31):
    for _ in range(5): fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3
a, b, c = eager(), graphed(), stepped()
same = [float((x - y).abs().max()) for x, y in zip(a, b)] + [float((x - y).abs().max()) for x, y in zip(a, c)]
print("B=%d n=%d %s K=%d: eager %.3f ms | graphed_icp (forward graph + backward graph, loss outside) %.3f ms | graphed_icp_step (one graph, loss inside) %.3f ms per forward + backward; max |difference| of T / source.grad / target.grad against eager: %s" % (B, n, typ, K, timed(eager), timed(graphed), timed(stepped), same), flush=True)
