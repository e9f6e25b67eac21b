"""What ICP.deterministic costs at the benchmark shape (256 x 16384, point-to-plane + huber + trim, K = 10, fwd + bwd), and that two runs agree bit for bit.
usage (MI355X): PYTHONPATH=. python scripts/det_cost.py > profiles/rNN_deterministic_cost.txt"""
import time
import torch
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs

B, n, K = 256, 16384, 10
src, tgt = make_pairs(B, n, n, seed=1, dtype=torch.float32)
src, tgt = src.cuda(), tgt.cuda()
T0 = torch.eye(4, device="cuda").repeat(B, 1, 1)
kw = dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=3)


def call(icp):
    s, t = src.detach().requires_grad_(True), tgt.detach().requires_grad_(True)
    o = icp.icp(s, t, T0, **kw)
    o["T"].sum().backward()
    return o, s.grad, t.grad


for det in (False, True):
    icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=1e-12)
    icp.const_iter, icp.deterministic = True, det
    for _ in range(4):
        call(icp)
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); call(icp); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    a, b = call(icp), call(icp)
    torch.cuda.synchronize()
    same = torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    d = max(float((a[1] - b[1]).abs().max() / a[1].abs().max()), float((a[2] - b[2]).abs().max() / a[2].abs().max()))
    print("deterministic=%-5s  %.3f ms per call (median of 7)   two runs bit-identical: %s   largest difference / largest gradient: %.2e" % (det, sorted(ts)[3] * 1e3, same, d))
