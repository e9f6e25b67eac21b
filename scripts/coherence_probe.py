"""Probe: how many queries could KEEP their match from one ICP iteration to the next, certified exactly?
A query's match y1 (distance d1, runner-up distance d2, both exact) survives a motion of the query by at most D if
   0.5 (d2 - D)^2 - 0.5 (d1 + D)^2 > 3 E,   E = 3e-6 (1 + 0.5 |x|^2)   (the rounding margin of a float32 score in centred coordinates).
Benchmark clouds, poses of a 10-iteration call."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs
B, n, K = 4, 16384, 10
src, tgt = make_pairs(B, n, n, seed=3)
src, tgt = src.cuda(), tgt.cuda()
T0 = torch.eye(4, device="cuda").repeat(B, 1, 1)
Ts = [T0]
for k in range(1, K + 1):
    icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=k, tolerance=1e-12); icp.const_iter = True
    Ts.append(icp.icp(src, tgt, T0, trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0})["T"].detach())
ctr = tgt[:, :, :3].median(dim=1).values
prev = None
for k in range(K):
    x = (src.double() @ Ts[k][:, :3, :3].double().transpose(1, 2) + Ts[k][:, None, :3, 3].double())
    d = torch.cdist(x, tgt[:, :, :3].double())
    v, i = torch.topk(d, 2, dim=2, largest=False)
    d1, d2, idx = v[:, :, 0], v[:, :, 1], i[:, :, 0]
    if prev is not None:
        xp, d1p, d2p, idxp = prev
        D = (x - xp).norm(dim=2)
        E = 3e-6 * (1 + 0.5 * ((xp - ctr[:, None, :].double()) ** 2).sum(-1))
        cert = ((d2p - D) > 0) & (0.5 * (d2p - D) ** 2 - 0.5 * (d1p + D) ** 2 > 3 * E)
        same = idx == idxp
        print("iteration %d: motion max %.2e median %.2e | certifiable %.2f %% (of them unchanged: %.4f %%) | matches unchanged overall %.2f %% | d1 median %.3f d2 median %.3f"
              % (k, float(D.max()), float(D.median()), 100 * float(cert.float().mean()), 100 * float(same[cert].float().mean()) if cert.any() else 0.0,
                 100 * float(same.float().mean()), float(d1.median()), float(d2.median())), flush=True)
    prev = (x, d1, d2, idx)
