"""Where the time goes inside dicp_icp_backward_run: per-block, per-iteration timestamps (s_memrealtime, 100 MHz) of the headline call.
usage: python scripts/run_phases.py [K] [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dicp_amd import _ops
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs
K = int(sys.argv[1]) if len(sys.argv) > 1 else 12
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
n = 16384
_ops.RUN_DEBUG = True
src, tgt = make_pairs(B, n, n, seed=3); src, tgt = src.cuda(), tgt.cuda()
T0 = torch.eye(4, device="cuda").repeat(B, 1, 1)
icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=1e-12); icp.const_iter = True
for rep in range(3):
    s, t = src.detach().requires_grad_(True), tgt.detach().requires_grad_(True)
    out = icp.icp(s, t, T0, trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0})
    out["T"].sum().backward()
    torch.cuda.synchronize()
d = icp.knn_stats["bwd_run_debug"].cpu().double() * 0.01        # microseconds
lo, hi = icp.knn_stats["bwd_run"]
print("run", (lo, hi), "timestamps", tuple(d.shape))
t0 = d[..., 0].min()
wait, comp, arr = d[..., 1] - d[..., 0], d[..., 2] - d[..., 1], d[..., 3] - d[..., 2]
its = d.shape[2]
print("per iteration of the run (mean over blocks; us):   top->matches known (step + wait) | adjoint of 4 slots | reduce + publish + arrive | whole iteration")
for i in range(its):
    print("  it %2d: %7.2f | %7.2f | %7.2f | %7.2f     (block start spread %.1f us)" % (i, wait[:, :, i].mean(), comp[:, :, i].mean(), arr[:, :, i].mean(),
          (d[:, :, i, 3] - d[:, :, i, 0]).mean(), d[:, :, i, 0].max() - d[:, :, i, 0].min()))
per_cloud = (d[:, :, -1, 3].amax(dim=1) - d[:, :, 0, 0].amin(dim=1))
print("a cloud's whole run: mean %.1f us, min %.1f, max %.1f; launch groups end at" % (per_cloud.mean(), per_cloud.min(), per_cloud.max()),
      [round(float(d[c0:c0 + 64, :, -1, 3].max() - t0), 1) for c0 in range(0, B, 64)], "us after the first block started")
