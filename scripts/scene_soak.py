"""Back-to-back calls on planar scenes (B = 256, n = m = 16384, K = 20, forward + backward), a finiteness check every 50 calls:
the soak that found the two races of round 6 (profiles/r06_scene_soak.txt).  python scripts/scene_soak.py <mode> <calls>, mode one of
default | notail | nocert | nograd | nof16 (combinations by substring: "notail_nof16")."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dicp_amd import _ops
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_scene_pairs
mode, calls = sys.argv[1], int(sys.argv[2])
B, n, K = 256, 16384, 20
S, T = make_scene_pairs(B, n, n, seed=3)
S, T = S.cuda(), T.cuda()
T0 = torch.eye(4, device="cuda").repeat(B, 1, 1)
if "nof16" in mode:
    _ops.F16_SWEEP = False
icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=1e-12); icp.const_iter = True
if "notail" in mode:
    icp._tuning["bwd_tail"] = False
if mode == "nocert":
    icp.reuse_matches = False
if mode == "nograd":
    pass
t0 = time.time()
bad = 0
for i in range(calls):
    if mode == "nograd":
        with torch.no_grad():
            o = icp.icp(S, T, T0, trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0})
    else:
        s, t = S.detach().requires_grad_(True), T.detach().requires_grad_(True)
        o = icp.icp(s, t, T0, trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0})
        o["T"].sum().backward()
    if i % 50 == 49:
        torch.cuda.synchronize()
        fin = bool(torch.isfinite(o["T"]).all()) and (mode == "nograd" or (bool(torch.isfinite(s.grad).all()) and bool(torch.isfinite(t.grad).all())))
        bad += 0 if fin else 1
        print("%s: %d calls, %.1f s, nonfinite checks %d" % (mode, i + 1, time.time() - t0, bad), flush=True)
torch.cuda.synchronize()
print("%s: done %d calls" % (mode, calls), flush=True)
