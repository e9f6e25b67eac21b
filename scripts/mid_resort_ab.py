"""BASELINE configs[1] (B=32 x 4096, pt2pt): query re-order schedule of the sweep, whole call fwd+bwd, interleaved on one box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs
def bench(B, n, typ, K, resort, reps=15):
    src, tgt = make_pairs(B, n, n, seed=3)
    src, tgt = src.cuda(), (tgt[:, :, :3].contiguous() if typ == "pt2pt" else tgt).cuda()
    T0 = torch.eye(4, device="cuda").repeat(B, 1, 1)
    icp = ICP(icp_type=typ, differentiable=True, max_iterations=K, tolerance=1e-12); icp.const_iter = True; icp._tuning["sweep_resort"] = resort
    kw = dict(trim_dist=5.0) if typ == "pt2pt" else dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0})
    def call():
        s, t = src.detach().requires_grad_(True), tgt.detach().requires_grad_(True)
        icp.icp(s, t, T0, **kw)["T"].sum().backward()
    for _ in range(4): call()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); call(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[reps // 2] * 1e3
for (B, n, typ, K) in ((32, 4096, "pt2pt", 10), (32, 4096, "pt2pt", 30), (64, 8192, "pt2pl", 10), (16, 16384, "pt2pl", 10)):
    for rnd in range(2):
        print("B=%d n=%d %s K=%d: " % (B, n, typ, K) + "  ".join("%s %.3f" % (r, bench(B, n, typ, K, r)) for r in ((0, 1, 2, 3), (0, 1, 2), (0, 2), (0, 1), (0,))), flush=True)
