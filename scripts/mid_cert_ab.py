"""BASELINE configs[1] (B=32 x 4096, pt2pt) and a 65536-point batch: the call with and without match certificates (same box, interleaved)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs
def bench(B, n, icp_type, K, reuse, reps=9):
    src, tgt = make_pairs(B, n, n, seed=3)
    src, tgt = src.cuda(), tgt.cuda()
    if icp_type == "pt2pt":
        tgt = tgt[:, :, :3].contiguous()
    T0 = torch.eye(4, device="cuda").repeat(B, 1, 1)
    icp = ICP(icp_type=icp_type, differentiable=True, max_iterations=K, tolerance=1e-12); icp.const_iter = True; icp.reuse_matches = reuse
    kw = dict(trim_dist=5.0) if icp_type == "pt2pt" else dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0})
    def call():
        s, t = src.detach().requires_grad_(True), tgt.detach().requires_grad_(True)
        icp.icp(s, t, T0, **kw)["T"].sum().backward()
    for _ in range(3): call()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); call(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[reps // 2] * 1e3
for (B, n, typ, K) in ((32, 4096, "pt2pt", 10), (32, 4096, "pt2pt", 30), (64, 8192, "pt2pl", 10), (64, 65536, "pt2pl", 5), (64, 65536, "pt2pl", 10)):
    r = [(bench(B, n, typ, K, True), bench(B, n, typ, K, False)) for _ in range(2)]
    print("B=%d n=%d %s K=%d: %s ms per call with certificates, %s without" % (B, n, typ, K, [round(a, 3) for a, _ in r], [round(b, 3) for _, b in r]), flush=True)
