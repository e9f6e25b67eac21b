"""Per-call timing of the headline call: forward / backward, with the backward's negligible-cotangent skip on and off.
usage: python scripts/run_timing.py [K ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs
B, n = 256, 16384
Ks = [int(a) for a in sys.argv[1:]] or [10, 20]
src, tgt = make_pairs(B, n, n, seed=3); src, tgt = src.cuda(), tgt.cuda()
T0 = torch.eye(4, device="cuda").repeat(B, 1, 1)
for K in Ks:
    for name, kw in (("full reverse sweep", dict(bwd_skip_eps=0.0)), ("negligible-cotangent skip", dict(bwd_skip_eps=None))):
        icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=1e-12); icp.const_iter = True
        for k, v in kw.items(): setattr(icp, k, v)
        def call():
            s, t = src.detach().requires_grad_(True), tgt.detach().requires_grad_(True)
            out = icp.icp(s, t, T0, trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}); torch.cuda.synchronize(); t1 = time.perf_counter()
            out["T"].sum().backward(); torch.cuda.synchronize(); return t1, s.grad
        for _ in range(4): call()
        fw, bw = [], []
        for _ in range(7):
            torch.cuda.synchronize(); t0 = time.perf_counter(); t1, g = call(); t2 = time.perf_counter(); fw.append(t1 - t0); bw.append(t2 - t1)
        fw.sort(); bw.sort()
        live = icp.knn_stats.get("bwd_live")
        print("K=%2d %-26s forward %.3f ms  backward %.3f ms  total %.3f ms  (%.0f cloud-it/s)  finite %s  clouds at work per backward iteration %s" % (
            K, name, fw[3] * 1e3, bw[3] * 1e3, (fw[3] + bw[3]) * 1e3, B * K / (fw[3] + bw[3]), bool(torch.isfinite(g).all()), None if live is None else live[:K].tolist()), flush=True)
