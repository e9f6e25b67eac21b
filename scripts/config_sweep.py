"""Throughput of the ICP call at the BASELINE.json configs other than the headline one (steady state, fwd+bwd)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs

def run(B, n, icp_type, K, knn, svd=False, fwd_only=False):
    src, tgt = make_pairs(B, n, n, seed=3)
    src, tgt = src.cuda(), tgt.cuda()
    if icp_type == "pt2pt":
        tgt = tgt[:, :, :3].contiguous()
    T0 = torch.eye(4, device="cuda").repeat(B, 1, 1)
    icp = ICP(icp_type=icp_type, differentiable=True, max_iterations=K, tolerance=1e-12)
    icp.const_iter = True
    icp.knn_variant = knn
    kw = dict(trim_dist=5.0) if icp_type == "pt2pt" else dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0})
    def call():
        s, t = src.detach().requires_grad_(not fwd_only), tgt.detach().requires_grad_(not fwd_only)
        if svd:
            ps, T = icp.pt2pt_dICP_SVD(s, t, T0, trim_dist=5.0)
        else:
            T = icp.icp(s, t, T0, **kw)["T"]
        if not fwd_only:
            T.sum().backward()
    for _ in range(2):
        call()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); call(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    t = sorted(ts)[2]
    print("B=%4d n=m=%6d %-5s %-9s K=%2d %s: %8.3f ms/call  %7.3f ms/iter  %10.0f cloud-it/s" %
          (B, n, icp_type, {0: "auto", 1: "brute", 2: "mfma", 3: "sweep"}[knn], K, "SVD " if svd else ("fwd " if fwd_only else "f+b "), t * 1e3, t * 1e3 / K, B * K / t), flush=True)

run(32, 4096, "pt2pt", 10, 0)            # configs[1] Gauss-Newton
run(32, 4096, "pt2pt", 10, 1)
run(32, 4096, "pt2pt", 10, 0, svd=True)  # configs[1] SVD step
run(1, 65, "pt2pl", 10, 0)               # one tiny pair (latency floor)
run(256, 16384, "pt2pl", 10, 0)          # configs[2]
run(256, 16384, "pt2pl", 10, 0, fwd_only=True)
run(64, 65536, "pt2pl", 5, 0)            # configs[3] cloud size (quarter batch)
run(64, 65536, "pt2pl", 5, 2)            # ... with the MFMA brute-force kNN the config names
run(2048, 16384, "pt2pl", 5, 0)          # configs[4]: the whole 8-GPU batch on one GPU (memory check)
