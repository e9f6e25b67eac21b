"""Pick the thread count for bench.py's cpu_baseline: time the oracle (1 cloud x 1 iteration fwd+bwd,
n=m=16384) at several torch thread counts on this host."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dicp_amd.synthetic import make_pairs
from oracle import dicp_oracle as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
src, tgt = make_pairs(1, n, n, seed=3)
for th in (8, 16, 32, 64, 128, 256):
    if th > (os.cpu_count() or 1):
        break
    torch.set_num_threads(th)
    best = 1e9
    for _ in range(2):
        s, t = src.clone().requires_grad_(True), tgt.clone().requires_grad_(True)
        t0 = time.time()
        r = O.icp_batched(s, t, torch.eye(4)[None], torch.ones(1, n), icp_type="pt2pl", differentiable=True, max_iterations=1,
                          tolerance=1e-12, trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=3, const_iter=True)
        r["T"].sum().backward()
        best = min(best, time.time() - t0)
    print("threads %3d: %.3f s per cloud-iteration -> %.2f cloud-it/s" % (th, best, 1.0 / best), flush=True)
