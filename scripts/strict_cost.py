"""What ICP.strict_errors costs: a backward pass that used the one-launch tail waits for its own kernels and reads the tail's error word before it returns, so
the host cannot run ahead of the GPU into the next step.  Calls back to back (one synchronisation at the END of 20 calls, as a training loop that never
looks at a result would run them), B = 256 x 16384 pt2pl + Huber, fwd + bwd.  usage: python scripts/strict_cost.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs
B, n = 256, 16384
src, tgt = make_pairs(B, n, n, seed=3); src, tgt = src.cuda(), tgt.cuda()
T0 = torch.eye(4, device="cuda").repeat(B, 1, 1)


def run(K, const_iter, strict, calls=20):
    icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=1e-12 if const_iter else 1e-4)
    icp.const_iter, icp.strict_errors = const_iter, strict

    def call():
        s, t = src.detach().requires_grad_(True), tgt.detach().requires_grad_(True)
        o = icp.icp(s, t, T0, trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0})
        o["T"].sum().backward()
        return o
    for _ in range(6):
        call()
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(calls):
            o = call()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / calls)
    return best * 1e3, int(icp.knn_stats.get("bwd_tail_from", 0))


for rnd in range(2):
    for K, ci in ((20, True), (10, True), (50, False)):
        a, ta = run(K, ci, True)
        b, tb = run(K, ci, False)
        print("%s K=%d: strict_errors=True %.3f ms per call | False %.3f ms   (+%.1f %%; the tail takes the iterations below %d)" % (
            "constant" if ci else "tolerance", K, a, b, 100.0 * (a - b) / b, ta), flush=True)
