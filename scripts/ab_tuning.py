"""A/B of the private mechanism switches (ICP._tuning) on one box: B = 256 x 16384 pt2pl huber, fwd + bwd, median of 9 calls after 4 warm-ups, interleaved.
usage: python scripts/ab_tuning.py [K] [switch ...]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
names = sys.argv[2:] or ["bwd_tail", "first_search", "cert_sets", "cert_hint", "cert_backoff", "plan_call"]
B, n = 256, 16384
src, tgt = make_pairs(B, n, n, seed=3); src, tgt = src.cuda(), tgt.cuda()
T0 = torch.eye(4, device="cuda").repeat(B, 1, 1)


def bench(switch, value):
    icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=1e-12); icp.const_iter = True
    if switch:
        icp._tuning[switch] = value

    def call():
        s, t = src.detach().requires_grad_(True), tgt.detach().requires_grad_(True)
        o = icp.icp(s, t, T0, trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}); o["T"].sum().backward(); return o
    for _ in range(4):
        call()
    ts = []
    for _ in range(9):
        torch.cuda.synchronize(); t0 = time.perf_counter(); call(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[4] * 1e3


for rnd in range(2):
    base = bench(None, None)
    print("K=%d all on: %.3f ms" % (K, base), flush=True)
    for nm in names:
        print("   %-14s off: %.3f ms  (%+.1f %%)" % (nm, bench(nm, False), 100 * (bench(nm, False) / base - 1)), flush=True)
