"""Host-side cost of a mid-size / tiny call: cProfile of the steady-state call, wall time per call, and the GPU's own busy time (kernel
durations from events around the call are not separable here: compare wall time with `rocprofv3 --kernel-trace` of the same call).
usage: python scripts/host_profile.py [B n icp_type K]"""
import sys, os, cProfile, pstats, time, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs
B, n, typ, K = (int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])) if len(sys.argv) > 4 else (32, 4096, "pt2pt", 10)
src, tgt = make_pairs(B, n, n, seed=3); src, tgt = src.cuda(), tgt.cuda()
if typ == "pt2pt":
    tgt = tgt[:, :, :3].contiguous()
T0 = torch.eye(4, device="cuda").repeat(B, 1, 1)
icp = ICP(icp_type=typ, differentiable=True, max_iterations=K, tolerance=1e-12); icp.const_iter = True
kw = dict(trim_dist=5.0) if typ == "pt2pt" else dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0})
def call():
    s, t = src.detach().requires_grad_(True), tgt.detach().requires_grad_(True)
    icp.icp(s, t, T0, **kw)["T"].sum().backward()
for _ in range(20): call()
torch.cuda.synchronize()
ts = []
for _ in range(30):
    t0 = time.perf_counter(); call(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
ts.sort()
# host-only time: enqueue without waiting for the GPU (the queue drains behind)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30): call()
t_host = (time.perf_counter() - t0) / 30
torch.cuda.synchronize()
print("B=%d n=%d %s K=%d: %.3f ms per call (median of 30, synchronised); %.3f ms of host time per call when 30 calls are enqueued back to back" % (B, n, typ, K, ts[15] * 1e3, t_host * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(50): call()
torch.cuda.synchronize(); pr.disable()
out = io.StringIO(); pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(22)
print("\n".join(l for l in out.getvalue().splitlines() if l.strip())[:6000])
