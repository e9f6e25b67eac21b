"""Host against GPU time of the headline call (256 x 16384 pt2pl + Huber, fwd + bwd): wall per call back to back, the host's share (time until icp() / backward()
return, nothing waited for), and the GPU's own (a call alone, events around it).  usage: python scripts/headline_host_time.py [K] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
B, n = 256, 16384
src, tgt = make_pairs(B, n, n, seed=3); src, tgt = src.cuda(), tgt.cuda()
T0 = torch.eye(4, device="cuda").repeat(B, 1, 1)
kw = dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0})
icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=1e-12); icp.const_iter = True


def call():
    s, t = src.detach().requires_grad_(True), tgt.detach().requires_grad_(True)
    a = time.perf_counter()
    o = icp.icp(s, t, T0, **kw)
    b = time.perf_counter()
    o["T"].sum().backward()
    return b - a, time.perf_counter() - b


for _ in range(6):
    call()
for rnd in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); hf = hb = 0.0
    for _ in range(reps):
        f, b = call(); hf += f; hb += b
    host = time.perf_counter() - t0
    torch.cuda.synchronize(); wall = time.perf_counter() - t0
    # one call alone, the GPU's own time between two events
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    gp = []
    for _ in range(5):
        torch.cuda.synchronize(); e0.record(); call(); e1.record(); torch.cuda.synchronize(); gp.append(e0.elapsed_time(e1))
    print("K=%d: wall %.3f ms per call back to back; host %.3f (forward %.3f, backward %.3f; the loop left the host after %.3f); one call alone on the GPU %.3f ms" % (
        K, wall / reps * 1e3, (hf + hb) / reps * 1e3, hf / reps * 1e3, hb / reps * 1e3, host / reps * 1e3, sorted(gp)[2]), flush=True)
