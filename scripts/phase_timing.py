"""Where does a 10-iteration ICP call spend wall time?  (host overhead vs kernels)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dicp_amd.ICP import ICP
from dicp_amd import _ops
from dicp_amd.synthetic import make_pairs
B, n, K = 256, 16384, 10
src, tgt = make_pairs(B, n, n, seed=3)
src, tgt = src.cuda(), tgt.cuda()
T0 = torch.eye(4, device="cuda").repeat(B, 1, 1)
icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=1e-12)
icp.const_iter = True
kw = dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=3)
def sync(): torch.cuda.synchronize()
for rep in range(3):
    s = src.detach().requires_grad_(True); t = tgt.detach().requires_grad_(True)
    sync(); t0 = time.perf_counter()
    out = icp.icp(s, t, T0, **kw)
    t1 = time.perf_counter(); sync(); t2 = time.perf_counter()
    loss = out["T"].sum()
    loss.backward()
    t3 = time.perf_counter(); sync(); t4 = time.perf_counter()
    print("fwd: host-return %.2f ms, gpu-done %.2f ms | bwd: host-return %.2f ms, gpu-done %.2f ms | total %.2f ms" %
          ((t1 - t0) * 1e3, (t2 - t0) * 1e3, (t3 - t2) * 1e3, (t4 - t2) * 1e3, (t4 - t0) * 1e3))
# pieces
sync(); t0 = time.perf_counter(); sw = _ops.SweepIndex(tgt); sync(); print("SweepIndex build %.2f ms" % ((time.perf_counter() - t0) * 1e3))
pose = torch.cat((torch.eye(3, device="cuda").reshape(9), torch.zeros(3, device="cuda"))).repeat(B, 1)
sync(); t0 = time.perf_counter(); qo = sw.query_order(src, pose); sync(); print("query_order %.2f ms" % ((time.perf_counter() - t0) * 1e3))
sync(); t0 = time.perf_counter(); b = icp._batch(src, tgt, T0, None); sync(); print("_batch %.2f ms" % ((time.perf_counter() - t0) * 1e3))
import cProfile, pstats
s = src.detach().requires_grad_(True); t = tgt.detach().requires_grad_(True)
pr = cProfile.Profile(); pr.enable()
out = icp.icp(s, t, T0, **kw); out["T"].sum().backward(); sync()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
