"""Ragged list inputs (targets padded with far rows, ICP.py:460): speed and pruning of the sweep path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs
B, n, K = 32, 16384, 10
src, tgt = make_pairs(B, n, n, seed=3)
g = torch.Generator().manual_seed(0)
lens = torch.randint(6000, n + 1, (B,), generator=g).tolist()
S = [src[b, :lens[b]].cuda() for b in range(B)]
Tg = [tgt[b, :max(5000, lens[b] - 3000)].cuda() for b in range(B)]
T0 = [torch.eye(4).cuda() for _ in range(B)]
icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=1e-12); icp.const_iter = True
def call():
    s = [x.detach().requires_grad_(True) for x in S]; t = [x.detach().requires_grad_(True) for x in Tg]
    out = icp.icp(s, t, T0, trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}); out["T"].sum().backward(); return out
for _ in range(3): call()
ts = []
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter(); call(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
pairs = float(icp.knn_stats["knn_pairs"].sum())
real = float(sum(a * max(5000, a - 3000) for a in lens))
print("ragged lists B=%d (6000..16384 source points, targets 3000 shorter): %.3f ms/iteration; pairs scored: %.2f %% of the padded n*m, %.2f %% of the real n_b*m_b"
      % (B, sorted(ts)[2] * 1e3 / K, 100 * pairs / (float(B) * n * n * K), 100 * pairs / (real * K)))
# the same clouds as one DENSE batch of the mean size, for the per-pair comparison
nm = int(sum(lens) / B); mm = int(sum(max(5000, a - 3000) for a in lens) / B)
Sd, Td = src[:, :nm].contiguous().cuda(), tgt[:, :mm].contiguous().cuda()
T0d = torch.eye(4).cuda().repeat(B, 1, 1)
def dense():
    s, t = Sd.detach().requires_grad_(True), Td.detach().requires_grad_(True)
    out = icp.icp(s, t, T0d, trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}); out["T"].sum().backward(); return out
for _ in range(3): dense()
td = []
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter(); dense(); torch.cuda.synchronize(); td.append(time.perf_counter() - t0)
print("dense batch of the mean size (%d x %d): %.3f ms/iteration" % (nm, mm, sorted(td)[2] * 1e3 / K))
