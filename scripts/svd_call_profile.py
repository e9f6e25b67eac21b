"""configs[1] with the SVD step (ICP.pt2pt_dICP_SVD, 32 x 4096 points, K = 10): wall, host time, and -- under rocprofv3 --kernel-trace -- the kernels of a call."""
import sys
import time

import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs

N, n, K = 32, 4096, 10
src, tgt = make_pairs(N, n, n, seed=2, dtype=torch.float32)
S, Tg = src.cuda(), tgt[:, :, :3].contiguous().cuda()
Ti = torch.eye(4).repeat(N, 1, 1).cuda()
icp = ICP(icp_type="pt2pt", differentiable=True, max_iterations=K, tolerance=1e-12)
icp.const_iter = True


def call():
    s, t = S.detach().requires_grad_(True), Tg.detach().requires_grad_(True)
    icp.pt2pt_dICP_SVD(s, t, Ti, trim_dist=5.0)[1].sum().backward()


for _ in range(20):
    call()
torch.cuda.synchronize()
reps = 100
t0 = time.perf_counter()
host = 0.0
for _ in range(reps):
    a = time.perf_counter()
    call()
    host += time.perf_counter() - a
torch.cuda.synchronize()
print("back to back: wall %.3f ms per call, host %.3f ms per call" % ((time.perf_counter() - t0) / reps * 1e3, host / reps * 1e3))
ts = []
for _ in range(20):
    torch.cuda.synchronize()
    a = time.perf_counter()
    call()
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - a)
ts.sort()
print("synchronised on both sides: median %.3f ms per call" % (ts[10] * 1e3))
if len(sys.argv) > 1:
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(100):
        call()
        torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(25)
