"""cProfile of the host side of tolerance-mode calls at the headline shape (B = 256 x 16384, tolerance 1e-4, up to 50 iterations): the mode in which the
host cannot run ahead of the GPU (ICP.py:259's all-converged check), so its time between learning K and the backward's first launch is exposed."""
import cProfile
import pstats
import sys

import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs

B, n = int((sys.argv[1:] + ["256"])[0]), 16384
src, tgt = make_pairs(B, n, n, seed=3, dtype=torch.float32)
S, Tg = src.cuda(), tgt.cuda()
Ti = torch.eye(4).repeat(B, 1, 1).cuda()
kw = dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=3)
icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=50, tolerance=1e-4)
icp.const_iter = False


def call():
    s, t = S.detach().requires_grad_(True), Tg.detach().requires_grad_(True)
    icp.icp(s, t, Ti, **kw)["T"].sum().backward()
    torch.cuda.synchronize()


for _ in range(10):
    call()
import time
t0 = time.perf_counter()
for _ in range(20):
    call()
print("wall %.3f ms per call" % ((time.perf_counter() - t0) / 20 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    call()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(32)
