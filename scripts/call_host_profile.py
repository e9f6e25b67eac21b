"""cProfile of the host side of eager configs[1] calls on the one-call path (32 x 4096 x 4096, K = 10, forward + backward)."""
import cProfile
import pstats
import sys

import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs

N, n, K = 32, 4096, 10
src, tgt = make_pairs(N, n, n, seed=3, dtype=torch.float32)
S, Tg = src.cuda().requires_grad_(True), tgt.cuda()
Ti = torch.eye(4).repeat(N, 1, 1).cuda().requires_grad_(True)
kw = dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=3)
icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=1e-12)
icp.const_iter = True
icp._tuning["one_call"] = (sys.argv[1:] + ["1"])[0] == "1"
for _ in range(30):
    icp.icp(S, Tg, Ti, **kw)["T"].sum().backward()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    icp.icp(S, Tg, Ti, **kw)["T"].sum().backward()
    torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
