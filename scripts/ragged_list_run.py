import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_independent_pairs
B, n = 256, 16384
RAG = os.environ.get("RAGGED", "1") == "1"
S, Tg = make_independent_pairs(B, n, n, seed=3, dtype=torch.float32, ragged=RAG)
if RAG:
    S, Tg = [x.cuda() for x in S], [x.cuda() for x in Tg]; T0 = [torch.eye(4, device="cuda")] * B
else:
    S, Tg = S.cuda(), Tg.cuda(); T0 = torch.eye(4, device="cuda").repeat(B, 1, 1)
icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=10, tolerance=1e-12); icp.const_iter = True
def call():
    if RAG:
        s_ = [x.detach().requires_grad_(True) for x in S]; t_ = [x.detach().requires_grad_(True) for x in Tg]
    else:
        s_, t_ = S.detach().requires_grad_(True), Tg.detach().requires_grad_(True)
    o = icp.icp(s_, t_, T0, trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=3)
    o["T"].sum().backward()
    torch.cuda.synchronize()
for _ in range(6): call()
