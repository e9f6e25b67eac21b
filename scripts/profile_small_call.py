import sys, os, cProfile, pstats, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs
src, tgt = make_pairs(1, 65, 65, seed=3); src, tgt = src.cuda(), tgt.cuda()
T0 = torch.eye(4, device="cuda").repeat(1, 1, 1)
icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=10, tolerance=1e-12); icp.const_iter = True
def call():
    s, t = src.detach().requires_grad_(True), tgt.detach().requires_grad_(True)
    icp.icp(s, t, T0, trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0})["T"].sum().backward()
for _ in range(20): call()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(50): call()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(40)
