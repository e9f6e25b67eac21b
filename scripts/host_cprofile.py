import sys, os, cProfile, pstats, io
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/dicp_amd") else os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs
B, n, typ, K = 32, 4096, "pt2pt", 10
src, tgt = make_pairs(B, n, n, seed=3); src, tgt = src.cuda(), tgt.cuda()
tgt = tgt[:, :, :3].contiguous()
T0 = torch.eye(4, device="cuda").repeat(B, 1, 1)
icp = ICP(icp_type=typ, differentiable=True, max_iterations=K, tolerance=1e-12); icp.const_iter = True
def call():
    s, t = src.detach().requires_grad_(True), tgt.detach().requires_grad_(True)
    icp.icp(s, t, T0, trim_dist=5.0)["T"].sum().backward()
for _ in range(20): call()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(200): call()
torch.cuda.synchronize(); pr.disable()
out = io.StringIO(); st = pstats.Stats(pr, stream=out); st.sort_stats("tottime").print_stats(45)
print("\n".join(l for l in out.getvalue().splitlines() if l.strip()))
