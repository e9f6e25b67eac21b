import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs
B, n, K = 32, 4096, 10
src, tgt = make_pairs(B, n, n, seed=3); src, tgt = src.cuda(), tgt[:, :, :3].contiguous().cuda()
T0 = torch.eye(4, device="cuda").repeat(B, 1, 1)
icp = ICP(icp_type="pt2pt", differentiable=True, max_iterations=K, tolerance=1e-12); icp.const_iter = True
def call():
    s, t = src.detach().requires_grad_(True), tgt.detach().requires_grad_(True)
    icp.icp(s, t, T0, trim_dist=5.0)["T"].sum().backward()
for _ in range(5): call()
torch.cuda.synchronize()
time.sleep(0.01)
call(); torch.cuda.synchronize()
if len(sys.argv) > 1:
    import cProfile, pstats
    for _ in range(3): call()
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(20): call()
    pr.disable(); torch.cuda.synchronize()
    st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(16); st.sort_stats("cumulative").print_stats(30)
    t0 = time.perf_counter()
    for _ in range(50): call()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("host time per call %.1f us; with final sync %.1f us" % ((t1 - t0) / 50 * 1e6, (t2 - t0) / 50 * 1e6))
