import sys, os, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs
B, n, K = 256, 65536, 5
src, tgt = make_pairs(B, n, n, seed=3); src, tgt = src.cuda(), tgt.cuda()
T0 = torch.eye(4, device="cuda").repeat(B, 1, 1)
for knn, name in ((0, "auto(sweep)"), (2, "mfma brute force")):
    icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=1e-12); icp.const_iter = True; icp.knn_variant = knn
    def call():
        s, t = src.detach().requires_grad_(True), tgt.detach().requires_grad_(True)
        icp.icp(s, t, T0, trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0})["T"].sum().backward()
    for _ in range(2): call()
    torch.cuda.synchronize(); ts = []
    for _ in range(3):
        t0 = time.perf_counter(); call(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    t = sorted(ts)[1]
    print("configs[3] B=%d n=m=%d pt2pl K=%d f+b, kNN %s: %.2f ms/call  %.3f ms/iteration  %.0f cloud-it/s  peak mem %.1f GB" % (B, n, K, name, t * 1e3, t * 1e3 / K, B * K / t, torch.cuda.max_memory_allocated() / 2**30), flush=True)
