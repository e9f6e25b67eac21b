"""A few headline-shaped calls (fwd + bwd) for a profiler run.  usage: python3 scripts/call_profile.py [scene|random|indep] [K] [calls] [B]
e.g.  rocprofv3 --kernel-trace --stats -d /tmp/prof -o s --output-format csv -- python3 scripts/call_profile.py scene 10 4"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs, make_scene_pairs, make_independent_pairs
kind = sys.argv[1] if len(sys.argv) > 1 else "scene"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 4
B = int(sys.argv[4]) if len(sys.argv) > 4 else 256
n = 16384
RAGGED = os.environ.get("DICP_RAGGED") == "1"      # (indep only: the clouds as Python lists of different lengths)
src, tgt = (make_independent_pairs(B, n, n, seed=3, ragged=RAGGED) if kind == "indep" else (make_scene_pairs if kind == "scene" else make_pairs)(B, n, n, seed=3))
if kind == "indep" and RAGGED:
    src, tgt = [x.cuda() for x in src], [x.cuda() for x in tgt]
    T0 = [torch.eye(4, device="cuda")] * B
else:
    src, tgt = src.cuda(), tgt.cuda()
    T0 = torch.eye(4, device="cuda").repeat(B, 1, 1)
TOL = os.environ.get("DICP_TOL")            # DICP_TOL=1e-4: a tolerance-mode call (up to K iterations, const_iter off)
icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=float(TOL) if TOL else 1e-12); icp.const_iter = TOL is None
icp.reuse_matches = os.environ.get("DICP_REUSE", "1") == "1"          # (DICP_REUSE=0: search everything in every iteration)
icp._tuning["cert_backoff"] = os.environ.get("DICP_BACKOFF", "1") == "1"
for _ in range(calls):
    s, t = ([x.detach().requires_grad_(True) for x in src], [x.detach().requires_grad_(True) for x in tgt]) if isinstance(src, list) else (src.detach().requires_grad_(True), tgt.detach().requires_grad_(True))
    out = icp.icp(s, t, T0, trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0})
    out["T"].sum().backward()
    torch.cuda.synchronize()
print("certificates %s, per-cloud switch %s; off for %s clouds" % (icp.reuse_matches, icp._tuning["cert_backoff"], int(icp.knn_stats["certs_off"].sum()) if "certs_off" in icp.knn_stats else None))
K = int(out["deltas"].shape[1])
pairs = float(icp.knn_stats["knn_pairs"].sum().item()) / K / (float(B) * n * n)
again = icp.knn_stats.get("searched_again")
print("%s K=%d B=%d: pairs scored %.2f %% of n*m per launch; units / queries searched again per iteration: %s / %s; backward live: %s" % (
    kind, K, B, 100 * pairs, None if again is None else again[:K, :64].sum(1).tolist(), None if again is None else again[:K, 64:].sum(1).tolist(),
    icp.knn_stats["bwd_live"][:K].tolist() if "bwd_live" in icp.knn_stats else None))
