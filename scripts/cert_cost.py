"""Forward call time with certificates (switch on / off) and without, on the cases where certificates could lose.  usage: python scripts/cert_cost.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from dicp_amd import _lib
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs, make_scene_pairs
import test_gpu_configs as TG
def fwd_ms(src, tgt, K, reuse, backoff):
    N = src.shape[0]
    icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=1e-12); icp.const_iter = True
    icp.knn_variant, icp.reuse_matches, icp._tuning["cert_backoff"] = _lib.KNN_SWEEP, reuse, backoff
    s, t, T0 = src.cuda(), tgt.cuda(), torch.eye(4).cuda().repeat(N, 1, 1)
    ts = []
    for _ in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter(); icp.icp(s, t, T0, **TG.KW); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    off = icp.knn_stats.get("certs_off")
    return sorted(ts)[3] * 1e3, (None if off is None else int(off.sum()))
cases = [(c,) + TG._cert_case(c, torch.float32) for c in ("near_duplicates", "slow_convergence", "far_from_origin", "duplicated_targets")]
cases.append(("scene B=256 K=10",) + make_scene_pairs(256, 16384, 16384, seed=3) + (10,))
cases.append(("random B=256 K=10",) + make_pairs(256, 16384, 16384, seed=3) + (10,))
cases.append(("random B=32 n=4096 pt2pl K=10",) + make_pairs(32, 4096, 4096, seed=3) + (10,))
for name, src, tgt, K in cases:
    a, _ = fwd_ms(src, tgt, K, False, True)
    b, _ = fwd_ms(src, tgt, K, True, False)
    c, off = fwd_ms(src, tgt, K, True, True)
    print("%-30s forward call, %2d iterations: without certificates %.3f ms | certificates, no switch %.3f ms (%+.1f %%) | with the per-cloud switch %.3f ms (%+.1f %%), off for %s of %d clouds"
          % (name, K, a, b, 100 * (b / a - 1), c, 100 * (c / a - 1), off, src.shape[0]), flush=True)
