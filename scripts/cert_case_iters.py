"""What a certificate hard case searches again per iteration.  usage: python scripts/cert_case_iters.py <case>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from dicp_amd import _lib
from dicp_amd.ICP import ICP
import test_gpu_configs as TG
src, tgt, K = TG._cert_case(sys.argv[1], torch.float32)
N, n = src.shape[0], src.shape[1]
icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=1e-12); icp.const_iter = True
icp.knn_variant, icp._tuning["cert_hint"] = _lib.KNN_SWEEP, False
out = icp.icp(src.cuda(), tgt.cuda(), torch.eye(4).cuda().repeat(N, 1, 1), **TG.KW); torch.cuda.synchronize()
a = icp.knn_stats["searched_again"]
print(sys.argv[1], "N", N, "n", n, "K", K, "units per cloud", (n + 127) // 128)
print(" units searched again per iteration :", a[:K, :64].sum(1).tolist())
print(" single queries per iteration       :", a[:K, 64:].sum(1).tolist())
print(" clouds off at the end              :", int(icp.knn_stats["certs_off"].sum()), "| step norms of cloud 0:", [float("%.2e" % v) for v in out["deltas"][0, :, :, 0].norm(dim=1).tolist()])
