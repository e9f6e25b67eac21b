"""Kernel-time breakdown of one certificate hard case, with and without certificates (rocprofv3 --kernel-trace --stats around it).
usage: python3 scripts/cert_case_trace.py <case> <reuse 0|1>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from dicp_amd import _lib
from dicp_amd.ICP import ICP
import test_gpu_configs as TG
src, tgt, K = TG._cert_case(sys.argv[1], torch.float32)
N = src.shape[0]
icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=1e-12); icp.const_iter = True
icp.knn_variant, icp.reuse_matches = _lib.KNN_SWEEP, sys.argv[2] == "1"
s, t, T0 = src.cuda(), tgt.cuda(), torch.eye(4).cuda().repeat(N, 1, 1)
for _ in range(6):
    icp.icp(s, t, T0, **TG.KW); torch.cuda.synchronize()
print(sys.argv[1], "reuse", sys.argv[2], "N", N, "n", src.shape[1], "K", K, "off", int(icp.knn_stats["certs_off"].sum()) if "certs_off" in icp.knn_stats else None)
