"""Search forms on make_independent_pairs (partially overlapping, independently sampled clouds with big start poses): ms per call, fwd + bwd, K = 10.
usage: python scripts/indep_forms.py [B n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dicp_amd import _lib, _ops
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_independent_pairs

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
K = 10
S, T = make_independent_pairs(B, n, n, seed=3, ragged=False)
S, T = S.cuda(), T.cuda()
T0 = torch.eye(4, device="cuda").repeat(B, 1, 1)
KW = dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=3)


def run(name, knn, reuse, f16_min=None):
    old = _ops.F16_SWEEP_MIN_TARGETS
    if f16_min is not None:
        _ops.F16_SWEEP_MIN_TARGETS = f16_min
    try:
        icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=1e-12)
        icp.const_iter, icp.knn_variant, icp.reuse_matches = True, knn, reuse
        ts = []
        for i in range(5):
            s, t = S.detach().requires_grad_(True), T.detach().requires_grad_(True)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = icp.icp(s, t, T0, **KW)
            out["T"].sum().backward()
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        fr = float(icp.knn_stats["knn_pairs"].sum().item()) / K / (float(B) * n * n) if "knn_pairs" in icp.knn_stats else float("nan")
        print("%-34s %8.2f ms per call  %8.0f cloud-it/s  pairs %.3f  T[0,0,3]=%.6f" % (name, sorted(ts)[2] * 1e3, B * K / sorted(ts)[2], fr, float(out["T"][0, 0, 3])), flush=True)
    finally:
        _ops.F16_SWEEP_MIN_TARGETS = old


run("sweep, certificates", _lib.KNN_SWEEP, True)
run("sweep, no certificates", _lib.KNN_SWEEP, False)
run("f16 sweep, certificates", _lib.KNN_SWEEP, True, 0)
run("f16 sweep, no certificates", _lib.KNN_SWEEP, False, 0)
run("brute force, matrix cores", _lib.KNN_MFMA, False)
run("brute force, VALU", _lib.KNN_VALU, False)
