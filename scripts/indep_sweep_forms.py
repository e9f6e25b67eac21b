"""VALU sweep against the matrix-core sweep on make_independent_pairs under the pose after k iterations: pairs scored and time per launch, identical matches.
usage: python scripts/indep_sweep_forms.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dicp_amd import _ops
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_independent_pairs

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = 16384
dev = "cuda"
S, T = make_independent_pairs(B, n, n, seed=3, ragged=False)
S, T = S.to(dev), T.to(dev)
T0 = torch.eye(4, device=dev).repeat(B, 1, 1).contiguous()


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


frame = _ops.search_frame(T, src=S, T_init=T0)
sw = _ops.SweepIndex(T, frame=frame)
for K in (0, 1, 3, 6):
    if K:
        icp = ICP(icp_type="pt2pl", differentiable=False, max_iterations=K, tolerance=1e-12); icp.const_iter = True
        Tk = icp.icp(S, T, T0, trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0})["T"]
    else:
        Tk = T0
    ps = _ops.search_pose(_ops._pose_from_T(Tk), frame)
    qo = sw.query_order(S, ps)
    res = {}
    for mf in (False, True):
        idx = torch.empty((B, n), dtype=torch.int32, device=dev); spos = torch.empty_like(idx)
        sw.pair_shards.zero_()
        sw.knn(S, ps, qo, out=idx, cfg=2, spos=spos, mfma=mf)
        torch.cuda.synchronize()
        fr = float(sw.pairs.item()) / (float(B) * n * n)
        t = timed(lambda: sw.knn(S, ps, qo, out=idx, cfg=2, spos=spos, mfma=mf))
        res[mf] = (idx.clone(), spos.clone(), fr, t)
    bad = int((res[False][0] != res[True][0]).sum()) + int((res[False][1] != res[True][1]).sum())
    print("pose after %d iterations: valu %.3f ms (%.2f %% of the pairs)  mfma %.3f ms (%.2f %%)  mismatches %d" % (
        K, res[False][3], 100 * res[False][2], res[True][3], 100 * res[True][2], bad), flush=True)
