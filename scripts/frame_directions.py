"""Which sort direction dicp_search_frame picks per generator, with and without the queries, and what the first search then scores.
usage: python scripts/frame_directions.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dicp_amd import _ops
from dicp_amd.synthetic import make_pairs, make_scene_pairs, make_independent_pairs

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = 16384
DIRS = torch.tensor([[1, 0, 0], [0, 1, 0], [0, 0, 1], [0.6, 0.64, 0.48], [0.6, -0.64, 0.48], [0.48, 0.6, -0.64]], dtype=torch.float32, device="cuda")


def which(F):
    return (F[:, :3] @ DIRS.T).argmax(dim=1)


for name, gen in (("make_pairs", lambda: make_pairs(B, n, n, seed=3)), ("make_scene_pairs", lambda: make_scene_pairs(B, n, n, seed=3)),
                  ("make_independent_pairs", lambda: make_independent_pairs(B, n, n, seed=3, ragged=False))):
    S, T = gen()
    S, T = S.cuda(), T.cuda()
    T0 = torch.eye(4, device="cuda").repeat(B, 1, 1).contiguous()
    for label, kw in (("target only", {}), ("with the queries", dict(src=S, T_init=T0))):
        F = _ops.search_frame(T, **kw)
        sw = _ops.SweepIndex(T, frame=F)
        pose = _ops.search_pose(_ops._pose_from_T(T0), F)
        qo = sw.query_order(S, pose)
        sw.knn(S, pose, qorder=qo, cfg=2)
        torch.cuda.synchronize()
        print("%-24s %-18s directions %s  pairs scored by iteration 0's search %.4f of n*m" % (
            name, label, torch.bincount(which(F), minlength=6).tolist(), float(sw.pairs) / (float(B) * n * n)), flush=True)
