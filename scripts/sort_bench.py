import os, sys, torch
sys.path.insert(0, "/root/repo")
from dicp_amd import _ops
from dicp_amd.synthetic import make_pairs
src, tgt = make_pairs(256, 16384, 16384, seed=3)
tgt = tgt.cuda()
for flag in (0, 1, 0, 1):
    _ops.NATIVE_SORT = flag
    ts = []
    for r in range(8):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); sw = _ops.SweepIndex(tgt); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    print("NATIVE_SORT=%d SweepIndex build: median %.3f ms min %.3f" % (flag, sorted(ts)[4], min(ts)))
