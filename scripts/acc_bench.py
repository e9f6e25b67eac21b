"""Where the forward accumulate kernel's time goes at the benchmark shape: real matches vs an identity index
(coalesced target rows), with and without the weight output."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dicp_amd import _lib, _loop, _ops
from dicp_amd.synthetic import make_pairs

B = int(os.environ.get("B", 256)); n = int(os.environ.get("NPTS", 16384)); rounds = int(os.environ.get("ROUNDS", 9))
lib = _lib.load()
src, tgt = make_pairs(B, n, n, seed=3)
src, tgt = src.cuda(), tgt.cuda()
c = tgt.shape[2]
dt, code = src.dtype, _lib.F32
sw = _ops.SweepIndex(tgt)
qo = sw.query_order(src, None)
idx = torch.empty((B, n), dtype=torch.int32, device="cuda")
pose = torch.tensor([[1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0]] * B, dtype=dt, device="cuda")
sw.knn(src, pose, qo, out=idx)
ident = torch.arange(n, dtype=torch.int32, device="cuda").repeat(B, 1).contiguous()
w0 = torch.ones((B, n), dtype=dt, device="cuda")
wout = torch.empty((B, n), dtype=dt, device="cuda")
P = _loop.LoopConfig(icp_type="pt2pl", differentiable=True, max_iterations=1, tolerance=0.0, trim_dist=5.0, loss_name="huber",
                    loss_metric=1.0, dim=3, const_iter=True, tanh_steepness=10.0, match_ratio_thresh=0.01).params()
p, st = _ops._p, _ops._stream()
nb = lib.dicp_accumulate_blocks(n)
part = torch.empty((B, nb, _lib.NACC_PAD), dtype=dt, device="cuda")


def run(ix, w):
    _lib.check(lib.dicp_accumulate(code, ctypes.byref(P), p(src), p(tgt), c, p(ix), p(pose), p(w0), None, None, B, n, n, p(part), p(w), n, st), "acc")


cases = {"real idx + w": lambda: run(idx, wout), "real idx, no w": lambda: run(idx, None),
         "identity idx + w": lambda: run(ident, wout), "rows mode (idx NULL) + w": lambda: run(None, wout)}
times = {k: [] for k in cases}
for rnd in range(rounds + 1):
    for name, fn in cases.items():
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        if rnd:
            times[name].append(a.elapsed_time(b))
for name, ts in times.items():
    ts = sorted(ts)
    print("%-28s median %.4f ms  min %.4f ms" % (name, ts[len(ts) // 2], ts[0]))
