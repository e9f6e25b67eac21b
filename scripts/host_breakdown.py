"""Where the HOST time of a call goes: every C-ABI entry point and the tensor allocations wrapped with a timer (steady state, calls enqueued
back to back so the GPU never makes the host wait).  usage: python scripts/host_breakdown.py [B n icp_type K]"""
import sys, os, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dicp_amd import _lib, _ops
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs
B, n, typ, K = (int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])) if len(sys.argv) > 4 else (32, 4096, "pt2pt", 10)
src, tgt = make_pairs(B, n, n, seed=3); src, tgt = src.cuda(), tgt.cuda()
if typ == "pt2pt":
    tgt = tgt[:, :, :3].contiguous()
T0 = torch.eye(4, device="cuda").repeat(B, 1, 1)
icp = ICP(icp_type=typ, differentiable=True, max_iterations=K, tolerance=1e-12); icp.const_iter = True
kw = dict(trim_dist=5.0) if typ == "pt2pt" else dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0})
def call():
    s, t = src.detach().requires_grad_(True), tgt.detach().requires_grad_(True)
    icp.icp(s, t, T0, **kw)["T"].sum().backward()
for _ in range(20): call()
torch.cuda.synchronize()
acc, cnt = collections.defaultdict(float), collections.defaultdict(int)
class Wrapped:
    def __init__(self, lib): self._lib = lib
    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        def timed(*a):
            t0 = time.perf_counter(); r = fn(*a); acc[name] += time.perf_counter() - t0; cnt[name] += 1; return r
        return timed
real = _lib.load()
wrapped = Wrapped(real)
_lib._lib = wrapped
for nm in ("empty", "zeros", "empty_like", "zeros_like"):
    f = getattr(torch, nm)
    def mk(f, nm):
        def g(*a, **k):
            t0 = time.perf_counter(); r = f(*a, **k); acc["torch." + nm] += time.perf_counter() - t0; cnt["torch." + nm] += 1; return r
        return g
    setattr(torch, nm, mk(f, nm))
R = 200
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(R): call()
host = (time.perf_counter() - t0) / R
torch.cuda.synchronize()
print("B=%d n=%d %s K=%d: %.1f us of host time per call (calls enqueued back to back, timers on)" % (B, n, typ, K, host * 1e6))
tot = 0.0
for name, t in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("  %-32s %5.1f calls  %7.1f us per call of the loop  (%.1f us each)" % (name, cnt[name] / R, t / R * 1e6, t / cnt[name] * 1e6))
    tot += t
print("  sum of the wrapped pieces: %.1f us; the rest (Python logic, autograd, struct building, views): %.1f us" % (tot / R * 1e6, (host - tot / R) * 1e6))
