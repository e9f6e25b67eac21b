"""Where the host's own ~1 ms of a headline call goes: every stage function of dicp_amd._loop (and ICP.dICP around them) wrapped in a timer, a synchronisation
before every call so that nothing waits for the GPU.  usage: python scripts/headline_host_stages.py [K]"""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dicp_amd._loop as L
import dicp_amd._ops as O
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs
K = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B, n = 256, 16384
src, tgt = make_pairs(B, n, n, seed=3); src, tgt = src.cuda(), tgt.cuda()
T0 = torch.eye(4, device="cuda").repeat(B, 1, 1)
kw = dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0})
acc = collections.OrderedDict()


def wrap(mod, name):
    f = getattr(mod, name)

    def g(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
    setattr(mod, name, g)


for nm in [x for x in dir(L) if x.startswith("_fwd_") or x.startswith("_bwd_")] + ["backward_once", "order_by_matches"]:
    wrap(L, nm)
wrap(O, "prebuild_search")
icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=1e-12); icp.const_iter = True
tot = {"icp()": 0.0, "T.sum()": 0.0, "backward()": 0.0}


def call(timed):
    s, t = src.detach().requires_grad_(True), tgt.detach().requires_grad_(True)
    torch.cuda.synchronize()
    a = time.perf_counter(); o = icp.icp(s, t, T0, **kw); b = time.perf_counter(); l = o["T"].sum(); c = time.perf_counter(); l.backward(); d = time.perf_counter()
    if timed:
        tot["icp()"] += b - a; tot["T.sum()"] += c - b; tot["backward()"] += d - c


for _ in range(8):
    call(False)
acc.clear()
R = 40
for _ in range(R):
    call(True)
print("K=%d, per call (us):" % K)
for k, v in tot.items():
    print("  %-28s %7.1f" % (k, v / R * 1e6))
for k, v in acc.items():
    print("    %-26s %7.1f" % (k, v / R * 1e6))
