"""Probe: does a hipGraph replay of the whole fixed-shape call (icp() + backward()) beat the stream launches?"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs
B, n, typ, K = (int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])) if len(sys.argv) > 4 else (256, 16384, "pt2pl", 10)
src, tgt = make_pairs(B, n, n, seed=3)
if typ == "pt2pt":
    tgt = tgt[:, :, :3].contiguous()
src, tgt = src.cuda().requires_grad_(True), tgt.cuda().requires_grad_(True)
T0 = torch.eye(4, device="cuda").repeat(B, 1, 1)
icp = ICP(icp_type=typ, differentiable=True, max_iterations=K, tolerance=1e-12)
icp.const_iter = True
kw = dict(trim_dist=5.0) if typ == "pt2pt" else dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0})
print("B=%d n=%d %s K=%d" % (B, n, typ, K))


def call():
    src.grad = None; tgt.grad = None
    out = icp.icp(src, tgt, T0, **kw)
    out["T"].sum().backward()
    return out["T"]


def timed(fn, reps=7):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3


for _ in range(4):
    call()
print("stream launches: %.3f ms per call" % timed(call))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        call()
torch.cuda.current_stream().wait_stream(s)
try:
    with torch.cuda.graph(g):
        T_static = call()
    ref = call().clone()
    g.replay(); torch.cuda.synchronize()
    print("graph replay == stream call:", float((T_static - ref).abs().max()))
    print("graph replay: %.3f ms per call" % timed(g.replay))
except Exception as e:
    print("capture failed:", type(e).__name__, str(e)[:300])
