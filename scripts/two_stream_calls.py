"""Two ICP objects on two streams, calls in flight at the same time (forward + backward each), against the same calls on the default stream: poses bit for bit,
gradients to rounding, no TailTimeout (two one-launch tails share the GPU: dicp_bwd_tail_max_blocks allows for one more launch of its kind).
usage: python scripts/two_stream_calls.py [calls]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs, make_scene_pairs
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B, n, K = 128, 16384, 10
data = [[x.cuda() for x in make_pairs(B, n, n, seed=5)], [x.cuda() for x in make_scene_pairs(B, n, n, seed=6)]]
T0 = torch.eye(4, device="cuda").repeat(B, 1, 1)
kw = dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0})
ref = []
for d in data:
    icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=1e-12); icp.const_iter = True
    for _ in range(3):
        s, t = d[0].clone().requires_grad_(True), d[1].clone().requires_grad_(True)
        o = icp.icp(s, t, T0, **kw); o["T"].sum().backward()
    ref.append((o["T"].detach().clone(), s.grad.clone(), t.grad.clone()))
torch.cuda.synchronize()
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
icps = []
for _ in range(2):
    icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=1e-12); icp.const_iter = True
    icps.append(icp)
bad = 0
t0 = time.time()
for i in range(calls):
    res = []
    for j in (0, 1):
        with torch.cuda.stream(streams[j]):
            s, t = data[j][0].clone().requires_grad_(True), data[j][1].clone().requires_grad_(True)
            o = icps[j].icp(s, t, T0, **kw)
            o["T"].sum().backward()
            res.append((o, s, t))
    if i % 20 == 19 or i == calls - 1:
        torch.cuda.synchronize()
        for j in (0, 1):
            o, s, t = res[j]
            ok = torch.equal(o["T"], ref[j][0])
            for g, e in ((s.grad, ref[j][1]), (t.grad, ref[j][2])):
                ok = ok and bool(torch.isfinite(g).all()) and float((g - e).abs().max()) <= 3e-5 * float(e.abs().max())
            bad += 0 if ok else 1
torch.cuda.synchronize()
for j in (0, 1):
    with torch.cuda.stream(streams[j]):
        icps[j].check_errors()
print("two streams: %d calls each in %.1f s, failed checks %d" % (calls, time.time() - t0, bad))
sys.exit(1 if bad else 0)
