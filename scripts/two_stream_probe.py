"""Would running the batch as G independent groups on G streams hide the dispatch gaps and the one-wave kernels?  Whole call fwd+bwd, headline shape."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs
B, n, K = 256, 16384, int(sys.argv[1]) if len(sys.argv) > 1 else 10
src, tgt = make_pairs(B, n, n, seed=3)
src, tgt = src.cuda(), tgt.cuda()
T0 = torch.eye(4, device="cuda").repeat(B, 1, 1)
KW = dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0})
def make():
    icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=1e-12); icp.const_iter = True
    return icp
def run(G):
    icps = [make() for _ in range(G)]
    streams = [torch.cuda.Stream() for _ in range(G)]
    bounds = [(g * B // G, (g + 1) * B // G) for g in range(G)]
    def call():
        s, t = src.detach().requires_grad_(True), tgt.detach().requires_grad_(True)
        cur = torch.cuda.current_stream()
        outs = []
        for g, (a, b) in enumerate(bounds):
            if G > 1:
                streams[g].wait_stream(cur)
                with torch.cuda.stream(streams[g]):
                    outs.append(icps[g].icp(s[a:b], t[a:b], T0[a:b], **KW)["T"].sum())
            else:
                outs.append(icps[g].icp(s[a:b], t[a:b], T0[a:b], **KW)["T"].sum())
        if G > 1:
            for st in streams: cur.wait_stream(st)
        sum(outs).backward()
        return s.grad
    for _ in range(5): call()
    ts = []
    for _ in range(9):
        torch.cuda.synchronize(); t0 = time.perf_counter(); call(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[4] * 1e3
for rnd in range(2):
    print("K=%d  " % K + "   ".join("%d group(s): %.3f ms" % (G, run(G)) for G in (1, 2, 3, 4)), flush=True)
