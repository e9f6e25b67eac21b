"""Where the certifying search goes (ICP._tuning["sweep_resort"] / ["cert_from"]): the headline shape in the three modes of the bench line, pairs and scenes,
median of 15 back-to-back timed calls after 5 warm-ups, variants interleaved.  usage: python scripts/cert_start_ab.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs, make_scene_pairs
B, n = 256, 16384
T0 = torch.eye(4, device="cuda").repeat(B, 1, 1)
VARIANTS = [("default (0,1,2,3)", {}), ("resort (0,1,2)", {"sweep_resort": (0, 1, 2)}), ("resort (0,1)", {"sweep_resort": (0, 1)}), ("resort (0,2)", {"sweep_resort": (0, 2)})]
if os.environ.get("AB") == "2":
    VARIANTS = [("default (0,1,2,3)", {}), ("(0,1,2) cert 3", {"sweep_resort": (0, 1, 2), "cert_from": 3}), ("(0,1,3)", {"sweep_resort": (0, 1, 3)}), ("(0,2,3)", {"sweep_resort": (0, 2, 3)}),
                ("(0,1) cert 3", {"sweep_resort": (0, 1), "cert_from": 3})]
MODES = [("K=10", 10, None), ("K=20", 20, None), ("tolerance", 50, 1e-4)]
if os.environ.get("ONLY_TOL") == "1":
    MODES = MODES[2:]


def bench(src, tgt, K, tol, tune):
    icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=tol if tol else 1e-12); icp.const_iter = tol is None
    icp._tuning.update(tune)

    def call():
        s, t = src.detach().requires_grad_(True), tgt.detach().requires_grad_(True)
        o = icp.icp(s, t, T0, trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}); o["T"].sum().backward(); return o
    for _ in range(5):
        o = call()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(15):
        o = call()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 15 * 1e3, int(o["deltas"].shape[1]), o["T"]


for gen_name, gen in (("pairs", make_pairs), ("scenes", make_scene_pairs)):
    src, tgt = gen(B, n, n, seed=3); src, tgt = src.cuda(), tgt.cuda()
    for mname, K, tol in MODES:
        ref = None
        for rnd in range(2):
            for vname, tune in VARIANTS:
                ms, Kx, T = bench(src, tgt, K, tol, tune)
                if ref is None:
                    ref = T
                print("%-7s %-10s %-18s %.3f ms per call (%d iterations, %.0f cloud-iterations/s)%s" % (gen_name, mname, vname, ms, Kx, B * Kx / ms * 1e3, "" if torch.equal(T, ref) else "   T DIFFERS"), flush=True)
