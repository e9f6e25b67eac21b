"""Back-to-back calls on ONE ICP object per mode: the headline shape on pairs, planar scenes and independent scans, constant iterations and tolerance mode, dense
batches and ragged lists, forward + backward.  The inputs never change, so every call must return the SAME bits for T (the forward has no atomics on floats) and
gradients that agree to rounding; anything else -- a non-finite value, a TailTimeout, a T that differs from the first call's -- is a race between launches or
between a call and the buffers an earlier call left behind.  usage: python scripts/soak_modes.py [calls] [mode-substring]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs, make_scene_pairs, make_independent_pairs
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 500
only = sys.argv[2] if len(sys.argv) > 2 else ""
B, n = 256, 16384
MODES = []
for gen in ("pairs", "scene", "indep"):
    for tol in (None, 1e-4):
        for ragged in (False, True):
            if ragged and gen != "indep":
                continue
            MODES.append((gen, tol, ragged))
MODES += [("pairs64", None, False), ("pairs_pt2pt", None, False), ("mid", None, False), ("mid", 1e-4, False)]
bad_total = 0
for gen, tol, ragged in MODES:
    name = "%s %s %s" % (gen, "tolerance" if tol else "K=10", "ragged lists" if ragged else "dense")
    if only and only not in name:
        continue
    typ, dt, Bm, nm = "pt2pl", torch.float32, B, n
    if gen == "pairs64":
        dt, Bm = torch.float64, 32
    if gen == "pairs_pt2pt":
        typ = "pt2pt"
    if gen == "mid":
        Bm, nm, typ = 32, 4096, "pt2pt"
    if gen == "indep":
        S, T = make_independent_pairs(Bm, nm, nm, seed=3, ragged=ragged)
    else:
        S, T = (make_scene_pairs if gen == "scene" else make_pairs)(Bm, nm, nm, seed=3, dtype=dt)
    if typ == "pt2pt" and not isinstance(T, list):
        T = T[:, :, :3].contiguous()
    if isinstance(S, list):
        S, T = [x.cuda() for x in S], [x.cuda() for x in T]
        T0 = [torch.eye(4, device="cuda", dtype=dt)] * Bm
    else:
        S, T = S.cuda(), T.cuda()
        T0 = torch.eye(4, device="cuda", dtype=dt).repeat(Bm, 1, 1)
    icp = ICP(icp_type=typ, differentiable=True, max_iterations=50 if tol else 10, tolerance=tol if tol else 1e-12)
    icp.const_iter = tol is None
    first, bad, t0 = None, 0, time.time()
    cnt = calls if gen != "indep" else max(50, calls // 4)
    try:
        for i in range(cnt):
            if isinstance(S, list):
                s, t = [x.detach().requires_grad_(True) for x in S], [x.detach().requires_grad_(True) for x in T]
            else:
                s, t = S.detach().requires_grad_(True), T.detach().requires_grad_(True)
            o = icp.icp(s, t, T0, trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0})
            Tout = o["T"] if torch.is_tensor(o["T"]) else torch.stack(list(o["T"]))
            Tout.sum().backward()
            if i == 0 or i % 25 == 24 or i == cnt - 1:
                gs = torch.cat([x.grad.reshape(-1) for x in s]) if isinstance(s, list) else s.grad.reshape(-1)
                gt = torch.cat([x.grad.reshape(-1) for x in t]) if isinstance(t, list) else t.grad.reshape(-1)
                ok = bool(torch.isfinite(Tout).all()) and bool(torch.isfinite(gs).all()) and bool(torch.isfinite(gt).all())
                if first is None:
                    first = (Tout.clone(), gs.clone(), gt.clone())
                else:
                    ok = ok and torch.equal(Tout, first[0])
                    for a, b in ((gs, first[1]), (gt, first[2])):
                        ok = ok and float((a - b).abs().max()) <= 5e-5 * float(b.abs().max())
                bad += 0 if ok else 1
        torch.cuda.synchronize()
        icp.icp(S, T, T0, trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0})      # (looks at the last passes' error words)
        msg = ""
    except Exception as e:      # noqa: BLE001 (a soak reports and goes on)
        bad += 1
        msg = "   %s: %s" % (type(e).__name__, str(e)[:200])
    bad_total += bad
    print("%-34s %5d calls  %6.1f s  failed checks %d%s" % (name, cnt, time.time() - t0, bad, msg), flush=True)
print("soak_modes: %s" % ("ok" if bad_total == 0 else "%d FAILED checks" % bad_total))
sys.exit(0 if bad_total == 0 else 1)
