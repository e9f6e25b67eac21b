"""Seeded soak of the match certificates: random shapes / types / modes, the certified loop against searching every query in every
iteration; every output must be identical bit for bit.  usage: python scripts/cert_soak.py [cases] [seed]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dicp_amd import _lib
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for c in range(cases):
    dtype = rng.choice([torch.float32, torch.float32, torch.float64])
    n = rng.choice([2048, 3000, 4096, 6000, 8192, 12000, 16384])
    m = max(2048, int(n * rng.choice([0.6, 1.0, 1.0, 1.5])))
    N = rng.choice([3, 8, 17, 40]) if n * m < 2e8 else rng.choice([3, 8, 17])
    if N * n * m < 1.2e8:
        N = int(1.2e8 // (n * m)) + 1                      # (KNN_AUTO takes the sweep from 1e8 pairs on; it is forced below anyway)
    typ = rng.choice(["pt2pl", "pt2pt"])
    K = rng.randint(7, 14)
    const_iter = rng.random() < 0.6
    ragged = rng.random() < 0.3
    noise = rng.choice([0.0, 0.01, 0.05])
    rot, trans = rng.choice([(0.02, 0.1), (0.05, 0.3), (0.2, 1.0)])
    loss = rng.choice([None, {"name": "huber", "metric": 1.0}, {"name": "cauchy", "metric": 0.5}])
    diff = rng.random() < 0.7
    src, tgt = make_pairs(N, n, m, seed=1000 + c, dtype=dtype, noise=noise, max_rot=rot, max_trans=trans)
    if typ == "pt2pt":
        tgt = tgt[:, :, :3].contiguous()
    if rng.random() < 0.3:
        off = torch.tensor([rng.uniform(-500, 500), rng.uniform(-500, 500), rng.uniform(-50, 50)], dtype=dtype)
        src = src + off; tgt[:, :, :3] += off
    outs = []
    for reuse in (False, True):
        icp = ICP(icp_type=typ, differentiable=diff, max_iterations=K, tolerance=1e-12 if const_iter else 1e-5)
        icp.const_iter = const_iter; icp.reuse_matches = reuse; icp.knn_variant = _lib.KNN_SWEEP
        if ragged:
            ls = [max(300, n - (977 * b) % (n // 2)) for b in range(N)]
            S = [src[b, :ls[b]].cuda().requires_grad_(True) for b in range(N)]
            T = [tgt[b, :max(2048, m - (613 * b) % (m // 3))].cuda().requires_grad_(True) for b in range(N)]
            T0 = [torch.eye(4, dtype=dtype).cuda()] * N
        else:
            S, T, T0 = src.cuda().requires_grad_(True), tgt.cuda().requires_grad_(True), torch.eye(4, dtype=dtype).cuda().repeat(N, 1, 1)
        trim = 5.0
        out = icp.icp(S, T, T0, trim_dist=trim, loss_fn=loss)
        out["T"].sum().backward()
        gs = torch.cat([x.grad.reshape(-1) for x in (S if ragged else [S])])
        outs.append((out, gs, icp.knn_stats))
    a, b = outs
    ok = all(torch.equal(a[0][k], b[0][k]) for k in ("T", "deltas", "weights", "costs", "pc")) and torch.equal(a[0]["stats"]["iterations"], b[0]["stats"]["iterations"])
    gtol = (1e-4 if dtype == torch.float32 else 1e-10) * max(1.0, float(a[1].abs().max()))
    gok = bool((((a[1] - b[1]).abs() <= gtol) | (torch.isnan(a[1]) & torch.isnan(b[1]))).all())      # (hard huber weights at a zero residual: NaN in the reference too)
    cnt = b[2].get("searched_again")
    used = "no certificates" if cnt is None else "units %d, queries %d searched again" % (int(cnt[:, :64].sum()), int(cnt[:, 64:].sum()))
    print("case %2d %s N=%d n=%d m=%d %s K=%d %s%s%s: %s, gradients %s (%s)" % (c, str(dtype)[6:], N, n, m, typ, K, "const" if const_iter else "tol", " ragged" if ragged else "",
          " diff" if diff else " hard", "IDENTICAL" if ok else "DIFFERENT", "ok" if gok else "OFF", used), flush=True)
    bad += (not ok) or (not gok)
print("%d of %d cases failed" % (bad, cases))
sys.exit(1 if bad else 0)
