"""Seeded soak of the match certificates (`-m gpu`): random shapes / dtypes / modes -- float32 and float64, both ICP types,
every loss, hard and soft weights, ragged lists, tolerance mode, clouds far from the origin -- the certified loop against
searching every query in every iteration (what the reference does, nn.py:32-35).  Every output must be identical bit for
bit, gradients to rounding.  scripts/cert_soak.py runs the same generator for any number of cases / other seeds."""
import random

import pytest
import torch

from dicp_amd import _lib
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs

DEV = "cuda"

pytestmark = pytest.mark.gpu
CASES = 24


def case_config(c, seed=1):
    """The c-th case of the stream seeded by `seed` (drawn sequentially, so case c does not depend on how many are run)."""
    rng = random.Random(seed)
    cfg = None
    for _ in range(c + 1):
        dtype = rng.choice([torch.float32, torch.float32, torch.float64])
        n = rng.choice([2048, 3000, 4096, 6000, 8192, 12000, 16384])
        m = max(2048, int(n * rng.choice([0.6, 1.0, 1.0, 1.5])))
        N = rng.choice([3, 8, 17, 40]) if n * m < 2e8 else rng.choice([3, 8, 17])
        if N * n * m < 1.2e8:
            N = int(1.2e8 // (n * m)) + 1
        cfg = dict(dtype=dtype, n=n, m=m, N=N, typ=rng.choice(["pt2pl", "pt2pt"]), K=rng.randint(7, 14), const_iter=rng.random() < 0.6,
                   ragged=rng.random() < 0.3, noise=rng.choice([0.0, 0.01, 0.05]), motion=rng.choice([(0.02, 0.1), (0.05, 0.3), (0.2, 1.0)]),
                   loss=rng.choice([None, {"name": "huber", "metric": 1.0}, {"name": "cauchy", "metric": 0.5}]), diff=rng.random() < 0.7,
                   offset=[rng.uniform(-500, 500), rng.uniform(-500, 500), rng.uniform(-50, 50)] if rng.random() < 0.3 else None)
    return cfg


def run_case(c, seed=1, dev="cuda"):
    """-> (identical, gradients_ok, description)"""
    g = case_config(c, seed)
    dtype, n, m, N, K = g["dtype"], g["n"], g["m"], g["N"], g["K"]
    src, tgt = make_pairs(N, n, m, seed=1000 * seed + c, dtype=dtype, noise=g["noise"], max_rot=g["motion"][0], max_trans=g["motion"][1])
    if g["typ"] == "pt2pt":
        tgt = tgt[:, :, :3].contiguous()
    if g["offset"] is not None:
        off = torch.tensor(g["offset"], dtype=dtype)
        src = src + off
        tgt[:, :, :3] += off
    outs = []
    for reuse in (False, True):
        icp = ICP(icp_type=g["typ"], differentiable=g["diff"], max_iterations=K, tolerance=1e-12 if g["const_iter"] else 1e-5)
        icp.const_iter, icp.reuse_matches, icp.knn_variant = g["const_iter"], reuse, _lib.KNN_SWEEP
        if g["ragged"]:
            ls = [max(300, n - (977 * b) % (n // 2)) for b in range(N)]
            S = [src[b, :ls[b]].to(dev).requires_grad_(True) for b in range(N)]
            T = [tgt[b, :max(2048, m - (613 * b) % (m // 3))].to(dev).requires_grad_(True) for b in range(N)]
            T0 = [torch.eye(4, dtype=dtype, device=dev)] * N
        else:
            S, T, T0 = src.to(dev).requires_grad_(True), tgt.to(dev).requires_grad_(True), torch.eye(4, dtype=dtype, device=dev).repeat(N, 1, 1)
        out = icp.icp(S, T, T0, trim_dist=5.0, loss_fn=g["loss"])
        out["T"].sum().backward()
        gs = torch.cat([x.grad.reshape(-1) for x in (S if g["ragged"] else [S])])
        gt = torch.cat([x.grad.reshape(-1) for x in (T if g["ragged"] else [T])])
        outs.append((out, gs, gt, dict(icp.knn_stats)))
    a, b = outs
    same = all(torch.equal(a[0][k], b[0][k]) for k in ("T", "deltas", "weights", "costs", "pc")) and torch.equal(a[0]["stats"]["iterations"], b[0]["stats"]["iterations"])
    gok = True
    for ga, gb in ((a[1], b[1]), (a[2], b[2])):
        gtol = (1e-4 if dtype == torch.float32 else 1e-10) * max(1.0, float(ga[torch.isfinite(ga)].abs().max()) if bool(torch.isfinite(ga).any()) else 1.0)
        # (hard Huber weights at an exactly zero residual: NaN in the reference too)
        gok = gok and bool((((ga - gb).abs() <= gtol) | (torch.isnan(ga) & torch.isnan(gb))).all())
    cnt = b[3].get("searched_again")
    used = "no certificates" if cnt is None else "units %d, queries %d searched again" % (int(cnt[:, :64].sum()), int(cnt[:, 64:].sum()))
    what = "%s N=%d n=%d m=%d %s K=%d %s%s%s%s" % (str(dtype)[6:], N, n, m, g["typ"], K, "const" if g["const_iter"] else "tol", " ragged" if g["ragged"] else "",
                                                   " diff" if g["diff"] else " hard", " offset" if g["offset"] is not None else "")
    return same, gok, what + " (" + used + ")"


@pytest.mark.parametrize("c", range(CASES))
def test_certified_loop_equals_searching_everything(c):
    same, gok, what = run_case(c)
    assert same, what
    assert gok, what


@pytest.mark.parametrize("seed,N,n,K,icp_type,dtype", [(11, 12, 16384, 10, "pt2pl", torch.float32), (12, 7, 8192, 14, "pt2pt", torch.float32),
                                                       (13, 5, 12000, 9, "pt2pl", torch.float64), (14, 20, 16384, 8, "pt2pl", torch.float32)])
def test_candidate_sets_on_planar_scenes(seed, N, n, K, icp_type, dtype):
    """Planar scenes are where matches have runner-ups within the scores' rounding (dense surfaces): with candidate sets (ICP._tuning["cert_sets"]) such queries are
    re-scored among four certified rows instead of searched.  Poses, weights, costs and the transformed cloud bit for bit those of searching every query
    in every iteration, gradients to rounding -- and the sets really are in use (single-query searches collapse after the iteration that makes them)."""
    from dicp_amd.synthetic import make_scene_pairs
    src, tgt = make_scene_pairs(N, n, n, seed=seed, dtype=dtype)
    if icp_type == "pt2pt":
        tgt = tgt[:, :, :3].contiguous()
    kw = dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=3)
    outs = []
    for reuse in (False, True):
        icp = ICP(icp_type=icp_type, differentiable=True, max_iterations=K, tolerance=1e-12)
        icp.const_iter, icp.reuse_matches, icp.knn_variant = True, reuse, _lib.KNN_SWEEP
        S, Tg = src.to(DEV).requires_grad_(True), tgt.to(DEV).requires_grad_(True)
        out = icp.icp(S, Tg, torch.eye(4, dtype=dtype, device=DEV).repeat(N, 1, 1), **kw)
        out["T"].sum().backward()
        outs.append((out, S.grad, Tg.grad, dict(icp.knn_stats)))
    for key in ("T", "deltas", "weights", "costs", "pc"):
        assert torch.equal(outs[0][0][key], outs[1][0][key]), key
    for i in (1, 2):
        scale = max(1.0, float(outs[0][i].abs().max()))
        assert float((outs[0][i] - outs[1][i]).abs().max()) <= (2e-6 if dtype == torch.float32 else 1e-12) * scale
    singles = outs[1][3]["searched_again"][:K, 64:].sum(1).tolist()
    if dtype == torch.float32:
        assert max(singles) > 50 * N and singles[-1] * 10 < max(singles), singles       # made once, then re-scored


@pytest.mark.parametrize("n,K", [(65536, 50), (16384, 130)])
def test_single_cloud_with_duplicated_targets_fills_its_guard_list(n, K):
    """One cloud (N = 1: the guard launches' work lists are sized per batch, and a single cloud can append an entry per unit of the sweep AND one per 64 of its
    candidate sets -- round 5's advice: twice the old capacity, the overflow was dropped silently).  Every target twice (each match has an exact runner-up: no
    certificate, a candidate set) and a planar scene (dense surfaces), a start pose that keeps it moving: the certified loop must still equal searching
    everything, bit for bit."""
    from dicp_amd.synthetic import make_scene_pairs
    src, tgt = make_scene_pairs(1, n, n // 2, seed=21, dtype=torch.float32, max_rot=0.1, max_trans=0.5)
    tgt = torch.cat((tgt, tgt), dim=1)[:, torch.randperm(n, generator=torch.Generator().manual_seed(3))].contiguous()
    kw = dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=3)
    outs = []
    for reuse in (False, True):
        icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=1e-12)
        icp.const_iter, icp.reuse_matches, icp.knn_variant = True, reuse, _lib.KNN_SWEEP
        S, Tg = src.to(DEV).requires_grad_(True), tgt.to(DEV).requires_grad_(True)
        out = icp.icp(S, Tg, torch.eye(4, device=DEV).repeat(1, 1, 1), **kw)
        out["T"].sum().backward()
        outs.append((out, S.grad, Tg.grad, dict(icp.knn_stats)))
    assert "searched_again" in outs[1][3] and "searched_again" not in outs[0][3]      # (certificates were in use)
    for key in ("T", "deltas", "weights", "costs", "pc"):
        assert torch.equal(outs[0][0][key], outs[1][0][key]), key
    for i in (1, 2):
        scale = max(1.0, float(outs[0][i].abs().max()))
        assert float((outs[0][i] - outs[1][i]).abs().max()) <= 2e-5 * scale
