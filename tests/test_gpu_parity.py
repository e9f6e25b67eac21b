"""Parity of the HIP path (libdicp_hip.so through dicp_amd) against the golden vectors
generated from the reference and against the CPU oracle.  Needs an MI355X: `-m gpu`.

Bars (BASELINE.json north_star): pose <= 1e-4, gradients <= 1e-3.  In float64 the kernels
are held far tighter (1e-9 .. 1e-11); float32 runs are held to the north-star bars against
the float64 reference.  kNN indices are integer work: exact.
"""
import numpy as np
import pytest
import torch

from dicp_amd import _lib, _loop, _ops
from dicp_amd.ICP import ICP
from dicp_amd.loss import loss
from dicp_amd.nn import nn
from dicp_amd.synthetic import make_pairs
from oracle import dicp_oracle as O
from oracle.se3 import tran2vec

pytestmark = pytest.mark.gpu
DEV = "cuda"


def t(a, dtype=None, grad=False, dev=DEV):
    x = torch.tensor(np.asarray(a), dtype=dtype, device=dev)
    return x.requires_grad_(True) if grad else x


def npy(x):
    return x.detach().cpu().numpy()


def check_result(res, g, prefix="", atol=1e-10):
    np.testing.assert_allclose(npy(res["T"]), g[prefix + "T"], rtol=0, atol=atol)
    assert res["deltas"].shape == g[prefix + "deltas"].shape
    np.testing.assert_allclose(npy(res["deltas"]), g[prefix + "deltas"], rtol=0, atol=atol)
    np.testing.assert_allclose(npy(res["costs"]), g[prefix + "costs"], rtol=1e-7, atol=atol)
    if prefix + "weights" in g:
        # weights see the pose through tanh(5(tau-|e|)): ~10x the pose error, and the kernels' exact
        # Rodrigues differs from torch.matrix_exp (the reference) by ~4e-12 per iteration
        np.testing.assert_allclose(npy(res["weights"]), g[prefix + "weights"], rtol=0, atol=10 * atol)
    if prefix + "pc" in g:
        np.testing.assert_allclose(npy(res["pc"]), g[prefix + "pc"], rtol=0, atol=atol)
    np.testing.assert_array_equal(npy(res["stats"]["converged"]), g[prefix + "stats_converged"])
    np.testing.assert_allclose(npy(res["stats"]["iterations"]), g[prefix + "stats_iterations"])
    np.testing.assert_allclose(npy(res["stats"]["matched_ratio"]), g[prefix + "stats_matched_ratio"], atol=1e-7)


# ------------------------------------------------------------------------- kNN
def exact_or_tied(idx_gpu, x, y, tol):
    """idx must equal the exact-f64 brute force unless the runner-up is within `tol` (squared distance)."""
    idx, best, second = O.knn_exact_f64(x.cpu(), y.cpu())
    got = idx_gpu.cpu().long()
    bad = got != idx
    if bad.any():
        d_got = ((x.cpu().double() - torch.gather(y.cpu()[:, :, :3].double(), 1, got.unsqueeze(-1).expand(-1, -1, 3))) ** 2).sum(-1)
        assert bool(((d_got - best)[bad] <= tol).all()), "kNN picked a point that is not (nearly) nearest"
    return int(bad.sum())


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("N,n,m", [(1, 1, 1), (3, 70, 90), (2, 257, 16), (5, 1000, 2049), (2, 2100, 4100)])
def test_knn_small_shapes(dtype, N, n, m):
    g = torch.Generator().manual_seed(N * 1000 + n)
    x = (torch.rand((N, n, 3), generator=g, dtype=torch.float64) * 10 - 5).to(dtype).to(DEV)
    y = (torch.rand((N, m, 6), generator=g, dtype=torch.float64) * 10 - 5).to(dtype).to(DEV)
    tgt4 = _ops.pack_target(y)
    idx = _ops.knn(x, None, tgt4, m, _lib.KNN_VALU)
    assert idx.dtype == torch.int32 and int(idx.min()) >= 0 and int(idx.max()) < m
    if dtype == torch.float64:
        assert torch.equal(idx.cpu().long(), O.nn_index(x.cpu(), y.cpu()))
    else:
        assert exact_or_tied(idx, x, y, 1e-4) <= max(1, N * n // 2000)
        assert torch.equal(_ops.knn(x, None, tgt4, m, _lib.KNN_MFMA).cpu(), idx.cpu()) or \
            exact_or_tied(_ops.knn(x, None, tgt4, m, _lib.KNN_MFMA), x, y, 1e-4) <= max(1, N * n // 2000)


def test_knn_ties_take_lowest_index():
    """torch.argmin semantics: duplicates of the nearest target -> first one."""
    y = torch.zeros((1, 40, 3), dtype=torch.float32, device=DEV)
    y[0, :, 0] = torch.arange(40, device=DEV) % 10          # each x value appears 4 times
    x = torch.tensor([[[3.1, 0, 0], [8.9, 0, 0], [0.2, 0, 0]]], dtype=torch.float32, device=DEV)
    tgt4 = _ops.pack_target(y)
    for variant in (_lib.KNN_VALU, _lib.KNN_MFMA):
        assert _ops.knn(x, None, tgt4, 40, variant).cpu().tolist() == [[3, 9, 0]]
    yd = y.double()
    assert _ops.knn(x.double(), None, _ops.pack_target(yd), 40, _lib.KNN_VALU).cpu().tolist() == [[3, 9, 0]]


def test_knn_fused_transform_and_pad_rows():
    """pose is applied inside the kernel (ICP.py:137); target pad rows (ICP.py:460) never win."""
    g = torch.Generator().manual_seed(5)
    x = torch.rand((4, 300, 3), generator=g, dtype=torch.float64) * 4
    y = torch.rand((4, 333, 6), generator=g, dtype=torch.float64) * 4
    y[:, 300:] = 4000.0                                        # what batch_size_handling pads with
    Tm = torch.stack([torch.tensor(np.linalg.inv(np.eye(4))) for _ in range(4)])
    ang = 0.3
    C = torch.tensor([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1.0]])
    r = torch.tensor([0.2, -0.1, 0.3])
    pose = torch.cat((C.reshape(9), r)).repeat(4, 1).to(DEV)
    idx = _ops.knn(x.to(DEV), pose, _ops.pack_target(y.to(DEV)), 333)
    want = O.nn_index(x @ C.T + r, y)
    assert torch.equal(idx.cpu().long(), want) and int(idx.max()) < 300
    del Tm


@pytest.mark.parametrize("variant", [_lib.KNN_VALU, _lib.KNN_MFMA])
def test_knn_big_launch_configs(variant):
    """Exercises the Q=4 / NB=8 instantiations (>= 1M queries), ragged n and m."""
    N, n, m = 66, 16001, 523
    src, tgt = make_pairs(N, n, m, seed=3, dtype=torch.float32)
    sd, td = src.to(DEV), tgt.to(DEV)
    idx = _ops.knn(sd, None, _ops.pack_target(td), m, variant)
    pick = [0, 17, 65]
    assert exact_or_tied(idx[pick], sd[pick], td[pick], 1e-4) <= 8
    # the mid-size instantiations
    idx2 = _ops.knn(sd[:40], None, _ops.pack_target(td[:40]), m, variant)
    assert torch.equal(idx2.cpu(), idx[:40].cpu())


# ------------------------------------------------------------- C1: tests/data pair
@pytest.mark.parametrize("name,icp_type,diff", [
    ("c1_pt2pt_diff", "pt2pt", True),
    ("c1_pt2pl_diff", "pt2pl", True),
    ("c1_pt2pt_hard", "pt2pt", False),
])
def test_c1_float64(golden, name, icp_type, diff):
    """tests/test_ICP.py:35-149 on the HIP path, float64 like the reference."""
    g = golden(name)
    trim, huber, tol, max_iter = g["params"]
    src, tgt = t(g["source"], grad=True), t(g["target"], grad=True)
    icp = ICP(icp_type=icp_type, differentiable=diff, max_iterations=int(max_iter), tolerance=float(tol))
    res = icp.icp(src, tgt, t(g["T_init"]), trim_dist=float(trim), loss_fn={"name": "huber", "metric": float(huber)}, dim=2)
    check_result(res, g)
    err = tran2vec(g["T_ts_true"] @ np.linalg.inv(npy(res["T"])[0]))
    assert np.linalg.norm(err) < float(tol)                                  # test_ICP.py:65-66
    assert np.allclose(npy(res["pc"])[0], g["target"][:, :3], atol=1e-5)     # test_ICP.py:69
    res["T"].sum().backward()
    np.testing.assert_allclose(npy(src.grad), g["grad_source"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(npy(tgt.grad), g["grad_target"], rtol=0, atol=1e-10)
    src2, tgt2 = t(g["source"], grad=True), t(g["target"], grad=True)
    res2 = icp.icp(src2, tgt2, t(g["T_init"]), trim_dist=float(trim), loss_fn={"name": "huber", "metric": float(huber)}, dim=2)
    (res2["pc"] ** 2).sum().backward()
    np.testing.assert_allclose(npy(src2.grad), g["grad_source_pc2"], rtol=0, atol=1e-8)
    np.testing.assert_allclose(npy(tgt2.grad), g["grad_target_pc2"], rtol=0, atol=1e-8)


@pytest.mark.parametrize("name,icp_type", [("c1_pt2pt_diff", "pt2pt"), ("c1_pt2pl_diff", "pt2pl")])
def test_c1_float32_meets_north_star(golden, name, icp_type):
    """float32 kernels vs the float64 reference: pose <= 1e-4, gradients <= 1e-3."""
    g = golden(name)
    trim, huber, tol, _ = g["params"]
    src, tgt = t(g["source"], torch.float32, grad=True), t(g["target"], torch.float32, grad=True)
    icp = ICP(icp_type=icp_type, differentiable=True, max_iterations=30, tolerance=float(tol))
    res = icp.icp(src, tgt, t(g["T_init"], torch.float32), trim_dist=float(trim), loss_fn={"name": "huber", "metric": float(huber)}, dim=2)
    assert res["T"].dtype == torch.float32
    np.testing.assert_allclose(npy(res["T"]), g["T"], rtol=0, atol=1e-4)
    res["T"].sum().backward()
    np.testing.assert_allclose(npy(src.grad), g["grad_source"], rtol=0, atol=1e-3)
    np.testing.assert_allclose(npy(tgt.grad), g["grad_target"], rtol=0, atol=1e-3)
    g32 = golden(name + "_f32")                       # and against the reference's own float32 run
    np.testing.assert_allclose(npy(res["T"]), g32["T"], rtol=0, atol=2e-5)


# --------------------------------------------------- reference input-handling tests
def test_ragged_list_batch(golden):
    """tests/test_ICP_inputs.py:36-110: batch == per-item loop == reference."""
    g = golden("input_types")
    S = [t(g["s0"]), t(g["s1"]), t(g["s2"])]
    Tg = [t(g["t0"]), t(g["t1"]), t(g["t2"])]
    T0 = torch.stack([torch.eye(4, dtype=torch.float64, device=DEV)] * 3)
    icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=25, tolerance=1e-8)
    kw = dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=2)
    batch = icp.icp(S, Tg, T0, **kw)
    check_result(batch, g, "batch_")
    for i in range(3):
        single = icp.icp(S[i], Tg[i], T0[i], **kw)
        check_result(single, g, "single%d_" % i)
        err = tran2vec(npy(single["T"])[0] @ np.linalg.inv(npy(batch["T"])[i]))
        assert np.linalg.norm(err) < 1e-8                                   # test_ICP_inputs.py:106-107
    g2 = golden("input_types_pt2pt")
    icp2 = ICP(icp_type="pt2pt", differentiable=True, max_iterations=25, tolerance=1e-8)
    check_result(icp2.icp(S, [x[:, :3] for x in Tg], list(T0), **kw), g2, "batch_")


def test_empty_clouds(golden, scan_map):
    """tests/test_ICP_inputs.py:113-155: missing data returns T_init."""
    scan, mp = scan_map
    g = golden("zero_inputs")
    S = [t(scan), [], []]
    Tg = [[], t(mp), []]
    T0 = torch.stack([torch.eye(4, dtype=torch.float64, device=DEV)] * 3)
    icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=25, tolerance=1e-8)
    batch = icp.icp(S, Tg, T0, trim_dist=5.0, loss_fn=None, dim=2)
    check_result(batch, g, "batch_")
    assert np.linalg.norm(npy(batch["T"]) - npy(T0)) < 1e-8
    for i in range(3):
        single = icp.icp(S[i], Tg[i], T0[i], trim_dist=5.0, loss_fn=None, dim=2)
        check_result(single, g, "single%d_" % i)


def test_weight_inputs(golden, scan_map):
    """tests/test_ICP_inputs.py:157-211 (+ the weight gradients no reference test checks)."""
    scan, mp = scan_map
    g = golden("weight_inputs")
    S = [t(scan[:, :3]), t(scan[:, :3]), t(np.vstack((scan[:, :3], g["junk"])))]
    Tg = [t(mp)] * 3
    W = [None, t(np.ones(65), grad=True), t(np.hstack((np.ones(65), np.zeros(10))), grad=True)]
    T0 = torch.stack([torch.eye(4, dtype=torch.float64, device=DEV)] * 3)
    icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=25, tolerance=1e-8)
    kw = dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=2)
    batch = icp.icp(S, Tg, T0, weight=W, **kw)
    check_result(batch, g, "batch_")
    batch["T"].sum().backward()
    np.testing.assert_allclose(npy(W[1].grad), g["grad_w1"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(npy(W[2].grad), g["grad_w2"], rtol=0, atol=1e-10)
    Ts = npy(batch["T"])
    assert np.linalg.norm(Ts[0] - Ts[1]) < 1e-8 and np.linalg.norm(Ts[0] - Ts[2]) < 1e-8   # test_ICP_inputs.py:210-211


def test_diff_vs_nondiff_and_padding(golden, scan_map):
    """tests/test_ICP_inputs.py:213-271."""
    scan, mp = scan_map
    g = golden("diff_vs_nondiff")
    s, tg = t(scan[:50, :3]), t(mp[:55])
    T0 = torch.eye(4, dtype=torch.float64, device=DEV)
    for lname, metric in (("huber", 1.0), ("cauchy", 0.5)):
        outs = {}
        for diff in (True, False):
            icp = ICP(icp_type="pt2pl", differentiable=diff, max_iterations=25, tolerance=1e-8)
            outs[diff] = icp.icp(s, tg, T0, trim_dist=5.0, loss_fn={"name": lname, "metric": metric}, dim=2)
            check_result(outs[diff], g, "%s_%s_" % (lname, "diff" if diff else "hard"))
        err = tran2vec(npy(outs[True]["T"])[0] @ np.linalg.inv(npy(outs[False]["T"])[0]))
        assert np.linalg.norm(err) < 1e-8
    gp = golden("padded_inputs")
    icp = ICP(icp_type="pt2pt", differentiable=False, max_iterations=25, tolerance=1e-8)
    icp.source_zeroes_are_pad = True
    plain = icp.icp(s, tg, T0, dim=2)
    padded = icp.icp(torch.cat((s, torch.zeros((20, 3), dtype=torch.float64, device=DEV))), tg, T0, dim=2)
    check_result(plain, gp, "plain_")
    check_result(padded, gp, "padded_")


def test_per_cloud_weight_with_zero_rows_as_pads(scan_map):
    """ICP.py:445-446: with source_zeroes_are_pad an (N,1) weight is multiplied into the (N,n) mask of non-zero rows and is a per-POINT weight from
    there on (ICP.py:248,269 then count points, not clouds) -- the call must report what the same call with that (N,n) weight reports."""
    scan, mp = scan_map
    pts = np.vstack((scan[:40, :3], np.zeros((9, 3))))
    S = torch.stack([t(pts), t(pts)])
    Tg = torch.stack([t(mp[:50]), t(mp[:50])])
    T0 = torch.stack([torch.eye(4, dtype=torch.float64, device=DEV)] * 2)
    w1 = torch.tensor([[1.0], [0.7]], dtype=torch.float64, device=DEV)
    outs = []
    for w in (w1, w1.expand(2, 49) * (S.norm(dim=2) != 0)):
        icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=12, tolerance=1e-12)
        icp.source_zeroes_are_pad = True
        outs.append(icp.icp(S, Tg, T0, weight=w.contiguous(), trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=2))
    for key in ("converged", "iterations", "matched_ratio"):
        assert torch.equal(outs[0]["stats"][key], outs[1]["stats"][key]), key
    assert torch.equal(outs[0]["T"], outs[1]["T"])
    assert bool((outs[0]["stats"]["matched_ratio"] <= 1.0).all())


# ------------------------------------------------ 3-D matrix with all four gradients
def matrix_keys(g):
    return sorted(k[:-len("__T")] for k in g if k.endswith("__T"))


def hard_branch_margins(g, key, metric):
    """Float32 can legitimately take the other side of a hard decision the float64 reference made: a where() branch of the
    hard Huber / trim weights (loss.py:32,56-58) or the argmin itself (nn.py:35).  For every cloud of a matrix scenario:
    the smallest distance of any point, at any iteration, from such a decision boundary (oracle run in float64 on the CPU:
    the checker, not the product)."""
    icp_type, mode, lname, trim, d = key.split("_")
    src = torch.tensor(g["source"])
    tgt = torch.tensor(g["target"] if icp_type == "pt2pl" else g["target"][:, :, :3])
    w = torch.tensor(g["weight"])
    rec = {}
    O.icp_batched(src, tgt, torch.tensor(g["T_init"]), w if icp_type == "pt2pl" else w.repeat_interleave(3, dim=1),
                  icp_type=icp_type, differentiable=(mode == "diff"), max_iterations=int(g["K"]), tolerance=1e-14,
                  trim_dist=(1.5 if trim == "trim" else None), loss_fn=None if lname == "none" else {"name": lname, "metric": metric},
                  dim=int(d[1]), const_iter=True, record=rec)
    if int(d[1]) == 2:                                      # the loop ran on the planar copies (ICP.py:107-116)
        src = src * torch.tensor([1.0, 1.0, 0.0], dtype=torch.float64)
        tgt = tgt * torch.tensor([1.0, 1.0, 0.0, 1.0, 1.0, 0.0], dtype=torch.float64)[:tgt.shape[2]]
    margin = torch.full((src.shape[0],), float("inf"), dtype=torch.float64)
    for idx, C, r in zip(rec["idx"], rec["C"], rec["r"]):
        ps = src @ C.transpose(1, 2) + r.reshape(-1, 1, 3)
        _, best, second = O.knn_exact_f64(ps, tgt)
        margin = torch.minimum(margin, (second - best).min(dim=1).values)
        nb = torch.gather(tgt, 1, idx.unsqueeze(-1).expand(-1, -1, tgt.shape[2]))
        e3 = ps - nb[:, :, :3]
        d3 = e3.norm(dim=2)
        en = (e3 * nb[:, :, 3:]).sum(-1).abs() if icp_type == "pt2pl" else d3
        if mode == "hard" and trim == "trim":
            margin = torch.minimum(margin, (d3 - 1.5).abs().min(dim=1).values)
        if mode == "hard" and lname in ("huber", "trim"):
            margin = torch.minimum(margin, (en - metric).abs().min(dim=1).values)
    return margin.numpy()


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("fixture", ["matrix3d", "matrix3d_trimloss"])
def test_matrix3d(golden, fixture, dtype):
    """matrix3d_trimloss: loss_fn={"name": "trim"} -- accepted exactly as ICP.py:157-160 / loss.py:15-16,43-58 do."""
    g = golden(fixture)
    K = int(g["K"])
    metric = float(g["loss_metric"]) if "loss_metric" in g else 0.3
    f64 = dtype == torch.float64
    held, skipped = 0, 0
    for key in matrix_keys(g):
        icp_type, mode, lname, trim, d = key.split("_")
        src = t(g["source"], dtype, grad=True)
        tgt = t(g["target"] if icp_type == "pt2pl" else g["target"][:, :, :3], dtype, grad=True)
        w = t(g["weight"], dtype, grad=True)
        T0 = t(g["T_init"], dtype, grad=True)
        icp = ICP(icp_type=icp_type, differentiable=(mode == "diff"), max_iterations=K, tolerance=1e-14)
        icp.const_iter = True
        res = icp.icp(src, tgt, T0, weight=w, trim_dist=(1.5 if trim == "trim" else None),
                      loss_fn=None if lname == "none" else {"name": lname, "metric": metric}, dim=int(d[1]))
        ((res["T"] * t(g["gT"], dtype)).sum() + (res["pc"] * t(g["gpc"], dtype)).sum()).backward()
        # float32: clouds whose float64 run passes within 1e-4 of a hard decision (a where() branch, the argmin) may
        # take the other branch; every other cloud is held to the north-star bars, hard-weight modes included
        clouds = np.arange(src.shape[0])
        if not f64:
            ok = hard_branch_margins(g, key, metric) > 1e-4
            held, skipped = held + int(ok.sum()), skipped + int((~ok).sum())
            clouds = clouds[ok]
        np.testing.assert_allclose(npy(res["T"])[clouds], g[key + "__T"][clouds], rtol=0, atol=1e-10 if f64 else 1e-4, err_msg=key)
        np.testing.assert_allclose(npy(res["deltas"])[clouds], g[key + "__deltas"][clouds], rtol=0, atol=1e-10 if f64 else 1e-4, err_msg=key)
        np.testing.assert_allclose(npy(res["weights"])[clouds, -1, :, 0], g[key + "__w_last"][clouds], rtol=0, atol=1e-10 if f64 else 2e-3, err_msg=key)
        np.testing.assert_allclose(npy(res["stats"]["iterations"]), g[key + "__stats_iterations"])
        for nm, leaf in (("source", src), ("target", tgt), ("weight", w), ("T_init", T0)):
            want = g[key + "__grad_" + nm]
            if f64:
                np.testing.assert_allclose(npy(leaf.grad), want, rtol=1e-8, atol=1e-9, err_msg=key + " " + nm)
            else:
                scale = max(1.0, float(np.abs(want).max()))      # north-star bar (1e-3) relative to the gradient's scale
                np.testing.assert_allclose(npy(leaf.grad)[clouds], want[clouds], rtol=0, atol=1e-3 * scale, err_msg=key + " " + nm)
    if not f64:
        assert held >= 5 * skipped, "too few float32 clouds clear of a hard decision boundary: %d held, %d skipped" % (held, skipped)


def test_loss_fn_trim_is_accepted(golden, scan_map):
    """The reference takes loss_fn={"name": "trim"} (ICP.py:157-160); an unknown name raises ValueError (loss.py:19)."""
    scan, mp = scan_map
    s, tg = t(scan[:, :3]), t(mp)
    T0 = torch.eye(4, dtype=torch.float64, device=DEV)
    icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=5, tolerance=1e-12)
    out = icp.icp(s, tg, T0, trim_dist=5.0, loss_fn={"name": "trim", "metric": 2.0}, dim=2)
    ref = O.icp_batched(s.cpu()[None], tg.cpu()[None], T0.cpu()[None], torch.ones(1, s.shape[0], dtype=torch.float64), icp_type="pt2pl",
                        differentiable=True, max_iterations=5, tolerance=1e-12, trim_dist=5.0, loss_fn={"name": "trim", "metric": 2.0}, dim=2)
    np.testing.assert_allclose(npy(out["T"]), ref["T"].numpy(), rtol=0, atol=1e-10)
    with pytest.raises(ValueError):
        icp.icp(s, tg, T0, loss_fn={"name": "tukey", "metric": 1.0})


def test_weight_tensor_dtype_and_shape():
    """a weight TENSOR of another dtype is cast (the reference promotes), one of the wrong shape raises
    (the reference's broadcast fails) -- it is never read as raw memory of the cloud dtype."""
    src, tgt = make_pairs(2, 300, 320, seed=5, dtype=torch.float64)
    src, tgt = src.to(DEV), tgt.to(DEV)
    T0 = torch.eye(4, dtype=torch.float64, device=DEV).repeat(2, 1, 1)
    w64 = torch.rand(2, 300, dtype=torch.float64, device=DEV) + 0.1
    icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=3, tolerance=1e-12)
    icp.const_iter = True
    a = icp.icp(src, tgt, T0, weight=w64.float().double(), trim_dist=5.0)
    w32 = w64.float().requires_grad_(True)
    b = icp.icp(src, tgt, T0, weight=w32, trim_dist=5.0)
    np.testing.assert_allclose(npy(b["T"]), npy(a["T"]), rtol=0, atol=1e-12)
    b["T"].sum().backward()
    assert w32.grad is not None and w32.grad.dtype == torch.float32 and bool(torch.isfinite(w32.grad).all())
    with pytest.raises(RuntimeError):
        icp.icp(src, tgt, T0, weight=w64[:, :200], trim_dist=5.0)
    with pytest.raises(AssertionError):                     # the reference's own check (ICP.py:326) catches the 2-D form first
        icp.icp(src[0], tgt[0], T0[0], weight=w64[0, :17], trim_dist=5.0)


# ---------------------------------------------------------------- nn / loss classes
def test_nn_class(golden):
    g = golden("nn_vectors")
    y = t(g["y"], grad=True)
    x = t(g["x"], grad=True)
    hard = nn(differentiable=False)
    nb = hard.find_nn(x, y)
    np.testing.assert_array_equal(npy(nb), g["nb"])
    (nb * t(g["cot"])).sum().backward()
    np.testing.assert_allclose(npy(y.grad), g["grad_y"], rtol=0, atol=1e-12)
    assert x.grad is None                                                  # argmin has no gradient
    np.testing.assert_array_equal(npy(hard.find_nn(t(g["x"]).transpose(1, 2), t(g["y"]).transpose(1, 2))), g["nb_T"])
    np.testing.assert_array_equal(npy(hard.find_nn(t(g["x"][0]), t(g["y"][0]))), g["nb_2d"])
    np.testing.assert_array_equal(npy(nn(differentiable=True, use_gumbel=False).find_nn(t(g["x"]), t(g["y"]))), g["nb"])
    # CPU tensors in, CPU tensors out (computed on the HIP device)
    out_cpu = hard.find_nn(torch.tensor(g["x"]), torch.tensor(g["y"]))
    assert out_cpu.device.type == "cpu"
    np.testing.assert_array_equal(out_cpu.numpy(), g["nb"])
    # Gumbel path with the reference's uniform draw injected
    xs, ys = t(g["x"], torch.float32, grad=True), t(g["y"], torch.float32, grad=True)
    soft = nn(differentiable=True, use_gumbel=True, eps=1e-10, tau=0.1)
    nb_soft = soft.find_nn(xs, ys, U=t(g["U"]))
    np.testing.assert_allclose(npy(nb_soft), g["nb_soft"], rtol=0, atol=2e-5)
    (nb_soft * t(g["cot"], torch.float32)).sum().backward()
    np.testing.assert_allclose(npy(xs.grad), g["grad_x_soft"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(npy(ys.grad), g["grad_y_soft"], rtol=1e-3, atol=1e-4)


def test_nn_known_answer(golden):
    """/root/reference/tests/test_nn.py:12-41 on the HIP device (default ctor = Gumbel path)."""
    g = golden("nn_vectors")
    torch.manual_seed(0)
    d = nn(differentiable=True)
    pts = t(g["kat_points"], grad=True)
    q = t(g["kat_query"], grad=True)
    near = d.find_nn(q, pts)
    assert torch.allclose(near[0, 0], t(g["kat_expect1"]), atol=1e-3)
    near.sum().backward()
    assert q.grad is not None and pts.grad is not None
    assert not torch.isnan(q.grad).any() and not torch.isnan(pts.grad).any()
    h = nn(differentiable=False)
    assert torch.equal(h.find_nn(q, pts)[0, 0], t(g["kat_expect1"]))
    pts2 = torch.cat((pts.detach(), t(g["kat_extra"]).view(1, -1)))
    assert torch.equal(h.find_nn(q, pts2)[0, 0], t(g["kat_expect2"]))


def test_loss_class(golden):
    g = golden("loss_vectors")
    for name, metric in (("huber", 1.0), ("cauchy", 0.5), ("trim", 2.0)):
        for diff in (True, False):
            for tag in ("e1", "e3", "eb"):
                key = "%s_%s_%s" % (name, "diff" if diff else "hard", tag)
                e = t(g[tag], grad=True)
                w = loss(name=name, metric=metric, differentiable=diff, tanh_steepness=5.0).get_weight(e)
                assert tuple(w.shape) == g[key].shape, key
                np.testing.assert_allclose(npy(w), g[key], rtol=0, atol=1e-13, err_msg=key)
                if key + "_grad" in g:
                    w.sum().backward()
                    np.testing.assert_allclose(npy(e.grad), g[key + "_grad"], rtol=0, atol=1e-12, err_msg=key)
    with pytest.raises(ValueError):
        loss("tukey").get_weight(t(g["e1"]))


# -------------------------------------------------------- oracle parity at larger sizes
@pytest.mark.parametrize("icp_type,variant", [("pt2pl", _lib.KNN_VALU), ("pt2pt", _lib.KNN_VALU), ("pt2pl", _lib.KNN_MFMA)])
def test_synthetic_float32_vs_oracle(icp_type, variant):
    """Benchmark-shaped input at a size the oracle finishes in seconds (B=4, 2048 pts)."""
    N, n, m, K = 4, 2048, 2048, 5
    src, tgt = make_pairs(N, n, m, seed=2, dtype=torch.float32)
    tg = tgt if icp_type == "pt2pl" else tgt[:, :, :3].contiguous()
    kw = dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=3)
    sc, tc = src.clone().requires_grad_(True), tg.clone().requires_grad_(True)
    rows = 3 if icp_type == "pt2pt" else 1
    ref = O.icp_batched(sc, tc, torch.eye(4).repeat(N, 1, 1), torch.ones(N, n * rows), icp_type=icp_type,
                        differentiable=True, max_iterations=K, tolerance=1e-12, const_iter=True, **kw)
    ref["T"].sum().backward()
    sd, td = src.to(DEV).requires_grad_(True), tg.to(DEV).requires_grad_(True)
    icp = ICP(icp_type=icp_type, differentiable=True, max_iterations=K, tolerance=1e-12)
    icp.const_iter = True
    icp.knn_variant = variant
    out = icp.icp(sd, td, torch.eye(4, device=DEV).repeat(N, 1, 1), **kw)
    out["T"].sum().backward()
    assert out["deltas"].shape == (N, K, 6, 1) and out["weights"].shape == (N, K, n * rows, 1)
    np.testing.assert_allclose(npy(out["T"]), npy(ref["T"]), rtol=0, atol=1e-4)
    np.testing.assert_allclose(npy(out["pc"]), npy(ref["pc"]), rtol=0, atol=2e-4)
    np.testing.assert_allclose(npy(out["costs"]), npy(ref["costs"]), rtol=2e-3, atol=1e-4)
    np.testing.assert_allclose(npy(sd.grad), npy(sc.grad), rtol=0, atol=1e-3)
    np.testing.assert_allclose(npy(td.grad), npy(tc.grad), rtol=0, atol=1e-3)
    assert float(npy(out["stats"]["iterations"]).min()) == K


def test_cpu_tensors_round_trip(golden):
    """The reference's tests pass CPU tensors: they are computed on the GPU and come back on the CPU."""
    g = golden("c1_pt2pl_diff")
    src = torch.tensor(g["source"], requires_grad=True)
    tgt = torch.tensor(g["target"], requires_grad=True)
    icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=100, tolerance=1e-10)
    res = icp.icp(src, tgt, torch.tensor(g["T_init"]), trim_dist=5.0, loss_fn={"name": "huber", "metric": 10.0}, dim=2)
    assert res["T"].device.type == "cpu" and res["stats"]["converged"].device.type == "cpu"
    np.testing.assert_allclose(res["T"].detach().numpy(), g["T"], rtol=0, atol=1e-10)
    res["T"].sum().backward()
    np.testing.assert_allclose(src.grad.numpy(), g["grad_source"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(tgt.grad.numpy(), g["grad_target"], rtol=0, atol=1e-10)


# ------------------------------------------------ BASELINE.json full sizes: size-independent properties
def sampled_exact(x, y, rows):
    """Exact f64 nearest neighbour of the sampled query rows against ALL targets: (idx, best, second)."""
    xs = x[rows].double()
    d = ((xs[:, None, :] - y[None, :, :3].double()) ** 2).sum(-1)
    v, i = torch.topk(d, 2, dim=1, largest=False)
    return torch.argmin(d, dim=1), v[:, 0], v[:, 1]


@pytest.mark.parametrize("n,variant", [(16384, _lib.KNN_VALU), (16384, _lib.KNN_MFMA), (65536, _lib.KNN_MFMA), (65536, _lib.KNN_VALU)])
def test_knn_full_size_properties(n, variant):
    """configs[2]/[3] cloud sizes (16384 and 65536 points): sampled exactness, permutation equivariance,
    self-match, pad rows never selected."""
    N = 16 if n == 16384 else 4
    src, tgt = make_pairs(N, n, n, seed=7, dtype=torch.float32)
    sd, td = src.to(DEV), tgt.to(DEV)
    tgt4 = _ops.pack_target(td)
    idx = _ops.knn(sd, None, tgt4, n, variant)
    assert int(idx.min()) >= 0 and int(idx.max()) < n
    g = torch.Generator().manual_seed(1)
    rows = torch.randint(0, n, (256,), generator=g).to(DEV)
    for b in (0, N - 1):
        want, best, second = sampled_exact(sd[b], td[b], rows)
        got = idx[b][rows].long()
        bad = got != want
        if bad.any():      # only allowed where the runner-up is within float32 resolution of the expanded form
            assert bool(((second - best)[bad] < 1e-4).all())
        assert int(bad.sum()) <= 2
    # permuting the queries permutes the answer; permuting the targets relabels it
    perm = torch.randperm(n, generator=g).to(DEV)
    idx_p = _ops.knn(sd[:, perm].contiguous(), None, tgt4, n, variant)
    assert torch.equal(idx_p, idx[:, perm])
    tperm = torch.randperm(n, generator=g).to(DEV)
    idx_t = _ops.knn(sd[:2], None, _ops.pack_target(td[:2, tperm].contiguous()), n, variant)
    same = tperm[idx_t.long()] == idx[:2].long()
    assert float(same.float().mean()) > 0.9999          # ties/near-ties may relabel
    # a cloud queried against itself returns the identity (distance exactly 0 beats everything)
    self_idx = _ops.knn(td[:2, :, :3].contiguous(), None, _ops.pack_target(td[:2]), n, variant)
    assert torch.equal(self_idx.cpu(), torch.arange(n, dtype=torch.int32).repeat(2, 1))


def test_full_size_icp_properties():
    """configs[2] shape (16384-pt clouds, pt2pl + huber + trim, fwd+bwd) on 8 clouds: recovers the planted
    pose, batch == per-item, already-aligned input stays put, gradients finite and batch-independent."""
    N, n = 8, 16384
    src, tgt = make_pairs(N, n, n, seed=9, dtype=torch.float32)
    sd, td = src.to(DEV).requires_grad_(True), tgt.to(DEV).requires_grad_(True)
    T0 = torch.eye(4, device=DEV).repeat(N, 1, 1)
    icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=12, tolerance=1e-12)
    icp.const_iter = True
    kw = dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=3)
    out = icp.icp(sd, td, T0, **kw)
    out["T"].sum().backward()
    # planted motion: make_pairs builds source = C^T (s_t - r): ICP must return pc ~ the noisy target picks
    nbr = nn(differentiable=False).find_nn(out["pc"].detach(), td.detach())[:, :, :3]
    assert float((out["pc"].detach() - nbr).norm(dim=2).mean()) < 0.03          # noise sigma 0.01 per axis
    assert float(out["deltas"][:, -1].abs().max()) < 1e-3                        # converged
    assert bool(torch.isfinite(sd.grad).all() and torch.isfinite(td.grad).all())
    # batch == single item (independent clouds)
    s1, t1 = src[3:4].to(DEV).requires_grad_(True), tgt[3:4].to(DEV).requires_grad_(True)
    one = icp.icp(s1, t1, T0[:1], **kw)
    one["T"].sum().backward()
    np.testing.assert_allclose(npy(one["T"])[0], npy(out["T"])[3], rtol=0, atol=1e-6)
    np.testing.assert_allclose(npy(s1.grad)[0], npy(sd.grad)[3], rtol=0, atol=1e-5)
    # idempotence: restarting from the solution moves by (almost) nothing
    again = icp.icp(sd.detach(), td.detach(), out["T"].detach(), **kw)
    assert float((again["T"] - out["T"].detach()).abs().max()) < 1e-4


def test_config2_point_to_point_batch32():
    """configs[1]: B=32 synthetic 4096-pt clouds, point-to-point; oracle parity on a 4-cloud slice."""
    N, n = 32, 4096
    src, tgt = make_pairs(N, n, n, seed=1, dtype=torch.float32)
    tg = tgt[:, :, :3].contiguous()
    icp = ICP(icp_type="pt2pt", differentiable=True, max_iterations=5, tolerance=1e-12)
    icp.const_iter = True
    out = icp.icp(src.to(DEV), tg.to(DEV), torch.eye(4, device=DEV).repeat(N, 1, 1), trim_dist=5.0)
    sl = slice(10, 14)
    ref = O.icp_batched(src[sl], tg[sl], torch.eye(4).repeat(4, 1, 1), torch.ones(4, 3 * n), icp_type="pt2pt",
                        differentiable=True, max_iterations=5, tolerance=1e-12, trim_dist=5.0, const_iter=True)
    np.testing.assert_allclose(npy(out["T"])[sl], npy(ref["T"]), rtol=0, atol=1e-4)
    np.testing.assert_allclose(npy(out["costs"])[sl], npy(ref["costs"]), rtol=2e-3, atol=1e-3)
    assert out["weights"].shape == (N, 5, 3 * n, 1)


def test_tolerance_stop_equals_const_iter(golden):
    """converged clouds are frozen, so running extra constant iterations changes nothing."""
    g = golden("c1_pt2pl_diff")
    src, tgt = t(g["source"]), t(g["target"])
    kw = dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 10.0}, dim=2)
    a = ICP(icp_type="pt2pl", max_iterations=100, tolerance=1e-10).icp(src, tgt, t(g["T_init"]), **kw)
    b_icp = ICP(icp_type="pt2pl", max_iterations=30, tolerance=1e-10)
    b_icp.const_iter = True
    b = b_icp.icp(src, tgt, t(g["T_init"]), **kw)
    assert a["deltas"].shape[1] == 6 and b["deltas"].shape[1] == 30
    np.testing.assert_allclose(npy(a["T"]), npy(b["T"]), rtol=0, atol=1e-12)
    assert float(npy(a["stats"]["iterations"])[0]) == 6.0 and float(npy(b["stats"]["iterations"])[0]) == 30.0


# ----------------------------------------------------------- exact sorted-sweep kNN (dicp_knn_sweep)
def sweep_knn(x, y, pose=None, sort_queries=True, cfg=0):
    sw = _ops.SweepIndex(y)
    return sw.knn(x, pose, sw.query_order(x, pose) if sort_queries else None, cfg=cfg), sw


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("N,n,m", [(1, 1, 1), (3, 70, 90), (2, 257, 16), (5, 1000, 2049), (2, 2100, 4100), (9, 130, 64)])
def test_sweep_knn_equals_brute_force(dtype, N, n, m):
    g = torch.Generator().manual_seed(N * 1000 + n + 1)
    x = (torch.rand((N, n, 3), generator=g, dtype=torch.float64) * 10 - 5).to(dtype).to(DEV)
    y = (torch.rand((N, m, 6), generator=g, dtype=torch.float64) * 10 - 5).to(dtype).to(DEV)
    brute = _ops.knn(x, None, _ops.pack_target(y), m, _lib.KNN_VALU)
    for cfg in (0, 1, 2, 4):                                      # every launch configuration of the tile sweep
        for sort_q in (True, False):
            got, sw = sweep_knn(x, y, sort_queries=sort_q, cfg=cfg)
            assert torch.equal(got, brute), (cfg, sort_q)        # same scores, same tie rule: bit-identical indices
    if dtype == torch.float64:
        assert torch.equal(brute.cpu().long(), O.nn_index(x.cpu(), y.cpu()))


QO_SIDE, QO_MID = 256, 2048 - 2 * 256        # kernels_setup.h: ordering buckets for the queries outside the targets' x range (either side), and inside it


def rank_buckets_ascend(sw, keys, qo):
    """The query order is a counting sort by bucket: inside the targets' x range the RANK bucket (rank = lower bound of the query's x among the
    sorted target keys, bucket = QO_SIDE + rank * (QO_MID - 1) // m), outside it QO_SIDE equal-width x buckets per side (one per 1/QO_SIDE of the
    targets' span, clamped a span away: round 6 -- the queries a partly overlapping scan has outside the other's footprint stay neighbours in x).
    Along the order the buckets must never decrease."""
    m = sw.m
    xs = sw.tgs4[:, :m, 0].contiguous()
    ks = torch.gather(keys, 1, qo.long())
    lo, hi = xs[:, :1], xs[:, m - 1:m]
    ks = torch.nan_to_num(ks, nan=0.0, posinf=3e38, neginf=-3e38)
    ks = torch.where(torch.isnan(torch.gather(keys, 1, qo.long())), lo.expand_as(ks), ks)      # NaN keys are parked in the first middle bucket
    rank = torch.searchsorted(xs, ks.contiguous())
    side = QO_SIDE / (hi - lo).clamp_min(1e-30)
    b = QO_SIDE + (rank * (QO_MID - 1)) // max(m, 1)
    b = torch.where(ks < lo, QO_SIDE - 1 - ((lo - ks) * side).clamp(0, QO_SIDE - 1).long(), b)
    b = torch.where(ks > hi, QO_SIDE + QO_MID + ((ks - hi) * side).clamp(0, QO_SIDE - 1).long(), b)
    # the kernel finds the rank through a float bucket table: allow one bucket of slack for its rounding at the edges
    return bool(((torch.cummax(b, dim=1).values - b) <= 1).all())


def test_query_order_beyond_the_lds_path():
    """More than 16384 queries per cloud take dicp_query_order's two-pass form: still a permutation in bucket order,
    and the sweep built on it still equals brute force."""
    g = torch.Generator().manual_seed(5)
    N, n, m = 2, 20011, 3000
    x = (torch.rand((N, n, 3), generator=g) * 8 - 4).to(DEV)
    y = (torch.rand((N, m, 3), generator=g) * 8 - 4).to(DEV)
    sw = _ops.SweepIndex(y)
    qo = sw.query_order(x, None)
    assert torch.equal(torch.sort(qo.long(), dim=1).values, torch.arange(n, device=DEV).repeat(N, 1))
    # (clouds of more than 16384 queries are ordered by equal-WIDTH x buckets of the targets' range, not by rank)
    lo, hi = y[:, :, 0].min(dim=1).values[:, None], y[:, :, 0].max(dim=1).values[:, None]
    ks = torch.gather(x[:, :, 0], 1, qo.long()).clamp(lo, hi)          # (queries outside the targets' x range: side buckets, by x as well)
    width = float(((hi - lo) / QO_MID).max())
    assert float((torch.cummax(ks, dim=1).values - ks).max()) <= 1.05 * width + 1e-4
    assert torch.equal(sw.knn(x, None, qo), _ops.knn(x, None, _ops.pack_target(y), m, _lib.KNN_VALU))


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_query_order_is_a_permutation_in_bucket_order(dtype):
    """dicp_query_order (counting sort by the rank bucket of x among the sorted targets): a permutation of the queries in
    ascending bucket, for any pose, with queries outside the target's x range and non-finite ones parked at the ends."""
    g = torch.Generator().manual_seed(11)
    N, n, m = 5, 3001, 777
    x = ((torch.rand((N, n, 3), generator=g, dtype=torch.float64) * 14 - 7)).to(dtype)
    x[0, 5] = float("nan")
    x[1, 9, 0] = float("inf")
    y = ((torch.rand((N, m, 3), generator=g, dtype=torch.float64) * 10 - 5)).to(dtype)
    xd, yd = x.to(DEV), y.to(DEV)
    sw = _ops.SweepIndex(yd)
    ang = 0.3
    pose = torch.tensor([np.cos(ang), -np.sin(ang), 0, np.sin(ang), np.cos(ang), 0, 0, 0, 1, 0.4, -0.1, 0.2], dtype=dtype).repeat(N, 1).to(DEV)
    for ps in (None, pose):
        qo = sw.query_order(xd, ps)
        assert qo.dtype == torch.int32 and qo.shape == (N, n)
        assert torch.equal(torch.sort(qo.long(), dim=1).values, torch.arange(n, device=DEV).repeat(N, 1))
        key = xd[:, :, 0] if ps is None else (xd * ps[:, None, 0:3]).sum(dim=2) + ps[:, None, 9]
        assert rank_buckets_ascend(sw, key.contiguous(), qo)
        exact = sw.query_order(xd, ps, exact=True)
        assert torch.equal(torch.sort(exact.long(), dim=1).values, torch.arange(n, device=DEV).repeat(N, 1))
        # reproducible (buckets hold their members in index order) and the slot-ordered copies are what they say
        wq = torch.rand((N, n), generator=g, dtype=torch.float64).to(dtype).to(DEV)
        qo, x_s, w_s = sw.query_order(xd, ps, w=wq, copies=True)
        gi = qo.long().unsqueeze(-1).expand(-1, -1, 3)
        assert torch.equal(torch.nan_to_num(x_s, nan=7.0), torch.nan_to_num(torch.gather(xd, 1, gi), nan=7.0))
        assert torch.equal(w_s, torch.gather(wq, 1, qo.long()))
    # reproducible when no bucket is crowded (buckets of up to 64 members are put in index order): queries inside the
    # target's x range, two calls -> the same permutation
    xin = (xd * 0.6).contiguous()
    xin[0, 5] = 0.0
    xin[1, 9, 0] = 0.0
    assert torch.equal(sw.query_order(xin, None, reproducible=True), sw.query_order(xin, None, reproducible=True))


def test_query_reorder_keeps_the_order_of_clouds_that_hardly_moved():
    """dicp_query_reorder: given the order made under an earlier pose of the call, a cloud whose points have moved by less than a tenth of a unit's x
    extent keeps it (a copy); a cloud that moved further is sorted again (a permutation in bucket order under the NEW pose)."""
    g = torch.Generator().manual_seed(17)
    N, n, m = 4, 8192, 8192
    x = (torch.rand((N, n, 3), generator=g) * 20 - 10).to(DEV)
    y = (torch.rand((N, m, 3), generator=g) * 20 - 10).to(DEV)
    sw = _ops.SweepIndex(y)
    eye = torch.tensor([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0], dtype=torch.float32).repeat(N, 1).to(DEV)
    first = sw.query_order(x, eye)
    moved = eye.clone()
    moved[0, 9] += 1e-4                                      # 0.1 mm: far below a tenth of 20 m * 128 / 8192 = 0.31 m
    moved[1, 9] += 0.5                                       # half a metre
    moved[2, 0], moved[2, 1], moved[2, 3], moved[2, 4] = float(np.cos(0.2)), float(-np.sin(0.2)), float(np.sin(0.2)), float(np.cos(0.2))      # 0.2 rad about z
    again = sw.query_order(x, moved, pose_prev=eye, order_prev=first)
    assert again.data_ptr() != first.data_ptr()
    for b in range(N):
        assert torch.equal(torch.sort(again[b].long()).values, torch.arange(n, device=DEV)), b
    assert torch.equal(again[0], first[0]) and torch.equal(again[3], first[3])            # kept
    assert not torch.equal(again[1], first[1]) and not torch.equal(again[2], first[2])    # sorted again ...
    key = (x * moved[:, None, 0:3]).sum(dim=2) + moved[:, None, 9]
    assert rank_buckets_ascend(sw, key.contiguous(), again)                               # ... under the new pose (the kept ones moved by less than a bucket)
    got = sw.knn(x, moved, again)
    assert torch.equal(got, _ops.knn(x, moved, _ops.pack_target(y), m, _lib.KNN_VALU))


def test_sweep_knn_ties_duplicates_and_pads():
    """Duplicated targets (exact score ties across chunks and tiles) resolve to the lowest ORIGINAL index,
    like torch.argmin; pad rows never win; a fused pose is honoured."""
    g = torch.Generator().manual_seed(3)
    base = torch.rand((2, 50, 3), generator=g, dtype=torch.float32) * 6
    y = base.repeat(1, 7, 1)[:, torch.randperm(350, generator=g)]            # every point 7 times, shuffled
    y = torch.cat((y, torch.full((2, 30, 3), 6000.0)), dim=1).to(DEV)        # + pad rows (ICP.py:460)
    x = (base[:, :40] + 0.01 * torch.rand((2, 40, 3), generator=g)).to(DEV)
    brute = _ops.knn(x, None, _ops.pack_target(y), 380, _lib.KNN_VALU)
    # among exact duplicates the kernels return the lowest index; the reference's pick depends on how its
    # BLAS rounds each column of the cdist matmul, so only "same point, nothing nearer" is comparable
    ref = O.nn_index(x.cpu().double(), y.cpu().double())
    pick = lambda ix: torch.gather(y.cpu(), 1, ix.cpu().long().unsqueeze(-1).expand(-1, -1, 3))
    assert torch.equal(pick(brute), pick(ref))
    first = torch.stack([torch.stack([(y[b] == y[b, brute[b, i]]).all(dim=1).nonzero()[0, 0] for i in range(40)]) for b in range(2)])
    assert torch.equal(brute.cpu().long(), first.cpu())
    for cfg in (1, 2, 4):
        got, _ = sweep_knn(x, y, cfg=cfg)
        assert torch.equal(got, brute), cfg
    ang = 0.4
    C = torch.tensor([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1.0]], dtype=torch.float32)
    r = torch.tensor([0.5, -0.2, 0.1])
    pose = torch.cat((C.reshape(9), r)).repeat(2, 1).to(DEV)
    for cfg in (0, 1):
        got, _ = sweep_knn(x, y, pose=pose, cfg=cfg)
        assert torch.equal(got, _ops.knn(x, pose, _ops.pack_target(y), 380, _lib.KNN_VALU))
    # degenerate: every target on one x plane (no pruning possible) and identical points
    flat = y.clone()
    flat[:, :, 0] = 1.0
    for cfg in (0, 1, 2):
        got, _ = sweep_knn(x, flat, cfg=cfg)
        assert torch.equal(got, _ops.knn(x, None, _ops.pack_target(flat), 380, _lib.KNN_VALU))
    # non-finite queries: every form answers 0 like the brute-force kernel (no neighbour)
    xn = x.clone()
    xn[0, 3] = float("nan")
    xn[1, 7, 1] = float("inf")
    bn = _ops.knn(xn, None, _ops.pack_target(y), 380, _lib.KNN_VALU)
    for cfg in (0, 2):
        got, _ = sweep_knn(xn, y, cfg=cfg)
        assert torch.equal(got, bn), cfg


def test_sweep_knn_full_size_and_pruning():
    """configs[2] cloud size: identical to brute force on every query, while scoring a small fraction of pairs."""
    N, n = 16, 16384
    src, tgt = make_pairs(N, n, n, seed=7, dtype=torch.float32)
    sd, td = src.to(DEV), tgt.to(DEV)
    brute = _ops.knn(sd, None, _ops.pack_target(td), n, _lib.KNN_VALU)
    got, sw = sweep_knn(sd, td)
    assert torch.equal(got, brute)
    frac = float(sw.pairs.item()) / (float(N) * n * n)
    assert frac < 0.25, frac
    got2, _ = sweep_knn(sd, td, sort_queries=False)                         # unsorted queries: still exact
    assert torch.equal(got2, brute)
    # near pose: a few dozen rows per query, and m = 4n
    near = td[:, :, :3] + 0.01 * torch.randn((N, n, 3), generator=torch.Generator().manual_seed(1)).to(DEV)
    bn = _ops.knn(near, None, _ops.pack_target(td), n, _lib.KNN_VALU)
    got4, sw4 = sweep_knn(near, td)
    assert torch.equal(got4, bn)
    assert float(sw4.pairs.item()) / (float(N) * n * n) < 0.03
    q4 = near[:, ::4].contiguous()
    assert torch.equal(sweep_knn(q4, td)[0], _ops.knn(q4, None, _ops.pack_target(td), n, _lib.KNN_VALU))


@pytest.mark.parametrize("offset,noise", [(0.0, 1e-2), (0.0, 1e-4), (60.0, 1e-2), (300.0, 1e-2), (300.0, 0.0), (2000.0, 1e-3)])
def test_sweep_prune_margin_near_pose_and_far_from_origin(offset, noise):
    """The sweep's prune margin (SweepEps) is sized from the rounding error of a score, which grows with 0.5|x|^2: clouds far
    from the origin, queries a hair away from (or exactly on) their neighbours, must still return the brute-force indices."""
    N, n = 8, 16384
    g = torch.Generator().manual_seed(int(offset) + 5)
    pts = (torch.rand((N, n, 3), generator=g) - 0.5) * 20 + torch.tensor([offset, -0.5 * offset, 0.25 * offset])
    td = torch.cat((pts, torch.nn.functional.normalize(torch.randn((N, n, 3), generator=g), dim=2)), dim=2).float().to(DEV)
    qd = (pts[:, torch.randperm(n, generator=g)] + noise * torch.randn((N, n, 3), generator=g)).float().to(DEV).contiguous()
    brute = _ops.knn(qd, None, _ops.pack_target(td), n, _lib.KNN_VALU)
    for cfg in (0, 1, 2, 4):
        got, _ = sweep_knn(qd, td, cfg=cfg)
        assert torch.equal(got, brute), (offset, noise, cfg, int((got != brute).sum()))


@pytest.mark.parametrize("window", [False, True])
@pytest.mark.parametrize("name,icp_type,diff", [("c1_pt2pt_diff", "pt2pt", True), ("c1_pt2pl_diff", "pt2pl", True)])
def test_icp_with_sweep_knn_matches_reference(golden, name, icp_type, diff, window):
    g = golden(name)
    trim, huber, tol, max_iter = g["params"]
    src, tgt = t(g["source"], grad=True), t(g["target"], grad=True)
    icp = ICP(icp_type=icp_type, differentiable=diff, max_iterations=int(max_iter), tolerance=float(tol))
    icp.knn_variant = _lib.KNN_SWEEP
    icp.bwd_window = window
    res = icp.icp(src, tgt, t(g["T_init"]), trim_dist=float(trim), loss_fn={"name": "huber", "metric": float(huber)}, dim=2)
    check_result(res, g)
    res["T"].sum().backward()
    np.testing.assert_allclose(npy(src.grad), g["grad_source"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(npy(tgt.grad), g["grad_target"], rtol=0, atol=1e-10)
    assert int(icp.knn_stats["knn_pairs"].sum().item()) > 0


def test_search_frame_centre_is_the_quantised_median_and_zero_keeps_the_bits():
    """dicp_search_frame, the centre part (directions off: Q = I, t = -c): coordinate-wise (lower) median of a stride sample of the cloud's
    rows, rounded to multiples of the quantum (0 near the origin); a ragged batch hands over the clouds' own lengths, so the reference's far
    pad rows (ICP.py:460) are not in the sample; with a zero frame every producer writes exactly what it writes without one."""
    g = torch.Generator().manual_seed(3)
    tgt = torch.rand((5, 777, 6), generator=g) * 20 - 10
    tgt[1, :, :3] += torch.tensor([1003.0, -37.0, 7.9])
    tgt[2, :, :3] += torch.tensor([-8.1, 24.3, 100000.0])
    tgt[3, 500:, :] = float("nan")                           # rows past the clouds' own lengths (never read) ...
    tgt[4, 200:, :] = float("nan")
    tr = torch.tensor([777, 777, 777, 500, 200], dtype=torch.int32, device=DEV)
    td = tgt.to(DEV)
    med = tgt[:, :, :3].float().sort(dim=1).values[:, (777 - 1) // 2]            # lower median per coordinate (all 777 rows sampled)
    med[3] = tgt[3, :500, :3].float().sort(dim=0).values[(500 - 1) // 2]
    med[4] = tgt[4, :200, :3].float().sort(dim=0).values[(200 - 1) // 2]
    eye = torch.eye(3).reshape(9)

    def centre(t, **kw):
        F = _ops.search_frame(t, directions=False, **kw).cpu()
        assert torch.equal(F[:, :9].float(), eye.repeat(F.shape[0], 1))
        return -F[:, 9:]
    exact = centre(td, quantum=0.0, tgt_rows=tr)
    assert torch.equal(exact, med), (exact, med)
    c = centre(td, quantum=16.0, tgt_rows=tr)
    assert torch.equal(c.double(), torch.round(med.double() / 16.0) * 16.0), (c, med)
    assert torch.equal(c[0], torch.zeros(3)) and torch.equal(c[3], torch.zeros(3)) and torch.equal(c[4], torch.zeros(3))
    td = torch.nan_to_num(td, nan=10000.0)                   # (the index checks below run dense)
    same = torch.full((1, 50, 3), 7.25)
    assert torch.equal(centre(same.to(DEV), quantum=0.0), same[:, 0])                   # all rows one point: they all vote
    big = (torch.rand((2, 10000, 3), generator=g) * 20 - 10 + torch.tensor([500.0, 0.0, -300.0]))
    step = (10000 + 1023) // 1024
    sample = big[:, ::step]
    assert torch.equal(centre(big.to(DEV), quantum=0.0), sample.sort(dim=1).values[:, (sample.shape[1] - 1) // 2])
    assert torch.equal(centre(big.double().to(DEV), quantum=0.0).float(), sample.sort(dim=1).values[:, (sample.shape[1] - 1) // 2])
    zero = torch.cat((eye, torch.zeros(3))).repeat(5, 1).to(DEV)
    assert torch.equal(_ops.pack_target(td), _ops.pack_target(td, zero))
    a, b = _ops.SweepIndex(td, sorted_rows=True), _ops.SweepIndex(td, sorted_rows=True, frame=zero)
    for name in ("tgs4", "tperm", "bucket", "brange", "keys", "tgt_s"):
        x, y = getattr(a, name), getattr(b, name)
        if name == "keys":                                                            # pad keys are NaN
            x, y = torch.nan_to_num(x, nan=7.0), torch.nan_to_num(y, nan=7.0)
        assert torch.equal(x, y), name
    # a real centre: rows and keys are the shifted ones, the full rows for the backward stay as given
    F = _ops.search_frame(td, directions=False)
    cc = -F[:, 9:]
    sw = _ops.SweepIndex(td, sorted_rows=True, frame=F)
    perm = sw.tperm[:, :777].long()
    rows = torch.gather(td, 1, perm.unsqueeze(-1).expand(-1, -1, 6))
    assert torch.equal(sw.tgt_s[:, :777, :6], rows)
    assert torch.equal(sw.tgs4[:, :777, :3], rows[:, :, :3] - cc[:, None, :])
    assert torch.equal(sw.keys[:, :777], rows[:, :, 0] - cc[:, None, 0])


def test_search_frame_picks_a_direction_no_wall_is_perpendicular_to():
    """dicp_search_frame, the direction part: volumetric clouds keep the identity (and with it every bit of the plain x sort); a planar scene
    with walls perpendicular to x (and to y) gets one of the oblique rotations; a slab of a cloud that is thin in x but long in z gets z first.
    Whatever the frame, Q is a rotation, and the sweep on it returns the brute-force kernel's indices in the same frame, index for index."""
    from dicp_amd.synthetic import make_scene_pairs
    vol_s, vol_t = make_pairs(3, 3000, 4096, seed=5)
    sc_s, sc_t = make_scene_pairs(3, 3000, 4096, seed=5)
    g = torch.Generator().manual_seed(1)
    slab = torch.rand((3, 4096, 6), generator=g)
    slab[:, :, 0] *= 0.5
    slab[:, :, 1] *= 2.0
    slab[:, :, 2] *= 30.0
    for name, tgt, src, want in (("volume", vol_t, vol_s, "identity"), ("scene", sc_t, sc_s, "oblique"), ("slab", slab, slab[:, :3000, :3].contiguous() + 0.01, "z")):
        td, sd = tgt.to(DEV), src.to(DEV)
        F = _ops.search_frame(td)
        Q = F[:, :9].reshape(3, 3, 3).double().cpu()
        assert float((Q @ Q.transpose(1, 2) - torch.eye(3, dtype=torch.float64)).abs().max()) < 1e-6 and bool((torch.linalg.det(Q) > 0.999).all()), name
        first = Q[:, 0]
        if want == "identity":
            assert torch.equal(Q, torch.eye(3, dtype=torch.float64).repeat(3, 1, 1)), (name, Q)
        elif want == "z":
            assert torch.equal(first, torch.tensor([0.0, 0.0, 1.0], dtype=torch.float64).repeat(3, 1)), (name, first)
        else:
            assert bool((first.abs().min(dim=1).values > 0.4).all()), (name, first)
        sw = _ops.SweepIndex(td, frame=F)
        pose = _ops.search_pose(None, F)
        brute = _ops.knn(sd, pose, _ops.pack_target(td, F), td.shape[1], _lib.KNN_VALU)
        assert torch.equal(sw.knn(sd, pose, sw.query_order(sd, pose)), brute), name
        # ... and they are the true neighbours (float64, original coordinates), up to float32 near-ties
        d = torch.cdist(sd[0].double(), td[0, :, :3].double())
        v, ix = torch.topk(d, 2, dim=1, largest=False)
        bad = brute[0].long() != ix[:, 0]
        assert float(bad.float().mean()) < 0.02 and bool(((v[:, 1] ** 2 - v[:, 0] ** 2)[bad] < 1e-3).all()), name


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_search_pose_from_T_init(dtype):
    """dicp_search_pose: [Q C | Q r + t] of T_init, the values dicp_loop_init writes as pose_search_0 -- for a pure translation frame
    (Q = I: [C | r - centre], exactly) and for a rotated one (the kernels' fma chains against float64)."""
    g = torch.Generator().manual_seed(2)
    N = 7
    T = torch.rand((N, 4, 4), generator=g, dtype=torch.float64).to(dtype).to(DEV)
    ctr = (torch.rand((N, 3), generator=g, dtype=torch.float64) * 1000).to(dtype).to(DEV)
    eye = torch.eye(3, dtype=dtype, device=DEV).reshape(1, 9).repeat(N, 1)
    frame = torch.cat((eye, -ctr), dim=1).contiguous()
    lib = _lib.load()
    out = torch.empty((N, 12), dtype=dtype, device=DEV)
    _lib.check(lib.dicp_search_pose(_ops._DT[dtype], _ops._p(T), _ops._p(frame), N, _ops._p(out), _ops._stream()), "dicp_search_pose")
    want = torch.cat((T[:, :3, :3].reshape(N, 9), T[:, :3, 3] - ctr), dim=1)
    assert torch.equal(out, want)
    w0 = torch.ones((N, 5), dtype=dtype, device=DEV)
    pose0, alive, nst, ps0 = (torch.empty(sh, dtype=dtype, device=DEV) for sh in ((N, 12), (N,), (N,), (N, 12)))
    _lib.check(lib.dicp_loop_init(_ops._DT[dtype], _ops._p(T), _ops._p(w0), 0.01, 1, N, 5, _ops._p(pose0), _ops._p(alive), _ops._p(nst),
                                  _ops._p(frame), _ops._p(ps0), None, None, None, 0, _ops._stream()), "dicp_loop_init")
    assert torch.equal(ps0, out) and torch.equal(pose0[:, :9], out[:, :9]) and torch.equal(pose0[:, 9:], T[:, :3, 3])
    _lib.check(lib.dicp_search_pose(_ops._DT[dtype], _ops._p(T), None, N, _ops._p(out), _ops._stream()), "dicp_search_pose")
    assert torch.equal(out, pose0)
    # a rotated frame
    Q = torch.linalg.qr(torch.rand((N, 3, 3), generator=g, dtype=torch.float64)).Q
    t = torch.rand((N, 3), generator=g, dtype=torch.float64) * 10
    fr = torch.cat((Q.reshape(N, 9), t), dim=1).to(dtype).to(DEV).contiguous()
    _lib.check(lib.dicp_search_pose(_ops._DT[dtype], _ops._p(T), _ops._p(fr), N, _ops._p(out), _ops._stream()), "dicp_search_pose")
    _lib.check(lib.dicp_loop_init(_ops._DT[dtype], _ops._p(T), _ops._p(w0), 0.01, 1, N, 5, _ops._p(pose0), _ops._p(alive), _ops._p(nst),
                                  _ops._p(fr), _ops._p(ps0), None, None, None, 0, _ops._stream()), "dicp_loop_init")
    assert torch.equal(ps0, out)
    Qd, td, Td = fr[:, :9].reshape(N, 3, 3).double(), fr[:, 9:].double(), T.double()
    want = torch.cat(((Qd @ Td[:, :3, :3]).reshape(N, 9), (Qd @ Td[:, :3, 3:]).squeeze(-1) + td), dim=1)
    np.testing.assert_allclose(npy(out), npy(want), rtol=0, atol=1e-5 if dtype == torch.float32 else 1e-13)


@pytest.mark.parametrize("offset", [0.0, 1000.0, 25000.0])
@pytest.mark.parametrize("icp_type,n", [("pt2pl", 3000), ("pt2pt", 200)])
def test_icp_far_from_the_origin_prunes_and_matches_brute_force(offset, icp_type, n):
    """Clouds in a map frame: the search is centred on the target cloud, so the sweep prunes as it does at the origin, and the
    sweep / brute-force / small-cloud paths (all centred alike) still agree; poses follow the float64 oracle as far as float32
    coordinates of that size allow."""
    N, K = 4, 5
    src, tgt = make_pairs(N, n, n, seed=9, dtype=torch.float32)
    shift = torch.tensor([offset, -0.5 * offset, 0.25 * offset])
    src = src + shift
    tgt = tgt.clone(); tgt[:, :, :3] += shift
    if icp_type == "pt2pt":
        tgt = tgt[:, :, :3].contiguous()
    kw = dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0})
    outs = []
    for variant in (_lib.KNN_VALU, _lib.KNN_SWEEP):
        sd, td = src.to(DEV).requires_grad_(True), tgt.to(DEV).requires_grad_(True)
        icp = ICP(icp_type=icp_type, differentiable=True, max_iterations=K, tolerance=1e-12)
        icp.const_iter = True
        icp.knn_variant = variant
        T0 = torch.eye(4, device=DEV).repeat(N, 1, 1)
        out = icp.icp(sd, td, T0, **kw)
        out["T"].sum().backward()
        outs.append((out, sd.grad, td.grad, dict(icp.knn_stats)))
    scale = max(1.0, offset)
    np.testing.assert_allclose(npy(outs[0][0]["T"]), npy(outs[1][0]["T"]), rtol=0, atol=4e-6 * scale)
    assert torch.equal(outs[0][0]["weights"] > 0, outs[1][0]["weights"] > 0)
    for k in (1, 2):
        assert torch.isfinite(outs[1][k]).all()
    if n >= 1000 and offset <= 1000.0:              # (25 km away float32 ICP itself is coarse -- 2 mm coordinates, a 25 km lever arm --
        frac = float(outs[1][3]["knn_pairs"].sum()) / (float(N) * n * n * K)       # and badly aligned clouds have far neighbours)
        assert frac < 0.25, frac                     # uncentred: 27 % at 1 km with 16384 points, everything at 10 km
    if offset > 1000.0:         # float32 normal equations with 25 km coordinates are not ICP any more (in the reference either)
        return
    ref = O.icp_batched(src.double(), tgt.double(), torch.eye(4, dtype=torch.float64).repeat(N, 1, 1),
                        torch.ones((N, n * (3 if icp_type == "pt2pt" else 1)), dtype=torch.float64),      # (one weight per residual row: ICP.py:508-509)
                        icp_type=icp_type, differentiable=True, max_iterations=K, tolerance=1e-12, const_iter=True, tanh_steepness=5.0, dim=3, **kw)
    # float32 coordinates carry ~6e-8 * |x| of rounding each: the pose error budget grows with the offset
    np.testing.assert_allclose(npy(outs[1][0]["T"]), ref["T"].numpy(), rtol=0, atol=1e-4 + 2e-5 * offset)


@pytest.mark.parametrize("window", [False, True])
def test_icp_sweep_equals_brute_on_synthetic(window):
    """Sweep kNN (+ the sorted-space windowed backward or the plain atomic one) against the brute-force path.
    K = 6 with the default re-sort at iteration 1: iteration 0 takes the atomic form, 1..5 the windowed one."""
    N, n, K = 6, 4096, 6
    src, tgt = make_pairs(N, n, n, seed=5, dtype=torch.float32)
    wgt = torch.rand((N, n), generator=torch.Generator().manual_seed(1)) * 0.5 + 0.5
    outs = []
    for variant in (_lib.KNN_VALU, _lib.KNN_SWEEP):
        sd, td, wd = src.to(DEV).requires_grad_(True), tgt.to(DEV).requires_grad_(True), wgt.to(DEV).requires_grad_(True)
        icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=1e-12)
        icp.const_iter = True
        icp.knn_variant = variant
        icp.bwd_window = window
        T0 = torch.eye(4, device=DEV).repeat(N, 1, 1).requires_grad_(True)
        out = icp.icp(sd, td, T0, weight=wd, trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0})
        out["T"].sum().backward()
        outs.append((out, sd.grad, td.grad, wd.grad, T0.grad))
    # same matches and the same per-point arithmetic; the sweep path takes the normal-equation sums in the search kernel's
    # blocks (query order) and the backward in sorted space, i.e. the same terms in a different grouping
    np.testing.assert_allclose(npy(outs[0][0]["T"]), npy(outs[1][0]["T"]), rtol=0, atol=2e-6)
    np.testing.assert_allclose(npy(outs[0][0]["weights"]), npy(outs[1][0]["weights"]), rtol=0, atol=2e-5)
    for k in (1, 2, 3, 4):
        assert torch.isfinite(outs[1][k]).all()
        np.testing.assert_allclose(npy(outs[0][k]), npy(outs[1][k]), rtol=0, atol=2e-6 * max(1.0, float(outs[0][k].abs().max())))


@pytest.mark.parametrize("dtype,icp_type,N,n,m,const_iter", [
    (torch.float64, "pt2pl", 3, 65, 65, True), (torch.float32, "pt2pt", 7, 200, 150, True),
    (torch.float64, "pt2pl", 2, 300, 333, False), (torch.float32, "pt2pl", 5, 128, 1000, False)])
def test_small_cloud_kernels_equal_multi_kernel_loop(dtype, icp_type, N, n, m, const_iter, monkeypatch):
    """icp_small_forward_kernel / icp_small_backward_kernel (one block per cloud, a whole chunk of iterations per launch)
    against the multi-kernel loop: same matches, same per-point arithmetic, same block reduction -> same results to
    rounding, in const-iteration mode and with the host's convergence checks (several chunks) alike."""
    src, tgt = make_pairs(N, n, m, seed=21, dtype=dtype)
    wgt = torch.rand((N, n), generator=torch.Generator().manual_seed(2), dtype=torch.float64).to(dtype) * 0.5 + 0.5
    outs = []
    for small in (False, True):
        sd, td, wd = src.to(DEV).requires_grad_(True), tgt.to(DEV).requires_grad_(True), wgt.to(DEV).requires_grad_(True)
        T0 = torch.eye(4, dtype=dtype, device=DEV).repeat(N, 1, 1).requires_grad_(True)
        icp = ICP(icp_type=icp_type, differentiable=True, max_iterations=9, tolerance=1e-12 if const_iter else (1e-7 if dtype == torch.float64 else 1e-4))
        icp.const_iter = const_iter
        icp.sync_every = 2
        icp._tuning["small_loop"] = small
        out = icp.icp(sd, td if icp_type == "pt2pl" else td[:, :, :3], T0, weight=wd, trim_dist=4.0, loss_fn={"name": "huber", "metric": 0.7})
        (out["T"][:, :3].sum() + out["pc"].mean()).backward()
        outs.append((out, sd.grad, td.grad, wd.grad, T0.grad))
    a, b = outs
    f64 = dtype == torch.float64
    assert a[0]["deltas"].shape == b[0]["deltas"].shape
    assert torch.equal(a[0]["stats"]["iterations"], b[0]["stats"]["iterations"]) and torch.equal(a[0]["stats"]["converged"], b[0]["stats"]["converged"])
    for key in ("T", "deltas", "weights", "costs"):
        np.testing.assert_allclose(npy(a[0][key]), npy(b[0][key]), rtol=0, atol=(1e-11 if f64 else 5e-6) * max(1.0, float(a[0][key].detach().abs().max())))
    for k in (1, 2, 3, 4):
        assert torch.isfinite(b[k]).all()
        np.testing.assert_allclose(npy(a[k]), npy(b[k]), rtol=0, atol=(1e-9 if f64 else 2e-4) * max(1.0, float(a[k].abs().max())))


@pytest.mark.parametrize("dtype,mode,n,m,local", [
    (torch.float32, "pt2pl", 5000, 5000, True),       # matches near the diagonal: the LDS window takes them
    (torch.float32, "pt2pt", 3000, 9000, True),       # m = 3n: fewer slots per block
    (torch.float64, "pt2pl", 2500, 4100, False),      # random matches: almost everything misses the window (global fallback)
    (torch.float32, "pt2pl", 300, 70, False),         # window covers the whole cloud
])
def test_windowed_backward_kernel_equals_atomic_kernel(dtype, mode, n, m, local):
    """dicp_accumulate_bwd_window (sorted space) + dicp_window_reduce / dicp_permute_add_rows against dicp_accumulate_bwd on the same
    matches, for arbitrary query orders / target orders / match positions (the kernel must not rely on locality)."""
    import ctypes
    lib = _lib.load()
    code = _ops._DT[dtype]
    N, c = 3, 6 if mode == "pt2pl" else 3
    cv = c
    gen = torch.Generator().manual_seed(n * 7 + m)
    src = torch.randn((N, n, 3), generator=gen, dtype=dtype).to(DEV)
    tgt = torch.randn((N, m, c), generator=gen, dtype=dtype).to(DEV)
    w0 = torch.rand((N, n), generator=gen, dtype=dtype).to(DEV)
    m_pad = lib.dicp_padded_targets(m)
    qorder = torch.stack([torch.randperm(n, generator=gen) for _ in range(N)]).to(torch.int32).to(DEV)
    tperm = torch.stack([torch.randperm(m_pad, generator=gen) for _ in range(N)]).to(torch.int32)
    # real rows first (any order), pads last -- like SweepIndex
    tperm = torch.stack([torch.cat((r[r < m], r[r >= m])) for r in tperm]).to(DEV)
    if local:
        ctr = (torch.arange(n)[None, :].float() * (m / n)).long()
        spos = (ctr + torch.randint(-300, 301, (N, n), generator=gen)).clamp(0, m - 1)
        spos[:, ::97] = torch.randint(0, m, (N, len(range(0, n, 97))), generator=gen)      # a few outliers
    else:
        spos = torch.randint(0, m, (N, n), generator=gen)
    spos = spos.to(torch.int32).to(DEV)
    spos[0, 5] = -1                                                                         # "no neighbour" -> sorted row 0
    # spos above is written per SLOT; the kernels index it per QUERY (like idx): query qorder[s] sits in slot s
    spos_q = torch.empty_like(spos).scatter_(1, qorder.long(), spos)
    pose = torch.tensor([[1, 0, 0, 0, 1, 0, 0, 0, 1, 0.1, -0.2, 0.3]] * N, dtype=dtype, device=DEV)
    alive = torch.tensor([1.0, 1.0, 0.0], dtype=dtype, device=DEV)
    gs = torch.randn((N, 36), generator=gen, dtype=dtype).to(DEV)
    gb = torch.randn((N, 6), generator=gen, dtype=dtype).to(DEV)
    P = _loop.LoopConfig(icp_type=mode, differentiable=True, max_iterations=1, tolerance=0.0, trim_dist=2.0, loss_name="huber", loss_metric=1.0,
                        dim=3, const_iter=True, tanh_steepness=10.0, match_ratio_thresh=0.01).params()
    st = _ops._stream()

    # atomic form on the original-order arrays: idx of slot s = tperm[spos[s]], at point qorder[s]
    sp_c = spos.clamp(min=0).long()
    idx_slot = torch.gather(tperm.long(), 1, sp_c)
    idx = torch.empty((N, n), dtype=torch.int64, device=DEV).scatter_(1, qorder.long(), idx_slot).to(torch.int32)
    nb = lib.dicp_accumulate_blocks(n)
    g1 = dict(gsrc=torch.zeros_like(src), gtgt=torch.zeros_like(tgt), gw=torch.zeros_like(w0),
              part=torch.zeros((N, nb, _lib.NBWD_PAD), dtype=dtype, device=DEV))
    _lib.check(lib.dicp_accumulate_bwd(code, ctypes.byref(P), _ops._p(src), _ops._p(tgt), c, _ops._p(idx), _ops._p(pose), _ops._p(w0),
                                       _ops._p(alive), _ops._p(gs), _ops._p(gb), None, N, n, m, _ops._p(g1["gsrc"]), _ops._p(g1["gtgt"]),
                                       _ops._p(g1["gw"]), _ops._p(g1["part"]), st), "dicp_accumulate_bwd")

    src_s = _ops._gather_rows_raw(src, qorder)
    w_s = _ops._gather_rows_raw(w0.unsqueeze(-1), qorder).squeeze(-1).contiguous()
    tgt_s = _ops._gather_rows_raw(tgt, tperm)
    nw = lib.dicp_window_blocks(code, n, m_pad)
    assert nw >= 1
    # the first launch overwrites (its accumulators start as garbage), the second adds
    gsrc_s, gw_s = torch.full_like(src, float("nan")), torch.full_like(w0, 777.0)
    wt = lib.dicp_window_rows(code)
    slab = torch.full((N, nw, wt, cv), float("nan"), dtype=dtype, device=DEV)
    gfar = torch.zeros((N, m_pad, cv), dtype=dtype, device=DEV)
    part = torch.zeros((N, nw, _lib.NBWD_PAD), dtype=dtype, device=DEV)
    # the windows are placed by a DIFFERENT set of matches than the ones being accumulated in the second call
    spos_ref = spos_q if local else torch.randint(0, m, (N, n), generator=gen).to(torch.int32).to(DEV)
    for call in range(2):   # overwrite, then accumulate: two calls = twice the gradient
        _lib.check(lib.dicp_accumulate_bwd_window(code, ctypes.byref(P), _ops._p(src_s), _ops._p(tgt_s), c, _ops._p(spos_q), _ops._p(spos_ref),
                                                  _ops._p(qorder), _ops._p(pose), _ops._p(w_s), _ops._p(alive), _ops._p(gs), _ops._p(gb), None, N, n, m_pad,
                                                  _ops._p(gsrc_s), _ops._p(slab), _ops._p(gfar), _ops._p(gw_s), _ops._p(part), int(call == 0), st),
                   "dicp_accumulate_bwd_window")
    # un-permute: = into garbage (dicp_permute_rows) for the points, += into zeros (dicp_permute_add_rows) for the weights
    gsrc, gw, gtgt = torch.full_like(src, float("nan")), torch.zeros_like(w0), torch.zeros_like(tgt)
    _lib.check(lib.dicp_permute_rows(code, _ops._p(gsrc_s), _ops._p(qorder), N, n, n, n, 3, 3, _ops._p(gsrc), n, 3, st), "permute")
    _lib.check(lib.dicp_permute_add_rows(code, _ops._p(gw_s), _ops._p(qorder), N, n, n, n, 1, 1, _ops._p(gw), n, 1, st), "permute")
    _lib.check(lib.dicp_window_reduce(code, _ops._p(slab), _ops._p(spos_ref), _ops._p(qorder), _ops._p(tperm), _ops._p(gfar), None, N, n, m, m_pad, cv,
                                      _ops._p(gtgt), c, 0, st), "dicp_window_reduce")
    tol = 1e-11 if dtype == torch.float64 else 2e-4
    scale = lambda a: max(1.0, float(a.abs().max()))
    for a, b, nm in ((gsrc, g1["gsrc"], "gsrc"), (gw, g1["gw"], "gw"), (gtgt, g1["gtgt"], "gtgt")):
        assert torch.isfinite(a).all(), nm
        assert float((a - 2 * b).abs().max()) <= tol * scale(b), nm
    pa, pb = part.sum(dim=1), g1["part"].sum(dim=1)
    assert float((pa - pb).abs().max()) <= tol * scale(pb) * 10
    assert float(gfar[:, m:].abs().max() if m_pad > m else 0.0) == 0.0         # pad rows are never matched


# ------------------------------------------------------------------- SVD point-to-point (a-12 / f-4)
def kabsch_np(p, y, w=None):
    """Published Kabsch/Umeyama: the rigid transform minimising sum w |C p + r - y|^2."""
    w = np.ones(len(p)) if w is None else w
    mus, mut = (w[:, None] * p).sum(0) / w.sum(), (w[:, None] * y).sum(0) / w.sum()
    W = ((w[:, None] * (y - mut)).T @ (p - mus)) / w.sum()
    U, S, Vt = np.linalg.svd(W)
    C = U @ np.diag([1, 1, np.linalg.det(U) * np.linalg.det(Vt)]) @ Vt
    return C, mut - C @ mus


def test_svd_icp_planar_matches_reference(golden, scan_map):
    """The reference's pt2pt_dICP_SVD on its own planar test pair (where its V-for-V^T slip is harmless)."""
    scan, mp = scan_map
    g = golden("svd_planar")
    icp = ICP(icp_type="pt2pt", differentiable=False, max_iterations=100, tolerance=1e-20)
    ps, T = icp.pt2pt_dICP_SVD(t(scan[:, :3]), t(mp[:, :3]), torch.eye(4, dtype=torch.float64, device=DEV))
    assert ps.shape == (65, 3) and T.shape == (4, 4)                       # unbatched in, unbatched out (ICP.py:591)
    np.testing.assert_allclose(npy(T), g["T"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(npy(ps), g["pc"], rtol=0, atol=1e-10)
    err = tran2vec(g["T_ts_true"] @ np.linalg.inv(npy(T)))
    assert np.linalg.norm(err) < 1e-10
    # and it agrees with the integrated Gauss-Newton pt2pt solver on the same data (SURVEY 8a-12 pin)
    gn = ICP(icp_type="pt2pt", differentiable=True, max_iterations=100, tolerance=1e-10)
    Tg = gn.icp(t(scan[:, :3]), t(mp[:, :3]), torch.eye(4, dtype=torch.float64, device=DEV), dim=2)["T"]
    np.testing.assert_allclose(npy(T), npy(Tg)[0], rtol=0, atol=1e-9)


def test_svd_icp_T_init_composition_matches_reference(golden, scan_map):
    """T_init != I: the reference does not move the points by it -- T_ts = T_total @ T_init, ps = T_total source
    (ICP.py:545-547,578).  Fixture generated by the reference's own function (make_golden.py:svd_tinit)."""
    scan, mp = scan_map
    g = golden("svd_tinit")
    icp = ICP(icp_type="pt2pt", differentiable=False, max_iterations=100, tolerance=1e-20)
    T0 = t(g["T_init"], grad=True)
    src = t(scan[:, :3], grad=True)
    ps, T = icp.pt2pt_dICP_SVD(src, t(mp[:, :3]), T0)
    np.testing.assert_allclose(npy(T), g["T"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(npy(ps), g["pc"], rtol=0, atol=1e-10)
    (T.sum() + ps.sum()).backward()                         # the product T_total @ T_init carries gradient to T_init
    assert T0.grad is not None and src.grad is not None and bool(torch.isfinite(T0.grad).all())
    icp.svd_seed_T_init = True                              # build-specific: T_init as the starting pose of the search
    ps2, T2 = icp.pt2pt_dICP_SVD(t(scan[:, :3]), t(mp[:, :3]), t(g["T_init"]))
    np.testing.assert_allclose(npy(ps2), g["pc"], rtol=0, atol=1e-9)       # same aligned cloud either way


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_svd_icp_3d_batched_vs_numpy_kabsch(dtype):
    """General 3-D clouds: each iterate must equal numpy Kabsch on the current correspondences, and the final
    pose must recover the planted motion -- the case the reference's own function gets wrong."""
    N, n, m = 3, 600, 700
    src, tgt = make_pairs(N, n, m, seed=21, dtype=torch.float64, max_rot=0.2, max_trans=0.5)
    w = torch.rand((N, n), generator=torch.Generator().manual_seed(0), dtype=torch.float64) + 0.1
    icp = ICP(icp_type="pt2pt", max_iterations=20, tolerance=1e-30)
    icp.const_iter = True
    ps, T = icp.pt2pt_dICP_SVD(src.to(dtype).to(DEV), tgt.to(dtype).to(DEV), torch.eye(4, dtype=dtype, device=DEV).repeat(N, 1, 1),
                               weight=w.to(dtype).to(DEV))
    assert ps.shape == (N, n, 3) and T.shape == (N, 4, 4)
    for b in range(N):
        C, r = np.eye(3), np.zeros(3)
        for _ in range(20):
            idx = O.nn_index(torch.tensor(src[b].numpy() @ C.T + r)[None], tgt[b:b + 1, :, :3])[0].numpy()
            C, r = kabsch_np(src[b].numpy(), tgt[b, idx, :3].numpy(), w[b].numpy())
        tol = 1e-9 if dtype == torch.float64 else 1e-4
        np.testing.assert_allclose(npy(T)[b, :3, :3], C, rtol=0, atol=tol)
        np.testing.assert_allclose(npy(T)[b, :3, 3], r, rtol=0, atol=tol)
        assert abs(np.linalg.det(npy(T)[b, :3, :3].astype(np.float64)) - 1.0) < 1e-5
    assert float((ps - tgt.to(dtype).to(DEV)[:, :, :3].mean()).abs().max()) < 50      # sane output


@pytest.mark.parametrize("knn", [_lib.KNN_VALU, _lib.KNN_SWEEP])
def test_svd_loop_freezes_each_pair_where_its_own_call_stops(knn):
    """The fused SVD loop (dicp_kabsch_forward) with the tolerance check on device: every pair of a batch stops at the iteration a
    call of its own stops at (ICP.py:585-586) -- same pose, same iteration count, same gradients -- however many iterations the
    slowest pair of the batch needs and however rarely the host looks (sync_every); ragged lists included."""
    N, n, m = 4, 900, 1100
    src, tgt = make_pairs(N, n, m, seed=77, dtype=torch.float64, max_rot=0.06, max_trans=0.3, noise=0.0)
    lens = [900, 500, 900, 120]
    S = [src[b, :lens[b]].to(DEV).requires_grad_(True) for b in range(N)]
    Tg = [tgt[b, :, :3].to(DEV).requires_grad_(True) for b in range(N)]
    T0 = [torch.eye(4, dtype=torch.float64, device=DEV) for _ in range(N)]
    outs = {}
    for every in (1, 5):
        icp = ICP(icp_type="pt2pt", max_iterations=40, tolerance=1e-18)
        icp.knn_variant = knn
        icp.sync_every = every
        ps, T = icp.pt2pt_dICP_SVD([x.detach().clone().requires_grad_(True) for x in S], Tg, T0)
        outs[every] = (T.detach().clone(), icp.svd_stats["iterations"].clone())
    assert torch.equal(outs[1][0], outs[5][0]) and torch.equal(outs[1][1], outs[5][1])
    icp = ICP(icp_type="pt2pt", max_iterations=40, tolerance=1e-18)
    icp.knn_variant = knn
    ps, T = icp.pt2pt_dICP_SVD(S, Tg, T0)
    (T * torch.arange(16, dtype=torch.float64, device=DEV).reshape(4, 4)).sum().backward()
    its = icp.svd_stats["iterations"]
    assert float(its.min()) < 40, its                        # pairs stop on their own (each at its own iteration: compared per item below)
    for b in range(N):
        s1, t1 = S[b].detach().clone().requires_grad_(True), Tg[b].detach().clone().requires_grad_(True)
        one = ICP(icp_type="pt2pt", max_iterations=40, tolerance=1e-18)
        one.knn_variant = knn
        p1, T1 = one.pt2pt_dICP_SVD(s1, t1, T0[b])
        (T1 * torch.arange(16, dtype=torch.float64, device=DEV).reshape(4, 4)).sum().backward()
        assert float(one.svd_stats["iterations"][0]) == float(its[b])
        np.testing.assert_allclose(npy(T)[b], npy(T1), rtol=0, atol=1e-12)
        np.testing.assert_allclose(npy(ps)[b, :lens[b]], npy(p1), rtol=0, atol=1e-11)
        np.testing.assert_allclose(npy(S[b].grad), npy(s1.grad), rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(npy(Tg[b].grad), npy(t1.grad), rtol=1e-9, atol=1e-11)


def test_svd_icp_gradients_vs_autograd():
    """Gradient of the final pose w.r.t. source, target and weight == torch autograd through one Kabsch solve
    on the final correspondences (the composed updates of the reference telescope to exactly that)."""
    N, n, m = 2, 300, 350
    src, tgt = make_pairs(N, n, m, seed=33, dtype=torch.float64, max_rot=0.15)
    tg3 = tgt[:, :, :3].contiguous()
    w = torch.rand((N, n), generator=torch.Generator().manual_seed(1), dtype=torch.float64) + 0.2
    sd, td, wd = src.to(DEV).requires_grad_(True), tg3.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    icp = ICP(icp_type="pt2pt", max_iterations=15, tolerance=1e-30)
    icp.const_iter = True
    ps, T = icp.pt2pt_dICP_SVD(sd, td, torch.eye(4, dtype=torch.float64, device=DEV).repeat(N, 1, 1), weight=wd)
    gT = torch.randn((N, 4, 4), generator=torch.Generator().manual_seed(2), dtype=torch.float64)
    gP = torch.randn((N, n, 3), generator=torch.Generator().manual_seed(3), dtype=torch.float64)
    ((T * gT.to(DEV)).sum() + (ps * gP.to(DEV)).sum()).backward()
    for b in range(N):
        # final correspondences = NN under the pose of the second-to-last iterate; recover them from the result
        sc, tc, wc = (x[b].clone().requires_grad_(True) for x in (src, tg3, w))
        C, r = np.eye(3), np.zeros(3)
        for _ in range(15):
            idx = O.nn_index(torch.tensor(src[b].numpy() @ C.T + r)[None], tg3[b:b + 1])[0]
            C, r = kabsch_np(src[b].numpy(), tg3[b, idx].numpy(), w[b].numpy())
        y = tc[idx]
        S0 = wc.sum()
        mus, mut = (wc[:, None] * sc).sum(0) / S0, (wc[:, None] * y).sum(0) / S0
        W = (wc[:, None, None] * y[:, :, None] * sc[:, None, :]).sum(0) / S0 - mut[:, None] * mus[None, :]
        U, S, Vh = torch.linalg.svd(W)
        Ct = U @ torch.diag(torch.stack([torch.ones((), dtype=torch.float64), torch.ones((), dtype=torch.float64), torch.det(U) * torch.det(Vh)])) @ Vh
        rt = mut - Ct @ mus
        Tt = torch.eye(4, dtype=torch.float64)
        Tt = torch.cat((torch.cat((Ct, rt[:, None]), dim=1), torch.tensor([[0, 0, 0, 1.0]], dtype=torch.float64)), dim=0)
        ((Tt * gT[b]).sum() + ((sc @ Ct.T + rt) * gP[b]).sum()).backward()
        np.testing.assert_allclose(npy(T)[b], Tt.detach().numpy(), rtol=0, atol=1e-10)
        np.testing.assert_allclose(npy(sd.grad)[b], sc.grad.numpy(), rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(npy(td.grad)[b], tc.grad.numpy(), rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(npy(wd.grad)[b], wc.grad.numpy(), rtol=1e-7, atol=1e-9)


def test_svd_icp_config2_shape_and_trim():
    """configs[1]: B=32 x 4096-pt clouds, point-to-point with the SVD step; trim gate and tolerance stop."""
    N, n = 32, 4096
    src, tgt = make_pairs(N, n, n, seed=1, dtype=torch.float32)
    icp = ICP(icp_type="pt2pt", max_iterations=30, tolerance=1e-3)
    ps, T = icp.pt2pt_dICP_SVD(src.to(DEV), tgt.to(DEV), torch.eye(4, device=DEV).repeat(N, 1, 1), trim_dist=5.0)
    gn = ICP(icp_type="pt2pt", differentiable=True, max_iterations=30, tolerance=1e-7)
    Tg = gn.icp(src.to(DEV), tgt[:, :, :3].contiguous().to(DEV), torch.eye(4, device=DEV).repeat(N, 1, 1), trim_dist=5.0)["T"]
    np.testing.assert_allclose(npy(T), npy(Tg), rtol=0, atol=2e-3)            # same optimum as Gauss-Newton pt2pt
    assert icp.svd_stats["costs"].shape[0] == N and float(icp.svd_stats["iterations"].max()) <= 30
    # empty / switched-off cloud: zero total weight -> identity step (no NaN)
    ps0, T0 = icp.pt2pt_dICP_SVD(src[:2].to(DEV), tgt[:2].to(DEV), torch.eye(4, device=DEV).repeat(2, 1, 1),
                                 weight=torch.zeros((2, n), device=DEV))
    assert torch.equal(T0, torch.eye(4, device=DEV).repeat(2, 1, 1))


# ----------------------------------------------------------------------- robustness of the call surface
def test_views_partial_grads_and_no_grad():
    """Non-contiguous inputs, gradients for a subset of the inputs, no_grad, tiny clouds."""
    N, n, m = 3, 200, 260
    src6, tgt = make_pairs(N, n, m, seed=40, dtype=torch.float64)
    src6 = torch.cat((src6, torch.zeros(N, n, 3, dtype=torch.float64)), dim=2)       # (N,n,6): xyz is a strided view
    kw = dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=3)
    w = torch.rand((N, n), generator=torch.Generator().manual_seed(4), dtype=torch.float64) + 0.5
    sc, tc, wc = src6.clone().requires_grad_(True), tgt.clone().requires_grad_(True), w.clone().requires_grad_(True)
    T0 = torch.eye(4, dtype=torch.float64).repeat(N, 1, 1)
    T0[:, :3, 3] = 0.05
    ref = O.icp_batched(sc[:, :, :3], tc, T0, wc, icp_type="pt2pl", differentiable=True, max_iterations=4, tolerance=1e-12,
                        const_iter=True, **kw)
    ref["T"].sum().backward()
    icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=4, tolerance=1e-12)
    icp.const_iter = True
    # (a) only the source wants a gradient; 6-column source -> strided xyz view inside
    sd = src6.to(DEV).requires_grad_(True)
    out = icp.icp(sd, tgt.to(DEV), T0.to(DEV), weight=w.to(DEV), **kw)
    out["T"].sum().backward()
    np.testing.assert_allclose(npy(out["T"]), npy(ref["T"]), rtol=0, atol=1e-10)
    np.testing.assert_allclose(npy(sd.grad), npy(sc.grad), rtol=0, atol=1e-9)
    assert float(sd.grad[:, :, 3:].abs().max()) == 0.0
    # (b) only the weight wants a gradient; target passed as a transposed-back (non-contiguous) view
    wd = w.to(DEV).requires_grad_(True)
    tview = tgt.to(DEV).transpose(1, 2).contiguous().transpose(1, 2)
    assert not tview.is_contiguous()
    out = icp.icp(src6.to(DEV), tview, T0.to(DEV), weight=wd, **kw)
    out["T"].sum().backward()
    np.testing.assert_allclose(npy(wd.grad), npy(wc.grad), rtol=0, atol=1e-9)
    # (c) T_init gradient
    Td = T0.to(DEV).requires_grad_(True)
    Tc = T0.clone().requires_grad_(True)
    O.icp_batched(src6[:, :, :3], tgt, Tc, w, icp_type="pt2pl", differentiable=True, max_iterations=4, tolerance=1e-12,
                  const_iter=True, **kw)["T"].sum().backward()
    icp.icp(src6.to(DEV), tgt.to(DEV), Td, weight=w.to(DEV), **kw)["T"].sum().backward()
    np.testing.assert_allclose(npy(Td.grad), npy(Tc.grad), rtol=0, atol=1e-9)
    # (d) no_grad: nothing is saved, outputs carry no graph
    with torch.no_grad():
        out = icp.icp(src6.to(DEV), tgt.to(DEV), T0.to(DEV), weight=w.to(DEV), **kw)
    assert not out["T"].requires_grad and not out["pc"].requires_grad
    np.testing.assert_allclose(npy(out["T"]), npy(ref["T"]), rtol=0, atol=1e-10)


@pytest.mark.parametrize("n,m", [(1, 1), (1, 70), (63, 5), (300, 64), (65, 65)])
def test_tiny_and_odd_sizes(n, m):
    g = torch.Generator().manual_seed(n * 100 + m)
    src = torch.rand((2, n, 3), generator=g, dtype=torch.float64)
    tgt = torch.rand((2, m, 6), generator=g, dtype=torch.float64)
    T0 = torch.eye(4, dtype=torch.float64).repeat(2, 1, 1)
    for icp_type in ("pt2pl", "pt2pt"):
        rows = 3 if icp_type == "pt2pt" else 1
        tg = tgt if icp_type == "pt2pl" else tgt[:, :, :3].contiguous()
        ref = O.icp_batched(src, tg, T0, torch.ones(2, n * rows, dtype=torch.float64), icp_type=icp_type, differentiable=True,
                            max_iterations=3, tolerance=1e-14, const_iter=True, trim_dist=5.0)
        icp = ICP(icp_type=icp_type, differentiable=True, max_iterations=3, tolerance=1e-14)
        icp.const_iter = True
        for variant in (_lib.KNN_VALU, _lib.KNN_SWEEP):
            icp.knn_variant = variant
            out = icp.icp(src.to(DEV), tg.to(DEV), T0.to(DEV), trim_dist=5.0)
            assert out["weights"].shape == (2, 3, n * rows, 1) and out["deltas"].shape == (2, 3, 6, 1)
            assert bool(torch.isfinite(out["T"]).all())
            # iteration 0 depends only on T_init: weights and cost must agree whatever the conditioning
            np.testing.assert_allclose(npy(out["weights"])[:, 0], npy(ref["weights"])[:, 0], rtol=0, atol=1e-12)
            np.testing.assert_allclose(npy(out["costs"])[:, 0], npy(ref["costs"])[:, 0], rtol=1e-10, atol=1e-14)
            if min(n, m) >= 64:     # fewer distinct correspondences than unknowns leaves only the 1e-12 regulariser
                np.testing.assert_allclose(npy(out["T"]), npy(ref["T"]), rtol=0, atol=1e-9)


def test_errors_on_device_inputs():
    src, tgt = make_pairs(1, 50, 60, seed=1, dtype=torch.float32)
    icp = ICP(icp_type="pt2pl")
    with pytest.raises(AssertionError):
        icp.icp(src.to(DEV), tgt[:, :, :3].to(DEV), torch.eye(4, device=DEV))          # pt2pl needs normals (ICP.py:103)
    with pytest.raises(AssertionError):
        icp.icp(src.to(DEV), tgt.double().to(DEV), torch.eye(4, device=DEV))           # dtype mismatch (ICP.py:96)
    with pytest.raises(ValueError):
        icp.icp(src.to(DEV), tgt.to(DEV), torch.eye(4, device=DEV), loss_fn={"name": "tukey", "metric": 1.0})   # loss.py:19
    with pytest.raises(TypeError):
        icp.icp(src.half().to(DEV), tgt.half().to(DEV), torch.eye(4, device=DEV).half())


# ------------------------------------------------------------- fused Gumbel-softmax soft kNN (a-15 / f-1)
@pytest.mark.parametrize("dtype,c", [(torch.float32, 6), (torch.float32, 3), (torch.float64, 6)])
def test_gumbel_kernels_vs_oracle_with_injected_noise(dtype, c):
    g = torch.Generator().manual_seed(12)
    N, n, m = 2, 300, 777                                   # several LDS tiles, ragged tails
    x = torch.rand((N, n, 3), generator=g, dtype=torch.float64) * 3
    y = torch.rand((N, m, c), generator=g, dtype=torch.float64) * 3
    U = torch.rand((N, n, m), generator=g, dtype=torch.float64)
    cot = torch.randn((N, n, c), generator=g, dtype=torch.float64)
    xc, yc = x.to(dtype).clone().requires_grad_(True), y.to(dtype).clone().requires_grad_(True)
    ref = O.nn_gumbel(xc, yc, 1e-10, 0.5, U=U.to(dtype))
    (ref * cot.to(dtype)).sum().backward()
    xd, yd = x.to(dtype).to(DEV).detach().requires_grad_(True), y.to(dtype).to(DEV).detach().requires_grad_(True)
    soft = nn(differentiable=True, use_gumbel=True, eps=1e-10, tau=0.5)
    out = soft.find_nn(xd, yd, U=U.to(dtype).to(DEV))
    (out * cot.to(dtype).to(DEV)).sum().backward()
    tol = dict(rtol=0, atol=1e-10) if dtype == torch.float64 else dict(rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(npy(out), ref.detach().numpy(), **tol)
    gt = dict(rtol=0, atol=1e-9) if dtype == torch.float64 else dict(rtol=2e-3, atol=2e-4)
    np.testing.assert_allclose(npy(xd.grad), xc.grad.numpy(), **gt)
    np.testing.assert_allclose(npy(yd.grad), yc.grad.numpy(), **gt)


def test_gumbel_in_kernel_noise_is_seeded_uniform_and_consistent():
    """U=None: the noise comes from the in-kernel counter hash.  Same torch seed -> same output; the backward
    regenerates the same noise (finite-difference check); the sampled correspondence follows softmax(-d^2)."""
    g = torch.Generator().manual_seed(2)
    x = (torch.rand((1, 40, 3), generator=g, dtype=torch.float64)).to(DEV)
    y = (torch.rand((1, 50, 6), generator=g, dtype=torch.float64)).to(DEV)
    soft = nn(differentiable=True, use_gumbel=True, eps=1e-10, tau=0.3)
    torch.manual_seed(7)
    a = soft.find_nn(x, y)
    torch.manual_seed(7)
    b = soft.find_nn(x, y)
    torch.manual_seed(8)
    c2 = soft.find_nn(x, y)
    assert torch.equal(a, b) and not torch.equal(a, c2)
    # gradient w.r.t. x and y by central differences at a fixed seed
    from dicp_amd._ops import gumbel_nn
    xr, yr = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
    w = torch.randn(a.shape, generator=torch.Generator().manual_seed(1), dtype=torch.float64).to(DEV)
    (gumbel_nn(xr, yr, 1e-10, 0.3, seed=99) * w).sum().backward()
    f = lambda xx, yy: float((gumbel_nn(xx, yy, 1e-10, 0.3, seed=99) * w).sum())
    h = 1e-6
    for (b_, i_, k_) in ((0, 3, 0), (0, 17, 2), (0, 39, 1)):
        xp, xm = x.clone(), x.clone()
        xp[b_, i_, k_] += h
        xm[b_, i_, k_] -= h
        assert abs((f(xp, y) - f(xm, y)) / (2 * h) - float(xr.grad[b_, i_, k_])) < 1e-5
    for (b_, j_, k_) in ((0, 0, 0), (0, 21, 4), (0, 49, 2)):
        yp, ym = y.clone(), y.clone()
        yp[b_, j_, k_] += h
        ym[b_, j_, k_] -= h
        assert abs((f(x, yp) - f(x, ym)) / (2 * h) - float(yr.grad[b_, j_, k_])) < 1e-5
    # distribution: as tau -> 0 the output is the row of argmax(-d^2 + g), a sample from softmax(-d^2)
    q = torch.zeros((1, 1, 3), dtype=torch.float64, device=DEV)
    t2 = torch.tensor([[[0.3, 0, 0], [1.0, 0, 0]]], dtype=torch.float64, device=DEV)
    hard = nn(differentiable=True, use_gumbel=True, eps=1e-10, tau=1e-3)
    torch.manual_seed(0)
    picks = torch.cat([hard.find_nn(q, t2)[:, :, 0] for _ in range(400)]).flatten()
    p_near = float((picks < 0.65).double().mean())
    want = float(torch.softmax(torch.tensor([-0.09, -1.0]), 0)[0])                 # 0.713
    assert abs(p_near - want) < 0.08


def test_gumbel_at_benchmark_cloud_size_runs():
    """16384-pt clouds: 2.7e8 noisy logits per cloud, never materialised."""
    src, tgt = make_pairs(2, 16384, 16384, seed=3, dtype=torch.float32)
    xd, yd = src.to(DEV).requires_grad_(True), tgt.to(DEV).requires_grad_(True)
    torch.manual_seed(0)
    out = nn(differentiable=True, use_gumbel=True, eps=1e-10, tau=0.1).find_nn(xd, yd)
    out.sum().backward()
    assert out.shape == (2, 16384, 6) and bool(torch.isfinite(out).all())
    assert bool(torch.isfinite(xd.grad).all() and torch.isfinite(yd.grad).all())
    # a convex combination of target rows: inside the targets' bounding box
    assert float(out[:, :, :3].abs().max()) <= float(tgt[:, :, :3].abs().max()) + 1e-4


@pytest.mark.parametrize("icp_type", ["pt2pl", "pt2pt"])
def test_icp_with_gumbel_correspondence_vs_oracle(icp_type, monkeypatch):
    """config functionality.gumbel = True (ICP.py:40-44): the loop runs on soft neighbours.  The oracle draws its
    noise with torch.rand (nn.py:60); both sides are fed the same pre-drawn U_k."""
    N, n, m, K = 2, 120, 150, 3
    src, tgt = make_pairs(N, n, m, seed=50, dtype=torch.float64, max_rot=0.03, max_trans=0.1)
    tg = tgt if icp_type == "pt2pl" else tgt[:, :, :3].contiguous()
    g = torch.Generator().manual_seed(3)
    Us = [torch.rand((N, n, m), generator=g, dtype=torch.float32) for _ in range(K)]
    w = torch.rand((N, n), generator=g, dtype=torch.float64) + 0.5
    kw = dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=3)
    # oracle with injected draws
    draws = iter(Us)
    monkeypatch.setattr(torch, "rand", lambda *a, **k: next(draws))
    sc, tc, wc = src.clone().requires_grad_(True), tg.clone().requires_grad_(True), w.clone().requires_grad_(True)
    T0c = torch.eye(4, dtype=torch.float64).repeat(N, 1, 1).requires_grad_(True)
    rows = 3 if icp_type == "pt2pt" else 1
    ref = O.icp_batched(sc, tc, T0c, wc.repeat_interleave(rows, dim=1), icp_type=icp_type, differentiable=True, max_iterations=K,
                        tolerance=1e-14, const_iter=True, use_gumbel=True, gumbel_eps=1e-10, gumbel_tau=0.1, **kw)
    monkeypatch.undo()
    gT = torch.randn((N, 4, 4), generator=g, dtype=torch.float64)
    (ref["T"] * gT).sum().backward()
    # HIP path
    icp = ICP(icp_type=icp_type, differentiable=True, max_iterations=K, tolerance=1e-14)
    icp.const_iter = True
    icp.nn.use_gumbel, icp.nn.eps, icp.nn.tau = True, 1e-10, 0.1
    icp.nn._inject_U = [u.double().to(DEV) for u in Us]
    sd, td, wd = src.to(DEV).requires_grad_(True), tg.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    T0d = torch.eye(4, dtype=torch.float64, device=DEV).repeat(N, 1, 1).requires_grad_(True)
    out = icp.icp(sd, td, T0d, weight=wd, **kw)
    assert type(out["T"].grad_fn).__name__ == "ICPLoopBackward"      # the library loop (dicp_icp_forward with DICP_KNN_GUMBEL): one node per call
    (out["T"] * gT.to(DEV)).sum().backward()
    # noise is float32 in the reference (torch.rand default dtype) and float64 here: agreement to ~1e-6
    np.testing.assert_allclose(npy(out["T"]), npy(ref["T"]), rtol=0, atol=2e-6)
    np.testing.assert_allclose(npy(out["deltas"]), npy(ref["deltas"]), rtol=0, atol=2e-6)
    np.testing.assert_allclose(npy(out["weights"]), npy(ref["weights"]), rtol=0, atol=1e-5)
    assert out["weights"].shape == ref["weights"].shape and out["costs"].shape == ref["costs"].shape
    for a, b in ((sd, sc), (td, tc), (wd, wc), (T0d, T0c)):
        scale = max(1.0, float(b.grad.abs().max()))
        np.testing.assert_allclose(npy(a.grad), npy(b.grad), rtol=0, atol=2e-5 * scale)


def test_sync_every_and_history_slabs_do_not_change_results(golden, monkeypatch):
    """f-3: the loop is enqueued in segments; checking convergence only every few iterations (converged clouds
    are frozen) and splitting the histories into several slabs must give the reference's result exactly."""
    g = golden("input_types")
    S = [t(g["s0"]), t(g["s1"]), t(g["s2"])]
    Tg = [t(g["t0"]), t(g["t1"]), t(g["t2"])]
    T0 = torch.stack([torch.eye(4, dtype=torch.float64, device=DEV)] * 3)
    kw = dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=2)
    for sync_every, slab in ((1, 1 << 29), (4, 1 << 29), (7, 1), (25, 1 << 29)):
        monkeypatch.setattr(_ops, "HIST_CHUNK_BYTES", slab)          # slab = 1 byte -> one iteration per slab
        icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=25, tolerance=1e-8)
        icp.sync_every = sync_every
        src = [x.clone().requires_grad_(True) for x in S]
        out = icp.icp(src, Tg, T0, **kw)
        check_result(out, g, "batch_")
        out["T"].sum().backward()
        if sync_every == 1:
            ref_grads = [x.grad.clone() for x in src]
        else:
            for a, b in zip(src, ref_grads):
                np.testing.assert_allclose(npy(a.grad), npy(b), rtol=0, atol=1e-12)


@pytest.mark.parametrize("m,c", [(1, 3), (63, 6), (64, 3), (1000, 6), (4097, 3), (16384, 6), (16321, 6)])
def test_sweep_sort_is_the_stable_torch_sort(m, c):
    """dicp_sweep_sort (LDS radix sort of the target x keys) against torch.sort(stable=True): same sorted keys, same
    permutation -- with duplicates, negative values, zeros of both signs, infinities and NaN among the keys -- and the
    SweepIndex built on either is the same index."""
    from dicp_amd import _ops, _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(m)
    N = 5
    tgt = torch.randn((N, m, c), generator=g) * 3
    tgt[1, :, 0] = torch.randint(-4, 4, (m,), generator=g).float()                    # heavy duplicates
    tgt[2, :, 0] = (tgt[2, :, 0] * 1e30).clamp(-3e38, 3e38)                           # large magnitudes
    if m >= 64:
        tgt[3, 5, 0], tgt[3, 9, 0], tgt[3, 11, 0], tgt[3, 20, 0] = 0.0, -0.0, float("inf"), float("-inf")
        tgt[3, 30, 0], tgt[3, 31, 0] = float("nan"), -float("nan")
        tgt[4, :, 0] = tgt[4, :, 0].abs().neg()                                        # all negative
    tgt = tgt.cuda()
    m_pad = lib.dicp_padded_targets(m)
    key = torch.full((N, m_pad), float("nan"), device="cuda")      # pad slots carry the largest key: after every real row, NaN rows included
    key[:, :m] = tgt[:, :, 0]
    ref_keys, ref_order = torch.sort(key, dim=1, stable=True)
    keys = torch.empty((N, m_pad), device="cuda")
    perm = torch.empty((N, m_pad), dtype=torch.int32, device="cuda")
    _lib.check(lib.dicp_sweep_sort(_lib.F32, _ops._p(tgt), c, None, None, N, m, m_pad, _ops._p(keys), _ops._p(perm), 0, None, None, None, 0, _ops._stream()), "sort")
    plain = [0, 1, 2, 4] if m >= 64 else list(range(N))                               # row 3 holds the special values
    assert torch.equal(perm.long()[plain], ref_order[plain])
    assert torch.equal(torch.nan_to_num(keys[plain], nan=7.0), torch.nan_to_num(ref_keys[plain], nan=7.0))
    assert bool(torch.isnan(keys[:, m:]).all()) and bool((p_real := perm.long()[:, :m]).max() < m) and bool(p_real.min() >= 0)   # sorted slots [0, m) are the real rows
    # every row, the special one included, by the properties of a stable sort under "NaN last, -0 == +0":
    p64 = perm.long()
    assert torch.equal(torch.sort(p64, dim=1).values, torch.arange(m_pad, device="cuda").expand(N, -1))
    got = torch.gather(key, 1, p64)
    assert torch.equal(torch.nan_to_num(keys, nan=7.0), torch.nan_to_num(got, nan=7.0))  # equal as floats (-0 == +0)
    a, b = keys[:, :-1], keys[:, 1:]
    assert bool(((a <= b) | torch.isnan(b)).all()) and bool((~torch.isnan(a) | torch.isnan(b)).all())
    tie = (a == b) | (torch.isnan(a) & torch.isnan(b))
    assert bool((p64[:, :-1] < p64[:, 1:])[tie].all())                                # ties keep their index order
    if m >= 64:
        tgt[3, 30:32, 0] = 1.0                                                         # the index itself: finite clouds
        tgt[3, 11, 0], tgt[3, 20, 0] = 2.0, -2.0
    # the index itself against one built from torch.sort's permutation
    a = _ops.SweepIndex(tgt, sorted_rows=True)
    key = torch.full((N, m_pad), float("nan"), device="cuda")
    key[:, :m] = tgt[:, :, 0]
    want = torch.sort(key, dim=1, stable=True).indices
    assert torch.equal(a.tperm.long(), want)
    rows = torch.gather(tgt, 1, want[:, :m].unsqueeze(-1).expand(-1, -1, c))
    assert torch.equal(a.tgt_s[:, :m, :c], rows) and torch.equal(a.tgs4[:, :m, :3], rows[:, :, :3])
    assert bool((a.bucket[:, 0] == 0).all()) and bool((a.bucket[:, -1] <= m).all()) and bool((a.bucket[:, 1:] >= a.bucket[:, :-1]).all())


@pytest.mark.parametrize("dtype,m", [(torch.float64, 300), (torch.float64, 16384), (torch.float32, 16385), (torch.float32, 40000), (torch.float64, 70001)])
def test_chunked_key_sort_beyond_the_lds_sort(dtype, m):
    """dicp_sweep_sort for float64 keys and for more than 16384 slots (chunked LSD radix through scratch): the stable
    ascending order torch.sort(stable=True) gives, duplicates, -0 / +0, infinities and NaNs included; pads after everything."""
    N, c = 3, 3
    g = torch.Generator().manual_seed(m)
    tgt = (torch.rand((N, m, c), generator=g, dtype=torch.float64) * 40 - 20).to(dtype)
    tgt[1, :, 0] = torch.randint(-3, 4, (m,), generator=g).to(dtype)                  # many equal keys: stability decides
    if m >= 64:
        tgt[2, 5, 0], tgt[2, 9, 0], tgt[2, 11, 0], tgt[2, 20, 0] = 0.0, -0.0, float("inf"), float("-inf")
        tgt[2, 30, 0], tgt[2, 31, 0] = float("nan"), -float("nan")
    tgt = tgt.cuda()
    sw = _ops.SweepIndex(tgt, sorted_rows=True)
    m_pad = sw.tperm.shape[1]
    key = torch.full((N, m_pad), float("nan"), dtype=dtype, device="cuda")
    key[:, :m] = tgt[:, :, 0]
    ref_keys, ref_order = torch.sort(key, dim=1, stable=True)
    assert torch.equal(sw.tperm.long()[:2], ref_order[:2])
    p64 = sw.tperm.long()
    assert torch.equal(torch.sort(p64, dim=1).values, torch.arange(m_pad, device="cuda").expand(N, -1))
    got = torch.gather(key, 1, p64)
    assert torch.equal(torch.nan_to_num(sw.keys, nan=7.0), torch.nan_to_num(got, nan=7.0))
    a_, b_ = sw.keys[:, :-1], sw.keys[:, 1:]
    assert bool(((a_ <= b_) | torch.isnan(b_)).all()) and bool((~torch.isnan(a_) | torch.isnan(b_)).all())
    tie = (a_ == b_) | (torch.isnan(a_) & torch.isnan(b_))
    assert bool((p64[:, :-1] < p64[:, 1:])[tie].all())                                # ties keep their index order
    assert bool((p64[:, :m] < m).all())                                               # sorted slots [0, m) are the real rows
    # and the search built on it is exact
    x = (torch.rand((N, 500, 3), generator=g, dtype=torch.float64) * 40 - 20).to(dtype).cuda()
    clean = tgt.clone()
    clean[2, :, 0] = torch.nan_to_num(clean[2, :, 0], nan=1.0, posinf=2.0, neginf=-2.0)
    swc = _ops.SweepIndex(clean)
    assert torch.equal(swc.knn(x, None, swc.query_order(x, None)), _ops.knn(x, None, _ops.pack_target(clean), m, _lib.KNN_VALU))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_pose_grad_in_out_kernels(dtype):
    """dicp_pose_grad_in / dicp_pose_grad_out (head and tail of ICPLoop.backward) against the tensor expressions
    they replace: gpose = [gT[:, :3, :3].ravel, gT[:, :3, 3]] in double; gT0 = the same layout from gpose + the summed
    pose slots of the last launch's partials, bottom row zero."""
    from dicp_amd import _ops, _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    N, nblk = 37, 5
    code = _lib.F32 if dtype == torch.float32 else _lib.F64
    gT = torch.randn((N, 4, 4), generator=g, dtype=dtype).to(DEV)
    gpose = torch.full((N, 12), float("nan"), dtype=torch.float64, device=DEV)
    st = _ops._stream()
    _lib.check(lib.dicp_pose_grad_in(code, _ops._p(gT), _ops._p(gpose), N, st), "in")
    ref = torch.cat((gT[:, :3, :3].reshape(N, 9), gT[:, :3, 3]), dim=1).to(torch.float64)
    assert torch.equal(gpose, ref)
    _lib.check(lib.dicp_pose_grad_in(code, None, _ops._p(gpose), N, st), "in")
    assert float(gpose.abs().max()) == 0.0
    gpose = torch.randn((N, 12), generator=g, dtype=torch.float64).to(DEV)
    part = torch.randn((N, nblk, _lib.NBWD_PAD), generator=g, dtype=dtype).to(DEV)
    for p_, nb in ((part, nblk), (None, 0)):
        out = torch.full((N, 4, 4), float("nan"), dtype=dtype, device=DEV)
        _lib.check(lib.dicp_pose_grad_out(code, _ops._p(gpose), _ops._p(p_), nb, _ops._p(out), N, st), "out")
        tot = gpose + (p_.to(torch.float64).sum(dim=1)[:, :12] if p_ is not None else 0.0)
        want = torch.zeros((N, 4, 4), dtype=torch.float64, device=DEV)
        want[:, :3, :3] = tot[:, :9].reshape(N, 3, 3)
        want[:, :3, 3] = tot[:, 9:]
        assert float((out.to(torch.float64) - want).abs().max()) <= (1e-6 if dtype == torch.float32 else 1e-14)
        assert float(out[:, 3].abs().max()) == 0.0


def test_nonfinite_target_x_keeps_its_sorted_slot_in_the_windowed_backward():
    """a target row whose x is +inf or NaN sorts above +max; pad slots must still come after it, so that its
    sorted position is < m and dicp_window_reduce writes its gradient row (the buffer is torch.empty)."""
    B, n = 3, 5000
    src, tgt = make_pairs(B, n, n, seed=3)
    tgt[2, 5, 0] = float("inf")
    tgt[0, 9, 0] = float("nan")
    junk = torch.full((B * n * 6 * 4,), 12345.0, device=DEV)      # what "uninitialised" would show next
    del junk
    T0 = torch.eye(4, device=DEV).repeat(B, 1, 1)
    grads = {}
    for knn in (_lib.KNN_VALU, _lib.KNN_SWEEP):
        icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=4, tolerance=1e-12)
        icp.const_iter = True
        icp.knn_variant = knn
        s, tg = src.to(DEV).requires_grad_(True), tgt.to(DEV).requires_grad_(True)
        out = icp.icp(s, tg, T0, trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0})
        out["T"].sum().backward()
        grads[knn] = (out["T"].detach().cpu(), tg.grad.detach().cpu())
        assert bool(torch.isfinite(out["T"][1]).all() and torch.isfinite(tg.grad[1]).all())
        assert not bool((tg.grad == 12345.0).any())
        for b, j in ((2, 5), (0, 9)):                                # never matched (or NaN): no stale memory either way
            row = tg.grad[b, j]
            assert bool(((row == 0) | ~torch.isfinite(row)).all()), (knn, b, j, row)
    np.testing.assert_allclose(grads[_lib.KNN_SWEEP][0][1].numpy(), grads[_lib.KNN_VALU][0][1].numpy(), rtol=0, atol=1e-6)
    np.testing.assert_allclose(grads[_lib.KNN_SWEEP][1][1].numpy(), grads[_lib.KNN_VALU][1][1].numpy(), rtol=0, atol=1e-5)


def test_icp_with_gumbel_in_tolerance_mode_and_without_injected_noise():
    """The Gumbel loop inside the library (DICP_KNN_GUMBEL) beyond the injected-noise parity case: tolerance mode (segments, frozen clouds, trimmed
    histories) and the in-kernel noise (seeded from torch's generator: same seed -> same call, and the backward regenerates what the forward drew)."""
    N, n, m = 3, 200, 230
    src, tgt = make_pairs(N, n, m, seed=52, dtype=torch.float64, max_rot=0.03, max_trans=0.1)
    kw = dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=3)

    def run(seed, const_iter):
        torch.manual_seed(seed)
        icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=12, tolerance=1e-14 if const_iter else 2e-2)
        icp.const_iter = const_iter
        icp.nn.use_gumbel, icp.nn.eps, icp.nn.tau = True, 1e-10, 0.05
        s, t = src.to(DEV).requires_grad_(True), tgt.to(DEV).requires_grad_(True)
        out = icp.icp(s, t, torch.eye(4, dtype=torch.float64, device=DEV).repeat(N, 1, 1), **kw)
        out["T"].sum().backward()
        return out, s.grad, t.grad

    a, b, c = run(5, True), run(5, True), run(6, True)
    assert torch.equal(a[0]["T"], b[0]["T"]) and torch.equal(a[1], b[1]) and float((a[2] - b[2]).abs().max()) <= 1e-12 * float(a[2].abs().max())
    assert not torch.equal(a[0]["T"], c[0]["T"])                        # another seed, another draw
    assert bool(torch.isfinite(a[1]).all()) and bool(torch.isfinite(a[2]).all()) and float(a[2].abs().max()) > 0
    d = run(5, False)
    K = d[0]["deltas"].shape[1]
    assert 1 <= K <= 12 and d[0]["weights"].shape[1] == K and d[0]["costs"].shape[1] == K
    assert bool(torch.isfinite(d[0]["T"]).all()) and bool(torch.isfinite(d[1]).all())
    assert torch.equal(d[0]["deltas"][:, 0], a[0]["deltas"][:, 0])     # (same seed: the first iteration is the constant-iteration call's; later ones freeze cloud by cloud)


@pytest.mark.gpu
def test_library_copy_and_zero_fill():
    """dicp_copy / dicp_zero (csrc/dicp_fill.h: the library's own fills and copies, kernels instead of the runtime's memset / memcpy): every size class, unaligned
    ends, and nothing outside the range touched."""
    import ctypes
    from dicp_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(9)
    for words in (1, 3, 4, 63, 64, 1000, 4096, 1 << 20, (1 << 22) + 5):
        src = torch.randint(-2 ** 31, 2 ** 31 - 1, (words + 8,), generator=g, dtype=torch.int64).to(torch.int32).cuda()
        for off in (0, 1, 3):          # 16-byte aligned or only 4-byte aligned
            dst = torch.full((words + 8,), 77, dtype=torch.int32, device="cuda")
            _lib.check(lib.dicp_copy(ctypes.c_void_p(dst.data_ptr() + 4 * (off + 1)), ctypes.c_void_p(src.data_ptr() + 4 * off), words * 4, None), "dicp_copy")
            assert torch.equal(dst[off + 1:off + 1 + words], src[off:off + words]) and bool((dst[:off + 1] == 77).all()) and bool((dst[off + 1 + words:] == 77).all())
            _lib.check(lib.dicp_zero(ctypes.c_void_p(dst.data_ptr() + 4 * (off + 1)), words * 4, None), "dicp_zero")
            assert bool((dst[off + 1:off + 1 + words] == 0).all()) and bool((dst[:off + 1] == 77).all()) and bool((dst[off + 1 + words:] == 77).all())
