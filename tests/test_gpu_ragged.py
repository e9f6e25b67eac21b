"""f-2: ragged batches without pad rows.  The reference pads every cloud of a list to the longest one
(zero source rows of weight 0, target rows that are copies of one far point: ICP.py:305-511) and then computes on the pads;
here the kernels are handed the clouds' own lengths (`src_rows` / `tgt_rows`) and never read, score or accumulate a row
beyond them, while the API-visible tensors keep the reference's padded shapes.  `-m gpu`."""
import numpy as np
import pytest
import torch

from dicp_amd import _lib, _ops
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs

pytestmark = pytest.mark.gpu
DEV = "cuda"


def npy(x):
    return x.detach().cpu().numpy()


def rows(v):
    return torch.tensor(v, dtype=torch.int32, device=DEV)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_search_kernels_with_row_counts_never_touch_a_pad(dtype):
    """Every search form with per-cloud row counts == the dense call on the cropped cloud; pad rows hold NaN (any read of
    one would poison the result) and the index slots of pad queries are left alone."""
    N, n, m = 5, 3000, 3500
    src, tgt = make_pairs(N, n, m, seed=31, dtype=dtype)
    src, tgt = src.to(DEV), tgt.to(DEV)
    ns, ms = [3000, 1, 777, 2048, 65], [3500, 900, 1, 64, 3000]
    for b in range(N):
        src[b, ns[b]:] = float("nan")
        tgt[b, ms[b]:] = float("nan")
    sr, tr = rows(ns), rows(ms)
    ang = 0.03
    pose = torch.tensor([np.cos(ang), -np.sin(ang), 0, np.sin(ang), np.cos(ang), 0, 0, 0, 1, 0.1, -0.2, 0.05], dtype=dtype, device=DEV).repeat(N, 1)
    want = []
    for b in range(N):
        sb, tb = src[b:b + 1, :ns[b]].contiguous(), tgt[b:b + 1, :ms[b]].contiguous()
        want.append(_ops.knn(sb, pose[b:b + 1], _ops.pack_target(tb), ms[b], _lib.KNN_VALU)[0])

    def check(idx, what):
        for b in range(N):
            assert torch.equal(idx[b, :ns[b]], want[b]), (what, b)
            assert bool((idx[b, ns[b]:] == -7).all()), (what, b, "pad queries must not be written")

    tgt4 = _ops.pack_target(tgt, None, tr)
    variants = [_lib.KNN_VALU | (c << 8) for c in (0, 1, 2, 3, 5, 11)] + ([_lib.KNN_MFMA, _lib.KNN_MFMA | (5 << 8)] if dtype == torch.float32 else [])
    for v in variants:
        idx = torch.full((N, n), -7, dtype=torch.int32, device=DEV)
        check(_ops.knn(src, pose, tgt4, m, v, out=idx, src_rows=sr, tgt_rows=tr), ("brute", v))
    sw = _ops.SweepIndex(tgt, sorted_rows=True, tgt_rows=tr)
    for b in range(N):                                       # sorted slots [0, m_b) are exactly the cloud's own rows
        assert bool((sw.tperm[b, :ms[b]] < ms[b]).all()) and not bool(torch.isnan(sw.tgs4[b, :ms[b]]).any())
        assert not bool(torch.isnan(sw.tgt_s[b, :ms[b]]).any())
    qo = sw.query_order(src, pose, src_rows=sr)
    assert torch.equal(torch.sort(qo.long(), dim=1).values, torch.arange(n, device=DEV).repeat(N, 1))      # always a permutation
    for b in range(N):
        assert bool((qo[b, :ns[b]] < ns[b]).all()) and torch.equal(qo[b, ns[b]:].long(), torch.arange(ns[b], n, device=DEV))
    for cfg in (0, 1, 2, 4):
        for order in (qo, None):
            idx = torch.full((N, n), -7, dtype=torch.int32, device=DEV)
            spos = torch.full((N, n), -7, dtype=torch.int32, device=DEV)
            check(sw.knn(src, pose, order, out=idx, cfg=cfg, spos=spos, src_rows=sr), ("sweep", cfg))
            for b in range(N):
                sp = spos[b, :ns[b]].long()
                assert torch.equal(sw.tperm[b][sp], want[b]) and bool((spos[b, ns[b]:] == -7).all())
    # the centre is taken from the cloud's own rows
    ctr = _ops.search_frame(tgt, quantum=0.0, tgt_rows=tr)
    for b in range(N):
        assert torch.equal(ctr[b], _ops.search_frame(tgt[b:b + 1, :ms[b]].contiguous(), quantum=0.0)[0])


@pytest.mark.parametrize("dtype,knn", [(torch.float64, _lib.KNN_SWEEP), (torch.float64, _lib.KNN_VALU), (torch.float32, _lib.KNN_SWEEP)])
def test_ragged_icp_equals_per_item_calls(dtype, knn):
    """A list of clouds of very different sizes through the whole call (sweep path with the windowed backward, and the
    brute-force path): every result of the batch == the per-item call of that pair, gradients included; the pad rows of
    the returned tensors are what the reference's padding yields (weight 0, pc of the zero point, zero gradient)."""
    lens_s, lens_t = [2600, 900, 1500, 64], [2500, 1100, 2600, 300]
    N, K = len(lens_s), 5
    src, tgt = make_pairs(N, 2600, 2600, seed=41, dtype=dtype)
    S = [src[b, :lens_s[b]].to(DEV).requires_grad_(True) for b in range(N)]
    Tg = [tgt[b, :lens_t[b]].to(DEV).requires_grad_(True) for b in range(N)]
    W = [(torch.rand(lens_s[b], generator=torch.Generator().manual_seed(b), dtype=torch.float64).to(dtype) * 0.5 + 0.5).to(DEV).requires_grad_(True) for b in range(N)]
    T0 = [torch.eye(4, dtype=dtype, device=DEV) for _ in range(N)]
    kw = dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=3)
    icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=1e-12)
    icp.const_iter = True
    icp.knn_variant = knn
    out = icp.icp(S, Tg, T0, weight=W, **kw)
    # (the loss leaves the pad rows of pc out: they are the transformed zero point, which a per-item call does not have)
    (out["T"].sum() + sum((out["pc"][b, :lens_s[b]] ** 2).sum() for b in range(N))).backward()
    assert out["pc"].shape == (N, 2600, 3) and out["weights"].shape == (N, K, 2600, 1)
    f64 = dtype == torch.float64
    for b in range(N):
        s1, t1, w1 = (x.detach().clone().requires_grad_(True) for x in (S[b], Tg[b], W[b]))
        one = icp.icp(s1, t1, T0[b], weight=w1, **kw)
        (one["T"].sum() + (one["pc"] ** 2).sum()).backward()
        tol = 1e-10 if f64 else 2e-5
        np.testing.assert_allclose(npy(out["T"])[b], npy(one["T"])[0], rtol=0, atol=tol)
        np.testing.assert_allclose(npy(out["deltas"])[b], npy(one["deltas"])[0], rtol=0, atol=tol)
        np.testing.assert_allclose(npy(out["costs"])[b], npy(one["costs"])[0], rtol=1e-6 if f64 else 1e-3, atol=tol)
        np.testing.assert_allclose(npy(out["weights"])[b, :, :lens_s[b]], npy(one["weights"])[0], rtol=0, atol=10 * tol)
        np.testing.assert_allclose(npy(out["pc"])[b, :lens_s[b]], npy(one["pc"])[0], rtol=0, atol=10 * tol)
        gtol = 1e-8 if f64 else 2e-3
        for got, want, nm in ((S[b].grad, s1.grad, "source"), (Tg[b].grad, t1.grad, "target"), (W[b].grad, w1.grad, "weight")):
            scale = max(1.0, float(want.abs().max()))
            np.testing.assert_allclose(npy(got), npy(want), rtol=0, atol=gtol * scale, err_msg="%s grad of cloud %d" % (nm, b))
        # pads: weight 0 in every iteration, pc = the transformed zero point (ICP.py:274 on the zero rows)
        assert float(out["weights"][b, :, lens_s[b]:].abs().max()) == 0.0 if lens_s[b] < 2600 else True
        if lens_s[b] < 2600:
            np.testing.assert_allclose(npy(out["pc"])[b, lens_s[b]:], np.broadcast_to(npy(out["T"])[b, :3, 3], (2600 - lens_s[b], 3)), rtol=0, atol=tol)
        np.testing.assert_allclose(npy(out["stats"]["matched_ratio"])[b], npy(one["stats"]["matched_ratio"])[0], rtol=0, atol=1e-6)


def test_ragged_batch_scores_only_real_pairs():
    """Per real pair a ragged batch is searched no harder than a dense one: the pairs scored stay a small fraction of the
    REAL n_b x m_b products (the padded products are 2.3x that here)."""
    B, n, K = 16, 16384, 4
    src, tgt = make_pairs(B, n, n, seed=3)
    g = torch.Generator().manual_seed(0)
    ls = torch.randint(5000, n + 1, (B,), generator=g).tolist()
    lt = [max(4000, v - 3000) for v in ls]
    S = [src[b, :ls[b]].to(DEV) for b in range(B)]
    Tg = [tgt[b, :lt[b]].to(DEV) for b in range(B)]
    icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=1e-12)
    icp.const_iter = True
    icp.knn_variant = _lib.KNN_SWEEP
    out = icp.icp(S, Tg, [torch.eye(4, device=DEV)] * B, trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0})
    assert bool(torch.isfinite(out["T"]).all())
    real = float(sum(a * b for a, b in zip(ls, lt)))
    frac = float(icp.knn_stats["knn_pairs"].sum().item()) / (real * K)
    assert frac < 0.12, frac          # (dense benchmark clouds: ~4 % over the first four iterations; short clouds have wider tiles relative to m)


def test_lists_are_packed_by_one_launch_like_pad_sequence():
    """ICP.py:305-511 pads a list of clouds with one op per cloud; device lists go through dicp_pack_list / dicp_unpack_list (one launch each way): the same padded
    batch as torch's pad_sequence + the reference's pad value, bit for bit, and the same gradients back to every cloud of the list."""
    import torch
    from dicp_amd import _ops
    g = torch.Generator().manual_seed(5)
    for dt in (torch.float32, torch.float64):
        clouds = [torch.randn((int(torch.randint(1, 700, (1,), generator=g)), 6), generator=g, dtype=dt).cuda().requires_grad_(True) for _ in range(37)]
        pad = torch.tensor(1234.5, dtype=dt, device="cuda")
        for cols in (3, 6):
            out = _ops.pack_list(clouds, cols, pad)
            ref = torch.nn.utils.rnn.pad_sequence([c[:, :cols] for c in clouds], batch_first=True)
            lens = torch.tensor([c.shape[0] for c in clouds], device="cuda")
            ref = torch.where((torch.arange(ref.shape[1], device="cuda")[None, :] < lens[:, None]).unsqueeze(-1), ref, pad)
            assert torch.equal(out, ref)
            wgt = torch.randn(out.shape, generator=g, dtype=dt).cuda()
            ga = torch.autograd.grad((out * wgt).sum(), clouds)
            gb = torch.autograd.grad((ref * wgt).sum(), clouds)
            assert all(torch.equal(a, b) for a, b in zip(ga, gb))
        assert torch.equal(_ops.pack_list(clouds, 3), torch.nn.utils.rnn.pad_sequence([c[:, :3] for c in clouds], batch_first=True))
    # a strided view (the first three columns of six-column clouds, every other row) is taken as it stands or refused, never misread
    views = [c.detach()[::2] for c in clouds]
    assert _ops.packable(views, (6,)) and torch.equal(_ops.pack_list(views, 6), torch.nn.utils.rnn.pad_sequence(views, batch_first=True))
    # the launch reads the clouds through raw pointers: a cloud of another dtype, too few columns or a strided column is refused by pack_list itself
    base = torch.randn((10, 6), generator=g, dtype=torch.float32).cuda()
    for bad in ([base, base.double()], [base[:, :2]], [base.t().contiguous().t()], [base, base.cpu()], [base.reshape(2, 5, 6)]):
        with pytest.raises(ValueError):
            _ops.pack_list(bad, 3)
    assert not _ops.packable([c.detach().t() for c in clouds], (6,)) and not _ops.packable([c.detach().cpu() for c in clouds], (6,))


@pytest.mark.gpu
@pytest.mark.parametrize("icp_type,const_iter", [("pt2pl", True), ("pt2pt", False)])
def test_padded_batch_with_row_counts_equals_the_list_call(icp_type, const_iter):
    """ICP.icp(..., source_rows=, target_rows=) (build-specific): ragged clouds as ONE padded batch with per-cloud row counts -- the same poses, per-iteration
    outputs and gradients (on the clouds' own rows; zero on the pads) as the list call the reference's API offers for them (ICP.py:305-511), to rounding: the
    sums are grouped by another query order (the list's target keeps one far pad row).  Whatever the pad rows hold."""
    N, n, m, K = 12, 6000, 7000, 8
    src, tgt = make_pairs(N, n, m, seed=23, dtype=torch.float32)
    if icp_type == "pt2pt":
        tgt = tgt[:, :, :3].contiguous()
    g = torch.Generator().manual_seed(4)
    ls = [int(v) for v in torch.randint(n // 2, n + 1, (N,), generator=g)]
    lt = [int(v) for v in torch.randint(m // 2, m + 1, (N,), generator=g)]
    ls[0], lt[1] = n, m
    kw = dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0})

    def new():
        icp = ICP(icp_type=icp_type, differentiable=True, max_iterations=K, tolerance=1e-12 if const_iter else 1e-5)
        icp.const_iter = const_iter
        return icp
    S = [src[b, :ls[b]].cuda().requires_grad_(True) for b in range(N)]
    T = [tgt[b, :lt[b]].cuda().requires_grad_(True) for b in range(N)]
    out_l = new().icp(S, T, [torch.eye(4, device="cuda")] * N, **kw)
    mask = (torch.arange(n, device="cuda")[None, :] < torch.tensor(ls, device="cuda")[:, None])        # (the loss looks at the clouds' own rows of pc only)
    (out_l["T"][:, :3].sum() + 1e-3 * ((out_l["pc"] * mask.unsqueeze(-1)) ** 2).sum()).backward()
    # the padded batch: garbage behind every cloud's own rows
    sp, tp = src.clone(), tgt.clone()
    for b in range(N):
        sp[b, ls[b]:] = 1e3 * torch.randn((n - ls[b], 3), generator=g)
        tp[b, lt[b]:] = float("nan")
    sp, tp = sp.cuda().requires_grad_(True), tp.cuda().requires_grad_(True)
    for rows_as in ("list", "device"):
        sp.grad = tp.grad = None
        sr = ls if rows_as == "list" else torch.tensor(ls, dtype=torch.int32, device="cuda")
        tr = lt if rows_as == "list" else torch.tensor(lt, dtype=torch.int64, device="cuda")
        out_p = new().icp(sp, tp, torch.eye(4, device="cuda").repeat(N, 1, 1), source_rows=sr, target_rows=tr, **kw)
        (out_p["T"][:, :3].sum() + 1e-3 * ((out_p["pc"] * mask.unsqueeze(-1)) ** 2).sum()).backward()
        assert out_p["deltas"].shape == out_l["deltas"].shape
        assert float((out_p["T"] - out_l["T"]).abs().max()) <= 2e-5
        assert float((out_p["deltas"] - out_l["deltas"]).abs().max()) <= 2e-5
        for b in range(N):
            for got, want in ((sp.grad[b, :ls[b]], S[b].grad), (tp.grad[b, :lt[b]], T[b].grad)):
                assert float((got - want).abs().max()) <= 2e-4 * max(1e-6, float(want.abs().max())), (b, rows_as)
            assert bool((sp.grad[b, ls[b]:] == 0).all()) and bool((torch.nan_to_num(tp.grad[b, lt[b]:]) == 0).all())
    with pytest.raises(ValueError):
        new().icp(sp, tp, torch.eye(4, device="cuda").repeat(N, 1, 1), source_rows=ls[:-1], **kw)
    with pytest.raises(ValueError):
        new().icp(sp, tp, torch.eye(4, device="cuda").repeat(N, 1, 1), source_rows=[0] + ls[1:], **kw)
    with pytest.raises(ValueError):
        new().icp(S, T, [torch.eye(4, device="cuda")] * N, source_rows=ls, **kw)
