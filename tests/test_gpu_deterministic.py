"""ICP.deterministic (`-m gpu`): the same gradient bits on every run.  The reference's backward is autograd over deterministic CPU ops
(/root/reference/dICP/ICP.py:132-260 differentiated by torch); the default GPU path sums a target row's contributions in the order the waves
reach it and adds the out-of-window ones with float atomics -- equal to rounding, not bit for bit (profiles/r04_soak.txt: 2.6e-4 of the largest
gradient between two runs of one call).  With the switch: bit-identical runs, the same results as the default path to rounding, and the oracle's."""
import numpy as np
import pytest
import torch

from dicp_amd import _lib
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs, make_independent_pairs
from oracle import dicp_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
KW = dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=3)


def run(src, tgt, K, deterministic, icp_type="pt2pl", weight=None, lists=None, kw=KW, const_iter=True, tol=1e-12):
    icp = ICP(icp_type=icp_type, differentiable=True, max_iterations=K, tolerance=tol)
    icp.const_iter, icp.deterministic = const_iter, deterministic
    if lists is not None:           # (src, tgt: lists of clouds of their own lengths)
        N, dtype = len(src), src[0].dtype
        S = [x.to(DEV).requires_grad_(True) for x in src]
        Tg = [x.to(DEV).requires_grad_(True) for x in tgt]
        Ti, W = [torch.eye(4, dtype=dtype, device=DEV)] * N, None
    else:
        N, dtype = src.shape[0], src.dtype
        if icp_type == "pt2pt":
            tgt = tgt[:, :, :3]
        S, Tg = src.to(DEV).requires_grad_(True), tgt.to(DEV).requires_grad_(True)
        Ti = torch.eye(4, dtype=dtype).repeat(N, 1, 1).to(DEV).requires_grad_(True)
        W = weight.to(DEV).requires_grad_(True) if weight is not None else None
    out = icp.icp(S, Tg, Ti, weight=W, **kw)
    (out["T"][:, :3].sum() + 0.1 * out["pc"].sum()).backward()
    torch.cuda.synchronize()
    leaves = (S if lists is not None else [S]) + (Tg if lists is not None else [Tg]) + ([] if lists is not None else [Ti]) + ([W] if W is not None else [])
    return out, [x.grad.detach().clone() for x in leaves]


def identical(a, b):
    return all(torch.equal(x, y) or bool((torch.isnan(x) == torch.isnan(y)).all() and torch.equal(torch.nan_to_num(x), torch.nan_to_num(y))) for x, y in zip(a, b))


def close(a, b, rtol):
    for x, y in zip(a, b):
        scale = max(1.0, float(x.abs().max()))
        assert float((x - y).abs().max()) <= rtol * scale, float((x - y).abs().max()) / scale


@pytest.mark.parametrize("offset", [0.0, 600.0])
def test_two_runs_are_bit_identical_float32(offset):
    """The soak's case: float32, clouds far from the origin (out-of-window rows and long lists are common there)."""
    src, tgt = make_pairs(24, 16384, 16384, seed=31, dtype=torch.float32)
    src, tgt = src.clone(), tgt.clone()
    src[:, :, :3] += offset
    tgt[:, :, :3] += offset
    o1, g1 = run(src, tgt, 8, True)
    o2, g2 = run(src, tgt, 8, True)
    assert identical(g1, g2)
    assert torch.equal(o1["T"], o2["T"])
    od, gd = run(src, tgt, 8, False)
    assert torch.equal(o1["T"], od["T"])                  # the forward does not depend on the switch
    close(g1, gd, 1e-3)


def test_hard_inputs_ragged_lists_bit_identical():
    """Independently sampled, partially overlapping clouds with metre-sized start poses, as ragged lists (test_ICP_inputs.py:36-103's shape of input)."""
    src, tgt = make_independent_pairs(6, 6000, 6000, seed=5, dtype=torch.float32)      # ragged: lists
    assert len({x.shape[0] for x in src}) > 1
    _, g1 = run(src, tgt, 6, True, lists=True)
    _, g2 = run(src, tgt, 6, True, lists=True)
    assert identical(g1, g2)
    _, gd = run(src, tgt, 6, False, lists=True)
    close(g1, gd, 2e-3)


@pytest.mark.parametrize("icp_type", ["pt2pl", "pt2pt"])
def test_float64_weights_against_default_and_oracle(icp_type):
    src, tgt = make_pairs(3, 2048, 2300, seed=17, dtype=torch.float64)
    w = torch.rand(3, 2048, dtype=torch.float64) * 0.5 + 0.5
    o1, g1 = run(src, tgt, 5, True, icp_type=icp_type, weight=w)
    o2, g2 = run(src, tgt, 5, True, icp_type=icp_type, weight=w)
    assert identical(g1, g2)
    _, gd = run(src, tgt, 5, False, icp_type=icp_type, weight=w)
    close(g1, gd, 1e-10)
    s_c, t_c = src.clone().requires_grad_(True), (tgt if icp_type == "pt2pl" else tgt[:, :, :3]).clone().requires_grad_(True)
    T_c, w_c = torch.eye(4, dtype=torch.float64).repeat(3, 1, 1).requires_grad_(True), w.clone().requires_grad_(True)
    ref = O.icp_batched(s_c, t_c, T_c, w_c if icp_type == "pt2pl" else w_c.repeat_interleave(3, dim=1), icp_type=icp_type, differentiable=True, max_iterations=5, tolerance=1e-12, const_iter=True, tanh_steepness=5.0, **KW)
    (ref["T"][:, :3].sum() + 0.1 * ref["pc"].sum()).backward()
    np.testing.assert_allclose(o1["T"].detach().cpu().numpy(), ref["T"].detach().numpy(), atol=1e-10)
    for g, r in zip(g1, (s_c.grad, t_c.grad, T_c.grad, w_c.grad)):
        np.testing.assert_allclose(g.cpu().numpy(), r.numpy(), atol=1e-8 * max(1.0, float(r.abs().max())))


def test_tolerance_mode_bit_identical():
    src, tgt = make_pairs(8, 8192, 8192, seed=3, dtype=torch.float32)
    o1, g1 = run(src, tgt, 30, True, const_iter=False, tol=1e-4)
    o2, g2 = run(src, tgt, 30, True, const_iter=False, tol=1e-4)
    assert o1["deltas"].shape == o2["deltas"].shape and identical(g1, g2)


def test_not_with_gumbel():
    icp = ICP(icp_type="pt2pt", differentiable=True, max_iterations=2, tolerance=1e-12)
    icp.nn.use_gumbel = True
    icp.deterministic = True
    src, tgt = make_pairs(1, 256, 256, seed=1, dtype=torch.float64)
    with pytest.raises(NotImplementedError):
        icp.icp(src.to(DEV).requires_grad_(True), tgt[:, :, :3].to(DEV), torch.eye(4, dtype=torch.float64, device=DEV).unsqueeze(0))


def test_unsupported_forms_are_refused_by_the_call_not_by_backward():
    """A caller who set ICP.deterministic hears about a form it does not cover from icp() itself -- before anything is enqueued, not from loss.backward() after
    the whole forward has run."""
    src, tgt = make_pairs(2, 4096, 4096, seed=2, dtype=torch.float32)
    icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=3, tolerance=1e-12)
    icp.const_iter, icp.deterministic = True, True
    T0 = torch.eye(4, device=DEV).repeat(2, 1, 1)
    import dicp_amd._loop as loop
    cfg_seen = {}
    real = loop.ICPLoop.apply

    def no_window(*a):       # (the public switches route a deterministic call to the supported form: take the window away behind them)
        cfg = [x for x in a if isinstance(x, loop.LoopConfig)][0]
        cfg.bwd_window = False
        cfg_seen["cfg"] = cfg
        return real(*a)
    loop.ICPLoop.apply = no_window
    try:
        with pytest.raises(NotImplementedError):
            icp.icp(src.to(DEV).requires_grad_(True), tgt.to(DEV), T0, trim_dist=5.0)
    finally:
        loop.ICPLoop.apply = real
    assert cfg_seen["cfg"].deterministic


def test_the_reference_pair_takes_the_deterministic_path_too():
    """tests/data's 65-point pair (test_ICP.py:35-117: float64, point-to-plane, dim 2): forced onto the sweep search + windowed backward, same results as the default
    (small-cloud) path to rounding, bit-identical between runs."""
    import os
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    scan = torch.from_numpy(np.load(os.path.join(g, "points_scan.npy"))).to(torch.float64)
    mp = torch.from_numpy(np.load(os.path.join(g, "points_map.npy"))).to(torch.float64)
    kw = dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 10.0}, dim=2)
    outs = []
    for det in (True, True, False):
        icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=6, tolerance=1e-12)
        icp.const_iter, icp.deterministic = True, det
        s, t = scan.to(DEV).requires_grad_(True), mp.to(DEV).requires_grad_(True)
        o = icp.icp(s, t, torch.eye(4, dtype=torch.float64, device=DEV), **kw)
        o["T"].sum().backward()
        outs.append((o["T"].detach().clone(), s.grad.clone(), t.grad.clone()))
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))
    for a, b in zip(outs[0], outs[2]):
        assert float((a - b).abs().max()) <= 1e-10 * max(1.0, float(b.abs().max()))


@pytest.mark.parametrize("dtype,N,n", [(torch.float32, 3, 5000), (torch.float32, 2, 16384), (torch.float32, 2, 20011), (torch.float64, 3, 3001)])
def test_match_order_is_the_stable_sort_of_the_matches(dtype, N, n):
    """dicp_match_order (the deterministic backward's slot order, round 6: the library's own stable radix sort in place of torch.argsort): the queries in
    ascending order of their reference match, equal matches in index order, queries without a match and a cloud's rows past its own last."""
    import ctypes
    g = torch.Generator().manual_seed(n)
    spos = torch.randint(0, n // 3, (N, n), generator=g, dtype=torch.int32)          # (many queries share a match: stability is visible)
    spos[:, ::97] = -1
    rows = torch.tensor([n - 7 * b - 1 for b in range(N)], dtype=torch.int32)
    lib = _lib.load()
    code = _lib.F32 if dtype == torch.float32 else _lib.F64
    for src_rows in (None, rows.to(DEV)):
        sd = spos.to(DEV)
        qo = torch.full((N, n), -5, dtype=torch.int32, device=DEV)
        nbytes = int(lib.dicp_match_order_scratch_bytes(code, N, n))
        scratch = torch.empty((nbytes,), dtype=torch.uint8, device=DEV)
        rc = lib.dicp_match_order(code, ctypes.c_void_p(sd.data_ptr()), ctypes.c_void_p(src_rows.data_ptr()) if src_rows is not None else None, N, n,
                                  ctypes.c_void_p(scratch.data_ptr()), nbytes, ctypes.c_void_p(qo.data_ptr()), None)
        assert rc == 0
        torch.cuda.synchronize()
        key = spos.to(torch.int64)
        key[key < 0] = 2 ** 40
        if src_rows is not None:
            key = torch.where(torch.arange(n)[None, :] < rows[:, None].to(torch.int64), key, torch.full_like(key, 2 ** 40))
        want = torch.argsort(key, dim=1, stable=True).to(torch.int32)
        assert torch.equal(qo.cpu(), want), (dtype, N, n, src_rows is not None)
