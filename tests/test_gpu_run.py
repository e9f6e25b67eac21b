"""Runs of iterations inside one launch (`-m gpu`): dicp_icp_backward_run keeps every slot's point, match row and accumulating
gradients on chip over the iterations after the last query re-ordering and chains accumulate_bwd(k) -> step_bwd(k-1) inside the
launch.  It is the same arithmetic per point as the per-iteration windowed launches (the reverse of ICP.py:132-260), grouped
differently: gradients must agree with the per-iteration path to rounding, with every input's gradient, in every mode."""
import numpy as np
import pytest
import torch

from dicp_amd import _lib
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs, make_scene_pairs
from oracle import dicp_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
KW = dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=3)


def npy(x):
    return x.detach().cpu().numpy()


def run_both(src, tgt, K, icp_type="pt2pl", kw=KW, weight=None, T0=None, ragged=None, diff=True, run_from=None, const_iter=True, tol=1e-12, loss_pc=False):
    N = src.shape[0]
    outs = {}
    for use_run in (False, True):
        icp = ICP(icp_type=icp_type, differentiable=diff, max_iterations=K, tolerance=tol)
        icp.const_iter, icp.bwd_run, icp.knn_variant, icp.bwd_run_from = const_iter, use_run, _lib.KNN_SWEEP, run_from
        if ragged is not None:
            S = [src[b, :ragged[b]].to(DEV).requires_grad_(True) for b in range(N)]
            Tg = [tgt[b, :max(2048, ragged[b] - 300)].to(DEV).requires_grad_(True) for b in range(N)]
            Ti = [torch.eye(4, dtype=src.dtype, device=DEV)] * N
            W = None
        else:
            S, Tg = src.to(DEV).requires_grad_(True), tgt.to(DEV).requires_grad_(True)
            Ti = (torch.eye(4, dtype=src.dtype).repeat(N, 1, 1) if T0 is None else T0).to(DEV).requires_grad_(True)
            W = weight.to(DEV).requires_grad_(True) if weight is not None else None
        out = icp.icp(S, Tg, Ti, weight=W, **kw)
        loss = out["T"].sum() + ((out["pc"] ** 2).sum() * 1e-3 if loss_pc else 0.0)
        loss.backward()
        grads = [torch.cat([x.grad.reshape(-1) for x in (S if ragged is not None else [S])]),
                 torch.cat([x.grad.reshape(-1) for x in (Tg if ragged is not None else [Tg])])]
        if ragged is None:
            grads.append(Ti.grad.reshape(-1))
            if W is not None:
                grads.append(W.grad.reshape(-1))
        outs[use_run] = (out, grads, dict(icp.knn_stats))
    return outs


def check(outs, rtol):
    a, b = outs[False], outs[True]
    assert "bwd_run" not in a[2] and "bwd_run" in b[2], "the run path was not taken"
    for key in ("T", "deltas", "weights", "costs", "pc"):
        assert torch.equal(a[0][key], b[0][key]), key          # (the forward is the same code either way)
    for ga, gb in zip(a[1], b[1]):
        assert bool(torch.isfinite(gb).all())
        np.testing.assert_allclose(npy(gb), npy(ga), rtol=0, atol=rtol * max(1.0, float(ga.abs().max())))
    return b[2]["bwd_run"]


@pytest.mark.parametrize("N,n,K,icp_type", [(40, 16384, 10, "pt2pl"), (72, 16384, 9, "pt2pl"), (33, 4096, 12, "pt2pt"), (8, 8192, 7, "pt2pl"), (130, 2048, 8, "pt2pt")])
def test_backward_run_equals_per_iteration_launches(N, n, K, icp_type):
    """Dense batches: fewer clouds than one launch holds, more than one launch holds (72 and 130: several launches, the last one partial),
    cloud counts that are no multiple of 8, both ICP types."""
    src, tgt = make_pairs(N, n, n, seed=41)
    if icp_type == "pt2pt":
        tgt = tgt[:, :, :3].contiguous()
    lo, hi = check(run_both(src, tgt, K, icp_type), 2e-6)
    assert hi == K and lo == 4


@pytest.mark.parametrize("icp_type,dim,loss,diff", [("pt2pl", 3, {"name": "cauchy", "metric": 0.5}, True), ("pt2pt", 2, {"name": "huber", "metric": 0.3}, True),
                                                     ("pt2pl", 2, {"name": "trim", "metric": 0.8}, True), ("pt2pl", 3, {"name": "huber", "metric": 0.5}, False),
                                                     ("pt2pt", 3, None, True)])
def test_backward_run_every_input_gradient(icp_type, dim, loss, diff):
    """A weight tensor and a T_init that take gradients, a loss on the transformed cloud as well as on the pose, the other losses / planar mode / hard weights."""
    N, n, K = 12, 4096, 9
    src, tgt = make_pairs(N, n, n, seed=43)
    if icp_type == "pt2pt":
        tgt = tgt[:, :, :3].contiguous()
    g = torch.Generator().manual_seed(5)
    w0 = 0.5 + 0.5 * torch.rand((N, n), generator=g)
    T0 = torch.eye(4).repeat(N, 1, 1)
    T0[:, :3, 3] = 0.02 * torch.randn((N, 3), generator=g)
    outs = run_both(src, tgt, K, icp_type, kw=dict(trim_dist=5.0, loss_fn=loss, dim=dim), weight=w0, T0=T0, diff=diff, loss_pc=True)
    a, b = outs[False], outs[True]
    assert "bwd_run" in b[2]
    for ga, gb in zip(a[1], b[1]):
        tol = 2e-5 * max(1.0, float(ga[torch.isfinite(ga)].abs().max()))
        assert bool((((ga - gb).abs() <= tol) | (torch.isnan(ga) & torch.isnan(gb))).all())


def test_backward_run_ragged_and_tolerance_mode():
    """Ragged lists (pad slots, the clouds' own lengths) and a tolerance-mode call (the run ends where the loop stopped)."""
    N, n = 24, 8192
    src, tgt = make_pairs(N, n, n, seed=47)
    lens = [n - (977 * b) % (n // 3) for b in range(N)]
    check(run_both(src, tgt, 9, ragged=lens), 2e-6)
    outs = run_both(src, tgt, 30, const_iter=False, tol=1e-6, run_from=2)      # (converges after ~6 iterations: the run starts early enough to exist)
    lo, hi = check(outs, 2e-6)
    assert lo == 2 and hi == outs[True][0]["deltas"].shape[1] < 30


def test_backward_run_when_matches_keep_changing():
    """A run that starts right after the first iteration on clouds far from their pose: matches change from iteration to iteration for many
    slots (each change reloads the row and hands the accumulated target-row gradient to the side buffer).  Same gradients."""
    N, n, K = 20, 8192, 8
    src, tgt = make_pairs(N, n, n, seed=49, max_rot=0.2, max_trans=1.0)
    lo, hi = check(run_both(src, tgt, K, run_from=1), 5e-6)
    assert (lo, hi) == (1, K)
    s2, t2 = make_scene_pairs(N, n, n, seed=51)
    check(run_both(s2, t2, K, run_from=2), 5e-6)


def test_backward_run_against_the_oracle():
    """The whole call with the run in its backward against the CPU oracle (2 clouds of a 40-cloud batch, 10 iterations)."""
    N, n, K = 40, 16384, 10
    src, tgt = make_pairs(N, n, n, seed=53)
    icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=1e-12)
    icp.const_iter = True
    S, Tg = src.to(DEV).requires_grad_(True), tgt.to(DEV).requires_grad_(True)
    out = icp.icp(S, Tg, torch.eye(4, device=DEV).repeat(N, 1, 1), **KW)
    out["T"].sum().backward()
    assert icp.knn_stats.get("bwd_run") == (4, K)
    s, t = src[30:32].clone().requires_grad_(True), tgt[30:32].clone().requires_grad_(True)
    ref = O.icp_batched(s, t, torch.eye(4).repeat(2, 1, 1), torch.ones(2, n), icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=1e-12,
                        const_iter=True, tanh_steepness=5.0, **KW)
    ref["T"].sum().backward()
    np.testing.assert_allclose(npy(out["T"][30:32]), ref["T"].detach().numpy(), rtol=0, atol=1e-4)
    for got, want in ((S.grad[30:32].cpu(), s.grad), (Tg.grad[30:32].cpu(), t.grad)):
        scale = max(1.0, float(want.abs().max()))
        err = (got - want).abs().amax(dim=2)
        assert float((err > 1e-3 * scale).float().mean()) < 1e-3
        assert float(err.median()) < 1e-5 * scale
