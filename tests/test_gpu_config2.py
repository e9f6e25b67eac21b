"""BASELINE configs[1] at its EXACT workload (`-m gpu`): B = 32 synthetic 4096-point clouds, point-to-point + Huber(1) + trim(5), K = 10 constant
iterations, float32, forward + backward of T.sum() -- the call bench.py's `value_c2` leg times, with the Gauss-Newton step and with the closed-form SVD step --
against the reference's op sequence in float64 (oracle/dicp_oracle.py; for the SVD step: numpy Kabsch iterations and torch autograd through the final solve,
SURVEY.md 8a-12) at north_star's bars: poses within 1e-4, gradients within 1e-3 of their scale."""
import numpy as np
import pytest
import torch

from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs
from oracle import dicp_oracle as O

DEV = "cuda"
pytestmark = pytest.mark.gpu
N, n, K = 32, 4096, 10
TRIM, LOSS = 5.0, {"name": "huber", "metric": 1.0}
SLICE = slice(0, N, 4)          # the clouds the CPU restatement is run for (8 of the 32: it takes seconds per cloud)


def npy(t):
    return t.detach().cpu().numpy()


def workload():
    src, tgt = make_pairs(N, n, n, seed=2, dtype=torch.float32)                 # bench.py other_configs: seed 2
    return src, tgt[:, :, :3].contiguous()


def test_gauss_newton_step_against_the_oracle():
    src, tg = workload()
    icp = ICP(icp_type="pt2pt", differentiable=True, max_iterations=K, tolerance=1e-12)
    icp.const_iter = True
    s, t = src.to(DEV).requires_grad_(True), tg.to(DEV).requires_grad_(True)
    out = icp.icp(s, t, torch.eye(4, device=DEV).repeat(N, 1, 1), trim_dist=TRIM, loss_fn=LOSS, dim=3)
    out["T"].sum().backward()
    assert out["deltas"].shape == (N, K, 6, 1) and out["weights"].shape == (N, K, 3 * n, 1)
    sc, tc = src[SLICE].double().requires_grad_(True), tg[SLICE].double().requires_grad_(True)
    nb = sc.shape[0]
    ref = O.icp_batched(sc, tc, torch.eye(4, dtype=torch.float64).repeat(nb, 1, 1), torch.ones(nb, 3 * n, dtype=torch.float64), icp_type="pt2pt", differentiable=True,
                        max_iterations=K, tolerance=1e-12, trim_dist=TRIM, loss_fn=LOSS, dim=3, const_iter=True, tanh_steepness=5.0)
    ref["T"].sum().backward()
    np.testing.assert_allclose(npy(out["T"])[SLICE].astype(np.float64), npy(ref["T"]), rtol=0, atol=1e-4)
    np.testing.assert_allclose(npy(out["costs"])[SLICE].astype(np.float64), npy(ref["costs"]), rtol=2e-3, atol=1e-3)
    for got, want in ((s.grad[SLICE], sc.grad), (t.grad[SLICE], tc.grad)):
        scale = max(1.0, float(want.abs().max()))
        assert float((got.cpu().double() - want).abs().max()) <= 1e-3 * scale, float((got.cpu().double() - want).abs().max()) / scale


def kabsch_np(p, y):
    mus, mut = p.mean(0), y.mean(0)
    W = (y - mut).T @ (p - mus) / len(p)
    U, _, Vt = np.linalg.svd(W)
    C = U @ np.diag([1, 1, np.linalg.det(U) * np.linalg.det(Vt)]) @ Vt
    return C, mut - C @ mus


def test_svd_step_against_kabsch_iterations_and_autograd():
    """ICP.pt2pt_dICP_SVD as bench.py calls it.  Reference per cloud, float64: K x { nearest neighbours under the current pose (oracle's nn_index), Kabsch on
    (source, matched rows) } -- every match lies far inside the trim gate on these clouds, so all weights are one --, and, for the gradient of T.sum(), torch
    autograd through the Kabsch solve on the final correspondences (the argmin carries no gradient, nn.py:35, and the composed updates telescope to that solve)."""
    src, tg = workload()
    icp = ICP(icp_type="pt2pt", differentiable=True, max_iterations=K, tolerance=1e-12)
    icp.const_iter = True
    s, t = src.to(DEV).requires_grad_(True), tg.to(DEV).requires_grad_(True)
    pc, T = icp.pt2pt_dICP_SVD(s, t, torch.eye(4, device=DEV).repeat(N, 1, 1), trim_dist=TRIM)
    T.sum().backward()
    for b in range(N)[SLICE]:
        p, y = src[b].double().numpy(), tg[b].double()
        C, r = np.eye(3), np.zeros(3)
        for _ in range(K):
            idx = O.nn_index(torch.tensor(p @ C.T + r)[None], y[None])[0]
            assert float(np.linalg.norm(p @ C.T + r - y[idx].numpy(), axis=1).max()) < TRIM
            C, r = kabsch_np(p, y[idx].numpy())
        sc, tc = src[b].double().requires_grad_(True), tg[b].double().requires_grad_(True)
        ym = tc[idx]
        mus, mut = sc.mean(0), ym.mean(0)
        W = (ym - mut).T @ (sc - mus) / n
        U, _, Vh = torch.linalg.svd(W)
        one = torch.ones((), dtype=torch.float64)
        Ct = U @ torch.diag(torch.stack([one, one, torch.det(U) * torch.det(Vh)])) @ Vh
        rt = mut - Ct @ mus
        (Ct.sum() + rt.sum() + 1.0).backward()                                  # T.sum() = sum C + sum r + 1
        Tt = np.eye(4)
        Tt[:3, :3], Tt[:3, 3] = Ct.detach().numpy(), rt.detach().numpy()
        np.testing.assert_allclose(npy(T)[b].astype(np.float64), Tt, rtol=0, atol=1e-4)
        np.testing.assert_allclose(npy(pc)[b].astype(np.float64), p @ Tt[:3, :3].T + Tt[:3, 3], rtol=0, atol=2e-4)
        for got, want in ((s.grad[b], sc.grad), (t.grad[b], tc.grad)):
            scale = max(1.0, float(want.abs().max()))
            assert float((got.cpu().double() - want).abs().max()) <= 1e-3 * scale, (b, float((got.cpu().double() - want).abs().max()) / scale)
