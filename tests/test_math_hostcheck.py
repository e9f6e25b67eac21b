"""CPU check of the closed-form arithmetic the HIP kernels execute.

``dicp_amd/csrc/dicp_math.h`` is compiled here with g++ (tests/hostcheck/, a TEST-ONLY
build) and driven through the same forward / reverse loop structure as the product;
results are held to the oracle's forward values and to its *autograd* gradients on the
reference-generated ``matrix3d`` scenarios.  This validates the backward formulas of
SURVEY.md section 8(a-11) without a GPU.
"""
import ctypes
import os
import shutil
import subprocess

import numpy as np
import pytest
import scipy.linalg
import torch

from oracle import dicp_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "hostcheck", "hostcheck.cpp")
LIB = os.path.join(HERE, "hostcheck", "libhostcheck.so")

pytestmark = pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")


class WeightParams(ctypes.Structure):
    _fields_ = [("mode", ctypes.c_int), ("trim_on", ctypes.c_int), ("differentiable", ctypes.c_int),
                ("loss", ctypes.c_int), ("trim_dist", ctypes.c_double), ("tanh_k", ctypes.c_double),
                ("loss_delta", ctypes.c_double), ("match_thresh", ctypes.c_double)]


@pytest.fixture(scope="module")
def hc():
    hdr = os.path.join(HERE, "..", "dicp_amd", "csrc", "dicp_math.h")
    if (not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(SRC), os.path.getmtime(hdr))):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-Wno-unknown-pragmas", "-o", LIB, SRC])
    lib = ctypes.CDLL(LIB)
    assert lib.hc_sizeof_params() == ctypes.sizeof(WeightParams)
    return lib


def ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def run_chain(hc, src, tgt, w0, T0, P, K, dim, gT, gpc):
    """Forward K iterations + reverse sweep for ONE cloud using the hostcheck math."""
    n, c = src.shape[0], tgt.shape[1]
    mask_s = np.array([1, 1, 0.0]) if dim == 2 else np.ones(3)
    mask_t = (np.array([1, 1, 0, 1, 1, 0.0]) if dim == 2 else np.ones(6))[:c]
    s = np.ascontiguousarray(src * mask_s)
    t = np.ascontiguousarray(tgt * mask_t)
    C = np.ascontiguousarray(T0[:3, :3]).copy()
    r = np.ascontiguousarray(T0[:3, 3]).copy()
    saved = []
    for _ in range(K):
        pt = s @ C.T + r
        idx = O.nn_index(torch.tensor(pt)[None], torch.tensor(t)[None])[0].numpy().astype(np.int32)
        acc = np.zeros(30)
        w = np.zeros(n)
        hc.hc_forward_f64(ctypes.byref(P), n, c, ptr(s), ptr(t), ptr(idx), ptr(C), ptr(r), ptr(w0), ptr(acc), ptr(w))
        d6, Cn, rn, Areg = np.zeros(6), np.zeros(9), np.zeros(3), np.zeros(36)
        hc.hc_step_forward(ptr(acc), dim, ptr(C), ptr(r), ptr(d6), ptr(Cn), ptr(rn), ptr(Areg))
        saved.append((idx, C.copy(), r.copy(), d6, Areg, acc, w))
        C, r = Cn.reshape(3, 3).copy(), rn.copy()
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = C, r
    # reverse sweep: pc = s C^T + r and T both carry cotangents
    gC = gT[:3, :3] + gpc.T @ s
    gr = gT[:3, 3] + gpc.sum(0)
    gs = gpc @ C
    gt = np.zeros_like(t)
    gw = np.zeros(n)
    for idx, Ck, rk, d6, Areg, acc, w in reversed(saved):
        Gs, gb, gCo, gro = np.zeros(36), np.zeros(6), np.zeros(9), np.zeros(3)
        hc.hc_step_backward(ptr(np.ascontiguousarray(gC)), ptr(np.ascontiguousarray(gr)), dim, ptr(Ck), ptr(d6), ptr(Areg),
                            ptr(Gs), ptr(gb), ptr(gCo), ptr(gro))
        hc.hc_backward_f64(ctypes.byref(P), n, c, ptr(s), ptr(t), ptr(idx), ptr(Ck), ptr(rk), ptr(w0), ptr(Gs), ptr(gb),
                           ptr(gs), ptr(gt), ptr(gw), ptr(gCo), ptr(gro))
        gC, gr = gCo.reshape(3, 3), gro
    gT0 = np.zeros((4, 4))
    gT0[:3, :3], gT0[:3, 3] = gC, gr
    return T, saved, gs * mask_s, gt * mask_t, gw, gT0


@pytest.mark.parametrize("fixture", ["matrix3d", "matrix3d_trimloss"])
def test_matrix3d_forward_and_backward(hc, golden, fixture):
    g = golden(fixture)
    K = int(g["K"])
    metric = float(g["loss_metric"]) if "loss_metric" in g else 0.3
    keys = sorted(k[:-len("__T")] for k in g if k.endswith("__T"))
    for key in keys:
        icp_type, mode, lname, trim, d = key.split("_")
        dim = int(d[1])
        P = WeightParams(mode=1 if icp_type == "pt2pl" else 0, trim_on=int(trim == "trim"),
                         differentiable=int(mode == "diff"),
                         loss={"none": 0, "huber": 1, "cauchy": 2, "trim": 3}[lname],
                         trim_dist=1.5, tanh_k=5.0, loss_delta=metric, match_thresh=0.0)
        tgt_all = g["target"] if icp_type == "pt2pl" else np.ascontiguousarray(g["target"][:, :, :3])
        for b in range(g["source"].shape[0]):
            T, saved, gs, gt, gw, gT0 = run_chain(hc, g["source"][b], tgt_all[b], np.ascontiguousarray(g["weight"][b]),
                                                  g["T_init"][b], P, K, dim, g["gT"][b], g["gpc"][b])
            np.testing.assert_allclose(T, g[key + "__T"][b], rtol=0, atol=1e-11, err_msg=key)
            deltas = np.stack([sv[3] for sv in saved])
            np.testing.assert_allclose(deltas, g[key + "__deltas"][b, :, :, 0], rtol=0, atol=1e-11, err_msg=key)
            costs = np.array([sv[5][27] for sv in saved])
            np.testing.assert_allclose(costs, g[key + "__costs"][b, :, 0], rtol=1e-10, atol=1e-13, err_msg=key)
            w_last = saved[-1][6]
            want_w = g[key + "__w_last"][b]
            np.testing.assert_allclose(w_last, want_w[::3] if icp_type == "pt2pt" else want_w, rtol=0, atol=1e-12)
            np.testing.assert_allclose(gs, g[key + "__grad_source"][b], rtol=1e-8, atol=1e-10, err_msg=key + " src")
            np.testing.assert_allclose(gt, g[key + "__grad_target"][b], rtol=1e-8, atol=1e-10, err_msg=key + " tgt")
            np.testing.assert_allclose(gw, g[key + "__grad_weight"][b], rtol=1e-8, atol=1e-10, err_msg=key + " w")
            np.testing.assert_allclose(gT0, g[key + "__grad_T_init"][b], rtol=1e-8, atol=1e-10, err_msg=key + " T0")


def test_solver_and_exp_small_angles(hc):
    """so3 series branch and the 3-dof sub-block path agree with numpy."""
    rng = np.random.RandomState(0)
    for dim in (2, 3):
        for scale in (1e-7, 1e-3, 0.5):
            J = rng.normal(size=(40, 6))
            e = rng.normal(size=40) * scale
            A = J.T @ J
            b = J.T @ e
            acc = np.zeros(30)
            k = 0
            for i in range(6):
                for j in range(i, 6):
                    acc[k] = A[i, j]
                    k += 1
            acc[21:27] = b
            C = np.eye(3).ravel().copy()
            r = np.zeros(3)
            d6, Cn, rn, Areg = np.zeros(6), np.zeros(9), np.zeros(3), np.zeros(36)
            hc.hc_step_forward(ptr(acc), dim, ptr(C), ptr(r), ptr(d6), ptr(Cn), ptr(rn), ptr(Areg))
            sl = [2, 3, 4] if dim == 2 else list(range(6))
            want = np.zeros(6)
            want[sl] = -np.linalg.solve(A[np.ix_(sl, sl)] + 1e-12 * np.eye(len(sl)), b[sl])
            np.testing.assert_allclose(d6, want, rtol=1e-9, atol=1e-15)
            Kx = np.array([[0, -d6[2], d6[1]], [d6[2], 0, -d6[0]], [-d6[1], d6[0], 0]])
            # closed-form Rodrigues is exact to rounding; torch.matrix_exp (what the reference calls,
            # ICP.py:210) is itself only good to ~4e-12 here, so it gets the looser bound.
            np.testing.assert_allclose(Cn.reshape(3, 3), scipy.linalg.expm(Kx).T, rtol=0, atol=2e-15)
            np.testing.assert_allclose(Cn.reshape(3, 3), torch.matrix_exp(torch.tensor(Kx)).numpy().T, rtol=0, atol=1e-10)
            np.testing.assert_allclose(rn, -d6[3:], rtol=0, atol=0)


def test_hard_huber_zero_residual_is_nan_like_reference(hc):
    """Reference quirk kept on purpose: with non-differentiable Huber an exactly-zero residual makes
    autograd produce 0*(-inf)=NaN (loss.py:32 through torch.where); the closed-form adjoint does too."""
    torch.manual_seed(0)
    tgt = torch.rand(1, 20, 6, dtype=torch.float64)
    tgt[:, :, 3:] /= tgt[:, :, 3:].norm(dim=2, keepdim=True)
    src = tgt[:, :10, :3].clone() + 0.01 * torch.rand(1, 10, 3, dtype=torch.float64)
    src[0, 0] = tgt[0, 0, :3]
    for icp_type in ("pt2pl", "pt2pt"):
        tg = tgt if icp_type == "pt2pl" else tgt[:, :, :3].contiguous()
        s, t = src.clone().requires_grad_(True), tg.clone().requires_grad_(True)
        rows = 3 if icp_type == "pt2pt" else 1
        out = O.icp_batched(s, t, torch.eye(4, dtype=torch.float64)[None], torch.ones(1, 10 * rows, dtype=torch.float64),
                            icp_type=icp_type, differentiable=False, max_iterations=1, loss_fn={"name": "huber", "metric": 1.0})
        out["T"].sum().backward()
        P = WeightParams(mode=1 if icp_type == "pt2pl" else 0, trim_on=0, differentiable=0, loss=1,
                         trim_dist=0.0, tanh_k=5.0, loss_delta=1.0, match_thresh=0.0)
        T, saved, gs, gt, gw, gT0 = run_chain(hc, src[0].numpy(), tg[0].numpy(), np.ones(10), np.eye(4), P, 1, 3,
                                              np.ones((4, 4)), np.zeros((10, 3)))
        np.testing.assert_array_equal(np.isnan(gs), np.isnan(s.grad[0].numpy()))
        ok = ~np.isnan(gs)
        np.testing.assert_allclose(gs[ok], s.grad[0].numpy()[ok], rtol=1e-8, atol=1e-12)
        assert np.isnan(gs[0]).all() and not np.isnan(gs[1:]).any()


def kabsch_torch(p, y, w):
    """Weighted Kabsch/Umeyama with torch ops (autograd reference): C = U diag(1,1,det U det V) V^T."""
    S0 = w.sum()
    mus, mut = (w[:, None] * p).sum(0) / S0, (w[:, None] * y).sum(0) / S0
    W = (w[:, None, None] * y[:, :, None] * p[:, None, :]).sum(0) / S0 - mut[:, None] * mus[None, :]
    U, S, Vh = torch.linalg.svd(W)
    D = torch.diag(torch.stack([torch.ones(()), torch.ones(()), torch.det(U) * torch.det(Vh)]).to(W.dtype))
    C = U @ D @ Vh
    return C, mut - C @ mus


def kabsch_sums(p, y, w):
    acc = np.zeros(18)
    acc[0] = w.sum()
    acc[1:4] = (w[:, None] * p).sum(0)
    acc[4:7] = (w[:, None] * y).sum(0)
    acc[7:16] = (w[:, None, None] * y[:, :, None] * p[:, None, :]).sum(0).ravel()
    acc[16] = (w * (p ** 2).sum(1)).sum()
    acc[17] = (w * (y ** 2).sum(1)).sum()
    return acc


def test_svd3_and_kabsch_forward_backward(hc):
    hc.hc_kabsch_forward.restype = ctypes.c_double
    rng = np.random.RandomState(2)
    for trial in range(40):
        A = rng.normal(size=(3, 3))
        if trial % 4 == 1:
            A[:, 2] = 0.0                         # rank 2 (planar clouds)
        if trial % 8 == 7:
            A = np.outer(rng.normal(size=3), rng.normal(size=3))   # rank 1
        U, S, V = np.zeros(9), np.zeros(3), np.zeros(9)
        hc.hc_svd3(ptr(np.ascontiguousarray(A)), ptr(U), ptr(S), ptr(V))
        U, V = U.reshape(3, 3), V.reshape(3, 3)
        np.testing.assert_allclose(U @ np.diag(S) @ V.T, A, atol=1e-13)
        np.testing.assert_allclose(U.T @ U, np.eye(3), atol=1e-13)
        np.testing.assert_allclose(V.T @ V, np.eye(3), atol=1e-13)
        np.testing.assert_allclose(S, np.linalg.svd(A, compute_uv=False), atol=1e-13)
    for trial in range(10):
        n = 30
        p = rng.normal(size=(n, 3)) * 2
        ang = rng.uniform(-1, 1, 3)
        Rt = scipy.linalg.expm(np.array([[0, -ang[2], ang[1]], [ang[2], 0, -ang[0]], [-ang[1], ang[0], 0]]))
        if trial == 9:
            Rt = Rt @ np.diag([1, 1, -1.0])       # a reflection in the data: d3 = -1 branch
        y = p @ Rt.T + rng.normal(size=3) + 0.05 * rng.normal(size=(n, 3))
        w = rng.uniform(0.1, 1.0, n)
        C, r, save = np.zeros(9), np.zeros(3), np.zeros(40)
        cost = hc.hc_kabsch_forward(ptr(kabsch_sums(p, y, w)), ptr(C), ptr(r), ptr(save))
        pt, yt, wt = (torch.tensor(a, requires_grad=True) for a in (p, y, w))
        Ct, rt = kabsch_torch(pt, yt, wt)
        np.testing.assert_allclose(C.reshape(3, 3), Ct.detach().numpy(), atol=1e-12)
        np.testing.assert_allclose(r, rt.detach().numpy(), atol=1e-12)
        np.testing.assert_allclose(np.linalg.det(C.reshape(3, 3)), 1.0, atol=1e-12)
        np.testing.assert_allclose(cost, float((wt * ((pt @ Ct.T + rt - yt) ** 2).sum(1)).sum()), rtol=1e-9, atol=1e-10)
        gC, gr = rng.normal(size=(3, 3)), rng.normal(size=3)
        ((Ct * torch.tensor(gC)).sum() + (rt * torch.tensor(gr)).sum()).backward()
        gacc = np.zeros(16)
        hc.hc_kabsch_backward(ptr(np.ascontiguousarray(gC)), ptr(gr), ptr(save), ptr(gacc))
        gM = gacc[7:16].reshape(3, 3)
        gp = w[:, None] * (gacc[1:4][None, :] + y @ gM)
        gy = w[:, None] * (gacc[4:7][None, :] + p @ gM.T)
        gw = gacc[0] + p @ gacc[1:4] + y @ gacc[4:7] + np.einsum("ia,ab,ib->i", y, gM, p)
        np.testing.assert_allclose(gp, pt.grad.numpy(), rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(gy, yt.grad.numpy(), rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(gw, wt.grad.numpy(), rtol=1e-8, atol=1e-10)
