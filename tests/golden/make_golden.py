#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE itself.

Runs only in the build container (needs /root/reference).  The reference's Python
never travels: what is committed is data -- inputs and the reference's outputs --
plus this script.  Usage:

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_golden.py

Scenarios mirror the reference's own tests (file:line cited per block) with the
unseeded random draws of tests/test_ICP_inputs.py:52,172 frozen, and add a 3-D
matrix of {pt2pt,pt2pl} x {diff,hard} x {none,huber,cauchy} x {trim,no trim} on
random clouds with gradients w.r.t. source, target, weight and T_init.
"""
import os
import sys

sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")       # must precede ROOT: the repo ships a drop-in alias package also named dICP

import numpy as np
import torch

from dICP.ICP import ICP as RefICP            # the reference
from dICP.nn import nn as RefNN
from dICP.loss import loss as RefLoss
import dICP as _ref_pkg
assert _ref_pkg.__file__.startswith("/root/reference/"), "golden vectors must come from the reference, not the alias package"
from oracle.se3 import vec2tran

torch.set_num_threads(4)
SCAN = np.load("/root/reference/tests/data/points_scan.npy")
MAP = np.load("/root/reference/tests/data/points_map.npy")


def npy(t):
    if t is None:
        return None
    if isinstance(t, torch.Tensor):
        return t.detach().cpu().numpy()
    return np.asarray(t)


def pack_result(res, prefix=""):
    out = {}
    for k in ("pc", "T", "costs", "deltas", "weights"):
        out[prefix + k] = npy(res[k])
    for k in ("converged", "iterations", "matched_ratio"):
        out[prefix + "stats_" + k] = npy(res["stats"][k])
    return out


def save(name, **arrs):
    arrs = {k: v for k, v in arrs.items() if v is not None}
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrs)
    print("wrote", name, len(arrs), "arrays")


# ---------------------------------------------------------------- C1 single pair
def c1(name, icp_type, diff, huber, dtype, max_iter=100, tol=1e-10, with_grad=True):
    """tests/test_ICP.py:35-78 (pt2pt diff), :80-117 (pt2pl diff), :119-149 (pt2pt hard)."""
    src = torch.tensor(SCAN[:, :3], dtype=dtype, requires_grad=True)
    tgt = torch.tensor(MAP[:, :3] if icp_type == "pt2pt" else MAP, dtype=dtype, requires_grad=True)
    T0 = torch.eye(4, dtype=dtype)
    icp = RefICP(icp_type=icp_type, differentiable=diff, max_iterations=max_iter, tolerance=tol)
    res = icp.icp(src, tgt, T0, trim_dist=5.0, loss_fn={"name": "huber", "metric": huber}, dim=2)
    out = pack_result(res)
    if with_grad:
        res["T"].sum().backward()
        out["grad_source"] = npy(src.grad)
        out["grad_target"] = npy(tgt.grad)
    # also a pc-driven gradient (exercises the pc output's graph)
    src2 = torch.tensor(SCAN[:, :3], dtype=dtype, requires_grad=True)
    tgt2 = torch.tensor(MAP[:, :3] if icp_type == "pt2pt" else MAP, dtype=dtype, requires_grad=True)
    res2 = icp.icp(src2, tgt2, T0, trim_dist=5.0, loss_fn={"name": "huber", "metric": huber}, dim=2)
    (res2["pc"] ** 2).sum().backward()
    out["grad_source_pc2"] = npy(src2.grad)
    out["grad_target_pc2"] = npy(tgt2.grad)
    T_true = np.linalg.inv(vec2tran([1.0, 1.0, 0, 0, 0, 0.1]))
    save(name, source=npy(src), target=npy(tgt), T_init=npy(T0), T_ts_true=T_true,
         params=np.array([5.0, huber, tol, max_iter]), **out)


# ------------------------------------------------------------- ragged list batch
def input_types():
    """tests/test_ICP_inputs.py:36-110 with the outlier draw (:52) frozen."""
    rng = np.random.RandomState(1234)
    dt = torch.float64
    outlier = rng.rand(1, 3) * 1000
    s1 = torch.cat((torch.tensor(SCAN[:50, :3]), torch.tensor(outlier)), dim=0)
    t1 = torch.tensor(MAP[:55, :])
    s2, t2 = torch.tensor(SCAN[:, :3]), torch.tensor(MAP)
    s3, t3 = torch.tensor(SCAN[:55, :3]), torch.tensor(MAP[:60, :])
    S, Tg = [s1, s2, s3], [t1, t2, t3]
    T0 = torch.stack([torch.eye(4, dtype=dt)] * 3)
    icp = RefICP(icp_type="pt2pl", differentiable=True, max_iterations=25, tolerance=1e-8)
    out = {}
    for i in range(3):
        r = icp.icp(S[i], Tg[i], T0[i], trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=2)
        out.update(pack_result(r, "single%d_" % i))
    r = icp.icp(S, Tg, T0, trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=2)
    out.update(pack_result(r, "batch_"))
    sb, tb, Tb, wb = icp.batch_size_handling(S, Tg, T0, None)
    save("input_types", s0=npy(s1), s1=npy(s2), s2=npy(s3), t0=npy(t1), t1=npy(t2), t2=npy(t3),
         bsh_source=npy(sb), bsh_target=npy(tb), bsh_T=npy(Tb), bsh_w=npy(wb), **out)
    # the pt2pt flavour of the same batching (weights repeated x3, ICP.py:508-509)
    icp2 = RefICP(icp_type="pt2pt", differentiable=True, max_iterations=25, tolerance=1e-8)
    sb, tb, Tb, wb = icp2.batch_size_handling(S, [t[:, :3] for t in Tg], list(T0), None)
    r = icp2.icp(S, [t[:, :3] for t in Tg], list(T0), trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=2)
    save("input_types_pt2pt", bsh_source=npy(sb), bsh_target=npy(tb), bsh_T=npy(Tb), bsh_w=npy(wb),
         **pack_result(r, "batch_"))


def zero_inputs():
    """tests/test_ICP_inputs.py:113-155."""
    dt = torch.float64
    S = [torch.tensor(SCAN), [], []]
    Tg = [[], torch.tensor(MAP), []]
    T0 = torch.stack([torch.eye(4, dtype=dt)] * 3)
    icp = RefICP(icp_type="pt2pl", differentiable=True, max_iterations=25, tolerance=1e-8)
    out = {}
    for i in range(3):
        r = icp.icp(S[i], Tg[i], T0[i], trim_dist=5.0, loss_fn=None, dim=2)
        out.update(pack_result(r, "single%d_" % i))
    r = icp.icp(S, Tg, T0, trim_dist=5.0, loss_fn=None, dim=2)
    out.update(pack_result(r, "batch_"))
    sb, tb, Tb, wb = icp.batch_size_handling(S, Tg, T0, None)
    save("zero_inputs", bsh_source=npy(sb), bsh_target=npy(tb), bsh_T=npy(Tb), bsh_w=npy(wb), **out)


def weight_inputs():
    """tests/test_ICP_inputs.py:157-211 with the junk points (:172) frozen."""
    rng = np.random.RandomState(4321)
    dt = torch.float64
    junk = rng.rand(10, 3)
    S = [torch.tensor(SCAN[:, :3]), torch.tensor(SCAN[:, :3]), torch.tensor(np.vstack((SCAN[:, :3], junk)))]
    Tg = [torch.tensor(MAP), torch.tensor(MAP), torch.tensor(MAP)]
    W = [None, torch.tensor(np.ones(65), requires_grad=True),
         torch.tensor(np.hstack((np.ones(65), np.zeros(10))), requires_grad=True)]
    T0 = torch.stack([torch.eye(4, dtype=dt)] * 3)
    icp = RefICP(icp_type="pt2pl", differentiable=True, max_iterations=25, tolerance=1e-8)
    out = {}
    for i in range(3):
        r = icp.icp(S[i], Tg[i], T0[i], weight=W[i], trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=2)
        out.update(pack_result(r, "single%d_" % i))
    r = icp.icp(S, Tg, T0, weight=W, trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=2)
    out.update(pack_result(r, "batch_"))
    r["T"].sum().backward()
    out["grad_w1"] = npy(W[1].grad)
    out["grad_w2"] = npy(W[2].grad)
    sb, tb, Tb, wb = icp.batch_size_handling(S, Tg, T0, W)
    save("weight_inputs", junk=junk, bsh_source=npy(sb), bsh_target=npy(tb), bsh_T=npy(Tb), bsh_w=npy(wb), **out)


def diff_vs_nondiff():
    """tests/test_ICP_inputs.py:213-252."""
    out = {}
    s = torch.tensor(SCAN[:50, :3])
    t = torch.tensor(MAP[:55, :])
    T0 = torch.eye(4, dtype=s.dtype)
    for lname, metric in (("huber", 1.0), ("cauchy", 0.5)):
        for diff in (True, False):
            icp = RefICP(icp_type="pt2pl", differentiable=diff, max_iterations=25, tolerance=1e-8)
            r = icp.icp(s, t, T0, trim_dist=5.0, loss_fn={"name": lname, "metric": metric}, dim=2)
            out.update(pack_result(r, "%s_%s_" % (lname, "diff" if diff else "hard")))
    save("diff_vs_nondiff", **out)


def padded_inputs():
    """tests/test_ICP_inputs.py:254-271."""
    s = torch.tensor(SCAN[:50, :3])
    t = torch.tensor(MAP[:55, :])
    T0 = torch.eye(4, dtype=s.dtype)
    sp = torch.cat((s, torch.zeros((20, 3))))
    icp = RefICP(icp_type="pt2pt", differentiable=False, max_iterations=25, tolerance=1e-8)
    icp.source_zeroes_are_pad = True
    a = icp.icp(s, t, T0, dim=2)
    b = icp.icp(sp, t, T0, dim=2)
    sb, tb, Tb, wb = icp.batch_size_handling(sp, t, T0, None)
    save("padded_inputs", bsh_source=npy(sb), bsh_target=npy(tb), bsh_T=npy(Tb), bsh_w=npy(wb),
         **pack_result(a, "plain_"), **pack_result(b, "padded_"))


# ------------------------------------------------------------------ 3-D matrix
def make_cloud(rng, N, n, m, noise=0.02):
    tgt = rng.uniform(-3, 3, size=(N, m, 3))
    nrm = rng.normal(size=(N, m, 3))
    nrm /= np.linalg.norm(nrm, axis=2, keepdims=True)
    pick = np.stack([rng.permutation(m)[:n] for _ in range(N)])
    src_t = np.take_along_axis(tgt, pick[:, :, None], axis=1) + noise * rng.normal(size=(N, n, 3))
    out_src = np.empty_like(src_t)
    for b in range(N):
        T = vec2tran(np.concatenate([rng.uniform(-0.15, 0.15, 3), rng.uniform(-0.06, 0.06, 3)]))
        out_src[b] = (src_t[b] - T[:3, 3]) @ T[:3, :3]      # p = C^T (s - r)
    return out_src, np.concatenate([tgt, nrm], axis=2)


def matrix3d():
    rng = np.random.RandomState(77)
    N, n, m, K = 3, 48, 60, 4
    src_np, tgt_np = make_cloud(rng, N, n, m)
    w_np = rng.uniform(0.2, 1.0, size=(N, n))
    T0_np = np.stack([vec2tran(np.concatenate([rng.uniform(-0.05, 0.05, 3), rng.uniform(-0.02, 0.02, 3)])) for _ in range(N)])
    gT = rng.normal(size=(N, 4, 4))     # fixed cotangent for T (richer than T.sum())
    gpc = rng.normal(size=(N, n, 3))
    arrs = dict(source=src_np, target=tgt_np, weight=w_np, T_init=T0_np, gT=gT, gpc=gpc, K=np.array(K))
    for dtype, tag in ((torch.float64, "f64"),):
        for icp_type in ("pt2pl", "pt2pt"):
            for diff in (True, False):
                for lname in ("none", "huber", "cauchy"):
                    for trim in (None, 1.5):
                        for dim in (3, 2):
                            if dim == 2 and not (lname == "huber" and trim is not None):
                                continue
                            key = "%s_%s_%s_%s_d%d" % (icp_type, "diff" if diff else "hard", lname,
                                                       "trim" if trim else "notrim", dim)
                            src = torch.tensor(src_np, dtype=dtype, requires_grad=True)
                            tg = tgt_np if icp_type == "pt2pl" else tgt_np[:, :, :3]
                            tgt = torch.tensor(tg, dtype=dtype, requires_grad=True)
                            w = torch.tensor(w_np, dtype=dtype, requires_grad=True)
                            T0 = torch.tensor(T0_np, dtype=dtype, requires_grad=True)
                            icp = RefICP(icp_type=icp_type, differentiable=diff, max_iterations=K, tolerance=1e-14)
                            icp.const_iter = True
                            lf = None if lname == "none" else {"name": lname, "metric": 0.3}
                            r = icp.icp(src, tgt, T0, weight=w, trim_dist=trim, loss_fn=lf, dim=dim)
                            obj = (r["T"] * torch.tensor(gT, dtype=dtype)).sum() + (r["pc"] * torch.tensor(gpc, dtype=dtype)).sum()
                            obj.backward()
                            for k, v in pack_result(r, key + "__").items():
                                if k.endswith("weights") or k.endswith("pc"):
                                    continue        # keep the file small; weights pinned elsewhere
                                arrs[k] = v
                            arrs[key + "__w_last"] = npy(r["weights"][:, -1, :, 0])
                            arrs[key + "__grad_source"] = npy(src.grad)
                            arrs[key + "__grad_target"] = npy(tgt.grad)
                            arrs[key + "__grad_weight"] = npy(w.grad)
                            arrs[key + "__grad_T_init"] = npy(T0.grad)
    save("matrix3d", **arrs)


def matrix3d_trimloss():
    """loss_fn={"name": "trim", ...} is a valid call (ICP.py:157-160 -> loss.py:15-16,43-58: the trim gate applied to the
    residual the robust loss sees, |e| for pt2pl and |e3| for pt2pt, on top of -- or instead of -- trim_dist).  Same clouds,
    weights, T_init and cotangents as matrix3d (same seed), loss metric 1.2."""
    rng = np.random.RandomState(77)
    N, n, m, K = 3, 48, 60, 4
    src_np, tgt_np = make_cloud(rng, N, n, m)
    w_np = rng.uniform(0.2, 1.0, size=(N, n))
    T0_np = np.stack([vec2tran(np.concatenate([rng.uniform(-0.05, 0.05, 3), rng.uniform(-0.02, 0.02, 3)])) for _ in range(N)])
    gT = rng.normal(size=(N, 4, 4))
    gpc = rng.normal(size=(N, n, 3))
    metric = 1.2
    arrs = dict(source=src_np, target=tgt_np, weight=w_np, T_init=T0_np, gT=gT, gpc=gpc, K=np.array(K), loss_metric=np.array(metric))
    dtype = torch.float64
    for icp_type in ("pt2pl", "pt2pt"):
        for diff in (True, False):
            for trim in (None, 1.5):
                for dim in (3, 2):
                    if dim == 2 and trim is None:
                        continue
                    key = "%s_%s_trim_%s_d%d" % (icp_type, "diff" if diff else "hard", "trim" if trim else "notrim", dim)
                    src = torch.tensor(src_np, dtype=dtype, requires_grad=True)
                    tg = tgt_np if icp_type == "pt2pl" else tgt_np[:, :, :3]
                    tgt = torch.tensor(tg, dtype=dtype, requires_grad=True)
                    w = torch.tensor(w_np, dtype=dtype, requires_grad=True)
                    T0 = torch.tensor(T0_np, dtype=dtype, requires_grad=True)
                    icp = RefICP(icp_type=icp_type, differentiable=diff, max_iterations=K, tolerance=1e-14)
                    icp.const_iter = True
                    r = icp.icp(src, tgt, T0, weight=w, trim_dist=trim, loss_fn={"name": "trim", "metric": metric}, dim=dim)
                    obj = (r["T"] * torch.tensor(gT, dtype=dtype)).sum() + (r["pc"] * torch.tensor(gpc, dtype=dtype)).sum()
                    obj.backward()
                    for k, v in pack_result(r, key + "__").items():
                        if k.endswith("weights") or k.endswith("pc"):
                            continue
                        arrs[k] = v
                    arrs[key + "__w_last"] = npy(r["weights"][:, -1, :, 0])
                    arrs[key + "__grad_source"] = npy(src.grad)
                    arrs[key + "__grad_target"] = npy(tgt.grad)
                    arrs[key + "__grad_weight"] = npy(w.grad)
                    arrs[key + "__grad_T_init"] = npy(T0.grad)
    save("matrix3d_trimloss", **arrs)


# -------------------------------------------------------------------- nn / loss
def nn_vectors():
    rng = np.random.RandomState(5)
    x = rng.uniform(-5, 5, size=(3, 70, 3))
    y = rng.uniform(-5, 5, size=(3, 90, 6))
    hard = RefNN(differentiable=False)
    yt = torch.tensor(y, requires_grad=True)
    nb = hard.find_nn(torch.tensor(x), yt)
    g = rng.normal(size=nb.shape)
    (nb * torch.tensor(g)).sum().backward()
    # layouts accepted by nn.py:94-125
    nb_T = hard.find_nn(torch.tensor(x).transpose(1, 2), torch.tensor(y).transpose(1, 2))
    nb_2d = hard.find_nn(torch.tensor(x[0]), torch.tensor(y[0]))
    # Gumbel path with the uniform draw injected (nn.py:60)
    U = rng.uniform(size=(3, 70, 90)).astype(np.float32)
    real_rand = torch.rand
    torch.rand = lambda *a, **k: torch.tensor(U)
    try:
        soft = RefNN(differentiable=True, use_gumbel=True, eps=1e-10, tau=0.1)
        xs = torch.tensor(x, dtype=torch.float32, requires_grad=True)
        ys = torch.tensor(y, dtype=torch.float32, requires_grad=True)
        nb_soft = soft.find_nn(xs, ys)
        (nb_soft * torch.tensor(g, dtype=torch.float32)).sum().backward()
    finally:
        torch.rand = real_rand
    # known-answer test of the reference, tests/test_nn.py:10,20-21,36-37
    pts = np.array([(5.0, 4.0, 0.0), (2.0, 6.0, 0.0), (13.0, 3.0, 0.0), (8.0, 7.0, 0.0), (3.0, 1.0, 0.0)], dtype=np.float32)
    save("nn_vectors", x=x, y=y, nb=npy(nb), cot=g, grad_y=npy(yt.grad), nb_T=npy(nb_T), nb_2d=npy(nb_2d),
         U=U, nb_soft=npy(nb_soft), grad_x_soft=npy(xs.grad), grad_y_soft=npy(ys.grad),
         kat_points=pts, kat_query=np.array([[9, 4, 0]], dtype=np.float32),
         kat_expect1=np.array([8, 7, 0], dtype=np.float32), kat_extra=np.array([10, 2, 0], dtype=np.float32),
         kat_expect2=np.array([10, 2, 0], dtype=np.float32))


def loss_vectors():
    rng = np.random.RandomState(9)
    e1 = np.concatenate([np.zeros((2, 1)), rng.normal(scale=2.0, size=(40, 1))])
    e3 = np.concatenate([np.zeros((2, 3)), rng.normal(scale=2.0, size=(40, 3))])
    eb = rng.normal(scale=2.0, size=(3, 17, 3))
    arrs = dict(e1=e1, e3=e3, eb=eb)
    for name, metric in (("huber", 1.0), ("cauchy", 0.5), ("trim", 2.0)):
        for diff in (True, False):
            for tag, e in (("e1", e1), ("e3", e3), ("eb", eb)):
                et = torch.tensor(e, requires_grad=True)
                w = RefLoss(name=name, metric=metric, differentiable=diff, tanh_steepness=5.0).get_weight(et)
                key = "%s_%s_%s" % (name, "diff" if diff else "hard", tag)
                arrs[key] = npy(w)
                if w.requires_grad:
                    w.sum().backward()
                    arrs[key + "_grad"] = npy(et.grad)
    save("loss_vectors", **arrs)


def svd_planar():
    """ICP.pt2pt_dICP_SVD (ICP.py:533-591) on the bundled planar pair -- the one setting where the
    reference's V-for-V^T composition is harmless."""
    src = torch.tensor(SCAN[:, :3])
    tgt = torch.tensor(MAP[:, :3])
    icp = RefICP(icp_type="pt2pt", differentiable=False, max_iterations=100, tolerance=1e-20)
    ps, T = icp.pt2pt_dICP_SVD(src, tgt, torch.eye(4, dtype=src.dtype))
    save("svd_planar", pc=npy(ps), T=npy(T), T_ts_true=np.linalg.inv(vec2tran([1.0, 1.0, 0, 0, 0, 0.1])))


def svd_tinit():
    """ICP.pt2pt_dICP_SVD with T_init != I (ICP.py:545-547,578): the points are NOT moved by T_init -- the search starts
    from the raw source and the result is T_ts = (product of the updates) @ T_init, ps = (product of the updates) source."""
    src = torch.tensor(SCAN[:, :3])
    tgt = torch.tensor(MAP[:, :3])
    T0 = torch.tensor(vec2tran([0.3, -0.2, 0, 0, 0, 0.05]))
    icp = RefICP(icp_type="pt2pt", differentiable=False, max_iterations=100, tolerance=1e-20)
    ps, T = icp.pt2pt_dICP_SVD(src, tgt, T0)
    save("svd_tinit", pc=npy(ps), T=npy(T), T_init=npy(T0))


if __name__ == "__main__":
    if len(sys.argv) > 1:                      # regenerate selected files only: make_golden.py svd_planar ...
        for fn in sys.argv[1:]:
            globals()[fn]()
        sys.exit(0)
    np.save(os.path.join(HERE, "points_scan.npy"), SCAN)
    np.save(os.path.join(HERE, "points_map.npy"), MAP)
    c1("c1_pt2pt_diff", "pt2pt", True, 1.0, torch.float64)
    c1("c1_pt2pl_diff", "pt2pl", True, 10.0, torch.float64)
    c1("c1_pt2pt_hard", "pt2pt", False, 10.0, torch.float64)
    c1("c1_pt2pt_diff_f32", "pt2pt", True, 1.0, torch.float32, max_iter=30)
    c1("c1_pt2pl_diff_f32", "pt2pl", True, 10.0, torch.float32, max_iter=30)
    input_types()
    zero_inputs()
    weight_inputs()
    diff_vs_nondiff()
    padded_inputs()
    matrix3d()
    matrix3d_trimloss()
    nn_vectors()
    loss_vectors()
    svd_planar()
    svd_tinit()
