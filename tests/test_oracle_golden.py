"""Pin the CPU oracle (oracle/dicp_oracle.py) to the reference's own outputs.

The golden vectors were produced by tests/golden/make_golden.py from the imported
reference; everything here runs on CPU (`-m "not gpu"`).
"""
import numpy as np
import pytest
import torch

from oracle import dicp_oracle as O
from oracle.se3 import tran2vec, vec2tran

T64 = dict(rtol=0, atol=1e-11)


def t(a, dtype=None, grad=False):
    x = torch.tensor(np.asarray(a), dtype=dtype)
    return x.requires_grad_(True) if grad else x


def check_result(res, g, prefix="", atol=1e-11, gatol=None):
    assert res["T"].shape == g[prefix + "T"].shape
    np.testing.assert_allclose(res["T"].detach().numpy(), g[prefix + "T"], rtol=0, atol=atol)
    np.testing.assert_allclose(res["deltas"].numpy(), g[prefix + "deltas"], rtol=0, atol=atol)
    np.testing.assert_allclose(res["costs"].numpy(), g[prefix + "costs"], rtol=1e-9, atol=atol)
    if prefix + "weights" in g:
        np.testing.assert_allclose(res["weights"].numpy(), g[prefix + "weights"], rtol=0, atol=atol)
    if prefix + "pc" in g:
        np.testing.assert_allclose(res["pc"].detach().numpy(), g[prefix + "pc"], rtol=0, atol=atol)
    np.testing.assert_array_equal(res["stats"]["converged"].numpy(), g[prefix + "stats_converged"])
    np.testing.assert_allclose(res["stats"]["iterations"].numpy(), g[prefix + "stats_iterations"])
    np.testing.assert_allclose(res["stats"]["matched_ratio"].numpy(), g[prefix + "stats_matched_ratio"], atol=1e-12)


@pytest.mark.parametrize("name,icp_type,diff", [
    ("c1_pt2pt_diff", "pt2pt", True),
    ("c1_pt2pl_diff", "pt2pl", True),
    ("c1_pt2pt_hard", "pt2pt", False),
])
def test_c1_matches_reference(golden, name, icp_type, diff):
    g = golden(name)
    trim, huber, tol, max_iter = g["params"]
    src = t(g["source"], grad=True)
    tgt = t(g["target"], grad=True)
    r = 3 if icp_type == "pt2pt" else 1
    w = torch.ones((1, src.shape[0] * r), dtype=src.dtype)
    res = O.icp_batched(src.unsqueeze(0), tgt.unsqueeze(0), t(g["T_init"]).unsqueeze(0), w,
                        icp_type=icp_type, differentiable=diff, max_iterations=int(max_iter),
                        tolerance=float(tol), trim_dist=float(trim),
                        loss_fn={"name": "huber", "metric": float(huber)}, dim=2)
    check_result(res, g)
    # the reference's own assertion (tests/test_ICP.py:65-66): pose == ground truth
    err = tran2vec(g["T_ts_true"] @ np.linalg.inv(res["T"][0].detach().numpy()))
    assert np.linalg.norm(err) < float(tol)
    res["T"].sum().backward()
    np.testing.assert_allclose(src.grad.numpy(), g["grad_source"], **T64)
    np.testing.assert_allclose(tgt.grad.numpy(), g["grad_target"], **T64)


def test_ground_truth_convention():
    """vec2tran reproduces SURVEY.md section 8a's T_ts_true (pins the pylgmath stand-in)."""
    T = np.linalg.inv(vec2tran([1.0, 1.0, 0, 0, 0, 0.1]))
    want = np.array([[0.99500417, 0.09983342, 0, -1.04829251],
                     [-0.09983342, 0.99500417, 0, -0.94837582],
                     [0, 0, 1, 0], [0, 0, 0, 1]])
    np.testing.assert_allclose(T, want, atol=5e-9)
    xi = np.array([0.3, -0.2, 0.5, 0.1, -0.4, 0.25])
    np.testing.assert_allclose(tran2vec(vec2tran(xi)).ravel(), xi, atol=1e-13)


@pytest.mark.parametrize("name,icp_type,loss_fn,wkey", [
    ("input_types", "pt2pl", {"name": "huber", "metric": 1.0}, None),
    ("input_types_pt2pt", "pt2pt", {"name": "huber", "metric": 1.0}, None),
    ("zero_inputs", "pt2pl", None, None),
    ("weight_inputs", "pt2pl", {"name": "huber", "metric": 1.0}, None),
])
def test_batched_scenarios(golden, name, icp_type, loss_fn, wkey):
    """tests/test_ICP_inputs.py scenarios, entered after the reference's batch_size_handling."""
    g = golden(name)
    res = O.icp_batched(t(g["bsh_source"]), t(g["bsh_target"]), t(g["bsh_T"]), t(g["bsh_w"]),
                        icp_type=icp_type, differentiable=True, max_iterations=25, tolerance=1e-8,
                        trim_dist=5.0, loss_fn=loss_fn, dim=2)
    check_result(res, g, "batch_")


def test_padded_inputs(golden):
    g = golden("padded_inputs")
    res = O.icp_batched(t(g["bsh_source"]), t(g["bsh_target"]), t(g["bsh_T"]), t(g["bsh_w"]),
                        icp_type="pt2pt", differentiable=False, max_iterations=25, tolerance=1e-8, dim=2)
    check_result(res, g, "padded_")


def matrix_keys(g):
    return sorted(k[:-len("__T")] for k in g if k.endswith("__T"))


@pytest.mark.parametrize("fixture,count", [("matrix3d", 28), ("matrix3d_trimloss", 12)])
def test_matrix3d(golden, fixture, count):
    """matrix3d_trimloss: loss_fn={"name": "trim"} (ICP.py:157-160 -> loss.py:15-16), a valid reference call."""
    g = golden(fixture)
    K = int(g["K"])
    metric = float(g["loss_metric"]) if "loss_metric" in g else 0.3
    keys = matrix_keys(g)
    assert len(keys) == count
    for key in keys:
        icp_type, mode, lname, trim, d = key.split("_")
        src = t(g["source"], grad=True)
        tgt = t(g["target"] if icp_type == "pt2pl" else g["target"][:, :, :3], grad=True)
        w = t(g["weight"], grad=True)
        T0 = t(g["T_init"], grad=True)
        w_in = w if icp_type == "pt2pl" else w.repeat_interleave(3, dim=1)
        res = O.icp_batched(src, tgt, T0, w_in, icp_type=icp_type, differentiable=(mode == "diff"),
                            max_iterations=K, tolerance=1e-14, trim_dist=(1.5 if trim == "trim" else None),
                            loss_fn=None if lname == "none" else {"name": lname, "metric": metric},
                            dim=int(d[1]), const_iter=True)
        check_result(res, g, key + "__")
        np.testing.assert_allclose(res["weights"][:, -1, :, 0].numpy(), g[key + "__w_last"], **T64)
        ((res["T"] * t(g["gT"])).sum() + (res["pc"] * t(g["gpc"])).sum()).backward()
        for nm, leaf in (("source", src), ("target", tgt), ("weight", w), ("T_init", T0)):
            np.testing.assert_allclose(leaf.grad.numpy(), g[key + "__grad_" + nm], rtol=1e-9, atol=1e-10,
                                       err_msg=key + " " + nm)


def test_nn_vectors(golden):
    g = golden("nn_vectors")
    y = t(g["y"], grad=True)
    nb = O.find_nn(t(g["x"]), y, differentiable=False)
    np.testing.assert_array_equal(nb.detach().numpy(), g["nb"])
    (nb * t(g["cot"])).sum().backward()
    np.testing.assert_allclose(y.grad.numpy(), g["grad_y"], **T64)
    nb_T = O.find_nn(t(g["x"]).transpose(1, 2), t(g["y"]).transpose(1, 2), differentiable=False)
    np.testing.assert_array_equal(nb_T.numpy(), g["nb_T"])
    np.testing.assert_array_equal(O.find_nn(t(g["x"][0]), t(g["y"][0]), differentiable=False).numpy(), g["nb_2d"])
    xs = t(g["x"], torch.float32, grad=True)
    ys = t(g["y"], torch.float32, grad=True)
    soft = O.find_nn(xs, ys, True, True, eps=1e-10, tau=0.1, U=t(g["U"]))
    np.testing.assert_allclose(soft.detach().numpy(), g["nb_soft"], rtol=0, atol=1e-6)
    (soft * t(g["cot"], torch.float32)).sum().backward()
    np.testing.assert_allclose(xs.grad.numpy(), g["grad_x_soft"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(ys.grad.numpy(), g["grad_y_soft"], rtol=1e-4, atol=1e-5)


def test_nn_known_answer(golden):
    """/root/reference/tests/test_nn.py:10,20-21,36-37 (hard path: the exact neighbour)."""
    g = golden("nn_vectors")
    pts = t(g["kat_points"])
    q = t(g["kat_query"])
    assert torch.equal(O.find_nn(q, pts, differentiable=False)[0, 0], t(g["kat_expect1"]))
    pts2 = torch.cat((pts, t(g["kat_extra"]).view(1, -1)))
    assert torch.equal(O.find_nn(q, pts2, differentiable=False)[0, 0], t(g["kat_expect2"]))


def test_loss_vectors(golden):
    g = golden("loss_vectors")
    for name, metric in (("huber", 1.0), ("cauchy", 0.5), ("trim", 2.0)):
        for diff in (True, False):
            for tag in ("e1", "e3", "eb"):
                key = "%s_%s_%s" % (name, "diff" if diff else "hard", tag)
                e = t(g[tag], grad=True)
                w = O.loss_weight(e, name, metric, diff, 5.0)
                np.testing.assert_allclose(w.detach().numpy(), g[key], rtol=0, atol=1e-14, err_msg=key)
                if key + "_grad" in g:
                    w.sum().backward()
                    np.testing.assert_allclose(e.grad.numpy(), g[key + "_grad"], rtol=0, atol=1e-13, err_msg=key)
    with pytest.raises(ValueError):
        O.loss_weight(t(g["e1"]), "tukey", 1.0)


def test_knn_exact_helper():
    gen = torch.Generator().manual_seed(3)
    x = torch.rand((2, 33, 3), generator=gen, dtype=torch.float64)
    y = torch.rand((2, 41, 6), generator=gen, dtype=torch.float64)
    idx, best, second = O.knn_exact_f64(x, y)
    assert torch.equal(idx, O.nn_index(x, y))
    assert bool((second >= best).all())


def test_chunked_argmin_is_the_same_oracle():
    """O.NN_CHUNK (memory bound for 65536-point clouds): same indices, same results, same gradients."""
    gen = torch.Generator().manual_seed(11)
    src = torch.rand((2, 700, 3), generator=gen) * 4
    tgt = torch.rand((2, 900, 6), generator=gen) * 4
    kw = dict(icp_type="pt2pl", differentiable=True, max_iterations=3, tolerance=1e-12, trim_dist=5.0,
              loss_fn={"name": "huber", "metric": 1.0}, const_iter=True)
    outs = []
    for chunk in (None, 128):
        O.NN_CHUNK = chunk
        try:
            s, t_ = src.clone().requires_grad_(True), tgt.clone().requires_grad_(True)
            r = O.icp_batched(s, t_, torch.eye(4).repeat(2, 1, 1), torch.ones(2, 700), **kw)
            r["T"].sum().backward()
            outs.append((r["T"].detach(), s.grad, t_.grad, O.nn_index(src, tgt)))
        finally:
            O.NN_CHUNK = None
    for a, b in zip(*outs):
        assert torch.equal(a, b)
