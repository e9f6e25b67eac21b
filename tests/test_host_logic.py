"""Host-side logic of dicp_amd (no GPU): batching, layout normalisation, argument checks,
and the rule that nothing computes without a HIP device."""
import numpy as np
import pytest
import torch

from dicp_amd.ICP import ICP
from dicp_amd.loss import loss
from dicp_amd.nn import nn
from oracle import dicp_oracle as O

HAS_GPU = torch.cuda.is_available()


def t(a):
    return torch.tensor(np.asarray(a))


def check_bsh(out, g):
    s, tg, T, w = out
    np.testing.assert_array_equal(s.detach().numpy(), g["bsh_source"])
    np.testing.assert_array_equal(tg.detach().numpy(), g["bsh_target"])
    np.testing.assert_array_equal(T.numpy(), g["bsh_T"])
    np.testing.assert_array_equal(w.detach().numpy(), g["bsh_w"])
    assert s.dtype == torch.tensor(g["bsh_source"]).dtype


def test_batching_ragged_lists(golden, scan_map):
    """tests/test_ICP_inputs.py:36-110 inputs -> the reference's batch_size_handling output."""
    g = golden("input_types")
    S = [t(g["s0"]), t(g["s1"]), t(g["s2"])]
    Tg = [t(g["t0"]), t(g["t1"]), t(g["t2"])]
    T0 = torch.stack([torch.eye(4, dtype=torch.float64)] * 3)
    check_bsh(ICP(icp_type="pt2pl").batch_size_handling(S, Tg, T0, None), g)
    g2 = golden("input_types_pt2pt")
    check_bsh(ICP(icp_type="pt2pt").batch_size_handling(S, [x[:, :3] for x in Tg], list(T0), None), g2)


def test_batching_empty_clouds(golden, scan_map):
    """tests/test_ICP_inputs.py:113-155."""
    scan, mp = scan_map
    g = golden("zero_inputs")
    T0 = torch.stack([torch.eye(4, dtype=torch.float64)] * 3)
    icp = ICP(icp_type="pt2pl")
    check_bsh(icp.batch_size_handling([t(scan), [], []], [[], t(mp), []], T0, None), g)
    # whole-input empties -> phony pair, identity T, zero weight (ICP.py:328-346)
    for src, tgt in (([], t(mp)), (t(scan), []), (None, t(mp)), ([], [])):
        s, tg, T, w = icp.batch_size_handling(src, tgt, torch.eye(4, dtype=torch.float64))
        assert s.shape == (1, 1, 3) and tg.shape == (1, 1, 6) and w.shape == (1, 1)
        assert s.dtype == torch.float32 and float(w.sum()) == 0.0
        assert torch.equal(T, torch.eye(4).unsqueeze(0))
    assert ICP(icp_type="pt2pt").batch_size_handling([], [])[3].shape == (1, 3)


def test_batching_weights_and_padding(golden, scan_map):
    """tests/test_ICP_inputs.py:157-211 and :254-271."""
    scan, mp = scan_map
    g = golden("weight_inputs")
    S = [t(scan[:, :3]), t(scan[:, :3]), t(np.vstack((scan[:, :3], g["junk"])))]
    Tg = [t(mp)] * 3
    W = [None, t(np.ones(65)).requires_grad_(True), t(np.hstack((np.ones(65), np.zeros(10)))).requires_grad_(True)]
    T0 = torch.stack([torch.eye(4, dtype=torch.float64)] * 3)
    out = ICP(icp_type="pt2pl").batch_size_handling(S, Tg, T0, W)
    check_bsh(out, g)
    out[3].sum().backward()                       # the weight graph survives batching
    assert W[1].grad is not None and float(W[1].grad.sum()) == 65.0
    gp = golden("padded_inputs")
    icp = ICP(icp_type="pt2pt", differentiable=False)
    icp.source_zeroes_are_pad = True
    sp = torch.cat((t(scan[:50, :3]), torch.zeros((20, 3), dtype=torch.float64)))
    check_bsh(icp.batch_size_handling(sp, t(mp[:55]), torch.eye(4, dtype=torch.float64), None), gp)


def test_batching_tensor_forms_and_errors(scan_map):
    scan, mp = scan_map
    icp = ICP()
    s3 = t(scan)[None].repeat(2, 1, 1)
    out = icp.batch_size_handling(s3, t(mp)[None].repeat(2, 1, 1), torch.eye(4, dtype=torch.float64)[None].repeat(2, 1, 1))
    assert out[0].shape == (2, 65, 3) and out[1].shape == (2, 65, 6) and out[3].shape == (2, 65)
    w = torch.rand(65, dtype=torch.float64)
    assert torch.equal(icp.batch_size_handling(t(scan), t(mp), None, w)[3], w[None])
    with pytest.raises(ValueError):
        icp.batch_size_handling(torch.zeros(5, 4), t(mp))                      # ICP.py:442
    with pytest.raises(ValueError):
        icp.batch_size_handling(t(scan), torch.zeros(5, 4, dtype=torch.float64))   # ICP.py:491
    with pytest.raises(ValueError):
        icp.batch_size_handling(t(scan), t(mp), torch.eye(3))                  # ICP.py:502
    with pytest.raises(ValueError):
        icp.batch_size_handling([t(scan[:, :3]), torch.zeros(4, 5, dtype=torch.float64)], [t(mp), t(mp)])   # ICP.py:384
    with pytest.raises(ValueError):
        icp.batch_size_handling([t(scan), t(scan)], [t(mp), t(mp[:, :3])])     # ICP.py:469
    with pytest.raises(AssertionError):
        icp.batch_size_handling([t(scan)], [t(mp)], None, [w, w])              # ICP.py:324
    with pytest.raises(AssertionError):
        icp.batch_size_handling([t(scan)], [t(mp)], None, [w[:5]])             # ICP.py:372
    with pytest.raises(AssertionError):
        icp.icp(t(scan), t(mp), torch.eye(4, dtype=torch.float64), dim=4)      # ICP.py:79


def test_config_surface():
    icp = ICP(icp_type="pt2pt", max_iterations=7, tolerance=1e-5, differentiable=False)
    assert (icp.icp_type, icp.max_iterations, icp.tolerance, icp.diff) == ("pt2pt", 7, 1e-5, False)
    assert icp.const_iter is False and icp.verbose is False and icp.target_pad_val == 1000
    assert icp.source_zeroes_are_pad is False and icp.match_ratio_thresh == 0.0
    assert icp.config["dICP"]["parameters"]["tanh_steepness"] == 5.0
    assert icp.nn.use_gumbel is False and icp.nn.eps == 1e-10 and icp.nn.tau == 0.1 and icp.nn.differentiable is False
    d = nn()
    assert d.differentiable and d.use_gumbel and d.eps == 1e-20 and d.tau == 0.1              # nn.py:5
    l = loss()
    assert (l.name, l.metric, l.differentiable, l.tanh_steepness) == ("huber", 1.0, False, 10.0)   # loss.py:4
    from dICP.ICP import ICP as Alias            # drop-in import path of the reference
    from dICP.visualization import plot_overlay  # noqa: F401  (tests/test_ICP.py:10)
    assert Alias is ICP


def test_handle_dimensions_matches_oracle():
    gen = torch.Generator().manual_seed(0)
    x3 = torch.rand((2, 9, 3), generator=gen)
    y6 = torch.rand((2, 11, 6), generator=gen)
    for x, y in ((x3, y6), (x3.transpose(1, 2), y6.transpose(1, 2)), (x3[0], y6[0]), (x3[0], y6[0].T),
                 (torch.rand((2, 6, 9), generator=gen), y6[:, :, :3])):
        a, b = nn._handle_dimensions(x, y)
        c, d = O.handle_dimensions(x, y)
        assert torch.equal(a, c) and torch.equal(b, d)
    with pytest.raises(IndexError):
        nn._handle_dimensions(x3[0].T, y6[0])        # nn.py:109 quirk
    with pytest.raises(AssertionError):
        nn._handle_dimensions(torch.rand(2, 9, 6), y6)   # nn.py:111


@pytest.mark.skipif(HAS_GPU, reason="checks the no-GPU behaviour")
def test_no_cpu_fallback(scan_map):
    """Without a HIP device every operator raises: there is no CPU compute path."""
    scan, mp = scan_map
    with pytest.raises(RuntimeError, match="no HIP device"):
        ICP().icp(t(scan), t(mp), torch.eye(4, dtype=torch.float64), dim=2)
    with pytest.raises(RuntimeError, match="no HIP device"):
        nn(differentiable=False).find_nn(t(scan[:, :3]), t(mp))
    with pytest.raises(RuntimeError, match="no HIP device"):
        loss("huber").get_weight(torch.rand(4, 3))
    with pytest.raises(ValueError):
        loss("tukey").get_weight(torch.rand(4, 3))


def test_graph_capture_refuses_host_drawn_gumbel_noise():
    """the Gumbel-softmax correspondence draws its per-iteration noise seeds on the host (ICPLoop.forward) and hands them to the kernels by
    value; a captured graph would replay one draw for ever.  graphed_icp / graphed_icp_step refuse it before anything touches a device."""
    import pytest
    from dicp_amd.ICP import ICP
    from dicp_amd.graphed import graphed_icp, graphed_icp_step
    icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=3, tolerance=1e-12)
    icp.const_iter = True
    icp.nn.use_gumbel = True
    x = torch.zeros((1, 8, 3))
    with pytest.raises(ValueError, match="Gumbel"):
        graphed_icp(icp, x, torch.zeros((1, 8, 6)), torch.eye(4).unsqueeze(0))
    with pytest.raises(ValueError, match="Gumbel"):
        graphed_icp_step(icp, lambda out: out["T"].sum(), x, torch.zeros((1, 8, 6)), torch.eye(4).unsqueeze(0))
    icp.const_iter = False
    with pytest.raises(ValueError, match="const_iter"):
        graphed_icp(icp, x, torch.zeros((1, 8, 6)), torch.eye(4).unsqueeze(0))


def test_resort_schedule_by_size(monkeypatch):
    """dicp_amd._loop.resort_schedule: an explicit schedule is taken as given; by default small calls without certificates re-order before iterations 0 and 1 only."""
    from dicp_amd import _loop
    monkeypatch.setattr(_loop, "CERT_MIN_WORK", 2.0e6)          # (the shipped threshold: tests/conftest.py runs this module with certificates at every size)
    assert _loop.resort_schedule((0, 2), 256, 16384, 10, True, None) == (0, 2)
    assert _loop.resort_schedule(None, 256, 16384, 10, True, None) == (0, 1, 2, 3)            # the headline shape: certificates pay, full schedule
    assert _loop.resort_schedule(None, 32, 4096, 10, True, None) == (0, 1)                     # configs[1]: 131072 points, no certificates at this size
    assert _loop.resort_schedule(None, 64, 8192, 10, True, None) == (0, 1, 2, 3)               # 524288 points: the full schedule stays best (profiles/r04_mid_size_resort.txt)
    assert _loop.resort_schedule(None, 32, 4096, 100, True, None) == (0, 1, 2, 3)              # a long call: 96 certified iterations x 131072 points pay for certificates
    assert _loop.resort_schedule(None, 32, 4096, 100, False, None) == (0, 1)                   # ... unless they are switched off
    assert _loop.resort_schedule(None, 1, 65, 2, True, None) == (0, 1)
    # a size where certificates do not pay under the full schedule (certifying search at iteration 3) but would under the short one (at 1) keeps the
    # full schedule: the short one was measured for calls without certificates only, and the loop derives the certificates from the schedule it is handed
    assert not _loop.certificates_pay(True, 13, 3, 49, 4096) and _loop.certificates_pay(True, 13, 1, 49, 4096)
    assert _loop.resort_schedule(None, 49, 4096, 13, True, None) == (0, 1, 2, 3)
