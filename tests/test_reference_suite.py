"""The reference's own test-suite scenarios, run the way a user of the reference runs them: through the
``dICP`` import path with CPU float64 tensors (the package computes on the MI355X and returns CPU tensors),
with the reference's assertions and tolerances.  One test here per test there:

    /root/reference/tests/test_ICP.py         :35 :80 :119
    /root/reference/tests/test_ICP_inputs.py  :36 :113 :157 :213 :254
    /root/reference/tests/test_nn.py          :12

pylgmath (absent here) is replaced by oracle/se3.py.  The unseeded random draws of the reference's tests are
seeded.  Needs a GPU: there is no CPU compute path.
"""
import numpy as np
import pytest
import torch

from dICP.ICP import ICP                       # the drop-in alias package
from dICP.nn import nn
from dICP.visualization import plot_overlay    # noqa: F401  (imported by the reference's tests, calls commented out there)
from oracle.se3 import tran2vec, vec2tran

pytestmark = pytest.mark.gpu
TOL_ICP, TOL_INPUTS = 1e-10, 1e-8                                  # the two files' `tolerance` fixtures


@pytest.fixture
def T_ts_true():
    return np.linalg.inv(vec2tran(np.array([1.0, 1.0, 0, 0, 0, 0.1])))      # test_ICP.py:45-47


def run_and_check(scan_map, T_ts_true, icp_type, differentiable, huber, check_grad):
    scan, mp = scan_map
    source = torch.tensor(scan[:, :3], requires_grad=True)
    target = torch.tensor(mp[:, :3] if icp_type == "pt2pt" else mp, requires_grad=True)
    T_init = torch.eye(4, dtype=source.dtype)
    icp = ICP(icp_type=icp_type, differentiable=differentiable, max_iterations=100, tolerance=TOL_ICP)
    res = icp.icp(source, target, T_init, trim_dist=5.0, loss_fn={"name": "huber", "metric": huber}, dim=2)
    err_T = tran2vec(T_ts_true @ np.linalg.inv(res["T"].detach().numpy()))
    assert np.linalg.norm(err_T) < TOL_ICP
    assert np.allclose(res["pc"].detach().numpy(), target[:, :3].detach().numpy(), atol=1e-5)
    if check_grad:
        res["T"].sum().backward()
        assert source.grad is not None and target.grad is not None
        assert not torch.isnan(source.grad).any() and not torch.isnan(target.grad).any()


def test_pt2pt_dICP(scan_map, T_ts_true):
    run_and_check(scan_map, T_ts_true, "pt2pt", True, 1.0, True)


def test_pt2pl_dICP(scan_map, T_ts_true):
    run_and_check(scan_map, T_ts_true, "pt2pl", True, 10.0, True)


def test_pt2pt_ICP(scan_map, T_ts_true):
    run_and_check(scan_map, T_ts_true, "pt2pt", False, 10.0, False)


def test_input_types(scan_map):
    scan, mp = scan_map
    rng = np.random.RandomState(0)
    source_1 = torch.cat((torch.tensor(scan[:50, :3]), torch.tensor(rng.rand(1, 3) * 1000)), dim=0)
    sources = [source_1, torch.tensor(scan[:, :3], requires_grad=True), torch.tensor(scan[:55, :3], requires_grad=True)]
    targets = [torch.tensor(mp[:55], requires_grad=True), torch.tensor(mp, requires_grad=True), torch.tensor(mp[:60], requires_grad=True)]
    T_inits = [torch.eye(4, dtype=torch.float64) for _ in range(3)]
    icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=25, tolerance=TOL_INPUTS)
    kw = dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=2)
    T_loop = np.zeros((3, 4, 4))
    ratio_loop = np.zeros(3)
    for i in range(3):
        r = icp.icp(sources[i], targets[i], T_inits[i], **kw)
        T_loop[i] = r["T"].detach().numpy()
        ratio_loop[i] = r["stats"]["matched_ratio"].item()
    batch = icp.icp(sources, targets, torch.stack(T_inits), **kw)
    err_T = tran2vec(T_loop @ np.linalg.inv(batch["T"].detach().numpy()))
    assert np.linalg.norm(err_T) < TOL_INPUTS
    assert np.linalg.norm(ratio_loop - batch["stats"]["matched_ratio"].detach().numpy()) < TOL_INPUTS


def test_zero_inputs(scan_map):
    scan, mp = scan_map
    sources = [torch.tensor(scan, requires_grad=True), [], []]
    targets = [[], torch.tensor(mp, requires_grad=True), []]
    T_stack = torch.stack([torch.eye(4, dtype=torch.float64)] * 3)
    icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=25, tolerance=TOL_INPUTS)
    T_loop = np.zeros((3, 4, 4))
    for i in range(3):
        T_loop[i] = icp.icp(sources[i], targets[i], T_stack[i], trim_dist=5.0, loss_fn=None, dim=2)["T"].detach().numpy()
    batch = icp.icp(sources, targets, T_stack, trim_dist=5.0, loss_fn=None, dim=2)["T"]
    assert np.linalg.norm(T_loop - T_stack.numpy()) < TOL_INPUTS
    assert np.linalg.norm(batch.detach().numpy() - T_stack.numpy()) < TOL_INPUTS


def test_weight_inputs(scan_map):
    scan, mp = scan_map
    rng = np.random.RandomState(1)
    sources = [torch.tensor(scan[:, :3], requires_grad=True), torch.tensor(scan[:, :3], requires_grad=True),
               torch.tensor(np.vstack((scan[:, :3], rng.rand(10, 3))), requires_grad=True)]
    targets = [torch.tensor(mp, requires_grad=True) for _ in range(3)]
    weights = [None, torch.tensor(np.ones(65), requires_grad=True),
               torch.tensor(np.hstack((np.ones(65), np.zeros(10))), requires_grad=True)]
    T_stack = torch.stack([torch.eye(4, dtype=torch.float64)] * 3)
    icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=25, tolerance=TOL_INPUTS)
    kw = dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=2)
    T_loop = np.zeros((3, 4, 4))
    for i in range(3):
        T_loop[i] = icp.icp(sources[i], targets[i], T_stack[i], weight=weights[i], **kw)["T"].detach().numpy()
    batch = icp.icp(sources, targets, T_stack, weight=weights, **kw)["T"].detach().numpy()
    assert np.linalg.norm(batch - T_loop) < TOL_INPUTS
    assert np.linalg.norm(T_loop[0] - T_loop[1]) < TOL_INPUTS and np.linalg.norm(T_loop[0] - T_loop[2]) < TOL_INPUTS


def test_diff_vs_nondiff_types(scan_map):
    scan, mp = scan_map
    source = torch.tensor(scan[:50, :3], requires_grad=True)
    target = torch.tensor(mp[:55], requires_grad=True)
    T_init = torch.eye(4, dtype=source.dtype)
    for loss_fn in ({"name": "huber", "metric": 1.0}, {"name": "cauchy", "metric": 0.5}):
        Ts = []
        for diff in (True, False):
            icp = ICP(icp_type="pt2pl", differentiable=diff, max_iterations=25, tolerance=TOL_INPUTS)
            Ts.append(icp.icp(source, target, T_init, trim_dist=5.0, loss_fn=loss_fn, dim=2)["T"].detach().numpy())
        assert np.linalg.norm(tran2vec(Ts[0] @ np.linalg.inv(Ts[1]))) < TOL_INPUTS


def test_padded_inputs(scan_map):
    scan, mp = scan_map
    source = torch.tensor(scan[:50, :3], requires_grad=True)
    target = torch.tensor(mp[:55], requires_grad=True)
    T_init = torch.eye(4, dtype=source.dtype)
    source_pad = torch.cat((source, torch.zeros((20, 3))))
    icp = ICP(icp_type="pt2pt", differentiable=False, max_iterations=25, tolerance=TOL_INPUTS)
    icp.source_zeroes_are_pad = True
    a = icp.icp(source, target, T_init, dim=2)["T"].detach().numpy()
    b = icp.icp(source_pad, target, T_init, dim=2)["T"].detach().numpy()
    assert np.linalg.norm(tran2vec(a @ np.linalg.inv(b))) < TOL_INPUTS


def test_diff_nn():
    """test_nn.py:12-41.  The default constructor selects the Gumbel path (nn.py:5); its exact `==` there is
    flaky by construction (~1.9 % of seeds), so the noise is seeded and the equality is held to
    float32 resolution."""
    torch.manual_seed(0)
    diff_nn = nn(differentiable=True)
    points = torch.tensor([(5.0, 4.0, 0.0), (2.0, 6.0, 0.0), (13.0, 3.0, 0.0), (8.0, 7.0, 0.0), (3.0, 1.0, 0.0)], requires_grad=True)
    query = torch.tensor([[9.0, 4.0, 0.0]], requires_grad=True)
    nearest1 = diff_nn.find_nn(query, points)
    assert torch.allclose(nearest1[0, 0], torch.tensor((8.0, 7.0, 0.0)), atol=1e-4)
    nearest1.sum().backward()
    assert query.grad is not None and points.grad is not None
    assert not torch.isnan(query.grad).any() and not torch.isnan(points.grad).any()
    points2 = torch.cat((points.detach(), torch.tensor((10.0, 2.0, 0.0)).view(1, -1)))
    nearest2 = diff_nn.find_nn(query, points2)
    assert torch.allclose(nearest2[0, 0], torch.tensor((10.0, 2.0, 0.0)), atol=1e-4)
