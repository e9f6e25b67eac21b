"""One eager call behind one library call per direction (`-m gpu`; include/dicp_hip.h dicp_call_*, dicp_amd/_call.py) against the per-buffer
sequence it replaces for mid-size calls (dicp_amd/_ops.py ICPLoop: ICP.py:49-303 for a dense batch, constant iteration count, sorted sweep).
Same kernels with the same arguments, so: every result bit for bit; the gradients to a few units of rounding (the slot order of the reverse sweep's
sums comes from a counting sort whose order inside a bucket is the hardware's, and the out-of-window rows of the target gradient are added with
float atomics: two passes of EITHER path differ by that much).  And: which calls take it, which do not."""
import pytest
import torch

from dicp_amd import _call, _lib, _loop, _ops
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs, make_scene_pairs

pytestmark = pytest.mark.gpu
DEV = "cuda"


def run(one_call, src, tgt, K, icp_type, kw, weight, T0, dtype, grads=True, loss_of=None, counts=None, knn=_lib.KNN_SWEEP):
    icp = ICP(icp_type=icp_type, differentiable=True, max_iterations=K, tolerance=1e-12)
    icp.const_iter, icp.knn_variant = True, knn
    icp._tuning["one_call"] = one_call
    N = src.shape[0]
    S = src.to(DEV).requires_grad_(grads)
    Tg = tgt.to(DEV).requires_grad_(grads)
    Ti = (torch.eye(4, dtype=dtype).repeat(N, 1, 1) if T0 is None else T0).to(DEV).requires_grad_(grads)
    W = weight.to(DEV).requires_grad_(grads) if weight is not None else None
    taken = []
    real = _call.CallLoop.apply
    _call.CallLoop.apply = staticmethod(lambda *a: (taken.append(1), real(*a))[1])
    try:
        outs, gs = [], []
        for rep in range(2):        # twice: the second backward pass has the first one's hint for its one-launch tail
            for t in (S, Tg, Ti, W):
                if t is not None:
                    t.grad = None
            out = icp.icp(S, Tg, Ti, weight=W, **kw)
            if grads:
                (out["T"].sum() if loss_of is None else loss_of(out)).backward()
                torch.cuda.synchronize()
            outs.append(out)
            gs.append([t.grad.clone() if t is not None and t.grad is not None else None for t in (S, Tg, Ti, W)])
    finally:
        _call.CallLoop.apply = real
    if counts is not None:
        counts.append(len(taken))
    return outs, gs, dict(icp.knn_stats)


CASES = {
    "pt2pl-f32": dict(icp_type="pt2pl", dtype=torch.float32, kw=dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=3), K=6),
    "pt2pt-f32-weights": dict(icp_type="pt2pt", dtype=torch.float32, kw=dict(trim_dist=2.0, loss_fn={"name": "cauchy", "metric": 0.5}, dim=3), K=5, weights=True),
    "pt2pl-f64-dim2": dict(icp_type="pt2pl", dtype=torch.float64, kw=dict(trim_dist=None, loss_fn=None, dim=2), K=4, weights=True),
    "pt2pt-f64-pose": dict(icp_type="pt2pt", dtype=torch.float64, kw=dict(trim_dist=5.0, loss_fn={"name": "trim", "metric": 1.0}, dim=3), K=7, pose=True),
    "one-iteration": dict(icp_type="pt2pl", dtype=torch.float32, kw=dict(trim_dist=5.0, loss_fn=None, dim=3), K=1),
    "pc-cotangent": dict(icp_type="pt2pl", dtype=torch.float32, kw=dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=3), K=5, pc_loss=True),
    "scene-f32": dict(icp_type="pt2pl", dtype=torch.float32, kw=dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=3), K=10, scene=True),
}


@pytest.mark.parametrize("name", list(CASES))
def test_one_call_equals_the_per_buffer_loop(name):
    case = CASES[name]
    dtype = case["dtype"]
    N, n, m = (4, 3000, 2500) if not case.get("scene") else (6, 4096, 4096)
    make = make_scene_pairs if case.get("scene") else make_pairs
    src, tgt = make(N, n, m, seed=11, dtype=dtype)
    if case["icp_type"] == "pt2pt":
        tgt = tgt[:, :, :3].contiguous()
    weight = (torch.rand((N, n), generator=torch.Generator().manual_seed(5), dtype=torch.float64) > 0.1).to(dtype) * 0.9 if case.get("weights") else None
    T0 = None
    if case.get("pose"):
        T0 = torch.eye(4, dtype=dtype).repeat(N, 1, 1)
        T0[:, :3, 3] = torch.tensor([0.05, -0.02, 0.01], dtype=dtype)
    loss_of = (lambda out: (out["pc"] ** 2).sum() * 1e-3 + out["T"].sum()) if case.get("pc_loss") else None
    counts = []
    a = run(True, src, tgt, case["K"], case["icp_type"], case["kw"], weight, T0, dtype, loss_of=loss_of, counts=counts)
    b = run(False, src, tgt, case["K"], case["icp_type"], case["kw"], weight, T0, dtype, loss_of=loss_of, counts=counts)
    assert counts == [2, 0]                                   # the first object took the one-call path both times, the second never
    for rep in range(2):
        oa, ob = a[0][rep], b[0][rep]
        for key in ("T", "pc", "costs", "deltas", "weights"):
            assert oa[key].shape == ob[key].shape and torch.equal(oa[key], ob[key]), (key, rep)
        for key in ("converged", "iterations", "matched_ratio"):
            assert torch.equal(oa["stats"][key], ob["stats"][key]), key
        for i, (ga, gb) in enumerate(zip(a[1][rep], b[1][rep])):
            assert (ga is None) == (gb is None), i
            if ga is None:
                continue
            scale = float(gb.abs().max())
            assert float((ga - gb).abs().max()) <= (1e-5 if dtype == torch.float32 else 1e-13) * max(scale, 1e-30), (i, rep, float((ga - gb).abs().max()), scale)
    # the books: pairs scored, where the reverse sweeps ended, the tail of the second pass
    pa, pb = int(a[2]["knn_pairs"].sum()), int(b[2]["knn_pairs"].sum())          # (the query order inside a bucket is the hardware's: a few units of the sweep more or less)
    assert pa > 0 and abs(pa - pb) <= 0.01 * pb
    assert torch.equal(a[2]["bwd_live"].cpu(), b[2]["bwd_live"].cpu())
    assert a[2]["bwd_tail_from"] == b[2]["bwd_tail_from"]


def test_no_gradients_wanted():
    src, tgt = make_pairs(3, 2048, 2048, seed=3, dtype=torch.float32)
    kw = dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=3)
    counts = []
    a = run(True, src, tgt, 5, "pt2pl", kw, None, None, torch.float32, grads=False, counts=counts)
    b = run(False, src, tgt, 5, "pt2pl", kw, None, None, torch.float32, grads=False, counts=counts)
    assert counts == [2, 0]
    for key in ("T", "pc", "costs", "deltas", "weights"):
        assert torch.equal(a[0][1][key], b[0][1][key]), key
    assert not a[0][1]["T"].requires_grad


def test_which_calls_take_it():
    """Only the calls ICPLoop would run without a host decision: not tolerance mode, not ragged lists, not the brute-force searches, not a
    shape whose match certificates pay, not a graph capture."""
    src, tgt = make_pairs(2, 1024, 1024, seed=4, dtype=torch.float32)
    kw = dict(trim_dist=5.0, loss_fn=None, dim=3)
    S, Tg, Ti = src.to(DEV), tgt.to(DEV), torch.eye(4).repeat(2, 1, 1).to(DEV)

    def cfg_of(icp):
        return _loop.LoopConfig(icp_type="pt2pl", differentiable=True, max_iterations=int(icp.max_iterations), tolerance=1e-12, trim_dist=5.0, loss_name=None,
                               loss_metric=1.0, dim=3, const_iter=bool(icp.const_iter), tanh_steepness=5.0, match_ratio_thresh=0.5, knn_variant=icp.knn_variant,
                               reuse_matches=bool(icp.reuse_matches))
    icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=5, tolerance=1e-12)
    icp.const_iter, icp.knn_variant = True, _lib.KNN_SWEEP
    assert _call.eligible(cfg_of(icp), S, Tg, Ti, None, True)
    icp.const_iter = False
    assert not _call.eligible(cfg_of(icp), S, Tg, Ti, None, True)
    icp.const_iter, icp.knn_variant = True, _lib.KNN_VALU
    assert not _call.eligible(cfg_of(icp), S, Tg, Ti, None, True)
    icp.knn_variant = _lib.KNN_AUTO            # 2 x 1024 x 1024: the brute-force search
    assert not _call.eligible(cfg_of(icp), S, Tg, Ti, None, True)
    icp.knn_variant = _lib.KNN_SWEEP
    cfg = cfg_of(icp)
    cfg.src_rows = torch.tensor([1024, 1000], dtype=torch.int32, device=DEV)
    assert not _call.eligible(cfg, S, Tg, Ti, None, True)
    assert not _call.eligible(cfg_of(icp), S, Tg[:, :, :3], Ti, None, True)               # pt2pl wants the normals' columns; a slice is not contiguous either
    assert not _call.eligible(cfg_of(icp), S.double(), Tg, Ti, None, True)
    big = ICP(icp_type="pt2pl", differentiable=True, max_iterations=20, tolerance=1e-12)
    big.const_iter, big.knn_variant = True, _lib.KNN_SWEEP
    Sb = torch.empty((64, 4096, 3), device=DEV)
    Tb = torch.empty((64, 4096, 6), device=DEV)
    assert not _call.eligible(cfg_of(big), Sb, Tb, torch.eye(4, device=DEV).repeat(64, 1, 1), None, True)       # 16 x 64 x 4096 certified point-iterations: certificates pay
    big.max_iterations = 6
    assert _call.eligible(cfg_of(big), Sb, Tb, torch.eye(4, device=DEV).repeat(64, 1, 1), None, True)
    # a list of clouds of different lengths goes through ICPLoop (its pad rows are the loop's business), and gives what it always gave
    out = ICP(icp_type="pt2pl", differentiable=True, max_iterations=3, tolerance=1e-12)
    out.const_iter, out.knn_variant = True, _lib.KNN_SWEEP
    res = out.icp([S[0], S[1][:900]], [Tg[0], Tg[1]], [Ti[0], Ti[1]], **kw)
    assert res["T"].shape == (2, 4, 4) and bool(torch.isfinite(res["T"]).all())


def test_kept_results_keep_only_themselves_and_an_edited_step_history_is_refused():
    """the non-differentiable results are views of a small allocation of their own -- a log of the costs of every training step must not keep
    the call's workspace (search structure, match history, sort scratch) alive; and `deltas`, which the reverse sweep reads, is saved through autograd:
    an in-place edit of it is refused instead of silently giving wrong gradients."""
    src, tgt = make_pairs(8, 4096, 4096, seed=8, dtype=torch.float32)
    icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=6, tolerance=1e-12)
    icp.const_iter, icp.knn_variant = True, _lib.KNN_SWEEP
    Tg, Ti = tgt.to(DEV), torch.eye(4).repeat(8, 1, 1).to(DEV)
    log = []
    torch.cuda.synchronize()
    base = None
    for step in range(6):
        S = src.to(DEV).requires_grad_(True)
        out = icp.icp(S, Tg, Ti, trim_dist=5.0, loss_fn=None, dim=3)
        out["T"].sum().backward()
        log.append(out["costs"])                        # what a training loop keeps per step
        log.append(out["stats"]["iterations"])
        del out, S
        torch.cuda.synchronize()
        if step == 1:
            base = torch.cuda.memory_allocated()
    grown = torch.cuda.memory_allocated() - base
    assert grown < 4 * (1 << 20), grown                 # four more steps' logs: kilobytes, not four workspaces of ~5 MB (+ 0.8 MB of weights each)
    S = src.to(DEV).requires_grad_(True)
    out = icp.icp(S, Tg, Ti, trim_dist=5.0, loss_fn=None, dim=3)
    out["deltas"].mul_(2.0)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        out["T"].sum().backward()


def test_results_outlive_the_call_and_an_edited_result_does_not_block_the_backward():
    src, tgt = make_pairs(2, 2048, 2048, seed=8, dtype=torch.float32)
    icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=4, tolerance=1e-12)
    icp.const_iter, icp.knn_variant = True, _lib.KNN_SWEEP
    S, Tg = src.to(DEV).requires_grad_(True), tgt.to(DEV)
    Ti = torch.eye(4).repeat(2, 1, 1).to(DEV)
    out = icp.icp(S, Tg, Ti, trim_dist=5.0, loss_fn=None, dim=3)
    keep = {k: out[k].clone() for k in ("T", "pc", "weights", "costs", "deltas")}
    out["weights"].zero_()                              # a caller's own bookkeeping on a non-differentiable result
    filler = [torch.full((1 << 20,), 7.0, device=DEV) for _ in range(8)]      # allocations that would land in a freed workspace
    out["T"].sum().backward()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(S.grad).all()) and float(S.grad.abs().max()) > 0
    for k in ("T", "pc", "costs", "deltas"):
        assert torch.equal(out[k], keep[k]), k
    del filler


def test_the_c_abi_alone():
    """INTEGRATION.md's stub: dicp_call_plan + dicp_call_forward bound with ctypes and nothing else of the package's host side -- the poses and steps
    ICP.icp returns for the same call, bit for bit."""
    import ctypes
    lib = _lib.load()
    N, n, m, K = 3, 2048, 2304, 5
    src, tgt = make_pairs(N, n, m, seed=21, dtype=torch.float32)
    S, Tg, Ti = src.to(DEV), tgt.to(DEV), torch.eye(4).repeat(N, 1, 1).to(DEV)
    call = _lib.Call(src=S.data_ptr(), tgt=Tg.data_ptr(), T_init=Ti.data_ptr(), w0=None, N=N, n=n, m=m, c=6, K=K, dim=3, need_grad=0, n_resort=3, flags=_lib.CALL_FIRST_SEARCH,
                     directions=int(_ops.FRAME_DIRECTIONS), quantum=_ops.CENTER_QUANTUM, tolerance=1e-12)
    for i in range(3):
        call.resort[i] = i + 1
    lay = _lib.CallLayout()
    assert lib.dicp_call_plan(_lib.F32, ctypes.byref(call), ctypes.byref(lay)) == 0
    ws = torch.empty(lay.total // 4, dtype=torch.float32, device=DEV)
    rs = torch.empty(lay.results_total // 4, dtype=torch.float32, device=DEV)
    call.workspace, call.results = ws.data_ptr(), rs.data_ptr()
    icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=1e-12)
    icp.const_iter, icp.knn_variant = True, _lib.KNN_SWEEP
    P = _lib.WeightParams(mode=_lib.PT2PL, trim_on=1, differentiable=1, loss=_lib.LOSS_HUBER, trim_dist=5.0, tanh_k=float(icp.config['dICP']['parameters']['tanh_steepness']),
                          loss_delta=1.0, match_thresh=float(icp.match_ratio_thresh))
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.dicp_call_forward(_lib.F32, ctypes.byref(P), ctypes.byref(call), st) == 0
    torch.cuda.synchronize()
    T = ws[lay.T // 4: lay.T // 4 + 16 * N].view(N, 4, 4)
    deltas = rs[lay.deltas // 4: lay.deltas // 4 + 6 * K * N].view(N, K, 6)
    with torch.no_grad():
        out = icp.icp(S, Tg, Ti, trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=3)
    assert torch.equal(T, out["T"]) and torch.equal(deltas, out["deltas"][..., 0])
    # a misaligned workspace is refused before anything is launched
    call.workspace = ws.data_ptr() + 16
    assert lib.dicp_call_forward(_lib.F32, ctypes.byref(P), ctypes.byref(call), st) == 5


@pytest.mark.parametrize("dtype,K,weights,seed_T", [(torch.float32, 10, False, False), (torch.float64, 4, True, False), (torch.float32, 1, False, True), (torch.float32, 2, True, False)])
def test_svd_step_one_call_equals_the_per_buffer_loop(dtype, K, weights, seed_T):
    """ICP.pt2pt_dICP_SVD (ICP.py:533-591) through dicp_kabsch_call_* against KabschLoop + transform_points: pose, cloud, costs, iterations and the gradients
    w.r.t. source, target and weight.  (The two differ in ONE rounding: the first search pose is a kernel's here and torch matmuls there; with the frame
    at the origin -- these clouds -- both are exact copies of the start pose.)"""
    N, n, m = 5, 3000, 2800
    src, tgt = make_pairs(N, n, m, seed=31, dtype=dtype)
    w = ((torch.rand((N, n), generator=torch.Generator().manual_seed(9), dtype=torch.float64) > 0.05).to(dtype) * 0.8) if weights else None
    T0 = torch.eye(4, dtype=dtype).repeat(N, 1, 1)
    T0[:, :3, 3] = torch.tensor([0.04, 0.01, -0.03], dtype=dtype)
    res = []
    for one in (True, False):
        icp = ICP(icp_type="pt2pt", differentiable=True, max_iterations=K, tolerance=1e-12)
        icp.const_iter, icp.knn_variant, icp.svd_seed_T_init = True, _lib.KNN_SWEEP, seed_T
        icp._tuning["one_call"] = one
        taken = []
        real = _call.KabschCall.apply
        _call.KabschCall.apply = staticmethod(lambda *a: (taken.append(1), real(*a))[1])
        try:
            S, Tg = src.to(DEV).requires_grad_(True), tgt.to(DEV).requires_grad_(True)
            W = w.to(DEV).requires_grad_(True) if w is not None else None
            pc, T = icp.pt2pt_dICP_SVD(S, Tg, T0.to(DEV), trim_dist=3.0, weight=W)
            (T.sum() + 1e-3 * (pc ** 2).sum()).backward()
            torch.cuda.synchronize()
        finally:
            _call.KabschCall.apply = real
        assert len(taken) == (1 if one else 0)
        res.append((pc.detach(), T.detach(), icp.svd_stats["costs"].clone(), icp.svd_stats["iterations"].clone(), S.grad, Tg.grad, W.grad if W is not None else None))
    a, b = res
    assert a[0].shape == b[0].shape == (N, n, 3) and a[2].shape == b[2].shape == (N, K)
    tol = 2e-6 if dtype == torch.float32 else 1e-13
    for i in (0, 1, 2):
        assert float((a[i] - b[i]).abs().max()) <= tol * max(1.0, float(b[i].abs().max())), i
    assert torch.equal(a[3], b[3])
    for i in (4, 5, 6):
        assert (a[i] is None) == (b[i] is None)
        if a[i] is not None:
            assert float((a[i] - b[i]).abs().max()) <= (2e-5 if dtype == torch.float32 else 1e-12) * max(1e-30, float(b[i].abs().max())), i


def test_svd_step_which_calls_take_it():
    src, tgt = make_pairs(2, 1024, 1024, seed=4, dtype=torch.float32)
    S, Tg, Ti, W = src.to(DEV), tgt[:, :, :3].contiguous().to(DEV), torch.eye(4).repeat(2, 1, 1).to(DEV), torch.ones((2, 1024), device=DEV)
    assert _call.kabsch_eligible(S, Tg, Ti, W, True, _lib.KNN_SWEEP, None, None)
    assert not _call.kabsch_eligible(S, Tg, Ti, W, False, _lib.KNN_SWEEP, None, None)                  # tolerance mode: the host reads the counters between segments
    assert not _call.kabsch_eligible(S, Tg, Ti, W, True, _lib.KNN_AUTO, None, None)                    # 2 x 1024 x 1024: brute force
    assert not _call.kabsch_eligible(S, Tg, Ti, W, True, _lib.KNN_SWEEP, torch.tensor([1024, 1000], dtype=torch.int32, device=DEV), None)
    assert not _call.kabsch_eligible(S, Tg, Ti.double(), W, True, _lib.KNN_SWEEP, None, None)


@pytest.mark.parametrize("icp_type", ["pt2pl", "pt2pt"])
def test_one_call_against_the_oracle(icp_type):
    """The one-call path held to the CPU restatement of the reference directly (float64: poses, steps, weights 1e-10; gradients 1e-9 of their size)."""
    from oracle import dicp_oracle as O
    N, n, m, K = 3, 2500, 2600, 5
    src, tgt = make_pairs(N, n, m, seed=41, dtype=torch.float64)
    if icp_type == "pt2pt":
        tgt = tgt[:, :, :3].contiguous()
    kw = dict(trim_dist=5.0, loss_fn={"name": "cauchy", "metric": 0.7}, dim=3)
    T0 = torch.eye(4, dtype=torch.float64).repeat(N, 1, 1)
    counts = []
    outs, gs, _ = run(True, src, tgt, K, icp_type, kw, None, None, torch.float64, counts=counts)
    assert counts == [2]
    s_c, t_c = src.clone().requires_grad_(True), tgt.clone().requires_grad_(True)
    ref = O.icp_batched(s_c, t_c, T0, torch.ones(N, n * (3 if icp_type == "pt2pt" else 1), dtype=torch.float64), icp_type=icp_type, differentiable=True, max_iterations=K,
                        tolerance=1e-12, const_iter=True, tanh_steepness=5.0, **kw)
    ref["T"].sum().backward()
    out = outs[1]
    assert float((out["T"].detach().cpu() - ref["T"].detach()).abs().max()) < 1e-10
    assert float((out["deltas"].detach().cpu() - ref["deltas"].detach()).abs().max()) < 1e-10
    assert float((out["weights"].detach().cpu() - ref["weights"].detach()).abs().max()) < 1e-9
    assert float((out["pc"].detach().cpu() - ref["pc"].detach()).abs().max()) < 1e-9
    for got, want in ((gs[1][0].cpu(), s_c.grad), (gs[1][1].cpu(), t_c.grad)):
        assert float((got - want).abs().max()) <= 1e-9 * max(1.0, float(want.abs().max()))
