"""Truncated reverse sweep of the backward pass (`-m gpu`; include/dicp_hip.h, dicp_loop_buffers.bwd.skip).  Going backwards through the
iterations (the reverse of ICP.py:132-260, which the reference leaves to autograd), the chain of pose cotangents shrinks by orders of
magnitude per iteration near the pose; a cloud's sweep ends at the iteration from which on nothing -- this iteration's own contribution and
the most any earlier one could add, given the steps that were recorded -- reaches 2^-22 (float32) / 2^-40 (float64) of the cloud's largest
contribution: below the result's own rounding.  Checked here: same gradients as the full reverse sweep (bwd_skip_eps = 0) in every mode
and path, the sweeps really do end early, and the cases where nothing may be dropped."""
import numpy as np
import pytest
import torch

from dicp_amd import _lib
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs, make_scene_pairs

pytestmark = pytest.mark.gpu
DEV = "cuda"
KW = dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=3)


def npy(x):
    return x.detach().cpu().numpy()


def both(src, tgt, K, icp_type="pt2pl", kw=KW, weight=None, T0=None, diff=True, knn=_lib.KNN_AUTO, const_iter=True, tol=1e-12, loss_of=None, lists=None):
    N, dtype = src.shape[0], src.dtype
    outs = {}
    for eps in (0.0, None):
        icp = ICP(icp_type=icp_type, differentiable=diff, max_iterations=K, tolerance=tol)
        icp.const_iter, icp.knn_variant, icp.bwd_skip_eps = const_iter, knn, eps
        if lists is not None:
            S = [src[b, :lists[b]].to(DEV).requires_grad_(True) for b in range(N)]
            Tg = [tgt[b, :max(2048, lists[b] - 300)].to(DEV).requires_grad_(True) for b in range(N)]
            Ti, W = [torch.eye(4, dtype=dtype, device=DEV)] * N, None
        else:
            S, Tg = src.to(DEV).requires_grad_(True), tgt.to(DEV).requires_grad_(True)
            Ti = (torch.eye(4, dtype=dtype).repeat(N, 1, 1) if T0 is None else T0).to(DEV).requires_grad_(True)
            W = weight.to(DEV).requires_grad_(True) if weight is not None else None
        out = icp.icp(S, Tg, Ti, weight=W, **kw)
        loss = out["T"].sum() if loss_of is None else loss_of(out)
        loss.backward()
        grads = [torch.cat([x.grad.reshape(-1) for x in (S if lists is not None else [S])]), torch.cat([x.grad.reshape(-1) for x in (Tg if lists is not None else [Tg])])]
        if lists is None:
            grads.append(Ti.grad.reshape(-1))
            if W is not None:
                grads.append(W.grad.reshape(-1))
        outs[eps] = (out, grads, dict(icp.knn_stats))
    return outs[0.0], outs[None]


def same_gradients(full, skip, rtol):
    """Every gradient of the truncated sweep against the full one, to rtol of the gradient's own size."""
    assert "bwd_live" not in full[2] and "bwd_live" in skip[2]
    for key in ("T", "deltas", "weights", "costs", "pc"):
        assert torch.equal(full[0][key], skip[0][key]), key
    for i, (ga, gb) in enumerate(zip(full[1], skip[1])):
        scale = max(1e-30, float(ga[torch.isfinite(ga)].abs().max())) if bool(torch.isfinite(ga).any()) else 1.0
        assert bool((((ga - gb).abs() <= rtol * scale) | (torch.isnan(ga) & torch.isnan(gb))).all()), (i, float((ga - gb).abs().max() / scale))
    return skip[2]["bwd_live"].cpu()


def ended_early(live, K, N, at_most, stragglers=0.0):
    """live[k] = clouds at work in iteration k of the backward: all of them in the last iterations, (next to) none before the last `at_most`
    (the sweeps of different clouds end an iteration or two apart; `stragglers`: the share of earlier cloud-iterations that may still be at work)."""
    live = live[:K].tolist()
    assert live[K - 1] == N, live
    early = sum(live[:max(0, K - at_most)])
    assert early <= stragglers * N * max(1, K - at_most), live
    return K - next(k for k in range(K) if live[k] > 0)


@pytest.mark.parametrize("dtype,N,n,K,icp_type,knn", [(torch.float32, 40, 16384, 12, "pt2pl", _lib.KNN_AUTO), (torch.float64, 12, 8192, 14, "pt2pl", _lib.KNN_AUTO),
                                                       (torch.float32, 33, 4096, 10, "pt2pt", _lib.KNN_AUTO), (torch.float32, 6, 3000, 9, "pt2pl", _lib.KNN_VALU),
                                                       (torch.float64, 5, 300, 12, "pt2pt", _lib.KNN_AUTO), (torch.float32, 9, 200, 10, "pt2pl", _lib.KNN_AUTO)])
def test_skip_changes_no_gradient(dtype, N, n, K, icp_type, knn):
    """Sweep path (windowed backward), brute-force path (atomic backward) and the one-block-per-cloud small kernels, float32 and float64: the
    gradients of source, target and T_init equal the full reverse sweep's far below the parity bars, while only the last few iterations of
    the call did any per-point work."""
    src, tgt = make_pairs(N, n, n, seed=61, dtype=dtype)
    if icp_type == "pt2pt":
        tgt = tgt[:, :, :3].contiguous()
    full, skip = both(src, tgt, K, icp_type, knn=knn)
    live = same_gradients(full, skip, 2e-6 if dtype == torch.float32 else 1e-11)
    ended_early(live, K, N, 4 if dtype == torch.float32 else 6)


@pytest.mark.parametrize("icp_type,dim,loss,diff", [("pt2pl", 3, {"name": "cauchy", "metric": 0.5}, True), ("pt2pt", 2, {"name": "huber", "metric": 0.3}, True),
                                                     ("pt2pl", 2, {"name": "trim", "metric": 0.8}, True), ("pt2pl", 3, {"name": "trim", "metric": 0.8}, False),
                                                     ("pt2pt", 3, None, True), ("pt2pl", 3, None, False)])
def test_skip_in_every_mode(icp_type, dim, loss, diff):
    """Weight tensor and T_init with gradients, the other losses, planar mode, hard trim weights."""
    N, n, K = 12, 4096, 10
    src, tgt = make_pairs(N, n, n, seed=63)
    if icp_type == "pt2pt":
        tgt = tgt[:, :, :3].contiguous()
    g = torch.Generator().manual_seed(5)
    w0 = 0.5 + 0.5 * torch.rand((N, n), generator=g)
    T0 = torch.eye(4).repeat(N, 1, 1)
    T0[:, :3, 3] = 0.02 * torch.randn((N, 3), generator=g)
    full, skip = both(src, tgt, K, icp_type, kw=dict(trim_dist=5.0, loss_fn=loss, dim=dim), weight=w0, T0=T0, diff=diff)
    live = same_gradients(full, skip, 2e-6)
    if dim == 3:
        ended_early(live, K, N, 5)
    else:       # planar mode converges more slowly on these clouds (three of the six directions are never corrected): sweeps end late, some not at all
        assert int(live[0]) < N, live.tolist()


def test_skip_with_a_loss_on_the_transformed_cloud():
    """A loss on `pc` as well as on T: its pose gradient is dominated by a part that is no rotation at all (|C p + r|^2 does not change under
    rotations about the centroid), which passes through every iteration and whose float32 rounding keeps re-entering the chain -- the sweep may
    then end late or not at all; whatever it does, the gradients are the full sweep's, T_init's included."""
    N, n, K = 12, 8192, 10
    src, tgt = make_pairs(N, n, n, seed=73)
    for lossf in (lambda o: o["T"].sum() + (o["pc"] ** 2).sum() * 1e-3, lambda o: ((o["pc"] - 1.0) ** 2).mean()):
        full, skip = both(src, tgt, K, loss_of=lossf)
        same_gradients(full, skip, 2e-6)
    full, skip = both(src, tgt, K)
    live = same_gradients(full, skip, 2e-6)
    ended_early(live, K, N, 4)
    assert float(skip[1][2].abs().max()) > 0.1             # T.sum() depends on the 3x3 block in directions that are no rotation: T_init's gradient is not small


def test_hard_huber_is_never_skipped():
    """Hard Huber weights: the reference's gradient is NaN at an exactly zero residual whatever the cotangent (autograd differentiates both
    where() branches) -- the skip is off for that mode, every iteration takes part."""
    N, n = 6, 4096
    src, tgt = make_pairs(N, n, n, seed=65)
    icp = ICP(icp_type="pt2pl", differentiable=False, max_iterations=8, tolerance=1e-12)
    icp.const_iter = True
    S, Tg = src.to(DEV).requires_grad_(True), tgt.to(DEV).requires_grad_(True)
    icp.icp(S, Tg, torch.eye(4, device=DEV).repeat(N, 1, 1), **KW)["T"].sum().backward()
    assert "bwd_live" not in icp.knn_stats


def test_skip_keeps_what_matters():
    """Nothing may be skipped where it counts: (a) far from the pose and few iterations (no contraction yet) every iteration takes part and
    the gradients are the full ones; (b) a loss that involves only some clouds: the others carry an exactly zero cotangent, are skipped
    outright and get exactly zero gradients; (c) tolerance mode, clouds frozen at different iterations; (d) ragged lists; (e) planar scenes."""
    N, n = 16, 8192
    src, tgt = make_pairs(N, n, n, seed=67, max_rot=0.25, max_trans=1.5)
    full, skip = both(src, tgt, 3)
    live = same_gradients(full, skip, 2e-6)
    assert live[:3].tolist() == [N, N, N]
    src, tgt = make_pairs(N, n, n, seed=69)
    full, skip = both(src, tgt, 9, loss_of=lambda out: out["T"][::4].sum())
    live = same_gradients(full, skip, 2e-6)
    assert live[:9].tolist()[-1] == N // 4
    gs = skip[1][0].reshape(N, n, 3)
    assert float(gs[1].abs().max()) == 0.0 and float(gs[0].abs().max()) > 0.0
    full, skip = both(src, tgt, 40, const_iter=False, tol=1e-5)
    same_gradients(full, skip, 2e-6)
    assert full[0]["deltas"].shape[1] < 40
    lens = [n - (977 * b) % (n // 3) for b in range(N)]
    full, skip = both(src, tgt, 9, lists=lens)
    same_gradients(full, skip, 2e-6)
    s2, t2 = make_scene_pairs(N, n, n, seed=71)
    full, skip = both(s2, t2, 10)
    live = same_gradients(full, skip, 2e-6)
    ended_early(live, 10, N, 5, stragglers=0.15)


def test_skip_at_the_benchmark_size():
    """B = 256 x 16384, K = 12 (what bench.py times): the last iterations of the backward do the work, the gradients are the full sweep's."""
    B, n, K = 256, 16384, 12
    src, tgt = make_pairs(B, n, n, seed=3)
    full, skip = both(src, tgt, K, loss_of=lambda out: out["T"].sum())
    live = same_gradients(full, skip, 2e-6)
    assert ended_early(live, K, B, 4, stragglers=0.01) >= 2


def _tail_call(icp, src, tgt, weight=None):
    S, Tg = src.to(DEV).requires_grad_(True), tgt.to(DEV).requires_grad_(True)
    Ti = torch.eye(4, dtype=src.dtype).repeat(src.shape[0], 1, 1).to(DEV).requires_grad_(True)
    W = weight.to(DEV).requires_grad_(True) if weight is not None else None
    out = icp.icp(S, Tg, Ti, weight=W, **KW)
    out["T"].sum().backward()
    torch.cuda.synchronize()
    return [S.grad, Tg.grad, Ti.grad] + ([W.grad] if W is not None else [])


@pytest.mark.parametrize("dtype,N,n,K,icp_type,weights", [(torch.float32, 24, 16384, 12, "pt2pl", False), (torch.float64, 10, 8192, 14, "pt2pl", True),
                                                          (torch.float32, 33, 4096, 10, "pt2pt", True)])
def test_tail_launch_changes_no_gradient(dtype, N, n, K, icp_type, weights):
    """dicp_loop_buffers.bwd.tail_from: from the second call of a shape on, the iterations the previous call's sweeps had ended at are one launch.
    (i) placed by the previous call, it holds only ended clouds: the gradients equal those of the per-iteration launches (the pass-through of the
    pose cotangent to T_init.grad included); (ii) placed too late on purpose -- every iteration but the last in the one launch, all clouds still at
    work in it -- each cloud is swept by one block, in another order of summation: the same gradients to rounding."""
    src, tgt = make_pairs(N, n, n, seed=67, dtype=dtype)
    if icp_type == "pt2pt":
        tgt = tgt[:, :, :3].contiguous()
    w = (0.5 + torch.rand((N, n), dtype=dtype, generator=torch.Generator().manual_seed(5))) if weights else None

    def make(tail):
        icp = ICP(icp_type=icp_type, differentiable=True, max_iterations=K, tolerance=1e-12)
        icp.const_iter, icp._tuning["bwd_tail"] = True, tail
        return icp
    ref_icp = make(False)
    ref = _tail_call(ref_icp, src, tgt, w)
    assert ref_icp._hints.newest_tail is None and not ref_icp._hints.tail
    icp = make(True)
    first = _tail_call(icp, src, tgt, w)                    # no earlier call: every iteration is its own pair of launches
    assert icp.knn_stats["bwd_tail_from"] == 0
    tiny = 1e-6 if dtype == torch.float32 else 1e-13
    for a, b in zip(ref, first):                            # (the same launches; float atomics on the out-of-window rows: not bit for bit)
        assert float((a - b).abs().max()) <= tiny * float(a.abs().max())
    live_ref = ref_icp.knn_stats["bwd_live"][:K].tolist()
    second = _tail_call(icp, src, tgt, w)
    t_from = icp.knn_stats["bwd_tail_from"]
    assert 0 < t_from < K and t_from == next(k for k in range(K) if live_ref[k] * 8 >= N) - 1, (t_from, live_ref)
    live = icp.knn_stats["bwd_live"][:K].tolist()
    assert live == live_ref, (live, live_ref)               # the one launch decides like the per-iteration ones
    for i, (a, b) in enumerate(zip(ref, second)):
        scale = float(a.abs().max())
        assert float((a - b).abs().max()) <= tiny * scale, (i, float((a - b).abs().max()) / scale)
    # (ii) a hint that is wrong: "only the last iteration was at work"
    host, done = icp._hints.newest_tail[:2]
    done.synchronize()
    host.zero_()
    host[K - 1] = N
    third = _tail_call(icp, src, tgt, w)
    assert icp.knn_stats["bwd_tail_from"] == K - 2
    assert icp.knn_stats["bwd_live"][:K].tolist() == live_ref
    assert int(icp.knn_stats["bwd_tail_error"].item()) == 0
    rnd = 1e-5 if dtype == torch.float32 else 1e-12         # (the same arithmetic in the same order; float atomics on the out-of-window rows)
    for i, (a, b) in enumerate(zip(ref, third)):
        scale = float(a.abs().max())
        assert float((a - b).abs().max()) <= rnd * scale, (i, float((a - b).abs().max()) / scale)
    fourth = _tail_call(icp, src, tgt, w)                   # ... and the hint has corrected itself
    assert icp.knn_stats["bwd_tail_from"] == t_from


def test_tail_launch_on_ragged_lists_and_in_tolerance_mode():
    """The one-launch tail with the clouds' own lengths (pad slots carry no work in window_body, inside the tail as outside it), and a tolerance-mode
    call (several runs of the backward: the tail belongs to the run that reaches iteration 0): gradients equal the per-iteration launches'."""
    N, n = 10, 12288
    src, tgt = make_pairs(N, n, n, seed=71)
    lens = [n - 517 * b for b in range(N)]
    for mode in ("ragged", "tolerance"):
        grads = {}
        for tail in (False, True):
            icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=12 if mode == "ragged" else 30, tolerance=1e-12 if mode == "ragged" else 1e-5)
            icp.const_iter, icp._tuning["bwd_tail"] = mode == "ragged", tail
            for rep in range(3):        # (the later calls have the earlier ones' hints; the third one's is made to place the tail high)
                if tail and rep == 2:
                    entry = icp._hints.newest_tail
                    entry[1].synchronize()
                    K_run = entry[2][2]
                    entry[0].zero_()
                    entry[0][K_run - 1] = N
                if mode == "ragged":
                    S = [src[b, :lens[b]].to(DEV).requires_grad_(True) for b in range(N)]
                    Tg = [tgt[b, :max(2048, lens[b] - 300)].to(DEV).requires_grad_(True) for b in range(N)]
                    out = icp.icp(S, Tg, [torch.eye(4, device=DEV)] * N, **KW)
                else:
                    S, Tg = [src.to(DEV).requires_grad_(True)], [tgt.to(DEV).requires_grad_(True)]
                    out = icp.icp(S[0], Tg[0], torch.eye(4, device=DEV).repeat(N, 1, 1), **KW)
                out["T"].sum().backward()
                torch.cuda.synchronize()
            if tail:
                assert icp.knn_stats["bwd_tail_from"] > 0 and int(icp.knn_stats["bwd_tail_error"].item()) == 0
            grads[tail] = [torch.cat([x.grad.reshape(-1) for x in S]), torch.cat([x.grad.reshape(-1) for x in Tg])]
        for a, b in zip(grads[False], grads[True]):
            assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max()), mode


def test_tail_launch_only_where_a_clouds_blocks_are_all_resident():
    """The one-launch tail lets a cloud's blocks wait for each other, which is only safe while they are all resident (dispatch is in index order and
    a cloud's blocks share an XCD): beyond dicp_bwd_tail_max_blocks blocks per cloud -- half of what an XCD holds of that kernel -- the host does not
    place it, whatever the hint says (float64 clouds of 32768 points have 43 blocks, an XCD holds 32), and the library refuses it."""
    import ctypes
    lib = _lib.load()
    cap32, cap64 = lib.dicp_bwd_tail_max_blocks(_lib.F32), lib.dicp_bwd_tail_max_blocks(_lib.F64)
    assert 16 <= cap32 <= 128 and 8 <= cap64 <= cap32, (cap32, cap64)
    N, n, K = 9, 32768, 8
    assert lib.dicp_window_blocks(_lib.F64, n, n) > cap64
    src, tgt = make_pairs(N, n, n, seed=91, dtype=torch.float64)
    icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=1e-12)
    icp.const_iter = True
    first = _tail_call(icp, src, tgt)
    host, done = icp._hints.newest_tail[:2]
    done.synchronize()
    host.zero_()
    host[K - 1] = N                                         # "only the last iteration was at work": asks for the tail at its highest
    second = _tail_call(icp, src, tgt)
    assert icp.knn_stats["bwd_tail_from"] == 0              # ... and does not get it
    for a, b in zip(first, second):
        assert float((a - b).abs().max()) <= 1e-12 * float(a.abs().max())
    icp.check_errors()
    # the same shape in float32 (32 blocks <= the cap) does take it, stragglers and all
    if lib.dicp_window_blocks(_lib.F32, n, n) <= cap32:
        icp32 = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=1e-12)
        icp32.const_iter = True
        a32 = _tail_call(icp32, src.float(), tgt.float())
        host, done = icp32._hints.newest_tail[:2]
        done.synchronize()
        host.zero_()
        host[K - 1] = N
        b32 = _tail_call(icp32, src.float(), tgt.float())
        assert icp32.knn_stats["bwd_tail_from"] == K - 2 and int(icp32.knn_stats["bwd_tail_error"].item()) == 0
        for a, b in zip(a32, b32):
            assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max())
        icp32.check_errors()


def test_a_wait_that_ran_out_is_raised_not_returned():
    """The error word of the tail launch travels with the live counters; a record that reports one makes the NEXT backward pass (and
    ICP.check_errors) raise _loop.TailTimeout.  (The wait itself cannot be made to run out on a healthy GPU: the word is raised by hand.)"""
    from dicp_amd._loop import TailTimeout
    N, n, K = 12, 8192, 8
    src, tgt = make_pairs(N, n, n, seed=93)
    icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=1e-12)
    icp.const_iter = True
    _tail_call(icp, src, tgt)
    _tail_call(icp, src, tgt)
    icp.check_errors()                                      # nothing to report
    rec = icp._hints.newest_tail
    rec[1].synchronize()
    rec[0][rec[3]] = 1                                      # what bwd_tail_kernel raises next to bwd_tail_arrive[N]
    rec[4] = False
    with pytest.raises(TailTimeout):
        _tail_call(icp, src, tgt)
    rec[4] = False
    with pytest.raises(TailTimeout):
        icp.check_errors()
    rec[0][rec[3]] = 0
    rec[4] = False
    _tail_call(icp, src, tgt)


def test_strict_errors_raise_in_the_failing_pass(monkeypatch):
    """ICP.strict_errors: the pass that used the tail waits for itself and raises before it returns a gradient (the error word is raised by hand: the
    library's memset of the pass's workspace is followed by a fill of that word, on the same stream, before the pass's kernels)."""
    from dicp_amd import _loop
    from dicp_amd._loop import TailTimeout
    N, n, K = 12, 8192, 8
    src, tgt = make_pairs(N, n, n, seed=93)
    icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=1e-12)
    icp.const_iter, icp.strict_errors = True, True
    _tail_call(icp, src, tgt)
    _tail_call(icp, src, tgt)                               # (strict and healthy: nothing raised)
    assert icp.knn_stats["bwd_tail_from"] > 0
    real = _loop._strict_tail_check
    seen = []

    def raised(cfg, word):
        if word is not None:
            word.fill_(1)                                   # what bwd_tail_kernel does when a wait runs out
            seen.append(1)
        return real(cfg, word)
    monkeypatch.setattr(_loop, "_strict_tail_check", raised)
    S = src.to(DEV).requires_grad_(True)
    Tg = tgt.to(DEV).requires_grad_(True)
    out = icp.icp(S, Tg, torch.eye(4, device=DEV).repeat(N, 1, 1), trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0})
    with pytest.raises(TailTimeout):
        out["T"].sum().backward()
    assert seen and S.grad is None and Tg.grad is None      # no gradient was handed to the caller
    icp.strict_errors = False
    monkeypatch.setattr(_loop, "_strict_tail_check", real)
    _tail_call(icp, src, tgt)


def test_two_backward_passes_on_two_streams_beside_a_busy_gpu():
    """Two ICP objects run forward + backward (tails placed, with stragglers: hints wrong on purpose) concurrently on two streams while a third
    stream keeps the GPU full of other work: same gradients as alone, no wait ran out."""
    N, n, K = 24, 16384, 10
    pairs = [make_pairs(N, n, n, seed=97 + i) for i in range(2)]
    icps, alone = [], []
    for src, tgt in pairs:
        icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=1e-12)
        icp.const_iter = True
        _tail_call(icp, src, tgt)
        alone.append(_tail_call(icp, src, tgt))
        icps.append(icp)
    streams = [torch.cuda.Stream() for _ in range(3)]
    busy_in = torch.randn((4096, 4096), device=DEV)
    results = [None, None]
    torch.cuda.synchronize()
    for rep in range(3):
        with torch.cuda.stream(streams[2]):                 # the third kernel(s): a chain of GEMMs that fills every CU for the duration
            acc = busy_in
            for _ in range(12):
                acc = torch.tanh(acc @ busy_in * 1e-3)
        for i, (src, tgt) in enumerate(pairs):
            with torch.cuda.stream(streams[i]):
                # per stream the hints start afresh (they are kept per device, stream and shape): two calls, the second with a wrong hint
                for call in range(2):
                    S, Tg = src.to(DEV).requires_grad_(True), tgt.to(DEV).requires_grad_(True)
                    Ti = torch.eye(4).repeat(N, 1, 1).to(DEV).requires_grad_(True)
                    out = icps[i].icp(S, Tg, Ti, **KW)
                    out["T"].sum().backward()
                    if call == 0:
                        streams[i].synchronize()
                        rec = icps[i]._hints.newest_tail
                        rec[1].synchronize()
                        rec[0].zero_()
                        rec[0][K - 1] = N
                results[i] = [S.grad, Tg.grad, Ti.grad]
        torch.cuda.synchronize()
        for i in range(2):
            assert icps[i].knn_stats["bwd_tail_from"] == K - 2 and int(icps[i].knn_stats["bwd_tail_error"].item()) == 0
            icps[i].check_errors()
            for a, b in zip(alone[i], results[i]):
                assert bool(torch.isfinite(b).all())
                assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max())
    del acc
