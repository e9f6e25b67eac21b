"""What an ICP object carries from call to call (CallHints: where the reverse sweep's one-launch tail starts, certificate pauses, the scoring forms' plan, the
"clouds keep moving" flag) is about TIME only (`-m gpu`).  A training loop hands such an object NEW clouds of the same shape every step, so every hint it acts on was
recorded on other data: here one object is called in rotation with batches of five different kinds, and every call is held to a hint-free object's call on the same
batch -- the forward's results bit for bit, the gradients to the rounding of sums taken in another order.  bench.py's `value_fresh_inputs` times the same rotation."""
import pytest
import torch

from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs, make_scene_pairs, make_independent_pairs

DEV = "cuda"
pytestmark = pytest.mark.gpu
KW = dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0}, dim=3)


def call(icp, S, T, T0):
    s, t = S.detach().requires_grad_(True), T.detach().requires_grad_(True)
    o = icp.icp(s, t, T0, **KW)
    o["T"].sum().backward()
    torch.cuda.synchronize()
    return o, s.grad, t.grad


@pytest.mark.parametrize("const_iter,K", [(True, 10), (False, 40)])
def test_rotating_batches_equal_hint_free_objects(const_iter, K):
    N, n = 64, 16384            # (big enough for every hint to be in play: match certificates, the matrix-core plan from 16384 targets on, the one-launch tail)
    batches = [make_pairs(N, n, n, seed=s) for s in (1, 2)] + [make_pairs(N, n, n, seed=3, max_rot=0.2, max_trans=1.0), make_scene_pairs(N, n, n, seed=4),
                                                               make_independent_pairs(N, n, n, seed=5, ragged=False)]
    batches = [(s.to(DEV), t.to(DEV)) for s, t in batches]
    T0 = torch.eye(4, device=DEV).repeat(N, 1, 1)

    def new():
        icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=1e-12 if const_iter else 1e-4)
        icp.const_iter = const_iter
        return icp
    want = [call(new(), S, T, T0) for S, T in batches]                         # hint-free: a new object per batch
    worn = new()
    for rnd in range(4):                                                        # the same object, 20 calls: every hint it uses comes from another batch
        for i, (S, T) in enumerate(batches):
            got = call(worn, S, T, T0)
            for key in ("T", "deltas", "weights", "costs", "pc"):
                assert torch.equal(got[0][key], want[i][0][key]), (rnd, i, key)
            for key in ("iterations", "converged", "matched_ratio"):
                assert torch.equal(got[0]["stats"][key], want[i][0]["stats"][key]), (rnd, i, key)
            for g, w in ((got[1], want[i][1]), (got[2], want[i][2])):
                scale = max(1.0, float(w.abs().max()))
                assert float((g - w).abs().max()) <= 2e-5 * scale, (rnd, i, float((g - w).abs().max()) / scale)


@pytest.mark.parametrize("tol", [None, 1e-4])
def test_calls_interleaved_with_their_backward_passes(tol):
    """Two calls of ONE object before either backward pass, the passes in either order, and a third call between them: every call's buffers, certificates and
    hint records are its own -- poses bit for bit and gradients to rounding as when each call is followed by its own backward."""
    from dicp_amd.synthetic import make_scene_pairs
    B, n, K = 32, 16384, 10 if tol is None else 50
    d1 = [x.cuda() for x in make_pairs(B, n, n, seed=5)]
    d2 = [x.cuda() for x in make_scene_pairs(B, n, n, seed=6)]
    T0 = torch.eye(4, device="cuda").repeat(B, 1, 1)
    kw = dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0})
    icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=tol if tol else 1e-12)
    icp.const_iter = tol is None
    ref = []
    for d in (d1, d2):
        for _ in range(2):
            s, t = d[0].clone().requires_grad_(True), d[1].clone().requires_grad_(True)
            o = icp.icp(s, t, T0, **kw)
            o["T"].sum().backward()
        ref.append((o["T"].detach().clone(), s.grad.clone(), t.grad.clone()))
    for rnd in range(3):
        s1, t1 = d1[0].clone().requires_grad_(True), d1[1].clone().requires_grad_(True)
        s2, t2 = d2[0].clone().requires_grad_(True), d2[1].clone().requires_grad_(True)
        o1 = icp.icp(s1, t1, T0, **kw)
        o2 = icp.icp(s2, t2, T0, **kw)
        if rnd == 0:
            (o1["T"].sum() + o2["T"].sum()).backward()
        elif rnd == 1:
            o2["T"].sum().backward()
            o1["T"].sum().backward()
        else:
            o1["T"].sum().backward()
            icp.icp(s1.detach(), t1.detach(), T0, **kw)
            o2["T"].sum().backward()
        torch.cuda.synchronize()
        assert torch.equal(o1["T"], ref[0][0]) and torch.equal(o2["T"], ref[1][0]), rnd
        for g, e in ((s1.grad, ref[0][1]), (t1.grad, ref[0][2]), (s2.grad, ref[1][1]), (t2.grad, ref[1][2])):
            assert float((g - e).abs().max()) <= 3e-5 * float(e.abs().max()), rnd


def test_tolerance_mode_reads_the_counters_from_mapped_words():
    """The reference's all-converged check (ICP.py:259) without a copy in the stream: the last launch of every segment stores the segment's counters -- clouds
    still moving per iteration -- to mapped host words tagged with the call (dicp_loop_buffers.counters_host / counters_tag).  After a call the words of its
    executed iterations carry the call's tag, the last executed iteration's count is zero and every earlier one's is not; the next call's tag differs."""
    import dicp_amd._loop as L
    N, n = 16, 8192
    src, tgt = make_pairs(N, n, n, seed=12)
    src, tgt = src.cuda(), tgt.cuda()
    T0 = torch.eye(4, device="cuda").repeat(N, 1, 1)
    icp = ICP(icp_type="pt2pl", differentiable=False, max_iterations=40, tolerance=1e-4)
    icp.const_iter = False
    tags = []
    for _ in range(3):
        out = icp.icp(src, tgt, T0, trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0})
        torch.cuda.synchronize()
        K = int(out["deltas"].shape[1])
        rec = L._TOL_WORDS[L.CallHints._where(src.device)]
        words = rec[1][:K].copy()
        tag = rec[2] << 20
        assert 1 < K < 40 and bool(((words & 0x7ff00000) == tag).all()) and bool((words >= 0).all())
        counts = words & 0xfffff
        assert int(counts[K - 1]) == 0 and bool((counts[:K - 1] > 0).all()) and int(counts[0]) == N
        assert bool((out["stats"]["iterations"] <= K).all())
        tags.append(tag)
    assert len(set(tags)) == 3


def test_backward_prepared_behind_a_tolerance_mode_forward(monkeypatch):
    """Tolerance mode enqueues the cotangent-free part of the reverse sweep behind the forward (dicp_loop_backward_prepare: the GPU idles there while the host
    returns and autograd starts): the pass that finds it made gives the gradients of the pass that makes it itself; with and without point weights."""
    import dicp_amd._loop as L
    N, n = 24, 16384
    src, tgt = make_pairs(N, n, n, seed=14)
    src, tgt = src.cuda(), tgt.cuda()
    T0 = torch.eye(4, device="cuda").repeat(N, 1, 1)
    wgt = (torch.rand((N, n), generator=torch.Generator().manual_seed(2)) * 0.5 + 0.5).cuda()
    for weight in (None, wgt):
        res = []
        for prepared in (True, False):
            if not prepared:
                monkeypatch.setattr(L, "_fwd_prepare_backward", lambda ctx, S, spos_of: setattr(ctx, "pre", None))
            icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=40, tolerance=1e-4)
            icp.const_iter = False
            seen = []
            real = L.backward_once
            monkeypatch.setattr(L, "backward_once", lambda lib, code, P, F, *a, **k: (seen.append(bool(F.src_s)), real(lib, code, P, F, *a, **k))[1])
            s, t = src.clone().requires_grad_(True), tgt.clone().requires_grad_(True)
            w = weight.clone().requires_grad_(True) if weight is not None else None
            out = icp.icp(s, t, T0, weight=w, trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0})
            out["T"].sum().backward()
            monkeypatch.undo()
            assert seen == [prepared]
            res.append((out["T"].detach(), s.grad, t.grad) + ((w.grad,) if w is not None else ()))
        assert torch.equal(res[0][0], res[1][0])
        for a, b in zip(res[0][1:], res[1][1:]):
            assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max())
