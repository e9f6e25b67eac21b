"""One eager ICP call behind ONE library call per direction (include/dicp_hip.h, dicp_call_*).

A mid-size call of the sweep path -- 32 clouds x 4096 points x 10 iterations is 0.55 ms of kernels -- spent more time than that in the interpreter,
which prepared ~30 buffers and ~12 library calls per direction (dicp_amd/_ops.py, ICPLoop).  For the calls that need none of ICPLoop's host decisions
(dense batch, constant iteration count, sorted sweep, no match certificates, every history in one slab) the same sequence of launches is made by
dicp_call_forward / dicp_call_backward on ONE allocation each; this module allocates, hands out the results as views and keeps autograd's books.
Every other call takes ICPLoop.  The two give the same results bit for bit (tests/test_gpu_call.py).

Lifetimes: the workspace (search structure, match history, pose histories, sort scratch: 18 MB at 32 x 4096, up to a few hundred MB at MAX_POINTS) lives as
long as the autograd node does -- the reverse sweep reads it -- and no longer: the non-differentiable results (deltas, weights, costs, iterations, matched_ratio,
converged) are views of a SECOND, small allocation of their own (dicp_call.results), so a training loop that keeps the costs of every step keeps nothing else.
`deltas`, the one result the reverse sweep reads, goes through save_for_backward: an in-place edit of it is refused by autograd's version check, as with ICPLoop.
"""
import ctypes

import torch

from . import _lib
from . import _loop
from . import _ops
from ._ops import _DT, _on, _p, _stream

MAX_POINTS = 1 << 20        # source points of a batch up to which a call is taken here: beyond, the kernels outlast the host anyway, and ICPLoop
                            # frees the search structure at the end of the call where this path's one allocation lives as long as any result does
_PLANS = {}      # (dtype, shape, ...) -> (Call prototype values, CallLayout): dicp_call_plan asked once per shape


def eligible(cfg, source, target, T_init, w0, need_grad):
    """True when ICPLoop would run this call as: sweep search, one dicp_icp_forward_plan, no certificates, windowed backward -- and nothing else."""
    if cfg.gumbel is not None or not cfg.const_iter or cfg.timing_events is not None or cfg.src_rows is not None or cfg.tgt_rows is not None:
        return False
    if not cfg.plan_call or (cfg.knn_variant & 0xff00) or (need_grad and not cfg.bwd_window):
        return False
    dt = source.dtype
    if dt not in _DT or not (source.is_cuda and target.is_cuda and T_init.is_cuda) or target.dtype != dt or T_init.dtype != dt:
        return False
    if not (source.is_contiguous() and target.is_contiguous() and T_init.is_contiguous()) or (w0 is not None and not (w0.is_cuda and w0.is_contiguous() and w0.dtype == dt)):
        return False
    N, n, _ = source.shape
    m, c = target.shape[1], target.shape[2]
    if N * n > MAX_POINTS or N * m > 4 * MAX_POINTS:
        return False
    if tuple(T_init.shape) != (N, 4, 4) or c != (6 if cfg.icp_type == "pt2pl" else 3):
        return False
    kind = cfg.knn_variant & 0xff
    if kind == _lib.KNN_AUTO:
        kind = _ops.auto_knn_kind(N, n, m)
    if kind != _lib.KNN_SWEEP:
        return False
    Kmax = int(cfg.max_iterations)
    resorts = [k for k in cfg.sweep_resort if 0 <= k < Kmax]
    if len([k for k in resorts if k > 0]) >= _lib.MAX_SEGMENTS:
        return False
    cert_from = (max(resorts) if resorts else 0) if cfg.cert_from is None else max(0, int(cfg.cert_from))
    if _loop.certificates_pay(cfg.reuse_matches, Kmax, cert_from, N, n):
        return False                # match certificates pay from there on: ICPLoop's business
    if _ops.F16_SWEEP and dt == torch.float32 and float(N) * n >= _ops.F16_SWEEP_MIN_QUERIES and m >= _ops.F16_SWEEP_MIN_TARGETS:
        return False                # the matrix-core scoring of big problems
    if _ops.HIST_CHUNK_BYTES // max(1, N * n * max(source.element_size(), 4)) < Kmax:
        return False                # histories in several slabs
    return not torch.cuda.is_current_stream_capturing()


def _plan(code, N, n, m, c, K, dim, need_grad, resort, flags):
    key = (code, N, n, m, c, K, dim, need_grad, resort, flags)
    got = _PLANS.get(key)
    if got is None:
        call = _lib.Call(N=N, n=n, m=m, c=c, K=K, dim=dim, need_grad=need_grad, n_resort=len(resort), flags=flags)
        for i, k in enumerate(resort):
            call.resort[i] = k
        lay = _lib.CallLayout()
        _lib.check(_lib.load().dicp_call_plan(code, ctypes.byref(call), ctypes.byref(lay)), "dicp_call_plan")
        if len(_PLANS) > 256:
            _PLANS.clear()
        got = _PLANS[key] = lay
    return got


class CallLoop(torch.autograd.Function):
    """ICPLoop's contract (same inputs, same eight outputs, same gradients) for the calls `eligible` admits."""

    @staticmethod
    def forward(ctx, source, target, T_init, w0, cfg):
        lib = _lib.load()
        dev, dt = source.device, source.dtype
        code, es = _DT[dt], source.element_size()
        N, n, _ = source.shape
        m, c = target.shape[1], target.shape[2]
        K = int(cfg.max_iterations)
        need_grad = int(any(ctx.needs_input_grad[:4]))
        stats = cfg.stats_out
        if stats is not None:
            for key in ("knn_pairs", "searched_again", "budgets", "bwd_live", "certs_off"):
                stats.pop(key, None)
        ctx.set_materialize_grads(False)
        resort = tuple(sorted(set(k for k in cfg.sweep_resort if 0 < k < K)))
        flags = (_lib.CALL_FIRST_SEARCH if cfg.first_search else 0) | (0 if cfg.small_loop else _lib.CALL_NO_SMALL_LOOP)
        L = _plan(code, N, n, m, c, K, int(cfg.dim), need_grad, resort, flags)
        P = cfg.params()
        with _on(dev):
            ws = torch.empty((L.total // es,), dtype=dt, device=dev)
            rs = torch.empty((L.results_total // es,), dtype=dt, device=dev)       # (the non-differentiable results: an allocation of their own, see the module's docstring)
            # (the two differentiable results are tensors of their own: autograd refuses the backward pass of a view whose base has been edited)
            T = torch.empty((N, 4, 4), dtype=dt, device=dev)
            pc = torch.empty((N, n, 3), dtype=dt, device=dev)
            call = _lib.Call(T_out=T.data_ptr(), pc_out=pc.data_ptr(), results=rs.data_ptr(), src=source.data_ptr(), tgt=target.data_ptr(), T_init=T_init.data_ptr(), w0=w0.data_ptr() if w0 is not None else None,
                             N=N, n=n, m=m, c=c, K=K, dim=int(cfg.dim), need_grad=need_grad, n_resort=len(resort), flags=flags,
                             directions=int(_ops.FRAME_DIRECTIONS), quantum=_ops.CENTER_QUANTUM, tolerance=float(cfg.tolerance), workspace=ws.data_ptr())
            for i, k in enumerate(resort):
                call.resort[i] = k
            _lib.check(lib.dicp_call_forward(code, ctypes.byref(P), ctypes.byref(call), _stream()), "dicp_call_forward")
        # each result a tensor on its own slice of `rs` with a version counter of its OWN (.data): editing the weights in place must not look like an edit of
        # the steps, which the reverse sweep reads and autograd therefore watches
        def piece(off, count, shape):
            return rs.narrow(0, off // es, count).view(shape).data
        deltas = piece(L.deltas, N * K * 6, (N, K, 6))
        weights = piece(L.weights, N * K * n, (N, K, n))
        costs = piece(L.costs, N * K, (N, K))
        iterations = piece(L.iterations, N, (N,))
        matched = piece(L.matched_ratio, N, (N,))
        conv = rs[L.converged // es:L.converged // es + (N + es - 1) // es].view(torch.uint8)[:N].bool()
        if stats is not None:
            stats["knn_pairs"] = ws[L.pairs // es:L.pairs // es + _lib.PAIR_SHARDS * 8 // es].view(torch.int64).clone()    # (a copy: the statistics must not keep the workspace alive)
        if need_grad:
            # (the workspace -- poses, matches, sorted rows: what the reverse sweep reads -- is the node's own; of the caller-visible results it reads
            #  `deltas`, which therefore goes through save_for_backward: autograd's version check refuses the pass if the caller edited it in place)
            ctx.save_for_backward(source, target, w0, deltas)
            ctx.ws, ctx.call, ctx.cfg, ctx.P, ctx.L = ws, call, cfg, P, L
        ctx.mark_non_differentiable(deltas, weights, costs, conv, iterations, matched)
        return T, pc, deltas, weights, costs, conv, iterations, matched

    @staticmethod
    def backward(ctx, gT, gpc, *_unused):
        src, tgt, w0, deltas = ctx.saved_tensors
        cfg, P, call, FL = ctx.cfg, ctx.P, ctx.call, ctx.L
        lib = _lib.load()
        dev, dt = src.device, src.dtype
        code, es = _DT[dt], src.element_size()
        N, n, _ = src.shape
        m = tgt.shape[1]
        K = call.K
        stats = cfg.stats_out
        want_tgt, want_w = bool(ctx.needs_input_grad[1]), bool(ctx.needs_input_grad[3] and w0 is not None)
        with _on(dev):
            st = _stream()
            gsrc_pc = None
            if gpc is not None:         # pc = C_K p + r_K: its cotangent reaches the source directly and the pose through T
                gsrc_pc = torch.empty_like(src)
                pcp = torch.empty((N, FL.nblk, _lib.NBWD_PAD), dtype=dt, device=dev)
                pose_K = ctx.ws.as_strided((N, 12), (12, 1), FL.poses // es + K * N * 12)
                _lib.check(lib.dicp_transform_points_bwd(code, _p(src), _p(pose_K), _p(gpc.contiguous()), _p(gsrc_pc), _p(pcp), N, n, st), "dicp_transform_points_bwd")
                gT_pc = _ops._pose_sums_to_gT(pcp, N, dt, dev, st)
                gT = gT_pc if gT is None else gT + gT_pc
            base = ctx.ws.data_ptr()
            F = _lib.LoopBackwardIn(src=src.data_ptr(), tgt_sorted=base + FL.tgt_sorted, w0=w0.data_ptr() if w0 is not None else None, tperm=base + FL.tperm,
                                    qorder=base + FL.orders + (FL.n_orders - 1) * N * n * 4, spos=base + FL.spos, poses=base + FL.poses, deltas=deltas.data_ptr(),
                                    areg=base + FL.areg, alive=base + FL.alive, N=N, n=n, m=m, c=call.c, K=K, K_cap=K, m_pad=FL.m_pad, dim=call.dim,
                                    knn_variant=_lib.KNN_SWEEP | ((1 << 25) if (call.flags & _lib.CALL_NO_SMALL_LOOP) else 0))
            gsrc, gtgt, gT0, gw = _loop.backward_once(lib, code, P, F, cfg, src, tgt, w0, gT, want_tgt, want_w)
            if gsrc_pc is not None:
                gsrc += gsrc_pc
        return gsrc, gtgt, gT0, gw, None


# ------------------------------------------------------------------ ICP.pt2pt_dICP_SVD (ICP.py:533-591), the same way
def kabsch_eligible(source, target, T_start, w0, const_iter, knn_variant, src_rows, tgt_rows):
    """True when KabschLoop would run this call as: sweep search, three segments, no host decision in between."""
    if not const_iter or src_rows is not None or tgt_rows is not None or (knn_variant & 0xff00):
        return False
    dt = source.dtype
    if dt not in _DT or not (source.is_cuda and target.is_cuda and T_start.is_cuda and w0.is_cuda) or target.dtype != dt or T_start.dtype != dt or w0.dtype != dt:
        return False
    N, n, _ = source.shape
    m = target.shape[1]
    if N * n > MAX_POINTS or N * m > 4 * MAX_POINTS or tuple(T_start.shape) != (N, 4, 4) or tuple(w0.shape) != (N, n):
        return False
    kind = knn_variant & 0xff
    if kind == _lib.KNN_AUTO:
        kind = _ops.auto_knn_kind(N, n, m)
    return kind == _lib.KNN_SWEEP and not torch.cuda.is_current_stream_capturing()


_KPLANS = {}


class KabschCall(torch.autograd.Function):
    """KabschLoop + the transformed cloud (ICP.py:581) as one node on one allocation: dicp_kabsch_call_forward / _backward.
    Inputs : source (N,n,3), target (N,m,c), T_start (N,4,4), w0 (N,n), K, trim_dist
    Outputs: T_found (N,4,4), pc (N,n,3) differentiable; costs (N,K), iterations (N)."""

    @staticmethod
    def forward(ctx, source, target, T_start, w0, K, trim_dist):
        lib = _lib.load()
        dev, dt = source.device, source.dtype
        code, es = _DT[dt], source.element_size()
        src, tgt, Ts, w0c = source.contiguous(), target.contiguous(), T_start.contiguous(), w0.contiguous()
        N, n, _ = src.shape
        m, c = tgt.shape[1], tgt.shape[2]
        K = int(K)
        trim_on = int(trim_dist is not None and trim_dist >= 0.0)
        key = (code, N, n, m, c, K)
        L = _KPLANS.get(key)
        if L is None:
            L = _lib.KabschCallLayout()
            _lib.check(lib.dicp_kabsch_call_plan(code, ctypes.byref(_lib.KabschCall(N=N, n=n, m=m, c=c, K=K)), ctypes.byref(L)), "dicp_kabsch_call_plan")
            if len(_KPLANS) > 256:
                _KPLANS.clear()
            _KPLANS[key] = L
        ctx.set_materialize_grads(False)
        with _on(dev):
            ws = torch.empty((L.total // es,), dtype=dt, device=dev)
            T = torch.empty((N, 4, 4), dtype=dt, device=dev)
            pc = torch.empty((N, n, 3), dtype=dt, device=dev)
            call = _lib.KabschCall(src=src.data_ptr(), tgt=tgt.data_ptr(), T_start=Ts.data_ptr(), w0=w0c.data_ptr(), N=N, n=n, m=m, c=c, K=K, trim_on=trim_on,
                                   directions=int(_ops.FRAME_DIRECTIONS), trim_dist=float(trim_dist) if trim_on else 0.0, quantum=_ops.CENTER_QUANTUM, tolerance=0.0,
                                   workspace=ws.data_ptr(), T_out=T.data_ptr(), pc_out=pc.data_ptr())
            _lib.check(lib.dicp_kabsch_call_forward(code, ctypes.byref(call), _stream()), "dicp_kabsch_call_forward")
        # (copies: these two small results must not keep the workspace -- the search structure and the sort scratch -- alive in a caller's log)
        costs = ws.as_strided((N, K), (K, 1), L.costs // es).clone()
        iterations = ws.as_strided((N,), (1,), L.iterations // es).clone()
        ctx.save_for_backward(src, tgt, w0c)
        ctx.ws, ctx.call, ctx.L = ws, call, L
        ctx.mark_non_differentiable(costs, iterations)
        return T, pc, costs, iterations

    @staticmethod
    def backward(ctx, gT, gpc, *_unused):
        src, tgt, w0c = ctx.saved_tensors
        call, L = ctx.call, ctx.L
        lib = _lib.load()
        dev, dt = src.device, src.dtype
        code, es = _DT[dt], src.element_size()
        N, n, _ = src.shape
        with _on(dev):
            st = _stream()
            gsrc_pc = None
            if gpc is not None:         # pc = C p + r under the pose found: to the source directly, to the pose through T
                gsrc_pc = torch.empty_like(src)
                pcp = torch.empty((N, L.nblk, _lib.NBWD_PAD), dtype=dt, device=dev)
                pose = ctx.ws.as_strided((N, 12), (12, 1), L.pose // es)
                _lib.check(lib.dicp_transform_points_bwd(code, _p(src), _p(pose), _p(gpc.contiguous()), _p(gsrc_pc), _p(pcp), N, n, st), "dicp_transform_points_bwd")
                gT_pc = _ops._pose_sums_to_gT(pcp, N, dt, dev, st)
                gT = gT_pc if gT is None else gT + gT_pc
            gTc = gT.contiguous() if gT is not None else None
            gsrc = torch.empty_like(src)
            gtgt = torch.empty_like(tgt) if ctx.needs_input_grad[1] else None
            gw = torch.empty_like(w0c) if ctx.needs_input_grad[3] else None
            G = _lib.KabschCallGrads(gT=gTc.data_ptr() if gTc is not None else None, gsrc=gsrc.data_ptr(), gtgt=gtgt.data_ptr() if gtgt is not None else None,
                                     gw=gw.data_ptr() if gw is not None else None)
            _lib.check(lib.dicp_kabsch_call_backward(code, ctypes.byref(call), ctypes.byref(G), st), "dicp_kabsch_call_backward")
            if gsrc_pc is not None:
                gsrc += gsrc_pc
        return gsrc, gtgt, None, gw, None, None
