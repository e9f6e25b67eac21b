"""The ICP loop as ONE autograd node (ICP.py:131-260), its call-to-call hints and its size policies: the host side of dicp_icp_forward / dicp_icp_backward.

PyTorch-ROCm is plumbing here (device memory, streams, the autograd graph); every arithmetic step runs in libdicp_hip.so.  Round 6: moved out of _ops.py and cut
into stages -- ICPLoop.forward = search set-up -> certificate policy -> loop state -> enqueue (one plan call, or segment by segment) -> finish; ICPLoop.backward =
one library call (backward_once) where every iteration takes the windowed form, else set-up -> runs -> finish.
"""
import ctypes
import time
from dataclasses import dataclass
from types import SimpleNamespace

import torch

from . import _lib
from . import _ops           # (its tunables are read through the module at call time: tests and scripts set them there)
from ._ops import (_DT, _LOSS, SweepIndex, _Arena, _gather_rows_raw, _on, _p, _pose_sums_to_gT, _segments, _stream, auto_knn_kind, f16_image,
                   pack_target, require_device, search_frame)


def form_tally_wanted(rec, have_image):
    """Whether a call's plain searches should tally their slabs' tiles per cloud (dicp_loop_buffers.search.form): when this call REPORTS (the first two calls of a shape
    and every sixteenth: CallHints.form_record), or when its searches choose each cloud's scoring form by the tallies -- the matrix-core image exists (or the record
    says the shape's slabs are long: it will) and no earlier report has given the shape a plan yet.  The tally is an atomic add per unit of the sweep onto one word
    per cloud: 0.03 ms of the search near the pose, 1.6 % of the benchmark's call (round 5, A/B on one box)."""
    if rec is None:
        return bool(have_image)
    reporting = rec["event"] is None and (rec["calls"] < 2 or rec["calls"] % 16 == 0)
    return bool(reporting or ((have_image or rec["long"]) and not rec.get("plan")))



class TailTimeout(RuntimeError):
    """A wait inside the one-launch tail of a backward pass ran out (dicp_hip.h, bwd_tail_arrive): that pass poisoned the gradients of the clouds
    concerned with NaN.  Raised by the pass itself when ICP.strict_errors is set (it then waits for its own kernels), else at the next backward pass of the
    same ICP object, or by ICP.check_errors()."""


def _strict_tail_check(cfg, word):
    """ICP.strict_errors: wait for the pass and look at its tail's error word (a (1,) int32 device tensor) now."""
    if cfg.strict_errors and word is not None and not torch.cuda.is_current_stream_capturing() and int(word.item()) != 0:
        raise TailTimeout("dicp_amd: a wait of this backward pass's one-launch tail ran out (the GPU was kept full by other work for ~0.5 s); its gradients "
                          "are NaN and were not returned.  Re-run the step, or set ICP._tuning['bwd_tail'] = False")


class CallHints:
    """What the earlier calls of ONE ICP object tell its later ones -- about time only, never about results (a stale or missing hint costs time; every
    search and every sweep is exact either way).  Private to the object (ICP._hints), one record per (device, stream, call shape):
      tail : where the previous backward passes' reverse sweeps ended (their live counters, copied to pinned memory behind their launches; the
             error word of the one-launch tail rides along and is checked when the record is read),
      cert : whether the shape's match certificates paid in the previous call (the per-cloud switch states, every sixteenth certified call)."""
    MAX_SHAPES = 16

    def __init__(self):
        self.tail, self.cert, self.form = {}, {}, {}
        self.newest_tail = None         # [pinned counters (Kmax + 1), event, (N, n, K), Kmax, looked at, serial]: the record of the last backward pass (tests)
        self.serial = 0

    @staticmethod
    def _where(dev):
        return (dev.index if dev.index is not None else torch.cuda.current_device(), int(torch.cuda.current_stream(dev).cuda_stream))

    def _slot(self, table, key, make):
        if key not in table:
            while len(table) >= self.MAX_SHAPES:
                table.pop(next(iter(table)))
            table[key] = make()
        return table[key]

    def tail_records(self, dev, shape, any_stream=False):
        """The records of this (device, stream, shape).  any_stream (inside a graph capture, which runs on a stream of its own): the list whose newest
        record is the newest of the shape on this device -- the warm-up calls' --, to be read only."""
        key = self._where(dev) + tuple(shape)
        if any_stream and not self.tail.get(key):
            same = [recs for k, recs in self.tail.items() if k[0] == key[0] and k[2:] == key[2:] and recs]
            if same:
                return max(same, key=lambda recs: recs[-1][5])
        return self._slot(self.tail, key, list)

    def form_record(self, dev, shape):
        """form : whether the previous call's plain searches had long slabs in any cloud (then this call's score such clouds on the matrix cores)."""
        return self._slot(self.form, self._where(dev) + tuple(shape), lambda: {"long": False, "host": None, "event": None, "calls": 0})

    def cert_record(self, dev, shape):
        return self._slot(self.cert, self._where(dev) + tuple(shape), lambda: {"skip": 0, "host": None, "event": None, "calls": 0})

    def check(self, wait=False):
        """Raise TailTimeout if a backward pass whose record has arrived (wait: of every pass so far) reported a wait that ran out."""
        for recs in self.tail.values():
            for rec in recs:
                if rec[1] is None or rec[4]:
                    continue
                if wait:
                    rec[1].synchronize()
                if rec[1].query():
                    rec[4] = True       # (looked at)
                    if int(rec[0][rec[3]]) != 0:
                        raise TailTimeout("dicp_amd: a wait of an earlier backward pass's one-launch tail ran out (the GPU was kept full by other work for "
                                          "~0.5 s); the gradients of that pass were poisoned with NaN.  Re-run it, or set ICP._tuning['bwd_tail'] = False"
                                          "  [report word 0x%08x: arrivals seen %d, block %d, generation %d; live clouds by iteration %s]"
                                          % (int(rec[0][rec[3]]) & 0xffffffff, (int(rec[0][rec[3]]) >> 16) & 0xfff, (int(rec[0][rec[3]]) >> 8) & 0xff,
                                             int(rec[0][rec[3]]) & 0xff, rec[0][:rec[3]].tolist()))


CERT_MIN_WORK = 2.0e6        # certified point-iterations (iterations after the certifying search x N x n) below which match certificates are not used
RESORT_SMALL_POINTS = 262144  # source points of a batch below which a call without certificates re-orders its queries before iterations 0 and 1 only: a re-ordering is
                              # ~23 us of latency whatever the size, and what it saves the next searches shrinks with the batch (profiles/r04_mid_size_resort.txt:
                              # 32 x 4096: 0.561 -> 0.514 ms per call; 64 x 8192: the full schedule stays best, 0.714 against 0.815)


def certificates_pay(reuse_matches, Kmax, cert_from, N, n):
    """The size policy of the match certificates: at least three certified iterations, and enough certified point-iterations to outweigh what they cost the host
    in buffers and set-up (CERT_MIN_WORK).  One definition for the loop, the one-call path's admission and the re-order schedule."""
    left = Kmax - 1 - cert_from
    return bool(reuse_matches) and left >= 3 and float(left) * N * n >= CERT_MIN_WORK


def resort_schedule(resort, N, n, Kmax, reuse_matches, cert_from):
    """The iterations before which the sweep re-orders its queries: `resort` as given, or (None) by the size of the call."""
    if resort is not None:
        return tuple(int(v) for v in resort)
    full, small = (0, 1, 2, 3), (0, 1)
    cf = max([k for k in full if k < Kmax] or [0]) if cert_from is None else max(0, int(cert_from))
    if certificates_pay(reuse_matches, Kmax, cf, N, n) or N * n >= RESORT_SMALL_POINTS:
        return full
    # the short schedule was measured for calls WITHOUT certificates: it is only taken where they do not pay under it either (the loop derives the
    # certifying search's iteration from the schedule it is handed -- with (0, 1) that is iteration 1, two certified iterations more than under `full`)
    cf_small = max([k for k in small if k < Kmax] or [0]) if cert_from is None else cf
    return full if certificates_pay(reuse_matches, Kmax, cf_small, N, n) else small




@dataclass
class LoopConfig:
    icp_type: str
    differentiable: bool
    max_iterations: int
    tolerance: float
    trim_dist: object          # None or float
    loss_name: object          # None | "huber" | "cauchy"
    loss_metric: float
    dim: int
    const_iter: bool
    tanh_steepness: float
    match_ratio_thresh: float
    knn_variant: int = _lib.KNN_AUTO
    sweep_resort: tuple = (0, 1, 2, 3)  # iterations at which the sweep kNN re-sorts its queries by x
    bwd_window: bool = True       # sweep path: backward in sorted space (LDS window + full-line atomic flush)
    stats_out: object = None      # optional dict: receives "knn_pairs" (pairs scored by the sweep kNN; int64 shards, sum them)
    hints: object = None          # optional CallHints of the calling ICP object (what its earlier calls tell this one about time)
    sync_every: object = None     # tolerance mode: iterations between the host's all-converged checks (None = auto)
    timing_events: object = None  # optional object with .handles(K) -> ctypes array of 4*K hipEvent_t (bench.py)
    prebuilt: object = None       # (target, SweepIndex) started by the caller before its own host work (prebuild_search)
    small_loop: bool = True       # small clouds: one block runs a cloud's whole chunk of iterations (icp_small_* kernels)
    cert_from: object = None      # iteration of the certifying search (None: the last re-ordering of the queries)
    gumbel: object = None         # (eps, tau, inject_U or None): the Gumbel-softmax correspondence (nn.py:43-70) instead of the nearest neighbour
    deterministic: bool = False   # backward: the same bits on every run (slot order from a stable sort, window rows summed in slot order, out-of-window rows without float atomics)
    bwd_tail: bool = True         # truncated reverse sweep: the iterations before the ones the previous call still worked at run as ONE launch (dicp_loop_buffers.bwd.tail_from)
    strict_errors: bool = False   # a pass that used that launch waits for itself and raises TailTimeout if a wait inside it ran out (ICP.strict_errors)
    plan_call: bool = True        # constant-iteration sweep calls: all segments behind one library call (dicp_icp_forward_plan)
    first_search: bool = True     # sweep path: iteration 0's search is enqueued right behind the index build
    cert_hint: bool = True        # a shape whose clouds ALL ended a call with their certificates switched off is searched plainly for the next 32 calls
    cert_sets: bool = True        # a query whose match has a runner-up within the scores' rounding keeps a SET of 4 candidate rows, re-scored per iteration instead of searched
    cert_backoff: bool = True     # match certificates are switched off per cloud, on device, when a certified iteration costs more than 60 % of a full search
    reuse_matches: bool = True    # sweep path: match certificates -- an iteration searches only the waves holding a query whose match is not proven
                                  # unchanged since the wave's last search (exact; knn_sweep_kernel CERT)
    bwd_skip_eps: object = None   # backward: an iteration whose normal-equation cotangent is below this fraction of the largest of the cloud's later
                                  # iterations adds nothing above rounding and is skipped for that cloud (None: 2^-22 for float32, 2^-40 for float64; 0: off)
    src_rows: object = None       # ragged batches: (N) int32 device tensors, rows of each source / target cloud that take part
    tgt_rows: object = None       # (ICP._batch: the clouds' own lengths; the kernels never touch a pad row)

    def params(self):
        return _lib.WeightParams(
            mode=_lib.PT2PL if self.icp_type == "pt2pl" else _lib.PT2PT,
            trim_on=int(self.trim_dist is not None and self.trim_dist >= 0.0),
            differentiable=int(self.differentiable),
            loss=_LOSS[self.loss_name],
            trim_dist=float(self.trim_dist if self.trim_dist is not None else 0.0),
            tanh_k=float(self.tanh_steepness),
            loss_delta=float(self.loss_metric),
            match_thresh=float(self.match_ratio_thresh))



def order_by_matches(src, spos_ref, m, m_pad, src_rows, tgt_rows):
    """(N,n) int32 slot order for the windowed backward: the queries by the sorted position of their reference match (dicp_query_order with spos_prev: equal-population
    buckets of the target's sorted rows; a cloud's pad rows last)."""
    N, n, _ = src.shape
    qo = torch.empty((N, n), dtype=torch.int32, device=src.device)
    unused = torch.ones((N, 2), dtype=src.dtype, device=src.device)      # (the x range of the targets: not read when the buckets come from the matches)
    with _on(src.device):
        _lib.check(_lib.load().dicp_query_order(_DT[src.dtype], _p(src), None, _p(unused), SweepIndex.NBKT, N, n, _p(qo), None, None, None, 0, _p(spos_ref), int(m_pad),
                                                None, None, int(m), _p(src_rows), _p(tgt_rows), _stream()), "dicp_query_order")
    return qo


def backward_once(lib, code, P, F, cfg, src, tgt, w0c, gT, want_tgt, want_w):
    """The reverse sweep of a sweep-path call whose iterations all take the windowed form inside one history slab, from ONE library call (dicp_loop_backward):
    F = _lib.LoopBackwardIn naming the forward's buffers.  Allocates the pass's workspace and results, places the one-launch tail by the hints of the previous
    calls of this shape and records this call's (ICPLoop.backward documents both).  -> (gsrc, gtgt, gT0, gw)"""
    dev, dt = src.device, src.dtype
    es = src.element_size()
    N, n, m, K, Kcap = F.N, F.n, F.m, F.K, F.K_cap
    stats = cfg.stats_out
    st = _stream()
    L = _lib.CallBackwardLayout()
    _lib.check(lib.dicp_loop_backward_plan(code, ctypes.byref(P), ctypes.byref(F), int(want_tgt), int(want_w), ctypes.byref(L)), "dicp_loop_backward_plan")
    eps = cfg.bwd_skip_eps
    if eps is None:
        eps = 2.0 ** -22 if dt == torch.float32 else 2.0 ** -40
    if cfg.loss_name == "huber" and not cfg.differentiable:       # (their reference gradient is NaN at an exactly zero residual whatever the cotangent)
        eps = 0.0
    ws = torch.empty((L.total // es,), dtype=dt, device=dev)
    if stats is not None and eps > 0.0:
        stats["bwd_live"] = ws[L.live // es:L.live // es + (Kcap + 1 + (es // 4) - 1) * 4 // es].view(torch.int32)[:Kcap + 1]
    tail_from, hints, entry = 0, None, None
    capturing = torch.cuda.is_current_stream_capturing()
    use_tail = eps > 0.0 and cfg.bwd_tail and cfg.hints is not None
    if use_tail:
        if not capturing:
            cfg.hints.check()
        hints = cfg.hints.tail_records(dev, (N, n, m, Kcap, dt), any_stream=capturing)
        hint = next((h for h in reversed(hints) if h[2] == (N, n, K) and (capturing or h[1].query())), None)
        if hint is not None and L.nblk_w <= lib.dicp_bwd_tail_max_blocks(code):
            counts = hint[0][:K].tolist()
            tail_from = min(K, max(0, next((k for k in range(K) if counts[k] * 8 >= N), K) - 1))
        if not capturing:
            if len(hints) >= 4:         # four pinned buffers in rotation: the oldest one is re-used once its copy has landed (and has been looked at)
                if hints[0][1].query() and hints[0][0].numel() >= Kcap + 1:
                    cfg.hints.check()
                    entry = hints.pop(0)
            else:
                entry = [torch.empty((max(Kcap + 1, 64),), dtype=torch.int32).pin_memory(), None, None, Kcap, False, 0]
    tail_word = None
    if tail_from > 0:
        a0 = L.arrive // es
        tail_word = ws[a0:a0 + (N + 1 + (es // 4) - 1) * 4 // es].view(torch.int32)[N:N + 1]
    if stats is not None:
        stats["bwd_tail_from"] = int(tail_from)
        stats["bwd_tail_error"] = tail_word         # (None: this pass had no one-launch tail -- an earlier pass's word must not stand for it)
    gsrc = torch.empty_like(src)
    gtgt = torch.empty_like(tgt) if want_tgt else None
    gw = torch.empty_like(w0c) if want_w else None
    gT0 = torch.empty((N, 4, 4), dtype=dt, device=dev)
    gTc = gT.contiguous() if gT is not None else None
    G = _lib.CallGrads(gT=gTc.data_ptr() if gTc is not None else None, gsrc=gsrc.data_ptr(), gtgt=gtgt.data_ptr() if want_tgt else None,
                       gT0=gT0.data_ptr(), gw=gw.data_ptr() if want_w else None, workspace=ws.data_ptr(), skip_eps=float(eps), tail_from=int(tail_from),
                       live_host=entry[0].data_ptr() if entry is not None else None)
    _lib.check(lib.dicp_loop_backward(code, ctypes.byref(P), ctypes.byref(F), ctypes.byref(G), st), "dicp_loop_backward")
    if entry is not None:
        entry[1] = torch.cuda.Event()
        entry[1].record()
        cfg.hints.serial += 1
        entry[2], entry[3], entry[4], entry[5] = (N, n, K), Kcap, False, cfg.hints.serial
        hints.append(entry)
        cfg.hints.newest_tail = entry
    _strict_tail_check(cfg, tail_word)
    return gsrc, gtgt, gT0, gw


# ------------------------------------------------------------------ ICPLoop.forward, stage by stage
# Every stage reads and extends ONE namespace S (the call's inputs, sizes and everything the earlier stages made); the stages are what the old 385-line
# forward did in this order, cut where its concerns change.
def _fwd_begin(ctx, source, target, T_init, w0, cfg):
    """The call's inputs and sizes."""
    for t, nm in ((source, "source"), (target, "target"), (T_init, "T_init")) + (((w0, "weight"),) if w0 is not None else ()):
        require_device(t, "ICP(" + nm + ")")
    S = SimpleNamespace(cfg=cfg, lib=_lib.load(), dev=source.device, dt=source.dtype, code=_DT[source.dtype], es=source.element_size(), T_init=T_init)
    S.N, S.n, _ = source.shape
    S.m, S.c = target.shape[1], target.shape[2]
    # w0 None = unit weights (weight=None on a tensor input): the kernels take w_init == NULL and read 4 bytes per point less
    S.src, S.tgt, S.w0c = source.contiguous(), target.contiguous(), (w0.contiguous() if w0 is not None else None)
    S.P = cfg.params()
    S.rows = 3 if cfg.icp_type == "pt2pt" else 1
    S.Kmax = int(cfg.max_iterations)
    assert S.Kmax >= 1, "max_iterations must be at least 1"
    S.need_grad = any(ctx.needs_input_grad[:4])
    if cfg.stats_out is not None:       # the statistics describe THIS call (an earlier call's certificate counters must not outlive it)
        for key in ("knn_pairs", "searched_again", "budgets", "bwd_live", "certs_off"):
            cfg.stats_out.pop(key, None)
    return S


def _fwd_search_setup(S):
    """Which search runs, its per-call structure (search frame, sorted rows, the matrix-core image) and the scoring form of the plain searches."""
    cfg, N, n, m, dt, dev = S.cfg, S.N, S.n, S.m, S.dt, S.dev
    kind = cfg.knn_variant & 0xff
    if cfg.gumbel is not None:
        kind = _lib.KNN_GUMBEL
    elif kind == _lib.KNN_AUTO:
        kind = auto_knn_kind(N, n, m)
    S.kind = kind
    S.owned = kind == _lib.KNN_SWEEP and S.need_grad and cfg.bwd_window
    if cfg.deterministic and S.need_grad and not S.owned:
        # (said here, before anything is enqueued -- not by loss.backward() after the whole forward has run)
        raise NotImplementedError("ICP.deterministic covers the sweep search with the windowed backward (knn_variant KNN_SWEEP, bwd_window): this call "
                                  "would take another form")
    sweep = None
    # (every search variant of a call scores in the SAME frame -- the one chosen for the queries' slabs, whether or not this variant sweeps: identical coordinates,
    #  identical scores, identical matches, near-ties included.  Round 6's extended fuzz had the brute-force path, in the target-only frame, take the other of two
    #  targets 6e-8 .. 2e-6 apart in squared distance in 5 of 552 cases.)
    with_q = S.T_init.dtype == dt and tuple(S.T_init.shape) == (N, 4, 4)
    if kind == _lib.KNN_SWEEP:
        pre = cfg.prebuilt
        if (pre is not None and pre[0].data_ptr() == S.tgt.data_ptr() and pre[0].shape == S.tgt.shape and pre[0].dtype == S.tgt.dtype
                and pre[1].tgt_s is not None and pre[1].tgt_rows is cfg.tgt_rows):
            sweep = pre[1]                           # started by the caller, under its host work
        else:
            sweep = SweepIndex(S.tgt, sorted_rows=True, tgt_rows=cfg.tgt_rows,
                               frame=search_frame(S.tgt, tgt_rows=cfg.tgt_rows, src=S.src if with_q else None, T_init=S.T_init if with_q else None, src_rows=cfg.src_rows))
    S.sweep = sweep
    # the searches run in the target cloud's search frame (dicp_search_frame): packed rows Q y + t, pose [Q C | Q r + t]
    S.soft = kind == _lib.KNN_GUMBEL        # soft correspondences: no search structure at all
    S.center = sweep.frame if sweep is not None else (None if S.soft else search_frame(S.tgt, tgt_rows=cfg.tgt_rows, src=S.src if with_q else None,
                                                                                   T_init=S.T_init if with_q else None, src_rows=cfg.src_rows))
    S.tgt4 = sweep.tgs4 if sweep is not None else (None if S.soft else pack_target(S.tgt, S.center, cfg.tgt_rows))
    S.m_pad = S.tgt4.shape[1] if S.tgt4 is not None else 0
    # the matrix-core searches' image of the packed rows (the sweep path: of the sorted rows, made with the index)
    img16 = f16_image(S.tgt4, m, cfg.tgt_rows) if kind == _lib.KNN_MFMA else (sweep.img16 if sweep is not None else None)
    # The scoring form of the plain searches (dicp_loop_buffers.search.form / search.form_plan).  Every plain search of a REPORTING call tallies its slabs' tiles per
    # cloud; what the tallies of an earlier call of this shape said (a hint, like the tail's and the certificates': it arrives through pinned memory, costs time
    # at worst) decides whether this call builds the matrix-core image for clouds too small to get one by size (some cloud's slabs were long: start poses a metre
    # off, a third of the source without counterpart) and which form each iteration's search takes for the whole batch; until a report has arrived the searches
    # choose per cloud from the previous search's tally (two launches, each taking its clouds).
    # (clouds of 32768 points and more score on the matrix cores in every plain search, as in round 4: there the per-cloud choice -- it sends a cloud whose
    #  slabs have become short back to the vector form -- cost 8-10 % of a 64 x 65536 call, profiles/r05_form_tally.txt)
    tally = (sweep is not None and _ops.F16_SWEEP and _ops.F16_SWEEP_ADAPTIVE and dt == torch.float32 and float(N) * n >= _ops.F16_SWEEP_MIN_QUERIES
             and not (cfg.knn_variant & 0xff00) and m < _ops.F16_SWEEP_STATIC_TARGETS)
    form_hint = None
    if tally and cfg.hints is not None and not torch.cuda.is_current_stream_capturing():
        form_hint = cfg.hints.form_record(dev, (N, n, m, dt))
        if form_hint["event"] is not None and form_hint["event"].query():
            rep = form_hint["host"].tolist()
            form_hint["long"] = bool(4 * rep[0] >= N)       # (a quarter of the clouds: the image costs every call 0.07 ms)
            form_hint["moving"] = bool(4 * rep[1] >= N)     # ... still had long slabs in the call's LAST plain search: they keep moving
            # the plan of the next calls: iteration k + 1 scores on the matrix cores if most clouds' slabs were long in iteration k's plain search
            # (-1: no plain search then -- a certified iteration -- : the loop decides as it does without a plan)
            form_hint["plan"] = [0] + [0 if c < 0 else (2 if 2 * c >= N else 1) for c in rep[2:2 + _ops.FORM_PLAN_ITERS]]
            form_hint["event"] = None
        if img16 is None and form_hint["long"]:
            img16 = sweep.make_image()
    S.form_plan = None
    if tally and img16 is not None and form_hint is not None and form_hint.get("plan"):
        # one form per iteration for the whole batch, from an earlier call's tallies (iterations beyond the report: as the last reported one)
        pl = form_hint["plan"]
        S.form_plan = (ctypes.c_int32 * S.Kmax)(*[(pl[k] if k < len(pl) else pl[-1]) for k in range(S.Kmax)])
    if tally and not form_tally_wanted(form_hint, img16 is not None):
        tally = False           # (nobody reads this call's tallies: no per-cloud choice inside it, no report after it -- the searches do not take them)
    S.img16, S.tally, S.form_hint = img16, tally, form_hint
    S.adaptive = bool(tally) and kind == _lib.KNN_SWEEP


def _fwd_certificate_policy(S):
    """Whether this call uses match certificates, and from which iteration (the size policy and the previous calls' hint)."""
    cfg, N, n, m, dt, dev, Kmax, sweep = S.cfg, S.N, S.n, S.m, S.dt, S.dev, S.Kmax, S.sweep
    S.keep_idx = (sweep is None or (S.need_grad and not S.owned)) and not S.soft     # (original indices: the brute-force searches and the atomic backward)
    # match certificates (sweep path): a motion budget per query, a filter value per unit of the sweep, per-cloud motion bounds, and what each iteration
    # searched again.  The searches before the LAST re-ordering of the queries run plain (a certifying search costs a quarter more, and its budgets would
    # not survive the steps of the first iterations: it takes three certified iterations to be worth it)
    resorts = [k for k in cfg.sweep_resort if 0 <= k < Kmax]
    S.cert_from = (max(resorts) if resorts else 0) if cfg.cert_from is None else max(0, int(cfg.cert_from))     # iteration of the certifying search
    # (... and the point-iterations they can save must outweigh what they cost the host in buffers and set-up: measured break-even, forward +
    #  backward, at ~2 M certified point-iterations -- 32 x 4096 x 10 iterations loses 6 %, x 20 iterations wins 6 %; profiles/r03_certificates_mid_sizes.txt)
    want_certs = sweep is not None and not (cfg.knn_variant & 0xff00) and not S.keep_idx and certificates_pay(cfg.reuse_matches, Kmax, S.cert_from, N, n)
    # Where EVERY cloud of the previous call of this shape ended with its certificates switched off (near-duplicated or duplicated targets: no match can be
    # proven; poses that keep moving: no budget survives), the next calls do not try: they search plainly -- the same results, without the certifying search
    # and the guard launches -- and after 32 calls they try again.  The previous call's switch states arrive through pinned memory, like the tail's hint.
    cert_hint = None
    clouds_moving = False       # the hint says: most clouds of this shape were still moving when their certificates were tried (they switched them off)
    if want_certs and cfg.cert_hint and cfg.cert_backoff and cfg.hints is not None and not torch.cuda.is_current_stream_capturing():
        cert_hint = cfg.hints.cert_record(dev, (N, n, m, Kmax, dt))
        if cert_hint["host"] is None or cert_hint["host"].shape[0] < N:
            cert_hint["host"] = torch.empty((N, 8), dtype=torch.int32).pin_memory()
        if cert_hint["skip"] > 0:
            cert_hint["skip"] -= 1
            want_certs = False
            clouds_moving = bool(cert_hint.get("moving", False))
        elif cert_hint["event"] is not None and cert_hint["event"].query():
            # every cloud off at the call's end (for good, or backed off: clouds that keep moving) -- or, in a batch so small that a launch is as
            # long as its slowest cloud (no more units than the GPU holds at once), ANY cloud whose certificates did not pay in two iterations
            hc = cert_hint["host"][:N]
            # (round 5: or most of them -- partially overlapping clouds that start a metre off: 247 of 256 ended a call switched off, and the call cost
            #  34.6 ms with certificates against 28.2 without, profiles/r05_independent_forms.txt)
            most_off = float((hc[:, 2] > 0).float().mean()) >= 0.5
            if most_off or (N * ((n + 127) // 128) <= 8192 and bool((hc[:, 7] >= 2).any())):
                cert_hint.update(skip=31, calls=0, moving=most_off)  # (the first certified call after the pause reports again)
                want_certs = False
                clouds_moving = most_off
            cert_hint["event"] = None
        if want_certs:
            cert_hint["calls"] += 1                                  # certified calls of this shape
    # (... or, in a call without certificates, the form hint: a quarter of the clouds of this shape still had long slabs in the previous call's last plain search)
    S.clouds_moving = bool(clouds_moving or (not want_certs and S.form_hint is not None and S.form_hint.get("moving", False)))
    S.want_certs, S.cert_hint = want_certs, cert_hint


def _fwd_loop_state(S):
    """Every buffer of the loop (histories in slabs, zero-initialised state from one arena, the certificates' arrays), pose_0, and where the segments are cut."""
    cfg, lib, N, n, m, c, dt, dev, es, Kmax, sweep = S.cfg, S.lib, S.N, S.n, S.m, S.c, S.dt, S.dev, S.es, S.Kmax, S.sweep
    S.nblk = lib.dicp_accumulate_blocks(n)
    S.poses = torch.empty((Kmax + 1, N, 12), dtype=dt, device=dev)
    S.poses_c = torch.empty((Kmax + 1, N, 12), dtype=dt, device=dev) if S.center is not None else None   # [Q C | Q r + t]: what the searches read
    S.alive = torch.empty((Kmax + 1, N), dtype=dt, device=dev)
    S.areg = torch.empty((Kmax, N, 36), dtype=torch.float64, device=dev) if S.need_grad else None
    S.n_start = torch.empty((N,), dtype=dt, device=dev)
    S.partials = torch.empty((N, S.nblk, _lib.NACC_PAD), dtype=dt, device=dev)
    # every zero-initialised piece of loop state comes out of ONE zeroed arena (one fill instead of seven)
    want_certs = S.want_certs
    arena = _Arena(dev)
    arena.take((N, Kmax, 6), dt)                            # deltas
    arena.take((N, Kmax), dt)                               # costs
    arena.take((N,), torch.uint8)                           # converged
    arena.take((N,), dt)                                    # iterations
    arena.take((N,), dt)                                    # matched ratio
    arena.take((N,), dt)                                    # n_matched
    arena.take((Kmax,), torch.int32)                        # clouds still moving, per iteration
    arena.take((Kmax, 128) if want_certs else (0,), torch.int32)
    arena.take((N, 8) if want_certs else (0,), torch.int32)
    arena.take((N, n) if want_certs else (0,), torch.int32)     # (row cache: matches a guard launch leaves for the accumulate of its iteration; zero = none)
    arena.take((Kmax + 1, 8) if want_certs else (0,), torch.int32)     # (lengths of the guard launches' work lists, per iteration)
    # per-cloud tallies of the plain searches' slab lengths (dicp_loop_buffers.search.form)
    arena.take((Kmax, N) if S.adaptive else (0,), torch.int32)
    arena.take((N,) if (want_certs and cfg.cert_sets) else (0,), torch.int32)     # (lengths of the clouds' candidate-set lists)
    (S.deltas, S.costs, S.converged, S.iterations, S.matched, S.n_matched, S.counters, cert_count, S.cert_cloud, cert_pend, cert_gcount, S.sweep_form,
     cert_scount) = arena.finish()
    S.certs = None
    if want_certs:
        units = (n + 63) // 64          # (units of the sweep's one-query-per-lane forms; the two-query form uses half of them)
        # (row cache of the certified iterations: the matched row of every query, a filter per 64 queries, and -- with gradients -- where each such
        #  group's matches lie in the history, which those iterations keep by reference: dicp_loop_buffers.hist.spos_of)
        S.certs = dict(q=torch.empty((N, n), dtype=dt, device=dev), qu=torch.empty((N, units), dtype=dt, device=dev), count=cert_count,
                       nbr=torch.empty((N, n, 6 if cfg.icp_type == "pt2pl" else 3), dtype=dt, device=dev), gdirty=torch.empty((N, units), dtype=torch.int32, device=dev),
                       cm=torch.empty((N, n), dtype=torch.int32, device=dev), glist=torch.empty((8, max(N, 2) * units), dtype=torch.int32, device=dev), gcount=cert_gcount,
                       slist=torch.empty((N, n), dtype=torch.int32, device=dev) if cfg.cert_sets else None, scount=cert_scount if cfg.cert_sets else None,
                       pend=cert_pend, of=torch.empty((Kmax + 1, N, units), dtype=torch.int32, device=dev) if S.need_grad else None,
                       set=torch.empty((N * n * (es + 16),), dtype=torch.uint8, device=dev) if cfg.cert_sets else None,     # candidate sets: (N,n) budgets + (N,n,4) rows
                       rmax=torch.empty((N, 4), dtype=dt, device=dev), dcum=torch.empty((N, 2 * (Kmax + 1)), dtype=dt, device=dev))
    certs = S.certs
    # pose_0, alive_0, n_start (ICP.py:124-129)
    _lib.check(lib.dicp_loop_init(S.code, _p(S.T_init.contiguous()), _p(S.w0c), float(cfg.match_ratio_thresh), S.rows, N, n,
                                  _p(S.poses), _p(S.alive), _p(S.n_start), _p(S.center), _p(S.poses_c),
                                  _p(S.src) if certs else None, _p(certs["rmax"]) if certs else None, _p(certs["dcum"]) if certs else None, 2 * (Kmax + 1), S.st),
               "dicp_loop_init")
    # sweep path: the matches are kept as SORTED positions (spos) -- accumulate gathers the sorted, sector-aligned rows with them and
    # the windowed backward consumes them; original indices (idx) are only kept for the brute-force searches and the atomic backward
    S.idx_once = torch.empty((N, n), dtype=torch.int32, device=dev) if (S.keep_idx and not S.need_grad) else None
    S.keep_spos = sweep is not None and S.need_grad       # per-iteration sorted positions (the windowed backward reads them; same layout as idx)
    S.spos_once = torch.empty((N, n), dtype=torch.int32, device=dev) if (sweep is not None and not S.need_grad) else None
    # histories in slabs of kc iterations: slab j covers iterations [j*kc, (j+1)*kc)
    per_iter = N * n * max(es, 4)
    S.kc = max(1, min(Kmax, _ops.HIST_CHUNK_BYTES // max(1, per_iter)))
    S.w_slabs, S.idx_slabs, S.spos_slabs = [], [], []
    cuts = list(range(0, Kmax, S.kc))
    if sweep is not None:
        cuts += list(cfg.sweep_resort) + ([S.cert_from] if certs is not None else [])
    if not cfg.const_iter:
        every = cfg.sync_every
        if every is None:
            every = 1 if float(N) * n * m >= _ops.SWEEP_MIN_PAIRS else 4
        cuts += list(range(0, Kmax, max(1, int(every))))
    S.segs = _segments(Kmax, cuts)
    ev = cfg.timing_events
    S.events = ev.handles(Kmax) if ev is not None else None
    S.gum = None
    if S.soft:    # Gumbel-softmax correspondences: the neighbour ROWS of every iteration and their log-sum-exp are the history the reverse sweep reads
        g_eps, g_tau, inject_U = cfg.gumbel
        S.nbr_hist = torch.empty((Kmax, N, n, c), dtype=dt, device=dev)
        S.lse_hist = torch.empty((Kmax, N, n), dtype=dt, device=dev)
        S.ps_t = torch.empty((N, n, 3), dtype=dt, device=dev)
        S.U_list = [u.to(device=dev, dtype=dt).contiguous() for u in inject_U[:Kmax]] if inject_U is not None else None
        assert S.U_list is None or len(S.U_list) >= Kmax, "one injected noise tensor per iteration"
        S.U_arr = (ctypes.c_void_p * Kmax)(*[u.data_ptr() for u in S.U_list]) if S.U_list is not None else None
        # in-kernel noise: one seed per iteration from torch's CPU generator (torch.manual_seed makes a call reproducible)
        S.seed_list = [int(v) & 0xFFFFFFFF for v in torch.randint(0, 2 ** 31 - 1, (Kmax,)).tolist()] if S.U_list is None else [0] * Kmax
        S.seeds = (ctypes.c_uint32 * Kmax)(*S.seed_list)
        S.gum = _lib.GumbelLoop(U=ctypes.cast(S.U_arr, ctypes.c_void_p) if S.U_arr is not None else None, seeds=ctypes.cast(S.seeds, ctypes.c_void_p),
                                eps=float(g_eps), tau=float(g_tau), ps_t=_p(S.ps_t), nbr=_p(S.nbr_hist), lse=_p(S.lse_hist))
        S.g_eps, S.g_tau = float(g_eps), float(g_tau)
    # iteration 0's search may already be running: prebuild_search enqueued it behind the index build (under T_init's search pose and the
    # first query order), so that the GPU has 0.4 ms of work while this function prepares the loop (dicp_loop_buffers.search.first_done)
    S.first = cfg.prebuilt[2] if (sweep is not None and cfg.prebuilt is not None and cfg.prebuilt[1] is sweep) else None
    S.have_first = (S.first is not None and S.first[0].data_ptr() == S.src.data_ptr() and S.first[0].shape == S.src.shape
                    and S.first[1].data_ptr() == S.T_init.data_ptr() and S.T_init.is_contiguous())
    S.first_spos = None
    if (S.have_first and len(S.first) > 3 and S.first[3] is not None and not S.keep_idx and S.events is None and not (certs is not None and S.cert_from <= 0)):
        S.first_spos = S.first[3]
        if S.adaptive and sweep.form0 is not None:
            _lib.check(S.lib.dicp_copy(_p(S.sweep_form[0]), _p(sweep.form0), N * 4, S.st), "dicp_copy")      # (that search's tally of its slabs: the scoring form of iteration 1's)
        if not S.keep_spos:
            S.spos_once = S.first_spos
    S.qorders, S.seg_q, S.done_segs = [], [], []          # distinct query orders of the sweep, which one each segment used, the segments that ran
    S.K = Kmax


def _fwd_loop_buffers(S, plan_call):
    """dicp_loop_buffers with the fields that do not change from segment to segment (building the struct is ~20 us of host time)."""
    cfg, sweep, certs = S.cfg, S.sweep, S.certs
    LB = _lib.LoopBuffers(
        src=_p(S.src), tgt=_p(S.tgt), w_init=_p(S.w0c), c=S.c, K=S.Kmax, search_knn_variant=S.kind | (cfg.knn_variant & 0xff00) | ((0 if cfg.small_loop else 1) << 25), search_m_pad=S.m_pad,
        search_tgt4=_p(S.tgt4), search_tperm=_p(sweep.tperm) if sweep else None,
        search_bucket=_p(sweep.bucket) if sweep else None, search_brange=_p(sweep.brange) if sweep else None,
        search_nbkt=SweepIndex.NBKT, hist_per_iter=int(S.need_grad), search_pairs=_p(sweep.pair_shards) if sweep else None,
        hist_poses=_p(S.poses), hist_deltas=_p(S.deltas), hist_costs=_p(S.costs), hist_areg=_p(S.areg), hist_alive=_p(S.alive), converged=_p(S.converged),
        iterations=_p(S.iterations), matched_ratio=_p(S.matched), n_start=_p(S.n_start), n_matched=_p(S.n_matched),
        search_tgt_sorted=_p(sweep.tgt_s) if sweep is not None else None, search_tgt_sorted_stride=sweep.row_stride if sweep is not None else 0,
        cert_rmax=_p(certs["rmax"]) if certs else None, cert_dcum=_p(certs["dcum"]) if certs else None,
        hist_w_iter=S.n, hist_w_stride=S.kc * S.n,
        partials=_p(S.partials), counters=_p(S.counters), events=S.events, search_frame=_p(S.center), search_poses=_p(S.poses_c),
        src_rows=_p(cfg.src_rows), tgt_rows=_p(cfg.tgt_rows), search_tgt_f16=_p(S.img16),
        search_form=_p(S.sweep_form) if S.adaptive else None, search_form_default=sweep.form_default if sweep is not None else 0,
        search_form_plan=ctypes.cast(S.form_plan, ctypes.c_void_p) if S.form_plan is not None else None)
    if plan_call:       # (one slab of every history: the plan call's segments all write into it)
        LB.hist.w = _p(S.w_slabs[0])
        LB.hist.spos = _p(S.spos_slabs[0]) if S.keep_spos else _p(S.spos_once)
        LB.hist.idx = (_p(S.idx_slabs[0]) if S.need_grad else _p(S.idx_once)) if S.keep_idx else None
        LB.search.first_done = int(S.first_spos is not None)
        LB.hist.spos_of = _p(certs["of"]) if certs else None
    if S.gum is not None:
        LB.search.gumbel = ctypes.cast(ctypes.pointer(S.gum), ctypes.c_void_p)
    return LB


def _fwd_new_slab(S, j):
    """History slab j: the weights in the layout the API returns (same cloud stride in every slab), the matches the backward will read."""
    kk = min(S.kc, S.Kmax - j * S.kc)
    S.w_slabs.append(torch.empty((S.N, S.kc, S.n), dtype=S.dt, device=S.dev))
    if S.need_grad and S.keep_idx:
        S.idx_slabs.append(torch.empty((kk, S.N, S.n), dtype=torch.int32, device=S.dev))
    if S.keep_spos:
        S.spos_slabs.append(torch.empty((kk, S.N, S.n), dtype=torch.int32, device=S.dev))
        if j == 0 and S.first_spos is not None:
            # (a kernel of the library, not torch's copy_: a contiguous device-to-device copy_ is a runtime memcpy, and under a hipGraph capture the runtime's
            #  memset / memcpy nodes are not reliably ordered against the kernel nodes around them: csrc/dicp_fill.h)
            _lib.check(S.lib.dicp_copy(_p(S.spos_slabs[0][0]), _p(S.first_spos), S.first_spos.numel() * 4, S.st), "dicp_copy")


def _fwd_enqueue_plan(S):
    """Constant-iteration calls of the sweep path with all histories in one slab: every segment and the query re-orderings between them behind ONE library
    call (dicp_icp_forward_plan) -- per segment the host spent ~40 us, which a mid-size call does not have."""
    cfg, certs, N, n = S.cfg, S.certs, S.N, S.n
    _fwd_new_slab(S, 0)
    SP = _lib.SegmentPlan(nseg=len(S.segs), cert_from=S.cert_from if certs is not None else -1, keys=_p(S.sweep.keys),
                          cert_q=_p(certs["q"]) if certs else None, cert_qu=_p(certs["qu"]) if certs else None,
                          cert_count=_p(certs["count"]) if certs else None, cert_cloud=_p(S.cert_cloud) if (certs and cfg.cert_backoff) else None,
                          cert_set=_p(certs["set"]) if certs else None, cert_nbr=_p(certs["nbr"]) if certs else None,
                          cert_gdirty=_p(certs["gdirty"]) if certs else None, cert_pend=_p(certs["pend"]) if certs else None,
                          cert_cm=_p(certs["cm"]) if certs else None, cert_glist=_p(certs["glist"]) if certs else None,
                          cert_gcount=_p(certs["gcount"]) if certs else None, cert_slist=_p(certs["slist"]) if certs else None,
                          cert_scount=_p(certs["scount"]) if certs else None)
    n_new = sum(1 for (k0, _) in S.segs if (k0 == 0 or k0 in cfg.sweep_resort)) - (1 if S.have_first else 0)
    fresh_orders = torch.empty((max(n_new, 1), N, n), dtype=torch.int32, device=S.dev)
    used, qorder = 0, None
    for si, (k0, k1) in enumerate(S.segs):
        SP.k0[si], SP.k1[si] = k0, k1
        if k0 == 0 or k0 in cfg.sweep_resort:
            if k0 == 0 and S.have_first:
                qorder, SP.new_order[si] = S.first[2], 0
            else:
                qorder, SP.new_order[si] = fresh_orders[used], 1
                used += 1
            S.qorders.append(qorder)
        else:
            SP.new_order[si] = 0
        SP.order[si] = qorder.data_ptr()
        S.seg_q.append(len(S.qorders) - 1)
        S.done_segs.append((k0, k1))
    LB = _fwd_loop_buffers(S, True)
    _lib.check(S.lib.dicp_icp_forward_plan(S.code, ctypes.byref(S.P), ctypes.byref(LB), ctypes.byref(SP), N, n, S.m, int(cfg.dim), 1, float(cfg.tolerance), S.st),
               "dicp_icp_forward_plan")


_TOL_WORDS = {}      # (device, stream) -> [pinned int32 words, their numpy view, serial of the last call]


def _tolerance_words(dev, Kmax):
    """Tolerance mode's mapped host words (dicp_loop_buffers.counters_host) for a call on this device and stream: every word -1, and the call's tag.  Calls on one
    stream follow each other, so they share the words; a late segment of the PREVIOUS call that is still running writes words with the previous tag, in stream
    order before any of this call's, and is ignored."""
    key = CallHints._where(dev)
    rec = _TOL_WORDS.get(key)
    if rec is None or rec[0].numel() < Kmax:
        t = torch.empty((max(64, Kmax),), dtype=torch.int32).pin_memory()
        # (the words a longer call outgrows are kept, not freed: a late segment of the previous call may still store to them, and the pinned allocator knows
        #  nothing of stores made by kernels -- it would hand the memory to the next pin_memory() at once)
        rec = _TOL_WORDS[key] = [t, t.numpy(), rec[2] if rec else 0, (rec[3] + [rec[0]]) if rec else []]
    rec[2] = (rec[2] + 1) % 0x7ff                     # (0x7ff itself is what -1 carries in the tag bits)
    rec[1][:Kmax] = -1
    return rec, rec[2] << 20


def _words_converged_at(pending, view, tag):
    """pending = (k0, k1): K = 1 + the first iteration of [k0, k1) after which no cloud was still moving (ICP.py:240,259), or None.  Waits for that segment's
    words only (they arrive behind its last step; the next segment is already enqueued)."""
    k0, k1 = pending
    t_end = None
    while True:
        vals = view[k0:k1]
        if bool(((vals >= 0) & ((vals & 0x7ff00000) == tag)).all()):
            break
        if t_end is None:
            t_end = time.monotonic() + 10.0
        elif time.monotonic() > t_end:
            torch.cuda.synchronize()
            vals = view[k0:k1]
            if not bool(((vals >= 0) & ((vals & 0x7ff00000) == tag)).all()):
                raise RuntimeError("dicp_amd: the convergence counters of iterations [%d, %d) never reached the host" % (k0, k1))
            break
    zero = (vals & 0xfffff) == 0
    return k0 + int(zero.argmax()) + 1 if bool(zero.any()) else None


def _fwd_enqueue_segments(S):
    """One dicp_icp_forward per segment: the host acts between them -- a new history slab, a re-ordering of the sweep's queries, or (tolerance mode) the
    reference's all-converged check, ICP.py:259."""
    cfg, certs, lib, N, n, kc, sweep = S.cfg, S.certs, S.lib, S.N, S.n, S.kc, S.sweep
    LB = LBref = None
    qorder, order_k = None, 0        # (order_k: the iteration whose search pose `qorder` was made -- or kept -- under)
    pending = None                   # tolerance mode: the segment whose convergence counters are still in flight
    words, tag = (None, 0) if cfg.const_iter else _tolerance_words(S.dev, S.Kmax)
    for (k0, k1) in S.segs:
        j = k0 // kc
        if j == len(S.w_slabs):
            _fwd_new_slab(S, j)
        use_certs = certs is not None and k0 >= S.cert_from
        if sweep is not None and (qorder is None or k0 in cfg.sweep_resort):
            # queries re-ordered by x under the current pose
            if k0 == 0 and S.have_first:
                qorder = S.first[2]                    # ordered under T_init by the caller (prebuild_search)
            else:
                ps_hist = S.poses_c if S.poses_c is not None else S.poses
                qorder = sweep.query_order(S.src, ps_hist[k0], src_rows=cfg.src_rows, pose_prev=ps_hist[order_k] if qorder is not None else None, order_prev=qorder)
            order_k = k0
            S.qorders.append(qorder)
        S.seg_q.append(len(S.qorders) - 1)
        base = j * kc                                         # virtual bases: slab pointer minus its first iteration
        if LB is None:
            LB = _fwd_loop_buffers(S, False)
            LBref = ctypes.byref(LB)
            if words is not None:
                LB.counters_host, LB.counters_tag = words[0].data_ptr(), tag
        LB.search.qorder = _p(qorder)
        LB.search.first_done = int(k0 == 0 and S.first_spos is not None)
        LB.hist.spos = ctypes.c_void_p(S.spos_slabs[j].data_ptr() - base * N * n * 4) if S.keep_spos else _p(S.spos_once)
        LB.hist.idx = (ctypes.c_void_p(S.idx_slabs[j].data_ptr() - base * N * n * 4) if S.need_grad else _p(S.idx_once)) if S.keep_idx else None
        LB.cert.q, LB.cert.qu, LB.cert.count = (_p(certs["q"]), _p(certs["qu"]), _p(certs["count"])) if use_certs else (None, None, None)
        LB.cert.cloud = _p(S.cert_cloud) if (use_certs and cfg.cert_backoff) else None
        LB.cert.set = _p(certs["set"]) if use_certs else None
        LB.cert.nbr, LB.cert.gdirty, LB.cert.pend, LB.cert.cm = (_p(certs["nbr"]), _p(certs["gdirty"]), _p(certs["pend"]), _p(certs["cm"])) if use_certs else (None, None, None, None)
        LB.cert.glist, LB.cert.gcount = (_p(certs["glist"]), _p(certs["gcount"])) if use_certs else (None, None)
        LB.cert.slist, LB.cert.scount = (_p(certs["slist"]), _p(certs["scount"])) if use_certs else (None, None)
        LB.hist.spos_of = _p(certs["of"]) if use_certs else None
        LB.cert.reset = int(k0 == S.cert_from)
        # (history in several slabs: a certified iteration finds the matches of the slab before its own through spos_prev_chunk)
        LB.hist.spos_floor = base
        LB.hist.spos_prev_chunk = ctypes.c_void_p(S.spos_slabs[j - 1].data_ptr() - (j - 1) * kc * N * n * 4) if (S.keep_spos and j > 0) else None
        LB.hist.w = ctypes.c_void_p(S.w_slabs[j].data_ptr() - base * n * S.es)
        LB.hist.w_prev0 = _p(S.w_slabs[(k0 - 1) // kc][:, (k0 - 1) % kc]) if k0 > 0 else None
        _lib.check(lib.dicp_icp_forward(S.code, ctypes.byref(S.P), LBref, N, n, S.m, int(cfg.dim), int(cfg.const_iter),
                                        float(cfg.tolerance), k0, k1, S.st), "dicp_icp_forward")
        S.done_segs.append((k0, k1))
        if not cfg.const_iter:
            # ICP.py:259: stop at the first iteration whose steps are ALL below tolerance.  The reference synchronises
            # every iteration for this; here the segment's last launch stores its counters to mapped host words (dicp_loop_buffers.counters_host:
            # no copy engine in the stream) and they are read one segment LATER, while the next segment is already running: no drained GPU, no
            # launch bubble.  The price is at most one segment of frozen no-op iterations past K (every cloud has converged,
            # so nothing moves), trimmed below exactly like the ones a sync_every > 1 leaves.
            if pending is not None:
                K_at = _words_converged_at(pending, words[1], tag)
                if K_at is not None:
                    S.K = K_at
                    pending = None
                    break
            pending = (k0, k1)
    if pending is not None:          # the last segment that ran
        K_at = _words_converged_at(pending, words[1], tag)
        if K_at is not None:
            S.K = K_at


def _fwd_finish(S):
    """ICP.py:267-281: the stats of the clouds that never converged, T from the last pose, pc; this call's statistics and its reports to the next calls."""
    cfg, lib, N, n, K, dt, dev, sweep, certs = S.cfg, S.lib, S.N, S.n, S.K, S.dt, S.dev, S.sweep, S.certs
    # (clouds that never converged report the matches of the LAST executed iteration; with sync_every > 1 a few frozen no-op iterations may have run
    #  past K, which leaves n_matched of such clouds unchanged or zero-weighted)
    T = torch.empty((N, 4, 4), dtype=dt, device=dev)
    _lib.check(lib.dicp_loop_finish(S.code, _p(S.poses[K]), _p(S.alive[K]), _p(S.n_start), _p(S.n_matched), K, N,
                                    _p(S.iterations), _p(S.matched), _p(T), S.st), "dicp_loop_finish")
    if sweep is not None and cfg.stats_out is not None:
        cfg.stats_out["knn_pairs"] = sweep.pair_shards    # device int64 shards: sum them after a sync
        if certs is not None:             # (Kmax, 128) int32: [:, :64].sum(1) = units, [:, 64:].sum(1) = single queries searched again per iteration
            cfg.stats_out["searched_again"] = certs["count"]
            cfg.stats_out["budgets"] = certs["q"]         # (N,n) by query: the budgets as the last iteration left them
            cfg.stats_out["certs_off"] = (S.cert_cloud[:, 2] > 0).to(torch.int32)  # (N) int32: 1 = the cloud's certificates were switched off during the call (they cost more than searching everything)
    cert_hint, form_hint = S.cert_hint, S.form_hint
    # (reports to later calls are copies to pinned memory behind an event: not from inside a graph capture, whose replays nobody would read them from)
    if (certs is not None and sweep is not None and cfg.stats_out is not None and cert_hint is not None and cert_hint["event"] is None and cert_hint["calls"] % 16 == 1
            and not torch.cuda.is_current_stream_capturing()):
        cert_hint["host"][:N].copy_(S.cert_cloud, non_blocking=True)        # (every sixteenth certified call: it is host time)
        cert_hint["event"] = torch.cuda.Event()
        cert_hint["event"].record()
    if (S.adaptive and form_hint is not None and form_hint["event"] is None and (form_hint["calls"] < 2 or form_hint["calls"] % 16 == 0)
            and not torch.cuda.is_current_stream_capturing()):
        # (the first two calls of a shape and every sixteenth after: it is host time) clouds with long slabs in any plain search of this call, in its
        # last one, and per iteration (the next calls' plan)
        if form_hint["host"] is None:
            form_hint["host"] = torch.empty((2 + _ops.FORM_PLAN_ITERS,), dtype=torch.int32).pin_memory()
        units128 = ((cfg.src_rows if cfg.src_rows is not None else n) + 127) // 128
        tallied = S.sweep_form[:K]
        searched = (tallied > 0).any(dim=1)                                    # iterations with a plain search
        last = torch.where(searched, torch.arange(1, tallied.shape[0] + 1, device=dev), 0).argmax()     # the last of them
        long_k = (tallied > _ops.FORM_TILES * units128).sum(dim=1, dtype=torch.int32)                        # clouds with long slabs, per iteration
        per_k = torch.full((_ops.FORM_PLAN_ITERS,), -1, dtype=torch.int32, device=dev)
        kk = min(K, _ops.FORM_PLAN_ITERS)
        per_k[:kk] = torch.where(searched[:kk], long_k[:kk], torch.full_like(long_k[:kk], -1))
        form_hint["host"].copy_(torch.cat((torch.stack(((tallied > _ops.FORM_TILES * units128).any(dim=0).sum(dtype=torch.int32),
                                                        (tallied[last] > _ops.FORM_TILES_MOVING * units128).sum(dtype=torch.int32))), per_k)), non_blocking=True)
        form_hint["event"] = torch.cuda.Event()
        form_hint["event"].record()
    if form_hint is not None:
        form_hint["calls"] += 1
    weights = (S.w_slabs[0] if len(S.w_slabs) == 1 else torch.cat(S.w_slabs, dim=1))[:, :K]
    # ICP.py:274: the transformed source, in this node too (one launch; its own autograd node cost a mid-size call 35 us of host time)
    pc = torch.empty_like(S.src)
    _lib.check(lib.dicp_transform_points(S.code, _p(S.src), _p(S.poses[K]), _p(pc), N, n, S.st), "dicp_transform_points")
    return T, pc, S.deltas[:, :K], weights, S.costs[:, :K]


def _fwd_save(ctx, S):
    """What the reverse sweep reads: inputs, pose / step / alive histories, the matches per iteration, the query orders, the sorted rows."""
    cfg, certs, sweep = S.cfg, S.certs, S.sweep
    spos_of = certs["of"] if (certs is not None and S.keep_spos) else None
    saved = ([S.src, S.tgt, S.w0c, S.poses, S.deltas, S.areg, S.alive] + S.idx_slabs + S.spos_slabs + S.qorders + ([sweep.tperm, sweep.tgt_s] if S.owned else [])
             + ([spos_of] if spos_of is not None else []))
    if S.soft:
        saved += [S.nbr_hist, S.lse_hist] + (S.U_list if S.U_list is not None else [])
    ctx.save_for_backward(*saved)
    ctx.cfg, ctx.K, ctx.P, ctx.Kmax = cfg, S.K, S.P, S.Kmax
    ctx.soft = (S.g_eps, S.g_tau, S.seed_list, len(S.U_list) if S.U_list is not None else 0) if S.soft else None
    ctx.layout = (len(S.idx_slabs), len(S.spos_slabs), len(S.qorders), S.kc, S.owned, S.m_pad, S.kind,
                  [(a, min(b, S.K), q) for (a, b), q in zip(S.done_segs, S.seg_q) if a < S.K])
    ctx.of_from = S.cert_from if spos_of is not None else None
    # Clouds that do not converge: the forward's last query order (iteration 3's) is stale by the last iteration -- a block's slots no longer match a
    # window of neighbouring target rows, and most contributions take the float atomics (0.43 instead of 0.15 ms per launch on independently sampled
    # clouds, profiles/r05_ragged_lists.txt).  The backward then orders its slots by the reference matches themselves (one counting-sort launch).
    ctx.bwd_reorder = bool(S.clouds_moving) and sweep is not None
    _fwd_prepare_backward(ctx, S, spos_of)


def _one_call_backward(cfg, owned, soft, n_spos_slabs, n_idx_slabs, c, cv, segs, K):
    """Whether the pass is ONE library call (dicp_loop_backward): every iteration takes the windowed form inside one history slab."""
    return bool(owned and soft is None and cfg.timing_events is None and n_spos_slabs == 1 and n_idx_slabs == 0 and c == cv and segs and K >= 1 and not cfg.deterministic)


def _loop_backward_in(cfg, src, tgt_s, w0c, tperm, qorder, spos, poses, deltas, areg, alive, m, K, Kmax, m_pad, kind, spos_of, of_from):
    N, n, _ = src.shape
    return _lib.LoopBackwardIn(src=src.data_ptr(), tgt_sorted=tgt_s.data_ptr(), w0=w0c.data_ptr() if w0c is not None else None, tperm=tperm.data_ptr(),
                               qorder=qorder.data_ptr(), spos=spos.data_ptr(), poses=poses.data_ptr(), deltas=deltas.data_ptr(), areg=areg.data_ptr(),
                               alive=alive.data_ptr(), src_rows=cfg.src_rows.data_ptr() if cfg.src_rows is not None else None,
                               tgt_rows=cfg.tgt_rows.data_ptr() if cfg.tgt_rows is not None else None, N=N, n=n, m=m, c=tgt_s.shape[2], K=K, K_cap=Kmax, m_pad=m_pad,
                               dim=int(cfg.dim), knn_variant=kind | ((0 if cfg.small_loop else 1) << 25),
                               spos_of=spos_of.data_ptr() if spos_of is not None else None, spos_of_from=int(of_from) if of_from is not None else 0)


def _fwd_prepare_backward(ctx, S, spos_of):
    """Tolerance mode: the part of the reverse sweep that needs no cotangent -- the source and its weights in slot order, the reference matches out of the history
    kept by reference -- goes to the GPU NOW, behind the forward (dicp_loop_backward_prepare).  The host has just waited for the iteration count; until the
    backward's first launch it returns from the call, the loss is taken and autograd starts its thread, ~0.2 ms in which the GPU has nothing else to do
    (profiles/r06_tolerance_gap.txt): 58 us of the pass at 256 x 16384 move into that gap.  Constant-iteration calls gain nothing from it (their GPU is never idle
    there) and would pay for it when no backward follows: they do not."""
    cfg, sweep = S.cfg, S.sweep
    ctx.pre = None
    cv = 6 if cfg.icp_type == "pt2pl" else 3
    if (cfg.const_iter or not S.need_grad or ctx.bwd_reorder or torch.cuda.is_current_stream_capturing()
            or not _one_call_backward(cfg, S.owned, ctx.soft, len(S.spos_slabs), len(S.idx_slabs), S.c, cv, ctx.layout[7], S.K)):
        return
    N, n, K = S.N, S.n, S.K
    F = _loop_backward_in(cfg, S.src, sweep.tgt_s, S.w0c, sweep.tperm, S.qorders[-1], S.spos_slabs[0], S.poses, S.deltas, S.areg, S.alive, S.m, K, S.Kmax, S.m_pad, S.kind,
                          spos_of, ctx.of_from)
    by_ref = spos_of is not None and K - 1 >= ctx.of_from
    src_s = torch.empty_like(S.src)
    w_s = torch.empty_like(S.w0c) if S.w0c is not None else None
    ref = torch.empty((N, n), dtype=torch.int32, device=S.dev) if by_ref else None
    _lib.check(S.lib.dicp_loop_backward_prepare(S.code, ctypes.byref(F), _p(src_s), _p(w_s), _p(ref), S.st), "dicp_loop_backward_prepare")
    ctx.pre = (src_s, w_s, ref, S.qorders[-1].data_ptr())


# ------------------------------------------------------------------ ICPLoop.backward, stage by stage (one namespace B, as the forward's S)
def _bwd_begin(ctx, gT, gpc):
    """The saved histories, and the part of the cotangent that arrives through pc = C_K p + r_K (it reaches the source directly and the pose through T)."""
    src, tgt, w0c, poses, deltas, areg, alive, *rest = ctx.saved_tensors
    B = SimpleNamespace(src=src, tgt=tgt, w0c=w0c, poses=poses, deltas=deltas, areg=areg, alive=alive, cfg=ctx.cfg, K=ctx.K, P=ctx.P, Kmax=ctx.Kmax)
    n_idx, n_spos, n_q, B.kc, B.owned, B.m_pad, B.kind, B.segs = ctx.layout
    B.idx_slabs, B.spos_slabs = rest[:n_idx], rest[n_idx:n_idx + n_spos]
    B.qorders = rest[n_idx + n_spos:n_idx + n_spos + n_q]
    B.of_from = getattr(ctx, "of_from", None)
    B.spos_of = None
    if B.of_from is not None:         # (the certified iterations' match history is kept by reference: dicp_loop_buffers.hist.spos_of)
        B.spos_of, rest = rest[-1], rest[:-1]
    B.tperm, B.tgt_s = (rest[-2], rest[-1]) if B.owned else (None, None)
    B.soft = getattr(ctx, "soft", None)
    if B.soft is not None:    # Gumbel-softmax correspondences: neighbour rows and log-sum-exp of every iteration (+ the injected noise)
        B.nbr_hist, B.lse_hist, *B.U_list = rest[n_idx + n_spos + n_q:]
    B.bwd_reorder = bool(getattr(ctx, "bwd_reorder", False))
    B.pre = getattr(ctx, "pre", None)
    B.lib = _lib.load()
    B.dev, B.dt = src.device, src.dtype
    B.code, B.es = _DT[B.dt], src.element_size()
    B.N, B.n, _ = src.shape
    B.m, B.c = tgt.shape[1], tgt.shape[2]
    B.cv = 6 if B.cfg.icp_type == "pt2pl" else 3
    B.want_tgt, B.want_w = bool(ctx.needs_input_grad[1]), bool(ctx.needs_input_grad[3] and w0c is not None)
    return B


def _bwd_pc_cotangent(B, gT, gpc):
    B.gsrc_pc = None
    if gpc is not None:
        B.gsrc_pc = torch.empty_like(B.src)
        pcp = torch.empty((B.N, B.lib.dicp_accumulate_blocks(B.n), _lib.NBWD_PAD), dtype=B.dt, device=B.dev)
        _lib.check(B.lib.dicp_transform_points_bwd(B.code, _p(B.src), _p(B.poses[B.K]), _p(gpc.contiguous()), _p(B.gsrc_pc), _p(pcp), B.N, B.n, B.st), "dicp_transform_points_bwd")
        gT_pc = _pose_sums_to_gT(pcp, B.N, B.dt, B.dev, B.st)
        gT = gT_pc if gT is None else gT + gT_pc
    return gT


def _bwd_one_call(B, gT):
    """Every iteration takes the windowed form inside one slab: the whole pass is one library call (dicp_loop_backward) on one allocation.
    (Tolerance mode feels it most: there the host cannot run ahead of the GPU, and what it does before the pass's first launch is exposed.)"""
    cfg, N, n, K = B.cfg, B.N, B.n, B.K
    qo_b = B.qorders[-1]
    if cfg.stats_out is not None:
        cfg.stats_out["bwd_reordered"] = B.bwd_reorder     # (the backward's slots were ordered by the reference matches: the clouds keep moving)
    if B.bwd_reorder:
        ref = B.spos_slabs[0][K - 1]
        if B.spos_of is not None and K - 1 >= B.of_from:
            ref = torch.empty((N, n), dtype=torch.int32, device=B.dev)
            _lib.check(B.lib.dicp_resolve_matches(_p(B.spos_slabs[0]), _p(B.spos_of), K - 1, _p(cfg.src_rows), N, n, _p(ref), B.st), "dicp_resolve_matches")
        qo_b = order_by_matches(B.src, ref, B.m, B.m_pad, cfg.src_rows, cfg.tgt_rows)
    F = _loop_backward_in(cfg, B.src, B.tgt_s, B.w0c, B.tperm, qo_b, B.spos_slabs[0], B.poses, B.deltas, B.areg, B.alive, B.m, K, B.Kmax, B.m_pad, B.kind, B.spos_of, B.of_from)
    if B.pre is not None and B.pre[3] == qo_b.data_ptr():        # (made behind the forward, in this slot order: dicp_loop_backward_prepare)
        F.src_s, F.w_s, F.spos_ref = _p(B.pre[0]), _p(B.pre[1]), _p(B.pre[2])
    return backward_once(B.lib, B.code, B.P, F, cfg, B.src, B.tgt, B.w0c, gT, B.want_tgt, B.want_w)


def _bwd_windowed_setup(B):
    """The windowed form's per-pass set-up: the reference matches that place the windows, the slot order, sorted copies, slot-order accumulators, slabs."""
    cfg, lib, N, n, m, dt, dev = B.cfg, B.lib, B.N, B.n, B.m, B.dt, B.dev
    # Two forms of accumulate_bwd.  Atomic form (dicp_accumulate_bwd): original order, no set-up.  Windowed form
    # (dicp_accumulate_bwd_window, sweep path): everything in sorted space -- sorted copies of the source /
    # weights, slot-order gradient accumulators, target rows in the sweep's order, per-block slabs for the target
    # gradient -- 2.4x faster per iteration for ~0.4 ms of set-up and un-permuting per call.  ONE slot order serves
    # every windowed iteration: the last query order of the forward (the best sorted under the final poses); the
    # matches are stored per query, so the forward may have searched those iterations in other orders.  Matches that
    # fall outside a window (early iterations, whose poses are still far) take the kernel's atomic side path.
    B.windowed = [bool(B.owned)] * len(B.segs)      # (measured: even iteration 0, whose matches lie far from the final ones, pays: 0.13 vs 0.28 ms;
                                                    #  a sweep call keeps sorted positions only, so it always takes the windowed form)
    B.only_windowed = all(B.windowed) and len(B.windowed) > 0
    B.all_windowed = B.only_windowed and B.c == B.cv      # every iteration takes the windowed form -> dicp_window_reduce writes gtgt
    B.gtgt = None
    if B.want_tgt:    # all windowed: dicp_window_reduce writes every element once, no zero fill needed
        B.gtgt = torch.empty_like(B.tgt) if B.all_windowed else torch.zeros_like(B.tgt)
    # likewise the source / weight gradients: un-permuted from the slot-order accumulators with = (dicp_permute_rows)
    B.gsrc = torch.empty_like(B.src) if B.only_windowed else torch.zeros_like(B.src)
    B.gw = (torch.empty_like(B.w0c) if B.only_windowed else torch.zeros_like(B.w0c)) if B.want_w else None
    B.nblk_a, B.nblk_w = lib.dicp_accumulate_blocks(n), lib.dicp_window_blocks(B.code, n, B.m_pad)
    B.det_row = B.det_val = B.qo = B.spos_ref = B.src_s = B.w_s = B.gsrc_s = B.gw_s = B.slab = B.gfar = None
    if cfg.deterministic and not B.only_windowed and len(B.segs) > 0:      # (the forward refuses such a call; no executed iteration: zero gradients, nothing to order)
        raise NotImplementedError("ICP.deterministic covers the sweep search with the windowed backward (knn_variant KNN_SWEEP, bwd_window): this call took another form")
    if not any(B.windowed):
        return
    kc = B.kc
    k_ref = max(b for (_, b, _), wf in zip(B.segs, B.windowed) if wf) - 1      # windows placed by the last iteration's matches
    spos_ref = B.spos_slabs[k_ref // kc][k_ref % kc]
    if B.spos_of is not None and k_ref >= B.of_from:    # (kept by reference: a plain array of them)
        spos_ref = torch.empty((N, n), dtype=torch.int32, device=dev)
        jr = k_ref // kc
        _lib.check(lib.dicp_resolve_matches(ctypes.c_void_p(B.spos_slabs[jr].data_ptr() - jr * kc * N * n * 4), _p(B.spos_of), k_ref, _p(cfg.src_rows), N, n, _p(spos_ref), B.st),
                   "dicp_resolve_matches")
    qo = B.qorders[len(B.qorders) - 1]
    if cfg.stats_out is not None:
        cfg.stats_out["bwd_reordered"] = B.bwd_reorder and not cfg.deterministic
    if B.bwd_reorder and not cfg.deterministic:
        qo = order_by_matches(B.src, spos_ref, m, B.m_pad, cfg.src_rows, cfg.tgt_rows)
    if cfg.deterministic:
        # The forward's query order comes from a counting sort whose order inside a bucket is the arrival order of LDS adds: fine for a search
        # (exact for any order), but the backward takes its sums by slot.  Any permutation serves as slot order: here a STABLE sort of the
        # queries by their reference match (the best locality the windows can have; a cloud's pad rows last, in index order).
        qo = torch.empty((N, n), dtype=torch.int32, device=dev)
        nbytes = int(lib.dicp_match_order_scratch_bytes(B.code, N, n))
        scratch = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
        _lib.check(lib.dicp_match_order(B.code, _p(spos_ref), _p(cfg.src_rows), N, n, _p(scratch), nbytes, _p(qo), B.st), "dicp_match_order")
        if B.want_tgt:
            B.det_row = torch.empty((N, n), dtype=torch.int32, device=dev)
            B.det_val = torch.empty((N, n, B.cv), dtype=dt, device=dev)
    B.qo, B.spos_ref = qo, spos_ref
    B.src_s = _gather_rows_raw(B.src, qo)
    B.w_s = _gather_rows_raw(B.w0c.unsqueeze(-1), qo).squeeze(-1) if B.w0c is not None else None
    # slot-order accumulators and slabs: the first windowed launch writes them (bwd_overwrite), no zero fill
    B.gsrc_s = torch.empty_like(B.src)
    B.gw_s = torch.empty_like(B.w0c) if B.want_w else None
    B.slab = torch.empty((N, B.nblk_w, lib.dicp_window_rows(B.code), B.cv), dtype=dt, device=dev) if B.want_tgt else None
    B.gfar = torch.zeros((N, B.m_pad, B.cv), dtype=dt, device=dev) if B.want_tgt else None


def _bwd_truncation_and_tail(B):
    """The truncated reverse sweep's state (dicp_loop_buffers.bwd.skip) and where the one-launch tail starts (from the previous calls' live counters)."""
    cfg, lib, N, n, K, Kmax, dt, dev = B.cfg, B.lib, B.N, B.n, B.K, B.Kmax, B.dt, B.dev
    # a cloud's sweep ends at the iteration from which on nothing reaches the result's own rounding; the iterations before it do no per-point work.
    # Not with hard Huber weights: their reference gradient is NaN at an exactly zero residual whatever the cotangent.
    eps = cfg.bwd_skip_eps
    if eps is None:
        eps = 2.0 ** -22 if dt == torch.float32 else 2.0 ** -40
    if (cfg.loss_name == "huber" and not cfg.differentiable) or B.soft is not None:    # (soft correspondences carry gradient themselves: nothing contracts the chain)
        eps = 0.0
    B.eps, B.skip = eps, None
    if eps > 0.0:
        sk_arena = _Arena(dev)
        sk_arena.take((N,), torch.float64)
        sk_arena.take((N,), torch.int32)
        sk_arena.take((Kmax + 1,), torch.int32)
        sk_arena.take((N + 1,), torch.int32)
        B.skip = sk_arena.finish()          # mref, decisions, live counters (+ the tail's error word), the one-launch tail's per-cloud counters (+ its error word)
        if cfg.stats_out is not None:
            cfg.stats_out["bwd_live"] = B.skip[2]     # (Kmax + 1) int32: clouds that did per-point work in iteration k of the backward; [Kmax]: a wait of the tail ran out
    # The ended iterations as ONE launch (dicp_loop_buffers.bwd.tail_from).  Where a sweep ends is decided on the device, while the host
    # enqueues; what the host can know is where the PREVIOUS call of this shape ended (its live counters, copied to pinned memory behind
    # that call's launches): the iterations at which fewer than an eighth of its clouds were still at work go to the one launch, which
    # sweeps a cloud that is at work after all with its blocks in step -- slower for that cloud (3x per iteration: profiles/r06_backward_forms.txt), exact either way.
    B.tail_from, B.hints = 0, None
    # (inside a graph capture no event may be queried: the hint of the warm-up calls is read as it stands -- torch's capture entry points
    #  synchronise first -- and none is recorded; a stale hint costs time, never correctness: a cloud at work in the tail is swept there)
    B.capturing = torch.cuda.is_current_stream_capturing()
    B.use_tail = B.skip is not None and B.only_windowed and cfg.bwd_tail and cfg.hints is not None and not cfg.deterministic
    if B.use_tail:
        if not B.capturing:
            cfg.hints.check()       # an earlier pass's tail ran out of patience: its gradients are NaN, and the caller hears about it here at the latest
        # (the newest hint that has ARRIVED: in a loop that never waits for the GPU the last call's own counters are still on their way)
        B.hints = cfg.hints.tail_records(dev, (N, n, B.m, Kmax, dt), any_stream=B.capturing)
        hint = next((h for h in reversed(B.hints) if h[2] == (N, n, K) and (B.capturing or h[1].query())), None)
        # the tail's blocks wait for each other: only where all of a cloud's blocks are resident at once (dicp_bwd_tail_max_blocks)
        if hint is not None and B.nblk_w <= lib.dicp_bwd_tail_max_blocks(B.code):
            live = hint[0][:K].tolist()
            B.tail_from = max(0, next((k for k in range(K) if live[k] * 8 >= N), K) - 1)      # (most sweeps have ended BEFORE the one launch starts)


def _bwd_runs(B, gpose, gtmp):
    """The reverse sweep itself: neighbouring segments of one form inside one history slab run as ONE library call (the forward cut them where the host
    had to act -- a new query order, a convergence check -- and none of that concerns the reverse sweep).  -> (gpose, form, have, folded)"""
    cfg, lib, N, n, Kmax, kc, dt, dev, skip = B.cfg, B.lib, B.N, B.n, B.Kmax, B.kc, B.dt, B.dev, B.skip
    gs = torch.empty((N, 36), dtype=dt, device=dev)
    gb = torch.empty((N, 6), dtype=dt, device=dev)
    bwd_flat = torch.empty((N * max(B.nblk_a, B.nblk_w) * _lib.NBWD_PAD,), dtype=dt, device=dev)
    B.bwdp = {False: bwd_flat[:N * B.nblk_a * _lib.NBWD_PAD].view(N, B.nblk_a, _lib.NBWD_PAD),
              True: bwd_flat[:N * B.nblk_w * _lib.NBWD_PAD].view(N, B.nblk_w, _lib.NBWD_PAD)}
    ev = cfg.timing_events
    events = ev.handles(Kmax) if ev is not None else None
    gum = None
    if B.soft is not None:
        g_eps, g_tau, seed_list, n_u = B.soft
        U_arr = (ctypes.c_void_p * Kmax)(*[u.data_ptr() for u in B.U_list[:Kmax]]) if n_u else None
        seeds = (ctypes.c_uint32 * Kmax)(*seed_list)
        ps_t, g_ps = torch.empty((N, n, 3), dtype=dt, device=dev), torch.empty((N, n, 3), dtype=dt, device=dev)
        g_nbr = torch.empty((N, n, B.c), dtype=dt, device=dev)
        gum = _lib.GumbelLoop(U=ctypes.cast(U_arr, ctypes.c_void_p) if U_arr is not None else None, seeds=ctypes.cast(seeds, ctypes.c_void_p),
                              eps=g_eps, tau=g_tau, ps_t=_p(ps_t), nbr=_p(B.nbr_hist), lse=_p(B.lse_hist), g_nbr=_p(g_nbr), g_ps=_p(g_ps))
    runs = []
    for (k0, k1, q), w_form in zip(reversed(B.segs), reversed(B.windowed)):
        if runs and runs[-1][3] == w_form and runs[-1][0] == k1 and (k0 // kc) == ((runs[-1][1] - 1) // kc) and k1 > k0:
            runs[-1] = (k0, runs[-1][1], q, w_form)
        else:
            runs.append((k0, k1, q, w_form))
    # (the one launch runs down to iteration 0: it belongs to the last run, and starts no higher than that run does)
    tail_from = B.tail_from = min(B.tail_from, runs[-1][1]) if (runs and runs[-1][0] == 0 and runs[-1][3]) else 0
    tail_part = torch.empty((N, B.nblk_w, _lib.NBWD_PAD), dtype=dt, device=dev) if tail_from > 0 else None
    if cfg.stats_out is not None:
        cfg.stats_out["bwd_tail_from"] = int(tail_from)
        # (1) int32: nonzero = a wait of the tail launch ran out (TailTimeout at the next pass); None: this pass had no one-launch tail
        cfg.stats_out["bwd_tail_error"] = skip[3][N:] if tail_from > 0 else None
    have, form, fresh, folded = 0, None, 1, False
    for (k0, k1, q, w_form) in runs:
        if have and w_form != form:     # the partials of the other form have another block count: fold them in here
            gpose += B.bwdp[form].sum(dim=1)[:, :12].to(torch.float64)
            have = 0
        form = w_form
        j = k0 // kc
        base = j * kc
        LB = _lib.LoopBuffers(
            src=_p(B.src_s) if w_form else _p(B.src), tgt=_p(B.tgt_s) if w_form else _p(B.tgt),
            w_init=_p(B.w_s) if w_form else _p(B.w0c), c=B.tgt_s.shape[2] if w_form else B.c, K=Kmax, search_knn_variant=B.kind | ((0 if cfg.small_loop else 1) << 25), search_m_pad=B.m_pad, hist_per_iter=1,
            search_qorder=_p(B.qo) if w_form else None,
            hist_spos=ctypes.c_void_p(B.spos_slabs[j].data_ptr() - base * N * n * 4) if w_form else None,
            bwd_spos_ref=_p(B.spos_ref) if w_form else None, bwd_gts_far=_p(B.gfar) if w_form else None,
            hist_spos_of=_p(B.spos_of) if (w_form and B.spos_of is not None) else None, hist_spos_of_from=int(B.of_from) if B.of_from is not None else 0,
            hist_poses=_p(B.poses), hist_deltas=_p(B.deltas), hist_areg=_p(B.areg), hist_alive=_p(B.alive),
            hist_idx=ctypes.c_void_p(B.idx_slabs[j].data_ptr() - base * N * n * 4) if B.idx_slabs else None, events=events,
            bwd_overwrite=fresh if (w_form and k1 > k0) else 0, src_rows=_p(cfg.src_rows), tgt_rows=_p(cfg.tgt_rows),
            bwd_skip=_p(skip[1]) if skip else None, bwd_mref=_p(skip[0]) if skip else None, bwd_live=_p(skip[2]) if skip else None,
            bwd_skip_eps=float(B.eps), bwd_tail_from=int(tail_from) if (w_form and k0 == 0) else 0,
            bwd_tail_partials=_p(tail_part), bwd_tail_arrive=_p(skip[3]) if skip else None,
            search_gumbel=ctypes.cast(ctypes.pointer(gum), ctypes.c_void_p) if gum is not None else None,
            bwd_det_far_row=_p(B.det_row) if w_form else None, bwd_det_far_val=_p(B.det_val) if w_form else None)
        fresh_was = bool(w_form and k1 > k0 and fresh)
        if w_form and k1 > k0:
            fresh = 0
        _lib.check(lib.dicp_icp_backward(B.code, ctypes.byref(B.P), ctypes.byref(LB), N, n, B.m, int(cfg.dim), _p(gpose), _p(gtmp), have,
                                         _p(gs), _p(gb), _p(B.gsrc_s) if w_form else _p(B.gsrc), _p(B.slab) if w_form else _p(B.gtgt),
                                         _p(B.gw_s) if w_form else _p(B.gw), _p(B.bwdp[form]), k0, k1, B.st), "dicp_icp_backward")
        have = 1
        if w_form and k0 == 0 and tail_from > 0:
            kt = min(tail_from, k1) - (1 if (fresh_was and tail_from >= k1) else 0)     # (dicp_icp_backward: the first windowed iteration is never the tail's)
            folded = kt > 0         # the tail launch left the cotangent of pose_0 with the last pose sums already in it
        if (k1 - k0) % 2:           # the library alternates the two buffers: odd chunk -> the result is in the other one
            gpose, gtmp = gtmp, gpose
    return gpose, form, have, folded


def _bwd_finish(B, gpose, form, have, folded):
    """This pass's report to the next ones (where its sweeps ended), the slot-order accumulators and slabs back into the inputs' own order, the cotangent of T_init."""
    cfg, lib, N, n, K, Kmax, dt, dev, skip = B.cfg, B.lib, B.N, B.n, B.K, B.Kmax, B.dt, B.dev, B.skip
    if B.use_tail and not B.capturing:
        hints, entry = B.hints, None
        if len(hints) >= 4:         # four pinned buffers in rotation: the oldest one is re-used once its copy has landed (and has been looked at)
            if hints[0][1].query() and hints[0][0].numel() >= Kmax + 1:
                cfg.hints.check()
                entry = hints.pop(0)
        else:
            entry = [torch.empty((max(Kmax + 1, 64),), dtype=torch.int32).pin_memory(), None, None, Kmax, False, 0]
        if entry is not None:
            entry[0][:Kmax + 1].copy_(skip[2], non_blocking=True)      # live counters + the tail's error word
            entry[1] = torch.cuda.Event()
            entry[1].record()
            cfg.hints.serial += 1
            entry[2], entry[3], entry[4], entry[5] = (N, n, K), Kmax, False, cfg.hints.serial
            hints.append(entry)
            cfg.hints.newest_tail = entry
    if any(B.windowed):               # slot s is source point qo[s]; slabs + out-of-window rows -> original target order
        permute = lib.dicp_permute_rows if B.only_windowed else lib.dicp_permute_add_rows
        _lib.check(permute(B.code, _p(B.gsrc_s), _p(B.qo), N, n, n, n, 3, 3, _p(B.gsrc), n, 3, B.st), "dicp_permute_rows")
        if B.want_w:
            _lib.check(permute(B.code, _p(B.gw_s), _p(B.qo), N, n, n, n, 1, 1, _p(B.gw), n, 1, B.st), "dicp_permute_rows")
        if B.want_tgt:
            _lib.check(lib.dicp_window_reduce(B.code, _p(B.slab), _p(B.spos_ref), _p(B.qo), _p(B.tperm), _p(B.gfar), _p(cfg.src_rows), N, n, B.m, B.m_pad, B.cv,
                                              _p(B.gtgt), B.c, int(B.all_windowed), B.st), "dicp_window_reduce")
    gT0 = torch.empty((N, 4, 4), dtype=dt, device=dev)      # final gpose + the last launch's pose partials
    if folded:
        have = 0
    _lib.check(lib.dicp_pose_grad_out(B.code, _p(gpose), _p(B.bwdp[form]) if have else None, B.bwdp[form].shape[1] if have else 0,
                                      _p(gT0), N, B.st), "dicp_pose_grad_out")
    return gT0


class ICPLoop(torch.autograd.Function):
    """The whole iteration loop of ICP.dICP (ICP.py:131-260) as ONE autograd node.

    forward : dicp_icp_forward enqueues K x { kNN -> accumulate -> step } back to back (no host work between
              iterations); it is called once per segment, segments being cut only where the host must act: a new
              history slab, a re-sort of the sweep's query order, or the reference's all-converged check (ICP.py:259,
              every `sync_every` iterations; converged clouds are frozen, so running a few extra iterations and
              trimming the histories afterwards gives the identical result).
    backward: dicp_icp_backward, K x { step_bwd -> accumulate_bwd } in reverse, recomputing per-point quantities
              from the saved (index, pose) histories instead of keeping autograd's intermediates.
    Inputs : source (N,n,3), target (N,m,c), T_init (N,4,4), w0 (N,n)  [one weight per POINT]
    Outputs: T (N,4,4) and pc (N,n,3) = the source under T (ICP.py:274), differentiable; deltas (N,K,6), weights (N,K,n), costs (N,K),
             converged (N) bool, iterations (N), matched_ratio (N)  (non-differentiable).
    """

    @staticmethod
    def forward(ctx, source, target, T_init, w0, cfg):
        S = _fwd_begin(ctx, source, target, T_init, w0, cfg)
        ctx.set_materialize_grads(False)    # no zero tensors for the six non-differentiable outputs (168 MB for the weights)
        with _on(S.dev):
            S.st = _stream()
            _fwd_search_setup(S)
            _fwd_certificate_policy(S)
            _fwd_loop_state(S)
            if cfg.const_iter and S.sweep is not None and S.kc >= S.Kmax and len(S.segs) <= _lib.MAX_SEGMENTS and cfg.plan_call:
                _fwd_enqueue_plan(S)
            else:
                _fwd_enqueue_segments(S)
            T, pc, deltas_out, weights, costs_out = _fwd_finish(S)
        if S.need_grad:
            _fwd_save(ctx, S)
        conv = S.converged.bool()
        ctx.mark_non_differentiable(deltas_out, weights, costs_out, conv, S.iterations, S.matched)
        return T, pc, deltas_out, weights, costs_out, conv, S.iterations, S.matched

    @staticmethod
    def backward(ctx, gT, gpc, *_unused):
        B = _bwd_begin(ctx, gT, gpc)
        cfg = B.cfg
        with _on(B.dev):
            B.st = _stream()
            gT = _bwd_pc_cotangent(B, gT, gpc)
            if _one_call_backward(cfg, B.owned, B.soft, len(B.spos_slabs), len(B.idx_slabs), B.c, B.cv, B.segs, B.K):
                gsrc, gtgt, gT0, gw = _bwd_one_call(B, gT)
                if B.gsrc_pc is not None:
                    gsrc += B.gsrc_pc
                return gsrc, gtgt, gT0, gw, None
            gpose = torch.empty((B.N, 12), dtype=torch.float64, device=B.dev)
            gtmp = torch.empty_like(gpose)
            _lib.check(B.lib.dicp_pose_grad_in(B.code, _p(gT.contiguous()) if gT is not None else None, _p(gpose), B.N, B.st), "dicp_pose_grad_in")
            _bwd_windowed_setup(B)
            _bwd_truncation_and_tail(B)
            gpose, form, have, folded = _bwd_runs(B, gpose, gtmp)
            gT0 = _bwd_finish(B, gpose, form, have, folded)
            if B.gsrc_pc is not None:
                B.gsrc += B.gsrc_pc
            if B.tail_from > 0:
                _strict_tail_check(cfg, B.skip[3][B.N:])
        return B.gsrc, B.gtgt, gT0, B.gw, None
