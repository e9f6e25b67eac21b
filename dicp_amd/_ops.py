"""Tensor-level wrappers over the C ABI and the autograd glue.

PyTorch-ROCm is plumbing here (device memory, streams, the autograd graph); every
arithmetic step of the ICP iteration runs in libdicp_hip.so.  All functions require
HIP-device tensors and raise otherwise -- there is no CPU compute path.
"""
import os
import ctypes

import torch

from . import _lib

_DT = {torch.float32: _lib.F32, torch.float64: _lib.F64}
SWEEP_MIN_TARGETS = 2048     # KNN_AUTO inside ICP: below this many targets per cloud the brute-force kernel is used
SWEEP_MIN_QUERIES = 256
F16_SWEEP = True             # float32 sweep path: the plain searches of big clouds score on the matrix cores (split-f16 filter + exact refine; same indices)
F16_SWEEP_MIN_QUERIES = 2 * 256 * 1024      # ... from the size at which the sweep works in units of 128 queries (sweep_auto_cfg)
F16_SWEEP_MIN_TARGETS = 16384               # ... and, for a cloud whose slabs have not been measured yet, from 16384 targets on.  Round 6 (profiles/r06_f16_sweep_crossover.txt, the
                                            # sweep form as round 5 left it): at 256 x 16384 the matrix cores win from ~7 tiles per unit of 128 queries on -- 1.38x under the start
                                            # pose (15 tiles), 1.23x after one iteration (10), level after two (5.5), 0.85x after three (4); round 4 had measured 0.92x at 16 tiles,
                                            # and the threshold stood at 32768 targets / 32 tiles until this round.  Smaller clouds: per cloud by the previous plain search's tally of
                                            # slab lengths (dicp_loop_buffers.search.form; FORM_TILES) -- clouds that start metres off, or a third of which has no counterpart
FORM_PLAN_ITERS = 6                         # iterations whose tallies a reporting call hands to the next calls' plan (dicp_loop_buffers.search.form_plan); later ones follow the last
FORM_TILES_MOVING = 32                      # tiles per unit in a call's LAST plain search from which on its clouds count as still moving (the backward then orders its slots by the matches)
F16_SWEEP_STATIC_TARGETS = 32768            # from here on every plain search of every cloud scores on the matrix cores (the per-cloud choice cost 8-10 % of a 64 x 65536 call,
                                            # profiles/r05_form_tally.txt); between the two thresholds iteration 0's search does and the later ones go by the tallies -- planar
                                            # scenes close their slabs to 5 tiles in one iteration and are then faster in the vector form (0.200 against 0.240 ms per search)
F16_SWEEP_ADAPTIVE = os.environ.get("DICP_F16_ADAPTIVE", "1") != "0"   # (the environment switch: scripts/ab_adaptive.sh)                 # (False: the form is chosen by the size alone, as in round 4)
FORM_TILES = 12                             # (kernels_search.h: tiles per unit of 128 queries from which on a cloud's plain searches score on the matrix cores)
SWEEP_MIN_PAIRS = 1e8        # ... and below this many (query,target) pairs per iteration.  Measured (profiles/r02_mid_size_paths.txt): with the
                             # native key sort the sweep's per-call set-up is ~0.1 ms, and it already wins at 32 x 2048^2 and 8 x 4096^2
                             # (0.090 vs 0.103 and 0.075 vs 0.121 ms per iteration, fwd+bwd); at 32 x 4096^2 (BASELINE configs[1]) 0.084 vs 0.175


def auto_knn_kind(N, n, m):
    """KNN_AUTO: exact slab-pruned search once the clouds are big enough to repay its per-call sort."""
    big = m >= SWEEP_MIN_TARGETS and n >= SWEEP_MIN_QUERIES and float(N) * n * m >= SWEEP_MIN_PAIRS
    return _lib.KNN_SWEEP if big else _lib.KNN_VALU
_LOSS = {None: _lib.LOSS_NONE, "huber": _lib.LOSS_HUBER, "cauchy": _lib.LOSS_CAUCHY, "trim": _lib.LOSS_TRIM}


def require_device(t, what):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError("dicp_amd.%s needs tensors on a HIP device (got %s); there is no CPU fallback"
                           % (what, getattr(t, "device", type(t))))
    if t.dtype not in _DT:
        raise TypeError("dicp_amd.%s supports float32/float64, got %s" % (what, t.dtype))


def compute_device():
    """The device the kernels run on; raises if this process cannot see an MI355X."""
    if not torch.cuda.is_available():
        raise RuntimeError("dicp_amd: no HIP device visible (torch.cuda.is_available() is False). "
                           "The ICP hot path only exists as gfx950 kernels; there is no CPU fallback.")
    return torch.device("cuda", torch.cuda.current_device())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """The current HIP stream of the current device as a handle (torch.cuda.current_stream() builds a Stream object: ~10 us per call, and
    a call of the loop asks a dozen times)."""
    if _raw_stream is not None and _raw_device is not None:
        return ctypes.c_void_p(_raw_stream(_raw_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


class _NoSwitch:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_SWITCH = _NoSwitch()


def _on(device):
    """`with _on(dev):` = torch.cuda.device(dev), without the two device switches (~5 us) when dev already is the current device -- a call of the
    loop enters nine such blocks, and a mid-size call is host-bound (scripts/host_breakdown.py)."""
    if _raw_device is not None and device.index is not None and _raw_device() == device.index:
        return _NO_SWITCH
    return torch.cuda.device(device)


# ------------------------------------------------------------------ thin kernels
def padded_targets(m):
    return _lib.load().dicp_padded_targets(int(m))


def accumulate_blocks(n):
    return _lib.load().dicp_accumulate_blocks(int(n))


CENTER_QUANTUM = 16.0    # metres; clouds whose median point is within half of it of the origin keep c = 0
FRAME_DIRECTIONS = True  # the search frame also picks the sort direction (False: the x axis, always)


def search_frame(tgt, quantum=None, tgt_rows=None, directions=None, src=None, T_init=None, src_rows=None):
    """(N,m,c) -> (N,12): the search frame x' = Q x + t of every cloud (dicp_search_frame), [Q row-major | t] with t = -Q c:
    c = the coordinate-wise median of (a sample of) the target, rounded to a multiple of `quantum`, so that clouds near the origin get exactly 0;
    Q = the rotation whose first row is the direction the sorted sweep prunes along -- the identity unless one of five other candidates spreads
    the cloud's points clearly better (planar scenes: a wall perpendicular to x sits in every slab that touches it).
    tgt_rows (N) int32, optional: rows of each cloud that take part (ragged batches).
    src (N,n,3) + T_init (N,4,4), optional: the queries the searches will run for -- the direction is then chosen by the rows THEIR slabs hold."""
    require_device(tgt, "search_frame")
    N, m, c = tgt.shape
    out = torch.empty((N, 12), dtype=tgt.dtype, device=tgt.device)
    with_q = src is not None and T_init is not None
    if with_q:
        src, T_init = src.contiguous(), T_init.contiguous()
    with _on(tgt.device):
        _lib.check(_lib.load().dicp_search_frame(_DT[tgt.dtype], _p(tgt), c, _p(tgt_rows), N, m, CENTER_QUANTUM if quantum is None else float(quantum),
                                                 int(FRAME_DIRECTIONS if directions is None else directions),
                                                 _p(src) if with_q else None, _p(src_rows) if with_q else None, int(src.shape[1]) if with_q else 0,
                                                 _p(T_init) if with_q else None, _p(out), _stream()), "dicp_search_frame")
    return out


def search_pose(pose, frame, N=None):
    """[Q C | Q r + t] (N,12): the pose a search is handed when its packed rows / index were built in `frame` (pose None = identity: the frame itself).
    (torch arithmetic: a test / nn.find_nn helper -- the loop's search poses are written by the kernels.)"""
    if pose is None:
        if frame is not None:
            return frame.clone()
        pose = torch.zeros((N, 12))
        pose[:, 0] = pose[:, 4] = pose[:, 8] = 1.0
        return pose
    if frame is None:
        return pose
    n = pose.shape[0]
    Q, t = frame[:, :9].reshape(n, 3, 3), frame[:, 9:]
    out = torch.empty_like(pose)
    out[:, :9] = torch.matmul(Q, pose[:, :9].reshape(n, 3, 3)).reshape(n, 9)
    out[:, 9:] = torch.matmul(Q, pose[:, 9:, None]).squeeze(-1) + t
    return out


def pack_target(tgt, frame=None, tgt_rows=None):
    """(N,m,c) -> (N,m_pad,4) rows [x,y,z,0.5|y|^2] of y (or of Q y + t: the caller then searches with [Q C | Q r + t])."""
    require_device(tgt, "pack_target")
    tgt = tgt.contiguous()
    N, m, c = tgt.shape
    m_pad = padded_targets(m)
    out = torch.empty((N, m_pad, 4), dtype=tgt.dtype, device=tgt.device)
    with _on(tgt.device):
        _lib.check(_lib.load().dicp_pack_target(_DT[tgt.dtype], _p(tgt), c, _p(frame), _p(tgt_rows), N, m, _p(out), m_pad, _stream()),
                   "dicp_pack_target")
    return out


def f16_image(tgt4, m, tgt_rows=None):
    """The split-f16 image of packed target rows (N,m_pad,4) float32 for the matrix-core search (dicp_knn_f16_pack): a uint8 tensor."""
    require_device(tgt4, "f16_image")
    if tgt4.dtype != torch.float32:
        raise TypeError("dicp_amd.f16_image: the matrix-core search is float32 only")
    lib = _lib.load()
    N, m_pad, _ = tgt4.shape
    img = torch.empty((int(lib.dicp_knn_f16_bytes(N, m_pad)),), dtype=torch.uint8, device=tgt4.device)
    with _on(tgt4.device):
        _lib.check(lib.dicp_knn_f16_pack(_p(tgt4), _p(tgt_rows), N, int(m), m_pad, _p(img), _stream()), "dicp_knn_f16_pack")
    return img


def f16_counters(image, N, m_pad):
    """(queries sent through the second filter pass, queries scored against every row): what the matrix-core searches since the image was packed
    could not settle from one pass (near-ties inside the filter's resolution; queries outside the f16 range).  Synchronises."""
    m_img = (int(m_pad) + 511) // 512 * 512
    meta = image[N * m_img * 32:N * m_img * 32 + N * 320].view(torch.int32).view(N, 80)
    return int(meta[:, 6].sum().item()), int(meta[:, 7].sum().item())


def knn(src, pose, tgt4, m, variant=_lib.KNN_AUTO, out=None, src_rows=None, tgt_rows=None, image=None):
    """Fused transform + brute-force 1-NN: (N,n,3), (N,12)|None, packed targets -> idx (N,n) int32.
    src_rows / tgt_rows (N) int32, optional: rows of each cloud that take part (the idx of other rows is not written).
    KNN_MFMA reads the split-f16 image of tgt4 (f16_image; built here when not given)."""
    require_device(src, "knn")
    N, n, _ = src.shape
    idx = out if out is not None else torch.empty((N, n), dtype=torch.int32, device=src.device)
    if (variant & 0xff) == _lib.KNN_MFMA and image is None and src.dtype == torch.float32:
        image = f16_image(tgt4, m, tgt_rows)
    with _on(src.device):
        _lib.check(_lib.load().dicp_knn(_DT[src.dtype], _p(src), _p(pose), _p(tgt4), _p(src_rows), _p(tgt_rows), N, n, m, tgt4.shape[1],
                                        _p(idx), variant, _p(image), _stream()), "dicp_knn")
    return idx


class SweepIndex:
    """Per-call search structure of the exact sorted-sweep kNN (dicp_knn_sweep): targets do not move during
    an ICP call, so they are sorted by x once (dicp_sweep_sort: native for every size and dtype)."""
    NBKT = 1024

    def __init__(self, tgt, sorted_rows=False, frame=None, tgt_rows=None, first_order=None, first_search=False, tally=True):
        """sorted_rows: also keep tgt_s (N,m_pad,row_stride), the full rows in sorted order (the loop's accumulate and the windowed backward gather them).
        frame (N,12): the index is built on Q y + t (keys, table and packed rows; tgt_s keeps the rows as given) and the
        searches must then be given the pose [Q C | Q r + t].
        tgt_rows (N) int32: rows of each cloud that take part (ragged batches); the searches are given the same counts.
        first_order = (source, T_init, src_rows): the frame is chosen here and everything -- frame, sort, rows, the search pose of iteration 0 and the
        first query order (self.first = (source, T_init, qorder, spos0)) -- goes out in ONE library call (dicp_sweep_setup).
        first_search: iteration 0's search is enqueued right behind it (spos0 (N,n): its matches as sorted positions; None otherwise): 0.4 ms of
        kernel at the benchmark shape under which the host prepares the loop (dicp_loop_buffers.search.first_done)."""
        require_device(tgt, "SweepIndex")
        tgt = tgt.contiguous()
        N, m, c = tgt.shape
        self.m = m
        self.frame, self.tgt_rows = frame, tgt_rows
        lib = _lib.load()
        dev, dt = tgt.device, tgt.dtype
        m_pad = lib.dicp_padded_targets(m)
        self.tgs4 = torch.empty((N, m_pad, 4), dtype=dt, device=dev)
        self.tperm = torch.empty((N, m_pad), dtype=torch.int32, device=dev)
        self.bucket = torch.empty((N, self.NBKT + 1), dtype=torch.int32, device=dev)
        self.brange = torch.empty((N, 2), dtype=dt, device=dev)
        self.keys = torch.empty((N, m_pad), dtype=dt, device=dev)          # sorted x keys: the rank search of query_order reads them
        # the full rows in sorted order.  Their stride is an argument of dicp_sweep_build (padding a row to one aligned 32-byte sector, stride 8, was
        # measured: 2 % SLOWER end to end -- the copy grows by a third and leaves the cache sooner, docs/HISTORY.md section 5), so rows stay packed
        self.row_stride = c
        self.tgt_s = torch.empty((N, m_pad, self.row_stride), dtype=dt, device=dev) if sorted_rows else None
        nbytes = int(lib.dicp_sweep_sort_scratch_bytes(_DT[dt], N, m_pad))     # float64 keys / more than 16384 slots: chunked sort through scratch
        scratch = torch.empty((nbytes,), dtype=torch.uint8, device=dev) if nbytes else None
        self.first = None
        self.pair_shards = torch.zeros((_lib.PAIR_SHARDS,), dtype=torch.int64, device=dev)
        # float32: the image of the sorted rows for the matrix-core scoring of the plain searches (dicp_knn_sweep's f16_image; big problems only:
        # the form works in units of 128 queries)
        self.img16 = None
        want_img = (bool(F16_SWEEP) and dt == torch.float32 and first_order is not None and float(N) * first_order[0].shape[1] >= F16_SWEEP_MIN_QUERIES
                    and m >= F16_SWEEP_MIN_TARGETS)
        self.form_default = int(m >= F16_SWEEP_MIN_TARGETS)      # the scoring form of a cloud whose slabs have not been measured yet
        self.form0 = None                                        # (N) int32: the first search's tally of tiles per cloud
        if first_order is not None:
            source, T_init, src_rows = first_order
            self.frame = torch.empty((N, 12), dtype=dt, device=dev)
            pose_s = torch.empty((N, 12), dtype=dt, device=dev)
            qorder = torch.empty((N, source.shape[1]), dtype=torch.int32, device=dev)
            with _on(dev):
                _lib.check(lib.dicp_sweep_setup(_DT[dt], _p(tgt), c, _p(tgt_rows), N, m, m_pad, CENTER_QUANTUM, int(FRAME_DIRECTIONS), _p(self.frame), _p(self.keys), _p(self.tperm),
                                                self.NBKT, _p(self.bucket), _p(self.brange), _p(scratch), nbytes, _p(self.tgs4), _p(self.tgt_s), self.row_stride,
                                                _p(source), _p(src_rows), source.shape[1], _p(T_init), _p(pose_s), _p(qorder), _stream()), "dicp_sweep_setup")
            if want_img:
                self.img16 = f16_image(self.tgs4, m, tgt_rows)
            spos0 = None
            if first_search:
                spos0 = torch.empty((N, source.shape[1]), dtype=torch.int32, device=dev)
                if tally and F16_SWEEP_ADAPTIVE and dt == torch.float32 and float(N) * source.shape[1] >= F16_SWEEP_MIN_QUERIES and bool(F16_SWEEP):
                    self.form0 = torch.zeros((N,), dtype=torch.int32, device=dev)     # (that search's tally of its slabs' tiles per cloud: dicp_loop_buffers.search.form)
                with _on(dev):
                    _lib.check(lib.dicp_knn_sweep(_DT[dt], _p(source), _p(pose_s), _p(self.tgs4), _p(self.tperm), _p(qorder), _p(self.bucket), _p(self.brange), self.NBKT,
                                                  _p(src_rows), _p(tgt_rows), N, source.shape[1], m, m_pad, None, _p(spos0), _p(self.pair_shards), 0,
                                                  _p(self.img16) if (self.form_default or not F16_SWEEP_ADAPTIVE) else None, None, _p(self.form0), self.form_default, _stream()), "dicp_knn_sweep")
            self.first = (source, T_init, qorder, spos0)
            return
        with _on(dev):
            _lib.check(lib.dicp_sweep_sort(_DT[dt], _p(tgt), c, _p(frame), _p(tgt_rows), N, m, m_pad, _p(self.keys), _p(self.tperm), self.NBKT,
                                           _p(self.bucket), _p(self.brange), _p(scratch), nbytes, _stream()), "dicp_sweep_sort")
            _lib.check(lib.dicp_sweep_build(_DT[dt], _p(tgt), c, _p(frame), _p(tgt_rows), _p(self.tperm), N, m, m_pad,
                                            _p(self.tgs4), _p(self.tgt_s), self.row_stride, _stream()), "dicp_sweep_build")

    @property
    def pairs(self):
        """(query,target) pairs scored so far (device scalar)."""
        return self.pair_shards.sum()

    def query_order(self, src, pose, exact=False, w=None, copies=False, reproducible=False, spos_prev=None, src_rows=None, pose_prev=None, order_prev=None):
        """Query indices in (approximately) ascending transformed x: keeps a wave's queries neighbours.  Default: a
        counting sort by the rank bucket of each query's x among the sorted target keys (dicp_query_order; equal-width x
        buckets for clouds beyond 16384 points) -- or, given spos_prev (the matches of an earlier iteration), by the rank of
        each query's previous match; exact=True: a full sort of the x keys.
        copies=True -> (qorder, src_s, w_s): also the source rows (and the weights w, if given) in that slot order."""
        N, n, _ = src.shape
        lib = _lib.load()
        with _on(src.device):
            if not exact and pose is not None and pose_prev is not None and order_prev is not None and not copies and spos_prev is None:
                # a re-ordering inside one call: clouds that have hardly moved since `order_prev` was made (under pose_prev) keep it (dicp_query_reorder)
                qorder = torch.empty((N, n), dtype=torch.int32, device=src.device)
                _lib.check(lib.dicp_query_reorder(_DT[src.dtype], _p(src), _p(pose), _p(pose_prev), _p(order_prev), _p(self.brange), self.NBKT, N, n, _p(qorder),
                                                  self.tgs4.shape[1], _p(self.keys), _p(self.bucket), self.m, _p(src_rows), _p(self.tgt_rows), _stream()), "dicp_query_reorder")
                return qorder
            if not exact:
                qorder = torch.empty((N, n), dtype=torch.int32, device=src.device)
                _lib.check(lib.dicp_query_order(_DT[src.dtype], _p(src), _p(pose), _p(self.brange), self.NBKT, N, n, _p(qorder),
                                                None, None, None, int(reproducible), _p(spos_prev), self.tgs4.shape[1],
                                                _p(self.keys), _p(self.bucket), self.m, _p(src_rows), _p(self.tgt_rows), _stream()),
                           "dicp_query_order")
                if copies:      # (the ordering kernel can write them itself, but one block per cloud gathers slowly: 115 vs 16 us)
                    return qorder, _gather_rows_raw(src, qorder), (_gather_rows_raw(w.unsqueeze(-1), qorder).squeeze(-1) if w is not None else None)
                return qorder
            assert src_rows is None, "exact query order: dense batches only (a test / tuning aid)"
            keys = torch.empty((N, n), dtype=src.dtype, device=src.device)
            _lib.check(lib.dicp_query_keys(_DT[src.dtype], _p(src), _p(pose), N, n, _p(keys), _stream()), "dicp_query_keys")
        qorder = torch.argsort(keys, dim=1).to(torch.int32)
        if copies:
            return qorder, _gather_rows_raw(src, qorder), (_gather_rows_raw(w.unsqueeze(-1), qorder).squeeze(-1) if w is not None else None)
        return qorder

    def make_image(self):
        """The split-f16 image of the sorted rows (float32): searches given it score on the matrix cores."""
        if self.img16 is None:
            self.img16 = f16_image(self.tgs4, self.m, self.tgt_rows)
        return self.img16

    def knn(self, src, pose, qorder=None, out=None, cfg=0, spos=None, src_rows=None, mfma=False):
        """src_rows (N) int32: rows of each source cloud that take part (qorder, if any, made with the same counts).
        mfma: score on the matrix cores (float32, the (2,8) configuration's units: cfg 0 on big problems, or 2)."""
        N, n, _ = src.shape
        idx = out if out is not None else torch.empty((N, n), dtype=torch.int32, device=src.device)
        with _on(src.device):
            _lib.check(_lib.load().dicp_knn_sweep(_DT[src.dtype], _p(src), _p(pose), _p(self.tgs4), _p(self.tperm), _p(qorder),
                                                  _p(self.bucket), _p(self.brange), self.NBKT, _p(src_rows), _p(self.tgt_rows), N, n, self.m, self.tgs4.shape[1],
                                                  _p(idx), _p(spos), _p(self.pair_shards), cfg, _p(self.make_image()) if mfma else None, None, None, 1, _stream()), "dicp_knn_sweep")
        return idx


class _GatherRows(torch.autograd.Function):
    """nn.py:37-38: neighbours = y[idx]; backward = scatter-add into y.grad."""

    @staticmethod
    def forward(ctx, y, idx):
        N, m, c = y.shape
        n = idx.shape[1]
        out = torch.empty((N, n, c), dtype=y.dtype, device=y.device)
        with _on(y.device):
            _lib.check(_lib.load().dicp_gather_rows(_DT[y.dtype], _p(y), _p(idx), N, n, m, c, _p(out), _stream()), "dicp_gather_rows")
        ctx.save_for_backward(idx)
        ctx.shape = (N, m, c)
        return out

    @staticmethod
    def backward(ctx, gout):
        (idx,) = ctx.saved_tensors
        N, m, c = ctx.shape
        gout = gout.contiguous()
        gy = torch.zeros((N, m, c), dtype=gout.dtype, device=gout.device)
        with _on(gout.device):
            _lib.check(_lib.load().dicp_scatter_add_rows(_DT[gout.dtype], _p(gout), _p(idx), N, idx.shape[1], m, c, _p(gy), _stream()),
                       "dicp_scatter_add_rows")
        return gy, None


def gather_rows(y, idx):
    return _GatherRows.apply(y.contiguous(), idx)


def _gather_rows_raw(y, idx):
    """y (N,m,c), idx (N,k) int32 -> (N,k,c) = y[b, clamp(idx[b,s])] (dicp_gather_rows, no autograd)."""
    y = y.contiguous()
    N, m, c = y.shape
    k = idx.shape[1]
    out = torch.empty((N, k, c), dtype=y.dtype, device=y.device)
    with _on(y.device):
        _lib.check(_lib.load().dicp_gather_rows(_DT[y.dtype], _p(y), _p(idx), N, k, m, c, _p(out), _stream()), "dicp_gather_rows")
    return out


class _PackList(torch.autograd.Function):
    """A list of clouds (len_i, c_i) -> ONE padded batch (N, n_max, cols) of their first `cols` columns, one launch (dicp_pack_list); the reverse is one launch too
    (dicp_unpack_list into ONE flat buffer: the clouds' gradients are views of it).  What ICP.py:305-511 does with one op per cloud: at 256 clouds ~800 launches
    going in and ~2300 coming back through autograd, 20 ms of host and GPU time per call (profiles/r05_ragged_lists.txt)."""

    @staticmethod
    def forward(ctx, cols, pad, *clouds):
        lib = _lib.load()
        N, dev, dt = len(clouds), clouds[0].device, clouds[0].dtype
        # (the launch reads the clouds through RAW pointers: whatever would make it read foreign memory is refused here, whoever the caller is)
        for t in clouds:
            if not (t.is_cuda and t.device == dev and t.dtype == dt and t.dim() == 2 and t.shape[1] >= cols and (t.shape[0] <= 1 or t.stride(1) == 1) and t.stride(0) >= t.shape[1]):
                raise ValueError("pack_list: every cloud must be a 2-D %s tensor on %s with unit column stride and at least %d columns" % (dt, dev, cols))
        lens = [int(t.shape[0]) for t in clouds]
        n_max = max(lens)
        # pointers, lengths and row strides travel as one small tensor (pageable host memory: the copy is synchronous for the host, 6 KB)
        meta = torch.tensor([t.data_ptr() for t in clouds] + lens + [int(t.stride(0)) for t in clouds], dtype=torch.int64).to(dev)
        lens_d, strides_d = meta[N:2 * N].to(torch.int32), meta[2 * N:].to(torch.int32)
        out = torch.empty((N, n_max, cols), dtype=dt, device=dev)
        with _on(dev):
            _lib.check(lib.dicp_pack_list(_DT[dt], _p(meta), _p(lens_d), _p(strides_d), N, n_max, int(cols), _p(out), _p(pad), _stream()), "dicp_pack_list")
        ctx.geom = (cols, n_max, lens, [int(t.shape[1]) for t in clouds], dt, dev)
        ctx.lens_d = lens_d
        return out

    @staticmethod
    def backward(ctx, gout):
        cols, n_max, lens, widths, dt, dev = ctx.geom
        lib = _lib.load()
        N, es = len(lens), torch.empty((), dtype=dt).element_size()
        sizes = [l * c for l, c in zip(lens, widths)]
        offs = [0]
        for sz in sizes:
            offs.append(offs[-1] + sz)
        flat = torch.empty((offs[-1],), dtype=dt, device=dev)
        meta = torch.tensor([flat.data_ptr() + o * es for o in offs[:-1]] + widths, dtype=torch.int64).to(dev)
        strides_d = meta[N:].to(torch.int32)
        g = gout.contiguous()
        with _on(dev):
            _lib.check(lib.dicp_unpack_list(_DT[dt], _p(g), _p(meta), _p(ctx.lens_d), _p(strides_d), N, n_max, int(cols), max(widths), _stream()), "dicp_unpack_list")
        grads = tuple(flat[offs[i]:offs[i + 1]].view(lens[i], widths[i]) if ctx.needs_input_grad[2 + i] else None for i in range(N))
        return (None, None) + grads


def packable(clouds, widths):
    """A list that dicp_pack_list can take as it stands: non-empty 2-D device tensors of one dtype and device with unit column stride and one of the given widths."""
    first = clouds[0]
    if not isinstance(first, torch.Tensor) or not first.is_cuda or first.dtype not in _DT:
        return False
    return all(isinstance(t, torch.Tensor) and t.is_cuda and t.dim() == 2 and t.shape[0] > 0 and t.shape[1] in widths and t.dtype == first.dtype
               and t.device == first.device and t.stride(1) == 1 for t in clouds)


def pack_list(clouds, cols, pad=None):
    """(N, n_max, cols): the clouds' first `cols` columns, rows past a cloud's own filled with the device scalar `pad` (None: zero).  Differentiable w.r.t. the clouds."""
    return _PackList.apply(int(cols), pad, *clouds)


class _GumbelNN(torch.autograd.Function):
    """nn.__diff_nn_gumbel (nn.py:43-70) through dicp_gumbel_nn / dicp_gumbel_nn_bwd."""

    @staticmethod
    def forward(ctx, x, y, U, seed, eps, tau):
        N, n, _ = x.shape
        m, c = y.shape[1], y.shape[2]
        out = torch.empty((N, n, c), dtype=x.dtype, device=x.device)
        lse = torch.empty((N, n), dtype=x.dtype, device=x.device)
        with _on(x.device):
            _lib.check(_lib.load().dicp_gumbel_nn(_DT[x.dtype], _p(x), _p(y), c, _p(U), seed, float(eps), float(tau), N, n, m,
                                                  _p(out), _p(lse), _stream()), "dicp_gumbel_nn")
        ctx.save_for_backward(x, y, out, lse, *([U] if U is not None else []))
        ctx.cfg = (seed, float(eps), float(tau), U is not None)
        return out

    @staticmethod
    def backward(ctx, gout):
        x, y, out, lse, *rest = ctx.saved_tensors
        seed, eps, tau, has_u = ctx.cfg
        U = rest[0] if has_u else None
        N, n, _ = x.shape
        m, c = y.shape[1], y.shape[2]
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gy = torch.empty_like(y) if ctx.needs_input_grad[1] else None
        if gx is None and gy is None:
            return None, None, None, None, None, None
        with _on(x.device):
            _lib.check(_lib.load().dicp_gumbel_nn_bwd(_DT[x.dtype], _p(x), _p(y), c, _p(U), seed, eps, tau, _p(out), _p(lse),
                                                      _p(gout.contiguous()), N, n, m, _p(gx), _p(gy), _stream()), "dicp_gumbel_nn_bwd")
        return gx, gy, None, None, None, None


def gumbel_nn(x, y, eps, tau, U=None, seed=None):
    """Soft neighbours (N,n,c).  U=None: noise from the in-kernel generator, seeded from torch's CPU generator
    (so torch.manual_seed makes it reproducible)."""
    require_device(x, "gumbel_nn")
    require_device(y, "gumbel_nn")
    if seed is None:
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
    if U is not None:
        U = U.to(device=x.device, dtype=x.dtype).contiguous()
    return _GumbelNN.apply(x.contiguous(), y.contiguous(), U, seed & 0xFFFFFFFF, eps, tau)


def _pose_sums_to_gT(partials, N, dt, dev, st):
    """(N,nblk,NBWD_PAD) per-block sums of [q-bar p^T | s-bar] -> gT (N,4,4): the library's own fixed-order reduction (dicp_pose_grad_out)."""
    lib = _lib.load()
    zero = torch.zeros((N, 12), dtype=torch.float64, device=dev)
    gT = torch.empty((N, 4, 4), dtype=dt, device=dev)
    _lib.check(lib.dicp_pose_grad_out(_DT[dt], _p(zero), _p(partials), partials.shape[1], _p(gT), N, st), "dicp_pose_grad_out")
    return gT


class _TransformPoints(torch.autograd.Function):
    """pc = C p + r for every point (ICP.py:137,274), differentiable w.r.t. the points and the pose: T (N,4,4), or (N,12) [C row-major | r]."""

    @staticmethod
    def forward(ctx, source, T):
        N, n, _ = source.shape
        src = source.contiguous()
        ctx.as_pose = T.dim() == 2
        pose = T.contiguous() if ctx.as_pose else _pose_from_T(T)
        out = torch.empty_like(src)
        with _on(src.device):
            _lib.check(_lib.load().dicp_transform_points(_DT[src.dtype], _p(src), _p(pose), _p(out), N, n, _stream()), "dicp_transform_points")
        ctx.save_for_backward(src, pose)
        return out

    @staticmethod
    def backward(ctx, gout):
        src, pose = ctx.saved_tensors
        N, n, _ = src.shape
        lib = _lib.load()
        gsrc = torch.empty_like(src) if ctx.needs_input_grad[0] else None
        partials = torch.empty((N, lib.dicp_accumulate_blocks(n), _lib.NBWD_PAD), dtype=src.dtype, device=src.device)
        gT = None
        with _on(src.device):
            st = _stream()
            _lib.check(lib.dicp_transform_points_bwd(_DT[src.dtype], _p(src), _p(pose), _p(gout.contiguous()), _p(gsrc), _p(partials),
                                                     N, n, st), "dicp_transform_points_bwd")
            if ctx.needs_input_grad[1]:
                gT = _pose_sums_to_gT(partials, N, src.dtype, src.device, st)
                if ctx.as_pose:
                    gT = _pose_from_T(gT)
        return gsrc, gT


def transform_points(source, T):
    require_device(source, "transform_points")
    return _TransformPoints.apply(source, T)


class _LossWeight(torch.autograd.Function):
    @staticmethod
    def forward(ctx, err2d, loss, diff, metric, tanh_k):
        rows, r = err2d.shape
        w = torch.empty((rows,), dtype=err2d.dtype, device=err2d.device)
        with _on(err2d.device):
            _lib.check(_lib.load().dicp_loss_weight(_DT[err2d.dtype], loss, int(diff), float(metric), float(tanh_k),
                                                    _p(err2d), rows, r, _p(w), _stream()), "dicp_loss_weight")
        ctx.save_for_backward(err2d)
        ctx.cfg = (loss, int(diff), float(metric), float(tanh_k))
        return w

    @staticmethod
    def backward(ctx, gw):
        (err2d,) = ctx.saved_tensors
        loss, diff, metric, tanh_k = ctx.cfg
        rows, r = err2d.shape
        gerr = torch.empty_like(err2d)
        gw = gw.contiguous()
        with _on(err2d.device):
            _lib.check(_lib.load().dicp_loss_weight_bwd(_DT[err2d.dtype], loss, diff, metric, tanh_k, _p(err2d), _p(gw),
                                                        rows, r, _p(gerr), _stream()), "dicp_loss_weight_bwd")
        return gerr, None, None, None, None


def loss_weight(err2d, name, diff, metric, tanh_k):
    require_device(err2d, "loss_weight")
    return _LossWeight.apply(err2d.contiguous(), _LOSS[name], diff, metric, tanh_k)


# --------------------------------------------------------------- the ICP loop
def prebuild_search(source, target, knn_variant, want_rows, T_init=None, src_rows=None, tgt_rows=None, first_search=False, tally=True):
    """Enqueue the per-call search structure of the sweep path (target sort + index build, ~0.15 ms of kernels) NOW, so that
    it runs under the host work the caller still has to do before the loop starts (a call that begins on an idle GPU is
    host-bound until its first long kernel).  Returns (target, SweepIndex) for LoopConfig.prebuilt, or None when the loop
    will not use the sweep.  ICPLoop.forward uses it only if it was built from the very tensor it is given."""
    N, n = source.shape[0], source.shape[1]
    kind = knn_variant & 0xff
    if kind == _lib.KNN_AUTO:
        kind = auto_knn_kind(N, n, target.shape[1])
    if kind != _lib.KNN_SWEEP or not target.is_cuda or not target.is_contiguous() or target.dtype not in _DT:
        return None
    with _on(target.device):
        # ... with the first query order, from T_init alone (the loop's own pose_0 does not exist yet): the queue then holds ~0.2 ms of work while
        # the host builds the loop state.  Frame, sort, rows, search pose and that order are ONE library call.
        if (T_init is not None and T_init.is_cuda and T_init.is_contiguous() and T_init.dtype == target.dtype and source.is_contiguous()
                and source.dtype == target.dtype and tuple(T_init.shape) == (N, 4, 4)):
            sweep = SweepIndex(target, sorted_rows=True, tgt_rows=tgt_rows, first_order=(source, T_init, src_rows),
                               first_search=bool(first_search) and not (knn_variant & 0xff00), tally=tally)
            return (target, sweep, sweep.first)
        sweep = SweepIndex(target, sorted_rows=True, frame=search_frame(target, tgt_rows=tgt_rows), tgt_rows=tgt_rows)
        return (target, sweep, None)


def _pose_from_T(T):
    N = T.shape[0]
    return torch.cat((T[:, :3, :3].reshape(N, 9), T[:, :3, 3]), dim=1).contiguous()


HIST_CHUNK_BYTES = 1 << 29      # per-iteration histories (indices, weights) are allocated in slabs of at most this size


def _segments(Kmax, extra_cuts):
    cuts = sorted(set([0, Kmax] + [c for c in extra_cuts if 0 < c < Kmax]))
    return list(zip(cuts[:-1], cuts[1:]))


class _Arena:
    """Zero-initialised device tensors carved out of one allocation: take() the shapes, finish() -> the tensors."""

    def __init__(self, dev):
        self.dev, self.specs, self.size = dev, [], 0

    def take(self, shape, dtype):
        nbytes = int(torch.empty((), dtype=dtype).element_size())
        for d in shape:
            nbytes *= int(d)
        self.specs.append((self.size, nbytes, shape, dtype))
        self.size += (nbytes + 255) // 256 * 256
        return None

    def finish(self):
        buf = torch.zeros((max(self.size, 1),), dtype=torch.uint8, device=self.dev)
        return [buf[o:o + nb].view(dt).view(shape) for (o, nb, shape, dt) in self.specs]


def _converged_at(pending):
    """pending = (k0, k1, pinned counters, event): K = 1 + the first iteration of [k0, k1) after which no cloud was still moving
    (ICP.py:240,259), or None.  Waits for that segment's copy only."""
    k0, k1, host_cnt, ev = pending
    ev.synchronize()
    zero = (host_cnt[k0:k1] == 0).nonzero()
    return k0 + int(zero[0, 0]) + 1 if zero.numel() else None


class KabschLoop(torch.autograd.Function):
    """Point-to-point ICP with the closed-form SVD step (reference: ICP.pt2pt_dICP_SVD, ICP.py:533-591), batched.

    forward : dicp_kabsch_forward enqueues K x { kNN -> dicp_kabsch_accumulate -> step } back to back, one call per segment
              (segments are cut where the host must act: a new query order of the sweep, or its all-converged check, which
              reads a segment's counters one segment later, like ICPLoop).  Every iterate is the absolute optimum
              T_k = argmin_T sum w |T p - y[idx_k]|^2, so the reference's composed updates telescope to the last one and
              the gradient is ONE Kabsch adjoint through (source, target[idx_K], weight).  A cloud that meets the tolerance is
              frozen on device at that pose (every pair stops where a call of its own would, ICP.py:585-586).
    Inputs : source (N,n,3), target (N,m,3|6), T_init (N,4,4) [the starting pose of the search], w0 (N,n)
    Outputs: T (N,4,4) differentiable; costs (N,K), iterations (N) non-differentiable.
    """

    @staticmethod
    def forward(ctx, source, target, T_init, w0, max_iterations, tolerance, trim_dist, const_iter, knn_variant, src_rows=None, tgt_rows=None, sync_every=None):
        for t, nm in ((source, "source"), (target, "target"), (T_init, "T_init"), (w0, "weight")):
            require_device(t, "pt2pt_dICP_SVD(" + nm + ")")
        lib = _lib.load()
        dev, dt = source.device, source.dtype
        code = _DT[dt]
        N, n, _ = source.shape
        m, c = target.shape[1], target.shape[2]
        src, tgt, w0c = source.contiguous(), target.contiguous(), w0.contiguous()
        trim_on = int(trim_dist is not None and trim_dist >= 0.0)
        trim = float(trim_dist) if trim_on else 0.0
        Kmax = int(max_iterations)
        assert Kmax >= 1, "max_iterations must be at least 1"
        with _on(dev):
            st = _stream()
            kind = knn_variant & 0xff
            if kind == _lib.KNN_AUTO:
                kind = auto_knn_kind(N, n, m)
            center = search_frame(tgt, tgt_rows=tgt_rows)       # the search frame, as in ICPLoop
            sweep = SweepIndex(tgt, frame=center, tgt_rows=tgt_rows) if kind == _lib.KNN_SWEEP else None
            tgt4 = sweep.tgs4 if sweep is not None else pack_target(tgt, center, tgt_rows)
            img16 = f16_image(tgt4, m, tgt_rows) if kind == _lib.KNN_MFMA else None
            nblk = lib.dicp_accumulate_blocks(n)
            pose = _pose_from_T(T_init)
            pose_s = search_pose(pose, center)
            pose_used = torch.empty_like(pose)
            partials = torch.empty((N, nblk, _lib.NACC_PAD), dtype=dt, device=dev)
            save = torch.empty((N, _lib.KAB_SAVE), dtype=torch.float64, device=dev)
            idx = torch.empty((N, n), dtype=torch.int32, device=dev)
            rows_live = src_rows.clone() if src_rows is not None else torch.full((N,), n, dtype=torch.int32, device=dev)
            arena = _Arena(dev)
            costs = arena.take((N, Kmax), dt)
            iterations = arena.take((N,), dt)
            counters = arena.take((Kmax,), torch.int32)
            costs, iterations, counters = arena.finish()
            cuts = [0, 1, 2] if sweep is not None else []
            if not const_iter:
                every = sync_every if sync_every is not None else (1 if float(N) * n * m >= SWEEP_MIN_PAIRS else 4)
                cuts += list(range(0, Kmax, max(1, int(every))))
            qorder, K = None, Kmax
            pending, host_cnt = None, None
            for (k0, k1) in _segments(Kmax, cuts):
                if sweep is not None and k0 < 2:
                    qorder = sweep.query_order(src, pose_s, src_rows=rows_live)
                KB = _lib.KabschBuffers(
                    src=_p(src), tgt=_p(tgt), w_init=_p(w0c), c=c, K=Kmax, knn_variant=kind | (knn_variant & 0xff00), m_pad=tgt4.shape[1],
                    tgt4=_p(tgt4), tperm=_p(sweep.tperm) if sweep else None, qorder=_p(qorder), bucket=_p(sweep.bucket) if sweep else None,
                    brange=_p(sweep.brange) if sweep else None, nbkt=SweepIndex.NBKT, pairs=_p(sweep.pair_shards) if sweep else None,
                    frame=_p(center), pose=_p(pose), pose_search=_p(pose_s), pose_used=_p(pose_used), idx=_p(idx), partials=_p(partials),
                    save=_p(save), costs=_p(costs), iterations=_p(iterations), rows_live=_p(rows_live), tgt_rows=_p(tgt_rows), counters=_p(counters), tgt_f16=_p(img16))
                _lib.check(lib.dicp_kabsch_forward(code, ctypes.byref(KB), N, n, m, trim_on, trim, int(const_iter), float(tolerance), k0, k1, st),
                           "dicp_kabsch_forward")
                if not const_iter:      # ICP.py:585-586 for the batch: stop once every pair has stopped (frozen clouds make the overshoot a no-op)
                    if pending is not None and _converged_at(pending) is not None:
                        K = _converged_at(pending)
                        pending = None
                        break
                    if host_cnt is None:
                        host_cnt = torch.empty((Kmax,), dtype=torch.int32, pin_memory=True)
                    host_cnt[k0:k1].copy_(counters[k0:k1], non_blocking=True)
                    seg_done = torch.cuda.Event()
                    seg_done.record()
                    pending = (k0, k1, host_cnt, seg_done)
            if pending is not None and _converged_at(pending) is not None:
                K = _converged_at(pending)
            iterations = torch.where(iterations == 0, torch.full_like(iterations, K), iterations)
            T = torch.zeros((N, 4, 4), dtype=dt, device=dev)
            T[:, :3, :3] = pose[:, :9].reshape(N, 3, 3)
            T[:, :3, 3] = pose[:, 9:]
            T[:, 3, 3] = 1.0
            costs = costs[:, :K].contiguous()
        ctx.save_for_backward(src, tgt, w0c, idx, pose_used, save)
        ctx.trim = (trim_on, trim)
        ctx.src_rows = src_rows
        ctx.mark_non_differentiable(costs, iterations)
        return T, costs, iterations

    @staticmethod
    def backward(ctx, gT, *_unused):
        src, tgt, w0c, idx, pose_prev, save = ctx.saved_tensors
        trim_on, trim = ctx.trim
        lib = _lib.load()
        dev, dt = src.device, src.dtype
        code = _DT[dt]
        N, n, _ = src.shape
        m, c = tgt.shape[1], tgt.shape[2]
        with _on(dev):
            st = _stream()
            gT = gT.contiguous()
            gpose = torch.cat((gT[:, :3, :3].reshape(N, 9), gT[:, :3, 3]), dim=1).contiguous()
            gacc = torch.empty((N, 16), dtype=dt, device=dev)
            _lib.check(lib.dicp_kabsch_step_bwd(code, _p(gpose), _p(save), _p(gacc), N, st), "dicp_kabsch_step_bwd")
            gsrc = torch.zeros_like(src)
            gtgt = torch.zeros_like(tgt) if ctx.needs_input_grad[1] else None
            gw = torch.zeros_like(w0c)
            _lib.check(lib.dicp_kabsch_bwd(code, _p(src), _p(tgt), c, _p(idx), _p(pose_prev), _p(w0c), trim_on, trim, _p(gacc), _p(ctx.src_rows),
                                           N, n, m, _p(gsrc), _p(gtgt), _p(gw), st), "dicp_kabsch_bwd")
        return gsrc, gtgt, None, gw, None, None, None, None, None, None, None, None


