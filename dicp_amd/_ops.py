"""Tensor-level wrappers over the C ABI and the autograd glue.

PyTorch-ROCm is plumbing here (device memory, streams, the autograd graph); every
arithmetic step of the ICP iteration runs in libdicp_hip.so.  All functions require
HIP-device tensors and raise otherwise -- there is no CPU compute path.
"""
import os
import ctypes
from dataclasses import dataclass

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
                                            # slab lengths (dicp_loop_buffers.sweep_form; FORM_TILES) -- clouds that start metres off, or a third of which has no counterpart
FORM_PLAN_ITERS = 6                         # iterations whose tallies a reporting call hands to the next calls' plan (dicp_loop_buffers.sweep_form_plan); later ones follow the last
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
        kernel at the benchmark shape under which the host prepares the loop (dicp_loop_buffers.first_search_done)."""
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
                    self.form0 = torch.zeros((N,), dtype=torch.int32, device=dev)     # (that search's tally of its slabs' tiles per cloud: dicp_loop_buffers.sweep_form)
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

    def query_order(self, src, pose, exact=False, w=None, copies=False, reproducible=False, spos_prev=None, src_rows=None):
        """Query indices in (approximately) ascending transformed x: keeps a wave's queries neighbours.  Default: a
        counting sort by the rank bucket of each query's x among the sorted target keys (dicp_query_order; equal-width x
        buckets for clouds beyond 16384 points) -- or, given spos_prev (the matches of an earlier iteration), by the rank of
        each query's previous match; exact=True: a full sort of the x keys.
        copies=True -> (qorder, src_s, w_s): also the source rows (and the weights w, if given) in that slot order."""
        N, n, _ = src.shape
        lib = _lib.load()
        with _on(src.device):
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
def form_tally_wanted(rec, have_image):
    """Whether a call's plain searches should tally their slabs' tiles per cloud (dicp_loop_buffers.sweep_form): when this call REPORTS (the first two calls of a shape
    and every sixteenth: CallHints.form_record), or when its searches choose each cloud's scoring form by the tallies -- the matrix-core image exists (or the record
    says the shape's slabs are long: it will) and no earlier report has given the shape a plan yet.  The tally is an atomic add per unit of the sweep onto one word
    per cloud: 0.03 ms of the search near the pose, 1.6 % of the benchmark's call (round 5, A/B on one box)."""
    if rec is None:
        return bool(have_image)
    reporting = rec["event"] is None and (rec["calls"] < 2 or rec["calls"] % 16 == 0)
    return bool(reporting or ((have_image or rec["long"]) and not rec.get("plan")))


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
                                          "~0.5 s); the gradients of that pass were poisoned with NaN.  Re-run it, or set ICP._tuning['bwd_tail'] = False")


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
    bwd_tail: bool = True         # truncated reverse sweep: the iterations before the ones the previous call still worked at run as ONE launch (dicp_loop_buffers.bwd_tail_from)
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
        if tail_from > 0:
            stats["bwd_tail_error"] = tail_word
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
        for t, nm in ((source, "source"), (target, "target"), (T_init, "T_init")) + (((w0, "weight"),) if w0 is not None else ()):
            require_device(t, "ICP(" + nm + ")")
        lib = _lib.load()
        dev, dt = source.device, source.dtype
        code, es = _DT[dt], source.element_size()
        N, n, _ = source.shape
        m, c = target.shape[1], target.shape[2]
        # w0 None = unit weights (weight=None on a tensor input): the kernels take w_init == NULL and read 4 bytes per point less
        src, tgt, w0c = source.contiguous(), target.contiguous(), (w0.contiguous() if w0 is not None else None)
        P = cfg.params()
        rows = 3 if cfg.icp_type == "pt2pt" else 1
        Kmax = int(cfg.max_iterations)
        assert Kmax >= 1, "max_iterations must be at least 1"
        need_grad = any(ctx.needs_input_grad[:4])
        if cfg.stats_out is not None:       # the statistics describe THIS call (an earlier call's certificate counters must not outlive it)
            for key in ("knn_pairs", "searched_again", "budgets", "bwd_live", "certs_off"):
                cfg.stats_out.pop(key, None)
        ctx.set_materialize_grads(False)    # no zero tensors for the six non-differentiable outputs (168 MB for the weights)

        with _on(dev):
            st = _stream()
            kind = cfg.knn_variant & 0xff
            if cfg.gumbel is not None:
                kind = _lib.KNN_GUMBEL
            elif kind == _lib.KNN_AUTO:
                kind = auto_knn_kind(N, n, m)
            owned = kind == _lib.KNN_SWEEP and need_grad and cfg.bwd_window
            sweep = None
            if kind == _lib.KNN_SWEEP:
                pre = cfg.prebuilt
                if (pre is not None and pre[0].data_ptr() == tgt.data_ptr() and pre[0].shape == tgt.shape and pre[0].dtype == tgt.dtype
                        and pre[1].tgt_s is not None and pre[1].tgt_rows is cfg.tgt_rows):
                    sweep = pre[1]                           # started by the caller, under its host work
                else:
                    with_q = T_init.dtype == dt and tuple(T_init.shape) == (N, 4, 4)
                    sweep = SweepIndex(tgt, sorted_rows=True, tgt_rows=cfg.tgt_rows,
                                       frame=search_frame(tgt, tgt_rows=cfg.tgt_rows, src=src if with_q else None, T_init=T_init if with_q else None, src_rows=cfg.src_rows))
            # the searches run in the target cloud's search frame (dicp_search_frame): packed rows Q y + t, pose [Q C | Q r + t]
            soft = kind == _lib.KNN_GUMBEL        # soft correspondences: no search structure at all
            center = sweep.frame if sweep is not None else (None if soft else search_frame(tgt, tgt_rows=cfg.tgt_rows))
            tgt4 = sweep.tgs4 if sweep is not None else (None if soft else pack_target(tgt, center, cfg.tgt_rows))
            m_pad = tgt4.shape[1] if tgt4 is not None else 0
            # the matrix-core searches' image of the packed rows (the sweep path: of the sorted rows, made with the index)
            img16 = f16_image(tgt4, m, cfg.tgt_rows) if kind == _lib.KNN_MFMA else (sweep.img16 if sweep is not None else None)
            # The per-cloud choice of the scoring form (dicp_loop_buffers.sweep_form).  Every plain search tallies its slabs' tiles per cloud; what the tallies of
            # the PREVIOUS call of this shape said (a hint, like the tail's and the certificates': it arrives through pinned memory, costs time at worst) decides
            # whether this call builds the matrix-core image for clouds too small to get one by size: where some cloud's slabs were long, the plain searches
            # launch both forms and each takes its clouds (start poses a metre off, a third of the source without counterpart: 28.2 -> 19.0 ms per call,
            # profiles/r05_independent_forms.txt); where none was, nothing is built and nothing more is launched.
            # (clouds that get the image by their size score on the matrix cores in every plain search, as in round 4: there the per-cloud choice -- it sends a
            #  cloud whose slabs have become short back to the vector form -- cost 8-10 % of a 64 x 65536 call, profiles/r05_form_tally.txt)
            tally = (sweep is not None and F16_SWEEP and F16_SWEEP_ADAPTIVE and dt == torch.float32 and float(N) * n >= F16_SWEEP_MIN_QUERIES
                     and not (cfg.knn_variant & 0xff00) and m < F16_SWEEP_STATIC_TARGETS)
            form_hint = None
            if tally and cfg.hints is not None and not torch.cuda.is_current_stream_capturing():
                form_hint = cfg.hints.form_record(dev, (N, n, m, dt))
                if form_hint["event"] is not None and form_hint["event"].query():
                    rep = form_hint["host"].tolist()
                    form_hint["long"] = bool(4 * rep[0] >= N)       # (a quarter of the clouds: the image costs every call 0.03 ms)
                    form_hint["moving"] = bool(4 * rep[1] >= N)     # ... still had long slabs in the call's LAST plain search: they keep moving
                    # the plan of the next calls: iteration k + 1 scores on the matrix cores if most clouds' slabs were long in iteration k's plain search
                    # (-1: no plain search then -- a certified iteration -- : the loop decides as it does without a plan)
                    form_hint["plan"] = [0] + [0 if c < 0 else (2 if 2 * c >= N else 1) for c in rep[2:2 + FORM_PLAN_ITERS]]
                    form_hint["event"] = None
                if img16 is None and form_hint["long"]:
                    img16 = sweep.make_image()
            form_plan = None
            if tally and img16 is not None and form_hint is not None and form_hint.get("plan"):
                # one form per iteration for the whole batch, from an earlier call's tallies (iterations beyond the report: as the last reported one)
                pl = form_hint["plan"]
                form_plan = (ctypes.c_int32 * Kmax)(*[(pl[k] if k < len(pl) else pl[-1]) for k in range(Kmax)])
            if tally and not form_tally_wanted(form_hint, img16 is not None):
                tally = False           # (nobody reads this call's tallies: no per-cloud choice inside it, no report after it -- the searches do not take them)
            nblk = lib.dicp_accumulate_blocks(n)
            poses = torch.empty((Kmax + 1, N, 12), dtype=dt, device=dev)
            poses_c = torch.empty((Kmax + 1, N, 12), dtype=dt, device=dev) if center is not None else None   # [Q C | Q r + t]: what the searches read
            alive = torch.empty((Kmax + 1, N), dtype=dt, device=dev)
            areg = torch.empty((Kmax, N, 36), dtype=torch.float64, device=dev) if need_grad else None
            n_start = torch.empty((N,), dtype=dt, device=dev)
            partials = torch.empty((N, nblk, _lib.NACC_PAD), dtype=dt, device=dev)
            # every zero-initialised piece of loop state comes out of ONE zeroed arena (one fill instead of seven)
            arena = _Arena(dev)
            deltas = arena.take((N, Kmax, 6), dt)
            costs = arena.take((N, Kmax), dt)
            converged = arena.take((N,), torch.uint8)
            iterations = arena.take((N,), dt)
            matched = arena.take((N,), dt)
            n_matched = arena.take((N,), dt)
            counters = arena.take((Kmax,), torch.int32)
            # match certificates (sweep path): a motion budget per query, a filter value per unit of the sweep, per-cloud motion bounds, and
            # what each iteration searched again.  The searches before the LAST re-ordering of the queries run plain (a certifying search costs
            # a quarter more, and its budgets would not survive the steps of the first iterations).
            keep_idx = (sweep is None or (need_grad and not owned)) and not soft     # (original indices: the brute-force searches and the atomic backward)
            # (the one certifying search costs a quarter more than a plain one: it takes three certified iterations to be worth it)
            resorts = [k for k in cfg.sweep_resort if 0 <= k < Kmax]
            cert_from = (max(resorts) if resorts else 0) if cfg.cert_from is None else max(0, int(cfg.cert_from))     # iteration of the certifying search
            # (... and the point-iterations they can save must outweigh what they cost the host in buffers and set-up: measured break-even, forward +
            #  backward, at ~2 M certified point-iterations -- 32 x 4096 x 10 iterations loses 6 %, x 20 iterations wins 6 %; profiles/r03_certificates_mid_sizes.txt)
            want_certs = sweep is not None and not (cfg.knn_variant & 0xff00) and not keep_idx and certificates_pay(cfg.reuse_matches, Kmax, cert_from, N, n)
            # Where EVERY cloud of the previous call of this shape ended with its certificates switched off (near-duplicated or duplicated targets: no
            # match can be proven; poses that keep moving: no budget survives), the next calls do not try: they search plainly -- the same results, without the certifying search and the guard
            # launches -- and after 32 calls they try again.  The previous call's switch states arrive through pinned memory, like the tail's hint.
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
            # (... or, in a call without certificates, the form hint: a quarter of the clouds of this shape had long slabs in a plain search of the previous call)
            clouds_moving = bool(clouds_moving or (not want_certs and form_hint is not None and form_hint.get("moving", False)))
            arena.take((Kmax, 128) if want_certs else (0,), torch.int32)
            arena.take((N, 8) if want_certs else (0,), torch.int32)
            arena.take((N, n) if want_certs else (0,), torch.int32)     # (row cache: matches a guard launch leaves for the accumulate of its iteration; zero = none)
            arena.take((Kmax + 1, 8) if want_certs else (0,), torch.int32)     # (lengths of the guard launches' work lists, per iteration)
            # per-cloud tallies of the plain searches' slab lengths: the scoring form of the next plain search (dicp_loop_buffers.sweep_form)
            adaptive = bool(tally) and kind == _lib.KNN_SWEEP
            arena.take((Kmax, N) if adaptive else (0,), torch.int32)
            arena.take((N,) if (want_certs and cfg.cert_sets) else (0,), torch.int32)     # (lengths of the clouds' candidate-set lists)
            deltas, costs, converged, iterations, matched, n_matched, counters, cert_count, cert_cloud, cert_pend, cert_gcount, sweep_form, cert_scount = arena.finish()
            # pose_0, alive_0, n_start (ICP.py:124-129)
            certs = None
            if want_certs:
                units = (n + 63) // 64          # (units of the sweep's one-query-per-lane forms; the two-query form uses half of them)
                # (row cache of the certified iterations: the matched row of every query, a filter per 64 queries, and -- with gradients -- where each such
                #  group's matches lie in the history, which those iterations keep by reference: dicp_loop_buffers.spos_of)
                certs = dict(q=torch.empty((N, n), dtype=dt, device=dev), qu=torch.empty((N, units), dtype=dt, device=dev), count=cert_count,
                             nbr=torch.empty((N, n, 6 if cfg.icp_type == "pt2pl" else 3), dtype=dt, device=dev), gdirty=torch.empty((N, units), dtype=torch.int32, device=dev),
                             cm=torch.empty((N, n), dtype=torch.int32, device=dev), glist=torch.empty((8, max(N, 2) * units), dtype=torch.int32, device=dev), gcount=cert_gcount,
                             slist=torch.empty((N, n), dtype=torch.int32, device=dev) if cfg.cert_sets else None, scount=cert_scount if cfg.cert_sets else None,
                             pend=cert_pend, of=torch.empty((Kmax + 1, N, units), dtype=torch.int32, device=dev) if need_grad else None,
                             set=torch.empty((N * n * (es + 16),), dtype=torch.uint8, device=dev) if cfg.cert_sets else None,     # candidate sets: (N,n) budgets + (N,n,4) rows
                             rmax=torch.empty((N, 4), dtype=dt, device=dev), dcum=torch.empty((N, 2 * (Kmax + 1)), dtype=dt, device=dev))
            _lib.check(lib.dicp_loop_init(code, _p(T_init.contiguous()), _p(w0c), float(cfg.match_ratio_thresh), rows, N, n,
                                          _p(poses), _p(alive), _p(n_start), _p(center), _p(poses_c),
                                          _p(src) if certs else None, _p(certs["rmax"]) if certs else None, _p(certs["dcum"]) if certs else None, 2 * (Kmax + 1), st),
                       "dicp_loop_init")
            # sweep path: the matches are kept as SORTED positions (spos) -- accumulate gathers the sorted, sector-aligned rows with them and
            # the windowed backward consumes them; original indices (idx) are only kept for the brute-force searches and the atomic backward
            idx_once = torch.empty((N, n), dtype=torch.int32, device=dev) if (keep_idx and not need_grad) else None
            keep_spos = sweep is not None and need_grad       # per-iteration sorted positions (the windowed backward reads them; same layout as idx)
            spos_once = torch.empty((N, n), dtype=torch.int32, device=dev) if (sweep is not None and not need_grad) else None

            # histories in slabs of kc iterations: slab j covers iterations [j*kc, (j+1)*kc)
            per_iter = N * n * max(es, 4)
            kc = max(1, min(Kmax, HIST_CHUNK_BYTES // max(1, per_iter)))
            w_slabs, idx_slabs, spos_slabs = [], [], []
            cuts = list(range(0, Kmax, kc))
            if sweep is not None:
                cuts += list(cfg.sweep_resort) + ([cert_from] if certs is not None else [])
            if not cfg.const_iter:
                every = cfg.sync_every
                if every is None:
                    every = 1 if float(N) * n * m >= SWEEP_MIN_PAIRS else 4
                cuts += list(range(0, Kmax, max(1, int(every))))
            ev = cfg.timing_events
            events = ev.handles(Kmax) if ev is not None else None

            gum = None
            if soft:    # Gumbel-softmax correspondences: the neighbour ROWS of every iteration and their log-sum-exp are the history the reverse sweep reads
                g_eps, g_tau, inject_U = cfg.gumbel
                nbr_hist = torch.empty((Kmax, N, n, c), dtype=dt, device=dev)
                lse_hist = torch.empty((Kmax, N, n), dtype=dt, device=dev)
                ps_t = torch.empty((N, n, 3), dtype=dt, device=dev)
                U_list = [u.to(device=dev, dtype=dt).contiguous() for u in inject_U[:Kmax]] if inject_U is not None else None
                assert U_list is None or len(U_list) >= Kmax, "one injected noise tensor per iteration"
                U_arr = (ctypes.c_void_p * Kmax)(*[u.data_ptr() for u in U_list]) if U_list is not None else None
                # in-kernel noise: one seed per iteration from torch's CPU generator (torch.manual_seed makes a call reproducible)
                seed_list = [int(v) & 0xFFFFFFFF for v in torch.randint(0, 2 ** 31 - 1, (Kmax,)).tolist()] if U_list is None else [0] * Kmax
                seeds = (ctypes.c_uint32 * Kmax)(*seed_list)
                gum = _lib.GumbelLoop(U=ctypes.cast(U_arr, ctypes.c_void_p) if U_arr is not None else None, seeds=ctypes.cast(seeds, ctypes.c_void_p),
                                      eps=float(g_eps), tau=float(g_tau), ps_t=_p(ps_t), nbr=_p(nbr_hist), lse=_p(lse_hist))
            qorders, seg_q = [], []          # distinct query orders of the sweep and which one each segment used
            qorder = None
            K = Kmax
            segs = _segments(Kmax, cuts)
            done_segs = []
            pending, host_cnt = None, None   # tolerance mode: the segment whose convergence counters are still in flight
            LB, LBref, Pref = None, None, ctypes.byref(P)
            # iteration 0's search may already be running: prebuild_search enqueued it behind the index build (under T_init's search pose and the
            # first query order), so that the GPU has 0.4 ms of work while this function prepares the loop (dicp_loop_buffers.first_search_done)
            first0 = cfg.prebuilt[2] if (sweep is not None and cfg.prebuilt is not None and cfg.prebuilt[1] is sweep) else None
            first_spos = None
            if (first0 is not None and len(first0) > 3 and first0[3] is not None and first0[0].data_ptr() == src.data_ptr() and first0[0].shape == src.shape
                    and first0[1].data_ptr() == T_init.data_ptr() and T_init.is_contiguous() and not keep_idx and events is None
                    and not (certs is not None and cert_from <= 0)):
                first_spos = first0[3]
                if adaptive and sweep.form0 is not None:
                    sweep_form[0].copy_(sweep.form0)        # (that search's tally of its slabs: the scoring form of iteration 1's)
                if not keep_spos:
                    spos_once = first_spos
            # constant-iteration calls of the sweep path with all histories in one slab: every segment and the query re-orderings between them
            # behind ONE library call (dicp_icp_forward_plan) -- per segment the host spent ~40 us, which a mid-size call does not have
            if cfg.const_iter and sweep is not None and kc >= Kmax and len(segs) <= _lib.MAX_SEGMENTS and cfg.plan_call:
                w_slabs.append(torch.empty((N, kc, n), dtype=dt, device=dev))
                if need_grad and keep_idx:
                    idx_slabs.append(torch.empty((Kmax, N, n), dtype=torch.int32, device=dev))
                if keep_spos:
                    spos_slabs.append(torch.empty((Kmax, N, n), dtype=torch.int32, device=dev))
                    if first_spos is not None:
                        spos_slabs[0][0].copy_(first_spos)
                first = cfg.prebuilt[2] if (cfg.prebuilt is not None and cfg.prebuilt[1] is sweep) else None
                have_first = (first is not None and first[0].data_ptr() == src.data_ptr() and first[0].shape == src.shape
                              and first[1].data_ptr() == T_init.data_ptr() and T_init.is_contiguous())
                SP = _lib.SegmentPlan(nseg=len(segs), cert_from=cert_from if certs is not None else -1, keys=_p(sweep.keys),
                                      cert_q=_p(certs["q"]) if certs else None, cert_qu=_p(certs["qu"]) if certs else None,
                                      cert_count=_p(certs["count"]) if certs else None, cert_cloud=_p(cert_cloud) if (certs and cfg.cert_backoff) else None,
                                      cert_set=_p(certs["set"]) if certs else None, cert_nbr=_p(certs["nbr"]) if certs else None,
                                      cert_gdirty=_p(certs["gdirty"]) if certs else None, cert_pend=_p(certs["pend"]) if certs else None,
                                      cert_cm=_p(certs["cm"]) if certs else None, cert_glist=_p(certs["glist"]) if certs else None,
                                      cert_gcount=_p(certs["gcount"]) if certs else None, cert_slist=_p(certs["slist"]) if certs else None,
                                      cert_scount=_p(certs["scount"]) if certs else None)
                n_new = sum(1 for (k0, _) in segs if (k0 == 0 or k0 in cfg.sweep_resort)) - (1 if have_first else 0)
                fresh_orders = torch.empty((max(n_new, 1), N, n), dtype=torch.int32, device=dev)
                used = 0
                for si, (k0, k1) in enumerate(segs):
                    SP.k0[si], SP.k1[si] = k0, k1
                    if k0 == 0 or k0 in cfg.sweep_resort:
                        if k0 == 0 and have_first:
                            qorder, SP.new_order[si] = first[2], 0
                        else:
                            qorder, SP.new_order[si] = fresh_orders[used], 1
                            used += 1
                        qorders.append(qorder)
                    else:
                        SP.new_order[si] = 0
                    SP.order[si] = qorder.data_ptr()
                    seg_q.append(len(qorders) - 1)
                    done_segs.append((k0, k1))
                LB = _lib.LoopBuffers(
                    src=_p(src), tgt=_p(tgt), w_init=_p(w0c), c=c, K=Kmax, knn_variant=kind | (cfg.knn_variant & 0xff00) | ((0 if cfg.small_loop else 1) << 25), m_pad=m_pad,
                    tgt4=_p(tgt4), tperm=_p(sweep.tperm), bucket=_p(sweep.bucket), brange=_p(sweep.brange),
                    nbkt=SweepIndex.NBKT, idx_per_iter=int(need_grad), pairs=_p(sweep.pair_shards),
                    poses=_p(poses), deltas=_p(deltas), costs=_p(costs), areg=_p(areg), alive=_p(alive), converged=_p(converged),
                    iterations=_p(iterations), matched_ratio=_p(matched), n_start=_p(n_start), n_matched=_p(n_matched),
                    tgt_sorted=_p(sweep.tgt_s), tgt_sorted_stride=sweep.row_stride,
                    rmax=_p(certs["rmax"]) if certs else None, dcum=_p(certs["dcum"]) if certs else None,
                    w_iter=n, w_stride=kc * n, w=_p(w_slabs[0]),
                    spos=_p(spos_slabs[0]) if keep_spos else _p(spos_once),
                    idx=(_p(idx_slabs[0]) if need_grad else _p(idx_once)) if keep_idx else None,
                    partials=_p(partials), counters=_p(counters), events=events, frame=_p(center), poses_search=_p(poses_c),
                    src_rows=_p(cfg.src_rows), tgt_rows=_p(cfg.tgt_rows), first_search_done=int(first_spos is not None), tgt_f16=_p(img16),
                    spos_of=_p(certs["of"]) if certs else None, sweep_form=_p(sweep_form) if adaptive else None, sweep_form_default=sweep.form_default,
                    sweep_form_plan=ctypes.cast(form_plan, ctypes.c_void_p) if form_plan is not None else None)
                _lib.check(lib.dicp_icp_forward_plan(code, Pref, ctypes.byref(LB), ctypes.byref(SP), N, n, m, int(cfg.dim), 1, float(cfg.tolerance), st),
                           "dicp_icp_forward_plan")
                segs = []
            for (k0, k1) in segs:
                j = k0 // kc
                if j == len(w_slabs):
                    kk = min(kc, Kmax - j * kc)
                    w_slabs.append(torch.empty((N, kc, n), dtype=dt, device=dev))     # (N,K,n) layout as returned; same cloud stride in every slab
                    if need_grad and keep_idx:
                        idx_slabs.append(torch.empty((kk, N, n), dtype=torch.int32, device=dev))
                    if keep_spos:
                        spos_slabs.append(torch.empty((kk, N, n), dtype=torch.int32, device=dev))
                        if j == 0 and first_spos is not None:
                            spos_slabs[0][0].copy_(first_spos)
                new_order = False
                use_certs = certs is not None and k0 >= cert_from
                if sweep is not None and (qorder is None or k0 in cfg.sweep_resort):
                    new_order = True
                    # queries re-ordered by x under the current pose
                    first = cfg.prebuilt[2] if (k0 == 0 and sweep is not None and cfg.prebuilt is not None and cfg.prebuilt[1] is sweep) else None
                    if (first is not None and first[0].data_ptr() == src.data_ptr() and first[0].shape == src.shape
                            and first[1].data_ptr() == T_init.data_ptr() and T_init.is_contiguous()):
                        qorder = first[2]                    # ordered under T_init by the caller (prebuild_search)
                    else:
                        qorder = sweep.query_order(src, (poses_c if poses_c is not None else poses)[k0], src_rows=cfg.src_rows)
                    qorders.append(qorder)
                seg_q.append(len(qorders) - 1)
                base = j * kc                                         # virtual bases: slab pointer minus its first iteration
                if LB is None:      # the fields that do not change from segment to segment (building the struct is ~20 us of host time)
                    LB = _lib.LoopBuffers(
                        src=_p(src), tgt=_p(tgt), w_init=_p(w0c), c=c, K=Kmax, knn_variant=kind | (cfg.knn_variant & 0xff00) | ((0 if cfg.small_loop else 1) << 25), m_pad=m_pad,
                        tgt4=_p(tgt4), tperm=_p(sweep.tperm) if sweep else None,
                        bucket=_p(sweep.bucket) if sweep else None, brange=_p(sweep.brange) if sweep else None,
                        nbkt=SweepIndex.NBKT, idx_per_iter=int(need_grad), pairs=_p(sweep.pair_shards) if sweep else None,
                        poses=_p(poses), deltas=_p(deltas), costs=_p(costs), areg=_p(areg), alive=_p(alive), converged=_p(converged),
                        iterations=_p(iterations), matched_ratio=_p(matched), n_start=_p(n_start), n_matched=_p(n_matched),
                        tgt_sorted=_p(sweep.tgt_s) if sweep is not None else None, tgt_sorted_stride=sweep.row_stride if sweep is not None else 0,
                        rmax=_p(certs["rmax"]) if certs else None, dcum=_p(certs["dcum"]) if certs else None,
                        w_iter=n, w_stride=kc * n,
                        partials=_p(partials), counters=_p(counters), events=events, frame=_p(center), poses_search=_p(poses_c),
                        src_rows=_p(cfg.src_rows), tgt_rows=_p(cfg.tgt_rows), tgt_f16=_p(img16),
                        sweep_form=_p(sweep_form) if adaptive else None, sweep_form_default=sweep.form_default if sweep is not None else 0,
                        sweep_form_plan=ctypes.cast(form_plan, ctypes.c_void_p) if form_plan is not None else None)
                    if gum is not None:
                        LB.gumbel = ctypes.cast(ctypes.pointer(gum), ctypes.c_void_p)
                    LBref = ctypes.byref(LB)
                LB.qorder = _p(qorder)
                LB.first_search_done = int(k0 == 0 and first_spos is not None)
                LB.spos = ctypes.c_void_p(spos_slabs[j].data_ptr() - base * N * n * 4) if keep_spos else _p(spos_once)
                LB.idx = (ctypes.c_void_p(idx_slabs[j].data_ptr() - base * N * n * 4) if need_grad else _p(idx_once)) if keep_idx else None
                LB.cert_q, LB.cert_qu, LB.cert_count = (_p(certs["q"]), _p(certs["qu"]), _p(certs["count"])) if use_certs else (None, None, None)
                LB.cert_cloud = _p(cert_cloud) if (use_certs and cfg.cert_backoff) else None
                LB.cert_set = _p(certs["set"]) if use_certs else None
                LB.cert_nbr, LB.cert_gdirty, LB.cert_pend, LB.cert_cm = (_p(certs["nbr"]), _p(certs["gdirty"]), _p(certs["pend"]), _p(certs["cm"])) if use_certs else (None, None, None, None)
                LB.cert_glist, LB.cert_gcount = (_p(certs["glist"]), _p(certs["gcount"])) if use_certs else (None, None)
                LB.cert_slist, LB.cert_scount = (_p(certs["slist"]), _p(certs["scount"])) if use_certs else (None, None)
                LB.spos_of = _p(certs["of"]) if use_certs else None
                LB.cert_reset = int(k0 == cert_from)
                # (history in several slabs: a certified iteration finds the matches of the slab before its own through spos_prev_chunk)
                LB.spos_floor = base
                LB.spos_prev_chunk = ctypes.c_void_p(spos_slabs[j - 1].data_ptr() - (j - 1) * kc * N * n * 4) if (keep_spos and j > 0) else None
                LB.w = ctypes.c_void_p(w_slabs[j].data_ptr() - base * n * es)
                LB.w_prev0 = _p(w_slabs[(k0 - 1) // kc][:, (k0 - 1) % kc]) if k0 > 0 else None
                _lib.check(lib.dicp_icp_forward(code, Pref, LBref, N, n, m, int(cfg.dim), int(cfg.const_iter),
                                                float(cfg.tolerance), k0, k1, st), "dicp_icp_forward")
                done_segs.append((k0, k1))
                if not cfg.const_iter:
                    # ICP.py:259: stop at the first iteration whose steps are ALL below tolerance.  The reference synchronises
                    # every iteration for this; here the counters of a segment travel to pinned host memory asynchronously
                    # and are read one segment LATER, while the next segment is already running: no drained GPU, no launch
                    # bubble.  The price is at most one segment of frozen no-op iterations past K (every cloud has converged,
                    # so nothing moves), trimmed below exactly like the ones a sync_every > 1 leaves.
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
            if pending is not None and _converged_at(pending) is not None:          # the last segment that ran
                K = _converged_at(pending)

            # ICP.py:267-281: stats of the clouds that never converged (they report the matches of the LAST executed
            # iteration; with sync_every > 1 a few frozen no-op iterations may have run past K, which leaves n_matched of
            # such clouds unchanged or zero-weighted) and T from the last pose
            T = torch.empty((N, 4, 4), dtype=dt, device=dev)
            _lib.check(lib.dicp_loop_finish(code, _p(poses[K]), _p(alive[K]), _p(n_start), _p(n_matched), K, N,
                                            _p(iterations), _p(matched), _p(T), st), "dicp_loop_finish")
            if sweep is not None and cfg.stats_out is not None:
                cfg.stats_out["knn_pairs"] = sweep.pair_shards    # device int64 shards: sum them after a sync
                if certs is not None:             # (Kmax, 128) int32: [:, :64].sum(1) = units, [:, 64:].sum(1) = single queries searched again per iteration
                    cfg.stats_out["searched_again"] = certs["count"]
                    cfg.stats_out["budgets"] = certs["q"]         # (N,n) by query: the budgets as the last iteration left them
                    cfg.stats_out["certs_off"] = (cert_cloud[:, 2] > 0).to(torch.int32)  # (N) int32: 1 = the cloud's certificates were switched off during the call (they cost more than searching everything)
                    if cert_hint is not None and cert_hint["event"] is None and cert_hint["calls"] % 16 == 1:     # (every sixteenth certified call: it is host time)
                        cert_hint["host"][:N].copy_(cert_cloud, non_blocking=True)
                        cert_hint["event"] = torch.cuda.Event()
                        cert_hint["event"].record()
            if adaptive and form_hint is not None and form_hint["event"] is None and (form_hint["calls"] < 2 or form_hint["calls"] % 16 == 0):
                # (the first two calls of a shape and every sixteenth after: it is host time) clouds with long slabs in any plain search of this call
                if form_hint["host"] is None:
                    form_hint["host"] = torch.empty((2 + FORM_PLAN_ITERS,), dtype=torch.int32).pin_memory()
                units128 = ((cfg.src_rows if cfg.src_rows is not None else n) + 127) // 128
                tallied = sweep_form[:K]
                searched = (tallied > 0).any(dim=1)                                    # iterations with a plain search
                last = torch.where(searched, torch.arange(1, tallied.shape[0] + 1, device=dev), 0).argmax()     # the last of them
                long_k = (tallied > FORM_TILES * units128).sum(dim=1, dtype=torch.int32)                        # clouds with long slabs, per iteration
                per_k = torch.full((FORM_PLAN_ITERS,), -1, dtype=torch.int32, device=dev)
                kk = min(K, FORM_PLAN_ITERS)
                per_k[:kk] = torch.where(searched[:kk], long_k[:kk], torch.full_like(long_k[:kk], -1))
                form_hint["host"].copy_(torch.cat((torch.stack(((tallied > FORM_TILES * units128).any(dim=0).sum(dtype=torch.int32),
                                                                (tallied[last] > FORM_TILES_MOVING * units128).sum(dtype=torch.int32))), per_k)), non_blocking=True)
                form_hint["event"] = torch.cuda.Event()
                form_hint["event"].record()
            if form_hint is not None:
                form_hint["calls"] += 1
            weights = (w_slabs[0] if len(w_slabs) == 1 else torch.cat(w_slabs, dim=1))[:, :K]
            deltas_out = deltas[:, :K]
            costs_out = costs[:, :K]
            # ICP.py:274: the transformed source, in this node too (one launch; its own autograd node cost a mid-size call 35 us of host time)
            pc = torch.empty_like(src)
            _lib.check(lib.dicp_transform_points(code, _p(src), _p(poses[K]), _p(pc), N, n, st), "dicp_transform_points")

        if need_grad:
            spos_of = certs["of"] if (certs is not None and keep_spos) else None
            saved = [src, tgt, w0c, poses, deltas, areg, alive] + idx_slabs + spos_slabs + qorders + ([sweep.tperm, sweep.tgt_s] if owned else []) + ([spos_of] if spos_of is not None else [])
            if soft:
                saved += [nbr_hist, lse_hist] + (U_list if U_list is not None else [])
            ctx.save_for_backward(*saved)
            ctx.cfg, ctx.K, ctx.P, ctx.Kmax = cfg, K, P, Kmax
            ctx.soft = (float(g_eps), float(g_tau), seed_list, len(U_list) if U_list is not None else 0) if soft else None
            ctx.layout = (len(idx_slabs), len(spos_slabs), len(qorders), kc, owned, m_pad, kind,
                          [(a, min(b, K), q) for (a, b), q in zip(done_segs, seg_q) if a < K])
            ctx.of_from = cert_from if spos_of is not None else None
            # Clouds that do not converge: the forward's last query order (iteration 3's) is stale by the last iteration -- a block's slots no longer match a
            # window of neighbouring target rows, and most contributions take the float atomics (0.43 instead of 0.15 ms per launch on independently sampled
            # clouds, profiles/r05_ragged_lists.txt).  The backward then orders its slots by the reference matches themselves (one counting-sort launch).
            ctx.bwd_reorder = bool(clouds_moving) and sweep is not None
        conv = converged.bool()
        ctx.mark_non_differentiable(deltas_out, weights, costs_out, conv, iterations, matched)
        return T, pc, deltas_out, weights, costs_out, conv, iterations, matched

    @staticmethod
    def backward(ctx, gT, gpc, *_unused):
        src, tgt, w0c, poses, deltas, areg, alive, *rest = ctx.saved_tensors
        cfg, K, P, Kmax = ctx.cfg, ctx.K, ctx.P, ctx.Kmax
        n_idx, n_spos, n_q, kc, owned, m_pad, kind, segs = ctx.layout
        idx_slabs, spos_slabs = rest[:n_idx], rest[n_idx:n_idx + n_spos]
        qorders = rest[n_idx + n_spos:n_idx + n_spos + n_q]
        of_from = getattr(ctx, "of_from", None)
        spos_of = None
        if of_from is not None:         # (the certified iterations' match history is kept by reference: dicp_loop_buffers.spos_of)
            spos_of, rest = rest[-1], rest[:-1]
        tperm, tgt_s = (rest[-2], rest[-1]) if owned else (None, None)
        soft = getattr(ctx, "soft", None)
        if soft is not None:    # Gumbel-softmax correspondences: neighbour rows and log-sum-exp of every iteration (+ the injected noise)
            nbr_hist, lse_hist, *U_list = rest[n_idx + n_spos + n_q:]
        lib = _lib.load()
        dev, dt = src.device, src.dtype
        code, es = _DT[dt], src.element_size()
        N, n, _ = src.shape
        m, c = tgt.shape[1], tgt.shape[2]
        with _on(dev):
            st = _stream()
            gsrc_pc = None
            if gpc is not None:         # pc = C_K p + r_K: its cotangent reaches the source directly and the pose through T
                gsrc_pc = torch.empty_like(src)
                pcp = torch.empty((N, lib.dicp_accumulate_blocks(n), _lib.NBWD_PAD), dtype=dt, device=dev)
                _lib.check(lib.dicp_transform_points_bwd(code, _p(src), _p(poses[K]), _p(gpc.contiguous()), _p(gsrc_pc), _p(pcp), N, n, st), "dicp_transform_points_bwd")
                gT_pc = _pose_sums_to_gT(pcp, N, dt, dev, st)
                gT = gT_pc if gT is None else gT + gT_pc
            want_tgt, want_w = ctx.needs_input_grad[1], ctx.needs_input_grad[3] and w0c is not None
            cv = 6 if cfg.icp_type == "pt2pl" else 3
            if owned and soft is None and cfg.timing_events is None and n_spos == 1 and n_idx == 0 and c == cv and segs and K >= 1 and not cfg.deterministic:
                # every iteration takes the windowed form inside one slab: the whole pass is one library call (dicp_loop_backward) on one allocation.
                # (Tolerance mode feels it most: there the host cannot run ahead of the GPU, and what it does before the pass's first launch is exposed.)
                qo_b = qorders[-1]
                if cfg.stats_out is not None:
                    cfg.stats_out["bwd_reordered"] = bool(getattr(ctx, "bwd_reorder", False))     # (the backward's slots were ordered by the reference matches: the clouds keep moving)
                if getattr(ctx, "bwd_reorder", False):
                    ref = spos_slabs[0][K - 1]
                    if spos_of is not None and K - 1 >= of_from:
                        ref = torch.empty((N, n), dtype=torch.int32, device=dev)
                        _lib.check(lib.dicp_resolve_matches(_p(spos_slabs[0]), _p(spos_of), K - 1, _p(cfg.src_rows), N, n, _p(ref), st), "dicp_resolve_matches")
                    qo_b = order_by_matches(src, ref, m, m_pad, cfg.src_rows, cfg.tgt_rows)
                F = _lib.LoopBackwardIn(src=src.data_ptr(), tgt_sorted=tgt_s.data_ptr(), w0=w0c.data_ptr() if w0c is not None else None, tperm=tperm.data_ptr(),
                                        qorder=qo_b.data_ptr(), spos=spos_slabs[0].data_ptr(), poses=poses.data_ptr(), deltas=deltas.data_ptr(), areg=areg.data_ptr(),
                                        alive=alive.data_ptr(), src_rows=cfg.src_rows.data_ptr() if cfg.src_rows is not None else None,
                                        tgt_rows=cfg.tgt_rows.data_ptr() if cfg.tgt_rows is not None else None, N=N, n=n, m=m, c=tgt_s.shape[2], K=K, K_cap=Kmax, m_pad=m_pad,
                                        dim=int(cfg.dim), knn_variant=kind | ((0 if cfg.small_loop else 1) << 25),
                                        spos_of=spos_of.data_ptr() if spos_of is not None else None, spos_of_from=int(of_from) if of_from is not None else 0)
                gsrc, gtgt, gT0, gw = backward_once(lib, code, P, F, cfg, src, tgt, w0c, gT, bool(want_tgt), bool(want_w))
                if gsrc_pc is not None:
                    gsrc += gsrc_pc
                return gsrc, gtgt, gT0, gw, None
            gpose = torch.empty((N, 12), dtype=torch.float64, device=dev)
            gtmp = torch.empty_like(gpose)
            _lib.check(lib.dicp_pose_grad_in(code, _p(gT.contiguous()) if gT is not None else None, _p(gpose), N, st), "dicp_pose_grad_in")
            all_windowed = None         # set below: every iteration takes the windowed form -> dicp_window_reduce writes gtgt
            gtgt = None
            # Two forms of accumulate_bwd.  Atomic form (dicp_accumulate_bwd): original order, no set-up.  Windowed form
            # (dicp_accumulate_bwd_window, sweep path): everything in sorted space -- sorted copies of the source /
            # weights, slot-order gradient accumulators, target rows in the sweep's order, per-block slabs for the target
            # gradient -- 2.4x faster per iteration for ~0.4 ms of set-up and un-permuting per call.  ONE slot order serves
            # every windowed iteration: the last query order of the forward (the best sorted under the final poses); the
            # matches are stored per query, so the forward may have searched those iterations in other orders.  Matches that
            # fall outside a window (early iterations, whose poses are still far) take the kernel's atomic side path.
            q_star = len(qorders) - 1
            windowed = [bool(owned)] * len(segs)        # (measured: even iteration 0, whose matches lie far from the final ones, pays: 0.13 vs 0.28 ms;
                                                        #  a sweep call keeps sorted positions only, so it always takes the windowed form)
            only_windowed = all(windowed) and len(windowed) > 0
            all_windowed = only_windowed and c == cv
            if want_tgt:    # all windowed: dicp_window_reduce writes every element once, no zero fill needed
                gtgt = torch.empty_like(tgt) if all_windowed else torch.zeros_like(tgt)
            # likewise the source / weight gradients: un-permuted from the slot-order accumulators with = (dicp_permute_rows)
            gsrc = torch.empty_like(src) if only_windowed else torch.zeros_like(src)
            gw = (torch.empty_like(w0c) if only_windowed else torch.zeros_like(w0c)) if want_w else None
            nblk_a, nblk_w = lib.dicp_accumulate_blocks(n), lib.dicp_window_blocks(code, n, m_pad)
            det_row = det_val = None
            if cfg.deterministic and not only_windowed:
                raise NotImplementedError("ICP.deterministic covers the sweep search with the windowed backward (knn_variant KNN_SWEEP, bwd_window): this call took another form")
            if any(windowed):
                k_ref = max(b for (_, b, _), wf in zip(segs, windowed) if wf) - 1      # windows placed by the last iteration's matches
                spos_ref = spos_slabs[k_ref // kc][k_ref % kc]
                if spos_of is not None and k_ref >= of_from:    # (kept by reference: a plain array of them)
                    spos_ref = torch.empty((N, n), dtype=torch.int32, device=dev)
                    jr = k_ref // kc
                    _lib.check(lib.dicp_resolve_matches(ctypes.c_void_p(spos_slabs[jr].data_ptr() - jr * kc * N * n * 4), _p(spos_of), k_ref, _p(cfg.src_rows), N, n, _p(spos_ref), st),
                               "dicp_resolve_matches")
                qo = qorders[q_star]
                if cfg.stats_out is not None:
                    cfg.stats_out["bwd_reordered"] = bool(getattr(ctx, "bwd_reorder", False)) and not cfg.deterministic
                if getattr(ctx, "bwd_reorder", False) and not cfg.deterministic:
                    qo = order_by_matches(src, spos_ref, m, m_pad, cfg.src_rows, cfg.tgt_rows)
                if cfg.deterministic:
                    # The forward's query order comes from a counting sort whose order inside a bucket is the arrival order of LDS adds: fine for a search
                    # (exact for any order), but the backward takes its sums by slot.  Any permutation serves as slot order: here a STABLE sort of the
                    # queries by their reference match (the best locality the windows can have; a cloud's pad rows last, in index order).
                    key = spos_ref.to(torch.int64)
                    if cfg.src_rows is not None:
                        key = torch.where(torch.arange(n, device=dev)[None, :] < cfg.src_rows[:, None].to(torch.int64), key, torch.full_like(key, 2 ** 40))
                    qo = torch.argsort(key, dim=1, stable=True).to(torch.int32)
                    if want_tgt:
                        det_row = torch.empty((N, n), dtype=torch.int32, device=dev)
                        det_val = torch.empty((N, n, cv), dtype=dt, device=dev)
                src_s = _gather_rows_raw(src, qo)
                w_s = _gather_rows_raw(w0c.unsqueeze(-1), qo).squeeze(-1) if w0c is not None else None
                # slot-order accumulators and slabs: the first windowed launch writes them (bwd_overwrite), no zero fill
                gsrc_s = torch.empty_like(src)
                gw_s = torch.empty_like(w0c) if want_w else None
                slab = torch.empty((N, nblk_w, lib.dicp_window_rows(code), cv), dtype=dt, device=dev) if want_tgt else None
                gfar = torch.zeros((N, m_pad, cv), dtype=dt, device=dev) if want_tgt else None
            gs = torch.empty((N, 36), dtype=dt, device=dev)
            gb = torch.empty((N, 6), dtype=dt, device=dev)
            bwd_flat = torch.empty((N * max(nblk_a, nblk_w) * _lib.NBWD_PAD,), dtype=dt, device=dev)
            bwdp = {False: bwd_flat[:N * nblk_a * _lib.NBWD_PAD].view(N, nblk_a, _lib.NBWD_PAD),
                    True: bwd_flat[:N * nblk_w * _lib.NBWD_PAD].view(N, nblk_w, _lib.NBWD_PAD)}
            ev = cfg.timing_events
            events = ev.handles(Kmax) if ev is not None else None
            # truncated reverse sweep (dicp_hip.h, dicp_loop_buffers.bwd_skip): a cloud's sweep ends at the iteration from which on nothing reaches the
            # result's own rounding; the iterations before it do no per-point work.  Not with hard Huber weights: their reference gradient is NaN at an exactly zero residual whatever the cotangent.
            eps = cfg.bwd_skip_eps
            if eps is None:
                eps = 2.0 ** -22 if dt == torch.float32 else 2.0 ** -40
            if (cfg.loss_name == "huber" and not cfg.differentiable) or soft is not None:    # (soft correspondences carry gradient themselves: nothing contracts the chain)
                eps = 0.0
            skip = None
            if eps > 0.0:
                sk_arena = _Arena(dev)
                sk_arena.take((N,), torch.float64)
                sk_arena.take((N,), torch.int32)
                sk_arena.take((Kmax + 1,), torch.int32)
                sk_arena.take((N + 1,), torch.int32)
                skip = sk_arena.finish()            # mref, decisions, live counters (+ the tail's error word), the one-launch tail's per-cloud counters (+ its error word)
                if cfg.stats_out is not None:
                    cfg.stats_out["bwd_live"] = skip[2]     # (Kmax + 1) int32: clouds that did per-point work in iteration k of the backward; [Kmax]: a wait of the tail ran out
            # The ended iterations as ONE launch (dicp_loop_buffers.bwd_tail_from).  Where a sweep ends is decided on the device, while the host
            # enqueues; what the host can know is where the PREVIOUS call of this shape ended (its live counters, copied to pinned memory behind
            # that call's launches): the iterations at which fewer than an eighth of its clouds were still at work go to the one launch, which
            # sweeps a cloud that is at work after all with one block -- slower for that cloud, exact either way.
            tail_from, hint = 0, None
            # (inside a graph capture no event may be queried: the hint of the warm-up calls is read as it stands -- torch's capture entry points
            #  synchronise first -- and none is recorded; a stale hint costs time, never correctness: a cloud at work in the tail is swept there)
            capturing = torch.cuda.is_current_stream_capturing()
            use_tail = skip is not None and only_windowed and cfg.bwd_tail and cfg.hints is not None and not cfg.deterministic
            if use_tail:
                if not capturing:
                    cfg.hints.check()       # an earlier pass's tail ran out of patience: its gradients are NaN, and the caller hears about it here at the latest
                # (the newest hint that has ARRIVED: in a loop that never waits for the GPU the last call's own counters are still on their way)
                hints = cfg.hints.tail_records(dev, (N, n, m, Kmax, dt), any_stream=capturing)
                hint = next((h for h in reversed(hints) if h[2] == (N, n, K) and (capturing or h[1].query())), None)
                # the tail's blocks wait for each other: only where all of a cloud's blocks are resident at once (dicp_bwd_tail_max_blocks)
                if hint is not None and nblk_w <= lib.dicp_bwd_tail_max_blocks(code):
                    live = hint[0][:K].tolist()
                    tail_from = max(0, next((k for k in range(K) if live[k] * 8 >= N), K) - 1)      # (most sweeps have ended BEFORE the one launch starts)
            gum = None
            if soft is not None:
                g_eps, g_tau, seed_list, n_u = soft
                U_arr = (ctypes.c_void_p * Kmax)(*[u.data_ptr() for u in U_list[:Kmax]]) if n_u else None
                seeds = (ctypes.c_uint32 * Kmax)(*seed_list)
                ps_t, g_ps = torch.empty((N, n, 3), dtype=dt, device=dev), torch.empty((N, n, 3), dtype=dt, device=dev)
                g_nbr = torch.empty((N, n, c), dtype=dt, device=dev)
                gum = _lib.GumbelLoop(U=ctypes.cast(U_arr, ctypes.c_void_p) if U_arr is not None else None, seeds=ctypes.cast(seeds, ctypes.c_void_p),
                                      eps=g_eps, tau=g_tau, ps_t=_p(ps_t), nbr=_p(nbr_hist), lse=_p(lse_hist), g_nbr=_p(g_nbr), g_ps=_p(g_ps))
            have, form, fresh = 0, None, 1
            # neighbouring segments of one form inside one history slab run as ONE library call (the forward cut them where the host had to
            # act -- a new query order, a convergence check -- and none of that concerns the reverse sweep)
            runs = []
            for (k0, k1, q), w_form in zip(reversed(segs), reversed(windowed)):
                if runs and runs[-1][3] == w_form and runs[-1][0] == k1 and (k0 // kc) == ((runs[-1][1] - 1) // kc) and k1 > k0:
                    runs[-1] = (k0, runs[-1][1], q, w_form)
                else:
                    runs.append((k0, k1, q, w_form))
            # (the one launch runs down to iteration 0: it belongs to the last run, and starts no higher than that run does)
            tail_from = min(tail_from, runs[-1][1]) if (runs and runs[-1][0] == 0 and runs[-1][3]) else 0
            tail_part = torch.empty((N, nblk_w, _lib.NBWD_PAD), dtype=dt, device=dev) if tail_from > 0 else None
            if cfg.stats_out is not None:
                cfg.stats_out["bwd_tail_from"] = int(tail_from)
                if tail_from > 0:
                    cfg.stats_out["bwd_tail_error"] = skip[3][N:]       # (1) int32: nonzero = a wait of the tail launch ran out (TailTimeout at the next pass)
            folded = False
            for (k0, k1, q, w_form) in runs:
                if have and w_form != form:     # the partials of the other form have another block count: fold them in here
                    gpose += bwdp[form].sum(dim=1)[:, :12].to(torch.float64)
                    have = 0
                form = w_form
                j = k0 // kc
                base = j * kc
                LB = _lib.LoopBuffers(
                    src=_p(src_s) if w_form else _p(src), tgt=_p(tgt_s) if w_form else _p(tgt),
                    w_init=_p(w_s) if w_form else _p(w0c), c=tgt_s.shape[2] if w_form else c, K=Kmax, knn_variant=kind | ((0 if cfg.small_loop else 1) << 25), m_pad=m_pad, idx_per_iter=1,
                    qorder=_p(qo) if w_form else None,
                    spos=ctypes.c_void_p(spos_slabs[j].data_ptr() - base * N * n * 4) if w_form else None,
                    spos_ref=_p(spos_ref) if w_form else None, gts_far=_p(gfar) if w_form else None,
                    spos_of=_p(spos_of) if (w_form and spos_of is not None) else None, spos_of_from=int(of_from) if of_from is not None else 0,
                    poses=_p(poses), deltas=_p(deltas), areg=_p(areg), alive=_p(alive),
                    idx=ctypes.c_void_p(idx_slabs[j].data_ptr() - base * N * n * 4) if idx_slabs else None, events=events,
                    bwd_overwrite=fresh if (w_form and k1 > k0) else 0, src_rows=_p(cfg.src_rows), tgt_rows=_p(cfg.tgt_rows),
                    bwd_skip=_p(skip[1]) if skip else None, bwd_mref=_p(skip[0]) if skip else None, bwd_live=_p(skip[2]) if skip else None,
                    bwd_skip_eps=float(eps), bwd_tail_from=int(tail_from) if (w_form and k0 == 0) else 0,
                    bwd_tail_partials=_p(tail_part), bwd_tail_arrive=_p(skip[3]) if skip else None,
                    gumbel=ctypes.cast(ctypes.pointer(gum), ctypes.c_void_p) if gum is not None else None,
                    det_far_row=_p(det_row) if w_form else None, det_far_val=_p(det_val) if w_form else None)
                fresh_was = bool(w_form and k1 > k0 and fresh)
                if w_form and k1 > k0:
                    fresh = 0
                _lib.check(lib.dicp_icp_backward(code, ctypes.byref(P), ctypes.byref(LB), N, n, m, int(cfg.dim), _p(gpose), _p(gtmp), have,
                                                 _p(gs), _p(gb), _p(gsrc_s) if w_form else _p(gsrc), _p(slab) if w_form else _p(gtgt),
                                                 _p(gw_s) if w_form else _p(gw), _p(bwdp[form]), k0, k1, st), "dicp_icp_backward")
                have = 1
                if w_form and k0 == 0 and tail_from > 0:
                    kt = min(tail_from, k1) - (1 if (fresh_was and tail_from >= k1) else 0)     # (dicp_icp_backward: the first windowed iteration is never the tail's)
                    folded = kt > 0         # the tail launch left the cotangent of pose_0 with the last pose sums already in it
                if (k1 - k0) % 2:           # the library alternates the two buffers: odd chunk -> the result is in the other one
                    gpose, gtmp = gtmp, gpose
            if use_tail and not capturing:
                entry = None
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
            if any(windowed):               # slot s is source point qo[s]; slabs + out-of-window rows -> original target order
                permute = lib.dicp_permute_rows if only_windowed else lib.dicp_permute_add_rows
                _lib.check(permute(code, _p(gsrc_s), _p(qo), N, n, n, n, 3, 3, _p(gsrc), n, 3, st), "dicp_permute_rows")
                if want_w:
                    _lib.check(permute(code, _p(gw_s), _p(qo), N, n, n, n, 1, 1, _p(gw), n, 1, st), "dicp_permute_rows")
                if want_tgt:
                    _lib.check(lib.dicp_window_reduce(code, _p(slab), _p(spos_ref), _p(qo), _p(tperm), _p(gfar), _p(cfg.src_rows), N, n, m, m_pad, cv,
                                                      _p(gtgt), c, int(all_windowed), st), "dicp_window_reduce")
            gT0 = torch.empty((N, 4, 4), dtype=dt, device=dev)      # final gpose + the last launch's pose partials
            if folded:
                have = 0
            _lib.check(lib.dicp_pose_grad_out(code, _p(gpose), _p(bwdp[form]) if have else None, bwdp[form].shape[1] if have else 0,
                                              _p(gT0), N, st), "dicp_pose_grad_out")
            if gsrc_pc is not None:
                gsrc += gsrc_pc
            if tail_from > 0:
                _strict_tail_check(cfg, skip[3][N:])
        return gsrc, gtgt, gT0, gw, None


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


