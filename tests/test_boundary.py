"""The drop-in boundary (no GPU needed): the C-ABI library loads, exports every symbol that
include/dicp_hip.h declares, rejects bad arguments without launching, and the product
package never touches the oracle."""
import ctypes
import os
import re

import pytest

from dicp_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    _lib.build()
    return _lib.load()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "dicp_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dicp_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree(lib):
    names = declared_symbols()
    assert len(names) >= 13
    assert sorted(_lib.EXPORTS) == names
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), "libdicp_hip.so does not export " + n


def test_struct_layouts_match_header(lib):
    assert ctypes.sizeof(_lib.WeightParams) == 4 * 4 + 4 * 8
    # dicp_step_io: field order and padding as the C compiler lays it out
    expect = ["partials", "nblk", "iter", "dim", "const_iter", "tolerance", "rows_per_point", "n", "pose_in", "pose_out",
              "delta", "delta_stride", "cost", "cost_prev", "cost_stride", "areg", "alive", "alive_out", "converged", "iterations",
              "matched_ratio", "n_start", "n_matched", "w_cur", "w_prev", "w_stride", "n_not_converged", "frame", "pose_search_out",
              "rmax", "dcum", "dcum_stride", "cert_cloud", "cert_qu", "cert_units", "glist_cap", "glist", "gcount", "cert_scount", "cert_slist", "w_copied"]
    assert [f[0] for f in _lib.StepIO._fields_] == expect
    hdr = open(os.path.join(ROOT, "include", "dicp_hip.h")).read()
    body = hdr[hdr.index("typedef struct dicp_step_io {"):hdr.index("} dicp_step_io;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"([A-Za-z_0-9]+)\s*;", body)
    assert fields == expect


def _header_fields(hdr, name):
    """The declared field names of `typedef struct <name> { ... } <name>;`, in order ("int64_t w_iter, w_stride" declares two)."""
    body = hdr[hdr.index("typedef struct %s {" % name):hdr.index("} %s;" % name)]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if decl:
            fields.extend(x.replace("*", " ").split()[-1] for x in decl.split(","))
    return fields


def test_loop_buffers_layout_matches_header(lib):
    """dicp_loop_buffers and its four sub-structs: the ctypes mirrors list the header's fields in the header's order (a mismatch would hand the
    library a pose history where it expects an index history), the struct carries the ABI version first, and the entry points refuse another one."""
    hdr = open(os.path.join(ROOT, "include", "dicp_hip.h")).read()
    for name, cls in (("dicp_search_buffers", _lib.SearchBuffers), ("dicp_cert_buffers", _lib.CertBuffers), ("dicp_history_buffers", _lib.HistoryBuffers),
                      ("dicp_bwd_buffers", _lib.BwdBuffers), ("dicp_loop_buffers", _lib.LoopBuffers)):
        assert [f[0] for f in cls._fields_] == _header_fields(hdr, name), name
    assert _lib.LoopBuffers._fields_[0][0] == "abi" and _lib.LoopBuffers().abi == _lib.ABI_VERSION
    assert [f[0] for f in _lib.LoopBuffers._fields_[-4:]] == ["search", "cert", "hist", "bwd"]
    # keyword construction reaches the sub-structs
    LB = _lib.LoopBuffers(c=6, search_m_pad=64, cert_reset=1, hist_w_iter=7, bwd_tail_from=3)
    assert (LB.c, LB.search.m_pad, LB.cert.reset, LB.hist.w_iter, LB.bwd.tail_from) == (6, 64, 1, 7, 3)
    # another layout's struct is refused before anything is read from it (DICP_ERR_ABI = 6)
    one, P = ctypes.c_void_p(64), _lib.WeightParams(mode=1, loss=0)
    LB = _lib.LoopBuffers(src=one, tgt=one, converged=one, iterations=one, matched_ratio=one, n_start=one, n_matched=one, partials=one, counters=one, K=4,
                          hist_poses=one, hist_deltas=one, hist_costs=one, hist_alive=one, hist_areg=one, hist_w=one, hist_idx=one, hist_w_iter=1, hist_w_stride=4)
    LB.abi = _lib.ABI_VERSION - 1
    assert lib.dicp_icp_forward(0, ctypes.byref(P), ctypes.byref(LB), 1, 1, 1, 3, 1, 1e-4, 0, 1, None) == 6
    assert lib.dicp_icp_forward_plan(0, ctypes.byref(P), ctypes.byref(LB), ctypes.byref(_lib.SegmentPlan(nseg=1)), 1, 1, 1, 3, 1, 1e-4, None) == 6
    assert lib.dicp_icp_backward(0, ctypes.byref(P), ctypes.byref(LB), 1, 1, 1, 3, one, one, 0, one, one, one, one, None, one, 0, 1, None) == 6


def test_new_entry_points_reject_bad_arguments(lib):
    """Set-up / windowed-backward entry points: null pointers, bad shapes and dtypes are rejected before any launch."""
    one, P = ctypes.c_void_p(64), _lib.WeightParams(mode=1, loss=0)
    assert [lib.dicp_window_blocks(0, n, 16384) for n in (0, 1, 16384)] == [0, 1, 16]
    assert lib.dicp_window_rows(0) == 1536 and lib.dicp_window_rows(1) == 768
    # dicp_sweep_build(dtype, tgt, c, center, tgt_rows, tperm, N, m, m_pad, tgs4, tgt_s, tgt_s_stride, stream)
    assert lib.dicp_sweep_build(0, None, 3, None, None, one, 1, 1, 64, one, None, 0, None) == 1
    assert lib.dicp_sweep_build(0, one, 4, None, None, one, 1, 1, 64, one, None, 0, None) == 2
    assert lib.dicp_sweep_build(0, one, 3, None, None, one, 1, 1, 64, ctypes.c_void_p(8), None, 0, None) == 5
    assert lib.dicp_sweep_build(0, one, 3, None, None, None, 1, 1, 64, one, None, 0, None) == 1           # the permutation comes from dicp_sweep_sort
    assert lib.dicp_sweep_build(0, one, 6, None, None, one, 1, 1, 64, one, one, 4, None) == 2              # sorted rows narrower than the rows
    # dicp_sweep_sort(dtype, tgt, c, center, tgt_rows, N, m, m_pad, keys, tperm, nbkt, bucket, brange, scratch, bytes, stream)
    assert lib.dicp_sweep_sort(0, None, 3, None, None, 1, 1, 64, one, one, 0, None, None, None, 0, None) == 1
    assert lib.dicp_sweep_sort(9, one, 3, None, None, 1, 1, 64, one, one, 0, None, None, None, 0, None) == 3
    assert lib.dicp_sweep_sort(0, one, 3, None, None, 1, 5, 128, one, one, 0, None, None, None, 0, None) == 2             # not the padded size
    assert lib.dicp_sweep_sort(0, one, 3, None, None, 1, 5, 64, one, one, 1024, one, None, None, 0, None) == 1            # a table needs its range too
    assert lib.dicp_sweep_sort(0, one, 3, None, None, 1, 16385, 16448, one, one, 0, None, None, None, 0, None) == 1       # beyond the LDS sort: scratch
    assert lib.dicp_sweep_sort(1, one, 3, None, None, 1, 1, 64, one, one, 0, None, None, one, 8, None) == 1                # float64 keys: scratch too small
    assert lib.dicp_sweep_sort_scratch_bytes(0, 4, 16384) == 0 and lib.dicp_sweep_sort_scratch_bytes(0, 4, 16448) >= 4 * 2 * 16448 * 8
    assert lib.dicp_sweep_sort_scratch_bytes(1, 4, 64) >= 4 * 2 * 64 * 12
    assert lib.dicp_query_order(0, one, None, None, 1024, 1, 1, one, None, None, None, 0, None, 0, None, None, 0, None, None, None) == 1
    assert lib.dicp_query_order(9, one, None, one, 1024, 1, 1, one, None, None, None, 0, None, 0, None, None, 0, None, None, None) == 3
    assert lib.dicp_query_order(0, one, None, one, 1024, 1, 1, one, None, None, None, 0, None, 0, one, one, 0, None, None, None) == 2      # keys without m
    assert lib.dicp_match_order(0, None, None, 1, 1, one, 1 << 20, one, None) == 1 and lib.dicp_match_order(0, one, None, 1, 1, one, 0, one, None) in (2, 5)      # (no matches; scratch too small or misaligned)
    assert lib.dicp_match_order_scratch_bytes(0, 0, 16) == 0 and lib.dicp_match_order_scratch_bytes(0, 2, 100) >= 2 * 100 * 12 + 2 * 128 * 8
    # dicp_query_reorder(dtype, src, pose, pose_prev, order_prev, brange, nbkt, N, n, qorder, m_pad, skeys, bucket, m, src_rows, tgt_rows, stream): both poses and the order before
    assert lib.dicp_query_reorder(0, one, None, one, one, one, 1024, 1, 1, one, 0, None, None, 0, None, None, None) == 1
    assert lib.dicp_query_reorder(0, one, one, one, None, one, 1024, 1, 1, one, 0, None, None, 0, None, None, None) == 1
    assert lib.dicp_query_reorder(0, one, one, one, one, one, 1024, 1, 1, one, 0, None, None, 0, None, None, None) == 1       # (in place: order_prev == qorder)
    assert lib.dicp_loop_init(0, one, one, 0.01, 2, 1, 1, one, one, one, None, None, None, None, None, 0, None) == 2
    assert lib.dicp_loop_init(0, None, one, 0.01, 1, 1, 1, one, one, one, one, one, None, None, None, 0, None) == 1
    assert lib.dicp_loop_init(0, one, one, 0.01, 1, 1, 1, one, one, one, None, None, None, one, None, 4, None) == 1     # rmax needs the points and dcum
    assert lib.dicp_loop_finish(0, one, one, one, one, 1, 1, None, one, one, None) == 1
    # centred search (center itself is optional everywhere)
    assert lib.dicp_search_frame(0, None, 3, None, 1, 1, 16.0, 1, None, None, 0, None, one, None) == 1
    assert lib.dicp_search_frame(0, one, 3, None, 1, 1, 16.0, 1, None, None, 0, None, None, None) == 1
    assert lib.dicp_search_frame(0, one, 2, None, 1, 1, 16.0, 1, None, None, 0, None, one, None) == 2
    assert lib.dicp_search_frame(0, one, 3, None, 1, 1, -1.0, 1, None, None, 0, None, one, None) == 2
    assert lib.dicp_search_frame(0, one, 3, None, 1, 1, 16.0, 1, one, None, 0, one, one, None) == 2      # queries without their count
    assert lib.dicp_search_frame(5, one, 3, None, 1, 1, 16.0, 1, None, None, 0, None, one, None) == 3
    assert lib.dicp_search_pose(0, None, None, 1, one, None) == 1 and lib.dicp_search_pose(0, one, None, 0, one, None) == 2
    # dicp_accumulate_bwd_window(dtype, prm, src_s, tgt_s, c, spos, spos_ref, qorder, pose, w_s, alive, gs, gb, src_rows, N, n, m_pad, gsrc_s, slab, far, gw_s, partials, ow, stream)
    assert lib.dicp_accumulate_bwd_window(0, ctypes.byref(P), one, one, 6, None, one, None, one, one, None, one, one, None, 1, 1, 64,
                                          one, None, None, None, one, 0, None) == 1
    assert lib.dicp_accumulate_bwd_window(0, ctypes.byref(P), one, one, 6, one, one, None, one, one, None, one, one, None, 1, 1, 63,
                                          one, None, None, None, one, 0, None) == 2
    assert lib.dicp_accumulate_bwd_window(0, ctypes.byref(P), one, one, 6, one, one, None, one, one, None, one, one, None, 1, 1, 64,
                                          one, one, None, None, one, 0, None) == 1      # a slab needs the side buffer too
    assert lib.dicp_window_reduce(0, one, one, None, one, None, None, 1, 1, 1, 64, 5, one, 6, 0, None) == 2
    assert lib.dicp_window_reduce(0, one, one, None, one, None, None, 1, 1, 1, 64, 6, one, 3, 0, None) == 2
    assert lib.dicp_permute_add_rows(0, one, one, 1, 2, 1, 2, 3, 3, one, 4, 3, None) == 2
    assert lib.dicp_permute_rows(0, one, one, 1, 2, 1, 2, 3, 3, one, 4, 3, None) == 2
    assert lib.dicp_copy(None, one, 16, None) == 1 and lib.dicp_copy(one, one, 6, None) == 5 and lib.dicp_zero(None, 16, None) == 1 and lib.dicp_zero(ctypes.c_void_p(66), 16, None) == 5
    assert lib.dicp_copy(one, one, 0, None) == 0 and lib.dicp_zero(one, 0, None) == 0                                    # (nothing to do: nothing launched)
    # dicp_loop_backward_prepare(dtype, fwd, src_s_out, w_s_out, spos_ref_out, stream): the forward's buffers it reads, its own outputs
    F = _lib.LoopBackwardIn(src=one, qorder=one, spos=one, N=1, n=1, m=1, c=6, K=2, K_cap=2, m_pad=64, dim=3)
    assert lib.dicp_loop_backward_prepare(0, ctypes.byref(F), None, None, None, None) == 1
    assert lib.dicp_loop_backward_prepare(9, ctypes.byref(F), one, None, None, None) == 3
    F.w0 = one
    assert lib.dicp_loop_backward_prepare(0, ctypes.byref(F), one, None, None, None) == 1           # weights need their copy too
    F.w0, F.spos_of, F.spos_of_from = None, one, 0
    assert lib.dicp_loop_backward_prepare(0, ctypes.byref(F), one, None, None, None) == 1           # a history by reference needs the resolved matches
    F.K = 3
    assert lib.dicp_loop_backward_prepare(0, ctypes.byref(F), one, None, one, None) == 2            # K beyond the histories
    assert lib.dicp_pose_grad_in(0, None, None, 1, None) == 1 and lib.dicp_pose_grad_in(7, None, one, 1, None) == 3
    assert lib.dicp_pose_grad_out(0, one, one, 0, one, 1, None) == 2 and lib.dicp_pose_grad_out(0, one, None, 0, None, 1, None) == 1
    # dicp_knn_sweep(dtype, src, pose, tgs4, tperm, qorder, bucket, brange, nbkt, src_rows, tgt_rows, N, n, m, m_pad, idx, spos, pairs, cfg, f16_image, form_in, form_out, form_default, stream)
    assert lib.dicp_knn_sweep(0, one, None, one, one, None, one, one, 1024, None, None, 1, 1, 1, 64, one, None, None, 99, None, None, None, 0, None) == 4
    assert lib.dicp_knn_sweep(0, one, None, one, one, None, one, one, 1024, None, None, 1, 1, 1, 64, None, None, None, 0, None, None, None, 0, None) == 1   # idx or spos
    assert lib.dicp_knn_sweep(0, one, None, one, one, None, one, one, 1024, None, None, 1, 1, 1, 64, one, None, None, 16, None, None, None, 0, None) == 4   # (the scan form is gone)
    assert lib.dicp_knn_sweep(0, one, None, one, one, None, one, one, 1024, None, None, 1, 1, 1, 64, one, None, None, 2 | 0x100, None, None, None, 0, None) == 4  # (the slot-ordered source copy is gone too)
    # dicp_call_*: plans are pure host arithmetic; a bad shape, a bad dtype, a missing buffer are refused before anything is touched
    call = _lib.Call(N=2, n=300, m=200, c=3, K=5, dim=3, need_grad=1, n_resort=2)
    call.resort[0], call.resort[1] = 1, 2
    lay = _lib.CallLayout()
    assert lib.dicp_call_plan(0, ctypes.byref(call), ctypes.byref(lay)) == 0
    assert lay.m_pad == 256 and lay.n_orders == 3 and lay.zeroed > 0 and lay.total > lay.spos >= lay.orders + 3 * 2 * 300 * 4
    offs = sorted(getattr(lay, k) for k in ("n_matched", "counters", "pairs"))
    assert offs[0] == 0 and offs[-1] < lay.zeroed <= lay.T and all(o % 256 == 0 for o in offs)       # the zeroed state leads, everything 256-byte aligned
    # the non-differentiable results: offsets into an allocation of their own, the zero-initialised ones first
    roffs = sorted(getattr(lay, k) for k in ("deltas", "costs", "converged", "iterations", "matched_ratio"))
    assert roffs[0] == 0 and roffs[-1] < lay.results_zeroed <= lay.weights < lay.results_total and lay.results_total - lay.weights >= 2 * 5 * 300 * 4
    lay64 = _lib.CallLayout()
    assert lib.dicp_call_plan(1, ctypes.byref(call), ctypes.byref(lay64)) == 0 and lay64.total > lay.total
    assert lib.dicp_call_plan(3, ctypes.byref(call), ctypes.byref(lay)) == 3 and lib.dicp_call_plan(0, None, ctypes.byref(lay)) == 1
    call.resort[1] = 1
    assert lib.dicp_call_plan(0, ctypes.byref(call), ctypes.byref(lay)) == 2             # not ascending
    call.resort[1] = 5
    assert lib.dicp_call_plan(0, ctypes.byref(call), ctypes.byref(lay)) == 2             # not inside (0, K)
    call.resort[1] = 2
    assert lib.dicp_call_forward(0, ctypes.byref(P), ctypes.byref(call), None) == 1      # no buffers
    blay = _lib.CallBackwardLayout()
    assert lib.dicp_call_backward_plan(0, ctypes.byref(P), ctypes.byref(call), 1, 0, ctypes.byref(blay)) == 0
    assert blay.far > 0 and blay.slab > 0 and blay.gw_s == 0 and blay.w_s == 0 and blay.live < blay.zeroed <= blay.gpose and blay.nblk_w == lib.dicp_window_blocks(0, 300, 256)
    G = _lib.CallGrads()
    assert lib.dicp_call_backward(0, ctypes.byref(P), ctypes.byref(call), ctypes.byref(G), None) == 1
    # dicp_loop_backward: the reverse sweep of a buffer-by-buffer forward from one call; the same layout as dicp_call_backward's for the same shape
    F = _lib.LoopBackwardIn(N=2, n=300, m=200, c=3, K=4, K_cap=5, m_pad=256, dim=3, knn_variant=3)
    bl2 = _lib.CallBackwardLayout()
    assert lib.dicp_loop_backward_plan(0, ctypes.byref(P), ctypes.byref(F), 1, 0, ctypes.byref(bl2)) == 0 and bl2.total == blay.total and bl2.live == blay.live
    assert lib.dicp_loop_backward(0, ctypes.byref(P), ctypes.byref(F), ctypes.byref(G), None) == 1        # no buffers named
    F.K = 6
    assert lib.dicp_loop_backward_plan(0, ctypes.byref(P), ctypes.byref(F), 1, 0, ctypes.byref(bl2)) == 2     # more iterations than the histories hold
    # dicp_kabsch_call_*: the same contract
    kc = _lib.KabschCall(N=2, n=300, m=200, c=3, K=5)
    kl = _lib.KabschCallLayout()
    assert lib.dicp_kabsch_call_plan(0, ctypes.byref(kc), ctypes.byref(kl)) == 0
    assert kl.m_pad == 256 and 0 == kl.costs < kl.iterations < kl.zeroed <= kl.frame and kl.total > kl.gacc > kl.gpose > kl.orders
    kc.c = 4
    assert lib.dicp_kabsch_call_plan(0, ctypes.byref(kc), ctypes.byref(kl)) == 2
    kc.c = 3
    assert lib.dicp_kabsch_call_plan(2, ctypes.byref(kc), ctypes.byref(kl)) == 3
    assert lib.dicp_kabsch_call_forward(0, ctypes.byref(kc), None) == 1
    assert lib.dicp_kabsch_call_backward(0, ctypes.byref(kc), ctypes.byref(_lib.KabschCallGrads()), None) == 1


def test_sizes_and_argument_checks(lib):
    assert lib.dicp_abi_version() == _lib.ABI_VERSION == 11
    assert [lib.dicp_padded_targets(m) for m in (0, 1, 64, 65, 129)] == [0, 64, 64, 128, 192]
    assert [lib.dicp_accumulate_blocks(n) for n in (0, 1, 512, 513, 16384)] == [0, 1, 1, 2, 32]
    # rejected before any launch (no GPU touched): null pointers, bad dtype / shapes / enums
    one = ctypes.c_void_p(16)
    # dicp_pack_target(dtype, tgt, c, center, tgt_rows, N, m, tgt4, m_pad, stream)
    assert lib.dicp_pack_target(0, None, 3, None, None, 1, 1, None, 64, None) == 1
    assert lib.dicp_pack_target(7, one, 3, None, None, 1, 1, one, 64, None) == 3
    assert lib.dicp_pack_target(0, one, 4, None, None, 1, 1, one, 64, None) == 2
    assert lib.dicp_pack_target(0, one, 3, None, None, 1, 1, one, 63, None) == 2
    # dicp_knn(dtype, src, pose, tgt4, src_rows, tgt_rows, N, n, m, m_pad, idx, variant, f16_image, stream)
    assert lib.dicp_knn(0, one, None, one, None, None, 1, 1, 1, 64, one, 9, None, None) == 4
    assert lib.dicp_knn(1, one, None, ctypes.c_void_p(32), None, None, 1, 1, 1, 64, one, _lib.KNN_MFMA, one, None) == 3
    assert lib.dicp_knn(0, one, None, ctypes.c_void_p(8), None, None, 1, 1, 1, 64, one, 0, None, None) == 5
    assert lib.dicp_knn(0, one, None, one, None, None, 1, 1, 1, 64, one, _lib.KNN_MFMA, None, None) == 1       # the matrix-core form needs its image
    # dicp_knn_f16_bytes(N, m_pad) / dicp_knn_f16_pack(tgt4, tgt_rows, N, m, m_pad, image, stream): 32 bytes per row, rows rounded up to 512, + 320 bytes per cloud + 8 per 64 rows
    assert lib.dicp_knn_f16_bytes(2, 64) == 2 * (512 * 32 + 320 + 64) and lib.dicp_knn_f16_bytes(1, 16384) == 16384 * 32 + 320 + 2048 and lib.dicp_knn_f16_bytes(0, 64) == 0
    assert lib.dicp_knn_f16_pack(None, None, 1, 1, 64, one, None) == 1
    assert lib.dicp_knn_f16_pack(one, None, 1, 65, 64, one, None) == 2
    assert lib.dicp_knn_f16_pack(ctypes.c_void_p(8), None, 1, 1, 64, one, None) == 5
    P = _lib.WeightParams(mode=1, loss=0)
    # dicp_accumulate(dtype, prm, src, tgt, c, idx, pose, w_init, alive, src_rows, N, n, m, partials, w_out, w_stride, stream)
    assert lib.dicp_accumulate(0, ctypes.byref(P), one, one, 3, one, one, one, None, None, 1, 1, 1, one, None, 0, None) == 2   # pt2pl needs normals (ICP.py:103)
    P.loss = 9
    assert lib.dicp_accumulate(0, ctypes.byref(P), one, one, 6, one, one, one, None, None, 1, 1, 1, one, None, 0, None) == 4
    P.loss = _lib.LOSS_TRIM                                                        # loss_fn "trim" is a loss like the others (loss.py:15-16)
    assert lib.dicp_accumulate(0, ctypes.byref(P), one, one, 6, one, one, one, None, None, 1, 1, 1, None, None, 0, None) == 1   # (only the null partials are left to object to; null weights = unit weights)
    assert lib.dicp_loss_weight(0, 0, 0, 1.0, 5.0, one, 4, 3, one, None) == 4      # unknown loss (loss.py:19)
    with pytest.raises(RuntimeError, match="rejected"):
        _lib.check(2, "x")
    with pytest.raises(RuntimeError, match="hipError_t 98"):
        _lib.check(-98, "x")


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "dicp_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "hostcheck" not in src or f == "dicp_math.h", f
    for f in os.listdir(os.path.join(ROOT, "dICP")):
        if not f.endswith(".py"):
            continue
        assert "oracle" not in open(os.path.join(ROOT, "dICP", f)).read()


def test_public_header_is_plain_c(tmp_path):
    """include/dicp_hip.h is the drop-in boundary: a C99 (and C++) translation unit that names the versioned struct and its sub-structs compiles against it alone --
    no HIP, no torch types in the signatures."""
    import shutil
    import subprocess
    src = tmp_path / "t.c"
    src.write_text('#include "dicp_hip.h"\n'
                   "int main(void) { dicp_loop_buffers b; b.abi = DICP_ABI_VERSION; b.search.m_pad = 1; b.cert.q = 0; b.hist.w_iter = 2; b.bwd.tail_from = 3;\n"
                   "                 dicp_step_io io; io.cert_slist = 0; (void)io; return (int)sizeof(b) & 0; }\n")
    inc = os.path.join(ROOT, "include")
    for cc, args in (("gcc", ["-std=c99", "-pedantic"]), ("g++", ["-std=c++17", "-x", "c++"])):
        if shutil.which(cc) is None:
            pytest.skip(cc + " not installed")
        r = subprocess.run([cc, "-Wall", "-Wextra", "-Werror", "-I", inc, "-fsyntax-only"] + args + [str(src)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
