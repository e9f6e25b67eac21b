"""ctypes binding of libdicp_hip.so (include/dicp_hip.h) and its in-tree build recipe.

There is no CPU compute path in this package: if the library is missing or no HIP
device is visible, every operator raises.
"""
import ctypes
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
LIB_PATH = os.environ.get("DICP_HIP_LIB") or os.path.join(_HERE, "libdicp_hip.so")   # env override: A/B builds
SOURCES = [os.path.join(_HERE, "csrc", f) for f in ("dicp_kernels.hip", "knn_f16.hip", "dicp_call.hip")]
HEADERS = ([os.path.join(_HERE, "csrc", f) for f in ("dicp_math.h", "dicp_common.h", "dicp_internal.h", "dicp_fill.h")]
           + [os.path.join(_HERE, "csrc", "kernels_%s.h" % f) for f in ("setup", "search", "setup_sort", "rows", "accumulate", "backward", "soft_svd", "host")]
           + [os.path.join(_ROOT, "include", "dicp_hip.h")])

F32, F64 = 0, 1
PT2PT, PT2PL = 0, 1
LOSS_NONE, LOSS_HUBER, LOSS_CAUCHY, LOSS_TRIM = 0, 1, 2, 3
KNN_AUTO, KNN_VALU, KNN_MFMA, KNN_SWEEP, KNN_GUMBEL = 0, 1, 2, 3, 4
NACC_PAD, NBWD_PAD, KAB_SAVE = 32, 16, 40
PAIR_SHARDS = 64      # DICP_PAIR_SHARDS
ABI_VERSION = 11
_ERRORS = {1: "null pointer", 2: "bad shape/stride", 3: "unsupported dtype", 4: "bad enum value", 5: "misaligned buffer"}

vp, i32, i64, f64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_double


class WeightParams(ctypes.Structure):
    """dicp_weight_params (include/dicp_hip.h)."""
    _fields_ = [("mode", i32), ("trim_on", i32), ("differentiable", i32), ("loss", i32),
                ("trim_dist", f64), ("tanh_k", f64), ("loss_delta", f64), ("match_thresh", f64)]


class StepIO(ctypes.Structure):
    """dicp_step_io (include/dicp_hip.h)."""
    _fields_ = [("partials", vp), ("nblk", i32), ("iter", i32), ("dim", i32), ("const_iter", i32),
                ("tolerance", f64), ("rows_per_point", i32), ("n", i32),
                ("pose_in", vp), ("pose_out", vp), ("delta", vp), ("delta_stride", i64),
                ("cost", vp), ("cost_prev", vp), ("cost_stride", i64), ("areg", vp), ("alive", vp), ("alive_out", vp),
                ("converged", vp), ("iterations", vp), ("matched_ratio", vp), ("n_start", vp),
                ("n_matched", vp), ("w_cur", vp), ("w_prev", vp), ("w_stride", i64), ("n_not_converged", vp),
                ("frame", vp), ("pose_search_out", vp), ("rmax", vp), ("dcum", vp), ("dcum_stride", i64), ("cert_cloud", vp),
                ("cert_qu", vp), ("cert_units", i32), ("glist_cap", i32), ("glist", vp), ("gcount", vp), ("cert_scount", vp), ("cert_slist", vp), ("w_copied", i32)]


class SearchBuffers(ctypes.Structure):
    """dicp_search_buffers (include/dicp_hip.h)."""
    _fields_ = [("knn_variant", i32), ("m_pad", i32), ("tgt4", vp), ("tperm", vp), ("qorder", vp), ("bucket", vp), ("brange", vp), ("nbkt", i32), ("pairs", vp), ("frame", vp), ("poses", vp),
                ("tgt_sorted", vp), ("tgt_sorted_stride", i32), ("gumbel", vp), ("first_done", i32), ("tgt_f16", vp), ("form", vp), ("form_default", i32), ("form_plan", vp)]


class CertBuffers(ctypes.Structure):
    """dicp_cert_buffers (include/dicp_hip.h)."""
    _fields_ = [("q", vp), ("qu", vp), ("set", vp), ("count", vp), ("rmax", vp), ("dcum", vp), ("reset", i32), ("cloud", vp), ("nbr", vp), ("gdirty", vp), ("pend", vp), ("glist", vp),
                ("gcount", vp), ("slist", vp), ("scount", vp), ("cm", vp)]


class HistoryBuffers(ctypes.Structure):
    """dicp_history_buffers (include/dicp_hip.h)."""
    _fields_ = [("per_iter", i32), ("spos", vp), ("poses", vp), ("deltas", vp), ("costs", vp), ("areg", vp), ("alive", vp), ("idx", vp), ("w", vp), ("w_iter", i64), ("w_stride", i64),
                ("w_prev0", vp), ("spos_prev_chunk", vp), ("spos_floor", i32), ("spos_of", vp), ("spos_of_from", i32)]


class BwdBuffers(ctypes.Structure):
    """dicp_bwd_buffers (include/dicp_hip.h)."""
    _fields_ = [("spos_ref", vp), ("gts_far", vp), ("overwrite", i32), ("skip", vp), ("mref", vp), ("live", vp), ("skip_eps", f64), ("tail_from", i32), ("tail_partials", vp),
                ("tail_arrive", vp), ("det_far_row", vp), ("det_far_val", vp)]


class LoopBuffers(ctypes.Structure):
    """dicp_loop_buffers (include/dicp_hip.h): one versioned struct, the buffers of the search / the certificates / the histories / the backward in sub-structs."""
    _fields_ = [("abi", i32), ("src", vp), ("tgt", vp), ("w_init", vp), ("c", i32), ("K", i32), ("converged", vp), ("iterations", vp), ("matched_ratio", vp), ("n_start", vp), ("n_matched", vp),
                ("partials", vp), ("counters", vp), ("counters_host", vp), ("counters_tag", i32), ("events", vp), ("src_rows", vp), ("tgt_rows", vp),
                ("search", SearchBuffers), ("cert", CertBuffers), ("hist", HistoryBuffers), ("bwd", BwdBuffers)]

    def __init__(self, **kw):
        """Keywords: the top-level fields by name, the sub-structs' as search_<field> / cert_<field> / hist_<field> / bwd_<field>."""
        super().__init__(abi=ABI_VERSION)
        subs = (self, self.search, self.cert, self.hist, self.bwd)
        where = LoopBuffers._KW
        for k, v in kw.items():
            if v is not None:           # (a fresh struct is all zeros)
                i, f = where[k]
                setattr(subs[i], f, v)


LoopBuffers._KW = dict([(f[0], (0, f[0])) for f in LoopBuffers._fields_[:-4]] +
                       [(g + "_" + f[0], (i + 1, f[0])) for i, (g, cls) in enumerate((("search", SearchBuffers), ("cert", CertBuffers), ("hist", HistoryBuffers), ("bwd", BwdBuffers)))
                        for f in cls._fields_])


class GumbelLoop(ctypes.Structure):
    """dicp_gumbel_loop (include/dicp_hip.h)."""
    _fields_ = [("U", vp), ("seeds", vp), ("eps", f64), ("tau", f64), ("ps_t", vp), ("nbr", vp), ("lse", vp), ("g_nbr", vp), ("g_ps", vp)]


MAX_SEGMENTS = 16
CERT_OFF_FOR_GOOD = 1 << 20     # dicp_loop_buffers.cert.cloud[:, 2]: the cloud's certificates are off for the rest of the call


class SegmentPlan(ctypes.Structure):
    """dicp_segment_plan (include/dicp_hip.h)."""
    _fields_ = [("nseg", i32), ("k0", i32 * MAX_SEGMENTS), ("k1", i32 * MAX_SEGMENTS), ("new_order", i32 * MAX_SEGMENTS),
                ("cert_from", i32), ("pad0", i32), ("order", vp * MAX_SEGMENTS), ("keys", vp), ("cert_q", vp), ("cert_qu", vp), ("cert_count", vp), ("cert_cloud", vp), ("cert_set", vp),
                ("cert_nbr", vp), ("cert_gdirty", vp), ("cert_pend", vp), ("cert_cm", vp), ("cert_glist", vp), ("cert_gcount", vp), ("cert_slist", vp), ("cert_scount", vp)]


class Call(ctypes.Structure):
    """dicp_call (include/dicp_hip.h)."""
    _fields_ = [("src", vp), ("tgt", vp), ("T_init", vp), ("w0", vp), ("N", i32), ("n", i32), ("m", i32), ("c", i32), ("K", i32), ("dim", i32), ("need_grad", i32),
                ("n_resort", i32), ("resort", i32 * MAX_SEGMENTS), ("flags", i32), ("directions", i32), ("quantum", f64), ("tolerance", f64), ("workspace", vp), ("results", vp), ("T_out", vp), ("pc_out", vp)]


_sz = ctypes.c_size_t


class CallLayout(ctypes.Structure):
    """dicp_call_layout (include/dicp_hip.h)."""
    _fields_ = ([(k, _sz) for k in ("total", "zeroed", "results_total", "results_zeroed", "T", "pc", "deltas", "weights", "costs", "converged", "iterations", "matched_ratio", "pairs", "n_matched", "counters",
                                    "poses", "poses_search", "alive", "areg", "n_start", "partials", "tgs4", "tperm", "bucket", "brange", "keys", "tgt_sorted", "scratch",
                                    "scratch_bytes", "frame", "pose_s", "orders", "spos")]
                + [("n_orders", i32), ("m_pad", i32), ("nblk", i32), ("pad0", i32)])


class CallGrads(ctypes.Structure):
    """dicp_call_grads (include/dicp_hip.h)."""
    _fields_ = [("gT", vp), ("gsrc", vp), ("gtgt", vp), ("gT0", vp), ("gw", vp), ("workspace", vp), ("skip_eps", f64), ("tail_from", i32), ("pad0", i32), ("live_host", vp)]


class CallBackwardLayout(ctypes.Structure):
    """dicp_call_backward_layout (include/dicp_hip.h)."""
    _fields_ = ([(k, _sz) for k in ("total", "zeroed", "live", "arrive", "mref", "decisions", "far", "gpose", "gtmp", "src_s", "w_s", "gsrc_s", "gw_s", "slab", "gs", "gb",
                                    "partials", "tail_partials", "spos_ref")] + [("nblk_w", i32), ("pad0", i32)])


class LoopBackwardIn(ctypes.Structure):
    """dicp_loop_backward_in (include/dicp_hip.h)."""
    _fields_ = ([(k, vp) for k in ("src", "tgt_sorted", "w0", "tperm", "qorder", "spos", "poses", "deltas", "areg", "alive", "src_rows", "tgt_rows", "spos_of")]
                + [(k, i32) for k in ("N", "n", "m", "c", "K", "K_cap", "m_pad", "dim", "knn_variant", "spos_of_from")]
                + [(k, vp) for k in ("src_s", "w_s", "spos_ref")])


class KabschCall(ctypes.Structure):
    """dicp_kabsch_call (include/dicp_hip.h)."""
    _fields_ = [("src", vp), ("tgt", vp), ("T_start", vp), ("w0", vp), ("N", i32), ("n", i32), ("m", i32), ("c", i32), ("K", i32), ("trim_on", i32), ("directions", i32), ("pad0", i32),
                ("trim_dist", f64), ("quantum", f64), ("tolerance", f64), ("workspace", vp), ("T_out", vp), ("pc_out", vp)]


class KabschCallLayout(ctypes.Structure):
    """dicp_kabsch_call_layout (include/dicp_hip.h)."""
    _fields_ = ([(k, _sz) for k in ("total", "zeroed", "costs", "iterations", "pairs", "counters", "frame", "keys", "tperm", "bucket", "brange", "tgs4", "scratch", "scratch_bytes",
                                    "pose", "pose_search", "pose_used", "partials", "save", "idx", "rows_live", "orders", "gpose", "gacc")] + [("m_pad", i32), ("nblk", i32)])


class KabschCallGrads(ctypes.Structure):
    """dicp_kabsch_call_grads (include/dicp_hip.h)."""
    _fields_ = [("gT", vp), ("gsrc", vp), ("gtgt", vp), ("gw", vp)]


CALL_FIRST_SEARCH, CALL_NO_SMALL_LOOP, CALL_NBKT = 1, 2, 1024


class KabschBuffers(ctypes.Structure):
    """dicp_kabsch_buffers (include/dicp_hip.h)."""
    _fields_ = [("src", vp), ("tgt", vp), ("w_init", vp), ("c", i32), ("K", i32), ("knn_variant", i32), ("m_pad", i32), ("tgt4", vp), ("tperm", vp),
                ("qorder", vp), ("bucket", vp), ("brange", vp), ("nbkt", i32), ("pad0", i32), ("pairs", vp), ("frame", vp), ("pose", vp),
                ("pose_search", vp), ("pose_used", vp), ("idx", vp), ("partials", vp), ("save", vp), ("costs", vp), ("iterations", vp),
                ("rows_live", vp), ("tgt_rows", vp), ("counters", vp), ("tgt_f16", vp)]


_SIGNATURES = {
    "dicp_abi_version": ([], ctypes.c_int),
    "dicp_copy": ([vp, vp, ctypes.c_size_t, vp], ctypes.c_int),
    "dicp_zero": ([vp, ctypes.c_size_t, vp], ctypes.c_int),
    "dicp_padded_targets": ([i32], ctypes.c_int),
    "dicp_accumulate_blocks": ([i32], ctypes.c_int),
    "dicp_pack_target": ([i32, vp, i32, vp, vp, i32, i32, vp, i32, vp], ctypes.c_int),
    "dicp_search_frame": ([i32, vp, i32, vp, i32, i32, f64, i32, vp, vp, i32, vp, vp, vp], ctypes.c_int),
    "dicp_knn_f16_bytes": ([i32, i32], ctypes.c_size_t),
    "dicp_knn_f16_pack": ([vp, vp, i32, i32, i32, vp, vp], ctypes.c_int),
    "dicp_knn_f16_probe": ([vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp], ctypes.c_int),
    "dicp_knn": ([i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, i32, vp, vp], ctypes.c_int),
    "dicp_knn_sweep": ([i32, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, i32, i32, i32, i32, vp, vp, vp, i32, vp, vp, vp, i32, vp], ctypes.c_int),
    "dicp_window_blocks": ([i32, i32, i32], ctypes.c_int),
    "dicp_window_rows": ([i32], ctypes.c_int),
    "dicp_bwd_tail_max_blocks": ([i32], ctypes.c_int),
    "dicp_sweep_sort_scratch_bytes": ([i32, i32, i32], ctypes.c_size_t),
    "dicp_sweep_sort": ([i32, vp, i32, vp, vp, i32, i32, i32, vp, vp, i32, vp, vp, vp, ctypes.c_size_t, vp], ctypes.c_int),
    "dicp_sweep_build": ([i32, vp, i32, vp, vp, vp, i32, i32, i32, vp, vp, i32, vp], ctypes.c_int),
    "dicp_sweep_setup": ([i32, vp, i32, vp, i32, i32, i32, f64, i32, vp, vp, vp, i32, vp, vp, vp, ctypes.c_size_t, vp, vp, i32, vp, vp, i32, vp, vp, vp, vp], ctypes.c_int),
    "dicp_query_order": ([i32, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, i32, vp, i32, vp, vp, i32, vp, vp, vp], ctypes.c_int),
    "dicp_query_reorder": ([i32, vp, vp, vp, vp, vp, i32, i32, i32, vp, i32, vp, vp, i32, vp, vp, vp], ctypes.c_int),
    "dicp_query_keys": ([i32, vp, vp, i32, i32, vp, vp], ctypes.c_int),
    "dicp_loop_init": ([i32, vp, vp, ctypes.c_double, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp], ctypes.c_int),
    "dicp_search_pose": ([i32, vp, vp, i32, vp, vp], ctypes.c_int),
    "dicp_loop_finish": ([i32, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp], ctypes.c_int),
    "dicp_accumulate_bwd_window": ([i32, ctypes.POINTER(WeightParams), vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, i32, vp], ctypes.c_int),
    "dicp_resolve_matches": ([vp, vp, i32, vp, i32, i32, vp, vp], ctypes.c_int),
    "dicp_match_order_scratch_bytes": ([i32, i32, i32], ctypes.c_size_t),
    "dicp_match_order": ([i32, vp, vp, i32, i32, vp, ctypes.c_size_t, vp, vp], ctypes.c_int),
    "dicp_window_reduce": ([i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, i32, i32, vp], ctypes.c_int),
    "dicp_permute_add_rows": ([i32, vp, vp, i32, i32, i32, i32, i32, i32, vp, i32, i32, vp], ctypes.c_int),
    "dicp_pose_grad_in": ([i32, vp, vp, i32, vp], ctypes.c_int),
    "dicp_pose_grad_out": ([i32, vp, vp, i32, vp, i32, vp], ctypes.c_int),
    "dicp_permute_rows": ([i32, vp, vp, i32, i32, i32, i32, i32, i32, vp, i32, i32, vp], ctypes.c_int),
    "dicp_gather_rows": ([i32, vp, vp, i32, i32, i32, i32, vp, vp], ctypes.c_int),
    "dicp_scatter_add_rows": ([i32, vp, vp, i32, i32, i32, i32, vp, vp], ctypes.c_int),
    "dicp_pack_list": ([i32, vp, vp, vp, i32, i32, i32, vp, vp, vp], ctypes.c_int),
    "dicp_unpack_list": ([i32, vp, vp, vp, vp, i32, i32, i32, i32, vp], ctypes.c_int),
    "dicp_accumulate": ([i32, ctypes.POINTER(WeightParams), vp, vp, i32, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, i64, vp], ctypes.c_int),
    "dicp_step": ([i32, ctypes.POINTER(StepIO), i32, vp], ctypes.c_int),
    "dicp_icp_forward": ([i32, ctypes.POINTER(WeightParams), ctypes.POINTER(LoopBuffers), i32, i32, i32, i32, i32, f64, i32, i32, vp], ctypes.c_int),
    "dicp_icp_forward_plan": ([i32, ctypes.POINTER(WeightParams), ctypes.POINTER(LoopBuffers), ctypes.POINTER(SegmentPlan), i32, i32, i32, i32, i32, f64, vp], ctypes.c_int),
    "dicp_icp_backward": ([i32, ctypes.POINTER(WeightParams), ctypes.POINTER(LoopBuffers), i32, i32, i32, i32, vp, vp, i32, vp, vp, vp, vp, vp, vp, i32, i32, vp], ctypes.c_int),
    "dicp_step_bwd": ([i32, vp, vp, i32, i32, vp, vp, i64, vp, vp, vp, vp, i32, vp], ctypes.c_int),
    "dicp_accumulate_bwd": ([i32, ctypes.POINTER(WeightParams), vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp], ctypes.c_int),
    "dicp_gumbel_nn": ([i32, vp, vp, i32, vp, ctypes.c_uint32, f64, f64, i32, i32, i32, vp, vp, vp], ctypes.c_int),
    "dicp_gumbel_nn_bwd": ([i32, vp, vp, i32, vp, ctypes.c_uint32, f64, f64, vp, vp, vp, i32, i32, i32, vp, vp, vp], ctypes.c_int),
    "dicp_kabsch_accumulate": ([i32, vp, vp, i32, vp, vp, vp, i32, f64, vp, i32, i32, i32, vp, vp], ctypes.c_int),
    "dicp_kabsch_step": ([i32, vp, i32, vp, vp, vp, i32, vp], ctypes.c_int),
    "dicp_kabsch_step_bwd": ([i32, vp, vp, vp, i32, vp], ctypes.c_int),
    "dicp_kabsch_forward": ([i32, ctypes.POINTER(KabschBuffers), i32, i32, i32, i32, f64, i32, f64, i32, i32, vp], ctypes.c_int),
    "dicp_kabsch_bwd": ([i32, vp, vp, i32, vp, vp, vp, i32, f64, vp, vp, i32, i32, i32, vp, vp, vp, vp], ctypes.c_int),
    "dicp_transform_points": ([i32, vp, vp, vp, i32, i32, vp], ctypes.c_int),
    "dicp_transform_points_bwd": ([i32, vp, vp, vp, vp, vp, i32, i32, vp], ctypes.c_int),
    "dicp_loss_weight": ([i32, i32, i32, f64, f64, vp, i64, i32, vp, vp], ctypes.c_int),
    "dicp_loss_weight_bwd": ([i32, i32, i32, f64, f64, vp, vp, i64, i32, vp, vp], ctypes.c_int),
    "dicp_call_plan": ([i32, ctypes.POINTER(Call), ctypes.POINTER(CallLayout)], ctypes.c_int),
    "dicp_call_forward": ([i32, ctypes.POINTER(WeightParams), ctypes.POINTER(Call), vp], ctypes.c_int),
    "dicp_call_backward_plan": ([i32, ctypes.POINTER(WeightParams), ctypes.POINTER(Call), i32, i32, ctypes.POINTER(CallBackwardLayout)], ctypes.c_int),
    "dicp_call_backward": ([i32, ctypes.POINTER(WeightParams), ctypes.POINTER(Call), ctypes.POINTER(CallGrads), vp], ctypes.c_int),
    "dicp_loop_backward_plan": ([i32, ctypes.POINTER(WeightParams), ctypes.POINTER(LoopBackwardIn), i32, i32, ctypes.POINTER(CallBackwardLayout)], ctypes.c_int),
    "dicp_loop_backward": ([i32, ctypes.POINTER(WeightParams), ctypes.POINTER(LoopBackwardIn), ctypes.POINTER(CallGrads), vp], ctypes.c_int),
    "dicp_loop_backward_prepare": ([i32, ctypes.POINTER(LoopBackwardIn), vp, vp, vp, vp], ctypes.c_int),
    "dicp_kabsch_call_plan": ([i32, ctypes.POINTER(KabschCall), ctypes.POINTER(KabschCallLayout)], ctypes.c_int),
    "dicp_kabsch_call_forward": ([i32, ctypes.POINTER(KabschCall), vp], ctypes.c_int),
    "dicp_kabsch_call_backward": ([i32, ctypes.POINTER(KabschCall), ctypes.POINTER(KabschCallGrads), vp], ctypes.c_int),
}
EXPORTS = tuple(_SIGNATURES)

_lib = None
_lock = threading.Lock()


FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=on", "-mllvm", "-amdgpu-mfma-vgpr-form", "-Wall", "-Wno-unused-function"]


def build(force=False, verbose=False):
    """Compile the HIP sources for gfx950 into dicp_amd/libdicp_hip.so (hipcc cross-compiles without a GPU): one object per translation
    unit (in parallel; only those whose source or a header is newer), then one link."""
    hdr_time = max(os.path.getmtime(p) for p in HEADERS + [os.path.abspath(__file__)])       # (this file holds the compiler flags)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(_HERE, "csrc", "_obj")
    os.makedirs(objdir, exist_ok=True)
    # -amdgpu-mfma-vgpr-form: the matrix-core search variants read their results with vector instructions right away: let the MFMA write VGPRs
    # directly (no v_accvgpr_read per result; profiles/r01_knn_variants_ab.txt); no effect on the other kernels
    # -ffp-contract=on: a*b+c fuses where the SOURCE writes it in one expression, never across statements after inlining ("fast", the
    # HIP default, did: the same point_forward then rounded differently in two instantiations of accumulate_kernel) -- results are a
    # function of the source, not of the optimiser's context
    jobs = []
    for src in SOURCES:
        obj = os.path.join(objdir, os.path.splitext(os.path.basename(src))[0] + ".o")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_time):
            jobs.append((src, obj))
    objs = [os.path.join(objdir, os.path.splitext(os.path.basename(src))[0] + ".o") for src in SOURCES]
    if not jobs and os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= max(os.path.getmtime(o) for o in objs):
        return LIB_PATH

    def compile_one(job):
        cmd = [hipcc] + FLAGS + ["-I", os.path.join(_ROOT, "include"), "-c", "-o", job[1] + ".tmp", job[0]]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        os.replace(job[1] + ".tmp", job[1])
    if jobs:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(jobs), 4)) as pool:
            list(pool.map(compile_one, jobs))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH + ".tmp"] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


def load():
    """Load the C-ABI library; raises (never falls back) if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    # libdicp_hip.so must share PyTorch's HIP runtime (PyTorch bundles its own libamdhip64.so.7):
    # importing torch first makes the loader resolve our NEEDED libamdhip64.so.7 to that copy.  A
    # second runtime (from /opt/rocm) in one process sees no device (hipErrorNoDevice).
    import torch  # noqa: F401
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    "dicp_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
            lib = ctypes.CDLL(LIB_PATH)
            for name, (args, res) in _SIGNATURES.items():
                fn = getattr(lib, name)
                fn.argtypes = args
                fn.restype = res
            if lib.dicp_abi_version() != ABI_VERSION:
                raise RuntimeError("dicp_amd: libdicp_hip.so ABI version mismatch; rebuild it")
            _lib = lib
    return _lib


def check(status, what):
    if status == 0:
        return
    if status > 0:
        raise RuntimeError("dicp_amd: %s rejected its arguments: %s" % (what, _ERRORS.get(status, status)))
    raise RuntimeError("dicp_amd: %s failed to launch: hipError_t %d" % (what, -status))
