// dicp_call_*: one eager ICP call of the sorted-sweep path behind ONE host call per direction (host code only: every kernel is reached through
// the entry points of dicp_hip.h).  A mid-size call -- 32 clouds of 4096 points, 10 iterations: 0.55 ms of kernels -- spent more time than that in
// the interpreter that prepared it buffer by buffer (profiles/r03: 0.96 ms per call); here the caller makes ONE allocation per direction, whose
// carving is dicp_call_plan's / dicp_call_backward_plan's, and this file does what dicp_amd/_ops.py's ICPLoop does for the same shapes, call for
// call and argument for argument (tests/test_gpu_call.py holds the two against each other bit for bit).
#include <hip/hip_runtime.h>
#include <string.h>
#include "dicp_hip.h"

namespace {

struct Carver {
    size_t off = 0;
    size_t take(size_t bytes) {
        size_t o = off;
        off += (bytes + 255) / 256 * 256;
        return o;
    }
};

inline size_t esize(int dtype) { return dtype == DICP_F64 ? 8 : 4; }

inline bool call_ok(int dtype, const dicp_call* c) {
    if (dtype != DICP_F32 && dtype != DICP_F64) return false;
    if (c->N < 1 || c->n < 1 || c->m < 1 || c->K < 1 || (c->c != 3 && c->c != 6) || (c->dim != 2 && c->dim != 3)) return false;
    if (c->n_resort < 0 || c->n_resort >= DICP_MAX_SEGMENTS) return false;
    int prev = 0;
    for (int i = 0; i < c->n_resort; ++i) {          // strictly ascending, inside (0, K)
        if (c->resort[i] <= prev || c->resort[i] >= c->K) return false;
        prev = c->resort[i];
    }
    return true;
}

inline char* at(const dicp_call* c, size_t off) { return (char*)c->workspace + off; }

}  // namespace

extern "C" {

int dicp_call_plan(int dtype, const dicp_call* c, dicp_call_layout* L) {
    if (!c || !L) return DICP_ERR_NULL;
    if (!call_ok(dtype, c)) return dtype != DICP_F32 && dtype != DICP_F64 ? DICP_ERR_DTYPE : DICP_ERR_SHAPE;
    const size_t es = esize(dtype), N = c->N, n = c->n, K = c->K;
    memset(L, 0, sizeof(*L));
    L->m_pad = dicp_padded_targets(c->m);
    L->nblk = dicp_accumulate_blocks(c->n);
    L->n_orders = 1 + c->n_resort;
    const size_t m_pad = L->m_pad;
    Carver w;
    // zero-initialised loop state first: one fill
    L->deltas = w.take(N * K * 6 * es);
    L->costs = w.take(N * K * es);
    L->converged = w.take(N);
    L->iterations = w.take(N * es);
    L->matched_ratio = w.take(N * es);
    L->n_matched = w.take(N * es);
    L->counters = w.take(K * 4);
    L->pairs = w.take(DICP_PAIR_SHARDS * 8);
    L->zeroed = w.off;
    L->T = w.take(N * 16 * es);
    L->pc = w.take(N * n * 3 * es);
    L->weights = w.take(N * K * n * es);
    L->poses = w.take((K + 1) * N * 12 * es);
    L->poses_search = w.take((K + 1) * N * 12 * es);
    L->alive = w.take((K + 1) * N * es);
    L->areg = c->need_grad ? w.take(K * N * 36 * 8) : 0;
    L->n_start = w.take(N * es);
    L->partials = w.take(N * L->nblk * DICP_NACC_PAD * es);
    L->tgs4 = w.take(N * m_pad * 4 * es);
    L->tperm = w.take(N * m_pad * 4);
    L->bucket = w.take(N * (DICP_CALL_NBKT + 1) * 4);
    L->brange = w.take(N * 2 * es);
    L->keys = w.take(N * m_pad * es);
    L->tgt_sorted = w.take(N * m_pad * c->c * es);
    L->scratch_bytes = dicp_sweep_sort_scratch_bytes(dtype, c->N, L->m_pad);
    L->scratch = L->scratch_bytes ? w.take(L->scratch_bytes) : 0;
    L->frame = w.take(N * 12 * es);
    L->pose_s = w.take(N * 12 * es);
    L->orders = w.take((size_t)L->n_orders * N * n * 4);
    L->spos = w.take((c->need_grad ? K : 1) * N * n * 4);
    L->total = w.off;
    return 0;
}

int dicp_call_forward(int dtype, const dicp_weight_params* prm, const dicp_call* c, void* stream) {
    if (!prm || !c || !c->src || !c->tgt || !c->T_init || !c->workspace) return DICP_ERR_NULL;
    dicp_call_layout L;
    if (int rc = dicp_call_plan(dtype, c, &L)) return rc;
    if (((uintptr_t)c->workspace & 255) != 0) return DICP_ERR_ALIGN;
    const size_t es = esize(dtype), N = c->N, n = c->n;
    const int K = c->K;
    hipStream_t st = (hipStream_t)stream;
    if (hipError_t e = hipMemsetAsync(c->workspace, 0, L.zeroed, st)) return -(int)e;
    int32_t* order0 = (int32_t*)at(c, L.orders);
    int32_t* spos = (int32_t*)at(c, L.spos);
    // frame, sort, rows, the search pose of iteration 0 and the first query order; iteration 0's search right behind them
    if (int rc = dicp_sweep_setup(dtype, c->tgt, c->c, nullptr, c->N, c->m, L.m_pad, c->quantum, c->directions, at(c, L.frame), at(c, L.keys),
                                  (int32_t*)at(c, L.tperm), DICP_CALL_NBKT, (int32_t*)at(c, L.bucket), at(c, L.brange), L.scratch_bytes ? at(c, L.scratch) : nullptr,
                                  L.scratch_bytes, at(c, L.tgs4), at(c, L.tgt_sorted), c->c, c->src, nullptr, c->n, c->T_init, at(c, L.pose_s), order0, stream))
        return rc;
    const bool first_search = (c->flags & DICP_CALL_FIRST_SEARCH) != 0;
    if (first_search)
        if (int rc = dicp_knn_sweep(dtype, c->src, at(c, L.pose_s), at(c, L.tgs4), (int32_t*)at(c, L.tperm), order0, (int32_t*)at(c, L.bucket), at(c, L.brange),
                                    DICP_CALL_NBKT, nullptr, nullptr, c->N, c->n, c->m, L.m_pad, nullptr, spos, (unsigned long long*)at(c, L.pairs), 0, nullptr, stream))
            return rc;
    if (int rc = dicp_loop_init(dtype, c->T_init, c->w0, prm->match_thresh, prm->mode == DICP_PT2PT ? 3 : 1, c->N, c->n, at(c, L.poses), at(c, L.alive),
                                at(c, L.n_start), at(c, L.frame), at(c, L.poses_search), nullptr, nullptr, nullptr, 2 * (K + 1), stream))
        return rc;
    // the segments: cut where the queries are re-ordered
    dicp_segment_plan SP;
    memset(&SP, 0, sizeof(SP));
    SP.cert_from = -1;
    SP.keys = at(c, L.keys);
    int cuts[DICP_MAX_SEGMENTS + 1], nc = 0;
    cuts[nc++] = 0;
    for (int i = 0; i < c->n_resort; ++i) cuts[nc++] = c->resort[i];
    cuts[nc] = K;
    SP.nseg = nc;
    for (int s = 0; s < nc; ++s) {
        SP.k0[s] = cuts[s];
        SP.k1[s] = cuts[s + 1];
        SP.new_order[s] = s > 0;                       // (segment 0 searches in the order dicp_sweep_setup made)
        SP.order[s] = order0 + (size_t)s * N * n;
    }
    dicp_loop_buffers B;
    memset(&B, 0, sizeof(B));
    B.src = c->src; B.tgt = c->tgt; B.w_init = c->w0; B.c = c->c; B.K = K;
    B.knn_variant = DICP_KNN_SWEEP | ((c->flags & DICP_CALL_NO_SMALL_LOOP) ? (1 << 25) : 0);
    B.m_pad = L.m_pad;
    B.tgt4 = at(c, L.tgs4); B.tperm = (int32_t*)at(c, L.tperm); B.bucket = (int32_t*)at(c, L.bucket); B.brange = at(c, L.brange);
    B.nbkt = DICP_CALL_NBKT; B.idx_per_iter = c->need_grad ? 1 : 0; B.pairs = (unsigned long long*)at(c, L.pairs);
    B.poses = at(c, L.poses); B.deltas = at(c, L.deltas); B.costs = at(c, L.costs); B.areg = c->need_grad ? (double*)at(c, L.areg) : nullptr;
    B.alive = at(c, L.alive); B.converged = (uint8_t*)at(c, L.converged); B.iterations = at(c, L.iterations); B.matched_ratio = at(c, L.matched_ratio);
    B.n_start = at(c, L.n_start); B.n_matched = at(c, L.n_matched);
    B.tgt_sorted = at(c, L.tgt_sorted); B.tgt_sorted_stride = c->c;
    B.w_iter = c->n; B.w_stride = (int64_t)K * c->n; B.w = at(c, L.weights);
    B.spos = spos;
    B.partials = at(c, L.partials); B.counters = (int32_t*)at(c, L.counters);
    B.frame = at(c, L.frame); B.poses_search = at(c, L.poses_search);
    B.first_search_done = first_search ? 1 : 0;
    if (int rc = dicp_icp_forward_plan(dtype, prm, &B, &SP, c->N, c->n, c->m, c->dim, 1, c->tolerance, stream)) return rc;
    char* pose_K = at(c, L.poses) + (size_t)K * N * 12 * es;
    if (int rc = dicp_loop_finish(dtype, pose_K, at(c, L.alive) + (size_t)K * N * es, at(c, L.n_start), at(c, L.n_matched), K, c->N, at(c, L.iterations),
                                  at(c, L.matched_ratio), c->T_out ? c->T_out : at(c, L.T), stream))
        return rc;
    return dicp_transform_points(dtype, c->src, pose_K, c->pc_out ? c->pc_out : at(c, L.pc), c->N, c->n, stream);
}

int dicp_call_backward_plan(int dtype, const dicp_weight_params* prm, const dicp_call* c, int want_tgt, int want_w, dicp_call_backward_layout* L) {
    if (!prm || !c || !L) return DICP_ERR_NULL;
    if (!call_ok(dtype, c)) return dtype != DICP_F32 && dtype != DICP_F64 ? DICP_ERR_DTYPE : DICP_ERR_SHAPE;
    const size_t es = esize(dtype), N = c->N, n = c->n, K = c->K;
    const int m_pad = dicp_padded_targets(c->m), cv = prm->mode == DICP_PT2PL ? 6 : 3;
    memset(L, 0, sizeof(*L));
    L->nblk_w = dicp_window_blocks(dtype, c->n, m_pad);
    const size_t nblk = (size_t)(L->nblk_w > dicp_accumulate_blocks(c->n) ? L->nblk_w : dicp_accumulate_blocks(c->n));
    Carver w;
    // zero-initialised: the truncated sweep's state, and the rows matched outside every window
    L->mref = w.take(N * 8);
    L->decisions = w.take(N * 4);
    L->live = w.take((K + 1) * 4);
    L->arrive = w.take((N + 1) * 4);
    L->far = want_tgt ? w.take(N * (size_t)m_pad * cv * es) : 0;
    L->zeroed = w.off;
    L->gpose = w.take(N * 12 * 8);
    L->gtmp = w.take(N * 12 * 8);
    L->src_s = w.take(N * n * 3 * es);
    L->w_s = c->w0 ? w.take(N * n * es) : 0;
    L->gsrc_s = w.take(N * n * 3 * es);
    L->gw_s = want_w ? w.take(N * n * es) : 0;
    L->slab = want_tgt ? w.take(N * (size_t)L->nblk_w * dicp_window_rows(dtype) * cv * es) : 0;
    L->gs = w.take(N * 36 * es);
    L->gb = w.take(N * 6 * es);
    L->partials = w.take(N * nblk * DICP_NBWD_PAD * es);
    L->tail_partials = w.take(N * (size_t)L->nblk_w * DICP_NBWD_PAD * es);
    L->total = w.off;
    return 0;
}

int dicp_call_backward(int dtype, const dicp_weight_params* prm, const dicp_call* c, const dicp_call_grads* g, void* stream) {
    if (!prm || !c || !g || !c->workspace || !g->workspace || !g->gsrc || !g->gT0 || !c->src || !c->tgt) return DICP_ERR_NULL;
    if (!c->need_grad) return DICP_ERR_ENUM;
    const int want_tgt = g->gtgt != nullptr, want_w = g->gw != nullptr;
    if (want_w && !c->w0) return DICP_ERR_NULL;
    dicp_call_layout F;
    dicp_call_backward_layout L;
    if (int rc = dicp_call_plan(dtype, c, &F)) return rc;
    if (int rc = dicp_call_backward_plan(dtype, prm, c, want_tgt, want_w, &L)) return rc;
    const int cv = prm->mode == DICP_PT2PL ? 6 : 3;
    if (c->c != cv) return DICP_ERR_SHAPE;           // (every element of gtgt is written once by dicp_window_reduce: the row is the gradient's row)
    if ((((uintptr_t)c->workspace | (uintptr_t)g->workspace) & 255) != 0) return DICP_ERR_ALIGN;
    const size_t N = c->N, n = c->n;
    const int K = c->K;
    const bool skip = g->skip_eps > 0.0;
    int tail_from = skip ? g->tail_from : 0;
    if (tail_from < 0) return DICP_ERR_SHAPE;
    if (tail_from > K) tail_from = K;
    hipStream_t st = (hipStream_t)stream;
    char* W = (char*)g->workspace;
    if (hipError_t e = hipMemsetAsync(W, 0, L.zeroed, st)) return -(int)e;
    double* gpose = (double*)(W + L.gpose);
    double* gtmp = (double*)(W + L.gtmp);
    if (int rc = dicp_pose_grad_in(dtype, g->gT, gpose, c->N, stream)) return rc;
    // ONE slot order serves every iteration: the last query order of the forward
    const int32_t* qo = (const int32_t*)at(c, F.orders) + (size_t)(F.n_orders - 1) * N * n;
    if (int rc = dicp_gather_rows(dtype, c->src, qo, c->N, c->n, c->n, 3, W + L.src_s, stream)) return rc;
    if (c->w0)
        if (int rc = dicp_gather_rows(dtype, c->w0, qo, c->N, c->n, c->n, 1, W + L.w_s, stream)) return rc;
    int32_t* spos = (int32_t*)at(c, F.spos);
    const int32_t* spos_ref = spos + (size_t)(K - 1) * N * n;      // windows placed by the last iteration's matches
    dicp_loop_buffers B;
    memset(&B, 0, sizeof(B));
    B.src = W + L.src_s; B.tgt = at(c, F.tgt_sorted); B.w_init = c->w0 ? W + L.w_s : nullptr; B.c = c->c; B.K = K;
    B.knn_variant = DICP_KNN_SWEEP | ((c->flags & DICP_CALL_NO_SMALL_LOOP) ? (1 << 25) : 0);
    B.m_pad = F.m_pad; B.idx_per_iter = 1; B.qorder = qo; B.spos = spos; B.spos_ref = spos_ref; B.gts_far = want_tgt ? W + L.far : nullptr;
    B.poses = at(c, F.poses); B.deltas = at(c, F.deltas); B.areg = (double*)at(c, F.areg); B.alive = at(c, F.alive);
    B.bwd_overwrite = 1;
    if (skip) {
        B.bwd_skip = (int32_t*)(W + L.decisions); B.bwd_mref = (double*)(W + L.mref); B.bwd_live = (int32_t*)(W + L.live);
        B.bwd_tail_arrive = (int32_t*)(W + L.arrive);
    }
    B.bwd_skip_eps = g->skip_eps;
    B.bwd_tail_from = tail_from;
    B.bwd_tail_partials = tail_from > 0 ? W + L.tail_partials : nullptr;
    if (int rc = dicp_icp_backward(dtype, prm, &B, c->N, c->n, c->m, c->dim, gpose, gtmp, 0, W + L.gs, W + L.gb, W + L.gsrc_s, want_tgt ? W + L.slab : nullptr,
                                   want_w ? W + L.gw_s : nullptr, W + L.partials, 0, K, stream))
        return rc;
    // the tail launch leaves the cotangent of pose_0 with the last pose sums already in it (the first iteration of the pass is never the tail's)
    const bool folded = tail_from > 0 && (tail_from - (tail_from >= K ? 1 : 0)) > 0;
    if (K % 2) { double* t = gpose; gpose = gtmp; gtmp = t; }
    if (skip && g->live_host)        // where the sweeps ended, for the next call's tail (and the tail's error word)
        if (hipError_t e = hipMemcpyAsync(g->live_host, W + L.live, (size_t)(K + 1) * 4, hipMemcpyDeviceToHost, st)) return -(int)e;
    if (int rc = dicp_permute_rows(dtype, W + L.gsrc_s, qo, c->N, c->n, c->n, c->n, 3, 3, g->gsrc, c->n, 3, stream)) return rc;
    if (want_w)
        if (int rc = dicp_permute_rows(dtype, W + L.gw_s, qo, c->N, c->n, c->n, c->n, 1, 1, g->gw, c->n, 1, stream)) return rc;
    if (want_tgt)
        if (int rc = dicp_window_reduce(dtype, W + L.slab, spos_ref, qo, (const int32_t*)at(c, F.tperm), W + L.far, nullptr, c->N, c->n, c->m, F.m_pad, cv,
                                        g->gtgt, c->c, 1, stream))
            return rc;
    return dicp_pose_grad_out(dtype, gpose, folded ? nullptr : W + L.partials, folded ? 0 : L.nblk_w, g->gT0, c->N, stream);
}

}  // extern "C"
