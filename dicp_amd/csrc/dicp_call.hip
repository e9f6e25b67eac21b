// dicp_call_*: one eager ICP call of the sorted-sweep path behind ONE host call per direction (host code only: every kernel is reached through
// the entry points of dicp_hip.h).  A mid-size call -- 32 clouds of 4096 points, 10 iterations: 0.55 ms of kernels -- spent more time than that in
// the interpreter that prepared it buffer by buffer (profiles/r03: 0.96 ms per call); here the caller makes ONE allocation per direction, whose
// carving is dicp_call_plan's / dicp_call_backward_plan's, and this file does what dicp_amd/_ops.py's ICPLoop does for the same shapes, call for
// call and argument for argument (tests/test_gpu_call.py holds the two against each other bit for bit).
#include <hip/hip_runtime.h>
#include <string.h>
#include "dicp_hip.h"
#include "dicp_fill.h"

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

inline int zero_fill(void* p, size_t bytes, hipStream_t st) { return dicp_fill::zero(p, bytes, st); }      // (a kernel, not hipMemsetAsync: dicp_fill.h)

inline char* at(const dicp_call* c, size_t off) { return (char*)c->workspace + off; }
inline char* res(const dicp_call* c, size_t off) { return (char*)c->results + off; }

// ---- the SVD loop's one-call form: three per-cloud trifles that were torch arithmetic on the host side
template <typename T>
__global__ void kabsch_prep_kernel(int32_t* __restrict__ rows_live, int N, int n) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < N) rows_live[b] = n;                                 // every source row takes part (a cloud's count drops to 0 when it is frozen)
}
// T (N,4,4) from the pose [C | r]; a cloud that never met the tolerance reports all K iterations (ICP.py:585-589)
template <typename T>
__global__ void kabsch_finish_kernel(const T* __restrict__ pose, T* __restrict__ iterations, T* __restrict__ T_out, int N, int K) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * 16) return;
    const int b = i >> 4, e = i & 15, row = e >> 2, col = e & 3;
    T v = row == 3 ? (col == 3 ? T(1) : T(0)) : (col == 3 ? pose[b * 12 + 9 + row] : pose[b * 12 + row * 3 + col]);
    T_out[i] = v;
    if (e == 0 && iterations[b] == T(0)) iterations[b] = T(K);
}
// the cotangent of the pose [C | r] out of the cotangent of T (NULL: zeros)
template <typename T>
__global__ void kabsch_gpose_kernel(const T* __restrict__ gT, T* __restrict__ gpose, int N) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * 12) return;
    const int b = i / 12, e = i % 12;
    gpose[i] = gT ? (e < 9 ? gT[b * 16 + (e / 3) * 4 + e % 3] : gT[b * 16 + (e - 9) * 4 + 3]) : T(0);
}

inline bool kabsch_call_ok(int dtype, const dicp_kabsch_call* c) {
    return (dtype == DICP_F32 || dtype == DICP_F64) && c->N >= 1 && c->n >= 1 && c->m >= 1 && c->K >= 1 && (c->c == 3 || c->c == 6);
}
inline char* kat(const dicp_kabsch_call* c, size_t off) { return (char*)c->workspace + off; }
inline int last_launch() { const hipError_t e = hipGetLastError(); return e == hipSuccess ? 0 : -(int)e; }

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
    // the non-differentiable results: an allocation of their own (dicp_call.results), the zero-initialised ones first
    Carver rs;
    L->deltas = rs.take(N * K * 6 * es);
    L->costs = rs.take(N * K * es);
    L->converged = rs.take(N);
    L->iterations = rs.take(N * es);
    L->matched_ratio = rs.take(N * es);
    L->results_zeroed = rs.off;
    L->weights = rs.take(N * K * n * es);
    L->results_total = rs.off;
    Carver w;
    // zero-initialised loop state first: one fill
    L->n_matched = w.take(N * es);
    L->counters = w.take(K * 4);
    L->pairs = w.take(DICP_PAIR_SHARDS * 8);
    L->zeroed = w.off;
    L->T = w.take(N * 16 * es);
    L->pc = w.take(N * n * 3 * es);
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
    if (!prm || !c || !c->src || !c->tgt || !c->T_init || !c->workspace || !c->results) return DICP_ERR_NULL;
    dicp_call_layout L;
    if (int rc = dicp_call_plan(dtype, c, &L)) return rc;
    if (((uintptr_t)c->workspace & 255) != 0 || ((uintptr_t)c->results & 255) != 0) return DICP_ERR_ALIGN;
    const size_t es = esize(dtype), N = c->N, n = c->n;
    const int K = c->K;
    hipStream_t st = (hipStream_t)stream;
    if (int rc = zero_fill(c->workspace, L.zeroed, st)) return rc;
    if (int rc = zero_fill(c->results, L.results_zeroed, st)) return rc;
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
                                    DICP_CALL_NBKT, nullptr, nullptr, c->N, c->n, c->m, L.m_pad, nullptr, spos, (unsigned long long*)at(c, L.pairs), 0, nullptr, nullptr, nullptr, 0, stream))
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
    B.abi = DICP_ABI_VERSION;
    B.src = c->src; B.tgt = c->tgt; B.w_init = c->w0; B.c = c->c; B.K = K;
    B.search.knn_variant = DICP_KNN_SWEEP | ((c->flags & DICP_CALL_NO_SMALL_LOOP) ? (1 << 25) : 0);
    B.search.m_pad = L.m_pad;
    B.search.tgt4 = at(c, L.tgs4); B.search.tperm = (int32_t*)at(c, L.tperm); B.search.bucket = (int32_t*)at(c, L.bucket); B.search.brange = at(c, L.brange);
    B.search.nbkt = DICP_CALL_NBKT; B.hist.per_iter = c->need_grad ? 1 : 0; B.search.pairs = (unsigned long long*)at(c, L.pairs);
    B.hist.poses = at(c, L.poses); B.hist.deltas = res(c, L.deltas); B.hist.costs = res(c, L.costs); B.hist.areg = c->need_grad ? (double*)at(c, L.areg) : nullptr;
    B.hist.alive = at(c, L.alive); B.converged = (uint8_t*)res(c, L.converged); B.iterations = res(c, L.iterations); B.matched_ratio = res(c, L.matched_ratio);
    B.n_start = at(c, L.n_start); B.n_matched = at(c, L.n_matched);
    B.search.tgt_sorted = at(c, L.tgt_sorted); B.search.tgt_sorted_stride = c->c;
    B.hist.w_iter = c->n; B.hist.w_stride = (int64_t)K * c->n; B.hist.w = res(c, L.weights);
    B.hist.spos = spos;
    B.partials = at(c, L.partials); B.counters = (int32_t*)at(c, L.counters);
    B.search.frame = at(c, L.frame); B.search.poses = at(c, L.poses_search);
    B.search.first_done = first_search ? 1 : 0;
    if (int rc = dicp_icp_forward_plan(dtype, prm, &B, &SP, c->N, c->n, c->m, c->dim, 1, c->tolerance, stream)) return rc;
    char* pose_K = at(c, L.poses) + (size_t)K * N * 12 * es;
    if (int rc = dicp_loop_finish(dtype, pose_K, at(c, L.alive) + (size_t)K * N * es, at(c, L.n_start), at(c, L.n_matched), K, c->N, res(c, L.iterations),
                                  res(c, L.matched_ratio), c->T_out ? c->T_out : at(c, L.T), stream))
        return rc;
    return dicp_transform_points(dtype, c->src, pose_K, c->pc_out ? c->pc_out : at(c, L.pc), c->N, c->n, stream);
}

// the reverse sweep of ONE windowed run over iterations [0, K) -- dicp_call_backward's and dicp_loop_backward's common body
static int backward_layout(int dtype, const dicp_weight_params* prm, int N_, int n_, int m_, int Kcap_, bool has_w0, int want_tgt, int want_w, dicp_call_backward_layout* L) {
    const size_t es = esize(dtype), N = N_, n = n_, K = Kcap_;
    const int m_pad = dicp_padded_targets(m_), cv = prm->mode == DICP_PT2PL ? 6 : 3;
    memset(L, 0, sizeof(*L));
    L->nblk_w = dicp_window_blocks(dtype, n_, m_pad);
    const size_t nblk = (size_t)(L->nblk_w > dicp_accumulate_blocks(n_) ? L->nblk_w : dicp_accumulate_blocks(n_));
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
    L->w_s = has_w0 ? w.take(N * n * es) : 0;
    L->gsrc_s = w.take(N * n * 3 * es);
    L->gw_s = want_w ? w.take(N * n * es) : 0;
    L->slab = want_tgt ? w.take(N * (size_t)L->nblk_w * dicp_window_rows(dtype) * cv * es) : 0;
    L->gs = w.take(N * 36 * es);
    L->gb = w.take(N * 6 * es);
    L->partials = w.take(N * nblk * DICP_NBWD_PAD * es);
    L->tail_partials = w.take(N * (size_t)L->nblk_w * DICP_NBWD_PAD * es);
    L->spos_ref = w.take(N * n * 4);
    L->total = w.off;
    return 0;
}

// the part of the pass that needs no cotangent: the slot-order copies, and the reference matches out of a history kept by reference
static int prepare_pass(int dtype, const dicp_loop_backward_in* f, void* src_s, void* w_s, int32_t* spos_ref, void* stream) {
    if (int rc = dicp_gather_rows(dtype, f->src, f->qorder, f->N, f->n, f->n, 3, src_s, stream)) return rc;
    if (f->w0)
        if (int rc = dicp_gather_rows(dtype, f->w0, f->qorder, f->N, f->n, f->n, 1, w_s, stream)) return rc;
    if (spos_ref)
        if (int rc = dicp_resolve_matches(f->spos, f->spos_of, f->K - 1, f->src_rows, f->N, f->n, spos_ref, stream)) return rc;
    return 0;
}

static int backward_once(int dtype, const dicp_weight_params* prm, const dicp_loop_backward_in* f, const dicp_call_grads* g, void* stream) {
    const int want_tgt = g->gtgt != nullptr, want_w = g->gw != nullptr;
    if (want_w && !f->w0) return DICP_ERR_NULL;
    const int cv = prm->mode == DICP_PT2PL ? 6 : 3;
    if (f->c != cv) return DICP_ERR_SHAPE;           // (every element of gtgt is written once by dicp_window_reduce: the row is the gradient's row)
    if (((uintptr_t)g->workspace & 255) != 0) return DICP_ERR_ALIGN;
    dicp_call_backward_layout L;
    backward_layout(dtype, prm, f->N, f->n, f->m, f->K_cap, f->w0 != nullptr, want_tgt, want_w, &L);
    const size_t N = f->N, n = f->n;
    const int K = f->K;
    const bool skip = g->skip_eps > 0.0;
    int tail_from = skip ? g->tail_from : 0;
    if (tail_from < 0) return DICP_ERR_SHAPE;
    if (tail_from > K) tail_from = K;
    hipStream_t st = (hipStream_t)stream;
    char* W = (char*)g->workspace;
    if (int rc = zero_fill(W, L.zeroed, st)) return rc;
    double* gpose = (double*)(W + L.gpose);
    double* gtmp = (double*)(W + L.gtmp);
    if (int rc = dicp_pose_grad_in(dtype, g->gT, gpose, f->N, stream)) return rc;
    // ONE slot order serves every iteration: the last query order of the forward
    const int32_t* qo = f->qorder;
    const bool prepared = f->src_s != nullptr;                       // (dicp_loop_backward_prepare ran behind the forward)
    if (prepared && f->w0 && !f->w_s) return DICP_ERR_NULL;
    const char* src_s = prepared ? (const char*)f->src_s : W + L.src_s;
    const char* w_s = f->w0 ? (prepared ? (const char*)f->w_s : W + L.w_s) : nullptr;
    // windows placed by the last iteration's matches (a plain array of them: the history may be kept by reference)
    const int32_t* spos_ref = f->spos + (size_t)(K - 1) * N * n;
    const bool by_reference = f->spos_of && K - 1 >= f->spos_of_from;
    if (prepared) {
        if (by_reference) { if (!f->spos_ref) return DICP_ERR_NULL; spos_ref = f->spos_ref; }
    } else {
        if (int rc = prepare_pass(dtype, f, W + L.src_s, f->w0 ? W + L.w_s : nullptr, by_reference ? (int32_t*)(W + L.spos_ref) : nullptr, stream)) return rc;
        if (by_reference) spos_ref = (const int32_t*)(W + L.spos_ref);
    }
    dicp_loop_buffers B;
    memset(&B, 0, sizeof(B));
    B.abi = DICP_ABI_VERSION;
    B.src = src_s; B.tgt = f->tgt_sorted; B.w_init = w_s; B.c = f->c; B.K = f->K_cap;
    B.search.knn_variant = f->knn_variant;
    B.search.m_pad = f->m_pad; B.hist.per_iter = 1; B.search.qorder = qo; B.hist.spos = (int32_t*)f->spos; B.bwd.spos_ref = spos_ref; B.bwd.gts_far = want_tgt ? W + L.far : nullptr;
    B.hist.spos_of = (int32_t*)f->spos_of; B.hist.spos_of_from = f->spos_of_from;
    B.hist.poses = (void*)f->poses; B.hist.deltas = (void*)f->deltas; B.hist.areg = (double*)f->areg; B.hist.alive = (void*)f->alive;
    B.src_rows = f->src_rows; B.tgt_rows = f->tgt_rows;
    B.bwd.overwrite = 1;
    if (skip) {
        B.bwd.skip = (int32_t*)(W + L.decisions); B.bwd.mref = (double*)(W + L.mref); B.bwd.live = (int32_t*)(W + L.live);
        B.bwd.tail_arrive = (int32_t*)(W + L.arrive);
    }
    B.bwd.skip_eps = g->skip_eps;
    B.bwd.tail_from = tail_from;
    B.bwd.tail_partials = tail_from > 0 ? W + L.tail_partials : nullptr;
    if (int rc = dicp_icp_backward(dtype, prm, &B, f->N, f->n, f->m, f->dim, gpose, gtmp, 0, W + L.gs, W + L.gb, W + L.gsrc_s, want_tgt ? W + L.slab : nullptr,
                                   want_w ? W + L.gw_s : nullptr, W + L.partials, 0, K, stream))
        return rc;
    // the tail launch leaves the cotangent of pose_0 with the last pose sums already in it (the first iteration of the pass is never the tail's)
    const bool folded = tail_from > 0 && (tail_from - (tail_from >= K ? 1 : 0)) > 0;
    if (K % 2) { double* t = gpose; gpose = gtmp; gtmp = t; }
    if (skip && g->live_host)        // where the sweeps ended, for the next call's tail (and the tail's error word)
        if (hipError_t e = hipMemcpyAsync(g->live_host, W + L.live, (size_t)(f->K_cap + 1) * 4, hipMemcpyDeviceToHost, st)) return -(int)e;
    if (int rc = dicp_permute_rows(dtype, W + L.gsrc_s, qo, f->N, f->n, f->n, f->n, 3, 3, g->gsrc, f->n, 3, stream)) return rc;
    if (want_w)
        if (int rc = dicp_permute_rows(dtype, W + L.gw_s, qo, f->N, f->n, f->n, f->n, 1, 1, g->gw, f->n, 1, stream)) return rc;
    if (want_tgt)
        if (int rc = dicp_window_reduce(dtype, W + L.slab, spos_ref, qo, f->tperm, W + L.far, f->src_rows, f->N, f->n, f->m, f->m_pad, cv,
                                        g->gtgt, f->c, 1, stream))
            return rc;
    return dicp_pose_grad_out(dtype, gpose, folded ? nullptr : W + L.partials, folded ? 0 : L.nblk_w, g->gT0, f->N, stream);
}

int dicp_call_backward_plan(int dtype, const dicp_weight_params* prm, const dicp_call* c, int want_tgt, int want_w, dicp_call_backward_layout* L) {
    if (!prm || !c || !L) return DICP_ERR_NULL;
    if (!call_ok(dtype, c)) return dtype != DICP_F32 && dtype != DICP_F64 ? DICP_ERR_DTYPE : DICP_ERR_SHAPE;
    return backward_layout(dtype, prm, c->N, c->n, c->m, c->K, c->w0 != nullptr, want_tgt, want_w, L);
}

int dicp_call_backward(int dtype, const dicp_weight_params* prm, const dicp_call* c, const dicp_call_grads* g, void* stream) {
    if (!prm || !c || !g || !c->workspace || !c->results || !g->workspace || !g->gsrc || !g->gT0 || !c->src || !c->tgt) return DICP_ERR_NULL;
    if (!c->need_grad) return DICP_ERR_ENUM;
    dicp_call_layout F;
    if (int rc = dicp_call_plan(dtype, c, &F)) return rc;
    if (((uintptr_t)c->workspace & 255) != 0) return DICP_ERR_ALIGN;
    dicp_loop_backward_in f;
    memset(&f, 0, sizeof(f));
    f.src = c->src; f.tgt_sorted = at(c, F.tgt_sorted); f.w0 = c->w0; f.tperm = (const int32_t*)at(c, F.tperm);
    f.qorder = (const int32_t*)at(c, F.orders) + (size_t)(F.n_orders - 1) * c->N * c->n;
    f.spos = (const int32_t*)at(c, F.spos); f.poses = at(c, F.poses); f.deltas = res(c, F.deltas); f.areg = (const double*)at(c, F.areg); f.alive = at(c, F.alive);
    f.N = c->N; f.n = c->n; f.m = c->m; f.c = c->c; f.K = c->K; f.K_cap = c->K; f.m_pad = F.m_pad; f.dim = c->dim;
    f.knn_variant = DICP_KNN_SWEEP | ((c->flags & DICP_CALL_NO_SMALL_LOOP) ? (1 << 25) : 0);
    return backward_once(dtype, prm, &f, g, stream);
}

// the same reverse sweep for a forward that was run buffer by buffer (dicp_icp_forward / _plan with every history in one slab): the caller names the buffers
int dicp_loop_backward_plan(int dtype, const dicp_weight_params* prm, const dicp_loop_backward_in* f, int want_tgt, int want_w, dicp_call_backward_layout* L) {
    if (!prm || !f || !L) return DICP_ERR_NULL;
    if (dtype != DICP_F32 && dtype != DICP_F64) return DICP_ERR_DTYPE;
    if (f->N < 1 || f->n < 1 || f->m < 1 || f->K < 1 || f->K_cap < f->K) return DICP_ERR_SHAPE;
    return backward_layout(dtype, prm, f->N, f->n, f->m, f->K_cap, f->w0 != nullptr, want_tgt, want_w, L);
}

int dicp_loop_backward_prepare(int dtype, const dicp_loop_backward_in* f, void* src_s_out, void* w_s_out, int32_t* spos_ref_out, void* stream) {
    if (!f || !f->src || !f->qorder || !f->spos || !src_s_out || (f->w0 && !w_s_out)) return DICP_ERR_NULL;
    if (dtype != DICP_F32 && dtype != DICP_F64) return DICP_ERR_DTYPE;
    if (f->N < 1 || f->n < 1 || f->K < 1 || f->K_cap < f->K) return DICP_ERR_SHAPE;
    const bool by_reference = f->spos_of && f->K - 1 >= f->spos_of_from;
    if (by_reference && !spos_ref_out) return DICP_ERR_NULL;
    return prepare_pass(dtype, f, src_s_out, w_s_out, by_reference ? spos_ref_out : nullptr, stream);
}

int dicp_loop_backward(int dtype, const dicp_weight_params* prm, const dicp_loop_backward_in* f, const dicp_call_grads* g, void* stream) {
    if (!prm || !f || !g || !g->workspace || !g->gsrc || !g->gT0) return DICP_ERR_NULL;
    if (!f->src || !f->tgt_sorted || !f->tperm || !f->qorder || !f->spos || !f->poses || !f->deltas || !f->areg || !f->alive) return DICP_ERR_NULL;
    if (dtype != DICP_F32 && dtype != DICP_F64) return DICP_ERR_DTYPE;
    if (f->N < 1 || f->n < 1 || f->m < 1 || f->K < 1 || f->K_cap < f->K || f->m_pad != dicp_padded_targets(f->m) || (f->dim != 2 && f->dim != 3)) return DICP_ERR_SHAPE;
    return backward_once(dtype, prm, f, g, stream);
}

// ---- ICP.pt2pt_dICP_SVD (ICP.py:533-591) for a dense batch on the sweep path with a constant iteration count, one call per direction
int dicp_kabsch_call_plan(int dtype, const dicp_kabsch_call* c, dicp_kabsch_call_layout* L) {
    if (!c || !L) return DICP_ERR_NULL;
    if (!kabsch_call_ok(dtype, c)) return dtype != DICP_F32 && dtype != DICP_F64 ? DICP_ERR_DTYPE : DICP_ERR_SHAPE;
    const size_t es = esize(dtype), N = c->N, n = c->n, K = c->K;
    memset(L, 0, sizeof(*L));
    L->m_pad = dicp_padded_targets(c->m);
    L->nblk = dicp_accumulate_blocks(c->n);
    const size_t m_pad = L->m_pad;
    Carver w;
    L->costs = w.take(N * K * es);
    L->iterations = w.take(N * es);
    L->counters = w.take(K * 4);
    L->pairs = w.take(DICP_PAIR_SHARDS * 8);
    L->zeroed = w.off;
    L->frame = w.take(N * 12 * es);
    L->keys = w.take(N * m_pad * es);
    L->tperm = w.take(N * m_pad * 4);
    L->bucket = w.take(N * (DICP_CALL_NBKT + 1) * 4);
    L->brange = w.take(N * 2 * es);
    L->tgs4 = w.take(N * m_pad * 4 * es);
    L->scratch_bytes = dicp_sweep_sort_scratch_bytes(dtype, c->N, L->m_pad);
    L->scratch = L->scratch_bytes ? w.take(L->scratch_bytes) : 0;
    L->pose = w.take(N * 12 * es);
    L->pose_search = w.take(N * 12 * es);
    L->pose_used = w.take(N * 12 * es);
    L->partials = w.take(N * L->nblk * DICP_NACC_PAD * es);
    L->save = w.take(N * DICP_KAB_SAVE * 8);
    L->idx = w.take(N * n * 4);
    L->rows_live = w.take(N * 4);
    L->orders = w.take(2 * N * n * 4);
    L->gpose = w.take(N * 12 * es);
    L->gacc = w.take(N * 16 * es);
    L->total = w.off;
    return 0;
}

int dicp_kabsch_call_forward(int dtype, const dicp_kabsch_call* c, void* stream) {
    if (!c || !c->src || !c->tgt || !c->T_start || !c->w0 || !c->workspace || !c->T_out) return DICP_ERR_NULL;
    dicp_kabsch_call_layout L;
    if (int rc = dicp_kabsch_call_plan(dtype, c, &L)) return rc;
    if (((uintptr_t)c->workspace & 255) != 0) return DICP_ERR_ALIGN;
    const size_t N = c->N, n = c->n;
    const int K = c->K;
    hipStream_t st = (hipStream_t)stream;
    if (int rc = zero_fill(c->workspace, L.zeroed, st)) return rc;
    int32_t* order0 = (int32_t*)kat(c, L.orders);
    int32_t* order1 = order0 + N * n;
    // frame, sort, packed rows (the Kabsch sums gather the rows as given: no sorted copy), search pose of the start, first query order
    if (int rc = dicp_sweep_setup(dtype, c->tgt, c->c, nullptr, c->N, c->m, L.m_pad, c->quantum, c->directions, kat(c, L.frame), kat(c, L.keys), (int32_t*)kat(c, L.tperm),
                                  DICP_CALL_NBKT, (int32_t*)kat(c, L.bucket), kat(c, L.brange), L.scratch_bytes ? kat(c, L.scratch) : nullptr, L.scratch_bytes,
                                  kat(c, L.tgs4), nullptr, c->c, c->src, nullptr, c->n, c->T_start, kat(c, L.pose_search), order0, stream))
        return rc;
    if (int rc = dicp_search_pose(dtype, c->T_start, nullptr, c->N, kat(c, L.pose), stream)) return rc;          // [C | r] itself
    (void)hipGetLastError();
    if (dtype == DICP_F32) kabsch_prep_kernel<float><<<(c->N + 255) / 256, 256, 0, st>>>((int32_t*)kat(c, L.rows_live), c->N, c->n);
    else                   kabsch_prep_kernel<double><<<(c->N + 255) / 256, 256, 0, st>>>((int32_t*)kat(c, L.rows_live), c->N, c->n);
    if (int rc = last_launch()) return rc;
    dicp_kabsch_buffers B;
    memset(&B, 0, sizeof(B));
    B.src = c->src; B.tgt = c->tgt; B.w_init = c->w0; B.c = c->c; B.K = K; B.knn_variant = DICP_KNN_SWEEP; B.m_pad = L.m_pad;
    B.tgt4 = kat(c, L.tgs4); B.tperm = (int32_t*)kat(c, L.tperm); B.bucket = (int32_t*)kat(c, L.bucket); B.brange = kat(c, L.brange); B.nbkt = DICP_CALL_NBKT;
    B.pairs = (unsigned long long*)kat(c, L.pairs); B.frame = kat(c, L.frame); B.pose = kat(c, L.pose); B.pose_search = kat(c, L.pose_search);
    B.pose_used = kat(c, L.pose_used); B.idx = (int32_t*)kat(c, L.idx); B.partials = kat(c, L.partials); B.save = (double*)kat(c, L.save);
    B.costs = kat(c, L.costs); B.iterations = kat(c, L.iterations); B.rows_live = (int32_t*)kat(c, L.rows_live); B.counters = (int32_t*)kat(c, L.counters);
    // segments [0,1) [1,2) [2,K): the queries are re-ordered under the pose before iteration 1 (the poses move most in the first step)
    const int cuts[4] = {0, K > 1 ? 1 : K, K > 2 ? 2 : K, K};
    for (int s = 0; s < 3; ++s) {
        const int k0 = cuts[s], k1 = cuts[s + 1];
        if (k1 <= k0) continue;
        if (k0 == 1)
            if (int rc = dicp_query_order(dtype, c->src, kat(c, L.pose_search), kat(c, L.brange), DICP_CALL_NBKT, c->N, c->n, order1, nullptr, nullptr, nullptr, 0, nullptr,
                                          L.m_pad, kat(c, L.keys), (int32_t*)kat(c, L.bucket), c->m, (int32_t*)kat(c, L.rows_live), nullptr, stream))
                return rc;
        B.qorder = k0 == 0 ? order0 : order1;
        if (int rc = dicp_kabsch_forward(dtype, &B, c->N, c->n, c->m, c->trim_on, c->trim_dist, 1, c->tolerance, k0, k1, stream)) return rc;
    }
    (void)hipGetLastError();
    if (dtype == DICP_F32) kabsch_finish_kernel<float><<<(c->N * 16 + 255) / 256, 256, 0, st>>>((const float*)kat(c, L.pose), (float*)kat(c, L.iterations), (float*)c->T_out, c->N, K);
    else                   kabsch_finish_kernel<double><<<(c->N * 16 + 255) / 256, 256, 0, st>>>((const double*)kat(c, L.pose), (double*)kat(c, L.iterations), (double*)c->T_out, c->N, K);
    if (int rc = last_launch()) return rc;
    if (c->pc_out) return dicp_transform_points(dtype, c->src, kat(c, L.pose), c->pc_out, c->N, c->n, stream);     // ICP.py:581
    return 0;
}

int dicp_kabsch_call_backward(int dtype, const dicp_kabsch_call* c, const dicp_kabsch_call_grads* g, void* stream) {
    if (!c || !g || !c->workspace || !c->src || !c->tgt || !c->w0 || !g->gsrc) return DICP_ERR_NULL;
    dicp_kabsch_call_layout L;
    if (int rc = dicp_kabsch_call_plan(dtype, c, &L)) return rc;
    const size_t es = esize(dtype), N = c->N, n = c->n;
    hipStream_t st = (hipStream_t)stream;
    (void)hipGetLastError();
    if (dtype == DICP_F32) kabsch_gpose_kernel<float><<<(c->N * 12 + 255) / 256, 256, 0, st>>>((const float*)g->gT, (float*)kat(c, L.gpose), c->N);
    else                   kabsch_gpose_kernel<double><<<(c->N * 12 + 255) / 256, 256, 0, st>>>((const double*)g->gT, (double*)kat(c, L.gpose), c->N);
    if (int rc = last_launch()) return rc;
    if (int rc = dicp_kabsch_step_bwd(dtype, kat(c, L.gpose), (const double*)kat(c, L.save), kat(c, L.gacc), c->N, stream)) return rc;
    // dicp_kabsch_bwd adds: the three gradients start at zero
    if (int rc = zero_fill(g->gsrc, N * n * 3 * es, st)) return rc;
    if (g->gtgt) if (int rc = zero_fill(g->gtgt, N * (size_t)c->m * c->c * es, st)) return rc;
    if (g->gw) if (int rc = zero_fill(g->gw, N * n * es, st)) return rc;
    return dicp_kabsch_bwd(dtype, c->src, c->tgt, c->c, (const int32_t*)kat(c, L.idx), kat(c, L.pose_used), c->w0, c->trim_on, c->trim_dist, kat(c, L.gacc), nullptr,
                           c->N, c->n, c->m, g->gsrc, g->gtgt, g->gw, stream);
}

}  // extern "C"
