// libdicp_hip.so — hand-written HIP kernels (gfx950 / MI355X) for the differentiable-ICP
// hot path, behind the C ABI declared in include/dicp_hip.h.
//
// Kernels (one section each):
//   pack_kernel            target rows -> [x,y,z,0.5|y|^2] (once per ICP call)
//   sweep_rows / sweep_buckets / query_order / query_keys / loop_init / loop_finish
//                          per-call set-up of the sorted-sweep search and of the loop state (everything after torch.sort)
//   knn_valu_kernel        fused transform + brute-force 1-NN, VALU FMA form, LDS-tiled targets
//   (knn_f16.hip)          the same search on the matrix cores: split-f16 filter (v_mfma_f32_32x32x16_f16) + exact float32 refine
//   knn_sweep_kernel       exact 1-NN with slab pruning over x-sorted targets
//   gather / scatter / permute_add
//                          row-indexed copies (nn.find_nn's gather and its backward; sorted copies and their undoing)
//   accumulate_kernel      residual/weights/Jacobian/normal-equation sums, per-block partials
//   step_kernel            per-cloud reduce + 6x6 solve + pose update + loop bookkeeping (step_body)
//   icp_small_forward / icp_small_backward
//                          small clouds: one block runs a cloud's whole chunk of iterations, forward and reverse
//   accumulate_bwd_kernel  adjoint of accumulate_kernel (recompute from idx and pose), row-coalesced float atomics
//   accumulate_bwd_window  the same adjoint in sorted space (sweep path): per-block LDS windows with per-row lists,
//                          per-block slabs, no float atomics on the common path; window_reduce sums the slabs
//   step_bwd_kernel        adjoint of step_kernel
//   gumbel_* kernels       Gumbel-softmax soft correspondence (online softmax) and its two-pass backward
//   kabsch_* kernels       closed-form SVD point-to-point step and its adjoint
//   transform / loss_weight kernels
//                          pc = C p + r and loss.get_weight for direct users of the classes
//
// Written for 64-wide wavefronts and 8 XCDs: block ids are dealt so that all blocks of
// one cloud land on one XCD (they share that cloud's targets in its L2).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "dicp_common.h"
#include "dicp_internal.h"
#include "dicp_fill.h"

namespace {

#include "kernels_setup.h"
#include "kernels_search.h"
#include "kernels_setup_sort.h"
#include "kernels_rows.h"
#include "kernels_accumulate.h"
#include "kernels_backward.h"
#include "kernels_soft_svd.h"
#include "kernels_host.h"

}  // namespace

// ======================================================================== C ABI
extern "C" {

int dicp_abi_version(void) { return DICP_ABI_VERSION; }
int dicp_copy(void* dst, const void* src, size_t bytes, void* stream) {
    if (!dst || !src) return DICP_ERR_NULL;
    if ((((uintptr_t)dst | (uintptr_t)src | bytes) & 3) != 0) return DICP_ERR_ALIGN;
    return dicp_fill::copy(dst, src, bytes, (hipStream_t)stream);
}
int dicp_zero(void* dst, size_t bytes, void* stream) {
    if (!dst) return DICP_ERR_NULL;
    if ((((uintptr_t)dst | bytes) & 3) != 0) return DICP_ERR_ALIGN;
    return dicp_fill::zero(dst, bytes, (hipStream_t)stream);
}
int dicp_padded_targets(int m) { return m <= 0 ? 0 : ((m + KNN_PAD - 1) / KNN_PAD) * KNN_PAD; }
int dicp_accumulate_blocks(int n) { return n <= 0 ? 0 : (n + ACC_PTS - 1) / ACC_PTS; }
int dicp_search_frame(int dtype, const void* tgt, int c, const int32_t* tgt_rows, int N, int m, double quantum, int directions,
                      const void* src, const int32_t* src_rows, int n, const void* T_init, void* frame, void* stream) {
    if (!tgt || !frame) return DICP_ERR_NULL;
    if (bad_dtype(dtype)) return DICP_ERR_DTYPE;
    if (N <= 0 || m <= 0 || c < 3 || !(quantum >= 0.0) || (src && T_init && n <= 0)) return DICP_ERR_SHAPE;
    if (!src || !T_init) { src = nullptr; T_init = nullptr; }          // (the queries count only with their pose)
    begin_launch();
    if (dtype == DICP_F32) search_frame_kernel<float><<<N, CC_THREADS, 0, (hipStream_t)stream>>>((const float*)tgt, c, m, tgt_rows, quantum, directions, (float*)frame, (const float*)src, n, src_rows, (const float*)T_init);
    else                   search_frame_kernel<double><<<N, CC_THREADS, 0, (hipStream_t)stream>>>((const double*)tgt, c, m, tgt_rows, quantum, directions, (double*)frame, (const double*)src, n, src_rows, (const double*)T_init);
    return launch_status();
}

int dicp_pack_target(int dtype, const void* tgt, int c, const void* frame, const int32_t* tgt_rows, int N, int m, void* tgt4, int m_pad, void* stream) {
    if (!tgt || !tgt4) return DICP_ERR_NULL;
    if (bad_dtype(dtype)) return DICP_ERR_DTYPE;
    if (N <= 0 || m <= 0 || (c != 3 && c != 6) || m_pad != dicp_padded_targets(m)) return DICP_ERR_SHAPE;
    if ((uintptr_t)tgt4 % (dtype == DICP_F32 ? 16 : 32)) return DICP_ERR_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    begin_launch();
    const int bpc = (int)blocks_for((size_t)m_pad);
    const unsigned g = grid_for(N, bpc);
    if (dtype == DICP_F32) pack_kernel<float><<<g, BLOCK, 0, st>>>((const float*)tgt, N, m, c, (float4*)tgt4, m_pad, bpc, (const float*)frame, tgt_rows);
    else                   pack_kernel<double><<<g, BLOCK, 0, st>>>((const double*)tgt, N, m, c, (double4*)tgt4, m_pad, bpc, (const double*)frame, tgt_rows);
    return launch_status();
}

size_t dicp_sweep_sort_scratch_bytes(int dtype, int N, int m_pad) {
    if (N <= 0 || m_pad <= 0 || (dtype == DICP_F32 && m_pad <= RS_MAX)) return 0;     // the one-block LDS sort needs none
    const size_t S = ((size_t)m_pad + GS_CHUNK - 1) / GS_CHUNK, passes = dtype == DICP_F32 ? 4 : 8;      // + the digit counts of the several-blocks-per-cloud form
    return (size_t)N * 2 * m_pad * ((dtype == DICP_F32 ? 4 : 8) + sizeof(int32_t)) + 256 + (S <= GS_MAX_CHUNKS ? passes * (size_t)N * S * S * 256 * sizeof(int32_t) : 0);
}

int dicp_sweep_sort(int dtype, const void* tgt, int c, const void* frame, const int32_t* tgt_rows, int N, int m, int m_pad, void* keys_sorted,
                    int32_t* tperm, int nbkt, int32_t* bucket, void* brange, void* scratch, size_t scratch_bytes, void* stream) {
    if (!tgt || !keys_sorted || !tperm || (bucket && !brange)) return DICP_ERR_NULL;
    if (bucket && nbkt <= 0) return DICP_ERR_SHAPE;
    if (bad_dtype(dtype)) return DICP_ERR_DTYPE;
    if (N <= 0 || m <= 0 || (c != 3 && c != 6) || m_pad != dicp_padded_targets(m)) return DICP_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DICP_F32 && m_pad <= RS_MAX) {                 // in LDS, the bucket table while the sorted keys are there
        begin_launch();
        sort_keys_kernel<<<N, RS_THREADS, 0, st>>>((const float*)tgt, c, N, m, m_pad, (float*)keys_sorted, tperm, nbkt, bucket, (float*)brange, (const float*)frame, tgt_rows);
        return launch_status();
    }
    if (!scratch || scratch_bytes < dicp_sweep_sort_scratch_bytes(dtype, N, m_pad)) return DICP_ERR_NULL;
    begin_launch();
    void* kb = (void*)(((uintptr_t)scratch + 15) & ~(uintptr_t)15);       // keys first (8-byte aligned), then the indices
    const int S = (m_pad + GS_CHUNK - 1) / GS_CHUNK;
    const bool several = S >= 2 && S <= GS_MAX_CHUNKS && (size_t)N * S <= 0x7fffffffu;      // several blocks per cloud, one launch per pass
#define DICP_BIGSORT(T, KT) do { \
        int32_t* ib = (int32_t*)((KT*)kb + (size_t)N * 2 * m_pad); \
        int32_t* hist = ib + (size_t)N * 2 * m_pad; \
        if (several) { \
            constexpr int PASSES = (int)sizeof(KT); \
            sort_big_keys_kernel<T><<<N * S, GS_THREADS, 0, st>>>((const T*)tgt, c, m, m_pad, S, (const T*)frame, tgt_rows, (KT*)kb, ib, hist, N); \
            for (int p = 0; p + 1 < PASSES; ++p) sort_big_pass_kernel<T, false><<<N * S, GS_THREADS, 0, st>>>(p, m_pad, S, (KT*)kb, ib, hist, N, (T*)keys_sorted, tperm); \
            sort_big_pass_kernel<T, true><<<N * S, GS_THREADS, 0, st>>>(PASSES - 1, m_pad, S, (KT*)kb, ib, hist, N, (T*)keys_sorted, tperm); \
        } else sort_keys_big_kernel<T><<<N, GS_THREADS, 0, st>>>((const T*)tgt, c, m, m_pad, (const T*)frame, tgt_rows, (T*)keys_sorted, tperm, (KT*)kb, ib); \
        if (bucket) sweep_buckets_kernel<T><<<N, BLOCK, 0, st>>>((const T*)keys_sorted, N, m, m_pad, nbkt, bucket, (T*)brange, tgt_rows); } while (0)
    if (dtype == DICP_F32) DICP_BIGSORT(float, unsigned); else DICP_BIGSORT(double, unsigned long long);
#undef DICP_BIGSORT
    return launch_status();
}

int dicp_sweep_build(int dtype, const void* tgt, int c, const void* frame, const int32_t* tgt_rows, const int32_t* tperm, int N, int m, int m_pad,
                     void* tgs4, void* tgt_s, int tgt_s_stride, void* stream) {
    // frame != NULL: the packed rows tgs4 are y - centre; tgt_s stays as given
    if (!tgt || !tgs4 || !tperm) return DICP_ERR_NULL;
    if (bad_dtype(dtype)) return DICP_ERR_DTYPE;
    if (N <= 0 || m <= 0 || (c != 3 && c != 6) || m_pad != dicp_padded_targets(m) || (tgt_s && tgt_s_stride < c)) return DICP_ERR_SHAPE;
    if ((uintptr_t)tgs4 % (dtype == DICP_F32 ? 16 : 32)) return DICP_ERR_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    begin_launch();
    const int bpc = (m_pad + BLOCK * 4 - 1) / (BLOCK * 4);
    const unsigned g = grid_for(N, bpc);
    if (dtype == DICP_F32) sweep_rows_kernel<float><<<g, BLOCK, 0, st>>>((const float*)tgt, tgt_rows, N, m, c, m_pad, bpc, (float4*)tgs4, tperm, (float*)tgt_s, tgt_s_stride, (const float*)frame);
    else                   sweep_rows_kernel<double><<<g, BLOCK, 0, st>>>((const double*)tgt, tgt_rows, N, m, c, m_pad, bpc, (double4*)tgs4, tperm, (double*)tgt_s, tgt_s_stride, (const double*)frame);
    return launch_status();
}

// The whole per-call set-up of the sweep path behind one call: search frame -> key sort -> sorted rows -> (given T_init and the queries)
// the search pose of iteration 0 and the first query order.  Five launches the host used to make one by one, with their glue, while the
// GPU sat idle at the start of a call (profiles/r03_timed_call_timeline.txt: 73 us before the first search).
int dicp_sweep_setup(int dtype, const void* tgt, int c, const int32_t* tgt_rows, int N, int m, int m_pad, double quantum, int directions,
                     void* frame, void* keys_sorted, int32_t* tperm, int nbkt, int32_t* bucket, void* brange, void* scratch, size_t scratch_bytes,
                     void* tgs4, void* tgt_s, int tgt_s_stride,
                     const void* src, const int32_t* src_rows, int n, const void* T_init, void* pose_search0, int32_t* qorder0, void* stream) {
    if (!frame) return DICP_ERR_NULL;
    int rc = dicp_search_frame(dtype, tgt, c, tgt_rows, N, m, quantum, directions, src, src_rows, n, T_init, frame, stream);
    if (!rc) rc = dicp_sweep_sort(dtype, tgt, c, frame, tgt_rows, N, m, m_pad, keys_sorted, tperm, nbkt, bucket, brange, scratch, scratch_bytes, stream);
    if (!rc) rc = dicp_sweep_build(dtype, tgt, c, frame, tgt_rows, tperm, N, m, m_pad, tgs4, tgt_s, tgt_s_stride, stream);
    if (!rc && T_init && pose_search0 && qorder0) {
        if (!src || n <= 0) return DICP_ERR_NULL;
        rc = dicp_search_pose(dtype, T_init, frame, N, pose_search0, stream);
        if (!rc) rc = dicp_query_order(dtype, src, pose_search0, brange, nbkt, N, n, qorder0, nullptr, nullptr, nullptr, 0, nullptr, m_pad, keys_sorted, bucket, m,
                                       src_rows, tgt_rows, stream);
    }
    return rc;
}

int dicp_query_keys(int dtype, const void* src, const void* pose, int N, int n, void* keys, void* stream) {
    if (!src || !keys) return DICP_ERR_NULL;
    if (bad_dtype(dtype)) return DICP_ERR_DTYPE;
    if (N <= 0 || n <= 0) return DICP_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    begin_launch();
    const int bpc = (int)blocks_for((size_t)n);
    const unsigned g = grid_for(N, bpc);
    if (dtype == DICP_F32) query_keys_kernel<float><<<g, BLOCK, 0, st>>>((const float*)src, (const float*)pose, N, n, bpc, (float*)keys);
    else                   query_keys_kernel<double><<<g, BLOCK, 0, st>>>((const double*)src, (const double*)pose, N, n, bpc, (double*)keys);
    return launch_status();
}

static int query_order_go(int dtype, const void* src, const void* pose, const void* brange, int nbkt, int N, int n, int32_t* qorder,
                          const void* w, void* src_s, void* w_s, int reproducible, const int32_t* spos_prev, int m_pad,
                          const void* skeys, const int32_t* bucket, int m, const int32_t* src_rows, const int32_t* tgt_rows, const void* pose_prev, const int32_t* order_prev, void* stream);
int dicp_query_order(int dtype, const void* src, const void* pose, const void* brange, int nbkt, int N, int n, int32_t* qorder,
                     const void* w, void* src_s, void* w_s, int reproducible, const int32_t* spos_prev, int m_pad,
                     const void* skeys, const int32_t* bucket, int m, const int32_t* src_rows, const int32_t* tgt_rows, void* stream) {
    return query_order_go(dtype, src, pose, brange, nbkt, N, n, qorder, w, src_s, w_s, reproducible, spos_prev, m_pad, skeys, bucket, m, src_rows, tgt_rows, nullptr, nullptr, stream);
}
int dicp_query_reorder(int dtype, const void* src, const void* pose, const void* pose_prev, const int32_t* order_prev, const void* brange, int nbkt, int N, int n,
                       int32_t* qorder, int m_pad, const void* skeys, const int32_t* bucket, int m, const int32_t* src_rows, const int32_t* tgt_rows, void* stream) {
    if (!pose || !pose_prev || !order_prev || order_prev == qorder) return DICP_ERR_NULL;
    return query_order_go(dtype, src, pose, brange, nbkt, N, n, qorder, nullptr, nullptr, nullptr, 0, nullptr, m_pad, skeys, bucket, m, src_rows, tgt_rows, pose_prev, order_prev, stream);
}
static int query_order_go(int dtype, const void* src, const void* pose, const void* brange, int nbkt, int N, int n, int32_t* qorder,
                          const void* w, void* src_s, void* w_s, int reproducible, const int32_t* spos_prev, int m_pad,
                          const void* skeys, const int32_t* bucket, int m, const int32_t* src_rows, const int32_t* tgt_rows, const void* pose_prev, const int32_t* order_prev, void* stream) {
    if (!src || !brange || !qorder || (w_s && !w)) return DICP_ERR_NULL;
    if (bad_dtype(dtype)) return DICP_ERR_DTYPE;
    if (N <= 0 || n <= 0 || nbkt <= 0 || ((spos_prev || skeys) && m_pad <= 0) || (skeys && (m <= 0 || m > m_pad))) return DICP_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    begin_launch();
#define DICP_QO(T, S) query_order_kernel<T, S><<<N, QO_THREADS, 0, st>>>((const T*)src, (const T*)pose, (const T*)brange, nbkt, N, n, qorder, \
        (const T*)w, (T*)src_s, (T*)w_s, reproducible, spos_prev, m_pad, (const T*)skeys, 1, m, bucket, src_rows, tgt_rows, (const T*)pose_prev, order_prev)
    if (dtype == DICP_F32) { if (n <= 16384) DICP_QO(float, 16384); else DICP_QO(float, 65536); }
    else                   { if (n <= 16384) DICP_QO(double, 16384); else DICP_QO(double, 65536); }
#undef DICP_QO
    return launch_status();
}

int dicp_loop_init(int dtype, const void* T_init, const void* w0, double thresh, int rows, int N, int n,
                   void* pose0, void* alive0, void* n_start, const void* frame, void* pose_search0,
                   const void* src, void* rmax, void* dcum, int dcum_stride, void* stream) {
    if (!T_init || !pose0 || !alive0 || !n_start || (rmax && (!src || !dcum))) return DICP_ERR_NULL;       // (w0 == NULL: unit weights)
    if (bad_dtype(dtype)) return DICP_ERR_DTYPE;
    if (N <= 0 || n <= 0 || (rows != 1 && rows != 3) || (rmax && dcum_stride < 2)) return DICP_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    begin_launch();
    if (dtype == DICP_F32) loop_init_kernel<float><<<N, LI_THREADS, 0, st>>>((const float*)T_init, (const float*)w0, (float)thresh, rows, n, (float*)pose0, (float*)alive0, (float*)n_start, (const float*)frame, (float*)pose_search0, (const float*)src, (float*)rmax, (float*)dcum, dcum_stride);
    else                   loop_init_kernel<double><<<N, LI_THREADS, 0, st>>>((const double*)T_init, (const double*)w0, thresh, rows, n, (double*)pose0, (double*)alive0, (double*)n_start, (const double*)frame, (double*)pose_search0, (const double*)src, (double*)rmax, (double*)dcum, dcum_stride);
    return launch_status();
}

int dicp_search_pose(int dtype, const void* T_init, const void* frame, int N, void* pose_search, void* stream) {
    if (!T_init || !pose_search) return DICP_ERR_NULL;
    if (bad_dtype(dtype)) return DICP_ERR_DTYPE;
    if (N <= 0) return DICP_ERR_SHAPE;
    begin_launch();
    const unsigned g = (unsigned)((N * 12 + BLOCK - 1) / BLOCK);
    if (dtype == DICP_F32) search_pose_kernel<float><<<g, BLOCK, 0, (hipStream_t)stream>>>((const float*)T_init, (const float*)frame, N, (float*)pose_search);
    else                   search_pose_kernel<double><<<g, BLOCK, 0, (hipStream_t)stream>>>((const double*)T_init, (const double*)frame, N, (double*)pose_search);
    return launch_status();
}

int dicp_loop_finish(int dtype, const void* pose_K, const void* alive_K, const void* n_start, const void* n_matched, int K, int N,
                     void* iterations, void* matched_ratio, void* T_out, void* stream) {
    if (!pose_K || !alive_K || !n_start || !n_matched || !iterations || !matched_ratio || !T_out) return DICP_ERR_NULL;
    if (bad_dtype(dtype)) return DICP_ERR_DTYPE;
    if (N <= 0 || K < 0) return DICP_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    begin_launch();
    const unsigned g = blocks_for((size_t)N);
    if (dtype == DICP_F32) loop_finish_kernel<float><<<g, BLOCK, 0, st>>>((const float*)pose_K, (const float*)alive_K, (const float*)n_start, (const float*)n_matched, K, N, (float*)iterations, (float*)matched_ratio, (float*)T_out);
    else                   loop_finish_kernel<double><<<g, BLOCK, 0, st>>>((const double*)pose_K, (const double*)alive_K, (const double*)n_start, (const double*)n_matched, K, N, (double*)iterations, (double*)matched_ratio, (double*)T_out);
    return launch_status();
}

int dicp_pose_grad_in(int dtype, const void* gT, double* gpose, int N, void* stream) {
    if (!gpose) return DICP_ERR_NULL;
    if (bad_dtype(dtype)) return DICP_ERR_DTYPE;
    if (N <= 0) return DICP_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    begin_launch();
    const unsigned g = blocks_for((size_t)N * 12);
    if (dtype == DICP_F32) pose_grad_in_kernel<float><<<g, BLOCK, 0, st>>>((const float*)gT, gpose, N);
    else                   pose_grad_in_kernel<double><<<g, BLOCK, 0, st>>>((const double*)gT, gpose, N);
    return launch_status();
}

int dicp_pose_grad_out(int dtype, const double* gpose, const void* bwd_partials, int nblk, void* gT0, int N, void* stream) {
    if (!gpose || !gT0) return DICP_ERR_NULL;
    if (bad_dtype(dtype)) return DICP_ERR_DTYPE;
    if (N <= 0 || (bwd_partials && nblk <= 0)) return DICP_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    begin_launch();
    const unsigned g = blocks_for((size_t)N * 16);
    if (dtype == DICP_F32) pose_grad_out_kernel<float><<<g, BLOCK, 0, st>>>(gpose, (const float*)bwd_partials, nblk, (float*)gT0, N);
    else                   pose_grad_out_kernel<double><<<g, BLOCK, 0, st>>>(gpose, (const double*)bwd_partials, nblk, (double*)gT0, N);
    return launch_status();
}

int dicp_knn_f16_probe(const void* src, const void* pose, const void* tgt4, const void* f16_image, const int32_t* src_rows, const int32_t* tgt_rows,
                       int N, int n, int m, int m_pad, float* out, void* stream) {
    return dicp_tu::knn_f16_probe(src, pose, tgt4, f16_image, src_rows, tgt_rows, N, n, m, m_pad, out, stream);
}
size_t dicp_knn_f16_bytes(int N, int m_pad) { return (N <= 0 || m_pad <= 0) ? 0 : dicp_tu::knn_f16_image_bytes(N, m_pad); }
int dicp_knn_f16_pack(const void* tgt4, const int32_t* tgt_rows, int N, int m, int m_pad, void* image, void* stream) {
    return dicp_tu::knn_f16_pack(tgt4, tgt_rows, N, m, m_pad, image, stream);
}

int dicp_knn(int dtype, const void* src, const void* pose, const void* tgt4, const int32_t* src_rows, const int32_t* tgt_rows,
             int N, int n, int m, int m_pad, int32_t* idx, int variant, const void* f16_image, void* stream) {
    if (!src || !tgt4 || !idx) return DICP_ERR_NULL;
    if (bad_dtype(dtype)) return DICP_ERR_DTYPE;
    if (N <= 0 || n <= 0 || m <= 0 || m_pad != dicp_padded_targets(m)) return DICP_ERR_SHAPE;
    const int kind = variant & 0xff, cfg = (variant >> 8) & 0xff;      // cfg != 0: fixed launch config (tuning)
    if (kind < DICP_KNN_AUTO || kind > DICP_KNN_MFMA || (variant >> 16)) return DICP_ERR_ENUM;
    if (kind == DICP_KNN_MFMA && dtype != DICP_F32) return DICP_ERR_DTYPE;
    if ((uintptr_t)tgt4 % (dtype == DICP_F32 ? 16 : 32)) return DICP_ERR_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    begin_launch();
    const Rows rw{src_rows, tgt_rows};
    if (kind == DICP_KNN_MFMA) {      // (the launch configuration bits are not used by this form)
        if (!f16_image) return DICP_ERR_NULL;
        return dicp_tu::knn_f16_brute(src, pose, tgt4, const_cast<void*>(f16_image), src_rows, tgt_rows, N, n, m, m_pad, idx, stream);
    }
    if (dtype == DICP_F32) return knn_valu_launch<float>(cfg, src, pose, tgt4, N, n, m, m_pad, idx, rw, st);
    return knn_valu_launch<double>(cfg, src, pose, tgt4, N, n, m, m_pad, idx, rw, st);
}

// Tile-sweep launch configurations (queries per lane, rows per chunk): 0 = chosen from the problem size,
// 1 = (1, 8), 2 = (2, 8) [the big-problem default], 4 = (1, 16) [float32: the small-problem default; float64: (1, 8)]
static int sweep_queries_per_lane(int cfg) { return cfg == 2 ? 2 : ((cfg == 1 || cfg == 4) ? 1 : 0); }
static int sweep_auto_cfg(int N, int n) { return ((long)N * n >= 2L * BLOCK * 1024) ? SWEEP_CFG_BIG : 4; }

// entries per guard work list (one list per XCD, clouds dealt by cloud & 7): a cloud appends at most one entry per unit of the sweep and one per 64 of its
// candidate sets, 2 ceil(n/64) in all; N ceil(n/64) holds that for every N >= 2, a single cloud needs twice its own (round 5's advice: entries past the
// capacity were dropped silently, and with them those sets' re-scoring)
static int guard_list_cap(int N, int n) { return (N > 2 ? N : 2) * ((n + WAVE - 1) / WAVE); }

struct CertArgs {             // certifying search: budgets (NULL q: plain search), motion bounds; guard: the launch of a certified iteration
    void* q; void* qu; const void* dcum; int dstride; int k; int32_t* count; bool guard; int32_t* cloud; void* set;
    int32_t* cm; int32_t* pend; int32_t* gdirty;     // the searches' own copy of the matches (by slot); guard launches: where a CHANGED match is left (SweepCert::pend)
    int32_t* slist; int32_t* scount;                 // the clouds' candidate-set lists (SweepCert::slist)
    const int32_t* glist; const int32_t* gcount; int glist_cap;     // guard launches: the work lists the previous step made
};

struct FormArgs { const int32_t* in; int32_t* out; int dflt; };       // per-cloud slab tallies of the loop's plain searches (dicp_loop_buffers.search.form)
static int sweep_launch(int dtype, const void* src, const void* pose, const void* tgs4, const int32_t* tperm,
                        const int32_t* qorder, const int32_t* bucket, const void* brange, int nbkt,
                        int N, int n, int m, int m_pad, int32_t* idx, int32_t* spos, unsigned long long* pairs, int cfg, Rows rw, hipStream_t st,
                        CertArgs ca = CertArgs{}, const void* f16_image = nullptr, FormArgs fa = FormArgs{nullptr, nullptr, 0}) {
    hipEvent_t ev0, ev1;
    take_launch_events(ev0, ev1);                                       // (null unless a timed loop set them for this launch)
    if (cfg == 0) cfg = sweep_auto_cfg(N, n);
    // plain float32 searches in units of 128 queries, given the image of the sorted rows: the scoring runs on the matrix cores (knn_f16.hip) -- for every
    // cloud, or (given the previous plain search's tally) for the clouds whose slabs were long, the others staying with the vector form: two launches, each
    // of which leaves the other's clouds at once
    const bool f16_ok = f16_image && dtype == DICP_F32 && !ca.q && cfg == SWEEP_CFG_BIG;
    const bool hybrid = f16_ok && fa.in;
    if (f16_ok && !hybrid)
        return dicp_tu::knn_f16_sweep(src, pose, tgs4, const_cast<void*>(f16_image), tperm, qorder, bucket, brange, nbkt, rw.src, rw.tgt, N, n, m, m_pad, idx, spos,
                                      pairs, nullptr, fa.out, FORM_TILES, 1, ev0, ev1, st);
    hipEvent_t ev_stop = ev1;
    if (hybrid) ev1 = nullptr;                                          // (the pair brackets both launches)
    const int Q = sweep_queries_per_lane(cfg);
    if (Q <= 0) return DICP_ERR_ENUM;
    const int units = (n + WAVE * Q - 1) / (WAVE * Q);                  // waves per cloud
    const int bpc = (units + BLOCK / WAVE - 1) / (BLOCK / WAVE);
#define DICP_SWEEP_ARGS(T) (const T*)src, (const T*)pose, (const typename V4<T>::type*)tgs4, tperm, qorder, bucket, (const T*)brange, nbkt, idx, spos, pairs, \
        N, n, m, m_pad, bpc, rw.src, rw.tgt
#define DICP_SWEEP_C(T, Q, CH, CERT, CT) hipExtLaunchKernelGGL((knn_sweep_kernel<T, Q, CH, CERT>), dim3(grid_for(N, bpc)), dim3(BLOCK), 0, st, ev0, ev1, 0, DICP_SWEEP_ARGS(T), CT)
#define DICP_SWEEP_L(T, Q, CH, CT) do { /* the guard: a grid the GPU holds at once (5 blocks per unit), or as many blocks as there are units */ \
        const long want = ((long)N * units + BLOCK / WAVE - 1) / (BLOCK / WAVE); const unsigned g8 = (unsigned)((want < 1280 ? want : 1280) + 7) / 8 * 8; \
        hipExtLaunchKernelGGL((knn_sweep_guard_kernel<T, Q, CH>), dim3(g8), dim3(BLOCK), 0, st, ev0, ev1, 0, DICP_SWEEP_ARGS(T), CT, ca.glist, ca.gcount, ca.glist_cap); } while (0)
#define DICP_SWEEP_CG(T, Q, CH, CT) do { if (ca.guard) DICP_SWEEP_L(T, Q, CH, CT); else DICP_SWEEP_C(T, Q, CH, true, CT); } while (0)
#define DICP_SWEEP(T, Q, CH) do { SweepCert<T> none{}; none.form_in = hybrid ? fa.in : nullptr; none.form_out = fa.out; none.form_mine = 0; none.form_default = fa.dflt; DICP_SWEEP_C(T, Q, CH, false, none); } while (0)
    if (ca.q) {             // certifying search, or the guard launch of a certified iteration
        if (!ca.qu || !ca.dcum || !spos || !qorder || (ca.guard && (!ca.glist || !ca.gcount))) return DICP_ERR_ENUM;
        if (dtype == DICP_F32) {
            SweepCert<float> c{(float*)ca.q, (float*)ca.qu, (const float*)ca.dcum, ca.dstride, ca.k, ca.count, ca.set, ca.cloud, ca.cm, ca.guard ? ca.pend : nullptr, ca.gdirty, (n + WAVE - 1) / WAVE, ca.slist, ca.scount};
            if (cfg == 2) DICP_SWEEP_CG(float, 2, 8, c); else if (cfg == 4) DICP_SWEEP_CG(float, 1, 16, c); else DICP_SWEEP_CG(float, 1, 8, c);
        } else {
            SweepCert<double> c{(double*)ca.q, (double*)ca.qu, (const double*)ca.dcum, ca.dstride, ca.k, ca.count, ca.set, ca.cloud, ca.cm, ca.guard ? ca.pend : nullptr, ca.gdirty, (n + WAVE - 1) / WAVE, ca.slist, ca.scount};
            if (cfg == 2) DICP_SWEEP_CG(double, 2, 8, c); else DICP_SWEEP_CG(double, 1, 8, c);
        }
        return launch_status();
    }
    if (dtype == DICP_F32) {
        if (cfg == 2) DICP_SWEEP(float, 2, 8); else if (cfg == 4) DICP_SWEEP(float, 1, 16); else DICP_SWEEP(float, 1, 8);
    } else {
        if (cfg == 2) DICP_SWEEP(double, 2, 8); else DICP_SWEEP(double, 1, 8);
    }
    if (hybrid) {
        if (const int rc = launch_status()) return rc;
        return dicp_tu::knn_f16_sweep(src, pose, tgs4, const_cast<void*>(f16_image), tperm, qorder, bucket, brange, nbkt, rw.src, rw.tgt, N, n, m, m_pad, idx, spos,
                                      pairs, fa.in, fa.out, FORM_TILES, fa.dflt, nullptr, ev_stop, st);
    }
#undef DICP_SWEEP
#undef DICP_SWEEP_C
#undef DICP_SWEEP_L
#undef DICP_SWEEP_CG
#undef DICP_SWEEP_ARGS
    return launch_status();
}

int dicp_knn_sweep(int dtype, const void* src, const void* pose, const void* tgs4, const int32_t* tperm,
                   const int32_t* qorder, const int32_t* bucket, const void* brange, int nbkt, const int32_t* src_rows, const int32_t* tgt_rows,
                   int N, int n, int m, int m_pad, int32_t* idx, int32_t* spos, unsigned long long* pairs, int cfg, const void* f16_image,
                   const int32_t* form_in, int32_t* form_out, int form_default, void* stream) {
    if (!src || !tgs4 || !tperm || !bucket || !brange || (!idx && !spos)) return DICP_ERR_NULL;
    if (bad_dtype(dtype)) return DICP_ERR_DTYPE;
    if (N <= 0 || n <= 0 || m <= 0 || nbkt <= 0 || m_pad != dicp_padded_targets(m)) return DICP_ERR_SHAPE;
    if ((uintptr_t)tgs4 % (dtype == DICP_F32 ? 16 : 32)) return DICP_ERR_ALIGN;
    begin_launch();
    return sweep_launch(dtype, src, pose, tgs4, tperm, qorder, bucket, brange, nbkt, N, n, m, m_pad, idx, spos, pairs, cfg, Rows{src_rows, tgt_rows}, (hipStream_t)stream,
                        CertArgs{}, f16_image, FormArgs{form_in, form_out, form_default});
}

int dicp_gather_rows(int dtype, const void* tgt, const int32_t* idx, int N, int n, int m, int c, void* out, void* stream) {
    if (!tgt || !idx || !out) return DICP_ERR_NULL;
    if (bad_dtype(dtype)) return DICP_ERR_DTYPE;
    if (N <= 0 || n <= 0 || m <= 0 || c <= 0) return DICP_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    begin_launch();
    if ((size_t)n * c > 0x7fffffffu) return DICP_ERR_SHAPE;
    const int bpc = (int)(((size_t)n * c + BLOCK * ROWS_U - 1) / (BLOCK * ROWS_U));
    const unsigned g = grid_for(N, bpc);
#define DICP_ROWS(K, T, C) K<T, C><<<g, BLOCK, 0, st>>>((const T*)tgt, idx, N, n, m, c, bpc, (T*)out)
#define DICP_ROWS_C(K, T) do { if (c == 1) DICP_ROWS(K, T, 1); else if (c == 3) DICP_ROWS(K, T, 3); else if (c == 6) DICP_ROWS(K, T, 6); \
        else DICP_ROWS(K, T, 0); } while (0)
    if (dtype == DICP_F32) DICP_ROWS_C(gather_kernel, float); else DICP_ROWS_C(gather_kernel, double);
#undef DICP_ROWS
    return launch_status();
}

int dicp_pack_list(int dtype, const void* const* ptrs, const int32_t* lens, const int32_t* strides, int N, int n_max, int cols, void* out, const void* pad, void* stream) {
    if (!ptrs || !lens || !strides || !out) return DICP_ERR_NULL;
    if (bad_dtype(dtype)) return DICP_ERR_DTYPE;
    if (N <= 0 || n_max <= 0 || cols <= 0) return DICP_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    begin_launch();
    const int bpc = (int)blocks_for((size_t)n_max * cols);
    if (dtype == DICP_F32) pack_list_kernel<float><<<grid_for(N, bpc), BLOCK, 0, st>>>((const float* const*)ptrs, lens, strides, N, n_max, cols, bpc, (float*)out, (const float*)pad);
    else                   pack_list_kernel<double><<<grid_for(N, bpc), BLOCK, 0, st>>>((const double* const*)ptrs, lens, strides, N, n_max, cols, bpc, (double*)out, (const double*)pad);
    return launch_status();
}
int dicp_unpack_list(int dtype, const void* gout, void* const* ptrs, const int32_t* lens, const int32_t* strides, int N, int n_max, int cols, int stride_max, void* stream) {
    if (!gout || !ptrs || !lens || !strides) return DICP_ERR_NULL;
    if (bad_dtype(dtype)) return DICP_ERR_DTYPE;
    if (N <= 0 || n_max <= 0 || cols <= 0 || stride_max < cols) return DICP_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    begin_launch();
    const int bpc = (int)blocks_for((size_t)n_max * stride_max);
    if (dtype == DICP_F32) unpack_list_kernel<float><<<grid_for(N, bpc), BLOCK, 0, st>>>((const float*)gout, (float* const*)ptrs, lens, strides, N, n_max, cols, bpc);
    else                   unpack_list_kernel<double><<<grid_for(N, bpc), BLOCK, 0, st>>>((const double*)gout, (double* const*)ptrs, lens, strides, N, n_max, cols, bpc);
    return launch_status();
}
int dicp_scatter_add_rows(int dtype, const void* gout, const int32_t* idx, int N, int n, int m, int c, void* gtgt, void* stream) {
    if (!gout || !idx || !gtgt) return DICP_ERR_NULL;
    if (bad_dtype(dtype)) return DICP_ERR_DTYPE;
    if (N <= 0 || n <= 0 || m <= 0 || c <= 0) return DICP_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    begin_launch();
    if ((size_t)n * c > 0x7fffffffu) return DICP_ERR_SHAPE;
    const int bpc = (int)blocks_for((size_t)n * c);
    const unsigned g = grid_for(N, bpc);
#define DICP_ROWS(K, T, C) K<T, C><<<g, BLOCK, 0, st>>>((const T*)gout, idx, N, n, m, c, bpc, (T*)gtgt)
    if (dtype == DICP_F32) DICP_ROWS_C(scatter_add_kernel, float); else DICP_ROWS_C(scatter_add_kernel, double);
#undef DICP_ROWS
#undef DICP_ROWS_C
    return launch_status();
}

static int check_params(const dicp_weight_params* p, int c) {
    if (!p) return DICP_ERR_NULL;
    if (p->mode != DICP_PT2PT && p->mode != DICP_PT2PL) return DICP_ERR_ENUM;
    if (p->loss < DICP_LOSS_NONE || p->loss > DICP_LOSS_TRIM) return DICP_ERR_ENUM;
    // c = elements per target row: 3 or 6 (normals in 3:6), or 4 / 8 for rows padded to 16 / 32 bytes (the sorted copies: one sector per gathered row)
    if (p->mode == DICP_PT2PL ? (c != 6 && c != 8) : (c != 3 && c != 4 && c != 6 && c != 8)) return DICP_ERR_SHAPE;   // ICP.py:103
    return 0;
}

// (ca: the certified iterations of the sweep loop -- idx is then ca->spos)
static int accumulate_go(int dtype, const dicp_weight_params* prm, const void* src, const void* tgt, int c,
                         const int32_t* idx, const void* pose, const void* w_init, const void* alive, const int32_t* src_rows,
                         int N, int n, int m, void* partials, void* w_out, int64_t w_stride, void* stream, const CertAcc* ca, const void* w_prev = nullptr) {
    if (const int e = check_params(prm, c)) return e;
    if (!src || !tgt || !partials) return DICP_ERR_NULL;       // (w_init == NULL: unit weights)
    if (bad_dtype(dtype)) return DICP_ERR_DTYPE;
    if (N <= 0 || n <= 0 || m <= 0 || (w_out && w_stride < n) || (!idx && !ca && m != n)) return DICP_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    begin_launch();
    const WeightParams P = to_params(prm);
    const int bpc = dicp_accumulate_blocks(n);
    const unsigned g = grid_for(N, bpc);
    hipEvent_t ev0, ev1;
    take_launch_events(ev0, ev1);                                       // (null unless a timed loop set them for this launch)
#define DICP_ACC(T, M, CERT, PS) hipExtLaunchKernelGGL((accumulate_kernel<T, M, CERT>), dim3(g), dim3(BLOCK), 0, st, ev0, ev1, 0, P, (const T*)src, (const T*)tgt, c, idx, (const T*)pose, \
        (const T*)w_init, (const T*)alive, N, n, m, bpc, (T*)partials, (T*)w_out, (long)w_stride, src_rows, PS, (const T*)w_prev)
#define DICP_ACC_T(T) do { \
        AccCert<T> ps{}; \
        if (ca) { \
            ps.spos = ca->spos; ps.hist = ca->hist; ps.hist_prev = ca->hist_prev; ps.of = ca->of; ps.k_floor = ca->k_floor; ps.N = N; ps.nwr = (n + WAVE - 1) / WAVE; ps.k = ca->k; \
            ps.nbr = (T*)ca->nbr; ps.gdirty = ca->gdirty; ps.pend = ca->pend; ps.cloud = ca->cloud; ps.fresh = ca->fresh; ps.units = ca->units; ps.sets = ca->sets; ps.scount = ca->scount; \
            if (P.mode == MODE_PT2PL) DICP_ACC(T, MODE_PT2PL, true, ps); else DICP_ACC(T, MODE_PT2PT, true, ps); \
        } else { if (P.mode == MODE_PT2PL) DICP_ACC(T, MODE_PT2PL, false, ps); else DICP_ACC(T, MODE_PT2PT, false, ps); } } while (0)
    if (dtype == DICP_F32) DICP_ACC_T(float); else DICP_ACC_T(double);
#undef DICP_ACC_T
#undef DICP_ACC
    return launch_status();
}

int dicp_accumulate(int dtype, const dicp_weight_params* prm, const void* src, const void* tgt, int c,
                    const int32_t* idx, const void* pose, const void* w_init, const void* alive, const int32_t* src_rows,
                    int N, int n, int m, void* partials, void* w_out, int64_t w_stride, void* stream) {
    return accumulate_go(dtype, prm, src, tgt, c, idx, pose, w_init, alive, src_rows, N, n, m, partials, w_out, w_stride, stream, nullptr);
}

int dicp_step(int dtype, const dicp_step_io* io, int N, void* stream) {
    if (!io || !io->partials || !io->pose_in || !io->pose_out || !io->delta || !io->cost || !io->alive || !io->alive_out ||
        !io->converged || !io->iterations || !io->matched_ratio || !io->n_start) return DICP_ERR_NULL;
    if (bad_dtype(dtype)) return DICP_ERR_DTYPE;
    if (N <= 0 || io->nblk <= 0 || (io->dim != 2 && io->dim != 3) || io->delta_stride < 6 || io->cost_stride < 1 ||
        (io->w_cur && io->w_stride < io->n)) return DICP_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    begin_launch();
    if (dtype == DICP_F32) step_kernel<float><<<N, WAVE, 0, st>>>(*io, N);
    else                   step_kernel<double><<<N, WAVE, 0, st>>>(*io, N);
    return launch_status();
}

struct SkipHost { int32_t* skip; double* mref; const void* alive_k; int32_t* live_k; double eps; int k; };     // SkipArgs, untyped (all NULL / 0: off)
static int step_bwd_go(int dtype, const double* gpose_in, const void* bwd_partials, int nblk, int dim,
                       const void* pose_k, const void* delta_k, int64_t delta_stride, const double* areg_k,
                       void* gs, void* gb, double* gpose_out, int N, void* stream, const SkipHost& sh) {
    if (!gpose_in || !pose_k || !delta_k || !areg_k || !gs || !gb || !gpose_out || (sh.skip && !sh.mref)) return DICP_ERR_NULL;
    if (bad_dtype(dtype)) return DICP_ERR_DTYPE;
    if (N <= 0 || (dim != 2 && dim != 3) || delta_stride < 6 || (bwd_partials && nblk <= 0)) return DICP_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    begin_launch();
    if (dtype == DICP_F32)
        step_bwd_kernel<float><<<N, WAVE, 0, st>>>(gpose_in, (const float*)bwd_partials, nblk, dim, (const float*)pose_k,
                                                  (const float*)delta_k, (long)delta_stride, areg_k, (float*)gs, (float*)gb, gpose_out, N,
                                                  SkipArgs<float>{sh.skip, sh.mref, (const float*)sh.alive_k, sh.live_k, sh.eps, sh.k});
    else
        step_bwd_kernel<double><<<N, WAVE, 0, st>>>(gpose_in, (const double*)bwd_partials, nblk, dim, (const double*)pose_k,
                                                   (const double*)delta_k, (long)delta_stride, areg_k, (double*)gs, (double*)gb, gpose_out, N,
                                                   SkipArgs<double>{sh.skip, sh.mref, (const double*)sh.alive_k, sh.live_k, sh.eps, sh.k});
    return launch_status();
}
int dicp_step_bwd(int dtype, const double* gpose_in, const void* bwd_partials, int nblk, int dim,
                  const void* pose_k, const void* delta_k, int64_t delta_stride, const double* areg_k,
                  void* gs, void* gb, double* gpose_out, int N, void* stream) {
    return step_bwd_go(dtype, gpose_in, bwd_partials, nblk, dim, pose_k, delta_k, delta_stride, areg_k, gs, gb, gpose_out, N, stream, SkipHost{});
}

static int accumulate_bwd_go(int dtype, const dicp_weight_params* prm, const void* src, const void* tgt, int c,
                             const int32_t* idx, const void* pose, const void* w_init, const void* alive,
                             const void* gs, const void* gb, const int32_t* src_rows, int N, int n, int m,
                             void* gsrc, void* gtgt, void* gw, void* bwd_partials, void* stream, const int32_t* skip);
int dicp_accumulate_bwd(int dtype, const dicp_weight_params* prm, const void* src, const void* tgt, int c,
                        const int32_t* idx, const void* pose, const void* w_init, const void* alive,
                        const void* gs, const void* gb, const int32_t* src_rows, int N, int n, int m,
                        void* gsrc, void* gtgt, void* gw, void* bwd_partials, void* stream) {
    return accumulate_bwd_go(dtype, prm, src, tgt, c, idx, pose, w_init, alive, gs, gb, src_rows, N, n, m, gsrc, gtgt, gw, bwd_partials, stream, nullptr);
}
static int accumulate_bwd_go(int dtype, const dicp_weight_params* prm, const void* src, const void* tgt, int c,
                             const int32_t* idx, const void* pose, const void* w_init, const void* alive,
                             const void* gs, const void* gb, const int32_t* src_rows, int N, int n, int m,
                             void* gsrc, void* gtgt, void* gw, void* bwd_partials, void* stream, const int32_t* skip) {
    if (const int e = check_params(prm, c)) return e;
    if (!src || !tgt || !gs || !gb || !gsrc || !bwd_partials || (gw && !w_init)) return DICP_ERR_NULL;
    if (bad_dtype(dtype)) return DICP_ERR_DTYPE;
    if (N <= 0 || n <= 0 || m <= 0 || (!idx && m != n)) return DICP_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    begin_launch();
    const WeightParams P = to_params(prm);
    const int bpc = dicp_accumulate_blocks(n);
    const unsigned g = grid_for(N, bpc);
#define DICP_BWD(T, M) accumulate_bwd_kernel<T, M><<<g, BLOCK, 0, st>>>(P, (const T*)src, (const T*)tgt, c, idx, (const T*)pose, \
        (const T*)w_init, (const T*)alive, (const T*)gs, (const T*)gb, N, n, m, bpc, (T*)gsrc, (T*)gtgt, (T*)gw, (T*)bwd_partials, src_rows, skip)
    if (dtype == DICP_F32) { if (P.mode == MODE_PT2PL) DICP_BWD(float, MODE_PT2PL); else DICP_BWD(float, MODE_PT2PT); }
    else                   { if (P.mode == MODE_PT2PL) DICP_BWD(double, MODE_PT2PL); else DICP_BWD(double, MODE_PT2PT); }
#undef DICP_BWD
    return launch_status();
}

int dicp_gumbel_nn(int dtype, const void* x, const void* y, int c, const void* U, uint32_t seed, double eps, double tau,
                   int N, int n, int m, void* out, void* lse, void* stream) {
    if (!x || !y || !out || !lse) return DICP_ERR_NULL;
    if (bad_dtype(dtype)) return DICP_ERR_DTYPE;
    if (N <= 0 || n <= 0 || m <= 0 || (c != 3 && c != 6) || !(tau > 0.0)) return DICP_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    begin_launch();
    const int bpc = (n + BLOCK - 1) / BLOCK;
    const unsigned g = grid_for(N, bpc);
#define DICP_GF(T, C) gumbel_fwd_kernel<T, C><<<g, BLOCK, 0, st>>>((const T*)x, (const T*)y, (const T*)U, seed, (T)eps, (T)(1.0 / tau), (T*)out, (T*)lse, N, n, m, bpc)
    if (dtype == DICP_F32) { if (c == 6) DICP_GF(float, 6); else DICP_GF(float, 3); }
    else                   { if (c == 6) DICP_GF(double, 6); else DICP_GF(double, 3); }
#undef DICP_GF
    return launch_status();
}

static int gumbel_nn_bwd_go(int dtype, const void* x, const void* y, int c, const void* U, uint32_t seed, double eps, double tau,
                            const void* out, const void* lse, const void* gout, int N, int n, int m, void* gx, void* gy, int add_gy, void* stream);
int dicp_gumbel_nn_bwd(int dtype, const void* x, const void* y, int c, const void* U, uint32_t seed, double eps, double tau,
                       const void* out, const void* lse, const void* gout, int N, int n, int m, void* gx, void* gy, void* stream) {
    return gumbel_nn_bwd_go(dtype, x, y, c, U, seed, eps, tau, out, lse, gout, N, n, m, gx, gy, 0, stream);
}
static int gumbel_nn_bwd_go(int dtype, const void* x, const void* y, int c, const void* U, uint32_t seed, double eps, double tau,
                            const void* out, const void* lse, const void* gout, int N, int n, int m, void* gx, void* gy, int add_gy, void* stream) {
    if (!x || !y || !out || !lse || !gout || (!gx && !gy)) return DICP_ERR_NULL;
    if (bad_dtype(dtype)) return DICP_ERR_DTYPE;
    if (N <= 0 || n <= 0 || m <= 0 || (c != 3 && c != 6) || !(tau > 0.0)) return DICP_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    begin_launch();
    const int bq = (n + BLOCK - 1) / BLOCK, bt = (m + BLOCK - 1) / BLOCK;
#define DICP_GB(T, C) do { \
        if (gx) gumbel_bwd_q_kernel<T, C><<<grid_for(N, bq), BLOCK, 0, st>>>((const T*)x, (const T*)y, (const T*)U, seed, (T)eps, (T)(1.0 / tau), (const T*)out, (const T*)lse, (const T*)gout, (T*)gx, N, n, m, bq); \
        if (gy) gumbel_bwd_t_kernel<T, C><<<grid_for(N, bt), BLOCK, 0, st>>>((const T*)x, (const T*)y, (const T*)U, seed, (T)eps, (T)(1.0 / tau), (const T*)out, (const T*)lse, (const T*)gout, (T*)gy, N, n, m, bt, add_gy); } while (0)
    if (dtype == DICP_F32) { if (c == 6) DICP_GB(float, 6); else DICP_GB(float, 3); }
    else                   { if (c == 6) DICP_GB(double, 6); else DICP_GB(double, 3); }
#undef DICP_GB
    return launch_status();
}

int dicp_kabsch_accumulate(int dtype, const void* src, const void* tgt, int c, const int32_t* idx, const void* pose,
                           const void* w_init, int trim_on, double trim_dist, const int32_t* src_rows, int N, int n, int m, void* partials, void* stream) {
    if (!src || !tgt || !idx || !w_init || !partials) return DICP_ERR_NULL;
    if (bad_dtype(dtype)) return DICP_ERR_DTYPE;
    if (N <= 0 || n <= 0 || m <= 0 || (c != 3 && c != 6)) return DICP_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    begin_launch();
    const int bpc = dicp_accumulate_blocks(n);
    if (dtype == DICP_F32) kabsch_accumulate_kernel<float><<<grid_for(N, bpc), BLOCK, 0, st>>>((const float*)src, (const float*)tgt, c, idx, (const float*)pose, (const float*)w_init, trim_on, (float)trim_dist, N, n, m, bpc, (float*)partials, src_rows);
    else                   kabsch_accumulate_kernel<double><<<grid_for(N, bpc), BLOCK, 0, st>>>((const double*)src, (const double*)tgt, c, idx, (const double*)pose, (const double*)w_init, trim_on, trim_dist, N, n, m, bpc, (double*)partials, src_rows);
    return launch_status();
}

int dicp_kabsch_step(int dtype, const void* partials, int nblk, void* pose_out, void* cost, double* save, int N, void* stream) {
    if (!partials || !pose_out) return DICP_ERR_NULL;
    if (bad_dtype(dtype)) return DICP_ERR_DTYPE;
    if (N <= 0 || nblk <= 0) return DICP_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    begin_launch();
    if (dtype == DICP_F32) kabsch_step_kernel<float><<<N, WAVE, 0, st>>>((const float*)partials, nblk, (float*)pose_out, (float*)cost, save, N);
    else                   kabsch_step_kernel<double><<<N, WAVE, 0, st>>>((const double*)partials, nblk, (double*)pose_out, (double*)cost, save, N);
    return launch_status();
}

// Iterations [k0, k1) of the SVD loop (ICP.py:549-586) enqueued back to back: K x { search -> 18 sums -> 3x3 SVD step }, no host work between.
int dicp_kabsch_forward(int dtype, const dicp_kabsch_buffers* B, int N, int n, int m, int trim_on, double trim_dist, int const_iter, double tolerance,
                        int k0, int k1, void* stream) {
    if (!B || !B->src || !B->tgt || !B->w_init || !B->tgt4 || !B->pose || !B->pose_used || !B->idx || !B->partials || !B->save || !B->costs ||
        !B->iterations || !B->rows_live || !B->counters) return DICP_ERR_NULL;
    if (bad_dtype(dtype)) return DICP_ERR_DTYPE;
    if (k0 < 0 || k1 > B->K || k0 > k1 || N <= 0 || n <= 0 || m <= 0 || (B->c != 3 && B->c != 6)) return DICP_ERR_SHAPE;
    const int kind = B->knn_variant & 0xff;
    if (kind == DICP_KNN_SWEEP && (!B->tperm || !B->bucket || !B->brange)) return DICP_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    const int nblk = dicp_accumulate_blocks(n);
    const void* pose_s = B->pose_search ? B->pose_search : B->pose;
    for (int k = k0; k < k1; ++k) {
        int rc;
        if (kind == DICP_KNN_SWEEP)
            rc = dicp_knn_sweep(dtype, B->src, pose_s, B->tgt4, B->tperm, B->qorder, B->bucket, B->brange, B->nbkt, B->rows_live, B->tgt_rows, N, n, m, B->m_pad,
                                B->idx, nullptr, B->pairs, (B->knn_variant >> 8) & 0xff, B->tgt_f16, nullptr, nullptr, 0, stream);
        else
            rc = dicp_knn(dtype, B->src, pose_s, B->tgt4, B->rows_live, B->tgt_rows, N, n, m, B->m_pad, B->idx, B->knn_variant & 0xffff, B->tgt_f16, stream);
        if (rc) return rc;
        rc = dicp_kabsch_accumulate(dtype, B->src, B->tgt, B->c, B->idx, B->pose, B->w_init, trim_on, trim_dist, B->rows_live, N, n, m, B->partials, stream);
        if (rc) return rc;
        begin_launch();
        if (dtype == DICP_F32)
            kabsch_loop_step_kernel<float><<<N, WAVE, 0, st>>>((const float*)B->partials, nblk, (float*)B->pose, (float*)B->pose_search, (float*)B->pose_used,
                                                               (const float*)B->frame, (float*)B->costs, (long)B->K, k, B->save, B->rows_live, (float*)B->iterations,
                                                               const_iter, tolerance, B->counters);
        else
            kabsch_loop_step_kernel<double><<<N, WAVE, 0, st>>>((const double*)B->partials, nblk, (double*)B->pose, (double*)B->pose_search, (double*)B->pose_used,
                                                                (const double*)B->frame, (double*)B->costs, (long)B->K, k, B->save, B->rows_live, (double*)B->iterations,
                                                                const_iter, tolerance, B->counters);
        rc = launch_status();
        if (rc) return rc;
    }
    return 0;
}

int dicp_kabsch_step_bwd(int dtype, const void* gpose, const double* save, void* gacc, int N, void* stream) {
    if (!gpose || !save || !gacc) return DICP_ERR_NULL;
    if (bad_dtype(dtype)) return DICP_ERR_DTYPE;
    if (N <= 0) return DICP_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    begin_launch();
    if (dtype == DICP_F32) kabsch_step_bwd_kernel<float><<<N, WAVE, 0, st>>>((const float*)gpose, save, (float*)gacc, N);
    else                   kabsch_step_bwd_kernel<double><<<N, WAVE, 0, st>>>((const double*)gpose, save, (double*)gacc, N);
    return launch_status();
}

int dicp_kabsch_bwd(int dtype, const void* src, const void* tgt, int c, const int32_t* idx, const void* pose, const void* w_init,
                    int trim_on, double trim_dist, const void* gacc, const int32_t* src_rows, int N, int n, int m, void* gsrc, void* gtgt, void* gw, void* stream) {
    if (!src || !tgt || !idx || !w_init || !gacc || !gsrc) return DICP_ERR_NULL;
    if (bad_dtype(dtype)) return DICP_ERR_DTYPE;
    if (N <= 0 || n <= 0 || m <= 0 || (c != 3 && c != 6)) return DICP_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    begin_launch();
    const int bpc = dicp_accumulate_blocks(n);
    if (dtype == DICP_F32) kabsch_bwd_kernel<float><<<grid_for(N, bpc), BLOCK, 0, st>>>((const float*)src, (const float*)tgt, c, idx, (const float*)pose, (const float*)w_init, trim_on, (float)trim_dist, (const float*)gacc, N, n, m, bpc, (float*)gsrc, (float*)gtgt, (float*)gw, src_rows);
    else                   kabsch_bwd_kernel<double><<<grid_for(N, bpc), BLOCK, 0, st>>>((const double*)src, (const double*)tgt, c, idx, (const double*)pose, (const double*)w_init, trim_on, trim_dist, (const double*)gacc, N, n, m, bpc, (double*)gsrc, (double*)gtgt, (double*)gw, src_rows);
    return launch_status();
}

int dicp_resolve_matches(const int32_t* spos, const int32_t* spos_of, int k, const int32_t* src_rows, int N, int n, int32_t* out, void* stream) {
    if (!spos || !out) return DICP_ERR_NULL;
    if (N <= 0 || n <= 0 || k < 0) return DICP_ERR_SHAPE;
    begin_launch();
    const int bpc = (int)blocks_for((size_t)n);
    const MatchHist h = spos_of ? MatchHist{spos, spos_of, k, N, n, (n + WAVE - 1) / WAVE} : plain_matches(spos + (size_t)k * N * n, N, n);
    resolve_matches_kernel<<<grid_for(N, bpc), BLOCK, 0, (hipStream_t)stream>>>(h, bpc, src_rows, out);
    return launch_status();
}

// The slot order of ICP.deterministic's backward: the queries in a STABLE order of their reference matches (dicp_sweep_sort's radix sort on the match
// positions as keys) -- the same permutation on every run, windows as local as they can be.  (Round 5 took it from torch.argsort: the one torch operator on
// the product's per-call path.)
static size_t match_order_parts(int dtype, int N, int n, size_t* keys3, size_t* keys_sorted, size_t* tperm, size_t* sort_scratch) {
    const size_t es = dtype == DICP_F32 ? 4 : 8, m_pad = (size_t)dicp_padded_targets(n);
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t off = 0;
    *keys3 = off; off = up(off + (size_t)N * n * 3 * es);
    *keys_sorted = off; off = up(off + (size_t)N * m_pad * es);
    *tperm = off; off = up(off + (size_t)N * m_pad * 4);
    *sort_scratch = off; off = up(off + dicp_sweep_sort_scratch_bytes(dtype, N, (int)m_pad));
    return off;
}
size_t dicp_match_order_scratch_bytes(int dtype, int N, int n) {
    if (bad_dtype(dtype) || N <= 0 || n <= 0) return 0;
    size_t a, b, c, d;
    return match_order_parts(dtype, N, n, &a, &b, &c, &d);
}
int dicp_match_order(int dtype, const int32_t* spos_ref, const int32_t* src_rows, int N, int n, void* scratch, size_t scratch_bytes, int32_t* qorder, void* stream) {
    if (!spos_ref || !scratch || !qorder) return DICP_ERR_NULL;
    if (bad_dtype(dtype)) return DICP_ERR_DTYPE;
    if (N <= 0 || n <= 0) return DICP_ERR_SHAPE;
    if ((uintptr_t)scratch & 255) return DICP_ERR_ALIGN;
    size_t k3, ks, tp, sc;
    if (scratch_bytes < match_order_parts(dtype, N, n, &k3, &ks, &tp, &sc)) return DICP_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    char* W = (char*)scratch;
    const int m_pad = dicp_padded_targets(n);
    begin_launch();
    const int bpc = (int)blocks_for((size_t)n);
    if (dtype == DICP_F32) match_keys_kernel<float><<<grid_for(N, bpc), BLOCK, 0, st>>>(spos_ref, src_rows, N, n, bpc, (float*)(W + k3));
    else                   match_keys_kernel<double><<<grid_for(N, bpc), BLOCK, 0, st>>>(spos_ref, src_rows, N, n, bpc, (double*)(W + k3));
    if (const int rc = launch_status()) return rc;
    const size_t sort_bytes = dicp_sweep_sort_scratch_bytes(dtype, N, m_pad);
    if (const int rc = dicp_sweep_sort(dtype, W + k3, 3, nullptr, nullptr, N, n, m_pad, W + ks, (int32_t*)(W + tp), 0, nullptr, nullptr, sort_bytes ? W + sc : nullptr, sort_bytes, stream)) return rc;
    // the first n entries of each cloud's sorted order (the slots past n are the sort's own pads, last): a strided device copy
    if (const int e = dicp_fill::copy_rows(qorder, (size_t)n * 4, W + tp, (size_t)m_pad * 4, (size_t)n * 4, (size_t)N, st)) return e;
    return 0;
}

int dicp_window_blocks(int dtype, int n, int m_pad) {
    if (n <= 0 || m_pad <= 0) return 0;
    const int spb = window_slots(dtype == DICP_F32 ? WindowRows<float>::v : WindowRows<double>::v, n, m_pad);
    return (n + spb - 1) / spb;
}

int dicp_window_rows(int dtype) { return dtype == DICP_F32 ? WindowRows<float>::v : WindowRows<double>::v; }

// The one-launch tail of the reverse sweep (bwd_tail_kernel) lets the blocks of a cloud wait for each other inside an ordinary launch.  That is
// only safe while ALL of a cloud's blocks can be resident at once: blocks are dispatched in index order and a cloud's blocks share an XCD
// (decode_block), so a cloud with more blocks than an XCD holds would leave its first ones waiting for blocks that can never start.  The most
// blocks per cloud the tail may be used with: HALF of what one XCD holds of that kernel (the occupancy query for the instantiation with the
// larger register footprint, x the XCD's compute units) -- the other half is room for a second such launch on another stream, or for an
// occupancy answer that is one block per unit too high.  0: never (query failed).  dicp_icp_backward refuses a tail beyond it.
int dicp_bwd_tail_max_blocks(int dtype) {
    constexpr int MAXDEV = 64;
    static int cap[MAXDEV][2];                               // per DEVICE of the process (0: not asked yet; the answer + 1 otherwise)
    if (bad_dtype(dtype)) return 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    if (dev >= 0 && dev < MAXDEV && cap[dev][dtype] > 0) return cap[dev][dtype] - 1;
    int cus = 0, xcds = 0, a = 0, b = 0;
    hipError_t e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&xcds, hipDeviceAttributeNumberOfXccs, dev);       // (the device's own count: 8 on MI355X)
    if (e == hipSuccess) {
        if (dtype == DICP_F32) {
            e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, bwd_tail_kernel<float, MODE_PT2PL, WindowRows<float>::v>, BLOCK, 0);
            if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, bwd_tail_kernel<float, MODE_PT2PT, WindowRows<float>::v>, BLOCK, 0);
        } else {
            e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, bwd_tail_kernel<double, MODE_PT2PL, WindowRows<double>::v>, BLOCK, 0);
            if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, bwd_tail_kernel<double, MODE_PT2PT, WindowRows<double>::v>, BLOCK, 0);
        }
    }
    if (e != hipSuccess) { (void)hipGetLastError(); return 0; }           // (not cached: no device yet)
    // (decode_block deals a cloud's blocks to ONE of 8 residue classes of the block index -- one XCD on an 8-XCD part; with another count the classes spread
    //  over the XCDs and the bound is the whole device's share of one class, which is smaller or equal: still safe)
    const int per_cu = a < b ? a : b;
    const int ans = (per_cu > 0 && xcds > 0 && cus >= xcds) ? (per_cu * (cus / (xcds > 8 ? xcds : 8))) / 2 : 0;
    if (dev >= 0 && dev < MAXDEV) cap[dev][dtype] = ans + 1;
    return ans;
}

static int accumulate_bwd_window_go(int dtype, const dicp_weight_params* prm, const void* src_s, const void* tgt_s, int c,
                                    const MatchHist spos, const int32_t* spos_ref, const int32_t* qorder, const void* pose, const void* w_s,
                                    const void* alive, const void* gs, const void* gb, const int32_t* src_rows, int N, int n, int m_pad, void* gsrc_s, void* slab,
                                    void* gts_far, void* gw_s, void* bwd_partials, int overwrite, void* stream, const int32_t* skip, int32_t* det_row = nullptr, void* det_val = nullptr);
int dicp_accumulate_bwd_window(int dtype, const dicp_weight_params* prm, const void* src_s, const void* tgt_s, int c,
                               const int32_t* spos, const int32_t* spos_ref, const int32_t* qorder, const void* pose, const void* w_s,
                               const void* alive, const void* gs, const void* gb, const int32_t* src_rows, int N, int n, int m_pad, void* gsrc_s, void* slab,
                               void* gts_far, void* gw_s, void* bwd_partials, int overwrite, void* stream) {
    if (!spos) return DICP_ERR_NULL;
    return accumulate_bwd_window_go(dtype, prm, src_s, tgt_s, c, plain_matches(spos, N, n), spos_ref, qorder, pose, w_s, alive, gs, gb, src_rows, N, n, m_pad, gsrc_s, slab, gts_far, gw_s,
                                    bwd_partials, overwrite, stream, nullptr);
}
static int accumulate_bwd_window_go(int dtype, const dicp_weight_params* prm, const void* src_s, const void* tgt_s, int c,
                                    const MatchHist spos, const int32_t* spos_ref, const int32_t* qorder, const void* pose, const void* w_s,
                                    const void* alive, const void* gs, const void* gb, const int32_t* src_rows, int N, int n, int m_pad, void* gsrc_s, void* slab,
                                    void* gts_far, void* gw_s, void* bwd_partials, int overwrite, void* stream, const int32_t* skip, int32_t* det_row, void* det_val) {
    if (const int e = check_params(prm, c)) return e;
    if ((det_row != nullptr) != (det_val != nullptr)) return DICP_ERR_NULL;
    if (!src_s || !tgt_s || !spos.base || !spos_ref || !pose || (gw_s && !w_s) || !gs || !gb || !gsrc_s || !bwd_partials || (slab && !gts_far))
        return DICP_ERR_NULL;
    if (bad_dtype(dtype)) return DICP_ERR_DTYPE;
    if (N <= 0 || n <= 0 || m_pad <= 0 || m_pad % KNN_PAD) return DICP_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    begin_launch();
    hipEvent_t ev0, ev1;
    take_launch_events(ev0, ev1);
    const WeightParams P = to_params(prm);
    const int bpc = dicp_window_blocks(dtype, n, m_pad);
    const unsigned g = grid_for(N, bpc);
#define DICP_WIN(T, M) do { if (overwrite) DICP_WIN_O(T, M, true); else DICP_WIN_O(T, M, false); } while (0)
#define DICP_WIN_O(T, M, OV) do { constexpr int WT = WindowRows<T>::v; const int spb = window_slots(WT, n, m_pad); \
        hipExtLaunchKernelGGL((accumulate_bwd_window_kernel<T, M, WT, OV>), dim3(g), dim3(BLOCK), 0, st, ev0, ev1, 0, P, (const T*)src_s, (const T*)tgt_s, c, spos, spos_ref, qorder, (const T*)pose, \
            (const T*)w_s, (const T*)alive, (const T*)gs, (const T*)gb, N, n, m_pad, spb, bpc, (T*)gsrc_s, (T*)slab, (T*)gts_far, (T*)gw_s, \
            (T*)bwd_partials, src_rows, skip, det_row, (T*)det_val); \
        if (det_row && slab) far_apply_kernel<T, (M == MODE_PT2PL ? 6 : 3)><<<N * FAR_RANGES, BLOCK, 0, st>>>(det_row, (const T*)det_val, (T*)gts_far, n, m_pad, src_rows, OV ? nullptr : skip); } while (0)
    if (dtype == DICP_F32) { if (P.mode == MODE_PT2PL) DICP_WIN(float, MODE_PT2PL); else DICP_WIN(float, MODE_PT2PT); }
    else                   { if (P.mode == MODE_PT2PL) DICP_WIN(double, MODE_PT2PL); else DICP_WIN(double, MODE_PT2PT); }
#undef DICP_WIN
#undef DICP_WIN_O
    return launch_status();
}

int dicp_window_reduce(int dtype, const void* slab, const int32_t* spos_ref, const int32_t* qorder, const int32_t* tperm, const void* gts_far,
                       const int32_t* src_rows, int N, int n, int m, int m_pad, int cv, void* gtgt, int c, int overwrite, void* stream) {
    if (!slab || !spos_ref || !tperm || !gtgt) return DICP_ERR_NULL;
    if (bad_dtype(dtype)) return DICP_ERR_DTYPE;
    if (N <= 0 || n <= 0 || m <= 0 || m_pad != dicp_padded_targets(m) || (cv != 3 && cv != 6) || c < cv) return DICP_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    begin_launch();
    const int bpc = dicp_window_blocks(dtype, n, m_pad);
    const int rpc = (m * cv + BLOCK * WR_U - 1) / (BLOCK * WR_U);
    const unsigned g = grid_for(N, rpc);
#define DICP_RED(T, CVV) window_reduce_kernel<T, WindowRows<T>::v, CVV><<<g, BLOCK, 0, st>>>((const T*)slab, spos_ref, qorder, tperm, (const T*)gts_far, N, n, m, m_pad, cv, \
        window_slots(WindowRows<T>::v, n, m_pad), bpc, rpc, (T*)gtgt, c, overwrite, src_rows)
    if (dtype == DICP_F32) { if (cv == 6) DICP_RED(float, 6); else DICP_RED(float, 3); }
    else                   { if (cv == 6) DICP_RED(double, 6); else DICP_RED(double, 3); }
#undef DICP_RED
    return launch_status();
}

static int permute_rows(int dtype, const void* in, const int32_t* perm, int N, int cnt, int in_rows, int perm_rows, int c_in, int cols,
                        void* out, int out_rows, int c_out, int overwrite, void* stream);
int dicp_permute_add_rows(int dtype, const void* in, const int32_t* perm, int N, int cnt, int in_rows, int perm_rows, int c_in, int cols,
                          void* out, int out_rows, int c_out, void* stream) {
    return permute_rows(dtype, in, perm, N, cnt, in_rows, perm_rows, c_in, cols, out, out_rows, c_out, 0, stream);
}
int dicp_permute_rows(int dtype, const void* in, const int32_t* perm, int N, int cnt, int in_rows, int perm_rows, int c_in, int cols,
                      void* out, int out_rows, int c_out, void* stream) {
    return permute_rows(dtype, in, perm, N, cnt, in_rows, perm_rows, c_in, cols, out, out_rows, c_out, 1, stream);
}
static int permute_rows(int dtype, const void* in, const int32_t* perm, int N, int cnt, int in_rows, int perm_rows, int c_in, int cols,
                        void* out, int out_rows, int c_out, int overwrite, void* stream) {
    if (!in || !perm || !out) return DICP_ERR_NULL;
    if (bad_dtype(dtype)) return DICP_ERR_DTYPE;
    if (N <= 0 || cnt <= 0 || cnt > in_rows || cnt > perm_rows || cols <= 0 || cols > c_in || cols > c_out || out_rows <= 0) return DICP_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    begin_launch();
    if ((size_t)cnt * cols > 0x7fffffffu) return DICP_ERR_SHAPE;
    const int bpc = (int)(((size_t)cnt * cols + BLOCK * ROWS_U - 1) / (BLOCK * ROWS_U));
    const unsigned g = grid_for(N, bpc);
#define DICP_PERM(T, C) permute_add_rows_kernel<T, C><<<g, BLOCK, 0, st>>>((const T*)in, perm, N, cnt, in_rows, perm_rows, c_in, cols, (T*)out, out_rows, c_out, bpc, overwrite)
#define DICP_PERM_C(T) do { if (cols == 1) DICP_PERM(T, 1); else if (cols == 3) DICP_PERM(T, 3); else if (cols == 6) DICP_PERM(T, 6); else DICP_PERM(T, 0); } while (0)
    if (dtype == DICP_F32) DICP_PERM_C(float); else DICP_PERM_C(double);
#undef DICP_PERM
#undef DICP_PERM_C
    return launch_status();
}

// One block per cloud runs the whole chunk when the packed targets fit comfortably in LDS and a block's brute-force
// search stays in the microseconds (SMALL_PAIRS pairs per iteration); anything bigger is better off spread over the chip.
constexpr long SMALL_PAIRS = 128L * 1024;    // measured break-even against the multi-kernel path: ~512 x 512 (profiles/r01_small_clouds.txt)
static bool small_loop_eligible(int dtype, int kind, int knn_variant, int n, int m_pad) {
    if (kind == DICP_KNN_SWEEP || kind == DICP_KNN_MFMA || ((knn_variant >> 25) & 1) || ((knn_variant >> 8) & 0xff)) return false;   // bit 25: switched off by the caller
    const size_t lds = (size_t)m_pad * (dtype == DICP_F32 ? 16 : 32);
    return lds <= 48 * 1024 && (long)n * m_pad <= SMALL_PAIRS;
}

static int transform_points_bwd_go(int dtype, const void* src, const void* pose, const void* gout, void* gsrc, void* partials, int N, int n, int add, void* stream);
// ------------------------------------------------------------------ whole-loop entry points
// The iteration loop of ICP.dICP (ICP.py:131-260) behind ONE call: K x { kNN -> accumulate -> step } are
// enqueued back to back on the stream with every piece of per-iteration state in caller-allocated buffers
// (pose / alive / index / weight histories indexed by iteration), so the host does no per-iteration work.
// The reference's per-iteration host check `all(converged)` (ICP.py:259) is the caller's business: it runs
// [k0,k1) chunks and reads counters[] between them (converged clouds are frozen, extra iterations are no-ops).
// tolerance mode: the segment's convergence counters to the host's mapped words (dicp_loop_buffers.counters_host), behind the segment's last step
static int report_counters(const dicp_loop_buffers* B, int const_iter, int k0, int k1, void* stream) {
    if (const_iter || !B->counters_host || k1 <= k0) return 0;
    begin_launch();
    counters_report_kernel<<<1, WAVE, 0, (hipStream_t)stream>>>(B->counters, B->counters_host, k0, k1 - k0, B->counters_tag);
    return launch_status();
}

int dicp_icp_forward(int dtype, const dicp_weight_params* prm, const dicp_loop_buffers* B, int N, int n, int m,
                     int dim, int const_iter, double tolerance, int k0, int k1, void* stream) {
    if (!prm || !B || !B->src || !B->tgt || !B->hist.poses || !B->hist.deltas || !B->hist.costs || !B->hist.alive ||
        !B->converged || !B->iterations || !B->matched_ratio || !B->n_start || !B->n_matched || !B->hist.w ||
        !B->partials || !B->counters) return DICP_ERR_NULL;
    if (B->abi != DICP_ABI_VERSION) return DICP_ERR_ABI;
    if (bad_dtype(dtype)) return DICP_ERR_DTYPE;
    if (k0 < 0 || k1 > B->K || k0 > k1 || N <= 0 || n <= 0 || m <= 0 || B->hist.w_stride < n || B->hist.w_iter < n) return DICP_ERR_SHAPE;
    const size_t es = dtype == DICP_F32 ? 4 : 8;
    const int kind = B->search.knn_variant & 0xff;
    const dicp_gumbel_loop* G = kind == DICP_KNN_GUMBEL ? B->search.gumbel : nullptr;
    if (kind == DICP_KNN_GUMBEL) { if (!G || !G->ps_t || !G->nbr || !G->lse || (!G->U && !G->seeds)) return DICP_ERR_NULL; }
    else if ((!B->hist.idx && !B->hist.spos) || (kind == DICP_KNN_SWEEP ? (!B->search.tperm || !B->search.bucket || !B->search.brange) : !B->search.tgt4)) return DICP_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    const int nblk = dicp_accumulate_blocks(n);
    if (!G && small_loop_eligible(dtype, kind, B->search.knn_variant, n, B->search.m_pad) && k1 > k0) {
        // small clouds: the whole chunk is ONE launch, one block per cloud (bit 25 of knn_variant switches this off)
        if (const int e = check_params(prm, B->c)) return e;
        begin_launch();
        const WeightParams P = to_params(prm);
        const size_t lds = (size_t)B->search.m_pad * (dtype == DICP_F32 ? sizeof(float4) : sizeof(double4));
#define DICP_SMALL(T, M) icp_small_forward_kernel<T, M><<<N, BLOCK, lds, st>>>(P, *B, N, n, m, dim, const_iter, tolerance, k0, k1)
        if (dtype == DICP_F32) { if (P.mode == MODE_PT2PL) DICP_SMALL(float, MODE_PT2PL); else DICP_SMALL(float, MODE_PT2PT); }
        else                   { if (P.mode == MODE_PT2PL) DICP_SMALL(double, MODE_PT2PL); else DICP_SMALL(double, MODE_PT2PT); }
#undef DICP_SMALL
        if (const int e = launch_status()) return e;
        return report_counters(B, const_iter, k0, k1, stream);
    }
    for (int k = k0; k < k1; ++k) {
        const char* pose_k = (const char*)B->hist.poses + (size_t)k * N * 12 * es;
        // the searches read [C | r - centre] when the caller keeps that second pose history (packed rows are then y - centre)
        const char* pose_s = B->search.poses ? (const char*)B->search.poses + (size_t)k * N * 12 * es : pose_k;
        int32_t* idx_k = B->hist.idx ? B->hist.idx + (B->hist.per_iter ? (size_t)k * N * n : 0) : nullptr;
        char* w_k = (char*)B->hist.w + (size_t)k * B->hist.w_iter * es;       // cloud stride B->hist.w_stride: (N,K,n) or (K,N,n) alike
        const char* w_prev_k = k > k0 ? (const char*)B->hist.w + (size_t)(k - 1) * B->hist.w_iter * es : (const char*)B->hist.w_prev0;     // (as make_step_io)
        const char* alive_k = (const char*)B->hist.alive + (size_t)k * N * es;
        if (B->events) {    // the sweep launch carries its two events itself; the brute-force forms are bracketed by records
            if (kind == DICP_KNN_SWEEP) set_launch_events((hipEvent_t)B->events[6 * k + 0], (hipEvent_t)B->events[6 * k + 1]);
            else if (hipEventRecord((hipEvent_t)B->events[6 * k + 0], st) != hipSuccess) return -(int)hipGetLastError();
        }
        int rc;
        if (kind == DICP_KNN_SWEEP) {
            // bits 8..15 of knn_variant optionally pin a tile-sweep launch configuration (0 = chosen from the problem size)
            int cfg = (B->search.knn_variant >> 8) & 0xff;
            int32_t* spos_k = B->hist.spos ? B->hist.spos + (B->hist.per_iter ? (size_t)k * N * n : 0) : nullptr;
            const bool sorted_rows = B->search.tgt_sorted && spos_k;      // accumulate gathers 32-byte aligned rows of the sorted copy at the sorted positions
            if (!sorted_rows && !B->hist.idx) { set_launch_events(nullptr, nullptr); return DICP_ERR_NULL; }
            // match certificates: only the units holding a query whose match is not proven unchanged are searched again
            const int cfg_plain = cfg ? cfg : sweep_auto_cfg(N, n);
            const bool cert = B->cert.q && B->cert.qu && B->cert.rmax && B->cert.dcum && spos_k && sorted_rows && !B->hist.idx && B->search.qorder && sweep_queries_per_lane(cfg_plain) > 0;
            // (the certified iterations keep a row cache and their match history by reference: accumulate_kernel)
            if (cert && (!B->cert.nbr || !B->cert.gdirty || !B->cert.pend || !B->cert.cm || !B->cert.glist || !B->cert.gcount || (B->cert.set && (!B->cert.slist || !B->cert.scount)) || (B->hist.per_iter && !B->hist.spos_of))) { set_launch_events(nullptr, nullptr); return DICP_ERR_NULL; }
            const int cert_units = cert ? (n + WAVE * sweep_queries_per_lane(cfg_plain) - 1) / (WAVE * sweep_queries_per_lane(cfg_plain)) : 0;
            const int glist_cap = guard_list_cap(N, n);
            const bool fresh = k == 0 || (k == k0 && B->cert.reset);           // a new query order: every query is searched, every budget written
            int32_t* count_k = B->cert.count ? B->cert.count + (size_t)k * 2 * CERT_SHARDS : nullptr;
            const bool searched = B->search.first_done && k == 0 && !cert && spos_k && !B->hist.idx;    // the caller ran iteration 0's search itself, ahead of this call
            if (searched) rc = 0;
            else if (cert) {
                // (nothing is copied from iteration to iteration, or from call to call: where a group of queries finds its matches is a word of spos_of)
                begin_launch();
                rc = sweep_launch(dtype, B->src, pose_s, B->search.tgt4, B->search.tperm, B->search.qorder, B->search.bucket, B->search.brange, B->search.nbkt, N, n, m, B->search.m_pad, nullptr, spos_k,
                                  B->search.pairs, cfg, Rows{B->src_rows, B->tgt_rows}, st, CertArgs{B->cert.q, B->cert.qu, B->cert.dcum, 2 * (B->K + 1), k, count_k, !fresh, B->cert.cloud, B->cert.set,
                                                                                              B->cert.cm, B->cert.pend, B->cert.gdirty, B->cert.set ? B->cert.slist : nullptr, B->cert.set ? B->cert.scount : nullptr,
                                                                                              B->cert.glist, B->cert.gcount ? B->cert.gcount + (size_t)k * 8 : nullptr, glist_cap});
            } else
            {
                // plain search.  The matrix-core form pays where a wave's slab is long -- big clouds, and clouds of any size whose queries are far from their
                // matches (the first iterations; start poses a metre off; parts of the source without counterpart in the target) -- and loses where it is a few
                // tiles.  Which form iteration k takes: the caller's plan (sweep_form_plan[k]: 1 vector, 2 matrix cores -- what the tallies of an EARLIER call
                // of the shape said about this iteration: one launch), else per cloud from the previous plain search's tally (both forms are launched, each takes
                // its clouds: 0.02-0.03 ms per search at 256 x 16384 for the launch that finds nothing to do), else the default.  Every plain search tallies its
                // slabs' tiles per cloud when given sweep_form.
                const int planned = (B->search.form_plan && B->search.tgt_f16) ? B->search.form_plan[k] : 0;
                const int32_t* form_in = (!planned && B->search.form && B->search.tgt_f16 && k > 0) ? B->search.form + (size_t)(k - 1) * N : nullptr;     // (all zeros behind a certified iteration: no tally)
                int32_t* form_out = B->search.form ? B->search.form + (size_t)k * N : nullptr;
                const void* img = planned ? (planned == 2 ? B->search.tgt_f16 : nullptr) : ((form_in || B->search.form_default || !B->search.form) ? B->search.tgt_f16 : nullptr);
                begin_launch();
                rc = sweep_launch(dtype, B->src, pose_s, B->search.tgt4, B->search.tperm, B->search.qorder, B->search.bucket, B->search.brange, B->search.nbkt, N, n, m, B->search.m_pad, B->hist.idx ? idx_k : nullptr, spos_k, B->search.pairs, cfg,
                                  Rows{B->src_rows, B->tgt_rows}, st, CertArgs{}, img, FormArgs{form_in, form_out, B->search.form_default});
            }
            set_launch_events(nullptr, nullptr);
            if (rc) return rc;
            if (B->events) set_launch_events((hipEvent_t)B->events[6 * k + 2], (hipEvent_t)B->events[6 * k + 3]);
            if (cert) {
                // the accumulate of a certified iteration checks every point's budget and searches the spent ones on the spot; its matches
                // are the start of the next iteration's (within this call)
                const CertAcc ca{spos_k, B->hist.spos, B->hist.spos_prev_chunk, B->hist.per_iter ? B->hist.spos_of : nullptr, B->hist.spos_floor, k,
                                 B->cert.nbr, B->cert.gdirty, B->cert.pend, B->cert.cloud, fresh ? 1 : 0, cert_units, B->cert.set ? 1 : 0, B->cert.scount};
                rc = accumulate_go(dtype, prm, B->src, B->search.tgt_sorted, B->search.tgt_sorted_stride, spos_k, pose_k, B->w_init, alive_k, B->src_rows, N, n, B->search.m_pad,
                                   B->partials, w_k, B->hist.w_stride, stream, &ca, w_prev_k);
            } else if (sorted_rows)
                rc = accumulate_go(dtype, prm, B->src, B->search.tgt_sorted, B->search.tgt_sorted_stride, spos_k, pose_k, B->w_init, alive_k, B->src_rows, N, n, B->search.m_pad,
                                   B->partials, w_k, B->hist.w_stride, stream, nullptr, w_prev_k);
            else
                rc = accumulate_go(dtype, prm, B->src, B->tgt, B->c, idx_k, pose_k, B->w_init, alive_k, B->src_rows, N, n, m, B->partials, w_k, B->hist.w_stride, stream, nullptr, w_prev_k);
            set_launch_events(nullptr, nullptr);
            if (rc) return rc;
        } else if (G) {
            // soft correspondences (nn.py:43-70 inside ICP.py:137-140): the neighbours are ROWS of their own, one per source point, kept per
            // iteration for the reverse sweep together with the log-sum-exp that lets it rebuild the probabilities
            char* nbr_k = (char*)G->nbr + (size_t)k * N * n * B->c * es;
            char* lse_k = (char*)G->lse + (size_t)k * N * n * es;
            rc = dicp_transform_points(dtype, B->src, pose_k, G->ps_t, N, n, stream);
            if (!rc) rc = dicp_gumbel_nn(dtype, G->ps_t, B->tgt, B->c, G->U ? G->U[k] : nullptr, G->seeds ? G->seeds[k] : 0u, G->eps, G->tau, N, n, m, nbr_k, lse_k, stream);
            if (rc) return rc;
            if (B->events) {
                if (hipEventRecord((hipEvent_t)B->events[6 * k + 1], st) != hipSuccess) return -(int)hipGetLastError();
                set_launch_events((hipEvent_t)B->events[6 * k + 2], (hipEvent_t)B->events[6 * k + 3]);
            }
            rc = accumulate_go(dtype, prm, B->src, nbr_k, B->c, nullptr, pose_k, B->w_init, alive_k, nullptr, N, n, n, B->partials, w_k, B->hist.w_stride, stream, nullptr, w_prev_k);
            set_launch_events(nullptr, nullptr);
            if (rc) return rc;
        } else {
            rc = dicp_knn(dtype, B->src, pose_s, B->search.tgt4, B->src_rows, B->tgt_rows, N, n, m, B->search.m_pad, idx_k, B->search.knn_variant & 0xffff, B->search.tgt_f16, stream);
            if (rc) return rc;
            if (B->events) {
                if (hipEventRecord((hipEvent_t)B->events[6 * k + 1], st) != hipSuccess) return -(int)hipGetLastError();
                set_launch_events((hipEvent_t)B->events[6 * k + 2], (hipEvent_t)B->events[6 * k + 3]);
            }
            rc = accumulate_go(dtype, prm, B->src, B->tgt, B->c, idx_k, pose_k, B->w_init, alive_k, B->src_rows, N, n, m, B->partials, w_k, B->hist.w_stride, stream, nullptr, w_prev_k);
            set_launch_events(nullptr, nullptr);
            if (rc) return rc;
        }
        dicp_step_io io = make_step_io(*B, k, k0, N, n, prm->mode, dim, const_iter, tolerance, es, nblk);
        io.w_copied = w_prev_k ? 1 : 0;
        if (kind == DICP_KNN_SWEEP && B->cert.q && B->cert.qu && B->cert.glist && B->cert.gcount && B->cert.rmax && B->cert.dcum && k + 1 < B->K) {
            // a certified iteration's step also makes the next guard launch its work list (dicp_step_io.glist)
            const int cfgp = ((B->search.knn_variant >> 8) & 0xff) ? ((B->search.knn_variant >> 8) & 0xff) : sweep_auto_cfg(N, n);
            const int Qp = sweep_queries_per_lane(cfgp);
            if (Qp > 0) {
                io.cert_qu = B->cert.qu; io.cert_units = (n + WAVE * Qp - 1) / (WAVE * Qp); io.glist_cap = guard_list_cap(N, n);
                io.glist = B->cert.glist; io.gcount = B->cert.gcount + (size_t)(k + 1) * 8;
                io.cert_scount = B->cert.set ? B->cert.scount : nullptr; io.cert_slist = B->cert.set ? B->cert.slist : nullptr;
            }
        }
        rc = dicp_step(dtype, &io, N, stream);
        if (rc) return rc;
    }
    return report_counters(B, const_iter, k0, k1, stream);
}

// The constant-iteration loop of the sweep path in ONE call: the segments dicp_icp_forward would be called for one by one -- cut where the
// queries are re-ordered and where the certificates start -- with the query re-orderings between them (dicp_query_order under the segment's
// first search pose).  What the host did per segment (four library calls and their glue for a 10-iteration call) is a tenth of a
// millisecond of a mid-size call that has half a millisecond of kernels (scripts/host_breakdown.py).  All histories in ONE slab: buf->spos /
// buf->idx / buf->w are their real bases.  Tolerance mode, where the host looks at the convergence counters between segments, keeps calling
// dicp_icp_forward itself.
int dicp_icp_forward_plan(int dtype, const dicp_weight_params* prm, const dicp_loop_buffers* buf, const dicp_segment_plan* S, int N, int n, int m,
                          int dim, int const_iter, double tolerance, void* stream) {
    if (!prm || !buf || !S) return DICP_ERR_NULL;
    if (buf->abi != DICP_ABI_VERSION) return DICP_ERR_ABI;
    if (bad_dtype(dtype)) return DICP_ERR_DTYPE;
    if (S->nseg <= 0 || S->nseg > DICP_MAX_SEGMENTS) return DICP_ERR_SHAPE;
    const size_t es = dtype == DICP_F32 ? 4 : 8;
    dicp_loop_buffers B = *buf;
    const int32_t* order_before = nullptr;      // the order in use before this segment's, and the iteration it was made (or kept) at
    int k_before = 0;
    for (int s = 0; s < S->nseg; ++s) {
        const int k0 = S->k0[s], k1 = S->k1[s];
        if (k0 < 0 || k1 > B.K || k0 >= k1 || (s > 0 && k0 != S->k1[s - 1])) return DICP_ERR_SHAPE;
        int32_t* qo = S->order[s];
        if (qo && S->new_order[s]) {
            if (!S->keys) return DICP_ERR_NULL;
            const char* poses_s = (const char*)(B.search.poses ? B.search.poses : B.hist.poses);
            // (a re-ordering: a cloud that has hardly moved since the order before keeps it, dicp_query_reorder)
            if (const int rc = query_order_go(dtype, B.src, poses_s + (size_t)k0 * N * 12 * es, B.search.brange, B.search.nbkt, N, n, qo, nullptr, nullptr, nullptr, 0, nullptr, B.search.m_pad,
                                              S->keys, B.search.bucket, m, B.src_rows, B.tgt_rows, (order_before && order_before != qo) ? poses_s + (size_t)k_before * N * 12 * es : nullptr,
                                              order_before != qo ? order_before : nullptr, stream)) return rc;
        }
        if (qo && qo != order_before) { order_before = qo; k_before = k0; }
        B.search.qorder = qo;
        const bool certs = S->cert_from >= 0 && k0 >= S->cert_from;
        B.cert.q = certs ? S->cert_q : nullptr; B.cert.qu = certs ? S->cert_qu : nullptr; B.cert.count = certs ? S->cert_count : nullptr;
        B.cert.cloud = certs ? S->cert_cloud : nullptr;
        B.cert.set = certs ? S->cert_set : nullptr;
        B.cert.nbr = certs ? S->cert_nbr : nullptr; B.cert.gdirty = certs ? S->cert_gdirty : nullptr; B.cert.pend = certs ? S->cert_pend : nullptr; B.cert.cm = certs ? S->cert_cm : nullptr; B.cert.glist = certs ? S->cert_glist : nullptr; B.cert.gcount = certs ? S->cert_gcount : nullptr; B.cert.slist = certs ? S->cert_slist : nullptr; B.cert.scount = certs ? S->cert_scount : nullptr;
        B.cert.reset = (S->cert_from >= 0 && k0 == S->cert_from) ? 1 : 0;
        B.hist.spos_prev_chunk = nullptr; B.hist.spos_floor = 0;
        B.hist.w_prev0 = k0 > 0 ? (const char*)B.hist.w + (size_t)(k0 - 1) * B.hist.w_iter * es : nullptr;
        if (const int rc = dicp_icp_forward(dtype, prm, &B, N, n, m, dim, const_iter, tolerance, k0, k1, stream)) return rc;
    }
    return 0;
}

// Reverse sweep for iterations k1-1 .. k0: K x { step_bwd -> accumulate_bwd }.  gpose (N,12) double holds the
// cotangent of pose_{k1} on entry; the cotangent of pose_{k0} (without the last accumulate_bwd partials, which stay in
// bwd_partials for the caller or the next chunk) is left in gpose when k1-k0 is even and in gpose_tmp when it is odd.
int dicp_icp_backward(int dtype, const dicp_weight_params* prm, const dicp_loop_buffers* B, int N, int n, int m, int dim,
                      double* gpose, double* gpose_tmp, int have_partials, void* gs, void* gb, void* gsrc, void* gtgt, void* gw,
                      void* bwd_partials, int k0, int k1, void* stream) {
    const dicp_gumbel_loop* G = (B && (B->search.knn_variant & 0xff) == DICP_KNN_GUMBEL) ? B->search.gumbel : nullptr;
    if (!prm || !B || !B->src || !B->tgt || !B->hist.poses || !B->hist.deltas || !B->hist.areg || !B->hist.alive || (!G && !B->hist.idx && !B->hist.spos) || (gw && !B->w_init) ||
        !gpose || !gpose_tmp || !gs || !gb || !gsrc || !bwd_partials) return DICP_ERR_NULL;
    if (B->abi != DICP_ABI_VERSION) return DICP_ERR_ABI;
    if ((B->search.knn_variant & 0xff) == DICP_KNN_GUMBEL && (!G || !G->ps_t || !G->nbr || !G->lse || !G->g_nbr || !G->g_ps || (!G->U && !G->seeds))) return DICP_ERR_NULL;
    if (bad_dtype(dtype)) return DICP_ERR_DTYPE;
    if (k0 < 0 || k1 > B->K || k0 > k1 || !B->hist.per_iter || (B->hist.spos && (B->search.m_pad <= 0 || !B->bwd.spos_ref))) return DICP_ERR_SHAPE;
    if (B->bwd.skip && (!B->bwd.mref || !(B->bwd.skip_eps >= 0.0))) return DICP_ERR_NULL;
    const size_t es = dtype == DICP_F32 ? 4 : 8;
    hipStream_t st = (hipStream_t)stream;
    const int nblk = B->hist.spos ? dicp_window_blocks(dtype, n, B->search.m_pad) : dicp_accumulate_blocks(n);
    {   // small clouds (atomic form only): the whole chunk is ONE launch, one block per cloud
        const size_t lds = (size_t)m * (prm->mode == DICP_PT2PL ? 6 : 3) * es;
        if (!G && !B->hist.spos && k1 > k0 && B->search.m_pad > 0 && lds <= 40 * 1024 &&
            small_loop_eligible(dtype, B->search.knn_variant & 0xff, B->search.knn_variant, n, B->search.m_pad)) {
            if (const int e = check_params(prm, B->c)) return e;
            begin_launch();
            const WeightParams P = to_params(prm);
            double* dst = ((k1 - k0) & 1) ? gpose_tmp : gpose;        // where the alternating buffers would have left it
#define DICP_SMALLB(T, M) icp_small_backward_kernel<T, M><<<N, BLOCK, lds, st>>>(P, *B, N, n, m, dim, gpose, dst, have_partials, \
                (T*)gsrc, (T*)gtgt, (T*)gw, (T*)bwd_partials, nblk, k0, k1)
            if (dtype == DICP_F32) { if (P.mode == MODE_PT2PL) DICP_SMALLB(float, MODE_PT2PL); else DICP_SMALLB(float, MODE_PT2PT); }
            else                   { if (P.mode == MODE_PT2PL) DICP_SMALLB(double, MODE_PT2PL); else DICP_SMALLB(double, MODE_PT2PT); }
#undef DICP_SMALLB
            return launch_status();
        }
    }
    double* gin = gpose;
    double* gout = gpose_tmp;
    // windowed form with the truncated sweep: the iterations below bwd_tail_from are ONE launch (bwd_tail_kernel)
    int kt = k0;
    if (B->hist.spos && B->bwd.skip && B->bwd.tail_from > k0) {
        if (B->bwd.det_far_row) return DICP_ERR_SHAPE;          // (deterministic target gradients: per-iteration launches only)
        if (k0 != 0) return DICP_ERR_SHAPE;                 // (the launch runs down to iteration 0 and folds the last pose sums into the cotangent)
        if (!B->bwd.tail_partials || !B->bwd.tail_arrive) return DICP_ERR_NULL;
        if (nblk > dicp_bwd_tail_max_blocks(dtype)) return DICP_ERR_SHAPE;      // (its blocks wait for each other: they must all be resident)
        kt = B->bwd.tail_from < k1 ? B->bwd.tail_from : k1;
        if (B->bwd.overwrite && kt >= k1) kt = k1 - 1;      // (the first windowed launch initialises the accumulators: it always runs)
    }
    for (int k = k1 - 1; k >= kt; --k) {
        const char* pose_k = (const char*)B->hist.poses + (size_t)k * N * 12 * es;
        const char* alive_k = (const char*)B->hist.alive + (size_t)k * N * es;
        const SkipHost sh{B->bwd.skip, B->bwd.mref, alive_k, B->bwd.live ? B->bwd.live + k : nullptr, B->bwd.skip_eps, k};
        int rc = step_bwd_go(dtype, gin, have_partials ? bwd_partials : nullptr, nblk, dim, pose_k,
                             (const char*)B->hist.deltas + (size_t)k * 6 * es, (int64_t)B->K * 6, B->hist.areg + (size_t)k * N * 36,
                             gs, gb, gout, N, stream, B->bwd.skip ? sh : SkipHost{});
        if (rc) return rc;
        if (B->events) {
            if (B->hist.spos) set_launch_events((hipEvent_t)B->events[6 * k + 4], (hipEvent_t)B->events[6 * k + 5]);
            else if (hipEventRecord((hipEvent_t)B->events[6 * k + 4], st) != hipSuccess) return -(int)hipGetLastError();
        }
        if (G) {
            // soft correspondences: the neighbour rows carry gradient themselves -- accumulate_bwd leaves it in g_nbr, the soft kNN's
            // adjoint takes it on to the transformed source (g_ps) and to the target (added up over the iterations), and the transform's
            // adjoint to the source and to the pose (its sums join accumulate_bwd's in bwd_partials)
            const char* nbr_k = (const char*)G->nbr + (size_t)k * N * n * B->c * es;
            const char* lse_k = (const char*)G->lse + (size_t)k * N * n * es;
            if (const int e = dicp_fill::zero(G->g_nbr, (size_t)N * n * B->c * es, st)) return e;
            rc = accumulate_bwd_go(dtype, prm, B->src, nbr_k, B->c, nullptr, pose_k, B->w_init, alive_k, gs, gb, nullptr, N, n, n, gsrc, G->g_nbr, gw, bwd_partials, stream, B->bwd.skip);
            if (!rc) rc = dicp_transform_points(dtype, B->src, pose_k, G->ps_t, N, n, stream);
            if (!rc) rc = gumbel_nn_bwd_go(dtype, G->ps_t, B->tgt, B->c, G->U ? G->U[k] : nullptr, G->seeds ? G->seeds[k] : 0u, G->eps, G->tau, nbr_k, lse_k, G->g_nbr,
                                           N, n, m, G->g_ps, gtgt, 1, stream);
            if (!rc) rc = transform_points_bwd_go(dtype, B->src, pose_k, G->g_ps, gsrc, bwd_partials, N, n, 1, stream);
        } else if (B->hist.spos)    // windowed form: src / w_init / tgt are the SORTED copies, gsrc / gw accumulate in slot order, gtgt is the slab
            rc = accumulate_bwd_window_go(dtype, prm, B->src, B->tgt, B->c,
                                          (B->hist.spos_of && k >= B->hist.spos_of_from) ? MatchHist{B->hist.spos, B->hist.spos_of, k, N, n, (n + WAVE - 1) / WAVE} : plain_matches(B->hist.spos + (size_t)k * N * n, N, n),
                                          B->bwd.spos_ref, B->search.qorder, pose_k, B->w_init,
                                          alive_k, gs, gb, B->src_rows, N, n, B->search.m_pad,
                                          gsrc, gtgt, B->bwd.gts_far, gw, bwd_partials, (B->bwd.overwrite && k == k1 - 1) ? 1 : 0, stream, B->bwd.skip, B->bwd.det_far_row, B->bwd.det_far_val);
        else
            rc = accumulate_bwd_go(dtype, prm, B->src, B->tgt, B->c, B->hist.idx + (size_t)k * N * n, pose_k, B->w_init,
                                   alive_k, gs, gb, B->src_rows, N, n, m, gsrc, gtgt, gw, bwd_partials, stream, B->bwd.skip);
        set_launch_events(nullptr, nullptr);
        if (rc) return rc;
        if (B->events && !B->hist.spos) { if (hipEventRecord((hipEvent_t)B->events[6 * k + 5], st) != hipSuccess) return -(int)hipGetLastError(); }
        have_partials = 1;
        double* t = gin; gin = gout; gout = t;
    }
    if (kt > k0) {
        if (const int e = check_params(prm, B->c)) return e;
        begin_launch();
        const WeightParams P = to_params(prm);
        double* dst = ((k1 - k0) & 1) ? gpose_tmp : gpose;            // where the alternating buffers would have left it
        if (gin == dst) {
            // (an even number of iterations left for the launch: it would read the cotangent from the buffer it writes the result to.  Block 0 of a cloud whose
            //  sweep ends at its first iteration here is through its product chain in microseconds and writes the cloud's result -- and a block of the same
            //  cloud that is dispatched after that (the grid is larger than what is resident at once) would start from the RESULT as its cotangent, reach
            //  another verdict than its siblings and wait for them in vain: the TailTimeout seen every few hundred calls on planar scenes in rounds 5 and 6.
            //  The launch reads from the other buffer instead: 12 doubles per cloud to copy.)
            double* other = gin == gpose ? gpose_tmp : gpose;
            if (const int e = dicp_fill::copy(other, gin, (size_t)N * 12 * sizeof(double), st)) return e;
            gin = other;
        }
        const unsigned g = grid_for(N, nblk);
#define DICP_TAILB(T, M) do { constexpr int WT = WindowRows<T>::v; \
        bwd_tail_kernel<T, M, WT><<<g, BLOCK, 0, st>>>(P, *B, N, n, dim, window_slots(WT, n, B->search.m_pad), nblk, gin, dst, have_partials, \
            (T*)gsrc, (T*)gtgt, (T*)gw, (T*)bwd_partials, (T*)B->bwd.tail_partials, B->bwd.tail_arrive, kt); } while (0)
        if (dtype == DICP_F32) { if (P.mode == MODE_PT2PL) DICP_TAILB(float, MODE_PT2PL); else DICP_TAILB(float, MODE_PT2PT); }
        else                   { if (P.mode == MODE_PT2PL) DICP_TAILB(double, MODE_PT2PL); else DICP_TAILB(double, MODE_PT2PT); }
#undef DICP_TAILB
        return launch_status();
    }
    // the two buffers alternate: after an odd number of iterations the result sits in gpose_tmp (no copy: the caller
    // swaps its two pointers, see dicp_hip.h)
    return 0;
}

int dicp_transform_points(int dtype, const void* src, const void* pose, void* out, int N, int n, void* stream) {
    if (!src || !pose || !out) return DICP_ERR_NULL;
    if (bad_dtype(dtype)) return DICP_ERR_DTYPE;
    if (N <= 0 || n <= 0) return DICP_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    begin_launch();
    const int bpc = dicp_accumulate_blocks(n);
    if (dtype == DICP_F32) transform_kernel<float><<<grid_for(N, bpc), BLOCK, 0, st>>>((const float*)src, (const float*)pose, (float*)out, N, n, bpc);
    else                   transform_kernel<double><<<grid_for(N, bpc), BLOCK, 0, st>>>((const double*)src, (const double*)pose, (double*)out, N, n, bpc);
    return launch_status();
}

int dicp_transform_points_bwd(int dtype, const void* src, const void* pose, const void* gout, void* gsrc, void* partials,
                              int N, int n, void* stream) {
    return transform_points_bwd_go(dtype, src, pose, gout, gsrc, partials, N, n, 0, stream);
}
static int transform_points_bwd_go(int dtype, const void* src, const void* pose, const void* gout, void* gsrc, void* partials, int N, int n, int add, void* stream) {
    if (!src || !pose || !gout || !partials) return DICP_ERR_NULL;
    if (bad_dtype(dtype)) return DICP_ERR_DTYPE;
    if (N <= 0 || n <= 0) return DICP_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    begin_launch();
    const int bpc = dicp_accumulate_blocks(n);
    if (dtype == DICP_F32) transform_bwd_kernel<float><<<grid_for(N, bpc), BLOCK, 0, st>>>((const float*)src, (const float*)pose, (const float*)gout, (float*)gsrc, (float*)partials, N, n, bpc, add);
    else                   transform_bwd_kernel<double><<<grid_for(N, bpc), BLOCK, 0, st>>>((const double*)src, (const double*)pose, (const double*)gout, (double*)gsrc, (double*)partials, N, n, bpc, add);
    return launch_status();
}

int dicp_loss_weight(int dtype, int loss, int differentiable, double metric, double tanh_k,
                     const void* err, int64_t rows, int r, void* w, void* stream) {
    if (!err || !w) return DICP_ERR_NULL;
    if (bad_dtype(dtype)) return DICP_ERR_DTYPE;
    if (loss < DICP_LOSS_HUBER || loss > DICP_LOSS_TRIM) return DICP_ERR_ENUM;    // loss.py:19
    if (rows <= 0 || r < 1 || r > 3) return DICP_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    begin_launch();
    const unsigned g = blocks_for((size_t)rows);
    if (dtype == DICP_F32) loss_weight_kernel<float><<<g, BLOCK, 0, st>>>(loss, differentiable, (float)metric, (float)tanh_k, (const float*)err, (long)rows, r, (float*)w);
    else                   loss_weight_kernel<double><<<g, BLOCK, 0, st>>>(loss, differentiable, metric, tanh_k, (const double*)err, (long)rows, r, (double*)w);
    return launch_status();
}

int dicp_loss_weight_bwd(int dtype, int loss, int differentiable, double metric, double tanh_k,
                         const void* err, const void* gw, int64_t rows, int r, void* gerr, void* stream) {
    if (!err || !gw || !gerr) return DICP_ERR_NULL;
    if (bad_dtype(dtype)) return DICP_ERR_DTYPE;
    if (loss < DICP_LOSS_HUBER || loss > DICP_LOSS_TRIM) return DICP_ERR_ENUM;
    if (rows <= 0 || r < 1 || r > 3) return DICP_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    begin_launch();
    const unsigned g = blocks_for((size_t)rows);
    if (dtype == DICP_F32) loss_weight_bwd_kernel<float><<<g, BLOCK, 0, st>>>(loss, differentiable, (float)metric, (float)tanh_k, (const float*)err, (const float*)gw, (long)rows, r, (float*)gerr);
    else                   loss_weight_bwd_kernel<double><<<g, BLOCK, 0, st>>>(loss, differentiable, metric, tanh_k, (const double*)err, (const double*)gw, (long)rows, r, (double*)gerr);
    return launch_status();
}

}  // extern "C"
