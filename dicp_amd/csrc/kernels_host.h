// libdicp_hip.so -- host-side launch helpers (timing events carried on dispatches, parameter conversion, launch configurations).
// Part of the one translation unit dicp_kernels.hip (included inside its anonymous namespace, in this order: kernels_setup.h, kernels_search.h, kernels_setup_sort.h, kernels_rows.h, kernels_accumulate.h, kernels_backward.h, kernels_soft_svd.h, kernels_host.h).
// ------------------------------------------------------------------- host helpers
// Timing events of the loop entry points (dicp_loop_buffers.events): the search and the windowed-backward launches
// carry their pair of events ON the dispatch (hipExtLaunchKernel: start / stop are taken from the kernel's own
// completion signal), where two hipEventRecord calls would put a barrier packet -- about 6 us of idle queue -- on
// either side of every launch they time.  The loop sets the pair, the next such launch of this host thread takes it.
thread_local hipEvent_t tl_launch_start = nullptr, tl_launch_stop = nullptr;
inline void set_launch_events(hipEvent_t a, hipEvent_t b) { tl_launch_start = a; tl_launch_stop = b; }
inline void take_launch_events(hipEvent_t& a, hipEvent_t& b) { a = tl_launch_start; b = tl_launch_stop; tl_launch_start = tl_launch_stop = nullptr; }
inline WeightParams to_params(const dicp_weight_params* p) {
    WeightParams P;
    P.mode = p->mode; P.trim_on = p->trim_on; P.differentiable = p->differentiable; P.loss = p->loss;
    P.trim_dist = p->trim_dist; P.tanh_k = p->tanh_k; P.loss_delta = p->loss_delta; P.match_thresh = p->match_thresh;
    return P;
}
inline unsigned blocks_for(size_t total) { return (unsigned)((total + BLOCK - 1) / BLOCK); }


struct CertAcc {              // what the accumulate of a certified iteration needs (AccCert, untyped)
    int32_t* spos; const int32_t* hist; const int32_t* hist_prev; int32_t* of; int k_floor, k;
    void* nbr; int32_t* gdirty; int32_t* pend; int32_t* cloud; int fresh, units, sets; int32_t* scount;
};

template <typename T, int Q, int CH, int MINW = 1>
void knn_valu_go(const void* src, const void* pose, const void* tgt4, int N, int n, int m, int m_pad, int32_t* idx, Rows rw, hipStream_t st) {
    using T4 = typename V4<T>::type;
    constexpr int TILE = sizeof(T) == 4 ? 2048 : 1024;      // 32 KiB of LDS either way
    const int bpc = (n + BLOCK * Q - 1) / (BLOCK * Q);
    knn_valu_kernel<T, Q, TILE, CH, MINW><<<grid_for(N, bpc), BLOCK, 0, st>>>((const T*)src, (const T*)pose, (const T4*)tgt4, idx, N, n, m, m_pad, bpc, rw.src, rw.tgt);
}

// cfg 0 = pick by problem size: enough blocks to fill 256 CUs first, then register-block queries to
// amortise the LDS broadcasts.  cfg 1.. = fixed (tuning / tests).
template <typename T>
int knn_valu_launch(int cfg, const void* src, const void* pose, const void* tgt4, int N, int n, int m, int m_pad, int32_t* idx, Rows rw, hipStream_t st) {
    const long q_total = (long)N * n;
    if (cfg == 0) {
        if (q_total >= 8L * BLOCK * 1024)      cfg = (sizeof(T) == 4) ? 11 : 3;    // Q=8, 16-target chunks (f32)
        else if (q_total >= 4L * BLOCK * 1024) cfg = (sizeof(T) == 4) ? 5 : 3;     // Q=4
        else if (q_total >= 2L * BLOCK * 1024) cfg = 2;
        else                                   cfg = 1;
    }
    switch (cfg) {
        case 1: knn_valu_go<T, 1, 8>(src, pose, tgt4, N, n, m, m_pad, idx, rw, st); break;
        case 2: knn_valu_go<T, 2, 8>(src, pose, tgt4, N, n, m, m_pad, idx, rw, st); break;
        case 3: knn_valu_go<T, 4, 8>(src, pose, tgt4, N, n, m, m_pad, idx, rw, st); break;
        case 5: knn_valu_go<T, 4, 16>(src, pose, tgt4, N, n, m, m_pad, idx, rw, st); break;
        case 11: knn_valu_go<T, 8, 16, (sizeof(T) == 4 ? 4 : 1)>(src, pose, tgt4, N, n, m, m_pad, idx, rw, st); break;
        default: return DICP_ERR_ENUM;
    }
    return launch_status();
}
