// libdicp_hip.so -- Gumbel-softmax soft correspondences, Kabsch / SVD point-to-point step, transform points, loss weights, pose gradient in / out.
// Part of the one translation unit dicp_kernels.hip (included inside its anonymous namespace, in this order: kernels_setup.h, kernels_search.h, kernels_setup_sort.h, kernels_rows.h, kernels_accumulate.h, kernels_backward.h, kernels_soft_svd.h, kernels_host.h).
// ------------------------------------------------------------- Gumbel-softmax soft kNN
// nn.__diff_nn_gumbel (nn.py:43-70): out_i = sum_j softmax_j((-|x_i - y_j|^2 + g_ij) / tau) * y_j with
// g = -log(-log(U + eps) + eps).  The reference materialises (N,n,m) distances, noise and probabilities; here
// the targets stream through LDS and each lane keeps an ONLINE softmax (running max, sum, weighted row) for
// its query.  Noise is either an injected U (N,n,m) -- what the parity tests use -- or generated in-kernel
// from a counter-based hash of (seed, cloud, i, j), so the backward passes can regenerate it instead of
// storing it.  Backward recomputes the probabilities from the saved log-sum-exp in two passes: one lane per
// query (x-bar) and one lane per target (y-bar, no atomics).
__device__ __forceinline__ unsigned mix32(unsigned v) {
    v ^= v >> 16; v *= 0x7feb352du; v ^= v >> 15; v *= 0x846ca68bu; v ^= v >> 16;
    return v;
}
template <typename T>
__device__ __forceinline__ T gumbel_uniform(const T* __restrict__ U, size_t off, unsigned key_bi, unsigned j) {
    if (U) return U[off];
    return T(mix32(key_bi ^ (j * 0xC2B2AE35u + 0x27D4EB2Fu)) >> 8) * T(1.0 / 16777216.0);      // [0,1) like torch.rand
}
__device__ __forceinline__ float  log_t(float v)  { return __logf(v); }
__device__ __forceinline__ double log_t(double v) { return log(v); }
__device__ __forceinline__ float  exp_t(float v)  { return __expf(v); }
__device__ __forceinline__ double exp_t(double v) { return exp(v); }

template <typename T>
__device__ __forceinline__ T gumbel_logit(const T* x, const T* y, T u, T eps, T inv_tau) {
    const T d0 = x[0] - y[0], d1 = x[1] - y[1], d2 = x[2] - y[2];
    const T g = -log_t(-log_t(u + eps) + eps);                                   // nn.py:62
    return (g - (d0 * d0 + d1 * d1 + d2 * d2)) * inv_tau;                         // nn.py:56-64
}

constexpr int GUM_TILE = 512;

template <typename T, int C>
__global__ __launch_bounds__(BLOCK) void gumbel_fwd_kernel(const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ U,
                                                           unsigned seed, T eps, T inv_tau, T* __restrict__ out, T* __restrict__ lse,
                                                           int N, int n, int m, int bpc) {
    __shared__ T ty[GUM_TILE * C];
    int cloud, blk;
    if (!decode_block(bpc, N, cloud, blk)) return;
    const int tid = threadIdx.x, i = blk * BLOCK + tid;
    const bool on = i < n;
    T xi[3] = {T(0), T(0), T(0)};
    if (on) { const T* xp = x + ((size_t)cloud * n + i) * 3; xi[0] = xp[0]; xi[1] = xp[1]; xi[2] = xp[2]; }
    const unsigned key = mix32(mix32(seed ^ ((unsigned)cloud * 0x9E3779B9u)) ^ ((unsigned)i * 0x85EBCA6Bu));
    const size_t urow = ((size_t)cloud * n + (on ? i : 0)) * m;
    T M = -inf_v<T>(), S = T(0), acc[C];
#pragma unroll
    for (int k = 0; k < C; ++k) acc[k] = T(0);
    const T* __restrict__ yc = y + (size_t)cloud * m * C;
    for (int base = 0; base < m; base += GUM_TILE) {
        const int len = min(GUM_TILE, m - base);
        for (int t = tid; t < len * C; t += BLOCK) ty[t] = yc[(size_t)base * C + t];
        __syncthreads();
        for (int j = 0; j < len; ++j) {
            const T* yj = ty + j * C;
            const T l = gumbel_logit(xi, yj, gumbel_uniform(U, urow + base + j, key, (unsigned)(base + j)), eps, inv_tau);
            const T Mn = l > M ? l : M;
            const T sc = exp_t(M - Mn), e = exp_t(l - Mn);                        // M = -inf first time: sc = 0
            S = S * sc + e;
#pragma unroll
            for (int k = 0; k < C; ++k) acc[k] = acc[k] * sc + e * yj[k];
            M = Mn;
        }
        __syncthreads();
    }
    if (on) {
        const T invS = T(1) / S;
        T* op = out + ((size_t)cloud * n + i) * C;
#pragma unroll
        for (int k = 0; k < C; ++k) op[k] = acc[k] * invS;                        // probs @ y, nn.py:65-68
        lse[(size_t)cloud * n + i] = M + log_t(S);
    }
}

// x-bar: one lane per query.
template <typename T, int C>
__global__ __launch_bounds__(BLOCK) void gumbel_bwd_q_kernel(const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ U,
                                                             unsigned seed, T eps, T inv_tau, const T* __restrict__ out,
                                                             const T* __restrict__ lse, const T* __restrict__ gout, T* __restrict__ gx,
                                                             int N, int n, int m, int bpc) {
    __shared__ T ty[GUM_TILE * C];
    int cloud, blk;
    if (!decode_block(bpc, N, cloud, blk)) return;
    const int tid = threadIdx.x, i = blk * BLOCK + tid;
    const bool on = i < n;
    const size_t q = (size_t)cloud * n + (on ? i : 0);
    T xi[3], go[C], D = T(0);
    xi[0] = x[q * 3]; xi[1] = x[q * 3 + 1]; xi[2] = x[q * 3 + 2];
#pragma unroll
    for (int k = 0; k < C; ++k) { go[k] = gout[q * C + k]; D += go[k] * out[q * C + k]; }
    const T L = lse[q];
    const unsigned key = mix32(mix32(seed ^ ((unsigned)cloud * 0x9E3779B9u)) ^ ((unsigned)i * 0x85EBCA6Bu));
    T g[3] = {T(0), T(0), T(0)};
    const T* __restrict__ yc = y + (size_t)cloud * m * C;
    for (int base = 0; base < m; base += GUM_TILE) {
        const int len = min(GUM_TILE, m - base);
        for (int t = tid; t < len * C; t += BLOCK) ty[t] = yc[(size_t)base * C + t];
        __syncthreads();
        for (int j = 0; j < len; ++j) {
            const T* yj = ty + j * C;
            const T l = gumbel_logit(xi, yj, gumbel_uniform(U, q * m + base + j, key, (unsigned)(base + j)), eps, inv_tau);
            const T p = exp_t(l - L);
            T gy = T(0);
#pragma unroll
            for (int k = 0; k < C; ++k) gy += go[k] * yj[k];
            const T dl = p * (gy - D);
            g[0] += dl * (xi[0] - yj[0]); g[1] += dl * (xi[1] - yj[1]); g[2] += dl * (xi[2] - yj[2]);
        }
        __syncthreads();
    }
    if (on) {
        const T f = -T(2) * inv_tau;
        gx[q * 3] = f * g[0]; gx[q * 3 + 1] = f * g[1]; gx[q * 3 + 2] = f * g[2];
    }
}

// y-bar: one lane per target, queries stream through LDS as [x(3), gout(C), lse, D].
template <typename T, int C>
__global__ __launch_bounds__(BLOCK) void gumbel_bwd_t_kernel(const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ U,
                                                             unsigned seed, T eps, T inv_tau, const T* __restrict__ out,
                                                             const T* __restrict__ lse, const T* __restrict__ gout, T* __restrict__ gy,
                                                             int N, int n, int m, int bpc, int add /* 1: gy += (a loop's iterations add up) */) {
    constexpr int R = C + 5;
    __shared__ T tq[GUM_TILE * R];
    int cloud, blk;
    if (!decode_block(bpc, N, cloud, blk)) return;
    const int tid = threadIdx.x, j = blk * BLOCK + tid;
    const bool on = j < m;
    const size_t tj = (size_t)cloud * m + (on ? j : 0);
    T yj[C], g[C];
#pragma unroll
    for (int k = 0; k < C; ++k) { yj[k] = y[tj * C + k]; g[k] = T(0); }
    const unsigned kc = mix32(seed ^ ((unsigned)cloud * 0x9E3779B9u));
    for (int base = 0; base < n; base += GUM_TILE) {
        const int len = min(GUM_TILE, n - base);
        for (int t = tid; t < len; t += BLOCK) {
            const size_t q = (size_t)cloud * n + base + t;
            T* r = tq + t * R;
            r[0] = x[q * 3]; r[1] = x[q * 3 + 1]; r[2] = x[q * 3 + 2];
            T D = T(0);
#pragma unroll
            for (int k = 0; k < C; ++k) { const T v = gout[q * C + k]; r[3 + k] = v; D += v * out[q * C + k]; }
            r[3 + C] = lse[q];
            r[4 + C] = D;
        }
        __syncthreads();
        for (int t = 0; t < len; ++t) {
            const T* r = tq + t * R;
            const int i = base + t;
            const unsigned key = mix32(kc ^ ((unsigned)i * 0x85EBCA6Bu));
            const T l = gumbel_logit(r, yj, gumbel_uniform(U, ((size_t)cloud * n + i) * m + (on ? j : 0), key, (unsigned)j), eps, inv_tau);
            const T p = exp_t(l - r[3 + C]);
            T gd = T(0);
#pragma unroll
            for (int k = 0; k < C; ++k) { gd += r[3 + k] * yj[k]; g[k] += p * r[3 + k]; }
            const T dl = p * (gd - r[4 + C]) * (T(2) * inv_tau);
            g[0] += dl * (r[0] - yj[0]); g[1] += dl * (r[1] - yj[1]); g[2] += dl * (r[2] - yj[2]);
        }
        __syncthreads();
    }
    if (on) {
#pragma unroll
        for (int k = 0; k < C; ++k) gy[tj * C + k] = add ? gy[tj * C + k] + g[k] : g[k];
    }
}

// ------------------------------------------------------------------ Kabsch / SVD path
// Point-to-point alignment in closed form (the step of the reference's pt2pt_dICP_SVD, ICP.py:533-591),
// batched and weighted.  accumulate: 18 sums per cloud; step: 3x3 SVD per cloud; bwd: one pass.
template <typename T>
__device__ __forceinline__ T kabsch_weight(const T* C, const T* r, const T* p, const T* y, T w0, int trim_on, T trim_dist) {
    if (!trim_on) return w0;
    T q[3];
    matvec3(C, p, q);
    const T e[3] = {q[0] + r[0] - y[0], q[1] + r[1] - y[1], q[2] + r[2] - y[2]};
    return (m_sqrt(dot3(e, e)) < trim_dist) ? w0 : T(0);      // hard gate on the CURRENT residual (not differentiated)
}

template <typename T>
__global__ __launch_bounds__(BLOCK) void kabsch_accumulate_kernel(const T* __restrict__ src, const T* __restrict__ tgt, int c,
                                                                  const int32_t* __restrict__ idx, const T* __restrict__ pose,
                                                                  const T* __restrict__ w_init, int trim_on, T trim_dist,
                                                                  int N, int n, int m, int bpc, T* __restrict__ partials, const int32_t* __restrict__ src_rows) {
    __shared__ T red[(BLOCK / WAVE) * NACC_PAD];
    int cloud, blk;
    if (!decode_block(bpc, N, cloud, blk)) return;
    T C[9], r[3];
    load_pose(pose, cloud, C, r);
    T acc[NKAB];
#pragma unroll
    for (int k = 0; k < NKAB; ++k) acc[k] = T(0);
    const int end = min(rows_of(src_rows, cloud, n), (blk + 1) * ACC_PTS);
    for (int i = blk * ACC_PTS + threadIdx.x; i < end; i += BLOCK) {
        const size_t pt = (size_t)cloud * n + i;
        const T* sp = src + pt * 3;
        const T p[3] = {sp[0], sp[1], sp[2]};
        const int j = idx ? min(max(idx[pt], 0), m - 1) : i;     // idx == NULL: tgt holds one row per source point
        const T* yp = tgt + ((size_t)cloud * m + j) * c;
        const T y[3] = {yp[0], yp[1], yp[2]};
        const T w = kabsch_weight(C, r, p, y, w_init[pt], trim_on, trim_dist);
        acc[KAB_S0] += w;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            acc[KAB_SP + a] += w * p[a];
            acc[KAB_SY + a] += w * y[a];
#pragma unroll
            for (int b = 0; b < 3; ++b) acc[KAB_M + a * 3 + b] += w * y[a] * p[b];
        }
        acc[KAB_PP] += w * dot3(p, p);
        acc[KAB_YY] += w * dot3(y, y);
    }
    block_reduce_store<T, NKAB, NACC_PAD>(acc, partials + ((size_t)cloud * bpc + blk) * NACC_PAD, red);
}

template <typename T>
__global__ __launch_bounds__(WAVE) void kabsch_step_kernel(const T* __restrict__ partials, int nblk, T* __restrict__ pose_out,
                                                           T* __restrict__ cost, double* __restrict__ save, int N) {
    __shared__ double sacc[NACC_PAD], ssave[KAB_SAVE], sC[9], sr[3];
    __shared__ double scost;
    const int cloud = blockIdx.x, tid = threadIdx.x;
    {
        const int slot_i = tid & 31, part = tid >> 5;
        const T* pp = partials + (size_t)cloud * nblk * NACC_PAD + slot_i;
        double s = 0.0;
        for (int b = part; b < nblk; b += 2) s += (double)pp[(size_t)b * NACC_PAD];
        s += __shfl_down(s, 32);
        if (tid < NACC_PAD) sacc[tid] = s;
    }
    __syncthreads();
    if (tid == 0) scost = kabsch_forward(sacc, sC, sr, ssave);
    __syncthreads();
    if (tid < 9) pose_out[(size_t)cloud * 12 + tid] = (T)sC[tid];
    if (tid < 3) pose_out[(size_t)cloud * 12 + 9 + tid] = (T)sr[tid];
    if (tid < KAB_SAVE && save) save[(size_t)cloud * KAB_SAVE + tid] = ssave[tid];
    if (tid == 0 && cost) cost[cloud] = (T)scost;
}

// The step of the fused loop (dicp_kabsch_forward): as kabsch_step_kernel, plus the loop's bookkeeping on device.  A cloud whose
// cost falls below the tolerance is FROZEN at that pose (rows_live = 0: the searches and sums of later iterations skip it, its
// matches / pose / SVD of the last active iteration stay for the backward) -- every pair stops where a call of its own would
// (ICP.py:585-586), and the iterations the host enqueues past that point before it notices are no-ops.
template <typename T>
__global__ __launch_bounds__(WAVE) void kabsch_loop_step_kernel(const T* __restrict__ partials, int nblk, T* __restrict__ pose, T* __restrict__ pose_search,
                                                                T* __restrict__ pose_used, const T* __restrict__ frame, T* __restrict__ costs, long cost_stride,
                                                                int k, double* __restrict__ save, int32_t* __restrict__ rows_live, T* __restrict__ iterations,
                                                                int const_iter, double tolerance, int32_t* __restrict__ counters) {
    __shared__ double sacc[NACC_PAD], ssave[KAB_SAVE], sC[9], sr[3];
    __shared__ double scost;
    const int cloud = blockIdx.x, tid = threadIdx.x;
    T* cst = costs + (size_t)cloud * cost_stride;
    if (rows_live[cloud] <= 0) {                            // frozen (or empty): the history repeats its last entry
        if (tid == 0) cst[k] = k > 0 ? cst[k - 1] : T(0);
        return;
    }
    {
        const int slot_i = tid & 31, part = tid >> 5;
        const T* pp = partials + (size_t)cloud * nblk * NACC_PAD + slot_i;
        double s = 0.0;
        for (int b = part; b < nblk; b += 2) s += (double)pp[(size_t)b * NACC_PAD];
        s += __shfl_down(s, 32);
        if (tid < NACC_PAD) sacc[tid] = s;
    }
    __syncthreads();
    if (tid == 0) scost = kabsch_forward(sacc, sC, sr, ssave);
    __syncthreads();
    T* ps = pose + (size_t)cloud * 12;
    if (tid < 12) pose_used[(size_t)cloud * 12 + tid] = ps[tid];        // the pose the matches were found under (what the backward re-derives the gate from)
    __syncthreads();
    if (tid < 12) {
        const T v = tid < 9 ? (T)sC[tid] : (T)sr[tid - 9];
        ps[tid] = v;
        if (pose_search) {
            T pw[12];
#pragma unroll
            for (int e = 0; e < 12; ++e) pw[e] = e < 9 ? (T)sC[e] : (T)sr[e - 9];
            pose_search[(size_t)cloud * 12 + tid] = frame_pose_entry<T>(frame ? frame + (size_t)cloud * 12 : nullptr, pw, tid);
        }
    }
    if (tid < KAB_SAVE) save[(size_t)cloud * KAB_SAVE + tid] = ssave[tid];
    if (tid == 0) {
        cst[k] = (T)scost;
        if (!const_iter && (double)(T)scost < tolerance) {              // ICP.py:585-586
            iterations[cloud] = (T)(k + 1);
            rows_live[cloud] = 0;
        } else if (counters) atomicAdd(counters + k, 1);
    }
}

template <typename T>
__global__ __launch_bounds__(WAVE) void kabsch_step_bwd_kernel(const T* __restrict__ gpose, const double* __restrict__ save,
                                                               T* __restrict__ gacc, int N) {
    __shared__ double sg[12], ssave[KAB_SAVE], sout[16];
    const int cloud = blockIdx.x, tid = threadIdx.x;
    if (tid < 12) sg[tid] = (double)gpose[(size_t)cloud * 12 + tid];
    if (tid < KAB_SAVE) ssave[tid] = save[(size_t)cloud * KAB_SAVE + tid];
    __syncthreads();
    if (tid == 0) kabsch_backward(sg, sg + 9, ssave, sout);
    __syncthreads();
    if (tid < 16) gacc[(size_t)cloud * 16 + tid] = (T)sout[tid];
}

template <typename T>
__global__ __launch_bounds__(BLOCK) void kabsch_bwd_kernel(const T* __restrict__ src, const T* __restrict__ tgt, int c,
                                                           const int32_t* __restrict__ idx, const T* __restrict__ pose,
                                                           const T* __restrict__ w_init, int trim_on, T trim_dist,
                                                           const T* __restrict__ gacc, int N, int n, int m, int bpc,
                                                           T* __restrict__ gsrc, T* __restrict__ gtgt, T* __restrict__ gw, const int32_t* __restrict__ src_rows) {
    int cloud, blk;
    if (!decode_block(bpc, N, cloud, blk)) return;
    T C[9], r[3], g[16];
    load_pose(pose, cloud, C, r);
#pragma unroll
    for (int k = 0; k < 16; ++k) g[k] = gacc[(size_t)cloud * 16 + k];
    const int end = min(rows_of(src_rows, cloud, n), (blk + 1) * ACC_PTS);
    for (int i = blk * ACC_PTS + threadIdx.x; i < end; i += BLOCK) {
        const size_t pt = (size_t)cloud * n + i;
        const T* sp = src + pt * 3;
        const T p[3] = {sp[0], sp[1], sp[2]};
        const int j = idx ? min(max(idx[pt], 0), m - 1) : i;     // idx == NULL: tgt holds one row per source point
        const size_t row = ((size_t)cloud * m + j) * c;
        const T y[3] = {tgt[row], tgt[row + 1], tgt[row + 2]};
        const T w0 = w_init[pt];
        const T w = kabsch_weight(C, r, p, y, w0, trim_on, trim_dist);
        T yMp = T(0);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            T gp = g[KAB_SP + a], gy = g[KAB_SY + a];
#pragma unroll
            for (int b = 0; b < 3; ++b) { gp += g[KAB_M + b * 3 + a] * y[b]; gy += g[KAB_M + a * 3 + b] * p[b]; yMp += y[a] * g[KAB_M + a * 3 + b] * p[b]; }
            gsrc[pt * 3 + a] += w * gp;
            if (gtgt) unsafeAtomicAdd(&gtgt[row + a], w * gy);
        }
        if (gw) gw[pt] += (w == w0 ? T(1) : T(0)) * (g[KAB_S0] + dot3(g + KAB_SP, p) + dot3(g + KAB_SY, y) + yMp);
    }
}

// ------------------------------------------------------------------ transform points
// pc = C p + r for every point (the returned cloud, ICP.py:274) and its adjoint.  A (N,n,3)x(3,3) bmm through a
// BLAS library costs 5x the time of streaming the 24 bytes per point.
template <typename T>
__global__ __launch_bounds__(BLOCK) void transform_kernel(const T* __restrict__ src, const T* __restrict__ pose, T* __restrict__ out,
                                                          int N, int n, int bpc) {
    int cloud, blk;
    if (!decode_block(bpc, N, cloud, blk)) return;
    T C[9], r[3];
    load_pose(pose, cloud, C, r);
    const int end = min(n, (blk + 1) * ACC_PTS);
    for (int i = blk * ACC_PTS + threadIdx.x; i < end; i += BLOCK) {
        const size_t pt = ((size_t)cloud * n + i) * 3;
        const T p[3] = {src[pt], src[pt + 1], src[pt + 2]};
        T q[3];
        matvec3(C, p, q);
        out[pt] = q[0] + r[0]; out[pt + 1] = q[1] + r[1]; out[pt + 2] = q[2] + r[2];
    }
}

template <typename T>
__global__ __launch_bounds__(BLOCK) void transform_bwd_kernel(const T* __restrict__ src, const T* __restrict__ pose, const T* __restrict__ gout,
                                                              T* __restrict__ gsrc, T* __restrict__ partials, int N, int n, int bpc,
                                                              int add /* 1: gsrc and partials are added to */) {
    __shared__ T red[(BLOCK / WAVE) * NBWD_PAD];
    __shared__ T sums[NBWD_PAD];
    int cloud, blk;
    if (!decode_block(bpc, N, cloud, blk)) return;
    T C[9], r[3];
    load_pose(pose, cloud, C, r);
    T acc[NBWD];
#pragma unroll
    for (int k = 0; k < NBWD; ++k) acc[k] = T(0);
    const int end = min(n, (blk + 1) * ACC_PTS);
    for (int i = blk * ACC_PTS + threadIdx.x; i < end; i += BLOCK) {
        const size_t pt = ((size_t)cloud * n + i) * 3;
        const T p[3] = {src[pt], src[pt + 1], src[pt + 2]};
        const T g[3] = {gout[pt], gout[pt + 1], gout[pt + 2]};
        if (gsrc) {
            const T v0 = C[0] * g[0] + C[3] * g[1] + C[6] * g[2], v1 = C[1] * g[0] + C[4] * g[1] + C[7] * g[2], v2 = C[2] * g[0] + C[5] * g[1] + C[8] * g[2];
            gsrc[pt]     = add ? gsrc[pt] + v0 : v0;
            gsrc[pt + 1] = add ? gsrc[pt + 1] + v1 : v1;
            gsrc[pt + 2] = add ? gsrc[pt + 2] + v2 : v2;
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
#pragma unroll
            for (int b = 0; b < 3; ++b) acc[a * 3 + b] += g[a] * p[b];
            acc[9 + a] += g[a];
        }
    }
    T* out = partials + ((size_t)cloud * bpc + blk) * NBWD_PAD;
    if (!add) { block_reduce_store<T, NBWD, NBWD_PAD>(acc, out, red); return; }
    block_reduce_store<T, NBWD, NBWD_PAD>(acc, sums, red);
    __syncthreads();
    if (threadIdx.x < NBWD_PAD) out[threadIdx.x] += sums[threadIdx.x];
}

// ------------------------------------------------------------------ loss weights
template <typename T>
__device__ __forceinline__ void loss_eval(int loss, int diff, T metric, T kk, const T* e, int r, T& w, T& en, T& th) {
    T s = T(0);
    for (int k = 0; k < r; ++k) s += e[k] * e[k];
    en = m_sqrt(s);
    th = T(0);
    if (loss == DICP_LOSS_HUBER) {
        if (diff) w = (metric * metric) / (metric * metric + en * en);
        else      w = (en > metric) ? metric / en : T(1);
    } else if (loss == DICP_LOSS_CAUCHY) {
        const T t = en / metric;
        w = T(1) / (T(1) + t * t);
    } else {   // trim
        if (diff) { th = m_tanh(kk * (metric - en) - T(3)); w = T(0.5) * th + T(0.5); }
        else      w = (en < metric) ? T(1) : T(0);
    }
}

template <typename T>
__global__ __launch_bounds__(BLOCK) void loss_weight_kernel(int loss, int diff, T metric, T kk, const T* __restrict__ err,
                                                            long rows, int r, T* __restrict__ w) {
    const long i = (long)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= rows) return;
    T e[3] = {T(0), T(0), T(0)};
    for (int k = 0; k < r; ++k) e[k] = err[i * r + k];
    T wv, en, th;
    loss_eval(loss, diff, metric, kk, e, r, wv, en, th);
    w[i] = wv;
}

template <typename T>
__global__ __launch_bounds__(BLOCK) void loss_weight_bwd_kernel(int loss, int diff, T metric, T kk, const T* __restrict__ err,
                                                                const T* __restrict__ gw, long rows, int r, T* __restrict__ gerr) {
    const long i = (long)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= rows) return;
    T e[3] = {T(0), T(0), T(0)};
    for (int k = 0; k < r; ++k) e[k] = err[i * r + k];
    T wv, en, th;
    loss_eval(loss, diff, metric, kk, e, r, wv, en, th);
    T dw = T(0);      // d w / d en
    if (loss == DICP_LOSS_HUBER) {
        if (diff) dw = -T(2) * en * wv * wv / (metric * metric);
        else      dw = hard_huber_slope(en, metric);
    } else if (loss == DICP_LOSS_CAUCHY) {
        dw = -T(2) * en * wv * wv / (metric * metric);
    } else if (diff) {
        dw = -T(0.5) * kk * (T(1) - th * th);
    }
    // torch's norm backward is e/|e| with 0 at e == 0; a NaN slope (hard huber at 0) still propagates
    for (int k = 0; k < r; ++k) gerr[i * r + k] = (en > T(0)) ? gw[i] * dw * e[k] / en : gw[i] * dw * T(0);
}

// ------------------------------------------------------- pose gradient in / out of the backward loop
// gpose (N,12) double = [dL/dC row-major, dL/dr] from the upstream gradient of T (N,4,4) (NULL: zeros), and back:
// gT0 (N,4,4) = the same layout from the final gpose plus the pose sums of the last accumulate_bwd's partials
// (slots 0..11 of each block's row; summed in block order, in double) -- the head and tail of ICPLoop.backward in
// one launch each instead of a dozen tensor ops.
template <typename T>
__global__ __launch_bounds__(BLOCK) void pose_grad_in_kernel(const T* __restrict__ gT, double* __restrict__ gpose, int N) {
    const int e = blockIdx.x * BLOCK + threadIdx.x;
    if (e >= N * 12) return;
    const int b = e / 12, k = e - b * 12;
    const int row = k < 9 ? k / 3 : k - 9, col = k < 9 ? k - (k / 3) * 3 : 3;
    gpose[e] = gT ? (double)gT[(size_t)b * 16 + row * 4 + col] : 0.0;
}
template <typename T>
__global__ __launch_bounds__(BLOCK) void pose_grad_out_kernel(const double* __restrict__ gpose, const T* __restrict__ bwd_partials, int nblk,
                                                              T* __restrict__ gT0, int N) {
    const int e = blockIdx.x * BLOCK + threadIdx.x;
    if (e >= N * 16) return;
    const int b = e >> 4, row = (e >> 2) & 3, col = e & 3;
    T out = T(0);
    if (row < 3) {
        const int k = col < 3 ? row * 3 + col : 9 + row;
        double v = gpose[(size_t)b * 12 + k];
        if (bwd_partials)
            for (int blk = 0; blk < nblk; ++blk) v += (double)bwd_partials[((size_t)b * nblk + blk) * NBWD_PAD + k];
        out = (T)v;
    }
    gT0[e] = out;
}
