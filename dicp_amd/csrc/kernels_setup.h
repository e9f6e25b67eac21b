// libdicp_hip.so -- per-call set-up: search frame, packed rows, sweep index (LDS key sort, sorted rows, bucket table), query order, loop init / finish.
// Part of the one translation unit dicp_kernels.hip (included inside its anonymous namespace, in this order: kernels_setup.h, kernels_search.h, kernels_setup_sort.h, kernels_rows.h, kernels_accumulate.h, kernels_backward.h, kernels_soft_svd.h, kernels_host.h).
// ------------------------------------------------------------------------- pack
template <typename T> __device__ __forceinline__ T big_v();
template <> __device__ __forceinline__ float  big_v<float>()  { return 3.402823466e+38f; }
template <> __device__ __forceinline__ double big_v<double>() { return 1.7976931348623157e+308; }

// one expression for 0.5|y|^2 wherever a target row is packed, so every kNN form sees bit-identical scores
// ctr (optional): the search runs in coordinates centred on the target cloud, rows are packed as y - ctr (section "centre" below)
// The SEARCH FRAME of a cloud (dicp_search_frame): x' = Q x + t, Q orthonormal (rows: the sort direction first), t = -Q c with c the
// cloud's centre.  F = [Q row-major (9) | t (3)].  Every search form reads only (search pose, packed rows), and both come from the two
// functions below, so a frame changes none of them and they all stay index-for-index identical.  An axis-aligned identity Q is applied as
// the plain subtraction it is: the same bits as the centred search had before frames existed, also for rows with non-finite coordinates
// (0 * inf in the general form would spread a NaN over the row).
template <typename T>
__device__ __forceinline__ bool frame_is_translation(const T* __restrict__ F) {
    return F[0] == T(1) && F[4] == T(1) && F[8] == T(1) && F[1] == T(0) && F[2] == T(0) && F[3] == T(0) && F[5] == T(0) && F[6] == T(0) && F[7] == T(0);
}
template <typename T>
__device__ __forceinline__ void frame_apply(const T* __restrict__ F, const T* y, T* out) {      // out = Q y + t
    if (!F) { out[0] = y[0]; out[1] = y[1]; out[2] = y[2]; return; }
    if (frame_is_translation(F)) { out[0] = y[0] + F[9]; out[1] = y[1] + F[10]; out[2] = y[2] + F[11]; return; }
#pragma unroll
    for (int k = 0; k < 3; ++k) out[k] = fma_t(F[3 * k], y[0], fma_t(F[3 * k + 1], y[1], fma_t(F[3 * k + 2], y[2], F[9 + k])));
}
// the pose a search is handed: [Q C | Q r + t] (entry e of its 12)
template <typename T>
__device__ __forceinline__ T frame_pose_entry(const T* __restrict__ F, const T* pose /* [C row-major | r] */, int e) {
    if (!F) return pose[e];
    if (frame_is_translation(F)) return e < 9 ? pose[e] : pose[e] + F[e];
    if (e < 9) { const int i = e / 3, j = e - 3 * i; return fma_t(F[3 * i], pose[j], fma_t(F[3 * i + 1], pose[3 + j], F[3 * i + 2] * pose[6 + j])); }
    const int i = e - 9;
    return fma_t(F[3 * i], pose[9], fma_t(F[3 * i + 1], pose[10], fma_t(F[3 * i + 2], pose[11], F[9 + i])));
}

template <typename T>
__device__ __forceinline__ typename V4<T>::type pack_row(const T* __restrict__ y, const T* __restrict__ frame = nullptr) {
    typename V4<T>::type v;
    T q[3];
    frame_apply(frame, y, q);
    v.x = q[0]; v.y = q[1]; v.z = q[2];
    v.w = T(0.5) * fma_t(v.z, v.z, fma_t(v.y, v.y, v.x * v.x));     // explicit fmas: no per-kernel contraction choices
    return v;
}

// ------------------------------------------------------------------------ search frame
// The search scores in the expanded form 0.5|y|^2 - x.y, whose rounding error -- and with it the sweep's prune margin -- grows
// with 0.5|x|^2: in a map frame a kilometre from the origin nothing is pruned any more (profiles/r01_offset_clouds.txt).  So the
// search runs in coordinates centred on the target cloud.  And the sorted sweep prunes along ONE direction: a wall perpendicular
// to it puts all of its points into every slab that touches it (planar scenes: 4.9 % of the pairs scored per launch against 1.4 %
// on volumetric clouds, profiles/r03_scene_kernel_stats_before.txt).  So the direction is chosen per cloud as well.  Both are one
// affine map, the cloud's SEARCH FRAME x' = Q x + t (frame_apply): packed rows hold Q y + t, the search kernels are handed the
// pose [Q C | Q r + t] (a second, search-only pose buffer).  Every search form reads only (pose, packed rows), so none of them
// changes and they all stay index-for-index identical.
//   c = the target's median point rounded to a multiple of `quantum`: clouds near the origin get c = 0;
//   Q = the candidate rotation (sort direction = its first row) whose keys spread the cloud's points best: the sum over a
//       256-bin histogram of the projected sample of count^2 -- proportional to the pairs a slab search scores -- is smallest;
//       candidates: the three axes (pure permutations of the coordinates) and three oblique directions no axis-aligned plane
//       is perpendicular to.  The identity keeps the job unless another candidate is 20 % better: volumetric clouds and clouds
//       near the origin get Q = I, t = 0 and with it exactly the bits they had without a frame.
constexpr int CC_THREADS = 1024;     // one block per cloud
constexpr int CC_SAMPLE = CC_THREADS;// rows looked at per cloud: one per thread, its three keys stay in registers
constexpr int SF_DIRS = 6;
__device__ __forceinline__ unsigned sortable_bits(float x);
// The centre only sizes a margin (it decides no result), but it has to sit INSIDE the cloud: a mean would be dragged away by
// stray returns.  So it is the coordinate-wise MEDIAN of a stride sample of at most CC_SAMPLE of the cloud's rows (rows 0, step,
// 2 step, ...; a ragged batch hands over the cloud's own length, so pad rows are not in it), found by a radix select (most
// significant byte first, the three axes side by side) over the order-preserving bit pattern of the float values (float is
// plenty: the centre is rounded to `quantum` anyway).
// Round 6: given the QUERIES too (src under T_init: dicp_sweep_setup has them) the cost of a direction is what it stands for, the rows a slab
// search scores: for a sample of the queries, the sample targets whose key lies within the query's own reach of its key -- the reach being
// the distance to its nearest SAMPLE target (an upper bound of its nearest target's).  The target-only cost above is the special case "every
// query sits on a target": scan pairs that overlap only partly are not that case -- the queries outside the target's footprint reach metres
// far, and along a direction oblique to the footprint's edge their slabs hold a third of the cloud (profiles/r06_search_direction.txt: a
// corridor scanned from two places 6 m apart, 33 % of all pairs scored per search with the target-only choice, 9 % along the corridor).
constexpr int SF_QUERIES = CC_THREADS / 4;     // sample queries per cloud: four threads each share the scan of the sample targets
template <typename T>
__global__ __launch_bounds__(CC_THREADS) void search_frame_kernel(const T* __restrict__ tgt, int c, int m, const int32_t* __restrict__ tgt_rows,
                                                                  double quantum, int directions, T* __restrict__ frame,
                                                                  const T* __restrict__ src, int n, const int32_t* __restrict__ src_rows, const T* __restrict__ T_init) {
    // rotations with det +1; row 0 = the sort direction.  0: identity, 1 / 2: y / z first (cyclic permutations), 3..5: oblique
    const double QS[SF_DIRS][9] = {
        {1, 0, 0, 0, 1, 0, 0, 0, 1}, {0, 1, 0, 0, 0, 1, 1, 0, 0}, {0, 0, 1, 1, 0, 0, 0, 1, 0},
        {0.6, 0.64, 0.48, 0.72953720414008516, -0.68394112888132985, 0, 0.32829174186303833, 0.35017785798724088, -0.87726848797845247},
        {0.6, -0.64, 0.48, -0.72953720414008516, -0.68394112888132985, 0, 0.32829174186303833, -0.35017785798724088, -0.87726848797845247},
        {0.48, 0.6, -0.64, 0, -0.72953720414008516, -0.68394112888132985, -0.87726848797845247, 0.32829174186303833, -0.35017785798724088}};
    __shared__ int hist[3][256];
    __shared__ unsigned sel_prefix[3];
    __shared__ int sel_want[3];
    __shared__ int dhist[SF_DIRS][256];
    __shared__ float s_ctr[3], s_ext[CC_THREADS / WAVE], s_cnt[CC_THREADS / WAVE];
    __shared__ int s_cost[SF_DIRS];
    __shared__ float4 s_t[CC_SAMPLE];            // the sample targets, centred (query-aware cost)
    __shared__ int s_cum[SF_DIRS][256 + 1];      // exclusive prefix sums of dhist
    const int cloud = blockIdx.x, tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid >> 6;
    const T* __restrict__ rows = tgt + (size_t)cloud * m * c;
    const int mc = max(rows_of(tgt_rows, cloud, m), 1);
    const int step = (mc + CC_SAMPLE - 1) / CC_SAMPLE, ms = (mc + step - 1) / step;
    const bool on = tid < ms;
    unsigned key[3] = {0u, 0u, 0u};
    float pt[3] = {0.f, 0.f, 0.f};
    if (on) {
        const T* r = rows + (size_t)tid * step * c;
        pt[0] = (float)r[0]; pt[1] = (float)r[1]; pt[2] = (float)r[2];
        key[0] = sortable_bits(pt[0]); key[1] = sortable_bits(pt[1]); key[2] = sortable_bits(pt[2]);
    }
    if (tid < 3) { sel_prefix[tid] = 0u; sel_want[tid] = (ms - 1) / 2; }    // lower median
    unsigned mask = 0u;
    for (int pass = 3; pass >= 0; --pass) {
        for (int d = tid; d < 3 * 256; d += CC_THREADS) (&hist[0][0])[d] = 0;
        __syncthreads();
        if (on) {
#pragma unroll
            for (int a = 0; a < 3; ++a)
                if ((key[a] & mask) == sel_prefix[a]) atomicAdd(&hist[a][(key[a] >> (8 * pass)) & 255u], 1);
        }
        __syncthreads();
        if (wave < 3) {                                                     // wave a selects axis a's byte: 4 bins per lane
            const int a = wave, want = sel_want[a];
            const int h0 = hist[a][4 * lane], h1 = hist[a][4 * lane + 1], h2 = hist[a][4 * lane + 2], h3 = hist[a][4 * lane + 3];
            int incl = h0 + h1 + h2 + h3;
            const int own = incl;
#pragma unroll
            for (int off = 1; off < WAVE; off <<= 1) { const int o = __shfl_up(incl, off); if (lane >= off) incl += o; }
            const unsigned long long over = __ballot(incl > want);         // first lane whose running count passes `want`
            const int L = over ? __ffsll((long long)over) - 1 : WAVE - 1;
            if (lane == L) {
                int w = want - (incl - own), d = 0;
                if (w >= h0) { w -= h0; d = 1; if (w >= h1) { w -= h1; d = 2; if (w >= h2) { w -= h2; d = 3; } } }
                sel_want[a] = w;
                sel_prefix[a] |= (unsigned)(4 * lane + d) << (8 * pass);
            }
        }
        mask |= 0xffu << (8 * pass);
        __syncthreads();
    }
    if (tid < 3) {
        unsigned u = sel_prefix[tid];
        u ^= (u >> 31) ? 0x80000000u : 0xffffffffu;                         // inverse of sortable_bits
        double v = (double)__uint_as_float(u);
        v = quantum > 0.0 ? rint(v / quantum) * quantum : v;
        s_ctr[tid] = (v == v && fabs(v) < 1e30) ? (float)(T)v : 0.f;        // non-finite input: no centring
    }
    for (int d = tid; d < SF_DIRS * 256; d += CC_THREADS) (&dhist[0][0])[d] = 0;
    __syncthreads();
    // ---- the sort direction: histograms of the sample's keys along every candidate, one bin width for all of them
    const float dx = pt[0] - s_ctr[0], dy = pt[1] - s_ctr[1], dz = pt[2] - s_ctr[2];
    const bool fin = on && fabsf(dx) < 1e30f && fabsf(dy) < 1e30f && fabsf(dz) < 1e30f;
    // the histograms' span: the sample's extent about the centre -- of the points within 32x the MEAN deviation (round 6: one stray return, or
    // the far pad row a ragged batch's cloud carries (max(source) * 1000: a thousand times the cloud), stretched the span until every real
    // point shared a bin or two and all directions cost the same; such points now land in the edge bins)
    const float dev = fin ? fmaxf(fabsf(dx), fmaxf(fabsf(dy), fabsf(dz))) : 0.f;
    float dsum = dev, dcnt = fin ? 1.f : 0.f;
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) { dsum += __shfl_xor(dsum, off); dcnt += __shfl_xor(dcnt, off); }
    if (lane == 0) { s_ext[wave] = dsum; s_cnt[wave] = dcnt; }
    __syncthreads();
    float mean_dev = 0.f, cnt_all = 0.f;
    for (int w = 0; w < CC_THREADS / WAVE; ++w) { mean_dev += s_ext[w]; cnt_all += s_cnt[w]; }
    mean_dev = cnt_all > 0.f ? mean_dev / cnt_all : 0.f;
    __syncthreads();
    float ext = (fin && dev <= 32.f * mean_dev) ? dev : 0.f;
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) ext = fmaxf(ext, __shfl_xor(ext, off));
    if (lane == 0) s_ext[wave] = ext;
    __syncthreads();
    float R2 = 0.f;
    for (int w = 0; w < CC_THREADS / WAVE; ++w) R2 = fmaxf(R2, s_ext[w]);
    R2 *= 1.7321f;                                                          // |d . (p - c)| <= sqrt(3) max |p - c|_inf
    int best = 0;
    if (directions && R2 > 0.f) {
        if (fin) {
            const float scale = 128.f / R2;
#pragma unroll
            for (int j = 0; j < SF_DIRS; ++j) {
                const float k = (float)QS[j][0] * dx + (float)QS[j][1] * dy + (float)QS[j][2] * dz;
                const int bin = min(max((int)((k + R2) * scale), 0), 255);
                atomicAdd(&dhist[j][bin], 1);
            }
        }
        const bool with_queries = src != nullptr && T_init != nullptr && n > 0;
        if (with_queries) s_t[tid] = fin ? make_float4(dx, dy, dz, 0.f) : make_float4(1e18f, 1e18f, 1e18f, 0.f);
        __syncthreads();
        if (wave < SF_DIRS) {
            const int h0 = dhist[wave][4 * lane], h1 = dhist[wave][4 * lane + 1], h2 = dhist[wave][4 * lane + 2], h3 = dhist[wave][4 * lane + 3];
            int cst = h0 * h0 + h1 * h1 + h2 * h2 + h3 * h3;
#pragma unroll
            for (int off = WAVE / 2; off > 0; off >>= 1) cst += __shfl_xor(cst, off);
            if (lane == 0) s_cost[wave] = with_queries ? 0 : cst;
            if (with_queries) {
                int incl = h0 + h1 + h2 + h3;
                const int own = incl;
#pragma unroll
                for (int off = 1; off < WAVE; off <<= 1) { const int o = __shfl_up(incl, off); if (lane >= off) incl += o; }
                const int ex = incl - own;
                s_cum[wave][4 * lane] = ex; s_cum[wave][4 * lane + 1] = ex + h0; s_cum[wave][4 * lane + 2] = ex + h0 + h1; s_cum[wave][4 * lane + 3] = ex + h0 + h1 + h2;
                if (lane == WAVE - 1) s_cum[wave][256] = incl;
            }
        }
        __syncthreads();
        if (with_queries) {
            // sample query tid / 4 under T_init, centred like the targets; its four threads share the sample targets
            const int nc = rows_of(src_rows, cloud, n);
            const int qstep = max((nc + SF_QUERIES - 1) / SF_QUERIES, 1), nq = (nc + qstep - 1) / qstep;
            const int qid = tid >> 2, part = tid & 3;
            bool qon = qid < nq;
            float qx = 0.f, qy = 0.f, qz = 0.f;
            if (qon) {
                const T* p = src + ((size_t)cloud * n + (size_t)qid * qstep) * 3;
                const T* M = T_init + (size_t)cloud * 16;
                const float p0 = (float)p[0], p1 = (float)p[1], p2 = (float)p[2];
                qx = (float)M[0] * p0 + (float)M[1] * p1 + (float)M[2] * p2 + (float)M[3] - s_ctr[0];
                qy = (float)M[4] * p0 + (float)M[5] * p1 + (float)M[6] * p2 + (float)M[7] - s_ctr[1];
                qz = (float)M[8] * p0 + (float)M[9] * p1 + (float)M[10] * p2 + (float)M[11] - s_ctr[2];
                qon = fabsf(qx) < 1e18f && fabsf(qy) < 1e18f && fabsf(qz) < 1e18f;
            }
            float d2 = 3e38f;
            if (qon) {      // (four targets per round, one 16-byte LDS read each: every sample slot past ms holds the far filler or is never reached)
                float da = 3e38f, db = 3e38f, dc = 3e38f, dd = 3e38f;
                int t = part;
                for (; t + 12 < ms; t += 16) {
                    const float4 a = s_t[t], b = s_t[t + 4], c4 = s_t[t + 8], e4 = s_t[t + 12];
                    da = fminf(da, (a.x - qx) * (a.x - qx) + (a.y - qy) * (a.y - qy) + (a.z - qz) * (a.z - qz));
                    db = fminf(db, (b.x - qx) * (b.x - qx) + (b.y - qy) * (b.y - qy) + (b.z - qz) * (b.z - qz));
                    dc = fminf(dc, (c4.x - qx) * (c4.x - qx) + (c4.y - qy) * (c4.y - qy) + (c4.z - qz) * (c4.z - qz));
                    dd = fminf(dd, (e4.x - qx) * (e4.x - qx) + (e4.y - qy) * (e4.y - qy) + (e4.z - qz) * (e4.z - qz));
                }
                for (; t < ms; t += 4) { const float4 a = s_t[t]; da = fminf(da, (a.x - qx) * (a.x - qx) + (a.y - qy) * (a.y - qy) + (a.z - qz) * (a.z - qz)); }
                d2 = fminf(fminf(da, db), fminf(dc, dd));
            }
            d2 = fminf(d2, __shfl_xor(d2, 1));
            d2 = fminf(d2, __shfl_xor(d2, 2));
            if (qon && d2 < 1e30f) {
                const float d = sqrtf(d2), scale = 128.f / R2;
#pragma unroll
                for (int j = 0; j < SF_DIRS; ++j) {             // (constant indices: the candidates' table stays in registers / literals)
                    if ((j & 3) != part) continue;
                    const float k = (float)QS[j][0] * qx + (float)QS[j][1] * qy + (float)QS[j][2] * qz;
                    const int lo = min(max((int)((k - d + R2) * scale), 0), 255), hi = min(max((int)((k + d + R2) * scale), 0), 255);
                    atomicAdd(&s_cost[j], s_cum[j][hi + 1] - s_cum[j][lo]);
                }
            }
            __syncthreads();
        }
        for (int j = 1; j < SF_DIRS; ++j) if (s_cost[j] < s_cost[best]) best = j;
        if (!(5 * (long long)s_cost[best] < 4 * (long long)s_cost[0])) best = 0;    // the identity keeps the job unless another is 20 % better
    }
    if (tid < 12) {
        T* F = frame + (size_t)cloud * 12;
        if (tid < 9) F[tid] = (T)QS[best][tid];
        else {          // t = -Q c, in T arithmetic (for Q = I: exactly -c)
            const int i = tid - 9;
            const T cx = (T)s_ctr[0], cy = (T)s_ctr[1], cz = (T)s_ctr[2];
            F[tid] = best == 0 ? -(i == 0 ? cx : (i == 1 ? cy : cz))
                               : -fma_t((T)QS[best][3 * i], cx, fma_t((T)QS[best][3 * i + 1], cy, (T)QS[best][3 * i + 2] * cz));
        }
    }
}

template <typename T>
__global__ __launch_bounds__(BLOCK) void pack_kernel(const T* __restrict__ tgt, int N, int m, int c,
                                                     typename V4<T>::type* __restrict__ out, int m_pad, int bpc,
                                                     const T* __restrict__ frame, const int32_t* __restrict__ tgt_rows) {
    int b, blk;                                             // all blocks of a cloud on one XCD (decode_block)
    if (!decode_block(bpc, N, b, blk)) return;
    const int j = blk * BLOCK + threadIdx.x;
    if (j >= m_pad) return;
    typename V4<T>::type v;
    if (j < rows_of(tgt_rows, b, m)) v = pack_row<T>(tgt + ((size_t)b * m + j) * c, frame ? frame + (size_t)b * 12 : nullptr);
    else { v.x = v.y = v.z = T(0); v.w = inf_v<T>(); }
    out[(size_t)b * m_pad + j] = v;
}

// ------------------------------------------------------------ sweep index / loop set-up
// What follows the key sort (dicp_sweep_sort) in the sorted-sweep search structure: the packed rows, and optionally the
// full rows, in sorted order.
template <typename T>
__global__ __launch_bounds__(BLOCK) void sweep_rows_kernel(const T* __restrict__ tgt, const int32_t* __restrict__ tgt_rows, int N, int m, int c,
                                                           int m_pad, int bpc, typename V4<T>::type* __restrict__ tgs4, const int32_t* __restrict__ tperm,
                                                           T* __restrict__ tgt_s /* optional (N,m_pad,rs): the full rows in sorted order */, int rs /* elements per row of tgt_s, >= c */,
                                                           const T* __restrict__ frame /* optional (N,12): tgs4 rows are Q y + t; tgt_s stays as given */) {
    constexpr int U = 4;                                    // rows per thread in flight (index -> row is a dependent pair)
    int b, blk;
    if (!decode_block(bpc, N, b, blk)) return;
    const int s0 = blk * (BLOCK * U) + threadIdx.x;
    const int mc = rows_of(tgt_rows, b, m);
    {
        int j[U];
        T y[U][6];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t at = (size_t)b * m_pad + min(s0 + u * BLOCK, m_pad - 1);
            j[u] = tperm[at];
        }
        const bool full = tgt_s && c == 6;                  // the normals are wanted too: read the whole row once
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const T* yp = tgt + ((size_t)b * m + (j[u] >= 0 && j[u] < mc ? j[u] : 0)) * c;   // pad slots repeat row 0 (never matched)
            y[u][0] = yp[0]; y[u][1] = yp[1]; y[u][2] = yp[2];
            if (full) { y[u][3] = yp[3]; y[u][4] = yp[4]; y[u][5] = yp[5]; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int sl = s0 + u * BLOCK;
            if (sl >= m_pad) continue;
            typename V4<T>::type v;
            if (j[u] >= 0 && j[u] < mc) v = pack_row<T>(y[u], frame ? frame + (size_t)b * 12 : nullptr);
            else { v.x = big_v<T>(); v.y = v.z = T(0); v.w = inf_v<T>(); }      // pads sort last and can never win
            tgs4[(size_t)b * m_pad + sl] = v;
            if (tgt_s) {
                T* o = tgt_s + ((size_t)b * m_pad + sl) * rs;
                o[0] = y[u][0]; o[1] = y[u][1]; o[2] = y[u][2];
                if (full) { o[3] = y[u][3]; o[4] = y[u][4]; o[5] = y[u][5]; }
                for (int k = c; k < rs; ++k) o[k] = T(0);
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(BLOCK) void sweep_buckets_kernel(const T* __restrict__ keys /* (N,m_pad) ascending */, int N, int m_full, int m_pad,
                                                              int nbkt, int32_t* __restrict__ bucket, T* __restrict__ brange, const int32_t* __restrict__ tgt_rows) {
    const int cloud = blockIdx.x;
    const int m = max(rows_of(tgt_rows, cloud, m_full), 1);
    const T* __restrict__ xs = keys + (size_t)cloud * m_pad;
    const T xlo = xs[0], span = xs[m - 1] - xlo;
    for (int b = threadIdx.x; b <= nbkt; b += BLOCK) {
        const T edge = fma_t(T(b), span / T(nbkt), xlo);    // (explicit fma: dicp_sweep_sort builds the same table from LDS)
        int lo = 0, hi = m;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (xs[mid] < edge) lo = mid + 1; else hi = mid; }
        bucket[(size_t)cloud * (nbkt + 1) + b] = lo;
    }
    if (threadIdx.x == 0) {
        brange[(size_t)cloud * 2] = xlo;
        brange[(size_t)cloud * 2 + 1] = span > T(0) ? T(nbkt) / span : T(0);
    }
}

// Stable sort of a cloud's target x keys (float) entirely in LDS: LSD radix sort, 8-bit digits, 4 passes, one block of
// 1024 threads per cloud, 16 keys per thread, up to 16384 keys (the pad slots carry +max and sort last; ties keep their
// index order, like torch.sort(stable=True), so the permutation is the one the rest of the path was built on).
// A pass never uses an atomic: a wave takes its 16 rounds of 64 keys in order; per round, 8 ballots tell every lane which
// lanes hold the same digit (rank inside the round = set bits below the lane), the first lane of every digit group
// advances the wave's per-digit counter in LDS, and after the rounds a block scan turns the 16 x 256 wave histograms
// into offsets.  Keys and indices stay in registers between passes; one LDS buffer (written at the new positions, read
// back in the striped order) is all the exchange space it takes.
constexpr int RS_THREADS = 1024, RS_PER = 16, RS_MAX = RS_THREADS * RS_PER;
__device__ __forceinline__ unsigned sortable_bits(float x) {      // order-preserving map float -> unsigned
    unsigned u = __float_as_uint(x + 0.0f);                        // -0 sorts as +0 (they compare equal; index order decides)
    u ^= (u >> 31) ? 0xffffffffu : 0x80000000u;
    return x != x ? 0xffffffffu : u;                               // NaN of either sign sorts last, as torch.sort has it
}
__global__ __launch_bounds__(RS_THREADS) void sort_keys_kernel(const float* __restrict__ tgt, int c, int N, int m_full, int m_pad,
                                                               float* __restrict__ keys_sorted, int32_t* __restrict__ tperm,
                                                               int nbkt, int32_t* __restrict__ bucket, float* __restrict__ brange,
                                                               const float* __restrict__ frame, const int32_t* __restrict__ tgt_rows) {
    __shared__ unsigned skey[RS_MAX];
    __shared__ unsigned short sidx[RS_MAX];
    __shared__ int cnt[RS_THREADS / WAVE][256];             // per wave, per digit: running count, then offset
    __shared__ int tot[256];
    const int cloud = blockIdx.x, tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid >> 6;
    const float* __restrict__ rows = tgt + (size_t)cloud * m_full * c;
    const int m = rows_of(tgt_rows, cloud, m_full);          // rows past the cloud's own length are pad slots too
    unsigned key[RS_PER];
    unsigned short idx[RS_PER];
#pragma unroll
    for (int e = 0; e < RS_PER; ++e) {                      // striped: position = wave * 1024 + e * 64 + lane
        const int pos = wave * (WAVE * RS_PER) + e * WAVE + lane;
        unsigned u = 0xffffffffu;                           // beyond m_pad: sentinel, sorts after everything
        // pad slots keep the largest key there is: with the stable order they follow EVERY real row, also one whose x is
        // +inf or NaN (which sort above +max) -- sorted positions [0, m) are exactly the real rows, whatever they hold
        if (pos < m) { float q[3]; frame_apply<float>(frame ? frame + (size_t)cloud * 12 : nullptr, rows + (size_t)pos * c, q); u = sortable_bits(q[0]); }   // (the packed rows' x)
        key[e] = u;
        idx[e] = (unsigned short)pos;
    }
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = pass * 8;
        for (int d = lane; d < 256; d += WAVE) cnt[wave][d] = 0;
        __builtin_amdgcn_wave_barrier();
        int rank[RS_PER];
#pragma unroll
        for (int e = 0; e < RS_PER; ++e) {
            const unsigned d = (key[e] >> shift) & 0xffu;
            unsigned long long same = ~0ull;                // lanes of this round holding the same digit
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const unsigned long long bal = __ballot((d >> b) & 1u);
                same &= ((d >> b) & 1u) ? bal : ~bal;
            }
            const int below = __builtin_amdgcn_mbcnt_hi((unsigned)(same >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)same, 0u));
            const int base = cnt[wave][d];                  // every lane of the group reads before its first lane writes
            __builtin_amdgcn_wave_barrier();
            if (below == 0) cnt[wave][d] = base + __popcll(same);
            __builtin_amdgcn_wave_barrier();
            rank[e] = base + below;
        }
        __syncthreads();
        // offsets: digit-major, wave-minor exclusive scan of the 256 x 16 counts
        if (tid < 256) {
            int s = 0;
            for (int w = 0; w < RS_THREADS / WAVE; ++w) { const int v = cnt[w][tid]; cnt[w][tid] = s; s += v; }
            tot[tid] = s;
        }
        __syncthreads();
        if (tid < WAVE) {                                   // exclusive scan of the 256 digit totals by one wave (4 per lane)
            int v[4], s = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[k] = tot[lane * 4 + k]; s += v[k]; }
            int inc = s;
#pragma unroll
            for (int off = 1; off < WAVE; off <<= 1) { const int o = __shfl_up(inc, off); if (lane >= off) inc += o; }
            int run = inc - s;
#pragma unroll
            for (int k = 0; k < 4; ++k) { tot[lane * 4 + k] = run; run += v[k]; }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < RS_PER; ++e) {
            const unsigned d = (key[e] >> shift) & 0xffu;
            const int pos = tot[d] + cnt[wave][d] + rank[e];
            skey[pos] = key[e];
            sidx[pos] = idx[e];
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < RS_PER; ++e) {
            const int pos = wave * (WAVE * RS_PER) + e * WAVE + lane;
            key[e] = skey[pos];
            idx[e] = sidx[pos];
        }
        __syncthreads();
    }
#pragma unroll
    for (int e = 0; e < RS_PER; ++e) {
        const int pos = wave * (WAVE * RS_PER) + e * WAVE + lane;
        if (pos < m_pad) {
            unsigned u = key[e];
            u ^= (u >> 31) ? 0x80000000u : 0xffffffffu;
            keys_sorted[(size_t)cloud * m_pad + pos] = __uint_as_float(u);
            tperm[(size_t)cloud * m_pad + pos] = (int32_t)idx[e];
        }
    }
    // the sweep's bucket table (what sweep_buckets_kernel computes from global memory) while the sorted keys are in LDS
    if (bucket) {
        auto key_at = [&](int i) { unsigned u = skey[i]; u ^= (u >> 31) ? 0x80000000u : 0xffffffffu; return __uint_as_float(u); };
        const float xlo = key_at(0), span = key_at(max(m, 1) - 1) - xlo;
        for (int b = tid; b <= nbkt; b += RS_THREADS) {
            const unsigned edge = sortable_bits(fma_t(float(b), span / float(nbkt), xlo));
            int lo = 0, hi = m;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (skey[mid] < edge) lo = mid + 1; else hi = mid; }
            bucket[(size_t)cloud * (nbkt + 1) + b] = lo;
        }
        if (tid == 0) {
            brange[(size_t)cloud * 2] = xlo;
            brange[(size_t)cloud * 2 + 1] = span > 0.f ? float(nbkt) / span : 0.f;
        }
    }
}

// sort key of the queries: their x coordinate under the given pose (NULL = identity)
template <typename T>
__global__ __launch_bounds__(BLOCK) void query_keys_kernel(const T* __restrict__ src, const T* __restrict__ pose, int N, int n, int bpc,
                                                           T* __restrict__ keys) {
    int b, blk;
    if (!decode_block(bpc, N, b, blk)) return;
    const int i = blk * BLOCK + threadIdx.x;
    if (i >= n) return;
    {
        const size_t t = (size_t)b * n + i;
        const T* p = src + t * 3;
        T x = p[0];
        if (pose) {
            const T* q = pose + (size_t)b * 12;
            x = fma_t(q[0], p[0], fma_t(q[1], p[1], fma_t(q[2], p[2], q[9])));
        }
        keys[t] = x;
    }
}

// Query order for the sweep: a counting sort of the queries by the bucket of their transformed x (equal-width buckets over
// the TARGET's x range, the same table geometry as the search's bucket index).  The order inside a bucket is arbitrary:
// the search is exact for any order, the order only keeps a wave's queries neighbours in x, and 16 unordered queries
// per bucket widen a wave's slab by a few rows.  One block per cloud, everything in LDS: ~20x cheaper than a full sort.
constexpr int QO_THREADS = 1024;
constexpr int QO_BUCKETS = 2048;
constexpr int QO_KEYS = 16384;      // sorted target keys kept in LDS for the rank search (64 KiB)
constexpr int QO_TABLE = 1024;      // ... and the coarse lower-bound table that brackets it
// Queries OUTSIDE the targets' x range (round 6).  Scan pairs that overlap only partly put a third of the source there; ordered by rank (or by buckets
// over the targets' range) they all shared the end bucket, in arrival order -- units of the sweep whose queries lay metres apart in x, each dragging the
// slab of the farthest (the matrix-core sweep, whose slab test is per lane, scored twice the pairs of the per-query test: profiles/r06_search_direction.txt).
// QO_SIDE ordering buckets on either side now hold them by x, one bucket per 1/QO_SIDE of the targets' span (clamped at one span away).
constexpr int QO_SIDE = 256;
constexpr int QO_MID = QO_BUCKETS - 2 * QO_SIDE;
__device__ __forceinline__ int qo_side_bucket(float dist_scaled) { return (int)fminf(fmaxf(dist_scaled, 0.f), (float)(QO_SIDE - 1)); }
// QO_STAGE: queries per cloud whose permutation is assembled in LDS (16384 -> 32 KiB, several clouds per CU; 65536 -> 128 KiB)
template <typename T, int QO_STAGE>
__global__ __launch_bounds__(QO_THREADS) void query_order_kernel(const T* __restrict__ src, const T* __restrict__ pose,
                                                                 const T* __restrict__ brange, int nbkt_range, int N, int n_full,
                                                                 int32_t* __restrict__ qorder, const T* __restrict__ w,
                                                                 T* __restrict__ src_s, T* __restrict__ w_s, int reproducible,
                                                                 const int32_t* __restrict__ spos_prev, int m_pad,
                                                                 const T* __restrict__ skeys, int kstride, int mt_full, const int32_t* __restrict__ table,
                                                                 const int32_t* __restrict__ src_rows, const int32_t* __restrict__ tgt_rows,
                                                                 const T* __restrict__ pose_prev = nullptr, const int32_t* __restrict__ order_prev = nullptr) {
    __shared__ int cnt[QO_BUCKETS];
    __shared__ int s_same;
    __shared__ int wsum[QO_THREADS / WAVE];
    __shared__ unsigned short stage[QO_STAGE];              // query ids (< 65536) by slot: the permutation is assembled here
    __shared__ float lkeys[QO_STAGE <= 16384 ? QO_KEYS : 1];
    __shared__ int ltab[QO_STAGE <= 16384 ? QO_TABLE + 1 : 1];
    const int cloud = blockIdx.x, tid = threadIdx.x;
    // ragged batches: the cloud's own queries [0, n) are ordered; rows n .. n_full - 1 (pads) keep their slots, so that qorder
    // stays a permutation of all n_full rows (the row copies and the un-permuting of the backward walk all of it)
    const int n = rows_of(src_rows, cloud, n_full), mt = max(rows_of(tgt_rows, cloud, mt_full), 1);
    src += (size_t)cloud * (n_full - n) * 3;                // (every access below is src + (cloud * n + i) * 3)
    if (w) w += (size_t)cloud * (n_full - n);
    if (spos_prev) spos_prev += (size_t)cloud * (n_full - n);
    qorder += (size_t)cloud * (n_full - n);
    if (src_s) src_s += (size_t)cloud * (n_full - n) * 3;
    if (w_s) w_s += (size_t)cloud * (n_full - n);
    for (int i = n + tid; i < n_full; i += QO_THREADS) {
        qorder[(size_t)cloud * n + i] = i;
        if (src_s) { const T* p = src + ((size_t)cloud * n + i) * 3; T* o = src_s + ((size_t)cloud * n + i) * 3; o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; }
        if (w_s) w_s[(size_t)cloud * n + i] = w[(size_t)cloud * n + i];
    }
    if (n <= 0) return;
    // A RE-ordering (dicp_query_reorder: given the order made under an earlier pose of the same call): a cloud whose points have moved by less than a tenth
    // of the x extent of a unit of the sweep since then keeps that order -- a copy instead of the counting sort (round 6: the re-orderings before iterations
    // 2 and 3 were 37 us each at the benchmark shape for clouds that had moved by a millimetre).  The bound: |dC|_F R + |dr| with R the radius of the targets'
    // x range about the frame's origin (the order only keeps a wave's queries neighbours: a stale one costs pairs, never results).
    if (pose && pose_prev && order_prev && !src_s && !w_s) {
        if (tid == 0) {
            const T* a = pose + (size_t)cloud * 12;
            const T* b = pose_prev + (size_t)cloud * 12;
            float dc = 0.f, dr = 0.f;
#pragma unroll
            for (int e = 0; e < 9; ++e) { const float d = (float)(a[e] - b[e]); dc += d * d; }
#pragma unroll
            for (int e = 9; e < 12; ++e) { const float d = (float)(a[e] - b[e]); dr += d * d; }
            const float x0 = (float)brange[(size_t)cloud * 2], ts = (float)brange[(size_t)cloud * 2 + 1];
            const float span = ts > 0.f ? (float)nbkt_range / ts : 0.f;
            const float moved = sqrtf(dc) * 1.7321f * fmaxf(fabsf(x0), fabsf(x0 + span)) + sqrtf(dr);
            s_same = (moved * (float)n < 0.1f * span * (float)(2 * WAVE)) ? 1 : 0;        // (NaN: re-order)
        }
        __syncthreads();
        if (s_same) {
            const int32_t* op = order_prev + (size_t)cloud * (n_full - n);
            for (int i = tid; i < n; i += QO_THREADS) qorder[(size_t)cloud * n + i] = op[(size_t)cloud * n + i];
            return;
        }
    }
    for (int b = tid; b < QO_BUCKETS; b += QO_THREADS) cnt[b] = 0;
    T q[4] = {T(1), T(0), T(0), T(0)};
    if (pose) { const T* pp = pose + (size_t)cloud * 12; q[0] = pp[0]; q[1] = pp[1]; q[2] = pp[2]; q[3] = pp[9]; }
    const T xlo = brange[(size_t)cloud * 2];
    const T tscale = brange[(size_t)cloud * 2 + 1];                                       // table buckets per unit x
    const T scale = tscale * (T(QO_MID) / T(nbkt_range));                                 // ordering buckets per unit x (the middle ones: inside the targets' range)
    // rank ordering: the cloud's sorted target x keys, as floats, in LDS (QO_KEYS of them: bigger clouds fall back to x buckets)
    const bool ranked = QO_STAGE <= 16384 && skeys && table && !spos_prev && mt <= QO_KEYS && nbkt_range <= QO_TABLE;
    if (ranked) {
        const T* __restrict__ keys = skeys + (size_t)cloud * m_pad * kstride;
        for (int j = tid; j < mt; j += QO_THREADS) lkeys[j] = (float)keys[(size_t)j * kstride];
        for (int j = tid; j <= nbkt_range; j += QO_THREADS) ltab[j] = table[(size_t)cloud * (nbkt_range + 1) + j];
    }
    const float span_w = tscale > T(0) ? (float)(T(nbkt_range) / tscale) : 0.f;          // the targets' x range as the bucket table has it
    const float side_w = span_w > 0.f ? (float)QO_SIDE / span_w : 0.f;                    // side buckets per unit x
    auto bucket_of = [&](int i) {
        if (spos_prev) {        // bucket = rank of the query's previous match among the sorted targets: equal-POPULATION buckets,
                                // whatever the density of the cloud along x (an outlier cannot coarsen them)
            const int sp = spos_prev[(size_t)cloud * n + i];
            return sp < 0 ? QO_BUCKETS - 1 : (int)(((long)min(sp, m_pad - 1) * QO_BUCKETS) / m_pad);
        }
        const T* p = src + ((size_t)cloud * n + i) * 3;
        const T x = fma_t(q[0], p[0], fma_t(q[1], p[1], fma_t(q[2], p[2], q[3])));
        const float d = (float)(x - xlo);
        if (d < 0.f) return QO_SIDE - 1 - qo_side_bucket(-d * side_w);                    // left of every target, by x
        if (d > span_w) return QO_SIDE + QO_MID + qo_side_bucket((d - span_w) * side_w);  // right of every target
        const float f = d * (float)scale;
        return QO_SIDE + (f > 0.f ? (f < (float)(QO_MID - 1) ? (int)f : QO_MID - 1) : 0); // (NaN -> the first middle bucket)
    };
    __syncthreads();
    // ranked: the ends of the targets' x range, leaving out ONE key per end that lies further from the rest than the rest is wide (a ragged batch's far
    // pad row: it would make "inside the range" of everything to its left)
    float rk_lo = 0.f, rk_hi = 0.f, rk_side = 0.f;
    if (ranked) {
        int jl = 0, jh = mt - 1;
        if (mt >= 3 && lkeys[mt - 1] - lkeys[mt - 2] > lkeys[mt - 2] - lkeys[0]) jh = mt - 2;
        if (mt >= 3 && lkeys[1] - lkeys[0] > lkeys[jh] - lkeys[1]) jl = 1;
        rk_lo = lkeys[jl]; rk_hi = lkeys[jh];
        rk_side = rk_hi > rk_lo ? (float)QO_SIDE / (rk_hi - rk_lo) : 0.f;
    }
    // one returning LDS add per query gives its bucket AND its rank inside the bucket; both stay in registers while
    // the counters are turned into offsets (LDS atomics are the cost of this kernel: ~137 cycles per wave-instruction)
    constexpr int PER = 16;                                 // register-resident up to PER * QO_THREADS queries per cloud
    int bk[PER], rk[PER];
    const bool small = n <= PER * QO_THREADS;
    if (small && ranked) {
        // rank of every query's x among the sorted target keys, from the LDS copy of the keys: a full binary search per
        // query (14 LDS reads; from global memory the same chain of dependent loads took 144 us per launch)
        // (every query's coordinates are loaded before the first search, and the rounds below are unrolled so that their LDS chains overlap: 43 -> 38 us per
        //  launch at the benchmark shape, round 6.  What is left is the LDS adds: ~137 cycles per wave-instruction on 2048 random counters)
        float xq[PER];
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            const int i = min(e * QO_THREADS + tid, n - 1);
            const T* p = src + ((size_t)cloud * n + i) * 3;
            xq[e] = (float)fma_t(q[0], p[0], fma_t(q[1], p[1], fma_t(q[2], p[2], q[3])));
        }
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            const int i = e * QO_THREADS + tid;
            int bb = -1, rr = 0;
            if (i < n) {
                const float x = xq[e];
                // the coarse table (also in LDS) brackets the lower bound: ~4 steps on an even cloud instead of 14
                float f = (x - (float)xlo) * (float)tscale;
                f = f > 0.f ? (f < (float)nbkt_range ? f : (float)nbkt_range) : 0.f;
                const int tb = (int)f;
                int lo = ltab[tb], hi = ltab[min(tb + 1, nbkt_range)];
                if (!(lo <= hi) || (lo > 0 && !(lkeys[lo - 1] < x)) || (hi < mt && lkeys[hi] < x)) { lo = 0; hi = mt; }  // rounding at an edge
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (lkeys[mid] < x) lo = mid + 1; else hi = mid; }
                if (x < rk_lo) bb = QO_SIDE - 1 - qo_side_bucket((rk_lo - x) * rk_side);
                else if (x > rk_hi) bb = QO_SIDE + QO_MID + qo_side_bucket((x - rk_hi) * rk_side);
                else bb = QO_SIDE + (int)(((long)lo * (QO_MID - 1)) / max(mt, 1));
                rr = atomicAdd(&cnt[bb], 1);
            }
            bk[e] = bb; rk[e] = rr;
        }
    } else if (small) {
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            const int i = e * QO_THREADS + tid;
            bk[e] = i < n ? bucket_of(i) : -1;
            rk[e] = i < n ? atomicAdd(&cnt[bk[e]], 1) : 0;
        }
    } else {
        for (int i = tid; i < n; i += QO_THREADS) atomicAdd(&cnt[bucket_of(i)], 1);
    }
    __syncthreads();
    // exclusive prefix sum of the QO_BUCKETS counters (two per thread)
    const int a0 = cnt[2 * tid], a1 = cnt[2 * tid + 1];
    int v = a0 + a1;
    const int lane = tid & (WAVE - 1), wave = tid >> 6;
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) { const int o = __shfl_up(v, off); if (lane >= off) v += o; }
    if (lane == WAVE - 1) wsum[wave] = v;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    const int excl = base + v - (a0 + a1);
    __syncthreads();
    cnt[2 * tid] = excl;
    cnt[2 * tid + 1] = excl + a0;
    __syncthreads();
    if (small) {
        // the permutation is assembled in LDS and leaves as coalesced rows: 4-byte stores scattered over the cloud's
        // slots cost a 64-byte memory write each (measured: 243 MB written for 17 MB of order, 60 us instead of ~15)
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            const int i = e * QO_THREADS + tid;
            if (i < n) stage[min(cnt[bk[e]] + rk[e], n - 1)] = (unsigned short)i;
        }
        __syncthreads();
        // the arrival order of the LDS adds is not reproducible: on request (15 us) put every bucket's members in ascending
        // query index (insertion sort, ~8 per bucket) so that the order -- and every sum taken in it -- is the same every run
        for (int b = tid; reproducible && b < QO_BUCKETS; b += QO_THREADS) {
            const int lo = cnt[b], hi = b + 1 < QO_BUCKETS ? cnt[b + 1] : n;
            if (hi - lo > 64) continue;                     // a crowd (one x plane; queries outside the targets' x range): left as it arrived
            for (int a = lo + 1; a < hi; ++a) {
                const unsigned short v = stage[a];
                int k = a - 1;
                while (k >= lo && stage[k] > v) { stage[k + 1] = stage[k]; --k; }
                stage[k + 1] = v;
            }
        }
        __syncthreads();
        for (int sl = tid; sl < n; sl += QO_THREADS) {
            const int i = stage[sl];
            qorder[(size_t)cloud * n + sl] = (int32_t)i;
            if (src_s) {                                    // the rows in slot order, for coalesced query loads (and the backward)
                const T* p = src + ((size_t)cloud * n + i) * 3;
                T* o = src_s + ((size_t)cloud * n + sl) * 3;
                o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
            }
            if (w_s) w_s[(size_t)cloud * n + sl] = w[(size_t)cloud * n + i];
        }
    } else if (n <= QO_STAGE && !src_s && !w_s) {           // two passes of LDS adds, permutation still assembled in LDS
        for (int i = tid; i < n; i += QO_THREADS) stage[min(atomicAdd(&cnt[bucket_of(i)], 1), n - 1)] = (unsigned short)i;
        __syncthreads();
        for (int sl = tid; sl < n; sl += QO_THREADS) qorder[(size_t)cloud * n + sl] = (int32_t)stage[sl];
    } else {
        for (int i = tid; i < n; i += QO_THREADS) {
            const int slot = min(atomicAdd(&cnt[bucket_of(i)], 1), n - 1);
            qorder[(size_t)cloud * n + slot] = i;
            if (src_s) {
                const T* p = src + ((size_t)cloud * n + i) * 3;
                T* o = src_s + ((size_t)cloud * n + slot) * 3;
                o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
            }
            if (w_s) w_s[(size_t)cloud * n + slot] = w[(size_t)cloud * n + i];
        }
    }
}

// first-iteration state of the loop: pose_0 from T_init, alive_0 = 1, n_start = rows * #(w0 > thresh)  (ICP.py:124-129)
constexpr int LI_THREADS = 1024;     // one block per cloud: its two passes over the cloud (weights, bounding box) are chains of loads
template <typename T>
__global__ __launch_bounds__(LI_THREADS) void loop_init_kernel(const T* __restrict__ T_init, const T* __restrict__ w0, T thresh, int rows, int n,
                                                          T* __restrict__ pose0, T* __restrict__ alive0, T* __restrict__ n_start,
                                                          const T* __restrict__ frame, T* __restrict__ pose_search0,
                                                          const T* __restrict__ src, T* __restrict__ rmax, T* __restrict__ dcum, int dstride) {
    __shared__ int cnt[LI_THREADS / WAVE];
    __shared__ T box[(LI_THREADS / WAVE) * 6];
    const int cloud = blockIdx.x, tid = threadIdx.x;
    if (rmax) {     // bounding box of the cloud -> (radius, midpoint): with them the step kernels bound how far ANY query moves between two
                    // poses (match certificates): dC p + dr = dC (p - p0) + (dC p0 + dr), so a cloud far from the origin costs nothing
        T lo[3] = {inf_v<T>(), inf_v<T>(), inf_v<T>()}, hi[3] = {-inf_v<T>(), -inf_v<T>(), -inf_v<T>()};
        for (int i = tid; i < n; i += LI_THREADS) {
            const T* p = src + ((size_t)cloud * n + i) * 3;
#pragma unroll
            for (int k = 0; k < 3; ++k) { const T v = p[k]; lo[k] = v < lo[k] ? v : lo[k]; hi[k] = v > hi[k] ? v : hi[k]; }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
#pragma unroll
            for (int off = WAVE / 2; off > 0; off >>= 1) {
                const T a = __shfl_down(lo[k], off), c = __shfl_down(hi[k], off);
                lo[k] = a < lo[k] ? a : lo[k]; hi[k] = c > hi[k] ? c : hi[k];
            }
        }
        if ((tid & (WAVE - 1)) == 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) { box[(tid >> 6) * 6 + k] = lo[k]; box[(tid >> 6) * 6 + 3 + k] = hi[k]; }
        }
        __syncthreads();
        if (tid == 0) {
            T d2 = T(0), p0[3], pn = T(0);
            for (int k = 0; k < 3; ++k) {
                T l = box[k], h = box[3 + k];
                for (int w = 1; w < LI_THREADS / WAVE; ++w) { l = box[w * 6 + k] < l ? box[w * 6 + k] : l; h = box[w * 6 + 3 + k] > h ? box[w * 6 + 3 + k] : h; }
                p0[k] = T(0.5) * (l + h);
                d2 += (h - l) * (h - l);
                pn += p0[k] * p0[k];
            }
            // (a cloud with an infinite coordinate: radius inf -> nothing is ever certified.  NaN points are skipped by the min / max
            //  comparisons above, so the box covers the finite points only; a NaN query scores NaN against every target and never gets a budget)
            const T rad = T(0.5) * m_sqrt(d2) * (T(1) + T(8) * CertUlp<T>::v) + T(8) * CertUlp<T>::v * m_sqrt(pn);
            T* ro = rmax + (size_t)cloud * 4;
            ro[0] = rad; ro[1] = p0[0]; ro[2] = p0[1]; ro[3] = p0[2];
            // (M_0, e_0): no motion yet; e_k = rounding of a transformed point C p + (r - centre) under pose k
            const T* Ti = T_init + (size_t)cloud * 16;
            dcum[(size_t)cloud * dstride] = T(0);
            const T* ct = frame ? frame + (size_t)cloud * 12 + 9 : nullptr;        // (|t| = |centre|: Q is orthonormal)
            const T cn = ct ? m_sqrt(ct[0] * ct[0] + ct[1] * ct[1] + ct[2] * ct[2]) : T(0);
            dcum[(size_t)cloud * dstride + 1] = T(8) * CertUlp<T>::v * (m_sqrt(pn) + rad + m_sqrt(Ti[3] * Ti[3] + Ti[7] * Ti[7] + Ti[11] * Ti[11]) + cn + T(1));
        }
    }
    int k = 0;
    if (w0) { for (int i = tid; i < n; i += LI_THREADS) k += w0[(size_t)cloud * n + i] > thresh ? 1 : 0; }
    else if (tid == 0) k = T(1) > thresh ? n : 0;            // w0 == NULL: unit weights
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) k += __shfl_down(k, off);
    if ((tid & (WAVE - 1)) == 0) cnt[tid >> 6] = k;
    __syncthreads();
    if (tid == 0) {
        int tot = 0;
        for (int w = 0; w < LI_THREADS / WAVE; ++w) tot += cnt[w];
        n_start[cloud] = (T)((long)tot * rows);
        alive0[cloud] = T(1);
    }
    if (tid < 12) {
        const T* M = T_init + (size_t)cloud * 16;
        const T v = tid < 9 ? M[(tid / 3) * 4 + tid % 3] : M[(tid - 9) * 4 + 3];
        pose0[(size_t)cloud * 12 + tid] = v;
        if (pose_search0) {
            const T ps[12] = {M[0], M[1], M[2], M[4], M[5], M[6], M[8], M[9], M[10], M[3], M[7], M[11]};
            pose_search0[(size_t)cloud * 12 + tid] = frame_pose_entry<T>(frame ? frame + (size_t)cloud * 12 : nullptr, ps, tid);
        }
    }
}

// [C | r - centre] straight from T_init (N,4,4): the search pose of iteration 0, for a caller that wants the first query order in
// the queue before the loop state exists (same values as loop_init_kernel writes)
template <typename T>
__global__ __launch_bounds__(BLOCK) void search_pose_kernel(const T* __restrict__ T_init, const T* __restrict__ frame, int N, T* __restrict__ out) {
    const int e = blockIdx.x * BLOCK + threadIdx.x;
    if (e >= N * 12) return;
    const int cloud = e / 12, k = e - cloud * 12;
    const T* M = T_init + (size_t)cloud * 16;
    const T ps[12] = {M[0], M[1], M[2], M[4], M[5], M[6], M[8], M[9], M[10], M[3], M[7], M[11]};
    out[e] = frame_pose_entry<T>(frame ? frame + (size_t)cloud * 12 : nullptr, ps, k);
}

template <typename T>
__global__ __launch_bounds__(BLOCK) void loop_finish_kernel(const T* __restrict__ pose_K, const T* __restrict__ alive_K, const T* __restrict__ n_start,
                                                            const T* __restrict__ n_matched, int K, int N, T* __restrict__ iterations,
                                                            T* __restrict__ matched_ratio, T* __restrict__ T_out) {
    const int cloud = blockIdx.x * BLOCK + threadIdx.x;
    if (cloud >= N) return;
    if (iterations[cloud] == T(0)) iterations[cloud] = (T)K;
    if (matched_ratio[cloud] == T(0)) {
        long start = (alive_K[cloud] != T(0)) ? (long)n_start[cloud] : 0;
        if (start == 0) start = 1;
        matched_ratio[cloud] = (T)((float)(long)n_matched[cloud] / (float)start);     // int64/int64 -> float32 in the reference
    }
    const T* q = pose_K + (size_t)cloud * 12;
    T* M = T_out + (size_t)cloud * 16;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        M[i * 4 + 0] = q[i * 3 + 0]; M[i * 4 + 1] = q[i * 3 + 1]; M[i * 4 + 2] = q[i * 3 + 2]; M[i * 4 + 3] = q[9 + i];
    }
    M[12] = M[13] = M[14] = T(0); M[15] = T(1);
}
