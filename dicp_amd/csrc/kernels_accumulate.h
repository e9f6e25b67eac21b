// libdicp_hip.so -- forward: block reductions, accumulate (residuals, weights, Jacobian, normal-equation sums), step, small-cloud loop.
// Part of the one translation unit dicp_kernels.hip (included inside its anonymous namespace, in this order: kernels_setup.h, kernels_search.h, kernels_setup_sort.h, kernels_rows.h, kernels_accumulate.h, kernels_backward.h, kernels_soft_svd.h, kernels_host.h).
// -------------------------------------------------------------------- reductions
// Sum NV per-thread values over the block; thread k < PAD writes slot k of out.
// The wave step is a reduce-scatter: a lane exchange costs an LDS-crossbar instruction (ds_bpermute), and NV full
// butterflies (6 NV of them: 174 for the 29 forward sums) made the reduction a fifth of accumulate_kernel's time at
// 1024 points per block.  Here each exchange also HALVES the values a lane carries -- the lane keeps the half its
// lane bit selects and adds the partner's copy of that half -- so 32 values take 16+8+4+2+1 exchanges, one more joins
// the two lanes that end up with the same value: 32 in all, every value summed in one fixed order.
template <typename T, int H>
__device__ __forceinline__ void halve_step(T* v, int lane) {      // v[0..2H) -> v[0..H): partner = lane ^ (2H) for H = 16..1
    const bool up = (lane & (2 * H)) != 0;
#pragma unroll
    for (int k = 0; k < H; ++k) {
        const T keep = up ? v[H + k] : v[k];
        const T give = up ? v[k] : v[H + k];
        v[k] = keep + __shfl_xor(give, 2 * H);
    }
}
template <typename T, int NV, int PAD, int NT>
__device__ __forceinline__ void block_reduce_store(T* v, T* __restrict__ out, T* lds /* [NT/WAVE][PAD] */) {
    static_assert(NV <= 32 && PAD >= NV, "reduce-scatter over 32 slots");
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid >> 6;
    if constexpr (NV <= 16) {
        // 16 slots (the backward's 12 pose sums): four halving steps over lane bits 4..1, then the two lane bits that are left (0 and 5) as plain
        // exchanges of the one value a lane still carries: 17 exchanges instead of 32
        T a[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) a[k] = k < NV ? v[k] : T(0);
        halve_step<T, 8>(a, lane);
        halve_step<T, 4>(a, lane);
        halve_step<T, 2>(a, lane);
        halve_step<T, 1>(a, lane);
        T x = a[0] + __shfl_xor(a[0], 1);
        x += __shfl_xor(x, 32);
        const int slot = ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
        if (!(lane & 33) && slot < NV) lds[wave * PAD + slot] = x;
        __syncthreads();
        if (tid < PAD) {
            T s = T(0);
            if (tid < NV) {
#pragma unroll
                for (int w = 0; w < NT / WAVE; ++w) s += lds[w * PAD + tid];
            }
            out[tid] = s;
        }
        return;
    }
    T a[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) a[k] = k < NV ? v[k] : T(0);
    halve_step<T, 16>(a, lane);     // lane bit 5 picks the half, ... lane bit 1 the last pair:
    halve_step<T, 8>(a, lane);      // lane L ends with slot (L >> 1) & 31 in bit order 5,4,3,2,1
    halve_step<T, 4>(a, lane);
    halve_step<T, 2>(a, lane);
    halve_step<T, 1>(a, lane);
    const T x = a[0] + __shfl_xor(a[0], 1);
    const int slot = ((lane >> 5) & 1) * 16 + ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
    if (!(lane & 1) && slot < NV) lds[wave * PAD + slot] = x;
    __syncthreads();
    if (tid < PAD) {
        T s = T(0);
        if (tid < NV) {
#pragma unroll
            for (int w = 0; w < NT / WAVE; ++w) s += lds[w * PAD + tid];
        }
        out[tid] = s;
    }
}

// -------------------------------------------------------------------- accumulate
// CERT (certified iterations of the sweep loop): the ROW CACHE form.  Near the pose a query's match does not change for many iterations (that is
// what the certificates prove), so gathering its 24-byte target row again in every iteration -- 39 bytes at the memory side, from a random place of
// the cloud's sorted rows -- and reading its budget and its match only to hand the match on to the next iteration's history slab was most of what such
// a launch moved (63 bytes per point where section 8d counts 44).  Here every query keeps its matched ROW next to it (nbr, by query: a stream), the
// match history is kept by reference (AccCert::of), and the certificates are the guard launch's business alone (knn_sweep_guard_kernel): a match it
// CHANGED waits in pend, its group of 64 queries is marked in gdirty.  A launch reads, per group, one mark and one history word; a marked group first
// takes its pending matches over (the group's matches into this iteration's slab, the new rows into the cache); then every point streams its
// coordinates and its cached row and writes its weight: 40 bytes per point.  The launch behind a search of EVERY query of a cloud (the certifying
// search: fresh; a cloud whose certificates are off or tried again) is all gather: it reads the iteration's own slab and fills the cache.
// The kernel carries no search code any more, and with it went 30 registers: occupancy, not bytes, is what these launches were short of (a chain of
// memory latencies; 94 -> 69 registers, 5 -> 7 waves per SIMD, 54 -> 31.5 us per certified launch: profiles/r05_accumulate_occupancy.txt).
constexpr bool PAIR_ROWS = true;     // (plain launches, 7 waves per SIMD: 58 -> 49 us: two lanes share the two 24-byte rows of their two points)
template <typename T, int NB> __device__ __forceinline__ void load_cached_row(const T* __restrict__ p, T* v) {
    if constexpr (NB == 6 && sizeof(T) == 4) {
        const float2* q = reinterpret_cast<const float2*>(p);      // (rows are 24 bytes: 8-byte aligned)
        const float2 a = q[0], b = q[1], c = q[2];
        v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y; v[4] = c.x; v[5] = c.y;
    } else {
#pragma unroll
        for (int k = 0; k < NB; ++k) v[k] = p[k];
    }
}
template <typename T, int NB> __device__ __forceinline__ void store_cached_row(T* __restrict__ p, const T* v) {
    if constexpr (NB == 6 && sizeof(T) == 4) {
        float2* q = reinterpret_cast<float2*>(p);
        q[0] = make_float2(v[0], v[1]); q[1] = make_float2(v[2], v[3]); q[2] = make_float2(v[4], v[5]);
    } else {
#pragma unroll
        for (int k = 0; k < NB; ++k) p[k] = v[k];
    }
}

#ifndef DICP_ACC_CERT_WAVES
#define DICP_ACC_CERT_WAVES 7
#endif
#ifndef DICP_ACC_PRELOAD
#define DICP_ACC_PRELOAD 2
#endif
template <typename T, int MODE, bool CERT = false>
__global__ __launch_bounds__(BLOCK, (CERT && sizeof(T) == 4) ? DICP_ACC_CERT_WAVES : 1) void accumulate_kernel(WeightParams P, const T* __restrict__ src, const T* __restrict__ tgt, int c /* elements per row of tgt */,
                                                           const int32_t* __restrict__ idx, const T* __restrict__ pose,
                                                           const T* __restrict__ w_init, const T* __restrict__ alive,
                                                           int N, int n, int m, int bpc, T* __restrict__ partials,
                                                           T* __restrict__ w_out, long w_stride, const int32_t* __restrict__ src_rows, AccCert<T> ps,
                                                           const T* __restrict__ w_prev /* optional: a frozen cloud (alive = 0) keeps its previous weights, ICP.py:224-226 */) {
    constexpr int NB = MODE == MODE_PT2PL ? 6 : 3;          // elements of a cached row
    constexpr int ROUNDS = ACC_PTS / BLOCK;
    __shared__ T red[(BLOCK / WAVE) * NACC_PAD];
    int cloud, blk;
    if (!decode_block(bpc, N, cloud, blk)) return;
    const int nc = rows_of(src_rows, cloud, n);             // ragged batches: rows past the cloud's own carry weight 0 (ICP.py:386-398)
    const int end = min(nc, (blk + 1) * ACC_PTS);
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // CERT: how this block's cloud is served (block-uniform).  cached: the matches are where the last iteration left them, but for what the guard
    // launch changed; otherwise the search of this iteration has just written every match of the cloud into the iteration's own slab (certifying
    // search, certificates off, certificates tried again): everything is gathered and cached.  What a wave needs to know about its (up to) four
    // groups is fetched at once, up front: the launch is a chain of memory latencies, not of bytes.
    bool cached = false;
    size_t of_k = 0;                                        // this cloud's row of `of` at iteration k
    unsigned live_mask = 0, todo_mask = 0;                  // per round of the wave: its group exists / has something to take over first (wave-uniform)
    int gslab[ROUNDS];                                      // the iteration whose slab holds the group's matches (wave-uniform)
#pragma unroll
    for (int t = 0; t < ROUNDS; ++t) gslab[t] = 0;
    if (CERT) {
        of_k = ((size_t)ps.k * ps.N + cloud) * ps.nwr;
        const int cstate = ps.cloud ? ps.cloud[(size_t)cloud * CERT_CLOUD + 2] : 0;
        int dirty[ROUNDS];
#pragma unroll
        for (int t = 0; t < ROUNDS; ++t) {                  // (read whatever the cloud's state turns out to be: one latency instead of two; the words exist)
            const int g0 = blk * ACC_PTS + t * BLOCK + wave * WAVE;
            const bool have = g0 < end;
            dirty[t] = (have && !ps.fresh) ? ps.gdirty[(size_t)cloud * ps.nwr + (g0 >> 6)] : 0;
            gslab[t] = (have && !ps.fresh && ps.of) ? ps.of[of_k + (g0 >> 6)] : ps.k;
            live_mask |= have ? (1u << t) : 0u;
        }
        cached = !ps.fresh && !(cstate > 0) && cstate != CERT_RECERTIFY;
        if (ps.cloud && !ps.fresh && blk == 0 && threadIdx.x == 0) { ps.cloud[(size_t)cloud * CERT_CLOUD + 3] = ps.units; ps.cloud[(size_t)cloud * CERT_CLOUD + 5] = ps.sets; }
        if (ps.fresh && ps.scount && blk == 0 && threadIdx.x == 0) ps.scount[cloud] = 0;       // (a new query order: the slots the set lists name are gone)
        if (cached) {
#pragma unroll
            for (int t = 0; t < ROUNDS; ++t)                // a marked group; a group whose slab belongs to the previous history chunk (copied into this iteration's)
                if (((live_mask >> t) & 1u) && (dirty[t] != 0 || (ps.of && gslab[t] < ps.k_floor))) todo_mask |= 1u << t;
        }
    }
    if (CERT && cached && todo_mask) {
#pragma unroll
        for (int t = 0; t < ROUNDS; ++t) {
            if (!((todo_mask >> t) & 1u)) continue;         // (wave-uniform)
            const int i = blk * ACC_PTS + t * BLOCK + (int)threadIdx.x;
            const int grp = (i - lane) >> 6;
            const int s = gslab[t];
            // the group's matches into this iteration's slab (a marked group: some of them change; and references never reach behind the history chunk)
            if (ps.of && s != ps.k) {
                const int32_t* from = (s < ps.k_floor ? ps.hist_prev : ps.hist) + ((size_t)s * ps.N + cloud) * n;
                if (i < end) ps.spos[(size_t)cloud * n + i] = from[i];
                if (lane == 0) ps.of[of_k + grp] = ps.k;
                gslab[t] = ps.k;
            }
            if (i < end) {
                const size_t pt = (size_t)cloud * n + i;
                const int pd = ps.pend[pt];
                if (pd != 0) {                              // the match the guard launch left: into the history, its row into the cache
                    ps.spos[pt] = pd - 2;
                    ps.pend[pt] = 0;
                    const T* yp = tgt + ((size_t)cloud * m + min(max(pd - 2, 0), m - 1)) * c;
                    T row[NB];
#pragma unroll
                    for (int e = 0; e < NB; ++e) row[e] = yp[e];
                    store_cached_row<T, NB>(ps.nbr + pt * NB, row);
                }
            }
            if (lane == 0) ps.gdirty[(size_t)cloud * ps.nwr + grp] = 0;
        }
    }
    T C[9], r[3];
    load_pose(pose, cloud, C, r);
    const T live = alive ? alive[cloud] : T(1);
    T acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = T(0);
    if (w_out)                                              // ... which is what the weight history reports for them
        for (int i = max(blk * ACC_PTS, nc) + threadIdx.x; i < min(n, (blk + 1) * ACC_PTS); i += BLOCK) w_out[(size_t)cloud * w_stride + i] = T(0);
    const int32_t* __restrict__ ix = CERT ? ps.spos : idx;
    // one point from its row: the sums, the weight
    auto point = [&](const T* p, const T* y, const T* nrm, size_t pt, int i) {
        PointState<T> s;
        point_forward<T, MODE>(P, C, r, p, y, nrm, (w_init ? w_init[pt] : T(1)) * live, acc, s);
        // (a frozen cloud: all its weights are zero, and the reference then keeps the previous iteration's -- written here, by 1024 threads per
        //  block instead of the step kernel's one wave per cloud: 97 us of every tolerance-mode iteration at the benchmark shape)
        if (w_out) w_out[(size_t)cloud * w_stride + i] = (w_prev && live == T(0)) ? w_prev[(size_t)cloud * w_stride + i] : s.w;
    };
    if (CERT && cached) {
        // Every thread sums its rounds in the order 0, 1, 2, 3, like a launch that gathers everything: the sums are bit for bit the same.  All
        // loads of all rounds go out before the first sum -- unconditionally, so that nothing but arithmetic lies between them (a round past the
        // block's end re-reads the last point's; branches between the rounds made the compiler wait for each round's loads before issuing the next)
        constexpr int PRE = DICP_ACC_PRELOAD < ROUNDS ? DICP_ACC_PRELOAD : ROUNDS;      // rounds whose loads are in flight together
#pragma unroll
        for (int t0 = 0; t0 < ROUNDS; t0 += PRE) {
            T pp[PRE][3], rw[PRE][NB];
#pragma unroll
            for (int u = 0; u < PRE; ++u) {
                const int i = blk * ACC_PTS + (t0 + u) * BLOCK + (int)threadIdx.x;
                const size_t pt = (size_t)cloud * n + min(i, end - 1);
                const T* sp = src + pt * 3;
                pp[u][0] = sp[0]; pp[u][1] = sp[1]; pp[u][2] = sp[2];
                load_cached_row<T, NB>(ps.nbr + pt * NB, rw[u]);
            }
#pragma unroll
            for (int u = 0; u < PRE; ++u) {
                const int t = t0 + u;
                const int i = blk * ACC_PTS + t * BLOCK + (int)threadIdx.x;
                if (ps.of && lane == 0 && ((live_mask >> t) & 1u))      // the next iteration finds the group's matches where this one did
                    ps.of[of_k + (size_t)ps.N * ps.nwr + ((i - lane) >> 6)] = gslab[t];      // (`of` has K + 1 rows)
                if (i < end) {
                    const T nrm[3] = {MODE == MODE_PT2PL ? rw[u][NB - 3] : T(0), MODE == MODE_PT2PL ? rw[u][NB - 2] : T(0), MODE == MODE_PT2PL ? rw[u][NB - 1] : T(0)};
                    point(pp[u], rw[u], nrm, (size_t)cloud * n + i, i);
                }
            }
        }
    } else {
        for (int base = blk * ACC_PTS; base < end; base += BLOCK) {        // (2 or 4 points in flight per thread measured slower: 78 / 85 vs 72 us)
            const int i = base + (int)threadIdx.x;
            const bool on = i < end;
            const size_t pt = (size_t)cloud * n + (on ? i : end - 1);
            const T* sp = src + pt * 3;
            const T p[3] = {sp[0], sp[1], sp[2]};
            T y[3], nrm[3] = {T(0), T(0), T(0)};
            const int jm = ix ? ix[pt] : (on ? i : end - 1);    // ix == NULL: tgt holds one row per source point
            const int j = min(max(jm, 0), m - 1);
            if (MODE == MODE_PT2PL && PAIR_ROWS) {
                // The 24-byte row gather: two lanes share the two rows of their two points -- each loads its half (12 bytes) of both, so a wave
                // instruction touches 32 rows instead of 64 (the gather is bound by the cache's look-ups per instruction, not by bytes), and the
                // halves change hands inside the lane pair.
                const int half = threadIdx.x & 1;
                const int je = __shfl(j, lane & ~1), jo = __shfl(j, lane | 1);
                const T* re = tgt + ((size_t)cloud * m + je) * c + 3 * half;
                const T* ro = tgt + ((size_t)cloud * m + jo) * c + 3 * half;
                const T e[3] = {re[0], re[1], re[2]}, o[3] = {ro[0], ro[1], ro[2]};
                const T pe[3] = {__shfl_xor(e[0], 1), __shfl_xor(e[1], 1), __shfl_xor(e[2], 1)};
                const T po[3] = {__shfl_xor(o[0], 1), __shfl_xor(o[1], 1), __shfl_xor(o[2], 1)};
                if (half == 0) { y[0] = e[0]; y[1] = e[1]; y[2] = e[2]; nrm[0] = pe[0]; nrm[1] = pe[1]; nrm[2] = pe[2]; }
                else           { y[0] = po[0]; y[1] = po[1]; y[2] = po[2]; nrm[0] = o[0]; nrm[1] = o[1]; nrm[2] = o[2]; }
            } else {
                const T* yp = tgt + ((size_t)cloud * m + j) * c;
                y[0] = yp[0]; y[1] = yp[1]; y[2] = yp[2];
                if (MODE == MODE_PT2PL) { nrm[0] = yp[3]; nrm[1] = yp[4]; nrm[2] = yp[5]; }
            }
            if (CERT && (i - lane) < end) {                 // the group's rows into the cache; its matches lie in this iteration's slab
                if (on) {
                    T row[NB];
                    row[0] = y[0]; row[1] = y[1]; row[2] = y[2];
                    if (MODE == MODE_PT2PL) { row[3] = nrm[0]; row[4] = nrm[1]; row[5] = nrm[2]; }
                    store_cached_row<T, NB>(ps.nbr + pt * NB, row);
                }
                if (lane == 0) {
                    const int grp = (i - lane) >> 6;
                    ps.gdirty[(size_t)cloud * ps.nwr + grp] = 0;
                    if (ps.of) { ps.of[of_k + grp] = ps.k; ps.of[of_k + (size_t)ps.N * ps.nwr + grp] = ps.k; }
                }
            }
            if (on) point(p, y, nrm, pt, i);
        }
    }
    block_reduce_store<T, NACC, NACC_PAD>(acc, partials + ((size_t)cloud * bpc + blk) * NACC_PAD, red);
}

// -------------------------------------------------------------------------- step
// One 64-thread block per cloud.  All small matrices live in LDS: private arrays with dynamic indexing
// would be scratch (global) memory, and this kernel is pure latency (it sits between two big launches).
// dicp_step_io of iteration k of a dicp_icp_forward chunk [k0, k1): one place for the host loop and the small-cloud kernel
__host__ __device__ inline dicp_step_io make_step_io(const dicp_loop_buffers& B, int k, int k0, int N, int n, int mode, int dim,
                                                     int const_iter, double tolerance, size_t es, int nblk) {
    dicp_step_io io;
    io.partials = B.partials; io.nblk = nblk; io.iter = k; io.dim = dim; io.const_iter = const_iter; io.tolerance = tolerance;
    io.rows_per_point = mode == DICP_PT2PT ? 3 : 1; io.n = n;
    io.pose_in = (const char*)B.hist.poses + (size_t)k * N * 12 * es; io.pose_out = (char*)B.hist.poses + (size_t)(k + 1) * N * 12 * es;
    io.frame = B.search.frame; io.pose_search_out = B.search.poses ? (char*)B.search.poses + (size_t)(k + 1) * N * 12 * es : nullptr;
    io.delta = (char*)B.hist.deltas + (size_t)k * 6 * es; io.delta_stride = (int64_t)B.K * 6;
    io.cost = (char*)B.hist.costs + (size_t)k * es; io.cost_prev = k > 0 ? (const char*)B.hist.costs + (size_t)(k - 1) * es : nullptr;
    io.cost_stride = B.K;
    io.areg = B.hist.areg ? B.hist.areg + (size_t)k * N * 36 : nullptr;
    io.alive = (const char*)B.hist.alive + (size_t)k * N * es; io.alive_out = (char*)B.hist.alive + (size_t)(k + 1) * N * es;
    io.converged = B.converged; io.iterations = B.iterations; io.matched_ratio = B.matched_ratio;
    io.n_start = B.n_start; io.n_matched = B.n_matched;
    io.w_cur = (char*)B.hist.w + (size_t)k * B.hist.w_iter * es;
    io.w_prev = k > k0 ? (const char*)B.hist.w + (size_t)(k - 1) * B.hist.w_iter * es : (const char*)B.hist.w_prev0; io.w_stride = B.hist.w_stride;
    io.n_not_converged = B.counters + k;
    io.rmax = B.cert.rmax; io.dcum = B.cert.dcum; io.dcum_stride = 2 * (B.K + 1);
    io.cert_cloud = B.cert.cloud;
    io.cert_qu = nullptr; io.cert_units = 0; io.glist_cap = 0; io.glist = nullptr; io.gcount = nullptr; io.cert_scount = nullptr; io.cert_slist = nullptr;      // (dicp_icp_forward fills them in for a certified iteration)
    io.w_copied = 0;
    return io;
}

// The step of one cloud, run by a whole block of NT >= 64 threads (its first wave does the work, everybody joins the
// barriers): the body of step_kernel, and of the small-cloud kernel that keeps a cloud in one block for a whole chunk.
template <typename T, int NT>
__device__ __forceinline__ void step_body(const dicp_step_io& io, int cloud, int tid) {
    __shared__ double sacc[NACC_PAD], sA[36], sAreg[36], spose[12], sout[24], smisc[16];
    __shared__ T sframe[12];
    __shared__ int scc[7];
    __shared__ int s_copy, s_next_state;
    __shared__ T s_dnext[2];
    if (tid < WAVE) {   // reduce the per-block partials: lane = (part, slot); fixed summation order -> bit-reproducible
        const int slot_i = tid & 31, part = tid >> 5;
        const T* pp = (const T*)io.partials + (size_t)cloud * io.nblk * NACC_PAD + slot_i;
        double s = 0.0;
        // (the kernel is a chain of latencies: all of a lane's loads are issued before the first add -- the order of the adds is unchanged)
        constexpr int UB = 8;
        for (int b0 = part; b0 < io.nblk; b0 += 2 * UB) {
            T v[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) v[u] = (b0 + 2 * u < io.nblk) ? pp[(size_t)(b0 + 2 * u) * NACC_PAD] : T(0);
#pragma unroll
            for (int u = 0; u < UB; ++u) if (b0 + 2 * u < io.nblk) s += (double)v[u];
        }
        s += __shfl_down(s, 32);
        if (tid < NACC_PAD) sacc[tid] = s;
        if (tid < 12) spose[tid] = (double)((const T*)io.pose_in)[(size_t)cloud * 12 + tid];
        // the scalars the serial part below needs, fetched by idle lanes while the partials arrive
        if (tid == 40) smisc[0] = (double)((const T*)io.alive)[cloud];
        if (tid == 41) smisc[1] = io.cost_prev ? (double)((const T*)io.cost_prev)[(size_t)cloud * io.cost_stride] : 0.0;
        if (tid >= 42 && tid < 46) smisc[2 + (tid - 42)] = io.dcum ? (double)((const T*)io.rmax)[(size_t)cloud * 4 + (tid - 42)] : 0.0;
        if (tid >= 46 && tid < 49) smisc[6 + (tid - 46)] = io.frame ? (double)((const T*)io.frame)[(size_t)cloud * 12 + 9 + (tid - 46)] : 0.0;   // t of the search frame
        if (tid == 49) smisc[9] = io.dcum ? (double)((const T*)io.dcum)[(size_t)cloud * io.dcum_stride + 2 * io.iter] : 0.0;
        if (tid == 50) smisc[10] = (double)((const T*)io.n_start)[cloud];
        if (tid == 51) smisc[11] = (double)((const T*)io.iterations)[cloud];
        if (tid == 52) smisc[12] = (double)((const T*)io.matched_ratio)[cloud];
        if (tid >= 12 && tid < 24 && io.frame) sframe[tid - 12] = ((const T*)io.frame)[(size_t)cloud * 12 + (tid - 12)];
        if (tid >= 24 && tid < 31 && io.cert_cloud) scc[tid - 24] = io.cert_cloud[(size_t)cloud * CERT_CLOUD + (tid - 24)];
    }
    __syncthreads();
    if (tid == 0) {
        double* d6 = sout;
        unpack_sym6(sacc + ACC_A, sA);
        // the solve alone first: delta is rounded to T like the reference's before it moves the pose
        step_solve(sA, sacc + ACC_B, io.dim, d6, sAreg);
        T* dout = (T*)io.delta + (size_t)cloud * io.delta_stride;
        double nrm2 = 0.0;
        for (int k = 0; k < 6; ++k) { const T v = (T)d6[k]; dout[k] = v; d6[k] = (double)v; nrm2 += d6[k] * d6[k]; }
        double R[9];
        so3_exp(d6, R);                                                   // ICP.py:210
        T* pout = (T*)io.pose_out + (size_t)cloud * 12;
        const double* C = spose;
        T pn[12];                                                         // the new pose, kept in registers for what follows
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                pn[i * 3 + j] = (T)(R[0 * 3 + i] * C[0 * 3 + j] + R[1 * 3 + i] * C[1 * 3 + j] + R[2 * 3 + i] * C[2 * 3 + j]);
        for (int k = 0; k < 3; ++k) pn[9 + k] = (T)(spose[9 + k] - d6[3 + k]);
        for (int k = 0; k < 12; ++k) pout[k] = pn[k];
        if (io.dcum) {      // match certificates: (M, e) of the new pose.  M += how far a query of this cloud can have moved between the two
                            // poses: dC p + dr = dC (p - p0) + (dC p0 + dr) <= |dC|_F radius + |dC p0 + dr|, rounded up (radius, p0: the cloud's
                            // bounding box from dicp_loop_init); e = the rounding of a point transformed with the new pose
            const double rad = smisc[2], p0[3] = {smisc[3], smisc[4], smisc[5]};
            double dC = 0.0, mv = 0.0, rn2 = 0.0;
            for (int i = 0; i < 3; ++i) {
                double m = (double)pn[9 + i] - spose[9 + i];
                for (int j = 0; j < 3; ++j) { const double d = (double)pn[i * 3 + j] - spose[i * 3 + j]; dC += d * d; m += d * p0[j]; }
                mv += m * m;
                rn2 += (double)pn[9 + i] * (double)pn[9 + i];
            }
            const double ulp = sizeof(T) == 4 ? 1.2e-7 : 2.3e-16;
            T* dc = (T*)io.dcum + (size_t)cloud * io.dcum_stride + 2 * io.iter;
            const T nxt = (T)((double)(T)smisc[9] + (sqrt(dC) * rad + sqrt(mv)) * 1.0001);
            dc[2] = nxt + m_abs(nxt) * (T)(4.0 * ulp);                    // (rounded up)
            s_dnext[0] = dc[2];
            const double cn = io.frame ? sqrt(smisc[6] * smisc[6] + smisc[7] * smisc[7] + smisc[8] * smisc[8]) : 0.0;      // (the search frame adds t, |t| = |centre|, to r)
            const double pnm = sqrt(p0[0] * p0[0] + p0[1] * p0[1] + p0[2] * p0[2]);
            dc[3] = (T)(8.0 * ulp * (pnm + rad + sqrt(rn2) + cn + 1.0) * 1.0001);
            s_dnext[1] = dc[3];
        }
        if (io.pose_search_out) {                                         // what the next search reads: [Q C | Q r + t] (the cloud's search frame)
            T* ps = (T*)io.pose_search_out + (size_t)cloud * 12;
            T Fr[12];                                                     // (fetched with the partials: this lane's chain waits for no load)
            for (int k = 0; k < 12; ++k) Fr[k] = io.frame ? sframe[k] : T(0);
            for (int k = 0; k < 12; ++k) ps[k] = frame_pose_entry<T>(io.frame ? Fr : nullptr, pn, k);
        }

        T cost = (T)sacc[ACC_COST];                                       // ICP.py:229-232
        if (io.cost_prev && cost == T(0)) cost = (T)smisc[1];
        ((T*)io.cost)[(size_t)cloud * io.cost_stride] = cost;

        const double nmatch = sacc[ACC_NMATCH];
        if (io.n_matched) ((T*)io.n_matched)[cloud] = (T)nmatch;
        const T alive_in = (T)smisc[0];
        T alive_next = alive_in;
        const bool hit = (double)(T)sqrt(nrm2) < io.tolerance;            // ICP.py:237-239
        if (hit) io.converged[cloud] = 1;
        else if (io.n_not_converged) atomicAdd(io.n_not_converged, 1);
        if (hit && !io.const_iter) {                                      // ICP.py:240-257
            T* it = (T*)io.iterations + cloud;
            if ((T)smisc[11] == T(0)) *it = (T)(io.iter + 1);
            T* mr = (T*)io.matched_ratio + cloud;
            if ((T)smisc[12] == T(0)) {
                float start = (alive_in != T(0)) ? (float)(T)smisc[10] : 0.f;
                if (start == 0.f) start = 1.f;
                *mr = (T)((float)nmatch / start);       // int64/int64 -> float32 in the reference
            }
            alive_next = T(0);
        }
        ((T*)io.alive_out)[cloud] = alive_next;
        s_copy = (io.w_cur && io.w_prev && sacc[ACC_SUMW] == 0.0 && !(io.w_copied && alive_in == T(0))) ? 1 : 0;   // ICP.py:224-226
        s_next_state = io.cert_cloud ? scc[2] : 0;
        if (io.cert_cloud) {
            // Match certificates must never cost more than searching everything.  What this iteration searched again for this cloud --
            // whole units (a certifying search of a unit costs ~1.3 plain ones) and single queries (one wave per query: ~0.12 of a unit's
            // search each, measured on planar scenes once the searches' statistics were counted per wave at the kernel's end, profiles/r03_scene_kernel_stats_tally.txt) -- against the
            // cloud's units: from 60 % of a full search on, the cloud's certificates are switched off for the rest of the call (the guard
            // launch then searches every unit plainly, the accumulate checks nothing).  Results do not depend on it: both are exact.
            // Two kinds of evidence.  Queries that got NO certificate in a search of every unit (no unit was searched AGAIN: cc[0] == 0) are
            // structural -- near-ties inside the rounding bound of a score, searched one by one in every iteration from now on: the cloud is
            // switched off for good.  A guarded iteration that searched much again counts as a strike; on the second in a row the cloud
            // is switched off for a while -- 2 iterations, doubling up to 16 -- and then certified afresh (CERT_RECERTIFY: one guard launch
            // of certifying sweeps): a cloud that is still moving when the certificates start must get them back once it has settled.
            int32_t* cc = io.cert_cloud + (size_t)cloud * CERT_CLOUD;
            const int c_units = scc[0], c_single = scc[1], state = scc[2], units = scc[3], c_back = scc[4];     // (read in the prologue)
            const bool sets = scc[5] != 0;      // candidate sets are kept: a query without a certificate of its own is searched ONCE more (for its set), not in every iteration
            if (units > 0) {
                const bool costly = 1.3 * c_units + 0.12 * c_single > 0.6 * units;
                int next = state;
                if (state >= CERT_OFF_FOR_GOOD) next = state;
                else if (state > 0) next = state > 1 ? state - 1 : CERT_RECERTIFY;
                else if (state == CERT_RECERTIFY) next = sets ? (costly ? -1 : 0) : ((0.12 * c_single > 0.6 * units) ? CERT_OFF_FOR_GOOD : 0);
                // (sets: making them costs one single-query search per query without a certificate, re-scoring them a twelfth of that per iteration --
                //  against one full search per iteration that only pays while such queries are the minority)
                else if (sets && 2 * scc[6] > io.n) next = CERT_OFF_FOR_GOOD;
                else if (!costly) next = 0;
                else if (c_units == 0 && !sets) next = CERT_OFF_FOR_GOOD;
                else if (state == -1 && sets && c_units == 0) next = CERT_OFF_FOR_GOOD;      // twice in a row costly by per-query work alone (no unit moved): structural
                else if (state == -1) { const int d = c_back > 0 ? min(2 * c_back, 16) : 2; cc[4] = d; next = d; }
                else next = -1;
                cc[2] = next;
                s_next_state = next;
                if (costly || state > 0) cc[7] += 1;            // iterations of this call in which the cloud's certificates did not pay (the host's call-to-call hint reads it)
                cc[0] = 0; cc[1] = 0; cc[3] = 0; cc[6] = 0;
            }
        }
    }
    __syncthreads();
    if (io.cert_qu && io.glist && tid < WAVE) {
        // the next iteration's guard launch: which of this cloud's units it has to look at (knn_sweep_guard_kernel decides again, from the same
        // numbers: the list only has to hold every unit it would not leave at once)
        const T spent = cert_spent<T>(s_dnext);
        const int st = s_next_state;
        const bool certs_on = st <= 0 && st != CERT_RECERTIFY;
        const T* qu = (const T*)io.cert_qu + (size_t)cloud * io.cert_units;
        int32_t* list = io.glist + (size_t)(cloud & 7) * io.glist_cap;
        for (int u0 = 0; u0 < io.cert_units; u0 += WAVE) {
            const int u = u0 + tid;
            bool work = false;
            if (u < io.cert_units) { const T v = qu[u]; work = !(certs_on && v >= T(0) && v > spent); }
            const unsigned long long mk = __ballot(work);
            if (mk) {                                               // (wave-uniform)
                int base = 0;
                if (tid == 0) base = atomicAdd(io.gcount + (cloud & 7), __popcll(mk));
                base = __shfl(base, 0);
                const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u));
                if (work && base + rank < io.glist_cap) list[base + rank] = cloud * io.cert_units + u;
            }
        }
        // ... and the cloud's standing candidate sets, 64 to an entry (guard_sets); a cloud that is searched as a whole has none left afterwards
        if (io.cert_scount) {
            if (!certs_on) { if (tid == 0) io.cert_scount[cloud] = 0; }
            else {
                const int nwr = (io.n + WAVE - 1) / WAVE;
                const int have = min(io.cert_scount[cloud], io.n);
                const int nch = (have + WAVE - 1) / WAVE;
                // The list is handed over in WHOLE chunks: the tail of the last one is filled with -1 (no slot) and the length moved up to it.  The guard launch
                // that re-scores these chunks also APPENDS to the list (a query that gets a new set): an append inside a chunk another of its waves is reading
                // showed that wave an entry whose slot, set budget and candidate rows were not written yet -- whatever the memory held, and a row gather from
                // there (round 6: a memory fault once in a few hundred calls on planar scenes, as soon as the buffers stopped being the previous call's own).
                if (io.cert_slist && nch > 0) {
                    const int upto = min(nch * WAVE, io.n);
                    for (int e = have + tid; e < upto; e += WAVE) io.cert_slist[(size_t)cloud * io.n + e] = -1;
                    if (tid == 0 && upto > have) io.cert_scount[cloud] = upto;
                }
                if (nch > 0) {
                    int base = 0;
                    if (tid == 0) base = atomicAdd(io.gcount + (cloud & 7), nch);
                    base = __shfl(base, 0);
                    for (int c = tid; c < nch; c += WAVE)
                        if (base + c < io.glist_cap) list[base + c] = -1 - (cloud * nwr + c);
                }
            }
        }
    }
    if (io.areg && tid < 36) io.areg[(size_t)cloud * 36 + tid] = sAreg[tid];
    if (s_copy) {
        T* wc = (T*)io.w_cur + (size_t)cloud * io.w_stride;
        const T* wp = (const T*)io.w_prev + (size_t)cloud * io.w_stride;
        for (int i = tid; i < io.n; i += NT) wc[i] = wp[i];
    }
}

template <typename T>
__global__ __launch_bounds__(WAVE) void step_kernel(dicp_step_io io, int N) {
    step_body<T, WAVE>(io, blockIdx.x, threadIdx.x);
}

// The convergence counters of iterations [k0, k0 + cnt) to the host's mapped words (dicp_loop_buffers.counters_host): system-scope stores, tagged with the call
__global__ __launch_bounds__(WAVE) void counters_report_kernel(const int32_t* __restrict__ counters, int32_t* __restrict__ host, int k0, int cnt, int tag) {
    for (int k = threadIdx.x; k < cnt; k += WAVE) {
        const int v = __hip_atomic_load(counters + k0 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(host + k0 + k, (tag & 0x7ff00000) | (v < 0xfffff ? v : 0xfffff), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ------------------------------------------------------------ whole loop, small clouds
// Clouds of a few hundred points (the reference's own 65-point test pair; batches of many small scans) are pure
// launch latency on the multi-kernel path: 3 dependent launches per iteration, each a few microseconds of work.
// Here ONE block owns a cloud for a whole chunk of iterations: packed targets staged in LDS once, then per iteration
// brute-force 1-NN (same score arithmetic and lowest-index rule as every other form), the accumulate pass, the block
// reduction and the step (step_body), with the pose handed from one iteration to the next through the pose history.
template <typename T, int MODE>
__global__ __launch_bounds__(BLOCK) void icp_small_forward_kernel(WeightParams P, dicp_loop_buffers B, int N, int n, int m, int dim,
                                                                  int const_iter, double tolerance, int k0, int k1) {
    using T4 = typename V4<T>::type;
    extern __shared__ __align__(32) unsigned char small_lds[];
    T4* tg = reinterpret_cast<T4*>(small_lds);
    __shared__ T red[(BLOCK / WAVE) * NACC_PAD];
    const int cloud = blockIdx.x, tid = threadIdx.x, c = B.c;
    const int nc = rows_of(B.src_rows, cloud, n), mc = max(rows_of(B.tgt_rows, cloud, m), 1);
    const int m_pad = min((mc + KNN_PAD - 1) / KNN_PAD * KNN_PAD, B.search.m_pad);     // ragged batches: the cloud's own rows only
    {
        const T4* __restrict__ g = (const T4*)B.search.tgt4 + (size_t)cloud * B.search.m_pad;
        for (int j = tid; j < m_pad; j += BLOCK) tg[j] = g[j];
    }
    __syncthreads();
    const T* __restrict__ src = (const T*)B.src + (size_t)cloud * n * 3;
    const T* __restrict__ tgt = (const T*)B.tgt + (size_t)cloud * m * c;
    const T* __restrict__ w_init = B.w_init ? (const T*)B.w_init + (size_t)cloud * n : nullptr;
    for (int k = k0; k < k1; ++k) {
        T C[9], r[3];
        load_pose((const T*)B.hist.poses + (size_t)k * N * 12, cloud, C, r);
        T Cs[9], rs[3];                                     // the search's pose: [Q C | Q r + t] (packed rows are Q y + t)
        {
            const T pw[12] = {C[0], C[1], C[2], C[3], C[4], C[5], C[6], C[7], C[8], r[0], r[1], r[2]};
            const T* F = B.search.frame ? (const T*)B.search.frame + (size_t)cloud * 12 : nullptr;
#pragma unroll
            for (int e = 0; e < 9; ++e) Cs[e] = frame_pose_entry<T>(F, pw, e);
#pragma unroll
            for (int e = 0; e < 3; ++e) rs[e] = frame_pose_entry<T>(F, pw, 9 + e);
        }
        const T live = ((const T*)B.hist.alive)[(size_t)k * N + cloud];
        int32_t* __restrict__ idx_k = B.hist.idx + (B.hist.per_iter ? (size_t)k * N * n : 0) + (size_t)cloud * n;
        T* __restrict__ w_k = (T*)B.hist.w + (size_t)k * B.hist.w_iter + (size_t)cloud * B.hist.w_stride;
        T acc[NACC];
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = T(0);
        for (int i = nc + tid; i < n; i += BLOCK) w_k[i] = T(0);
        for (int i = tid; i < nc; i += BLOCK) {
            const T p[3] = {src[i * 3], src[i * 3 + 1], src[i * 3 + 2]};
            T nx[3];
            query_point(Cs, rs, p, nx);
            T best = inf_v<T>();
            int bj = 0;
            for (int j = 0; j < m_pad; j += 4) {            // m_pad is a multiple of 64; ascending, strict <: lowest index on ties
                const T s0 = score<T, T4>(nx, tg[j]), s1 = score<T, T4>(nx, tg[j + 1]);
                const T s2 = score<T, T4>(nx, tg[j + 2]), s3 = score<T, T4>(nx, tg[j + 3]);
                if (s0 < best) { best = s0; bj = j; }
                if (s1 < best) { best = s1; bj = j + 1; }
                if (s2 < best) { best = s2; bj = j + 2; }
                if (s3 < best) { best = s3; bj = j + 3; }
            }
            bj = min(bj, mc - 1);
            idx_k[i] = bj;
            const T* yp = tgt + (size_t)bj * c;
            const T y[3] = {yp[0], yp[1], yp[2]};
            T nrm[3] = {T(0), T(0), T(0)};
            if (MODE == MODE_PT2PL) { nrm[0] = yp[3]; nrm[1] = yp[4]; nrm[2] = yp[5]; }
            PointState<T> st;
            point_forward<T, MODE>(P, C, r, p, y, nrm, (w_init ? w_init[i] : T(1)) * live, acc, st);
            w_k[i] = st.w;
        }
        block_reduce_store<T, NACC, NACC_PAD>(acc, (T*)B.partials + (size_t)cloud * NACC_PAD, red);
        __threadfence_block();
        __syncthreads();
        const dicp_step_io io = make_step_io(B, k, k0, N, n, MODE == MODE_PT2PT ? DICP_PT2PT : DICP_PT2PL, dim, const_iter, tolerance, sizeof(T), 1);
        step_body<T, BLOCK>(io, cloud, tid);
        __threadfence_block();                              // pose / alive / weights of iteration k+1 are read next
        __syncthreads();
    }
}
