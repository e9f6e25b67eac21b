// libdicp_hip.so -- backward: accumulate_bwd (row atomics), windowed form + window reduce, step_bwd, small-cloud backward, one-launch tail.
// Part of the one translation unit dicp_kernels.hip (included inside its anonymous namespace, in this order: kernels_setup.h, kernels_search.h, kernels_setup_sort.h, kernels_rows.h, kernels_accumulate.h, kernels_backward.h, kernels_soft_svd.h, kernels_host.h).
// ---------------------------------------------------------------- accumulate bwd
// Target gradients are a scatter-add of one 12/24-byte row per source point.  Float atomics execute at the
// memory side in 64-byte requests, and 64 lanes adding to 64 different rows cost 64 requests per
// instruction (MI355X_MICROARCH.md, Global float atomics).  So each wave first transposes its 64 rows
// through LDS: in the add instructions lane l carries element l of the flattened [point][column] list, i.e.
// the CV floats of one row sit in CV consecutive lanes and leave L2 as one (sometimes two) requests.
template <typename T, int MODE>
__global__ __launch_bounds__(BLOCK) void accumulate_bwd_kernel(WeightParams P, const T* __restrict__ src, const T* __restrict__ tgt, int c,
                                                               const int32_t* __restrict__ idx, const T* __restrict__ pose,
                                                               const T* __restrict__ w_init, const T* __restrict__ alive,
                                                               const T* __restrict__ gs, const T* __restrict__ gb,
                                                               int N, int n, int m, int bpc,
                                                               T* __restrict__ gsrc, T* __restrict__ gtgt, T* __restrict__ gw,
                                                               T* __restrict__ bwd_partials, const int32_t* __restrict__ src_rows,
                                                               const int32_t* __restrict__ skip /* optional (N): step_bwd found this iteration's cotangent negligible */) {
    constexpr int CV = (MODE == MODE_PT2PL) ? 6 : 3;        // gradient columns per target row
    __shared__ T red[(BLOCK / WAVE) * NBWD_PAD];
    __shared__ T stage_v[(BLOCK / WAVE) * WAVE * CV];
    __shared__ int stage_j[BLOCK];
    int cloud, blk;
    if (!decode_block(bpc, N, cloud, blk)) return;
    if (skip && skip[cloud]) {                              // nothing this cloud would add is above rounding: zero sums for the next step_bwd, done
        if (threadIdx.x < NBWD_PAD) bwd_partials[((size_t)cloud * bpc + blk) * NBWD_PAD + threadIdx.x] = T(0);
        return;
    }
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid >> 6;
    T C[9], r[3], Gs[36], Gb[6];
    load_pose(pose, cloud, C, r);
#pragma unroll
    for (int k = 0; k < 36; ++k) Gs[k] = gs[(size_t)cloud * 36 + k];
#pragma unroll
    for (int k = 0; k < 6; ++k) Gb[k] = gb[(size_t)cloud * 6 + k];
    const T live = alive ? alive[cloud] : T(1);
    T acc[NBWD];
#pragma unroll
    for (int k = 0; k < NBWD; ++k) acc[k] = T(0);
    T* sv = stage_v + wave * (WAVE * CV);
    int* sj = stage_j + wave * WAVE;
    T* grow = gtgt ? gtgt + (size_t)cloud * m * c : nullptr;
    const int end = min(rows_of(src_rows, cloud, n), (blk + 1) * ACC_PTS);     // (rows past the cloud's own: weight 0, no gradient)
    for (int base = blk * ACC_PTS; base < end; base += BLOCK) {     // trip count is block-uniform
        const int i = base + tid;
        const bool on = i < end;
        T gy[3] = {T(0), T(0), T(0)}, gn[3] = {T(0), T(0), T(0)};
        int j = -1;
        if (on) {
            const size_t pt = (size_t)cloud * n + i;
            const T* sp = src + pt * 3;
            const T p[3] = {sp[0], sp[1], sp[2]};
            j = idx ? min(max(idx[pt], 0), m - 1) : i;
            const T* yp = tgt + ((size_t)cloud * m + j) * c;
            const T y[3] = {yp[0], yp[1], yp[2]};
            T nrm[3] = {T(0), T(0), T(0)};
            if (MODE == MODE_PT2PL) { nrm[0] = yp[3]; nrm[1] = yp[4]; nrm[2] = yp[5]; }
            T gp[3], gw0;
            point_backward<T, MODE>(P, C, r, p, y, nrm, (w_init ? w_init[pt] : T(1)) * live, Gs, Gb, gp, gy, gn, gw0, acc, acc + 9);
            T* gsp = gsrc + pt * 3;
            gsp[0] += gp[0]; gsp[1] += gp[1]; gsp[2] += gp[2];
            if (gw) gw[pt] += gw0 * live;
        }
        if (grow) {
            sj[lane] = j;
            sv[lane * CV + 0] = gy[0]; sv[lane * CV + 1] = gy[1]; sv[lane * CV + 2] = gy[2];
            if (MODE == MODE_PT2PL) { sv[lane * CV + 3] = gn[0]; sv[lane * CV + 4] = gn[1]; sv[lane * CV + 5] = gn[2]; }
            __builtin_amdgcn_wave_barrier();                // same-wave LDS hand-off: DS ops retire in order
#pragma unroll
            for (int t = 0; t < CV; ++t) {
                const int e = t * WAVE + lane;
                const int pnt = e / CV, col = e - pnt * CV;
                const int jj = sj[pnt];
                if (jj >= 0) unsafeAtomicAdd(&grow[(size_t)jj * c + col], sv[e]);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    block_reduce_store<T, NBWD, NBWD_PAD>(acc, bwd_partials + ((size_t)cloud * bpc + blk) * NBWD_PAD, red);
}

// Windowed form of the same backward, for the sorted-sweep path.  Everything is in SORTED space: slot s of a
// cloud is the s-th query in the x-order the sweep used, spos[s] the sorted position of its neighbour, and
// src_s / w_s / tgt_s are the caller's copies in those orders, so every stream is coalesced.  Queries that are
// neighbours in x match targets that are neighbours in x: a block of `spb` consecutive slots covers a window of
// WT consecutive sorted target rows.  Its threads leave their target-row contributions in LDS and thread each one
// onto a per-row list (ONE LDS exchange per slot: head[row] <-> slot); then every row is summed by the one thread
// that owns it and added to the block's OWN slab (N, blocks, WT, CV) with plain read-modify-writes -- no float
// atomics anywhere on the common path.  Measured at the benchmark shape: global float atomics for the flush cost
// 0.145 ms per launch and do not overlap the streams (a CU's vector-memory path is in order), and 6 LDS float
// atomics per slot (ds_add_f32) cost 0.10 ms -- about 137 cycles per wave-instruction.
// The windows of neighbouring blocks overlap; dicp_window_reduce sums the slabs into the target gradient once per
// call.  The window origins come from spos_ref (the matches of ONE reference iteration, the same for every launch
// that adds into a slab), so a slab row means the same target row in every iteration.  A match outside the window
// (outlier, or an iteration whose matches moved) goes to gts_far with atomics: locality only decides the speed.
template <typename T> struct WindowRows;
template <> struct WindowRows<float>  { static constexpr int v = 1536; };    // 36 KiB of rows at 6 columns: 4 blocks per CU
template <> struct WindowRows<double> { static constexpr int v = 768; };

// slots per block: two thirds of the window for the span of the block's own slots (slots * m/n sorted targets),
// one third for the spread of the matches around the diagonal (measured at the benchmark shape: median 80 rows,
// 99th percentile 486)
__host__ __device__ inline int window_slots(int WT, int n, int m_pad) {
    long s = (long)(WT - WT / 3) * n / (m_pad > 0 ? m_pad : 1);
    s = (s / BLOCK) * BLOCK;
    return (int)(s < BLOCK ? BLOCK : (s > 4 * BLOCK ? 4 * BLOCK : s));        // <= SPB of the kernel
}

// Where the matches of one iteration are found: a plain (N,n) array (of == NULL: base itself), or the history kept by reference
// (dicp_loop_buffers.hist.spos_of): the matches of queries [64g, 64g+64) of cloud b at iteration k lie in the slab of iteration of[(k N + b) nwr + g].
struct MatchHist { const int32_t* base; const int32_t* of; int k, N, n, nwr; };
__device__ __forceinline__ int match_at(const MatchHist& h, int cloud, int q) {
    if (!h.of) return h.base[(size_t)cloud * h.n + q];
    const int s = h.of[((size_t)h.k * h.N + cloud) * h.nwr + (q >> 6)];
    return h.base[((size_t)s * h.N + cloud) * h.n + q];
}
__host__ __device__ inline MatchHist plain_matches(const int32_t* spos, int N, int n) { return MatchHist{spos, nullptr, 0, N, n, (n + WAVE - 1) / WAVE}; }

// first sorted row of block blk's window: centred on the reference neighbour of the block's middle slot
// (robust against outliers at the ends), a multiple of 16 rows
__device__ __forceinline__ int window_origin(const int32_t* __restrict__ sp_ref_c, const int32_t* __restrict__ qo_c,
                                             int blk, int spb, int n, int m_pad, int WT) {      // n: the cloud's own slots
    if (m_pad <= WT || n <= 0) return 0;
    const int mid = min(blk * spb + spb / 2, n - 1);
    const int ctr = max(sp_ref_c[qo_c ? min(max(qo_c[mid], 0), n - 1) : mid], 0);
    return min(max(ctr - WT / 2, 0), m_pad - WT) & ~15;
}

// One block's share of one iteration (the body of accumulate_bwd_window_kernel, and of the tail launch that runs a cloud's remaining iterations):
// gs / gb = THIS cloud's cotangents of the normal equations, part_out = this block's row of the pose partial sums.
template <typename T, int MODE, int WT, bool overwrite>
__device__ __forceinline__ void window_body(const WeightParams& P, const T* __restrict__ src_s, const T* __restrict__ tgt_s, int c,
                                            const MatchHist spos, const int32_t* __restrict__ spos_ref,
                                            const int32_t* __restrict__ qorder,
                                            const T* __restrict__ pose, const T* __restrict__ w_s, const T* __restrict__ alive,
                                            const T* gs, const T* gb, int n, int m_pad, int spb, int bpc,
                                            T* __restrict__ gsrc_s, T* __restrict__ slab /* (N,bpc,WT,CV) */,
                                            T* __restrict__ gts_far /* (N,m_pad,CV) */,
                                            T* __restrict__ gw_s, T* part_out, const int32_t* __restrict__ src_rows, int cloud, int blk,
                                            int32_t* __restrict__ det_row = nullptr /* (N,n), with det_val (N,n,CV): deterministic target gradients (dicp_loop_buffers.bwd.det_far_row) */,
                                            T* __restrict__ det_val = nullptr) {
    // overwrite: first launch into uninitialised accumulators -- gsrc_s / gw_s / the slab windows are written, not added to
    constexpr int CV = (MODE == MODE_PT2PL) ? 6 : 3;
    __shared__ T red[(BLOCK / WAVE) * NBWD_PAD];
    constexpr int SPB = 4 * BLOCK;                          // window_slots() never exceeds this
    __shared__ T contrib[SPB * CV];                         // target-row contribution of each of the block's slots
    __shared__ int head[WT], next[SPB];                     // per window row: list of the slots that matched it (a slot whose match lies OUTSIDE the window keeps the row there)
    __shared__ int far_list[SPB], far_n;                    // ... and the slots with such matches
    const int tid = threadIdx.x;
    const int nc = rows_of(src_rows, cloud, n);             // ragged batches: slots past the cloud's own carry weight 0: no work, zero gradient
    const int s0 = blk * spb, s1 = min(nc, s0 + spb), s1_all = min(n, s0 + spb);
    const int32_t* __restrict__ qo_c = qorder ? qorder + (size_t)cloud * n : nullptr;  // slot -> query (spos is indexed by query)
    constexpr int U = 4;                                    // slots per thread, all in flight: spb <= U * BLOCK = SPB
    // the two dependent index chains (slot -> query -> match, and the same for the window origin) start first and
    // run under everything else the prologue loads
    bool on[U];
    int pos[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int s = s0 + u * BLOCK + tid;
        on[u] = s < s1;
        const int sq = on[u] ? s : s0;
        pos[u] = qo_c ? min(max(qo_c[sq], 0), n - 1) : sq;
    }
    const int lo = window_origin(spos_ref + (size_t)cloud * n, qo_c, blk, spb, nc, m_pad, WT);
    const int hi = min(lo + WT, m_pad);
#pragma unroll
    for (int u = 0; u < U; ++u) pos[u] = on[u] ? min(max(match_at(spos, cloud, pos[u]), 0), m_pad - 1) : 0;     // -1 (no neighbour: non-finite input) -> row 0
    if (slab)
        for (int k = tid; k < hi - lo; k += BLOCK) head[k] = -1;
    if (tid == 0) far_n = 0;
    T C[9], r[3], Gs[36], Gb[6];
    load_pose(pose, cloud, C, r);
#pragma unroll
    for (int k = 0; k < 36; ++k) Gs[k] = gs[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) Gb[k] = gb[k];
    const T live = alive ? alive[cloud] : T(1);
    T acc[NBWD];
#pragma unroll
    for (int k = 0; k < NBWD; ++k) acc[k] = T(0);
    __syncthreads();
    T* gfar = gts_far ? gts_far + (size_t)cloud * m_pad * CV : nullptr;
    {                                                       // every load is issued before its first dependent use
        const int base = s0;                                // (the block is latency-bound)
        T p[U][3], y[U][3], nrm[U][3], wv[U], g0[U][3], gwv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int s = base + u * BLOCK + tid;
            const size_t pt = (size_t)cloud * n + (on[u] ? s : s0);
            const T* spp = src_s + pt * 3;
            p[u][0] = spp[0]; p[u][1] = spp[1]; p[u][2] = spp[2];
            wv[u] = w_s ? w_s[pt] : T(1);
            const T* gsp = gsrc_s + pt * 3;
            g0[u][0] = g0[u][1] = g0[u][2] = gwv[u] = T(0);
            if (!overwrite) {
                g0[u][0] = gsp[0]; g0[u][1] = gsp[1]; g0[u][2] = gsp[2];
                if (gw_s) gwv[u] = gw_s[pt];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const T* yp = tgt_s + ((size_t)cloud * m_pad + pos[u]) * c;
            y[u][0] = yp[0]; y[u][1] = yp[1]; y[u][2] = yp[2];
            nrm[u][0] = nrm[u][1] = nrm[u][2] = T(0);
            if (MODE == MODE_PT2PL) { nrm[u][0] = yp[3]; nrm[u][1] = yp[4]; nrm[u][2] = yp[5]; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!on[u]) {
                if (base + u * BLOCK + tid < s1_all) {
                    const size_t pz = (size_t)cloud * n + base + u * BLOCK + tid;
                    if (overwrite) {                        // a pad slot of a ragged batch: its accumulators start at zero
                        gsrc_s[pz * 3] = gsrc_s[pz * 3 + 1] = gsrc_s[pz * 3 + 2] = T(0);
                        if (gw_s) gw_s[pz] = T(0);
                    }
                    if (det_row && slab) det_row[pz] = -1;
                }
                continue;
            }
            const size_t pt = (size_t)cloud * n + base + u * BLOCK + tid;
            T gp[3], gy[3], gn[3], gw0;
            point_backward<T, MODE>(P, C, r, p[u], y[u], nrm[u], wv[u] * live, Gs, Gb, gp, gy, gn, gw0, acc, acc + 9);
            T* gsp = gsrc_s + pt * 3;
            gsp[0] = g0[u][0] + gp[0]; gsp[1] = g0[u][1] + gp[1]; gsp[2] = g0[u][2] + gp[2];
            if (gw_s) gw_s[pt] = gwv[u] + gw0 * live;
            if (slab) {
                if (pos[u] >= lo && pos[u] < hi) {
                    const int sl = base - s0 + u * BLOCK + tid;     // < spb <= SPB
                    T* row = contrib + sl * CV;
                    row[0] = gy[0]; row[1] = gy[1]; row[2] = gy[2];
                    if (MODE == MODE_PT2PL) { row[3] = gn[0]; row[4] = gn[1]; row[5] = gn[2]; }
                    next[sl] = atomicExch(&head[pos[u] - lo], sl);
                    if (det_row) det_row[pt] = -1;
                } else if (det_row) {                       // deterministic: left by its slot for far_apply_kernel, which adds a cloud's in slot order
                    det_row[pt] = pos[u];
                    T* dv = det_val + pt * CV;
                    dv[0] = gy[0]; dv[1] = gy[1]; dv[2] = gy[2];
                    if (MODE == MODE_PT2PL) { dv[3] = gn[0]; dv[4] = gn[1]; dv[5] = gn[2]; }
                }
#ifdef DICP_FAR_DIRECT      // (A/B builds, scripts/build_variant.sh: round 5's form -- every lane adds its own row's floats)
                else {
                    T* row = gfar + (size_t)pos[u] * CV;
                    unsafeAtomicAdd(&row[0], gy[0]); unsafeAtomicAdd(&row[1], gy[1]); unsafeAtomicAdd(&row[2], gy[2]);
                    if (MODE == MODE_PT2PL) { unsafeAtomicAdd(&row[3], gn[0]); unsafeAtomicAdd(&row[4], gn[1]); unsafeAtomicAdd(&row[5], gn[2]); }
                }
#else
                else {
                    // outside the window: float atomics into gts_far -- issued below with a row's CV floats in CV consecutive lanes.  One lane adding its own
                    // row's CV floats makes every wave instruction touch 64 rows, 64 memory-side requests (MI355X_MICROARCH.md, global float atomics); on
                    // clouds that keep moving MOST contributions of the early iterations come this way (a launch took 0.38 ms there, more than the
                    // all-atomic kernel's 0.28: profiles/r06_backward_far_rows.txt)
                    const int sl = base - s0 + u * BLOCK + tid;
                    T* row = contrib + sl * CV;
                    row[0] = gy[0]; row[1] = gy[1]; row[2] = gy[2];
                    if (MODE == MODE_PT2PL) { row[3] = gn[0]; row[4] = gn[1]; row[5] = gn[2]; }
                    next[sl] = pos[u];
                    far_list[atomicAdd(&far_n, 1)] = sl;
                }
#endif
            }
        }
    }
    __syncthreads();
    if (slab && gfar) {     // the out-of-window contributions: element e of the flattened [far slot][column] list per lane
        const int ne = far_n * CV;
        for (int e = tid; e < ne; e += BLOCK) {
            const int f = e / CV, col = e - f * CV, sl = far_list[f];
            unsafeAtomicAdd(gfar + (size_t)next[sl] * CV + col, contrib[sl * CV + col]);
        }
    }
    if (slab) {     // one thread per window row; this block is the only writer of its slab rows
        T* out = slab + ((size_t)cloud * bpc + blk) * (WT * CV);
        for (int rr = tid; rr < hi - lo; rr += BLOCK) {
            int h = head[rr];
            if (h < 0 && !overwrite) continue;
            T sum[CV];
#pragma unroll
            for (int k = 0; k < CV; ++k) sum[k] = T(0);
            if (det_row) {
                // deterministic: the list is in the order the block's waves reached the row; its slots are summed in ASCENDING order instead (lists are one to
                // three slots long on clouds that converge: a selection pass per slot).  A LONG list -- many queries of one neighbourhood matched to one row: the
                // part of a cloud that has no counterpart in the target, all of it matched to the target's rim -- would cost its length squared: its slots are
                // handed to far_apply_kernel instead (left by slot like an out-of-window contribution), which adds them in slot order whatever their number.
                constexpr int DET_LIST_MAX = 24;
                int len = 0;
                for (int h2 = h; h2 >= 0 && len <= DET_LIST_MAX; ++len) h2 = next[h2];
                if (len > DET_LIST_MAX) {
                    for (int h2 = h, g2 = 0; h2 >= 0 && g2 < SPB; ++g2) {
                        const size_t pt = (size_t)cloud * n + s0 + h2;
                        det_row[pt] = lo + rr;
#pragma unroll
                        for (int k = 0; k < CV; ++k) det_val[pt * CV + k] = contrib[h2 * CV + k];
                        h2 = next[h2];
                    }
                } else {
                    int last = -1;
                    for (int guard = 0; guard < SPB; ++guard) {
                        int best = 0x7fffffff;
                        for (int h2 = h, g2 = 0; h2 >= 0 && g2 < SPB; ++g2) { if (h2 > last && h2 < best) best = h2; h2 = next[h2]; }
                        if (best == 0x7fffffff) break;
#pragma unroll
                        for (int k = 0; k < CV; ++k) sum[k] += contrib[best * CV + k];
                        last = best;
                    }
                }
            } else {
                for (int guard = 0; h >= 0 && guard < SPB; ++guard) {       // every slot is on at most one list
#pragma unroll
                    for (int k = 0; k < CV; ++k) sum[k] += contrib[h * CV + k];
                    h = next[h];
                }
            }
#pragma unroll
            for (int k = 0; k < CV; ++k) out[rr * CV + k] = overwrite ? sum[k] : out[rr * CV + k] + sum[k];
        }
    }
    block_reduce_store<T, NBWD, NBWD_PAD>(acc, part_out, red);
}

template <typename T, int MODE, int WT, bool overwrite>
__global__ __launch_bounds__(BLOCK) void accumulate_bwd_window_kernel(WeightParams P, const T* __restrict__ src_s, const T* __restrict__ tgt_s, int c,
                                                                      const MatchHist spos, const int32_t* __restrict__ spos_ref,
                                                                      const int32_t* __restrict__ qorder,
                                                                      const T* __restrict__ pose, const T* __restrict__ w_s, const T* __restrict__ alive,
                                                                      const T* __restrict__ gs, const T* __restrict__ gb,
                                                                      int N, int n, int m_pad, int spb, int bpc,
                                                                      T* __restrict__ gsrc_s, T* __restrict__ slab /* (N,bpc,WT,CV) */,
                                                                      T* __restrict__ gts_far /* (N,m_pad,CV) */,
                                                                      T* __restrict__ gw_s, T* __restrict__ bwd_partials, const int32_t* __restrict__ src_rows,
                                                                      const int32_t* __restrict__ skip /* optional (N), see accumulate_bwd_kernel */,
                                                                      int32_t* __restrict__ det_row, T* __restrict__ det_val) {
    int cloud, blk;
    if (!decode_block(bpc, N, cloud, blk)) return;
    T* part_out = bwd_partials + ((size_t)cloud * bpc + blk) * NBWD_PAD;
    if (!overwrite && skip && skip[cloud]) {                // (the first launch initialises the accumulators: it always runs)
        if (threadIdx.x < NBWD_PAD) part_out[threadIdx.x] = T(0);
        return;
    }
    window_body<T, MODE, WT, overwrite>(P, src_s, tgt_s, c, spos, spos_ref, qorder, pose, w_s, alive, gs + (size_t)cloud * 36, gb + (size_t)cloud * 6,
                                        n, m_pad, spb, bpc, gsrc_s, slab, gts_far, gw_s, part_out, src_rows, cloud, blk, det_row, det_val);
}

// Deterministic mode: the contributions of one iteration that were left by slot (det_row >= 0: the sorted target row, det_val: its CV values: matches outside their
// block's window, and the slots of window rows with long lists), into gts_far.  A block per (cloud, eighth of its target rows); the cloud's slots pass through LDS
// 256 at a time, in order, and every thread adds the entries of the rows it owns (row mod 256) as it meets them: two entries of one row are added by one thread,
// the lower slot first, whatever their number (a wave per cloud that walked the entries one by one took 16 ms per launch where half of the slots were such
// entries; this form 0.1 ms, profiles/r05_deterministic_cost.txt).
constexpr int FAR_RANGES = 8;
template <typename T, int CV>
__global__ __launch_bounds__(BLOCK) void far_apply_kernel(const int32_t* __restrict__ det_row, const T* __restrict__ det_val, T* __restrict__ gts_far, int n, int m_pad,
                                                          const int32_t* __restrict__ src_rows, const int32_t* __restrict__ skip) {
    __shared__ int rows_l[BLOCK];
    const int cloud = blockIdx.x / FAR_RANGES, rg = blockIdx.x % FAR_RANGES, tid = threadIdx.x;
    if (skip && skip[cloud]) return;                        // (the window launch did not run for this cloud: its records are an earlier iteration's)
    const int nc = rows_of(src_rows, cloud, n);
    const int per = ((m_pad + FAR_RANGES - 1) / FAR_RANGES + BLOCK - 1) / BLOCK * BLOCK;
    const int lo = rg * per, hi = min(lo + per, m_pad);
    T* gfar = gts_far + (size_t)cloud * m_pad * CV;
    for (int s0 = 0; s0 < nc; s0 += BLOCK) {                // (block-uniform trip count)
        const int s = s0 + tid;
        const int r = s < nc ? det_row[(size_t)cloud * n + s] : -1;
        const bool mine = r >= lo && r < hi;
        __syncthreads();                                    // (the previous chunk has been read by everybody)
        rows_l[tid] = mine ? r : -1;
        if (!__syncthreads_or(mine ? 1 : 0)) continue;
        for (int j0 = 0; j0 < BLOCK; j0 += 8) {             // (eight entries per round of LDS reads: one at a time the loop was a chain of LDS latencies)
            int rj[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) rj[u] = rows_l[j0 + u];          // (the same words for every thread: broadcasts)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (rj[u] >= 0 && ((rj[u] - lo) & (BLOCK - 1)) == tid) {
                    const T* v = det_val + ((size_t)cloud * n + s0 + j0 + u) * CV;
                    T* o = gfar + (size_t)rj[u] * CV;
#pragma unroll
                    for (int k = 0; k < CV; ++k) o[k] += v[k];
                }
            }
        }
    }
}

// gtgt[b][tperm[s]][col] += gts_far[b][s][col] + sum over the blocks whose window covers sorted row s of their
// slab rows: the once-per-call end of the windowed backward (also undoes the sorted target order).
constexpr int WR_U = 4;      // gradient elements per thread
template <typename T, int WT, int CV>
__global__ __launch_bounds__(BLOCK) void window_reduce_kernel(const T* __restrict__ slab, const int32_t* __restrict__ spos_ref,
                                                              const int32_t* __restrict__ qorder,
                                                              const int32_t* __restrict__ tperm, const T* __restrict__ gts_far,
                                                              int N, int n, int m, int m_pad, int cv, int spb, int bpc, int rpc,
                                                              T* __restrict__ gtgt, int c, int overwrite, const int32_t* __restrict__ src_rows) {
    constexpr int MAXB = 256;                               // window blocks per cloud handled per pass
    __shared__ int origin[MAXB];
    int cloud, rb;
    if (!decode_block(rpc, N, cloud, rb)) return;
    const int tid = threadIdx.x;
    const int e0 = rb * (BLOCK * WR_U);                        // this block's elements of the (m*cv) row-major gradient
    T acc[WR_U];
#pragma unroll
    for (int u = 0; u < WR_U; ++u) acc[u] = T(0);
    // the loads that do not wait for the window origins (two dependent index loads) go out first and run under them
    int dst[WR_U];
#pragma unroll
    for (int u = 0; u < WR_U; ++u) {
        const int e = min(e0 + u * BLOCK + tid, m * cv - 1);
        const int s = e / CV;
        dst[u] = tperm[(size_t)cloud * m_pad + s];
        if (gts_far) acc[u] = gts_far[((size_t)cloud * m_pad + s) * cv + (e - s * CV)];
    }
    static_assert(MAXB == BLOCK, "one window block per thread in the relevance pass below");
    __shared__ int rlist[MAXB], wcnt[BLOCK / WAVE];
    // the sorted rows this block's elements lie in
    const int r_lo = e0 / CV, r_hi = min(e0 + BLOCK * WR_U - 1, m * cv - 1) / CV;
    for (int b0 = 0; b0 < bpc; b0 += MAXB) {
        __syncthreads();
        const int nb = min(MAXB, bpc - b0);
        int org = 0;
        if (tid < nb)
            origin[tid] = org = window_origin(spos_ref + (size_t)cloud * n, qorder ? qorder + (size_t)cloud * n : nullptr, b0 + tid, spb, rows_of(src_rows, cloud, n), m_pad, WT);
        // Which windows cover a row is arithmetic on the origins; only those are loaded (two or three of the cloud's blocks, in ascending block order: the order of
        // the sums is what it was).  Round 5: the kernel used to issue a predicated load per (element, block) -- 64 load instructions per thread for ~9 real loads,
        // 117 us per call; it is bound by instructions, not bytes.  Round 6: the blocks whose window reaches this block's ~170 rows at all are listed first (one
        // thread per window block, ballots: the list stays in ascending block order) and every element is held against THAT list -- three blocks, not all of the
        // cloud's: at configs[3]'s size (64 window blocks per cloud) the launch took 866 us for 1.2 GB.
        const bool rel = tid < nb && org <= r_hi && org + WT > r_lo;
        const unsigned long long rmask = __ballot(rel);
        if ((tid & (WAVE - 1)) == 0) wcnt[tid >> 6] = __popcll(rmask);
        __syncthreads();
        int base = 0, nrel = 0;
#pragma unroll
        for (int w = 0; w < BLOCK / WAVE; ++w) { if (w < (tid >> 6)) base += wcnt[w]; nrel += wcnt[w]; }
        if (rel) rlist[base + __popcll(rmask & ((1ull << (tid & (WAVE - 1))) - 1ull))] = tid;
        __syncthreads();
        for (int bb = 0; bb < nrel; bb += 32) {
            unsigned cover[WR_U];
#pragma unroll
            for (int u = 0; u < WR_U; ++u) cover[u] = 0u;
            const int nbb = min(32, nrel - bb);
            for (int k = 0; k < nbb; ++k) {
                const int lo = origin[rlist[bb + k]];
#pragma unroll
                for (int u = 0; u < WR_U; ++u) {
                    const int sr = min(e0 + u * BLOCK + tid, m * cv - 1) / CV;
                    cover[u] |= (sr >= lo && sr < lo + WT) ? (1u << k) : 0u;
                }
            }
            while (__any((cover[0] | cover[1] | cover[2] | cover[3]) != 0u)) {       // (rounds of one load per element, all of a thread's in flight together)
                T v[WR_U];
#pragma unroll
                for (int u = 0; u < WR_U; ++u) {
                    v[u] = T(0);
                    if (cover[u]) {
                        const int k = __ffs((int)cover[u]) - 1;
                        cover[u] &= cover[u] - 1;
                        const int e = min(e0 + u * BLOCK + tid, m * cv - 1);
                        const int b = rlist[bb + k];
                        v[u] = slab[((size_t)cloud * bpc + b0 + b) * (WT * CV) + (size_t)(e - origin[b] * CV)];
                    }
                }
#pragma unroll
                for (int u = 0; u < WR_U; ++u) acc[u] += v[u];
            }
        }
    }
#pragma unroll
    for (int u = 0; u < WR_U; ++u) {
        const int e = e0 + u * BLOCK + tid;
        if (e >= m * cv) continue;
        const int s = e / CV, col = e - s * CV;
        const T v = acc[u];
        const int j = dst[u];
        if (j >= 0 && j < m) {
            T* o = gtgt + ((size_t)cloud * m + j) * c + col;
            *o = overwrite ? v : *o + v;                    // overwrite: every row of gtgt[:, :, :cv] is written exactly once
        }
    }
}

// out[b][perm[b][s]][0..cols) += in[b][s][0..cols) for s < cnt: undoes a sorted order.  perm must be injective per
// cloud (plain read-modify-write, no atomics).
template <typename T, int C>
__global__ __launch_bounds__(BLOCK) void permute_add_rows_kernel(const T* __restrict__ in, const int32_t* __restrict__ perm,
                                                                 int N, int cnt, int in_rows, int perm_rows, int c_in, int cols,
                                                                 T* __restrict__ out, int out_rows, int c_out, int bpc, int overwrite) {
    const unsigned total = (unsigned)cnt * (unsigned)cols;
    int b, blk;
    if (!decode_block(bpc, N, b, blk)) return;
    const unsigned e0 = (unsigned)blk * (BLOCK * ROWS_U) + threadIdx.x;
    {
        int j[ROWS_U], k[ROWS_U];
        T v[ROWS_U], o[ROWS_U];
        bool ok[ROWS_U];
#pragma unroll
        for (int u = 0; u < ROWS_U; ++u) {
            const unsigned e = min(e0 + u * BLOCK, total - 1);
            int s;
            split_cols<C>(e, cols, s, k[u]);
            j[u] = perm[(size_t)b * perm_rows + s];
            v[u] = in[((size_t)b * in_rows + s) * c_in + k[u]];
            ok[u] = e0 + u * BLOCK < total && j[u] >= 0 && j[u] < out_rows;
        }
#pragma unroll
        for (int u = 0; u < ROWS_U; ++u) o[u] = (ok[u] && !overwrite) ? out[((size_t)b * out_rows + j[u]) * c_out + k[u]] : T(0);
#pragma unroll
        for (int u = 0; u < ROWS_U; ++u)
            if (ok[u]) out[((size_t)b * out_rows + j[u]) * c_out + k[u]] = o[u] + v[u];
    }
}

// out[b][q] = the match of query q of cloud b at the iteration h stands for: a plain (N,n) array out of the history kept by reference (the window
// origins of the backward pass and dicp_window_reduce read the reference iteration's matches at a few places per block)
__global__ __launch_bounds__(BLOCK) void resolve_matches_kernel(const MatchHist h, int bpc, const int32_t* __restrict__ src_rows, int32_t* __restrict__ out) {
    int cloud, blk;
    if (!decode_block(bpc, h.N, cloud, blk)) return;
    const int q = blk * BLOCK + threadIdx.x;
    // (ragged batches: the rows past a cloud's own have no match, and the groups past them no word in `of`)
    if (q < h.n) out[(size_t)cloud * h.n + q] = q < rows_of(src_rows, cloud, h.n) ? match_at(h, cloud, q) : -1;
}

// keys[b][q][0] = the sorted position of query q's match as a float (a cloud's pad rows and queries without a match: the largest finite value, so that a
// stable sort leaves them last in index order): the keys dicp_match_order sorts.  Match positions are below 2^24: exact in float32.
template <typename T>
__global__ __launch_bounds__(BLOCK) void match_keys_kernel(const int32_t* __restrict__ spos_ref, const int32_t* __restrict__ src_rows, int N, int n, int bpc, T* __restrict__ keys3) {
    int cloud, blk;
    if (!decode_block(bpc, N, cloud, blk)) return;
    const int q = blk * BLOCK + threadIdx.x;
    if (q >= n) return;
    const int v = q < rows_of(src_rows, cloud, n) ? spos_ref[(size_t)cloud * n + q] : -1;
    T* o = keys3 + ((size_t)cloud * n + q) * 3;
    o[0] = v >= 0 ? (T)v : big_v<T>();
    o[1] = o[2] = T(0);
}

// ---------------------------------------------------------------------- step bwd
// Truncated reverse sweep.  Going backwards through the iterations, what iteration k adds to every gradient is LINEAR in the cotangent
// (G_A + G_A^T, g_b)_k of its normal equations, with coefficients (the per-point Jacobians, residuals, weights) of the same size in every
// iteration.  A Gauss-Newton step near its fixed point is a strong contraction -- the new pose hardly depends on the old one -- so the chain
// of pose cotangents shrinks by ~2e-4 per iteration (oracle, float64, random clouds and planar scenes: the gradient through the last
// 1 / 2 / 3 / 4 iterations only differs from the full one by 2e-4 / 4e-8 / 1e-11 / 2e-15 of its size, profiles/r03_cotangent_decay.txt):
// all but the last few iterations of a call add less than the rounding error of the sums they are added to.  step_bwd measures it on device,
// per cloud, in the data's own units (A = the iteration's normal matrix, sum u j_a^2 on its diagonal; s_a = sqrt(A_aa)):
//     m_k = max( max_ab |G_ab| s_a s_b , max_a |g_a| s_a )                    what iteration k itself adds
//     w_k = max_a |g_a| s_a  x  max_{k' < k, b} |delta_k',b| s_b              the most any EARLIER iteration could add: G_A = -(g delta^T + delta g^T)
//                                                                             multiplies the chain by that iteration's step, which is 1e6 times
//                                                                             larger at the start of a call than at its end, and the chain itself
//                                                                             cannot grow by more than O(1) per iteration (x16 allowed below)
// and ends the cloud's reverse sweep at iteration k -- this and every earlier iteration do no per-point work; of the pose cotangent only the
// part that does not go through the normal equations travels on (pose_pass_through) -- when  16 max(m_k, w_k) <= eps x (the largest m of
// the cloud's later iterations).  eps is a few units of the result
// type's roundoff (2^-22 for float32, 2^-40 for float64 from the host side): what is dropped is below the resolution of the sums it would be
// added to.  The sweep cannot be resumed after a skipped iteration (the partial sums a skipped iteration would have produced are what makes
// the chain shrink), hence "ends".  Iterations at which the cloud was already frozen (alive = 0) are skipped without ending anything: their
// weights are zero and every term of the adjoint is exactly zero.  A NaN measure never ends a sweep.  Hard Huber weights are excluded by the
// caller (their reference gradient is NaN at an exactly zero residual whatever the cotangent).
template <typename T> struct SkipArgs {
    int32_t* skip;           // (N) zero-initialised per backward pass: 0 = take part, 1 = frozen at this iteration, 2 = the cloud's sweep has ended (sticky).
                             // Written by step_bwd, read by the accumulate_bwd launch that follows (NULL: feature off)
    double* mref;            // (N) zero-initialised per backward pass: the largest m so far
    const T* alive_k;        // (N) or NULL
    int32_t* live_k;         // optional counter: clouds that take part in this iteration
    double eps;
    int k;                   // this iteration (delta_k - 6 j = the step of iteration k - j)
};
template <typename T>
__device__ __forceinline__ int skip_decision(const double* Gs, const double* Gb, const double* Areg, const double* dmax /* [6]: max |delta| of the earlier iterations */,
                                             int dim, int cloud, const SkipArgs<T>& sk, bool live /* alive_k != 0 */, double ref /* mref[cloud] */) {
    const int D = dim == 2 ? 3 : 6, OFF = dim == 2 ? 2 : 0;   // (Areg is compact, leading dimension 6; Gs / Gb / delta sit at their slots)
    if (!live) return 1;
    double sa[6], m = 0.0, gmax = 0.0, amp = 0.0;
    bool nan = false;
    for (int i = 0; i < D; ++i) sa[i] = sqrt(fabs(Areg[i * 6 + i]));
    for (int i = 0; i < D; ++i) {
        const double vb = fabs(Gb[i + OFF]) * sa[i], va = dmax[i + OFF] * sa[i];
        nan = nan || !(vb == vb) || !(va == va);
        gmax = vb > gmax ? vb : gmax;
        amp = va > amp ? va : amp;
        for (int j = 0; j < D; ++j) {
            const double v = fabs(Gs[(i + OFF) * 6 + (j + OFF)]) * sa[i] * sa[j];
            nan = nan || !(v == v);
            m = v > m ? v : m;
        }
    }
    m = gmax > m ? gmax : m;
    const double worst = gmax * amp > m ? gmax * amp : m;
    if (!nan && 16.0 * worst <= sk.eps * ref) return 2;
    if (!nan && m > ref) sk.mref[cloud] = m;
    if (sk.live_k) atomicAdd(sk.live_k, 1);
    return 0;
}

// What is left of step_backward for a cloud whose sweep has ended: the part of the pose cotangent that does not go through the normal
// equations, C_new = exp(delta^)^T C -> gC = R gCn, gr = grn.  It must go on: a loss may depend on the 3x3 block of T in directions that are
// no rotation at all (T.sum() does), and those pass through every iteration unchanged down to the gradient of T_init.
DICP_HD void pose_pass_through(const double* gCn, const double* grn, const double* delta6, double* gC, double* gr) {
    double R[9];
    so3_exp(delta6, R);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            gC[i * 3 + j] = R[i * 3 + 0] * gCn[0 * 3 + j] + R[i * 3 + 1] * gCn[1 * 3 + j] + R[i * 3 + 2] * gCn[2 * 3 + j];
    for (int i = 0; i < 3; ++i) gr[i] = grn[i];
}

template <typename T>
__global__ __launch_bounds__(WAVE) void step_bwd_kernel(const double* __restrict__ gpose_in, const T* __restrict__ bwd_partials,
                                                        int nblk, int dim, const T* __restrict__ pose_k,
                                                        const T* __restrict__ delta_k, long delta_stride,
                                                        const double* __restrict__ areg_k, T* __restrict__ gs,
                                                        T* __restrict__ gb, double* __restrict__ gpose_out, int N, SkipArgs<T> sk) {
    __shared__ double sg[NBWD_PAD], sC[9], sd[6], sAreg[36], sGs[36], sGb[6], sgo[12], sdmax[6], smref;
    __shared__ int salive, sended;
    const int cloud = blockIdx.x, tid = threadIdx.x;
    if (sk.skip) {                                          // the largest step of the EARLIER iterations, per component (lanes over iterations)
        double dm[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        for (int j = 1 + tid; j <= sk.k; j += WAVE) {
            const T* dp = delta_k + (size_t)cloud * delta_stride - (size_t)j * 6;
#pragma unroll
            for (int i = 0; i < 6; ++i) { const double v = fabs((double)dp[i]); dm[i] = v > dm[i] ? v : dm[i]; }
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
#pragma unroll
            for (int off = WAVE / 2; off > 0; off >>= 1) { const double o = __shfl_down(dm[i], off); dm[i] = o > dm[i] ? o : dm[i]; }
        }
        if (tid == 0) {
#pragma unroll
            for (int i = 0; i < 6; ++i) sdmax[i] = dm[i];
        }
    }
    {
        const int slot_i = tid & 15, part = tid >> 4;       // 4 partial sums per slot
        double s = 0.0;
        if (bwd_partials && slot_i < NBWD) {
            const T* pp = bwd_partials + (size_t)cloud * nblk * NBWD_PAD + slot_i;
            constexpr int UB = 4;                           // (all of a lane's loads in flight before the first add; same order of adds)
            for (int b0 = part; b0 < nblk; b0 += 4 * UB) {
                T v[UB];
#pragma unroll
                for (int u = 0; u < UB; ++u) v[u] = (b0 + 4 * u < nblk) ? pp[(size_t)(b0 + 4 * u) * NBWD_PAD] : T(0);
#pragma unroll
                for (int u = 0; u < UB; ++u) if (b0 + 4 * u < nblk) s += (double)v[u];
            }
        }
        s += __shfl_down(s, 32);
        s += __shfl_down(s, 16);
        if (tid < NBWD) sg[tid] = s + gpose_in[(size_t)cloud * 12 + tid];
        if (tid < 9) sC[tid] = (double)pose_k[(size_t)cloud * 12 + tid];
        if (tid < 6) sd[tid] = (double)delta_k[(size_t)cloud * delta_stride + tid];
        if (tid < 36) sAreg[tid] = areg_k[(size_t)cloud * 36 + tid];
        if (tid == 40 && sk.skip) smref = sk.mref[cloud];
        if (tid == 41 && sk.skip) salive = (!sk.alive_k || sk.alive_k[cloud] != T(0)) ? 1 : 0;
        if (tid == 42) sended = (sk.skip && sk.skip[cloud] == 2) ? 1 : 0;
    }
    __syncthreads();
    if (sended) {                                           // this cloud's reverse sweep has ended (skip_decision): only the pass-through part goes on
        if (tid == 0) {                                     // (the accumulate_bwd blocks of an ended cloud published zero sums: sg is the incoming cotangent)
            double g[12], d[6], go[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) g[k] = sg[k];
#pragma unroll
            for (int k = 0; k < 6; ++k) d[k] = sd[k];
            pose_pass_through(g, g + 9, d, go, go + 9);
#pragma unroll
            for (int k = 0; k < 12; ++k) gpose_out[(size_t)cloud * 12 + k] = go[k];
        }
        return;
    }
    if (tid == 0) {     // (operands in registers: the adjoint reads each of them many times, and an LDS read is ~64 cycles of a one-lane chain)
        double g[12], C[9], d[6], A[36], Gs[36], Gb[6], go[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) g[k] = sg[k];
#pragma unroll
        for (int k = 0; k < 9; ++k) C[k] = sC[k];
#pragma unroll
        for (int k = 0; k < 6; ++k) d[k] = sd[k];
#pragma unroll
        for (int k = 0; k < 36; ++k) A[k] = sAreg[k];
        step_backward(g, g + 9, dim, C, d, A, Gs, Gb, go, go + 9);
#pragma unroll
        for (int k = 0; k < 36; ++k) sGs[k] = Gs[k];
#pragma unroll
        for (int k = 0; k < 6; ++k) sGb[k] = Gb[k];
#pragma unroll
        for (int k = 0; k < 12; ++k) sgo[k] = go[k];
    }
    __syncthreads();
    if (sk.skip) {
        // the measures of skip_decision, by the lanes (one entry of G_A / g_b each) from the LDS copies: inside the one-lane section above
        // they cost it its registers (592 bytes of scratch in a serial chain: the kernel went from 8 to 30 us)
        const int D = dim == 2 ? 3 : 6, OFF = dim == 2 ? 2 : 0;
        double v = 0.0, vb = 0.0, va = 0.0;
        if (tid < 36) {
            const int i = tid / 6, j = tid - 6 * i;
            if (i < D && j < D) v = fabs(sGs[(i + OFF) * 6 + (j + OFF)]) * sqrt(fabs(sAreg[i * 6 + i])) * sqrt(fabs(sAreg[j * 6 + j]));
        } else if (tid < 42) {
            const int i = tid - 36;
            if (i < D) { const double sa = sqrt(fabs(sAreg[i * 6 + i])); vb = fabs(sGb[i + OFF]) * sa; va = sdmax[i + OFF] * sa; }
        }
        const bool nan = __any(!(v == v) || !(vb == vb) || !(va == va)) != 0;
        double m = v > vb ? v : vb, gmax = vb, amp = va;
#pragma unroll
        for (int off = WAVE / 2; off > 0; off >>= 1) {
            const double a = __shfl_xor(m, off), b = __shfl_xor(gmax, off), c = __shfl_xor(amp, off);
            m = a > m ? a : m; gmax = b > gmax ? b : gmax; amp = c > amp ? c : amp;
        }
        if (tid == 0) {
            int verdict = 0;
            if (!salive) verdict = 1;
            else {
                const double worst = gmax * amp > m ? gmax * amp : m;
                if (!nan && 16.0 * worst <= sk.eps * smref) verdict = 2;
                else {
                    if (!nan && m > smref) sk.mref[cloud] = m;
                    if (sk.live_k) atomicAdd(sk.live_k, 1);
                }
            }
            sk.skip[cloud] = verdict;       // (verdict 2: sgo is already what passes through; gs / gb are written but no block will read them)
        }
    }
    if (tid < 36) gs[(size_t)cloud * 36 + tid] = (T)sGs[tid];
    if (tid < 6) gb[(size_t)cloud * 6 + tid] = (T)sGb[tid];
    if (tid < 12) gpose_out[(size_t)cloud * 12 + tid] = sgo[tid];
}

// The reverse sweep of small clouds: what icp_small_forward_kernel is to the forward.  One block owns a cloud for a whole
// chunk of iterations, in reverse: step_bwd (cotangent of the pose -> cotangents of the normal equations, first thread),
// accumulate_bwd (per-point adjoint; source / weight gradients straight to memory, the block is their only writer; target
// gradients into an LDS copy of the cloud's rows, added to memory once at the end), block reduction of the pose
// cotangent sums, next iteration.  Two launches per iteration become one launch per chunk.
template <typename T, int MODE>
__global__ __launch_bounds__(BLOCK) void icp_small_backward_kernel(WeightParams P, dicp_loop_buffers B, int N, int n, int m, int dim,
                                                                   const double* __restrict__ gpose_in, double* __restrict__ gpose_out,
                                                                   int have_partials, T* __restrict__ gsrc, T* __restrict__ gtgt,
                                                                   T* __restrict__ gw, T* __restrict__ bwd_partials, int nblk, int k0, int k1) {
    constexpr int CV = (MODE == MODE_PT2PL) ? 6 : 3;
    extern __shared__ __align__(16) unsigned char small_bwd_lds[];
    T* gt = reinterpret_cast<T*>(small_bwd_lds);            // (m, CV) target-gradient rows of this cloud
    __shared__ double sg[NBWD_PAD], sC[9], sd[6], sAreg[36], sGs[36], sGb[6], sgo[12], spart[NBWD_PAD], sdmax[6], sR[WAVE * 9];
    constexpr int NT = BLOCK;
    __shared__ T red[(NT / WAVE) * NBWD_PAD];
    __shared__ T part[NBWD_PAD];
    __shared__ int s_skip;
    const int cloud = blockIdx.x, tid = threadIdx.x, c = B.c;
    const int nc = rows_of(B.src_rows, cloud, n);           // ragged batches: rows past the cloud's own carry no gradient
    bool ended = B.bwd.skip && B.bwd.skip[cloud] == 2;      // (this cloud's reverse sweep ended in an earlier chunk)
    if (gtgt)
        for (int e = tid; e < m * CV; e += NT) gt[e] = T(0);
    if (tid < 12) sgo[tid] = gpose_in[(size_t)cloud * 12 + tid];
    if (tid >= WAVE && tid < WAVE + NBWD_PAD) {             // (second wave: 16 lanes, one slot each, loads of all blocks in flight together when they are few)
        const int slot = tid - WAVE;
        double s = 0.0;
        if (have_partials && slot < NBWD)
            for (int b = 0; b < nblk; ++b) s += (double)bwd_partials[((size_t)cloud * nblk + b) * NBWD_PAD + slot];
        spart[slot] = s;
        part[slot] = T(0);
    }
    const T* __restrict__ src = (const T*)B.src + (size_t)cloud * n * 3;
    const T* __restrict__ tgt = (const T*)B.tgt + (size_t)cloud * m * c;
    const T* __restrict__ w_init = B.w_init ? (const T*)B.w_init + (size_t)cloud * n : nullptr;
    const T* __restrict__ dlt = (const T*)B.hist.deltas + (size_t)cloud * B.K * 6;
    __syncthreads();
    for (int k = k1 - 1; k >= k0; --k) {
        if (ended) {
            // The sweep has ended: only the pass-through part of the pose cotangent goes on (pose_pass_through), through ALL the
            // remaining iterations at once: gC <- R_k gC with R_k = exp(delta_k^), the rotations by the lanes, the chain by one.
            if (tid < 12) sgo[tid] += spart[tid];           // (the sums of the last launch / iteration before the end: zero for an ended cloud, added for form's sake)
            __syncthreads();
            for (int kb = k; kb >= k0; kb -= WAVE) {
                const int cnt = min(WAVE, kb - k0 + 1);
                if (tid < cnt) {
                    const T* dp = dlt + (size_t)(kb - tid) * 6;
                    const double d[6] = {(double)dp[0], (double)dp[1], (double)dp[2], (double)dp[3], (double)dp[4], (double)dp[5]};
                    double R[9];
                    so3_exp(d, R);
#pragma unroll
                    for (int e = 0; e < 9; ++e) sR[tid * 9 + e] = R[e];
                }
                __syncthreads();
                if (tid < 3) {                              // column tid of gC: the three columns are independent chains
                    double v0 = sgo[0 * 3 + tid], v1 = sgo[1 * 3 + tid], v2 = sgo[2 * 3 + tid];
                    for (int t = 0; t < cnt; ++t) {
                        const double* R = sR + t * 9;
                        const double a = R[0] * v0 + R[1] * v1 + R[2] * v2, b = R[3] * v0 + R[4] * v1 + R[5] * v2, cc = R[6] * v0 + R[7] * v1 + R[8] * v2;
                        v0 = a; v1 = b; v2 = cc;
                    }
                    sgo[0 * 3 + tid] = v0; sgo[1 * 3 + tid] = v1; sgo[2 * 3 + tid] = v2;
                }
                __syncthreads();
            }
            if (tid < NBWD_PAD) { spart[tid] = 0.0; part[tid] = T(0); }
            __syncthreads();
            break;
        }
        const T* pose_k = (const T*)B.hist.poses + (size_t)k * N * 12;
        if (tid < NBWD) sg[tid] = spart[tid] + sgo[tid];
        if (tid < 9) sC[tid] = (double)pose_k[(size_t)cloud * 12 + tid];
        if (tid < 6) sd[tid] = (double)dlt[(size_t)k * 6 + tid];
        if (tid < 36) sAreg[tid] = B.hist.areg[((size_t)k * N + cloud) * 36 + tid];
        if (B.bwd.skip && tid >= WAVE && tid < 2 * WAVE) {  // the largest step of the EARLIER iterations, per component (second wave: lanes over iterations)
            double dm[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            for (int j = tid - WAVE; j < k; j += WAVE) {
#pragma unroll
                for (int i = 0; i < 6; ++i) { const double v = fabs((double)dlt[(size_t)j * 6 + i]); dm[i] = v > dm[i] ? v : dm[i]; }
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) {
#pragma unroll
                for (int off = WAVE / 2; off > 0; off >>= 1) { const double o = __shfl_down(dm[i], off); dm[i] = o > dm[i] ? o : dm[i]; }
            }
            if (tid == WAVE) {
#pragma unroll
                for (int i = 0; i < 6; ++i) sdmax[i] = dm[i];
            }
        }
        __syncthreads();
        if (tid == 0) {
            step_backward(sg, sg + 9, dim, sC, sd, sAreg, sGs, sGb, sgo, sgo + 9);
            s_skip = 0;
            if (B.bwd.skip) {
                const SkipArgs<T> sk{B.bwd.skip, B.bwd.mref, (const T*)B.hist.alive + (size_t)k * N, B.bwd.live ? B.bwd.live + k : nullptr, B.bwd.skip_eps, k};
                s_skip = skip_decision(sGs, sGb, sAreg, sdmax, dim, cloud, sk, sk.alive_k[cloud] != T(0), B.bwd.mref[cloud]);
                B.bwd.skip[cloud] = s_skip;
            }
        }
        __syncthreads();
        if (s_skip) {       // 2: the cloud's reverse sweep ends here (see skip_decision) -- 1: frozen at this iteration, every term is exactly zero
            if (tid < NBWD_PAD) { spart[tid] = 0.0; part[tid] = T(0); }
            __syncthreads();
            if (s_skip == 2) { ended = true; }
            continue;
        }
        T C[9], r[3], Gs[36], Gb[6];
        load_pose(pose_k, cloud, C, r);
#pragma unroll
        for (int a = 0; a < 36; ++a) Gs[a] = (T)sGs[a];     // rounded to T like the gs / gb buffers of the multi-kernel path
#pragma unroll
        for (int a = 0; a < 6; ++a) Gb[a] = (T)sGb[a];
        const T live = ((const T*)B.hist.alive)[(size_t)k * N + cloud];
        const int32_t* __restrict__ idx_k = B.hist.idx + (size_t)k * N * n + (size_t)cloud * n;
        T acc[NBWD];
#pragma unroll
        for (int a = 0; a < NBWD; ++a) acc[a] = T(0);
        for (int i = tid; i < nc; i += NT) {
            const T p[3] = {src[i * 3], src[i * 3 + 1], src[i * 3 + 2]};
            const int j = min(max(idx_k[i], 0), m - 1);
            const T* yp = tgt + (size_t)j * c;
            const T y[3] = {yp[0], yp[1], yp[2]};
            T nrm[3] = {T(0), T(0), T(0)};
            if (MODE == MODE_PT2PL) { nrm[0] = yp[3]; nrm[1] = yp[4]; nrm[2] = yp[5]; }
            T gp[3], gy[3], gn[3], gw0;
            point_backward<T, MODE>(P, C, r, p, y, nrm, (w_init ? w_init[i] : T(1)) * live, Gs, Gb, gp, gy, gn, gw0, acc, acc + 9);
            T* gsp = gsrc + ((size_t)cloud * n + i) * 3;
            gsp[0] += gp[0]; gsp[1] += gp[1]; gsp[2] += gp[2];
            if (gw) gw[(size_t)cloud * n + i] += gw0 * live;
            if (gtgt) {
                T* row = gt + j * CV;
                atomicAdd(&row[0], gy[0]); atomicAdd(&row[1], gy[1]); atomicAdd(&row[2], gy[2]);
                if (MODE == MODE_PT2PL) { atomicAdd(&row[3], gn[0]); atomicAdd(&row[4], gn[1]); atomicAdd(&row[5], gn[2]); }
            }
        }
        block_reduce_store<T, NBWD, NBWD_PAD, NT>(acc, part, red);
        __syncthreads();
        if (tid < NBWD_PAD) spart[tid] = (double)part[tid];
        __syncthreads();
    }
    if (tid < 12) gpose_out[(size_t)cloud * 12 + tid] = sgo[tid];
    // the last accumulate_bwd's sums stay in bwd_partials (block 0 of nblk; the others are zero) for the caller / next chunk
    for (int e = tid; e < nblk * NBWD_PAD; e += NT)
        bwd_partials[(size_t)cloud * nblk * NBWD_PAD + e] = e < NBWD_PAD ? part[e] : T(0);
    if (gtgt)
        for (int e = tid; e < m * CV; e += NT) {
            const int j = e / CV, col = e - j * CV;
            gtgt[((size_t)cloud * m + j) * c + col] += gt[e];
        }
}

// The TAIL of the windowed reverse sweep of big clouds (dicp_loop_buffers.bwd.tail_from): the iterations k1-1 .. 0 in ONE launch.
// With the truncated sweep the iterations before the last few are, for almost every cloud, nothing but the pass-through of the pose
// cotangent -- yet a pair of dependent launches each (21 us of dispatch per iteration at the benchmark shape: a third of a K = 20
// backward).  Here, on accumulate_bwd_window's grid: block 0 of an ended cloud multiplies the cotangent through all its remaining
// iterations (the rotations exp(delta_k^) by the lanes, one product chain), its other blocks leave at once.  A cloud that is still at
// work (a straggler, or a cloud whose sweep ends in these iterations) is swept by ITS blocks together, iteration by iteration: every
// block runs the cloud's step_bwd itself -- same inputs, same instructions, same verdicts in all of them, so nothing has to be handed
// from one block to the others -- then its own share of accumulate_bwd_window (window_body), publishes its pose sums and waits until
// all of the cloud's blocks have published theirs (one counter per cloud; sums double-buffered by generation, so a block that is ahead
// never overwrites what a block behind still reads).  Blocks wait only for blocks of their own cloud, whose indices are all inside one
// group of 8 bpc consecutive blocks (decode_block): dispatch is in index order, so the lowest unfinished group is always resident as a
// whole and makes progress (dicp_bwd_tail_max_blocks refuses grids where it could not be) -- and every wait is bounded anyway: a block whose
// wait runs out raises the error words and folds NaN from there on, so the cloud's gradients come out NaN, never as plausible wrong numbers.
// On exit gpose_out holds the cotangent of pose_0 INCLUDING the last pose sums (dicp_pose_grad_out is then called without partials).
// The hand-off of the pose sums (round 5: the C++ memory model's own form): the publishing block's stores, a workgroup barrier, then ONE lane's
// agent-scope RELEASE fence + relaxed agent-scope add to the cloud's counter; the waiting lane polls the counter with relaxed agent-scope loads,
// then an agent-scope ACQUIRE fence, then the workgroup barrier behind which the block reads the sums.  (Rounds 3-4 used sc1 accesses counted
// behind s_waitcnt alone -- MI355X_MICROARCH's measured hand-off -- because the fences write back / invalidate the XCD's L2 under this kernel's
// gradient traffic; they are paid by the clouds that are still at work inside this launch only, which are few: profiles/r05_tail_handoff.txt.)
// The words themselves stay agent-scope atomic accesses (sc1: served at the memory side, never from a stale L1 line).
__device__ __forceinline__ void coherent_store(float* p, float v)   { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void coherent_store(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float  coherent_load(const float* p)  { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double coherent_load(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// (float32: compiled for two waves per SIMD -- left to itself the pt2pl form took 257 registers, one block per CU, and with it half the residency cap of dicp_bwd_tail_max_blocks)
template <typename T, int MODE, int WT>
__global__ __launch_bounds__(BLOCK, sizeof(T) == 4 ? 2 : 1) void bwd_tail_kernel(WeightParams P, dicp_loop_buffers B, int N, int n, int dim, int spb, int bpc,
                                                         const double* __restrict__ gpose_in, double* __restrict__ gpose_out, int have_partials,
                                                         T* __restrict__ gsrc_s, T* __restrict__ slab, T* __restrict__ gw_s,
                                                         T* part0 /* bwd_partials: the sums on entry, then the even generations */, T* part1 /* the odd generations */,
                                                         int32_t* arrive /* (N + 1) zeros: blocks that have published, per cloud; [N] = error word */, int k1) {
    __shared__ double sg[NBWD_PAD], sC[9], sd[6], sAreg[36], sGs[36], sGb[6], sgo[12], sdmax[6], sR[WAVE * 9], smref;
    __shared__ T sGsT[36], sGbT[6], spub[NBWD_PAD];
    __shared__ int s_verdict, s_alive, s_timeout;
    int cloud, blk;
    if (!decode_block(bpc, N, cloud, blk)) return;
    const int tid = threadIdx.x;
    bool ended = B.bwd.skip[cloud] == 2;                    // (decided by an earlier launch: the same for all of the cloud's blocks)
    if (ended && blk != 0) return;
    if (tid < 12) sgo[tid] = gpose_in[(size_t)cloud * 12 + tid];
    if (tid == 32) smref = B.bwd.mref[cloud];
    if (tid == 33) s_timeout = 0;
    const T* __restrict__ dlt = (const T*)B.hist.deltas + (size_t)cloud * B.K * 6;
    const T* cur = have_partials ? part0 : nullptr;         // the cloud's bpc rows of pose sums still to be added to the cotangent (NULL: zeros)
    int gen = 0;
    __syncthreads();
    // sg[0..12) = the cotangent + the sums of the last accumulate_bwd, in step_bwd_kernel's order (every thread calls; ends with a barrier)
    auto fold = [&](const T* rows) {
        if (tid < WAVE) {
            const int slot_i = tid & 15, part = tid >> 4;
            double s = 0.0;
            if (rows && slot_i < NBWD) {
                const T* pp = rows + (size_t)cloud * bpc * NBWD_PAD + slot_i;
                constexpr int UB = 4;
                for (int b0 = part; b0 < bpc; b0 += 4 * UB) {
                    T v[UB];
#pragma unroll
                    for (int u = 0; u < UB; ++u) v[u] = (b0 + 4 * u < bpc) ? coherent_load(pp + (size_t)(b0 + 4 * u) * NBWD_PAD) : T(0);
#pragma unroll
                    for (int u = 0; u < UB; ++u) if (b0 + 4 * u < bpc) s += (double)v[u];
                }
            }
            s += __shfl_down(s, 32);
            s += __shfl_down(s, 16);
            // a wait of this block ran out: the sums it would fold are not known to be complete.  Nothing plausible leaves this launch for the
            // cloud any more -- every sum this block folds from here on is NaN (and with it its share of the gradients, the sums it publishes to the
            // cloud's other blocks and the cloud's pose cotangent); the error words make the host raise (dicp_hip.h, bwd_tail_arrive)
            if (tid < NBWD) sg[tid] = s_timeout ? __builtin_nan("") : s + sgo[tid];
        }
        __syncthreads();
    };
    for (int k = k1 - 1; k >= 0; --k) {
        if (ended) {        // (block 0 only) the pass-through part of the pose cotangent through ALL the remaining iterations: gC <- exp(delta_k^) gC
            fold(cur);
            cur = nullptr;
            if (tid < 12) sgo[tid] = sg[tid];
            __syncthreads();
            for (int kb = k; kb >= 0; kb -= WAVE) {
                const int cnt = min(WAVE, kb + 1);
                if (tid < cnt) {
                    const T* dp = dlt + (size_t)(kb - tid) * 6;
                    const double d[6] = {(double)dp[0], (double)dp[1], (double)dp[2], (double)dp[3], (double)dp[4], (double)dp[5]};
                    double R[9];
                    so3_exp(d, R);
#pragma unroll
                    for (int e = 0; e < 9; ++e) sR[tid * 9 + e] = R[e];
                }
                __syncthreads();
                if (tid < 3) {                              // column tid of gC: three independent chains
                    double v0 = sgo[0 * 3 + tid], v1 = sgo[1 * 3 + tid], v2 = sgo[2 * 3 + tid];
                    for (int t = 0; t < cnt; ++t) {
                        const double* R = sR + t * 9;
                        const double a = R[0] * v0 + R[1] * v1 + R[2] * v2, b = R[3] * v0 + R[4] * v1 + R[5] * v2, cc = R[6] * v0 + R[7] * v1 + R[8] * v2;
                        v0 = a; v1 = b; v2 = cc;
                    }
                    sgo[0 * 3 + tid] = v0; sgo[1 * 3 + tid] = v1; sgo[2 * 3 + tid] = v2;
                }
                __syncthreads();
            }
            break;
        }
        // ---- step_bwd of iteration k: by every block of the cloud alike
        const T* pose_k = (const T*)B.hist.poses + (size_t)k * N * 12;
        const T* alive_k = (const T*)B.hist.alive + (size_t)k * N;
        if (tid >= WAVE && tid < 2 * WAVE) {                // (second wave, under the first one's loads) the largest step of the EARLIER iterations, per component
            double dm[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            for (int j = tid - WAVE; j < k; j += WAVE) {
#pragma unroll
                for (int i = 0; i < 6; ++i) { const double v = fabs((double)dlt[(size_t)j * 6 + i]); dm[i] = v > dm[i] ? v : dm[i]; }
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) {
#pragma unroll
                for (int off = WAVE / 2; off > 0; off >>= 1) { const double o = __shfl_down(dm[i], off); dm[i] = o > dm[i] ? o : dm[i]; }
            }
            if (tid == WAVE) {
#pragma unroll
                for (int i = 0; i < 6; ++i) sdmax[i] = dm[i];
            }
        }
        if (tid >= 2 * WAVE && tid < 2 * WAVE + 9) sC[tid - 2 * WAVE] = (double)pose_k[(size_t)cloud * 12 + (tid - 2 * WAVE)];
        if (tid >= 2 * WAVE + 16 && tid < 2 * WAVE + 22) sd[tid - 2 * WAVE - 16] = (double)dlt[(size_t)k * 6 + (tid - 2 * WAVE - 16)];
        if (tid >= 3 * WAVE && tid < 3 * WAVE + 36) sAreg[tid - 3 * WAVE] = B.hist.areg[((size_t)k * N + cloud) * 36 + (tid - 3 * WAVE)];
        if (tid == 3 * WAVE + 40) s_alive = alive_k[cloud] != T(0) ? 1 : 0;
        fold(cur);
        cur = nullptr;
        if (tid == 0) {     // (operands in registers, as in step_bwd_kernel)
            double g[12], C[9], d[6], A[36], Gs[36], Gb[6], go[12];
#pragma unroll
            for (int e = 0; e < 12; ++e) g[e] = sg[e];
#pragma unroll
            for (int e = 0; e < 9; ++e) C[e] = sC[e];
#pragma unroll
            for (int e = 0; e < 6; ++e) d[e] = sd[e];
#pragma unroll
            for (int e = 0; e < 36; ++e) A[e] = sAreg[e];
            step_backward(g, g + 9, dim, C, d, A, Gs, Gb, go, go + 9);
#pragma unroll
            for (int e = 0; e < 36; ++e) sGs[e] = Gs[e];
#pragma unroll
            for (int e = 0; e < 6; ++e) sGb[e] = Gb[e];
#pragma unroll
            for (int e = 0; e < 12; ++e) sgo[e] = go[e];
        }
        __syncthreads();
        if (tid < WAVE) {   // the measures of skip_decision by the lanes, as in step_bwd_kernel
            const int D = dim == 2 ? 3 : 6, OFF = dim == 2 ? 2 : 0;
            double v = 0.0, vb = 0.0, va = 0.0;
            if (tid < 36) {
                const int i = tid / 6, j = tid - 6 * i;
                if (i < D && j < D) v = fabs(sGs[(i + OFF) * 6 + (j + OFF)]) * sqrt(fabs(sAreg[i * 6 + i])) * sqrt(fabs(sAreg[j * 6 + j]));
            } else if (tid < 42) {
                const int i = tid - 36;
                if (i < D) { const double sa = sqrt(fabs(sAreg[i * 6 + i])); vb = fabs(sGb[i + OFF]) * sa; va = sdmax[i + OFF] * sa; }
            }
            const bool nan = __any(!(v == v) || !(vb == vb) || !(va == va)) != 0;
            double mm = v > vb ? v : vb, gmax = vb, amp = va;
#pragma unroll
            for (int off = WAVE / 2; off > 0; off >>= 1) {
                const double a = __shfl_xor(mm, off), b = __shfl_xor(gmax, off), cc = __shfl_xor(amp, off);
                mm = a > mm ? a : mm; gmax = b > gmax ? b : gmax; amp = cc > amp ? cc : amp;
            }
            if (tid == 0) {
                int verdict = 0;
                if (!s_alive) verdict = 1;
                else {
                    const double worst = gmax * amp > mm ? gmax * amp : mm;
                    if (!nan && 16.0 * worst <= B.bwd.skip_eps * smref) verdict = 2;
                    else {
                        // (kept in the block until the launch ends: a sibling block that STARTS after block 0 has been here -- dispatch is in index order, not
                        //  simultaneous -- would read the raised value where block 0 compared with the old one, could reach another verdict for this very
                        //  iteration and leave, and the cloud's other blocks would wait for it until their patience ran out: the TailTimeout seen once in a
                        //  thousand calls on planar scenes, rounds 5 and 6)
                        if (!nan && mm > smref) smref = mm;
                        if (blk == 0 && B.bwd.live) atomicAdd(B.bwd.live + k, 1);
                    }
                }
                if (blk == 0) B.bwd.skip[cloud] = verdict;  // (nobody reads it again in this launch: the cloud's blocks all hold the same verdict)
                s_verdict = verdict;
            }
            if (tid < 36) sGsT[tid] = (T)sGs[tid];          // rounded to T like the gs / gb buffers of the per-iteration launches
            if (tid < 6) sGbT[tid] = (T)sGb[tid];
        }
        __syncthreads();
        const int verdict = s_verdict;
        if (verdict == 2) {                                 // the cloud's sweep ends here: what is left is block 0's product chain
            if (blk != 0) return;
            ended = true;
            continue;
        }
        if (verdict == 1) continue;                         // frozen at this iteration: every term is exactly zero, and so are its sums
        // ---- this block's share of accumulate_bwd of iteration k
        ++gen;
        T* out = (gen & 1) ? part1 : part0;
        const MatchHist mh = (B.hist.spos_of && k >= B.hist.spos_of_from) ? MatchHist{B.hist.spos, B.hist.spos_of, k, N, n, (n + WAVE - 1) / WAVE} : plain_matches(B.hist.spos + (size_t)k * N * n, N, n);
        window_body<T, MODE, WT, false>(P, (const T*)B.src, (const T*)B.tgt, B.c, mh, B.bwd.spos_ref, B.search.qorder, pose_k, (const T*)B.w_init, alive_k,
                                        sGsT, sGbT, n, B.search.m_pad, spb, bpc, gsrc_s, slab, (T*)B.bwd.gts_far, gw_s, spub, B.src_rows, cloud, blk);
        // ---- publish the pose sums; wait until all of the cloud's blocks have published theirs.  The hand-off is a handful of words: they are
        // written and read as agent-scope atomics (coherent where they live; a release / acquire FENCE at agent scope writes back and
        // invalidates the whole L2 -- tens of microseconds under this kernel's gradient traffic), each store complete (the workgroup-scope
        // release: s_waitcnt) before the block is counted.
        __syncthreads();
        if (tid < NBWD_PAD) {
            coherent_store(out + ((size_t)cloud * bpc + blk) * NBWD_PAD + tid, spub[tid]);
            // every store of the hand-off has left this wave before the block is counted (written as asm: the compiler's own wait after a
            // fence can be dropped when it believes the wave's memory counter is already empty)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (tid == 0) {
            // release: everything this block stored before the barrier above is visible at agent scope before its arrival is.  (The wait is written as asm:
            // the compiler may drop its own behind the write-back when it believes the wave's memory counter empty -- MI355X_MICROARCH, compiler hazard.)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(arrive + cloud, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!(k == 0 && blk != 0) && !s_timeout) {      // (after the last iteration only block 0 still needs the sums; a block waits in vain at most once)
                const int want = gen * bpc;
                int spins = 0;
                while (__hip_atomic_load(arrive + cloud, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                    if (++spins > (1 << 20)) {              // ~0.5 s: the cloud's other blocks are not running (dicp_bwd_tail_max_blocks keeps that from happening)
                        // (nonzero = a wait ran out; the words say whose, for the report: cloud and iteration | arrivals seen, block, generation)
                        atomicExch(arrive + N, 0x40000000 | ((cloud & 0xffff) << 8) | (k & 0xff));
                        if (B.bwd.live) atomicExch(B.bwd.live + B.K, 0x40000000 | ((__hip_atomic_load(arrive + cloud, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xfff) << 16) | ((blk & 0xff) << 8) | (gen & 0xff));
                        s_timeout = 1;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(8);
                }
                // acquire: what the other blocks released before the arrivals just seen is visible to this block behind the barrier below
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        if (k == 0 && blk != 0) return;
        __syncthreads();
        cur = out;
    }
    if (blk == 0) {
        fold(cur);
        if (tid < 12) gpose_out[(size_t)cloud * 12 + tid] = sg[tid];
        if (tid == 0) B.bwd.mref[cloud] = smref;            // (the largest contribution measure so far, for a later chunk's launches)
    }
}
