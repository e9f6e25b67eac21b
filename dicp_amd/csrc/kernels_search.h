// libdicp_hip.so -- the searches: brute-force 1-NN (VALU), exact sorted sweep with match certificates, guard launch, single-query search.
// Part of the one translation unit dicp_kernels.hip (included inside its anonymous namespace, in this order: kernels_setup.h, kernels_search.h, kernels_setup_sort.h, kernels_rows.h, kernels_accumulate.h, kernels_backward.h, kernels_soft_svd.h, kernels_host.h).
// ------------------------------------------------------------------- kNN (VALU)
// Each lane owns Q queries and walks every target of its cloud; targets are staged once
// per block through LDS and read back as wave-wide broadcasts.  Per chunk of 8 targets the
// lane only tracks the running minimum VALUE (v_min3) and which chunk last improved it;
// the index inside that chunk is recovered once at the end (strict <, ascending order,
// so ties resolve to the lowest index exactly like torch.argmin).
template <typename T, int Q, int TILE, int CH, int MINW>
__global__ __launch_bounds__(BLOCK, MINW) void knn_valu_kernel(const T* __restrict__ src, const T* __restrict__ pose,
                                                         const typename V4<T>::type* __restrict__ tgt4,
                                                         int32_t* __restrict__ idx, int N, int n_full, int m_full, int m_pad_full, int bpc,
                                                         const int32_t* __restrict__ src_rows, const int32_t* __restrict__ tgt_rows) {
    using T4 = typename V4<T>::type;
    __shared__ T4 tile[TILE];
    int cloud, blk;
    if (!decode_block(bpc, N, cloud, blk)) return;
    const int tid = threadIdx.x;
    // ragged batches: this cloud's own lengths (the packed rows past m are pad rows: they are not even read)
    const int n = rows_of(src_rows, cloud, n_full), m = max(rows_of(tgt_rows, cloud, m_full), 1);
    const int m_pad = min((m + KNN_PAD - 1) / KNN_PAD * KNN_PAD, m_pad_full);
    if (blk * (BLOCK * Q) >= n) return;                     // (block-uniform)
    T C[9], r[3];
    load_pose(pose, cloud, C, r);

    T nx[Q][3], best[Q];
    int bchunk[Q];
#pragma unroll
    for (int qi = 0; qi < Q; ++qi) {
        const int i = blk * (BLOCK * Q) + qi * BLOCK + tid;
        T p[3] = {T(0), T(0), T(0)};
        if (i < n) {
            const T* sp = src + ((size_t)cloud * n_full + i) * 3;
            p[0] = sp[0]; p[1] = sp[1]; p[2] = sp[2];
        }
        query_point(C, r, p, nx[qi]);                       // ICP.py:137
        best[qi] = inf_v<T>();
        bchunk[qi] = 0;
    }

    const T4* __restrict__ tg = tgt4 + (size_t)cloud * m_pad_full;
    for (int base = 0; base < m_pad; base += TILE) {
        const int len = min(TILE, m_pad - base);            // multiple of 16
        for (int t = tid; t < len; t += BLOCK) tile[t] = tg[base + t];
        __syncthreads();
        for (int j0 = 0; j0 < len; j0 += CH) {
            T4 y[CH];
#pragma unroll
            for (int k = 0; k < CH; ++k) y[k] = tile[j0 + k];
#pragma unroll
            for (int qi = 0; qi < Q; ++qi) {
                T c = best[qi];
#pragma unroll
                for (int k = 0; k < CH; ++k) c = min_t(c, score<T, T4>(nx[qi], y[k]));
                bchunk[qi] = (c < best[qi]) ? base + j0 : bchunk[qi];
                best[qi] = c;                               // c = min(best, chunk): no select needed
            }
        }
        __syncthreads();
    }

#pragma unroll
    for (int qi = 0; qi < Q; ++qi) {
        const int i = blk * (BLOCK * Q) + qi * BLOCK + tid;
        if (i < n) {
            const T4* cp = tg + bchunk[qi];
            T bv = inf_v<T>();
            int bj = bchunk[qi];
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                const T s = score<T, T4>(nx[qi], cp[k]);
                if (s < bv) { bv = s; bj = bchunk[qi] + k; }
            }
            idx[(size_t)cloud * n_full + i] = min(bj, m - 1);
        }
    }
}

// ------------------------------------------------------------------ kNN (sweep)
// Exact 1-NN with slab pruning (same answer and tie rule as the brute-force kernels, far fewer pairs):
// the targets of a cloud are sorted by x ONCE per ICP call (they do not move); a wave owns 64*Q queries
// that are neighbours in x, starts at the target tile under them and sweeps tiles outwards, right and
// left alternately.  A side stops when its next tile starts further away in x alone than every query's
// current best distance: score(y) = 0.5|x-y|^2 - 0.5|x|^2 >= 0.5 (edge - x.x)^2 - 0.5|x|^2.
// The bound is applied with a safety margin far above the rounding error of a score, so a skipped
// target can never beat the kept minimum; exact score ties (duplicates) are detected and resolved to the
// lowest ORIGINAL index by a rare re-scan of the visited range.
template <typename T> struct SweepEps;
// How large the margin has to be (u = 2^-24; D = 0.5|x-y|^2, h = 0.5|x|^2; score() is three fmas on top of the stored 0.5|y|^2):
//   computed score of a target  >=  D(1 - 15u) - h(1 + 21u)        (3u on each of the four terms, 3u on the stored 0.5|y|^2,
//                                                                     |y| <= |x| + sqrt(2D), 4 sqrt(hD) <= 2h + 2D)
//   computed bound lb           <=  (0.5 dx^2 (1 + 3u) - h(1 - 3u))(1 + u),   D >= 0.5 dx^2 beyond the edge
// => a skipped target scores above `best` whenever  lb > best + 29u h + 13u |best|  = best + 1.8e-6 h + ...;  3e-6 keeps 1.7x of that.
template <> struct SweepEps<float>  { static constexpr float  v = 3e-6f; };
template <> struct SweepEps<double> { static constexpr double v = 1e-10; };

template <typename T, int NV, int PAD, int NT = BLOCK>
__device__ __forceinline__ void block_reduce_store(T* v, T* __restrict__ out, T* lds);

// Launch configuration of the tile sweep, measured at the benchmark shape (profiles/r01_sweep_configs_ab.txt): 2 queries per
// lane with 8-row chunks wins at every iteration once the per-chunk bookkeeping is three lane operations, and it wants
// registers rather than occupancy: 5 waves/SIMD (96 VGPRs, no scratch) beats 6 (80 VGPRs: the tie state spills).
constexpr int SWEEP_CFG_BIG = 2;            // (Q, CH) = (2, 8)
constexpr int SWEEP_MINW_Q2C8 = 5;

// Match certificates (temporal coherence, exact).  Between two ICP iterations near the pose a query moves by ~1e-7 m while the
// runner-up of its match is ~0.4 m further away: the argmin cannot have changed, and that can be PROVEN per query from what the
// search already knows.  A certifying search also tracks the second-smallest score it saw and stops a side only behind a wider
// margin; from   H1 = upper bound of the match's half squared distance (score + 0.5|x|^2 + E),   H2 = lower bound of every OTHER
// target's (the runner-up among the scored rows, minus E; half the squared x-distance to the first unscored row on either side),
// E the rounding bound of a score (the prune margin's),  d = sqrt(2H),  it derives   A = H2 - H1 - 3E   and   S = d1 + d2.
// After the query has moved by at most D, every other target is at least (d2 - D) away and the match at most (d1 + D): the match's
// computed score stays strictly the smallest -- ties and the lowest-index rule cannot come into play -- while   A - D S > 0.
// The step kernels keep, per cloud and iteration, M_k = a bound of how far any of its queries has moved since iteration 0
// (sum of |dC|_F max|p| + |dr|) and e_k = the rounding of a transformed point; a search at iteration k0 leaves per query the BUDGET
//     q = M_k0 + A / S - e_k0        (rounded down; -1: no certificate),
// and at iteration k the match is proven unchanged while   M_k + e_k < q   -- no record of when the query was last searched.
// The loop then runs, per iteration:  a guard launch (one wave per unit of the sweep: units with many spent budgets are searched
// again as units), and the forward accumulate, which checks each point's budget where it reads the point's match and searches the
// few spent ones on the spot (search_point).  Measured on the benchmark clouds: from the second certified iteration on, 0.18 % of
// the queries are searched again per iteration (near-ties inside the rounding bound, far from the cloud's centre).
template <typename T> struct SweepCert {
    T* q;                           // (N,n) budgets by SLOT (position in the query order of the certified iterations): only the searches read them
    T* qu;                          // (N,units): per unit of the sweep, the smallest budget if every query of the unit has one that stood when it was last
                                    // looked at, else 0 ("look at this unit": a query searched one by one, a candidate set to re-score) -- a filter, never a proof
    const T* dcum; int dstride;     // (N,dstride): (M_k, e_k) pairs per iteration
    int k;                          // this iteration
    int32_t* count;                 // (128) or NULL: [0,64) units searched again, [64,128) single queries, sharded by block
    void* set;                      // optional candidate sets (see search_point): (N,n) T set budgets by slot, then (N,n,4) int32 sorted positions
    int32_t* cloud;                 // (N,CERT_CLOUD) or NULL, per cloud: [0] units / [1] single queries searched again in this iteration; [2] the
                                    // state the step kernel keeps: 0 on, -1 on with one strike, k > 0 off for k more iterations (CERT_OFF_FOR_GOOD:
                                    // for the rest of the call), CERT_RECERTIFY: this iteration's guard searches every unit with certifying
                                    // sweeps; [3] its units (written by the searches: "a certified iteration ran"); [4] the last back-off length
    int32_t* cm;                    // (N,n) by slot: the query's current match (sorted position, -1 none), kept by the searches: a search that finds the
                                    // same match again changes nothing anywhere else
    int32_t* pend;                  // (N,n) zeros, by QUERY: a match that a guard launch CHANGED is left here (match + 2) for the accumulate of the same
                                    // iteration, which owns the history's slabs and the cached rows (accumulate_kernel) ...
    int32_t* gdirty;                // (N,nwr) by group of 64 queries: ... and is told so here (1: some pend of the group is set)
    int nwr;
    int32_t* slist; int32_t* scount;    // (N,n) / (N) zeros: per cloud, the slots that were given a candidate set (appended by the search that made it; entries whose set
                                    // no longer stands are skipped, the list is emptied when the whole cloud is searched again).  The guard launch re-scores the
                                    // standing sets 64 to a wave from this list: a unit whose only open queries have sets is not looked at for them
    // Plain searches of the loop, scoring form per cloud (dicp_loop_buffers.search.form): form_out[cloud] += the 64-row tiles this unit's slab had; a launch given
    // form_in (the previous plain search's tally) leaves a cloud alone unless its slabs were long (form_mine = 1: the matrix-core form) / short (0: this one)
    const int32_t* form_in; int32_t* form_out; int form_mine, form_default;     // (form_default: the form of a cloud without a tally -- 0 in form_in: no plain search before)
};
constexpr int FORM_TILES = 12;      // tiles per unit (in the PREVIOUS plain search) from which on a cloud's plain searches score on the matrix cores: a wave's fixed cost there
                                    // (prologue, margins, the exact refine of the winners' rows) is ~6 tiles' worth of VALU scoring (profiles/r06_f16_sweep_crossover.txt: 1.23x at
                                    // 10 tiles, level at 5.5, 0.85x at 4), and slabs halve from one early iteration to the next; round 4's form lost up to 16 tiles and the
                                    // threshold stood at 32 until round 6
__device__ __forceinline__ bool form_is_mine(const int32_t* __restrict__ form_in, int form_mine, int form_default, int cloud, int queries) {
    if (!form_in) return true;
    const int units = (queries + 2 * WAVE - 1) / (2 * WAVE), tally = form_in[cloud];     // (units of 128 queries: the matrix-core form's)
    const bool lng = tally > 0 ? tally > FORM_TILES * units : form_default != 0;
    return lng == (form_mine != 0);
}
constexpr int CERT_CANDS = 4;       // rows of a candidate set
template <typename T> __device__ __forceinline__ T* set_budgets(void* set) { return (T*)set; }
template <typename T> __device__ __forceinline__ int32_t* set_cands(void* set, int N, int n) { return (int32_t*)((char*)set + (size_t)N * n * sizeof(T)); }
constexpr int CERT_MARGIN = 6;      // prune margin of a certifying search, in units of the plain one: the slab ends where H > H1 + 6E, so an
                                    // unscored row alone still leaves A = 2E (the certificate needs H2 - H1 > 4E + D S); 8: the search 4 % slower,
                                    // 8 % fewer single searches in the iteration after it -- a wash (A/B on one box)
constexpr int CERT_SHARDS = 64;
constexpr int CERT_CLOUD = 8;       // ints per cloud of the per-cloud certificate state (SweepCert::cloud)
constexpr int CERT_OFF_FOR_GOOD = 1 << 20, CERT_RECERTIFY = -2;
constexpr int CERT_SLOT_MAX = 16;   // a unit with more spent budgets than this is searched again as a unit (guard launch), the others' queries one by one

template <typename T>
__device__ __forceinline__ T cert_budget(T A, T S, T H1, T hx, const T* __restrict__ dk /* (M_k, e_k) */) {
    if (!(A > T(0))) return T(-1);
    // how far the query may move: A / S, and never further than a fifth of max(d1, |x|) -- the last term of A covers the rounding of the
    // scores AFTER the move only while 0.5 |x|^2 and the match's half squared distance have not grown past 1.8x, which this cap guarantees
    // (15 (t + .2)^2 + 21 (1.2)^2 <= 1.8 (15 t^2 + 21) and 15 (1.2)^2 + 21 (t + .2)^2 <= 1.8 (15 + 21 t^2) for every t in [0, 1])
    const T cap = T(0.2) * max_t(m_sqrt(T(2) * H1), m_sqrt(T(2) * hx));
    const T slack = (S > T(0) && A < inf_v<T>()) ? min_t(A / S, cap) : cap;
    return (slack + dk[0]) * (T(1) - T(8) * CertUlp<T>::v) - dk[1] * (T(1) + T(8) * CertUlp<T>::v);
}
// The budget of a query from what its search knows: bv = the match's score, s2 = the smallest score of any other SCORED row (inf: none),
// h_unscored = a lower bound of the half squared distance of every row that was not scored (inf: all were), hx = 0.5 |x|^2.
// Rounding model (u = unit roundoff; three fmas on a stored 0.5|y|^2, |y| <= |x| + d):  |score - (D - h)| <= 15u D + 21u h;  with the
// rounding of hx and of the sum,  |(score + hx) - D| <= 16u D + 24u h.  Taken x1.5 for H1 (above D1) and H2 (below D2); the two scores
// compared AFTER the move err by 2 (15u D' + 21u h') (1 + 16u) with D', h' <= 1.8x (the cap in cert_budget): 54u D1 + 76u h, taken x1.2.
template <typename T>
__device__ __forceinline__ T cert_from_scores(T bv, T s2, T hx, T h_unscored, const T* __restrict__ dk, T& H1_out) {
    const T u = T(0.5) * CertUlp<T>::v;
    const T H1r = max_t(bv + hx, T(0));
    const T H1 = H1r + (T(24) * u * H1r + T(36) * u * hx);
    H1_out = H1;
    T H2 = inf_v<T>();
    if (s2 < inf_v<T>()) { const T H2r = s2 + hx; H2 = H2r - (T(24) * u * m_abs(H2r) + T(36) * u * hx); }
    H2 = min_t(H2, h_unscored);
    T A = inf_v<T>(), S = T(0);                                  // (no other target at all: only the cap limits the budget)
    if (H2 < inf_v<T>()) { A = (H2 - H1) - (T(65) * u * H1 + T(91) * u * hx); S = m_sqrt(T(2) * H1) + m_sqrt(T(2) * max_t(H2, T(0))); }
    return cert_budget(A, S, H1, hx, dk);
}
// "no certificate, searched at iteration k": never above cert_spent(), and told apart from a budget that was spent before this iteration
template <typename T> __device__ __forceinline__ T cert_mark(int k) { return T(-(k + 2)); }
// what a budget is compared with at iteration k:  budget > cert_spent(...)  <=>  the match stands
template <typename T>
__device__ __forceinline__ T cert_spent(const T* __restrict__ dk) { return (dk[0] + dk[1]) * (T(1) + T(8) * CertUlp<T>::v); }

// wave-wide minimum (all lanes get it)
template <typename T> __device__ __forceinline__ T wave_min(T v) {
#pragma unroll
    for (int o = WAVE / 2; o > 0; o >>= 1) v = min_t(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
    for (int o = WAVE / 2; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
    return v;
}

// each wave keeps the last NT tiles it scored in a ring: near the pose that is the whole visited range, and the
// epilogue then re-scores the winning chunk out of LDS instead of gathering its rows (Q x CH 16-byte gathers per lane)
template <typename T> struct SweepRing { static constexpr int NT = sizeof(T) == 4 ? 6 : 3; };

// The search of ONE unit (64*Q consecutive slots of a cloud's query order) by one wave; `ring`: the wave's NT tiles of LDS.
// STATE: the search runs inside the certified loop and keeps its state (budgets or marks, the searches' copy of the matches, pending matches); the plain
// kernel outside it is compiled without (its registers are the headline's: at 96 it spilled 40 bytes with that code merely present)
template <typename T, int Q, int CH, bool CERT, bool STATE = true>
__device__ __forceinline__ void sweep_unit(const T* __restrict__ src, const T* __restrict__ pose,
                                           const typename V4<T>::type* __restrict__ tgs4,
                                           const int32_t* __restrict__ tperm, const int32_t* __restrict__ qorder,
                                           const int32_t* __restrict__ bucket, const T* __restrict__ brange, int nbkt,
                                           int32_t* __restrict__ idx, int32_t* __restrict__ spos,
                                           unsigned long long* __restrict__ pairs,
                                           int n_full, int m_full, int m_pad,
                                           const int32_t* __restrict__ src_rows, const int32_t* __restrict__ tgt_rows, const SweepCert<T>& ct,
                                           const int cloud, const int unit, typename V4<T>::type* __restrict__ ring, const bool to_pend = false) {
    using T4 = typename V4<T>::type;
    constexpr int NT = SweepRing<T>::NT;
    const int lane = threadIdx.x & (WAVE - 1);
    // ragged batches: this cloud's own lengths.  Its queries are the first n slots of qorder, its targets the first m sorted rows
    const int n = rows_of(src_rows, cloud, n_full), m = max(rows_of(tgt_rows, cloud, m_full), 1);
    const bool idle_wave = unit * (WAVE * Q) >= n;
    if (idle_wave) return;                                  // whole wave idle (no block-level sync anywhere below)
    T C[9], r[3];
    load_pose(pose, cloud, C, r);

    T nx[Q][3], xq[Q], hx[Q], best[Q];
    T sec[Q];                         // CERT: second-smallest chunk minimum seen
    int qi[Q], mi[Q], c1[Q], c2[Q];   // c1: chunk that set the minimum; c2: a second chunk with an EQUAL minimum; mi: the match
    T tb[Q], ob[Q];                   // tie records carry the minimum they were made at and count only if it is still the
                                      // final one (nothing to reset when the minimum moves): tb for c2; ob: three or more
                                      // chunks tied, resolved by re-scanning the visited range
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int pos = unit * (WAVE * Q) + q * WAVE + lane;
        qi[q] = -1; mi[q] = 0;
        T p[3] = {T(0), T(0), T(0)};
        if (pos < n) {
            qi[q] = qorder ? qorder[(size_t)cloud * n_full + pos] : pos;
            const T* sp = src + ((size_t)cloud * n_full + qi[q]) * 3;
            p[0] = sp[0]; p[1] = sp[1]; p[2] = sp[2];
        }
        query_point(C, r, p, nx[q]);
        const T v[3] = {-nx[q][0], -nx[q][1], -nx[q][2]};
        xq[q] = v[0];
        hx[q] = T(0.5) * (v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        best[q] = inf_v<T>();
        sec[q] = inf_v<T>();
        c1[q] = 0; c2[q] = -1;
        tb[q] = ob[q] = -inf_v<T>();
    }
    // idle slots of a partial last wave take a real query's values (their own first one, else lane 0's:
    // lane 0 of a live wave always holds a real query) so that they never hold the sweep open
    {
        const T b0 = __shfl(nx[0][0], 0), b1 = __shfl(nx[0][1], 0), b2 = __shfl(nx[0][2], 0), bx = __shfl(xq[0], 0), bh = __shfl(hx[0], 0);
        if (qi[0] < 0) { nx[0][0] = b0; nx[0][1] = b1; nx[0][2] = b2; xq[0] = bx; hx[0] = bh; }
#pragma unroll
        for (int q = 1; q < Q; ++q)
            if (qi[q] < 0) { nx[q][0] = nx[0][0]; nx[q][1] = nx[0][1]; nx[q][2] = nx[0][2]; xq[q] = xq[0]; hx[q] = hx[0]; }
    }

    const T4* __restrict__ tg = tgs4 + (size_t)cloud * m_pad;
    const int ntiles = min((m + WAVE - 1) / WAVE, m_pad / WAVE);      // (the sorted rows past m are pad rows)
    // start under the wave's middle query: coarse bucket table of lower_bound positions (built once per call)
    const T xc = __shfl(xq[Q / 2], WAVE / 2);
    const T xlo = brange[(size_t)cloud * 2], inv = brange[(size_t)cloud * 2 + 1];
    T fb = (xc - xlo) * inv;
    fb = fb < T(0) ? T(0) : (fb > T(nbkt) ? T(nbkt) : fb);
    int start = bucket[(size_t)cloud * (nbkt + 1) + (int)fb];
    {   // an uneven cloud can put thousands of targets into one equal-width table bucket: finish the lower bound there
        // (wave-uniform; on even clouds a bucket is a fraction of a tile and this costs nothing)
        int hi = bucket[(size_t)cloud * (nbkt + 1) + min((int)fb + 1, nbkt)];
        while (hi - start > WAVE) {
            const int mid = (start + hi) >> 1;
            if (tg[mid].x < xc) start = mid + 1; else hi = mid;
        }
    }
    int tR = min(max(start / WAVE, 0), ntiles - 1), tL = tR - 1;
    if (idle_wave) { tR = ntiles; tL = -1; }
    int visR = tR, visL = tR;                               // tiles [visL, visR) have been scored
    const int t0 = tR;                                      // tile t sits in ring slot (t - t0) mod NT
    int sR = 0, sL = NT - 1;
    T edgeR = -inf_v<T>(), edgeL = inf_v<T>();
    // both directions keep their next tile in flight while the current one is being scored
    T4 preR = tg[(size_t)tR * WAVE + lane];
    T4 preL = tg[(size_t)max(tL, 0) * WAVE + lane];

    auto process = [&](const T4& mine, int t, int slot) {
        T4* tile = ring + slot * WAVE;
        tile[lane] = mine;
        __builtin_amdgcn_wave_barrier();
#pragma unroll 1
        for (int j0 = 0; j0 < WAVE; j0 += CH) {
            T4 y[CH];
#pragma unroll
            for (int k = 0; k < CH; ++k) y[k] = tile[j0 + k];
            const int chunk = t * WAVE + j0;
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                T cm = score<T, T4>(nx[q], y[0]);
#pragma unroll
                for (int k = 1; k < CH; ++k) cm = min_t(cm, score<T, T4>(nx[q], y[k]));
                // common path: compare, select the chunk, min -- three lane operations.  With 64 lanes a chunk lowers
                // SOMEBODY's minimum most of the time near the pose, so a wave-uniform "anything changed?" branch around
                // a longer update was taken almost always; only exact ties (duplicated targets) are rare, and they alone
                // sit behind the wave-uniform branch.  The prune threshold is derived from best where it is used.
                if (__builtin_expect(__any(cm == best[q]) != 0, 0)) {
                    asm volatile("" ::: "memory");          // keep this a real (wave-uniform) branch, not predicated code
                    if (cm == best[q] && cm < inf_v<T>()) {
                        if (c2[q] >= 0 && tb[q] == best[q]) ob[q] = best[q];
                        else { c2[q] = chunk; tb[q] = best[q]; }
                    }
                }
                if (CERT) sec[q] = min_t(sec[q], max_t(best[q], cm));      // (two smallest of the chunk minima so far)
                const bool lt = cm < best[q];
                c1[q] = lt ? chunk : c1[q];
                best[q] = lt ? cm : best[q];
            }
        }
        __builtin_amdgcn_wave_barrier();
    };
    auto prunable = [&](T edge, bool right) {
        bool ok = true;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const T dx = right ? edge - xq[q] : xq[q] - edge;
            const T lb = T(0.5) * dx * dx - hx[q];
            const T thr = best[q] + (CERT ? T(CERT_MARGIN) : T(1)) * SweepEps<T>::v * (T(1) + m_abs(best[q]) + hx[q]);      // best + margin
            ok = ok && (dx > T(0)) && (lb > thr);
        }
        return __all(ok) != 0;
    };

    bool cutR = false, cutL = false;                        // a side ended by the bound (unscored rows remain beyond its edge), not by the array
    while (tR < ntiles || tL >= 0) {
        if (tR < ntiles) {
            if (prunable(edgeR, true)) { tR = ntiles; cutR = true; }
            else {
                const T4 cur = preR;
                if (tR + 1 < ntiles) preR = tg[(size_t)(tR + 1) * WAVE + lane];
                process(cur, tR, sR);
                sR = sR + 1 == NT ? 0 : sR + 1;
                edgeR = __shfl(cur.x, WAVE - 1);
                visR = ++tR;
            }
        }
        if (tL >= 0) {
            if (prunable(edgeL, false)) { tL = -1; cutL = true; }
            else {
                const T4 cur = preL;
                if (tL >= 1) preL = tg[(size_t)(tL - 1) * WAVE + lane];
                process(cur, tL, sL);
                sL = sL == 0 ? NT - 1 : sL - 1;
                edgeL = __shfl(cur.x, 0);
                visL = tL--;
            }
        }
    }

    const int32_t* __restrict__ pm = tperm + (size_t)cloud * m_pad;
    T qmin = inf_v<T>();                                    // CERT: smallest budget this wave wrote,
    int nunc = 0;                                           // ... and how many of this lane's queries got none
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        if (qi[q] < 0) continue;
        T bv = inf_v<T>(), rv = inf_v<T>();                  // rv: smallest score among the re-scored rows other than the winner (CERT)
        int bo = 0x7fffffff, bs = 0;
        auto consider = [&](int j, const T4& row) {          // lowest ORIGINAL index among equal scores; the permutation
            const T sc = score<T, T4>(nx[q], row);          // is only read for the winner and on (rare) exact ties
            if (CERT) rv = min_t(rv, max_t(bv, sc));
            if (sc < bv) { bv = sc; bs = j; bo = -1; }
            else if (sc == bv && sc < inf_v<T>()) {
                if (bo < 0) bo = pm[bs];
                const int o = pm[j];
                if (o < bo) { bo = o; bs = j; }
            }
        };
        auto consider_chunk = [&](int c) {
            const int t = c >> 6;
            // still in the ring: visited, and neither t + NT nor t - NT was scored (either would have taken its slot)
            if (t >= visL && t < visR && t + NT >= visR && t - NT < visL) {
                const T4* rp = ring + ((unsigned)(t - t0 + NT * (1 << 24)) % NT) * WAVE + (c & (WAVE - 1));
#pragma unroll
                for (int k = 0; k < CH; ++k) consider(c + k, rp[k]);
            } else {
#pragma unroll
                for (int k = 0; k < CH; ++k) consider(c + k, tg[c + k]);
            }
        };
        if (ob[q] != best[q]) {
            consider_chunk(c1[q]);
            if (c2[q] >= 0 && tb[q] == best[q]) consider_chunk(c2[q]);
        } else {
            // >= 3 chunks share the minimum (duplicated targets): rare, re-scan what this wave visited
            for (int j = visL * WAVE; j < visR * WAVE; ++j) consider(j, tg[j]);
        }
        if (bo < 0) bo = idx ? pm[bs] : (bv < inf_v<T>() ? 0 : 0x7fffffff);   // (0x7fffffff: nothing finite was seen; without idx the
                                                                                // original index is only looked up on exact ties)
        mi[q] = (bo == 0x7fffffff) ? 0 : min(max(bo, 0), m - 1);
        if (idx) idx[(size_t)cloud * n_full + qi[q]] = mi[q];
        // sorted position of the winner (indexed like idx, by the query): what the windowed backward consumes.  Inside a certified iteration
        // (to_pend) a match that CHANGED is left for the accumulate that follows, which owns the history's slabs and the cached rows
        const int val = (bo == 0x7fffffff || bo >= m) ? -1 : bs;
        const size_t at = (size_t)cloud * n_full + unit * (WAVE * Q) + q * WAVE + lane;       // the query's slot: the certificates' own arrays go by it
        if (STATE && to_pend) {
            if (ct.cm[at] != val) {
                ct.pend[(size_t)cloud * n_full + qi[q]] = val + 2;
                ct.gdirty[(size_t)cloud * ct.nwr + (qi[q] >> 6)] = 1;
            }
        } else if (spos) spos[(size_t)cloud * n_full + qi[q]] = val;
        if (STATE && ct.cm) ct.cm[at] = val;
        if (CERT) {
            T bq = T(-1);
            if (bo != 0x7fffffff && ob[q] != best[q]) {     // (three or more tied chunks: no certificate)
                const T eps = SweepEps<T>::v;
                const T s2 = min_t(sec[q], rv);             // runner-up among the scored rows: other chunks, and the winner's own
                T hu = inf_v<T>();                          // rows beyond a side that the bound ended: at least 0.5 dx^2 away
                if (cutR) { const T dx = edgeR - xq[q]; hu = min_t(hu, dx > T(0) ? T(0.5) * dx * dx * (T(1) - T(8) * eps) : T(0)); }
                if (cutL) { const T dx = xq[q] - edgeL; hu = min_t(hu, dx > T(0) ? T(0.5) * dx * dx * (T(1) - T(8) * eps) : T(0)); }
                T H1c;
                bq = cert_from_scores(bv, s2, hx[q], hu, ct.dcum + (size_t)cloud * ct.dstride + 2 * ct.k, H1c);
            }
            // no certificate: -(k + 2) says "searched at iteration k" -- spent for every later iteration
            ct.q[at] = bq > T(0) ? bq : cert_mark<T>(ct.k);
            if (ct.set) set_budgets<T>(ct.set)[at] = T(-1);      // (a new search: whatever candidate set the query had is void)
            if (bq > T(0)) qmin = min_t(qmin, bq); else ++nunc;
        } else if (STATE && ct.q) {
            ct.q[at] = cert_mark<T>(ct.k);                      // plain search of a unit inside a certified loop
            if (ct.set) set_budgets<T>(ct.set)[at] = T(-1);
        }
    }
    if (CERT) {
        // the unit's filter value: its smallest budget -- or 0 ("look at me every iteration") when a query has no certificate at all: the
        // guard launch of the next iteration then searches it on its own (and tries a candidate set for it)
        qmin = wave_min(qmin);
        int tot = 0;
#pragma unroll
        for (int q = 0; q < Q; ++q) tot += __popcll(__ballot(nunc > q));
        if (lane == 0) ct.qu[(size_t)cloud * ((n_full + WAVE * Q - 1) / (WAVE * Q)) + unit] = tot > 0 ? T(0) : qmin;
        // per cloud, for the step kernel's "are certificates worth it here?": queries that got no certificate will be searched one by one in
        // every later iteration (near-ties inside the rounding bound of a score: dense surfaces far from the centre, duplicated targets)
        if (ct.cloud && lane == 0) {
            if (tot && !ct.set) atomicAdd(ct.cloud + (size_t)cloud * CERT_CLOUD + 1, tot);      // (with candidate sets they are searched ONCE more, and counted then)
            if (tot && ct.set) atomicAdd(ct.cloud + (size_t)cloud * CERT_CLOUD + 6, tot);       // ... but a cloud where MOST queries came back without one is not worth the sets
            if (unit == 0) { ct.cloud[(size_t)cloud * CERT_CLOUD + 3] = (n_full + WAVE * Q - 1) / (WAVE * Q); ct.cloud[(size_t)cloud * CERT_CLOUD + 5] = ct.set ? 1 : 0; }
        }
    }
    // sharded: one counter serialises ~12 ns per add, which at 65k waves would outlast the kernel itself
    if (pairs && lane == 0 && !idle_wave)
        atomicAdd(pairs + (blockIdx.x & (DICP_PAIR_SHARDS - 1)), (unsigned long long)(visR - visL) * WAVE * WAVE * Q);
    if (ct.form_out && lane == 0 && (Q > 1 || !(unit & 1))) atomicAdd(ct.form_out + cloud, visR - visL);      // (per 128 queries: the one-query-per-lane forms tally every other unit)
}

#define DICP_SWEEP_PARAMS const T* __restrict__ src, const T* __restrict__ pose, const typename V4<T>::type* __restrict__ tgs4, \
        const int32_t* __restrict__ tperm, const int32_t* __restrict__ qorder, const int32_t* __restrict__ bucket, const T* __restrict__ brange, int nbkt, \
        int32_t* __restrict__ idx, int32_t* __restrict__ spos, unsigned long long* __restrict__ pairs, \
        int N, int n_full, int m_full, int m_pad, int bpc, const int32_t* __restrict__ src_rows, const int32_t* __restrict__ tgt_rows, SweepCert<T> ct
#define DICP_SWEEP_MINW ((Q == 2 && CH == 8 && sizeof(T) == 4) ? SWEEP_MINW_Q2C8 : 1)

// Every unit of every cloud: block (cloud, blk) of the XCD-aware grid, one unit per wave.
// The certifying search and the guard launch carry more state than the plain search (the runner-up, the budgets; both forms of the unit search, the
// single-query search): at the plain kernel's 5 waves per SIMD they kept 64 / 132 bytes per lane in scratch memory; at 4 they keep none
// (profiles/r05_kernel_resources.txt).
#ifndef DICP_CERT_MINW
#define DICP_CERT_MINW 4
#endif
template <typename T, int Q, int CH, bool CERT>
__global__ __launch_bounds__(BLOCK, (CERT && DICP_SWEEP_MINW > DICP_CERT_MINW) ? DICP_CERT_MINW : DICP_SWEEP_MINW) void knn_sweep_kernel(DICP_SWEEP_PARAMS) {
    __shared__ typename V4<T>::type tiles[BLOCK / WAVE][SweepRing<T>::NT * WAVE];
    int cloud, blk;
    if (!decode_block(bpc, N, cloud, blk)) return;
    if (!CERT && !form_is_mine(ct.form_in, ct.form_mine, ct.form_default, cloud, rows_of(src_rows, cloud, n_full))) return;      // (this cloud's slabs were long: the matrix-core launch has it)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));     // (wave-uniform, and told so: the unit's number then lives in a scalar register)
    sweep_unit<T, Q, CH, CERT, CERT>(src, pose, tgs4, tperm, qorder, bucket, brange, nbkt, idx, spos, pairs, n_full, m_full, m_pad, src_rows, tgt_rows, ct,
                                     cloud, blk * (BLOCK / WAVE) + wave, tiles[wave]);
}

// What the search of one query needs besides the query (guard launches of the certified loop).
template <typename T> struct SearchCtx {
    const typename V4<T>::type* tgs4; const int32_t* tperm; const int32_t* bucket; const T* brange; int nbkt;
    const int32_t* tgt_rows; int m_full, m_pad;
    SweepCert<T> ct;
};

// What the accumulate of a certified iteration needs (accumulate_kernel CERT: the row cache and the match history kept by reference).
template <typename T> struct AccCert {
    int32_t* spos;                                  // (N,n) this iteration's slab of the match history (or the one reused buffer)
    // The match history is kept BY REFERENCE: of[(k N + cloud) nwr + g] = the iteration whose slab holds the matches of queries [64 g, 64 g + 64) at
    // iteration k.  A certified iteration that changes no match of such a group writes nothing but that word; one that does copies the group into its own
    // slab first.  hist = the virtual base of the slabs of this history chunk (iteration s at hist + s N n), hist_prev that of the chunk before it
    // (references never reach further back: the first iteration of a chunk copies every group), k_floor the chunk's first iteration.
    const int32_t* hist; const int32_t* hist_prev; int32_t* of; int k_floor, N, nwr, k;
    T* nbr;                                         // (N,n,NB) row cache: the matched target row of every query (NB = 6 pt2pl / 3 pt2pt elements), rewritten with the match
    int32_t* gdirty;                                // (N,nwr): 1 = the guard launch of this iteration changed a match of the group (SweepCert::pend)
    int32_t* pend;                                  // (N,n): those matches (match + 2; 0: none)
    int32_t* cloud;                                 // optional (N,CERT_CLOUD): the per-cloud switch's state in [2] (SweepCert::cloud); [3] / [5] are written here ("a certified
                                                    // iteration ran": the cloud's units, whether candidate sets are kept) -- the guard launch has no block per cloud
    int fresh;                                      // the search of this iteration has just written EVERY match into `spos` (the certifying search): everything is gathered and cached
    int units, sets;
    int32_t* scount;                                // (N) lengths of the candidate-set lists (SweepCert::scount): emptied behind a certifying search of everything
};

// The search of ONE query by one wave (all lanes carry the same arguments): the query's previous match, scored under the current
// pose, bounds the best score from above, and with it the slab of sorted rows that can hold the new match -- the same bound the
// sweep prunes with, so a row outside the slab can never beat the kept minimum.  The lanes score the slab's rows 64 at a time with
// the score() every search form uses; equal scores resolve to the lowest ORIGINAL index: index for index the match of a full
// search.  Returns the match's sorted position (-1: none) and leaves the query's new budget in `budget`.
template <typename T>
__device__ __forceinline__ int search_point(const SearchCtx<T>& ps, const int cloud, const T* nx, const int prev, T& budget, unsigned long long& rows_scored,
                                            T& set_budget, int* cset /* [CERT_CANDS], wave-uniform */) {
    using T4 = typename V4<T>::type;
    const int lane = threadIdx.x & (WAVE - 1);
    const int m = max(rows_of(ps.tgt_rows, cloud, ps.m_full), 1);
    const T xq = -nx[0];
    const T hx = T(0.5) * (nx[0] * nx[0] + nx[1] * nx[1] + nx[2] * nx[2]);
    const T4* __restrict__ tg = ps.tgs4 + (size_t)cloud * ps.m_pad;
    const int32_t* __restrict__ pm = ps.tperm + (size_t)cloud * ps.m_pad;
    const T eps = SweepEps<T>::v;

    // the slab: rows whose x alone does not put them beyond the previous match's score (+ the certifying margin)
    const T ub = (prev >= 0 && prev < m) ? score<T, T4>(nx, tg[prev]) : inf_v<T>();
    int r0 = 0, r1 = m;
    T h_edge = inf_v<T>();                                      // lower bound of the half squared distance of every row outside the slab
    const T inv = ps.brange[(size_t)cloud * 2 + 1];
    if (ub < inf_v<T>() && inv > T(0)) {
        const T thr = ub + T(CERT_MARGIN) * eps * (T(1) + m_abs(ub) + hx);
        const T R = m_sqrt(max_t(T(2) * (thr + hx), T(0))) * (T(1) + T(4) * eps);      // 0.5 dx^2 - hx > thr  for every |dx| > R
        const T xlo = ps.brange[(size_t)cloud * 2];
        const int32_t* __restrict__ bk = ps.bucket + (size_t)cloud * (ps.nbkt + 1);
        // bucket b of the table starts at the lower bound of xlo + b / inv (sweep_buckets_kernel); the index of a value and the
        // table's edges are rounded differently by far less than one bucket, so one bucket more on either side is a superset
        T fa = (xq - R - xlo) * inv - T(1), fb = (xq + R - xlo) * inv + T(2);
        fa = fa < T(0) ? T(0) : (fa > T(ps.nbkt) ? T(ps.nbkt) : fa);
        r0 = min(bk[(int)fa], m);
        r1 = fb >= T(ps.nbkt) ? m : min(max(bk[(int)fb], r0), m);
        // an equal-width table bucket can hold thousands of rows (an uneven cloud; a ragged batch's one far pad row stretches the table's
        // span a thousandfold, so every real row sits in a bucket or two): finish both bounds by bisection on the sorted keys, as
        // sweep_unit does for its start.  Rows left of xq - R and right of xq + R cannot beat the kept minimum (thr), so any r0 at or
        // below the first row with x >= xq - R and any r1 at or above the first row with x > xq + R keep the search exact.
        // (wave-uniform; on even clouds the slab is already a few tiles and the loops do not run)
        if (r1 - r0 > 4 * WAVE) {
            const T xa = xq - R, xb = xq + R;
            int lo = r0, hi = r1;
            while (hi - lo > WAVE) { const int mid = (lo + hi) >> 1; if (tg[mid].x < xa) lo = mid + 1; else hi = mid; }
            r0 = max(lo - 1, r0);                               // (one row of slack: the keys were rounded when they were packed)
            lo = r0; hi = r1;
            while (hi - lo > WAVE) { const int mid = (lo + hi) >> 1; if (tg[mid].x <= xb) lo = mid + 1; else hi = mid; }
            r1 = min(hi + 1, r1);
        }
        if (r0 > 0 || r1 < m) h_edge = (thr + hx) * (T(1) - T(8) * eps);
    }

    T b1 = inf_v<T>(), b2 = inf_v<T>(), b3 = inf_v<T>();       // this lane's three smallest scores (the third: a bound only)
    int j1 = -1, j2 = -1;
    constexpr int U = 4;                                        // rows in flight per lane: a wide slab is a few round trips, not one per 64 rows
    for (int j = r0 + lane; j < r1; j += U * WAVE) {
        T4 y[U];
#pragma unroll
        for (int u = 0; u < U; ++u) y[u] = tg[min(j + u * WAVE, r1 - 1)];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int jj = j + u * WAVE;
            const T sc = jj < r1 ? score<T, T4>(nx, y[u]) : inf_v<T>();
            if (sc < b1) { b3 = b2; b2 = b1; j2 = j1; b1 = sc; j1 = jj; }
            else if (sc == b1 && sc < inf_v<T>()) { b3 = b2; b2 = b1; if (pm[jj] < pm[j1]) { j2 = j1; j1 = jj; } else j2 = jj; }
            else if (sc < b2) { b3 = b2; b2 = sc; j2 = jj; }
            else if (sc < b3) b3 = sc;
        }
    }
    const T bv = wave_min(b1);
    int bs = -1;
    budget = T(-1);
    if (bv < inf_v<T>()) {                                      // (wave-uniform)
        const bool cand = b1 == bv;
        const unsigned long long cm = __ballot(cand);
        int win;
        if (__popcll(cm) == 1) win = __ffsll((long long)cm) - 1;
        else {                                                  // equal scores in several lanes: lowest original index
            const int o = cand ? pm[j1] : 0x7fffffff;
            const int omin = wave_min(o);
            win = __ffsll((long long)__ballot(cand && o == omin)) - 1;
        }
        bs = __shfl(j1, win);
        const T s2 = wave_min(lane == win ? b2 : b1);
        T H1c;
        budget = cert_from_scores(bv, s2, hx, h_edge, ps.ct.dcum + (size_t)cloud * ps.ct.dstride + 2 * ps.ct.k, H1c);
        // No certificate for the match alone (a runner-up inside the rounding allowance of the scores: dense surfaces, duplicated targets):
        // a certificate for a SET.  The CERT_CANDS smallest scores' rows are kept; s_rest bounds every other row from below (what the lanes
        // have left of their three smallest, and the slab's edge).  While the query has moved by less than the budget that (match, s_rest)
        // give -- the same inequality as above with the runner-up replaced by the best row OUTSIDE the set -- the old match still scores
        // strictly below every outside row, so the new match is the best of the set under the same score() and tie rule: CERT_CANDS rows
        // to re-score per iteration instead of a search.
        set_budget = T(-1);
        if (ps.ct.set && !(budget > T(0))) {
            cset[0] = bs;
            T r1 = b1, r2 = b2;
            int i1 = j1, i2 = j2;
            if (lane == win) { r1 = b2; i1 = j2; r2 = inf_v<T>(); }      // (the winner's own entry is used up)
#pragma unroll
            for (int c = 1; c < CERT_CANDS; ++c) {
                const T mn = wave_min(r1);
                cset[c] = -1;
                if (mn < inf_v<T>()) {                              // (wave-uniform)
                    const int L0 = __ffsll((long long)__ballot(r1 == mn)) - 1;
                    cset[c] = __shfl(i1, L0);
                    if (lane == L0) { r1 = r2; i1 = i2; r2 = inf_v<T>(); }
                }
            }
            const T s_rest = wave_min(min_t(r1, b3));
            T H1s;
            set_budget = cert_from_scores(bv, s_rest, hx, h_edge, ps.ct.dcum + (size_t)cloud * ps.ct.dstride + 2 * ps.ct.k, H1s);
        }
    } else set_budget = T(-1);
    rows_scored += (unsigned long long)(r1 - r0);             // (the caller counts the searches and adds everything to the statistics ONCE, at its end:
                                                                //  a wave's loads return behind its earlier atomics, and a cloud's word is one address)
    return bs;          // (sorted slots [0,m) hold the cloud's own rows: a row found is a real one)
}

// Guard of a certified iteration: one wave per unit of the sweep, as in knn_sweep_kernel.  It owns the certificates: a unit whose filter value
// stands (every query has a budget, and the smallest of them is not spent) leaves after one load.  Of the others, a unit with more than
// CERT_SLOT_MAX spent budgets is searched again as a unit (cheaper per query than one by one, and what keeps a batch that suddenly moves far
// from falling back on single searches).  The rest is dealt with here, query by query (until round 5 inside the accumulate that follows, whose
// registers -- and with them the occupancy of every launch that had nothing to search -- that code decided): a standing candidate set is
// re-scored by the query's own lane (four gathered rows), a query without budget or set is searched by the whole wave (search_point).  Whatever
// match CHANGES is left in pend for the accumulate, its group of 64 queries marked in gdirty; the certificates' own state (budgets, sets,
// current matches) goes by slot, so a unit's share of it is one coalesced piece.
template <typename T, int Q, int CH>
__device__ __forceinline__ void guard_unit(DICP_SWEEP_PARAMS, const int cloud, const int unit, typename V4<T>::type* __restrict__ ring) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int units = (n_full + WAVE * Q - 1) / (WAVE * Q);
    const int n = rows_of(src_rows, cloud, n_full);
    if (unit * (WAVE * Q) >= n) return;                        // (a unit past the cloud's own rows)
    const T* dk = ct.dcum + (size_t)cloud * ct.dstride + 2 * ct.k;
    const T spent = cert_spent(dk);
    const T step = ct.k > 0 ? dk[0] - dk[-2] : inf_v<T>();     // how far the cloud's queries can have moved in the last step
    T* qu = ct.qu + (size_t)cloud * units + unit;
    const T v = *qu;
    bool plain = false, research = true;
    const int cstate = ct.cloud ? ct.cloud[(size_t)cloud * CERT_CLOUD + 2] : 0;
    // what the unit's queries need: 0 nothing (the budget stands) / 1 the candidate set re-scored / 2 a search of their own
    int need[Q], cmv[Q], qiv[Q];
    T bud[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) { need[q] = 0; bud[q] = inf_v<T>(); cmv[q] = -1; qiv[q] = 0; }
    T* __restrict__ qs = ct.set ? set_budgets<T>(ct.set) : nullptr;
    int32_t* __restrict__ cands = ct.set ? set_cands<T>(ct.set, N, n_full) : nullptr;
    if (cstate > 0) plain = true;                               // this cloud's certificates are off (step kernel): every unit, plainly
    else if (cstate == CERT_RECERTIFY) plain = false;           // ... and this is the iteration that tries them again: every unit, certifying
    else if (v < T(0)) plain = step > -v;                       // plain mode (below): certify again once the steps are at most -v
    else {
        if (v > spent) return;
        int bad = 0, live = 0;
        // every piece of the unit's state goes by slot: all of it at once, whatever the budgets turn out to say (a unit that is looked at mostly has
        // a candidate set to re-score: budget -> set budget -> candidates -> rows was a chain of dependent loads, the launch's whole time near the pose)
        T bq[Q], sq[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int pos = unit * (WAVE * Q) + q * WAVE + lane;
            const size_t at = (size_t)cloud * n_full + min(pos, n - 1);
            bq[q] = ct.q[at];
            sq[q] = qs ? qs[at] : T(-2);
            cmv[q] = ct.cm[at];
            qiv[q] = qorder ? qorder[at] : min(pos, n - 1);
        }
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int pos = unit * (WAVE * Q) + q * WAVE + lane;
            if (pos < n) {
                const T b = bq[q];
                bud[q] = b;
                ++live;
                if (!(b > spent)) {                             // no certificate of its own: a candidate set that still stands is as good (re-scored below)
                    // A budget that the poses' motion has spent counts against the unit, as ever.  A query that never had a certificate of its own
                    // (b < 0: a mark) may have a candidate set: one that stands is as good as a budget; none tried yet (-1): its search below
                    // will try; "no set either" (-2) or a spent set count against the unit.
                    const T sb = (qs && b < T(0)) ? sq[q] : T(-2);
                    if (sb > spent) bud[q] = sb;                // (a standing set: re-scored from the cloud's list, guard_sets; as good as a budget here)
                    else { need[q] = 2; if (sb != T(-1)) ++bad; }
                }
            }
        }
        int nbad = 0, nlive = 0;
#pragma unroll
        for (int q = 0; q < Q; ++q) { nbad += __popcll(__ballot(bad > q)); nlive += __popcll(__ballot(live > q)); }
        research = nbad > CERT_SLOT_MAX;
        if (research) {
            // three quarters of the last search's budgets did not survive one step, and the steps are not shrinking fast (less than halved
            // since the one before): certifying this unit is wasted work while the cloud moves like this.  It is searched plainly (cheaper,
            // no budgets) until the steps have halved.
            const T step_before = ct.k > 1 ? dk[-2] - dk[-4] : inf_v<T>();
            plain = 4 * nbad >= 3 * nlive && step > T(0) && step < inf_v<T>() && T(2) * step > step_before;
            if (plain && lane == 0) *qu = -T(0.5) * step;
        }
    }
    if (research) {
        // a unit searched again leaves the matches it CHANGED for the accumulate (pend) -- unless the whole cloud is searched (certificates off, or
        // tried again): that accumulate then reads the iteration's own slab for the cloud, as after the certifying search
        const bool to_pend = ct.pend && !(cstate > 0 || cstate == CERT_RECERTIFY);
        if (plain) sweep_unit<T, Q, CH, false>(src, pose, tgs4, tperm, qorder, bucket, brange, nbkt, idx, spos, pairs, n_full, m_full, m_pad, src_rows, tgt_rows, ct,
                                               cloud, unit, ring, to_pend);
        else       sweep_unit<T, Q, CH, true>(src, pose, tgs4, tperm, qorder, bucket, brange, nbkt, idx, spos, pairs, n_full, m_full, m_pad, src_rows, tgt_rows, ct,
                                              cloud, unit, ring, to_pend);
        // the counters LAST: vector memory operations return in order, so a unit that counted itself first waited for its add -- one of up to
        // 128 to the same word when a whole cloud is searched again -- before its first load came back (a cloud with its certificates off:
        // 53 us per launch instead of the plain kernel's 31)
        if (lane == 0 && ct.count) atomicAdd(ct.count + (blockIdx.x & (CERT_SHARDS - 1)), 1);
        if (lane == 0 && ct.cloud) atomicAdd(ct.cloud + (size_t)cloud * CERT_CLOUD, 1);
        return;
    }
    // ---- the unit's open queries, one by one
    const SearchCtx<T> sc{tgs4, tperm, bucket, brange, nbkt, tgt_rows, m_full, m_pad, ct};
    T C[9], r[3];
    load_pose(pose, cloud, C, r);
    unsigned long long rows_scored = 0;
    int singles = 0;
    T qmin = inf_v<T>();
    bool open = false;                                          // this lane keeps a query that must be searched again in the next iteration
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int pos = unit * (WAVE * Q) + q * WAVE + lane;
        const size_t at = (size_t)cloud * n_full + min(pos, n - 1);
        const int qi = qiv[q], cur = cmv[q];
        T nx[3] = {T(0), T(0), T(0)};
        if (need[q]) {
            const T* sp = src + ((size_t)cloud * n_full + qi) * 3;
            const T p[3] = {sp[0], sp[1], sp[2]};
            query_point(C, r, p, nx);
        }
        int found = cur;
        // spent, never certifiable, NaN: searched by the whole wave, one query at a time
        unsigned long long todo = __ballot(need[q] == 2);
        T nb = T(-1), ns = T(-2);
        int nc[CERT_CANDS];
#pragma unroll
        for (int c = 0; c < CERT_CANDS; ++c) nc[c] = -1;
        while (todo) {                                          // (wave-uniform)
            const int L = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const T nq[3] = {__shfl(nx[0], L), __shfl(nx[1], L), __shfl(nx[2], L)};
            T got, gs;
            int gc[CERT_CANDS];
            const int fnd = search_point<T>(sc, cloud, nq, __shfl(cur, L), got, rows_scored, gs, gc);
            ++singles;
            if (lane == L) {
                found = fnd; nb = got; ns = gs > T(0) ? gs : T(-2);       // (-2: searched, no set either)
#pragma unroll
                for (int c = 0; c < CERT_CANDS; ++c) nc[c] = gc[c];
            }
        }
        if (need[q] == 2) {
            ct.q[at] = nb;
            if (qs) {
                if (!(nb > T(0)) && ns > T(0)) {                    // a candidate set: onto the cloud's list (full: no set, the query is searched in every iteration)
                    const int e = atomicAdd(ct.scount + cloud, 1);
                    if (e < n_full) ct.slist[(size_t)cloud * n_full + e] = (int)(at - (size_t)cloud * n_full);
                    else ns = T(-2);
                }
                qs[at] = nb > T(0) ? T(-1) : ns;
                if (ns > T(0)) {
#pragma unroll
                    for (int c = 0; c < CERT_CANDS; ++c) cands[at * CERT_CANDS + c] = nc[c];
                }
            }
            bud[q] = nb > spent ? nb : (ns > spent ? ns : nb);
            if (!(bud[q] > spent)) open = true;
        }
        if (need[q] && found != cur) {                          // a match that changed: for the accumulate of this iteration
            ct.cm[at] = found;
            if (ct.pend) {
                ct.pend[(size_t)cloud * n_full + qi] = found + 2;
                ct.gdirty[(size_t)cloud * ct.nwr + (qi >> 6)] = 1;
            } else if (spos) spos[(size_t)cloud * n_full + qi] = found;
        }
        if (pos < n && bud[q] > spent) qmin = min_t(qmin, bud[q]);
    }
    // the unit's filter afresh: its smallest budget, or 0 while a query of it has to be looked at in every iteration
    qmin = wave_min(qmin);
    const bool any_open = __any(open) != 0;
    if (lane == 0) *qu = any_open ? T(0) : qmin;
    // the statistics last (a wave's loads return behind its earlier atomics)
    const int eq = singles;
    if (eq > 0 && lane == 0) {
        if (pairs && rows_scored) atomicAdd(pairs + (blockIdx.x & (DICP_PAIR_SHARDS - 1)), rows_scored);
        if (ct.count && singles) atomicAdd(ct.count + CERT_SHARDS + (blockIdx.x & (CERT_SHARDS - 1)), singles);
        if (ct.cloud) atomicAdd(ct.cloud + (size_t)cloud * CERT_CLOUD + 1, eq);
    }
}

// The standing candidate sets of one cloud, 64 to a wave (entries [64 chunk, 64 chunk + 64) of the cloud's list): the new match of each is the set's best
// row under this iteration's pose -- same score(), equal scores -> lowest original index; the set's first row is the old match: never empty.  Four gathered
// rows per query instead of a search; a re-scored set counts a twelfth of a single-query search for the per-cloud switch.
template <typename T>
__device__ __forceinline__ void guard_sets(const T* __restrict__ src, const T* __restrict__ pose, const typename V4<T>::type* __restrict__ tgs4,
                                           const int32_t* __restrict__ tperm, const int32_t* __restrict__ qorder, int32_t* __restrict__ spos,
                                           int N, int n_full, int m_pad, const SweepCert<T>& ct, const int cloud, const int chunk) {
    using T4 = typename V4<T>::type;
    const int lane = threadIdx.x & (WAVE - 1);
    // (entries of this chunk were all written before this launch: the step kernel hands the list over in whole chunks, their tails filled with -1, and this
    //  launch's own appends go behind them)
    const int cnt = min(ct.scount[cloud], n_full), e = chunk * WAVE + lane;
    const int slot = e < cnt ? ct.slist[(size_t)cloud * n_full + e] : -1;
    const bool have = slot >= 0 && slot < n_full;
    const size_t at = (size_t)cloud * n_full + (have ? slot : 0);
    const T spent = cert_spent(ct.dcum + (size_t)cloud * ct.dstride + 2 * ct.k);
    const T b = ct.q[at], sb = set_budgets<T>(ct.set)[at];
    const bool on = have && !(b > spent) && b < T(0) && sb > spent;        // (the set still stands, and the query has no budget of its own)
    int rescored = 0;
    if (on) {
        const int32_t* cd = set_cands<T>(ct.set, N, n_full) + at * CERT_CANDS;
        const int cur = ct.cm[at], qi = qorder ? qorder[at] : (int)(at - (size_t)cloud * n_full);
        int cj[CERT_CANDS];
#pragma unroll
        for (int c = 0; c < CERT_CANDS; ++c) cj[c] = cd[c];
        const T4* __restrict__ tg = tgs4 + (size_t)cloud * m_pad;
        const int32_t* __restrict__ pm = tperm + (size_t)cloud * m_pad;
        T4 row[CERT_CANDS];
#pragma unroll
        for (int c = 0; c < CERT_CANDS; ++c) row[c] = tg[max(cj[c], 0)];       // (all four gathers in flight together)
        const T* sp = src + ((size_t)cloud * n_full + qi) * 3;
        const T p[3] = {sp[0], sp[1], sp[2]};
        T C[9], r[3], nx[3];
        load_pose(pose, cloud, C, r);
        query_point(C, r, p, nx);
        T best = inf_v<T>();
        int bj = max(cj[0], 0);
#pragma unroll
        for (int c = 0; c < CERT_CANDS; ++c) {
            const T scv = cj[c] >= 0 ? score<T, T4>(nx, row[c]) : inf_v<T>();
            if (scv < best) { best = scv; bj = cj[c]; }
            else if (scv == best && scv < inf_v<T>() && pm[cj[c]] < pm[bj]) bj = cj[c];
        }
        if (bj != cur) {                                            // a match that changed: for the accumulate of this iteration
            ct.cm[at] = bj;
            if (ct.pend) { ct.pend[(size_t)cloud * n_full + qi] = bj + 2; ct.gdirty[(size_t)cloud * ct.nwr + (qi >> 6)] = 1; }
            else if (spos) spos[(size_t)cloud * n_full + qi] = bj;
        }
        rescored = 1;
    }
    const int resc = __popcll(__ballot(rescored != 0));
    if (lane == 0 && ct.cloud && resc >= 12) atomicAdd(ct.cloud + (size_t)cloud * CERT_CLOUD + 1, resc / 12);
}

// The guard launch: a small grid of waves working through the lists the previous iteration's step kernel made (dicp_step_io.glist: the units that
// have anything to do, one list per XCD so that a cloud's units are searched on the XCD whose L2 holds its rows).  Near the pose the lists are
// (next to) empty and the launch is one load per wave.
template <typename T, int Q, int CH>
__global__ __launch_bounds__(BLOCK, (DICP_SWEEP_MINW > DICP_CERT_MINW ? DICP_CERT_MINW : DICP_SWEEP_MINW)) void knn_sweep_guard_kernel(DICP_SWEEP_PARAMS, const int32_t* __restrict__ glist, const int32_t* __restrict__ gcount, int glist_cap) {
    __shared__ typename V4<T>::type tiles[BLOCK / WAVE][SweepRing<T>::NT * WAVE];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int xcd = blockIdx.x & 7;
    const int units = (n_full + WAVE * Q - 1) / (WAVE * Q);
    const int count = min(gcount[xcd], glist_cap);
    const int32_t* __restrict__ list = glist + (size_t)xcd * glist_cap;
    const int stride = (int)(gridDim.x >> 3) * (BLOCK / WAVE);
    for (int e = (int)(blockIdx.x >> 3) * (BLOCK / WAVE) + wave; e < count; e += stride) {      // (wave-uniform)
        const int entry = list[e];
        if (entry < 0) {                                            // 64 of a cloud's candidate sets: -1 - (cloud nwr + chunk)
            const int id = -1 - entry, cloud = id / ct.nwr;
            if (cloud < N && ct.set && ct.slist) guard_sets<T>(src, pose, tgs4, tperm, qorder, spos, N, n_full, m_pad, ct, cloud, id - cloud * ct.nwr);
            continue;
        }
        const int cloud = entry / units, unit = entry - cloud * units;
        if (cloud >= N) continue;
        guard_unit<T, Q, CH>(src, pose, tgs4, tperm, qorder, bucket, brange, nbkt, idx, spos, pairs, N, n_full, m_full, m_pad, bpc, src_rows, tgt_rows, ct,
                             cloud, unit, tiles[wave]);
    }
}
#undef DICP_SWEEP_PARAMS
