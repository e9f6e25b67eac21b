// libdicp_hip.so -- per-call set-up: the key sort beyond the LDS sort (chunked radix sort through scratch, several blocks per cloud).
// Part of the one translation unit dicp_kernels.hip (included inside its anonymous namespace, in this order: kernels_setup.h, kernels_search.h, kernels_setup_sort.h, kernels_rows.h, kernels_accumulate.h, kernels_backward.h, kernels_soft_svd.h, kernels_host.h).
// ------------------------------------------------------------- key sort beyond the LDS sort
// Stable sort of a cloud's target x keys for clouds sort_keys_kernel cannot take: float64 keys, or more than 16384 slots.
// One block of 1024 threads per cloud, LSD radix over the order-preserving bit pattern of the key (4 or 8 digits of 8
// bits), keys and indices in global ping-pong buffers from the caller's scratch.  A pass = digit histogram of all M slots,
// then the slots chunk by chunk (16384 at a time, in order): per wave and round the ballot ranking of sort_keys_kernel,
// wave counts scanned per digit, scatter to base[digit] + offset; the bases advance from chunk to chunk, so equal keys
// keep their index order across chunks too.  (Written for the cell ids of the grid search experiment of round 2,
// profiles/r02_grid_knn_experiment.txt; with it no torch.sort is left on the ICP path.)
constexpr int GS_THREADS = 1024, GS_PER = 16, GS_CHUNK = GS_THREADS * GS_PER;
template <typename T> struct SortKey;
template <> struct SortKey<float> {
    using type = unsigned;
    static __device__ __forceinline__ unsigned of(float x) { return sortable_bits(x); }
    static __device__ __forceinline__ float back(unsigned u) { u ^= (u >> 31) ? 0x80000000u : 0xffffffffu; return __uint_as_float(u); }
    static __device__ __forceinline__ unsigned back_bits(unsigned u) { return u ^ ((u >> 31) ? 0x80000000u : 0xffffffffu); }
};
template <> struct SortKey<double> {
    using type = unsigned long long;
    static __device__ __forceinline__ unsigned long long of(double x) {
        unsigned long long u = (unsigned long long)__double_as_longlong(x + 0.0);      // -0 sorts as +0
        u ^= (u >> 63) ? ~0ull : 0x8000000000000000ull;
        return x != x ? ~0ull : u;                                                     // NaN of either sign sorts last
    }
    static __device__ __forceinline__ double back(unsigned long long u) { u ^= (u >> 63) ? 0x8000000000000000ull : ~0ull; return __longlong_as_double((long long)u); }
    static __device__ __forceinline__ unsigned long long back_bits(unsigned long long u) { return u ^ ((u >> 63) ? 0x8000000000000000ull : ~0ull); }
};

template <typename T>
__global__ __launch_bounds__(GS_THREADS) void sort_keys_big_kernel(const T* __restrict__ tgt, int c, int m_full, int m_pad, const T* __restrict__ frame, const int32_t* __restrict__ tgt_rows,
                                                                   T* __restrict__ keys_sorted, int32_t* __restrict__ tperm,
                                                                   typename SortKey<T>::type* __restrict__ gkey /* (N,2,m_pad) */, int32_t* __restrict__ gidx /* (N,2,m_pad) */) {
    using KT = typename SortKey<T>::type;
    __shared__ int cnt[GS_THREADS / WAVE][256];
    __shared__ int base[256], ctot[256];
    const int cloud = blockIdx.x, tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid >> 6;
    const T* __restrict__ rows = tgt + (size_t)cloud * m_full * c;
    const int m = rows_of(tgt_rows, cloud, m_full);
    const T* __restrict__ Fc = frame ? frame + (size_t)cloud * 12 : nullptr;
    KT* kbuf[2] = {gkey + (size_t)cloud * 2 * m_pad, gkey + (size_t)cloud * 2 * m_pad + m_pad};
    int32_t* ibuf[2] = {gidx + (size_t)cloud * 2 * m_pad, gidx + (size_t)cloud * 2 * m_pad + m_pad};
    // pass-0 input: the keys in slot order; pad slots keep the largest key there is (after every real row, NaN rows included)
    for (int j = tid; j < m_pad; j += GS_THREADS) {
        T q[3] = {T(0), T(0), T(0)};
        if (j < m) frame_apply<T>(Fc, rows + (size_t)j * c, q);
        kbuf[0][j] = j < m ? SortKey<T>::of(q[0]) : ~(KT)0;
        ibuf[0][j] = j;
    }
    __syncthreads();
    constexpr int PASSES = (int)sizeof(KT);
    for (int pass = 0; pass < PASSES; ++pass) {
        const int shift = pass * 8;
        const KT* kin = kbuf[pass & 1];
        KT* kout = kbuf[(pass & 1) ^ 1];
        const int32_t* iin = ibuf[pass & 1];
        int32_t* iout = ibuf[(pass & 1) ^ 1];
        if (tid < 256) base[tid] = 0;
        __syncthreads();
        for (int j = tid; j < m_pad; j += GS_THREADS) atomicAdd(&base[(unsigned)(kin[j] >> shift) & 0xffu], 1);
        __syncthreads();
        if (tid < WAVE) {                                       // exclusive scan of the 256 digit totals (4 per lane)
            int v[4], s = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[k] = base[lane * 4 + k]; s += v[k]; }
            int inc = s;
#pragma unroll
            for (int off = 1; off < WAVE; off <<= 1) { const int o = __shfl_up(inc, off); if (lane >= off) inc += o; }
            int run = inc - s;
#pragma unroll
            for (int k = 0; k < 4; ++k) { base[lane * 4 + k] = run; run += v[k]; }
        }
        __syncthreads();
        for (int c0 = 0; c0 < m_pad; c0 += GS_CHUNK) {
            KT key[GS_PER];
            int idx[GS_PER], rank[GS_PER];
            for (int d = lane; d < 256; d += WAVE) cnt[wave][d] = 0;
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int e = 0; e < GS_PER; ++e) {                  // striped: position = c0 + wave * 1024 + e * 64 + lane
                const int pos = c0 + wave * (WAVE * GS_PER) + e * WAVE + lane;
                const bool on = pos < m_pad;
                key[e] = on ? kin[pos] : (KT)0;
                idx[e] = on ? iin[pos] : -1;
                const unsigned d = on ? ((unsigned)(key[e] >> shift) & 0xffu) : 0x100u;     // 0x100: no slot here
                unsigned long long same = __ballot(on);
                if (!on) same = ~same;
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    const unsigned long long bal = __ballot((d >> b) & 1u);
                    same &= ((d >> b) & 1u) ? bal : ~bal;
                }
                const int below = __builtin_amdgcn_mbcnt_hi((unsigned)(same >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)same, 0u));
                const int bs = on ? cnt[wave][d & 0xff] : 0;    // every lane of a group reads before its first lane writes
                __builtin_amdgcn_wave_barrier();
                if (on && below == 0) cnt[wave][d] = bs + __popcll(same);
                __builtin_amdgcn_wave_barrier();
                rank[e] = bs + below;
            }
            __syncthreads();
            if (tid < 256) {
                int s = 0;
                for (int w = 0; w < GS_THREADS / WAVE; ++w) { const int v = cnt[w][tid]; cnt[w][tid] = s; s += v; }
                ctot[tid] = s;
            }
            __syncthreads();
#pragma unroll
            for (int e = 0; e < GS_PER; ++e) {
                if (idx[e] < 0) continue;
                const unsigned d = (unsigned)(key[e] >> shift) & 0xffu;
                const int pos = base[d] + cnt[wave][d] + rank[e];
                kout[pos] = key[e];
                iout[pos] = idx[e];
            }
            __syncthreads();
            if (tid < 256) base[tid] += ctot[tid];
            __syncthreads();
        }
    }
    // an even number of passes: the result is back in buffer 0
    for (int s = tid; s < m_pad; s += GS_THREADS) {
        keys_sorted[(size_t)cloud * m_pad + s] = SortKey<T>::back(kbuf[0][s]);
        tperm[(size_t)cloud * m_pad + s] = ibuf[0][s];
    }
}

// The same sort with SEVERAL blocks per cloud (a cloud of 65536 targets on one CU: 0.71 ms for 64 clouds; the other 192 CUs idle).
// Block (cloud, s) owns chunk s of GS_CHUNK consecutive slots of the pass's input.  A pass is one launch: every block reads the digit
// counts of all of the cloud's chunks (hist[pass], written by the launch before), derives where each digit of ITS chunk starts --
// all smaller digits of the cloud, then the same digit in the chunks before it: stable --, ranks its chunk with sort_keys_big_kernel's
// wave ballots and scatters.  The counts of the NEXT pass are a by-product of the scatter: a slot's destination chunk and next digit
// are known, counted in an LDS table and written out as this block's own row hist[pass + 1][cloud][s][destination chunk][digit] -- plain
// stores, no global atomics (1024 of them per block cost a pass 35 us), nothing to zero; the next launch sums the S rows.  The first
// launch makes the keys and counts digit 0; the last one writes keys_sorted / tperm.  1 + sizeof(key) launches, no waiting inside.
constexpr int GS_MAX_CHUNKS = 8;        // hist is chunks x chunks x 256 per cloud and pass (beyond 131072 slots: the one-block kernel)
template <typename T>
__global__ __launch_bounds__(GS_THREADS) void sort_big_keys_kernel(const T* __restrict__ tgt, int c, int m_full, int m_pad, int S, const T* __restrict__ frame,
                                                                   const int32_t* __restrict__ tgt_rows, typename SortKey<T>::type* __restrict__ gkey,
                                                                   int32_t* __restrict__ gidx, int32_t* __restrict__ hist /* (passes, N, S, S, 256) */, int N) {
    using KT = typename SortKey<T>::type;
    __shared__ int h[256];
    const int cloud = blockIdx.x / S, s = blockIdx.x - cloud * S, tid = threadIdx.x;
    const T* __restrict__ rows = tgt + (size_t)cloud * m_full * c;
    const int m = rows_of(tgt_rows, cloud, m_full);
    const T* __restrict__ Fc = frame ? frame + (size_t)cloud * 12 : nullptr;
    KT* k0 = gkey + (size_t)cloud * 2 * m_pad;
    int32_t* i0 = gidx + (size_t)cloud * 2 * m_pad;
    if (tid < 256) h[tid] = 0;
    __syncthreads();
    for (int j = s * GS_CHUNK + tid; j < min(m_pad, (s + 1) * GS_CHUNK); j += GS_THREADS) {
        T q[3] = {T(0), T(0), T(0)};
        if (j < m) frame_apply<T>(Fc, rows + (size_t)j * c, q);
        const KT key = j < m ? SortKey<T>::of(q[0]) : ~(KT)0;       // pad slots keep the largest key there is (after every real row, NaN rows included)
        k0[j] = key;
        i0[j] = j;
        atomicAdd(&h[(unsigned)key & 0xffu], 1);
    }
    __syncthreads();
    if (tid < 256)          // pass 0 reads the slots where they are: chunk s holds what chunk s counted
        for (int q = 0; q < S; ++q) hist[((((size_t)cloud) * S + s) * S + q) * 256 + tid] = q == s ? h[tid] : 0;
}

// The scatter is staged through LDS: a slot's rank inside its chunk's digit order is known before anything is written, so the chunk is
// first put in that order in LDS (half a chunk at a time, 32-bit words: keys, then indices) and then written out by consecutive lanes
// -- a digit's slots of a chunk go to consecutive addresses, so the stores of a wave are runs instead of 64 scattered words (a pass over
// uniformly distributed digits: 62 -> 3x us for 64 clouds of 65536).
template <typename T, bool LAST>
__global__ __launch_bounds__(GS_THREADS) void sort_big_pass_kernel(int pass, int m_pad, int S, typename SortKey<T>::type* __restrict__ gkey, int32_t* __restrict__ gidx,
                                                                   int32_t* __restrict__ hist, int N, T* __restrict__ keys_sorted, int32_t* __restrict__ tperm) {
    using KT = typename SortKey<T>::type;
    constexpr int HALF = GS_CHUNK / 2;
    __shared__ unsigned stage[HALF];                            // (its first 16 KiB double as the per-wave digit counts until the ranks are final)
    __shared__ int base[256], dstart[256];
    __shared__ int nh[LAST ? 1 : GS_MAX_CHUNKS * 256];
    int (*cnt)[256] = reinterpret_cast<int (*)[256]>(stage);
    static_assert(sizeof(int) * (GS_THREADS / WAVE) * 256 <= sizeof(unsigned) * HALF, "the counts fit the staging buffer");
    const int cloud = blockIdx.x / S, s = blockIdx.x - cloud * S, tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid >> 6;
    const int shift = pass * 8;
    const KT* kin = gkey + (size_t)cloud * 2 * m_pad + (size_t)(pass & 1) * m_pad;
    KT* kout = gkey + (size_t)cloud * 2 * m_pad + (size_t)((pass & 1) ^ 1) * m_pad;
    const int32_t* iin = gidx + (size_t)cloud * 2 * m_pad + (size_t)(pass & 1) * m_pad;
    int32_t* iout = gidx + (size_t)cloud * 2 * m_pad + (size_t)((pass & 1) ^ 1) * m_pad;
    const int32_t* hp = hist + ((size_t)pass * N + cloud) * S * S * 256;
    if (!LAST)
        for (int e = tid; e < S * 256; e += GS_THREADS) nh[e] = 0;
    if (tid < 256) {        // digit tid: all of the cloud's slots with it, and those in the chunks before this one (rows: who counted them)
        int tot = 0, before = 0;
        for (int r = 0; r < S; ++r)
            for (int q = 0; q < S; ++q) { const int v = hp[(r * S + q) * 256 + tid]; tot += v; before += q < s ? v : 0; }
        base[tid] = tot;
        dstart[tid] = before;       // (parked until the scan below has read base)
    }
    __syncthreads();
    if (tid < WAVE) {                                           // exclusive scan of the 256 digit totals (4 per lane)
        int v[4], t = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[k] = base[lane * 4 + k]; t += v[k]; }
        int inc = t;
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) { const int o = __shfl_up(inc, off); if (lane >= off) inc += o; }
        int run = inc - t;
#pragma unroll
        for (int k = 0; k < 4; ++k) { base[lane * 4 + k] = run + dstart[lane * 4 + k]; run += v[k]; }
    }
    const int c0 = s * GS_CHUNK;
    const int live = min(GS_CHUNK, m_pad - c0);                 // slots of this chunk
    KT key[GS_PER];
    int idx[GS_PER], rank[GS_PER];
    for (int d = lane; d < 256; d += WAVE) cnt[wave][d] = 0;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int e = 0; e < GS_PER; ++e) {                  // striped: position = c0 + wave * 1024 + e * 64 + lane
        const int pos = c0 + wave * (WAVE * GS_PER) + e * WAVE + lane;
        const bool on = pos < m_pad;
        key[e] = on ? kin[pos] : (KT)0;
        idx[e] = on ? iin[pos] : -1;
        const unsigned d = on ? ((unsigned)(key[e] >> shift) & 0xffu) : 0x100u;     // 0x100: no slot here
        unsigned long long same = __ballot(on);
        if (!on) same = ~same;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long bal = __ballot((d >> b) & 1u);
            same &= ((d >> b) & 1u) ? bal : ~bal;
        }
        const int below = __builtin_amdgcn_mbcnt_hi((unsigned)(same >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)same, 0u));
        const int bs = on ? cnt[wave][d & 0xff] : 0;    // every lane of a group reads before its first lane writes
        __builtin_amdgcn_wave_barrier();
        if (on && below == 0) cnt[wave][d] = bs + __popcll(same);
        __builtin_amdgcn_wave_barrier();
        rank[e] = bs + below;
    }
    __syncthreads();
    if (tid < 256) {                                            // per digit: the waves' starts inside the digit, and the chunk's total
        int t = 0;
        for (int w = 0; w < GS_THREADS / WAVE; ++w) { const int v = cnt[w][tid]; cnt[w][tid] = t; t += v; }
        dstart[tid] = t;
    }
    __syncthreads();
    if (tid < WAVE) {                                           // where each digit starts in the chunk's own digit order
        int v[4], t = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[k] = dstart[lane * 4 + k]; t += v[k]; }
        int inc = t;
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) { const int o = __shfl_up(inc, off); if (lane >= off) inc += o; }
        int run = inc - t;
#pragma unroll
        for (int k = 0; k < 4; ++k) { dstart[lane * 4 + k] = run; run += v[k]; }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < GS_PER; ++e) {                          // rank -> place in the chunk's digit order
        if (idx[e] < 0) { rank[e] = -1; continue; }
        const unsigned d = (unsigned)(key[e] >> shift) & 0xffu;
        rank[e] += dstart[d] + cnt[wave][d];
    }
    __syncthreads();                                            // (the counts are dead: the buffer is the stage now)
    // a staged word's digit tells where it goes: base[d] + (place - dstart[d])
    constexpr int WORDS = (int)(sizeof(KT) / 4);
    for (int half = 0; half < 2; ++half) {
        const int lo = half * HALF;
        if (lo >= live) break;
        int dest[HALF / GS_THREADS];
        // keys: the word that holds the current digit first (it places the slot), then the other word of a 64-bit key
#pragma unroll
        for (int wsel = 0; wsel < WORDS + 1; ++wsel) {          // WORDS key words, then the index
            const bool is_idx = wsel == WORDS;
            const int word = wsel == 0 ? (shift >= 32 ? 1 : 0) : (WORDS == 2 && wsel == 1 ? (shift >= 32 ? 0 : 1) : 0);
#pragma unroll
            for (int e = 0; e < GS_PER; ++e) {
                const int at = rank[e] - lo;
                if (rank[e] >= 0 && at >= 0 && at < HALF) stage[at] = is_idx ? (unsigned)idx[e] : (unsigned)(key[e] >> (32 * word));
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < HALF / GS_THREADS; ++u) {
                const int at = u * GS_THREADS + tid;
                if (lo + at >= live) continue;
                const unsigned w = stage[at];
                if (wsel == 0) {
                    const unsigned d = (w >> (shift & 31)) & 0xffu;
                    dest[u] = base[d] + (lo + at - dstart[d]);
                    if (!LAST && WORDS == 1) atomicAdd(&nh[(dest[u] / GS_CHUNK) * 256 + ((w >> ((shift + 8) & 31)) & 0xffu)], 1);
                }
                const int pos = dest[u];
                if (is_idx) {
                    if (LAST) tperm[(size_t)cloud * m_pad + pos] = (int32_t)w; else iout[pos] = (int32_t)w;
                } else if (WORDS == 1) {
                    if (LAST) reinterpret_cast<unsigned*>(keys_sorted)[(size_t)cloud * m_pad + pos] = (unsigned)SortKey<T>::back_bits((KT)w);
                    else reinterpret_cast<unsigned*>(kout)[pos] = w;
                } else {
                    // 64-bit keys travel as two words; the finished key is put back into floating point by the caller's last sweep below
                    reinterpret_cast<unsigned*>(kout)[2 * (size_t)pos + word] = w;
                }
            }
            __syncthreads();
        }
    }
    if (WORDS == 2) {
        // 64-bit keys: the next pass's counts and the result conversion need the whole key: one more look at what this block wrote
        // would race with other blocks' writes into kout -- so count from the registers instead (the slot's destination is recomputed)
#pragma unroll
        for (int e = 0; e < GS_PER; ++e) {
            if (rank[e] < 0) continue;
            const unsigned d = (unsigned)(key[e] >> shift) & 0xffu;
            const int pos = base[d] + (rank[e] - dstart[d]);
            if (!LAST) atomicAdd(&nh[(pos / GS_CHUNK) * 256 + ((unsigned)(key[e] >> (shift + 8)) & 0xffu)], 1);
            else keys_sorted[(size_t)cloud * m_pad + pos] = SortKey<T>::back(key[e]);
        }
    }
    if (!LAST) {
        __syncthreads();
        int32_t* hn = hist + (((size_t)(pass + 1) * N + cloud) * S + s) * S * 256;
        for (int e = tid; e < S * 256; e += GS_THREADS) hn[e] = nh[e];
    }
}
