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

namespace {

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
template <typename T>
__global__ __launch_bounds__(CC_THREADS) void search_frame_kernel(const T* __restrict__ tgt, int c, int m, const int32_t* __restrict__ tgt_rows,
                                                                  double quantum, int directions, T* __restrict__ frame) {
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
    __shared__ float s_ctr[3], s_ext[CC_THREADS / WAVE];
    __shared__ int s_cost[SF_DIRS];
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
    float ext = fin ? fmaxf(fabsf(dx), fmaxf(fabsf(dy), fabsf(dz))) : 0.f;
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
        __syncthreads();
        if (wave < SF_DIRS) {
            int cst = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int h = dhist[wave][4 * lane + q]; cst += h * h; }
#pragma unroll
            for (int off = WAVE / 2; off > 0; off >>= 1) cst += __shfl_xor(cst, off);
            if (lane == 0) s_cost[wave] = cst;
        }
        __syncthreads();
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
// QO_STAGE: queries per cloud whose permutation is assembled in LDS (16384 -> 32 KiB, several clouds per CU; 65536 -> 128 KiB)
template <typename T, int QO_STAGE>
__global__ __launch_bounds__(QO_THREADS) void query_order_kernel(const T* __restrict__ src, const T* __restrict__ pose,
                                                                 const T* __restrict__ brange, int nbkt_range, int N, int n_full,
                                                                 int32_t* __restrict__ qorder, const T* __restrict__ w,
                                                                 T* __restrict__ src_s, T* __restrict__ w_s, int reproducible,
                                                                 const int32_t* __restrict__ spos_prev, int m_pad,
                                                                 const T* __restrict__ skeys, int kstride, int mt_full, const int32_t* __restrict__ table,
                                                                 const int32_t* __restrict__ src_rows, const int32_t* __restrict__ tgt_rows) {
    __shared__ int cnt[QO_BUCKETS];
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
    for (int b = tid; b < QO_BUCKETS; b += QO_THREADS) cnt[b] = 0;
    T q[4] = {T(1), T(0), T(0), T(0)};
    if (pose) { const T* pp = pose + (size_t)cloud * 12; q[0] = pp[0]; q[1] = pp[1]; q[2] = pp[2]; q[3] = pp[9]; }
    const T xlo = brange[(size_t)cloud * 2];
    const T tscale = brange[(size_t)cloud * 2 + 1];                                       // table buckets per unit x
    const T scale = tscale * (T(QO_BUCKETS) / T(nbkt_range));                             // ordering buckets per unit x
    // rank ordering: the cloud's sorted target x keys, as floats, in LDS (QO_KEYS of them: bigger clouds fall back to x buckets)
    const bool ranked = QO_STAGE <= 16384 && skeys && table && !spos_prev && mt <= QO_KEYS && nbkt_range <= QO_TABLE;
    if (ranked) {
        const T* __restrict__ keys = skeys + (size_t)cloud * m_pad * kstride;
        for (int j = tid; j < mt; j += QO_THREADS) lkeys[j] = (float)keys[(size_t)j * kstride];
        for (int j = tid; j <= nbkt_range; j += QO_THREADS) ltab[j] = table[(size_t)cloud * (nbkt_range + 1) + j];
    }
    auto bucket_of = [&](int i) {
        if (spos_prev) {        // bucket = rank of the query's previous match among the sorted targets: equal-POPULATION buckets,
                                // whatever the density of the cloud along x (an outlier cannot coarsen them)
            const int sp = spos_prev[(size_t)cloud * n + i];
            return sp < 0 ? QO_BUCKETS - 1 : (int)(((long)min(sp, m_pad - 1) * QO_BUCKETS) / m_pad);
        }
        const T* p = src + ((size_t)cloud * n + i) * 3;
        const T x = fma_t(q[0], p[0], fma_t(q[1], p[1], fma_t(q[2], p[2], q[3])));
        T f = (x - xlo) * scale;
        f = f > T(0) ? (f < T(QO_BUCKETS - 1) ? f : T(QO_BUCKETS - 1)) : T(0);             // NaN and out-of-range -> end buckets
        return (int)f;
    };
    __syncthreads();
    // one returning LDS add per query gives its bucket AND its rank inside the bucket; both stay in registers while
    // the counters are turned into offsets (LDS atomics are the cost of this kernel: ~137 cycles per wave-instruction)
    constexpr int PER = 16;                                 // register-resident up to PER * QO_THREADS queries per cloud
    int bk[PER], rk[PER];
    const bool small = n <= PER * QO_THREADS;
    if (small && ranked) {
        // rank of every query's x among the sorted target keys, from the LDS copy of the keys: a full binary search per
        // query (14 LDS reads; from global memory the same chain of dependent loads took 144 us per launch)
#pragma unroll 1
        for (int e = 0; e < PER; ++e) {
            const int i = e * QO_THREADS + tid;
            int bb = -1, rr = 0;
            if (i < n) {
                const T* p = src + ((size_t)cloud * n + i) * 3;
                const float x = (float)fma_t(q[0], p[0], fma_t(q[1], p[1], fma_t(q[2], p[2], q[3])));
                // the coarse table (also in LDS) brackets the lower bound: ~4 steps on an even cloud instead of 14
                float f = (x - (float)xlo) * (float)tscale;
                f = f > 0.f ? (f < (float)nbkt_range ? f : (float)nbkt_range) : 0.f;
                const int tb = (int)f;
                int lo = ltab[tb], hi = ltab[min(tb + 1, nbkt_range)];
                if (!(lo <= hi) || (lo > 0 && !(lkeys[lo - 1] < x)) || (hi < mt && lkeys[hi] < x)) { lo = 0; hi = mt; }  // rounding at an edge
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (lkeys[mid] < x) lo = mid + 1; else hi = mid; }
                bb = (int)(((long)lo * (QO_BUCKETS - 1)) / max(mt, 1));
                rr = atomicAdd(&cnt[bb], 1);
            }
#pragma unroll
            for (int k = 0; k < PER; ++k) if (k == e) { bk[k] = bb; rk[k] = rr; }
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
    T* q;                           // (N,n) budgets by QUERY (like spos)
    T* qu;                          // (N,units): per unit of the sweep, a lower bound of its certified queries' budgets (a filter, never a proof)
    const T* dcum; int dstride;     // (N,dstride): (M_k, e_k) pairs per iteration
    int k;                          // this iteration
    int32_t* count;                 // (128) or NULL: [0,64) units searched again, [64,128) single queries, sharded by block
    void* set;                      // optional candidate sets (see search_point): (N,n) T set budgets by query, then (N,n,4) int32 sorted positions
    int32_t* cloud;                 // (N,CERT_CLOUD) or NULL, per cloud: [0] units / [1] single queries searched again in this iteration; [2] the
                                    // state the step kernel keeps: 0 on, -1 on with one strike, k > 0 off for k more iterations (CERT_OFF_FOR_GOOD:
                                    // for the rest of the call), CERT_RECERTIFY: this iteration's guard searches every unit with certifying
                                    // sweeps; [3] its units (written by the searches: "a certified iteration ran"); [4] the last back-off length
};
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
template <typename T, int Q, int CH, bool CERT>
__device__ __forceinline__ void sweep_unit(const T* __restrict__ src, const T* __restrict__ pose,
                                           const typename V4<T>::type* __restrict__ tgs4,
                                           const int32_t* __restrict__ tperm, const int32_t* __restrict__ qorder,
                                           const int32_t* __restrict__ bucket, const T* __restrict__ brange, int nbkt,
                                           int32_t* __restrict__ idx, int32_t* __restrict__ spos,
                                           unsigned long long* __restrict__ pairs,
                                           int n_full, int m_full, int m_pad, int src_sorted,
                                           const int32_t* __restrict__ src_rows, const int32_t* __restrict__ tgt_rows, const SweepCert<T>& ct,
                                           const int cloud, const int unit, typename V4<T>::type* __restrict__ ring) {
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
            const T* sp = src + ((size_t)cloud * n_full + (src_sorted ? pos : qi[q])) * 3;      // src_sorted: rows already in slot order
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
        // sorted position of the winner (indexed like idx, by the query): what the windowed backward consumes
        if (spos) spos[(size_t)cloud * n_full + qi[q]] = (bo == 0x7fffffff || bo >= m) ? -1 : bs;
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
            // no certificate: -(k + 2) says "searched at iteration k" -- spent for every later iteration, not searched twice in this one
            ct.q[(size_t)cloud * n_full + qi[q]] = bq > T(0) ? bq : cert_mark<T>(ct.k);
            if (ct.set) set_budgets<T>(ct.set)[(size_t)cloud * n_full + qi[q]] = T(-1);      // (a new match: whatever candidate set the query had is void)
            if (bq > T(0)) qmin = min_t(qmin, bq); else ++nunc;
        } else if (ct.q) {
            ct.q[(size_t)cloud * n_full + qi[q]] = cert_mark<T>(ct.k);                      // plain search of a unit inside a certified loop
            if (ct.set) set_budgets<T>(ct.set)[(size_t)cloud * n_full + qi[q]] = T(-1);
        }
    }
    if (CERT) {
        // the unit's filter value: its smallest budget -- or 0 ("look at me every iteration") when more queries than the accumulate
        // should search one by one have no certificate at all
        qmin = wave_min(qmin);
        int tot = 0;
#pragma unroll
        for (int q = 0; q < Q; ++q) tot += __popcll(__ballot(nunc > q));
        if (lane == 0) ct.qu[(size_t)cloud * ((n_full + WAVE * Q - 1) / (WAVE * Q)) + unit] = tot > CERT_SLOT_MAX ? T(0) : qmin;
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
}

#define DICP_SWEEP_PARAMS const T* __restrict__ src, const T* __restrict__ pose, const typename V4<T>::type* __restrict__ tgs4, \
        const int32_t* __restrict__ tperm, const int32_t* __restrict__ qorder, const int32_t* __restrict__ bucket, const T* __restrict__ brange, int nbkt, \
        int32_t* __restrict__ idx, int32_t* __restrict__ spos, unsigned long long* __restrict__ pairs, \
        int N, int n_full, int m_full, int m_pad, int bpc, int src_sorted, const int32_t* __restrict__ src_rows, const int32_t* __restrict__ tgt_rows, SweepCert<T> ct
#define DICP_SWEEP_MINW ((Q == 2 && CH == 8 && sizeof(T) == 4) ? SWEEP_MINW_Q2C8 : 1)

// Every unit of every cloud: block (cloud, blk) of the XCD-aware grid, one unit per wave.
template <typename T, int Q, int CH, bool CERT>
__global__ __launch_bounds__(BLOCK, DICP_SWEEP_MINW) void knn_sweep_kernel(DICP_SWEEP_PARAMS) {
    __shared__ typename V4<T>::type tiles[BLOCK / WAVE][SweepRing<T>::NT * WAVE];
    int cloud, blk;
    if (!decode_block(bpc, N, cloud, blk)) return;
    const int wave = threadIdx.x >> 6;
    sweep_unit<T, Q, CH, CERT>(src, pose, tgs4, tperm, qorder, bucket, brange, nbkt, idx, spos, pairs, n_full, m_full, m_pad, src_sorted, src_rows, tgt_rows, ct,
                               cloud, blk * (BLOCK / WAVE) + wave, tiles[wave]);
}

// What the on-the-spot search of one query needs besides the query (certifying loop only).
template <typename T> struct PointSearch {
    const T* pose;                                  // (N,12) search pose of this iteration
    const typename V4<T>::type* tgs4; const int32_t* tperm; const int32_t* bucket; const T* brange; int nbkt;
    const int32_t* tgt_rows; int m_full, m_pad;
    unsigned long long* pairs;
    SweepCert<T> ct;                                // ct.dcum == NULL: no budget is checked (the iteration's search has just written them)
    int32_t* spos;                                  // (N,n) this iteration's matches: read, and rewritten where a query is searched
    int32_t* spos_next;                             // optional (N,n): the next iteration's, started as a copy of this one's
};

// The search of ONE query by one wave (all lanes carry the same arguments): the query's previous match, scored under the current
// pose, bounds the best score from above, and with it the slab of sorted rows that can hold the new match -- the same bound the
// sweep prunes with, so a row outside the slab can never beat the kept minimum.  The lanes score the slab's rows 64 at a time with
// the score() every search form uses; equal scores resolve to the lowest ORIGINAL index: index for index the match of a full
// search.  Returns the match's sorted position (-1: none) and leaves the query's new budget in `budget`.
template <typename T>
__device__ __forceinline__ int search_point(const PointSearch<T>& ps, const int cloud, const T* nx, const int prev, T& budget, unsigned long long& rows_scored,
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

// Guard of a certified iteration: one wave per unit of the sweep, as in knn_sweep_kernel.  A unit none of whose certified queries can
// have spent its budget leaves at once; of the others, the ones with more than CERT_SLOT_MAX spent budgets are searched again as a
// unit (cheaper per query than one by one, and what keeps a batch that suddenly moves far from falling back on single searches);
// the rest is left to the accumulate that follows, which searches spent queries on the spot.
template <typename T, int Q, int CH>
__global__ __launch_bounds__(BLOCK, DICP_SWEEP_MINW) void knn_sweep_guard_kernel(DICP_SWEEP_PARAMS) {
    __shared__ typename V4<T>::type tiles[BLOCK / WAVE][SweepRing<T>::NT * WAVE];
    int cloud, blk;
    if (!decode_block(bpc, N, cloud, blk)) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & (WAVE - 1);
    const int unit = blk * (BLOCK / WAVE) + wave, units = (n_full + WAVE * Q - 1) / (WAVE * Q);
    const int n = rows_of(src_rows, cloud, n_full);
    if (unit * (WAVE * Q) >= n) return;
    const T* dk = ct.dcum + (size_t)cloud * ct.dstride + 2 * ct.k;
    const T spent = cert_spent(dk);
    const T step = ct.k > 0 ? dk[0] - dk[-2] : inf_v<T>();     // how far the cloud's queries can have moved in the last step
    T* qu = ct.qu + (size_t)cloud * units + unit;
    const T v = *qu;
    if (ct.cloud && unit == 0 && lane == 0) { ct.cloud[(size_t)cloud * CERT_CLOUD + 3] = units; ct.cloud[(size_t)cloud * CERT_CLOUD + 5] = ct.set ? 1 : 0; }
    bool plain;
    const int cstate = ct.cloud ? ct.cloud[(size_t)cloud * CERT_CLOUD + 2] : 0;
    if (cstate > 0) plain = true;                               // this cloud's certificates are off (step kernel): every unit, plainly
    else if (cstate == CERT_RECERTIFY) plain = false;           // ... and this is the iteration that tries them again: every unit, certifying
    else if (v < T(0)) plain = step > -v;                       // plain mode (below): certify again once the steps are at most -v
    else {
        if (v > spent) return;
        int bad = 0, live = 0;
        T qmin = inf_v<T>();
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int pos = unit * (WAVE * Q) + q * WAVE + lane;
            if (pos < n) {
                const size_t at = (size_t)cloud * n_full + qorder[(size_t)cloud * n_full + pos];
                const T b = ct.q[at];
                ++live;
                if (b > spent) qmin = min_t(qmin, b);
                else {                                          // no certificate of its own: a candidate set that still stands is as good (the accumulate re-scores it)
                    // A budget that the poses' motion has spent counts against the unit, as ever.  A query that never had a certificate of its own
                    // (b < 0: a mark) may have a candidate set: one that stands is as good as a budget; none tried yet (-1): the accumulate's
                    // search of this query will try; "no set either" (-2) or a spent set count against the unit.
                    const T sb = (ct.set && b < T(0)) ? set_budgets<T>(ct.set)[at] : T(-2);
                    if (sb > spent) qmin = min_t(qmin, sb); else if (sb != T(-1)) ++bad;
                }
            }
        }
        int nbad = 0, nlive = 0;
#pragma unroll
        for (int q = 0; q < Q; ++q) { nbad += __popcll(__ballot(bad > q)); nlive += __popcll(__ballot(live > q)); }
        if (nbad <= CERT_SLOT_MAX) {                            // the few spent ones are left to the accumulate
            qmin = wave_min(qmin);
            if (lane == 0) *qu = qmin;
            return;
        }
        // three quarters of the last search's budgets did not survive one step, and the steps are not shrinking fast (less than halved
        // since the one before): certifying this unit is wasted work while the cloud moves like this.  It is searched plainly (cheaper,
        // no budgets) until the steps have halved.
        const T step_before = ct.k > 1 ? dk[-2] - dk[-4] : inf_v<T>();
        plain = 4 * nbad >= 3 * nlive && step > T(0) && step < inf_v<T>() && T(2) * step > step_before;
        if (plain && lane == 0) *qu = -T(0.5) * step;
    }
    if (plain) sweep_unit<T, Q, CH, false>(src, pose, tgs4, tperm, qorder, bucket, brange, nbkt, idx, spos, pairs, n_full, m_full, m_pad, src_sorted, src_rows, tgt_rows, ct,
                                           cloud, unit, tiles[wave]);
    else       sweep_unit<T, Q, CH, true>(src, pose, tgs4, tperm, qorder, bucket, brange, nbkt, idx, spos, pairs, n_full, m_full, m_pad, src_sorted, src_rows, tgt_rows, ct,
                                          cloud, unit, tiles[wave]);
    // the counters LAST: vector memory operations return in order, so a unit that counted itself first waited for its add -- one of up to
    // 128 to the same word when a whole cloud is searched again -- before its first load came back (a cloud with its certificates off:
    // 53 us per launch instead of the plain kernel's 31)
    if (lane == 0 && ct.count) atomicAdd(ct.count + (blockIdx.x & (CERT_SHARDS - 1)), 1);
    if (lane == 0 && ct.cloud) atomicAdd(ct.cloud + (size_t)cloud * CERT_CLOUD, 1);
}
#undef DICP_SWEEP_PARAMS

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

// ------------------------------------------------------------- gather / scatter
// Row-indexed copies.  One thread per ELEMENT (consecutive lanes walk a row, so reads of a row and writes of the
// output are as coalesced as the data allows); all blocks of a cloud run on ONE XCD (decode_block): the rows they
// pick at random then come out of one L2 instead of being fetched into eight.
template <int C>
__device__ __forceinline__ void split_cols(unsigned e, int c, int& row, int& col) {
    if (C > 0) { row = (int)(e / (unsigned)C); col = (int)(e - (unsigned)row * C); }
    else       { row = (int)(e / (unsigned)c); col = (int)(e - (unsigned)row * (unsigned)c); }
}

constexpr int ROWS_U = 4;       // elements per thread: both loads of an element depend on each other (index, then row), so
                                // the kernels are pure latency unless each thread keeps several elements in flight

template <typename T, int C>
__global__ __launch_bounds__(BLOCK) void gather_kernel(const T* __restrict__ tgt, const int32_t* __restrict__ idx,
                                                       int N, int n, int m, int c, int bpc, T* __restrict__ out) {
    const unsigned total = (unsigned)n * (unsigned)c;
    int b, blk;
    if (!decode_block(bpc, N, b, blk)) return;
    const unsigned e0 = (unsigned)blk * (BLOCK * ROWS_U) + threadIdx.x;
    {
        int j[ROWS_U], k[ROWS_U];
#pragma unroll
        for (int u = 0; u < ROWS_U; ++u) {
            const unsigned e = min(e0 + u * BLOCK, total - 1);
            int i;
            split_cols<C>(e, c, i, k[u]);
            j[u] = min(max(idx[(size_t)b * n + i], 0), m - 1);
        }
        T v[ROWS_U];
#pragma unroll
        for (int u = 0; u < ROWS_U; ++u) v[u] = tgt[((size_t)b * m + j[u]) * c + k[u]];
#pragma unroll
        for (int u = 0; u < ROWS_U; ++u)
            if (e0 + u * BLOCK < total) out[(size_t)b * total + e0 + u * BLOCK] = v[u];
    }
}

template <typename T, int C>
__global__ __launch_bounds__(BLOCK) void scatter_add_kernel(const T* __restrict__ gout, const int32_t* __restrict__ idx,
                                                            int N, int n, int m, int c, int bpc, T* __restrict__ gtgt) {
    int b, blk;
    if (!decode_block(bpc, N, b, blk)) return;
    const unsigned e = (unsigned)blk * BLOCK + threadIdx.x;
    if (e >= (unsigned)n * (unsigned)c) return;
    int i, k;
    split_cols<C>(e, c, i, k);
    {
        const int j = min(max(idx[(size_t)b * n + i], 0), m - 1);
        unsafeAtomicAdd(&gtgt[((size_t)b * m + j) * c + k], gout[(size_t)b * n * c + e]);
    }
}

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
// CERT (certified iterations of the sweep loop; idx = this iteration's sorted positions): the point's budget is checked where its
// match is read, a spent one is searched on the spot by the whole wave (search_point), and the matches are handed on to the next
// iteration's buffer.
constexpr bool PAIR_ROWS = true;     // (plain launches, 7 waves per SIMD: 58 -> 49 us; certified ones, 5 waves because of their search code: 55 -> 55 -- and 6 or 7 waves spill: 115 / 130 us)
template <typename T, int MODE, bool CERT = false>
__global__ __launch_bounds__(BLOCK, (CERT && sizeof(T) == 4) ? 5 : 1) void accumulate_kernel(WeightParams P, const T* __restrict__ src, const T* __restrict__ tgt, int c /* elements per row of tgt */,
                                                           const int32_t* __restrict__ idx, const T* __restrict__ pose,
                                                           const T* __restrict__ w_init, const T* __restrict__ alive,
                                                           int N, int n, int m, int bpc, T* __restrict__ partials,
                                                           T* __restrict__ w_out, long w_stride, const int32_t* __restrict__ src_rows, PointSearch<T> ps,
                                                           const T* __restrict__ w_prev /* optional: a frozen cloud (alive = 0) keeps its previous weights, ICP.py:224-226 */) {
    __shared__ T red[(BLOCK / WAVE) * NACC_PAD];
    __shared__ short set_list[CERT ? BLOCK / WAVE : 1][CERT ? ACC_PTS / (BLOCK / WAVE) : 1];      // per wave: its points (offsets in the block's range) with a standing candidate set
    int cloud, blk;
    if (!decode_block(bpc, N, cloud, blk)) return;
    const int nc = rows_of(src_rows, cloud, n);             // ragged batches: rows past the cloud's own carry weight 0 (ICP.py:386-398)
    const int end = min(nc, (blk + 1) * ACC_PTS);
    unsigned long long rows_scored = 0;                     // this wave's on-the-spot searches: rows scored, searches made (wave-uniform)
    int singles = 0, rescored = 0;                          // ... and this LANE's candidate sets re-scored
    if (CERT && ps.ct.dcum && !(ps.ct.cloud && ps.ct.cloud[(size_t)cloud * CERT_CLOUD + 2] > 0)) {      // (certificates off for this cloud: the guard launch has just searched every unit)
        // first the budgets of this block's points (before the sums' registers are live): spent ones are searched again, one query at a
        // time by the whole wave; the thread that owns the point rewrites its match and budget and reads them back below
        const int lane = threadIdx.x & (WAVE - 1);
        const T spent = cert_spent(ps.ct.dcum + (size_t)cloud * ps.ct.dstride + 2 * ps.ct.k);
        constexpr int ROUNDS = ACC_PTS / BLOCK;
        T b[ROUNDS];
#pragma unroll
        for (int t = 0; t < ROUNDS; ++t) {
            const int i = blk * ACC_PTS + t * BLOCK + threadIdx.x;
            b[t] = i < end ? ps.ct.q[(size_t)cloud * n + i] : inf_v<T>();
        }
        T* __restrict__ qs = ps.ct.set ? set_budgets<T>(ps.ct.set) : nullptr;           // candidate sets (search_point): budgets by query, then the rows
        int32_t* __restrict__ cands = ps.ct.set ? set_cands<T>(ps.ct.set, N, n) : nullptr;
        bool isset[ROUNDS];
#pragma unroll
        for (int t = 0; t < ROUNDS; ++t) isset[t] = false;
        if (qs) {
            // Standing candidate sets first, ALL rounds of the wave at once: the points with one (8 % of them on scanned surfaces, in every wave)
            // are listed in LDS and re-scored by consecutive lanes -- one chain of dependent loads (set budget, rows, scores) per 64 such
            // points instead of one per round of the block (planar scenes: 110 -> 98 us per launch; the plain accumulate: 58).
            const int wv = threadIdx.x >> 6;
            T sbv[ROUNDS];
#pragma unroll
            for (int t = 0; t < ROUNDS; ++t) {
                const int i = blk * ACC_PTS + t * BLOCK + (int)threadIdx.x;
                const bool open = i < end && !(b[t] > spent) && b[t] != cert_mark<T>(ps.ct.k);
                sbv[t] = open ? qs[(size_t)cloud * n + i] : T(-1);
            }
            int total = 0;
#pragma unroll
            for (int t = 0; t < ROUNDS; ++t) {
                isset[t] = sbv[t] > spent;
                const unsigned long long mk = __ballot(isset[t]);
                if (isset[t]) set_list[wv][total + __builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u))] = (short)(t * BLOCK + (int)threadIdx.x);
                total += __popcll(mk);
            }
            if (total) {                                        // (wave-uniform)
                __builtin_amdgcn_wave_barrier();
                using T4 = typename V4<T>::type;
                T Cs[9], rs[3];
                load_pose(ps.pose, cloud, Cs, rs);
                const T4* __restrict__ tg = ps.tgs4 + (size_t)cloud * ps.m_pad;
                const int32_t* __restrict__ pm = ps.tperm + (size_t)cloud * ps.m_pad;
                for (int s0 = 0; s0 < total; s0 += WAVE) {
                    const int kk = s0 + lane;
                    if (kk < total) {
                        const size_t pt = (size_t)cloud * n + blk * ACC_PTS + set_list[wv][kk];
                        const T* sp = src + pt * 3;
                        const T p[3] = {sp[0], sp[1], sp[2]};
                        const int32_t* cd = cands + pt * CERT_CANDS;
                        int cj[CERT_CANDS];
                        T4 row[CERT_CANDS];
#pragma unroll
                        for (int c = 0; c < CERT_CANDS; ++c) cj[c] = cd[c];
#pragma unroll
                        for (int c = 0; c < CERT_CANDS; ++c) row[c] = tg[max(cj[c], 0)];       // (all four gathers in flight together)
                        T nx[3];
                        query_point(Cs, rs, p, nx);
                        // the new match is the set's best row (same score(), equal scores -> lowest original index; the set's first row is the old match: never empty)
                        T best = inf_v<T>();
                        int bj = max(cj[0], 0);
#pragma unroll
                        for (int c = 0; c < CERT_CANDS; ++c) {
                            const T sc = cj[c] >= 0 ? score<T, T4>(nx, row[c]) : inf_v<T>();
                            if (sc < best) { best = sc; bj = cj[c]; }
                            else if (sc == best && sc < inf_v<T>() && pm[cj[c]] < pm[bj]) bj = cj[c];
                        }
                        ps.spos[pt] = bj;
                        ++rescored;
                    }
                }
                __threadfence_block();                          // (the matches are read back by the points' own lanes below)
            }
        }
#pragma unroll 1
        for (int t = 0; t < ROUNDS; ++t) {
            const bool redo = !(b[t] > spent) && b[t] != cert_mark<T>(ps.ct.k) && !isset[t];     // spent, never certifiable, NaN -- unless this iteration's
            unsigned long long todo = __ballot(redo);                                             // guard launch has just searched it, or its candidate set stands
            if (!todo) continue;                                // (wave-uniform; the common case)
            const size_t pt = (size_t)cloud * n + min(blk * ACC_PTS + t * BLOCK + (int)threadIdx.x, n - 1);
            T p[3] = {T(0), T(0), T(0)}, nb = T(-1), ns = T(-2);
            int j = -1, nc[CERT_CANDS];
#pragma unroll
            for (int c = 0; c < CERT_CANDS; ++c) nc[c] = -1;
            if (redo) { const T* sp = src + pt * 3; p[0] = sp[0]; p[1] = sp[1]; p[2] = sp[2]; j = ps.spos[pt]; }
            T Cs[9], rs[3];
            load_pose(ps.pose, cloud, Cs, rs);
            while (todo) {
                const int L = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const T pq[3] = {__shfl(p[0], L), __shfl(p[1], L), __shfl(p[2], L)};
                T nx[3], got, gs;
                int gc[CERT_CANDS];
                query_point(Cs, rs, pq, nx);
                const int found = search_point<T>(ps, cloud, nx, __shfl(j, L), got, rows_scored, gs, gc);
                ++singles;
                if (lane == L) {
                    j = found; nb = got; ns = gs > T(0) ? gs : T(-2);       // (-2: searched, no set either)
#pragma unroll
                    for (int c = 0; c < CERT_CANDS; ++c) nc[c] = gc[c];
                }
            }
            if (redo) {
                ps.spos[pt] = j;
                ps.ct.q[pt] = nb;
                if (qs) {
                    qs[pt] = nb > T(0) ? T(-1) : ns;
                    if (ns > T(0)) {
#pragma unroll
                        for (int c = 0; c < CERT_CANDS; ++c) cands[pt * CERT_CANDS + c] = nc[c];
                    }
                }
            }
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
    for (int base = blk * ACC_PTS; base < end; base += BLOCK) {            // (2 or 4 points in flight per thread measured slower: 78 / 85 vs 72 us)
        const int i = base + (int)threadIdx.x;
        const bool on = i < end;
        const size_t pt = (size_t)cloud * n + (on ? i : end - 1);
        const T* sp = src + pt * 3;
        const T p[3] = {sp[0], sp[1], sp[2]};
        const int jm = ix ? ix[pt] : (on ? i : end - 1);    // ix == NULL: tgt holds one row per source point
        if (CERT && ps.spos_next && on) ps.spos_next[pt] = jm;
        const int j = min(max(jm, 0), m - 1);
        T y[3], nrm[3] = {T(0), T(0), T(0)};
        if (MODE == MODE_PT2PL && PAIR_ROWS) {
            // The 24-byte row gather: two lanes share the two rows of their two points -- each loads its half (12 bytes) of both, so a wave
            // instruction touches 32 rows instead of 64 (the gather is bound by the cache's look-ups per instruction, not by bytes), and the
            // halves change hands inside the lane pair.
            const int half = threadIdx.x & 1;
            const int je = __shfl(j, (int)(threadIdx.x & (WAVE - 1)) & ~1), jo = __shfl(j, (int)(threadIdx.x & (WAVE - 1)) | 1);
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
        if (!on) continue;
        PointState<T> s;
        point_forward<T, MODE>(P, C, r, p, y, nrm, (w_init ? w_init[pt] : T(1)) * live, acc, s);
        // (a frozen cloud: all its weights are zero, and the reference then keeps the previous iteration's -- written here, by 1024 threads per
        //  block instead of the step kernel's one wave per cloud: 97 us of every tolerance-mode iteration at the benchmark shape)
        if (w_out) w_out[(size_t)cloud * w_stride + i] = (w_prev && live == T(0)) ? w_prev[(size_t)cloud * w_stride + i] : s.w;
    }
    block_reduce_store<T, NACC, NACC_PAD>(acc, partials + ((size_t)cloud * bpc + blk) * NACC_PAD, red);
    if (CERT) {             // the statistics of this wave's on-the-spot searches, after everything else
        // a re-scored candidate set costs about a twelfth of a single-query search (4 gathered rows against a slab): counted as such for the switch
        int resc = rescored;
#pragma unroll
        for (int o = WAVE / 2; o > 0; o >>= 1) resc += __shfl_xor(resc, o);
        const int eq = singles + resc / 12;
        if (eq > 0 && (threadIdx.x & (WAVE - 1)) == 0) {
            if (ps.pairs && rows_scored) atomicAdd(ps.pairs + (blockIdx.x & (DICP_PAIR_SHARDS - 1)), rows_scored);
            if (ps.ct.count && singles) atomicAdd(ps.ct.count + CERT_SHARDS + (blockIdx.x & (CERT_SHARDS - 1)), singles);
            if (ps.ct.cloud) atomicAdd(ps.ct.cloud + (size_t)cloud * CERT_CLOUD + 1, eq);
        }
    }
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
    io.pose_in = (const char*)B.poses + (size_t)k * N * 12 * es; io.pose_out = (char*)B.poses + (size_t)(k + 1) * N * 12 * es;
    io.frame = B.frame; io.pose_search_out = B.poses_search ? (char*)B.poses_search + (size_t)(k + 1) * N * 12 * es : nullptr;
    io.delta = (char*)B.deltas + (size_t)k * 6 * es; io.delta_stride = (int64_t)B.K * 6;
    io.cost = (char*)B.costs + (size_t)k * es; io.cost_prev = k > 0 ? (const char*)B.costs + (size_t)(k - 1) * es : nullptr;
    io.cost_stride = B.K;
    io.areg = B.areg ? B.areg + (size_t)k * N * 36 : nullptr;
    io.alive = (const char*)B.alive + (size_t)k * N * es; io.alive_out = (char*)B.alive + (size_t)(k + 1) * N * es;
    io.converged = B.converged; io.iterations = B.iterations; io.matched_ratio = B.matched_ratio;
    io.n_start = B.n_start; io.n_matched = B.n_matched;
    io.w_cur = (char*)B.w + (size_t)k * B.w_iter * es;
    io.w_prev = k > k0 ? (const char*)B.w + (size_t)(k - 1) * B.w_iter * es : (const char*)B.w_prev0; io.w_stride = B.w_stride;
    io.n_not_converged = B.counters + k;
    io.rmax = B.rmax; io.dcum = B.dcum; io.dcum_stride = 2 * (B.K + 1);
    io.cert_cloud = B.cert_cloud;
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
    __shared__ int s_copy;
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
        double* d6 = sout; double* Cn = sout + 6; double* rn = sout + 15;
        unpack_sym6(sacc + ACC_A, sA);
        // solve with the pose untouched first so delta can be rounded to T like the reference's
        step_forward(sA, sacc + ACC_B, io.dim, spose, spose + 9, d6, Cn, rn, sAreg);
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
            const double cn = io.frame ? sqrt(smisc[6] * smisc[6] + smisc[7] * smisc[7] + smisc[8] * smisc[8]) : 0.0;      // (the search frame adds t, |t| = |centre|, to r)
            const double pnm = sqrt(p0[0] * p0[0] + p0[1] * p0[1] + p0[2] * p0[2]);
            dc[3] = (T)(8.0 * ulp * (pnm + rad + sqrt(rn2) + cn + 1.0) * 1.0001);
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
                if (costly || state > 0) cc[7] += 1;            // iterations of this call in which the cloud's certificates did not pay (the host's call-to-call hint reads it)
                cc[0] = 0; cc[1] = 0; cc[3] = 0; cc[6] = 0;
            }
        }
    }
    __syncthreads();
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
    const int m_pad = min((mc + KNN_PAD - 1) / KNN_PAD * KNN_PAD, B.m_pad);     // ragged batches: the cloud's own rows only
    {
        const T4* __restrict__ g = (const T4*)B.tgt4 + (size_t)cloud * B.m_pad;
        for (int j = tid; j < m_pad; j += BLOCK) tg[j] = g[j];
    }
    __syncthreads();
    const T* __restrict__ src = (const T*)B.src + (size_t)cloud * n * 3;
    const T* __restrict__ tgt = (const T*)B.tgt + (size_t)cloud * m * c;
    const T* __restrict__ w_init = B.w_init ? (const T*)B.w_init + (size_t)cloud * n : nullptr;
    for (int k = k0; k < k1; ++k) {
        T C[9], r[3];
        load_pose((const T*)B.poses + (size_t)k * N * 12, cloud, C, r);
        T Cs[9], rs[3];                                     // the search's pose: [Q C | Q r + t] (packed rows are Q y + t)
        {
            const T pw[12] = {C[0], C[1], C[2], C[3], C[4], C[5], C[6], C[7], C[8], r[0], r[1], r[2]};
            const T* F = B.frame ? (const T*)B.frame + (size_t)cloud * 12 : nullptr;
#pragma unroll
            for (int e = 0; e < 9; ++e) Cs[e] = frame_pose_entry<T>(F, pw, e);
#pragma unroll
            for (int e = 0; e < 3; ++e) rs[e] = frame_pose_entry<T>(F, pw, 9 + e);
        }
        const T live = ((const T*)B.alive)[(size_t)k * N + cloud];
        int32_t* __restrict__ idx_k = B.idx + (B.idx_per_iter ? (size_t)k * N * n : 0) + (size_t)cloud * n;
        T* __restrict__ w_k = (T*)B.w + (size_t)k * B.w_iter + (size_t)cloud * B.w_stride;
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
                                            const int32_t* __restrict__ spos, const int32_t* __restrict__ spos_ref,
                                            const int32_t* __restrict__ qorder,
                                            const T* __restrict__ pose, const T* __restrict__ w_s, const T* __restrict__ alive,
                                            const T* gs, const T* gb, int n, int m_pad, int spb, int bpc,
                                            T* __restrict__ gsrc_s, T* __restrict__ slab /* (N,bpc,WT,CV) */,
                                            T* __restrict__ gts_far /* (N,m_pad,CV) */,
                                            T* __restrict__ gw_s, T* part_out, const int32_t* __restrict__ src_rows, int cloud, int blk) {
    // overwrite: first launch into uninitialised accumulators -- gsrc_s / gw_s / the slab windows are written, not added to
    constexpr int CV = (MODE == MODE_PT2PL) ? 6 : 3;
    __shared__ T red[(BLOCK / WAVE) * NBWD_PAD];
    constexpr int SPB = 4 * BLOCK;                          // window_slots() never exceeds this
    __shared__ T contrib[SPB * CV];                         // target-row contribution of each of the block's slots
    __shared__ int head[WT], next[SPB];                     // per window row: list of the slots that matched it
    const int tid = threadIdx.x;
    const int nc = rows_of(src_rows, cloud, n);             // ragged batches: slots past the cloud's own carry weight 0: no work, zero gradient
    const int s0 = blk * spb, s1 = min(nc, s0 + spb), s1_all = min(n, s0 + spb);
    const int32_t* __restrict__ sp_c = spos + (size_t)cloud * n;
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
    for (int u = 0; u < U; ++u) pos[u] = on[u] ? min(max(sp_c[pos[u]], 0), m_pad - 1) : 0;     // -1 (no neighbour: non-finite input) -> row 0
    if (slab)
        for (int k = tid; k < hi - lo; k += BLOCK) head[k] = -1;
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
                if (overwrite && base + u * BLOCK + tid < s1_all) {      // a pad slot of a ragged batch: its accumulators start at zero
                    const size_t pz = (size_t)cloud * n + base + u * BLOCK + tid;
                    gsrc_s[pz * 3] = gsrc_s[pz * 3 + 1] = gsrc_s[pz * 3 + 2] = T(0);
                    if (gw_s) gw_s[pz] = T(0);
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
                } else {
                    T* row = gfar + (size_t)pos[u] * CV;
                    unsafeAtomicAdd(&row[0], gy[0]); unsafeAtomicAdd(&row[1], gy[1]); unsafeAtomicAdd(&row[2], gy[2]);
                    if (MODE == MODE_PT2PL) { unsafeAtomicAdd(&row[3], gn[0]); unsafeAtomicAdd(&row[4], gn[1]); unsafeAtomicAdd(&row[5], gn[2]); }
                }
            }
        }
    }
    __syncthreads();
    if (slab) {     // one thread per window row; this block is the only writer of its slab rows
        T* out = slab + ((size_t)cloud * bpc + blk) * (WT * CV);
        for (int rr = tid; rr < hi - lo; rr += BLOCK) {
            int h = head[rr];
            if (h < 0 && !overwrite) continue;
            T sum[CV];
#pragma unroll
            for (int k = 0; k < CV; ++k) sum[k] = T(0);
            for (int guard = 0; h >= 0 && guard < SPB; ++guard) {       // every slot is on at most one list
#pragma unroll
                for (int k = 0; k < CV; ++k) sum[k] += contrib[h * CV + k];
                h = next[h];
            }
#pragma unroll
            for (int k = 0; k < CV; ++k) out[rr * CV + k] = overwrite ? sum[k] : out[rr * CV + k] + sum[k];
        }
    }
    block_reduce_store<T, NBWD, NBWD_PAD>(acc, part_out, red);
}

template <typename T, int MODE, int WT, bool overwrite>
__global__ __launch_bounds__(BLOCK) void accumulate_bwd_window_kernel(WeightParams P, const T* __restrict__ src_s, const T* __restrict__ tgt_s, int c,
                                                                      const int32_t* __restrict__ spos, const int32_t* __restrict__ spos_ref,
                                                                      const int32_t* __restrict__ qorder,
                                                                      const T* __restrict__ pose, const T* __restrict__ w_s, const T* __restrict__ alive,
                                                                      const T* __restrict__ gs, const T* __restrict__ gb,
                                                                      int N, int n, int m_pad, int spb, int bpc,
                                                                      T* __restrict__ gsrc_s, T* __restrict__ slab /* (N,bpc,WT,CV) */,
                                                                      T* __restrict__ gts_far /* (N,m_pad,CV) */,
                                                                      T* __restrict__ gw_s, T* __restrict__ bwd_partials, const int32_t* __restrict__ src_rows,
                                                                      const int32_t* __restrict__ skip /* optional (N), see accumulate_bwd_kernel */) {
    int cloud, blk;
    if (!decode_block(bpc, N, cloud, blk)) return;
    T* part_out = bwd_partials + ((size_t)cloud * bpc + blk) * NBWD_PAD;
    if (!overwrite && skip && skip[cloud]) {                // (the first launch initialises the accumulators: it always runs)
        if (threadIdx.x < NBWD_PAD) part_out[threadIdx.x] = T(0);
        return;
    }
    window_body<T, MODE, WT, overwrite>(P, src_s, tgt_s, c, spos, spos_ref, qorder, pose, w_s, alive, gs + (size_t)cloud * 36, gb + (size_t)cloud * 6,
                                        n, m_pad, spb, bpc, gsrc_s, slab, gts_far, gw_s, part_out, src_rows, cloud, blk);
}

// gtgt[b][tperm[s]][col] += gts_far[b][s][col] + sum over the blocks whose window covers sorted row s of their
// slab rows: the once-per-call end of the windowed backward (also undoes the sorted target order).
constexpr int WR_U = 4;      // gradient elements per thread
constexpr int WR_B = 8;      // window blocks per round of loads (16: 160 us instead of 118 -- registers)
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
    for (int b0 = 0; b0 < bpc; b0 += MAXB) {
        __syncthreads();
        for (int b = tid; b < min(MAXB, bpc - b0); b += BLOCK)
            origin[b] = window_origin(spos_ref + (size_t)cloud * n, qorder ? qorder + (size_t)cloud * n : nullptr, b0 + b, spb, rows_of(src_rows, cloud, n), m_pad, WT);
        __syncthreads();
        const int nb = min(MAXB, bpc - b0);
        // the block loop is the OUTER one: all of a thread's elements have their (predicated) loads of WR_B window blocks in
        // flight together -- the kernel is bound by how many dependent rounds of loads a thread makes, not by bytes
        for (int bb = 0; bb < nb; bb += WR_B) {
            T v[WR_U][WR_B];
#pragma unroll
            for (int u = 0; u < WR_U; ++u) {
                const int e = min(e0 + u * BLOCK + tid, m * cv - 1);
                const int s = e / CV;
#pragma unroll
                for (int k = 0; k < WR_B; ++k) {
                    const int b = min(bb + k, nb - 1);
                    const int lo = origin[b];
                    const bool cov = bb + k < nb && s >= lo && s < lo + WT;
                    v[u][k] = cov ? slab[((size_t)cloud * bpc + b0 + b) * (WT * CV) + (size_t)(e - lo * CV)] : T(0);
                }
            }
#pragma unroll
            for (int u = 0; u < WR_U; ++u)
#pragma unroll
                for (int k = 0; k < WR_B; ++k) acc[u] += v[u][k];
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
    bool ended = B.bwd_skip && B.bwd_skip[cloud] == 2;      // (this cloud's reverse sweep ended in an earlier chunk)
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
    const T* __restrict__ dlt = (const T*)B.deltas + (size_t)cloud * B.K * 6;
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
        const T* pose_k = (const T*)B.poses + (size_t)k * N * 12;
        if (tid < NBWD) sg[tid] = spart[tid] + sgo[tid];
        if (tid < 9) sC[tid] = (double)pose_k[(size_t)cloud * 12 + tid];
        if (tid < 6) sd[tid] = (double)dlt[(size_t)k * 6 + tid];
        if (tid < 36) sAreg[tid] = B.areg[((size_t)k * N + cloud) * 36 + tid];
        if (B.bwd_skip && tid >= WAVE && tid < 2 * WAVE) {  // the largest step of the EARLIER iterations, per component (second wave: lanes over iterations)
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
            if (B.bwd_skip) {
                const SkipArgs<T> sk{B.bwd_skip, B.bwd_mref, (const T*)B.alive + (size_t)k * N, B.bwd_live ? B.bwd_live + k : nullptr, B.bwd_skip_eps, k};
                s_skip = skip_decision(sGs, sGb, sAreg, sdmax, dim, cloud, sk, sk.alive_k[cloud] != T(0), B.bwd_mref[cloud]);
                B.bwd_skip[cloud] = s_skip;
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
        const T live = ((const T*)B.alive)[(size_t)k * N + cloud];
        const int32_t* __restrict__ idx_k = B.idx + (size_t)k * N * n + (size_t)cloud * n;
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

// The TAIL of the windowed reverse sweep of big clouds (dicp_loop_buffers.bwd_tail_from): the iterations k1-1 .. 0 in ONE launch.
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
// whole and makes progress -- and every wait is bounded anyway (on running out it raises the error word and goes on: wrong sums, no hang).
// On exit gpose_out holds the cotangent of pose_0 INCLUDING the last pose sums (dicp_pose_grad_out is then called without partials).
// A word handed from one block to another inside a launch: agent-scope atomic accesses (sc1: coherent across the XCDs' L2s)
__device__ __forceinline__ void coherent_store(float* p, float v)   { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void coherent_store(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float  coherent_load(const float* p)  { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double coherent_load(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <typename T, int MODE, int WT>
__global__ __launch_bounds__(BLOCK) void bwd_tail_kernel(WeightParams P, dicp_loop_buffers B, int N, int n, int dim, int spb, int bpc,
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
    bool ended = B.bwd_skip[cloud] == 2;                    // (decided by an earlier launch: the same for all of the cloud's blocks)
    if (ended && blk != 0) return;
    if (tid < 12) sgo[tid] = gpose_in[(size_t)cloud * 12 + tid];
    if (tid == 32) smref = B.bwd_mref[cloud];
    if (tid == 33) s_timeout = 0;
    const T* __restrict__ dlt = (const T*)B.deltas + (size_t)cloud * B.K * 6;
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
        const T* pose_k = (const T*)B.poses + (size_t)k * N * 12;
        const T* alive_k = (const T*)B.alive + (size_t)k * N;
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
        if (tid >= 3 * WAVE && tid < 3 * WAVE + 36) sAreg[tid - 3 * WAVE] = B.areg[((size_t)k * N + cloud) * 36 + (tid - 3 * WAVE)];
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
                    if (!nan && 16.0 * worst <= B.bwd_skip_eps * smref) verdict = 2;
                    else {
                        if (!nan && mm > smref) { smref = mm; if (blk == 0) B.bwd_mref[cloud] = mm; }
                        if (blk == 0 && B.bwd_live) atomicAdd(B.bwd_live + k, 1);
                    }
                }
                if (blk == 0) B.bwd_skip[cloud] = verdict;  // (nobody reads it again in this launch: the cloud's blocks all hold the same verdict)
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
        window_body<T, MODE, WT, false>(P, (const T*)B.src, (const T*)B.tgt, B.c, B.spos + (size_t)k * N * n, B.spos_ref, B.qorder, pose_k, (const T*)B.w_init, alive_k,
                                        sGsT, sGbT, n, B.m_pad, spb, bpc, gsrc_s, slab, (T*)B.gts_far, gw_s, spub, B.src_rows, cloud, blk);
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
            __hip_atomic_fetch_add(arrive + cloud, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!(k == 0 && blk != 0) && !s_timeout) {      // (after the last iteration only block 0 still needs the sums; a block waits in vain at most once)
                const int want = gen * bpc;
                int spins = 0;
                while (__hip_atomic_load(arrive + cloud, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                    if (++spins > (1 << 20)) {              // ~0.5 s: the cloud's other blocks are not running (dicp_bwd_tail_max_blocks keeps that from happening)
                        atomicExch(arrive + N, 1);
                        if (B.bwd_live) atomicExch(B.bwd_live + B.K, 1);
                        s_timeout = 1;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(8);
                }
            }
            asm volatile("" ::: "memory");                  // (the sums are read with agent-scope loads after the barrier below: nothing to invalidate)
        }
        if (k == 0 && blk != 0) return;
        __syncthreads();
        cur = out;
    }
    if (blk == 0) {
        fold(cur);
        if (tid < 12) gpose_out[(size_t)cloud * 12 + tid] = sg[tid];
    }
}

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


struct CertAcc {              // what the accumulate of a certified iteration needs for its on-the-spot searches (PointSearch, untyped)
    const void* pose_search; const void* tgs4; const int32_t* tperm; const int32_t* bucket; const void* brange; int nbkt;
    const int32_t* tgt_rows; int m_full, m_pad; unsigned long long* pairs;
    void* q; void* qu; const void* dcum; int dstride, k; int32_t* count;
    int32_t* spos; int32_t* spos_next; int32_t* cloud; void* set;
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

}  // namespace

// ======================================================================== C ABI
extern "C" {

int dicp_abi_version(void) { return DICP_ABI_VERSION; }
int dicp_padded_targets(int m) { return m <= 0 ? 0 : ((m + KNN_PAD - 1) / KNN_PAD) * KNN_PAD; }
int dicp_accumulate_blocks(int n) { return n <= 0 ? 0 : (n + ACC_PTS - 1) / ACC_PTS; }
int dicp_search_frame(int dtype, const void* tgt, int c, const int32_t* tgt_rows, int N, int m, double quantum, int directions, void* frame, void* stream) {
    if (!tgt || !frame) return DICP_ERR_NULL;
    if (bad_dtype(dtype)) return DICP_ERR_DTYPE;
    if (N <= 0 || m <= 0 || c < 3 || !(quantum >= 0.0)) return DICP_ERR_SHAPE;
    begin_launch();
    if (dtype == DICP_F32) search_frame_kernel<float><<<N, CC_THREADS, 0, (hipStream_t)stream>>>((const float*)tgt, c, m, tgt_rows, quantum, directions, (float*)frame);
    else                   search_frame_kernel<double><<<N, CC_THREADS, 0, (hipStream_t)stream>>>((const double*)tgt, c, m, tgt_rows, quantum, directions, (double*)frame);
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
    int rc = dicp_search_frame(dtype, tgt, c, tgt_rows, N, m, quantum, directions, frame, stream);
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

int dicp_query_order(int dtype, const void* src, const void* pose, const void* brange, int nbkt, int N, int n, int32_t* qorder,
                     const void* w, void* src_s, void* w_s, int reproducible, const int32_t* spos_prev, int m_pad,
                     const void* skeys, const int32_t* bucket, int m, const int32_t* src_rows, const int32_t* tgt_rows, void* stream) {
    if (!src || !brange || !qorder || (w_s && !w)) return DICP_ERR_NULL;
    if (bad_dtype(dtype)) return DICP_ERR_DTYPE;
    if (N <= 0 || n <= 0 || nbkt <= 0 || ((spos_prev || skeys) && m_pad <= 0) || (skeys && (m <= 0 || m > m_pad))) return DICP_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    begin_launch();
#define DICP_QO(T, S) query_order_kernel<T, S><<<N, QO_THREADS, 0, st>>>((const T*)src, (const T*)pose, (const T*)brange, nbkt, N, n, qorder, \
        (const T*)w, (T*)src_s, (T*)w_s, reproducible, spos_prev, m_pad, (const T*)skeys, 1, m, bucket, src_rows, tgt_rows)
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

struct CertArgs {             // certifying search: budgets (NULL q: plain search), motion bounds; guard: the launch of a certified iteration
    void* q; void* qu; const void* dcum; int dstride; int k; int32_t* count; bool guard; int32_t* cloud; void* set;
};

static int sweep_launch(int dtype, const void* src, const void* pose, const void* tgs4, const int32_t* tperm,
                        const int32_t* qorder, const int32_t* bucket, const void* brange, int nbkt,
                        int N, int n, int m, int m_pad, int32_t* idx, int32_t* spos, unsigned long long* pairs, int cfg, Rows rw, hipStream_t st,
                        CertArgs ca = CertArgs{}, const void* f16_image = nullptr) {
    hipEvent_t ev0, ev1;
    take_launch_events(ev0, ev1);                                       // (null unless a timed loop set them for this launch)
    const int src_sorted = (cfg & DICP_SWEEP_SRC_SORTED) ? 1 : 0;      // src holds the rows in qorder's slot order
    cfg &= ~DICP_SWEEP_SRC_SORTED;
    if (src_sorted && !qorder) return DICP_ERR_NULL;
    if (cfg == 0) cfg = sweep_auto_cfg(N, n);
    // plain float32 searches in units of 128 queries, given the image of the sorted rows: the scoring runs on the matrix cores (knn_f16.hip)
    if (f16_image && dtype == DICP_F32 && !ca.q && cfg == SWEEP_CFG_BIG)
        return dicp_tu::knn_f16_sweep(src, pose, tgs4, const_cast<void*>(f16_image), tperm, qorder, bucket, brange, nbkt, rw.src, rw.tgt, N, n, m, m_pad, idx, spos,
                                      pairs, src_sorted, ev0, ev1, st);
    const int Q = sweep_queries_per_lane(cfg);
    if (Q <= 0) return DICP_ERR_ENUM;
    const int units = (n + WAVE * Q - 1) / (WAVE * Q);                  // waves per cloud
    const int bpc = (units + BLOCK / WAVE - 1) / (BLOCK / WAVE);
#define DICP_SWEEP_ARGS(T) (const T*)src, (const T*)pose, (const typename V4<T>::type*)tgs4, tperm, qorder, bucket, (const T*)brange, nbkt, idx, spos, pairs, \
        N, n, m, m_pad, bpc, src_sorted, rw.src, rw.tgt
#define DICP_SWEEP_C(T, Q, CH, CERT, CT) hipExtLaunchKernelGGL((knn_sweep_kernel<T, Q, CH, CERT>), dim3(grid_for(N, bpc)), dim3(BLOCK), 0, st, ev0, ev1, 0, DICP_SWEEP_ARGS(T), CT)
#define DICP_SWEEP_L(T, Q, CH, CT) hipExtLaunchKernelGGL((knn_sweep_guard_kernel<T, Q, CH>), dim3(grid_for(N, bpc)), dim3(BLOCK), 0, st, ev0, ev1, 0, DICP_SWEEP_ARGS(T), CT)
#define DICP_SWEEP_CG(T, Q, CH, CT) do { if (ca.guard) DICP_SWEEP_L(T, Q, CH, CT); else DICP_SWEEP_C(T, Q, CH, true, CT); } while (0)
#define DICP_SWEEP(T, Q, CH) do { SweepCert<T> none{}; DICP_SWEEP_C(T, Q, CH, false, none); } while (0)
    if (ca.q) {             // certifying search, or the guard launch of a certified iteration
        if (!ca.qu || !ca.dcum || !spos || !qorder) return DICP_ERR_ENUM;
        if (dtype == DICP_F32) {
            SweepCert<float> c{(float*)ca.q, (float*)ca.qu, (const float*)ca.dcum, ca.dstride, ca.k, ca.count, ca.set, ca.cloud};
            if (cfg == 2) DICP_SWEEP_CG(float, 2, 8, c); else if (cfg == 4) DICP_SWEEP_CG(float, 1, 16, c); else DICP_SWEEP_CG(float, 1, 8, c);
        } else {
            SweepCert<double> c{(double*)ca.q, (double*)ca.qu, (const double*)ca.dcum, ca.dstride, ca.k, ca.count, ca.set, ca.cloud};
            if (cfg == 2) DICP_SWEEP_CG(double, 2, 8, c); else DICP_SWEEP_CG(double, 1, 8, c);
        }
        return launch_status();
    }
    if (dtype == DICP_F32) {
        if (cfg == 2) DICP_SWEEP(float, 2, 8); else if (cfg == 4) DICP_SWEEP(float, 1, 16); else DICP_SWEEP(float, 1, 8);
    } else {
        if (cfg == 2) DICP_SWEEP(double, 2, 8); else DICP_SWEEP(double, 1, 8);
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
                   int N, int n, int m, int m_pad, int32_t* idx, int32_t* spos, unsigned long long* pairs, int cfg, const void* f16_image, void* stream) {
    if (!src || !tgs4 || !tperm || !bucket || !brange || (!idx && !spos)) return DICP_ERR_NULL;
    if (bad_dtype(dtype)) return DICP_ERR_DTYPE;
    if (N <= 0 || n <= 0 || m <= 0 || nbkt <= 0 || m_pad != dicp_padded_targets(m)) return DICP_ERR_SHAPE;
    if ((uintptr_t)tgs4 % (dtype == DICP_F32 ? 16 : 32)) return DICP_ERR_ALIGN;
    begin_launch();
    return sweep_launch(dtype, src, pose, tgs4, tperm, qorder, bucket, brange, nbkt, N, n, m, m_pad, idx, spos, pairs, cfg, Rows{src_rows, tgt_rows}, (hipStream_t)stream,
                        CertArgs{}, f16_image);
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
        PointSearch<T> ps{}; \
        if (ca) { \
            ps.pose = (const T*)ca->pose_search; ps.tgs4 = (const typename V4<T>::type*)ca->tgs4; ps.tperm = ca->tperm; ps.bucket = ca->bucket; ps.brange = (const T*)ca->brange; \
            ps.nbkt = ca->nbkt; ps.tgt_rows = ca->tgt_rows; ps.m_full = ca->m_full; ps.m_pad = ca->m_pad; ps.pairs = ca->pairs; \
            ps.ct = SweepCert<T>{(T*)ca->q, (T*)ca->qu, (const T*)ca->dcum, ca->dstride, ca->k, ca->count, ca->set, ca->cloud}; ps.spos = ca->spos; ps.spos_next = ca->spos_next; \
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
                                B->idx, nullptr, B->pairs, (B->knn_variant >> 8) & 0xff, B->tgt_f16, stream);
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
    static int cap[2] = {-1, -1};
    if (bad_dtype(dtype)) return 0;
    if (cap[dtype] >= 0) return cap[dtype];
    int dev = 0, cus = 0, a = 0, b = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
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
    const int per_cu = a < b ? a : b, xcds = 8;
    cap[dtype] = (per_cu > 0 && cus >= xcds) ? (per_cu * (cus / xcds)) / 2 : 0;
    return cap[dtype];
}

static int accumulate_bwd_window_go(int dtype, const dicp_weight_params* prm, const void* src_s, const void* tgt_s, int c,
                                    const int32_t* spos, const int32_t* spos_ref, const int32_t* qorder, const void* pose, const void* w_s,
                                    const void* alive, const void* gs, const void* gb, const int32_t* src_rows, int N, int n, int m_pad, void* gsrc_s, void* slab,
                                    void* gts_far, void* gw_s, void* bwd_partials, int overwrite, void* stream, const int32_t* skip);
int dicp_accumulate_bwd_window(int dtype, const dicp_weight_params* prm, const void* src_s, const void* tgt_s, int c,
                               const int32_t* spos, const int32_t* spos_ref, const int32_t* qorder, const void* pose, const void* w_s,
                               const void* alive, const void* gs, const void* gb, const int32_t* src_rows, int N, int n, int m_pad, void* gsrc_s, void* slab,
                               void* gts_far, void* gw_s, void* bwd_partials, int overwrite, void* stream) {
    return accumulate_bwd_window_go(dtype, prm, src_s, tgt_s, c, spos, spos_ref, qorder, pose, w_s, alive, gs, gb, src_rows, N, n, m_pad, gsrc_s, slab, gts_far, gw_s,
                                    bwd_partials, overwrite, stream, nullptr);
}
static int accumulate_bwd_window_go(int dtype, const dicp_weight_params* prm, const void* src_s, const void* tgt_s, int c,
                                    const int32_t* spos, const int32_t* spos_ref, const int32_t* qorder, const void* pose, const void* w_s,
                                    const void* alive, const void* gs, const void* gb, const int32_t* src_rows, int N, int n, int m_pad, void* gsrc_s, void* slab,
                                    void* gts_far, void* gw_s, void* bwd_partials, int overwrite, void* stream, const int32_t* skip) {
    if (const int e = check_params(prm, c)) return e;
    if (!src_s || !tgt_s || !spos || !spos_ref || !pose || (gw_s && !w_s) || !gs || !gb || !gsrc_s || !bwd_partials || (slab && !gts_far))
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
            (T*)bwd_partials, src_rows, skip); } while (0)
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
int dicp_icp_forward(int dtype, const dicp_weight_params* prm, const dicp_loop_buffers* B, int N, int n, int m,
                     int dim, int const_iter, double tolerance, int k0, int k1, void* stream) {
    if (!prm || !B || !B->src || !B->tgt || !B->poses || !B->deltas || !B->costs || !B->alive ||
        !B->converged || !B->iterations || !B->matched_ratio || !B->n_start || !B->n_matched || !B->w ||
        !B->partials || !B->counters) return DICP_ERR_NULL;
    if (bad_dtype(dtype)) return DICP_ERR_DTYPE;
    if (k0 < 0 || k1 > B->K || k0 > k1 || N <= 0 || n <= 0 || m <= 0 || B->w_stride < n || B->w_iter < n) return DICP_ERR_SHAPE;
    const size_t es = dtype == DICP_F32 ? 4 : 8;
    const int kind = B->knn_variant & 0xff;
    const dicp_gumbel_loop* G = kind == DICP_KNN_GUMBEL ? B->gumbel : nullptr;
    if (kind == DICP_KNN_GUMBEL) { if (!G || !G->ps_t || !G->nbr || !G->lse || (!G->U && !G->seeds)) return DICP_ERR_NULL; }
    else if ((!B->idx && !B->spos) || (kind == DICP_KNN_SWEEP ? (!B->tperm || !B->bucket || !B->brange) : !B->tgt4)) return DICP_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    const int nblk = dicp_accumulate_blocks(n);
    if (!G && small_loop_eligible(dtype, kind, B->knn_variant, n, B->m_pad) && k1 > k0) {
        // small clouds: the whole chunk is ONE launch, one block per cloud (bit 25 of knn_variant switches this off)
        if (const int e = check_params(prm, B->c)) return e;
        begin_launch();
        const WeightParams P = to_params(prm);
        const size_t lds = (size_t)B->m_pad * (dtype == DICP_F32 ? sizeof(float4) : sizeof(double4));
#define DICP_SMALL(T, M) icp_small_forward_kernel<T, M><<<N, BLOCK, lds, st>>>(P, *B, N, n, m, dim, const_iter, tolerance, k0, k1)
        if (dtype == DICP_F32) { if (P.mode == MODE_PT2PL) DICP_SMALL(float, MODE_PT2PL); else DICP_SMALL(float, MODE_PT2PT); }
        else                   { if (P.mode == MODE_PT2PL) DICP_SMALL(double, MODE_PT2PL); else DICP_SMALL(double, MODE_PT2PT); }
#undef DICP_SMALL
        return launch_status();
    }
    for (int k = k0; k < k1; ++k) {
        const char* pose_k = (const char*)B->poses + (size_t)k * N * 12 * es;
        // the searches read [C | r - centre] when the caller keeps that second pose history (packed rows are then y - centre)
        const char* pose_s = B->poses_search ? (const char*)B->poses_search + (size_t)k * N * 12 * es : pose_k;
        int32_t* idx_k = B->idx ? B->idx + (B->idx_per_iter ? (size_t)k * N * n : 0) : nullptr;
        char* w_k = (char*)B->w + (size_t)k * B->w_iter * es;       // cloud stride B->w_stride: (N,K,n) or (K,N,n) alike
        const char* w_prev_k = k > k0 ? (const char*)B->w + (size_t)(k - 1) * B->w_iter * es : (const char*)B->w_prev0;     // (as make_step_io)
        const char* alive_k = (const char*)B->alive + (size_t)k * N * es;
        if (B->events) {    // the sweep launch carries its two events itself; the brute-force forms are bracketed by records
            if (kind == DICP_KNN_SWEEP) set_launch_events((hipEvent_t)B->events[6 * k + 0], (hipEvent_t)B->events[6 * k + 1]);
            else if (hipEventRecord((hipEvent_t)B->events[6 * k + 0], st) != hipSuccess) return -(int)hipGetLastError();
        }
        int rc;
        if (kind == DICP_KNN_SWEEP) {
            // bits 8..15 of knn_variant optionally pin a tile-sweep launch configuration (0 = chosen from the problem size)
            int cfg = (B->knn_variant >> 8) & 0xff;
            int32_t* spos_k = B->spos ? B->spos + (B->idx_per_iter ? (size_t)k * N * n : 0) : nullptr;
            const bool sorted_rows = B->tgt_sorted && spos_k;      // accumulate gathers 32-byte aligned rows of the sorted copy at the sorted positions
            if (!sorted_rows && !B->idx) { set_launch_events(nullptr, nullptr); return DICP_ERR_NULL; }
            // src_s: the source rows in qorder's slot order (dicp_query_order wrote them): coalesced query loads
            const void* qsrc = B->src;
            if (B->src_s && B->qorder) { qsrc = B->src_s; cfg |= DICP_SWEEP_SRC_SORTED; }
            // match certificates: only the units holding a query whose match is not proven unchanged are searched again
            const int cfg_plain = (cfg & ~DICP_SWEEP_SRC_SORTED) ? (cfg & ~DICP_SWEEP_SRC_SORTED) : sweep_auto_cfg(N, n);
            const bool cert = B->cert_q && B->cert_qu && B->rmax && B->dcum && spos_k && sorted_rows && !B->idx && B->qorder && sweep_queries_per_lane(cfg_plain) > 0;
            const bool fresh = k == 0 || (k == k0 && B->cert_reset);           // a new query order: every query is searched, every budget written
            int32_t* count_k = B->cert_count ? B->cert_count + (size_t)k * 2 * CERT_SHARDS : nullptr;
            const bool searched = B->first_search_done && k == 0 && !cert && spos_k && !B->idx;    // the caller ran iteration 0's search itself, ahead of this call
            if (searched) rc = 0;
            else if (cert) {
                if (!fresh && k == k0 && B->idx_per_iter) {     // this call's first matches start as the previous call's last (later ones: handed on by accumulate)
                    if (!B->spos_prev0) { set_launch_events(nullptr, nullptr); return DICP_ERR_NULL; }
                    if (hipMemcpyAsync(spos_k, B->spos_prev0, (size_t)N * n * sizeof(int32_t), hipMemcpyDeviceToDevice, st) != hipSuccess) {
                        set_launch_events(nullptr, nullptr);        // (thread-local: the next launch on this thread must not carry them)
                        return -(int)hipGetLastError();
                    }
                }
                begin_launch();
                rc = sweep_launch(dtype, qsrc, pose_s, B->tgt4, B->tperm, B->qorder, B->bucket, B->brange, B->nbkt, N, n, m, B->m_pad, nullptr, spos_k,
                                  B->pairs, cfg, Rows{B->src_rows, B->tgt_rows}, st, CertArgs{B->cert_q, B->cert_qu, B->dcum, 2 * (B->K + 1), k, count_k, !fresh, B->cert_cloud, B->cert_set});
            } else
            rc = dicp_knn_sweep(dtype, qsrc, pose_s, B->tgt4, B->tperm, B->qorder, B->bucket, B->brange, B->nbkt, B->src_rows, B->tgt_rows, N, n, m, B->m_pad,
                                B->idx ? idx_k : nullptr, spos_k, B->pairs, cfg, B->tgt_f16, stream);
            set_launch_events(nullptr, nullptr);
            if (rc) return rc;
            if (B->events) set_launch_events((hipEvent_t)B->events[6 * k + 2], (hipEvent_t)B->events[6 * k + 3]);
            if (cert) {
                // the accumulate of a certified iteration checks every point's budget and searches the spent ones on the spot; its matches
                // are the start of the next iteration's (within this call)
                const CertAcc ca{pose_s, B->tgt4, B->tperm, B->bucket, B->brange, B->nbkt, B->tgt_rows, m, B->m_pad, B->pairs,
                                 B->cert_q, B->cert_qu, fresh ? nullptr : B->dcum, 2 * (B->K + 1), k, count_k,
                                 spos_k, (B->idx_per_iter && k + 1 < k1) ? spos_k + (size_t)N * n : nullptr, B->cert_cloud, B->cert_set};
                rc = accumulate_go(dtype, prm, B->src, B->tgt_sorted, B->tgt_sorted_stride, spos_k, pose_k, B->w_init, alive_k, B->src_rows, N, n, B->m_pad,
                                   B->partials, w_k, B->w_stride, stream, &ca, w_prev_k);
            } else if (sorted_rows)
                rc = accumulate_go(dtype, prm, B->src, B->tgt_sorted, B->tgt_sorted_stride, spos_k, pose_k, B->w_init, alive_k, B->src_rows, N, n, B->m_pad,
                                   B->partials, w_k, B->w_stride, stream, nullptr, w_prev_k);
            else
                rc = accumulate_go(dtype, prm, B->src, B->tgt, B->c, idx_k, pose_k, B->w_init, alive_k, B->src_rows, N, n, m, B->partials, w_k, B->w_stride, stream, nullptr, w_prev_k);
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
            rc = accumulate_go(dtype, prm, B->src, nbr_k, B->c, nullptr, pose_k, B->w_init, alive_k, nullptr, N, n, n, B->partials, w_k, B->w_stride, stream, nullptr, w_prev_k);
            set_launch_events(nullptr, nullptr);
            if (rc) return rc;
        } else {
            rc = dicp_knn(dtype, B->src, pose_s, B->tgt4, B->src_rows, B->tgt_rows, N, n, m, B->m_pad, idx_k, B->knn_variant & 0xffff, B->tgt_f16, stream);
            if (rc) return rc;
            if (B->events) {
                if (hipEventRecord((hipEvent_t)B->events[6 * k + 1], st) != hipSuccess) return -(int)hipGetLastError();
                set_launch_events((hipEvent_t)B->events[6 * k + 2], (hipEvent_t)B->events[6 * k + 3]);
            }
            rc = accumulate_go(dtype, prm, B->src, B->tgt, B->c, idx_k, pose_k, B->w_init, alive_k, B->src_rows, N, n, m, B->partials, w_k, B->w_stride, stream, nullptr, w_prev_k);
            set_launch_events(nullptr, nullptr);
            if (rc) return rc;
        }
        dicp_step_io io = make_step_io(*B, k, k0, N, n, prm->mode, dim, const_iter, tolerance, es, nblk);
        io.w_copied = w_prev_k ? 1 : 0;
        rc = dicp_step(dtype, &io, N, stream);
        if (rc) return rc;
    }
    return 0;
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
    if (bad_dtype(dtype)) return DICP_ERR_DTYPE;
    if (S->nseg <= 0 || S->nseg > DICP_MAX_SEGMENTS) return DICP_ERR_SHAPE;
    const size_t es = dtype == DICP_F32 ? 4 : 8;
    dicp_loop_buffers B = *buf;
    for (int s = 0; s < S->nseg; ++s) {
        const int k0 = S->k0[s], k1 = S->k1[s];
        if (k0 < 0 || k1 > B.K || k0 >= k1 || (s > 0 && k0 != S->k1[s - 1])) return DICP_ERR_SHAPE;
        int32_t* qo = S->order[s];
        if (qo && S->new_order[s]) {
            if (!S->keys) return DICP_ERR_NULL;
            const char* pose_s = (const char*)(B.poses_search ? B.poses_search : B.poses) + (size_t)k0 * N * 12 * es;
            if (const int rc = dicp_query_order(dtype, B.src, pose_s, B.brange, B.nbkt, N, n, qo, nullptr, nullptr, nullptr, 0, nullptr, B.m_pad,
                                                S->keys, B.bucket, m, B.src_rows, B.tgt_rows, stream)) return rc;
        }
        B.qorder = qo;
        const bool certs = S->cert_from >= 0 && k0 >= S->cert_from;
        B.cert_q = certs ? S->cert_q : nullptr; B.cert_qu = certs ? S->cert_qu : nullptr; B.cert_count = certs ? S->cert_count : nullptr;
        B.cert_cloud = certs ? S->cert_cloud : nullptr;
        B.cert_set = certs ? S->cert_set : nullptr;
        B.cert_reset = (S->cert_from >= 0 && k0 == S->cert_from) ? 1 : 0;
        B.spos_prev0 = (B.spos && B.idx_per_iter && k0 > 0) ? B.spos + (size_t)(k0 - 1) * N * n : nullptr;
        B.w_prev0 = k0 > 0 ? (const char*)B.w + (size_t)(k0 - 1) * B.w_iter * es : nullptr;
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
    const dicp_gumbel_loop* G = (B && (B->knn_variant & 0xff) == DICP_KNN_GUMBEL) ? B->gumbel : nullptr;
    if (!prm || !B || !B->src || !B->tgt || !B->poses || !B->deltas || !B->areg || !B->alive || (!G && !B->idx && !B->spos) || (gw && !B->w_init) ||
        !gpose || !gpose_tmp || !gs || !gb || !gsrc || !bwd_partials) return DICP_ERR_NULL;
    if ((B->knn_variant & 0xff) == DICP_KNN_GUMBEL && (!G || !G->ps_t || !G->nbr || !G->lse || !G->g_nbr || !G->g_ps || (!G->U && !G->seeds))) return DICP_ERR_NULL;
    if (bad_dtype(dtype)) return DICP_ERR_DTYPE;
    if (k0 < 0 || k1 > B->K || k0 > k1 || !B->idx_per_iter || (B->spos && (B->m_pad <= 0 || !B->spos_ref))) return DICP_ERR_SHAPE;
    if (B->bwd_skip && (!B->bwd_mref || !(B->bwd_skip_eps >= 0.0))) return DICP_ERR_NULL;
    const size_t es = dtype == DICP_F32 ? 4 : 8;
    hipStream_t st = (hipStream_t)stream;
    const int nblk = B->spos ? dicp_window_blocks(dtype, n, B->m_pad) : dicp_accumulate_blocks(n);
    {   // small clouds (atomic form only): the whole chunk is ONE launch, one block per cloud
        const size_t lds = (size_t)m * (prm->mode == DICP_PT2PL ? 6 : 3) * es;
        if (!G && !B->spos && k1 > k0 && B->m_pad > 0 && lds <= 40 * 1024 &&
            small_loop_eligible(dtype, B->knn_variant & 0xff, B->knn_variant, n, B->m_pad)) {
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
    if (B->spos && B->bwd_skip && B->bwd_tail_from > k0) {
        if (k0 != 0) return DICP_ERR_SHAPE;                 // (the launch runs down to iteration 0 and folds the last pose sums into the cotangent)
        if (!B->bwd_tail_partials || !B->bwd_tail_arrive) return DICP_ERR_NULL;
        if (nblk > dicp_bwd_tail_max_blocks(dtype)) return DICP_ERR_SHAPE;      // (its blocks wait for each other: they must all be resident)
        kt = B->bwd_tail_from < k1 ? B->bwd_tail_from : k1;
        if (B->bwd_overwrite && kt >= k1) kt = k1 - 1;      // (the first windowed launch initialises the accumulators: it always runs)
    }
    for (int k = k1 - 1; k >= kt; --k) {
        const char* pose_k = (const char*)B->poses + (size_t)k * N * 12 * es;
        const char* alive_k = (const char*)B->alive + (size_t)k * N * es;
        const SkipHost sh{B->bwd_skip, B->bwd_mref, alive_k, B->bwd_live ? B->bwd_live + k : nullptr, B->bwd_skip_eps, k};
        int rc = step_bwd_go(dtype, gin, have_partials ? bwd_partials : nullptr, nblk, dim, pose_k,
                             (const char*)B->deltas + (size_t)k * 6 * es, (int64_t)B->K * 6, B->areg + (size_t)k * N * 36,
                             gs, gb, gout, N, stream, B->bwd_skip ? sh : SkipHost{});
        if (rc) return rc;
        if (B->events) {
            if (B->spos) set_launch_events((hipEvent_t)B->events[6 * k + 4], (hipEvent_t)B->events[6 * k + 5]);
            else if (hipEventRecord((hipEvent_t)B->events[6 * k + 4], st) != hipSuccess) return -(int)hipGetLastError();
        }
        if (G) {
            // soft correspondences: the neighbour rows carry gradient themselves -- accumulate_bwd leaves it in g_nbr, the soft kNN's
            // adjoint takes it on to the transformed source (g_ps) and to the target (added up over the iterations), and the transform's
            // adjoint to the source and to the pose (its sums join accumulate_bwd's in bwd_partials)
            const char* nbr_k = (const char*)G->nbr + (size_t)k * N * n * B->c * es;
            const char* lse_k = (const char*)G->lse + (size_t)k * N * n * es;
            if (hipMemsetAsync(G->g_nbr, 0, (size_t)N * n * B->c * es, st) != hipSuccess) return -(int)hipGetLastError();
            rc = accumulate_bwd_go(dtype, prm, B->src, nbr_k, B->c, nullptr, pose_k, B->w_init, alive_k, gs, gb, nullptr, N, n, n, gsrc, G->g_nbr, gw, bwd_partials, stream, B->bwd_skip);
            if (!rc) rc = dicp_transform_points(dtype, B->src, pose_k, G->ps_t, N, n, stream);
            if (!rc) rc = gumbel_nn_bwd_go(dtype, G->ps_t, B->tgt, B->c, G->U ? G->U[k] : nullptr, G->seeds ? G->seeds[k] : 0u, G->eps, G->tau, nbr_k, lse_k, G->g_nbr,
                                           N, n, m, G->g_ps, gtgt, 1, stream);
            if (!rc) rc = transform_points_bwd_go(dtype, B->src, pose_k, G->g_ps, gsrc, bwd_partials, N, n, 1, stream);
        } else if (B->spos)    // windowed form: src / w_init / tgt are the SORTED copies, gsrc / gw accumulate in slot order, gtgt is the slab
            rc = accumulate_bwd_window_go(dtype, prm, B->src, B->tgt, B->c, B->spos + (size_t)k * N * n, B->spos_ref, B->qorder, pose_k, B->w_init,
                                          alive_k, gs, gb, B->src_rows, N, n, B->m_pad,
                                          gsrc, gtgt, B->gts_far, gw, bwd_partials, (B->bwd_overwrite && k == k1 - 1) ? 1 : 0, stream, B->bwd_skip);
        else
            rc = accumulate_bwd_go(dtype, prm, B->src, B->tgt, B->c, B->idx + (size_t)k * N * n, pose_k, B->w_init,
                                   alive_k, gs, gb, B->src_rows, N, n, m, gsrc, gtgt, gw, bwd_partials, stream, B->bwd_skip);
        set_launch_events(nullptr, nullptr);
        if (rc) return rc;
        if (B->events && !B->spos) { if (hipEventRecord((hipEvent_t)B->events[6 * k + 5], st) != hipSuccess) return -(int)hipGetLastError(); }
        have_partials = 1;
        double* t = gin; gin = gout; gout = t;
    }
    if (kt > k0) {
        if (const int e = check_params(prm, B->c)) return e;
        begin_launch();
        const WeightParams P = to_params(prm);
        double* dst = ((k1 - k0) & 1) ? gpose_tmp : gpose;            // where the alternating buffers would have left it
        const unsigned g = grid_for(N, nblk);
#define DICP_TAILB(T, M) do { constexpr int WT = WindowRows<T>::v; \
        bwd_tail_kernel<T, M, WT><<<g, BLOCK, 0, st>>>(P, *B, N, n, dim, window_slots(WT, n, B->m_pad), nblk, gin, dst, have_partials, \
            (T*)gsrc, (T*)gtgt, (T*)gw, (T*)bwd_partials, (T*)B->bwd_tail_partials, B->bwd_tail_arrive, kt); } while (0)
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
