// The search on the matrix cores (round 4): split-f16 FILTER on v_mfma_f32_32x32x16_f16 + exact float32 REFINE.
//
// The reference's hot line is the K = 5 contraction inside torch.cdist (/root/reference/dICP/nn.py:32): score(x, y) = 0.5|y|^2 - x.y.
// gfx950's f32 MFMA runs at the vector rate (round 1: it never paid); its f16 MFMA is 16x faster and accumulates in f32.  So:
//   FILTER  x and y are split into two f16 terms each (x = xh + xl to 2^-22, after a per-cloud power-of-two scale s that puts the
//           cloud's extent at 2^10..2^11), and ONE 32x32x16 MFMA scores 32 targets x 32 queries with the K = 16 slots
//               A (target):  yh0 yh1 yh2 | yl0 yl1 yl2 | yh0 yh1 yh2 | yl0 yl1 yl2 | H0 H1 H2 | 0        H = three-term split of h s^2 / P
//               B (query):   xh0 xh1 xh2 | xh0 xh1 xh2 | xl0 xl1 xl2 | xl0 xl1 xl2 | P  P  P  | 0        x = -(C p + r) s,  P = 2^13
//           = s^2 (x.y + h) up to the error bound E below.  Lane l of the result holds 16 targets of ONE query (column l & 31), so the
//           running minimum is lane-local: per chunk of F16_CHT MFMAs a lane reduces its 16 F16_CHT scores (v_min3), and keeps the two
//           smallest chunk minima b1 <= b2 and the chunk of b1 (v_med3, compare, select, min).
//   REFINE  |filter - score()| <= E for every pair (E per query, below), so the true winner j* -- the argmin of the float32 score() every
//           other search form computes, lowest index among equals -- lies in the chunk of b1 unless another chunk's minimum is within
//           2E of it:  b2 - b1 > 2E  =>  re-score that one chunk (16 F16_CHT rows) with score() itself, ascending, strict <: bit-identical
//           to knn_valu_kernel.  Otherwise (a near-tie inside the filter's resolution, a query outside the f16 range, nothing finite)
//           the wave scores every row of the cloud for that query exactly, 64 rows at a time.
// The bound (u = 2^-24; T = sum_i |x_i y_i| + h <= |x| |y| + 0.5 |y|^2, unscaled):
//     |score() - (h - x.y)|          <= 3.01 u T                  three fmas on the stored h
//     split:  |x^ y^ - x y|          <= 2 (4u) |x_i y_i| + phi (|x|_1 + |y|_1),  phi = 2^-24 / s  (f16 denormals are honoured: measured,
//                                                                 scripts/ubench/mfma_f16_ubench.hip; the term covers their rounding)
//     h split (three terms)          <= u h + hphi,               hphi = 3 * 2^-12 / s^2
//     MFMA accumulation              <= 20 u T                    measured 4.97 u T at worst over 4e5 dot products of wide range and
//                                                                 heavy cancellation (same ubench); 4x that is assumed -- and the WHOLE bound was then
//                                                                 searched for its worst case: knn_f16_probe_kernel below scores every pair of adversarial
//                                                                 clouds (queries on targets km from the origin, f16-denormal low terms, extents at the
//                                                                 scale's boundaries, rows at 16x the extent, queries at the edge of the f16 range,
//                                                                 cancelling products) as the searches do: max |filter - score()| / E = 0.34 over 5.6e9
//                                                                 pairs (profiles/r05_knn_f16_bound_search.txt; tests/test_gpu_f16.py holds <= 0.5)
//   E = 32 u T + phi (|x|_1 + sqrt3 |y|) + hphi, with |y| <= |x| + sqrt(2 D) and D the candidate's half squared distance, itself bounded
//   from the filter's own minimum (D <= b1 / s^2 + 0.5|x|^2, taken with 2^-9 relative slack, which the code checks E against).
// Rows the f16 range cannot hold next to the rest of the cloud (the reference's pad rows at max(source) * 1000, ICP.py:460: a cloud with up
// to 64 rows beyond 16x the rest's extent) are left out of the image and scored exactly for every query at the end ("far rows").
#include <stdlib.h>
#include <type_traits>

#include "dicp_common.h"
#include "dicp_internal.h"
#include "dicp_fill.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int F16_META = dicp_tu::KNN_F16_META;
constexpr int F16_STAGE = dicp_tu::KNN_F16_STAGE_ROWS / 32;   // A tiles (32 rows, 1 KiB) per LDS stage
constexpr int F16_G = 4;                // B tiles (32 queries each) per wave: 128 queries, like a unit of the sweep
constexpr int F16_CHT = 4;              // A tiles per chunk of the lane-local bookkeeping
constexpr int F16_FAR_MAX = 64;
constexpr int F16_NTC = 8;              // sweep form: 64-row tiles whose float32 rows a wave keeps in LDS for its refine
#ifndef DICP_F16_SCAN_MAX
#define DICP_F16_SCAN_MAX 6
#endif
constexpr int F16_SCAN_MAX = DICP_F16_SCAN_MAX;         // sweep form: up to this many queries of a wave with three or more candidate pieces are scanned exactly instead of filtered again
enum { FM_S = 0, FM_INV_S2 = 1, FM_PHI = 2, FM_NFAR = 3, FM_HPHI = 4, FM_FAR_ABOVE = 5, FM_AGAIN = 6 /* queries sent through pass 2, added up */,
       FM_SCAN = 7 /* queries scored against every row */, FM_FAR0 = 8 };
constexpr float F16_P = 8192.f;
constexpr float F16_U = 5.9604644775390625e-8f;     // 2^-24
constexpr float F16_CREL = 32.f;

// row (inside its 32-row tile) held by MFMA row rho of the A operand: half h of the RESULT's lanes then owns rows 16h .. 16h + 15, in
// register order (result register i of lane half h is MFMA row (i & 3) + 8 (i >> 2) + 4 h)
__device__ __forceinline__ int f16_row_of(int rho) { return ((rho >> 2) & 1) * 16 + (rho >> 3) * 4 + (rho & 3); }

constexpr int F16_SCALE_THREADS = 1024;      // one block per cloud: three passes over its rows, 16 rows per thread and pass at 16384
__device__ __forceinline__ float block_max(float v, float* red) {
#pragma unroll
    for (int o = WAVE / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    __syncthreads();
    if ((threadIdx.x & (WAVE - 1)) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = red[0];
    for (int w = 1; w < F16_SCALE_THREADS / WAVE; ++w) r = fmaxf(r, red[w]);
    return r;
}

// per cloud: the scale and the far rows.  One block per cloud.
// Which clouds a launch concerns (the loop's per-cloud choice of the scoring form, dicp_loop_buffers.search.form): those whose last plain search tallied more than
// `tiles` 64-row tiles per unit of 128 queries (no tally: if dflt).
struct FormPick {
    const int32_t* tally; const int32_t* src_rows; int n_full, tiles, dflt;
};
__device__ __forceinline__ bool form_long(const FormPick& f, int cloud) {
    if (!f.tally) return true;
    const int t = f.tally[cloud], units = (rows_of(f.src_rows, cloud, f.n_full) + 32 * F16_G - 1) / (32 * F16_G);
    return t > 0 ? t > f.tiles * units : f.dflt != 0;
}

__global__ __launch_bounds__(F16_SCALE_THREADS) void knn_f16_scale_kernel(const float4* __restrict__ rows4, const int32_t* __restrict__ tgt_rows, int m_full, int m_pad,
                                                              float* __restrict__ meta_all) {
    __shared__ float red[F16_SCALE_THREADS / WAVE];
    __shared__ int s_cnt;
    const int cloud = blockIdx.x, tid = threadIdx.x;
    const int m = min(max(rows_of(tgt_rows, cloud, m_full), 1), m_pad);
    const float4* __restrict__ tg = rows4 + (size_t)cloud * m_pad;
    float* mt = meta_all + (size_t)cloud * F16_META;
    int32_t* mi = (int32_t*)mt;
    auto row_max = [&](int r) {
        const float4 v = tg[r];
        return (v.w < __builtin_huge_valf()) ? fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fabsf(v.z)) : -1.f;     // (0.5|y|^2 finite => the row is)
    };
    // a row that repeats the row before it bit for bit can never be the answer (lowest index among equal scores): the reference's pad rows --
    // copies of ONE far point, ICP.py:460,472-477 -- count as one far row
    auto repeats = [&](int r) {
        if (r == 0) return false;
        const float4 v = tg[r], p = tg[r - 1];
        return __float_as_uint(v.x) == __float_as_uint(p.x) && __float_as_uint(v.y) == __float_as_uint(p.y) && __float_as_uint(v.z) == __float_as_uint(p.z) &&
               __float_as_uint(v.w) == __float_as_uint(p.w);
    };
    float mx = 0.f;
    for (int r = tid; r < m; r += F16_SCALE_THREADS) mx = fmaxf(mx, row_max(r));
    const float Minf = block_max(mx, red);
    // the extent WITHOUT the rows beyond Minf / 16, and how many those are
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    const float cut = Minf * 0.0625f;
    float m2 = 0.f;
    int above = 0;
    for (int r = tid; r < m; r += F16_SCALE_THREADS) { const float v = row_max(r); if (v > cut) above += repeats(r) ? 0 : 1; else m2 = fmaxf(m2, v); }
    if (above) atomicAdd(&s_cnt, above);
    const float M2 = block_max(m2, red);
    const int cnt = s_cnt;
    __syncthreads();
    const bool use_far = cnt > 0 && cnt <= F16_FAR_MAX && M2 > 0.f && cnt * 64 <= m;
    const float M = use_far ? M2 : Minf;
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    if (use_far)
        for (int r = tid; r < m; r += F16_SCALE_THREADS)
            if (row_max(r) > cut && !repeats(r)) { const int slot = atomicAdd(&s_cnt, 1); if (slot < F16_FAR_MAX) mi[FM_FAR0 + slot] = r; }
    __syncthreads();
    if (tid == 0) {
        int ex = 0;
        float s = 1.f;
        if (M > 0.f && M < __builtin_huge_valf()) {
            (void)frexpf(M, &ex);                                   // M = f 2^ex, f in [0.5, 1): M 2^(11 - ex) in [2^10, 2^11): queries up to 32x
            const int e = min(max(11 - ex, -60), 60);               // the cloud's extent away still fit the f16 range
            s = ldexpf(1.f, e);
        }
        mt[FM_S] = s;
        mt[FM_INV_S2] = (1.f / s) * (1.f / s);
        mt[FM_PHI] = F16_U / s;
        mi[FM_NFAR] = use_far ? min(s_cnt, F16_FAR_MAX) : 0;
        mt[FM_HPHI] = 3.f * 0.000244140625f * ((1.f / s) * (1.f / s));      // 3 * 2^-12 / s^2
        mt[FM_FAR_ABOVE] = use_far ? cut : __builtin_huge_valf();
        mi[FM_AGAIN] = mi[FM_SCAN] = 0;
    }
}

// the image: one thread per (row, k half) -> 16 bytes, written in MFMA operand order (a wave's load of a tile is one contiguous KiB)
__global__ __launch_bounds__(BLOCK) void knn_f16_image_kernel(const float4* __restrict__ rows4, const int32_t* __restrict__ tgt_rows, int m_full, int m_pad, int m_img,
                                                              const float* __restrict__ meta_all, uint4* __restrict__ image, int tiles_per_cloud, int N,
                                                              float* __restrict__ edges /* (N, tiles_per_cloud / 2, 2) */) {
    const int cloud = blockIdx.y;
    const int tile = blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6), lane = threadIdx.x & (WAVE - 1);
    if (tile >= tiles_per_cloud) return;
    const int m = min(max(rows_of(tgt_rows, cloud, m_full), 1), m_pad);
    const float* mt = meta_all + (size_t)cloud * F16_META;
    const float s = mt[FM_S], far_above = mt[FM_FAR_ABOVE];
    const int r = tile * 32 + f16_row_of(lane & 31), kh = lane >> 5;
    // x of the first / last row of the 64-row tile (as the packed rows hold it: pad rows of a sorted batch carry the largest key)
    if (kh == 0 && r < m_pad && (r & 63) == 0) edges[((size_t)cloud * (tiles_per_cloud / 2) + (r >> 6)) * 2] = rows4[(size_t)cloud * m_pad + r].x;
    if (kh == 0 && r < m_pad && (r & 63) == 63) edges[((size_t)cloud * (tiles_per_cloud / 2) + (r >> 6)) * 2 + 1] = rows4[(size_t)cloud * m_pad + r].x;
    _Float16 yh[3] = {0, 0, 0}, yl[3] = {0, 0, 0}, H[3] = {(_Float16)__builtin_huge_valf(), 0, 0};
    if (r < m) {
        const float4 v = rows4[(size_t)cloud * m_pad + r];
        const float big = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fabsf(v.z));
        if (v.w < __builtin_huge_valf() && !(big > far_above)) {
            const float c[3] = {v.x * s, v.y * s, v.z * s};
#pragma unroll
            for (int k = 0; k < 3; ++k) { yh[k] = (_Float16)c[k]; yl[k] = (_Float16)(c[k] - (float)yh[k]); }
            const float hs = v.w * (s * s * (1.f / F16_P));
            H[0] = (_Float16)hs;
            const float r1 = hs - (float)H[0];
            H[1] = (_Float16)r1;
            H[2] = (_Float16)(r1 - (float)H[1]);
        }
    }
    half8 a;
    if (kh == 0) { a[0] = yh[0]; a[1] = yh[1]; a[2] = yh[2]; a[3] = yl[0]; a[4] = yl[1]; a[5] = yl[2]; a[6] = yh[0]; a[7] = yh[1]; }
    else         { a[0] = yh[2]; a[1] = yl[0]; a[2] = yl[1]; a[3] = yl[2]; a[4] = H[0];  a[5] = H[1];  a[6] = H[2];  a[7] = (_Float16)0.f; }
    uint4 out;
    __builtin_memcpy(&out, &a, 16);
    image[((size_t)cloud * tiles_per_cloud + tile) * 64 + lane] = out;
}

// the query's B fragment: lane half 0 holds k = 0..7, half 1 k = 8..15.  ok = false: the scaled query does not fit the f16 range (further than
// ~30x the cloud's extent from its centre, or not finite): its fragment is all zeros (finite filter values that mean nothing) and the caller
// must take the exact scan for it
__device__ __forceinline__ half8 f16_query_fragment(const float* nx, float s, int kh, bool& ok) {
    _Float16 xh[3], xl[3];
    ok = true;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float c = nx[k] * s;
        if (!(fabsf(c) <= 60000.f)) { ok = false; c = 0.f; }
        xh[k] = (_Float16)c; xl[k] = (_Float16)(c - (float)xh[k]);
    }
    if (!ok) { xh[0] = xh[1] = xh[2] = xl[0] = xl[1] = xl[2] = (_Float16)0.f; }
    half8 b;
    const _Float16 P = (_Float16)F16_P;
    if (kh == 0) { b[0] = xh[0]; b[1] = xh[1]; b[2] = xh[2]; b[3] = xh[0]; b[4] = xh[1]; b[5] = xh[2]; b[6] = xl[0]; b[7] = xl[1]; }
    else         { b[0] = xl[2]; b[1] = xl[0]; b[2] = xl[1]; b[3] = xl[2]; b[4] = P;     b[5] = P;     b[6] = P;     b[7] = (_Float16)0.f; }
    return b;
}

// an upper bound of sqrt(v), v >= 0: the hardware's v_sqrt_f32 (1 ulp) with room to spare (the correctly rounded sqrtf is a dozen instructions, and
// the bounds that use it are taken a per cent above what is needed anyway); tiny arguments get an absolute 1e-18
__device__ __forceinline__ float f16_sqrt_up(float v) { return __builtin_amdgcn_sqrtf(v) * 1.00001f + 1e-18f; }

// 2E in the filter's own (scaled) units for a query whose smallest filter value is b1 (scaled); < 0: no bound (exact scan)
__device__ __forceinline__ float f16_margin(const float* nx, float b1, float s, float inv_s2, float phi, float hphi) {
    const float x1 = fabsf(nx[0]) + fabsf(nx[1]) + fabsf(nx[2]);
    const float hx = 0.5f * (nx[0] * nx[0] + nx[1] * nx[1] + nx[2] * nx[2]);
    const float b1u = b1 * inv_s2;
    const float D0 = fmaxf(b1u + hx, 0.f);
    const float slack = 0.001953125f * (D0 + hx);                   // 2^-9 (D0 + hx)
    const float Dup = D0 + slack;
    const float X2 = f16_sqrt_up(2.f * hx);
    const float Y = X2 + f16_sqrt_up(2.f * Dup);
    const float Tup = X2 * Y + 0.5f * Y * Y;
    const float E = (F16_CREL * F16_U * Tup + phi * (x1 + 1.7321f * Y) + hphi) * 1.01f;
    if (!(4.f * E <= slack) || !(E < __builtin_huge_valf())) return -1.f;      // (also NaN / inf: a query outside the f16 range)
    return 2.f * E * (s * s) * 1.0001f;
}
__device__ __forceinline__ float f16_margin(const float* nx, float b1, const float* __restrict__ mt) {
    return f16_margin(nx, b1, mt[FM_S], mt[FM_INV_S2], mt[FM_PHI], mt[FM_HPHI]);
}

// the value the lane 32 away holds (a query's two lanes are l and l ^ 32): v_permlane32_swap exchanges the upper half of one register with the lower
// half of another inside the vector unit -- the generic shuffle goes through the LDS crossbar (ds_bpermute), a hundred cycles of latency each time
__device__ __forceinline__ unsigned swap32(unsigned v) {
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);      // r[0] = {lower, lower}, r[1] = {upper, upper} of v
    return (threadIdx.x & 32) ? r[0] : r[1];
}
__device__ __forceinline__ float swap32(float v) { return __uint_as_float(swap32(__float_as_uint(v))); }
__device__ __forceinline__ int swap32(int v) { return (int)swap32((unsigned)v); }

// v_min / v_max on values that are never signalling NaNs (the compiler's fminf / fmaxf canonicalise both inputs first: two more instructions each)
__device__ __forceinline__ float vmin(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

__device__ __forceinline__ float f16_chunk_min(const f32x16& d, float cm) {
#pragma unroll
    for (int i = 0; i < 16; i += 2) cm = __builtin_fminf(__builtin_fminf(cm, d[i]), d[i + 1]);
    return cm;
}

// The lane-local bookkeeping per chunk: the three smallest chunk minima b1 <= b2 <= b3 and the chunks of the first two.
struct F16Track {
    float b1, b2, b3;
    int id1, id2;
    __device__ __forceinline__ void init() { b1 = b2 = b3 = __builtin_huge_valf(); id1 = id2 = 0; }
    // (written as instructions: the compiler turned the nested selects into divergent branches, which also kept it from interleaving the
    //  MFMAs with the minima of the previous results)
    __device__ __forceinline__ void add(float cm, int chunk /* in a vector register */) {
        asm("v_cmp_lt_f32 vcc, %5, %1\n\t"            // cm < b2
            "v_cndmask_b32 %4, %4, %6, vcc\n\t"       //   id2 = chunk
            "v_cmp_lt_f32 vcc, %5, %0\n\t"            // cm < b1
            "v_cndmask_b32 %4, %4, %3, vcc\n\t"       //   id2 = id1
            "v_cndmask_b32 %3, %3, %6, vcc\n\t"       //   id1 = chunk
            "v_med3_f32 %2, %1, %5, %2\n\t"           // b3 = cm clamped into [b2, b3]: the third smallest of the four
            "v_med3_f32 %1, %0, %5, %1\n\t"           // b2 likewise
            "v_min_f32 %0, %0, %5"
            : "+v"(b1), "+v"(b2), "+v"(b3), "+v"(id1), "+v"(id2) : "v"(cm), "v"(chunk) : "vcc");
    }
};

// stage st of the cloud's image -> registers (issued early), registers -> LDS (after the stage in LDS has been used)
// (four named registers, not an array: the array form was left in scratch memory -- 64 bytes per lane written and read back per stage, 2 GB of HBM writes per
//  launch at 256 x 16384 x 16384, and an s_waitcnt right behind the loads it was meant to hide; profiles/r04_pmc_hbm_traffic.json at 8ee49b5 shows it)
static_assert(F16_STAGE * 64 / BLOCK == 4, "DICP_F16_FETCH / _COMMIT move four 16-byte pieces per thread");
#define DICP_F16_FETCH(st_)  { const uint4* f_ = img + (size_t)(st_) * (F16_STAGE * 64) + tid; pre0 = f_[0]; pre1 = f_[BLOCK]; pre2 = f_[2 * BLOCK]; pre3 = f_[3 * BLOCK]; }
#define DICP_F16_COMMIT(buf_) { uint4* c_ = &lds[buf_][tid]; c_[0] = pre0; c_[BLOCK] = pre1; c_[2 * BLOCK] = pre2; c_[3 * BLOCK] = pre3; }

// every n x m pair.  Block = 4 waves x 128 queries; the cloud's image streams through LDS in stages of 512 rows, double-buffered.
//   pass 1   the filter over all tiles (four B tiles per wave), lane-local bookkeeping per chunk of F16_CHT tiles;
//   refine   b2 - b1 > 2E: the winner's chunk is re-scored exactly; b3 - b1 > 2E: the two candidate chunks are;
//   pass 2   queries with three or more chunks inside 2E (dense surfaces: a fifth of the queries of a planar scene have a second candidate
//            there, a few per cent a third): the block streams the image once more, each wave with ONE B tile made of up to 32 such queries,
//            and every lane re-scores exactly the 16-row pieces whose filter minimum is within 2E of the query's b1 -- a quarter of pass 1's
//            matrix work per round of 32 queries per wave, whatever the number of candidates;
//   scan     queries the filter has no bound for (outside the f16 range, nothing finite): the wave scores every row, 64 at a time.
template <int MINW>
__global__ __launch_bounds__(BLOCK, MINW) void knn_f16_kernel(const float* __restrict__ src, const float* __restrict__ pose, const float4* __restrict__ tgt4,
                                                              const uint4* __restrict__ image, float* __restrict__ meta_all, int32_t* __restrict__ idx,
                                                              int N, int n_full, int m_full, int m_pad_full, int tiles_per_cloud, int bpc,
                                                              const int32_t* __restrict__ src_rows, const int32_t* __restrict__ tgt_rows) {
    __shared__ uint4 lds[2][F16_STAGE * 64];
    __shared__ float4 qlist[BLOCK / WAVE][32];          // pass 2: (query, threshold) of the wave's B tile
    __shared__ int qslot[BLOCK / WAVE][32];             // ... and where its result goes: (g << 8) | column
    __shared__ int s_rounds;
    int cloud, blk;
    if (!decode_block(bpc, N, cloud, blk)) return;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid >> 6;
    const int n = rows_of(src_rows, cloud, n_full), m = min(max(rows_of(tgt_rows, cloud, m_full), 1), m_pad_full);
    if (blk * (BLOCK / WAVE) * (32 * F16_G) >= n) return;           // (block-uniform)
    const int col = lane & 31, kh = lane >> 5;
    const int qwave = (blk * (BLOCK / WAVE) + wave) * (32 * F16_G);
    float* __restrict__ mt = meta_all + (size_t)cloud * F16_META;
    const float s = mt[FM_S];
    float C[9], r[3];
    load_pose(pose, cloud, C, r);
    if (tid == 0) s_rounds = 0;

    float nx[F16_G][3];
    F16Track tr[F16_G];
    half8 b[F16_G];
    int q_ok = 0;                                                   // bit g: the query fits the filter's range
#pragma unroll
    for (int g = 0; g < F16_G; ++g) {
        const int i = qwave + g * 32 + col;
        float p[3] = {0.f, 0.f, 0.f};
        if (i < n) {
            const float* sp = src + ((size_t)cloud * n_full + i) * 3;
            p[0] = sp[0]; p[1] = sp[1]; p[2] = sp[2];
        }
        query_point(C, r, p, nx[g]);                                // ICP.py:137
        bool ok;
        b[g] = f16_query_fragment(nx[g], s, kh, ok);
        q_ok |= ok ? (1 << g) : 0;
        tr[g].init();
    }

    const uint4* __restrict__ img = image + (size_t)cloud * tiles_per_cloud * 64;
    const int nst = (m + 32 * F16_STAGE - 1) / (32 * F16_STAGE);      // stages that hold a real row
    f32x16 zero;
#pragma unroll
    for (int i = 0; i < 16; ++i) zero[i] = 0.f;
    uint4 pre0, pre1, pre2, pre3;
    DICP_F16_FETCH(0)
    DICP_F16_COMMIT(0)
    __syncthreads();
    // ---- pass 1
    for (int st = 0; st < nst; ++st) {
        const int cur = st & 1;
        if (st + 1 < nst) { DICP_F16_FETCH(st + 1) }
#pragma unroll 1
        for (int t = 0; t < F16_STAGE; t += F16_CHT) {
            half8 a[F16_CHT];
#pragma unroll
            for (int c = 0; c < F16_CHT; ++c) { const uint4 v = lds[cur][(t + c) * 64 + lane]; __builtin_memcpy(&a[c], &v, 16); }
            int chunk = st * (F16_STAGE / F16_CHT) + t / F16_CHT;
            asm("v_mov_b32 %0, %1" : "=v"(chunk) : "s"(chunk));       // (a vector register: v_cndmask's second source)
#pragma unroll
            for (int g = 0; g < F16_G; ++g) {
                float cm = __builtin_huge_valf();
#pragma unroll
                for (int c = 0; c < F16_CHT; ++c)
                    cm = f16_chunk_min(__builtin_amdgcn_mfma_f32_32x32x16_f16(a[c], b[g], zero, 0, 0, 0), cm);
                tr[g].add(cm, chunk);
            }
        }
        if (st + 1 < nst) { DICP_F16_COMMIT(cur ^ 1) }
        __syncthreads();
    }

    // ---- refine
    const float4* __restrict__ tg = tgt4 + (size_t)cloud * m_pad_full;
    const int nfar = ((const int32_t*)mt)[FM_NFAR];
    float bv[F16_G], thr[F16_G];
    int bj[F16_G];
    int n_scan = 0, n_again = 0, my_unsure = 0;          // (my_unsure: bit g = this query goes to pass 2)
    auto rescore = [&](const float* q, int r0, int cnt, float& v, int& j) {      // rows r0 .. r0 + cnt - 1, ascending
#pragma unroll 4
        for (int k = 0; k < cnt; ++k) {
            const int rr = r0 + k;
            if (rr < m) {
                const float sc = score<float, float4>(q, tg[rr]);
                if (sc < v || (sc == v && rr < j)) { v = sc; j = rr; }
            }
        }
    };
#pragma unroll
    for (int g = 0; g < F16_G; ++g) {
        const int i = qwave + g * 32 + col;
        // the query's two lanes (the two halves of every tile) merge their bookkeeping: the three smallest of the six values, the chunks
        // (and lane halves) of the first two
        const F16Track me = tr[g];
        F16Track ot;
        ot.b1 = swap32(me.b1); ot.b2 = swap32(me.b2); ot.b3 = swap32(me.b3);
        ot.id1 = swap32(me.id1); ot.id2 = swap32(me.id2);
        // (lo, hi): lo holds the smaller b1 (lane half 0 on equal values: any consistent choice -- equal values are a near-tie anyway)
        const bool me_lo = me.b1 < ot.b1 || (me.b1 == ot.b1 && kh == 0);
        const F16Track lo = me_lo ? me : ot, hi = me_lo ? ot : me;
        const int lo_h = me_lo ? kh : (kh ^ 1), hi_h = lo_h ^ 1;
        const float B1 = lo.b1;
        const bool second_lo = lo.b2 < hi.b1;                       // the second smallest: lo's second or hi's first
        const float B2 = second_lo ? lo.b2 : hi.b1;
        const int c1 = lo.id1, h1 = lo_h, c2 = second_lo ? lo.id2 : hi.id1, h2 = second_lo ? lo_h : hi_h;
        const float B3 = second_lo ? __builtin_fminf(lo.b3, hi.b1) : __builtin_fminf(lo.b2, hi.b2);
        const float margin = f16_margin(nx[g], B1, mt);
        const bool bounded = ((q_ok >> g) & 1) && (B1 < __builtin_huge_valf()) && (B1 > -__builtin_huge_valf()) && margin >= 0.f;
        const int ncand = !bounded ? 0 : (B2 - B1 > margin ? 1 : (B3 - B1 > margin ? 2 : 3));
        bv[g] = __builtin_huge_valf();
        bj[g] = 0;
        thr[g] = B1 + margin;
        if (ncand == 1) {           // the winner's 16 F16_CHT rows by the query's two lanes (tiles 0..1 / 2..3 of the chunk)
            for (int c = kh * (F16_CHT / 2); c < (kh + 1) * (F16_CHT / 2); ++c) rescore(nx[g], (c1 * F16_CHT + c) * 32 + h1 * 16, 16, bv[g], bj[g]);
        } else if (ncand == 2) {    // one candidate chunk per lane
            const int cc = kh ? c2 : c1, hh = kh ? h2 : h1;
            for (int c = 0; c < F16_CHT; ++c) rescore(nx[g], (cc * F16_CHT + c) * 32 + hh * 16, 16, bv[g], bj[g]);
        }
        if (ncand == 1 || ncand == 2) {
            const float ov = swap32(bv[g]);
            const int oj = swap32(bj[g]);
            if (ov < bv[g] || (ov == bv[g] && oj < bj[g])) { bv[g] = ov; bj[g] = oj; }
        }
        if (ncand == 3 && i < n) my_unsure |= 1 << g;
        // queries the filter has no bound for: the wave scores every row for them, one query at a time (lane half 0 asks)
        unsigned long long need = __ballot(ncand == 0 && kh == 0 && i < n);
        while (need) {
            const int L = __builtin_ctzll(need);
            need &= need - 1;
            const float q[3] = {__shfl(nx[g][0], L), __shfl(nx[g][1], L), __shfl(nx[g][2], L)};
            float v = __builtin_huge_valf();
            int j = 0x7fffffff;
            for (int rr = lane; rr < m; rr += WAVE) {
                const float sc = score<float, float4>(q, tg[rr]);
                if (sc < v) { v = sc; j = rr; }
            }
#pragma unroll
            for (int o = WAVE / 2; o > 0; o >>= 1) {
                const float ov = __shfl_xor(v, o);
                const int oj = __shfl_xor(j, o);
                if (ov < v || (ov == v && oj < j)) { v = ov; j = oj; }
            }
            if (col == (L & 31)) { bv[g] = v; bj[g] = (j == 0x7fffffff) ? 0 : j; }
            ++n_scan;
        }
    }

    // ---- pass 2: rounds of up to 32 queries per wave
    int cnt = 0, base[F16_G];        // this wave's pass-2 queries: base[g] = how many precede B tile g's
#pragma unroll
    for (int g = 0; g < F16_G; ++g) { base[g] = cnt; cnt += __popcll(__ballot(((my_unsure >> g) & 1) && kh == 0)); }
    n_again = cnt;
    if (lane == 0 && cnt) atomicMax(&s_rounds, (cnt + 31) / 32);
    __syncthreads();
    const int rounds = s_rounds;
    for (int rd = 0; rd < rounds; ++rd) {
        // the wave's B tile of this round: its pass-2 queries number 32 rd .. 32 rd + 31
#pragma unroll
        for (int g = 0; g < F16_G; ++g) {
            const bool mineg = ((my_unsure >> g) & 1) && kh == 0;
            const unsigned long long mask = __ballot(mineg);
            const int rank = base[g] + __popcll(mask & ((1ull << lane) - 1)) - 32 * rd;
            if (mineg && rank >= 0 && rank < 32) {
                qlist[wave][rank] = make_float4(nx[g][0], nx[g][1], nx[g][2], thr[g]);
                qslot[wave][rank] = (g << 8) | col;
            }
        }
        const int have = min(max(cnt - 32 * rd, 0), 32);
        __builtin_amdgcn_wave_barrier();
        float q[3] = {0.f, 0.f, 0.f}, tau = -__builtin_huge_valf();
        if (col < have) { const float4 e = qlist[wave][col]; q[0] = e.x; q[1] = e.y; q[2] = e.z; tau = e.w; }
        bool ok2;
        const half8 b2 = f16_query_fragment(q, s, kh, ok2);          // (pass-2 queries are bounded ones: they fit)
        float v2 = __builtin_huge_valf();
        int j2 = 0;
        DICP_F16_FETCH(0)
        DICP_F16_COMMIT(0)
        __syncthreads();
        for (int st = 0; st < nst; ++st) {
            const int cur = st & 1;
            if (st + 1 < nst) { DICP_F16_FETCH(st + 1) }
            if (have) {
#pragma unroll 2
                for (int t = 0; t < F16_STAGE; ++t) {
                    half8 a;
                    { const uint4 v = lds[cur][t * 64 + lane]; __builtin_memcpy(&a, &v, 16); }
                    const float cm = f16_chunk_min(__builtin_amdgcn_mfma_f32_32x32x16_f16(a, b2, zero, 0, 0, 0), __builtin_huge_valf());
                    if (cm <= tau) rescore(q, (st * F16_STAGE + t) * 32 + kh * 16, 16, v2, j2);       // (rare: this lane's 16 rows of the tile, exactly)
                }
            }
            if (st + 1 < nst) { DICP_F16_COMMIT(cur ^ 1) }
            __syncthreads();
        }
        {   // the query's two lanes merge; the result goes back to the lanes that own the query
            const float ov = swap32(v2);
            const int oj = swap32(j2);
            if (ov < v2 || (ov == v2 && oj < j2)) { v2 = ov; j2 = oj; }
        }
        for (int e = 0; e < have; ++e) {
            const int slot = qslot[wave][e];                         // (wave-uniform)
            const float rv = __shfl(v2, e);
            const int rj = __shfl(j2, e);
#pragma unroll
            for (int g = 0; g < F16_G; ++g)
                if ((slot >> 8) == g && col == (slot & 0xff)) { bv[g] = rv; bj[g] = rj; }
        }
        __builtin_amdgcn_wave_barrier();
    }

    // ---- far rows (left out of the image) are scored exactly for every query; results
#pragma unroll
    for (int g = 0; g < F16_G; ++g) {
        const int i = qwave + g * 32 + col;
        for (int f = 0; f < nfar; ++f) {
            const int rr = ((const int32_t*)mt)[FM_FAR0 + f];
            if (rr < m) {
                const float sc = score<float, float4>(nx[g], tg[rr]);
                if (sc < bv[g] || (sc == bv[g] && rr < bj[g])) { bv[g] = sc; bj[g] = rr; }
            }
        }
        if (kh == 0 && i < n) idx[(size_t)cloud * n_full + i] = min(bj[g], m - 1);
    }
    if (lane == 0 && n_again) atomicAdd((int*)mt + FM_AGAIN, n_again);
    if (lane == 0 && n_scan) atomicAdd((int*)mt + FM_SCAN, n_scan);
}
#undef DICP_F16_FETCH
#undef DICP_F16_COMMIT

// ------------------------------------------------------------------ the same filter inside the exact sorted sweep
// knn_sweep_kernel's search (dicp_kernels.hip: targets sorted by x once per call, a wave owns 128 queries that are neighbours in x and sweeps
// 64-row tiles outwards from the tile under them, a side ending where x-distance alone exceeds every query's best) with the SCORING on the
// matrix cores: per 64-row tile two A tiles straight from the image (global memory, no LDS) x the wave's four B tiles = 8 MFMAs, lane-local
// bookkeeping per (tile, lane half).  The refine is the brute-force kernel's (one candidate piece / two / a second filter pass over the visited
// tiles / the exact scan), with the sweep's tie rule: the lowest ORIGINAL index among equal scores.  Plain searches only (no certificates).
// A side ends when  0.5 dx^2 > kappa1 (b1 / s^2 + 0.5|x|^2) + kappa0  for every query of the wave, dx its x-distance to the side's last scored
// row: the right side is what a skipped row's computed score would have to beat, with the filter's error E(D) <= alpha |x|^2 + beta D + gamma
// (linear in the candidate's half squared distance D; E as in the header, sqrt(2D) <= 0.5 + D) and the rounding model of knn_sweep_kernel's
// margin (computed score of a row >= D (1 - 15u) - 0.5|x|^2 (1 + 21u)) solved for D:  beta = 2^-17 + 2 phi, kappa1 = 1 + 2.02 beta (never, when
// beta >= 1/4), kappa0 = 1.01 kappa1 (2^-15 0.5|x|^2 + phi (|x|_1 + 1.7321 |x| + 0.87) + hphi).
struct SweepBest {                   // exact (score, sorted position, original index or -1 = not looked up yet)
    float v; int s, o;
};
__device__ __forceinline__ void sweep_consider(SweepBest& b, const float* q, const float4& row, int j, const int32_t* __restrict__ pm) {
    const float sc = score<float, float4>(q, row);
    if (sc < b.v) { b.v = sc; b.s = j; b.o = -1; }
    else if (sc == b.v && sc < __builtin_huge_valf()) {
        if (b.o < 0) b.o = pm[b.s];
        const int o = pm[j];
        if (o < b.o) { b.o = o; b.s = j; }
    }
}
__device__ __forceinline__ void sweep_merge(SweepBest& b, float ov, int os, int oo, const int32_t* __restrict__ pm) {
    if (ov < b.v) { b.v = ov; b.s = os; b.o = oo; }
    else if (ov == b.v && ov < __builtin_huge_valf() && os != b.s) {
        if (b.o < 0) b.o = pm[b.s];
        if (oo < 0) oo = pm[os];
        if (oo < b.o) { b.o = oo; b.s = os; }
    }
}

__device__ __forceinline__ void f16_sweep_unit(const float* __restrict__ src, const float* __restrict__ pose, const float4* __restrict__ tgs4,
                                                                    const uint4* __restrict__ image, float* __restrict__ meta_all,
                                                                    const int32_t* __restrict__ tperm, const int32_t* __restrict__ qorder,
                                                                    const int32_t* __restrict__ bucket, const float* __restrict__ brange, int nbkt,
                                                                    int32_t* __restrict__ idx, int32_t* __restrict__ spos, unsigned long long* __restrict__ pairs,
                                                                    int N, int n_full, int m_full, int m_pad, int tiles_per_cloud, int bpc,
                                                                    const int32_t* __restrict__ src_rows, const int32_t* __restrict__ tgt_rows,
                                                                    const float* __restrict__ edges_all, int32_t* __restrict__ form_out, const int cloud, const int blk) {
    __shared__ float4 qlist[BLOCK / WAVE][32];
    __shared__ int qslot[BLOCK / WAVE][32];
    // the float32 rows of the F16_NTC tiles around the wave's first one (copied by LDS-DMA beside the scoring): where the winners are -- a
    // match is at most its own distance away in x -- so the refine reads its rows from LDS instead of gathering 512 bytes per query from L2
    // (2 GB per launch at the benchmark shape: that gather was half of the kernel)
    __shared__ __align__(256) float4 rowcache[BLOCK / WAVE][F16_NTC * WAVE];      // (256-byte aligned: a run's swizzled places are its base XOR a 4-bit field, below)
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x >> 6;
    const int unit = blk * (BLOCK / WAVE) + wave;
    const int n = rows_of(src_rows, cloud, n_full), m = min(max(rows_of(tgt_rows, cloud, m_full), 1), m_pad);
    if (unit * (32 * F16_G) >= n) return;                           // whole wave idle (no block-level synchronisation anywhere below)
    const int col = lane & 31, kh = lane >> 5;
    float* __restrict__ mt = meta_all + (size_t)cloud * F16_META;
    const float s = mt[FM_S], inv_s2 = mt[FM_INV_S2], phi = mt[FM_PHI], hphi = mt[FM_HPHI];
    float C[9], r[3];
    load_pose(pose, cloud, C, r);

    float nx[F16_G][3], xq[F16_G], k0[F16_G];
    int qi[F16_G], q_ok = 0;          // (the compiler parks three values -- 16 bytes per lane -- in scratch memory across the tile loop: stored once, loaded once.  Fetching the
                                      //  queries again behind the loop instead frees the registers -- ScratchSize 0 -- and runs 2.5-6 % SLOWER on one box: profiles/r05_kernel_resources.txt)
    F16Track tr[F16_G];
    half8 b[F16_G];
    const float beta = 7.62939453125e-6f + 2.f * phi;               // 2^-17 + 2 phi
    const float kappa1 = beta < 0.25f ? 1.f + 2.02f * beta : __builtin_huge_valf();
    const float kA = inv_s2 * kappa1;
#pragma unroll
    for (int g = 0; g < F16_G; ++g) {
        const int pos = unit * (32 * F16_G) + g * 32 + col;
        qi[g] = -1;
        float p[3] = {0.f, 0.f, 0.f};
        if (pos < n) {
            qi[g] = qorder ? qorder[(size_t)cloud * n_full + pos] : pos;
            const float* sp = src + ((size_t)cloud * n_full + qi[g]) * 3;
            p[0] = sp[0]; p[1] = sp[1]; p[2] = sp[2];
        }
        query_point(C, r, p, nx[g]);
    }
    {   // idle slots of a partial last wave take a real query's values (lane 0 of B tile 0 always holds one): they never hold the sweep open
        const float a0 = __shfl(nx[0][0], 0), a1 = __shfl(nx[0][1], 0), a2 = __shfl(nx[0][2], 0);
#pragma unroll
        for (int g = 0; g < F16_G; ++g)
            if (qi[g] < 0) { nx[g][0] = a0; nx[g][1] = a1; nx[g][2] = a2; }
    }
    float xmax = -__builtin_huge_valf(), xmin = __builtin_huge_valf();
#pragma unroll
    for (int g = 0; g < F16_G; ++g) {
        xq[g] = -nx[g][0];
        const float hx = 0.5f * (nx[g][0] * nx[g][0] + nx[g][1] * nx[g][1] + nx[g][2] * nx[g][2]);
        const float x1 = fabsf(nx[g][0]) + fabsf(nx[g][1]) + fabsf(nx[g][2]);
        k0[g] = hx * kappa1 + 1.01f * kappa1 * (3.0517578125e-5f * hx + phi * (x1 + 1.7321f * f16_sqrt_up(2.f * hx) + 0.87f) + hphi);     // kappa1 hx + kappa0
        xmax = fmaxf(xmax, xq[g]); xmin = fminf(xmin, xq[g]);
        bool ok;
        b[g] = f16_query_fragment(nx[g], s, kh, ok);
        q_ok |= ok ? (1 << g) : 0;
        if (!ok || !(k0[g] < __builtin_huge_valf())) k0[g] = __builtin_huge_valf();      // never ends a side: such a wave visits every tile, and the query is scanned exactly
        tr[g].init();
    }

    const float4* __restrict__ tg = tgs4 + (size_t)cloud * m_pad;
    const uint4* __restrict__ img = image + (size_t)cloud * tiles_per_cloud * 64;      // 64-row tile t = A tiles 2t, 2t + 1 = 128 uint4
    const int ntiles = min((m + WAVE - 1) / WAVE, m_pad / WAVE);
    const float xc = __shfl(xq[F16_G / 2], 0);                      // the wave's middle query
    const float xlo = brange[(size_t)cloud * 2], inv = brange[(size_t)cloud * 2 + 1];
    float fb = (xc - xlo) * inv;
    fb = fb < 0.f ? 0.f : (fb > (float)nbkt ? (float)nbkt : fb);
    int start = bucket[(size_t)cloud * (nbkt + 1) + (int)fb];
    {
        int hi = bucket[(size_t)cloud * (nbkt + 1) + min((int)fb + 1, nbkt)];
        while (hi - start > WAVE) {
            const int mid = (start + hi) >> 1;
            if (tg[mid].x < xc) start = mid + 1; else hi = mid;
        }
    }
    // (wave-uniform from here on, and told so: the tile counters then live in scalar registers, the loop is scalar control flow and the tiles' edge
    //  values come through the scalar cache -- as vector loads they queued behind the prefetched tiles and every slab test waited for those)
    start = __builtin_amdgcn_readfirstlane(start);
    int tR = min(max(start / WAVE, 0), ntiles - 1), tL = tR - 1;
    int visR = tR, visL = tR;
    float edgeR = -__builtin_huge_valf(), edgeL = __builtin_huge_valf();
    typedef const __attribute__((address_space(4))) float* cfloat_p;
    const cfloat_p edges = (cfloat_p)(uintptr_t)(edges_all + (size_t)cloud * (tiles_per_cloud / 2) * 2);
    uint4 preR0 = img[(size_t)tR * 128 + lane], preR1 = img[(size_t)tR * 128 + 64 + lane];
    uint4 preL0 = img[(size_t)max(tL, 0) * 128 + lane], preL1 = img[(size_t)max(tL, 0) * 128 + 64 + lane];
    f32x16 zero;
#pragma unroll
    for (int i = 0; i < 16; ++i) zero[i] = 0.f;

    const int tc0 = tR - F16_NTC / 2;                               // the row cache holds tiles [tc0, tc0 + F16_NTC)
    auto process = [&](const uint4& u0, const uint4& u1, int t) {
        half8 a0, a1;
        __builtin_memcpy(&a0, &u0, 16); __builtin_memcpy(&a1, &u1, 16);
        if (t - tc0 >= 0 && t - tc0 < F16_NTC)      // (slot `lane` of the tile's 64 takes the row whose swizzled place it is: cache_slot)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tg + (size_t)t * WAVE + (lane & 48) + ((lane & 15) ^ (((t - tc0) * 4 + (lane >> 4)) & 15))),
                                             (__attribute__((address_space(3))) void*)&rowcache[wave][(t - tc0) * WAVE], 16, 0, 0);
        int chunk;
        asm("v_mov_b32 %0, %1" : "=v"(chunk) : "s"(t));
#pragma unroll
        for (int g = 0; g < F16_G; ++g) {
            float cm = f16_chunk_min(__builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b[g], zero, 0, 0, 0), __builtin_huge_valf());
            cm = f16_chunk_min(__builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b[g], zero, 0, 0, 0), cm);
            tr[g].add(cm, chunk);
        }
    };
    // what a skipped row would have to beat, per lane: the largest of its four queries' bounds.  Taken ONCE per round of the loop below (a right and a left
    // tile): the left side's test then uses bounds that the right tile of the same round may just have lowered -- a stale bound is only larger, so a side
    // ends at most one tile later than it could, and the 20 vector instructions of the second evaluation per round are gone (round 6: the loop is bound by
    // its vector instructions, 20 per MFMA; profiles/r06_knn_f16_sweep.txt)
    float dreq = __builtin_huge_valf();
    auto bound = [&]() {
        // (a query's best so far is the smaller of its two lanes' -- each lane sees half of every tile's rows)
        float d = -__builtin_huge_valf();
#pragma unroll
        for (int g = 0; g < F16_G; ++g) d = vmax(d, fmaf(vmin(tr[g].b1, swap32(tr[g].b1)), kA, k0[g]));
        return d;
    };
    auto prunable = [&](float edge, bool right) {
        const float dx = right ? edge - xmax : xmin - edge;
        return __all(dx > 0.f && 0.5f * dx * dx > dreq) != 0;       // (inf thresholds: never)
    };
#if defined(DICP_F16_ABLATE) && DICP_F16_ABLATE == 1      // (timing builds only, scripts/f16_ablate.sh: the prologue alone)
    if (spos && lane == 0) spos[(size_t)cloud * n_full + unit] = (int)(xq[0] + k0[1]) + tR; return;
#endif
    while (tR < ntiles || tL >= 0) {
        dreq = bound();
        if (tR < ntiles) {
            if (prunable(edgeR, true)) tR = ntiles;
            else {
                const uint4 c0 = preR0, c1 = preR1;
                if (tR + 1 < ntiles) { preR0 = img[(size_t)(tR + 1) * 128 + lane]; preR1 = img[(size_t)(tR + 1) * 128 + 64 + lane]; }
                edgeR = edges[2 * tR + 1];
                process(c0, c1, tR);
                visR = ++tR;
            }
        }
        if (tL >= 0) {
            if (prunable(edgeL, false)) tL = -1;
            else {
                const uint4 c0 = preL0, c1 = preL1;
                if (tL >= 1) { preL0 = img[(size_t)(tL - 1) * 128 + lane]; preL1 = img[(size_t)(tL - 1) * 128 + 64 + lane]; }
                edgeL = edges[2 * tL];
                process(c0, c1, tL);
                visL = tL--;
            }
        }
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // (the row cache's copies have landed)
    __builtin_amdgcn_wave_barrier();
#if defined(DICP_F16_ABLATE) && DICP_F16_ABLATE == 2      // (timing builds only: prologue + the sweep)
    if (spos && kh == 0) { for (int g = 0; g < F16_G; ++g) if (qi[g] >= 0) spos[(size_t)cloud * n_full + qi[g]] = tr[g].id1 + (int)tr[g].b2 + (int)tr[g].b3 + tr[g].id2 + visR - visL; } return;
#endif
    // Where a cached row lies.  The refine reads RUNS of 16 rows, every lane its own run, all lanes row k of their run at the same time: laid out in row
    // order those reads fall on the same four LDS banks for every run (a run is 256 bytes = all 64 banks) and were served one run at a time -- bank
    // conflicts on 81 % of the LDS cycles of this kernel (profiles/r04_knn_c4_65536_pmc.txt).  Row i of run R (of the 4 F16_NTC runs of the cache) is kept
    // in place i ^ (R & 15) of its run instead: lanes on different runs now read different banks (runs R and R + 16 still share theirs).
    auto cache_slot = [&](int sl, int in_tile) { const int R = sl * 4 + (in_tile >> 4); return R * 16 + ((in_tile & 15) ^ (R & 15)); };
    auto row_at = [&](int rr) {                                     // a visited row: from the cache when its tile is in it
        const int sl = (rr >> 6) - tc0;
        return (sl >= 0 && sl < F16_NTC) ? rowcache[wave][cache_slot(sl, rr & 63)] : tg[rr];
    };
    // ---- refine.  A piece = the 2 x 16 rows of one (tile, lane half): rows 64 t + 32 a + 16 h + i, a = 0, 1
    // Written for latency: a wave passes here once, with three others per SIMD at best to hide behind -- so no scratch memory (register arrays
    // are only ever indexed by constants), the cloud's constants in scalar registers, and the rows of a query fetched eight at a time.
    const int32_t* __restrict__ pm = tperm + (size_t)cloud * m_pad;
    const int nfar = ((const int32_t*)mt)[FM_NFAR];
    // (the best row per query as three plain register arrays, indexed by constants only: as an array of structs picked from inside the rare loops below
    //  it lived in scratch memory -- 128 bytes per lane, 6.9 M scratch writes per launch, a reload behind every update; profiles/r04_knn_c4_65536_pmc.txt)
    float bV[F16_G];
    int bS[F16_G], bO[F16_G];
    float thr[F16_G];
    int run0[F16_G], run1[F16_G];       // this lane's rows to re-score per query: up to two runs of 16 (-1: none)
    int n_scan = 0, my_unsure = 0, my_scan = 0, ties = 0;       // (ties: bit g = this query's rows are taken again by the careful loop below)
#pragma unroll
    for (int g = 0; g < F16_G; ++g) {
        const F16Track me = tr[g];
        F16Track ot;
        ot.b1 = swap32(me.b1); ot.b2 = swap32(me.b2); ot.b3 = swap32(me.b3);
        ot.id1 = swap32(me.id1); ot.id2 = swap32(me.id2);
        const bool me_lo = me.b1 < ot.b1 || (me.b1 == ot.b1 && kh == 0);
        const F16Track lo = me_lo ? me : ot, hi = me_lo ? ot : me;
        const int lo_h = me_lo ? kh : (kh ^ 1), hi_h = lo_h ^ 1;
        const float B1 = lo.b1;
        const bool second_lo = lo.b2 < hi.b1;
        const float B2 = second_lo ? lo.b2 : hi.b1;
        const int c1 = lo.id1, h1 = lo_h, c2 = second_lo ? lo.id2 : hi.id1, h2 = second_lo ? lo_h : hi_h;
        const float B3 = second_lo ? __builtin_fminf(lo.b3, hi.b1) : __builtin_fminf(lo.b2, hi.b2);
        const float margin = f16_margin(nx[g], B1, s, inv_s2, phi, hphi);
        const bool bounded = ((q_ok >> g) & 1) && (B1 < __builtin_huge_valf()) && (B1 > -__builtin_huge_valf()) && margin >= 0.f;
        const int ncand = !bounded ? 0 : (B2 - B1 > margin ? 1 : (B3 - B1 > margin ? 2 : 3));
        bV[g] = __builtin_huge_valf(); bS[g] = 0; bO[g] = 0x7fffffff;
        thr[g] = B1 + margin;
        run0[g] = run1[g] = -1;
        if (ncand == 1) run0[g] = c1 * 64 + 32 * kh + 16 * h1;                            // the winner's piece, half each
        else if (ncand == 2) { const int cc = kh ? c2 : c1, hh = kh ? h2 : h1; run0[g] = cc * 64 + 16 * hh; run1[g] = run0[g] + 32; }
        if (ncand == 3 && qi[g] >= 0) my_unsure |= 1 << g;
        if (ncand == 0 && qi[g] >= 0 && kh == 0) my_scan |= 1 << g;
    }
    // A few queries of the wave with three or more candidate pieces (dense surfaces: a runner-up chunk within the filter's resolution): their visited rows are
    // scored exactly, 64 at a time, by the whole wave -- the second filter pass below walks the visited tiles one dependent load at a time for up to 32 such
    // queries at once, ~40 us of a wave's time even for ONE (round 6, planar scenes: 0.3 % of the queries took it and the launch was 0.17 ms longer for them:
    // profiles/r06_f16_sweep_crossover.txt); it is kept for waves with many.
    {
        int tot = 0;
#pragma unroll
        for (int g = 0; g < F16_G; ++g) tot += __popcll(__ballot(((my_unsure >> g) & 1) && kh == 0));
        if (tot <= F16_SCAN_MAX) { if (kh == 0) my_scan |= my_unsure; my_unsure = 0; }
    }
    // the rows, out of the wave's LDS cache (a run outside it is flagged for the careful loop).  Per run of 16 rows: the scores (independent), their
    // minimum as a tree, the first and the last row that has it -- a serial compare-and-select chain over the rows was three times the instructions,
    // and a wave's instruction count is what this part of the kernel costs.  Rows past the cloud's own are pad rows (score +inf) by construction.
    // Two rows with the minimum (duplicated targets) only raise the flag: ties go by the lowest ORIGINAL index, which the careful loop looks up.
#pragma unroll
    for (int g = 0; g < F16_G; ++g) {
#pragma unroll
        for (int rn = 0; rn < 2; ++rn) {
            const int r0 = rn ? run1[g] : run0[g];
            if (!__any(r0 >= 0)) continue;                          // (second runs: near-ties only)
            const int sl = (max(r0, 0) >> 6) - tc0;
            const bool cached = sl >= 0 && sl < F16_NTC;
            const bool far_run = r0 >= 0 && !cached;                // (a lane whose run is not in the cache gathers it from the sorted rows themselves)
            const int run = min(max(sl, 0), F16_NTC - 1) * 4 + ((max(r0, 0) & 63) >> 4);      // (r0 is a multiple of 16)
            // LDS byte address of the run's place 0, XOR its swizzle: place i ^ (run & 15) of a 256-byte-aligned run is ONE v_xor with an inline constant away
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            typedef const __attribute__((address_space(3))) f32x4* lds_row_p;
            const unsigned pre = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)&rowcache[wave][run * 16] ^ ((unsigned)(run & 15) << 4);
            float sc[16];
#pragma unroll
            for (int kb = 0; kb < 16; kb += 8) {
                float4 rw[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) { const f32x4 v = *(lds_row_p)(uintptr_t)(pre ^ ((unsigned)(kb + k) << 4)); rw[k] = make_float4(v[0], v[1], v[2], v[3]); }
                if (__any(far_run)) {
                    if (far_run) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) rw[k] = tg[r0 + kb + k];        // (r0 + 15 < m_pad: a run lies inside its tile)
                    }
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) sc[kb + k] = score<float, float4>(nx[g], rw[k]);
            }
            float mn = __builtin_fminf(sc[0], sc[1]);
#pragma unroll
            for (int k = 2; k < 16; k += 2) mn = __builtin_fminf(__builtin_fminf(mn, sc[k]), sc[k + 1]);
            int first = 15, last = 0;
#pragma unroll
            for (int k = 14; k >= 0; --k) first = (sc[k] == mn) ? k : first;
#pragma unroll
            for (int k = 1; k < 16; ++k) last = (sc[k] == mn) ? k : last;
            const bool use = r0 >= 0 && mn < __builtin_huge_valf();
            if (use && (first != last || mn == bV[g])) ties |= 1 << g;
            const bool lt = use && mn < bV[g];
            bV[g] = lt ? mn : bV[g];
            bS[g] = lt ? r0 + first : bS[g];
        }
    }
#if defined(DICP_F16_ABLATE) && DICP_F16_ABLATE == 3      // (timing builds only: everything up to and including the winners' rows)
    if (spos && kh == 0) { for (int g = 0; g < F16_G; ++g) if (qi[g] >= 0) spos[(size_t)cloud * n_full + qi[g]] = bS[g] + ties; } return;
#endif
    // The rare paths below run over the four B tiles in a LOOP (one copy of the code, the tile's values picked with selects): unrolled four times they
    // made the kernel 40 KB of instructions that every wave streamed through once -- the instruction fetches cost more than the scoring.
    static_assert(F16_G == 4, "pick4 below picks one of four");
#define DICP_PICK4(a, g) ((g) == 0 ? (a)[0] : ((g) == 1 ? (a)[1] : ((g) == 2 ? (a)[2] : (a)[3])))      /* selects over constant indices: the arrays stay in registers */
    auto put_best = [&](int g, bool mine, const SweepBest& w) {
#pragma unroll
        for (int i = 0; i < F16_G; ++i) {
            const bool on = g == i && mine;
            bV[i] = on ? w.v : bV[i]; bS[i] = on ? w.s : bS[i]; bO[i] = on ? w.o : bO[i];
        }
    };
    // exact ties inside a lane's rows (duplicated targets): those rows again, with the rule -- the lowest ORIGINAL index among equal scores
    if (__any(ties != 0)) {
#pragma unroll 1
        for (int g = 0; g < F16_G; ++g) {
            const bool mine = (ties >> g) & 1;
            if (!__any(mine)) continue;
            float qq[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) qq[k] = g == 0 ? nx[0][k] : (g == 1 ? nx[1][k] : (g == 2 ? nx[2][k] : nx[3][k]));
            SweepBest w;
            w.v = __builtin_huge_valf(); w.s = 0; w.o = 0x7fffffff;
#pragma unroll 1
            for (int rn = 0; rn < 2; ++rn) {
                const int r0 = mine ? (rn ? DICP_PICK4(run1, g) : DICP_PICK4(run0, g)) : -1;
#pragma unroll 1
                for (int k = 0; k < 16; ++k)
                    if (r0 >= 0 && r0 + k < m) sweep_consider(w, qq, row_at(r0 + k), r0 + k, pm);
            }
            put_best(g, mine, w);
        }
    }
#pragma unroll
    for (int g = 0; g < F16_G; ++g) {
        if (!((ties >> g) & 1) && bV[g] < __builtin_huge_valf()) bO[g] = -1;      // (original index not looked up yet)
        // (one or two candidate pieces: the query's two lanes hold a part each.  Lanes of a pair took the same branch: their tracks were merged)
        if (__any(run0[g] >= 0)) {
            SweepBest t;
            t.v = bV[g]; t.s = bS[g]; t.o = bO[g];
            sweep_merge(t, swap32(bV[g]), swap32(bS[g]), swap32(bO[g]), pm);
            bV[g] = t.v; bS[g] = t.s; bO[g] = t.o;
        }
    }
    // queries the filter has no bound for: every visited row, 64 at a time (such a wave never pruned: it visited every tile)
    if (__any(my_scan != 0)) {
#pragma unroll 1
        for (int g = 0; g < F16_G; ++g) {
            unsigned long long need = __ballot((my_scan >> g) & 1);
            float qq[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) qq[k] = g == 0 ? nx[0][k] : (g == 1 ? nx[1][k] : (g == 2 ? nx[2][k] : nx[3][k]));
            while (need) {
                const int L = __builtin_ctzll(need);
                need &= need - 1;
                const float q[3] = {__shfl(qq[0], L), __shfl(qq[1], L), __shfl(qq[2], L)};
                SweepBest w;
                w.v = __builtin_huge_valf(); w.s = 0; w.o = 0x7fffffff;
                {   // (four rows in flight per lane: one row per round trip made this a chain of dependent loads)
                    const int r_end = min(visR * WAVE, m);
                    int rr = visL * WAVE + lane;
#pragma unroll 1
                    for (; rr + 3 * WAVE < r_end; rr += 4 * WAVE) {
                        const float4 y0 = tg[rr], y1 = tg[rr + WAVE], y2 = tg[rr + 2 * WAVE], y3 = tg[rr + 3 * WAVE];
                        sweep_consider(w, q, y0, rr, pm); sweep_consider(w, q, y1, rr + WAVE, pm);
                        sweep_consider(w, q, y2, rr + 2 * WAVE, pm); sweep_consider(w, q, y3, rr + 3 * WAVE, pm);
                    }
#pragma unroll 1
                    for (; rr < r_end; rr += WAVE) sweep_consider(w, q, tg[rr], rr, pm);
                }
#pragma unroll 1
                for (int o = WAVE / 2; o > 0; o >>= 1) sweep_merge(w, __shfl_xor(w.v, o), __shfl_xor(w.s, o), __shfl_xor(w.o, o), pm);
                put_best(g, col == (L & 31), w);
                ++n_scan;
            }
        }
    }

    // ---- second filter pass over the visited tiles for the queries with three or more candidate pieces: one B tile of up to 32 of them per round
    int cnt = 0;
    if (__any(my_unsure != 0)) {
        int base[F16_G];
#pragma unroll
        for (int g = 0; g < F16_G; ++g) { base[g] = cnt; cnt += __popcll(__ballot(((my_unsure >> g) & 1) && kh == 0)); }
#pragma unroll 1
        for (int rd = 0; rd * 32 < cnt; ++rd) {
#pragma unroll 1
            for (int g = 0; g < F16_G; ++g) {
                const bool mineg = ((my_unsure >> g) & 1) && kh == 0;
                const unsigned long long mask = __ballot(mineg);
                const int rank = DICP_PICK4(base, g) + __popcll(mask & ((1ull << lane) - 1)) - 32 * rd;
                if (mineg && rank >= 0 && rank < 32) {
                    float qq[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) qq[k] = g == 0 ? nx[0][k] : (g == 1 ? nx[1][k] : (g == 2 ? nx[2][k] : nx[3][k]));
                    qlist[wave][rank] = make_float4(qq[0], qq[1], qq[2], DICP_PICK4(thr, g));
                    qslot[wave][rank] = (g << 8) | col;
                }
            }
            const int have = min(cnt - 32 * rd, 32);
            __builtin_amdgcn_wave_barrier();
            float q[3] = {0.f, 0.f, 0.f}, tau = -__builtin_huge_valf();
            if (col < have) { const float4 e = qlist[wave][col]; q[0] = e.x; q[1] = e.y; q[2] = e.z; tau = e.w; }
            bool ok2;
            const half8 b2 = f16_query_fragment(q, s, kh, ok2);
            SweepBest w;
            w.v = __builtin_huge_valf(); w.s = 0; w.o = 0x7fffffff;
#pragma unroll 1
            for (int t = visL * 2; t < visR * 2; ++t) {             // A tiles of the visited range
                half8 a;
                { const uint4 v = img[(size_t)t * 64 + lane]; __builtin_memcpy(&a, &v, 16); }
                const float cm = f16_chunk_min(__builtin_amdgcn_mfma_f32_32x32x16_f16(a, b2, zero, 0, 0, 0), __builtin_huge_valf());
                if (cm <= tau) {
#pragma unroll 1
                    for (int k = 0; k < 16; ++k) { const int rr = t * 32 + 16 * kh + k; if (rr < m) sweep_consider(w, q, tg[rr], rr, pm); }
                }
            }
            sweep_merge(w, swap32(w.v), swap32(w.s), swap32(w.o), pm);
#pragma unroll 1
            for (int e = 0; e < have; ++e) {
                const int slot = qslot[wave][e];
                SweepBest r;
                r.v = __shfl(w.v, e); r.s = __shfl(w.s, e); r.o = __shfl(w.o, e);
                put_best(slot >> 8, col == (slot & 0xff), r);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }

    // ---- far rows (left out of the image), then the results as knn_sweep_kernel writes them
    if (nfar > 0) {
#pragma unroll 1
        for (int g = 0; g < F16_G; ++g) {
            float qq[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) qq[k] = g == 0 ? nx[0][k] : (g == 1 ? nx[1][k] : (g == 2 ? nx[2][k] : nx[3][k]));
            SweepBest w;
            w.v = DICP_PICK4(bV, g); w.s = DICP_PICK4(bS, g); w.o = DICP_PICK4(bO, g);
#pragma unroll 1
            for (int f = 0; f < nfar; ++f) {
                const int rr = ((const int32_t*)mt)[FM_FAR0 + f];
                if (rr < m) sweep_consider(w, qq, tg[rr], rr, pm);
            }
            put_best(g, true, w);
        }
    }
#pragma unroll
    for (int g = 0; g < F16_G; ++g) {
        if (qi[g] < 0 || kh != 0) continue;
        const bool found = bV[g] < __builtin_huge_valf();
        int bo = bO[g];
        if (found && bo < 0) bo = pm[bS[g]];
        if (!found) bo = 0x7fffffff;
        if (idx) idx[(size_t)cloud * n_full + qi[g]] = (bo == 0x7fffffff) ? 0 : min(max(bo, 0), m - 1);
        if (spos) spos[(size_t)cloud * n_full + qi[g]] = (bo == 0x7fffffff || bo >= m) ? -1 : bS[g];
    }
    if (lane == 0) {
        if (cnt) atomicAdd((int*)mt + FM_AGAIN, cnt);
        if (n_scan) atomicAdd((int*)mt + FM_SCAN, n_scan);
        if (pairs) atomicAdd(pairs + (blockIdx.x & (DICP_PAIR_SHARDS - 1)), (unsigned long long)(visR - visL) * WAVE * (32 * F16_G));
        if (form_out) atomicAdd(form_out + cloud, visR - visL);
    }
}
#undef DICP_PICK4

// The launch: one block per 4 units of every cloud; given the clouds' tallies (the loop's per-cloud choice of the form) the blocks of a cloud whose slabs were
// short leave at once -- the vector form's launch has it.
template <int MINW>
__global__ __launch_bounds__(BLOCK, MINW) void knn_f16_sweep_kernel(const float* __restrict__ src, const float* __restrict__ pose, const float4* __restrict__ tgs4,
                                                                    const uint4* __restrict__ image, float* __restrict__ meta_all,
                                                                    const int32_t* __restrict__ tperm, const int32_t* __restrict__ qorder,
                                                                    const int32_t* __restrict__ bucket, const float* __restrict__ brange, int nbkt,
                                                                    int32_t* __restrict__ idx, int32_t* __restrict__ spos, unsigned long long* __restrict__ pairs,
                                                                    int N, int n_full, int m_full, int m_pad, int tiles_per_cloud, int bpc,
                                                                    const int32_t* __restrict__ src_rows, const int32_t* __restrict__ tgt_rows,
                                                                    const float* __restrict__ edges_all, FormPick pick, int32_t* __restrict__ form_out) {
    int cloud, blk;
    if (!decode_block(bpc, N, cloud, blk)) return;
    if (!form_long(pick, cloud)) return;
    f16_sweep_unit(src, pose, tgs4, image, meta_all, tperm, qorder, bucket, brange, nbkt, idx, spos, pairs, N, n_full, m_full, m_pad, tiles_per_cloud, bpc,
                   src_rows, tgt_rows, edges_all, form_out, cloud, blk);
}

// ------------------------------------------------------------------ the filter's error bound, held to account (a test aid: dicp_knn_f16_probe)
// For EVERY (query, image row) pair of a cloud: the filter value exactly as the searches compute it (the same image tile, the same query fragment, one
// v_mfma_f32_32x32x16_f16 into a zero accumulator) against score() -- the float32 value every other search form compares -- and against the bound E of this
// file's header evaluated for THAT pair (T = sum |x_i y_i| + h).  Per cloud: the largest |filter / s^2 - score()| / E, that error, its E, and the pairs
// checked.  The searches rely on ratio <= 1; the header's MFMA term assumes 4x the worst accumulation error that was measured, so a healthy margin is < 0.5.
// One wave = 32 queries against all tiles of the cloud.
__global__ __launch_bounds__(BLOCK) void knn_f16_probe_kernel(const float* __restrict__ src, const float* __restrict__ pose, const float4* __restrict__ tgt4,
                                                               const uint4* __restrict__ image, const float* __restrict__ meta_all, int N, int n_full, int m_full,
                                                               int m_pad, int tiles_per_cloud, int bpc, const int32_t* __restrict__ src_rows,
                                                               const int32_t* __restrict__ tgt_rows, float* __restrict__ out /* (N,4) */) {
    int cloud, blk;
    if (!decode_block(bpc, N, cloud, blk)) return;
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x >> 6, col = lane & 31, kh = lane >> 5;
    const int n = rows_of(src_rows, cloud, n_full), m = min(max(rows_of(tgt_rows, cloud, m_full), 1), m_pad);
    const int q = (blk * (BLOCK / WAVE) + wave) * 32 + col;
    if ((blk * (BLOCK / WAVE) + wave) * 32 >= n) return;
    const float* mt = meta_all + (size_t)cloud * F16_META;
    const float s = mt[FM_S], inv_s2 = mt[FM_INV_S2], phi = mt[FM_PHI], hphi = mt[FM_HPHI];
    float C[9], r[3];
    load_pose(pose, cloud, C, r);
    float p[3] = {0.f, 0.f, 0.f}, nx[3];
    if (q < n) { const float* sp = src + ((size_t)cloud * n_full + q) * 3; p[0] = sp[0]; p[1] = sp[1]; p[2] = sp[2]; }
    query_point(C, r, p, nx);
    bool ok;
    const half8 b = f16_query_fragment(nx, s, kh, ok);
    ok = ok && q < n;
    const float x1 = fabsf(nx[0]) + fabsf(nx[1]) + fabsf(nx[2]);
    const float4* __restrict__ tg = tgt4 + (size_t)cloud * m_pad;
    const uint4* __restrict__ img = image + (size_t)cloud * tiles_per_cloud * 64;
    f32x16 zero;
#pragma unroll
    for (int i = 0; i < 16; ++i) zero[i] = 0.f;
    double best = 0.0, best_err = 0.0, best_E = 0.0;
    unsigned long long cnt = 0;
    const int ntiles = min((m + 31) / 32, tiles_per_cloud);
    for (int t = 0; t < ntiles; ++t) {
        half8 a;
        { const uint4 v = img[(size_t)t * 64 + lane]; __builtin_memcpy(&a, &v, 16); }
        const f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, zero, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = t * 32 + 16 * kh + i;
            const float f = d[i];
            if (!ok || row >= m || !(f < __builtin_huge_valf()) || !(f > -__builtin_huge_valf())) continue;      // (rows left out of the image carry +inf)
            const float4 y = tg[row];
            const float sc = score<float, float4>(nx, y);
            const double err = fabs((double)f * (double)inv_s2 - (double)sc);
            const double T = fabs((double)nx[0] * y.x) + fabs((double)nx[1] * y.y) + fabs((double)nx[2] * y.z) + (double)y.w;
            const double ynorm = sqrt(2.0 * (double)y.w);
            const double E = (double)F16_CREL * (double)F16_U * T + (double)phi * ((double)x1 + 1.7321 * ynorm) + (double)hphi;
            const double ratio = err / E;
            ++cnt;
            if (ratio > best) { best = ratio; best_err = err; best_E = E; }
        }
    }
#pragma unroll 1
    for (int o = WAVE / 2; o > 0; o >>= 1) {
        const double ob = __shfl_xor(best, o), oe = __shfl_xor(best_err, o), oE = __shfl_xor(best_E, o);
        cnt += __shfl_xor(cnt, o);
        if (ob > best) { best = ob; best_err = oe; best_E = oE; }
    }
    if (lane == 0) {
        // (non-negative floats order like their bit patterns: the maximum as an integer atomic; the error and its bound of the wave that raised it last)
        const float bf = (float)best;
        const int old = atomicMax((int*)out + (size_t)cloud * 4, __float_as_int(bf));
        if (__float_as_int(bf) > old) { out[(size_t)cloud * 4 + 1] = (float)best_err; out[(size_t)cloud * 4 + 2] = (float)best_E; }
        atomicAdd(out + (size_t)cloud * 4 + 3, (float)cnt);
    }
}

}  // namespace

namespace dicp_tu {

int knn_f16_probe(const void* src, const void* pose, const void* tgt4, const void* image, const int32_t* src_rows, const int32_t* tgt_rows,
                  int N, int n, int m, int m_pad, float* out, void* stream) {
    if (!src || !tgt4 || !image || !out) return DICP_ERR_NULL;
    if (N <= 0 || n <= 0 || m <= 0 || m_pad < m) return DICP_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int m_img = knn_f16_image_rows(m_pad), tiles = m_img / 32;
    const float* meta = (const float*)((const char*)image + knn_f16_meta_offset(N, m_pad));
    const int bpc = (n + (BLOCK / WAVE) * 32 - 1) / ((BLOCK / WAVE) * 32);
    begin_launch();
    if (const int e = dicp_fill::zero(out, (size_t)N * 4 * sizeof(float), st)) return e;
    knn_f16_probe_kernel<<<grid_for(N, bpc), BLOCK, 0, st>>>((const float*)src, (const float*)pose, (const float4*)tgt4, (const uint4*)image, meta, N, n, m, m_pad, tiles, bpc,
                                                             src_rows, tgt_rows, out);
    return launch_status();
}

int knn_f16_pack(const void* rows4, const int32_t* tgt_rows, int N, int m_full, int m_pad, void* image, void* stream) {
    if (!rows4 || !image) return DICP_ERR_NULL;
    if (N <= 0 || m_full <= 0 || m_pad < m_full || (m_pad % 32)) return DICP_ERR_SHAPE;
    if (((uintptr_t)rows4 % 16) || ((uintptr_t)image % 16)) return DICP_ERR_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    const int m_img = knn_f16_image_rows(m_pad), tiles = m_img / 32;
    float* meta = (float*)((char*)image + knn_f16_meta_offset(N, m_pad));
    float* edges = (float*)((char*)image + knn_f16_edges_offset(N, m_pad));
    begin_launch();
    knn_f16_scale_kernel<<<N, F16_SCALE_THREADS, 0, st>>>((const float4*)rows4, tgt_rows, m_full, m_pad, meta);
    knn_f16_image_kernel<<<dim3((tiles + BLOCK / WAVE - 1) / (BLOCK / WAVE), N), BLOCK, 0, st>>>((const float4*)rows4, tgt_rows, m_full, m_pad, m_img, meta, (uint4*)image, tiles, N, edges);
    return launch_status();
}

int knn_f16_brute(const void* src, const void* pose, const void* tgt4, void* image, const int32_t* src_rows, const int32_t* tgt_rows,
                  int N, int n, int m, int m_pad, int32_t* idx, void* stream) {
    if (!src || !tgt4 || !image || !idx) return DICP_ERR_NULL;
    if (N <= 0 || n <= 0 || m <= 0 || m_pad < m) return DICP_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int m_img = knn_f16_image_rows(m_pad), tiles = m_img / 32;
    float* meta = (float*)((char*)image + knn_f16_meta_offset(N, m_pad));        // (its two counters are added to)
    const int bpc = (n + (BLOCK / WAVE) * 32 * F16_G - 1) / ((BLOCK / WAVE) * 32 * F16_G);
    begin_launch();
    knn_f16_kernel<4><<<grid_for(N, bpc), BLOCK, 0, st>>>((const float*)src, (const float*)pose, (const float4*)tgt4, (const uint4*)image, meta, idx,
                                                         N, n, m, m_pad, tiles, bpc, src_rows, tgt_rows);
    return launch_status();
}

int knn_f16_sweep(const void* src, const void* pose, const void* tgs4, void* image, const int32_t* tperm, const int32_t* qorder, const int32_t* bucket,
                  const void* brange, int nbkt, const int32_t* src_rows, const int32_t* tgt_rows, int N, int n, int m, int m_pad, int32_t* idx, int32_t* spos,
                  unsigned long long* pairs, const int32_t* form_in, int32_t* form_out, int form_tiles, int form_default, void* ev0, void* ev1, void* stream) {
    if (!src || !tgs4 || !image || !tperm || !bucket || !brange || (!idx && !spos)) return DICP_ERR_NULL;
    if (N <= 0 || n <= 0 || m <= 0 || m_pad < m || (m_pad % WAVE)) return DICP_ERR_SHAPE;
    const int m_img = knn_f16_image_rows(m_pad), tiles = m_img / 32;
    float* meta = (float*)((char*)image + knn_f16_meta_offset(N, m_pad));
    const float* edges = (const float*)((const char*)image + knn_f16_edges_offset(N, m_pad));
    const int units = (n + 32 * F16_G - 1) / (32 * F16_G);
    const int bpc = (units + BLOCK / WAVE - 1) / (BLOCK / WAVE);
    const FormPick pick{form_in, src_rows, n, form_tiles, form_default};
    begin_launch();
    hipExtLaunchKernelGGL((knn_f16_sweep_kernel<4>), dim3(grid_for(N, bpc)), dim3(BLOCK), 0, (hipStream_t)stream, (hipEvent_t)ev0, (hipEvent_t)ev1, 0,
                          (const float*)src, (const float*)pose, (const float4*)tgs4, (const uint4*)image, meta, tperm, qorder, bucket, (const float*)brange, nbkt,
                          idx, spos, pairs, N, n, m, m_pad, tiles, bpc, src_rows, tgt_rows, edges, pick, form_out);
    return launch_status();
}

}  // namespace dicp_tu
