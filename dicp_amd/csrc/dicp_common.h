// Shared by the translation units of libdicp_hip.so: launch geometry, the XCD-aware block -> cloud map, ragged-batch row counts,
// and the ONE definition of the query and of a score that every search form uses (so that all of them see bit-identical values).
// Everything here has internal linkage (anonymous namespace): each .hip file gets its own copy.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "../../include/dicp_hip.h"
#include "dicp_math.h"

using namespace dicp;

namespace {

constexpr int BLOCK = 256;
constexpr int WAVE = 64;
#ifndef DICP_ACC_PTS
#define DICP_ACC_PTS 512
#endif
constexpr int ACC_PTS = DICP_ACC_PTS;  // source points per accumulate block
constexpr int KNN_PAD = 64;            // m_pad granularity: 4 MFMA tiles of 16 targets / largest VALU chunk

template <typename T> struct V4;
template <> struct V4<float>  { using type = float4; };
template <> struct V4<double> { using type = double4; };

template <typename T> __device__ __forceinline__ T inf_v();
template <> __device__ __forceinline__ float  inf_v<float>()  { return __builtin_huge_valf(); }
template <> __device__ __forceinline__ double inf_v<double>() { return __builtin_huge_val(); }
// (a little above) the machine epsilon: roundings of the match-certificate bookkeeping are covered with multiples of it
template <typename T> struct CertUlp;
template <> struct CertUlp<float>  { static constexpr float  v = 1.2e-7f; };
template <> struct CertUlp<double> { static constexpr double v = 2.3e-16; };

__device__ __forceinline__ float  fma_t(float a, float b, float c)    { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float  min_t(float a, float b)   { return __builtin_fminf(a, b); }
__device__ __forceinline__ double min_t(double a, double b) { return __builtin_fmin(a, b); }
__device__ __forceinline__ float  max_t(float a, float b)   { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ double max_t(double a, double b) { return __builtin_fmax(a, b); }

// Blocks b and b+8 share an XCD (round-robin dispatch; speed only, never correctness):
// give every cloud's blocks the same b % 8.
__device__ __forceinline__ bool decode_block(int bpc, int N, int& cloud, int& blk) {
    const int b = blockIdx.x;
    const int i = b >> 3;
    cloud = (i / bpc) * 8 + (b & 7);
    blk = i % bpc;
    return cloud < N;
}
inline unsigned grid_for(int N, int bpc) { return 8u * (unsigned)((N + 7) / 8) * (unsigned)bpc; }

// Ragged batches (ICP.py:305-511 pads every cloud to the longest): rows[cloud] = leading rows of the cloud that take part
// (NULL: all `full` of them).  The kernels never read, score or accumulate a row beyond it.
__device__ __forceinline__ int rows_of(const int32_t* __restrict__ rows, int cloud, int full) {
    return rows ? min(max(rows[cloud], 0), full) : full;
}

template <typename T>
__device__ __forceinline__ void load_pose(const T* __restrict__ pose, int cloud, T* C, T* r) {
    if (pose) {
        const T* p = pose + (size_t)cloud * 12;
#pragma unroll
        for (int k = 0; k < 9; ++k) C[k] = p[k];
        r[0] = p[9]; r[1] = p[10]; r[2] = p[11];
    } else {
#pragma unroll
        for (int k = 0; k < 9; ++k) C[k] = (k % 4 == 0) ? T(1) : T(0);
        r[0] = r[1] = r[2] = T(0);
    }
}

// -(C p + r), the query every kNN form scores with: ONE explicit fma chain, so that all forms (VALU, packed, MFMA,
// sweep, scan) see bit-identical queries whatever the compiler would contract in their different surroundings
template <typename T>
__device__ __forceinline__ void query_point(const T* C, const T* r, const T* p, T* nx) {
#pragma unroll
    for (int k = 0; k < 3; ++k)
        nx[k] = -fma_t(C[3 * k], p[0], fma_t(C[3 * k + 1], p[1], fma_t(C[3 * k + 2], p[2], r[k])));
}

// ------------------------------------------------------------------- kNN scores
// score(x, y) = 0.5|y|^2 - x.y  = 0.5(|x-y|^2 - |x|^2): same argmin as the distance.
template <typename T, typename T4>
__device__ __forceinline__ T score(const T* nx, const T4& y) {
    return fma_t(nx[0], y.x, fma_t(nx[1], y.y, fma_t(nx[2], y.z, y.w)));
}

// ------------------------------------------------------------------- host helpers
// hipGetLastError() is sticky per host thread and the HIP runtime is shared with PyTorch, which can
// leave an unrelated error behind: every entry point clears it (begin_launch) before launching and
// reads it back (launch_status) after, so the status returned is that of OUR launch only.
inline void begin_launch() { (void)hipGetLastError(); }
inline int launch_status() {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}
inline bool bad_dtype(int d) { return d != DICP_F32 && d != DICP_F64; }

struct Rows { const int32_t* src; const int32_t* tgt; };     // optional per-cloud row counts of a ragged batch

}  // namespace
