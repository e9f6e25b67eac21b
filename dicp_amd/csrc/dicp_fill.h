// Zero fills and small device-to-device copies as KERNELS of this library -- not hipMemsetAsync / hipMemcpyAsync.  Inside a captured hipGraph (dicp_amd/graphed.py
// captures whole calls) the runtime's memset node is not ordered against the kernel nodes around it when a replay starts on an idle GPU: the zero fill of a
// backward pass's counters and side buffer ran under the kernels that were already adding to them, and every replay that followed a synchronisation returned
// garbage target gradients while replays back to back were right (round 6; tests/test_gpu_configs.py::test_captured_step_after_a_synchronisation).  A kernel node
// is ordered like every other kernel of the stream.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dicp_fill {

static __global__ __launch_bounds__(256) void zero16_kernel(uint4* __restrict__ p, size_t n16) {
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = z;
}
static __global__ __launch_bounds__(256) void zero4_kernel(uint32_t* __restrict__ p, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) p[i] = 0u;
}
// rows x row_words 32-bit words, source and destination pitches in words (a plain copy: rows = 1)
static __global__ __launch_bounds__(256) void copy4_kernel(uint32_t* __restrict__ dst, size_t dst_pitch, const uint32_t* __restrict__ src, size_t src_pitch, size_t row_words, size_t rows) {
    const size_t total = row_words * rows;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t r = i / row_words, c = i - r * row_words;
        dst[r * dst_pitch + c] = src[r * src_pitch + c];
    }
}
static __global__ __launch_bounds__(256) void copy16_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
inline unsigned grid_of(size_t items) { const size_t g = (items + 255) / 256; return (unsigned)(g < 1 ? 1 : (g > 8192 ? 8192 : g)); }

// -> 0 or -(hipError).  bytes and p must be multiples of 4 (every buffer of this library is)
inline int zero(void* p, size_t bytes, hipStream_t st) {
    if (!bytes) return 0;
    (void)hipGetLastError();
    if (!(((uintptr_t)p | bytes) & 15)) zero16_kernel<<<grid_of(bytes / 16), 256, 0, st>>>((uint4*)p, bytes / 16);
    else if (!(((uintptr_t)p | bytes) & 3)) zero4_kernel<<<grid_of(bytes / 4), 256, 0, st>>>((uint32_t*)p, bytes / 4);
    else return -(int)hipErrorInvalidValue;
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}
inline int copy_rows(void* dst, size_t dst_pitch_bytes, const void* src, size_t src_pitch_bytes, size_t row_bytes, size_t rows, hipStream_t st) {
    if (!row_bytes || !rows) return 0;
    if ((((uintptr_t)dst | (uintptr_t)src | dst_pitch_bytes | src_pitch_bytes | row_bytes) & 3)) return -(int)hipErrorInvalidValue;
    (void)hipGetLastError();
    if (rows == 1 && !(((uintptr_t)dst | (uintptr_t)src | row_bytes) & 15)) copy16_kernel<<<grid_of(row_bytes / 16), 256, 0, st>>>((uint4*)dst, (const uint4*)src, row_bytes / 16);
    else copy4_kernel<<<grid_of(row_bytes / 4 * rows), 256, 0, st>>>((uint32_t*)dst, dst_pitch_bytes / 4, (const uint32_t*)src, src_pitch_bytes / 4, row_bytes / 4, rows);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}
inline int copy(void* dst, const void* src, size_t bytes, hipStream_t st) { return copy_rows(dst, bytes, src, bytes, bytes, 1, st); }

}  // namespace dicp_fill
