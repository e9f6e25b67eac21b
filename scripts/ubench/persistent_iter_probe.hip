// Would a certified ICP iteration be cheaper inside ONE launch?  One 1024-thread block per cloud (256 clouds = 256 CUs) runs K iterations of
// { per-point pass over the cloud (12-byte point, 4-byte match, gathered 24-byte row, 4-byte weight out; the arithmetic of dicp_math.h's point_forward),
// block reduction of the 30 sums, a serial step by one lane, new pose } against the same work as K x { 4096-block pass, 256-block step } launches.
// Build: hipcc -O3 --offload-arch=gfx950 -I ../../dicp_amd/csrc -I ../../include -o persistent_iter_probe persistent_iter_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
#include "dicp_math.h"
using namespace dicp;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr int NPT = 16384, NCL = 256;

__device__ __forceinline__ void reduce30(float* acc, float* lds, int nt, float* out) {      // block sum of 30 values -> out[0..30) (order not matched to the product)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int k = 0; k < NACC; ++k) {
        float v = acc[k];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) lds[wave * 32 + k] = v;
    }
    __syncthreads();
    if (tid < NACC) { float s = 0.f; for (int w = 0; w < nt / 64; ++w) s += lds[w * 32 + tid]; out[tid] = s; }
    __syncthreads();
}
__device__ __forceinline__ void fake_step(const float* sums, float* pose /* 12 */, int work) {     // one lane: a chain of dependent double operations (~5 us), then the pose
    double x = sums[0] + 1e-3;
    for (int i = 0; i < work; ++i) x = fma(x, 0.99999, 1e-7 * sums[i % 30]);
    for (int k = 0; k < 12; ++k) pose[k] = (k % 4 == 0 && k < 9) ? 1.f + (float)(x * 1e-12) : (float)(x * 1e-12);
}
template <int NT>
__device__ __forceinline__ void point_pass(const WeightParams& P, const float* src, const float* rows, const int* spos, const float* pose, float* w_out, int lo, int hi, float* acc) {
    float C[9], r[3];
    for (int k = 0; k < 9; ++k) C[k] = pose[k];
    for (int k = 0; k < 3; ++k) r[k] = pose[9 + k];
    for (int i = lo + (int)threadIdx.x; i < hi; i += NT) {
        const float p[3] = {src[i * 3], src[i * 3 + 1], src[i * 3 + 2]};
        const int j = spos[i];
        const float* yp = rows + (size_t)j * 6;
        const float y[3] = {yp[0], yp[1], yp[2]}, nrm[3] = {yp[3], yp[4], yp[5]};
        PointState<float> s;
        point_forward<float, MODE_PT2PL>(P, C, r, p, y, nrm, 1.f, acc, s);
        w_out[i] = s.w;
    }
}
__global__ __launch_bounds__(1024) void persistent(WeightParams P, const float* src, const float* rows, const int* spos, float* poses, float* w_hist, int K, int work) {
    __shared__ float lds[16 * 32], sums[32], spose[12];
    const int cloud = blockIdx.x;
    if (threadIdx.x < 12) spose[threadIdx.x] = poses[cloud * 12 + threadIdx.x];
    __syncthreads();
    for (int k = 0; k < K; ++k) {
        float acc[NACC];
        for (int q = 0; q < NACC; ++q) acc[q] = 0.f;
        point_pass<1024>(P, src + (size_t)cloud * NPT * 3, rows + (size_t)cloud * NPT * 6, spos + (size_t)cloud * NPT, spose, w_hist + ((size_t)k * NCL + cloud) * NPT, 0, NPT, acc);
        reduce30(acc, lds, 1024, sums);
        if (threadIdx.x == 0) fake_step(sums, spose, work);
        __syncthreads();
    }
    if (threadIdx.x < 12) poses[cloud * 12 + threadIdx.x] = spose[threadIdx.x];
}
__global__ __launch_bounds__(256) void pass_kernel(WeightParams P, const float* src, const float* rows, const int* spos, const float* poses, float* w_out, float* partials) {
    __shared__ float lds[4 * 32];
    const int b = blockIdx.x, i8 = b >> 3, cloud = (i8 / 16) * 8 + (b & 7), blk = i8 % 16;       // (the product's XCD-aware block -> cloud map)
    float acc[NACC];
    for (int q = 0; q < NACC; ++q) acc[q] = 0.f;
    point_pass<256>(P, src + (size_t)cloud * NPT * 3, rows + (size_t)cloud * NPT * 6, spos + (size_t)cloud * NPT, poses + cloud * 12, w_out + (size_t)cloud * NPT, blk * 1024, blk * 1024 + 1024, acc);
    reduce30(acc, lds, 256, partials + ((size_t)cloud * 16 + blk) * 32);
}
__global__ __launch_bounds__(64) void step_kernel(const float* partials, float* poses, int work) {
    __shared__ float sums[32];
    const int cloud = blockIdx.x;
    if (threadIdx.x < 30) { float s = 0.f; for (int b = 0; b < 16; ++b) s += partials[((size_t)cloud * 16 + b) * 32 + threadIdx.x]; sums[threadIdx.x] = s; }
    __syncthreads();
    if (threadIdx.x == 0) { float pose[12]; fake_step(sums, pose, work); for (int k = 0; k < 12; ++k) poses[cloud * 12 + k] = pose[k]; }
}
int main() {
    const int K = 16;
    std::mt19937 rng(1);
    std::uniform_real_distribution<float> U(-10.f, 10.f);
    std::vector<float> src((size_t)NCL * NPT * 3), rows((size_t)NCL * NPT * 6), poses(NCL * 12, 0.f);
    std::vector<int> spos((size_t)NCL * NPT);
    for (auto& v : src) v = U(rng);
    for (size_t i = 0; i < rows.size(); i += 6) { rows[i] = U(rng); rows[i + 1] = U(rng); rows[i + 2] = U(rng); rows[i + 3] = 0.6f; rows[i + 4] = 0.64f; rows[i + 5] = 0.48f; }
    for (auto& v : spos) v = rng() % NPT;
    for (int c = 0; c < NCL; ++c) { poses[c * 12] = poses[c * 12 + 4] = poses[c * 12 + 8] = 1.f; }
    float *dsrc, *drows, *dposes, *dw, *dpart; int* dspos;
    CHECK(hipMalloc(&dsrc, src.size() * 4)); CHECK(hipMalloc(&drows, rows.size() * 4)); CHECK(hipMalloc(&dposes, poses.size() * 4));
    CHECK(hipMalloc(&dw, (size_t)K * NCL * NPT * 4)); CHECK(hipMalloc(&dpart, (size_t)NCL * 16 * 32 * 4)); CHECK(hipMalloc(&dspos, spos.size() * 4));
    CHECK(hipMemcpy(dsrc, src.data(), src.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(drows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dposes, poses.data(), poses.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dspos, spos.data(), spos.size() * 4, hipMemcpyHostToDevice));
    WeightParams P; P.mode = MODE_PT2PL; P.trim_on = 1; P.differentiable = 1; P.loss = 1; P.trim_dist = 5.0; P.tanh_k = 5.0; P.loss_delta = 1.0; P.match_thresh = 0.0;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int work : {500, 2000}) {
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipEventRecord(e0));
            persistent<<<NCL, 1024>>>(P, dsrc, drows, dspos, dposes, dw, K, work);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 2) printf("persistent (1 block of 1024 threads per cloud), serial step of %4d dependent ops: %7.1f us per iteration\n", work, ms * 1e3 / K);
        }
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipEventRecord(e0));
            for (int k = 0; k < K; ++k) {
                pass_kernel<<<NCL * 16, 256>>>(P, dsrc, drows, dspos, dposes, dw + (size_t)k * NCL * NPT, dpart);
                step_kernel<<<NCL, 64>>>(dpart, dposes, work);
            }
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 2) printf("two launches per iteration (4096 x 256 threads, then 256 x 64),   step of %4d dependent ops: %7.1f us per iteration\n", work, ms * 1e3 / K);
        }
    }
    return 0;
}
