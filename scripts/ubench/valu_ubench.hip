// Micro-benchmarks that size the roofs the kNN kernel lives under (run on the MI355X box):
// f32 VALU issue rate (v_fma / v_pk_fma / v_min3), f32 MFMA 16x16x4 rate, MFMA+VALU co-issue,
// and the clock the chip holds under each load (s_memtime vs s_memrealtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0 v_fma_f32, 1 v_pk_fma_f32, 2 v_min3_f32, 3 mfma only, 4 mfma + VF fma per mfma (same wave),
// 5 odd waves mfma / even waves fma (different waves, same SIMD when >= 8 waves per block)
template <int MODE, int VF>
__global__ __launch_bounds__(512) void kern(float* out, unsigned long long* clk, int iters) {
    const int tid = threadIdx.x;
    float a[16];
    f32x2 p[16];
    f32x4 acc[8];
    for (int j = 0; j < 16; ++j) { a[j] = tid * 0.001f + j; p[j] = {a[j], a[j] + 1.f}; }
    for (int j = 0; j < 8; ++j) acc[j] = {0.f, 0.f, 0.f, 0.f};
    const float x = 1.0001f + tid * 1e-7f, y = 0.5f;
    const f32x2 x2 = {x, x}, y2 = {y, y};
    const int wave = tid >> 6;
    const bool mf = (MODE == 3) || (MODE == 4) || (MODE == 5 && ((wave >> 2) & 1));   // waves 4..7 = 2nd wave of each SIMD
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || (MODE == 5 && !mf)) {
#pragma unroll
            for (int j = 0; j < 16; ++j) a[j] = __builtin_fmaf(a[j], x, y);
        } else if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 16; ++j) p[j] = p[j] * x2 + y2;
        } else if (MODE == 2) {
#pragma unroll
            for (int j = 0; j < 16; ++j) a[j] = __builtin_fminf(__builtin_fminf(a[j], x + j), a[(j + 1) & 15]);
        } else if (mf) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, a[j], acc[j], 0, 0, 0);
                if (MODE == 4) {
#pragma unroll
                    for (int v = 0; v < VF; ++v) a[8 + ((j * VF + v) & 7)] = __builtin_fmaf(a[8 + ((j * VF + v) & 7)], x, y);
                }
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int j = 0; j < 16; ++j) s += a[j] + p[j][0] + p[j][1];
    for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[blockIdx.x * blockDim.x + tid] = s;
    if (tid == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int MODE, int VF>
void run(const char* name, int threads, int blocks_per_cu, double flop_per_iter_lane, double instr_per_iter) {
    const int iters = 20000, grid = 256 * blocks_per_cu;
    float* out; unsigned long long* clk;
    hipMalloc(&out, sizeof(float) * grid * threads);
    hipMalloc(&clk, sizeof(unsigned long long) * 2 * grid);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    kern<MODE, VF><<<grid, threads>>>(out, clk, iters);                  // warm
    hipDeviceSynchronize();
    hipEventRecord(a);
    kern<MODE, VF><<<grid, threads>>>(out, clk, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> h(2 * grid);
    hipMemcpy(h.data(), clk, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost);
    std::vector<double> cyc, ghz;
    for (int i = 0; i < grid; ++i) { cyc.push_back((double)h[2 * i]); ghz.push_back((double)h[2 * i] / ((double)h[2 * i + 1] * 10.0) ); }
    std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
    const double c = cyc[grid / 2], g = ghz[grid / 2];
    const double waves_per_simd = threads / 64.0 * blocks_per_cu / 4.0;
    const double tf = flop_per_iter_lane * iters * (double)grid * threads / (ms * 1e-3) / 1e12;
    printf("%-34s thr %4d x%d/CU (%4.1f w/SIMD)  %8.3f ms  clk %.2f GHz  %6.1f TF   cycles/instr/SIMD %.2f\n",
           name, threads, blocks_per_cu, waves_per_simd, ms, g, tf, c / (instr_per_iter * iters * waves_per_simd));
    hipFree(out); hipFree(clk);
}

int main() {
    for (int bpc : {1, 2, 4}) {
        run<0, 0>("v_fma_f32 x16", 256, bpc, 32, 16);
        run<1, 0>("v_pk_fma_f32 x16", 256, bpc, 64, 16);
        run<2, 0>("v_min/min3 x16", 256, bpc, 0, 16);
    }
    run<0, 0>("v_fma_f32 x16", 512, 4, 32, 16);
    run<3, 0>("mfma16x16x4 x8", 256, 1, 8 * 2048 / 64.0, 8);
    run<3, 0>("mfma16x16x4 x8", 256, 2, 8 * 2048 / 64.0, 8);
    run<3, 0>("mfma16x16x4 x8", 256, 4, 8 * 2048 / 64.0, 8);
    run<4, 2>("mfma + 2 fma (same wave)", 256, 2, 8 * 2048 / 64.0 + 8 * 2 * 2, 8 * 3);
    run<4, 4>("mfma + 4 fma (same wave)", 256, 2, 8 * 2048 / 64.0 + 8 * 4 * 2, 8 * 5);
    run<4, 8>("mfma + 8 fma (same wave)", 256, 2, 8 * 2048 / 64.0 + 8 * 8 * 2, 8 * 9);
    run<4, 12>("mfma + 12 fma (same wave)", 256, 2, 8 * 2048 / 64.0 + 8 * 12 * 2, 8 * 13);
    run<4, 16>("mfma + 16 fma (same wave)", 256, 2, 8 * 2048 / 64.0 + 8 * 16 * 2, 8 * 17);
    run<5, 0>("mfma waves | fma waves (512thr)", 512, 1, (8 * 2048 / 64.0 + 32) / 2, 12);
    run<5, 0>("mfma waves | fma waves (512thr)", 512, 2, (8 * 2048 / 64.0 + 32) / 2, 12);
    return 0;
}
