// Does vector work overlap the f16 matrix pipe?  One loop iteration = one v_mfma_f32_32x32x16_f16 + V vector instructions, either independent of the
// MFMA's result (IND) or reading it (v_min3 chain over the 16 result registers).  Cycles per iteration per SIMD from s_memtime (real shader cycles).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int V, int MODE>    // MODE 0: V independent v_fma; 1: V independent v_min3; 2: v_min3 chain on the PREVIOUS MFMA's result (8 ops) + (V-8) independent min3
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* clk, int iters) {
    const int tid = threadIdx.x;
    half8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(float)(tid % 7 + j); b[j] = (_Float16)(float)(tid % 5 - j); }
    f32x16 d0, d1;
    for (int i = 0; i < 16; ++i) { d0[i] = 0.f; d1[i] = 0.f; }
    float x[16], y = 1.0001f, z = 0.5f, cm = 1e30f;
    for (int j = 0; j < 16; ++j) x[j] = tid * 0.01f + j;
    f32x16 zero; for (int i = 0; i < 16; ++i) zero[i] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x16& dst = h ? d1 : d0;
            f32x16& src = h ? d0 : d1;
            dst = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, zero, 0, 0, 0);
            if (MODE == 2) {
#pragma unroll
                for (int i = 0; i < 16; i += 2) cm = __builtin_fminf(__builtin_fminf(cm, src[i]), src[i + 1]);
#pragma unroll
                for (int j = 0; j < V - 8; ++j) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(x[j & 15]) : "v"(y), "v"(z));
            } else {
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j & 15]) : "v"(y), "v"(z));
                    else           asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(x[j & 15]) : "v"(y), "v"(z));
                }
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, V > 0 ? V : 1, 0);
            a[0] = (_Float16)(float)((it + h) & 3);     // (keeps the MFMAs from being hoisted / merged)
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = cm;
    for (int j = 0; j < 16; ++j) s += x[j] + d0[j] + d1[j];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) clk[blockIdx.x] = t1 - t0;
}

template <int V, int MODE>
void run(const char* name) {
    const int iters = 20000;
    for (int bpcu : {1, 2, 4}) {
        const int grid = 256 * bpcu;
        float* out; unsigned long long* clk;
        hipMalloc(&out, sizeof(float) * grid * 256); hipMalloc(&clk, 8 * grid);
        k<V, MODE><<<grid, 256>>>(out, clk, iters); hipDeviceSynchronize();
        k<V, MODE><<<grid, 256>>>(out, clk, iters); hipDeviceSynchronize();
        std::vector<unsigned long long> h(grid);
        hipMemcpy(h.data(), clk, 8 * grid, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        printf("%-46s V=%2d  %d wave/SIMD: %6.1f cycles per (MFMA + V ops) per SIMD\n", name, V, bpcu, (double)h[grid / 2] / iters / bpcu);
        hipFree(out); hipFree(clk);
    }
}
int main() {
    run<0, 0>("MFMA alone");
    run<4, 0>("MFMA + independent v_fma"); run<8, 0>("MFMA + independent v_fma"); run<12, 0>("MFMA + independent v_fma"); run<16, 0>("MFMA + independent v_fma");
    run<4, 1>("MFMA + independent v_min3"); run<8, 1>("MFMA + independent v_min3"); run<12, 1>("MFMA + independent v_min3");
    run<8, 2>("MFMA + v_min3 chain over the previous result"); run<12, 2>("MFMA + v_min3 chain over the previous result (+4 independent)");
    return 0;
}
