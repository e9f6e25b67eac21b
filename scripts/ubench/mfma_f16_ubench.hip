// Micro-benchmark / probe behind the split-f16 matrix-core search (round 4): what v_mfma_f32_32x32x16_f16 does with
// the operands the filter kernel feeds it, and what its VALU epilogue costs.
//   1. layout:   A[row = lane&31][k = 8(lane>>5)+j], B[k][col = lane&31], D[row = (reg&3)+8(reg>>2)+4(lane>>5)][col = lane&31]
//                checked with exact integer data and an asymmetric B
//   2. denormals: are f16 denormal operands honoured (v_cvt_f16_f32 producing them, the MFMA consuming them)?
//   3. accumulation: error of the 16-term dot product against the exact one, in units of u * sum|a_k b_k| (u = 2^-24),
//                over random operands with wide dynamic range and heavy cancellation -- the constant of the filter's bound
//   4. rate:     the scoring loop itself (A tile from LDS, four B tiles in registers, per-lane chunk minimum + top-2
//                bookkeeping per chunk of 1 / 2 / 4 MFMAs) at 2 / 3 / 4 waves per SIMD
// Build: hipcc -O3 --offload-arch=gfx950 -o mfma_f16_ubench mfma_f16_ubench.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// ------------------------------------------------------------------ 1-3: one MFMA on given operands
__global__ void one_mfma(const _Float16* A /* 32 x 16 */, const _Float16* B /* 16 x 32 */, float* D /* 32 x 32 */) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    half8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = A[r * 16 + 8 * h + j]; b[j] = B[(8 * h + j) * 32 + r]; }
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int i = 0; i < 16; ++i) D[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = c[i];
}
__global__ void cvt_probe(const float* in, _Float16* out, int n) {
    const int i = threadIdx.x;
    if (i < n) out[i] = (_Float16)in[i];
}

// ------------------------------------------------------------------ 4: the scoring loop
// CHT = MFMAs (A tiles) per chunk and B tile; the per-chunk bookkeeping is: c = chunk minimum; b2 = med3(b1, c, b2);
// id = c < b1 ? chunk : id; b1 = min(b1, c)
template <int CHT, int MINW, int BOOK, int BATCH>
__global__ __launch_bounds__(256, MINW) void rate_kernel(const uint4* __restrict__ img, int ntiles, float4* __restrict__ out, const float* __restrict__ q) {
    constexpr int STAGE = 16;                      // A tiles (1 KiB each) per LDS stage
    __shared__ uint4 lds[STAGE * 64];
    const int tid = threadIdx.x, lane = tid & 63;
    half8 b[4];
    float b1[4], b2[4];
    int id[4];
    for (int g = 0; g < 4; ++g) {
        for (int j = 0; j < 8; ++j) b[g][j] = (_Float16)q[(tid * 8 + j + g * 1031 + blockIdx.x * 17) % 4096];
        b1[g] = b2[g] = __builtin_huge_valf();
        id[g] = 0;
    }
    f32x16 zero;
    for (int i = 0; i < 16; ++i) zero[i] = 0.f;
    for (int base = 0; base < ntiles; base += STAGE) {
        for (int t = tid; t < STAGE * 64; t += 256) lds[t] = img[(size_t)base * 64 + t];
        __syncthreads();
#pragma unroll 1
        for (int t = 0; t < STAGE; t += CHT) {
            half8 a[CHT];
#pragma unroll
            for (int c = 0; c < CHT; ++c) { const uint4 v = lds[(t + c) * 64 + lane]; __builtin_memcpy(&a[c], &v, 16); }
            if (BATCH) {
                // all four B tiles' MFMAs of an A tile first (four accumulators in flight), their minima after
                float cm[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) cm[g] = __builtin_huge_valf();
#pragma unroll
                for (int c = 0; c < CHT; ++c) {
                    f32x16 d[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) d[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[c], b[g], zero, 0, 0, 0);
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int i = 0; i < 16; i += 2) cm[g] = __builtin_fminf(__builtin_fminf(cm[g], d[g][i]), d[g][i + 1]);
                    // the order asked of the scheduler: two MFMAs ahead, then each accumulator's minima beside the next MFMA
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (BOOK) {
                        b2[g] = __builtin_amdgcn_fmed3f(b1[g], cm[g], b2[g]);
                        id[g] = cm[g] < b1[g] ? base + t : id[g];
                    }
                    b1[g] = __builtin_fminf(b1[g], cm[g]);
                }
            } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float cm = __builtin_huge_valf();
#pragma unroll
                for (int c = 0; c < CHT; ++c) {
                    const f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[c], b[g], zero, 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 16; i += 2) cm = __builtin_fminf(__builtin_fminf(cm, d[i]), d[i + 1]);
                }
                if (BOOK) {
                    b2[g] = __builtin_amdgcn_fmed3f(b1[g], cm, b2[g]);
                    id[g] = cm < b1[g] ? base + t : id[g];
                }
                b1[g] = __builtin_fminf(b1[g], cm);
            }
            }
        }
        __syncthreads();
    }
    for (int g = 0; g < 4; ++g) out[(blockIdx.x * 4 + g) * 256 + tid] = make_float4(b1[g], b2[g], (float)id[g], 0.f);
}

template <int CHT, int MINW, int BOOK, int BATCH>
void run_rate(const char* name, const uint4* img, int ntiles, const float* q, int blocks) {
    float4* out;
    CHECK(hipMalloc(&out, sizeof(float4) * blocks * 4 * 256));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    rate_kernel<CHT, MINW, BOOK, BATCH><<<blocks, 256>>>(img, ntiles, out, q);
    CHECK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        rate_kernel<CHT, MINW, BOOK, BATCH><<<blocks, 256>>>(img, ntiles, out, q);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    const double mfmas = (double)blocks * 4 /*waves*/ * 4 /*B tiles*/ * ntiles;
    const double pairs = mfmas * 1024.0;
    printf("%-28s batch %d chunk %d MFMA, minw %d: %8.3f ms  %6.2f Tpairs/s  %5.1f cycles/MFMA/SIMD at 2.4 GHz   (6.87e10 pairs = %.3f ms)\n",
           name, BATCH, CHT, MINW, best, pairs / (best * 1e-3) / 1e12, best * 1e-3 * 2.4e9 * 1024.0 / mfmas, 6.87e10 / (pairs / best));
    CHECK(hipFree(out));
}


// ------------------------------------------------------------------ 5: issue rate of the candidate epilogue instructions
// 16 independent chains per lane; OP: 0 v_fma_f32, 1 v_min_f32, 2 v_min3_f32, 3 v_med3_f32, 4 v_min_u32, 5 v_min3_u32, 6 v_pk_min_f16, 7 v_cmp_lt_f32 + v_cndmask_b32,
// 8 v_max3_f32, 9 v_min_i32, 10 v_pk_add_f32 (one packed op = two values)
template <int OP>
__global__ __launch_bounds__(256) void valu_rate_kernel(float* out, int iters) {
    float a[16], x = 1.0f + threadIdx.x * 1e-6f, y = 0.999f;
    for (int j = 0; j < 16; ++j) a[j] = threadIdx.x * 0.37f + j;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(x), "v"(y));
            if (OP == 1) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[j]) : "v"(x));
            if (OP == 2) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(x), "v"(y));
            if (OP == 3) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(x), "v"(y));
            if (OP == 4) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[j]) : "v"(x));
            if (OP == 5) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(x), "v"(y));
            if (OP == 6) asm volatile("v_pk_min_f16 %0, %0, %1" : "+v"(a[j]) : "v"(x));
            if (OP == 7) asm volatile("v_cmp_lt_f32 vcc, %1, %0\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[j]) : "v"(x), "v"(y) : "vcc");
            if (OP == 8) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(x), "v"(y));
            if (OP == 9) asm volatile("v_min_i32 %0, %0, %1" : "+v"(a[j]) : "v"(x));
        }
    }
    float s = 0.f;
    for (int j = 0; j < 16; ++j) s += a[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP>
void run_valu(const char* name) {
    float* out;
    const int iters = 4000;
    for (int bpcu : {1, 2, 4, 8}) {
        const int blocks = 256 * bpcu;
        CHECK(hipMalloc(&out, sizeof(float) * blocks * 256));
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        valu_rate_kernel<OP><<<blocks, 256>>>(out, iters);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        valu_rate_kernel<OP><<<blocks, 256>>>(out, iters);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double instr_per_simd = (double)iters * 16 * bpcu;          // wave-instructions issued on one SIMD (bpcu waves per SIMD)
        printf("   %-28s %d wave/SIMD: %7.3f ms  %5.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", name, bpcu, ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
        CHECK(hipFree(out));
    }
}

static float h2f(_Float16 h) { return (float)h; }

int main() {
    std::mt19937_64 rng(1234);
    _Float16 *dA, *dB; float* dD;
    CHECK(hipMalloc(&dA, 32 * 16 * 2)); CHECK(hipMalloc(&dB, 16 * 32 * 2)); CHECK(hipMalloc(&dD, 32 * 32 * 4));
    std::vector<_Float16> A(512), B(512);
    std::vector<float> D(1024);
    auto mfma = [&]() {
        CHECK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice));
        one_mfma<<<1, 64>>>(dA, dB, dD);
        CHECK(hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost));
    };
    // 1. layout (exact small integers, asymmetric)
    for (int r = 0; r < 32; ++r) for (int k = 0; k < 16; ++k) A[r * 16 + k] = (_Float16)(float)((r * 7 + k * 3) % 11 - 5);
    for (int k = 0; k < 16; ++k) for (int c = 0; c < 32; ++c) B[k * 32 + c] = (_Float16)(float)((k * 5 + c * 2 + (c > 20)) % 13 - 6);
    mfma();
    int bad = 0;
    for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) {
        double s = 0; for (int k = 0; k < 16; ++k) s += (double)h2f(A[r * 16 + k]) * h2f(B[k * 32 + c]);
        bad += (D[r * 32 + c] != (float)s);
    }
    printf("1. layout check: %d of 1024 entries differ (0 = the operand/result maps are as documented)\n", bad);

    // 2. denormals
    {
        float in[4] = {3.0e-6f, 6.0e-8f, 2.9e-8f, 1.0e-5f}; float* din; _Float16* dout; _Float16 o[4];
        CHECK(hipMalloc(&din, 16)); CHECK(hipMalloc(&dout, 8));
        CHECK(hipMemcpy(din, in, 16, hipMemcpyHostToDevice));
        cvt_probe<<<1, 64>>>(din, dout, 4);
        CHECK(hipMemcpy(o, dout, 8, hipMemcpyDeviceToHost));
        printf("2. v_cvt_f16_f32 of 3e-6, 6e-8, 2.9e-8, 1e-5 -> %.4g %.4g %.4g %.4g (f16 denormals are multiples of 5.96e-8)\n", h2f(o[0]), h2f(o[1]), h2f(o[2]), h2f(o[3]));
        for (auto& v : A) v = (_Float16)0.f;
        for (auto& v : B) v = (_Float16)0.f;
        unsigned short bits = 0x0011; _Float16 den; memcpy(&den, &bits, 2);      // 17 * 2^-24
        A[0] = den; B[0] = (_Float16)1024.f;                                      // D[0][0] = 17 * 2^-14 if honoured
        A[16 + 1] = (_Float16)1024.f; B[32 + 1] = den;                            // D[1][1]: denormal on the B side
        mfma();
        printf("   MFMA with a denormal operand: A-side %.6g, B-side %.6g (honoured: %.6g)\n", D[0], D[33], 17.0 / 16384.0);
    }

    // 3. accumulation error
    {
        double worst_sum = 0, worst_res = 0;
        std::uniform_real_distribution<double> U(-1, 1);
        for (int trial = 0; trial < 400; ++trial) {
            const int mode = trial % 4;
            for (int r = 0; r < 32; ++r) for (int k = 0; k < 16; ++k) {
                double v = U(rng) * std::ldexp(1.0, mode == 0 ? 0 : (int)(U(rng) * 12));
                if (mode == 2 && (k & 1)) v = -(double)h2f(A[r * 16 + k - 1]) * (1 + 1e-3 * U(rng));        // cancellation
                A[r * 16 + k] = (_Float16)(float)v;
            }
            for (int k = 0; k < 16; ++k) for (int c = 0; c < 32; ++c) {
                double v = U(rng) * std::ldexp(1.0, mode == 0 ? 0 : (int)(U(rng) * 12));
                if (mode == 2 && (k & 1)) v = (double)h2f(B[(k - 1) * 32 + c]);
                if (mode == 3) v = (k < 12) ? v : 8192.0;                                                      // like the 0.5|y|^2 slots
                B[k * 32 + c] = (_Float16)(float)v;
            }
            mfma();
            for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) {
                double s = 0, sa = 0;
                for (int k = 0; k < 16; ++k) { const double p = (double)h2f(A[r * 16 + k]) * h2f(B[k * 32 + c]); s += p; sa += std::fabs(p); }
                if (!(sa > 0) || !std::isfinite(D[r * 32 + c])) continue;
                const double err = std::fabs((double)D[r * 32 + c] - s);
                worst_sum = std::max(worst_sum, err / (sa * 5.9604644775390625e-8));
                if (std::fabs(s) > 0) worst_res = std::max(worst_res, err / (std::fabs(s) * 5.9604644775390625e-8));
            }
        }
        printf("3. accumulation: max |mfma - exact| = %.3f u * sum|a_k b_k|   (%.3g u * |result|), u = 2^-24, 400 x 1024 dot products\n", worst_sum, worst_res);
    }

    // 4. rate
    {
        const int ntiles = 512, blocks = 256 * 16;          // 16384 targets; 16 blocks of 512 queries per CU
        std::vector<uint4> img((size_t)ntiles * 64);
        std::uniform_int_distribution<unsigned> R(0, 0xffffffffu);
        for (auto& v : img) {                                // random finite halves (exponent field < 0x1f)
            unsigned w[4]; for (int i = 0; i < 4; ++i) { unsigned x = R(rng); x &= 0xbbffbbffu; w[i] = x; }
            v = make_uint4(w[0], w[1], w[2], w[3]);
        }
        std::vector<float> q(4096 + 8);
        for (auto& v : q) v = (float)((double)(rng() % 2000) / 100.0 - 10.0);
        uint4* dimg; float* dq;
        CHECK(hipMalloc(&dimg, img.size() * 16)); CHECK(hipMalloc(&dq, q.size() * 4));
        CHECK(hipMemcpy(dimg, img.data(), img.size() * 16, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dq, q.data(), q.size() * 4, hipMemcpyHostToDevice));
        printf("4. scoring loop: %d blocks x 4 waves x 4 B tiles x %d A tiles (= 256 clouds of 16384 x 16384 pairs = 6.87e10 pairs)\n", blocks, ntiles);
        run_rate<1, 2, 0, 0>("min only", dimg, ntiles, dq, blocks);
        run_rate<1, 2, 1, 0>("min + top-2 + chunk id", dimg, ntiles, dq, blocks);
        run_rate<2, 2, 1, 0>("min + top-2 + chunk id", dimg, ntiles, dq, blocks);
        run_rate<4, 2, 1, 0>("min + top-2 + chunk id", dimg, ntiles, dq, blocks);
        run_rate<1, 4, 1, 0>("min + top-2 + chunk id", dimg, ntiles, dq, blocks);
        run_rate<2, 4, 1, 0>("min + top-2 + chunk id", dimg, ntiles, dq, blocks);
        run_rate<4, 4, 1, 0>("min + top-2 + chunk id", dimg, ntiles, dq, blocks);
        run_rate<1, 6, 1, 0>("min + top-2 + chunk id", dimg, ntiles, dq, blocks);
        run_rate<2, 6, 1, 0>("min + top-2 + chunk id", dimg, ntiles, dq, blocks);
        run_rate<4, 6, 1, 0>("min + top-2 + chunk id", dimg, ntiles, dq, blocks);
        run_rate<1, 8, 1, 0>("min + top-2 + chunk id", dimg, ntiles, dq, blocks);
        run_rate<2, 8, 1, 0>("min + top-2 + chunk id", dimg, ntiles, dq, blocks);
        run_rate<4, 8, 1, 0>("min + top-2 + chunk id", dimg, ntiles, dq, blocks);
        run_rate<1, 2, 1, 1>("min + top-2 + chunk id", dimg, ntiles, dq, blocks);
        run_rate<2, 2, 1, 1>("min + top-2 + chunk id", dimg, ntiles, dq, blocks);
        run_rate<4, 2, 1, 1>("min + top-2 + chunk id", dimg, ntiles, dq, blocks);
        run_rate<1, 4, 1, 1>("min + top-2 + chunk id", dimg, ntiles, dq, blocks);
        run_rate<2, 4, 1, 1>("min + top-2 + chunk id", dimg, ntiles, dq, blocks);
        run_rate<4, 4, 1, 1>("min + top-2 + chunk id", dimg, ntiles, dq, blocks);
        run_rate<1, 6, 1, 1>("min + top-2 + chunk id", dimg, ntiles, dq, blocks);
        run_rate<2, 6, 1, 1>("min + top-2 + chunk id", dimg, ntiles, dq, blocks);
        run_rate<4, 6, 1, 1>("min + top-2 + chunk id", dimg, ntiles, dq, blocks);
        run_rate<1, 8, 1, 1>("min + top-2 + chunk id", dimg, ntiles, dq, blocks);
    }
    printf("5. issue rate of single instructions (16 independent chains per lane)\n");
    run_valu<0>("v_fma_f32"); run_valu<1>("v_min_f32"); run_valu<2>("v_min3_f32"); run_valu<3>("v_med3_f32"); run_valu<4>("v_min_u32");
    run_valu<5>("v_min3_u32"); run_valu<6>("v_pk_min_f16"); run_valu<7>("v_cmp_lt_f32 + v_cndmask"); run_valu<8>("v_max3_f32"); run_valu<9>("v_min_i32");
    return 0;
}
