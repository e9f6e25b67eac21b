// Micro-benchmark (MI355X box): what a per-cloud iteration barrier costs inside ONE persistent launch.
//
// The candidate design for the certified iterations of the ICP loop keeps a cloud's points in registers over many
// iterations; per iteration the cloud's G blocks each publish a 32-float partial sum, the block that arrives last
// reduces them, takes the cloud's step (a serial chain of f64 work by one lane) and publishes the new pose, the others wait.
// Hand-off form (MI355X_MICROARCH.md, "Workgroup dispatch ... inter-workgroup visibility", first row of the sc1 table):
// payload by sc1 (write-through) stores of whole 128-byte lines, the storing wave's s_waitcnt vmcnt(0), ONE agent-scope
// returning atomic add per block; the last arriver (told by the returned value) reads the partials with sc1 loads;
// the pose goes back the same way behind an sc1 flag that the others poll with sc1 loads + s_sleep (bounded spin).
// Measured: microseconds per iteration for G = 16, 1..5 blocks per CU, with a light or heavier compute phase, with and
// without a 16-byte-per-thread streaming store per iteration (the weight history), stale-data detection on every word.
// Also: what an EMPTY launch of a big grid costs back to back (the price of always-enqueued fallback launches).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int BLOCK = 256, WAVE = 64, G = 16, SPIN_MAX = 1 << 22;

__device__ __forceinline__ float ld_sc1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int ld_sc1(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// partial[(group, blk)][32], pose[group][16], counter[group] (own 128-byte line), flag[group] (own line), err[0..3]
template <int WORK, bool STREAM>
__global__ __launch_bounds__(BLOCK, 4) void run_kernel(float* __restrict__ partial, float* __restrict__ pose, int* __restrict__ counter, int* __restrict__ flag,
                                                       int* __restrict__ err, float* __restrict__ stream, int iters, int groups) {
    __shared__ float red[4][32];
    __shared__ float spose[16];
    __shared__ int s_last;
    const int b = blockIdx.x, i = b >> 3;
    const int group = (i / G) * 8 + (b & 7), blk = i % G;      // all blocks of a group share blockIdx % 8 (one XCD under round-robin dispatch)
    if (group >= groups) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float st[36];                                               // the "points" a thread keeps over all iterations
#pragma unroll
    for (int k = 0; k < 36; ++k) st[k] = 1e-3f * (float)((tid * 36 + k + blk * 977 + group * 31) % 1013);
    float pz[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) pz[k] = 0.f;
    for (int it = 0; it < iters; ++it) {
        // compute phase: WORK rounds of 36 fmas per thread against the current pose, 32 sums per thread
        float acc[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) acc[k] = 0.f;
#pragma unroll 1
        for (int w = 0; w < WORK; ++w) {
#pragma unroll
            for (int k = 0; k < 36; ++k) acc[k & 31] = __builtin_fmaf(st[k], pz[k % 12] + 1.0f, acc[k & 31]);
        }
        if (STREAM) {       // the weight history: 4 floats per thread per iteration, plain coalesced stores, never read back in the kernel
            float4* o = reinterpret_cast<float4*>(stream) + ((size_t)it * gridDim.x + b) * BLOCK + tid;
            *o = make_float4(acc[0], acc[1], acc[2], acc[3]);
        }
        // block reduce (plain butterflies: the real kernel has its own cheaper form), slot k -> lane k of wave 0
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            float v = acc[k];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
            if (lane == 0) red[wave][k] = v;
        }
        __syncthreads();
        if (wave == 0) {
            float* mine = partial + ((size_t)group * G + blk) * 32;
            if (lane < 32) {
                float v = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
                if (lane == 31) v = (float)(it + 1);            // tag: which iteration this record belongs to (stale detection)
                st_sc1(mine + lane, v);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            int last = 0;
            if (lane == 0) {
                const int old = __hip_atomic_fetch_add(counter + group * 32, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = (old == G * (it + 1) - 1) ? 1 : 0;
            }
            last = __shfl(last, 0);
            if (last) {
                // the cloud's step: read all G records (sc1), fixed-order f64 sums, a serial f64 chain by one lane, publish
                double s = 0.0;
                int bad = 0;
                const float* rec = partial + (size_t)group * G * 32;
                if (lane < 32) {
                    float v[G];
#pragma unroll
                    for (int g = 0; g < G; ++g) v[g] = ld_sc1(rec + g * 32 + lane);
#pragma unroll
                    for (int g = 0; g < G; ++g) { s += (double)v[g]; if (lane == 31 && v[g] != (float)(it + 1)) ++bad; }
                }
                if (bad) atomicAdd(err + 0, bad);
                double chain = s;
                if (lane == 0) {                                // ~ a 6x6 pivoted solve + Rodrigues: a few hundred dependent f64 operations
#pragma unroll 1
                    for (int k = 0; k < 300; ++k) chain = chain * 0.999 + 1.0 / (1.0 + chain * chain);
                }
                chain = __shfl(chain, 0);
                if (lane < 16) st_sc1(pose + (size_t)group * 32 + lane, lane < 12 ? (float)(1e-6 * chain) + (float)(it + 1) * 1e-3f : (float)(it + 1));
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) st_sc1(flag + group * 32, it + 1);
            } else if (lane == 0) {
                int spins = 0;
                while (ld_sc1(flag + group * 32) < it + 1) {
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > SPIN_MAX) { atomicAdd(err + 1, 1); break; }      // every wave reaches an exit
                }
            }
            // the polling / publishing wave loads the pose behind its own poll
            if (lane < 16) spose[lane] = ld_sc1(pose + (size_t)group * 32 + lane);
        }
        __syncthreads();
        if (spose[15] != (float)(it + 1) && tid == 0) atomicAdd(err + 2, 1);       // a stale pose record
#pragma unroll
        for (int k = 0; k < 12; ++k) pz[k] = spose[k];
        __syncthreads();
    }
    if (tid == 0 && pz[0] == 12345.f) err[3] = 1;               // (keeps the state live)
    float keep = 0.f;
#pragma unroll
    for (int k = 0; k < 36; ++k) keep += st[k];
    if (keep == 12345.f) err[3] = 2;
}

__global__ void empty_kernel(const int* __restrict__ flag, int* out) {
    if (flag[blockIdx.x & 255] == 12345) out[0] = 1;
}

template <int WORK, bool STREAM>
static void run(const char* name, int blocks_per_cu, int iters) {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, run_kernel<WORK, STREAM>, BLOCK, 0));
    if (blocks_per_cu > occ) { printf("%-34s %d blocks/CU: occupancy API says %d -- skipped\n", name, blocks_per_cu, occ); return; }
    const int groups = (cus * blocks_per_cu / G) / 8 * 8;      // whole groups, a multiple of 8 (the XCD mapping)
    const int grid = groups * G;
    float *partial, *pose, *stream = nullptr;
    int *counter, *flag, *err;
    CK(hipMalloc(&partial, (size_t)groups * G * 32 * 4)); CK(hipMalloc(&pose, (size_t)groups * 32 * 4));
    CK(hipMalloc(&counter, (size_t)groups * 32 * 4)); CK(hipMalloc(&flag, (size_t)groups * 32 * 4)); CK(hipMalloc(&err, 16));
    if (STREAM) CK(hipMalloc(&stream, (size_t)iters * grid * BLOCK * 16));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    std::vector<float> ms;
    int herr[4] = {0, 0, 0, 0};
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipMemset(counter, 0, (size_t)groups * 32 * 4)); CK(hipMemset(flag, 0, (size_t)groups * 32 * 4)); CK(hipMemset(err, 0, 16));
        CK(hipMemset(pose, 0, (size_t)groups * 32 * 4));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        run_kernel<WORK, STREAM><<<grid, BLOCK>>>(partial, pose, counter, flag, err, stream, iters, groups);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float t; CK(hipEventElapsedTime(&t, a, b));
        if (rep) ms.push_back(t);
        int e[4]; CK(hipMemcpy(e, err, 16, hipMemcpyDeviceToHost));
        for (int k = 0; k < 4; ++k) herr[k] += e[k];
    }
    std::sort(ms.begin(), ms.end());
    printf("%-34s %d blocks/CU (API max %d), %4d groups x %d blocks: %7.2f us per iteration (median of 3, %d iterations)  stale records %d, spin timeouts %d, stale poses %d\n",
           name, blocks_per_cu, occ, groups, G, ms[1] * 1e3 / iters, iters, herr[0], herr[1], herr[2]);
    fflush(stdout);
    CK(hipFree(partial)); CK(hipFree(pose)); CK(hipFree(counter)); CK(hipFree(flag)); CK(hipFree(err));
    if (stream) CK(hipFree(stream));
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s, %d CUs\n", prop.name, prop.multiProcessorCount);
    const int iters = 64;
    for (int bpc = 1; bpc <= 4; bpc *= 2) run<1, false>("light phase (36 fma/thread)", bpc, iters);
    for (int bpc = 1; bpc <= 4; bpc *= 2) run<16, false>("16 x 36 fma/thread", bpc, iters);
    for (int bpc = 1; bpc <= 4; bpc *= 2) run<16, true>("16 x 36 fma + 16 B/thread store", bpc, iters);
    run<48, true>("48 x 36 fma + 16 B/thread store", 4, iters);
    // empty launches back to back
    int *flag, *out;
    CK(hipMalloc(&flag, 1024)); CK(hipMalloc(&out, 4)); CK(hipMemset(flag, 0, 1024));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int grid : {256, 4096, 8192}) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(a));
            for (int k = 0; k < 45; ++k) empty_kernel<<<grid, BLOCK>>>(flag, out);
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            float t; CK(hipEventElapsedTime(&t, a, b));
            if (rep) printf("empty launch, grid %5d x 256: %.2f us each (45 back to back)\n", grid, t * 1e3 / 45);
        }
    }
    return 0;
}
