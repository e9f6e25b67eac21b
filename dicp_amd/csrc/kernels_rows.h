// libdicp_hip.so -- row-indexed copies: gather / scatter-add / permute (nn.find_nn, sorted copies and their undoing).
// Part of the one translation unit dicp_kernels.hip (included inside its anonymous namespace, in this order: kernels_setup.h, kernels_search.h, kernels_setup_sort.h, kernels_rows.h, kernels_accumulate.h, kernels_backward.h, kernels_soft_svd.h, kernels_host.h).
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

// ------------------------------------------------------------- lists of clouds <-> one padded batch
// ICP.py:305-511 (batch_size_handling) pads a LIST of clouds to the longest, one torch op per cloud: 256 clouds are ~800 copy / fill launches going in and -- autograd
// through them -- ~2300 coming back: 20 ms of a 40 ms call on the hard benchmark clouds (profiles/r05_ragged_lists.txt).  Here the list is a table of pointers:
// pack: out[b][i][k] = rows_b[i][k] for i < lens[b] (k < cols of the row's first `cols`), else *pad (NULL: zero); unpack (the adjoint): grads_b[i][k] = gout[b][i][k]
// for k < cols, zero in the row's further columns.  One thread per element of the padded batch.
template <typename T>
__global__ __launch_bounds__(BLOCK) void pack_list_kernel(const T* const* __restrict__ ptrs, const int32_t* __restrict__ lens, const int32_t* __restrict__ strides,
                                                          int N, int n_max, int cols, int bpc, T* __restrict__ out, const T* __restrict__ pad) {
    int b, blk;
    if (!decode_block(bpc, N, b, blk)) return;
    const unsigned e = (unsigned)blk * BLOCK + threadIdx.x;
    if (e >= (unsigned)n_max * (unsigned)cols) return;
    const int i = (int)(e / (unsigned)cols), k = (int)(e - (unsigned)i * cols);
    T v = pad ? *pad : T(0);
    if (i < lens[b]) v = ptrs[b][(size_t)i * strides[b] + k];
    out[(size_t)b * n_max * cols + e] = v;
}
template <typename T>
__global__ __launch_bounds__(BLOCK) void unpack_list_kernel(const T* __restrict__ gout, T* const* __restrict__ ptrs, const int32_t* __restrict__ lens,
                                                            const int32_t* __restrict__ strides, int N, int n_max, int cols, int bpc) {
    int b, blk;
    if (!decode_block(bpc, N, b, blk)) return;
    const int stride = strides[b];                          // (elements per row of the cloud's gradient: its own column count)
    const unsigned e = (unsigned)blk * BLOCK + threadIdx.x;
    if (e >= (unsigned)n_max * (unsigned)stride) return;
    const int i = (int)(e / (unsigned)stride), k = (int)(e - (unsigned)i * stride);
    if (i < lens[b]) ptrs[b][(size_t)i * stride + k] = k < cols ? gout[((size_t)b * n_max + i) * cols + k] : T(0);
}
