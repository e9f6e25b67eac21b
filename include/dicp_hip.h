/* dicp_hip.h — C ABI of libdicp_hip.so: the MI355X (gfx950) hot path of differentiable ICP.
 *
 * The reference (utiasASRL/dICP) has no FFI: its boundary is the Python call surface and
 * the seam the kernels sit behind is inside dICP/ICP.py's loop.  Each entry point below
 * cites the reference lines it replaces (paths relative to /root/reference).  The Python
 * host side (dicp_amd/_lib.py) binds these with ctypes; INTEGRATION.md shows the stub a
 * reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch-ROCm allocations);
 *     the library allocates nothing, keeps no state, and never synchronises;
 *   - `stream` is a hipStream_t (pass torch.cuda.current_stream().cuda_stream);
 *   - dtype: DICP_F32 or DICP_F64 selects the scalar type T of all `void*` tensors
 *     (the reference's tests run float64: tests/test_ICP.py:41-42);
 *   - tensors are dense row-major unless a stride argument says otherwise;
 *   - ragged batches (the reference pads every cloud of a list to the longest, ICP.py:305-511): `src_rows` / `tgt_rows`
 *     (N) int32, optional, are the leading rows of each cloud that take part.  NULL = all n / m of them.  No kernel reads,
 *     scores or accumulates a row beyond them; outputs for such rows are what the reference's padding yields (weight 0,
 *     zero gradient).  To reproduce the reference bit for bit pass tgt_rows = real rows + 1: its pad rows are copies of
 *     ONE far point (ICP.py:460,472-477), and one copy takes part exactly like all of them;
 *   - return value: 0 = ok, DICP_ERR_* (>0) = rejected argument (nothing launched),
 *     <0 = -(hipError_t) from the launch.  Nothing throws across the boundary.
 */
#ifndef DICP_HIP_H
#define DICP_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DICP_ABI_VERSION 11  /* 11: dicp_search_frame takes the queries (src under T_init): the sort direction is chosen for THEIR slabs; dicp_query_reorder (a re-ordering
                                keeps the order of a cloud that hardly moved); dicp_loop_buffers.search.form_plan (one scoring form per iteration from an earlier call's tallies); dicp_match_order (the deterministic backward's slot order, natively);
                                dicp_loop_buffers is ONE VERSIONED struct: `abi` first (the entry points return DICP_ERR_ABI for another layout), its ~80 fields in four sub-structs
                                (search / cert / hist / bwd); dicp_step_io.cert_slist.
                                10: dicp_pack_list / dicp_unpack_list (lists of clouds to one padded batch and back, one launch each).
                                9: dicp_loop_buffers.bwd.det_far_row / det_far_val (deterministic target gradients of the windowed backward).
                                8: certified iterations keep a row cache and their match history by reference (dicp_loop_buffers.hist.spos_of / cert_nbr / cert_gdirty /
                                cert_pend / cert_cm, spos_prev_chunk + spos_floor instead of spos_prev0; dicp_resolve_matches).
                                7: dicp_call_* (one eager call of the sweep path behind one host call per direction).
                                6: dicp_bwd_tail_max_blocks (the one-launch tail only where a cloud's blocks are all resident; a wait that runs out poisons the
                                cloud's gradients with NaN and raises bwd_live[K] next to bwd_tail_arrive[N]); bwd_live is (K + 1).
                                5: dicp_loop_buffers.bwd.tail_from (the ended iterations of the truncated reverse sweep as one launch).
                                4: dicp_loop_buffers.bwd.skip / bwd_mref / bwd_live / bwd_skip_eps (truncated reverse sweep of dicp_icp_backward); cert_cloud (per-cloud switch of the
                                match certificates); dicp_cloud_center -> dicp_search_frame: centre AND sort direction as one affine map per cloud, (N,12).
                                3: per-cloud row counts of ragged batches (src_rows / tgt_rows) on every entry point of the path; the centred and
                                uncentred forms of an entry point are one (center may be NULL); the key sort is native for every size and dtype;
                                timing events are 6 per iteration; the scan / packed / fused-accumulate search forms are gone */

enum { DICP_F32 = 0, DICP_F64 = 1 };
enum { DICP_PT2PT = 0, DICP_PT2PL = 1 };                 /* ICP(icp_type=...)  ICP.py:15,101-105 */
enum { DICP_LOSS_NONE = 0, DICP_LOSS_HUBER = 1, DICP_LOSS_CAUCHY = 2, DICP_LOSS_TRIM = 3 };
enum { DICP_KNN_AUTO = 0, DICP_KNN_VALU = 1, DICP_KNN_MFMA = 2, DICP_KNN_SWEEP = 3 /* via dicp_knn_sweep */,
       DICP_KNN_GUMBEL = 4 /* dicp_icp_forward / _backward only: soft correspondences, dicp_loop_buffers.search.gumbel */ };
enum { DICP_ERR_NULL = 1, DICP_ERR_SHAPE = 2, DICP_ERR_DTYPE = 3, DICP_ERR_ENUM = 4, DICP_ERR_ALIGN = 5, DICP_ERR_ABI = 6 /* dicp_loop_buffers.abi is not this library's */ };

/* Accumulator layout of one (cloud, block) partial: see dicp_amd/csrc/dicp_math.h */
#define DICP_NACC_PAD 32   /* 21 A-upper, 6 b, cost, sum w, #matched, 2 pad */
#define DICP_NBWD_PAD 16   /* 9 C-bar, 3 r-bar, 4 pad */

/* loss.py:4 constructor arguments + the ICP.py:152-160 call-site switches. */
typedef struct dicp_weight_params {
    int32_t mode;            /* DICP_PT2PT | DICP_PT2PL */
    int32_t trim_on;         /* trim_dist is not None and >= 0            ICP.py:153 */
    int32_t differentiable;  /* ICP(differentiable=...)                   ICP.py:38  */
    int32_t loss;            /* DICP_LOSS_NONE | _HUBER | _CAUCHY | _TRIM (loss.py:12-17)   ICP.py:158 */
    double trim_dist;
    double tanh_k;           /* config tanh_steepness                     ICP.py:119 */
    double loss_delta;       /* loss_fn["metric"]                                    */
    double match_thresh;     /* config matched_ratio_thresh               ICP.py:37  */
} dicp_weight_params;

int dicp_abi_version(void);
/* Plain device-to-device copy / zero fill of `bytes` (multiples of 4, 4-byte aligned) as KERNELS of this library, for callers whose calls may be captured into a
 * hipGraph: the runtime's memset node (hipMemsetAsync under capture) is not ordered against the kernel nodes around it when a replay starts on an idle GPU
 * (round 6, csrc/dicp_fill.h); every fill and copy inside the library goes the same way. */
int dicp_copy(void* dst, const void* src, size_t bytes, void* stream);
int dicp_zero(void* dst, size_t bytes, void* stream);

/* Rows of target points padded for the kNN kernels: returns m rounded up to 64. */
int dicp_padded_targets(int m);
/* Blocks per cloud used by dicp_accumulate / dicp_accumulate_bwd for n source points. */
int dicp_accumulate_blocks(int n);

/* The search frame.  The searches score in the expanded form 0.5|y|^2 - x.y (the reference's own: nn.py:32), whose rounding error grows
 * with 0.5|x|^2 -- and with it the prune margin of dicp_knn_sweep: a cloud a kilometre from the origin is searched almost exhaustively.
 * And the sorted sweep prunes along ONE direction: a wall perpendicular to it sits in every slab that touches it.  Both are met by one
 * affine map per cloud, x' = Q x + t:  frame (N,12) T = [Q row-major (9) | t (3)], Q a rotation whose first row is the sort direction,
 * t = -Q c.  dicp_search_frame writes it: c = the coordinate-wise median of a stride sample of at most 1024 of the cloud's rows, rounded to a
 * multiple of `quantum` (0: not rounded; float precision); Q = (directions != 0) the candidate -- identity, the two other axis orders, three
 * oblique directions -- whose projected keys spread the sample best (smallest sum of squared counts of a 256-bin histogram: proportional to
 * the pairs a slab search scores), the identity unless another candidate is 20 % better; directions == 0: Q = I.
 * Given the queries as well (src (N,n,3) with T_init (N,4,4) row-major, optional src_rows; both NULL: the target-only choice) the cost of a
 * candidate is the rows the slabs of a sample of 256 queries hold along it, a query's reach being the distance to its nearest sample target:
 * scan pairs that overlap only partly reach metres far outside the common footprint, and a direction oblique to its edge then puts a third of
 * the target into those slabs.  Points beyond 32x the sample's mean deviation from the centre (a ragged cloud's far pad row) do not stretch
 * the histograms' span.
 * Entry points given `frame` pack rows as Q y + t, and the caller hands the searches the pose [Q C | Q r + t] (dicp_loop_buffers.search.poses;
 * dicp_loop_init / the step kernels write it).  Every search form reads only (pose, packed rows): with the same frame they return the same
 * indices as each other; Q = I is applied as the plain subtraction it is, and with t == 0 too (clouds near the origin, given a quantum) the
 * results are exactly those of frame == NULL. */
int dicp_search_frame(int dtype, const void* tgt, int c, const int32_t* tgt_rows, int N, int m, double quantum, int directions,
                      const void* src, const int32_t* src_rows, int n, const void* T_init, void* frame, void* stream);

/* Once per ICP call: tgt (N,m,c) -> tgt4 (N,m_pad,4) rows [x,y,z,0.5|y|^2] (of Q y + t when frame != NULL), pad rows
 * [0,0,0,+inf].  The norms are the ||y||^2 column that torch.cdist's matmul path builds
 * on every call (nn.py:32 -> ATen _euclidean_dist).  c in {3,6}. */
int dicp_pack_target(int dtype, const void* tgt, int c, const void* frame, const int32_t* tgt_rows, int N, int m, void* tgt4, int m_pad, void* stream);

/* The matrix-core form of the brute-force search (DICP_KNN_MFMA, float32): a split-f16 FILTER on v_mfma_f32_32x32x16_f16 and an exact float32
 * REFINE with the score every other form computes -- index for index the result of DICP_KNN_VALU (csrc/knn_f16.hip has the error bound).
 * It reads, next to tgt4, an IMAGE of the packed rows in MFMA operand order: dicp_knn_f16_bytes(N, m_pad) bytes (16-byte aligned), built once
 * per call by dicp_knn_f16_pack from tgt4 (N,m_pad,4) float32 with the SAME tgt_rows the search is given.  Replaces the K = 5 bmm inside
 * torch.cdist, nn.py:32. */
size_t dicp_knn_f16_bytes(int N, int m_pad);
int dicp_knn_f16_pack(const void* tgt4, const int32_t* tgt_rows, int N, int m, int m_pad, void* image, void* stream);
/* A TEST AID, not on the path of nn.py:32: the matrix-core filter's error bound held to account.  For every (query, image row) pair of every cloud the
 * filter value exactly as DICP_KNN_MFMA / the matrix-core sweep compute it, against the float32 score of the other search forms and the bound E of
 * csrc/knn_f16.hip evaluated for that pair.  out (N,4) float32: [max over pairs of |filter - score| / E, that error, its E, pairs checked].  The searches
 * are index-identical to DICP_KNN_VALU while the first stays <= 1 (tests/test_gpu_f16.py searches for its maximum on adversarial clouds). */
int dicp_knn_f16_probe(const void* src, const void* pose, const void* tgt4, const void* f16_image, const int32_t* src_rows, const int32_t* tgt_rows,
                       int N, int n, int m, int m_pad, float* out, void* stream);

/* Fused transform + brute-force 1-NN: replaces ICP.py:137 (ps_t = C p + r) followed by
 * nn.find_nn's cdist -> argmin, nn.py:32-35 / 83-86.  Never materialises (N,n,m).
 *   src (N,n,3); pose (N,12) = [C row-major (9), r (3)] or NULL for identity;
 *   idx (N,n) int32, ties -> lowest index.  variant: DICP_KNN_* in the low byte (MFMA is f32 only and needs f16_image, NULL otherwise);
 *   bits 8..15 optionally pin a launch configuration of the VALU form (0 = chosen from the problem size). */
int dicp_knn(int dtype, const void* src, const void* pose, const void* tgt4, const int32_t* src_rows, const int32_t* tgt_rows,
             int N, int n, int m, int m_pad, int32_t* idx, int variant, const void* f16_image, void* stream);

/* Set-up of the sorted-sweep search structure, ONCE per ICP call (targets do not move between iterations).
 * dicp_sweep_sort: keys_sorted (N,m_pad) T and tperm (N,m_pad) as a STABLE ascending sort of the target x keys (first coordinate of Q y + t) gives them;
 *   the m_pad - m pad slots (and, in a ragged batch, the slots past a cloud's own rows) keep the largest key: they follow every real row,
 *   NaN rows included.  float32 clouds of up to 16384 slots: an LSD radix sort in LDS, one block per cloud; float64 keys or more slots: the
 *   same sort chunk by chunk through `scratch` (dicp_sweep_sort_scratch_bytes(dtype, N, m_pad) bytes; 0 = none needed).  Given bucket
 *   (N,nbkt+1) and brange (N,2) it also builds the search's coarse table: bucket[b] = #rows with x < xlo + b / inv, brange = [xlo, inv].
 * dicp_sweep_build: from tperm, tgs4 (N,m_pad,4) = the packed rows in sorted order (pads [max,0,0,+inf]) and, optionally, tgt_s
 *   (N,m_pad,tgt_s_stride) = the full target rows in sorted order (row s = tgt[tperm[s]], as given: not in the search frame; tgt_s_stride >= c elements per
 *   row, the rest zero: 8 for c = 6 / 4 for c = 3 makes every row ONE aligned 32- / 16-byte sector for the gathers of dicp_accumulate and
 *   dicp_accumulate_bwd_window, which take such rows with c = the stride). */
size_t dicp_sweep_sort_scratch_bytes(int dtype, int N, int m_pad);
int dicp_sweep_sort(int dtype, const void* tgt, int c, const void* frame, const int32_t* tgt_rows, int N, int m, int m_pad, void* keys_sorted,
                    int32_t* tperm, int nbkt, int32_t* bucket, void* brange, void* scratch, size_t scratch_bytes, void* stream);
int dicp_sweep_build(int dtype, const void* tgt, int c, const void* frame, const int32_t* tgt_rows, const int32_t* tperm, int N, int m, int m_pad,
                     void* tgs4, void* tgt_s, int tgt_s_stride, void* stream);
/* The whole per-call set-up of the sweep path in one call: dicp_search_frame -> dicp_sweep_sort -> dicp_sweep_build and, given T_init
 * (N,4,4), the queries src (N,n,3) and two outputs, dicp_search_pose (pose_search0 (N,12)) -> dicp_query_order (qorder0 (N,n): the query
 * order of iteration 0).  Arguments as those entry points'. */
int dicp_sweep_setup(int dtype, const void* tgt, int c, const int32_t* tgt_rows, int N, int m, int m_pad, double quantum, int directions,
                     void* frame, void* keys_sorted, int32_t* tperm, int nbkt, int32_t* bucket, void* brange, void* scratch, size_t scratch_bytes,
                     void* tgs4, void* tgt_s, int tgt_s_stride,
                     const void* src, const int32_t* src_rows, int n, const void* T_init, void* pose_search0, int32_t* qorder0, void* stream);

/* keys (N,n) = x coordinate of every source point under pose (NULL = identity): the sort key of the query order. */
int dicp_query_keys(int dtype, const void* src, const void* pose, int N, int n, void* keys, void* stream);
/* qorder (N,n) = the queries in ascending bucket of their x under pose (counting sort over equal-width buckets of the
 * target's x range [brange from dicp_sweep_build, nbkt its bucket count]; arbitrary order inside a bucket).  A cheap
 * replacement for an exact sort of dicp_query_keys: the sweep is exact for ANY query order, the order is speed only.
 * Optional: src_s (N,n,3) / w_s (N,n) = the source rows / the per-point weights w (N,n) in that slot order, for coalesced
 * loads of dicp_accumulate_bwd_window (measured: one block
 * per cloud gathers slowly, 115 vs 16 us; dicp_gather_rows does it better).  reproducible != 0: buckets of up to 64
 * members are put in index order, so the permutation is the same on every run (needed only when sums are taken in
 * this order).  spos_prev (N,n), optional: the queries' matches of an earlier iteration
 * (dicp_knn_sweep's spos): the bucket is then the match's rank among the m_pad sorted targets instead of the query's x --
 * equal-population buckets, robust against uneven density along x (but one pose late).  keys_sorted (N,m_pad; the
 * sorted keys dicp_sweep_build was given) + bucket (from dicp_sweep_build; m real targets), optional: the bucket is the RANK of the query's x among the sorted target keys under the given pose (a
 * binary search in an LDS copy of the keys; clouds of up to 16384 queries and targets, bigger ones keep the x buckets):
 * equal-population buckets without the lag -- what the ICP loop uses (0.58 vs 1.50 ms/iteration on clouds with a dense
 * blob and one far outlier, profiles/r01_uneven_clouds.txt).  Ragged batches: a cloud's own queries fill the first src_rows[b] slots, the
 * rows past them follow in index order, so qorder is always a permutation of all n rows. */
int dicp_query_order(int dtype, const void* src, const void* pose, const void* brange, int nbkt, int N, int n, int32_t* qorder,
                     const void* w, void* src_s, void* w_s, int reproducible, const int32_t* spos_prev, int m_pad,
                     const void* keys_sorted, const int32_t* bucket, int m, const int32_t* src_rows, const int32_t* tgt_rows, void* stream);
/* A RE-ordering inside one call: as dicp_query_order (no copies, no previous matches), given the order that was made under an earlier pose of the same call.
 * A cloud whose points have moved by less than a tenth of the x extent of a unit of the sweep between the two poses (|dC|_F R + |dr|, R the radius of the
 * targets' x range) keeps that order -- qorder gets a copy of order_prev -- instead of being sorted again; order_prev != qorder.  The order only keeps a wave's
 * queries neighbours in x: which clouds were sorted again changes pairs scored, never results.  dicp_icp_forward_plan re-orders this way. */
int dicp_query_reorder(int dtype, const void* src, const void* pose, const void* pose_prev, const int32_t* order_prev, const void* brange, int nbkt, int N, int n,
                       int32_t* qorder, int m_pad, const void* skeys, const int32_t* bucket, int m, const int32_t* src_rows, const int32_t* tgt_rows, void* stream);

/* Exact 1-NN with slab pruning: same result (and lowest-index tie rule) as dicp_knn, far fewer pairs.
 * The caller prepares, ONCE per ICP call (targets do not move between iterations):
 *   tgs4  (N,m_pad,4)  the rows of dicp_pack_target re-ordered by ascending x (pad rows last, x = +max);
 *   tperm (N,m_pad)    original row index of each sorted row;
 *   bucket (N,nbkt+1)  bucket[b] = #rows with x < xlo + b/inv  (lower_bound table), brange (N,2) = [xlo, inv];
 *   qorder (N,n)       optional: query indices in ascending x (under any recent pose) so that a wave's
 *                      queries are neighbours; NULL = natural order (still exact, less pruning).
 * spos (N,n), optional: spos[b][i] = SORTED position of the neighbour of query i (-1 if none; indexed like idx):
 *   what dicp_accumulate_bwd_window -- and dicp_accumulate on the sorted rows -- consume.  idx may be NULL when spos is given (the original
 *   index then costs a look-up only on exact ties).
 * pairs: optional DICP_PAIR_SHARDS device counters; their sum += number of (query,target) pairs actually scored
 *   (roofline accounting; sharded because adds to ONE address serialise at ~12 ns each).
 * cfg: 0 = launch configuration chosen from the problem size; 1, 2, 4 pin one (queries per lane, rows per chunk) = (1,8), (2,8), (1,16).
 * f16_image: optional (float32): the image dicp_knn_f16_pack made of tgs4 (the SORTED packed rows, same tgt_rows).  The scoring of the (2,8)
 *   configuration's units -- 128 queries per wave, what big problems get -- then runs on the matrix cores (split-f16 filter + exact float32
 *   refine, csrc/knn_f16.hip): the same idx / spos, index for index.
 * form_in / form_out (N) int32, optional; form_default: the scoring form per CLOUD.  The matrix-core form pays where a wave's slab is long -- big clouds, or clouds of
 *   any size whose queries lie far from their matches -- and loses where it is a few tiles.  Every unit adds the 64-row tiles its slab had to form_out[cloud].
 *   Given f16_image AND form_in (such a tally of an earlier search of the same clouds), both forms are launched and each takes its clouds: the matrix cores those
 *   with more than 20 tiles per unit in form_in (a cloud whose tally is 0: if form_default != 0), the vector form the others.  Same idx / spos either way. */
#define DICP_PAIR_SHARDS 64
int dicp_knn_sweep(int dtype, const void* src, const void* pose, const void* tgs4, const int32_t* tperm,
                   const int32_t* qorder, const int32_t* bucket, const void* brange, int nbkt, const int32_t* src_rows, const int32_t* tgt_rows,
                   int N, int n, int m, int m_pad, int32_t* idx, int32_t* spos, unsigned long long* pairs, int cfg, const void* f16_image,
                   const int32_t* form_in, int32_t* form_out, int form_default, void* stream);

/* Gather whole target rows at idx (nn.py:37-38 / 89-90) and its backward, a scatter-add
 * into a zero-initialised (N,m,c) buffer (autograd's gather backward). */
int dicp_gather_rows(int dtype, const void* tgt, const int32_t* idx, int N, int n, int m, int c, void* out, void* stream);
/* Lists of clouds <-> one padded batch (ICP.py:305-511, batch_size_handling: the reference pads a list to its longest cloud with one op per cloud).
 * dicp_pack_list: out (N,n_max,cols)[b][i][k] = rows_b[i * strides[b] + k] for i < lens[b], else *pad (a DEVICE scalar; NULL: zero).  ptrs (N) device array of the
 * clouds' base pointers (dtype elements), lens / strides (N) int32 device arrays: rows and elements per row of each cloud (strides[b] >= cols: only the first
 * `cols` columns of a row are taken).  dicp_unpack_list, its adjoint: the cloud's gradient rows_b (lens[b], strides[b]) contiguous = gout[b][i][k] in the first
 * `cols` columns and zero beyond them; stride_max = max strides[b] sizes the launch.  One launch each whatever N. */
int dicp_pack_list(int dtype, const void* const* ptrs, const int32_t* lens, const int32_t* strides, int N, int n_max, int cols, void* out, const void* pad, void* stream);
int dicp_unpack_list(int dtype, const void* gout, void* const* ptrs, const int32_t* lens, const int32_t* strides, int N, int n_max, int cols, int stride_max, void* stream);
int dicp_scatter_add_rows(int dtype, const void* gout, const int32_t* idx, int N, int n, int m, int c, void* gtgt, void* stream);

/* One pass over the source points: residuals (ICP.py:143-149), trim and robust weights
 * (loss.py:21-58 via ICP.py:152-160), weight combine (ICP.py:162-169,194-196), Jacobian
 * rows (ICP.py:171-183) and the normal-equation sums A = J_w^T J_w, b = J_w^T e_w
 * (ICP.py:198-201) plus cost (ICP.py:229), sum(w) and #(w > thresh) (ICP.py:225,247).
 *   tgt (N,m,c) with c = 6 for pt2pl (normals in 3:6), 3 or 6 for pt2pt -- or 8 / 4: the same rows padded to 32 / 16 bytes (dicp_sweep_build's tgt_s,
 *   gathered at the SORTED positions dicp_knn_sweep wrote: idx = spos, m = m_pad); idx == NULL (then m must equal n)
 *   means tgt already holds ONE ROW PER SOURCE POINT -- the soft neighbours of dicp_gumbel_nn;
 *   w_init (N,n), NULL = unit weights (the reference's weight=None); alive (N) multiplies w_init (the zeroing of ICP.py:256-257), may be NULL;
 *   partials (N, nblk, DICP_NACC_PAD) with nblk = dicp_accumulate_blocks(n);
 *   w_out: cloud b's n weights are written at w_out + b*w_stride (elements); may be NULL. */
int dicp_accumulate(int dtype, const dicp_weight_params* prm, const void* src, const void* tgt, int c,
                    const int32_t* idx, const void* pose, const void* w_init, const void* alive, const int32_t* src_rows,
                    int N, int n, int m, void* partials, void* w_out, int64_t w_stride, void* stream);

/* Per-cloud state advanced by dicp_step (one ICP iteration's tail, ICP.py:198-260). */
typedef struct dicp_step_io {
    const void* partials;    /* (N,nblk,32) T from dicp_accumulate */
    int32_t nblk;
    int32_t iter;            /* ii, 0-based */
    int32_t dim;             /* 2 or 3                                    ICP.py:186-189,203-207 */
    int32_t const_iter;      /* config const_iter                         ICP.py:240 */
    double tolerance;        /* ICP(tolerance=...)                        ICP.py:239 */
    int32_t rows_per_point;  /* 3 for pt2pt, 1 for pt2pl (weights layout ICP.py:164-165) */
    int32_t n;               /* source points per cloud */
    const void* pose_in;     /* (N,12) T */
    void* pose_out;          /* (N,12) T   C <- exp(dphi^)^T C, r <- r - dr    ICP.py:209-217 */
    void* delta;             /* T: cloud b's 6 numbers at delta + b*delta_stride       ICP.py:220 */
    int64_t delta_stride;
    void* cost;              /* T: cloud b's cost at cost + b*cost_stride              ICP.py:229-234 */
    const void* cost_prev;   /* same layout, previous iteration, or NULL */
    int64_t cost_stride;
    double* areg;            /* (N,36) the regularised matrix that was inverted (for backward) */
    const void* alive;       /* (N) T in:  w_init multiplier of THIS iteration        ICP.py:256-257 */
    void* alive_out;         /* (N) T out: multiplier of the next one (0 once converged; may alias alive) */
    uint8_t* converged;      /* (N) in/out                                 ICP.py:239 */
    void* iterations;        /* (N) T in/out                               ICP.py:245 */
    void* matched_ratio;     /* (N) T in/out                               ICP.py:247-251 */
    const void* n_start;     /* (N) T: #(w_init > thresh) of the ORIGINAL w_init, in weight rows */
    void* n_matched;         /* (N) T out: #(w > thresh) this iteration (for ICP.py:268-271) */
    void* w_cur;             /* this iteration's weights (cloud stride w_stride) or NULL */
    const void* w_prev;      /* previous iteration's, or NULL              ICP.py:224-226 */
    int64_t w_stride;
    int32_t* n_not_converged;/* device counter for this iteration, pre-zeroed: += 1 per cloud with |delta| >= tol */
    const void* frame;       /* optional (N,12) T: the search frame (dicp_search_frame) */
    void* pose_search_out;   /* optional (N,12) T: [Q C' | Q r' + t], what the NEXT search reads (NULL: not kept) */
    const void* rmax;        /* optional (N,4) T: bounding radius and midpoint of each source cloud (dicp_loop_init) */
    void* dcum;              /* optional T: cloud b's (motion bound, point rounding) pairs at dcum + b*dcum_stride: [2(iter+1)] = [2 iter] + how far any
                                query of the cloud can have moved between pose_in and pose_out, [2(iter+1)+1] = the rounding of a point transformed
                                with pose_out (match certificates, dicp_loop_buffers.cert.q) */
    int64_t dcum_stride;
    int32_t* cert_cloud;     /* optional (N,8): per-cloud counters of the match certificates (dicp_loop_buffers.cert.cloud); the step decides
                                from them whether the cloud's certificates stay on */
    /* Match certificates, optional: the step also makes the NEXT iteration's guard launch its work list -- the units whose filter value (cert_qu) does not stand
       under the motion bound it has just written, or all units of a cloud whose certificates are off / tried again.  A guard launch with one wave per unit
       spent 12 us finding out that it had nothing to do (the footprint of the search code it carries: 4096 blocks in 3.2 rounds); with the list it is a
       small grid that reads one counter. */
    const void* cert_qu;     /* (N, cert_units) T */
    int32_t cert_units;      /* units per cloud of the sweep's launch configuration */
    int32_t glist_cap;       /* entries per list */
    int32_t* glist;          /* (8, glist_cap) int32: entry = cloud * cert_units + unit, list = cloud & 7 (the XCD the cloud's blocks run on) */
    int32_t* gcount;         /* (8) zeros: entries in each list of the next iteration */
    int32_t* cert_scount;    /* optional (N): lengths of the clouds' candidate-set lists (dicp_loop_buffers.cert.scount): 64 sets make one more entry, -1 - (cloud * ceil(n/64) + chunk) */
    int32_t* cert_slist;     /* with cert_scount, (N, n): the lists themselves -- the step fills the tail of a list's last chunk of 64 with -1 and moves the length up to it, so that the
                                next guard launch's appends never land inside a chunk that launch is re-scoring */
    int32_t w_copied;        /* 1: the accumulate launch of this iteration already wrote w_prev into w_cur for the clouds that are frozen (alive = 0;
                                dicp_icp_forward does): the step then has nothing to copy for them (ICP.py:224-226) */
} dicp_step_io;

/* Reduce the partials, solve the 6x6 (3x3 for dim 2) system (ICP.py:200-201), update the
 * pose (ICP.py:209-217) and do the loop bookkeeping of ICP.py:219-257 on device. */
int dicp_step(int dtype, const dicp_step_io* io, int N, void* stream);

/* Caller-allocated state of one whole ICP call, indexed by iteration k = 0..K-1 (T = dtype's scalar). */
/* The Gumbel-softmax correspondence (nn.py:43-70, config functionality.gumbel) inside the ICP loop: iteration k is
   dicp_transform_points -> dicp_gumbel_nn -> dicp_accumulate on the soft neighbour rows -> dicp_step, and in reverse
   dicp_step_bwd -> dicp_accumulate_bwd (gradient of the rows) -> dicp_gumbel_nn_bwd (to the transformed source and the target) ->
   dicp_transform_points_bwd (to the source and the pose).  U and seeds are HOST arrays (read while the call enqueues). */
typedef struct dicp_gumbel_loop {
    const void* const* U;    /* optional (K) device pointers, each (N,n,m): the uniform noise of iteration k; NULL: in-kernel noise from seeds[k] */
    const uint32_t* seeds;   /* (K), used when U is NULL */
    double eps, tau;
    void* ps_t;              /* (N,n,3) scratch: the transformed source of the iteration at hand */
    void* nbr;               /* (K,N,n,c) the soft neighbour rows of every iteration (the reverse sweep reads them) */
    void* lse;               /* (K,N,n) their log-sum-exp */
    void* g_nbr;             /* dicp_icp_backward: (N,n,c) scratch */
    void* g_ps;              /* dicp_icp_backward: (N,n,3) scratch */
} dicp_gumbel_loop;

/* dicp_loop_buffers.search -- the nearest-neighbour search of the loop: the packed / sorted target, the sweep's index and query order, the search frame, the scoring form */
typedef struct dicp_search_buffers {
    int32_t knn_variant;     /* DICP_KNN_VALU | _MFMA (uses tgt4) or DICP_KNN_SWEEP (uses tgt4 = tgs4 + the arrays below); bits 8..15: optional
                                launch configuration (as dicp_knn / dicp_knn_sweep); bit 25: never take the one-block-per-cloud small path */
    int32_t m_pad;
    const void* tgt4;        /* dicp_pack_target output, or its x-sorted form for the sweep */
    const int32_t* tperm;    /* sweep only */
    const int32_t* qorder;   /* sweep only, may be NULL */
    const int32_t* bucket;   /* sweep only */
    const void* brange;      /* sweep only */
    int32_t nbkt;
    unsigned long long* pairs; /* sweep only, optional: DICP_PAIR_SHARDS counters */
    const void* frame;       /* optional (N,12) T: the search frame (dicp_search_frame); tgt4 / the sweep index were then
                                built with it */
    void* poses;             /* optional (K+1,N,12) T: [Q C | Q r + t] per iteration, written by dicp_loop_init (k = 0) and
                                the step kernels; the searches read it instead of poses.  NULL: they read poses */
    const void* tgt_sorted;  /* sweep only, optional (N,m_pad,tgt_sorted_stride): dicp_sweep_build's tgt_s.  With it (and spos) the forward accumulate
                                gathers the match rows from the sorted copy at spos -- one aligned sector per row -- and idx may be NULL */
    int32_t tgt_sorted_stride;
    const dicp_gumbel_loop* gumbel; /* knn_variant DICP_KNN_GUMBEL: the soft correspondences' buffers (idx / spos / tgt4 are then unused; gtgt of dicp_icp_backward
                                is (N,m,c) zeros and is added to; no truncated sweep: the matches themselves carry gradient) */
    int32_t first_done;      /* sweep path: 1 = the matches of iteration 0 are already in spos (the caller enqueued dicp_knn_sweep under pose_search[0] and
                                   the first query order itself, right behind the index build, so that the search runs while the host is still
                                   preparing the loop): dicp_icp_forward then starts iteration 0 at its accumulate */
    const void* tgt_f16;     /* optional, float32: the split-f16 image of tgt4 (dicp_knn_f16_pack, with the same tgt_rows).  DICP_KNN_MFMA needs it; on the
                                sweep path (tgt4 = the sorted rows) the plain searches of big problems score on the matrix cores when it is given */
    int32_t* form;           /* sweep path, optional (K, N) int32 ZEROS: the plain search of iteration k tallies its slabs' tiles per cloud in row k and, given tgt_f16,
                                takes the scoring form of every cloud from row k-1 (dicp_knn_sweep's form_in / form_out; a row of zeros: no plain search then).
                                Without tgt_f16 the tallies are only kept (a caller may decide from them whether the next call of the shape gets the image) */
    int32_t form_default;    /* the form of a cloud without a tally: 0 vector, 1 matrix cores */
    const int32_t* form_plan; /* optional HOST array (K) int32, read during the call: the form of iteration k's plain search for EVERY cloud -- 1 vector, 2 matrix
                                cores (needs tgt_f16), 0 as above (by the tallies / the default).  A caller that has seen an earlier call's tallies of the same shape
                                plans with them: one launch per search instead of the two of the per-cloud choice */
} dicp_search_buffers;

/* dicp_loop_buffers.cert -- match certificates (sweep path, optional: q == NULL = off): which queries' matches are PROVEN unchanged, so that an iteration searches only the others */
typedef struct dicp_cert_buffers {
    void* q;                 /* sweep only, optional (N,n) T, by query: match certificates ("budgets": see knn_sweep_kernel).  With them the first iteration
                                of a query order searches every query and writes its budget; a later one searches only the queries whose match is
                                not PROVEN unchanged -- whole units in a guard launch where many are, the others inside dicp_accumulate's launch
                                (exact: same indices as a full search).  Needs spos (and no idx), tgt_sorted, qorder, cert.qu, rmax, dcum */
    void* qu;                /* (N, ceil(n/64)) T scratch */
    void* set;               /* optional, N*n*(sizeof(T) + 16) bytes of scratch: candidate sets -- a query whose match has a runner-up inside the rounding allowance
                                of the scores (dense surfaces, duplicated targets) gets no certificate of its own; the search that finds this keeps the
                                rows of its 4 smallest scores and a budget from the best row OUTSIDE that set, and while the budget stands the accumulate
                                re-scores those 4 rows instead of searching (exact: the match stays strictly below every outside row) */
    int32_t* count;          /* optional (K,128) zeros: per iteration, units searched again [0,64) and single queries [64,128), sharded by block */
    void* rmax;              /* (N,4) T from dicp_loop_init: bounding radius and midpoint of each source cloud */
    void* dcum;              /* (N, 2(K+1)) T: per iteration (motion bound since iteration 0, rounding of a transformed point); dicp_loop_init
                                writes iteration 0's, the step kernels the rest */
    int32_t reset;           /* 1: qorder is new in this call's first iteration: that iteration searches every query */
    int32_t* cloud;          /* optional (N,8) zeros, per cloud: units / single queries searched again in the current iteration, and the state the
                                step kernel keeps from them: certificates that cost more than 60 % of a full search are switched off -- for good where
                                the evidence is structural (queries without any certificate after a search of every unit), for 2, 4, .. 16 iterations
                                and then certified afresh where a guarded iteration was costly twice in a row (a cloud that is still moving).
                                While off, every unit of the cloud is searched plainly and nothing is checked: results are the same either way */
    void* nbr;               /* (N, n, 6 | 3) T scratch (pt2pl | pt2pt): the matched target row of every query, rewritten where a match changes */
    int32_t* gdirty;         /* (N, ceil(n/64)) int32 scratch: 1 = the guard launch of the iteration changed a match among these 64 consecutive queries */
    int32_t* pend;           /* (N, n) int32 ZEROS, by query: a match the guard launch CHANGED, left for the accumulate of the same iteration (match + 2) */
    int32_t* glist;          /* (8, max(N, 2) * ceil(n/64)) int32 scratch: the guard launches' work lists (dicp_step_io.glist), rewritten by every step */
    int32_t* gcount;         /* (K + 1, 8) int32 ZEROS: their lengths, per iteration */
    int32_t* slist;          /* with cert.set, (N, n) int32 scratch: per cloud, the slots that were given a candidate set -- the guard launch re-scores the standing
                                ones 64 to a wave from it */
    int32_t* scount;         /* (N) int32 ZEROS: its lengths */
    int32_t* cm;             /* (N, n) int32 scratch, by slot of the query order: the searches' own copy of the current matches.  The certificates' state
                                (cert.q, cert.set, cert.cm) goes by SLOT: the guard launch, which alone reads it, takes a unit's share as one coalesced piece */
} dicp_cert_buffers;

/* dicp_loop_buffers.hist -- what every iteration leaves for the result and for the reverse sweep: poses, steps, costs, matches, weights */
typedef struct dicp_history_buffers {
    int32_t per_iter;        /* 1: idx / spos are (K,N,n) and every iteration keeps its own (needed for backward); 0: (N,n) reused */
    int32_t* spos;           /* sweep only, optional (K,N,n): per-iteration sorted match positions.  Non-NULL in dicp_icp_backward
                                selects dicp_accumulate_bwd_window: src / w_init / tgt are then the SORTED copies it documents,
                                gsrc / gw accumulate in slot order, gtgt is the slab, bwd_partials has dicp_window_blocks blocks */
    void* poses;             /* (K+1,N,12): poses[0] = initial pose, poses[k+1] written by iteration k */
    void* deltas;            /* (N,K,6) */
    void* costs;             /* (N,K) */
    double* areg;            /* (K,N,36) or NULL when no backward is needed */
    void* alive;             /* (K+1,N): alive[0] = 1, alive[k+1] written by iteration k */
    int32_t* idx;            /* (K,N,n) or (N,n); optional on the sweep path when tgt_sorted and spos are given */
    void* w;                 /* weights of every iteration: iteration k, cloud b at w + k*w_iter + b*w_stride (elements);
                                (N,K,n): w_iter = n, w_stride = K*n.  May be a per-slab virtual base: only [k0,k1) is touched */
    int64_t w_iter, w_stride;
    const void* w_prev0;     /* weights of iteration k0-1 (cloud stride w_stride too), or NULL when k0 == 0   ICP.py:224-226 */
    const int32_t* spos_prev_chunk; /* certified iterations, histories in several slabs: the virtual base of the slabs BEFORE the one `spos` addresses
                                (iteration s < spos_floor at spos_prev_chunk + s*N*n), NULL when spos_floor == 0 */
    int32_t spos_floor;      /* first iteration of the history slab `spos` addresses (0: one slab) */
    /* Certified iterations keep the match history BY REFERENCE and the matched rows in a cache (accumulate_kernel, csrc/kernels_accumulate.h).
       Near the pose almost no match changes from one iteration to the next: copying every match into the next iteration's slab, reading every
       budget and gathering every 24-byte row again was 19 of the 63 bytes per point such a launch moved (section 8d counts 44). */
    int32_t* spos_of;        /* (K+1, N, ceil(n/64)) int32, needed with cert.q when hist.per_iter: spos_of[k][b][g] = the iteration whose slab of `spos` holds
                                the matches of queries [64g, 64g+64) of cloud b at iteration k >= spos_of_from.  Written by the certified iterations' accumulate
                                (rows k and k+1); dicp_icp_backward reads the matches through it */
    int32_t spos_of_from;    /* dicp_icp_backward: iterations below it have complete slabs of their own (the searches before the certificates start); the
                                forward ignores it */
} dicp_history_buffers;

/* dicp_loop_buffers.bwd -- dicp_icp_backward only: the windowed form's placement and side buffer, the truncated reverse sweep, its one-launch tail, deterministic target gradients */
typedef struct dicp_bwd_buffers {
    const int32_t* spos_ref; /* backward, windowed form: (N,n) reference matches that place the windows; qorder = its slot order */
    void* gts_far;           /* backward, windowed form: (N,m_pad,CV) atomically accumulated out-of-window rows */
    int32_t overwrite;       /* dicp_icp_backward, windowed form: 1 = gsrc / gw / the slab (gtgt) are uninitialised and this call's
                                first launch (iteration k1-1) writes them instead of adding; 0 = they are accumulators */
    /* dicp_icp_backward, optional: truncated reverse sweep.  Going backwards, iteration k adds to every gradient something LINEAR in the
       cotangent (G_A + G_A^T, g_b)_k of its normal equations, and the chain of pose cotangents shrinks by orders of magnitude per iteration near
       the pose (a Gauss-Newton step near its fixed point is a strong contraction).  step_bwd measures, per cloud and in the data's own units,
       what iteration k adds (m_k = max |G_ab| s_a s_b, |g_a| s_a with s_a = sqrt(A_aa)) and the most any earlier iteration could add
       (|g_a| s_a times the largest recorded step |delta_k',b| s_b of the iterations before k), and ENDS the cloud's sweep at iteration k -- this
       and all earlier iterations do no per-point work; of the pose cotangent only the part that does not go through the normal equations
       (C_new = exp(delta^)^T C: gC = R gCn, gr = grn) travels on to the gradient of T_init -- when 16x the larger of the two is
       below bwd.skip_eps times the largest m of the cloud's later iterations.  The host side passes a few units of the result type's roundoff
       (2^-22 for float32, 2^-40 for float64): what is dropped is below the resolution of the sums it would be added to.  Iterations at which a
       cloud was frozen (alive = 0) are skipped too (every term is exactly zero).  bwd.skip NULL = off: every iteration runs, like autograd. */
    int32_t* skip;           /* (N) zero-initialised once per backward pass (all chunks of it share it): 0 take part / 1 frozen at this iteration / 2 sweep ended */
    double* mref;            /* (N) zero-initialised once per backward pass */
    int32_t* live;           /* optional (K + 1) zeros: clouds that took part in iteration k; [K]: raised with bwd.tail_arrive[N] (so that one copy brings the host both) */
    double skip_eps;
    int32_t tail_from;       /* windowed form with bwd.skip: the iterations k < bwd.tail_from run as ONE launch (0: every iteration is its own pair of
                                launches).  Before the last few iterations almost every cloud's sweep has ended, and a pair of dependent launches per
                                iteration is pure dispatch time.  On accumulate_bwd_window's grid, that launch multiplies an ended cloud's pose cotangent
                                through all its remaining iterations; a cloud that is still at work is swept by its own blocks together, iteration by
                                iteration (each block runs the cloud's step_bwd itself, then its share of accumulate_bwd_window, and waits on a per-cloud
                                counter for the others' pose sums): the same arithmetic as the per-iteration launches.  The call must then run down to
                                k0 = 0, and the cotangent it leaves INCLUDES the last pose sums (dicp_pose_grad_out without partials).  The first
                                iteration of a backward pass (bwd.overwrite) always takes the per-iteration launches.  The caller picks the iteration
                                from where the previous call's sweeps ended. */
    void* tail_partials;     /* (N, dicp_window_blocks, DICP_NBWD_PAD): the second buffer of pose sums of that launch */
    int32_t* tail_arrive;    /* (N + 1) zeros per backward pass: its per-cloud counters; [N] is raised if a wait ran out.  That cannot happen while a cloud's
                                blocks are all resident, which dicp_bwd_tail_max_blocks guarantees for a launch that has the GPU to itself (or shares it with one
                                more of its kind); a GPU kept full by other work for longer than the wait's bound (~0.5 s) can still make one run out.  The
                                block then stops waiting for good and folds NaN from there on: the cloud's pose cotangent and (part of) its point gradients
                                come out NaN, never as plausible wrong numbers, and the caller must treat the error word as a failed call.
                                (Nonzero = failed.  The words also say whose wait it was, for the report: [N] = 0x40000000 | cloud << 8 | iteration;
                                bwd.live[K] = 0x40000000 | arrivals seen << 16 | block << 8 | generation.)
                                Hand-off between the blocks (hardware assumption, gfx950): the pose sums are written and read with agent-scope atomic
                                accesses (sc1: served by the memory side, coherent across the XCDs' L2s) and counted with an agent-scope atomic add after
                                the storing wave's s_waitcnt vmcnt(0) -- MI355X_MICROARCH's measured hand-off form, not the C++ memory model's release /
                                acquire pair (an agent-scope fence writes back / invalidates the XCD's whole L2: 70 us per iteration against 26) */
    int32_t* det_far_row;    /* dicp_icp_backward, windowed form, optional (N,n) int32 + bwd.det_far_val (N,n,cv): DETERMINISTIC target gradients.  Without them the
                                contributions to a target row are summed in the order the block's waves happened to reach it, and those whose match lies outside the
                                block's window are added with float atomics: two runs differ in the last bits.  With them every window row sums its slots in
                                ascending slot order, and an out-of-window contribution is left here by its slot (row, values) and added by one launch per
                                iteration that walks a cloud's slots in order (one lane per target row): the same bits on every run, given the same slot order. */
    void* det_far_val;
} dicp_bwd_buffers;

/* The buffers of the loop entry points (dicp_icp_forward / _forward_plan / dicp_icp_backward), one versioned struct: `abi` must be DICP_ABI_VERSION (the entry
 * points return DICP_ERR_ABI otherwise: a caller built against another layout would hand the library a pose history where it expects an index history). */
typedef struct dicp_loop_buffers {
    int32_t abi;             /* DICP_ABI_VERSION of the header the caller was built against */
    const void* src;         /* (N,n,3) */
    const void* tgt;         /* (N,m,c) */
    const void* w_init;      /* (N,n); NULL = unit weights (the reference's weight=None: 4 bytes per point and launch less to read) */
    int32_t c;
    int32_t K;               /* capacity of the histories (= max_iterations) */
    uint8_t* converged;      /* (N) zero-initialised */
    void* iterations;        /* (N) zero-initialised */
    void* matched_ratio;     /* (N) zero-initialised */
    const void* n_start;     /* (N) */
    void* n_matched;         /* (N) */
    void* partials;          /* (N, dicp_accumulate_blocks(n), DICP_NACC_PAD) scratch */
    int32_t* counters;       /* (K) zero-initialised: #clouds with |delta| >= tol at iteration k */
    int32_t* counters_host;  /* optional (K) int32 in PINNED (device-mapped) host memory: a dicp_icp_forward call that is not const_iter ends with one small launch
                                that stores counters_tag | min(counters[k], 0xfffff) to counters_host[k] for its iterations [k0,k1) -- the host's all-converged check
                                (ICP.py:259) reads the words one segment later, without a copy engine in the stream (a hipMemcpyAsync per segment cost a blit and its
                                completion signal: 10 us of every iteration of the reference's default mode).  The caller fills the words with -1 before the call */
    int32_t counters_tag;    /* bits 20..30: which call the words belong to (a late segment of the object's PREVIOUS call may still be writing when the host prepares the next) */
    void** events;           /* optional 6*K hipEvent_t: [6k] before / [6k+1] after the kNN of iteration k, [6k+2] / [6k+3] its
                                accumulate (forward), [6k+4] / [6k+5] its accumulate_bwd (backward); NULL = none.  The sweep,
                                accumulate and windowed-backward launches take their pair as the start / stop events of the
                                dispatch (hipExtLaunchKernel), the other forms are bracketed by hipEventRecord */
    const int32_t* src_rows; /* optional (N): rows of each source cloud that take part (ragged batches); qorder, if any, from dicp_query_order
                                with the same counts */
    const int32_t* tgt_rows; /* optional (N): rows of each target cloud that take part; tgt4 / the sweep index built with the same counts */
    dicp_search_buffers search;
    dicp_cert_buffers cert;
    dicp_history_buffers hist;
    dicp_bwd_buffers bwd;
} dicp_loop_buffers;

/* Head and tail of the backward loop.  dicp_pose_grad_in: gpose (N,12) double = [dL/dC row-major (9), dL/dr (3)] taken from the
 * upstream gradient gT (N,4,4) of the result T (NULL = zeros).  dicp_pose_grad_out: gT0 (N,4,4) = the same entries of
 * gpose + the sum over the nblk rows of bwd_partials (N,nblk,DICP_NBWD_PAD) of their pose slots 0..11 (the partials of the
 * last dicp_accumulate_bwd* launch, not yet folded in; NULL = none), bottom row 0: the gradient w.r.t. T_init. */
int dicp_pose_grad_in(int dtype, const void* gT, double* gpose, int N, void* stream);
int dicp_pose_grad_out(int dtype, const double* gpose, const void* bwd_partials, int nblk, void* gT0, int N, void* stream);

/* Loop state before iteration 0 (ICP.py:124-129): pose0 (N,12) from T_init (N,4,4), alive0 (N) = 1,
 * n_start (N) = rows * #(w0 > thresh), rows = 3 for pt2pt, 1 for pt2pl.  And after the last executed iteration K
 * (ICP.py:267-281): iterations / matched_ratio of clouds that never converged, T_out (N,4,4) from pose_K. */
int dicp_loop_init(int dtype, const void* T_init, const void* w0, double thresh, int rows, int N, int n,
                   void* pose0, void* alive0, void* n_start, const void* frame /* optional */, void* pose_search0 /* optional: [Q C_0 | Q r_0 + t] */,
                   const void* src, void* rmax, void* dcum, int dcum_stride /* optional (match certificates): rmax (N,4) = bounding radius and midpoint of each cloud of
                   src (N,n,3), dcum (N,dcum_stride >= 2): [0] = 0, [1] = the rounding of a point transformed with pose0 */, void* stream);
/* pose_search (N,12) = [Q C_0 | Q r_0 + t] from T_init (N,4,4) alone: the same values, for a caller that orders the first queries
 * before the loop state exists (frame optional) */
int dicp_search_pose(int dtype, const void* T_init, const void* frame, int N, void* pose_search, void* stream);
int dicp_loop_finish(int dtype, const void* pose_K, const void* alive_K, const void* n_start, const void* n_matched, int K, int N,
                     void* iterations, void* matched_ratio, void* T_out, void* stream);

/* Iterations [k0,k1) of the loop (ICP.py:131-260), enqueued back to back: no host work between iterations. */
int dicp_icp_forward(int dtype, const dicp_weight_params* prm, const dicp_loop_buffers* buf, int N, int n, int m,
                     int dim, int const_iter, double tolerance, int k0, int k1, void* stream);
/* The constant-iteration loop of the sweep path in ONE call: the segments [k0[s], k1[s]) dicp_icp_forward would be called for one after the
 * other (cut where the queries are re-ordered and where the match certificates start), the query re-orderings between them included
 * (new_order[s] != 0: dicp_query_order under the segment's first search pose into order[s]).  buf as for dicp_icp_forward with
 * every history in one slab (spos / idx / w are their real bases); qorder, cert_q / cert_qu / cert_count / cert_cloud / cert_nbr / cert_gdirty / cert_pend / cert_cm, cert_reset, spos_prev_chunk / spos_floor
 * and w_prev0 of `buf` are ignored: the call derives them per segment (certificates from iteration cert_from on; cert_from < 0: none).
 * The reference's per-iteration host check (ICP.py:259) needs the host between segments: tolerance mode keeps calling dicp_icp_forward. */
#define DICP_MAX_SEGMENTS 16
typedef struct dicp_segment_plan {
    int32_t nseg;
    int32_t k0[DICP_MAX_SEGMENTS], k1[DICP_MAX_SEGMENTS];
    int32_t new_order[DICP_MAX_SEGMENTS];    /* 1: order[s] is computed at the segment's start */
    int32_t cert_from;
    int32_t pad0;
    int32_t* order[DICP_MAX_SEGMENTS];       /* (N,n) each: the query order the segment searches in (several segments may share one), NULL: none */
    const void* keys;        /* (N,m_pad) sorted target x keys (dicp_sweep_sort): the rank search of dicp_query_order */
    void* cert_q; void* cert_qu; int32_t* cert_count; int32_t* cert_cloud; void* cert_set;
    void* cert_nbr; int32_t* cert_gdirty; int32_t* cert_pend; int32_t* cert_cm; int32_t* cert_glist; int32_t* cert_gcount; int32_t* cert_slist; int32_t* cert_scount;     /* as dicp_loop_buffers' (its spos_of is taken from `buf`) */
} dicp_segment_plan;
int dicp_icp_forward_plan(int dtype, const dicp_weight_params* prm, const dicp_loop_buffers* buf, const dicp_segment_plan* plan, int N, int n, int m,
                          int dim, int const_iter, double tolerance, void* stream);
/* Reverse sweep over iterations k1-1..k0.  gpose/gpose_tmp (N,12) double: gpose holds the cotangent of pose_k1 on entry;
 * the two alternate, so the cotangent of pose_k0 is left in gpose when k1-k0 is even and in gpose_tmp when it is odd
 * (no copy: swap the two pointers for the next chunk); gs (N,36), gb (N,6) scratch T;
 * gsrc/gtgt/gw accumulate like dicp_accumulate_bwd; bwd_partials (N,nblk,DICP_NBWD_PAD) carries the last
 * accumulate_bwd's C-bar/r-bar sums across chunks (have_partials: it holds valid sums on entry). */
int dicp_icp_backward(int dtype, const dicp_weight_params* prm, const dicp_loop_buffers* buf, int N, int n, int m, int dim,
                      double* gpose, double* gpose_tmp, int have_partials, void* gs, void* gb, void* gsrc, void* gtgt, void* gw,
                      void* bwd_partials, int k0, int k1, void* stream);
/* out (N,n) = the matches of iteration k out of the history `spos` (virtual base: iteration s at spos + s*N*n) kept by reference through spos_of
 * (dicp_loop_buffers.hist.spos_of; NULL: iteration k's own slab, a plain copy): what places the windows of the backward pass (spos_ref). */
int dicp_resolve_matches(const int32_t* spos, const int32_t* spos_of, int k, const int32_t* src_rows /* optional (N): -1 for the rows past a cloud's own */,
                         int N, int n, int32_t* out, void* stream);
/* The slot order of a DETERMINISTIC windowed backward (dicp_loop_buffers.bwd.det_far_row): qorder (N,n) = the queries in a stable order of their reference matches
 * spos_ref (N,n) (sorted positions; -1 and a cloud's rows past src_rows: last, in index order) -- the same permutation on every run.  dicp_sweep_sort's stable
 * radix sort on the positions as keys; scratch: dicp_match_order_scratch_bytes(dtype, N, n) bytes, 256-byte aligned. */
size_t dicp_match_order_scratch_bytes(int dtype, int N, int n);
int dicp_match_order(int dtype, const int32_t* spos_ref, const int32_t* src_rows, int N, int n, void* scratch, size_t scratch_bytes, int32_t* qorder, void* stream);


/* Backward of dicp_step for iteration k.  gpose_in (N,12) double = cotangent of pose_out
 * that flowed through later iterations; bwd_partials (N,nblk,DICP_NBWD_PAD) T = the
 * C-bar/r-bar sums dicp_accumulate_bwd produced for iteration k+1 (NULL for the last).
 * Out: gs (N,36) T = G_A + G_A^T, gb (N,6) T, gpose_out (N,12) double. */
int dicp_step_bwd(int dtype, const double* gpose_in, const void* bwd_partials, int nblk, int dim,
                  const void* pose_k, const void* delta_k, int64_t delta_stride, const double* areg_k,
                  void* gs, void* gb, double* gpose_out, int N, void* stream);

/* Backward of dicp_accumulate (the reference uses stock autograd through
 * ICP.py:137-201).  Recomputes the forward quantities from (src,tgt,idx,pose,w_init).
 *   gsrc (N,n,3) +=, gw (N,n) +=, gtgt (N,m,c) += via atomics (zero-initialised by caller;
 *   may be NULL if target needs no gradient), bwd_partials (N,nblk,DICP_NBWD_PAD) written. */
int dicp_accumulate_bwd(int dtype, const dicp_weight_params* prm, const void* src, const void* tgt, int c,
                        const int32_t* idx, const void* pose, const void* w_init, const void* alive,
                        const void* gs, const void* gb, const int32_t* src_rows, int N, int n, int m,
                        void* gsrc, void* gtgt, void* gw, void* bwd_partials, void* stream);

/* Windowed form of dicp_accumulate_bwd for the sorted-sweep path, entirely in SORTED space and without global
 * atomics on the common path:
 *   slot s of cloud b = query qorder[b][s], qorder (N,n) being ANY order that keeps x-neighbours together under the
 *   poses involved (NULL = identity; it need not be the order the forward search used);
 *   src_s (N,n,3) = src rows in slot order, w_s (N,n) likewise; tgt_s (N,m_pad,c) = target rows in the sweep's
 *   sorted order (row s = tgt[tperm[s]]); spos (N,n) as written by dicp_knn_sweep for THIS iteration (by query);
 *   spos_ref (N,n) = the spos of ONE reference iteration, the same in every launch that adds into a given slab
 *   (it places each block's window of dicp_window_rows consecutive sorted target rows; may equal spos).
 * Accumulates (+=) gsrc_s (N,n,3), gw_s (N,n) in slot order.  Target gradients (slab = NULL: not wanted): each of the
 * dicp_window_blocks(dtype,n,m_pad) blocks per cloud sums its matches in LDS and adds its window to its own rows
 * of slab (N, blocks, dicp_window_rows(dtype), CV) with plain read-modify-writes (CV = 6 for pt2pl, 3 for pt2pt);
 * matches outside a window are added to gts_far (N,m_pad,CV) with atomics.  After the last iteration
 * dicp_window_reduce adds slab + gts_far into gtgt (N,m,c) in the ORIGINAL target order (+=, call it once per
 * slab; pass gts_far with one of them; overwrite != 0: = instead of +=, when nothing else has been added to gtgt).  bwd_partials: (N, dicp_window_blocks, DICP_NBWD_PAD).
 * dicp_accumulate_bwd_window's own overwrite != 0: the FIRST launch into gsrc_s / gw_s / slab, which then need no
 * zero fill (it writes every slot and every window row instead of adding to them); gts_far is always added to. */
int dicp_window_blocks(int dtype, int n, int m_pad);
int dicp_window_rows(int dtype);
/* the most dicp_window_blocks(...) per cloud with which dicp_loop_buffers.bwd.tail_from may be used on the current device (0: never): its launch lets a
 * cloud's blocks wait for each other, so they must all be resident -- half of what one XCD holds of that kernel.  dicp_icp_backward returns
 * DICP_ERR_SHAPE for a tail beyond it. */
int dicp_bwd_tail_max_blocks(int dtype);
int dicp_accumulate_bwd_window(int dtype, const dicp_weight_params* prm, const void* src_s, const void* tgt_s, int c,
                               const int32_t* spos, const int32_t* spos_ref, const int32_t* qorder, const void* pose, const void* w_s,
                               const void* alive, const void* gs, const void* gb, const int32_t* src_rows, int N, int n, int m_pad, void* gsrc_s, void* slab,
                               void* gts_far, void* gw_s, void* bwd_partials, int overwrite, void* stream);
int dicp_window_reduce(int dtype, const void* slab, const int32_t* spos_ref, const int32_t* qorder, const int32_t* tperm, const void* gts_far,
                       const int32_t* src_rows, int N, int n, int m, int m_pad, int cv, void* gtgt, int c, int overwrite, void* stream);

/* out[b][perm[b][s]][k] += in[b][s][k] for s < cnt, k < cols.  in (N,in_rows,c_in), perm (N,perm_rows) injective per
 * cloud (plain read-modify-write), out (N,out_rows,c_out).  Undoes a sorted order. */
int dicp_permute_add_rows(int dtype, const void* in, const int32_t* perm, int N, int cnt, int in_rows, int perm_rows, int c_in, int cols,
                          void* out, int out_rows, int c_out, void* stream);
/* The same with = instead of += (perm a bijection onto the out rows: nothing of out[:, :, :cols] needs initialising). */
int dicp_permute_rows(int dtype, const void* in, const int32_t* perm, int N, int cnt, int in_rows, int perm_rows, int c_in, int cols,
                      void* out, int out_rows, int c_out, void* stream);

/* Gumbel-softmax soft correspondence, nn.__diff_nn_gumbel (nn.py:43-70), without the (N,n,m) tensors:
 *   out (N,n,c) = softmax_j((-|x_i - y_j|^2 + g_ij)/tau) @ y,  g = -log(-log(U + eps) + eps)   (nn.py:56-68)
 *   x (N,n,3), y (N,m,c) with c in {3,6}; lse (N,n) = log-sum-exp of the logits (kept for the backward).
 *   U (N,n,m) injects the uniform draw of nn.py:60; NULL = generated in-kernel from a counter hash of
 *   (seed, cloud, i, j) -- the backward regenerates it from the same seed instead of storing it.
 * Backward: gx (N,n,3) and/or gy (N,m,c) are WRITTEN (either may be NULL). */
int dicp_gumbel_nn(int dtype, const void* x, const void* y, int c, const void* U, uint32_t seed, double eps, double tau,
                   int N, int n, int m, void* out, void* lse, void* stream);
int dicp_gumbel_nn_bwd(int dtype, const void* x, const void* y, int c, const void* U, uint32_t seed, double eps, double tau,
                       const void* out, const void* lse, const void* gout, int N, int n, int m, void* gx, void* gy, void* stream);

/* Closed-form point-to-point step (Kabsch / SVD), the solver of ICP.pt2pt_dICP_SVD (ICP.py:533-591): batched,
 * weighted, with the rotation composed as U diag(1,1,det U det V) V^T (the reference multiplies by V where V^T
 * is required, ICP.py:566-570 -- only correct for planar data).
 *   accumulate: per-block sums [sum w, sum w p (3), sum w y (3), sum w y p^T (9), sum w|p|^2, sum w|y|^2] in
 *               partials (N, dicp_accumulate_blocks(n), DICP_NACC_PAD); w = w_init, times a hard gate
 *               |C p + r - y| < trim_dist when trim_on;
 *   step:       pose_out (N,12) = [C, r] minimising sum w |C p + r - y|^2; cost (N) = that minimum (ICP.py:585);
 *               save (N,DICP_KAB_SAVE) doubles for the backward pass (may be NULL);
 *   step_bwd:   gpose (N,12) T -> gacc (N,16) T = cotangents of the 16 leading sums;
 *   bwd:        gsrc (N,n,3) +=, gtgt (N,m,c) += (atomics, may be NULL), gw (N,n) += (may be NULL). */
#define DICP_KAB_SAVE 40
int dicp_kabsch_accumulate(int dtype, const void* src, const void* tgt, int c, const int32_t* idx, const void* pose,
                           const void* w_init, int trim_on, double trim_dist, const int32_t* src_rows, int N, int n, int m, void* partials, void* stream);
int dicp_kabsch_step(int dtype, const void* partials, int nblk, void* pose_out, void* cost, double* save, int N, void* stream);
int dicp_kabsch_step_bwd(int dtype, const void* gpose, const double* save, void* gacc, int N, void* stream);

/* The whole SVD loop (ICP.py:549-586) behind one call per segment, like dicp_icp_forward: iterations [k0,k1) = K x { search -> dicp_kabsch_accumulate ->
 * step } enqueued back to back, convergence on device.  A cloud whose cost sum w|T p - nn|^2 falls below `tolerance` (ICP.py:585; never under
 * const_iter) is frozen at that pose: rows_live[b] becomes 0, so later searches and sums skip it, and its matches (idx), the pose they were found
 * under (pose_used) and the SVD (save) of its last active iteration stay for dicp_kabsch_step_bwd / dicp_kabsch_bwd.  counters[k] = clouds still
 * moving after iteration k: the caller reads them (asynchronously) to end the loop; iterations past that point are no-ops. */
typedef struct dicp_kabsch_buffers {
    const void* src;         /* (N,n,3) */
    const void* tgt;         /* (N,m,c) */
    const void* w_init;      /* (N,n) */
    int32_t c;
    int32_t K;               /* capacity of the cost history (= max_iterations) */
    int32_t knn_variant;     /* as dicp_loop_buffers */
    int32_t m_pad;
    const void* tgt4;        /* packed rows (sorted for the sweep), built with `frame` */
    const int32_t* tperm;    /* sweep only */
    const int32_t* qorder;   /* sweep only, may be NULL */
    const int32_t* bucket;   /* sweep only */
    const void* brange;      /* sweep only */
    int32_t nbkt;
    int32_t pad0;
    unsigned long long* pairs;   /* sweep only, optional */
    const void* frame;       /* optional (N,12): the search frame */
    void* pose;              /* (N,12) in/out: the current pose [C | r] */
    void* pose_search;       /* optional (N,12) in/out: [Q C | Q r + t], what the searches read (NULL: they read pose) */
    void* pose_used;         /* (N,12) out: the pose of each cloud's last active iteration BEFORE its step */
    int32_t* idx;            /* (N,n) out: the matches of each cloud's last active iteration */
    void* partials;          /* (N, dicp_accumulate_blocks(n), DICP_NACC_PAD) scratch */
    double* save;            /* (N,DICP_KAB_SAVE) out */
    void* costs;             /* (N,K) out */
    void* iterations;        /* (N) zero-initialised: k+1 of the iteration a cloud converged at */
    int32_t* rows_live;      /* (N) in/out: source rows of each cloud that take part (n, or the cloud's own length); 0 once frozen */
    const int32_t* tgt_rows; /* optional (N) */
    int32_t* counters;       /* (K) zero-initialised */
    const void* tgt_f16;     /* knn_variant DICP_KNN_MFMA: the split-f16 image of tgt4 (dicp_knn_f16_pack) */
} dicp_kabsch_buffers;
int dicp_kabsch_forward(int dtype, const dicp_kabsch_buffers* buf, int N, int n, int m, int trim_on, double trim_dist, int const_iter, double tolerance,
                        int k0, int k1, void* stream);
int dicp_kabsch_bwd(int dtype, const void* src, const void* tgt, int c, const int32_t* idx, const void* pose, const void* w_init,
                    int trim_on, double trim_dist, const void* gacc, const int32_t* src_rows, int N, int n, int m, void* gsrc, void* gtgt, void* gw, void* stream);

/* The returned cloud pc = C p + r (ICP.py:274) and its adjoint: gsrc (N,n,3) = C^T gout (may be NULL);
 * partials (N, dicp_accumulate_blocks(n), DICP_NBWD_PAD) = per-block [sum gout p^T (9), sum gout (3)] for the pose. */
int dicp_transform_points(int dtype, const void* src, const void* pose, void* out, int N, int n, void* stream);
int dicp_transform_points_bwd(int dtype, const void* src, const void* pose, const void* gout, void* gsrc, void* partials,
                              int N, int n, void* stream);

/* loss(name, metric, differentiable, tanh_steepness).get_weight(err), loss.py:11-58, for
 * callers that use the class directly.  err (rows,r), r in {1,3}; w (rows). */
int dicp_loss_weight(int dtype, int loss, int differentiable, double metric, double tanh_k,
                     const void* err, int64_t rows, int r, void* w, void* stream);
int dicp_loss_weight_bwd(int dtype, int loss, int differentiable, double metric, double tanh_k,
                         const void* err, const void* gw, int64_t rows, int r, void* gerr, void* stream);

/* ---- One eager ICP call behind ONE host call per direction (ICP.py:49-303 for a dense batch on the sorted-sweep path, constant iteration count, no match
 * certificates: the mid-size calls -- 32 clouds x 4096 points x 10 iterations is 0.55 ms of kernels -- whose time was the host's, preparing ~30 buffers
 * and ~12 library calls from the interpreter).  The caller makes ONE allocation per direction; dicp_call_plan / dicp_call_backward_plan say how big and
 * where the results lie in it; dicp_call_forward runs dicp_sweep_setup -> dicp_knn_sweep (iteration 0) -> dicp_loop_init -> dicp_icp_forward_plan ->
 * dicp_loop_finish -> dicp_transform_points on it, dicp_call_backward runs dicp_pose_grad_in -> dicp_gather_rows -> dicp_icp_backward (windowed form,
 * truncated sweep, one-launch tail) -> dicp_permute_rows -> dicp_window_reduce -> dicp_pose_grad_out.  Same kernels, same arguments, same results bit for
 * bit as the per-buffer sequence (dicp_amd/_ops.py ICPLoop), which stays the path of every other kind of call. */
#define DICP_CALL_NBKT 1024
enum { DICP_CALL_FIRST_SEARCH = 1 /* iteration 0's search right behind the index build */, DICP_CALL_NO_SMALL_LOOP = 2 /* knn_variant bit 25 */ };
typedef struct dicp_call {
    const void* src;         /* (N,n,3) */
    const void* tgt;         /* (N,m,c) */
    const void* T_init;      /* (N,4,4) */
    const void* w0;          /* (N,n) or NULL = unit weights */
    int32_t N, n, m, c;
    int32_t K;               /* iterations: every one of them runs (the reference's const_iter) */
    int32_t dim;             /* 3, or 2 (ICP.py:107-116) */
    int32_t need_grad;       /* 1: keep what dicp_call_backward reads (per-iteration matches, the regularised normal matrices) */
    int32_t n_resort;        /* iterations in (0, K), ascending, before which the sweep re-orders its queries under the current pose (at most DICP_MAX_SEGMENTS - 1) */
    int32_t resort[DICP_MAX_SEGMENTS];
    int32_t flags;           /* DICP_CALL_* */
    int32_t directions;      /* dicp_search_frame's */
    double quantum;          /* dicp_search_frame's */
    double tolerance;        /* ICP.py:237: a cloud whose step falls below it is frozen */
    void* workspace;         /* dicp_call_layout.total bytes, 256-byte aligned; dicp_call_backward reads what dicp_call_forward left in it */
    void* results;           /* dicp_call_layout.results_total bytes, 256-byte aligned: the non-differentiable results (deltas, costs, converged, iterations,
                                matched_ratio, weights), in an allocation of their OWN -- a caller that keeps one of them (a per-step log of the costs) must not
                                keep the search structure, the match history and the sort scratch alive with it, and one that edits them must not be able to
                                touch what the reverse sweep reads (it reads `deltas`: the binding keeps that tensor's version) */
    void* T_out;             /* optional (N,4,4) / (N,n,3): the two differentiable results go here instead of into the workspace (a binding whose autograd */
    void* pc_out;            /* must not see them as views of one buffer that also holds results a caller may edit) */
} dicp_call;
typedef struct dicp_call_layout {       /* byte offsets into dicp_call.workspace -- and, for the six results named below, into dicp_call.results */
    size_t total, zeroed;    /* the workspace's size; its first `zeroed` bytes are cleared by dicp_call_forward itself */
    size_t results_total, results_zeroed;   /* the same for dicp_call.results */
    size_t T, pc;            /* (N,4,4) T, (N,n,3) T in the workspace (unused when T_out / pc_out are given) */
    size_t deltas, weights, costs, converged, iterations, matched_ratio;   /* in dicp_call.results: (N,K,6) T, (N,K,n) T, (N,K) T, (N) uint8, (N) T, (N) T */
    size_t pairs;            /* DICP_PAIR_SHARDS uint64: pairs scored by the searches */
    size_t n_matched, counters, poses, poses_search, alive, areg, n_start, partials, tgs4, tperm, bucket, brange, keys, tgt_sorted, scratch, scratch_bytes,
           frame, pose_s, orders, spos;      /* loop state and search structure (as dicp_loop_buffers names them) */
    int32_t n_orders, m_pad, nblk, pad0;
} dicp_call_layout;
int dicp_call_plan(int dtype, const dicp_call* call, dicp_call_layout* layout);
int dicp_call_forward(int dtype, const dicp_weight_params* prm, const dicp_call* call, void* stream);

typedef struct dicp_call_grads {
    const void* gT;          /* (N,4,4) cotangent of T, or NULL = zeros (the cotangent of pc reaches the pose through dicp_transform_points_bwd: the caller adds it to gT) */
    void* gsrc;              /* (N,n,3) written */
    void* gtgt;              /* (N,m,c) written, or NULL: not wanted (c = 6 for pt2pl, 3 for pt2pt: the rows ARE the gradient's rows) */
    void* gT0;               /* (N,4,4) written: the gradient w.r.t. T_init */
    void* gw;                /* (N,n) written, or NULL */
    void* workspace;         /* dicp_call_backward_layout.total bytes, 256-byte aligned */
    double skip_eps;         /* dicp_loop_buffers.bwd.skip_eps; 0: every iteration runs */
    int32_t tail_from;       /* dicp_loop_buffers.bwd.tail_from (0: none; the caller checks dicp_bwd_tail_max_blocks against nblk_w) */
    int32_t pad0;
    int32_t* live_host;      /* optional PINNED host memory, (K + 1) int32: receives bwd_live (+ the tail's error word) behind the pass's launches */
} dicp_call_grads;
typedef struct dicp_call_backward_layout {
    size_t total, zeroed;
    size_t live, arrive;     /* (K + 1) / (N + 1) int32: dicp_loop_buffers.bwd.live / bwd_tail_arrive of the pass */
    size_t mref, decisions, far, gpose, gtmp, src_s, w_s, gsrc_s, gw_s, slab, gs, gb, partials, tail_partials;
    size_t spos_ref;         /* (N,n) int32: the matches of the last executed iteration as a plain array (dicp_resolve_matches: they place the windows) */
    int32_t nblk_w, pad0;    /* dicp_window_blocks of the shape */
} dicp_call_backward_layout;
int dicp_call_backward_plan(int dtype, const dicp_weight_params* prm, const dicp_call* call, int want_tgt, int want_w, dicp_call_backward_layout* layout);
int dicp_call_backward(int dtype, const dicp_weight_params* prm, const dicp_call* call, const dicp_call_grads* grads, void* stream);
/* The same reverse sweep behind one call for a forward that was run buffer by buffer (dicp_icp_forward / dicp_icp_forward_plan on the sweep path, every history in
 * one slab): the caller names the forward's buffers.  K iterations were executed (tolerance mode: fewer than the histories' capacity K_cap, which is what their
 * strides and the live counters' length follow). */
typedef struct dicp_loop_backward_in {
    const void* src;         /* (N,n,3) */
    const void* tgt_sorted;  /* (N,m_pad,c): dicp_sweep_build's tgt_s */
    const void* w0;          /* (N,n) or NULL */
    const int32_t* tperm;    /* (N,m_pad) */
    const int32_t* qorder;   /* (N,n): the slot order of the pass (the forward's last query order) */
    const int32_t* spos;     /* (K_cap,N,n): sorted match positions per iteration, by query */
    const void* poses;       /* (K_cap+1,N,12) */
    const void* deltas;      /* (N,K_cap,6) */
    const double* areg;      /* (K_cap,N,36) */
    const void* alive;       /* (K_cap+1,N) */
    const int32_t* src_rows; /* optional (N) */
    const int32_t* tgt_rows; /* optional (N) */
    const int32_t* spos_of;  /* optional (K_cap+1,N,ceil(n/64)): the history is kept by reference from iteration spos_of_from on (dicp_loop_buffers.hist.spos_of) */
    int32_t N, n, m, c, K, K_cap, m_pad, dim;
    int32_t knn_variant;     /* as dicp_loop_buffers */
    int32_t spos_of_from;
    /* optional, all or none: what dicp_loop_backward_prepare made ahead of the pass -- the part of it that needs no cotangent */
    const void* src_s;       /* (N,n,3): src in the slot order `qorder` */
    const void* w_s;         /* (N,n): w0 in that order (NULL with w0 == NULL) */
    const int32_t* spos_ref; /* (N,n): the matches of iteration K - 1 as a plain array (NULL when that iteration's slab is its own: the pass reads it in place) */
} dicp_loop_backward_in;
/* The set-up of dicp_loop_backward that depends on the forward alone -- slot-order copies of the source and its weights, the reference matches resolved out of a
 * history kept by reference -- into caller buffers, to be named in `fwd` (src_s / w_s / spos_ref) for the pass.  A caller whose GPU would idle between the forward
 * and the backward (the reference's default mode: the host waits for the iteration count, returns, the loss is taken, autograd starts its thread: ~0.2 ms)
 * enqueues it behind the forward; 58 us of the pass at 256 x 16384.  spos_ref_out may be NULL (and fwd->spos_ref then stays NULL) when K - 1 < spos_of_from. */
int dicp_loop_backward_prepare(int dtype, const dicp_loop_backward_in* fwd, void* src_s_out, void* w_s_out, int32_t* spos_ref_out, void* stream);
int dicp_loop_backward_plan(int dtype, const dicp_weight_params* prm, const dicp_loop_backward_in* fwd, int want_tgt, int want_w, dicp_call_backward_layout* layout);
int dicp_loop_backward(int dtype, const dicp_weight_params* prm, const dicp_loop_backward_in* fwd, const dicp_call_grads* grads, void* stream);

/* The same for ICP.pt2pt_dICP_SVD (ICP.py:533-591; BASELINE configs[1]'s "HIP kNN + 3x3 SVD"): a dense batch on the sweep path, every one of K iterations run.
 * dicp_kabsch_call_forward: dicp_sweep_setup -> dicp_search_pose -> 3 x { dicp_query_order (before iteration 1) -> dicp_kabsch_forward } -> T -> dicp_transform_points;
 * dicp_kabsch_call_backward: the cotangent of the found pose -> dicp_kabsch_step_bwd -> dicp_kabsch_bwd (gradients written, not added). */
typedef struct dicp_kabsch_call {
    const void* src;         /* (N,n,3) */
    const void* tgt;         /* (N,m,c) */
    const void* T_start;     /* (N,4,4): the starting pose of the search */
    const void* w0;          /* (N,n) */
    int32_t N, n, m, c;
    int32_t K;
    int32_t trim_on;         /* matches farther than trim_dist are gated out */
    int32_t directions;      /* dicp_search_frame's */
    int32_t pad0;
    double trim_dist;
    double quantum;          /* dicp_search_frame's */
    double tolerance;        /* unused while every iteration runs; kept for the segment calls */
    void* workspace;         /* dicp_kabsch_call_layout.total bytes, 256-byte aligned */
    void* T_out;             /* (N,4,4): the pose found */
    void* pc_out;            /* optional (N,n,3): the source under it (ICP.py:581) */
} dicp_kabsch_call;
typedef struct dicp_kabsch_call_layout {
    size_t total, zeroed;
    size_t costs, iterations;    /* results: (N,K) T, (N) T */
    size_t pairs, counters, frame, keys, tperm, bucket, brange, tgs4, scratch, scratch_bytes, pose, pose_search, pose_used, partials, save, idx, rows_live, orders, gpose, gacc;
    int32_t m_pad, nblk;
} dicp_kabsch_call_layout;
typedef struct dicp_kabsch_call_grads {
    const void* gT;          /* (N,4,4) cotangent of the pose found, or NULL = zeros */
    void* gsrc;              /* (N,n,3) written */
    void* gtgt;              /* (N,m,c) written, or NULL */
    void* gw;                /* (N,n) written, or NULL */
} dicp_kabsch_call_grads;
int dicp_kabsch_call_plan(int dtype, const dicp_kabsch_call* call, dicp_kabsch_call_layout* layout);
int dicp_kabsch_call_forward(int dtype, const dicp_kabsch_call* call, void* stream);
int dicp_kabsch_call_backward(int dtype, const dicp_kabsch_call* call, const dicp_kabsch_call_grads* grads, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DICP_HIP_H */
