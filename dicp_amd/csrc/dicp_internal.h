// Entry points that one translation unit of libdicp_hip.so offers the others (not part of the C ABI).
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace dicp_tu {

// ---- knn_f16.hip: the matrix-core search (split-f16 filter on v_mfma_f32_32x32x16_f16 + exact float32 refine)
// The image of a batch of packed target rows [x, y, z, 0.5|y|^2] (dicp_pack_target / the sweep's sorted rows): per cloud m_img = rows rounded
// up to 512, 32 bytes per row in MFMA operand order, followed by KNN_F16_META 4-byte words per cloud (scale, error terms, far rows) and by the
// x of the first and the last row of every 64-row tile (2 floats per tile: what the sorted sweep's slab bound reads, through the scalar cache).
constexpr int KNN_F16_META = 80;
constexpr int KNN_F16_STAGE_ROWS = 512;
inline int knn_f16_image_rows(int m_pad) { return (m_pad + KNN_F16_STAGE_ROWS - 1) / KNN_F16_STAGE_ROWS * KNN_F16_STAGE_ROWS; }
inline size_t knn_f16_image_bytes(int N, int m_pad) { return (size_t)N * ((size_t)knn_f16_image_rows(m_pad) * 32 + KNN_F16_META * 4 + (size_t)knn_f16_image_rows(m_pad) / 64 * 8); }
inline size_t knn_f16_meta_offset(int N, int m_pad) { return (size_t)N * knn_f16_image_rows(m_pad) * 32; }
inline size_t knn_f16_edges_offset(int N, int m_pad) { return knn_f16_meta_offset(N, m_pad) + (size_t)N * KNN_F16_META * 4; }
// rows4: (N, m_pad) float4 rows; tgt_rows: optional own lengths (rows beyond them, and rows whose 0.5|y|^2 is not finite, never match)
int knn_f16_pack(const void* rows4, const int32_t* tgt_rows, int N, int m_full, int m_pad, void* image, void* stream);
// all n x m pairs of every cloud; idx (N,n) as dicp_knn writes it.  Words 6 / 7 of a cloud's meta record count the queries that needed the second
// filter pass / the exact scan of every row (added up over the launches since the pack)
int knn_f16_probe(const void* src, const void* pose, const void* tgt4, const void* image, const int32_t* src_rows, const int32_t* tgt_rows,
                  int N, int n, int m, int m_pad, float* out, void* stream);
int knn_f16_brute(const void* src, const void* pose, const void* tgt4, void* image, const int32_t* src_rows, const int32_t* tgt_rows,
                  int N, int n, int m, int m_pad, int32_t* idx, void* stream);

// the exact sorted sweep (dicp_knn_sweep's plain search, units of 128 queries) with the scoring on the matrix cores; image = the image of the SORTED packed
// rows tgs4; ev0 / ev1: optional hipEvent_t carried on the dispatch.  form_in / form_out / form_tiles / form_default: optional (N) tallies of 64-row tiles per
// cloud -- given form_in the launch takes only the clouds with more than form_tiles tiles per unit in it (a cloud without a tally: if form_default); form_out is
// added to (dicp_loop_buffers.search.form)
int knn_f16_sweep(const void* src, const void* pose, const void* tgs4, void* image, const int32_t* tperm, const int32_t* qorder, const int32_t* bucket,
                  const void* brange, int nbkt, const int32_t* src_rows, const int32_t* tgt_rows, int N, int n, int m, int m_pad, int32_t* idx, int32_t* spos,
                  unsigned long long* pairs, const int32_t* form_in, int32_t* form_out, int form_tiles, int form_default, void* ev0, void* ev1, void* stream);

}  // namespace dicp_tu
