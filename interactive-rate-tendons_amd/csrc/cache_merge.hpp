// cache_merge.hpp -- device-side union of the voxel sets of an edge's FK samples: the roadmap's
// edge voxel cache (VoxelEnvironment.cpp:406-422 ORs the sampled backbones into one VoxelOctree;
// VoxelCachedLazyPRM.cpp:1751-1775 stores it per edge).  Separate translation unit (rocPRIM's sort /
// scan / reduce-by-key templates compile independently of the FK kernels).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace trk {

// Scratch owned by the merge; grows on demand, freed by merge_free.
struct MergeScratch {
  int64_t cap_items = 0, cap_nnz = 0, cap_group_items = 0, cap_group_edges = 0;
  size_t cap_tmp = 0;
  int64_t *cnt = nullptr, *offs = nullptr;        // [items + 1]
  uint64_t *keys[2] = {nullptr, nullptr};         // [nnz] (edge << id_bits) | block id
  uint64_t *vals[2] = {nullptr, nullptr};         // [nnz] masks
  uint64_t *ukeys = nullptr, *uvals = nullptr;    // [nnz] unique keys, OR-ed masks
  uint32_t *uids = nullptr;                       // [nnz] block id of each unique key
  int32_t *ecount = nullptr;                      // [edges]
  int64_t cap_edges = 0;
  uint64_t *scalars = nullptr;                    // [0] unique count, [1] overflow flag
  void *tmp = nullptr;
  // grouping of the items by edge: edge ids as sort keys, item numbers, the rank of every item, entries per edge and
  // the segment (= edge) offsets of the entry arrays
  uint32_t *ekey[2] = {nullptr, nullptr};         // [items]
  int32_t *order[2] = {nullptr, nullptr};         // [items]
  int32_t *rank = nullptr;                        // [items]
  unsigned long long *etot = nullptr;             // [edges + 1]
  unsigned int *seg = nullptr;                    // [edges + 1]
  int32_t *ibeg = nullptr, *iend = nullptr;       // [edges] every edge's run of grouped positions
};

// Per-sample block lists (ids / masks stored as columns [maxB][ld], counts[i] entries, counts < 0 =
// overflow) -> per-edge sorted, duplicate-free lists on the device:
//   ms.uids / ms.uvals [*n_unique] ordered by (edge, block id), ms.ecount[e] entries per edge.
// `pool` items: item i is the list of pool sample d_item_src[i] (i itself when d_item_src is null) taken as part of
// edge d_sample_edge[i] (< 0: skipped) -- with an explicit source a sample (a roadmap vertex) can join many edges.
// Returns hipSuccess, or an error; *overflow != 0 when a sample's list had overflowed.
hipError_t merge_edge_caches(MergeScratch &ms, const uint32_t *d_ids, const uint64_t *d_masks, const int32_t *d_counts,
                             const int32_t *d_item_src, const int32_t *d_sample_edge, int64_t pool, int64_t ld, uint32_t n_blocks,
                             int64_t n_edges, int64_t *n_unique, int *overflow, hipStream_t stream);
void merge_free(MergeScratch &ms);

// Undirected edge list of a k-nearest-neighbour table (connectVertices over every vertex's neighbours with the
// `if (!getEdge(v, n))` test of VoxelCachedLazyPRM.cpp:1491-1502): the pairs (i, idx[i][p]) with idx >= 0 and idx != i,
// each once as (lo, hi), ordered by (lo, hi).  d_idx: [n][k] on the device.  d_edges (2 x capacity int32) receives
// min(count, capacity) pairs; *n_edges the count.
hipError_t knn_edge_list(MergeScratch &ms, const int32_t *d_idx, int64_t n, int k, int32_t *d_edges, int64_t capacity,
                         int64_t *n_edges, hipStream_t stream);

// The states ordered by one weighted coordinate, for the neighbour search (knn_kernel.hpp): key = key_scale * states[i][key_col]
// -- a term of the compound metric, so |key difference| <= distance.  d_sorted [n][S] = the states in ascending order of the key,
// d_xs [n] those keys, d_perm [n] the original index of each (stable: equal keys keep their order).  All arrays on the device;
// d_keys_tmp [n] doubles and d_perm_tmp [n] int32 are scratch.
hipError_t sort_states_by_key(MergeScratch &ms, const double *d_states, int64_t n, int S, int key_col, double key_scale, double *d_sorted,
                              double *d_xs, int32_t *d_perm, double *d_keys_tmp, int32_t *d_perm_tmp, hipStream_t stream);

// The neighbour search's cell grid (knn_kernel.hpp): two coordinates of the state, each scaled to a TERM of the compound metric
// (a tension as it is, the retraction times its weight), so that |key difference| <= distance on either of them; C x B cells of
// uniform width over the keys' ranges.  knn_cell_of is monotone non-decreasing in the key, so every candidate whose key lies in
// [lo, hi] has its cell index in [cell(lo), cell(hi)]: the search visits exactly those cells.
struct KnnCells {
  int32_t col0, col1;           // state columns of the first two keys (col1 = -1: one key only)
  int32_t C, B;                 // cells along key 0 / key 1
  double scale0, scale1;        // key = scale * state[col]
  double lo0, inv0, lo1, inv1;  // cell = clamp((int)((key - lo) * inv), 0, cells - 1)
  // an optional third key (the wave-per-query search, knn_kernel.hpp: knn_wave_query): cell id = (c0 B + c1) A + c2; col2 = -1, A = 1: none
  int32_t col2, A;
  double scale2, lo2, inv2;
};
__host__ __device__ inline int knn_cell_of(double key, double lo, double inv, int cells) {
  const double t = (key - lo) * inv;
  int c = t > 0.0 ? (t < (double)cells ? (int)t : cells - 1) : 0;      // NaN -> 0
  return c < cells - 1 ? c : cells - 1;
}
// d_sorted [n][S] = the states ordered by cell id (key-0 cell major), d_perm [n] the original index of each, d_cellstart [C B + 1]
// the first sorted position of every cell.  g comes in with col / scale set; C, B (cells of about cell_width along each key's
// range in the data, at most max_cells; 1 along a key without spread), lo and inv are filled in.  d_keys: two scratch arrays of
// n uint32, d_perm_tmp n int32, d_cellstart C B A + 1 <= max_cells^3 + 1 int32 (max_cells^2 + 1 without a third key).  Synchronises `stream` once (the key ranges).
hipError_t sort_states_by_cells(MergeScratch &ms, const double *d_states, int64_t n, int S, KnnCells &g, double cell_width, int max_cells, double *d_sorted,
                                int32_t *d_perm, int32_t *d_cellstart, uint32_t *d_keys[2], int32_t *d_perm_tmp, hipStream_t stream);

// Retraction-enabled robots: the order in which the verdict-only kernel takes a batch.  A wave of the retraction kernel runs
// from its LONGEST backbone's base to the tip (tip-aligned iterations, fk_retract_kernel.hpp), shorter backbones idling
// until their rows come up: with retractions ~ U[0, L] in arrival order every wave pays for a full-length backbone and half
// of its lane-steps are masked off.  d_perm [n] = configuration indices ordered by retraction (state coordinate S - 1,
// clamped to [0, L], 256 levels: one radix pass), longest first -- a wave then holds backbones of one length.
// d_keys / d_vals: two scratch arrays of n each; *perm_out points at the one holding the result.
// d_wave_k_begin (optional, [ceil(n / 64)]): per wave of the ordered batch a step of the shared grid before which none of its backbones
// has begun (0 when the grid does not take exactly one step per row behind its first interval: one_step_per_row).
hipError_t retraction_order(MergeScratch &ms, const double *d_states, int64_t n, int S, double L, uint32_t *const d_keys[2],
                            int32_t *const d_vals[2], const int32_t **perm_out, hipStream_t stream, double dL = 0.0, int k_first = 0,
                            bool one_step_per_row = false, int32_t *d_wave_k_begin = nullptr);

}  // namespace trk
