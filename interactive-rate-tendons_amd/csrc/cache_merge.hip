// cache_merge.hip -- see cache_merge.hpp.  Pipeline, all on one stream:
//   counts -> exclusive scan -> (edge, block id) keys + masks -> radix sort on the key bits in use
//   -> reduce-by-key with OR -> block ids and per-edge counts.
// HBM-bound integer work: ~16 B per entry per sort pass, (id_bits + edge_bits) / 8 passes.
#include "cache_merge.hpp"

#include <cstdlib>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>
#include <rocprim/device/device_reduce_by_key.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>

namespace trk {
namespace {

// item i = the block list of pool sample src[i] (i itself when src is null) as part of edge item_edge[i] (< 0: of none).
// The items are first GROUPED by edge (a radix sort of the 10^6-odd (edge, item) pairs): the entries of an edge are then
// one segment of the entry arrays and only need sorting by block id WITHIN their segment -- ~300 entries, one in-LDS
// block sort of rocPRIM's segmented sort -- instead of a global sort of 10^8 (edge, block) keys over all their bits.
__global__ __launch_bounds__(256) void merge_item_keys(const int32_t *__restrict__ item_edge, int64_t n_items, uint32_t *__restrict__ ekey,
                                                       int32_t *__restrict__ order) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_items) return;
  ekey[i] = (uint32_t)item_edge[i];                 // items of no edge (< 0) sort to the end
  order[i] = (int32_t)i;
}

// p = position in grouped order: item order[p] of edge ekey[p]
__global__ __launch_bounds__(256) void merge_counts(const int32_t *__restrict__ counts, const int32_t *__restrict__ src,
                                                    const uint32_t *__restrict__ ekey, const int32_t *__restrict__ order,
                                                    int64_t n_items, int64_t *__restrict__ cnt, int32_t *__restrict__ rank,
                                                    unsigned long long *__restrict__ etot, uint64_t *__restrict__ scalars,
                                                    int32_t *__restrict__ ibeg, int32_t *__restrict__ iend) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= n_items) return;
  const int32_t i = order[p];
  rank[i] = (int32_t)p;
  if ((int32_t)ekey[p] < 0) { cnt[p] = 0; return; }
  // the edge's run of grouped positions [ibeg, iend) (both zeroed by the host: an edge without items keeps an empty run)
  if (p == 0 || ekey[p - 1] != ekey[p]) ibeg[ekey[p]] = (int32_t)p;
  if (p == n_items - 1 || ekey[p + 1] != ekey[p]) iend[ekey[p]] = (int32_t)(p + 1);
  const int c = counts[src ? src[i] : i];
  if (c < 0) scalars[1] = 1;
  cnt[p] = c > 0 ? c : 0;
  if (c > 0) atomicAdd(&etot[ekey[p]], (unsigned long long)c);
}

__global__ __launch_bounds__(256) void merge_segments(const unsigned long long *__restrict__ eoff, int64_t n, unsigned int *__restrict__ seg) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e < n) seg[e] = (unsigned int)eoff[e];
}

// Entries of 64 consecutive items (pool order: their columns are adjacent, the reads coalesce) through LDS to each item's
// run in the grouped entry arrays (32 consecutive entries per store).  Written thread-per-item as before, every 8-byte store
// opened a cache line of its own: 0.65 TB/s for a kernel that moves 5 GB.
constexpr int MK_TILE = 32;
__global__ __launch_bounds__(256) void merge_keys(const uint32_t *__restrict__ ids, const uint64_t *__restrict__ masks,
                                                  const int64_t *__restrict__ cnt, const int64_t *__restrict__ offs,
                                                  const int32_t *__restrict__ rank, const int32_t *__restrict__ src,
                                                  const int32_t *__restrict__ item_edge, int64_t n_items,
                                                  int64_t ld, int id_bits, uint64_t *__restrict__ keys, uint64_t *__restrict__ vals) {
  __shared__ uint32_t t_id[64][MK_TILE + 1];
  __shared__ uint64_t t_mask[64][MK_TILE + 1];
  __shared__ int64_t s_off[64];
  __shared__ int32_t s_cnt[64];
  __shared__ uint64_t s_hi[64];
  __shared__ int s_max;
  const int t = threadIdx.x, lane = t & 63, kq = t >> 6;
  const int64_t i0 = (int64_t)blockIdx.x * 64;
  if (t == 0) s_max = 0;
  __syncthreads();
  int64_t col = 0;
  int c = 0;
  {
    const int64_t i = i0 + lane;
    if (i < n_items) {
      const int64_t p = rank[i];
      c = (int)cnt[p];
      col = src ? src[i] : i;
      if (kq == 0) { s_off[lane] = offs[p]; s_cnt[lane] = c; s_hi[lane] = (uint64_t)(uint32_t)item_edge[i] << id_bits; }
    } else if (kq == 0) { s_cnt[lane] = 0; s_off[lane] = 0; s_hi[lane] = 0; }
    if (kq == 0 && c > 0) atomicMax(&s_max, c);
  }
  __syncthreads();
  const int maxc = s_max;
  for (int k0 = 0; k0 < maxc; k0 += MK_TILE) {
    for (int k = k0 + kq; k < k0 + MK_TILE; k += 4)
      if (k < c) { t_id[lane][k - k0] = ids[(int64_t)k * ld + col]; t_mask[lane][k - k0] = masks[(int64_t)k * ld + col]; }
    __syncthreads();
    // wave kq writes the runs of items kq, kq + 4, ...: two items per pass, 32 entries each
    for (int r = kq * 2 + (lane >> 5); r < 64; r += 8) {
      const int k = k0 + (lane & 31);
      if (k < s_cnt[r]) {
        keys[s_off[r] + k] = s_hi[r] | t_id[r][lane & 31];
        vals[s_off[r] + k] = t_mask[r][lane & 31];
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void merge_finish(const uint64_t *__restrict__ ukeys, const uint64_t *__restrict__ scalars,
                                                    int id_bits, uint32_t *__restrict__ uids, int32_t *__restrict__ ecount) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)scalars[0]) return;
  const uint64_t k = ukeys[i];
  uids[i] = (uint32_t)(k & (((uint64_t)1 << id_bits) - 1));
  atomicAdd(&ecount[k >> id_bits], 1);
}

__global__ __launch_bounds__(256) void knn_edge_keys(const int32_t *__restrict__ idx, int64_t n, int k, uint64_t *__restrict__ keys) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n * k) return;
  const int64_t i = t / k;
  const int64_t j = idx[t];
  // entries without a neighbour (and the vertex itself) sort to the end and are cut off
  keys[t] = (j < 0 || j == i) ? ~0ull : ((uint64_t)(i < j ? i : j) << 32) | (uint64_t)(i < j ? j : i);
}

__global__ __launch_bounds__(256) void knn_edge_unpack(const uint64_t *__restrict__ keys, int64_t m, int32_t *__restrict__ edges) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= m) return;
  edges[2 * t] = (int32_t)(keys[t] >> 32);
  edges[2 * t + 1] = (int32_t)(keys[t] & 0xffffffffull);
}


// The union of an edge's block lists in ONE kernel, a wave per edge: every (block id, mask) entry of the edge's items goes
// into an open-addressing hash table in LDS (compare-and-swap claims the slot of a block id, an atomic OR adds the mask),
// the occupied slots are gathered, ordered by block id with a bitonic sort in LDS and written as the edge's list into its
// segment of the entry arrays (the segment holds all its entries, so the duplicate-free list fits).  It replaces the gather
// of the entries into segments, the segmented radix sort and the reduce-by-key (~11 ms of a 588 k-edge roadmap's 17 ms of
// voxelise-and-merge; the sort path below stays as the fallback for an edge with more than EU_MAXLOAD distinct blocks).
constexpr int EU_SLOTS = 512, EU_MAXLOAD = 384;
constexpr uint32_t EU_EMPTY = 0xffffffffu;          // no block has this id (ids < Nb^3 <= 2^30)

__global__ __launch_bounds__(64) void edge_union(const uint32_t *__restrict__ ids, const uint64_t *__restrict__ masks, const int32_t *__restrict__ counts,
                                                 const int32_t *__restrict__ src, const int32_t *__restrict__ ibeg, const int32_t *__restrict__ iend,
                                                 const int32_t *__restrict__ order, int64_t ld, const unsigned long long *__restrict__ eoff_in, uint32_t *__restrict__ tmp_ids,
                                                 uint64_t *__restrict__ tmp_masks, int32_t *__restrict__ ecount, uint64_t *__restrict__ scalars,
                                                 int max_load) {
  __shared__ uint32_t tkey[EU_SLOTS];
  __shared__ unsigned long long tmask[EU_SLOTS];
  __shared__ uint32_t lid[EU_SLOTS];
  __shared__ unsigned long long lmask[EU_SLOTS];
  __shared__ int s_unique, s_ovf;
  const int64_t e = blockIdx.x;
  const int lane = threadIdx.x;
  for (int s = lane; s < EU_SLOTS; s += 64) { tkey[s] = EU_EMPTY; tmask[s] = 0ull; }
  if (lane == 0) { s_unique = 0; s_ovf = 0; }
  __syncthreads();
  const int64_t p0 = ibeg[e], p1 = iend[e];         // (a binary search of the grouped keys here was most of the kernel: 44 dependent loads per wave)
  // eight items at a time, eight lanes each (an item holds ~35 entries: eight lanes waste few turns on its last round, and
  // the items of an edge are mostly neighbouring pool slots -- neighbouring columns, so one row of eight items is one 32-byte read)
  const int grp = lane >> 3, gl = lane & 7;
  for (int64_t p = p0 + grp; p < p1; p += 8) {
    const int32_t i = order[p];
    const int64_t col = src ? src[i] : i;
    const int c = counts[col];
    for (int k = gl; k < c; k += 8) {
      const uint32_t id = ids[(int64_t)k * ld + col];
      const unsigned long long m = masks[(int64_t)k * ld + col];
      uint32_t h = (id * 2654435761u) >> 23;                       // 9 bits
      int probe = 0;
      for (; probe < EU_SLOTS; probe++) {
        const uint32_t old = atomicCAS(&tkey[h], EU_EMPTY, id);
        if (old == EU_EMPTY) atomicAdd(&s_unique, 1);
        if (old == EU_EMPTY || old == id) { atomicOr(&tmask[h], m); break; }
        h = (h + 1) & (EU_SLOTS - 1);
      }
      if (probe == EU_SLOTS) s_ovf = 1;                             // the table is full
    }
  }
  __syncthreads();
  if (s_ovf != 0 || s_unique > max_load) {
    if (lane == 0) { ecount[e] = 0; atomicOr((unsigned long long *)&scalars[1], 2ull); }
    return;
  }
  // the occupied slots, densely
  int n = 0;
  for (int s0 = 0; s0 < EU_SLOTS; s0 += 64) {
    const uint32_t key = tkey[s0 + lane];
    const bool occ = key != EU_EMPTY;
    const unsigned long long b = __ballot(occ);
    if (occ) {
      const int pos = n + __popcll(b & (((unsigned long long)1 << lane) - 1));
      lid[pos] = key; lmask[pos] = tmask[s0 + lane];
    }
    n += __popcll(b);
  }
  int np2 = 1;
  while (np2 < n) np2 <<= 1;
  for (int t = n + lane; t < np2; t += 64) lid[t] = EU_EMPTY;        // padding sorts to the end
  __syncthreads();
  for (int k = 2; k <= np2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = lane; t < (np2 >> 1); t += 64) {
        const int a = ((t & ~(j - 1)) << 1) | (t & (j - 1)), b = a | j;
        const bool up = (a & k) == 0;
        const uint32_t ia = lid[a], ib = lid[b];
        if ((ia > ib) == up) {
          const unsigned long long ma = lmask[a], mb = lmask[b];
          lid[a] = ib; lid[b] = ia; lmask[a] = mb; lmask[b] = ma;
        }
      }
      __syncthreads();
    }
  const int64_t base = (int64_t)eoff_in[e];
  for (int t = lane; t < n; t += 64) { tmp_ids[base + t] = lid[t]; tmp_masks[base + t] = lmask[t]; }
  if (lane == 0) ecount[e] = n;
}

// the edges' lists from their segments to their final, contiguous places: a wave per edge
__global__ __launch_bounds__(64) void edge_union_compact(const uint32_t *__restrict__ tmp_ids, const uint64_t *__restrict__ tmp_masks,
                                                         const unsigned long long *__restrict__ eoff_in, const int32_t *__restrict__ ecount,
                                                         const int64_t *__restrict__ eoff_out, uint32_t *__restrict__ uids, uint64_t *__restrict__ uvals) {
  const int64_t e = blockIdx.x;
  const int n = ecount[e];
  const int64_t from = (int64_t)eoff_in[e], to = eoff_out[e];
  for (int t = threadIdx.x; t < n; t += 64) { uids[to + t] = tmp_ids[from + t]; uvals[to + t] = tmp_masks[from + t]; }
}

struct BitOr { __host__ __device__ uint64_t operator()(uint64_t a, uint64_t b) const { return a | b; } };

int bits_for(uint64_t n) { int b = 1; while (((uint64_t)1 << b) < n) b++; return b; }

template <typename T>
hipError_t grow(T **p, size_t count) {
  if (*p) { (void)hipFree(*p); *p = nullptr; }
  return hipMalloc((void **)p, (count ? count : 1) * sizeof(T));
}

#define MERGE_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return e_; } while (0)

}  // namespace

void merge_free(MergeScratch &ms) {
  void *ptrs[] = {ms.cnt, ms.offs, ms.keys[0], ms.keys[1], ms.vals[0], ms.vals[1], ms.ukeys, ms.uvals, ms.uids, ms.ecount, ms.scalars, ms.tmp,
                  ms.ekey[0], ms.ekey[1], ms.order[0], ms.order[1], ms.rank, ms.etot, ms.seg, ms.ibeg, ms.iend};
  for (void *p : ptrs) if (p) (void)hipFree(p);
  ms = MergeScratch{};
}

hipError_t merge_edge_caches(MergeScratch &ms, const uint32_t *d_ids, const uint64_t *d_masks, const int32_t *d_counts,
                             const int32_t *d_item_src, const int32_t *d_sample_edge, int64_t pool, int64_t ld, uint32_t n_blocks,
                             int64_t n_edges, int64_t *n_unique, int *overflow, hipStream_t stream) {
  *n_unique = 0; *overflow = 0;
  if (pool <= 0 || n_edges <= 0) return hipSuccess;
  const int id_bits = bits_for(n_blocks), e_bits = bits_for((uint64_t)n_edges);
  if (id_bits + e_bits > 64) return hipErrorInvalidValue;
  if (ms.cap_items < pool + 1) {
    MERGE_TRY(grow(&ms.cnt, (size_t)pool + 1));
    MERGE_TRY(grow(&ms.offs, (size_t)pool + 1));
    ms.cap_items = pool + 1;
  }
  if (ms.cap_edges < n_edges + 1) { MERGE_TRY(grow(&ms.ecount, (size_t)n_edges + 1)); ms.cap_edges = n_edges + 1; }
  if (!ms.scalars) MERGE_TRY(grow(&ms.scalars, 2));
  auto need_tmp = [&](size_t bytes) -> hipError_t {
    if (ms.cap_tmp >= bytes) return hipSuccess;
    MERGE_TRY(grow((char **)&ms.tmp, bytes + bytes / 4));
    ms.cap_tmp = bytes + bytes / 4;
    return hipSuccess;
  };
  const dim3 gp((unsigned)((pool + 255) / 256)), b256(256);
  if (ms.cap_group_items < pool) {
    for (int b = 0; b < 2; b++) { MERGE_TRY(grow(&ms.ekey[b], (size_t)pool)); MERGE_TRY(grow(&ms.order[b], (size_t)pool)); }
    MERGE_TRY(grow(&ms.rank, (size_t)pool));
    ms.cap_group_items = pool;
  }
  if (ms.cap_group_edges < n_edges + 1) {
    MERGE_TRY(grow(&ms.etot, (size_t)n_edges + 1));
    MERGE_TRY(grow(&ms.seg, (size_t)n_edges + 1));
    MERGE_TRY(grow(&ms.ibeg, (size_t)n_edges + 1));
    MERGE_TRY(grow(&ms.iend, (size_t)n_edges + 1));
    ms.cap_group_edges = n_edges + 1;
  }

  MERGE_TRY(hipMemsetAsync(ms.scalars, 0, 2 * sizeof(uint64_t), stream));
  MERGE_TRY(hipMemsetAsync(ms.ecount, 0, (size_t)n_edges * sizeof(int32_t), stream));
  MERGE_TRY(hipMemsetAsync(ms.cnt + pool, 0, sizeof(int64_t), stream));
  MERGE_TRY(hipMemsetAsync(ms.etot, 0, ((size_t)n_edges + 1) * sizeof(unsigned long long), stream));
  MERGE_TRY(hipMemsetAsync(ms.ibeg, 0, (size_t)n_edges * sizeof(int32_t), stream));
  MERGE_TRY(hipMemsetAsync(ms.iend, 0, (size_t)n_edges * sizeof(int32_t), stream));
  // group the items by edge
  hipLaunchKernelGGL(merge_item_keys, gp, b256, 0, stream, d_sample_edge, pool, ms.ekey[0], ms.order[0]);
  MERGE_TRY(hipGetLastError());
  size_t bytes = 0;
  rocprim::double_buffer<uint32_t> ekb(ms.ekey[0], ms.ekey[1]);
  rocprim::double_buffer<int32_t> orb(ms.order[0], ms.order[1]);
  MERGE_TRY(rocprim::radix_sort_pairs(nullptr, bytes, ekb, orb, (size_t)pool, 0u, 32u, stream));
  MERGE_TRY(need_tmp(bytes));
  MERGE_TRY(rocprim::radix_sort_pairs(ms.tmp, bytes, ekb, orb, (size_t)pool, 0u, 32u, stream));
  hipLaunchKernelGGL(merge_counts, gp, b256, 0, stream, d_counts, d_item_src, ekb.current(), orb.current(), pool, ms.cnt, ms.rank, ms.etot,
                     ms.scalars, ms.ibeg, ms.iend);
  MERGE_TRY(hipGetLastError());
  bytes = 0;
  MERGE_TRY(rocprim::exclusive_scan(nullptr, bytes, ms.cnt, ms.offs, (int64_t)0, (size_t)pool + 1, rocprim::plus<int64_t>(), stream));
  MERGE_TRY(need_tmp(bytes));
  MERGE_TRY(rocprim::exclusive_scan(ms.tmp, bytes, ms.cnt, ms.offs, (int64_t)0, (size_t)pool + 1, rocprim::plus<int64_t>(), stream));
  bytes = 0;
  MERGE_TRY(rocprim::exclusive_scan(nullptr, bytes, ms.etot, ms.etot, 0ull, (size_t)n_edges + 1, rocprim::plus<unsigned long long>(), stream));
  MERGE_TRY(need_tmp(bytes));
  MERGE_TRY(rocprim::exclusive_scan(ms.tmp, bytes, ms.etot, ms.etot, 0ull, (size_t)n_edges + 1, rocprim::plus<unsigned long long>(), stream));
  hipLaunchKernelGGL(merge_segments, dim3((unsigned)((n_edges + 1 + 255) / 256)), b256, 0, stream, ms.etot, n_edges + 1, ms.seg);
  MERGE_TRY(hipGetLastError());
  int64_t nnz = 0;
  uint64_t sc[2];
  MERGE_TRY(hipMemcpyAsync(&nnz, ms.offs + pool, sizeof(int64_t), hipMemcpyDeviceToHost, stream));
  MERGE_TRY(hipMemcpyAsync(sc, ms.scalars, sizeof(sc), hipMemcpyDeviceToHost, stream));
  MERGE_TRY(hipStreamSynchronize(stream));
  if (sc[1]) { *overflow = 1; return hipSuccess; }
  if (nnz == 0) return hipSuccess;
  if (nnz >= ((int64_t)1 << 32)) return hipErrorInvalidValue;          // segment offsets are 32-bit (the callers' chunks stay far below)
  if (ms.cap_nnz < nnz) {
    const size_t want = (size_t)nnz + (size_t)nnz / 4 + 1024;
    for (int b = 0; b < 2; b++) { MERGE_TRY(grow(&ms.keys[b], want)); MERGE_TRY(grow(&ms.vals[b], want)); }
    MERGE_TRY(grow(&ms.ukeys, want));
    MERGE_TRY(grow(&ms.uvals, want));
    MERGE_TRY(grow(&ms.uids, want));
    ms.cap_nnz = (int64_t)want;
  }
  // the union kernel (TENDON_HIP_MERGE=sort: the sort path only; it also takes over when an edge overflows the kernel's table)
  const char *merge_mode = std::getenv("TENDON_HIP_MERGE");          // (read per call: the tests switch between the two paths)
  const bool sort_only = merge_mode && std::strcmp(merge_mode, "sort") == 0;
  if (!sort_only && n_edges < ((int64_t)1 << 31) && pool + 1 >= n_edges + 1) {
    uint32_t *tmp_ids = (uint32_t *)ms.keys[0];
    uint64_t *tmp_masks = ms.vals[0];
    int max_load = EU_MAXLOAD;
    if (const char *e = std::getenv("TENDON_HIP_MERGE_MAXLOAD")) {     // (tests: a small value sends the call through the overflow fallback)
      const int v = std::atoi(e);
      if (v >= 1 && v < EU_MAXLOAD) max_load = v;
    }
    hipLaunchKernelGGL(edge_union, dim3((unsigned)n_edges), dim3(64), 0, stream, d_ids, d_masks, d_counts, d_item_src, ms.ibeg, ms.iend, orb.current(),
                       ld, ms.etot, tmp_ids, tmp_masks, ms.ecount, ms.scalars, max_load);
    MERGE_TRY(hipGetLastError());
    // final offsets of the edges' lists (ms.cnt is free again: [pool + 1] >= [n_edges + 1]; its element n_edges is the total)
    MERGE_TRY(hipMemsetAsync(ms.ecount + n_edges, 0, sizeof(int32_t), stream));
    bytes = 0;
    MERGE_TRY(rocprim::exclusive_scan(nullptr, bytes, ms.ecount, ms.cnt, (int64_t)0, (size_t)n_edges + 1, rocprim::plus<int64_t>(), stream));
    MERGE_TRY(need_tmp(bytes));
    MERGE_TRY(rocprim::exclusive_scan(ms.tmp, bytes, ms.ecount, ms.cnt, (int64_t)0, (size_t)n_edges + 1, rocprim::plus<int64_t>(), stream));
    hipLaunchKernelGGL(edge_union_compact, dim3((unsigned)n_edges), dim3(64), 0, stream, tmp_ids, tmp_masks, ms.etot, ms.ecount, ms.cnt, ms.uids, ms.uvals);
    MERGE_TRY(hipGetLastError());
    int64_t total = 0;
    MERGE_TRY(hipMemcpyAsync(&total, ms.cnt + n_edges, sizeof(int64_t), hipMemcpyDeviceToHost, stream));
    MERGE_TRY(hipMemcpyAsync(sc, ms.scalars, sizeof(sc), hipMemcpyDeviceToHost, stream));
    MERGE_TRY(hipStreamSynchronize(stream));
    if (!(sc[1] & 2ull)) { *n_unique = total; return hipSuccess; }
    // an edge with more distinct blocks than the table takes: everything again on the sort path (ms.cnt is rebuilt below)
    MERGE_TRY(hipMemsetAsync(ms.scalars, 0, 2 * sizeof(uint64_t), stream));
    MERGE_TRY(hipMemsetAsync(ms.ecount, 0, (size_t)n_edges * sizeof(int32_t), stream));
    MERGE_TRY(hipMemsetAsync(ms.etot, 0, ((size_t)n_edges + 1) * sizeof(unsigned long long), stream));
    hipLaunchKernelGGL(merge_counts, gp, b256, 0, stream, d_counts, d_item_src, ekb.current(), orb.current(), pool, ms.cnt, ms.rank, ms.etot, ms.scalars,
                       ms.ibeg, ms.iend);
    MERGE_TRY(hipGetLastError());
    MERGE_TRY(hipMemsetAsync(ms.cnt + pool, 0, sizeof(int64_t), stream));
    bytes = 0;
    MERGE_TRY(rocprim::exclusive_scan(nullptr, bytes, ms.etot, ms.etot, 0ull, (size_t)n_edges + 1, rocprim::plus<unsigned long long>(), stream));
    MERGE_TRY(need_tmp(bytes));
    MERGE_TRY(rocprim::exclusive_scan(ms.tmp, bytes, ms.etot, ms.etot, 0ull, (size_t)n_edges + 1, rocprim::plus<unsigned long long>(), stream));
  }
  hipLaunchKernelGGL(merge_keys, dim3((unsigned)((pool + 63) / 64)), b256, 0, stream, d_ids, d_masks, ms.cnt, ms.offs, ms.rank, d_item_src,
                     d_sample_edge, pool, ld, id_bits, ms.keys[0], ms.vals[0]);
  MERGE_TRY(hipGetLastError());

  // every edge's entries by block id: the edge part of the key is the same within a segment
  rocprim::double_buffer<uint64_t> kb(ms.keys[0], ms.keys[1]), vb(ms.vals[0], ms.vals[1]);
  bytes = 0;
  MERGE_TRY(rocprim::segmented_radix_sort_pairs(nullptr, bytes, kb, vb, (unsigned int)nnz, (unsigned int)n_edges, ms.seg, ms.seg + 1, 0u,
                                                (unsigned)id_bits, stream));
  MERGE_TRY(need_tmp(bytes));
  MERGE_TRY(rocprim::segmented_radix_sort_pairs(ms.tmp, bytes, kb, vb, (unsigned int)nnz, (unsigned int)n_edges, ms.seg, ms.seg + 1, 0u,
                                                (unsigned)id_bits, stream));

  bytes = 0;
  MERGE_TRY(rocprim::reduce_by_key(nullptr, bytes, kb.current(), vb.current(), (size_t)nnz, ms.ukeys, ms.uvals, ms.scalars, BitOr(),
                                   rocprim::equal_to<uint64_t>(), stream));
  MERGE_TRY(need_tmp(bytes));
  MERGE_TRY(rocprim::reduce_by_key(ms.tmp, bytes, kb.current(), vb.current(), (size_t)nnz, ms.ukeys, ms.uvals, ms.scalars, BitOr(),
                                   rocprim::equal_to<uint64_t>(), stream));
  hipLaunchKernelGGL(merge_finish, dim3((unsigned)((nnz + 255) / 256)), b256, 0, stream, ms.ukeys, ms.scalars, id_bits, ms.uids, ms.ecount);
  MERGE_TRY(hipGetLastError());
  MERGE_TRY(hipMemcpyAsync(sc, ms.scalars, sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
  MERGE_TRY(hipStreamSynchronize(stream));
  *n_unique = (int64_t)sc[0];
  return hipSuccess;
}

hipError_t knn_edge_list(MergeScratch &ms, const int32_t *d_idx, int64_t n, int k, int32_t *d_edges, int64_t capacity,
                         int64_t *n_edges, hipStream_t stream) {
  *n_edges = 0;
  const int64_t nnz = n * (int64_t)k;
  if (nnz <= 0) return hipSuccess;
  if (ms.cap_nnz < nnz) {
    const size_t want = (size_t)nnz + (size_t)nnz / 4 + 1024;
    for (int b = 0; b < 2; b++) { MERGE_TRY(grow(&ms.keys[b], want)); MERGE_TRY(grow(&ms.vals[b], want)); }
    MERGE_TRY(grow(&ms.ukeys, want));
    MERGE_TRY(grow(&ms.uvals, want));
    MERGE_TRY(grow(&ms.uids, want));
    ms.cap_nnz = (int64_t)want;
  }
  if (!ms.scalars) MERGE_TRY(grow(&ms.scalars, 2));
  auto need_tmp = [&](size_t bytes) -> hipError_t {
    if (ms.cap_tmp >= bytes) return hipSuccess;
    MERGE_TRY(grow((char **)&ms.tmp, bytes + bytes / 4));
    ms.cap_tmp = bytes + bytes / 4;
    return hipSuccess;
  };
  const dim3 b256(256), g((unsigned)((nnz + 255) / 256));
  hipLaunchKernelGGL(knn_edge_keys, g, b256, 0, stream, d_idx, n, k, ms.keys[0]);
  MERGE_TRY(hipGetLastError());
  rocprim::double_buffer<uint64_t> kb(ms.keys[0], ms.keys[1]);
  size_t bytes = 0;
  MERGE_TRY(rocprim::radix_sort_keys(nullptr, bytes, kb, (size_t)nnz, 0u, 64u, stream));
  MERGE_TRY(need_tmp(bytes));
  MERGE_TRY(rocprim::radix_sort_keys(ms.tmp, bytes, kb, (size_t)nnz, 0u, 64u, stream));
  bytes = 0;
  MERGE_TRY(rocprim::unique(nullptr, bytes, kb.current(), ms.ukeys, ms.scalars, (size_t)nnz, rocprim::equal_to<uint64_t>(), stream));
  MERGE_TRY(need_tmp(bytes));
  MERGE_TRY(rocprim::unique(ms.tmp, bytes, kb.current(), ms.ukeys, ms.scalars, (size_t)nnz, rocprim::equal_to<uint64_t>(), stream));
  uint64_t cnt = 0, last = 0;
  MERGE_TRY(hipMemcpyAsync(&cnt, ms.scalars, sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
  MERGE_TRY(hipStreamSynchronize(stream));
  if (cnt > 0) {                                   // the all-ones key (no neighbour / self), if present, is the last one
    MERGE_TRY(hipMemcpyAsync(&last, ms.ukeys + (cnt - 1), sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
    MERGE_TRY(hipStreamSynchronize(stream));
    if (last == ~0ull) cnt--;
  }
  *n_edges = (int64_t)cnt;
  const int64_t m = (int64_t)cnt < capacity ? (int64_t)cnt : capacity;
  if (m > 0) {
    hipLaunchKernelGGL(knn_edge_unpack, dim3((unsigned)((m + 255) / 256)), b256, 0, stream, ms.ukeys, m, d_edges);
    MERGE_TRY(hipGetLastError());
  }
  return hipSuccess;
}


namespace {
__global__ __launch_bounds__(256) void key_coordinate_keys(const double *__restrict__ states, int64_t n, int S, int col, double scale,
                                                           double *__restrict__ keys, int32_t *__restrict__ vals) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { keys[i] = scale * states[i * S + col]; vals[i] = (int32_t)i; }       // the product the search kernel forms for its queries
}
__global__ __launch_bounds__(256) void gather_states(const double *__restrict__ states, const int32_t *__restrict__ perm, int64_t n, int S,
                                                     double *__restrict__ sorted) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t < n * S) { const int64_t j = t / S; sorted[t] = states[(int64_t)perm[j] * S + (t - j * S)]; }
}
}  // namespace

hipError_t sort_states_by_key(MergeScratch &ms, const double *d_states, int64_t n, int S, int key_col, double key_scale, double *d_sorted,
                              double *d_xs, int32_t *d_perm, double *d_keys_tmp, int32_t *d_perm_tmp, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(key_coordinate_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_states, n, S, key_col, key_scale,
                     d_keys_tmp, d_perm_tmp);
  MERGE_TRY(hipGetLastError());
  size_t bytes = 0;
  MERGE_TRY(rocprim::radix_sort_pairs(nullptr, bytes, d_keys_tmp, d_xs, d_perm_tmp, d_perm, (size_t)n, 0u, 64u, stream));
  if (ms.cap_tmp < bytes) {
    MERGE_TRY(grow((char **)&ms.tmp, bytes + bytes / 4));
    ms.cap_tmp = bytes + bytes / 4;
  }
  MERGE_TRY(rocprim::radix_sort_pairs(ms.tmp, bytes, d_keys_tmp, d_xs, d_perm_tmp, d_perm, (size_t)n, 0u, 64u, stream));
  hipLaunchKernelGGL(gather_states, dim3((unsigned)((n * S + 255) / 256)), dim3(256), 0, stream, d_states, d_perm, n, S, d_sorted);
  return hipGetLastError();
}

// ---- the neighbour search's cell grid (knn_kernel.hpp) ------------------------------------------------------------------
namespace {
// order-preserving map of a double onto uint64 (for atomic min / max)
__device__ __forceinline__ unsigned long long ordered_bits(double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__host__ __device__ __forceinline__ double from_ordered_bits(unsigned long long o) {
  const unsigned long long b = (o >> 63) ? (o & 0x7fffffffffffffffull) : ~o;
  double v;
  __builtin_memcpy(&v, &b, 8);
  return v;
}
// min / max of the key coordinates over all states (NaN keys are skipped: they land in cell 0 and are never within a radius)
__global__ __launch_bounds__(256) void key_ranges(const double *__restrict__ states, int64_t n, int S, KnnCells g,
                                                  unsigned long long *__restrict__ mm /* [6]: min0, max0, min1, max1, min2, max2 as ordered bits */) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  double lo[3] = {1.0 / 0.0, 1.0 / 0.0, 1.0 / 0.0}, hi[3] = {-1.0 / 0.0, -1.0 / 0.0, -1.0 / 0.0};
  if (i < n) {
    const int col[3] = {g.col0, g.col1, g.col2};
    const double sc[3] = {g.scale0, g.scale1, g.scale2};
#pragma unroll
    for (int a = 0; a < 3; a++)
      if (col[a] >= 0) { const double kx = sc[a] * states[i * S + col[a]]; if (kx == kx) { lo[a] = kx; hi[a] = kx; } }
  }
#pragma unroll
  for (int a = 0; a < 3; a++)
    for (int o = 32; o > 0; o >>= 1) { lo[a] = fmin(lo[a], __shfl_xor(lo[a], o, 64)); hi[a] = fmax(hi[a], __shfl_xor(hi[a], o, 64)); }
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int a = 0; a < 3; a++) { atomicMin(&mm[2 * a], ordered_bits(lo[a])); atomicMax(&mm[2 * a + 1], ordered_bits(hi[a])); }
  }
}
__global__ __launch_bounds__(256) void cell_keys(const double *__restrict__ states, int64_t n, int S, KnnCells g, uint32_t *__restrict__ keys,
                                                 int32_t *__restrict__ vals) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double k0 = g.scale0 * states[i * S + g.col0];
  const double k1 = g.col1 >= 0 ? g.scale1 * states[i * S + g.col1] : 0.0;
  const double k2 = g.col2 >= 0 ? g.scale2 * states[i * S + g.col2] : 0.0;
  keys[i] = (uint32_t)((knn_cell_of(k0, g.lo0, g.inv0, g.C) * g.B + knn_cell_of(k1, g.lo1, g.inv1, g.B)) * g.A + knn_cell_of(k2, g.lo2, g.inv2, g.A));
  vals[i] = (int32_t)i;
}
// cellstart[c] = first sorted position whose cell id is >= c (c = 0 .. C B A)
__global__ __launch_bounds__(256) void cell_starts(const uint32_t *__restrict__ sorted_keys, int64_t n, int n_cells, int32_t *__restrict__ cellstart) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c > n_cells) return;
  int64_t a = 0, b = n;
  while (a < b) { const int64_t mid = (a + b) >> 1; if (sorted_keys[mid] < (uint32_t)c) a = mid + 1; else b = mid; }
  cellstart[c] = (int32_t)a;
}
}  // namespace

hipError_t sort_states_by_cells(MergeScratch &ms, const double *d_states, int64_t n, int S, KnnCells &g, double cell_width, int max_cells, double *d_sorted,
                                int32_t *d_perm, int32_t *d_cellstart, uint32_t *d_keys[2], int32_t *d_perm_tmp, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  if (!ms.scalars) MERGE_TRY(grow(&ms.scalars, 2));
  // key ranges: six ordered-bit words in a scratch of their own (the merge's scalars hold only two)
  unsigned long long *mm = nullptr;
  MERGE_TRY(hipMalloc((void **)&mm, 6 * sizeof(unsigned long long)));
  const unsigned long long init[6] = {~0ull, 0ull, ~0ull, 0ull, ~0ull, 0ull};
  hipError_t e = hipMemcpyAsync(mm, init, sizeof(init), hipMemcpyHostToDevice, stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(key_ranges, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_states, n, S, g, mm);
    e = hipGetLastError();
  }
  unsigned long long got[6];
  if (e == hipSuccess) e = hipMemcpyAsync(got, mm, sizeof(got), hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  (void)hipFree(mm);
  MERGE_TRY(e);
  const double a0 = from_ordered_bits(got[0]), b0 = from_ordered_bits(got[1]), a1 = from_ordered_bits(got[2]), b1 = from_ordered_bits(got[3]);
  const double a2 = from_ordered_bits(got[4]), b2 = from_ordered_bits(got[5]);
  auto cells = [&](double lo, double hi) {
    if (!(hi > lo) || !(cell_width > 0.0)) return 1;
    const double c = std::ceil((hi - lo) / cell_width);
    return (int)(c < 1.0 ? 1.0 : (c > (double)max_cells ? (double)max_cells : c));
  };
  g.C = cells(a0, b0);
  g.B = g.col1 >= 0 ? cells(a1, b1) : 1;
  g.A = g.col2 >= 0 ? cells(a2, b2) : 1;
  g.lo0 = (b0 >= a0) ? a0 : 0.0; g.inv0 = g.C > 1 ? (double)g.C / (b0 - a0) : 0.0;
  g.lo1 = (b1 >= a1) ? a1 : 0.0; g.inv1 = g.B > 1 ? (double)g.B / (b1 - a1) : 0.0;
  g.lo2 = (b2 >= a2) ? a2 : 0.0; g.inv2 = g.A > 1 ? (double)g.A / (b2 - a2) : 0.0;
  hipLaunchKernelGGL(cell_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_states, n, S, g, d_keys[0], d_perm_tmp);
  MERGE_TRY(hipGetLastError());
  int bits = 1;
  while ((1 << bits) < g.C * g.B * g.A) bits++;
  size_t bytes = 0;
  MERGE_TRY(rocprim::radix_sort_pairs(nullptr, bytes, d_keys[0], d_keys[1], d_perm_tmp, d_perm, (size_t)n, 0u, (unsigned)bits, stream));
  if (ms.cap_tmp < bytes) {
    MERGE_TRY(grow((char **)&ms.tmp, bytes + bytes / 4));
    ms.cap_tmp = bytes + bytes / 4;
  }
  MERGE_TRY(rocprim::radix_sort_pairs(ms.tmp, bytes, d_keys[0], d_keys[1], d_perm_tmp, d_perm, (size_t)n, 0u, (unsigned)bits, stream));
  hipLaunchKernelGGL(cell_starts, dim3((unsigned)((g.C * g.B * g.A + 1 + 255) / 256)), dim3(256), 0, stream, d_keys[1], n, g.C * g.B * g.A, d_cellstart);
  hipLaunchKernelGGL(gather_states, dim3((unsigned)((n * S + 255) / 256)), dim3(256), 0, stream, d_states, d_perm, n, S, d_sorted);
  return hipGetLastError();
}

namespace {
__global__ __launch_bounds__(256) void retraction_keys(const double *__restrict__ states, int64_t n, int S, double inv_L, uint32_t *__restrict__ keys,
                                                       int32_t *__restrict__ vals) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double f = states[i * S + (S - 1)] * inv_L;
  f = f > 0.0 ? (f < 1.0 ? f : 1.0) : 0.0;          // NaN -> 0
  keys[i] = (uint32_t)(f * 255.0);
  vals[i] = (int32_t)i;
}
}  // namespace

// wave w of the ordered batch holds sorted positions [64 w, 64 w + 64): its smallest level bounds every retraction in it from below,
// i.e. its longest backbone from above: s_start >= level / 255 * L (the key is the truncation of s / L * 255).  Rows are spaced dL
// from the tip, a backbone starting at s has its base in row >= floor(s / dL) - 1 (the first interval is at most 1.5 dL long), and
// behind the grid's own first interval step k_first + r - 2 ends in row r: nothing of the wave happens before step
// k_first + floor(s_low / dL) - 3.
__global__ __launch_bounds__(256) void retraction_wave_begin(const uint32_t *__restrict__ sorted_keys, int64_t n, double L, double dL, int k_first,
                                                             int32_t *__restrict__ out) {
  const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (w * 64 >= n) return;
  const double s_low = (double)sorted_keys[w * 64] * (1.0 / 255.0) * L;       // ascending order: the wave's first key is its smallest
  const int rows = (int)(s_low / dL) - 3;
  out[w] = rows > 0 ? k_first + rows : 0;
}

hipError_t retraction_order(MergeScratch &ms, const double *d_states, int64_t n, int S, double L, uint32_t *const d_keys[2],
                            int32_t *const d_vals[2], const int32_t **perm_out, hipStream_t stream, double dL, int k_first, bool one_step_per_row,
                            int32_t *d_wave_k_begin) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(retraction_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_states, n, S, 1.0 / L, d_keys[0], d_vals[0]);
  MERGE_TRY(hipGetLastError());
  rocprim::double_buffer<uint32_t> kb(d_keys[0], d_keys[1]);
  rocprim::double_buffer<int32_t> vb(d_vals[0], d_vals[1]);
  size_t bytes = 0;
  MERGE_TRY(rocprim::radix_sort_pairs(nullptr, bytes, kb, vb, (size_t)n, 0u, 8u, stream));
  if (ms.cap_tmp < bytes) {
    MERGE_TRY(grow((char **)&ms.tmp, bytes + bytes / 4));
    ms.cap_tmp = bytes + bytes / 4;
  }
  MERGE_TRY(rocprim::radix_sort_pairs(ms.tmp, bytes, kb, vb, (size_t)n, 0u, 8u, stream));
  *perm_out = vb.current();
  if (d_wave_k_begin) {
    const int64_t nw = (n + 63) / 64;
    if (one_step_per_row) {
      hipLaunchKernelGGL(retraction_wave_begin, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, stream, kb.current(), n, L, dL, k_first, d_wave_k_begin);
      MERGE_TRY(hipGetLastError());
    } else {
      MERGE_TRY(hipMemsetAsync(d_wave_k_begin, 0, (size_t)nw * sizeof(int32_t), stream));
    }
  }
  return hipSuccess;
}

}  // namespace trk
