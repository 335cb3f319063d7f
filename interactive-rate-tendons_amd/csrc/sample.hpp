// sample.hpp -- createRoadmap phase 1 on the device (motion-planning/VoxelCachedLazyPRM.cpp:1415-1455: every new vertex is a
// rejection-sampling loop  sampleUniform -> fk -> is_valid_shape -> voxelize -> collides  repeated until a state is accepted):
// a counter-based candidate generator (the candidate sequence is a pure function of (seed, global candidate index), so it is the
// same for every batch size, every GPU count and on the host), and the order-preserving compaction of the accepted candidates.
// Separate translation unit; tendon_hip.hip owns the entry points (tr_candidate_states*, tr_validate_candidates_dev,
// tr_compact_rows_dev, tr_sample_valid_vertices*).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "tr_types.hpp"

namespace trk {

// Sampling box of the state space (motion-planning/Problem.cpp:101-163: tensions in [0, max_tension], rotation in [-pi, pi),
// retraction in [0, L]): state[d] = lo[d] + u * (hi[d] - lo[d]), u in [0, 1) with 53 random bits; product and sum rounded
// separately (no FMA), so that a host generator reproduces every bit.
struct SampleBox {
  double lo[TRK_MAX_TENDONS + 2];
  double span[TRK_MAX_TENDONS + 2];      // hi - lo, formed once on the host
  int32_t S;
  int32_t pad_;
};

// Device counters of one sampling run (tr_sample_valid_vertices): read back once per batch.
struct SampleCounters {
  unsigned long long have;               // accepted so far (<= n_want)
  unsigned long long tried;              // candidates consumed: index after the one that completed the set, or all so far
  unsigned long long batch_total;        // accepted candidates in the last batch (before clipping to n_want)
  unsigned long long pad_;
};

// states [count][S] of candidates first .. first + count - 1 of the sequence `seed`
void launch_candidate_states(uint64_t seed, uint64_t first, int64_t count, const SampleBox &box, double *d_states, hipStream_t s);

// Order-preserving compaction of the rows whose mask bit is set (bit i & 63 of word i >> 6), i in [0, count):
// row i (row_doubles doubles of d_rows, and 3 doubles of d_tips when given) goes to output position *d_have + (number of set bits
// below i), positions >= capacity are dropped.  d_index_out (optional) receives index_base + i.  Afterwards
// d_counters->have = min(capacity, have + total), ->batch_total = total, and ->tried = index_base + (i of the row that filled
// position capacity - 1) + 1 if the batch reached capacity, else index_base + count.
// d_wprefix: scratch of ceil(count / 64) uint32.
void launch_compact_rows(const uint64_t *d_mask, int64_t count, uint64_t index_base, const double *d_rows, int row_doubles,
                         const double *d_tips, int64_t capacity, double *d_rows_out, double *d_tips_out, int64_t *d_index_out,
                         SampleCounters *d_counters, uint32_t *d_wprefix, hipStream_t s);

// Signature rows on the wire (the sharded build's one large all-gather): consecutive backbone points lie at most dL <= one voxel edge
// apart, so consecutive cells of a row differ by -1, 0 or +1 per axis.  A packed row = the first point's signature word, then 6
// bits per further point (2 per axis: difference + 1), 16 points to three words; padded to an even number of words.  A row with a
// SIG_BAD word or a larger step cannot be coded: it is counted in *d_bad (the caller then sends the rows as they are).
__host__ __device__ inline int sig_packed_words(int n_points) { const int w = 1 + 3 * ((n_points - 1 + 15) / 16); return (w + 1) & ~1; }
void launch_pack_signatures(const uint32_t *d_sig, int64_t n_rows, int n_points, int64_t sig_stride, uint32_t *d_packed, unsigned long long *d_bad,
                            hipStream_t s);
void launch_unpack_signatures(const uint32_t *d_packed, int64_t n_rows, int n_points, int64_t sig_stride, uint32_t *d_sig, hipStream_t s);

}  // namespace trk
