// sample.hip -- candidate generator and accepted-set compaction of createRoadmap's vertex phase (see sample.hpp).
#include "sample.hpp"
#include <algorithm>

namespace trk {

namespace {

// Philox-4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11): a keyed bijection of
// 128-bit counters; counter = (candidate index lo, hi, coordinate pair, 0), key = seed.  Restated from the paper's round
// function and constants; tests/test_host.py holds the paper's known-answer vectors for the host mirror and
// tests/test_gpu_sampling.py compares this kernel with that mirror bit for bit.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ double unit53(uint32_t hi, uint32_t lo) {          // 53 bits -> [0, 1)
  return (double)((((uint64_t)hi << 32) | lo) >> 11) * 0x1.0p-53;
}

// one thread per (candidate, coordinate pair): two coordinates from one Philox block, 16 contiguous bytes per thread
__global__ __launch_bounds__(256) void candidate_states_kernel(uint64_t seed, uint64_t first, int64_t count, SampleBox box,
                                                               double *__restrict__ states) {
  const int S = box.S, npair = (S + 1) >> 1;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= count * npair) return;
  const int64_t i = t / npair;
  const int j = (int)(t - i * npair);
  const uint64_t idx = first + (uint64_t)i;
  uint32_t r[4];
  philox4x32_10((uint32_t)idx, (uint32_t)(idx >> 32), (uint32_t)j, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
  const int d = 2 * j;
  double *row = states + i * S;
  row[d] = __dadd_rn(box.lo[d], __dmul_rn(unit53(r[0], r[1]), box.span[d]));
  if (d + 1 < S) row[d + 1] = __dadd_rn(box.lo[d + 1], __dmul_rn(unit53(r[2], r[3]), box.span[d + 1]));
}

__device__ __forceinline__ uint64_t word_at(const uint64_t *mask, int64_t w, int64_t count) {
  uint64_t v = mask[w];
  const int64_t rest = count - w * 64;              // bits of the last word beyond `count` are not rows
  if (rest < 64) v &= (rest <= 0) ? 0ull : ((1ull << rest) - 1ull);
  return v;
}

// exclusive prefix of the words' popcounts inside blocks of 256 words; block totals to bsum
__global__ __launch_bounds__(256) void compact_word_scan(const uint64_t *__restrict__ mask, int64_t nw, int64_t count,
                                                         uint32_t *__restrict__ wprefix, uint32_t *__restrict__ bsum) {
  __shared__ uint32_t wsum[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t w = (int64_t)blockIdx.x * 256 + tid;
  const uint32_t c = w < nw ? (uint32_t)__popcll(word_at(mask, w, count)) : 0u;
  uint32_t incl = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t v = __shfl_up(incl, o, 64);
    if (lane >= o) incl += v;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  uint32_t base = 0;
  for (int q = 0; q < wave; q++) base += wsum[q];
  if (w < nw) wprefix[w] = base + incl - c;
  if (tid == 255) bsum[blockIdx.x] = base + incl;
}

// exclusive scan of the block totals in place (one workgroup, tiles of 1024); grand total to the counters
__global__ __launch_bounds__(1024) void compact_block_scan(uint32_t *__restrict__ bsum, int64_t nb, SampleCounters *__restrict__ ctr) {
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t running_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) running_s = 0;
  __syncthreads();
  for (int64_t t0 = 0; t0 < nb; t0 += 1024) {
    const int64_t i = t0 + tid;
    const uint32_t c = i < nb ? bsum[i] : 0u;
    uint32_t incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t v = __shfl_up(incl, o, 64);
      if (lane >= o) incl += v;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t base = running_s;
    for (int q = 0; q < wave; q++) base += wsum[q];
    if (i < nb) bsum[i] = base + incl - c;
    __syncthreads();
    if (tid == 1023) running_s = base + incl;
    __syncthreads();
  }
  if (tid == 0) ctr->batch_total = running_s;
}

__global__ __launch_bounds__(256) void compact_scatter(const uint64_t *__restrict__ mask, int64_t count, uint64_t index_base,
                                                       const double *__restrict__ rows, int rd, const double *__restrict__ tips,
                                                       int64_t capacity, double *__restrict__ rows_out, double *__restrict__ tips_out,
                                                       int64_t *__restrict__ index_out, SampleCounters *__restrict__ ctr,
                                                       const uint32_t *__restrict__ wprefix, const uint32_t *__restrict__ bsum) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  const int64_t w = i >> 6;
  const uint64_t word = word_at(mask, w, count);
  const int b = (int)(i & 63);
  if (!((word >> b) & 1ull)) return;
  const int64_t pos = (int64_t)ctr->have + bsum[w >> 8] + wprefix[w] + __popcll(word & ((1ull << b) - 1ull));
  if (pos >= capacity) return;
  if (rows_out) for (int k = 0; k < rd; k++) rows_out[pos * rd + k] = rows[i * rd + k];
  if (tips_out && tips) { tips_out[3 * pos] = tips[3 * i]; tips_out[3 * pos + 1] = tips[3 * i + 1]; tips_out[3 * pos + 2] = tips[3 * i + 2]; }
  if (index_out) index_out[pos] = (int64_t)(index_base + (uint64_t)i);
  if (pos == capacity - 1) ctr->tried = index_base + (uint64_t)i + 1ull;
}

__global__ void compact_finish(SampleCounters *ctr, int64_t capacity, uint64_t index_base, int64_t count) {
  const unsigned long long sum = ctr->have + ctr->batch_total;
  if (sum < (unsigned long long)capacity) { ctr->have = sum; ctr->tried = index_base + (uint64_t)count; }
  else ctr->have = (unsigned long long)capacity;         // ->tried was written by the row that filled the last position
}

// one thread per (row, group of 16 points): points 16 g + 1 .. 16 g + 16 against their predecessors; the row's words 16 g .. 16 g + 15
// come in as four 16-byte loads (rows start on 64-byte boundaries), word 16 g + 16 as one more
__global__ __launch_bounds__(256) void pack_signatures_kernel(const uint32_t *__restrict__ sig, int64_t n_rows, int P, int64_t stride, int G, int PW,
                                                              uint32_t *__restrict__ packed, unsigned long long *__restrict__ bad) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n_rows * G) return;
  const int64_t row = t / G;
  const int g = (int)(t - row * G);
  const uint32_t *in = sig + row * stride;
  uint32_t w[17];
  {
    const uint4 *in4 = reinterpret_cast<const uint4 *>(in + 16 * g);
#pragma unroll
    for (int q = 0; q < 4; q++) { const uint4 v = in4[q]; w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w; }
    w[16] = 16 * g + 16 < P ? in[16 * g + 16] : 0u;
  }
  bool ok = !(w[0] & (1u << 30));
  // three words: five 6-bit codes each (bits 0 .. 29), the sixteenth code's three 2-bit fields in their top bits
  uint32_t w3[3] = {0u, 0u, 0u};
#pragma unroll
  for (int j = 0; j < 16; j++) {
    const int pt = 16 * g + 1 + j;
    uint32_t code = 0x15u;                                     // (padding past the last point: no movement)
    if (pt < P) {
      const uint32_t cur = w[j + 1], prev = w[j];
      const uint32_t dx = (cur & 1023u) + 1u - (prev & 1023u), dy = ((cur >> 10) & 1023u) + 1u - ((prev >> 10) & 1023u),
                     dz = ((cur >> 20) & 1023u) + 1u - ((prev >> 20) & 1023u);
      ok = ok && !(cur & (1u << 30)) && dx <= 2u && dy <= 2u && dz <= 2u;
      code = (dx & 3u) | ((dy & 3u) << 2) | ((dz & 3u) << 4);
    }
    if (j < 15) w3[j / 5] |= code << (6 * (j % 5));
    else { w3[0] |= (code & 3u) << 30; w3[1] |= ((code >> 2) & 3u) << 30; w3[2] |= ((code >> 4) & 3u) << 30; }
  }
  uint32_t *out = packed + row * PW;
  if (g == 0) out[0] = w[0];
  out[1 + 3 * g] = w3[0]; out[2 + 3 * g] = w3[1]; out[3 + 3 * g] = w3[2];
  if (g == G - 1 && ((1 + 3 * G) & 1)) out[1 + 3 * G] = 0u;    // the pad word
  if (!ok) atomicAdd(bad, 1ull);
}

__device__ __forceinline__ uint32_t sig_code_at(const uint32_t *w3, int j) {
  if (j < 15) return (w3[j / 5] >> (6 * (j % 5))) & 63u;
  return (w3[0] >> 30) | ((w3[1] >> 30) << 2) | ((w3[2] >> 30) << 4);
}

// one thread per (row, 16 words of the row): words 16 g .. 16 g + 15 = points 16 g .. 16 g + 15 leave as four 16-byte stores.  Point 16 g
// = the first cell + every step of the code groups before g; the points after it take the first fifteen steps of code group g.
__global__ __launch_bounds__(256) void unpack_signatures_kernel(const uint32_t *__restrict__ packed, int64_t n_rows, int P, int64_t stride, int G, int PW,
                                                                uint32_t *__restrict__ sig) {
  const int TPR = (P + 15) / 16;                               // threads per row
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n_rows * TPR) return;
  const int64_t row = t / TPR;
  const int g = (int)(t - row * TPR);
  const uint32_t *in = packed + row * PW;
  const uint32_t first = in[0];
  int cx = (int)(first & 1023u), cy = (int)((first >> 10) & 1023u), cz = (int)((first >> 20) & 1023u);
  for (int q = 0; q < g; q++) {
    const uint32_t w3[3] = {in[1 + 3 * q], in[2 + 3 * q], in[3 + 3 * q]};
    // the sum of sixteen 2-bit fields per axis, minus sixteen: fields of one axis sit 6 bits apart
    int sx = 0, sy = 0, sz = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) { const uint32_t c = sig_code_at(w3, j); sx += (int)(c & 3u); sy += (int)((c >> 2) & 3u); sz += (int)((c >> 4) & 3u); }
    cx += sx - 16; cy += sy - 16; cz += sz - 16;
  }
  uint32_t w[16];
  uint32_t w3[3] = {0x15555555u, 0x15555555u, 0x15555555u};
  if (g < G) { w3[0] = in[1 + 3 * g]; w3[1] = in[2 + 3 * g]; w3[2] = in[3 + 3 * g]; }
#pragma unroll
  for (int j = 0; j < 16; j++) {
    const int pt = 16 * g + j;
    w[j] = pt < P ? (((uint32_t)cx & 1023u) | (((uint32_t)cy & 1023u) << 10) | (((uint32_t)cz & 1023u) << 20)) : 0u;
    if (j < 15) { const uint32_t c = sig_code_at(w3, j); cx += (int)(c & 3u) - 1; cy += (int)((c >> 2) & 3u) - 1; cz += (int)((c >> 4) & 3u) - 1; }
  }
  uint4 *out4 = reinterpret_cast<uint4 *>(sig + row * stride + 16 * g);
#pragma unroll
  for (int q = 0; q < 4; q++) out4[q] = uint4{w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]};
}

}  // namespace

void launch_pack_signatures(const uint32_t *d_sig, int64_t n_rows, int n_points, int64_t sig_stride, uint32_t *d_packed, unsigned long long *d_bad,
                            hipStream_t s) {
  if (n_rows <= 0) return;
  const int G = (n_points - 1 + 15) / 16;
  const int64_t threads = n_rows * std::max(G, 1);
  hipLaunchKernelGGL(pack_signatures_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, d_sig, n_rows, n_points, sig_stride, std::max(G, 1),
                     sig_packed_words(n_points), d_packed, d_bad);
}
void launch_unpack_signatures(const uint32_t *d_packed, int64_t n_rows, int n_points, int64_t sig_stride, uint32_t *d_sig, hipStream_t s) {
  if (n_rows <= 0) return;
  const int G = (n_points - 1 + 15) / 16;
  const int64_t threads = n_rows * ((n_points + 15) / 16);
  hipLaunchKernelGGL(unpack_signatures_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, d_packed, n_rows, n_points, sig_stride, std::max(G, 1),
                     sig_packed_words(n_points), d_sig);
}

void launch_candidate_states(uint64_t seed, uint64_t first, int64_t count, const SampleBox &box, double *d_states, hipStream_t s) {
  if (count <= 0) return;
  const int64_t threads = count * ((box.S + 1) / 2);
  hipLaunchKernelGGL(candidate_states_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, seed, first, count, box, d_states);
}

void launch_compact_rows(const uint64_t *d_mask, int64_t count, uint64_t index_base, const double *d_rows, int row_doubles,
                         const double *d_tips, int64_t capacity, double *d_rows_out, double *d_tips_out, int64_t *d_index_out,
                         SampleCounters *d_counters, uint32_t *d_wprefix, hipStream_t s) {
  if (count <= 0) return;
  const int64_t nw = (count + 63) / 64, nb = (nw + 255) / 256;
  uint32_t *bsum = d_wprefix + nw;
  hipLaunchKernelGGL(compact_word_scan, dim3((unsigned)nb), dim3(256), 0, s, d_mask, nw, count, d_wprefix, bsum);
  hipLaunchKernelGGL(compact_block_scan, dim3(1), dim3(1024), 0, s, bsum, nb, d_counters);
  hipLaunchKernelGGL(compact_scatter, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, d_mask, count, index_base, d_rows,
                     row_doubles, d_tips, capacity, d_rows_out, d_tips_out, d_index_out, d_counters, d_wprefix, bsum);
  hipLaunchKernelGGL(compact_finish, dim3(1), dim3(1), 0, s, d_counters, capacity, index_base, count);
}

}  // namespace trk
