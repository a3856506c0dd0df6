// RNA records (sequence + dot-bracket text) -> model-ready graph arrays, on the device.
//
// Replaces GraphBuilder._build_full (src/ginfinity/graph.py:494-561: node features 496-514,
// typed directed edges 516-546, pair table 737-747) and the concatenation into one shard
// (graph.py:346-412) for unsliced records.  Integer / one-hot work: the arrays are
// bit-identical to the reference's, including the order of the edges
//   per record: [backbone fwd | backbone rev | pair fwd (opens ascending) | pair rev |
//                skip-2 (i->i+2, i+2->i interleaved)].
// The positional columns (numpy float32 sin / cos of the relative position, graph.py:508-513)
// are an INPUT: they are the reference's host numpy values, not recomputed here.
//
// One launch, two kinds of workgroup:
//   * edge blocks: one wave per record, 64 nucleotides per step.  Bracket matching: a ')'
//     at nesting level l pairs with the most recent '(' of level l.  Inside a step that is
//     a scan over the step's open brackets with v_readlane; across steps the most recent
//     open bracket of every level is carried in a per-wave LDS array.  The record's dot-bracket text
//     is fetched 1,024 characters at a time (16 independent loads into four registers),
//     so the step-to-step chain holds no global load and never waits for its own edge
//     stores: a 4,096-nt record is 64 short steps.
//   * feature blocks: one thread per output float (a feature row depends on its own
//     nucleotide only), fully coalesced; they also check the alphabet.
// No workgroup barrier anywhere: waves are independent.
#include "gfy_common.h"

namespace gfy {
namespace {

// Nesting levels carried per wave, 16 KB of (position, rank) pairs: every record the
// reference accepts fits (MAXIMUM_LENGTH_NT = 4096, _validation.py, so depth <= 2048); a
// deeper nest is reported through first_invalid like any other unusable record.
constexpr int kLdsLevels = 2049;
constexpr int kWaves = 4;
constexpr int kStash = 1024;       // characters of one record held in LDS at a time
constexpr int kFeatureItems = 8;   // output floats per thread of a feature block

// record that owns node `node` (node_ptr rebased to node_ptr[0]); error path only
__device__ int owner_of(const int64_t* node_ptr, int records, int64_t node) {
  int lo = 0, hi = records - 1;
  const int64_t first = node_ptr[0];
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (node_ptr[mid] - first <= node) lo = mid; else hi = mid - 1;
  }
  return lo;
}

__device__ __forceinline__ void feature_block(
    int block, const uint8_t* __restrict__ bases, const uint8_t* __restrict__ marks,
    const int64_t* __restrict__ node_ptr, int records, int64_t nodes, int struct_states,
    int positional_cols, const float* __restrict__ positional,
    float* __restrict__ features, uint32_t* first_invalid) {
  const int dim = 4 + struct_states + positional_cols;
  const int64_t total = nodes * dim;
  const int64_t begin = (int64_t)block * (256 * kFeatureItems) + threadIdx.x;
#pragma unroll
  for (int k = 0; k < kFeatureItems; ++k) {
    const int64_t idx = begin + k * 256;
    if (idx >= total) break;
    const int64_t row = idx / dim;
    const int col = (int)(idx - row * dim);
    const uint8_t letter = bases[row], mark = marks[row];
    const int code = letter == 'A' ? 0 : letter == 'C' ? 1 : letter == 'G' ? 2
                   : letter == 'U' ? 3 : -1;
    const int state = mark == '(' ? 0 : mark == ')' ? 2 : mark == '.' ? 1 : -1;
    float v;
    if (col < 4) v = col == code ? 1.f : 0.f;
    else if (col < 4 + struct_states)
      v = struct_states == 1 ? (state != 1 ? 1.f : 0.f) : (col - 4 == state ? 1.f : 0.f);
    else
      v = positional[row * positional_cols + (col - 4 - struct_states)];
    features[idx] = v;
    if (col == 0 && (code < 0 || state < 0))
      atomicMin(first_invalid, (uint32_t)owner_of(node_ptr, records, row));
  }
}

__global__ __launch_bounds__(64 * kWaves) void k_build_graphs(
    const uint8_t* __restrict__ bases, const uint8_t* __restrict__ marks,
    const int64_t* __restrict__ node_ptr, const int64_t* __restrict__ edge_ptr,
    int records, int64_t nodes, int edge_blocks, int struct_states, int positional_cols,
    int skip2, const float* __restrict__ positional, float* __restrict__ features,
    int32_t* __restrict__ src, int32_t* __restrict__ dst, uint8_t* __restrict__ types,
    uint32_t* __restrict__ first_invalid) {
  __shared__ int2 lds_stack[kWaves][kLdsLevels];
  if ((int)blockIdx.x >= edge_blocks) {
    feature_block(blockIdx.x - edge_blocks, bases, marks, node_ptr, records, nodes,
                  struct_states, positional_cols, positional, features, first_invalid);
    return;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = blockIdx.x * kWaves + wave;
  if (r >= records) return;
  // Plain (not volatile) LDS accesses: volatile makes hipcc drain vmcnt — every edge store of
  // the previous step — before each of them, which was 1.6 us per step.  The writes and
  // reads of one wave stay in program order; wave_barrier() keeps the compiler from moving
  // them across the points where another lane's value is expected.
  int2* stack = lds_stack[wave];

  const int64_t n0 = node_ptr[0], e0 = edge_ptr[0];
  const int base = (int)(node_ptr[r] - n0);
  const int length = (int)(node_ptr[r + 1] - node_ptr[r]);
  const int eb = (int)(edge_ptr[r] - e0);
  const int edges = (int)(edge_ptr[r + 1] - edge_ptr[r]);
  const int head = length > 1 ? length - 1 : 0;
  const int skip = (skip2 && length > 2) ? length - 2 : 0;
  const int pair_edges = edges - 2 * head - 2 * skip;
  bool bad = length < 0 || pair_edges < 0 || (pair_edges & 1);
  const int pairs = bad ? 0 : pair_edges >> 1;
  const int e_pair = eb + 2 * head, e_skip = e_pair + 2 * pairs;

  int depth = 0, opens = 0;
  for (int c0 = 0; c0 < length; c0 += kStash) {
    // the next 1,024 characters of the record: 16 independent byte loads per lane, kept
    // packed in four registers (lane L owns characters c0 + 64 k + L)
    uint32_t ch[kStash / 64], packed[kStash / 256] = {};
#pragma unroll
    for (int k = 0; k < kStash / 64; ++k) {
      const int p = c0 + 64 * k + lane;                    // clamped, not predicated: the
      ch[k] = marks[base + (p < length ? p : length - 1)];   // loads stay independent
    }
#pragma unroll
    for (int k = 0; k < kStash / 64; ++k) {
      const uint32_t c = c0 + 64 * k + lane < length ? ch[k] : (uint32_t)'.';
      packed[k >> 2] |= c << (8 * (k & 3));
    }

    const int c1 = length - c0 < kStash ? length : c0 + kStash;
    for (int c = c0; c < c1; c += 64) {
      const int p = c + lane;
      const int k = (c - c0) >> 6;                       // wave-uniform
      const uint32_t word = (k >> 2) == 0 ? packed[0] : (k >> 2) == 1 ? packed[1]
                          : (k >> 2) == 2 ? packed[2] : packed[3];
      const uint32_t mark = (word >> (8 * (k & 3))) & 0xFFu;
      const bool open = mark == '(', close = mark == ')';

      const unsigned long long opened = __ballot(open), closed = __ballot(close);
      const unsigned long long below = (1ull << lane) - 1ull, upto = below | (1ull << lane);
      const int level = depth + __popcll(opened & upto) - __popcll(closed & upto)
                      + (close ? 1 : 0);          // a ')' closes the level it came from
      const int rank = opens + __popcll(opened & below);   // index among the record's '('

      // lanes of this step whose '(' has my level, as a 64-bit mask per lane: one compare and
      // one select per open bracket of the step.  For a ')' the mate is the highest such
      // lane below me; a '(' is superseded (not carried) if there is one above me.
      unsigned long long same = 0;
      for (unsigned long long m = opened; m; m &= m - 1ull) {
        const int i = __builtin_amdgcn_readfirstlane(__builtin_ctzll(m));
        const int level_i = __builtin_amdgcn_readlane(level, i);
        same |= level_i == level ? (1ull << i) : 0ull;
      }
      const unsigned long long lower = same & below;
      const int mate = (close && lower) ? 63 - __builtin_clzll(lower) : -1;
      const bool superseded = open && (same & ~upto) != 0;
      const int mate_rank = __shfl(rank, mate < 0 ? 0 : mate, 64);
      __builtin_amdgcn_wave_barrier();
      int partner = -1, partner_rank = 0;
      if (close) {
        if (level < 1) {
          bad = true;
        } else if (mate >= 0) {
          partner = c + mate;
          partner_rank = mate_rank;
        } else if (level < kLdsLevels) {
          const int2 entry = stack[level];
          partner = entry.x;
          partner_rank = entry.y;
        } else {
          bad = true;                             // deeper than any legal record
        }
      }
      // carried entries are read above, replaced below (same wave: program order)
      __builtin_amdgcn_wave_barrier();
      if (open && !superseded && level < kLdsLevels) stack[level] = make_int2(p, rank);
      __builtin_amdgcn_wave_barrier();

      if (p < head) {
        src[eb + p] = base + p;            dst[eb + p] = base + p + 1;
        types[eb + p] = 0;
        src[eb + head + p] = base + p + 1; dst[eb + head + p] = base + p;
        types[eb + head + p] = 1;
      }
      if (close && level >= 1) {
        if (partner < 0 || partner >= p || partner_rank < 0 || partner_rank >= pairs) {
          bad = true;
        } else {
          const int forward = e_pair + partner_rank, reverse = forward + pairs;
          src[forward] = base + partner;   dst[forward] = base + p;       types[forward] = 2;
          src[reverse] = base + p;         dst[reverse] = base + partner; types[reverse] = 3;
        }
      }
      if (p < skip) {
        const int at = e_skip + 2 * p;
        src[at] = base + p;                dst[at] = base + p + 2;        types[at] = 4;
        src[at + 1] = base + p + 2;        dst[at + 1] = base + p;        types[at + 1] = 5;
      }

      depth += __popcll(opened) - __popcll(closed);
      opens += __popcll(opened);
      if (depth < 0) bad = true;
    }
  }
  if (depth != 0 || opens != pairs) bad = true;
  if (__any(bad) && lane == 0) atomicMin(first_invalid, (uint32_t)r);
}

}  // namespace

int launch_build_graphs(const uint8_t* bases, const uint8_t* marks, const int64_t* node_ptr,
                        const int64_t* edge_ptr, int64_t records, int64_t n, int64_t e,
                        int struct_states, int positional_cols, int skip2,
                        const float* positional, float* features, int32_t* edge_index,
                        uint8_t* edge_types, int32_t* first_invalid, hipStream_t s) {
  GFY_CHECK_HIP(hipMemsetAsync(first_invalid, 0xFF, sizeof(int32_t), s));   // -1 = all valid
  if (records == 0) return GFY_OK;
  const int dim = 4 + struct_states + positional_cols;
  const unsigned edge_blocks = (unsigned)((records + kWaves - 1) / kWaves);
  const unsigned feature_blocks =
      (unsigned)((n * dim + 256 * kFeatureItems - 1) / (256 * kFeatureItems));
  hipLaunchKernelGGL(k_build_graphs, dim3(edge_blocks + feature_blocks), dim3(64 * kWaves), 0,
                     s, bases, marks, node_ptr, edge_ptr, (int)records, n, (int)edge_blocks,
                     struct_states, positional_cols, skip2, positional, features, edge_index,
                     edge_index + e, edge_types, (uint32_t*)first_invalid);
  GFY_CHECK_HIP(hipGetLastError());
  return GFY_OK;
}

}  // namespace gfy
