// Stable LSD radix sort of 64-bit keys (optionally with a float payload) for gfx950, used by the
// Laplacian builder.  Internal to libmagnet_hip.so.
#pragma once
#include <cstddef>
#include <cstdint>
#include <hip/hip_runtime.h>

namespace mhsort {

// bytes of scratch (tile histograms and their scan) for n keys
size_t workspace_bytes(int64_t n);

// number of 8-bit passes for `bits` significant key bits
inline int passes_for(int bits) { return (bits + 7) / 8; }

// Optional rewrite of the keys on their way through the first pass of a sort() call: the keys arrive ordered by the
// field (key >> field_shift) & field_mask, starts[f] is the index of the first key of field value f (starts[f + 1] -
// starts[f] keys carry it), and every key gets min(that count, cap) written into its bits from tag_shift upwards (bits
// the sort does not look at and that are zero on input).  With `counts` set the count of field value f is counts[f]
// instead (a table from elsewhere: the keys at hand need not be all keys that carry f) and `starts` is not read.
struct Tag {
    const int32_t* starts;
    const int32_t* counts;
    int field_shift;
    uint32_t field_mask;
    int tag_shift;
    uint32_t cap;
};

// Source of a sort's keys when they are not in memory yet: the directed records of the Laplacian builder, made from the
// edge lists inside the FIRST pass (its histogram and its scatter), so that the records are never written unsorted.
// Record p < cnt[0] is the forward record of list 0's entry p, record cnt[0] + e the backward record of list 1's entry e
// (cnt[d] = n_in[d] + n_self[d]; entries n_in[d] <= e < cnt[d] are the self-loops (e - n_in[d], e - n_in[d], 1) of
// gcn_appr).  key = row << (cb + 1) | col << 1 | dir; an endpoint outside [0, n_nodes) sets bit 0 of *err and the record
// is made harmless, a row outside [row_lo, row_hi) sets bit 1.
struct RecordSource {
    const void* row[2];
    const void* col[2];
    const float* w[2];
    int64_t n_in[2], n_self[2];
    int idx64;
    int cb;
    int32_t n_nodes, row_lo, row_hi;
    int32_t* err;
};

// Sorts n keys by the `bits` key bits starting at bit `first_bit` (lower bits are carried along and, the
// sort being stable, keep their input order).  The input is in (a, va); passes alternate between the
// two buffers, so the result is in (a, va) when passes_for(bits) is even and in (b, vb) when it is odd.
// va / vb may both be null (keys only).  Returns a hipError_t.
// `tag` (keys-only sorts): see Tag.  `gen`: the keys (and, when va is given, the payload) are not in (a, va) but made
// by the first pass from the record source; (a, va) are then only the buffers the second pass writes to.
hipError_t sort(uint64_t* a, uint64_t* b, float* va, float* vb, int64_t n, int first_bit, int bits, void* workspace,
                size_t workspace_bytes, hipStream_t stream, const Tag* tag = nullptr, const RecordSource* gen = nullptr);

}  // namespace mhsort
