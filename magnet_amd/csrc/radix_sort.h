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
// the sort does not look at and that are zero on input).
struct Tag {
    const int32_t* starts;
    int field_shift;
    uint32_t field_mask;
    int tag_shift;
    uint32_t cap;
};

// Sorts n keys by the `bits` key bits starting at bit `first_bit` (lower bits are carried along and, the
// sort being stable, keep their input order).  The input is in (a, va); passes alternate between the
// two buffers, so the result is in (a, va) when passes_for(bits) is even and in (b, vb) when it is odd.
// va / vb may both be null (keys only).  Returns a hipError_t.
// `tag` (keys-only sorts): see Tag.
hipError_t sort(uint64_t* a, uint64_t* b, float* va, float* vb, int64_t n, int first_bit, int bits, void* workspace,
                size_t workspace_bytes, hipStream_t stream, const Tag* tag = nullptr);

}  // namespace mhsort
