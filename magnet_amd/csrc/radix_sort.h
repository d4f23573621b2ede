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

// Sorts n keys by the `bits` key bits starting at bit `first_bit` (lower bits are carried along and, the
// sort being stable, keep their input order).  The input is in (a, va); passes alternate between the
// two buffers, so the result is in (a, va) when passes_for(bits) is even and in (b, vb) when it is odd.
// va / vb may both be null (keys only).  Returns a hipError_t.
hipError_t sort(uint64_t* a, uint64_t* b, float* va, float* vb, int64_t n, int first_bit, int bits, void* workspace,
                size_t workspace_bytes, hipStream_t stream);

}  // namespace mhsort
