// Exclusive scan of 0/1 byte flags on gfx950: 4096-flag tiles, wavefront shuffle scans inside a tile,
// tile sums scanned by one workgroup.  Shared by the Laplacian builder (CSR slot of every kept entry)
// and the halo plan (slot of every referenced remote column).
//
//   flags  uint8  [n_tiles * kScanTile]   (padding past the last flag must be zero)
//   tiles  uint32 [n_tiles + 1]
//   pos    uint32 [m + 1]                 pos[i] = number of set flags before i; pos[m] = total
//
//   scan_tile_sums<<<n_tiles, 256>>>(flags, tiles);
//   scan_tiles<<<1, 1024>>>(tiles, n_tiles, total);      // *total (device int64) = number of set flags
//   scan_tile_write<<<n_tiles, 256>>>(flags, tiles, pos, m);
#pragma once
#include <cstdint>

#include <hip/hip_runtime.h>

namespace mhscan {
namespace {

constexpr int kBlock = 256;
constexpr int kScanItems = 16;
constexpr int kScanTile = kBlock * kScanItems;   // 4096 flags per workgroup

__device__ inline uint32_t sum_bytes16(const uint4 v) {
    // flags are 0/1 bytes: add the four bytes of each word with a multiply
    auto s4 = [](uint32_t x) { return (x * 0x01010101u) >> 24; };
    return s4(v.x) + s4(v.y) + s4(v.z) + s4(v.w);
}

__device__ inline uint32_t wave_inclusive_scan(uint32_t v, int lane) {
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t t = __shfl_up(v, off, 64);
        if (lane >= off) v += t;
    }
    return v;
}

__global__ void __launch_bounds__(kBlock) scan_tile_sums(const uint8_t* __restrict__ flags, uint32_t* __restrict__ tile_sums) {
    __shared__ uint32_t wsum[kBlock / 64];
    const uint4* f = reinterpret_cast<const uint4*>(flags + size_t(blockIdx.x) * kScanTile);
    uint32_t v = sum_bytes16(f[threadIdx.x]);
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// one workgroup: tile_sums[0..n) -> exclusive prefix in place, total to tile_sums[n] and *total
__global__ void __launch_bounds__(1024) scan_tiles(uint32_t* __restrict__ tile_sums, int64_t n, int64_t* total) {
    __shared__ uint32_t wtot[16];
    __shared__ uint32_t carry_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int64_t base = 0; base < n; base += 1024) {
        int64_t i = base + threadIdx.x;
        uint32_t v = (i < n) ? tile_sums[i] : 0u;
        uint32_t inc = wave_inclusive_scan(v, lane);
        if (lane == 63) wtot[wave] = inc;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wave; ++w) woff += wtot[w];
        uint32_t carry = carry_s;
        if (i < n) tile_sums[i] = carry + woff + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) { tile_sums[n] = carry_s; *total = int64_t(carry_s); }
}

__global__ void __launch_bounds__(kBlock) scan_tile_write(const uint8_t* __restrict__ flags,
                                                         const uint32_t* __restrict__ tile_sums,
                                                         uint32_t* __restrict__ pos, int64_t m) {
    __shared__ uint32_t wtot[kBlock / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t base = size_t(blockIdx.x) * kScanTile + size_t(threadIdx.x) * kScanItems;
    uint4 fv = *reinterpret_cast<const uint4*>(flags + base);
    uint32_t tsum = sum_bytes16(fv);
    uint32_t inc = wave_inclusive_scan(tsum, lane);
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    uint32_t off = tile_sums[blockIdx.x] + inc - tsum;
    for (int w = 0; w < wave; ++w) off += wtot[w];
    uint32_t words[4] = {fv.x, fv.y, fv.z, fv.w};
    uint32_t out[kScanItems];
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        out[i] = off;
        off += (words[i >> 2] >> ((i & 3) * 8)) & 0xFFu;
    }
    if (base + kScanItems <= size_t(m) + 1) {
        uint4* o = reinterpret_cast<uint4*>(pos + base);
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = make_uint4(out[4 * i], out[4 * i + 1], out[4 * i + 2], out[4 * i + 3]);
    } else {
#pragma unroll
        for (int i = 0; i < kScanItems; ++i)
            if (base + i <= size_t(m)) pos[base + i] = out[i];
    }
}

}  // namespace
}  // namespace mhscan
