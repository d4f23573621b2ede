// Halo plan of one rank's row block of the Laplacian (row-partitioned multi-GPU propagation) on gfx950.
//
// The block's CSR holds GLOBAL column ids; the rank owns columns [c0, c1) (= its own rows) and every other
// column it references is a feature row that must arrive from a peer before the SpMM can gather it.
// The plan is, from the column ids alone:
//   halo_ids  ascending global ids of the referenced columns outside [c0, c1)          ("need" set)
//   peer_ptr  halo_ids[peer_ptr[p] : peer_ptr[p+1]] are the rows owned by peer p (rows bounds[p] .. bounds[p+1])
//   col_local the block's columns renumbered for a compact feature buffer [own rows | halo rows]:
//             own column c -> c - c0, remote column c -> (c1 - c0) + its slot in halo_ids
// so that a peer's rows land contiguously, in the order of halo_ids, right where the SpMM reads them.
//
// Steps: one flag byte per global column (marked by every nonzero outside the own range), the exclusive
// flag scan shared with the Laplacian builder, then three independent fill kernels.
// The single-device reference has no counterpart (reference src/sparse_Magnet.py:23-24).
#include <cstdint>

#include <hip/hip_runtime.h>

#include "common.h"
#include "scan.h"

namespace {

constexpr int kBlock = 256;
using mhscan::kScanTile;

struct HaloHeader {      // first 256 bytes of the workspace
    int32_t err;         // != 0: a column id outside [0, n_cols)
    int32_t pad;
    int64_t n_halo;
};

struct HaloLayout {
    int64_t n_tiles, n_pad;
    size_t off_hdr, off_flags, off_pos, off_tiles, total;
};

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

HaloLayout halo_layout(int32_t n_cols) {
    HaloLayout L{};
    L.n_tiles = int64_t(n_cols) / kScanTile + 1;          // covers positions 0 .. n_cols (pos[n_cols] = total)
    L.n_pad = L.n_tiles * kScanTile;
    size_t o = 0;
    L.off_hdr = o;    o += 256;
    L.off_flags = o;  o += align_up(size_t(L.n_pad));
    L.off_pos = o;    o += align_up(size_t(L.n_pad + 1) * 4);
    L.off_tiles = o;  o += align_up(size_t(L.n_tiles + 1) * 4);
    L.total = o;
    return L;
}

inline unsigned grid_for(int64_t n, int64_t cap = int64_t(1) << 20) {
    int64_t g = (n + kBlock - 1) / kBlock;
    if (g < 1) g = 1;
    return unsigned(g < cap ? g : cap);
}

__global__ void __launch_bounds__(kBlock) mark_remote(const int32_t* __restrict__ col_idx, int64_t nnz, int32_t n_cols,
                                                      int32_t c0, int32_t c1, uint8_t* flags, HaloHeader* hdr) {
    for (int64_t j = int64_t(blockIdx.x) * kBlock + threadIdx.x; j < nnz; j += int64_t(gridDim.x) * kBlock) {
        const int32_t c = col_idx[j];
        if (c < 0 || c >= n_cols) { atomicOr(&hdr->err, 1); continue; }
        // many lanes may mark the same column; they all store the same byte
        if (c < c0 || c >= c1) __hip_atomic_store(&flags[c], uint8_t(1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ void __launch_bounds__(kBlock) write_halo_ids(const uint8_t* __restrict__ flags, const uint32_t* __restrict__ pos,
                                                         int32_t n_cols, int64_t n_halo, int32_t* __restrict__ halo_ids) {
    for (int64_t c = int64_t(blockIdx.x) * kBlock + threadIdx.x; c < n_cols; c += int64_t(gridDim.x) * kBlock)
        if (flags[c] && int64_t(pos[c]) < n_halo) halo_ids[pos[c]] = int32_t(c);
}

__global__ void __launch_bounds__(kBlock) write_peer_ptr(const uint32_t* __restrict__ pos, const int32_t* __restrict__ bounds,
                                                         int32_t world, int32_t n_cols, int32_t* __restrict__ peer_ptr) {
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p > world) return;
    int32_t b = bounds[p];
    b = b < 0 ? 0 : (b > n_cols ? n_cols : b);
    peer_ptr[p] = int32_t(pos[b]);
}

__global__ void __launch_bounds__(kBlock) renumber_columns(const int32_t* __restrict__ col_idx, int64_t nnz, int32_t c0, int32_t c1,
                                                           const uint32_t* __restrict__ pos, int32_t* __restrict__ col_local) {
    for (int64_t j = int64_t(blockIdx.x) * kBlock + threadIdx.x; j < nnz; j += int64_t(gridDim.x) * kBlock) {
        const int32_t c = col_idx[j];
        col_local[j] = (c >= c0 && c < c1) ? c - c0 : (c1 - c0) + int32_t(pos[c]);
    }
}

}  // namespace

extern "C" size_t mh_halo_plan_workspace_bytes(int32_t n_cols) {
    if (n_cols < 0) return 0;
    return halo_layout(n_cols).total;
}

extern "C" int mh_halo_plan_count(const int32_t* col_idx, int64_t nnz, int32_t n_cols, int32_t c0, int32_t c1, void* workspace,
                                  size_t workspace_bytes, int64_t* n_halo_host, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    MH_REQUIRE(nnz >= 0 && n_cols >= 0, "negative size");
    MH_REQUIRE(0 <= c0 && c0 <= c1 && c1 <= n_cols, "own column range [%d, %d) outside [0, %d]", c0, c1, n_cols);
    MH_REQUIRE(workspace && n_halo_host, "workspace/n_halo_host is null");
    MH_REQUIRE(nnz == 0 || col_idx, "col_idx is null");
    MH_REQUIRE(mh_aligned16(workspace), "workspace must be 16-byte aligned");
    HaloLayout L = halo_layout(n_cols);
    if (workspace_bytes < L.total)
        return mh::fail(MH_ERR_WORKSPACE, "workspace %zu < required %zu bytes", workspace_bytes, L.total);
    char* ws = static_cast<char*>(workspace);
    HaloHeader* hdr = reinterpret_cast<HaloHeader*>(ws + L.off_hdr);
    uint8_t* flags = reinterpret_cast<uint8_t*>(ws + L.off_flags);
    uint32_t* pos = reinterpret_cast<uint32_t*>(ws + L.off_pos);
    uint32_t* tiles = reinterpret_cast<uint32_t*>(ws + L.off_tiles);
    MH_HIP(hipMemsetAsync(hdr, 0, 256, stream));
    MH_HIP(hipMemsetAsync(flags, 0, size_t(L.n_pad), stream));
    if (nnz > 0) {
        mark_remote<<<grid_for(nnz, 1 << 16), kBlock, 0, stream>>>(col_idx, nnz, n_cols, c0, c1, flags, hdr);
        MH_LAUNCH_CHECK("mark_remote");
    }
    mhscan::scan_tile_sums<<<unsigned(L.n_tiles), kBlock, 0, stream>>>(flags, tiles);
    MH_LAUNCH_CHECK("scan_tile_sums");
    mhscan::scan_tiles<<<1, 1024, 0, stream>>>(tiles, L.n_tiles, &hdr->n_halo);
    MH_LAUNCH_CHECK("scan_tiles");
    mhscan::scan_tile_write<<<unsigned(L.n_tiles), kBlock, 0, stream>>>(flags, tiles, pos, int64_t(n_cols));
    MH_LAUNCH_CHECK("scan_tile_write");
    HaloHeader h{};
    MH_HIP(hipMemcpyAsync(&h, hdr, sizeof(HaloHeader), hipMemcpyDeviceToHost, stream));
    MH_HIP(hipStreamSynchronize(stream));
    if (h.err) return mh::fail(MH_ERR_INDEX, "a column id is outside [0, %d)", n_cols);
    *n_halo_host = h.n_halo;
    return MH_OK;
}

extern "C" int mh_halo_plan_fill(const int32_t* col_idx, int64_t nnz, int32_t n_cols, int32_t c0, int32_t c1,
                                 const int32_t* bounds, int32_t world, void* workspace, size_t workspace_bytes, int64_t n_halo,
                                 int32_t* halo_ids, int32_t* peer_ptr, int32_t* col_local, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    MH_REQUIRE(nnz >= 0 && n_cols >= 0 && n_halo >= 0 && world >= 1, "bad sizes");
    MH_REQUIRE(0 <= c0 && c0 <= c1 && c1 <= n_cols, "own column range [%d, %d) outside [0, %d]", c0, c1, n_cols);
    MH_REQUIRE(workspace, "workspace is null");
    HaloLayout L = halo_layout(n_cols);
    if (workspace_bytes < L.total)
        return mh::fail(MH_ERR_WORKSPACE, "workspace %zu < required %zu bytes", workspace_bytes, L.total);
    char* ws = static_cast<char*>(workspace);
    const uint8_t* flags = reinterpret_cast<const uint8_t*>(ws + L.off_flags);
    const uint32_t* pos = reinterpret_cast<const uint32_t*>(ws + L.off_pos);
    if (n_halo > 0) {
        MH_REQUIRE(halo_ids, "halo_ids is null");
        write_halo_ids<<<grid_for(n_cols, 1 << 16), kBlock, 0, stream>>>(flags, pos, n_cols, n_halo, halo_ids);
        MH_LAUNCH_CHECK("write_halo_ids");
    }
    if (peer_ptr) {
        MH_REQUIRE(bounds, "bounds is null");
        write_peer_ptr<<<grid_for(world + 1), kBlock, 0, stream>>>(pos, bounds, world, n_cols, peer_ptr);
        MH_LAUNCH_CHECK("write_peer_ptr");
    }
    if (col_local && nnz > 0) {
        MH_REQUIRE(col_idx, "col_idx is null");
        renumber_columns<<<grid_for(nnz, 1 << 16), kBlock, 0, stream>>>(col_idx, nnz, c0, c1, pos, col_local);
        MH_LAUNCH_CHECK("renumber_columns");
    }
    return MH_OK;
}
