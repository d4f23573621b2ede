// Magnetic Laplacian builder for gfx950 (MI355X).
//
// Computes what hermitian_decomp_sparse (reference src/utils/hermitian.py:136-168)
// returns for norm=True, laplacian=True, as CSR with int32 indices and complex64
// values, entirely on the GPU:
//
//   1. every input edge (i, j, w) becomes two directed 64-bit records
//        key = row << (cb+1) | col << 1 | dir       (cb = bits of n_nodes-1)
//      (i, j, dir=0) "forward" and (j, i, dir=1) "backward";
//   2. one stable LSD radix sort over the 2*cb+1 live key bits (radix_sort.hip: tile histograms,
//      column scan, ballot-match ranking; duplicate edges keep their input order);
//   3. per (row, col) segment: w_fwd, w_bwd summed in fp32 in input order,
//      s = (w_fwd + w_bwd)/2, delta = w_fwd - w_bwd, keep = (delta != 0 && s != 0);
//   4. degrees d_row = sum_col s (fp64 over the row's records, or half the record count
//      when unweighted), d == 0 -> 1, dinv = d^-1/2;
//   5. exclusive scan of the keep flags (wavefront shuffle scan inside 4096-flag tiles,
//      tile sums scanned by one workgroup) gives every kept entry its CSR slot and every
//      row its row_ptr;
//   6. emission: col_idx, val = -(2/lmax) * cis(fl32(fl32(2 pi q) * delta)) * ((dinv_r * s) * dinv_c),
//      plus the diagonal (2/lmax - 1) when lmax != 2.
//
// Everything is deterministic (no atomics on data, fixed summation orders).
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include <hip/hip_runtime.h>

#include "common.h"
#include "radix_sort.h"
#include "scan.h"

namespace {

constexpr int kBlock = 256;
using mhscan::kScanTile;
using mhscan::scan_tile_sums;
using mhscan::scan_tiles;
using mhscan::scan_tile_write;

struct Header {          // first 256 bytes of the workspace
    int32_t err;         // != 0: an endpoint was out of range
    int32_t pad;
    int64_t kept;        // number of kept off-diagonal entries
};

struct Layout {
    int64_t e_total;     // edges incl. gcn self-loops
    int64_t m;           // directed records = 2 * e_total
    int64_t m_pad;       // m rounded up to the scan tile
    int64_t n_tiles;
    int cb;              // bits per node id
    size_t off_hdr, off_keys0, off_keys1, off_w0, off_w1, off_flags, off_pos, off_recptr, off_dinv,
        off_tiles, off_sort, sort_bytes, total;
};

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

int node_bits(int32_t n_nodes) {
    int cb = 1;
    while (cb < 31 && (int64_t(1) << cb) < int64_t(n_nodes)) ++cb;
    return cb;
}

Layout make_layout(int64_t n_edges, int32_t n_nodes, bool weighted, bool gcn) {
    Layout L{};
    L.e_total = n_edges + (gcn ? int64_t(n_nodes) : 0);
    L.m = 2 * L.e_total;
    L.n_tiles = L.m / kScanTile + 1;              // covers positions 0..m (pos[m] = total)
    L.m_pad = L.n_tiles * kScanTile;
    L.cb = node_bits(n_nodes);
    size_t o = 0;
    L.off_hdr = o;      o += 256;
    L.off_keys0 = o;    o += align_up(size_t(L.m_pad) * 8);
    L.off_keys1 = o;    o += align_up(size_t(L.m_pad) * 8);
    L.off_w0 = o;       o += weighted ? align_up(size_t(L.m_pad) * 4) : 0;
    L.off_w1 = o;       o += weighted ? align_up(size_t(L.m_pad) * 4) : 0;
    L.off_flags = o;    o += align_up(size_t(L.m_pad));
    L.off_pos = o;      o += align_up(size_t(L.m_pad + 1) * 4);
    L.off_recptr = o;   o += align_up(size_t(n_nodes + 1) * 4);
    L.off_dinv = o;     o += align_up(size_t(n_nodes) * 4 + 4);
    L.off_tiles = o;    o += align_up(size_t(L.n_tiles + 1) * 4);
    L.off_sort = o;
    L.sort_bytes = mhsort::workspace_bytes(L.m > 0 ? L.m : 1);
    o += align_up(L.sort_bytes);
    L.total = o;
    return L;
}

// ---------------------------------------------------------------------------
// 1. records
// ---------------------------------------------------------------------------
template <typename IdxT>
__global__ void __launch_bounds__(kBlock) make_records(const IdxT* __restrict__ row, const IdxT* __restrict__ col,
                                                      const float* __restrict__ weight, int64_t n_edges,
                                                      int64_t e_total, int32_t n_nodes, int cb,
                                                      uint64_t* __restrict__ keys, float* __restrict__ w_out,
                                                      Header* hdr) {
    int64_t e = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (e >= e_total) return;
    int64_t i, j;
    float w = 1.0f;
    if (e < n_edges) {
        i = int64_t(row[e]); j = int64_t(col[e]);
        if (weight) w = weight[e];
    } else {                       // gcn_appr self-loop of weight 1 (hermitian.py:144-145)
        i = j = e - n_edges;
    }
    if (i < 0 || j < 0 || i >= n_nodes || j >= n_nodes) {
        atomicOr(&hdr->err, 1);    // flag only; the record is made harmless below
        i = j = 0; w = 0.0f;
    }
    uint64_t ui = uint64_t(i), uj = uint64_t(j);
    keys[e] = (ui << (cb + 1)) | (uj << 1);
    keys[e_total + e] = (uj << (cb + 1)) | (ui << 1) | 1ull;
    if (w_out) { w_out[e] = w; w_out[e_total + e] = w; }
}

// ---------------------------------------------------------------------------
// 3a. first record of every row: rec_ptr[r] = first p with row(p) >= r ; rec_ptr[n] = m
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) row_starts(const uint64_t* __restrict__ keys, int64_t m, int32_t n_nodes,
                                                     int cb, int32_t* __restrict__ rec_ptr) {
    int64_t p = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (p > m) return;
    int64_t r_cur = (p < m) ? int64_t(keys[p] >> (cb + 1)) : int64_t(n_nodes);
    int64_t r_prev = (p > 0) ? int64_t(keys[p - 1] >> (cb + 1)) : -1;
    for (int64_t r = r_prev + 1; r <= r_cur; ++r) rec_ptr[r] = int32_t(p);
}

// ---------------------------------------------------------------------------
// 4. degrees -> d^-1/2.  One wavefront per row; weighted rows are summed in fp64 with a fixed
// lane-strided order and a butterfly, so the result does not depend on scheduling.
// ---------------------------------------------------------------------------
__device__ inline float inv_sqrt_degree(double sum) {
    float d = float(0.5 * sum);
    if (d == 0.0f) d = 1.0f;                       // hermitian.py:151
    return float(1.0 / sqrt(double(d)));           // d^-1/2 rounded once (hermitian.py:152)
}

__global__ void __launch_bounds__(kBlock) degree_inv_sqrt_weighted(const int32_t* __restrict__ rec_ptr,
                                                                   const float* __restrict__ w_sorted,
                                                                   int32_t n_nodes, float* __restrict__ dinv) {
    const int lane = threadIdx.x & 63;
    int64_t r = (int64_t(blockIdx.x) * kBlock + threadIdx.x) >> 6;
    if (r >= n_nodes) return;
    int32_t a = rec_ptr[r], b = rec_ptr[r + 1];
    double part = 0.0;
    for (int32_t p = a + lane; p < b; p += 64) part += double(w_sorted[p]);
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
    if (lane == 0) dinv[r] = inv_sqrt_degree(part);
}

// unweighted: every record has weight 1, so the degree is half the row's record count
__global__ void __launch_bounds__(kBlock) degree_inv_sqrt_counts(const int32_t* __restrict__ rec_ptr,
                                                                 int32_t n_nodes, float* __restrict__ dinv) {
    int64_t r = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (r >= n_nodes) return;
    dinv[r] = inv_sqrt_degree(double(rec_ptr[r + 1] - rec_ptr[r]));
}

// ---------------------------------------------------------------------------
// 3b. (row, col) segments.  The thread on a segment's first record walks the segment:
// forward records come first (they precede the backward ones in the input and the sort is stable;
// dir is the key's lowest bit), each side is summed in fp32 in
// stored (= input) order.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) segments(const uint64_t* __restrict__ keys, const float* __restrict__ w_sorted,
                                                   int64_t m, float* __restrict__ seg_s, float* __restrict__ seg_d,
                                                   uint8_t* __restrict__ flags) {
    int64_t p = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (p >= m) return;
    uint64_t k = keys[p];
    uint64_t pair = k >> 1;
    bool head = (p == 0) || ((keys[p - 1] >> 1) != pair);
    uint8_t keep = 0;
    if (head) {
        float fwd = 0.0f, bwd = 0.0f;
        int64_t t = p;
        uint64_t kt = k;
        while (true) {
            float w = w_sorted ? w_sorted[t] : 1.0f;
            if (kt & 1ull) bwd += w; else fwd += w;
            ++t;
            if (t >= m) break;
            kt = keys[t];
            if ((kt >> 1) != pair) break;
        }
        float s = 0.5f * (fwd + bwd);                  // hermitian.py:147
        float d = fwd - bwd;                           // hermitian.py:157
        seg_s[p] = s; seg_d[p] = d;
        keep = (d != 0.0f && s != 0.0f) ? 1 : 0;       // exact zeros are not stored by the CSR binops
    }
    flags[p] = keep;
}

__global__ void __launch_bounds__(kBlock) write_row_ptr(const int32_t* __restrict__ rec_ptr, const uint32_t* __restrict__ pos,
                                                        int32_t n_nodes, int diag, int32_t* __restrict__ row_ptr) {
    int64_t r = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (r > n_nodes) return;
    row_ptr[r] = int32_t(pos[rec_ptr[r]]) + (diag ? int32_t(r) : 0);
}

// ---------------------------------------------------------------------------
// 6. emission
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) emit_entries(const uint64_t* __restrict__ keys, const uint8_t* __restrict__ flags,
                                                       const uint32_t* __restrict__ pos, const float* __restrict__ seg_s,
                                                       const float* __restrict__ seg_d, const float* __restrict__ dinv,
                                                       int64_t m, int cb, float two_pi_q, float scale, int diag,
                                                       int64_t nnz, int32_t* __restrict__ col_idx,
                                                       float2* __restrict__ val) {
    int64_t p = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (p >= m || !flags[p]) return;
    uint64_t k = keys[p];
    int32_t r = int32_t(k >> (cb + 1));
    int32_t c = int32_t((k >> 1) & ((1ull << cb) - 1ull));
    int64_t idx = int64_t(pos[p]);
    if (diag) idx += int64_t(r) + (c > r ? 1 : 0);
    if (idx >= nnz) return;                            // caller's arrays are shorter than the count phase reported
    float a = (dinv[r] * seg_s[p]) * dinv[c];          // (D A_s) D, hermitian.py:154
    float theta = two_pi_q * seg_d[p];                 // fp32 product, hermitian.py:157
    float sn, cs;
    sincosf(theta, &sn, &cs);                          // full-range accurate sincos, hermitian.py:158
    float re = -(cs * a), im = -(sn * a);              // I - Theta.*A_s off the diagonal, hermitian.py:164
    col_idx[idx] = c;
    val[idx] = make_float2(re * scale, im * scale);    // (2/lmax) L - I, hermitian.py:167
}

// diagonal (2/lmax - 1) of every row; its slot follows the row's kept entries with col < r
__global__ void __launch_bounds__(kBlock) emit_diagonal(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ pos,
                                                        int64_t m, int32_t n_nodes, int cb, float diag_val, int64_t nnz,
                                                        int32_t* __restrict__ col_idx, float2* __restrict__ val) {
    int64_t r = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (r >= n_nodes) return;
    uint64_t target = (uint64_t(r) << (cb + 1)) | (uint64_t(r) << 1);
    int64_t lo = 0, hi = m;                             // lower_bound(keys, target)
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < target) lo = mid + 1; else hi = mid;
    }
    int64_t idx = int64_t(pos[lo]) + r;
    if (idx >= nnz) return;
    col_idx[idx] = int32_t(r);
    val[idx] = make_float2(diag_val, 0.0f);
}

inline unsigned grid_for(int64_t n) { return unsigned((n + kBlock - 1) / kBlock); }

}  // namespace

extern "C" size_t mh_lap_workspace_bytes(int64_t n_edges, int32_t n_nodes, int weighted, int gcn_appr) {
    if (n_edges < 0 || n_nodes < 0) return 0;
    return make_layout(n_edges, n_nodes, weighted != 0, gcn_appr != 0).total;
}

extern "C" int mh_lap_build_count(const void* row, const void* col, int idx64, const float* weight,
                                  int64_t n_edges, int32_t n_nodes, double max_eigen, int gcn_appr,
                                  void* workspace, size_t workspace_bytes, int32_t* row_ptr,
                                  int64_t* nnz_host, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    MH_REQUIRE(n_edges >= 0 && n_nodes >= 0, "negative size");
    MH_REQUIRE(n_edges == 0 || (row && col), "row/col is null");
    MH_REQUIRE(workspace && row_ptr && nnz_host, "workspace/row_ptr/nnz_host is null");
    MH_REQUIRE(max_eigen > 0.0, "max_eigen must be positive");
    const bool weighted = weight != nullptr, gcn = gcn_appr != 0;
    Layout L = make_layout(n_edges, n_nodes, weighted, gcn);
    if (L.m >= (int64_t(1) << 31) - 1 - n_nodes) return mh::fail(MH_ERR_RANGE, "2*edges does not fit int32 CSR indices");
    if (workspace_bytes < L.total)
        return mh::fail(MH_ERR_WORKSPACE, "workspace %zu < required %zu bytes", workspace_bytes, L.total);
    MH_REQUIRE(mh_aligned16(workspace), "workspace must be 16-byte aligned");
    char* ws = static_cast<char*>(workspace);
    Header* hdr = reinterpret_cast<Header*>(ws + L.off_hdr);
    uint64_t* keys0 = reinterpret_cast<uint64_t*>(ws + L.off_keys0);
    uint64_t* keys1 = reinterpret_cast<uint64_t*>(ws + L.off_keys1);
    float* w0 = weighted ? reinterpret_cast<float*>(ws + L.off_w0) : nullptr;
    float* w1 = weighted ? reinterpret_cast<float*>(ws + L.off_w1) : nullptr;
    uint8_t* flags = reinterpret_cast<uint8_t*>(ws + L.off_flags);
    uint32_t* pos = reinterpret_cast<uint32_t*>(ws + L.off_pos);
    int32_t* rec_ptr = reinterpret_cast<int32_t*>(ws + L.off_recptr);
    float* dinv = reinterpret_cast<float*>(ws + L.off_dinv);
    uint32_t* tiles = reinterpret_cast<uint32_t*>(ws + L.off_tiles);
    const float scale = float(2.0 / max_eigen);                 // the double quotient rounded once (hermitian.py:167)
    const int diag = (scale * 1.0f - 1.0f) != 0.0f;

    MH_HIP(hipMemsetAsync(hdr, 0, 256, stream));
    MH_HIP(hipMemsetAsync(flags, 0, size_t(L.m_pad), stream));
    // The sort key is (row, col) = key bits [1, 2cb+1).  The direction bit 0 is NOT sorted: all forward
    // records are written before all backward records and the sort is stable, so inside a (row, col)
    // segment the forward records still come first, each side in input order.
    // An odd number of sort passes moves the records keys0 -> keys1; for an even number they start in keys1.
    const bool start_in_0 = mhsort::passes_for(2 * L.cb) & 1;
    uint64_t* rec_keys = start_in_0 ? keys0 : keys1;
    float* rec_w = weighted ? (start_in_0 ? w0 : w1) : nullptr;
    if (L.e_total > 0) {
        if (idx64)
            make_records<int64_t><<<grid_for(L.e_total), kBlock, 0, stream>>>(
                static_cast<const int64_t*>(row), static_cast<const int64_t*>(col), weight, n_edges, L.e_total,
                n_nodes, L.cb, rec_keys, rec_w, hdr);
        else
            make_records<int32_t><<<grid_for(L.e_total), kBlock, 0, stream>>>(
                static_cast<const int32_t*>(row), static_cast<const int32_t*>(col), weight, n_edges, L.e_total,
                n_nodes, L.cb, rec_keys, rec_w, hdr);
        MH_LAUNCH_CHECK("make_records");
        // passes alternate between the two buffers; the records were written so that the result lands in keys1
        uint64_t* other = rec_keys == keys0 ? keys1 : keys0;
        float* other_w = weighted ? (rec_w == w0 ? w1 : w0) : nullptr;
        MH_HIP(mhsort::sort(rec_keys, other, rec_w, other_w, L.m, 1, 2 * L.cb, ws + L.off_sort, L.sort_bytes, stream));
    }
    row_starts<<<grid_for(L.m + 1), kBlock, 0, stream>>>(keys1, L.m, n_nodes, L.cb, rec_ptr);
    MH_LAUNCH_CHECK("row_starts");
    if (n_nodes > 0) {
        if (weighted)
            degree_inv_sqrt_weighted<<<grid_for(int64_t(n_nodes) * 64), kBlock, 0, stream>>>(rec_ptr, w1, n_nodes, dinv);
        else
            degree_inv_sqrt_counts<<<grid_for(n_nodes), kBlock, 0, stream>>>(rec_ptr, n_nodes, dinv);
        MH_LAUNCH_CHECK("degree_inv_sqrt");
    }
    float* seg_s = reinterpret_cast<float*>(keys0);             // keys0 is free after the sort
    float* seg_d = seg_s + L.m_pad;
    if (L.m > 0) {
        segments<<<grid_for(L.m), kBlock, 0, stream>>>(keys1, w1, L.m, seg_s, seg_d, flags);
        MH_LAUNCH_CHECK("segments");
    }
    scan_tile_sums<<<unsigned(L.n_tiles), kBlock, 0, stream>>>(flags, tiles);
    MH_LAUNCH_CHECK("scan_tile_sums");
    scan_tiles<<<1, 1024, 0, stream>>>(tiles, L.n_tiles, &hdr->kept);
    MH_LAUNCH_CHECK("scan_tiles");
    scan_tile_write<<<unsigned(L.n_tiles), kBlock, 0, stream>>>(flags, tiles, pos, L.m);
    MH_LAUNCH_CHECK("scan_tile_write");
    write_row_ptr<<<grid_for(int64_t(n_nodes) + 1), kBlock, 0, stream>>>(rec_ptr, pos, n_nodes, diag, row_ptr);
    MH_LAUNCH_CHECK("write_row_ptr");
    Header h{};
    MH_HIP(hipMemcpyAsync(&h, hdr, sizeof(Header), hipMemcpyDeviceToHost, stream));
    MH_HIP(hipStreamSynchronize(stream));
    if (h.err) return mh::fail(MH_ERR_INDEX, "an edge endpoint is outside [0, %d)", n_nodes);
    *nnz_host = h.kept + (diag ? int64_t(n_nodes) : 0);
    return MH_OK;
}

extern "C" int mh_lap_build_fill(int64_t n_edges, int32_t n_nodes, double q, double max_eigen, int weighted_,
                                 int gcn_appr, void* workspace, size_t workspace_bytes, int64_t nnz,
                                 int32_t* col_idx, float* val, float* dinv_out, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    MH_REQUIRE(nnz >= 0, "negative nnz");
    MH_REQUIRE(n_edges >= 0 && n_nodes >= 0 && workspace, "bad size or null workspace");
    MH_REQUIRE(max_eigen > 0.0, "max_eigen must be positive");
    const bool weighted = weighted_ != 0, gcn = gcn_appr != 0;
    Layout L = make_layout(n_edges, n_nodes, weighted, gcn);
    if (workspace_bytes < L.total)
        return mh::fail(MH_ERR_WORKSPACE, "workspace %zu < required %zu bytes", workspace_bytes, L.total);
    char* ws = static_cast<char*>(workspace);
    const uint64_t* keys1 = reinterpret_cast<const uint64_t*>(ws + L.off_keys1);
    const uint8_t* flags = reinterpret_cast<const uint8_t*>(ws + L.off_flags);
    const uint32_t* pos = reinterpret_cast<const uint32_t*>(ws + L.off_pos);
    const float* dinv = reinterpret_cast<const float*>(ws + L.off_dinv);
    const float* seg_s = reinterpret_cast<const float*>(ws + L.off_keys0);
    const float* seg_d = seg_s + L.m_pad;
    const float scale = float(2.0 / max_eigen);
    const float diag_val = scale * 1.0f - 1.0f;
    const int diag = diag_val != 0.0f;
    // 2*pi*q as the reference forms it: a double product rounded to fp32 when it meets the
    // float32 matrix (complex64 scalar), hermitian.py:157
    const float two_pi_q = float(2.0 * 3.14159265358979323846 * q);
    if (L.m > 0 && nnz > 0) {
        MH_REQUIRE(col_idx && val, "col_idx/val is null");
        MH_REQUIRE((reinterpret_cast<uintptr_t>(val) & 7u) == 0, "val must be 8-byte aligned");
        emit_entries<<<grid_for(L.m), kBlock, 0, stream>>>(keys1, flags, pos, seg_s, seg_d, dinv, L.m, L.cb, two_pi_q,
                                                           scale, diag, nnz, col_idx, reinterpret_cast<float2*>(val));
        MH_LAUNCH_CHECK("emit_entries");
    }
    if (diag && n_nodes > 0) {
        MH_REQUIRE(col_idx && val && nnz >= n_nodes, "col_idx/val is null or nnz too small for the diagonal");
        emit_diagonal<<<grid_for(n_nodes), kBlock, 0, stream>>>(keys1, pos, L.m, n_nodes, L.cb, diag_val, nnz, col_idx,
                                                                reinterpret_cast<float2*>(val));
        MH_LAUNCH_CHECK("emit_diagonal");
    }
    if (dinv_out && n_nodes > 0)
        MH_HIP(hipMemcpyAsync(dinv_out, dinv, size_t(n_nodes) * 4, hipMemcpyDeviceToDevice, stream));
    return MH_OK;
}
