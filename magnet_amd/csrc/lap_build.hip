// Magnetic Laplacian builder for gfx950 (MI355X).
//
// Computes what hermitian_decomp_sparse (reference src/utils/hermitian.py:136-168)
// returns for norm=True, laplacian=True, as CSR with int32 indices and complex64
// values, entirely on the GPU:
//
//   1. every input edge (i, j, w) becomes two directed 64-bit records
//        key = row << (cb+1) | col << 1 | dir       (cb = bits of n_nodes-1)
//      (i, j, dir=0) "forward" and (j, i, dir=1) "backward"; the records are made inside the first pass of the
//      sort (its histogram and its scatter read the edge lists), so they are never written unsorted;
//   2. one stable LSD radix sort over the 2*cb live (row, col) key bits (radix_sort.hip: tile histograms,
//      column scan, ballot-match ranking; duplicate edges keep their input order).  Unweighted whole-graph
//      builds whose column digits fill whole passes stop once after the column passes, when the records are
//      grouped by column: the size of a column's group IS that node's record count (every record (i, j) has its
//      mirror (j, i)), i.e. its degree, and it is written into the key's spare top bits on the way through the
//      next pass ("tagged" keys) - the emission then computes d^-1/2 of an entry's column from the tag instead
//      of gathering it from the degree table, the one access of the build that has no order;
//   3. per (row, col) segment: w_fwd, w_bwd summed in fp32 in input order,
//      s = (w_fwd + w_bwd)/2, delta = w_fwd - w_bwd, keep = (delta != 0 && s != 0);
//   4. degrees d_row = sum_col s (fp64 over the row's records, or half the record count
//      when unweighted), d == 0 -> 1, dinv = d^-1/2;
//   5. the keep flags are one bit per record (a wavefront's ballot = one 64-bit word per 64 records); an exclusive
//      scan of the words' population counts gives every word its first CSR slot, and a record's slot is that plus
//      the set bits below it in its word - no per-record flag or slot array exists;
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
using mhscan::scan_tiles;

constexpr int kLongRun = 64;          // (row, col) runs longer than this are summed by a whole wavefront
constexpr int kTagBits = 15;          // spare key bits that carry a column's record count (tagged keys)
constexpr uint32_t kTagCap = (1u << kTagBits) - 1u;     // this value = "too many to fit: look the degree up"

struct Header {          // first 256 bytes of the workspace
    int32_t err;         // bit 0: an endpoint outside [0, n_nodes); bit 1: a record outside the row block
    int32_t n_long;      // number of (row, col) runs longer than kLongRun records
    int64_t kept;        // number of kept off-diagonal entries
};

constexpr int kWordsPerThread = 8;
constexpr int kWordTile = kBlock * kWordsPerThread;      // 2048 flag words (131072 records) per workgroup of the scan

struct Layout {
    int64_t m;           // directed records
    int64_t m_pad;       // m rounded up to a multiple of 4096 records (the two key buffers then lie 32 KiB-aligned apart)
    int64_t n_words;     // 64-bit flag words incl. padding: a multiple of kWordTile covering positions 0..m
    int64_t n_tiles;     // n_words / kWordTile
    int cb;              // bits per node id
    size_t off_hdr, off_keys0, off_keys1, off_w0, off_w1, off_bits, off_wpos, off_recptr, off_dinv,
        off_tiles, off_long, off_tagtab, off_sort, sort_bytes, total;
};

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

int node_bits(int32_t n_nodes) {
    int cb = 1;
    while (cb < 31 && (int64_t(1) << cb) < int64_t(n_nodes)) ++cb;
    return cb;
}

// workspace for m directed records over n_nodes nodes
Layout make_layout(int64_t m, int32_t n_nodes, bool weighted) {
    Layout L{};
    L.m = m;
    L.m_pad = (m / 4096 + 1) * 4096;
    L.n_tiles = (m / 64 + 1 + kWordTile - 1) / kWordTile;      // words 0 .. m / 64 (position m needs its word)
    L.n_words = L.n_tiles * kWordTile;
    L.cb = node_bits(n_nodes);
    size_t o = 0;
    L.off_hdr = o;      o += 256;
    L.off_keys0 = o;    o += align_up(size_t(L.m_pad) * 8);
    L.off_keys1 = o;    o += align_up(size_t(L.m_pad) * 8);
    L.off_w0 = o;       o += weighted ? align_up(size_t(L.m_pad) * 4) : 0;
    L.off_w1 = o;       o += weighted ? align_up(size_t(L.m_pad) * 4) : 0;
    L.off_bits = o;     o += align_up(size_t(L.n_words) * 8);
    L.off_wpos = o;     o += align_up(size_t(L.n_words) * 4);
    L.off_recptr = o;   o += align_up(size_t(n_nodes + 1) * 4);
    L.off_dinv = o;     o += align_up(size_t(n_nodes) * 4 + 4);
    L.off_tiles = o;    o += align_up(size_t(L.n_tiles + 1) * 4);
    L.off_long = o;     o += align_up(size_t(L.m / (kLongRun + 1) + 1) * 4);
    L.off_tagtab = o;   o += align_up(size_t(kTagCap) * 4);
    L.off_sort = o;
    L.sort_bytes = mhsort::workspace_bytes(L.m > 0 ? L.m : 1);
    o += align_up(L.sort_bytes);
    L.total = o;
    return L;
}

// Tagged keys (see the header): unweighted builds of the whole graph, when the column bits fill whole sort passes (so the
// sort can stop between columns and rows at no extra pass) and the key has kTagBits spare bits.
inline bool use_tags(const Layout& L, bool weighted, bool whole_graph) {
    return whole_graph && !weighted && L.m > 0 && 2 * mhsort::passes_for(L.cb) == mhsort::passes_for(2 * L.cb) &&
           2 * L.cb + 1 + kTagBits <= 64;
}

inline int64_t full_records(int64_t n_edges, int32_t n_nodes, bool gcn) {
    return 2 * (n_edges + (gcn ? int64_t(n_nodes) : 0));
}

// ---------------------------------------------------------------------------
// 3a. first record of every row: rec_ptr[r] = first p with row(p) >= r ; rec_ptr[n] = m.  One binary search
// per row (neighbouring rows end next to each other, so a wave's searches share their cache lines); cost
// does not depend on how the records are spread over the rows.
// ---------------------------------------------------------------------------
// The same search finds the first record of every COLUMN while the records are ordered by column only (between the
// column passes and the row passes of the sort): `shift` / `mask` select the field the records are ordered by.
__global__ void __launch_bounds__(kBlock) row_starts(const uint64_t* __restrict__ keys, int64_t m, int32_t n_nodes,
                                                     int shift, uint64_t mask, int32_t* __restrict__ rec_ptr) {
    int64_t r = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (r > n_nodes) return;
    const uint64_t target = uint64_t(r);
    int64_t lo = 0, hi = m;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (((keys[mid] >> shift) & mask) < target) lo = mid + 1; else hi = mid;
    }
    rec_ptr[r] = int32_t(lo);
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

// Weighted degrees: the row's weights summed in double precision.  A wavefront takes four rows, 16 lanes each (the
// average row has a few dozen records); a row of more than kShortRow records is left to the whole wavefront afterwards,
// and one of more than kWaveRow to the whole workgroup, with four independent partial sums per lane so that the loads
// are in flight together (a hub row walked by 16 or 64 lanes would set the kernel's duration: one chain of dependent
// iterations).  The summation order is a fixed function of the row's length, so a row's value is the same in a
// whole-graph build and in a row-block build.
constexpr int kShortRow = 512;
constexpr int kWaveRow = 8192;

__global__ void __launch_bounds__(kBlock) degree_inv_sqrt_weighted(const int32_t* __restrict__ rec_ptr,
                                                                   const float* __restrict__ w_sorted,
                                                                   int32_t n_nodes, float* __restrict__ dinv) {
    constexpr int kRowsPerBlock = kBlock / 16;
    __shared__ int32_t hub_a[kRowsPerBlock], hub_b[kRowsPerBlock], hub_r[kRowsPerBlock];
    __shared__ int hub_n;
    __shared__ double wave_sum[kBlock / 64];
    if (threadIdx.x == 0) hub_n = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, sl = lane & 15;
    const int64_t r = int64_t(blockIdx.x) * kRowsPerBlock + (threadIdx.x >> 4);
    int32_t a = 0, b = 0;
    if (r < n_nodes) { a = rec_ptr[r]; b = rec_ptr[r + 1]; }
    const bool is_hub = b - a > kWaveRow;
    const bool is_long = b - a > kShortRow && !is_hub;
    if (is_hub && sl == 0) {
        const int i = atomicAdd(&hub_n, 1);
        hub_a[i] = a; hub_b[i] = b; hub_r[i] = int32_t(r);
    }
    double part = 0.0;
    if (b - a <= kShortRow)
        for (int32_t p = a + sl; p < b; p += 16) part += double(w_sorted[p]);
    for (int off = 8; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);      // stays inside the 16 lanes
    if (r < n_nodes && b - a <= kShortRow && sl == 0) dinv[r] = inv_sqrt_degree(part);
    uint64_t todo = __ballot(is_long && sl == 0);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int32_t la = __shfl(a, src, 64), lb = __shfl(b, src, 64);
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int32_t p = la + lane;
        for (; p + 192 < lb; p += 256) {
            s0 += double(w_sorted[p]); s1 += double(w_sorted[p + 64]);
            s2 += double(w_sorted[p + 128]); s3 += double(w_sorted[p + 192]);
        }
        for (; p < lb; p += 64) s0 += double(w_sorted[p]);
        double s = (s0 + s1) + (s2 + s3);
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        const int64_t lr = __shfl(r, src, 64);
        if (lane == 0) dinv[lr] = inv_sqrt_degree(s);
    }
    __syncthreads();
    const int hubs = hub_n;
    for (int i = 0; i < hubs; ++i) {
        const int32_t la = hub_a[i], lb = hub_b[i];
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int32_t p = la + int32_t(threadIdx.x);
        for (; p + 3 * kBlock < lb; p += 4 * kBlock) {
            s0 += double(w_sorted[p]); s1 += double(w_sorted[p + kBlock]);
            s2 += double(w_sorted[p + 2 * kBlock]); s3 += double(w_sorted[p + 3 * kBlock]);
        }
        for (; p < lb; p += kBlock) s0 += double(w_sorted[p]);
        double s = (s0 + s1) + (s2 + s3);
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) wave_sum[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int w = 0; w < kBlock / 64; ++w) t += wave_sum[w];
            dinv[hub_r[i]] = inv_sqrt_degree(t);
        }
        __syncthreads();
    }
}

// d^-1/2 for every record count a tagged key can carry (the emission looks a column's value up here instead of
// repeating the double-precision square root and division per entry; same function, same bits)
__global__ void __launch_bounds__(kBlock) tag_table_kernel(float* __restrict__ table) {
    const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
    if (t < kTagCap) table[t] = inv_sqrt_degree(double(t));
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
// dir is the key's lowest bit), each side is summed in fp32 in stored (= input) order.
// A run longer than kLongRun records (the same directed pair listed hundreds of times) is not walked
// by one thread: its head is put on a list and a whole wavefront sums it (segments_long).
// ---------------------------------------------------------------------------
__device__ inline void segment_values(float fwd, float bwd, float& s, float& d) {
    s = 0.5f * (fwd + bwd);                        // hermitian.py:147
    d = fwd - bwd;                                 // hermitian.py:157
}
__device__ inline uint8_t segment_kept(float s, float d) {
    return (d != 0.0f && s != 0.0f) ? 1 : 0;       // exact zeros are not stored by the CSR binops
}

// CSR slot of record p: kept records before its word (scan of the words' population counts) + set bits below it
__device__ inline uint32_t slot_of(const uint64_t* __restrict__ bits, const uint32_t* __restrict__ word_pos, int64_t p) {
    const uint64_t below = (1ull << (p & 63)) - 1ull;
    return word_pos[p >> 6] + uint32_t(__popcll(bits[p >> 6] & below));
}

// What a wavefront knows about the runs among its 64 consecutive records p = base + lane, from three ballots: which
// lanes hold a record (a prefix), which records start a (row, col) run, which are backward records.
struct WaveRuns {
    uint64_t in_mask, head_mask, dir_mask;
    uint64_t next_key;       // the record after the wave's 64 (read up front, beside the wave's own records)
    bool has_next;
};

// 64-record groups a wavefront of segments / emit_entries takes: their loads are issued together, so that four times
// the bytes are in flight per wave (one group per wave left both kernels waiting on memory latency, not bandwidth)
constexpr int kGroups = 4;

// k / in = the lane's record and whether p < m; every lane of the wave calls this.  `before` = the key of the record
// in front of the group (used by lane 0; any value when p - lane == 0), `next_key` / `has_next` = the record after it.
__device__ inline WaveRuns wave_runs(int64_t p, int lane, uint64_t k, bool in, uint64_t before, uint64_t next_key,
                                     bool has_next, bool& head) {
    WaveRuns r;
    r.has_next = has_next;
    r.next_key = next_key;
    const uint32_t plo = __shfl_up(uint32_t(k), 1, 64), phi = __shfl_up(uint32_t(k >> 32), 1, 64);
    uint64_t prev = (uint64_t(phi) << 32) | plo;
    if (lane == 0) prev = before;
    head = in && (p == 0 || (prev >> 1) != (k >> 1));
    r.in_mask = __ballot(in);
    r.head_mask = __ballot(head);
    r.dir_mask = __ballot(in && (k & 1ull));
    return r;
}

// the records of a wavefront's kGroups groups, and the keys around them, loaded in one go
struct WaveRecords {
    uint64_t k[kGroups];
    float w[kGroups];
    bool in[kGroups];
    uint64_t before, after;
    bool has_after;
};

__device__ inline WaveRecords load_records(const uint64_t* __restrict__ keys, const float* __restrict__ w_sorted, int64_t m,
                                           int64_t base, int lane) {
    WaveRecords R;
#pragma unroll
    for (int u = 0; u < kGroups; ++u) {
        const int64_t p = base + u * 64 + lane;
        R.in[u] = p < m;
        R.k[u] = R.in[u] ? keys[p] : 0ull;
        R.w[u] = (R.in[u] && w_sorted) ? w_sorted[p] : 1.0f;
    }
    R.before = base > 0 ? keys[base - 1] : 0ull;
    const int64_t after = base + 64 * kGroups;
    R.has_after = after < m;
    R.after = R.has_after ? keys[after] : 0ull;
    return R;
}

__device__ inline uint64_t lane_value(uint64_t v, int src_lane) {      // v of lane src_lane (0 or 63), wave-uniform
    const uint32_t lo = src_lane == 0 ? __builtin_amdgcn_readfirstlane(uint32_t(v)) : __builtin_amdgcn_readlane(uint32_t(v), 63);
    const uint32_t hi = src_lane == 0 ? __builtin_amdgcn_readfirstlane(uint32_t(v >> 32)) : __builtin_amdgcn_readlane(uint32_t(v >> 32), 63);
    return (uint64_t(hi) << 32) | lo;
}

// group u of a wavefront's records: its run structure
__device__ inline WaveRuns group_runs(const WaveRecords& R, int u, int64_t m, int64_t base, int lane, bool& head) {
    const uint64_t before = u == 0 ? R.before : lane_value(R.k[u > 0 ? u - 1 : 0], 63);
    const bool last = u + 1 == kGroups;
    const uint64_t next_key = last ? R.after : lane_value(R.k[last ? u : u + 1], 0);
    const bool has_next = last ? R.has_after : (base + (u + 1) * 64 < m);
    return wave_runs(base + u * 64 + lane, lane, R.k[u], R.in[u], before, next_key, has_next, head);
}

// Sums of the run that starts at this lane's record (meaningful for run heads; every lane of the wave calls this, `want`
// says whether the lane needs its sums).  The run is walked in stored order, forward weights (dir bit 0) and backward
// weights added separately - the fp32 sequence of the reference's duplicate summation.  Inside the wave the run's extent
// comes from the ballots: unit weights are counted (exact in fp32), real weights arrive by lane shuffles; a run that
// reaches the end of the wave is continued from memory.  Returns false when the run is longer than kLongRun records
// (its sums then come from segments_long).
__device__ inline bool run_sums(const uint64_t* __restrict__ keys, const float* __restrict__ w_sorted, int64_t m, int64_t p,
                                int lane, uint64_t k, float w, const WaveRuns& r, bool want, float& fwd, float& bwd) {
    int end, len;
    if (r.head_mask == r.in_mask) {
        // every record of the wave is a run of its own (the usual case: duplicate edges are rare) - wave-uniform test
        end = lane == 63 ? 64 : lane + 1;
        len = 1;
        fwd = (k & 1ull) ? 0.0f : w;
        bwd = (k & 1ull) ? w : 0.0f;
    } else if (!w_sorted) {
        const uint64_t above = lane == 63 ? 0ull : (~0ull << (lane + 1));
        const uint64_t stop = (r.head_mask | ~r.in_mask) & above;     // the next run, or the end of the records
        end = stop ? __ffsll((unsigned long long)stop) - 1 : 64;
        const uint64_t range = (end == 64 ? ~0ull : ((1ull << end) - 1ull)) & ~((1ull << lane) - 1ull);
        len = end - lane;
        fwd = float(__popcll(range & ~r.dir_mask));
        bwd = float(__popcll(range & r.dir_mask));
    } else {
        const uint64_t above = lane == 63 ? 0ull : (~0ull << (lane + 1));
        const uint64_t stop = (r.head_mask | ~r.in_mask) & above;
        end = stop ? __ffsll((unsigned long long)stop) - 1 : 64;
        len = end - lane;
        fwd = (k & 1ull) ? 0.0f : w;
        bwd = (k & 1ull) ? w : 0.0f;
        int longest = want ? len : 0;                   // the longest wanted run of the wave
        for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(longest, off, 64); longest = o > longest ? o : longest; }
        longest = __builtin_amdgcn_readfirstlane(longest);
        for (int j = 1; j < longest; ++j) {             // wave-uniform trip count
            const float nw = __shfl_down(w, j, 64);
            if (want && lane + j < end) {
                if ((r.dir_mask >> (lane + j)) & 1ull) bwd += nw; else fwd += nw;
            }
        }
    }
    const uint64_t pair = k >> 1;
    if (want && end == 64 && r.has_next && (r.next_key >> 1) == pair) {   // the run continues past the wave: from memory
        int64_t t = p - lane + 64;
        while (t < m) {
            const uint64_t kt = keys[t];
            if ((kt >> 1) != pair) break;
            if (len == kLongRun) return false;
            const float wt = w_sorted ? w_sorted[t] : 1.0f;
            if (kt & 1ull) bwd += wt; else fwd += wt;
            ++len; ++t;
        }
    }
    return true;
}

// Keep flags, one bit per record: a short run's head computes its sums and the keep rule; the wave's ballot is the flag
// word of its 64 records.  Nothing else is stored for short runs: the emission walks them again (from its own registers),
// which is cheaper than writing s and delta for every record and reading them back.  Long runs go on a list; their flag
// bit is set by segments_long.
__global__ void __launch_bounds__(kBlock) segments(const uint64_t* __restrict__ keys, const float* __restrict__ w_sorted,
                                                   int64_t m, uint64_t* __restrict__ bits, uint32_t* __restrict__ long_list,
                                                   Header* hdr) {
    const int lane = threadIdx.x & 63;
    const int64_t base = ((int64_t(blockIdx.x) * kBlock + threadIdx.x) >> 6) * (64 * kGroups);
    if (base >= m) return;                              // the whole wavefront lies past the records
    const WaveRecords R = load_records(keys, w_sorted, m, base, lane);
#pragma unroll
    for (int u = 0; u < kGroups; ++u) {
        const int64_t p = base + u * 64 + lane;
        if (p - lane >= m) break;                       // wave-uniform
        bool head;
        const WaveRuns runs = group_runs(R, u, m, base, lane, head);
        float fwd, bwd;
        const bool is_short = run_sums(keys, w_sorted, m, p, lane, R.k[u], R.w[u], runs, head, fwd, bwd);
        bool keep = false;
        if (head) {
            if (is_short) {
                float sv, dv;
                segment_values(fwd, bwd, sv, dv);
                keep = segment_kept(sv, dv) != 0;
            } else {
                long_list[atomicAdd(&hdr->n_long, 1)] = uint32_t(p);     // flag bit set by segments_long
            }
        }
        const uint64_t word = __ballot(keep);
        if (lane == 0) bits[p >> 6] = word;
    }
}

// One wavefront per long run, 64 records per step: the lanes load the step's records side by side (coalesced),
// then the weights are added in stored order, the same fp32 sequence the walk of a short run
// performs (every lane carries the same running sums), at one shuffle + add per record instead of one
// dependent memory access per record.  Unweighted runs are counted.
// The grid is fixed; the waves stride over the list whose length the kernel before left in the header.
__global__ void __launch_bounds__(kBlock) segments_long(const uint64_t* __restrict__ keys, const float* __restrict__ w_sorted,
                                                        int64_t m, float* __restrict__ seg_s, float* __restrict__ seg_d,
                                                        uint64_t* __restrict__ bits, const uint32_t* __restrict__ long_list,
                                                        const Header* hdr) {
    const int lane = threadIdx.x & 63;
    const int n_long = hdr->n_long;
    const int n_waves = int(gridDim.x) * (kBlock / 64);
    for (int li = int(blockIdx.x) * (kBlock / 64) + (threadIdx.x >> 6); li < n_long; li += n_waves) {
        const int64_t p = int64_t(long_list[li]);
        const uint64_t pair = keys[p] >> 1;
        float fwd = 0.0f, bwd = 0.0f;
        for (int64_t base = p;; base += 64) {
            const int64_t t = base + lane;
            uint64_t kt = 0;
            bool in_run = false;
            float w = 1.0f;
            if (t < m) { kt = keys[t]; in_run = (kt >> 1) == pair; }
            if (in_run && w_sorted) w = w_sorted[t];
            const uint64_t run_mask = __ballot(in_run);                   // a prefix of the lanes: the records are sorted
            const uint64_t bwd_mask = __ballot(in_run && (kt & 1ull));
            const int cnt = __popcll(run_mask);
            if (w_sorted) {
                for (int j = 0; j < cnt; ++j) {
                    const float wj = __shfl(w, j, 64);
                    if ((bwd_mask >> j) & 1ull) bwd += wj; else fwd += wj;
                }
            } else {
                fwd += float(__popcll(run_mask & ~bwd_mask));
                bwd += float(__popcll(bwd_mask));
            }
            if (cnt < 64) break;                                          // the run ended inside these 64 records
        }
        if (lane == 0) {
            float sv, dv;
            segment_values(fwd, bwd, sv, dv);
            seg_s[p] = sv; seg_d[p] = dv;
            if (segment_kept(sv, dv)) atomicOr(reinterpret_cast<unsigned long long*>(bits + (p >> 6)), 1ull << (p & 63));
        }
    }
}

// Scan of the flag words' population counts: word_pos[i] = kept records before word i.  Tile sums, one workgroup over
// the tile sums (mhscan::scan_tiles), then the words' prefixes.
__global__ void __launch_bounds__(kBlock) word_tile_sums(const uint64_t* __restrict__ bits, uint32_t* __restrict__ tile_sums) {
    __shared__ uint32_t wsum[kBlock / 64];
    const uint64_t* b = bits + size_t(blockIdx.x) * kWordTile + size_t(threadIdx.x) * kWordsPerThread;
    uint32_t v = 0;
#pragma unroll
    for (int i = 0; i < kWordsPerThread; ++i) v += uint32_t(__popcll(b[i]));
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ void __launch_bounds__(kBlock) word_tile_write(const uint64_t* __restrict__ bits, const uint32_t* __restrict__ tile_sums,
                                                         uint32_t* __restrict__ word_pos) {
    __shared__ uint32_t wtot[kBlock / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t base = size_t(blockIdx.x) * kWordTile + size_t(threadIdx.x) * kWordsPerThread;
    uint32_t c[kWordsPerThread], tsum = 0;
#pragma unroll
    for (int i = 0; i < kWordsPerThread; ++i) { c[i] = uint32_t(__popcll(bits[base + i])); tsum += c[i]; }
    uint32_t inc = mhscan::wave_inclusive_scan(tsum, lane);
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    uint32_t off = tile_sums[blockIdx.x] + inc - tsum;
    for (int w = 0; w < wave; ++w) off += wtot[w];
#pragma unroll
    for (int i = 0; i < kWordsPerThread; ++i) { word_pos[base + i] = off; off += c[i]; }
}

// row_ptr of the rows [row_lo, row_hi]: CSR slot of the row's first record (+ one diagonal per earlier row)
__global__ void __launch_bounds__(kBlock) write_row_ptr(const int32_t* __restrict__ rec_ptr, const uint64_t* __restrict__ bits,
                                                        const uint32_t* __restrict__ word_pos, int32_t row_lo, int32_t row_hi,
                                                        int diag, int32_t* __restrict__ row_ptr) {
    int64_t r = int64_t(row_lo) + int64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (r > row_hi) return;
    row_ptr[r - row_lo] = int32_t(slot_of(bits, word_pos, rec_ptr[r])) + (diag ? int32_t(r - row_lo) : 0);
}

// ---------------------------------------------------------------------------
// 6. emission: a wavefront takes kGroups flag words (64 records each)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) emit_entries(const uint64_t* __restrict__ keys, const float* __restrict__ w_sorted,
                                                       const uint64_t* __restrict__ bits,
                                                       const uint32_t* __restrict__ word_pos, const float* __restrict__ seg_s,
                                                       const float* __restrict__ seg_d, const float* __restrict__ dinv,
                                                       const float* __restrict__ tag_table,
                                                       int64_t m, int cb, int32_t row_lo, float two_pi_q, float scale, int diag,
                                                       int64_t nnz, int tagged, int32_t* __restrict__ col_idx,
                                                       float2* __restrict__ val) {
    const int lane = threadIdx.x & 63;
    const int64_t base = ((int64_t(blockIdx.x) * kBlock + threadIdx.x) >> 6) * (64 * kGroups);
    if (base >= m) return;
    const WaveRecords R = load_records(keys, w_sorted, m, base, lane);
    uint64_t words[kGroups];
    uint32_t first_slot[kGroups];
#pragma unroll
    for (int u = 0; u < kGroups; ++u) {
        const bool any = base + u * 64 < m;
        words[u] = any ? bits[(base >> 6) + u] : 0ull;
        first_slot[u] = any ? word_pos[(base >> 6) + u] : 0u;
    }
    const uint64_t node_mask = (1ull << cb) - 1ull;
    // d^-1/2 of the kept entries' rows and columns depends on the keys alone: all groups' lookups are issued here,
    // together, instead of one group's after another's arithmetic
    float d_row[kGroups], d_col[kGroups];
#pragma unroll
    for (int u = 0; u < kGroups; ++u) {
        const uint64_t k = R.k[u];
        const bool kept = (words[u] >> lane) & 1ull;
        d_row[u] = d_col[u] = 0.0f;
        if (kept) {
            // a column that occurs in a record has at least one record: a zero tag is a key that was not tagged
            const uint32_t tag = tagged ? uint32_t(k >> (2 * cb + 1)) : 0u;
            d_row[u] = dinv[int32_t((k >> (cb + 1)) & node_mask)];
            d_col[u] = (tag != 0u && tag < kTagCap) ? tag_table[tag]      // the column's record count travelled in the key
                                                    : dinv[int32_t((k >> 1) & node_mask)];
        }
    }
#pragma unroll
    for (int u = 0; u < kGroups; ++u) {
        const uint64_t word = words[u];
        if (word == 0ull) continue;                     // no kept entry among these 64 records (wave-uniform)
        const int64_t p = base + u * 64 + lane;
        const uint64_t k = R.k[u];
        const bool kept = (word >> lane) & 1ull;        // kept entries are run heads
        bool head;
        const WaveRuns runs = group_runs(R, u, m, base, lane, head);
        float fwd, bwd, sv, dv;
        const bool is_short = run_sums(keys, w_sorted, m, p, lane, k, R.w[u], runs, kept, fwd, bwd);
        if (!kept) continue;
        if (is_short) segment_values(fwd, bwd, sv, dv);
        else { sv = seg_s[p]; dv = seg_d[p]; }          // a long run: summed by segments_long
        int32_t r = int32_t((k >> (cb + 1)) & node_mask);
        int32_t c = int32_t((k >> 1) & node_mask);
        int64_t idx = int64_t(first_slot[u]) +
                      int64_t(__builtin_amdgcn_mbcnt_hi(uint32_t(word >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(word), 0u)));
        if (diag) idx += int64_t(r - row_lo) + (c > r ? 1 : 0);
        if (idx >= nnz) continue;                       // caller's arrays are shorter than the count phase reported
        float a = (d_row[u] * sv) * d_col[u];           // (D A_s) D, hermitian.py:154
        float theta = two_pi_q * dv;                    // fp32 product, hermitian.py:157
        float sn, cs;
        sincosf(theta, &sn, &cs);                       // full-range accurate sincos, hermitian.py:158
        float re = -(cs * a), im = -(sn * a);           // I - Theta.*A_s off the diagonal, hermitian.py:164
        col_idx[idx] = c;
        val[idx] = make_float2(re * scale, im * scale); // (2/lmax) L - I, hermitian.py:167
    }
}

// diagonal (2/lmax - 1) of every row of the block; its slot follows the row's kept entries with col < r
__global__ void __launch_bounds__(kBlock) emit_diagonal(const uint64_t* __restrict__ keys, const uint64_t* __restrict__ bits,
                                                        const uint32_t* __restrict__ word_pos,
                                                        int64_t m, int32_t row_lo, int32_t row_hi, int cb, float diag_val,
                                                        int64_t nnz, int32_t* __restrict__ col_idx, float2* __restrict__ val) {
    int64_t r = int64_t(row_lo) + int64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (r >= row_hi) return;
    uint64_t target = (uint64_t(r) << (cb + 1)) | (uint64_t(r) << 1);
    const uint64_t live = (1ull << (2 * cb + 1)) - 1ull;   // (row, col, dir): tagged keys carry a count above these bits
    int64_t lo = 0, hi = m;                             // lower_bound(keys, target)
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if ((keys[mid] & live) < target) lo = mid + 1; else hi = mid;
    }
    int64_t idx = int64_t(slot_of(bits, word_pos, lo)) + (r - row_lo);
    if (idx >= nnz) return;
    col_idx[idx] = int32_t(r);
    val[idx] = make_float2(diag_val, 0.0f);
}

inline unsigned grid_for(int64_t n) { return unsigned((n + kBlock - 1) / kBlock); }

struct Work {            // the workspace cut up
    Header* hdr; uint64_t *keys0, *keys1; float *w0, *w1; uint64_t* bits; uint32_t* word_pos; int32_t* rec_ptr; float* dinv;
    uint32_t* tiles; uint32_t* long_list; float* tag_table; char* sort;
};

Work cut(void* workspace, const Layout& L, bool weighted) {
    char* ws = static_cast<char*>(workspace);
    Work w{};
    w.hdr = reinterpret_cast<Header*>(ws + L.off_hdr);
    w.keys0 = reinterpret_cast<uint64_t*>(ws + L.off_keys0);
    w.keys1 = reinterpret_cast<uint64_t*>(ws + L.off_keys1);
    w.w0 = weighted ? reinterpret_cast<float*>(ws + L.off_w0) : nullptr;
    w.w1 = weighted ? reinterpret_cast<float*>(ws + L.off_w1) : nullptr;
    w.bits = reinterpret_cast<uint64_t*>(ws + L.off_bits);
    w.word_pos = reinterpret_cast<uint32_t*>(ws + L.off_wpos);
    w.rec_ptr = reinterpret_cast<int32_t*>(ws + L.off_recptr);
    w.dinv = reinterpret_cast<float*>(ws + L.off_dinv);
    w.tiles = reinterpret_cast<uint32_t*>(ws + L.off_tiles);
    w.long_list = reinterpret_cast<uint32_t*>(ws + L.off_long);
    w.tag_table = reinterpret_cast<float*>(ws + L.off_tagtab);
    w.sort = ws + L.off_sort;
    return w;
}

struct EdgeList { const void* row; const void* col; const float* w; int64_t n; int64_t n_self; };

// The sort key is (row, col) = key bits [1, 2cb+1).  The direction bit 0 is NOT sorted: all forward
// records come before all backward records in the record order and the sort is stable, so inside a (row, col)
// segment the forward records still come first, each side in input order.
// The records are made by the sort's first pass (mhsort::RecordSource).  An odd number of passes moves them
// "first buffer" -> keys1 when the pass that makes them writes into the buffer chosen here, so that the sorted
// records always end in keys1 / w1.
// `node_counts` (tagged sorts only): the record count of every node when the records at hand are not all records of the
// graph (a row block) - the column groups then say nothing about a node's degree and the caller's table is used.
int records_and_sort(const Layout& L, const Work& W, const EdgeList& fwd, const EdgeList& bwd, int idx64, bool weighted,
                     bool tagged, int32_t n_nodes, int32_t row_lo, int32_t row_hi, hipStream_t stream,
                     const int32_t* node_counts = nullptr) {
    if (L.m <= 0) return MH_OK;
    const bool start_in_0 = mhsort::passes_for(2 * L.cb) & 1;
    uint64_t* rec_keys = start_in_0 ? W.keys0 : W.keys1;          // the buffer the first pass does NOT write
    float* rec_w = weighted ? (start_in_0 ? W.w0 : W.w1) : nullptr;
    uint64_t* other = rec_keys == W.keys0 ? W.keys1 : W.keys0;
    float* other_w = weighted ? (rec_w == W.w0 ? W.w1 : W.w0) : nullptr;
    mhsort::RecordSource src{};
    src.row[0] = fwd.row; src.col[0] = fwd.col; src.w[0] = fwd.w; src.n_in[0] = fwd.n; src.n_self[0] = fwd.n_self;
    src.row[1] = bwd.row; src.col[1] = bwd.col; src.w[1] = bwd.w; src.n_in[1] = bwd.n; src.n_self[1] = bwd.n_self;
    src.idx64 = idx64; src.cb = L.cb; src.n_nodes = n_nodes; src.row_lo = row_lo; src.row_hi = row_hi;
    src.err = &W.hdr->err;
    if (!tagged) {
        MH_HIP(mhsort::sort(rec_keys, other, rec_w, other_w, L.m, 1, 2 * L.cb, W.sort, L.sort_bytes, stream, nullptr, &src));
    } else {
        // column passes; group sizes of the columns; row passes, the first of which writes the sizes into the keys
        const uint64_t node_mask = (1ull << L.cb) - 1ull;
        MH_HIP(mhsort::sort(rec_keys, other, nullptr, nullptr, L.m, 1, L.cb, W.sort, L.sort_bytes, stream, nullptr, &src));
        uint64_t* by_col = (mhsort::passes_for(L.cb) & 1) ? other : rec_keys;
        uint64_t* spare = by_col == rec_keys ? other : rec_keys;
        if (!node_counts) {
            row_starts<<<grid_for(int64_t(n_nodes) + 1), kBlock, 0, stream>>>(by_col, L.m, n_nodes, 1, node_mask, W.rec_ptr);
            MH_LAUNCH_CHECK("column_starts");
        }
        const mhsort::Tag tag{W.rec_ptr, node_counts, 1, uint32_t(node_mask), 2 * L.cb + 1, kTagCap};
        MH_HIP(mhsort::sort(by_col, spare, nullptr, nullptr, L.m, L.cb + 1, L.cb, W.sort, L.sort_bytes, stream, &tag));
    }
    MH_LAUNCH_CHECK("sort");
    return MH_OK;
}

// everything of the count phase after the sort; rows [row_lo, row_hi) are the rows the records belong to
int count_sorted(const Layout& L, const Work& W, bool weighted, bool have_row_starts, int32_t n_nodes, int32_t row_lo, int32_t row_hi,
                 int diag, int32_t* row_ptr, float* dinv_block, int64_t* nnz_host, hipStream_t stream) {
    // A tagged whole-graph build has the row starts already: they are the column starts found between the sort's column
    // and row passes (node j has as many records in row j as in column j, so both are the same running sum).
    if (!have_row_starts) {
        row_starts<<<grid_for(int64_t(n_nodes) + 1), kBlock, 0, stream>>>(W.keys1, L.m, n_nodes, L.cb + 1, (1ull << L.cb) - 1ull,
                                                                          W.rec_ptr);
        MH_LAUNCH_CHECK("row_starts");
    }
    if (n_nodes > 0) {
        if (weighted)
            degree_inv_sqrt_weighted<<<grid_for(int64_t(n_nodes) * 16), kBlock, 0, stream>>>(W.rec_ptr, W.w1, n_nodes, W.dinv);
        else
            degree_inv_sqrt_counts<<<grid_for(n_nodes), kBlock, 0, stream>>>(W.rec_ptr, n_nodes, W.dinv);
        MH_LAUNCH_CHECK("degree_inv_sqrt");
    }
    float* seg_s = reinterpret_cast<float*>(W.keys0);           // keys0 is free after the sort
    float* seg_d = seg_s + L.m_pad;
    if (L.m > 0) {
        segments<<<grid_for((L.m + kGroups - 1) / kGroups), kBlock, 0, stream>>>(W.keys1, W.w1, L.m, W.bits, W.long_list, W.hdr);
        MH_LAUNCH_CHECK("segments");
        segments_long<<<1024, kBlock, 0, stream>>>(W.keys1, W.w1, L.m, seg_s, seg_d, W.bits, W.long_list, W.hdr);
        MH_LAUNCH_CHECK("segments_long");
    }
    word_tile_sums<<<unsigned(L.n_tiles), kBlock, 0, stream>>>(W.bits, W.tiles);
    MH_LAUNCH_CHECK("word_tile_sums");
    scan_tiles<<<1, 1024, 0, stream>>>(W.tiles, L.n_tiles, &W.hdr->kept);
    MH_LAUNCH_CHECK("scan_tiles");
    word_tile_write<<<unsigned(L.n_tiles), kBlock, 0, stream>>>(W.bits, W.tiles, W.word_pos);
    MH_LAUNCH_CHECK("word_tile_write");
    write_row_ptr<<<grid_for(int64_t(row_hi - row_lo) + 1), kBlock, 0, stream>>>(W.rec_ptr, W.bits, W.word_pos, row_lo, row_hi, diag,
                                                                                 row_ptr);
    MH_LAUNCH_CHECK("write_row_ptr");
    if (dinv_block && row_hi > row_lo)
        MH_HIP(hipMemcpyAsync(dinv_block, W.dinv + row_lo, size_t(row_hi - row_lo) * 4, hipMemcpyDeviceToDevice, stream));
    Header h{};
    MH_HIP(hipMemcpyAsync(&h, W.hdr, sizeof(Header), hipMemcpyDeviceToHost, stream));
    MH_HIP(hipStreamSynchronize(stream));
    if (h.err & 1) return mh::fail(MH_ERR_INDEX, "an edge endpoint is outside [0, %d)", n_nodes);
    if (h.err & 2) return mh::fail(MH_ERR_INDEX, "an edge does not belong to the row block [%d, %d)", row_lo, row_hi);
    *nnz_host = h.kept + (diag ? int64_t(row_hi - row_lo) : 0);
    return MH_OK;
}

int fill_sorted(const Layout& L, const Work& W, bool tagged, int32_t row_lo, int32_t row_hi, double q, double max_eigen,
                int64_t nnz, const float* dinv, int32_t* col_idx, float* val, hipStream_t stream) {
    const float* seg_s = reinterpret_cast<const float*>(W.keys0);
    const float* seg_d = seg_s + L.m_pad;
    const float scale = float(2.0 / max_eigen);                 // the double quotient rounded once (hermitian.py:167)
    const float diag_val = scale * 1.0f - 1.0f;
    const int diag = diag_val != 0.0f;
    // 2*pi*q as the reference forms it: a double product rounded to fp32 when it meets the
    // float32 matrix (complex64 scalar), hermitian.py:157
    const float two_pi_q = float(2.0 * 3.14159265358979323846 * q);
    if (L.m > 0 && nnz > 0) {
        MH_REQUIRE(col_idx && val, "col_idx/val is null");
        MH_REQUIRE((reinterpret_cast<uintptr_t>(val) & 7u) == 0, "val must be 8-byte aligned");
        if (tagged) {
            tag_table_kernel<<<grid_for(kTagCap), kBlock, 0, stream>>>(W.tag_table);
            MH_LAUNCH_CHECK("tag_table");
        }
        emit_entries<<<grid_for((L.m + kGroups - 1) / kGroups), kBlock, 0, stream>>>(W.keys1, W.w1, W.bits, W.word_pos, seg_s, seg_d, dinv, W.tag_table, L.m, L.cb, row_lo,
                                                           two_pi_q, scale, diag, nnz, tagged ? 1 : 0, col_idx,
                                                           reinterpret_cast<float2*>(val));
        MH_LAUNCH_CHECK("emit_entries");
    }
    if (diag && row_hi > row_lo) {
        MH_REQUIRE(col_idx && val && nnz >= row_hi - row_lo, "col_idx/val is null or nnz too small for the diagonal");
        emit_diagonal<<<grid_for(row_hi - row_lo), kBlock, 0, stream>>>(W.keys1, W.bits, W.word_pos, L.m, row_lo, row_hi, L.cb,
                                                                        diag_val, nnz, col_idx, reinterpret_cast<float2*>(val));
        MH_LAUNCH_CHECK("emit_diagonal");
    }
    return MH_OK;
}

int prepare(const Layout& L, void* workspace, size_t workspace_bytes, const Work& W, int64_t n_rows_out, hipStream_t stream) {
    if (L.m >= (int64_t(1) << 31) - 1 - n_rows_out) return mh::fail(MH_ERR_RANGE, "2*edges does not fit int32 CSR indices");
    if (workspace_bytes < L.total)
        return mh::fail(MH_ERR_WORKSPACE, "workspace %zu < required %zu bytes", workspace_bytes, L.total);
    MH_REQUIRE(mh_aligned16(workspace), "workspace must be 16-byte aligned");
    MH_HIP(hipMemsetAsync(W.hdr, 0, 256, stream));
    MH_HIP(hipMemsetAsync(W.bits, 0, size_t(L.n_words) * 8, stream));
    return MH_OK;
}

}  // namespace

extern "C" size_t mh_lap_workspace_bytes(int64_t n_edges, int32_t n_nodes, int weighted, int gcn_appr) {
    if (n_edges < 0 || n_nodes < 0) return 0;
    return make_layout(full_records(n_edges, n_nodes, gcn_appr != 0), n_nodes, weighted != 0).total;
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
    Layout L = make_layout(full_records(n_edges, n_nodes, gcn), n_nodes, weighted);
    Work W = cut(workspace, L, weighted);
    int rc = prepare(L, workspace, workspace_bytes, W, n_nodes, stream);
    if (rc != MH_OK) return rc;
    const int diag = (float(2.0 / max_eigen) * 1.0f - 1.0f) != 0.0f;
    EdgeList e{row, col, weight, n_edges, gcn ? int64_t(n_nodes) : 0};
    rc = records_and_sort(L, W, e, e, idx64, weighted, use_tags(L, weighted, true), n_nodes, 0, n_nodes, stream);
    if (rc != MH_OK) return rc;
    return count_sorted(L, W, weighted, use_tags(L, weighted, true), n_nodes, 0, n_nodes, diag, row_ptr, nullptr, nnz_host, stream);
}

extern "C" int mh_lap_build_fill(int64_t n_edges, int32_t n_nodes, double q, double max_eigen, int weighted_,
                                 int gcn_appr, void* workspace, size_t workspace_bytes, int64_t nnz,
                                 int32_t* col_idx, float* val, float* dinv_out, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    MH_REQUIRE(nnz >= 0, "negative nnz");
    MH_REQUIRE(n_edges >= 0 && n_nodes >= 0 && workspace, "bad size or null workspace");
    MH_REQUIRE(max_eigen > 0.0, "max_eigen must be positive");
    const bool weighted = weighted_ != 0;
    Layout L = make_layout(full_records(n_edges, n_nodes, gcn_appr != 0), n_nodes, weighted);
    if (workspace_bytes < L.total)
        return mh::fail(MH_ERR_WORKSPACE, "workspace %zu < required %zu bytes", workspace_bytes, L.total);
    Work W = cut(workspace, L, weighted);
    int rc = fill_sorted(L, W, use_tags(L, weighted, true), 0, n_nodes, q, max_eigen, nnz, W.dinv, col_idx, val, stream);
    if (rc != MH_OK) return rc;
    if (dinv_out && n_nodes > 0)
        MH_HIP(hipMemcpyAsync(dinv_out, W.dinv, size_t(n_nodes) * 4, hipMemcpyDeviceToDevice, stream));
    return MH_OK;
}

// ---- one rank's row block of the Laplacian ---------------------------------------------------------
extern "C" size_t mh_lap_block_workspace_bytes(int64_t n_fwd, int64_t n_bwd, int32_t n_nodes, int weighted) {
    if (n_fwd < 0 || n_bwd < 0 || n_nodes < 0) return 0;
    return make_layout(n_fwd + n_bwd, n_nodes, weighted != 0).total;
}

extern "C" int mh_lap_block_count(const void* fwd_row, const void* fwd_col, const float* fwd_w, int64_t n_fwd,
                                  const void* bwd_row, const void* bwd_col, const float* bwd_w, int64_t n_bwd, int idx64,
                                  int weighted_, int32_t n_nodes, int32_t row_lo, int32_t row_hi, double max_eigen,
                                  void* workspace, size_t workspace_bytes, int32_t* row_ptr_block, float* dinv_block,
                                  int64_t* nnz_host, void* stream_) {
    return mh_lap_block_count_tagged(fwd_row, fwd_col, fwd_w, n_fwd, bwd_row, bwd_col, bwd_w, n_bwd, idx64, weighted_, n_nodes,
                                     row_lo, row_hi, max_eigen, nullptr, workspace, workspace_bytes, row_ptr_block, dinv_block,
                                     nnz_host, stream_);
}

extern "C" int mh_lap_block_count_tagged(const void* fwd_row, const void* fwd_col, const float* fwd_w, int64_t n_fwd,
                                         const void* bwd_row, const void* bwd_col, const float* bwd_w, int64_t n_bwd, int idx64,
                                         int weighted_, int32_t n_nodes, int32_t row_lo, int32_t row_hi, double max_eigen,
                                         const int32_t* node_records, void* workspace, size_t workspace_bytes,
                                         int32_t* row_ptr_block, float* dinv_block, int64_t* nnz_host, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    MH_REQUIRE(n_fwd >= 0 && n_bwd >= 0 && n_nodes >= 0, "negative size");
    MH_REQUIRE(0 <= row_lo && row_lo <= row_hi && row_hi <= n_nodes, "row block [%d, %d) outside [0, %d]", row_lo, row_hi, n_nodes);
    MH_REQUIRE((n_fwd == 0 || (fwd_row && fwd_col)) && (n_bwd == 0 || (bwd_row && bwd_col)), "row/col is null");
    MH_REQUIRE(workspace && row_ptr_block && nnz_host, "workspace/row_ptr_block/nnz_host is null");
    MH_REQUIRE(max_eigen > 0.0, "max_eigen must be positive");
    const bool weighted = weighted_ != 0;
    MH_REQUIRE(weighted || (!fwd_w && !bwd_w), "weights passed to an unweighted build");
    Layout L = make_layout(n_fwd + n_bwd, n_nodes, weighted);
    Work W = cut(workspace, L, weighted);
    int rc = prepare(L, workspace, workspace_bytes, W, row_hi - row_lo, stream);
    if (rc != MH_OK) return rc;
    const int diag = (float(2.0 / max_eigen) * 1.0f - 1.0f) != 0.0f;
    EdgeList f{fwd_row, fwd_col, fwd_w, n_fwd, 0}, b{bwd_row, bwd_col, bwd_w, n_bwd, 0};
    // the columns' record counts travel in the keys when the caller knows them (the block's own records do not: a column's
    // other records lie in other blocks); same conditions as for the whole graph.  The rows of the block are complete,
    // so their starts and degrees come from the block's records as ever.
    const bool tagged = node_records != nullptr && use_tags(L, weighted, true);
    rc = records_and_sort(L, W, f, b, idx64, weighted, tagged, n_nodes, row_lo, row_hi, stream, node_records);
    if (rc != MH_OK) return rc;
    return count_sorted(L, W, weighted, false, n_nodes, row_lo, row_hi, diag, row_ptr_block, dinv_block, nnz_host, stream);
}

extern "C" int mh_lap_block_fill(int64_t n_fwd, int64_t n_bwd, int32_t n_nodes, int32_t row_lo, int32_t row_hi, double q,
                                 double max_eigen, int weighted_, void* workspace, size_t workspace_bytes, int64_t nnz,
                                 const float* dinv_full, int32_t* col_idx, float* val, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    MH_REQUIRE(nnz >= 0 && n_fwd >= 0 && n_bwd >= 0 && n_nodes >= 0 && workspace, "bad size or null workspace");
    MH_REQUIRE(0 <= row_lo && row_lo <= row_hi && row_hi <= n_nodes, "row block [%d, %d) outside [0, %d]", row_lo, row_hi, n_nodes);
    MH_REQUIRE(max_eigen > 0.0, "max_eigen must be positive");
    MH_REQUIRE(dinv_full || n_nodes == 0, "dinv_full is null");
    const bool weighted = weighted_ != 0;
    Layout L = make_layout(n_fwd + n_bwd, n_nodes, weighted);
    if (workspace_bytes < L.total)
        return mh::fail(MH_ERR_WORKSPACE, "workspace %zu < required %zu bytes", workspace_bytes, L.total);
    Work W = cut(workspace, L, weighted);
    // keys that mh_lap_block_count_tagged tagged carry a non-zero count above their live bits; untagged ones carry zero
    return fill_sorted(L, W, use_tags(L, weighted, true), row_lo, row_hi, q, max_eigen, nnz, dinv_full, col_idx, val, stream);
}
