// Stable LSD radix sort, at most 8 bits per pass (the key bits are spread evenly over the passes), for the (row, col, dir) edge records of the Laplacian
// builder on gfx950.
//
// Per pass, three steps over tiles of 8192 keys (512 threads = 8 wavefronts x 16 rounds x 64 lanes):
//   1. tile_histogram   256-bin digit histogram of every tile (LDS atomics), stored tile-major so
//                       that a tile's 256 counters are one contiguous 1 KiB line;
//   2. scan_apply       column-wise exclusive scan of the [tiles][256] matrix in digit-major order: the histogram
//                       kernel also adds each tile's counts to its chunk's totals (256 tiles per chunk); one workgroup
//                       per chunk sums the totals before it, scans the 256 digit totals and turns its tiles'
//                       counters into global offsets;
//   3. scatter          each wavefront ranks its 1024 keys round by round with ballot "match" masks
//                       (8 ballots give the lanes holding the same digit; the population count below
//                       a lane is its rank), waves are offset against each other per digit, the tile is
//                       laid out digit-sorted in LDS and written to global memory in runs of equal
//                       digit.  Memory order = (wave, round, lane) = index order, so the sort is stable.
// 2 reads + 1 write of the keys per pass; no inter-workgroup communication inside a launch.
#include "radix_sort.h"

namespace mhsort {
namespace {

constexpr int kThreads = 512;
constexpr int kWaves = kThreads / 64;
constexpr int kItems = 16;
constexpr int kTile = kThreads * kItems;     // 8192 keys
constexpr int kBins = 256;
constexpr int kChunk = 256;                  // tiles per scan chunk
constexpr int kXcds = 8;                     // accelerator dies of the device; workgroup i runs on die i % 8

__device__ inline uint32_t digit_of(uint64_t k, int shift, uint32_t mask) { return uint32_t(k >> shift) & mask; }

// Record p of a RecordSource (see radix_sort.h); FLAG: report bad endpoints / rows in *err (done by one of the two
// kernels that visit every record).
template <typename IdxT, bool FLAG>
__device__ inline uint64_t record_key(const RecordSource& g, int64_t p, float& w) {
    const int64_t cnt0 = g.n_in[0] + g.n_self[0];
    const int dir = p >= cnt0 ? 1 : 0;
    const int64_t e = p - (dir ? cnt0 : 0);
    int64_t i, j;
    w = 1.0f;
    if (e < g.n_in[dir]) {
        i = int64_t(static_cast<const IdxT*>(g.row[dir])[e]);
        j = int64_t(static_cast<const IdxT*>(g.col[dir])[e]);
        if (g.w[dir]) w = g.w[dir][e];
    } else {
        i = j = e - g.n_in[dir];
    }
    if (i < 0 || j < 0 || i >= g.n_nodes || j >= g.n_nodes) {
        if (FLAG) atomicOr(g.err, 1);      // flag only; the record is made harmless
        i = j = g.row_lo; w = 0.0f;
    }
    const int64_t r = dir ? j : i, c = dir ? i : j;
    if (FLAG && (r < g.row_lo || r >= g.row_hi)) atomicOr(g.err, 2);
    return (uint64_t(r) << (g.cb + 1)) | (uint64_t(c) << 1) | uint64_t(dir);
}

// The same records for a stretch that lies wholly inside list `dir`'s edges (no self-loop entries, one direction): the
// common case of all but a few tiles.  `e0` is the list entry of this lane's first record; the ITEMS records of a
// lane are 64 entries apart.  Same keys, weights and error flags as record_key, without its per-record case analysis.
template <typename IdxT, bool FLAG, int ITEMS, bool WEIGHTS>
__device__ inline void record_keys_inside(const RecordSource& g, int dir, int64_t e0, uint64_t (&key)[ITEMS], float* w) {
    const IdxT* __restrict__ rp = static_cast<const IdxT*>(g.row[dir]) + e0;
    const IdxT* __restrict__ cp = static_cast<const IdxT*>(g.col[dir]) + e0;
    const float* __restrict__ wp = g.w[dir] ? g.w[dir] + e0 : nullptr;
    IdxT a[ITEMS], b[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) { a[i] = rp[i * 64]; b[i] = cp[i * 64]; }
    const bool check_rows = g.row_lo != 0 || g.row_hi != g.n_nodes;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        int64_t ii = int64_t(a[i]), jj = int64_t(b[i]);
        float wv = 1.0f;
        if (WEIGHTS && wp) wv = wp[i * 64];
        if (uint64_t(ii) >= uint64_t(g.n_nodes) || uint64_t(jj) >= uint64_t(g.n_nodes)) {
            if (FLAG) atomicOr(g.err, 1);
            ii = jj = g.row_lo; wv = 0.0f;
        }
        const int64_t r = dir ? jj : ii, c = dir ? ii : jj;
        if (FLAG && check_rows && (r < g.row_lo || r >= g.row_hi)) atomicOr(g.err, 2);
        key[i] = (uint64_t(r) << (g.cb + 1)) | (uint64_t(c) << 1) | uint64_t(dir);
        if (WEIGHTS) w[i] = wv;
    }
}

// which list a stretch of `len` records starting at record p0 lies wholly inside (its edges, not its self-loop entries):
// 0, 1, or -1 for neither
__device__ inline int list_of_stretch(const RecordSource& g, int64_t p0, int64_t len, int64_t& e_first) {
    const int64_t cnt0 = g.n_in[0] + g.n_self[0];
    if (p0 + len <= g.n_in[0]) { e_first = p0; return 0; }
    if (p0 >= cnt0 && p0 + len <= cnt0 + g.n_in[1]) { e_first = p0 - cnt0; return 1; }
    return -1;
}

// GEN: 0 = keys in memory, 1 / 2 = records made from int32 / int64 edge lists
template <int GEN>
__device__ inline uint64_t key_at(const uint64_t* __restrict__ keys, const RecordSource& g, int64_t idx) {
    float w;
    if (GEN == 1) return record_key<int32_t, false>(g, idx, w);
    if (GEN == 2) return record_key<int64_t, false>(g, idx, w);
    return keys[idx];
}

template <int GEN>
__global__ void __launch_bounds__(kThreads) tile_histogram(const uint64_t* __restrict__ keys, int64_t n, int shift,
                                                           uint32_t mask, uint32_t* __restrict__ hist,
                                                           uint32_t* __restrict__ chunk_sum, const RecordSource gen) {
    __shared__ uint32_t bins[kBins];
    if (threadIdx.x < kBins) bins[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = int64_t(blockIdx.x) * kTile;
    if (GEN == 0 && base + kTile <= n && (reinterpret_cast<uintptr_t>(keys) & 15u) == 0) {
        // a full tile of keys in memory: 16 bytes per lane and load, all loads issued before the first count
        // (a histogram does not care which lane sees which key)
        const ulonglong2* __restrict__ pairs = reinterpret_cast<const ulonglong2*>(keys + base);
        ulonglong2 k[kItems / 2];
#pragma unroll
        for (int i = 0; i < kItems / 2; ++i) k[i] = pairs[i * kThreads + threadIdx.x];
#pragma unroll
        for (int i = 0; i < kItems / 2; ++i) {
            atomicAdd(&bins[digit_of(k[i].x, shift, mask)], 1u);
            atomicAdd(&bins[digit_of(k[i].y, shift, mask)], 1u);
        }
    } else {
        int64_t e_first = 0;
        const int inside = (GEN != 0 && base + kTile <= n) ? list_of_stretch(gen, base, kTile, e_first) : -1;
        if (GEN != 0 && inside >= 0) {
            // a full tile inside one edge list (a histogram does not care which lane sees which record)
            uint64_t k[kItems];
            const int64_t e0 = e_first + int64_t(threadIdx.x >> 6) * (64 * kItems) + (threadIdx.x & 63);
            if (GEN == 1) record_keys_inside<int32_t, false, kItems, false>(gen, inside, e0, k, nullptr);
            else record_keys_inside<int64_t, false, kItems, false>(gen, inside, e0, k, nullptr);
#pragma unroll
            for (int i = 0; i < kItems; ++i) atomicAdd(&bins[digit_of(k[i], shift, mask)], 1u);
        } else {
#pragma unroll
            for (int i = 0; i < kItems; ++i) {
                int64_t idx = base + i * kThreads + threadIdx.x;
                if (idx < n) atomicAdd(&bins[digit_of(key_at<GEN>(keys, gen, idx), shift, mask)], 1u);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < kBins) {
        const uint32_t c = bins[threadIdx.x];
        hist[int64_t(blockIdx.x) * kBins + threadIdx.x] = c;
        // per digit, the sum over the chunk's tiles (integer adds: the order does not matter)
        if (c) atomicAdd(&chunk_sum[int64_t(blockIdx.x / kChunk) * kBins + threadIdx.x], c);
    }
}

// hist[t][d] (counts) -> global offset of (digit d, tile t), digit-major: all of digit 0, then digit 1, ...  One workgroup
// per chunk of 256 tiles, one thread per digit.  chunk_sum[c][d] holds the chunks' totals (tile_histogram): every workgroup
// adds up the chunks before its own and all chunks (the digit's total), scans the 256 totals, and walks its tiles.
__global__ void __launch_bounds__(kBins) scan_apply(uint32_t* __restrict__ hist, int64_t n_tiles,
                                                    const uint32_t* __restrict__ chunk_sum, int64_t n_chunks) {
    __shared__ uint32_t wave_total[kBins / 64];
    const int d = threadIdx.x, lane = d & 63, w = d >> 6;
    uint32_t before = 0, total = 0;
    // every load below is independent of the one before it; they are issued kBatch at a time, so the kernel spends
    // (chunks + tiles per chunk) / kBatch memory latencies instead of one per load
    constexpr int kBatch = 32;
    for (int64_t c0 = 0; c0 < n_chunks; c0 += kBatch) {
        uint32_t v[kBatch];
#pragma unroll
        for (int j = 0; j < kBatch; ++j) v[j] = c0 + j < n_chunks ? chunk_sum[(c0 + j) * kBins + d] : 0u;
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            before += c0 + j < int64_t(blockIdx.x) ? v[j] : 0u;
            total += v[j];
        }
    }
    uint32_t inc = total;                               // exclusive scan of the digit totals
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(inc, off, 64);
        if (lane >= off) inc += t;
    }
    if (lane == 63) wave_total[w] = inc;
    __syncthreads();
    uint32_t digit_base = inc - total;
    for (int ww = 0; ww < kBins / 64; ++ww) digit_base += ww < w ? wave_total[ww] : 0u;
    const int64_t t0 = int64_t(blockIdx.x) * kChunk;
    const int64_t t1 = t0 + kChunk < n_tiles ? t0 + kChunk : n_tiles;
    uint32_t run = digit_base + before;
    for (int64_t t = t0; t < t1; t += kBatch) {
        uint32_t v[kBatch];
#pragma unroll
        for (int j = 0; j < kBatch; ++j) v[j] = t + j < t1 ? hist[(t + j) * kBins + d] : 0u;
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            if (t + j < t1) hist[(t + j) * kBins + d] = run;
            run += v[j];
        }
    }
}

// digit of a key given as its two halves; HI: the digit lies in the upper half (shift >= 32)
template <bool HI>
__device__ inline uint32_t digit_of2(uint32_t lo, uint32_t hi, int shift, uint32_t mask) {
    const uint32_t v = HI ? hi >> (shift - 32) : __builtin_amdgcn_alignbit(hi, lo, uint32_t(shift));
    return v & mask;
}

// (plo, phi) &= the lanes of the wavefront whose digit `d` equals this lane's in its low NB bits (NB = 7 or 8).
// Per bit b: t = all ones when the lane's bit is set (v_bfe_i32), the ballot of that bit (v_cmp into a scalar register
// pair), and per 32-bit half p &= ~(ballot ^ t) as one three-input bit operation (table 0x90 over (p, ballot, t)):
// 4 vector instructions per bit.  Written out by hand: left to the compiler the same source took 9 per bit plus wait
// states after every compare.  A compare's scalar result is read no sooner than three instructions later (two wait
// states are required between a vector compare and a vector read of its scalar result).
template <int NB>
__device__ inline void match_lanes(uint32_t d, uint32_t& plo, uint32_t& phi) {
    uint32_t t0, t1, t2, t3, t4, t5, t6, t7;
    if (NB == 8) {
        asm volatile(
            "v_bfe_i32 %[t0], %[d], 0, 1\n\t"
            "v_bfe_i32 %[t1], %[d], 1, 1\n\t"
            "v_bfe_i32 %[t2], %[d], 2, 1\n\t"
            "v_bfe_i32 %[t3], %[d], 3, 1\n\t"
            "v_cmp_ne_u32_e64 s[80:81], 0, %[t0]\n\t"
            "v_cmp_ne_u32_e64 s[82:83], 0, %[t1]\n\t"
            "v_cmp_ne_u32_e64 s[84:85], 0, %[t2]\n\t"
            "v_cmp_ne_u32_e64 s[86:87], 0, %[t3]\n\t"
            "v_bfe_i32 %[t4], %[d], 4, 1\n\t"
            "v_bfe_i32 %[t5], %[d], 5, 1\n\t"
            "v_bfe_i32 %[t6], %[d], 6, 1\n\t"
            "v_bfe_i32 %[t7], %[d], 7, 1\n\t"
            "v_cmp_ne_u32_e64 s[88:89], 0, %[t4]\n\t"
            "v_cmp_ne_u32_e64 s[90:91], 0, %[t5]\n\t"
            "v_cmp_ne_u32_e64 s[92:93], 0, %[t6]\n\t"
            "v_cmp_ne_u32_e64 s[94:95], 0, %[t7]\n\t"
            "v_bitop3_b32 %[plo], %[plo], s80, %[t0] bitop3:0x90\n\t"
            "v_bitop3_b32 %[phi], %[phi], s81, %[t0] bitop3:0x90\n\t"
            "v_bitop3_b32 %[plo], %[plo], s82, %[t1] bitop3:0x90\n\t"
            "v_bitop3_b32 %[phi], %[phi], s83, %[t1] bitop3:0x90\n\t"
            "v_bitop3_b32 %[plo], %[plo], s84, %[t2] bitop3:0x90\n\t"
            "v_bitop3_b32 %[phi], %[phi], s85, %[t2] bitop3:0x90\n\t"
            "v_bitop3_b32 %[plo], %[plo], s86, %[t3] bitop3:0x90\n\t"
            "v_bitop3_b32 %[phi], %[phi], s87, %[t3] bitop3:0x90\n\t"
            "v_bitop3_b32 %[plo], %[plo], s88, %[t4] bitop3:0x90\n\t"
            "v_bitop3_b32 %[phi], %[phi], s89, %[t4] bitop3:0x90\n\t"
            "v_bitop3_b32 %[plo], %[plo], s90, %[t5] bitop3:0x90\n\t"
            "v_bitop3_b32 %[phi], %[phi], s91, %[t5] bitop3:0x90\n\t"
            "v_bitop3_b32 %[plo], %[plo], s92, %[t6] bitop3:0x90\n\t"
            "v_bitop3_b32 %[phi], %[phi], s93, %[t6] bitop3:0x90\n\t"
            "v_bitop3_b32 %[plo], %[plo], s94, %[t7] bitop3:0x90\n\t"
            "v_bitop3_b32 %[phi], %[phi], s95, %[t7] bitop3:0x90\n\t"
            : [plo] "+v"(plo), [phi] "+v"(phi), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3),
              [t4] "=&v"(t4), [t5] "=&v"(t5), [t6] "=&v"(t6), [t7] "=&v"(t7)
            : [d] "v"(d)
            : "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95");
    } else {
        asm volatile(
            "v_bfe_i32 %[t0], %[d], 0, 1\n\t"
            "v_bfe_i32 %[t1], %[d], 1, 1\n\t"
            "v_bfe_i32 %[t2], %[d], 2, 1\n\t"
            "v_bfe_i32 %[t3], %[d], 3, 1\n\t"
            "v_cmp_ne_u32_e64 s[80:81], 0, %[t0]\n\t"
            "v_cmp_ne_u32_e64 s[82:83], 0, %[t1]\n\t"
            "v_cmp_ne_u32_e64 s[84:85], 0, %[t2]\n\t"
            "v_cmp_ne_u32_e64 s[86:87], 0, %[t3]\n\t"
            "v_bfe_i32 %[t4], %[d], 4, 1\n\t"
            "v_bfe_i32 %[t5], %[d], 5, 1\n\t"
            "v_bfe_i32 %[t6], %[d], 6, 1\n\t"
            "v_cmp_ne_u32_e64 s[88:89], 0, %[t4]\n\t"
            "v_cmp_ne_u32_e64 s[90:91], 0, %[t5]\n\t"
            "v_cmp_ne_u32_e64 s[92:93], 0, %[t6]\n\t"
            "v_bitop3_b32 %[plo], %[plo], s80, %[t0] bitop3:0x90\n\t"
            "v_bitop3_b32 %[phi], %[phi], s81, %[t0] bitop3:0x90\n\t"
            "v_bitop3_b32 %[plo], %[plo], s82, %[t1] bitop3:0x90\n\t"
            "v_bitop3_b32 %[phi], %[phi], s83, %[t1] bitop3:0x90\n\t"
            "v_bitop3_b32 %[plo], %[plo], s84, %[t2] bitop3:0x90\n\t"
            "v_bitop3_b32 %[phi], %[phi], s85, %[t2] bitop3:0x90\n\t"
            "v_bitop3_b32 %[plo], %[plo], s86, %[t3] bitop3:0x90\n\t"
            "v_bitop3_b32 %[phi], %[phi], s87, %[t3] bitop3:0x90\n\t"
            "v_bitop3_b32 %[plo], %[plo], s88, %[t4] bitop3:0x90\n\t"
            "v_bitop3_b32 %[phi], %[phi], s89, %[t4] bitop3:0x90\n\t"
            "v_bitop3_b32 %[plo], %[plo], s90, %[t5] bitop3:0x90\n\t"
            "v_bitop3_b32 %[phi], %[phi], s91, %[t5] bitop3:0x90\n\t"
            "v_bitop3_b32 %[plo], %[plo], s92, %[t6] bitop3:0x90\n\t"
            "v_bitop3_b32 %[phi], %[phi], s93, %[t6] bitop3:0x90\n\t"
            : [plo] "+v"(plo), [phi] "+v"(phi), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3),
              [t4] "=&v"(t4), [t5] "=&v"(t5), [t6] "=&v"(t6)
            : [d] "v"(d)
            : "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93");
    }
}

struct ScatterShared {
    uint32_t cnt[kWaves][kBins];
    uint32_t digit_start[kBins];
    uint32_t gdelta[kBins];          // global offset of a digit's run minus its start inside the tile
    uint32_t wave_total[kBins / 64];
    uint64_t skeys[kTile];
};

// One tile of a pass.  FULL: no key of the tile is out of range (every tile but the last).  NB: digit bits that are
// matched (8, or 7 for the 7-bit digits of 42-bit keys).  HI: see digit_of2.
template <bool PAIRS, int TAG, int NB, bool HI, bool FULL, int GEN>
__device__ inline void scatter_tile(ScatterShared& sh, float* svals, const uint64_t* __restrict__ kin,
                                    uint64_t* __restrict__ kout, const float* __restrict__ vin, float* __restrict__ vout,
                                    int64_t n, int shift, uint32_t mask, uint32_t my_goff, const Tag& tag,
                                    const RecordSource& gen, int64_t tile) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t tile_base = tile * kTile;
    const int64_t chunk_base = tile_base + int64_t(w) * (64 * kItems);
    uint64_t key[kItems];
    float val[PAIRS ? kItems : 1];
    uint32_t rank[kItems];
    int64_t e_first = 0;
    const int inside = (GEN != 0 && FULL) ? list_of_stretch(gen, tile_base, kTile, e_first) : -1;
    if (GEN != 0 && inside >= 0) {
        const int64_t e0 = e_first + int64_t(w) * (64 * kItems) + lane;
        if (GEN == 1) record_keys_inside<int32_t, true, kItems, PAIRS>(gen, inside, e0, key, val);
        else record_keys_inside<int64_t, true, kItems, PAIRS>(gen, inside, e0, key, val);
    } else
#pragma unroll
    for (int i = 0; i < kItems; ++i) {
        int64_t idx = chunk_base + i * 64 + lane;
        bool ok = FULL || idx < n;
        if (GEN != 0) {
            float wv = 0.f;
            key[i] = ~0ull;
            if (ok) key[i] = GEN == 1 ? record_key<int32_t, true>(gen, idx, wv) : record_key<int64_t, true>(gen, idx, wv);
            if (PAIRS) val[i] = wv;
        } else {
            key[i] = ok ? kin[idx] : ~0ull;
            if (PAIRS) val[i] = ok ? vin[idx] : 0.f;
        }
        if (TAG && ok) {
            // the keys arrive grouped by the field, so neighbouring lanes read the same few entries of starts[]
            const uint32_t f = uint32_t(key[i] >> tag.field_shift) & tag.field_mask;
            // TAG 1: the size of the field value's group among these keys; TAG 2: the caller's table (compile-time
            // variants: a run-time choice here cost the whole-graph build a tenth of its time)
            const uint32_t cnt = TAG == 2 ? uint32_t(tag.counts[f]) : uint32_t(tag.starts[f + 1] - tag.starts[f]);
            key[i] |= uint64_t(cnt < tag.cap ? cnt : tag.cap) << tag.tag_shift;
        }
    }
    // Ranking.  Per round the 64 keys of a wavefront are matched by digit with one ballot per digit bit: a lane keeps,
    // in two 32-bit halves, the lanes that agree with it on every bit so far (match_lanes); the lanes of that set below
    // it are its rank inside the round (mbcnt), the set's size advances the wave's running count of the digit.
#pragma unroll
    for (int i = 0; i < kItems; ++i) {
        const uint32_t d = digit_of2<HI>(uint32_t(key[i]), uint32_t(key[i] >> 32), shift, mask);
        uint32_t plo = ~0u, phi = ~0u;
        bool ok = true;
        if (!FULL) {
            ok = chunk_base + i * 64 + lane < n;
            const uint64_t live = __ballot(ok);
            plo = uint32_t(live); phi = uint32_t(live >> 32);
        }
        match_lanes<NB>(d, plo, phi);
        const uint32_t r = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
        // cnt[w][*] is private to this wavefront: every lane of a digit group reads the running count, then the
        // group's lowest lane advances it for the next round.  The wave barriers keep the compiler from moving
        // the LDS store across the other lanes' loads (the lanes of a wave execute in lockstep).
        const uint32_t base = sh.cnt[w][d];
        __builtin_amdgcn_wave_barrier();
        if (ok && r == 0) sh.cnt[w][d] = base + uint32_t(__popc(plo) + __popc(phi));   // lowest lane of the group
        __builtin_amdgcn_wave_barrier();
        rank[i] = base + r;
    }
    __syncthreads();
    if (tid < kBins) {                     // per digit: offsets of the waves against each other, tile count
        uint32_t run = 0;
#pragma unroll
        for (int ww = 0; ww < kWaves; ++ww) { uint32_t t = sh.cnt[ww][tid]; sh.cnt[ww][tid] = run; run += t; }
        sh.digit_start[tid] = run;
    }
    __syncthreads();
    {                                      // exclusive scan of the 256 tile counts: shuffle scan per wave, 4 wave totals
        uint32_t v = 0, inc = 0;
        if (tid < kBins) {
            v = sh.digit_start[tid];
            inc = v;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t t = __shfl_up(inc, off, 64);
                if (lane >= off) inc += t;
            }
            if (lane == 63) sh.wave_total[w] = inc;
        }
        __syncthreads();
        if (tid < kBins) {
            uint32_t before = 0;
#pragma unroll
            for (int ww = 0; ww < kBins / 64; ++ww) before += ww < w ? sh.wave_total[ww] : 0u;
            const uint32_t start = before + inc - v;
            sh.gdelta[tid] = my_goff - start;
#pragma unroll
            for (int ww = 0; ww < kWaves; ++ww) sh.cnt[ww][tid] += start;     // slot of (wave, digit) inside the tile
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kItems; ++i) {
        if (FULL || chunk_base + i * 64 + lane < n) {
            const uint32_t d = digit_of2<HI>(uint32_t(key[i]), uint32_t(key[i] >> 32), shift, mask);
            const uint32_t pos = sh.cnt[w][d] + rank[i];
            sh.skeys[pos] = key[i];
            if (PAIRS) svals[pos] = val[i];
        }
    }
    __syncthreads();
    // the digit-sorted tile to global memory, in runs of equal digit (fixed trip count: the LDS reads and the stores of
    // a full tile are issued in batches)
    const int n_valid = FULL ? kTile : int(n - tile_base);
#pragma unroll
    for (int i = 0; i < kItems; ++i) {
        const int j = i * kThreads + tid;
        if (FULL || j < n_valid) {
            const uint64_t k = sh.skeys[j];
            const uint32_t d = digit_of2<HI>(uint32_t(k), uint32_t(k >> 32), shift, mask);
            const int64_t out = int64_t(uint32_t(sh.gdelta[d] + uint32_t(j)));
            kout[out] = k;
            if (PAIRS) vout[out] = svals[j];
        }
    }
}

template <bool PAIRS, int TAG, int NB, bool HI, int GEN>
__global__ void __launch_bounds__(kThreads) scatter(const uint64_t* __restrict__ kin, uint64_t* __restrict__ kout,
                                                    const float* __restrict__ vin, float* __restrict__ vout, int64_t n,
                                                    int shift, uint32_t mask, const uint32_t* __restrict__ offs, const Tag tag,
                                                    const RecordSource gen, int64_t n_tiles) {
    __shared__ ScatterShared sh;
    __shared__ float svals[PAIRS ? kTile : 1];
    const int tid = threadIdx.x;
    // Workgroups go to the eight XCDs round robin.  Each XCD takes one contiguous eighth of the tiles, so that the
    // tiles in flight on an XCD are neighbours: their runs of one digit are neighbours in the output too, and the
    // XCD's L2 collects spans of many KiB per digit instead of every eighth 256-byte run.
    const int64_t per_xcd = (n_tiles + kXcds - 1) / kXcds;
    const int64_t tile = int64_t(blockIdx.x % kXcds) * per_xcd + blockIdx.x / kXcds;
    if (tile >= n_tiles) return;
    for (int i = tid; i < kWaves * kBins; i += kThreads) (&sh.cnt[0][0])[i] = 0;
    uint32_t my_goff = 0;
    if (tid < kBins) my_goff = offs[tile * kBins + tid];
    __syncthreads();
    if (tile * kTile + kTile <= n)
        scatter_tile<PAIRS, TAG, NB, HI, true, GEN>(sh, svals, kin, kout, vin, vout, n, shift, mask, my_goff, tag, gen, tile);
    else
        scatter_tile<PAIRS, TAG, NB, HI, false, GEN>(sh, svals, kin, kout, vin, vout, n, shift, mask, my_goff, tag, gen, tile);
}

inline unsigned scatter_grid(unsigned tiles) { return (tiles + kXcds - 1) / kXcds * kXcds; }

template <bool PAIRS, int TAG>
void launch_scatter(unsigned tiles, hipStream_t stream, const uint64_t* a, uint64_t* b, const float* va, float* vb, int64_t n,
                    int shift, uint32_t mask, int nb, const uint32_t* hist, const Tag& tag) {
    const bool hi = shift >= 32;
    const RecordSource none{};
    // digits of 7 bits (42-bit keys in 6 passes) are matched with 7 ballots; every other width with 8 (the bits above
    // the digit are zero in every lane and match trivially)
    if (nb == 7) {
        if (hi) scatter<PAIRS, TAG, 7, true, 0><<<scatter_grid(tiles), kThreads, 0, stream>>>(a, b, va, vb, n, shift, mask, hist, tag, none, int64_t(tiles));
        else scatter<PAIRS, TAG, 7, false, 0><<<scatter_grid(tiles), kThreads, 0, stream>>>(a, b, va, vb, n, shift, mask, hist, tag, none, int64_t(tiles));
    } else {
        if (hi) scatter<PAIRS, TAG, 8, true, 0><<<scatter_grid(tiles), kThreads, 0, stream>>>(a, b, va, vb, n, shift, mask, hist, tag, none, int64_t(tiles));
        else scatter<PAIRS, TAG, 8, false, 0><<<scatter_grid(tiles), kThreads, 0, stream>>>(a, b, va, vb, n, shift, mask, hist, tag, none, int64_t(tiles));
    }
}

// first pass of a sort whose keys come from a record source (low key bits: HI is false)
template <bool PAIRS, int GEN>
void launch_scatter_gen(unsigned tiles, hipStream_t stream, uint64_t* b, float* vb, int64_t n, int shift, uint32_t mask, int nb,
                        const uint32_t* hist, const RecordSource& gen) {
    if (nb == 7) scatter<PAIRS, 0, 7, false, GEN><<<scatter_grid(tiles), kThreads, 0, stream>>>(nullptr, b, nullptr, vb, n, shift, mask, hist, Tag{}, gen, int64_t(tiles));
    else scatter<PAIRS, 0, 8, false, GEN><<<scatter_grid(tiles), kThreads, 0, stream>>>(nullptr, b, nullptr, vb, n, shift, mask, hist, Tag{}, gen, int64_t(tiles));
}

inline size_t align_up(size_t x) { return (x + 255) / 256 * 256; }

}  // namespace

constexpr int kMaxPasses = 8;              // 64 key bits

size_t workspace_bytes(int64_t n) {
    int64_t tiles = (n + kTile - 1) / kTile;
    if (tiles < 1) tiles = 1;
    int64_t chunks = (tiles + kChunk - 1) / kChunk;
    return align_up(size_t(tiles) * kBins * 4) + align_up(size_t(chunks) * kBins * 4 * kMaxPasses);
}

hipError_t sort(uint64_t* a, uint64_t* b, float* va, float* vb, int64_t n, int first_bit, int bits, void* workspace,
                size_t ws_bytes, hipStream_t stream, const Tag* tag, const RecordSource* gen) {
    if (n <= 0 || bits <= 0) return hipSuccess;
    if (tag && (va || gen)) return hipErrorInvalidValue;
    if (ws_bytes < workspace_bytes(n) || n >= (int64_t(1) << 32)) return hipErrorInvalidValue;
    const int64_t tiles = (n + kTile - 1) / kTile;
    const int64_t chunks = (tiles + kChunk - 1) / kChunk;
    uint32_t* hist = static_cast<uint32_t*>(workspace);
    uint32_t* chunk_all = reinterpret_cast<uint32_t*>(static_cast<char*>(workspace) + align_up(size_t(tiles) * kBins * 4));
    const bool pairs = va != nullptr;
    if (passes_for(bits) > kMaxPasses) return hipErrorInvalidValue;
    // the chunk totals of every pass are accumulated by that pass's histogram: one clear for all of them
    hipError_t e0 = hipMemsetAsync(chunk_all, 0, size_t(chunks) * kBins * 4 * passes_for(bits), stream);
    if (e0 != hipSuccess) return e0;
    int pass = 0;
    // equal digit widths over the passes (42 key bits -> 6 passes of 7 bits): fewer bins per pass
    // means longer runs of equal digit per tile, i.e. longer contiguous writes
    const int width = (bits + passes_for(bits) - 1) / passes_for(bits);
    for (int done = 0; done < bits; done += width) {
        const int shift = first_bit + done;
        const int nb = bits - done < width ? bits - done : width;
        const uint32_t mask = (1u << nb) - 1u;
        const bool from_source = gen != nullptr && done == 0;
        if (from_source && shift + nb > 32) return hipErrorInvalidValue;
        uint32_t* chunk = chunk_all + size_t(pass++) * size_t(chunks) * kBins;
        if (!from_source) tile_histogram<0><<<unsigned(tiles), kThreads, 0, stream>>>(a, n, shift, mask, hist, chunk, RecordSource{});
        else if (gen->idx64) tile_histogram<2><<<unsigned(tiles), kThreads, 0, stream>>>(nullptr, n, shift, mask, hist, chunk, *gen);
        else tile_histogram<1><<<unsigned(tiles), kThreads, 0, stream>>>(nullptr, n, shift, mask, hist, chunk, *gen);
        scan_apply<<<unsigned(chunks), kBins, 0, stream>>>(hist, tiles, chunk, chunks);
        if (from_source) {
            if (pairs) {
                if (gen->idx64) launch_scatter_gen<true, 2>(unsigned(tiles), stream, b, vb, n, shift, mask, nb, hist, *gen);
                else launch_scatter_gen<true, 1>(unsigned(tiles), stream, b, vb, n, shift, mask, nb, hist, *gen);
            } else {
                if (gen->idx64) launch_scatter_gen<false, 2>(unsigned(tiles), stream, b, vb, n, shift, mask, nb, hist, *gen);
                else launch_scatter_gen<false, 1>(unsigned(tiles), stream, b, vb, n, shift, mask, nb, hist, *gen);
            }
        }
        else if (pairs) launch_scatter<true, 0>(unsigned(tiles), stream, a, b, va, vb, n, shift, mask, nb, hist, Tag{});
        else if (tag && done == 0 && tag->counts) launch_scatter<false, 2>(unsigned(tiles), stream, a, b, nullptr, nullptr, n, shift, mask, nb, hist, *tag);
        else if (tag && done == 0) launch_scatter<false, 1>(unsigned(tiles), stream, a, b, nullptr, nullptr, n, shift, mask, nb, hist, *tag);
        else launch_scatter<false, 0>(unsigned(tiles), stream, a, b, nullptr, nullptr, n, shift, mask, nb, hist, Tag{});
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        uint64_t* t = a; a = b; b = t;
        float* tv = va; va = vb; vb = tv;
    }
    return hipSuccess;
}

}  // namespace mhsort
