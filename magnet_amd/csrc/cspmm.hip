// Fused complex CSR SpMM for gfx950 (MI355X): one Chebyshev order of MagNet's propagation.
//
//     Y = alpha * (L X) + beta1 * Z1 + beta2 * Z2  [+ bias][complex ReLU]
//
// replaces the four real torch.spmm of process() (reference src/layer/sparse_magnet.py:9-17):
// every stored nonzero (int32 column, complex64 value) is read once and applied to the real
// and the imaginary block of the gathered feature row in the same pass.
//
// Mapping (wave = 64 lanes):
//   * one wavefront owns one output row at a time (rows are dealt round-robin over all
//     waves of the grid, which also spreads hub rows);
//   * a feature row is float32 [2][F] (F real parts, then F imaginary parts).  A lane owns
//     4 consecutive features: it loads one float4 from the real block and one from the
//     imaginary block of the gathered row (each wave-instruction reads up to 1 KiB
//     contiguous), so LPN = F/4 lanes cover a nonzero and G = 64/LPN nonzeros are in flight
//     per wave-instruction; F > 256 gives a lane several such float4 columns;
//   * UNROLL nonzeros per group are issued back to back before the first use, which keeps
//     2*UNROLL 16-byte gathers per lane outstanding;
//   * group partial sums are combined with xor-butterflies at the end of the row; the
//     epilogue (axpy with up to two other matrices, bias, complex ReLU + mask) runs on the
//     accumulators before the single store of the output row;
//   * hub rows (power-law graphs have rows with 10^5 nonzeros) would serialise on one wave, so a
//     per-matrix plan lists the rows longer than 2*chunk nonzeros and cuts them into chunks of
//     `chunk` nonzeros: the row kernel skips them, a chunk kernel gives every chunk its own wave
//     and stores raw partial sums, a combine kernel adds a row's partials in chunk order and runs
//     the epilogue.
// No float atomics, fixed summation order: results are bitwise reproducible.
#include <cstdint>
#include <atomic>
#include <cstdlib>
#include <mutex>

#include <hip/hip_runtime.h>

#include "common.h"

// Hub-row partial sums of mh_cspmm_planned_noalloc: one copy of this array lives on every GPU the
// library is loaded on, from module load until the process ends.  Host code never uses the name as
// an address; static_scratch() resolves the per-device address once with hipGetSymbolAddress.
static __device__ __align__(256) float scratch[MH_NOALLOC_SCRATCH_BYTES / sizeof(float)];

namespace {

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / 64;

struct SpmmArgs {
    const int32_t* row_ptr;
    const int32_t* col_idx;
    const float2* val;
    int32_t n_rows;
    const float* X; int64_t ldx, imx;
    float* Y; int64_t ldy, imy;
    float alpha;
    const float* Z1; int64_t ldz1, imz1; float beta1;
    const float* Z2; int64_t ldz2, imz2; float beta2;
    const float* bias;
    int relu;
    uint8_t* mask;
    int64_t mask_ld;             // row stride of the mask (= the full feature width; a column slice keeps it)
    int32_t F;
    // rows longer than long_threshold nonzeros are cut into chunks of `chunk` nonzeros, one wave each
    // (n_long == 0: every row is handled whole by the row kernel)
    const int32_t* long_row;     // [n_long]   row id
    const int32_t* long_base;    // [n_long]   first chunk of the row
    const int32_t* long_cnt;     // [n_long]   number of chunks of the row
    const int32_t* chunk_long;   // [n_chunks] index into long_* of the row a chunk belongs to
    int32_t n_long, n_chunks, chunk, long_threshold;
    float* scratch;              // [n_chunks][2][F] partial sums
};

__device__ inline float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
// gathered feature rows: read once per nonzero with no reuse worth caching at graph scale
template <bool NT>
__device__ inline float4 ldx4(const float* p) {
    if constexpr (NT) {
        typedef float v4f __attribute__((ext_vector_type(4)));
        v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    } else {
        return *reinterpret_cast<const float4*>(p);
    }
}
__device__ inline void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

__device__ inline void cfma(float4& are, float4& aim, float2 v, float4 xr, float4 xi) {
    are.x = fmaf(v.x, xr.x, are.x); are.y = fmaf(v.x, xr.y, are.y); are.z = fmaf(v.x, xr.z, are.z); are.w = fmaf(v.x, xr.w, are.w);
    are.x = fmaf(-v.y, xi.x, are.x); are.y = fmaf(-v.y, xi.y, are.y); are.z = fmaf(-v.y, xi.z, are.z); are.w = fmaf(-v.y, xi.w, are.w);
    aim.x = fmaf(v.x, xi.x, aim.x); aim.y = fmaf(v.x, xi.y, aim.y); aim.z = fmaf(v.x, xi.z, aim.z); aim.w = fmaf(v.x, xi.w, aim.w);
    aim.x = fmaf(v.y, xr.x, aim.x); aim.y = fmaf(v.y, xr.y, aim.y); aim.z = fmaf(v.y, xr.z, aim.z); aim.w = fmaf(v.y, xr.w, aim.w);
}

__device__ inline float4 shfl_xor4(float4 v, int off) {
    return make_float4(__shfl_xor(v.x, off, 64), __shfl_xor(v.y, off, 64), __shfl_xor(v.z, off, 64), __shfl_xor(v.w, off, 64));
}
__device__ inline float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ inline float4 axpy4(float a, float4 x, float4 y) {
    return make_float4(fmaf(a, x.x, y.x), fmaf(a, x.y, y.y), fmaf(a, x.z, y.z), fmaf(a, x.w, y.w));
}
__device__ inline float4 scale4(float a, float4 x) { return make_float4(a * x.x, a * x.y, a * x.z, a * x.w); }

// LPN: lanes that cover one nonzero's feature row (power of two, <= 64)
// CH : float4 columns per lane (1 unless F > 256)
template <int LPN, int CH>
struct Lanes {
    int lg, g;
    bool col_ok[CH];
    int64_t foff[CH];
    __device__ explicit Lanes(int F) {
        const int lane = threadIdx.x & 63;
        lg = lane & (LPN - 1);
        g = lane / LPN;
        const int nvec = F >> 2;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            int cv = lg + c * LPN;
            col_ok[c] = cv < nvec;
            foff[c] = col_ok[c] ? int64_t(cv) * 4 : 0;     // idle lanes re-read column 0 (never stored)
        }
    }
};

// sum_j val[j] * X[col[j]] over nonzeros [start, end) of one row, into per-lane accumulators;
// afterwards every lane of group 0 holds the sum over all groups.
template <int LPN, int CH, int UNROLL, bool NT>
__device__ inline void accumulate(const SpmmArgs& a, const Lanes<LPN, CH>& L, int start, int end, float4 (&are)[CH],
                                  float4 (&aim)[CH]) {
    constexpr int G = 64 / LPN;
    const int g = L.g;
#pragma unroll
    for (int c = 0; c < CH; ++c) { are[c] = make_float4(0.f, 0.f, 0.f, 0.f); aim[c] = are[c]; }
    int jb = start;
    // full batches: every group has UNROLL in-range nonzeros
    for (; jb + UNROLL * G <= end; jb += UNROLL * G) {
        int cidx[UNROLL]; float2 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            int j = jb + u * G + g;
            cidx[u] = a.col_idx[j];
            v[u] = a.val[j];
        }
        float4 xr[UNROLL][CH], xi[UNROLL][CH];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const float* xrow = a.X + int64_t(cidx[u]) * a.ldx;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                xr[u][c] = ldx4<NT>(xrow + L.foff[c]);
                xi[u][c] = ldx4<NT>(xrow + a.imx + L.foff[c]);
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) cfma(are[c], aim[c], v[u], xr[u][c], xi[u][c]);
    }
    // tail: one nonzero per group per step, out-of-range groups contribute zero
    for (; jb < end; jb += G) {
        int j = jb + g;
        bool ok = j < end;
        int jj = ok ? j : start;
        int cj = a.col_idx[jj];
        float2 v = a.val[jj];
        if (!ok) v = make_float2(0.f, 0.f);
        const float* xrow = a.X + int64_t(cj) * a.ldx;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            float4 xr = ldx4<NT>(xrow + L.foff[c]);
            float4 xi = ldx4<NT>(xrow + a.imx + L.foff[c]);
            cfma(are[c], aim[c], v, xr, xi);
        }
    }
    // combine the G groups
#pragma unroll
    for (int off = LPN; off < 64; off <<= 1) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            are[c] = add4(are[c], shfl_xor4(are[c], off));
            aim[c] = add4(aim[c], shfl_xor4(aim[c], off));
        }
    }
}

// alpha * acc + beta1 Z1 + beta2 Z2 [+ bias][complex ReLU, mask] -> Y, for one float4 column of row r
__device__ inline void epilogue(const SpmmArgs& a, int r, int64_t foff, float4 are, float4 aim) {
    float4 yr = scale4(a.alpha, are), yi = scale4(a.alpha, aim);
    if (a.Z1) {
        const float* z = a.Z1 + int64_t(r) * a.ldz1 + foff;
        yr = axpy4(a.beta1, ld4(z), yr); yi = axpy4(a.beta1, ld4(z + a.imz1), yi);
    }
    if (a.Z2) {
        const float* z = a.Z2 + int64_t(r) * a.ldz2 + foff;
        yr = axpy4(a.beta2, ld4(z), yr); yi = axpy4(a.beta2, ld4(z + a.imz2), yi);
    }
    if (a.bias) {
        float4 b = ld4(a.bias + foff);
        yr = add4(yr, b); yi = add4(yi, b);
    }
    if (a.relu) {
        // mask = 1.0*(real >= 0); both parts are multiplied by it (sparse_magnet.py:80-81)
        float4 m = make_float4(yr.x >= 0.f ? 1.f : 0.f, yr.y >= 0.f ? 1.f : 0.f, yr.z >= 0.f ? 1.f : 0.f,
                               yr.w >= 0.f ? 1.f : 0.f);
        yr = make_float4(m.x * yr.x, m.y * yr.y, m.z * yr.z, m.w * yr.w);
        yi = make_float4(m.x * yi.x, m.y * yi.y, m.z * yi.z, m.w * yi.w);
        if (a.mask) {
            uint32_t bits = (m.x != 0.f ? 1u : 0u) | (m.y != 0.f ? 0x100u : 0u) | (m.z != 0.f ? 0x10000u : 0u) |
                            (m.w != 0.f ? 0x1000000u : 0u);
            *reinterpret_cast<uint32_t*>(a.mask + int64_t(r) * a.mask_ld + foff) = bits;
        }
    }
    float* y = a.Y + int64_t(r) * a.ldy + foff;
    st4(y, yr); st4(y + a.imy, yi);
}

// one wave per row (rows dealt round-robin over the grid's waves); rows in the plan are skipped
template <int LPN, int CH, int UNROLL, bool NT>
__global__ void __launch_bounds__(kBlock) cspmm_vec4(const SpmmArgs a) {
    const Lanes<LPN, CH> L(a.F);
    const int wave0 = __builtin_amdgcn_readfirstlane(int(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6)));
    const int n_waves = int(gridDim.x) * kWavesPerBlock;
    for (int r = wave0; r < a.n_rows; r += n_waves) {
        const int start = a.row_ptr[r], end = a.row_ptr[r + 1];
        if (a.n_long > 0 && end - start > a.long_threshold) continue;
        float4 are[CH], aim[CH];
        accumulate<LPN, CH, UNROLL, NT>(a, L, start, end, are, aim);
        if (L.g == 0) {
#pragma unroll
            for (int c = 0; c < CH; ++c)
                if (L.col_ok[c]) epilogue(a, r, L.foff[c], are[c], aim[c]);
        }
    }
}

// one wave per chunk of a long row: raw partial sums to scratch[chunk][2][F]
template <int LPN, int CH, int UNROLL, bool NT>
__global__ void __launch_bounds__(kBlock) cspmm_vec4_chunks(const SpmmArgs a) {
    const Lanes<LPN, CH> L(a.F);
    const int c0 = __builtin_amdgcn_readfirstlane(int(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6)));
    if (c0 >= a.n_chunks) return;
    const int li = a.chunk_long[c0];
    const int r = a.long_row[li];
    const int start = a.row_ptr[r] + (c0 - a.long_base[li]) * a.chunk;
    const int row_end = a.row_ptr[r + 1];
    const int end = start + a.chunk < row_end ? start + a.chunk : row_end;
    float4 are[CH], aim[CH];
    accumulate<LPN, CH, UNROLL, NT>(a, L, start, end, are, aim);
    if (L.g == 0) {
        float* p = a.scratch + int64_t(c0) * 2 * a.F;
#pragma unroll
        for (int c = 0; c < CH; ++c)
            if (L.col_ok[c]) { st4(p + L.foff[c], are[c]); st4(p + a.F + L.foff[c], aim[c]); }
    }
}

// one wave per long row: partials added in chunk order, then the epilogue
__global__ void __launch_bounds__(kBlock) cspmm_vec4_combine(const SpmmArgs a) {
    const int lane = threadIdx.x & 63;
    const int li = __builtin_amdgcn_readfirstlane(int(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6)));
    if (li >= a.n_long) return;
    const int r = a.long_row[li], base = a.long_base[li], cnt = a.long_cnt[li];
    const int nvec = a.F >> 2;
    for (int cv = lane; cv < nvec; cv += 64) {
        const int64_t foff = int64_t(cv) * 4;
        float4 are = make_float4(0.f, 0.f, 0.f, 0.f), aim = are;
        for (int p = 0; p < cnt; ++p) {
            const float* s = a.scratch + int64_t(base + p) * 2 * a.F + foff;
            are = add4(are, ld4(s)); aim = add4(aim, ld4(s + a.F));
        }
        epilogue(a, r, foff, are, aim);
    }
}

// Any F (also F % 4 != 0 or unaligned operands): one feature per lane, LPN lanes per nonzero,
// feature chunks of LPN handled one after the other.
template <int LPN>
__global__ void __launch_bounds__(kBlock) cspmm_scalar(const SpmmArgs a) {
    constexpr int G = 64 / LPN;
    const int lane = threadIdx.x & 63;
    const int lg = lane & (LPN - 1);
    const int g = lane / LPN;
    const int wave0 = __builtin_amdgcn_readfirstlane(int(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6)));
    const int n_waves = int(gridDim.x) * kWavesPerBlock;
    for (int r = wave0; r < a.n_rows; r += n_waves) {
        const int start = a.row_ptr[r], end = a.row_ptr[r + 1];
        for (int f0 = 0; f0 < a.F; f0 += LPN) {
            const int f = f0 + lg;
            const bool fok = f < a.F;
            const int fs = fok ? f : 0;
            float are = 0.f, aim = 0.f;
            for (int jb = start; jb < end; jb += G) {
                int j = jb + g;
                bool ok = j < end;
                int jj = ok ? j : start;
                int cj = a.col_idx[jj];
                float2 v = a.val[jj];
                if (!ok) v = make_float2(0.f, 0.f);
                const float* xrow = a.X + int64_t(cj) * a.ldx + fs;
                float xr = xrow[0], xi = xrow[a.imx];
                are = fmaf(v.x, xr, are); are = fmaf(-v.y, xi, are);
                aim = fmaf(v.x, xi, aim); aim = fmaf(v.y, xr, aim);
            }
#pragma unroll
            for (int off = LPN; off < 64; off <<= 1) {
                are += __shfl_xor(are, off, 64);
                aim += __shfl_xor(aim, off, 64);
            }
            if (g == 0 && fok) {
                float yr = a.alpha * are, yi = a.alpha * aim;
                if (a.Z1) {
                    const float* z = a.Z1 + int64_t(r) * a.ldz1 + f;
                    yr = fmaf(a.beta1, z[0], yr); yi = fmaf(a.beta1, z[a.imz1], yi);
                }
                if (a.Z2) {
                    const float* z = a.Z2 + int64_t(r) * a.ldz2 + f;
                    yr = fmaf(a.beta2, z[0], yr); yi = fmaf(a.beta2, z[a.imz2], yi);
                }
                if (a.bias) { float b = a.bias[f]; yr += b; yi += b; }
                if (a.relu) {
                    float m = yr >= 0.f ? 1.f : 0.f;
                    yr *= m; yi *= m;
                    if (a.mask) a.mask[int64_t(r) * a.mask_ld + f] = m != 0.f ? 1 : 0;
                }
                float* y = a.Y + int64_t(r) * a.ldy + f;
                y[0] = yr; y[a.imy] = yi;
            }
        }
    }
}

// ---- plan: which rows are long, and where their chunks live ---------------------------------
__global__ void __launch_bounds__(kBlock) plan_count_kernel(const int32_t* __restrict__ row_ptr, int32_t n_rows,
                                                            int32_t threshold, int32_t chunk, int32_t* counters) {
    int r = blockIdx.x * kBlock + threadIdx.x;
    if (r >= n_rows) return;
    int len = row_ptr[r + 1] - row_ptr[r];
    if (len > threshold) {
        atomicAdd(&counters[0], 1);
        atomicAdd(&counters[1], (len + chunk - 1) / chunk);
    }
}

// The slot a long row gets in the lists depends on scheduling, the results do not: a row's chunks
// are contiguous and are combined in chunk order.
__global__ void __launch_bounds__(kBlock) plan_fill_kernel(const int32_t* __restrict__ row_ptr, int32_t n_rows,
                                                           int32_t threshold, int32_t chunk, int32_t* cursors,
                                                           int32_t* long_row, int32_t* long_base, int32_t* long_cnt,
                                                           int32_t* chunk_long) {
    int r = blockIdx.x * kBlock + threadIdx.x;
    if (r >= n_rows) return;
    int len = row_ptr[r + 1] - row_ptr[r];
    if (len > threshold) {
        int n = (len + chunk - 1) / chunk;
        int li = atomicAdd(&cursors[0], 1);
        int base = atomicAdd(&cursors[1], n);
        long_row[li] = r; long_base[li] = base; long_cnt[li] = n;
        for (int t = 0; t < n; ++t) chunk_long[base + t] = li;
    }
}

// Upper bound of workgroups of the row kernel for the calling thread's next launches (mh_cspmm_set_grid_cap);
// 0 = the library default.
thread_local int32_t t_grid_cap = 0;

unsigned spmm_grid(int32_t n_rows) {
    int64_t want = (int64_t(n_rows) + kWavesPerBlock - 1) / kWavesPerBlock;
    static int64_t cap = -1;
    if (cap < 0) {
        const char* e = getenv("MH_SPMM_GRID_CAP");      // tuning knob: 0 = one wave per row
        cap = e ? atoll(e) : int64_t(256) * 8 * 4;
    }
    if (want < 1) want = 1;
    if (t_grid_cap > 0) return unsigned(want < t_grid_cap ? want : t_grid_cap);
    return unsigned((cap == 0 || want < cap) ? want : cap);
}

bool use_nt() {
    // tuning knob: MH_SPMM_NT=1 gathers feature rows with non-temporal loads.  Off by default: it was
    // measured slightly slower at 10M nodes and much slower whenever hot rows fit the caches.
    static int mode = -1;
    if (mode < 0) {
        const char* e = getenv("MH_SPMM_NT");
        mode = (e && e[0] == '1') ? 1 : 0;
    }
    return mode == 1;
}

template <int LPN, int CH, int UNROLL>
void launch_vec(const SpmmArgs& a, hipStream_t s) {
    const unsigned g_chunks = unsigned((a.n_chunks + kWavesPerBlock - 1) / kWavesPerBlock);
    if (use_nt()) {
        cspmm_vec4<LPN, CH, UNROLL, true><<<spmm_grid(a.n_rows), kBlock, 0, s>>>(a);
        if (a.n_long > 0) cspmm_vec4_chunks<LPN, CH, UNROLL, true><<<g_chunks, kBlock, 0, s>>>(a);
    } else {
        cspmm_vec4<LPN, CH, UNROLL, false><<<spmm_grid(a.n_rows), kBlock, 0, s>>>(a);
        if (a.n_long > 0) cspmm_vec4_chunks<LPN, CH, UNROLL, false><<<g_chunks, kBlock, 0, s>>>(a);
    }
    if (a.n_long > 0)
        cspmm_vec4_combine<<<unsigned((a.n_long + kWavesPerBlock - 1) / kWavesPerBlock), kBlock, 0, s>>>(a);
}
template <int LPN>
void launch_scalar(const SpmmArgs& a, hipStream_t s) {
    cspmm_scalar<LPN><<<spmm_grid(a.n_rows), kBlock, 0, s>>>(a);
}

bool mult4(int64_t x) { return (x & 3) == 0; }

}  // namespace

namespace {

int run_spmm(SpmmArgs a, hipStream_t stream) {
    const int32_t F = a.F;
    bool vec = mult4(F) && mult4(a.ldx) && mult4(a.imx) && mult4(a.ldy) && mult4(a.imy) && mh_aligned16(a.X) &&
               mh_aligned16(a.Y) && (!a.Z1 || (mult4(a.ldz1) && mult4(a.imz1) && mh_aligned16(a.Z1))) &&
               (!a.Z2 || (mult4(a.ldz2) && mult4(a.imz2) && mh_aligned16(a.Z2))) && (!a.bias || mh_aligned16(a.bias)) &&
               (!a.mask || (reinterpret_cast<uintptr_t>(a.mask) & 3u) == 0) && (!a.scratch || mh_aligned16(a.scratch));
    if (!vec || F > 1024) a.n_long = 0;            // the plan is used by the single-pass vector path only
    if (vec && F <= 1024) {
        int nvec = F >> 2;
        if (nvec <= 1) launch_vec<1, 1, 4>(a, stream);
        else if (nvec <= 2) launch_vec<2, 1, 4>(a, stream);
        else if (nvec <= 4) launch_vec<4, 1, 4>(a, stream);
        else if (nvec <= 8) launch_vec<8, 1, 4>(a, stream);
        else if (nvec <= 16) launch_vec<16, 1, 4>(a, stream);
        else if (nvec <= 32) launch_vec<32, 1, 4>(a, stream);
        else if (nvec <= 64) launch_vec<64, 1, 4>(a, stream);
        else if (nvec <= 128) launch_vec<64, 2, 2>(a, stream);
        else launch_vec<64, 4, 1>(a, stream);
    } else if (vec) {
        // wider than one pass: column slices of 1024 features, each a full pass over the CSR
        for (int32_t f0 = 0; f0 < F; f0 += 1024) {
            SpmmArgs s = a;
            int32_t fw = F - f0 < 1024 ? F - f0 : 1024;
            s.X = a.X + f0; s.Y = a.Y + f0; s.F = fw;
            if (a.Z1) s.Z1 = a.Z1 + f0;
            if (a.Z2) s.Z2 = a.Z2 + f0;
            if (a.bias) s.bias = a.bias + f0;
            if (a.mask) s.mask = a.mask + f0;
            int nvec = fw >> 2;
            if (nvec <= 64) launch_vec<64, 1, 4>(s, stream);
            else if (nvec <= 128) launch_vec<64, 2, 2>(s, stream);
            else launch_vec<64, 4, 1>(s, stream);
        }
    } else {
        if (F <= 1) launch_scalar<1>(a, stream);
        else if (F <= 2) launch_scalar<2>(a, stream);
        else if (F <= 4) launch_scalar<4>(a, stream);
        else if (F <= 8) launch_scalar<8>(a, stream);
        else if (F <= 16) launch_scalar<16>(a, stream);
        else if (F <= 32) launch_scalar<32>(a, stream);
        else launch_scalar<64>(a, stream);
    }
    MH_LAUNCH_CHECK("cspmm");
    return MH_OK;
}

}  // namespace

extern "C" int mh_cspmm_planned(const int32_t* row_ptr, const int32_t* col_idx, const float* val, int32_t n_rows,
                                const float* X, int64_t ldx, int64_t imx, float* Y, int64_t ldy, int64_t imy,
                                float alpha, const float* Z1, int64_t ldz1, int64_t imz1, float beta1,
                                const float* Z2, int64_t ldz2, int64_t imz2, float beta2, const float* bias, int relu,
                                uint8_t* mask_out, int32_t F, const int32_t* plan, int32_t n_long, int32_t n_chunks,
                                int32_t chunk, float* scratch, size_t scratch_bytes, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    MH_REQUIRE(n_rows >= 0 && F >= 1, "n_rows=%d F=%d", n_rows, F);
    if (n_rows == 0) return MH_OK;
    MH_REQUIRE(row_ptr && X && Y, "row_ptr/X/Y is null");
    MH_REQUIRE(Y != X, "Y must not alias X");
    MH_REQUIRE((reinterpret_cast<uintptr_t>(val) & 7u) == 0, "val must be 8-byte aligned");
    MH_REQUIRE(ldx >= F && ldy >= F && imx >= F && imy >= F, "ld/im smaller than F");
    SpmmArgs a{};
    a.row_ptr = row_ptr; a.col_idx = col_idx; a.val = reinterpret_cast<const float2*>(val); a.n_rows = n_rows;
    a.X = X; a.ldx = ldx; a.imx = imx;
    a.Y = Y; a.ldy = ldy; a.imy = imy; a.alpha = alpha;
    a.Z1 = Z1; a.ldz1 = ldz1; a.imz1 = imz1; a.beta1 = beta1;
    a.Z2 = Z2; a.ldz2 = ldz2; a.imz2 = imz2; a.beta2 = beta2;
    a.bias = bias; a.relu = relu; a.mask = relu ? mask_out : nullptr; a.mask_ld = F; a.F = F;
    if (plan && n_long > 0) {
        MH_REQUIRE(n_chunks >= n_long && chunk >= 64, "bad plan sizes: n_long=%d n_chunks=%d chunk=%d", n_long, n_chunks, chunk);
        MH_REQUIRE(scratch != nullptr, "scratch is null");
        if (scratch_bytes < size_t(n_chunks) * 2 * size_t(F) * sizeof(float))
            return mh::fail(MH_ERR_WORKSPACE, "scratch %zu < required %zu bytes", scratch_bytes,
                            size_t(n_chunks) * 2 * size_t(F) * sizeof(float));
        a.long_row = plan + 2; a.long_base = a.long_row + n_long; a.long_cnt = a.long_base + n_long;
        a.chunk_long = a.long_cnt + n_long;
        a.n_long = n_long; a.n_chunks = n_chunks; a.chunk = chunk; a.long_threshold = 2 * chunk;
        a.scratch = scratch;
    }
    return run_spmm(a, stream);
}

namespace {

constexpr int kMaxDevices = 64;
std::atomic<float*> g_static_scratch[kMaxDevices];      // device address of `scratch` on each GPU, resolved on first use
std::mutex g_static_scratch_mu;

// address of this GPU's copy of `scratch` (the current device's).  Once a device's address is known the call is one
// atomic load; only the first call per device takes the mutex and asks the runtime.
int static_scratch(float** out) {
    int dev = -1;
    MH_HIP(hipGetDevice(&dev));
    MH_REQUIRE(dev >= 0 && dev < kMaxDevices, "device ordinal %d outside [0, %d)", dev, kMaxDevices);
    float* p = g_static_scratch[dev].load(std::memory_order_acquire);
    if (!p) {
        std::lock_guard<std::mutex> lock(g_static_scratch_mu);
        p = g_static_scratch[dev].load(std::memory_order_relaxed);
        if (!p) {
            void* sym = nullptr;
            MH_HIP(hipGetSymbolAddress(&sym, HIP_SYMBOL(scratch)));
            MH_REQUIRE(sym != nullptr, "hipGetSymbolAddress returned a null address");
            p = static_cast<float*>(sym);
            g_static_scratch[dev].store(p, std::memory_order_release);
        }
    }
    *out = p;
    return MH_OK;
}

}  // namespace

extern "C" int mh_cspmm_noalloc_init(void) {
    float* p = nullptr;
    return static_scratch(&p);
}

extern "C" int mh_cspmm_planned_noalloc(const int32_t* row_ptr, const int32_t* col_idx, const float* val,
                                        int32_t n_rows, const float* X, int64_t ldx, int64_t imx, float* Y,
                                        int64_t ldy, int64_t imy, float alpha, const float* Z1, int64_t ldz1,
                                        int64_t imz1, float beta1, const float* Z2, int64_t ldz2, int64_t imz2,
                                        float beta2, const float* bias, int relu, uint8_t* mask_out, int32_t F,
                                        const int32_t* plan, int32_t n_long, int32_t n_chunks, int32_t chunk,
                                        void* stream_) {
    float* ptr = nullptr;
    size_t need = 0;
    if (plan && n_long > 0) {
        MH_REQUIRE(n_chunks >= n_long && F >= 1, "bad plan sizes: n_long=%d n_chunks=%d F=%d", n_long, n_chunks, F);
        need = size_t(n_chunks) * 2 * size_t(F) * sizeof(float);
        if (need > size_t(MH_NOALLOC_SCRATCH_BYTES))
            return mh::fail(MH_ERR_WORKSPACE,
                            "hub-row partial sums need %zu bytes, the built-in scratch holds %zu (n_chunks=%d, F=%d): "
                            "use mh_cspmm_planned with caller memory", need, size_t(MH_NOALLOC_SCRATCH_BYTES), n_chunks, F);
        int rc = static_scratch(&ptr);
        if (rc != MH_OK) return rc;
    }
    // every chunk's slot is written by the chunk kernel before the combine kernel reads it on the same
    // stream, so nothing left over from an earlier call is ever read and nothing needs clearing
    return mh_cspmm_planned(row_ptr, col_idx, val, n_rows, X, ldx, imx, Y, ldy, imy, alpha, Z1, ldz1, imz1, beta1, Z2,
                            ldz2, imz2, beta2, bias, relu, mask_out, F, plan, n_long, n_chunks, chunk, ptr, need, stream_);
}

extern "C" int mh_cspmm(const int32_t* row_ptr, const int32_t* col_idx, const float* val, int32_t n_rows,
                        const float* X, int64_t ldx, int64_t imx, float* Y, int64_t ldy, int64_t imy, float alpha,
                        const float* Z1, int64_t ldz1, int64_t imz1, float beta1, const float* Z2, int64_t ldz2,
                        int64_t imz2, float beta2, const float* bias, int relu, uint8_t* mask_out, int32_t F,
                        void* stream_) {
    return mh_cspmm_planned(row_ptr, col_idx, val, n_rows, X, ldx, imx, Y, ldy, imy, alpha, Z1, ldz1, imz1, beta1, Z2,
                            ldz2, imz2, beta2, bias, relu, mask_out, F, nullptr, 0, 0, 0, nullptr, 0, stream_);
}

extern "C" int mh_cspmm_set_grid_cap(int32_t workgroups) {
    MH_REQUIRE(workgroups >= 0, "workgroups=%d", workgroups);
    t_grid_cap = workgroups;
    return MH_OK;
}

extern "C" size_t mh_cspmm_plan_ints(int32_t n_long, int32_t n_chunks) {
    return size_t(2) + size_t(3) * size_t(n_long > 0 ? n_long : 0) + size_t(n_chunks > 0 ? n_chunks : 0);
}

extern "C" int mh_cspmm_plan_count(const int32_t* row_ptr, int32_t n_rows, int32_t chunk, int32_t* counters2,
                                   int32_t* n_long_host, int32_t* n_chunks_host, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    MH_REQUIRE(n_rows >= 0 && chunk >= 64, "n_rows=%d chunk=%d (chunk must be >= 64)", n_rows, chunk);
    MH_REQUIRE(n_long_host && n_chunks_host, "null output");
    *n_long_host = 0; *n_chunks_host = 0;
    if (n_rows == 0) return MH_OK;
    MH_REQUIRE(row_ptr && counters2, "null pointer");
    MH_HIP(hipMemsetAsync(counters2, 0, 2 * sizeof(int32_t), stream));
    plan_count_kernel<<<unsigned((n_rows + kBlock - 1) / kBlock), kBlock, 0, stream>>>(row_ptr, n_rows, 2 * chunk, chunk, counters2);
    MH_LAUNCH_CHECK("plan_count");
    int32_t h[2] = {0, 0};
    MH_HIP(hipMemcpyAsync(h, counters2, sizeof(h), hipMemcpyDeviceToHost, stream));
    MH_HIP(hipStreamSynchronize(stream));
    *n_long_host = h[0]; *n_chunks_host = h[1];
    return MH_OK;
}

extern "C" int mh_cspmm_plan_fill(const int32_t* row_ptr, int32_t n_rows, int32_t chunk, int32_t n_long,
                                  int32_t n_chunks, int32_t* plan, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    MH_REQUIRE(n_rows >= 0 && chunk >= 64 && n_long >= 0 && n_chunks >= n_long, "bad sizes");
    if (n_rows == 0 || n_long == 0) return MH_OK;
    MH_REQUIRE(row_ptr && plan, "null pointer");
    MH_HIP(hipMemsetAsync(plan, 0, 2 * sizeof(int32_t), stream));
    int32_t* long_row = plan + 2;
    int32_t* long_base = long_row + n_long;
    int32_t* long_cnt = long_base + n_long;
    int32_t* chunk_long = long_cnt + n_long;
    plan_fill_kernel<<<unsigned((n_rows + kBlock - 1) / kBlock), kBlock, 0, stream>>>(row_ptr, n_rows, 2 * chunk, chunk, plan,
                                                                                     long_row, long_base, long_cnt, chunk_long);
    MH_LAUNCH_CHECK("plan_fill");
    return MH_OK;
}

// ---- whole recursions in one call (K SpMM launches on the stream) ---------------------------
namespace {
struct PlanArgs { const int32_t* plan; int32_t n_long, n_chunks, chunk; float* scratch; size_t scratch_bytes; };
int spmm_sq(const int32_t* rp, const int32_t* ci, const float* val, int32_t n, const float* X, float* Y, float alpha,
            const float* Z1, float b1, const float* Z2, float b2, const float* bias, int relu, uint8_t* mask, int32_t F,
            const PlanArgs& p, void* stream) {
    const int64_t ld = 2 * int64_t(F), im = F;
    return mh_cspmm_planned(rp, ci, val, n, X, ld, im, Y, ld, im, alpha, Z1, ld, im, b1, Z2, ld, im, b2, bias, relu, mask, F,
                            p.plan, p.n_long, p.n_chunks, p.chunk, p.scratch, p.scratch_bytes, stream);
}
}  // namespace

extern "C" int mh_cheb_terms(const int32_t* row_ptr, const int32_t* col_idx, const float* val, int32_t n,
                             float* Z, int64_t order_stride, int32_t n_terms, int32_t F, const int32_t* plan,
                             int32_t n_long, int32_t n_chunks, int32_t chunk, float* scratch, size_t scratch_bytes,
                             void* stream) {
    MH_REQUIRE(n >= 0 && n_terms >= 1 && F >= 1, "n=%d n_terms=%d F=%d", n, n_terms, F);
    if (n == 0 || n_terms == 1) return MH_OK;
    MH_REQUIRE(Z != nullptr, "Z is null");
    MH_REQUIRE(order_stride >= int64_t(n) * 2 * F, "order_stride smaller than one matrix");
    PlanArgs p{plan, n_long, n_chunks, chunk, scratch, scratch_bytes};
    for (int k = 1; k < n_terms; ++k) {
        float* zk = Z + int64_t(k) * order_stride;
        const float* zp = Z + int64_t(k - 1) * order_stride;
        int rc = (k == 1) ? spmm_sq(row_ptr, col_idx, val, n, zp, zk, 1.0f, nullptr, 0.f, nullptr, 0.f, nullptr, 0, nullptr, F, p, stream)
                          : spmm_sq(row_ptr, col_idx, val, n, zp, zk, 2.0f, Z + int64_t(k - 2) * order_stride, -1.0f,
                                    nullptr, 0.f, nullptr, 0, nullptr, F, p, stream);
        if (rc != MH_OK) return rc;
    }
    return MH_OK;
}

extern "C" int mh_clenshaw(const int32_t* row_ptr, const int32_t* col_idx, const float* val, int32_t n, const float* Y,
                           int64_t order_stride, int32_t n_terms, float* work, float* out, const float* bias, int relu,
                           uint8_t* mask_out, int32_t F, const int32_t* plan, int32_t n_long, int32_t n_chunks,
                           int32_t chunk, float* scratch, size_t scratch_bytes, void* stream) {
    MH_REQUIRE(n >= 0 && n_terms >= 1 && F >= 1, "n=%d n_terms=%d F=%d", n, n_terms, F);
    if (n == 0) return MH_OK;
    MH_REQUIRE(Y && out, "Y/out is null");
    MH_REQUIRE(order_stride >= int64_t(n) * 2 * F, "order_stride smaller than one matrix");
    const int64_t mat = int64_t(n) * 2 * F;
    PlanArgs p{plan, n_long, n_chunks, chunk, scratch, scratch_bytes};
    if (n_terms == 1) {
        MH_HIP(hipMemcpyAsync(out, Y, size_t(mat) * sizeof(float), hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
        return mh_bias_relu(out, 2 * int64_t(F), F, bias, relu, mask_out, n, F, stream);
    }
    MH_REQUIRE(n_terms == 2 || work != nullptr, "work (two matrices) is needed for n_terms > 2");
    // b_K = Y_K ; b_k = Y_k + 2 L b_{k+1} - b_{k+2} ; result = Y_0 + L b_1 - b_2
    const float* b1 = Y + int64_t(n_terms - 1) * order_stride;
    const float* b2 = nullptr;
    int slot = 0;
    for (int k = n_terms - 2; k >= 1; --k) {
        float* nb = work + int64_t(slot) * mat;
        int rc = spmm_sq(row_ptr, col_idx, val, n, b1, nb, 2.0f, Y + int64_t(k) * order_stride, 1.0f, b2, -1.0f, nullptr, 0,
                         nullptr, F, p, stream);
        if (rc != MH_OK) return rc;
        b2 = b1; b1 = nb; slot ^= 1;
    }
    return spmm_sq(row_ptr, col_idx, val, n, b1, out, 1.0f, Y, 1.0f, b2, -1.0f, bias, relu, mask_out, F, p, stream);
}
