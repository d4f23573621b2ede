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
//     accumulators before the single store of the output row.
// No atomics, fixed summation order: results are bitwise reproducible.
#include <cstdint>

#include <hip/hip_runtime.h>

#include "common.h"

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
    int32_t F;
};

__device__ inline float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
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
template <int LPN, int CH, int UNROLL>
__global__ void __launch_bounds__(kBlock) cspmm_vec4(const SpmmArgs a) {
    constexpr int G = 64 / LPN;
    const int lane = threadIdx.x & 63;
    const int lg = lane & (LPN - 1);
    const int g = lane / LPN;
    const int nvec = a.F >> 2;
    const int wave0 = __builtin_amdgcn_readfirstlane(int((blockIdx.x * kBlock + threadIdx.x) >> 6));
    const int n_waves = int(gridDim.x) * kWavesPerBlock;

    bool col_ok[CH];
    int64_t foff[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        int cv = lg + c * LPN;
        col_ok[c] = cv < nvec;
        foff[c] = col_ok[c] ? int64_t(cv) * 4 : 0;     // idle lanes re-read column 0 (never stored)
    }

    for (int r = wave0; r < a.n_rows; r += n_waves) {
        const int start = a.row_ptr[r], end = a.row_ptr[r + 1];
        float4 are[CH], aim[CH];
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
                    xr[u][c] = ld4(xrow + foff[c]);
                    xi[u][c] = ld4(xrow + a.imx + foff[c]);
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
                float4 xr = ld4(xrow + foff[c]);
                float4 xi = ld4(xrow + a.imx + foff[c]);
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
        if (g == 0) {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if (!col_ok[c]) continue;
                float4 yr = scale4(a.alpha, are[c]), yi = scale4(a.alpha, aim[c]);
                if (a.Z1) {
                    const float* z = a.Z1 + int64_t(r) * a.ldz1 + foff[c];
                    yr = axpy4(a.beta1, ld4(z), yr); yi = axpy4(a.beta1, ld4(z + a.imz1), yi);
                }
                if (a.Z2) {
                    const float* z = a.Z2 + int64_t(r) * a.ldz2 + foff[c];
                    yr = axpy4(a.beta2, ld4(z), yr); yi = axpy4(a.beta2, ld4(z + a.imz2), yi);
                }
                if (a.bias) {
                    float4 b = ld4(a.bias + foff[c]);
                    yr = add4(yr, b); yi = add4(yi, b);
                }
                if (a.relu) {
                    // mask = 1.0*(real >= 0); both parts are multiplied by it (sparse_magnet.py:80-81)
                    float4 m = make_float4(yr.x >= 0.f ? 1.f : 0.f, yr.y >= 0.f ? 1.f : 0.f,
                                           yr.z >= 0.f ? 1.f : 0.f, yr.w >= 0.f ? 1.f : 0.f);
                    yr = make_float4(m.x * yr.x, m.y * yr.y, m.z * yr.z, m.w * yr.w);
                    yi = make_float4(m.x * yi.x, m.y * yi.y, m.z * yi.z, m.w * yi.w);
                    if (a.mask) {
                        uint32_t bits = (m.x != 0.f ? 1u : 0u) | (m.y != 0.f ? 0x100u : 0u) |
                                        (m.z != 0.f ? 0x10000u : 0u) | (m.w != 0.f ? 0x1000000u : 0u);
                        *reinterpret_cast<uint32_t*>(a.mask + int64_t(r) * a.F + foff[c]) = bits;
                    }
                }
                float* y = a.Y + int64_t(r) * a.ldy + foff[c];
                st4(y, yr); st4(y + a.imy, yi);
            }
        }
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
    const int wave0 = __builtin_amdgcn_readfirstlane(int((blockIdx.x * kBlock + threadIdx.x) >> 6));
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
                    if (a.mask) a.mask[int64_t(r) * a.F + f] = m != 0.f ? 1 : 0;
                }
                float* y = a.Y + int64_t(r) * a.ldy + f;
                y[0] = yr; y[a.imy] = yi;
            }
        }
    }
}

unsigned spmm_grid(int32_t n_rows) {
    int64_t want = (int64_t(n_rows) + kWavesPerBlock - 1) / kWavesPerBlock;
    int64_t cap = 256 * 8 * 4;      // four rounds of the blocks the chip holds (256 CUs x 8 blocks)
    if (want < 1) want = 1;
    return unsigned(want < cap ? want : cap);
}

template <int LPN, int CH, int UNROLL>
void launch_vec(const SpmmArgs& a, hipStream_t s) {
    cspmm_vec4<LPN, CH, UNROLL><<<spmm_grid(a.n_rows), kBlock, 0, s>>>(a);
}
template <int LPN>
void launch_scalar(const SpmmArgs& a, hipStream_t s) {
    cspmm_scalar<LPN><<<spmm_grid(a.n_rows), kBlock, 0, s>>>(a);
}

bool mult4(int64_t x) { return (x & 3) == 0; }

}  // namespace

extern "C" int mh_cspmm(const int32_t* row_ptr, const int32_t* col_idx, const float* val, int32_t n_rows,
                        const float* X, int64_t ldx, int64_t imx, float* Y, int64_t ldy, int64_t imy, float alpha,
                        const float* Z1, int64_t ldz1, int64_t imz1, float beta1, const float* Z2, int64_t ldz2,
                        int64_t imz2, float beta2, const float* bias, int relu, uint8_t* mask_out, int32_t F,
                        void* stream_) {
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
    a.bias = bias; a.relu = relu; a.mask = relu ? mask_out : nullptr; a.F = F;

    bool vec = mult4(F) && mult4(ldx) && mult4(imx) && mult4(ldy) && mult4(imy) && mh_aligned16(X) && mh_aligned16(Y) &&
               (!Z1 || (mult4(ldz1) && mult4(imz1) && mh_aligned16(Z1))) &&
               (!Z2 || (mult4(ldz2) && mult4(imz2) && mh_aligned16(Z2))) && (!bias || mh_aligned16(bias)) &&
               (!a.mask || (reinterpret_cast<uintptr_t>(a.mask) & 3u) == 0);
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
        if (a.mask) return mh::fail(MH_ERR_ARG, "mask_out is not supported for F > 1024");
        for (int32_t f0 = 0; f0 < F; f0 += 1024) {
            SpmmArgs s = a;
            int32_t fw = F - f0 < 1024 ? F - f0 : 1024;
            s.X = a.X + f0; s.Y = a.Y + f0; s.F = fw;
            if (a.Z1) s.Z1 = a.Z1 + f0;
            if (a.Z2) s.Z2 = a.Z2 + f0;
            if (a.bias) s.bias = a.bias + f0;
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
