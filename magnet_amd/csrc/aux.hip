// Layout / activation helpers and the synthetic-graph generator of libmagnet_hip.so (gfx950).
#include <cstdint>

#include <hip/hip_runtime.h>

#include "common.h"

namespace mh {

char* error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace mh

extern "C" int mh_version(void) { return 300; }
extern "C" const char* mh_last_error(void) { return mh::error_buffer(); }

namespace {

constexpr int kBlock = 256;

inline unsigned grid_for(int64_t n, int64_t cap = int64_t(1) << 30) {
    int64_t g = (n + kBlock - 1) / kBlock;
    if (g < 1) g = 1;
    return unsigned(g < cap ? g : cap);
}

// two planar matrices -> row-planar [n][2][F]; one thread per output float (or float4)
template <bool VEC>
__global__ void __launch_bounds__(kBlock) pack_kernel(const float* __restrict__ re, int64_t ld_re,
                                                      const float* __restrict__ im_, int64_t ld_im,
                                                      float* __restrict__ out, int64_t ld, int64_t im,
                                                      int64_t n_rows, int32_t F) {
    const int per_row = VEC ? (F >> 2) : F;             // work items per half row
    const int64_t total = n_rows * 2 * per_row;
    for (int64_t t = int64_t(blockIdx.x) * kBlock + threadIdx.x; t < total; t += int64_t(gridDim.x) * kBlock) {
        int64_t r = t / (2 * per_row);
        int rem = int(t - r * 2 * per_row);
        int part = rem / per_row;
        int c = rem - part * per_row;
        const float* src = part ? im_ + r * ld_im : re + r * ld_re;
        float* dst = out + r * ld + (part ? im : 0);
        if (VEC) reinterpret_cast<float4*>(dst)[c] = reinterpret_cast<const float4*>(src)[c];
        else dst[c] = src[c];
    }
}

template <bool VEC>
__global__ void __launch_bounds__(kBlock) bias_relu_kernel(float* __restrict__ Y, int64_t ld, int64_t im,
                                                           const float* __restrict__ bias, int relu,
                                                           uint8_t* __restrict__ mask, int64_t n_rows, int32_t F) {
    const int per_row = VEC ? (F >> 2) : F;
    const int64_t total = n_rows * per_row;
    for (int64_t t = int64_t(blockIdx.x) * kBlock + threadIdx.x; t < total; t += int64_t(gridDim.x) * kBlock) {
        int64_t r = t / per_row;
        int c = int(t - r * per_row);
        float* yr_p = Y + r * ld + (VEC ? 4 * c : c);
        float* yi_p = yr_p + im;
        if (VEC) {
            float4 yr = *reinterpret_cast<float4*>(yr_p), yi = *reinterpret_cast<float4*>(yi_p);
            if (bias) {
                float4 b = *reinterpret_cast<const float4*>(bias + 4 * c);
                yr.x += b.x; yr.y += b.y; yr.z += b.z; yr.w += b.w;
                yi.x += b.x; yi.y += b.y; yi.z += b.z; yi.w += b.w;
            }
            if (relu) {
                float mx = yr.x >= 0.f ? 1.f : 0.f, my = yr.y >= 0.f ? 1.f : 0.f;
                float mz = yr.z >= 0.f ? 1.f : 0.f, mw = yr.w >= 0.f ? 1.f : 0.f;
                yr.x *= mx; yr.y *= my; yr.z *= mz; yr.w *= mw;
                yi.x *= mx; yi.y *= my; yi.z *= mz; yi.w *= mw;
                if (mask)
                    *reinterpret_cast<uint32_t*>(mask + r * F + 4 * c) =
                        (mx != 0.f ? 1u : 0u) | (my != 0.f ? 0x100u : 0u) | (mz != 0.f ? 0x10000u : 0u) |
                        (mw != 0.f ? 0x1000000u : 0u);
            }
            *reinterpret_cast<float4*>(yr_p) = yr; *reinterpret_cast<float4*>(yi_p) = yi;
        } else {
            float yr = *yr_p, yi = *yi_p;
            if (bias) { yr += bias[c]; yi += bias[c]; }
            if (relu) {
                float m = yr >= 0.f ? 1.f : 0.f;
                yr *= m; yi *= m;
                if (mask) mask[r * F + c] = m != 0.f ? 1 : 0;
            }
            *yr_p = yr; *yi_p = yi;
        }
    }
}

// row-planar -> two planar matrices; one thread per float (or float4)
template <bool VEC>
__global__ void __launch_bounds__(kBlock) unpack_kernel(const float* __restrict__ in, int64_t ld, int64_t im,
                                                        float* __restrict__ re, int64_t ld_re, float* __restrict__ im_,
                                                        int64_t ld_im, int64_t n_rows, int32_t F) {
    const int per_row = VEC ? (F >> 2) : F;
    const int64_t total = n_rows * 2 * per_row;
    for (int64_t t = int64_t(blockIdx.x) * kBlock + threadIdx.x; t < total; t += int64_t(gridDim.x) * kBlock) {
        int64_t r = t / (2 * per_row);
        int rem = int(t - r * 2 * per_row);
        int part = rem / per_row, c = rem - part * per_row;
        const float* src = in + r * ld + (part ? im : 0);
        float* dst = part ? im_ + r * ld_im : re + r * ld_re;
        if (VEC) reinterpret_cast<float4*>(dst)[c] = reinterpret_cast<const float4*>(src)[c];
        else dst[c] = src[c];
    }
}

// rows idx[i] of a row-planar matrix <-> row i of a dense buffer [n_idx][2][F].  VEC: `lpr` lanes (a power of two,
// at most 64) move one row as float4 items over both parts, 64 / lpr rows per wavefront; otherwise one wave per row,
// one float per lane and step.
template <bool GATHER, bool VEC>
__global__ void __launch_bounds__(kBlock) move_rows_kernel(float* __restrict__ X, int64_t ld, int64_t im,
                                                           const int32_t* __restrict__ idx, int64_t n_idx,
                                                           float* __restrict__ buf, int32_t F, int lpr) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t(blockIdx.x) * kBlock + threadIdx.x) >> 6;
    const int64_t n_waves = (int64_t(gridDim.x) * kBlock) >> 6;
    if (VEC) {
        const int per_wave = 64 / lpr, sub = lane / lpr, l = lane & (lpr - 1);
        const int nvec = F >> 2;
        for (int64_t i = wave * per_wave + sub; i < n_idx; i += n_waves * per_wave) {
            float* xr = X + int64_t(idx[i]) * ld;
            float* b = buf + i * 2 * F;
            for (int t = l; t < 2 * nvec; t += lpr) {
                const int part = t >= nvec ? 1 : 0, c = t - part * nvec;
                float4* px = reinterpret_cast<float4*>(xr + (part ? im : 0)) + c;
                float4* pb = reinterpret_cast<float4*>(b + (part ? F : 0)) + c;
                if (GATHER) *pb = *px; else *px = *pb;
            }
        }
    } else {
        for (int64_t i = wave; i < n_idx; i += n_waves) {
            float* xr = X + int64_t(idx[i]) * ld;
            float* b = buf + i * 2 * F;
            for (int c = lane; c < F; c += 64) {
                if (GATHER) { b[c] = xr[c]; b[F + c] = xr[im + c]; }
                else { xr[c] = b[c]; xr[im + c] = b[F + c]; }
            }
        }
    }
}

// out = G * mask on both parts (out may be G itself); one thread per feature (or four)
template <bool VEC>
__global__ void __launch_bounds__(kBlock) relu_bwd_kernel(const float* G, int64_t ld, int64_t im,
                                                          const uint8_t* __restrict__ mask, float* out, int64_t ldo,
                                                          int64_t imo, int64_t n_rows, int32_t F) {
    const int per_row = VEC ? (F >> 2) : F;
    const int64_t total = n_rows * per_row;
    for (int64_t t = int64_t(blockIdx.x) * kBlock + threadIdx.x; t < total; t += int64_t(gridDim.x) * kBlock) {
        int64_t r = t / per_row;
        int c = int(t - r * per_row);
        if (VEC) {
            const uint32_t m = *reinterpret_cast<const uint32_t*>(mask + r * F + 4 * c);
            const float mx = (m & 0xffu) ? 1.f : 0.f, my = (m & 0xff00u) ? 1.f : 0.f;
            const float mz = (m & 0xff0000u) ? 1.f : 0.f, mw = (m & 0xff000000u) ? 1.f : 0.f;
            const float* g = G + r * ld + 4 * c;
            float4 gr = *reinterpret_cast<const float4*>(g), gi = *reinterpret_cast<const float4*>(g + im);
            gr.x *= mx; gr.y *= my; gr.z *= mz; gr.w *= mw;
            gi.x *= mx; gi.y *= my; gi.z *= mz; gi.w *= mw;
            float* o = out + r * ldo + 4 * c;
            *reinterpret_cast<float4*>(o) = gr; *reinterpret_cast<float4*>(o + imo) = gi;
        } else {
            float m = mask[r * F + c] ? 1.f : 0.f;
            const float* g = G + r * ld + c;
            float* o = out + r * ldo + c;
            float gr = g[0] * m, gi = g[im] * m;
            o[0] = gr; o[imo] = gi;
        }
    }
}

// Philox4x32-10, counter = (lo, hi, 0, 0), key = (seed_lo, seed_hi)
__device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t k0, uint32_t k1, uint32_t& o0, uint32_t& o1) {
    uint32_t c2 = 0, c3 = 0;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        uint64_t p0 = uint64_t(0xD2511F53u) * c0, p1 = uint64_t(0xCD9E8D57u) * c2;
        uint32_t n0 = uint32_t(p1 >> 32) ^ c1 ^ k0, n1 = uint32_t(p1);
        uint32_t n2 = uint32_t(p0 >> 32) ^ c3 ^ k1, n3 = uint32_t(p0);
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o0 = c0; o1 = c1;
}

__global__ void __launch_bounds__(kBlock) powerlaw_kernel(int32_t n_nodes, int64_t n_edges, uint64_t seed,
                                                          int64_t first_edge, int32_t* __restrict__ src,
                                                          int32_t* __restrict__ dst) {
    for (int64_t e = int64_t(blockIdx.x) * kBlock + threadIdx.x; e < n_edges; e += int64_t(gridDim.x) * kBlock) {
        uint64_t id = uint64_t(first_edge + e);
        uint32_t r0, r1;
        philox4x32_10(uint32_t(id), uint32_t(id >> 32), uint32_t(seed), uint32_t(seed >> 32), r0, r1);
        double u1 = (double(r0) + 0.5) * (1.0 / 4294967296.0), u2 = (double(r1) + 0.5) * (1.0 / 4294967296.0);
        int64_t s = int64_t(double(n_nodes) * u1 * u1), d = int64_t(double(n_nodes) * u2 * u2);
        if (s > n_nodes - 1) s = n_nodes - 1;
        if (d > n_nodes - 1) d = n_nodes - 1;
        src[e] = int32_t(s); dst[e] = int32_t(d);
    }
}

bool mult4(int64_t x) { return (x & 3) == 0; }

}  // namespace

extern "C" int mh_pack(const float* re, int64_t ld_re, const float* im_, int64_t ld_im, float* out, int64_t ld,
                       int64_t im, int64_t n_rows, int32_t F, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    MH_REQUIRE(n_rows >= 0 && F >= 1, "n_rows=%lld F=%d", (long long)n_rows, F);
    if (n_rows == 0) return MH_OK;
    MH_REQUIRE(re && im_ && out, "null pointer");
    MH_REQUIRE(ld_re >= F && ld_im >= F && ld >= F && im >= F, "stride smaller than F");
    bool vec = mult4(F) && mult4(ld_re) && mult4(ld_im) && mult4(ld) && mult4(im) && mh_aligned16(re) &&
               mh_aligned16(im_) && mh_aligned16(out);
    int64_t items = n_rows * 2 * (vec ? F / 4 : F);
    if (vec) pack_kernel<true><<<grid_for(items, 1 << 20), kBlock, 0, stream>>>(re, ld_re, im_, ld_im, out, ld, im, n_rows, F);
    else pack_kernel<false><<<grid_for(items, 1 << 20), kBlock, 0, stream>>>(re, ld_re, im_, ld_im, out, ld, im, n_rows, F);
    MH_LAUNCH_CHECK("pack");
    return MH_OK;
}

extern "C" int mh_bias_relu(float* Y, int64_t ld, int64_t im, const float* bias, int relu, uint8_t* mask_out,
                            int64_t n_rows, int32_t F, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    MH_REQUIRE(n_rows >= 0 && F >= 1, "n_rows=%lld F=%d", (long long)n_rows, F);
    if (n_rows == 0 || (!bias && !relu)) return MH_OK;
    MH_REQUIRE(Y, "Y is null");
    MH_REQUIRE(ld >= F && im >= F, "stride smaller than F");
    uint8_t* mask = relu ? mask_out : nullptr;
    bool vec = mult4(F) && mult4(ld) && mult4(im) && mh_aligned16(Y) && (!bias || mh_aligned16(bias)) &&
               (!mask || (reinterpret_cast<uintptr_t>(mask) & 3u) == 0);
    int64_t items = n_rows * (vec ? F / 4 : F);
    if (vec) bias_relu_kernel<true><<<grid_for(items, 1 << 20), kBlock, 0, stream>>>(Y, ld, im, bias, relu, mask, n_rows, F);
    else bias_relu_kernel<false><<<grid_for(items, 1 << 20), kBlock, 0, stream>>>(Y, ld, im, bias, relu, mask, n_rows, F);
    MH_LAUNCH_CHECK("bias_relu");
    return MH_OK;
}

extern "C" int mh_relu_bwd_to(const float* G, int64_t ld, int64_t im, const uint8_t* mask, float* out, int64_t ldo,
                              int64_t imo, int64_t n_rows, int32_t F, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    MH_REQUIRE(n_rows >= 0 && F >= 1, "n_rows=%lld F=%d", (long long)n_rows, F);
    if (n_rows == 0) return MH_OK;
    MH_REQUIRE(G && mask && out, "null pointer");
    MH_REQUIRE(ld >= F && im >= F && ldo >= F && imo >= F, "stride smaller than F");
    const bool vec = mult4(F) && mult4(ld) && mult4(im) && mult4(ldo) && mult4(imo) && mh_aligned16(G) && mh_aligned16(out) &&
                     (reinterpret_cast<uintptr_t>(mask) & 3u) == 0;
    if (vec) relu_bwd_kernel<true><<<grid_for(n_rows * (F / 4), 1 << 20), kBlock, 0, stream>>>(G, ld, im, mask, out, ldo, imo, n_rows, F);
    else relu_bwd_kernel<false><<<grid_for(n_rows * F, 1 << 20), kBlock, 0, stream>>>(G, ld, im, mask, out, ldo, imo, n_rows, F);
    MH_LAUNCH_CHECK("relu_bwd");
    return MH_OK;
}

extern "C" int mh_relu_bwd(float* G, int64_t ld, int64_t im, const uint8_t* mask, int64_t n_rows, int32_t F,
                           void* stream_) {
    return mh_relu_bwd_to(G, ld, im, mask, G, ld, im, n_rows, F, stream_);
}

extern "C" int mh_powerlaw_edges(int32_t n_nodes, int64_t n_edges, uint64_t seed, int64_t first_edge, int32_t* src,
                                 int32_t* dst, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    MH_REQUIRE(n_nodes >= 1 && n_edges >= 0 && first_edge >= 0, "bad sizes");
    if (n_edges == 0) return MH_OK;
    MH_REQUIRE(src && dst, "null pointer");
    powerlaw_kernel<<<grid_for(n_edges, 1 << 16), kBlock, 0, stream>>>(n_nodes, n_edges, seed, first_edge, src, dst);
    MH_LAUNCH_CHECK("powerlaw_edges");
    return MH_OK;
}

// Read-only pass over a buffer: every lane keeps kProbeUnroll independent 16-byte loads in flight.
constexpr int kProbeUnroll = 8;
__global__ void __launch_bounds__(kBlock) stream_read_kernel(const float4* __restrict__ p, int64_t n_vec,
                                                             float* __restrict__ sink) {
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    float acc = 0.0f;
    for (; i + (kProbeUnroll - 1) * stride < n_vec; i += kProbeUnroll * stride) {
        float4 v[kProbeUnroll];
#pragma unroll
        for (int u = 0; u < kProbeUnroll; ++u) v[u] = p[i + u * stride];
#pragma unroll
        for (int u = 0; u < kProbeUnroll; ++u) acc += (v[u].x + v[u].y) + (v[u].z + v[u].w);
    }
    for (; i < n_vec; i += stride) { float4 v = p[i]; acc += (v.x + v.y) + (v.z + v.w); }
    if (acc == 1.2345678e30f) *sink = acc;      // keeps the loads alive; never true for finite data in practice
}

extern "C" int mh_stream_read(const void* buf, size_t bytes, float* sink, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    MH_REQUIRE(buf && sink, "null pointer");
    MH_REQUIRE((reinterpret_cast<uintptr_t>(buf) & 15) == 0, "buffer must be 16-byte aligned");
    int64_t n_vec = int64_t(bytes / 16);
    if (n_vec == 0) return MH_OK;
    stream_read_kernel<<<grid_for(n_vec / kProbeUnroll + 1, 1 << 14), kBlock, 0, stream>>>(
        static_cast<const float4*>(buf), n_vec, sink);
    MH_LAUNCH_CHECK("stream_read");
    return MH_OK;
}

extern "C" int mh_unpack(const float* in, int64_t ld, int64_t im, float* re, int64_t ld_re, float* im_, int64_t ld_im,
                         int64_t n_rows, int32_t F, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    MH_REQUIRE(n_rows >= 0 && F >= 1, "n_rows=%lld F=%d", (long long)n_rows, F);
    if (n_rows == 0) return MH_OK;
    MH_REQUIRE(in && re && im_, "null pointer");
    MH_REQUIRE(ld_re >= F && ld_im >= F && ld >= F && im >= F, "stride smaller than F");
    const bool vec = mult4(F) && mult4(ld_re) && mult4(ld_im) && mult4(ld) && mult4(im) && mh_aligned16(in) &&
                     mh_aligned16(re) && mh_aligned16(im_);
    if (vec) unpack_kernel<true><<<grid_for(n_rows * 2 * (F / 4), 1 << 20), kBlock, 0, stream>>>(in, ld, im, re, ld_re, im_, ld_im, n_rows, F);
    else unpack_kernel<false><<<grid_for(n_rows * 2 * F, 1 << 20), kBlock, 0, stream>>>(in, ld, im, re, ld_re, im_, ld_im, n_rows, F);
    MH_LAUNCH_CHECK("unpack");
    return MH_OK;
}

namespace {

// lanes that move one row of 2 * F / 4 float4 items: the smallest power of two that covers them, at most 64
inline int lanes_per_row(int32_t F) {
    int lpr = 1;
    while (lpr < 64 && lpr < 2 * (F >> 2)) lpr <<= 1;
    return lpr;
}

template <bool GATHER>
int move_rows(float* X, int64_t ld, int64_t im, const int32_t* idx, int64_t n_idx, float* buf, int32_t F, hipStream_t stream) {
    const bool vec = mult4(F) && mult4(ld) && mult4(im) && mh_aligned16(X) && mh_aligned16(buf);
    if (vec) {
        const int lpr = lanes_per_row(F);
        move_rows_kernel<GATHER, true><<<grid_for(n_idx * lpr, 1 << 16), kBlock, 0, stream>>>(X, ld, im, idx, n_idx, buf, F, lpr);
    } else {
        move_rows_kernel<GATHER, false><<<grid_for(n_idx * 64, 1 << 16), kBlock, 0, stream>>>(X, ld, im, idx, n_idx, buf, F, 64);
    }
    return MH_OK;
}

}  // namespace

extern "C" int mh_gather_rows(const float* X, int64_t ld, int64_t im, const int32_t* idx, int64_t n_idx, float* buf,
                              int32_t F, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    MH_REQUIRE(n_idx >= 0 && F >= 1, "n_idx=%lld F=%d", (long long)n_idx, F);
    if (n_idx == 0) return MH_OK;
    MH_REQUIRE(X && idx && buf, "null pointer");
    MH_REQUIRE(ld >= F && im >= F, "stride smaller than F");
    move_rows<true>(const_cast<float*>(X), ld, im, idx, n_idx, buf, F, stream);
    MH_LAUNCH_CHECK("gather_rows");
    return MH_OK;
}

extern "C" int mh_scatter_rows(const float* buf, const int32_t* idx, int64_t n_idx, float* X, int64_t ld, int64_t im,
                               int32_t F, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    MH_REQUIRE(n_idx >= 0 && F >= 1, "n_idx=%lld F=%d", (long long)n_idx, F);
    if (n_idx == 0) return MH_OK;
    MH_REQUIRE(X && idx && buf, "null pointer");
    MH_REQUIRE(ld >= F && im >= F, "stride smaller than F");
    move_rows<false>(X, ld, im, idx, n_idx, const_cast<float*>(buf), F, stream);
    MH_LAUNCH_CHECK("scatter_rows");
    return MH_OK;
}
