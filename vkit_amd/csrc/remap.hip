// cv.remap / cv.warpAffine / cv.warpPerspective (INTER_LINEAR, BORDER_CONSTANT 0) on gfx950.
// One lane per destination pixel, 64 consecutive pixels of a row per wavefront; the arithmetic is
// the integer fixed-point pipeline of OpenCV's imgwarp.cpp (see oracle/vkx_oracle.c for citations).
#include "vkx_internal.h"

namespace {

struct CoordMap {      // cv.remap: coordinates come from two float planes
    static constexpr bool kTile2D = false;      // the map is read row-major: a wavefront walks 64 columns of a row
    const float *mx, *my;
    ptrdiff_t stride;
    struct Column {};
    struct Rows {};
    __device__ __forceinline__ Column column(int) const { return Column(); }
    __device__ __forceinline__ Rows rows(int, int) const { return Rows(); }
    __device__ __forceinline__ void at(const Column &, const Rows &, int, int x, int y, int &X, int &Y) const { (*this)(x, y, X, Y); }
    __device__ __forceinline__ void operator()(int x, int y, int &X, int &Y) const
    {
        X = vkd::cv_round(mx[(ptrdiff_t)y * stride + x] * 32.f);
        Y = vkd::cv_round(my[(ptrdiff_t)y * stride + x] * 32.f);
    }
};

struct CoordAffine {   // warpAffine: inverse matrix, AB_BITS = 10 fixed point
    // a rotated / sheared source footprint: 64 destination pixels of ONE row reach a slanted strip of the source (6 degrees:
    // 8 source rows, ~32 cache lines per tap-load instruction); a wavefront therefore takes 16 columns x 4 rows per instruction
    // (5 - 6 lines), four wavefronts side by side so that their 48-byte row segments complete cache lines on the way out
    static constexpr bool kTile2D = true;
    double m[6];
    // adelta[x] / bdelta[x] of cv::warpAffine depend on the column only: a lane that walks several rows of its column
    // computes them once
    struct Column { int adelta, bdelta; };
    __device__ __forceinline__ Column column(int x) const
    {
        return Column{vkd::cv_round(m[0] * x * 1024), vkd::cv_round(m[3] * x * 1024)};
    }
    // X0 / Y0 of cv::warpAffine depend on the row only: lane r of the wavefront computes those of row y0 + r (one double
    // evaluation per wavefront instead of one per row), every lane reads them back with v_readlane
    struct Rows { int X0, Y0; };
    __device__ __forceinline__ Rows rows(int y0, int lane) const
    {
        const int y = y0 + (lane & 15);
        return Rows{vkd::cv_round((m[1] * y + m[2]) * 1024) + 16, vkd::cv_round((m[4] * y + m[5]) * 1024) + 16};
    }
    // `row`: the lane's row inside the tile (0 .. 15), per lane: the row terms come from the lane that computed them
    __device__ __forceinline__ void at(const Column &c, const Rows &r, int row, int, int, int &X, int &Y) const
    {
        X = (__builtin_amdgcn_ds_bpermute(row << 2, r.X0) + c.adelta) >> 5;
        Y = (__builtin_amdgcn_ds_bpermute(row << 2, r.Y0) + c.bdelta) >> 5;
    }
    __device__ __forceinline__ void operator()(int x, int y, int &X, int &Y) const
    {
        const int adelta = vkd::cv_round(m[0] * x * 1024);
        const int bdelta = vkd::cv_round(m[3] * x * 1024);
        const int X0 = vkd::cv_round((m[1] * y + m[2]) * 1024) + 16;
        const int Y0 = vkd::cv_round((m[4] * y + m[5]) * 1024) + 16;
        X = (X0 + adelta) >> 5;
        Y = (Y0 + bdelta) >> 5;
    }
};

struct CoordPerspective { // warpPerspective: inverse matrix, per pixel in double, 32x32 blocks
    static constexpr bool kTile2D = true;
    double m[9];
    int bw0;
    struct Column {};
    struct Rows {};
    __device__ __forceinline__ Column column(int) const { return Column(); }
    __device__ __forceinline__ Rows rows(int, int) const { return Rows(); }
    __device__ __forceinline__ void at(const Column &, const Rows &, int, int x, int y, int &X, int &Y) const { (*this)(x, y, X, Y); }
    __device__ __forceinline__ void operator()(int x, int y, int &X, int &Y) const
    {
        const int xb = (x / bw0) * bw0, x1 = x - xb;
        const double X0 = m[0] * xb + m[1] * y + m[2];
        const double Y0 = m[3] * xb + m[4] * y + m[5];
        const double W0 = m[6] * xb + m[7] * y + m[8];
        double W = W0 + m[6] * x1;
        W = W ? 32 / W : 0;
        const double fX = fmax((double)INT_MIN, fmin((double)INT_MAX, (X0 + m[0] * x1) * W));
        const double fY = fmax((double)INT_MIN, fmin((double)INT_MAX, (Y0 + m[3] * x1) * W));
        X = vkd::cv_round(fX);
        Y = vkd::cv_round(fY);
    }
};

typedef unsigned long long u64_u1 __attribute__((aligned(1)));
typedef uint32_t u32_u1 __attribute__((aligned(1)));

constexpr int kRgbRows = 4;      // rows a wavefront of the RGB path walks

// The RGB path for coordinate generators with a slanted footprint (Coord::kTile2D): a workgroup covers 64 x 16 destination pixels,
// wavefront w the columns [16 w, 16 w + 16); lane = (column lane & 15, row lane >> 4) and a lane walks the rows 4 rr + (lane >> 4),
// rr = 0 .. 3: every tap-load instruction of a wavefront reaches a 16 x 4 block of the destination.  Arithmetic and stores as in the
// row-major path below.
// OFF32: both planes are smaller than 4 GiB and the source pitch is below 2^24 (checked on the host): byte offsets are formed with
// full-rate 24-bit multiplies in 32 bits on the uniform base pointers instead of 64-bit multiply-adds per lane.
template <class Coord, bool OFF32>
__device__ __forceinline__ void sample_rgb_tile2d(const uint8_t *__restrict__ src, int sh, int sw, ptrdiff_t sstride,
                                                  uint8_t *__restrict__ dst, int dh, int dw, ptrdiff_t dstride, const Coord &coord)
{
    const int lane = threadIdx.x, lx = lane & 15, ly = lane >> 4;
    const int x = blockIdx.x * 64 + threadIdx.y * 16 + lx;
    const int y0 = blockIdx.y * 4 * kRgbRows;
    if (y0 >= dh) return;                       // uniform over the workgroup
    const bool active = x < dw;
    const typename Coord::Column col = coord.column(x);
    const typename Coord::Rows rws = coord.rows(y0, lane);
    const uint32_t src_mis = (uint32_t)(uintptr_t)src & 3u;       // (uniform)
    const uint8_t *src4 = src - src_mis;
    int Xs[kRgbRows], Ys[kRgbRows];
    unsigned long long ta[kRgbRows], tb[kRgbRows];
    bool inside[kRgbRows];
#pragma unroll
    for (int rr = 0; rr < kRgbRows; rr++) {
        const int row = 4 * rr + ly, yy = min(y0 + row, dh - 1);
        Xs[rr] = Ys[rr] = 0;
        coord.at(col, rws, row, x, yy, Xs[rr], Ys[rr]);       // (every lane: the row terms travel by a wavefront shuffle)
        const int sx = Xs[rr] >> 5, sy = Ys[rr] >> 5;
        inside[rr] = active && (unsigned)sx < (unsigned)max(sw - (OFF32 ? 3 : 2), 0) && (unsigned)sy < (unsigned)(sh - 1);
        ta[rr] = tb[rr] = 0;
        if (inside[rr]) {
            if constexpr (OFF32) {
                // 0 <= sy < 2^15, 0 < sstride < 2^24, sh * sstride + 8 <= 2^32
                const uint32_t o0 = __umul24((uint32_t)sy, (uint32_t)sstride) + __umul24((uint32_t)sx, 3u);
                // The six tap bytes of a row as three ALIGNED dwords + a byte alignment in registers: an 8-byte load at an arbitrary
                // byte address costs the texture addresser several accesses per lane (warpAffine 8192^2: 0.216 ms with the
                // unaligned pair, 0.179 with this).  `inside` leaves the last three source columns to the rim sampler, so the
                // twelve bytes stay inside the row; a source that is not 4-byte aligned is read from its aligned-down base.
                typedef uint32_t u32x3 __attribute__((ext_vector_type(3), aligned(4)));
                const uint32_t oa = o0 + src_mis, ob = oa + (uint32_t)sstride;
                const u32x3 a = *(const u32x3 *)(src4 + (size_t)(oa & ~3u)), b = *(const u32x3 *)(src4 + (size_t)(ob & ~3u));
                ta[rr] = (uint64_t)__builtin_amdgcn_alignbyte(a.y, a.x, oa & 3u) | ((uint64_t)__builtin_amdgcn_alignbyte(a.z, a.y, oa & 3u) << 32);
                tb[rr] = (uint64_t)__builtin_amdgcn_alignbyte(b.y, b.x, ob & 3u) | ((uint64_t)__builtin_amdgcn_alignbyte(b.z, b.y, ob & 3u) << 32);
            } else {
                const uint8_t *q = src + (ptrdiff_t)sy * sstride + (ptrdiff_t)sx * 3;   // 3 sx + 8 <= 3 sw
                ta[rr] = *(const u64_u1 *)q;
                tb[rr] = *(const u64_u1 *)(q + sstride);
            }
        }
    }
#pragma unroll
    for (int rr = 0; rr < kRgbRows; rr++) {
        const int yy = y0 + 4 * rr + ly;
        const bool row_ok = active && yy < dh;
        uint32_t P = 0;                     // r | g << 8 | b << 16
        if (inside[rr]) {
            const int fx = Xs[rr] & 31, fy = Ys[rr] & 31;
            const uint32_t wx = (uint32_t)(32 - fx) | ((uint32_t)fx << 24);
            const uint32_t t0 = (uint32_t)ta[rr], t1 = (uint32_t)(ta[rr] >> 32), b0 = (uint32_t)tb[rr], b1 = (uint32_t)(tb[rr] >> 32);
            const uint32_t ar = __builtin_amdgcn_udot4(t0, wx, 0u, false);
            const uint32_t ag = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(t1, t0, 1), wx, 0u, false);
            const uint32_t ab = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(t1, t0, 2), wx, 0u, false);
            const uint32_t wy0 = (uint32_t)(32 - fy), wy1 = (uint32_t)fy;
            const uint32_t br = __builtin_amdgcn_udot4(b0, wx, 0u, false);
            const uint32_t bg = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(b1, b0, 1), wx, 0u, false);
            const uint32_t bb = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(b1, b0, 2), wx, 0u, false);
            const uint32_t r = (__umul24(br, wy1) + __umul24(ar, wy0) + 512u) >> 10;
            const uint32_t g = (__umul24(bg, wy1) + __umul24(ag, wy0) + 512u) >> 10;
            const uint32_t b = (__umul24(bb, wy1) + __umul24(ab, wy0) + 512u) >> 10;
            P = r | (g << 8) | (b << 16);
        } else if (row_ok) {
            uint8_t px[3];
            vkd::sample_u8<3>(src, sh, sw, sstride, Xs[rr], Ys[rr], px);
            P = (uint32_t)px[0] | ((uint32_t)px[1] << 8) | ((uint32_t)px[2] << 16);
        }
        // (lanes with x & 3 == 3 -- the last of every 16-lane row among them -- never use their neighbour's pixel)
        const uint32_t Pn = (uint32_t)__builtin_amdgcn_mov_dpp((int)P, 0x130 /* wave_shl:1: from lane + 1 */, 0xf, 0xf, true);
        if (!row_ok) continue;
        const int m = x & 3;
        if (x < (dw & ~3)) {
            if (m < 3) {
                const uint32_t wv = (P >> (8 * m)) | (Pn << (24 - 8 * m));
                // byte 3 x + m of the row: (x >> 2) * 12 + 4 m
                if constexpr (OFF32) *(u32_u1 *)(dst + (size_t)((uint32_t)yy * (uint32_t)dstride + (uint32_t)(3 * x + m))) = wv;
                else *(u32_u1 *)(dst + (ptrdiff_t)yy * dstride + (ptrdiff_t)(x >> 2) * 12 + m * 4) = wv;
            }
        } else {
            uint8_t *d = dst + (ptrdiff_t)yy * dstride + (ptrdiff_t)x * 3;
            d[0] = (uint8_t)P; d[1] = (uint8_t)(P >> 8); d[2] = (uint8_t)(P >> 16);
        }
    }
}

template <int CN, class Coord, bool OFF32 = false>
__global__ void __launch_bounds__(256) k_sample_u8(const uint8_t *__restrict__ src, int sh, int sw, ptrdiff_t sstride,
                                                   uint8_t *__restrict__ dst, int dh, int dw, ptrdiff_t dstride,
                                                   Coord coord)
{
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if constexpr (CN == 3) {
        // RGB: the gather and the store of the fused chain kernel (fused.hip, phases C and E).  A pixel whose 2 x 2 taps lie
        // inside the source takes two unaligned 8-byte loads (6 bytes each are the tap pair of a row), horizontal pairs as
        // v_dot4_u32_u8, the vertical pair as 24-bit multiply-adds: the same integer as the four weighted taps.  Four
        // neighbouring lanes write their 12 bytes as three dwords.
        // a wavefront walks kRgbRows consecutive rows of its 64 columns: per-column coordinate terms are computed once
        const int lane = threadIdx.x;
        if constexpr (Coord::kTile2D) {
            sample_rgb_tile2d<Coord, OFF32>(src, sh, sw, sstride, dst, dh, dw, dstride, coord);
            return;
        }
        const bool active = x < dw;
        const typename Coord::Column col = coord.column(x);
        const int y0 = (blockIdx.y * 4 + threadIdx.y) * kRgbRows;
        if (y0 >= dh) return;                   // uniform over the wavefront
        const typename Coord::Rows rws = coord.rows(y0, lane);
        // all rows' coordinates and tap loads first (eight 8-byte loads in flight per lane), the arithmetic after
        int Xs[kRgbRows], Ys[kRgbRows];
        unsigned long long ta[kRgbRows], tb[kRgbRows];
        bool inside[kRgbRows];
#pragma unroll
        for (int rr = 0; rr < kRgbRows; rr++) {
            const int yy = min(y0 + rr, dh - 1);
            Xs[rr] = Ys[rr] = 0;
            if (active) coord.at(col, rws, rr, x, yy, Xs[rr], Ys[rr]);
            const int sx = Xs[rr] >> 5, sy = Ys[rr] >> 5;
            // (row-major footprint, cv.remap: the unaligned pair is the faster form here -- aligned dwords measured 0.289 against
            //  0.27 ms at 8192^2)
            inside[rr] = active && (unsigned)sx < (unsigned)max(sw - 2, 0) && (unsigned)sy < (unsigned)(sh - 1);
            ta[rr] = tb[rr] = 0;
            if (inside[rr]) {
                const uint8_t *q = src + (ptrdiff_t)sy * sstride + (ptrdiff_t)sx * 3;   // 3 sx + 8 <= 3 sw
                ta[rr] = *(const u64_u1 *)q;
                tb[rr] = *(const u64_u1 *)(q + sstride);
            }
        }
#pragma unroll
        for (int rr = 0; rr < kRgbRows; rr++) {
            const int yy = y0 + rr;
            if (yy >= dh) break;                // uniform over the wavefront
            uint32_t P = 0;                     // r | g << 8 | b << 16
            if (inside[rr]) {
                const int fx = Xs[rr] & 31, fy = Ys[rr] & 31;
                const uint32_t wx = (uint32_t)(32 - fx) | ((uint32_t)fx << 24);
                const uint32_t t0 = (uint32_t)ta[rr], t1 = (uint32_t)(ta[rr] >> 32), b0 = (uint32_t)tb[rr], b1 = (uint32_t)(tb[rr] >> 32);
                const uint32_t ar = __builtin_amdgcn_udot4(t0, wx, 0u, false);
                const uint32_t ag = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(t1, t0, 1), wx, 0u, false);
                const uint32_t ab = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(t1, t0, 2), wx, 0u, false);
                const uint32_t wy0 = (uint32_t)(32 - fy), wy1 = (uint32_t)fy;
                const uint32_t br = __builtin_amdgcn_udot4(b0, wx, 0u, false);
                const uint32_t bg = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(b1, b0, 1), wx, 0u, false);
                const uint32_t bb = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(b1, b0, 2), wx, 0u, false);
                const uint32_t r = (__umul24(br, wy1) + __umul24(ar, wy0) + 512u) >> 10;
                const uint32_t g = (__umul24(bg, wy1) + __umul24(ag, wy0) + 512u) >> 10;
                const uint32_t b = (__umul24(bb, wy1) + __umul24(ab, wy0) + 512u) >> 10;
                P = r | (g << 8) | (b << 16);
            } else if (active) {
                uint8_t px[3];
                vkd::sample_u8<3>(src, sh, sw, sstride, Xs[rr], Ys[rr], px);
                P = (uint32_t)px[0] | ((uint32_t)px[1] << 8) | ((uint32_t)px[2] << 16);
            }
            // lane 63 (x & 3 == 3) never uses its neighbour's pixel
            const uint32_t Pn = (uint32_t)__builtin_amdgcn_mov_dpp((int)P, 0x130 /* wave_shl:1: from lane + 1 */, 0xf, 0xf, true);
            if (!active) continue;
            uint8_t *drow = dst + (ptrdiff_t)yy * dstride;
            const int m = x & 3;
            if (x < (dw & ~3)) {
                if (m < 3) *(u32_u1 *)(drow + (ptrdiff_t)(x >> 2) * 12 + m * 4) = (P >> (8 * m)) | (Pn << (24 - 8 * m));
            } else {
                uint8_t *d = drow + (ptrdiff_t)x * 3;
                d[0] = (uint8_t)P; d[1] = (uint8_t)(P >> 8); d[2] = (uint8_t)(P >> 16);
            }
        }
        return;
    } else {
        if (x >= dw || y >= dh) return;
        int X, Y;
        coord(x, y, X, Y);
        uint8_t px[CN];
        vkd::sample_u8<CN>(src, sh, sw, sstride, X, Y, px);
        uint8_t *d = dst + (ptrdiff_t)y * dstride + (ptrdiff_t)x * CN;
#pragma unroll
        for (int k = 0; k < CN; k++) d[k] = px[k];
    }
}

template <class Coord>
__global__ void __launch_bounds__(256) k_sample_f32(const float *__restrict__ src, int sh, int sw, ptrdiff_t sstride,
                                                    float *__restrict__ dst, int dh, int dw, ptrdiff_t dstride,
                                                    Coord coord)
{
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x >= dw || y >= dh) return;
    int X, Y;
    coord(x, y, X, Y);
    dst[(ptrdiff_t)y * dstride + x] = vkd::sample_f32(src, sh, sw, sstride, X, Y);
}

template <class Coord>
int launch_u8(vkx_ctx *ctx, const uint8_t *src, int sh, int sw, int cn, ptrdiff_t sstride, uint8_t *dst, int dh,
              int dw, ptrdiff_t dstride, const Coord &coord)
{
    VKX_REQUIRE(ctx && src && dst, "NULL argument");
    VKX_REQUIRE(sh > 0 && sw > 0 && dh >= 0 && dw >= 0, "bad shape");
    VKX_REQUIRE(sh <= 32767 && sw <= 32767, "source larger than 32767 px (cv.remap limit)");
    if (dh == 0 || dw == 0) return VKX_OK;
    dim3 block(64, 4), grid(vkx_blocks(dw, 64), vkx_blocks(dh, 4));
    switch (cn) {
    case 1: { VKX_TIMED(ctx, "k_sample_u8"); k_sample_u8<1, Coord><<<grid, block, 0, ctx->stream>>>(src, sh, sw, sstride, dst, dh, dw, dstride, coord); } break;
    case 3: {
        VKX_TIMED(ctx, "k_sample_u8");
        const dim3 g3(grid.x, vkx_blocks(dh, 4 * kRgbRows));
        const bool off32 = Coord::kTile2D && sstride > 0 && sstride < (1 << 24) && dstride > 0 && (long long)sh * sstride + 16 <= 0xffffffffLL &&
                           (long long)dh * dstride + 4 <= 0xffffffffLL;
        if (off32) k_sample_u8<3, Coord, true><<<g3, block, 0, ctx->stream>>>(src, sh, sw, sstride, dst, dh, dw, dstride, coord);
        else k_sample_u8<3, Coord><<<g3, block, 0, ctx->stream>>>(src, sh, sw, sstride, dst, dh, dw, dstride, coord);
    } break;
    case 4: { VKX_TIMED(ctx, "k_sample_u8"); k_sample_u8<4, Coord><<<grid, block, 0, ctx->stream>>>(src, sh, sw, sstride, dst, dh, dw, dstride, coord); } break;
    default: vkx_set_error("unsupported channel count %d", cn); return VKX_ERR_UNSUPPORTED;
    }
    VKX_LAUNCH_CHECK();
    return VKX_OK;
}

template <class Coord>
int launch_f32(vkx_ctx *ctx, const float *src, int sh, int sw, ptrdiff_t sstride, float *dst, int dh, int dw,
               ptrdiff_t dstride, const Coord &coord)
{
    VKX_REQUIRE(ctx && src && dst, "NULL argument");
    VKX_REQUIRE(sh > 0 && sw > 0 && dh >= 0 && dw >= 0, "bad shape");
    VKX_REQUIRE(sh <= 32767 && sw <= 32767, "source larger than 32767 px (cv.remap limit)");
    if (dh == 0 || dw == 0) return VKX_OK;
    dim3 block(64, 4), grid(vkx_blocks(dw, 64), vkx_blocks(dh, 4));
    { VKX_TIMED(ctx, "k_sample_f32"); k_sample_f32<Coord><<<grid, block, 0, ctx->stream>>>(src, sh, sw, sstride, dst, dh, dw, dstride, coord); }
    VKX_LAUNCH_CHECK();
    return VKX_OK;
}

// cv::warpAffine's in-place inversion of the forward 2x3 matrix (double).
CoordAffine make_affine(const double Mf[6])
{
    CoordAffine c;
    double M[6];
    for (int i = 0; i < 6; i++) M[i] = Mf[i];
    double D = M[0] * M[4] - M[1] * M[3];
    D = D != 0 ? 1. / D : 0;
    const double A11 = M[4] * D, A22 = M[0] * D;
    M[0] = A11; M[1] *= -D;
    M[3] *= -D; M[4] = A22;
    const double b1 = -M[0] * M[2] - M[1] * M[5];
    const double b2 = -M[3] * M[2] - M[4] * M[5];
    M[2] = b1; M[5] = b2;
    for (int i = 0; i < 6; i++) c.m[i] = M[i];
    return c;
}

// cv::invert of a 3x3 double matrix (cofactors times 1/det; zeros when singular) and the block width of
// WarpPerspectiveInvoker.
CoordPerspective make_perspective(const double S[9], int dh, int dw)
{
    CoordPerspective c;
    auto at = [&](int r, int col) { return S[r * 3 + col]; };
    double d = at(0, 0) * (at(1, 1) * at(2, 2) - at(1, 2) * at(2, 1)) -
               at(0, 1) * (at(1, 0) * at(2, 2) - at(1, 2) * at(2, 0)) +
               at(0, 2) * (at(1, 0) * at(2, 1) - at(1, 1) * at(2, 0));
    if (d == 0.) {
        for (int i = 0; i < 9; i++) c.m[i] = 0;
    } else {
        d = 1. / d;
        c.m[0] = (at(1, 1) * at(2, 2) - at(1, 2) * at(2, 1)) * d;
        c.m[1] = (at(0, 2) * at(2, 1) - at(0, 1) * at(2, 2)) * d;
        c.m[2] = (at(0, 1) * at(1, 2) - at(0, 2) * at(1, 1)) * d;
        c.m[3] = (at(1, 2) * at(2, 0) - at(1, 0) * at(2, 2)) * d;
        c.m[4] = (at(0, 0) * at(2, 2) - at(0, 2) * at(2, 0)) * d;
        c.m[5] = (at(0, 2) * at(1, 0) - at(0, 0) * at(1, 2)) * d;
        c.m[6] = (at(1, 0) * at(2, 1) - at(1, 1) * at(2, 0)) * d;
        c.m[7] = (at(0, 1) * at(2, 0) - at(0, 0) * at(2, 1)) * d;
        c.m[8] = (at(0, 0) * at(1, 1) - at(0, 1) * at(1, 0)) * d;
    }
    const int BLOCK_SZ = 32;
    const int bh0 = BLOCK_SZ / 2 < dh ? BLOCK_SZ / 2 : (dh > 0 ? dh : 1);
    c.bw0 = BLOCK_SZ * BLOCK_SZ / bh0 < dw ? BLOCK_SZ * BLOCK_SZ / bh0 : (dw > 0 ? dw : 1);
    return c;
}

} // namespace

VKX_EXPORT int vkx_remap_u8_dev(vkx_ctx *ctx, const uint8_t *src, int sh, int sw, int cn, ptrdiff_t src_stride,
                                const float *map_x, const float *map_y, ptrdiff_t map_stride_el, uint8_t *dst,
                                int dh, int dw, ptrdiff_t dst_stride)
{
    VKX_REQUIRE(map_x && map_y, "NULL map");
    return launch_u8(ctx, src, sh, sw, cn, src_stride, dst, dh, dw, dst_stride, CoordMap{map_x, map_y, map_stride_el});
}

VKX_EXPORT int vkx_remap_f32_dev(vkx_ctx *ctx, const float *src, int sh, int sw, ptrdiff_t src_stride_el,
                                 const float *map_x, const float *map_y, ptrdiff_t map_stride_el, float *dst, int dh,
                                 int dw, ptrdiff_t dst_stride_el)
{
    VKX_REQUIRE(map_x && map_y, "NULL map");
    return launch_f32(ctx, src, sh, sw, src_stride_el, dst, dh, dw, dst_stride_el,
                      CoordMap{map_x, map_y, map_stride_el});
}

VKX_EXPORT int vkx_remap_multi_dev(vkx_ctx *ctx, const vkx_elem *elems, int n_elems, int sh, int sw, const float *map_x,
                                   const float *map_y, ptrdiff_t map_stride_el, int dh, int dw)
{
    VKX_REQUIRE(ctx && elems && map_x && map_y, "NULL argument");
    VKX_REQUIRE(n_elems >= 1, "no elements");
    const CoordMap map{map_x, map_y, map_stride_el};
    for (int i = 0; i < n_elems; i++) {
        const vkx_elem &e = elems[i];
        VKX_REQUIRE(e.src && e.dst, "NULL element plane");
        int rc;
        if (e.is_f32) {
            VKX_REQUIRE(e.cn == 1, "float32 elements are single channel");
            rc = launch_f32(ctx, (const float *)e.src, sh, sw, e.src_stride, (float *)e.dst, dh, dw, e.dst_stride, map);
        } else {
            rc = launch_u8(ctx, (const uint8_t *)e.src, sh, sw, e.cn, e.src_stride, (uint8_t *)e.dst, dh, dw, e.dst_stride, map);
        }
        if (rc) return rc;
    }
    return VKX_OK;
}

VKX_EXPORT int vkx_warp_affine_u8_dev(vkx_ctx *ctx, const uint8_t *src, int sh, int sw, int cn, ptrdiff_t src_stride,
                                      const double M[6], uint8_t *dst, int dh, int dw, ptrdiff_t dst_stride)
{
    VKX_REQUIRE(M != nullptr, "NULL matrix");
    return launch_u8(ctx, src, sh, sw, cn, src_stride, dst, dh, dw, dst_stride, make_affine(M));
}

VKX_EXPORT int vkx_warp_affine_f32_dev(vkx_ctx *ctx, const float *src, int sh, int sw, ptrdiff_t src_stride_el,
                                       const double M[6], float *dst, int dh, int dw, ptrdiff_t dst_stride_el)
{
    VKX_REQUIRE(M != nullptr, "NULL matrix");
    return launch_f32(ctx, src, sh, sw, src_stride_el, dst, dh, dw, dst_stride_el, make_affine(M));
}

VKX_EXPORT int vkx_warp_perspective_u8_dev(vkx_ctx *ctx, const uint8_t *src, int sh, int sw, int cn,
                                           ptrdiff_t src_stride, const double M[9], uint8_t *dst, int dh, int dw,
                                           ptrdiff_t dst_stride)
{
    VKX_REQUIRE(M != nullptr, "NULL matrix");
    return launch_u8(ctx, src, sh, sw, cn, src_stride, dst, dh, dw, dst_stride, make_perspective(M, dh, dw));
}

VKX_EXPORT int vkx_warp_perspective_f32_dev(vkx_ctx *ctx, const float *src, int sh, int sw, ptrdiff_t src_stride_el,
                                            const double M[9], float *dst, int dh, int dw, ptrdiff_t dst_stride_el)
{
    VKX_REQUIRE(M != nullptr, "NULL matrix");
    return launch_f32(ctx, src, sh, sw, src_stride_el, dst, dh, dw, dst_stride_el, make_perspective(M, dh, dw));
}
