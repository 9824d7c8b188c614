// Photometric members of the distortion chain on gfx950 (per-pixel integer / float32 work, HBM bound).
// Arithmetic specification and reference citations: oracle/vkx_oracle.c.
#include "vkx_internal.h"
#include "vkx_color.h"

#include <algorithm>

#include <float.h>

#include <math.h>
#include <string.h>

namespace {

constexpr int kMaxKsize = 31;
struct BlurKernel {
    uint16_t k[kMaxKsize]; // unsigned 8.8 fixed point, sums to 256
    int kw, kh;
};

// getGaussianKernelBitExact + getGaussianKernelFixedPoint_ED (OpenCV >= 4.2): double kernel, error-diffused
// 8.8 quantisation, centre tap = 256 - 2 * sum(others).
int gaussian_kernel_q8(int n, double sigma, uint16_t *kq)
{
    if (n <= 0 || (n & 1) == 0 || n > kMaxKsize || !(sigma > 0)) return -1;
    const double scale2X = -0.125 / (sigma * sigma);
    const int n2 = (n - 1) / 2;
    double values[kMaxKsize], kd[kMaxKsize];
    double sum = 0;
    for (int i = 0, x = 1 - n; i < n2; i++, x += 2) {
        const double t = exp((double)(x * x) * scale2X);
        values[i] = t;
        sum += t;
    }
    sum *= 2;
    sum += 1;
    const double mul1 = 1. / sum;
    for (int i = 0; i < n2; i++) { kd[i] = values[i] * mul1; }
    double err = 0;
    long long isum = 0;
    for (int i = 0; i < n2; i++) {
        const double adj = kd[i] * 256. + err;
        const long long v0 = (long long)nearbyint(adj);
        err = adj - (double)v0;
        kq[i] = (uint16_t)v0;
        kq[n - 1 - i] = (uint16_t)v0;
        isum += v0;
    }
    kq[n2] = (uint16_t)(256 - 2 * isum);
    return 0;
}

__device__ __forceinline__ int reflect101(int p, int len)
{
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do {
        if (p < 0) p = -p;
        else p = 2 * (len - 1) - p;
    } while ((unsigned)p >= (unsigned)len);
    return p;
}

// Horizontal u8 x 8.8 -> 8.8 then vertical 8.8 x 8.8 -> 16.16, (v + 2^15) >> 16, BORDER_REFLECT_101.
template <int CN>
__global__ void __launch_bounds__(256) k_gaussian_blur(const uint8_t *__restrict__ src, int h, int w, ptrdiff_t sstride,
                                                       uint8_t *__restrict__ dst, ptrdiff_t dstride, BlurKernel K)
{
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x >= w || y >= h) return;
    const int rx = K.kw / 2, ry = K.kh / 2;
    uint32_t acc[CN];
#pragma unroll
    for (int c = 0; c < CN; c++) acc[c] = 0;
    for (int j = 0; j < K.kh; j++) {
        const uint8_t *row = src + (ptrdiff_t)reflect101(y + j - ry, h) * sstride;
        uint32_t hacc[CN];
#pragma unroll
        for (int c = 0; c < CN; c++) hacc[c] = 0;
        for (int i = 0; i < K.kw; i++) {
            const uint8_t *p = row + (ptrdiff_t)reflect101(x + i - rx, w) * CN;
            const uint32_t kx = K.kw == 1 ? 256u : K.k[i];
#pragma unroll
            for (int c = 0; c < CN; c++) hacc[c] += kx * p[c];
        }
        const uint32_t ky = K.kh == 1 ? 256u : K.k[j];
#pragma unroll
        for (int c = 0; c < CN; c++) acc[c] += ky * min(hacc[c], 65535u);
    }
    uint8_t *d = dst + (ptrdiff_t)y * dstride + (ptrdiff_t)x * CN;
#pragma unroll
    for (int c = 0; c < CN; c++) d[c] = (uint8_t)vkd::clamp_u8((int)((acc[c] + 32768u) >> 16));
}

// The same arithmetic as k_gaussian_blur for kernels up to 7 x 7, tiled: a workgroup of 4 wavefronts owns a window of
// 64 columns (lane = column, the middle 64 - 2 rx are outputs) x 32 + 2 ry rows.  Every pixel is loaded once (row
// coalesced), the horizontal pass runs on wavefront shuffles, its 8.8 sums go through LDS, the vertical pass reads
// them at lane stride 1 -- the standalone form of phases D / E of the fused chain kernel.
constexpr int kBlurTileH = 32, kBlurRMax = 3;

template <int CN>
__global__ void __launch_bounds__(256) k_gaussian_blur_tiled(const uint8_t *__restrict__ src, int h, int w, ptrdiff_t sstride,
                                                             uint8_t *__restrict__ dst, ptrdiff_t dstride, BlurKernel K)
{
    __shared__ uint16_t hs[(kBlurTileH + 2 * kBlurRMax) * 64 * CN];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = K.kw / 2;                       // kw == kh here
    const int tw = 64 - 2 * r;
    const int x0 = blockIdx.x * tw, y0 = blockIdx.y * kBlurTileH;
    const int gx = reflect101(x0 - r + lane, w);   // the column this lane holds during the horizontal pass
    const int rows = min(kBlurTileH, h - y0) + 2 * r;
    uint32_t kq[2 * kBlurRMax + 1];
#pragma unroll
    for (int i = 0; i < 2 * kBlurRMax + 1; i++) kq[i] = i < K.kw ? K.k[i] : 0;
    for (int row = wave; row < rows; row += 4) {
        const uint8_t *p = src + (ptrdiff_t)reflect101(y0 - r + row, h) * sstride + (ptrdiff_t)gx * CN;
        uint32_t px[CN], acc[CN];
#pragma unroll
        for (int c = 0; c < CN; c++) { px[c] = p[c]; acc[c] = 0; }
#pragma unroll
        for (int i = 0; i < 2 * kBlurRMax + 1; i++) {
            if (i < K.kw) {
                const int from = min(max(lane + i - r, 0), 63);
#pragma unroll
                for (int c = 0; c < CN; c++) acc[c] += kq[i] * (uint32_t)__shfl((int)px[c], from, 64);
            }
        }
#pragma unroll
        for (int c = 0; c < CN; c++) hs[(row * 64 + lane) * CN + c] = (uint16_t)min(acc[c], 65535u);
    }
    __syncthreads();
    const int ox = lane - r, x = x0 + ox;
    if (ox < 0 || ox >= tw || x >= w) return;
    for (int orow = wave; orow < rows - 2 * r; orow += 4) {
        uint32_t acc[CN];
#pragma unroll
        for (int c = 0; c < CN; c++) acc[c] = 0;
#pragma unroll
        for (int j = 0; j < 2 * kBlurRMax + 1; j++) {
            if (j < K.kh) {
#pragma unroll
                for (int c = 0; c < CN; c++) acc[c] += kq[j] * hs[((orow + j) * 64 + lane) * CN + c];
            }
        }
        uint8_t *d = dst + (ptrdiff_t)(y0 + orow) * dstride + (ptrdiff_t)x * CN;
#pragma unroll
        for (int c = 0; c < CN; c++) d[c] = (uint8_t)vkd::clamp_u8((int)((acc[c] + 32768u) >> 16));
    }
}

// RGB form of the tiled blur with the packing of the fused chain kernel (fused.hip, phases D / E): r and b ride in the two
// 16-bit halves of one dword through the horizontal pass (255 * 256 < 2^16: the halves never carry), so a tap is two
// ds_bpermute on byte addresses computed once + two 24-bit multiply-adds instead of three __shfl (which rebuild their index
// math on every call) + three multiplies; four neighbouring lanes store their 12 bytes as three dwords.
typedef uint32_t blur_u32_u1 __attribute__((aligned(1)));

template <int R>
__global__ void __launch_bounds__(256) k_gaussian_blur_rgb(const uint8_t *__restrict__ src, int h, int w, ptrdiff_t sstride,
                                                           uint8_t *__restrict__ dst, ptrdiff_t dstride, BlurKernel K)
{
    // Round 4: no LDS, no barrier.  A wavefront owns 8 output rows of a 64-column window (lane = column, the middle 64 - 2 R are
    // outputs): it loads the 8 + 2 R rows it needs -- one unaligned dword per pixel --, runs their horizontal passes with
    // whole-wavefront DPP shifts (the LDS pipe, which ds_bpermute taps and the staging of the sums kept busy, was the bound of the
    // round-2 kernel), keeps the 8.8 sums of consecutive row PAIRS packed per channel (row | next row << 16) in registers, and takes
    // the vertical pass two taps at a time with v_dot2_u32_u16 -- the register form of phases D / E of the fused chain kernel.
    constexpr int KS = 2 * R + 1, TW = 64 - 2 * R, NR = 8 + 2 * R, NP = NR / 2;
    static_assert(NR % 2 == 0, "whole row pairs");
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * kBlurTileH + 8 * wave;
    if (y0 >= h) return;
    const int gx = reflect101(x0 - R + lane, w);
    const bool last = gx == w - 1;          // the image's last column reads the dword that ENDS with its pixel
    uint32_t kq[KS];
#pragma unroll
    for (int i = 0; i < KS; i++) kq[i] = K.k[i];
    int ry[NR];                                      // wavefront-uniform: the reflection loop only runs on the image's first / last bands
    if (y0 - R >= 0 && y0 + 8 + R <= h) {
#pragma unroll
        for (int r = 0; r < NR; r++) ry[r] = y0 - R + r;
    } else {
#pragma unroll
        for (int r = 0; r < NR; r++) ry[r] = reflect101(y0 - R + r, h);
    }
    const uint32_t voff = (uint32_t)gx * 3u - (last ? 1u : 0u);       // scalar row base + 32-bit lane offset: saddr loads
    uint32_t px[NR];
#pragma unroll
    for (int r = 0; r < NR; r++) px[r] = *(const blur_u32_u1 *)(src + (ptrdiff_t)ry[r] * sstride + voff);
    uint32_t pr[NP], pg[NP], pb[NP];                 // per row pair: (row 2 j | row 2 j + 1 << 16) of the horizontal sums
#pragma unroll
    for (int j = 0; j < NP; j++) {
        uint32_t arb[2], ag[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const uint32_t v = last ? px[2 * j + u] >> 8 : px[2 * j + u];
            const uint32_t rb = v & 0x00ff00ffu, g = (v >> 8) & 0xffu;
            uint32_t srb = __umul24(kq[R], rb), sg = __umul24(kq[R], g);      // both halves at once: k <= 256, each half <= 255
            uint32_t lrb = rb, lg = g, rrb = rb, rg = g;
#pragma unroll
            for (int d = 1; d <= R; d++) {
                lrb = (uint32_t)__builtin_amdgcn_mov_dpp((int)lrb, 0x138 /* wave_shr:1: from lane - 1; lane 0 reads 0 */, 0xf, 0xf, true);
                lg = (uint32_t)__builtin_amdgcn_mov_dpp((int)lg, 0x138, 0xf, 0xf, true);
                rrb = (uint32_t)__builtin_amdgcn_mov_dpp((int)rrb, 0x130 /* wave_shl:1: from lane + 1; lane 63 reads 0 */, 0xf, 0xf, true);
                rg = (uint32_t)__builtin_amdgcn_mov_dpp((int)rg, 0x130, 0xf, 0xf, true);
                srb += __umul24(kq[R - d], lrb) + __umul24(kq[R + d], rrb);
                sg += __umul24(kq[R - d], lg) + __umul24(kq[R + d], rg);
            }
            arb[u] = srb; ag[u] = sg;                // lanes the shifts ran dry on (0 .. R - 1, 64 - R .. 63) are halo columns
        }
        pr[j] = __builtin_amdgcn_perm(arb[1], arb[0], 0x05040100u);     // low halves: r of both rows
        pb[j] = __builtin_amdgcn_perm(arb[1], arb[0], 0x07060302u);     // high halves: b
        pg[j] = ag[0] | (ag[1] << 16);
    }
    const int ox = lane - R, x = x0 + ox;
    const bool ocol = ox >= 0 && ox < TW && x < w;
    const int full4 = x0 + ((min(TW, w - x0) >> 2) << 2);    // columns of this tile covered by whole 4-pixel groups (TW % 4 == 2 for R = 1, 3)
    // 12-byte groups start at the tile's first output column (lane R): m = group phase of this lane
    const int m = ox & 3;
    const bool grouped = ocol && x < full4, gstore = grouped && m < 3, bstore = ocol && !grouped;
    const uint32_t goff = (uint32_t)(x0 * 3 + (ox >> 2) * 12 + m * 4), boff = (uint32_t)x * 3u;
    const uint32_t sh0 = 8 * m, sh1 = 24 - 8 * m;
#pragma unroll
    for (int t = 0; t < 8; t++) {
        // taps kq[0 .. KS) sit on rows t .. t + 2 R of the loaded ones: pair (t + i) >> 1, half (t + i) & 1
        uint32_t ar = 32768u, ag = 32768u, ab = 32768u;
#pragma unroll
        for (int q = 0; q <= R; q++) {
            const int i0 = 2 * q - (t & 1), i1 = i0 + 1;             // tap indices of the pair's low / high half
            const uint32_t wq = (i0 >= 0 && i0 < KS ? kq[i0 < 0 ? 0 : i0] : 0u) | ((i1 < KS ? kq[i1 < KS ? i1 : 0] : 0u) << 16);
            const int j = (t >> 1) + q;
            ar = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, wq), __builtin_bit_cast(us2, pr[j]), ar, false);
            ag = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, wq), __builtin_bit_cast(us2, pg[j]), ag, false);
            ab = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, wq), __builtin_bit_cast(us2, pb[j]), ab, false);
        }
        const uint32_t P = (ar >> 16) | ((ag >> 16) << 8) | (ab & 0xff0000u);     // sums stay below 2^24: no clamp needed
        const uint32_t Pn = (uint32_t)__builtin_amdgcn_mov_dpp((int)P, 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
        if (y0 + t >= h) break;                      // wavefront-uniform
        uint8_t *drow = dst + (ptrdiff_t)(y0 + t) * dstride;
        if (gstore) *(blur_u32_u1 *)(drow + goff) = (P >> sh0) | (Pn << sh1);
        if (bstore) {
            uint8_t *d = drow + boff;
            d[0] = (uint8_t)P; d[1] = (uint8_t)(P >> 8); d[2] = (uint8_t)(P >> 16);
        }
    }
}

// ---- cv.filter2D(uint8 image, -1, float32 kernel) (defocus_blur / motion_blur, photometric/blur.py:85-192) -------
// Correlation anchored at the kernel centre, BORDER_REFLECT_101; per pixel the non-zero taps in row-major order,
// s += k * float(px) with separate roundings, then cvRound + saturate.  One workgroup = a 64 x 16 output tile: the
// tile plus its halo is staged in LDS once (reflection resolved while staging), every lane walks the taps out of LDS.
constexpr int kF2dMaxK = 15;                       // kernels up to 15 x 15
constexpr int kF2dTileW = 64, kF2dTileH = 16;
struct F2dKernel {
    int kh, kw;
    float k[kF2dMaxK * kF2dMaxK];
};

template <int CN>
__global__ void __launch_bounds__(256) k_filter2d_u8(const uint8_t *__restrict__ src, int h, int w, ptrdiff_t sstride,
                                                     uint8_t *__restrict__ dst, ptrdiff_t dstride, F2dKernel K)
{
    __shared__ uint8_t tile[(kF2dTileH + kF2dMaxK - 1) * (kF2dTileW + kF2dMaxK - 1) * CN];
    const int ay = K.kh / 2, ax = K.kw / 2;
    const int x0 = blockIdx.x * kF2dTileW, y0 = blockIdx.y * kF2dTileH;
    const int tw = kF2dTileW + K.kw - 1, th = kF2dTileH + K.kh - 1;
    for (int i = threadIdx.x; i < tw * th; i += 256) {
        const int ty = i / tw, tx = i - ty * tw;
        const uint8_t *p = src + (ptrdiff_t)reflect101(y0 + ty - ay, h) * sstride + (ptrdiff_t)reflect101(x0 + tx - ax, w) * CN;
#pragma unroll
        for (int c = 0; c < CN; c++) tile[i * CN + c] = p[c];
    }
    __syncthreads();
    const int lx = threadIdx.x & 63, x = x0 + lx;
    if (x >= w) return;
    for (int ly = threadIdx.x >> 6; ly < kF2dTileH; ly += 4) {
        const int y = y0 + ly;
        if (y >= h) break;
        float acc[CN];
#pragma unroll
        for (int c = 0; c < CN; c++) acc[c] = 0.f;
        for (int ky = 0; ky < K.kh; ky++)
            for (int kx = 0; kx < K.kw; kx++) {
                const float f = K.k[ky * K.kw + kx];
                if (f == 0) continue;                       // uniform: the reference drops zero taps from its list
                const uint8_t *t = tile + ((ly + ky) * tw + lx + kx) * CN;
#pragma unroll
                for (int c = 0; c < CN; c++) {
                    const float term = f * (float)t[c];
                    acc[c] = acc[c] + term;
                }
            }
        uint8_t *d = dst + (ptrdiff_t)y * dstride + (ptrdiff_t)x * CN;
#pragma unroll
        for (int c = 0; c < CN; c++) d[c] = (uint8_t)vkd::clamp_u8(vkd::cv_round(acc[c]));
    }
}

// ---- cvtColor RGB <-> HSV_FULL (uint8) -----------------------------------------------------------------------
// RGB->HSV is OpenCV's integer LUT division; the two 256-entry tables (cvRound of doubles) are built on the
// host once and passed in device memory.
struct HsvTables {
    int sdiv[256];
    int hdiv[256];
};

__device__ __forceinline__ void rgb2hsv_px(const int *__restrict__ sdiv, const int *__restrict__ hdiv, int r, int g, int b,
                                           int &H, int &S, int &V)
{
    vkd::rgb2hsv_full(sdiv, hdiv, r, g, b, H, S, V);
}

__device__ __forceinline__ void hsv2rgb_px(int H, int S, int V, int &r, int &g, int &b) { vkd::hsv2rgb_full(H, S, V, r, g, b); }

// mode 0: color_shift (RGB -> HSV, H += delta mod 256, HSV -> RGB); 1: RGB -> HSV; 2: HSV -> RGB
template <int MODE>
__global__ void __launch_bounds__(256) k_hsv(const uint8_t *__restrict__ src, int h, int w, ptrdiff_t sstride,
                                             uint8_t *__restrict__ dst, ptrdiff_t dstride, int delta,
                                             const HsvTables *__restrict__ T)
{
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x >= w || y >= h) return;
    const uint8_t *s = src + (ptrdiff_t)y * sstride + (ptrdiff_t)x * 3;
    uint8_t *d = dst + (ptrdiff_t)y * dstride + (ptrdiff_t)x * 3;
    int a = s[0], b = s[1], c = s[2];
    if (MODE == 0 || MODE == 1) {
        int H, S, V;
        rgb2hsv_px(T->sdiv, T->hdiv, a, b, c, H, S, V);
        a = H; b = S; c = V;
    }
    if (MODE == 0) {
        int hh = (a + delta) % 256;
        if (hh < 0) hh += 256;
        a = hh;
    }
    if (MODE == 0 || MODE == 2) {
        int r, g, bl;
        hsv2rgb_px(a, b, c, r, g, bl);
        a = r; b = g; c = bl;
    }
    d[0] = (uint8_t)a; d[1] = (uint8_t)b; d[2] = (uint8_t)c;
}

// RGB2HLS_f / HLS2RGB_f through the 8-bit wrappers (color_hsv.simd.hpp, scalar path), hrange 256, no FMA.
// a / b correctly rounded for NORMAL operands whose quotient is normal too (here: 1/255 <= b <= 2, 0 < a <= 60): the core of the
// sequence the compiler emits for an IEEE float32 division -- reciprocal, one Newton step, quotient, two residual corrections --
// without the operand scaling and the special-case fix-up, which these ranges never need (tests/test_gpu_hls_exhaustive.py runs all
// 2^24 colours against the oracle's plain division).
__device__ __forceinline__ float div_normal(float a, float b)
{
    float y = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, y, 1.0f);
    y = __builtin_fmaf(e, y, y);
    float q = a * y;
    const float r = __builtin_fmaf(-b, q, a);
    q = __builtin_fmaf(r, y, q);
    const float r2 = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(r2, y, q);
}

// cvRound of a value known to lie within int32: round half to even
__device__ __forceinline__ int round_small(float v) { return __float2int_rn(v); }

__device__ __forceinline__ void rgb2hls_px(int R, int G, int B, int &H, int &L, int &S)
{
    const float r = R * (1.f / 255.f), g = G * (1.f / 255.f), b = B * (1.f / 255.f);
    float h = 0.f, s = 0.f;
    const float vmax = fmaxf(r, fmaxf(g, b)), vmin = fminf(r, fminf(g, b));
    float diff = vmax - vmin;
    const float sum = vmax + vmin;
    const float l = sum * 0.5f;
    if (diff > FLT_EPSILON) {
        // one division serves both branches of s: the denominator is selected first (2 - vmax - vmin rounds as the reference's)
        const float den = l < 0.5f ? sum : (2 - vmax - vmin);
        s = div_normal(diff, den);
        diff = div_normal(60.f, diff);
        if (vmax == r) h = (g - b) * diff;
        else if (vmax == g) h = (b - r) * diff + 120.f;
        else h = (r - g) * diff + 240.f;
        if (h < 0.f) h += 360.f;
    }
    const float hscale = 256.f / 360.f;
    H = min(round_small(h * hscale), 255);      // 0 <= h < 360 + ulps: only the upper clamp can act
    L = round_small(l * 255.f);                 // 0 <= l, s <= 1 (s up to an ulp above): the rounded product is 0 .. 255
    S = min(round_small(s * 255.f), 255);
}

__device__ __forceinline__ void hls2rgb_px(int H, int L, int S, int &R, int &G, int &B)
{
    float h = (float)H;
    const float l = L * (1.f / 255.f), s = S * (1.f / 255.f);
    float b = l, g = l, r = l;
    if (s != 0) {
        const float hscale = 6.f / 256.f;
        const float p2 = l <= 0.5f ? l * (1 + s) : l + s - l * s;
        const float p1 = 2 * l - p2;
        h *= hscale;               // 0 <= h < 6 for every 8-bit hue
        const int sector = (int)h; // h >= 0: truncation is the floor
        h -= (float)sector;
        // sector_data (b, g, r), the table of the HSV inverse: 0 (t1,t3,t0) 1 (t1,t0,t2) 2 (t3,t0,t1) 3 (t0,t2,t1) 4 (t0,t1,t3) 5 (t2,t1,t0)
        // = (t1, odd ? t0 : t3, odd ? t2 : t0) rotated by sector / 2; t2 = p1 + (p2 - p1)(1 - h) serves the odd sectors only, t3 the even
        const bool odd = sector & 1;
        const int rot = sector >> 1;
        const float t0 = p2, t1 = p1, tt = p1 + (p2 - p1) * (odd ? 1 - h : h);
        const float u0 = t1, u1 = odd ? t0 : tt, u2 = odd ? tt : t0;
        b = rot == 0 ? u0 : (rot == 1 ? u1 : u2);
        g = rot == 0 ? u1 : (rot == 1 ? u2 : u0);
        r = rot == 0 ? u2 : (rot == 1 ? u0 : u1);
    }
    // the channels lie in [0, 1] up to a few ulps: the rounded product needs the clamp only in name
    R = vkd::clamp_u8(round_small(r * 255.f));
    G = vkd::clamp_u8(round_small(g * 255.f));
    B = vkd::clamp_u8(round_small(b * 255.f));
}

// RGB2Gray<uchar>: 15-bit fixed point
__device__ __forceinline__ int rgb2gray_px(int R, int G, int B) { return (R * 9798 + G * 19235 + B * 3735 + (1 << 14)) >> 15; }

// mode 0: brightness_shift (RGB -> HLS, L = clip(L + delta), HLS -> RGB); 1: RGB -> HLS; 2: HLS -> RGB;
//      3: RGB -> GRAY (1 channel out); 4: GRAY -> RGB; 5: color_balance (w0 * gray + w1 * px, clip, truncate)
template <int MODE>
__global__ void __launch_bounds__(256) k_cvt(const uint8_t *__restrict__ src, int h, int w, ptrdiff_t sstride,
                                             uint8_t *__restrict__ dst, ptrdiff_t dstride, int delta, float w0, float w1)
{
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x >= w || y >= h) return;
    constexpr int SCN = MODE == 4 ? 1 : 3, DCN = MODE == 3 ? 1 : 3;
    const uint8_t *s = src + (ptrdiff_t)y * sstride + (ptrdiff_t)x * SCN;
    uint8_t *d = dst + (ptrdiff_t)y * dstride + (ptrdiff_t)x * DCN;
    if (MODE == 4) {
        const uint8_t v = s[0];
        d[0] = v; d[1] = v; d[2] = v;
        return;
    }
    int a = s[0], b = s[1], c = s[2];
    if (MODE == 3) {
        d[0] = (uint8_t)rgb2gray_px(a, b, c);
        return;
    }
    if (MODE == 5) {
        const float gray = (float)rgb2gray_px(a, b, c);
        const int in[3] = {a, b, c};
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float t0 = w0 * gray, t1 = w1 * (float)in[k];
            float v = t0 + t1;
            v = v < 0.f ? 0.f : (v > 255.f ? 255.f : v);
            d[k] = (uint8_t)v;
        }
        return;
    }
    if (MODE == 0 || MODE == 1) {
        int H, L, S;
        rgb2hls_px(a, b, c, H, L, S);
        a = H; b = L; c = S;
    }
    if (MODE == 0 && delta != 0) b = vkd::clamp_u8(b + delta);
    if (MODE == 0 || MODE == 2) {
        int R, G, B;
        hls2rgb_px(a, b, c, R, G, B);
        a = R; b = G; c = B;
    }
    d[0] = (uint8_t)a; d[1] = (uint8_t)b; d[2] = (uint8_t)c;
}

// RGBA conversions of Image.to_target_mode_image (element/image.py:188-216): 6 RGBA -> RGB (alpha dropped), 7 RGB -> RGBA
// (alpha 255), 8 GRAY -> RGBA, 9 RGBA -> GRAY (the RGB2Gray weights on the first three channels)
template <int MODE>
__global__ void __launch_bounds__(256) k_cvt_alpha(const uint8_t *__restrict__ src, int h, int w, ptrdiff_t sstride,
                                                   uint8_t *__restrict__ dst, ptrdiff_t dstride)
{
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x >= w || y >= h) return;
    constexpr int SCN = MODE == 6 || MODE == 9 ? 4 : (MODE == 7 ? 3 : 1), DCN = MODE == 6 ? 3 : (MODE == 9 ? 1 : 4);
    const uint8_t *s = src + (ptrdiff_t)y * sstride + (ptrdiff_t)x * SCN;
    uint8_t *d = dst + (ptrdiff_t)y * dstride + (ptrdiff_t)x * DCN;
    if (MODE == 6) { d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; }
    else if (MODE == 7) { d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = 255; }
    else if (MODE == 8) { const uint8_t v = s[0]; d[0] = v; d[1] = v; d[2] = v; d[3] = 255; }
    else d[0] = (uint8_t)rgb2gray_px(s[0], s[1], s[2]);
}

// uint8(clip(w0 * a + w1 * b, 0, 255)) on the channels of `chmask` (0 = all), b copied elsewhere: color_balance on any
// image mode (photometric/color.py:371-396: float32 products rounded separately, clip, truncation)
__global__ void __launch_bounds__(256) k_blend_u8(const uint8_t *__restrict__ a, ptrdiff_t astride, const uint8_t *__restrict__ b,
                                                  ptrdiff_t bstride, int h, int wc, int cn, float w0, float w1, unsigned chmask,
                                                  uint8_t *__restrict__ dst, ptrdiff_t dstride)
{
    const int xe = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (xe >= wc || y >= h) return;
    const int c = xe % cn;
    const uint8_t vb = b[(ptrdiff_t)y * bstride + xe];
    uint8_t out = vb;
    if (chmask == 0 || ((chmask >> c) & 1u)) {
        const float t0 = w0 * (float)a[(ptrdiff_t)y * astride + xe], t1 = w1 * (float)vb;
        float v = t0 + t1;
        v = v < 0.f ? 0.f : (v > 255.f ? 255.f : v);
        out = (uint8_t)v;
    }
    dst[(ptrdiff_t)y * dstride + xe] = out;
}

// uint8(clip((1 - m) * px + m * fog, 0, 255)) with a float32 weight plane m [h, w] shared by the cn channels and one
// float32 fog value per channel: fog on GRAYSCALE images, whose fog value is fractional (photometric/effect.py:194-197)
__global__ void __launch_bounds__(256) k_fog_f32(const uint8_t *__restrict__ src, ptrdiff_t sstride, const float *__restrict__ m,
                                                 ptrdiff_t mstride, int h, int w, int cn, float f0, float f1, float f2, float f3,
                                                 uint8_t *__restrict__ dst, ptrdiff_t dstride)
{
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x >= w || y >= h) return;
    const float mk = m[(ptrdiff_t)y * mstride + x], inv = 1.0f - mk;
    const float fog[4] = {f0, f1, f2, f3};
    for (int c = 0; c < cn; c++) {
        const float t0 = inv * (float)src[(ptrdiff_t)y * sstride + (ptrdiff_t)x * cn + c], t1 = mk * fog[c];
        float v = t0 + t1;
        v = v < 0.f ? 0.f : (v > 255.f ? 255.f : v);
        dst[(ptrdiff_t)y * dstride + (ptrdiff_t)x * cn + c] = (uint8_t)v;
    }
}

__global__ void __launch_bounds__(256) k_mean_shift(const uint8_t *__restrict__ src, int h, int w, int cn, ptrdiff_t sstride,
                                                    uint8_t *__restrict__ dst, ptrdiff_t dstride, int delta, int has_thr,
                                                    int thr, int cycle, unsigned chmask)
{
    const int xe = blockIdx.x * 64 + threadIdx.x; // element index within the row
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (xe >= w * cn || y >= h) return;
    const int c = xe % cn;
    int v = src[(ptrdiff_t)y * sstride + xe];
    if ((chmask == 0 || ((chmask >> c) & 1u)) && delta != 0) {
        bool apply = true;
        if (has_thr) apply = delta > 0 ? (v <= thr) : (thr <= v);
        if (apply) v += delta;
        if (cycle) { v %= 256; if (v < 0) v += 256; }
        else v = vkd::clamp_u8(v);
    }
    dst[(ptrdiff_t)y * dstride + xe] = (uint8_t)v;
}

__global__ void __launch_bounds__(256) k_add_noise(const uint8_t *__restrict__ src, int h, int wc, ptrdiff_t sstride,
                                                   const int16_t *__restrict__ noise, ptrdiff_t nstride,
                                                   uint8_t *__restrict__ dst, ptrdiff_t dstride)
{
    const int xe = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (xe >= wc || y >= h) return;
    const int v = (int16_t)((int16_t)src[(ptrdiff_t)y * sstride + xe] + noise[(ptrdiff_t)y * nstride + xe]);
    dst[(ptrdiff_t)y * dstride + xe] = (uint8_t)vkd::clamp_u8(v);
}

// complement / posterization / channel_permutation (photometric/color.py:299-357, 423-432): one value per lane.
__global__ void __launch_bounds__(256) k_pointwise(const uint8_t *__restrict__ src, int h, int w, int cn, ptrdiff_t sstride,
                                                   uint8_t *__restrict__ dst, ptrdiff_t dstride, int op, int p0, int p1,
                                                   unsigned chmask)
{
    const int xe = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (xe >= w * cn || y >= h) return;
    const int c = xe % cn;
    const uint8_t *row = src + (ptrdiff_t)y * sstride;
    int v = row[xe];
    const bool on = chmask == 0 || ((chmask >> c) & 1u);
    if (op == VKX_POINT_COMPLEMENT) {
        if (on && (p0 < 0 || (p1 ? v <= p0 : p0 <= v))) v = 255 - v;
    } else if (op == VKX_POINT_POSTERIZE) {
        if (on) v &= (0xFF >> p0) << p0;
    } else {                                     // VKX_POINT_PERMUTE: out[c] = in[perm[c]], 2 bits per channel
        v = row[xe - c + ((p0 >> (2 * c)) & 3)];
    }
    dst[(ptrdiff_t)y * dstride + xe] = (uint8_t)v;
}

// impulse_noise (photometric/noise.py:125-150): per-PIXEL selector 0 keep / 1 salt (255) / 2 pepper (0).
__global__ void __launch_bounds__(256) k_impulse_noise(const uint8_t *__restrict__ src, int h, int w, int cn, ptrdiff_t sstride,
                                                       const uint8_t *__restrict__ sel, ptrdiff_t sel_stride,
                                                       uint8_t *__restrict__ dst, ptrdiff_t dstride)
{
    const int xe = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (xe >= w * cn || y >= h) return;
    const int m = sel[(ptrdiff_t)y * sel_stride + xe / cn];
    const uint8_t v = src[(ptrdiff_t)y * sstride + xe];
    dst[(ptrdiff_t)y * dstride + xe] = m == 1 ? (uint8_t)255 : (m == 2 ? (uint8_t)0 : v);
}

// speckle_noise (photometric/noise.py:172-183): float32(px) + float32(px) * float64 noise is a float64 expression
// in numpy; product and sum round separately; clip to [0, 255]; astype(uint8) truncates.
__global__ void __launch_bounds__(256) k_speckle_noise(const uint8_t *__restrict__ src, int h, int wc, ptrdiff_t sstride,
                                                       const double *__restrict__ noise, ptrdiff_t nstride,
                                                       uint8_t *__restrict__ dst, ptrdiff_t dstride)
{
    const int xe = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (xe >= wc || y >= h) return;
    const double m = (double)src[(ptrdiff_t)y * sstride + xe];
    const double t = m * noise[(ptrdiff_t)y * nstride + xe];
    double v = m + t;
    v = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v);   // np.clip; NaN (never produced by rng.normal) would pass through
    dst[(ptrdiff_t)y * dstride + xe] = (uint8_t)v;
}

// Per-channel histogram (the reduction half of boundary_equalization / histogram_equalization, photometric/color.py:
// 214-285): a workgroup accumulates into LDS counters and flushes once.  Integer counts: exact in any order.
__global__ void __launch_bounds__(256) k_histogram(const uint8_t *__restrict__ src, int h, int w, int cn, ptrdiff_t sstride,
                                                   int *__restrict__ hist /* [cn][256] */)
{
    __shared__ int lh[4 * 256];
    for (int i = threadIdx.x; i < cn * 256; i += 256) lh[i] = 0;
    __syncthreads();
    const int wc = w * cn;
    for (int y = blockIdx.y; y < h; y += gridDim.y) {
        const uint8_t *row = src + (ptrdiff_t)y * sstride;
        for (int xe = blockIdx.x * 256 + threadIdx.x; xe < wc; xe += gridDim.x * 256)
            atomicAdd(&lh[(xe % cn) * 256 + row[xe]], 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < cn * 256; i += 256)
        if (lh[i]) atomicAdd(&hist[i], lh[i]);
}

// dst[c] = lut[c][src[c]] on the selected channels (the per-value half of the two equalisations).
__global__ void __launch_bounds__(256) k_apply_lut(const uint8_t *__restrict__ src, int h, int w, int cn, ptrdiff_t sstride,
                                                   uint8_t *__restrict__ dst, ptrdiff_t dstride,
                                                   const uint8_t *__restrict__ lut /* [cn][256] */, unsigned chmask)
{
    const int xe = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (xe >= w * cn || y >= h) return;
    const int c = xe % cn;
    const uint8_t v = src[(ptrdiff_t)y * sstride + xe];
    dst[(ptrdiff_t)y * dstride + xe] = (chmask == 0 || ((chmask >> c) & 1u)) ? lut[c * 256 + v] : v;
}

// mat[pos_y, pos_x] (numpy advanced indexing with two index planes): the pixel shuffle of glass_blur,
// photometric/blur.py:204-250.  Indices are validated on the device: an out-of-range entry raises the flag.
template <int CN>
__global__ void __launch_bounds__(256) k_gather(const uint8_t *__restrict__ src, int sh, int sw, ptrdiff_t sstride,
                                                const int32_t *__restrict__ pos_y, const int32_t *__restrict__ pos_x,
                                                ptrdiff_t pstride, uint8_t *__restrict__ dst, int dh, int dw,
                                                ptrdiff_t dstride, int *__restrict__ bad)
{
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x >= dw || y >= dh) return;
    const int sy = pos_y[(ptrdiff_t)y * pstride + x], sx = pos_x[(ptrdiff_t)y * pstride + x];
    uint8_t *d = dst + (ptrdiff_t)y * dstride + (ptrdiff_t)x * CN;
    if ((unsigned)sy >= (unsigned)sh || (unsigned)sx >= (unsigned)sw) {
        atomicExch(bad, 1);
#pragma unroll
        for (int c = 0; c < CN; c++) d[c] = 0;
        return;
    }
    const uint8_t *p = src + (ptrdiff_t)sy * sstride + (ptrdiff_t)sx * CN;
#pragma unroll
    for (int c = 0; c < CN; c++) d[c] = p[c];
}

// np.clip(int64 samples, 0, 255).astype(uint8): the tail of poisson_noise (photometric/noise.py:81-91); the samples
// themselves are rng.poisson draws of the caller's numpy Generator.
__global__ void __launch_bounds__(256) k_saturate_i64(const long long *__restrict__ src, size_t n, uint8_t *__restrict__ dst)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long v = src[i];
    dst[i] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// fill_np_array blend of one value: trunc(fl32(fl32(1 - a) * dst) + fl32(a * val)), products rounded separately.
__device__ __forceinline__ uint8_t blend_u8(uint8_t d, uint8_t v, float w1)
{
    const float w0 = 1.0f - w1;
    const float t0 = w0 * (float)d;
    const float t1 = w1 * (float)v;
    const float s = t0 + t1;
    return (uint8_t)s;
}

struct StreakParams {
    int thickness, step, dash_step, dash_gap, dash;
    int enable_vert, enable_hori;
    int copy;       // alpha == 1.0: plain masked copy
    float alpha;
    uint8_t color[4];
};

constexpr int kStreakRows = 16;     // rows a wavefront walks: the column's stripe phases are computed once

template <int CN>
__global__ void __launch_bounds__(256) k_line_streak(uint8_t *img, int h, int w, ptrdiff_t stride, StreakParams P)
{
    // The stripes cover a fraction thickness / (thickness + gap) of the columns and of the rows: a wavefront walks
    // kStreakRows rows of its 64 columns, the column's phase in the stripe / dash periods is a per-lane constant, the row's
    // is uniform over the wavefront, and rows / lanes outside every stripe cost a compare each.
    const int x = blockIdx.x * 64 + threadIdx.x;
    if (x >= w) return;
    const bool col_v = P.enable_vert && (x % P.step) < P.thickness;          // the column lies in a vertical stripe
    const bool col_gap = P.dash && (x % P.dash_step) < P.dash_gap;           // ... in a dash gap of the horizontal ones
    const int y0 = (blockIdx.y * 4 + threadIdx.y) * kStreakRows;
    uint8_t *d = img + (ptrdiff_t)y0 * stride + (ptrdiff_t)x * CN;
    for (int r = 0; r < kStreakRows; r++, d += stride) {
        const int y = y0 + r;
        if (y >= h) break;
        bool mv = col_v, mh = P.enable_hori && (y % P.step) < P.thickness;
        if (P.dash) {
            if ((y % P.dash_step) < P.dash_gap) mv = false;
            if (col_gap) mh = false;
        }
        if (!mv && !mh) continue;
#pragma unroll
        for (int c = 0; c < CN; c++) {
            uint8_t v = d[c];
            // vertical stripes first, then horizontal ones: crossings are blended twice (streak.py:96-99)
            if (mv) v = P.copy ? P.color[c] : blend_u8(v, P.color[c], P.alpha);
            if (mh) v = P.copy ? P.color[c] : blend_u8(v, P.color[c], P.alpha);
            d[c] = v;
        }
    }
}

// ---- dense planes: 16 bytes per lane ---------------------------------------------------------------------------------
// The kernels above take any row pitch and move one byte (or one pixel) per lane: at 8192^2 they reach 1.0 - 1.6 TB/s,
// bound by the number of memory instructions.  A dense plane (row pitch == row bytes, 16-byte aligned base pointers) is one
// flat byte string: a lane takes one aligned 16-byte group of it, so a wavefront moves 1 KiB per load.  Same arithmetic
// per byte.  The channel of byte j of group t is (16 t + j) mod cn.
__device__ __forceinline__ int first_channel16(size_t t, int cn) { return cn == 3 ? (int)(t % 3) : 0; }   // 16 = 1 mod 3; 1, 2, 4 divide 16

template <class F>
__device__ __forceinline__ void map_bytes16(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, size_t n, int cn, F f)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x, i0 = t * 16;
    if (i0 >= n) return;
    int c = first_channel16(t, cn);
    if (i0 + 16 <= n) {
        const uint4 in = *(const uint4 *)(src + i0);
        uint32_t w[4] = {in.x, in.y, in.z, in.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t o = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                o |= (uint32_t)(f((int)((w[k] >> (8 * b)) & 0xffu), c, 4 * k + b) & 0xff) << (8 * b);
                c = c + 1 == cn ? 0 : c + 1;
            }
            w[k] = o;
        }
        *(uint4 *)(dst + i0) = make_uint4(w[0], w[1], w[2], w[3]);
    } else {
        for (int j = 0; i0 + j < n; j++) {
            dst[i0 + j] = (uint8_t)f((int)src[i0 + j], c, j);
            c = c + 1 == cn ? 0 : c + 1;
        }
    }
}

__global__ void __launch_bounds__(256) k_mean_shift16(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, size_t n, int cn,
                                                      int delta, int has_thr, int thr, int cycle, unsigned chmask)
{
    map_bytes16(src, dst, n, cn, [=](int v, int c, int) {
        if ((chmask == 0 || ((chmask >> c) & 1u)) && delta != 0) {
            bool apply = true;
            if (has_thr) apply = delta > 0 ? (v <= thr) : (thr <= v);
            if (apply) v += delta;
            v = cycle ? (v & 255) : vkd::clamp_u8(v);      // python's % 256 of a two's complement int is & 255
        }
        return v;
    });
}

__global__ void __launch_bounds__(256) k_pointwise16(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, size_t n, int cn,
                                                     int op, int p0, int p1, unsigned chmask)
{
    // complement / posterization (the permutation needs whole pixels: k_pointwise)
    map_bytes16(src, dst, n, cn, [=](int v, int c, int) {
        const bool on = chmask == 0 || ((chmask >> c) & 1u);
        if (op == VKX_POINT_COMPLEMENT) {
            if (on && (p0 < 0 || (p1 ? v <= p0 : p0 <= v))) v = 255 - v;
        } else if (on) {
            v &= (0xFF >> p0) << p0;
        }
        return v;
    });
}

__global__ void __launch_bounds__(256) k_apply_lut16(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, size_t n, int cn,
                                                     const uint8_t *__restrict__ lut /* [cn][256] */, unsigned chmask)
{
    __shared__ uint32_t table_w[256];       // the table may sit in mapped host memory: whole dwords, one round trip per workgroup
    if ((int)threadIdx.x < cn * 64) table_w[threadIdx.x] = ((const uint32_t *)lut)[threadIdx.x];
    __syncthreads();
    const uint8_t *table = (const uint8_t *)table_w;
    map_bytes16(src, dst, n, cn, [&](int v, int c, int) {
        return (chmask == 0 || ((chmask >> c) & 1u)) ? (int)table[c * 256 + v] : v;
    });
}

// ... and up to eight dense single-channel planes, each through its own table, in one launch (blockIdx.y = plane): PageResizingStep
// binarises four masks before and after their resize -- eight launches of a few microseconds each as two.
struct LutPlanes {
    const uint8_t *src[8];
    uint8_t *dst[8];
    unsigned long long n[8];
};
__global__ void __launch_bounds__(256) k_apply_lut16_planes(LutPlanes P, const uint8_t *__restrict__ luts /* [planes][256] */)
{
    __shared__ uint32_t table_w[64];
    const int p = blockIdx.y;
    if ((size_t)blockIdx.x * 4096 >= P.n[p]) return;          // (uniform per workgroup: before the barrier)
    if (threadIdx.x < 64) table_w[threadIdx.x] = ((const uint32_t *)(luts + 256 * p))[threadIdx.x];
    __syncthreads();
    const uint8_t *table = (const uint8_t *)table_w;
    map_bytes16(P.src[p], P.dst[p], (size_t)P.n[p], 1, [&](int v, int, int) { return (int)table[v]; });
}

__global__ void __launch_bounds__(256) k_add_noise16(const uint8_t *__restrict__ src, const int16_t *__restrict__ noise,
                                                     uint8_t *__restrict__ dst, size_t n)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x, i0 = t * 16;
    if (i0 >= n) return;
    if (i0 + 16 <= n) {
        const uint4 in = *(const uint4 *)(src + i0);
        const uint4 na = *(const uint4 *)(noise + i0), nb = *(const uint4 *)(noise + i0 + 8);
        const uint32_t w[4] = {in.x, in.y, in.z, in.w};
        const uint32_t z[8] = {na.x, na.y, na.z, na.w, nb.x, nb.y, nb.z, nb.w};
        uint32_t o[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            o[k] = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int j = 4 * k + b;
                const int16_t nz = (int16_t)(z[j >> 1] >> (16 * (j & 1)));
                // int16 arithmetic like the reference's (uint8 -> int16) + int16 sum: wraps at 16 bits
                const int v = (int16_t)((int16_t)((w[k] >> (8 * b)) & 0xffu) + nz);
                o[k] |= (uint32_t)vkd::clamp_u8(v) << (8 * b);
            }
        }
        *(uint4 *)(dst + i0) = make_uint4(o[0], o[1], o[2], o[3]);
    } else {
        for (size_t i = i0; i < n; i++) dst[i] = (uint8_t)vkd::clamp_u8((int16_t)((int16_t)src[i] + noise[i]));
    }
}

// impulse_noise on a dense plane: a lane takes 16 PIXELS (16 selector bytes, 16 CN pixel bytes)
template <int CN>
__global__ void __launch_bounds__(256) k_impulse_noise16(const uint8_t *__restrict__ src, const uint8_t *__restrict__ sel,
                                                         uint8_t *__restrict__ dst, size_t npix)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x, p0 = t * 16;
    if (p0 >= npix) return;
    if (p0 + 16 <= npix) {
        const uint4 sv = *(const uint4 *)(sel + p0);
        const uint32_t m[4] = {sv.x, sv.y, sv.z, sv.w};
        uint32_t w[4 * CN];
#pragma unroll
        for (int q = 0; q < CN; q++) {
            const uint4 in = *(const uint4 *)(src + p0 * CN + 16 * q);
            w[4 * q] = in.x; w[4 * q + 1] = in.y; w[4 * q + 2] = in.z; w[4 * q + 3] = in.w;
        }
#pragma unroll
        for (int j = 0; j < 16 * CN; j++) {
            const int px = j / CN;
            const uint32_t s = (m[px >> 2] >> (8 * (px & 3))) & 0xffu;
            const uint32_t keep = (w[j >> 2] >> (8 * (j & 3))) & 0xffu;
            const uint32_t v = s == 1 ? 255u : (s == 2 ? 0u : keep);
            w[j >> 2] = (w[j >> 2] & ~(0xffu << (8 * (j & 3)))) | (v << (8 * (j & 3)));
        }
#pragma unroll
        for (int q = 0; q < CN; q++) *(uint4 *)(dst + p0 * CN + 16 * q) = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
    } else {
        for (size_t p = p0; p < npix; p++) {
            const int s = sel[p];
            for (int c = 0; c < CN; c++) dst[p * CN + c] = s == 1 ? (uint8_t)255 : (s == 2 ? (uint8_t)0 : src[p * CN + c]);
        }
    }
}

// RGB pixel operators on a dense plane: a lane takes 4 pixels = 12 bytes = 3 dwords; the HSV division tables and the hue
// sector selectors sit in LDS (colour shift: the arithmetic of the fused chain kernel, vkd::hue_shift_packed).
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));

template <class F>
__device__ __forceinline__ void map_rgb4(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, size_t npix, F f)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x, p0 = t * 4;
    if (p0 >= npix) return;
    if (p0 + 4 <= npix) {
        const u32x3 in = *(const u32x3 *)(src + p0 * 3);
        const uint32_t w0 = in.x, w1 = in.y, w2 = in.z;
        // bytes: r0 g0 b0 r1 | g1 b1 r2 g2 | b2 r3 g3 b3;  f returns r | g << 8 | b << 16
        const uint32_t q0 = f(w0 & 0xffu, (w0 >> 8) & 0xffu, (w0 >> 16) & 0xffu);
        const uint32_t q1 = f(w0 >> 24, w1 & 0xffu, (w1 >> 8) & 0xffu);
        const uint32_t q2 = f((w1 >> 16) & 0xffu, w1 >> 24, w2 & 0xffu);
        const uint32_t q3 = f((w2 >> 8) & 0xffu, (w2 >> 16) & 0xffu, w2 >> 24);
        u32x3 out;
        out.x = q0 | (q1 << 24);
        out.y = (q1 >> 8) | (q2 << 16);
        out.z = (q2 >> 16) | (q3 << 8);
        *(u32x3 *)(dst + p0 * 3) = out;
    } else {
        for (size_t p = p0; p < npix; p++) {
            const uint32_t q = f(src[p * 3], src[p * 3 + 1], src[p * 3 + 2]);
            dst[p * 3] = (uint8_t)q; dst[p * 3 + 1] = (uint8_t)(q >> 8); dst[p * 3 + 2] = (uint8_t)(q >> 16);
        }
    }
}

template <int MODE>
__global__ void __launch_bounds__(256) k_hsv4(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, size_t npix, int delta,
                                              const HsvTables *__restrict__ T)
{
    __shared__ int lsdiv[256], lhdiv[256];
    __shared__ uint32_t lsel[8];
    if (MODE != 2) {
        lsdiv[threadIdx.x] = T->sdiv[threadIdx.x];
        lhdiv[threadIdx.x] = T->hdiv[threadIdx.x];
        if (threadIdx.x < 8) lsel[threadIdx.x] = vkd::kHsvSelectors[threadIdx.x];
        __syncthreads();
    }
    map_rgb4(src, dst, npix, [&](uint32_t a, uint32_t b, uint32_t c) -> uint32_t {
        if (MODE == 0) return vkd::hue_shift_packed(lsdiv, lhdiv, lsel, delta, (int)a, (int)b, (int)c);
        int x, y, z;
        if (MODE == 1) vkd::rgb2hsv_full(lsdiv, lhdiv, (int)a, (int)b, (int)c, x, y, z);
        else vkd::hsv2rgb_full((int)a, (int)b, (int)c, x, y, z);
        return (uint32_t)x | ((uint32_t)y << 8) | ((uint32_t)z << 16);
    });
}

// brightness_shift / RGB <-> HLS / color_balance (k_cvt modes 0, 1, 2, 5)
template <int MODE>
__global__ void __launch_bounds__(256) k_cvt4(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, size_t npix, int delta,
                                              float w0, float w1)
{
    map_rgb4(src, dst, npix, [=](uint32_t ua, uint32_t ub, uint32_t uc) -> uint32_t {
        int a = (int)ua, b = (int)ub, c = (int)uc;
        if (MODE == 5) {
            const float gray = (float)rgb2gray_px(a, b, c);
            const int in[3] = {a, b, c};
            uint32_t q = 0;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const float t0 = w0 * gray, t1 = w1 * (float)in[k];
                float v = t0 + t1;
                v = v < 0.f ? 0.f : (v > 255.f ? 255.f : v);
                q |= (uint32_t)(uint8_t)v << (8 * k);
            }
            return q;
        }
        if (MODE == 0 || MODE == 1) {
            int H, L, S;
            rgb2hls_px(a, b, c, H, L, S);
            a = H; b = L; c = S;
        }
        if (MODE == 0 && delta != 0) b = vkd::clamp_u8(b + delta);
        if (MODE == 0 || MODE == 2) {
            int R, G, B;
            hls2rgb_px(a, b, c, R, G, B);
            a = R; b = G; c = B;
        }
        return (uint32_t)a | ((uint32_t)b << 8) | ((uint32_t)c << 16);
    });
}

// Histogram of a dense plane: a few hundred workgroups stride over the 16-byte groups, every wavefront counts into its
// own LDS copy (fewer same-address collisions), one flush per workgroup (the per-row version above flushes 768 counters from
// thousands of workgroups: device-scope atomics on 768 addresses were its whole run time).
__global__ void __launch_bounds__(256) k_histogram16(const uint8_t *__restrict__ src, size_t n, int cn, int *__restrict__ hist)
{
    __shared__ int lh[4][4 * 256];
    for (int i = threadIdx.x; i < 4 * 4 * 256; i += 256) (&lh[0][0])[i] = 0;
    __syncthreads();
    int *mine = lh[threadIdx.x >> 6];
    const size_t groups = (n + 15) / 16, stride = (size_t)gridDim.x * 256;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < groups; t += stride) {
        const size_t i0 = t * 16;
        int c = first_channel16(t, cn);
        if (i0 + 16 <= n) {
            const uint4 in = *(const uint4 *)(src + i0);
            const uint32_t w[4] = {in.x, in.y, in.z, in.w};
#pragma unroll
            for (int j = 0; j < 16; j++) {
                atomicAdd(&mine[c * 256 + (int)((w[j >> 2] >> (8 * (j & 3))) & 0xffu)], 1);
                c = c + 1 == cn ? 0 : c + 1;
            }
        } else {
            for (size_t i = i0; i < n; i++) {
                atomicAdd(&mine[c * 256 + src[i]], 1);
                c = c + 1 == cn ? 0 : c + 1;
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < cn * 256; i += 256) {
        const int v = lh[0][i] + lh[1][i] + lh[2][i] + lh[3][i];
        if (v) atomicAdd(&hist[i], v);
    }
}

// a plane qualifies when its rows follow each other without gaps and every pointer is aligned for the widest access
inline bool dense16(const void *a, ptrdiff_t a_pitch, const void *b, ptrdiff_t b_pitch, size_t row_bytes, int h, unsigned align = 16)
{
    return (h == 1 || ((size_t)a_pitch == row_bytes && (size_t)b_pitch == row_bytes)) &&
           (((uintptr_t)a | (uintptr_t)b) & (align - 1)) == 0;
}

int check_plane(vkx_ctx *ctx, const void *src, const void *dst, int h, int w)
{
    VKX_REQUIRE(ctx && src && dst, "NULL argument");
    VKX_REQUIRE(h >= 0 && w >= 0, "bad shape");
    return VKX_OK;
}

} // namespace

VKX_EXPORT int vkx_gaussian_blur_u8_dev(vkx_ctx *ctx, const uint8_t *src, int h, int w, int cn, ptrdiff_t src_stride,
                                        int ksize, double sigma, uint8_t *dst, ptrdiff_t dst_stride)
{
    int rc = check_plane(ctx, src, dst, h, w);
    if (rc) return rc;
    VKX_REQUIRE(src != dst, "gaussian blur cannot run in place");
    if (h == 0 || w == 0) return VKX_OK;
    BlurKernel K;
    K.kw = w == 1 ? 1 : ksize;
    K.kh = h == 1 ? 1 : ksize;
    if (ksize == 1) { K.kw = K.kh = 1; }
    if (K.kw > 1 || K.kh > 1) {
        if (gaussian_kernel_q8(ksize, sigma, K.k)) {
            vkx_set_error("gaussian blur: unsupported ksize=%d sigma=%g (odd ksize <= %d, sigma > 0)", ksize, sigma,
                          kMaxKsize);
            return VKX_ERR_UNSUPPORTED;
        }
    }
    if (K.kw == K.kh && K.kw > 1 && K.kw <= 2 * kBlurRMax + 1 && (cn == 1 || cn == 3 || cn == 4)) {
        const int tw = 64 - 2 * (K.kw / 2);
        dim3 tgrid(vkx_blocks(w, tw), vkx_blocks(h, kBlurTileH));
        VKX_TIMED(ctx, "k_gaussian_blur");
        switch (cn) {
        case 1: k_gaussian_blur_tiled<1><<<tgrid, 256, 0, ctx->stream>>>(src, h, w, src_stride, dst, dst_stride, K); break;
        case 3:
            if (w < 2) k_gaussian_blur_tiled<3><<<tgrid, 256, 0, ctx->stream>>>(src, h, w, src_stride, dst, dst_stride, K);   // the rgb kernel loads dwords
            else if (K.kw == 3) k_gaussian_blur_rgb<1><<<tgrid, 256, 0, ctx->stream>>>(src, h, w, src_stride, dst, dst_stride, K);
            else if (K.kw == 5) k_gaussian_blur_rgb<2><<<tgrid, 256, 0, ctx->stream>>>(src, h, w, src_stride, dst, dst_stride, K);
            else k_gaussian_blur_rgb<3><<<tgrid, 256, 0, ctx->stream>>>(src, h, w, src_stride, dst, dst_stride, K);
            break;
        default: k_gaussian_blur_tiled<4><<<tgrid, 256, 0, ctx->stream>>>(src, h, w, src_stride, dst, dst_stride, K); break;
        }
        VKX_LAUNCH_CHECK();
        return VKX_OK;
    }
    dim3 block(64, 4), grid(vkx_blocks(w, 64), vkx_blocks(h, 4));
    switch (cn) {
    case 1: { VKX_TIMED(ctx, "k_gaussian_blur"); k_gaussian_blur<1><<<grid, block, 0, ctx->stream>>>(src, h, w, src_stride, dst, dst_stride, K); } break;
    case 3: { VKX_TIMED(ctx, "k_gaussian_blur"); k_gaussian_blur<3><<<grid, block, 0, ctx->stream>>>(src, h, w, src_stride, dst, dst_stride, K); } break;
    case 4: { VKX_TIMED(ctx, "k_gaussian_blur"); k_gaussian_blur<4><<<grid, block, 0, ctx->stream>>>(src, h, w, src_stride, dst, dst_stride, K); } break;
    default: vkx_set_error("unsupported channel count %d", cn); return VKX_ERR_UNSUPPORTED;
    }
    VKX_LAUNCH_CHECK();
    return VKX_OK;
}

VKX_EXPORT int vkx_filter2d_u8_dev(vkx_ctx *ctx, const uint8_t *src, int h, int w, int cn, ptrdiff_t src_stride,
                                   const float *kernel_host, int kh, int kw, uint8_t *dst, ptrdiff_t dst_stride)
{
    int rc = check_plane(ctx, src, dst, h, w);
    if (rc) return rc;
    VKX_REQUIRE(kernel_host && kh >= 1 && kw >= 1 && kh <= kF2dMaxK && kw <= kF2dMaxK, "kernel of 1..15 rows and columns");
    VKX_REQUIRE(cn == 1 || cn == 3 || cn == 4, "1, 3 or 4 channels");
    if (h == 0 || w == 0) return VKX_OK;
    F2dKernel K;
    K.kh = kh; K.kw = kw;
    for (int i = 0; i < kF2dMaxK * kF2dMaxK; i++) K.k[i] = i < kh * kw ? kernel_host[i] : 0.f;
    dim3 grid(vkx_blocks(w, kF2dTileW), vkx_blocks(h, kF2dTileH));
    VKX_TIMED(ctx, "k_filter2d");
    switch (cn) {
    case 1: k_filter2d_u8<1><<<grid, 256, 0, ctx->stream>>>(src, h, w, src_stride, dst, dst_stride, K); break;
    case 3: k_filter2d_u8<3><<<grid, 256, 0, ctx->stream>>>(src, h, w, src_stride, dst, dst_stride, K); break;
    default: k_filter2d_u8<4><<<grid, 256, 0, ctx->stream>>>(src, h, w, src_stride, dst, dst_stride, K); break;
    }
    VKX_LAUNCH_CHECK();
    return VKX_OK;
}

int vkx_gaussian_kernel_q8_host(int n, double sigma, uint16_t *kq) { return gaussian_kernel_q8(n, sigma, kq); }

static int hsv_tables(vkx_ctx *ctx, const HsvTables **out)
{
    if (!ctx->tables_ready) {
        int rc = vkx_scratch_reserve(ctx, &ctx->tables, sizeof(HsvTables));
        if (rc) return rc;
        HsvTables T;
        T.sdiv[0] = T.hdiv[0] = 0;
        for (int i = 1; i < 256; i++) {
            T.sdiv[i] = (int)nearbyint((255 << 12) / (1. * i));
            T.hdiv[i] = (int)nearbyint((256 << 12) / (6. * i));
        }
        VKX_HIP(hipMemcpyAsync(ctx->tables.ptr, &T, sizeof T, hipMemcpyHostToDevice, ctx->stream));
        VKX_HIP(hipStreamSynchronize(ctx->stream)); // T lives on this stack frame
        ctx->tables_ready = true;
    }
    *out = (const HsvTables *)ctx->tables.ptr;
    return VKX_OK;
}

template <int MODE>
static int launch_hsv(vkx_ctx *ctx, const uint8_t *src, int h, int w, ptrdiff_t src_stride, int delta, uint8_t *dst,
                      ptrdiff_t dst_stride)
{
    int rc = check_plane(ctx, src, dst, h, w);
    if (rc) return rc;
    if (h == 0 || w == 0) return VKX_OK;
    const HsvTables *T = nullptr;
    rc = hsv_tables(ctx, &T);
    if (rc) return rc;
    if (dense16(src, src_stride, dst, dst_stride, (size_t)w * 3, h, 4)) {
        const size_t npix = (size_t)h * w;
        { VKX_TIMED(ctx, "k_hsv"); k_hsv4<MODE><<<vkx_blocks((npix + 3) / 4, 256), 256, 0, ctx->stream>>>(src, dst, npix, delta, T); }
        VKX_LAUNCH_CHECK();
        return VKX_OK;
    }
    dim3 block(64, 4), grid(vkx_blocks(w, 64), vkx_blocks(h, 4));
    { VKX_TIMED(ctx, "k_hsv"); k_hsv<MODE><<<grid, block, 0, ctx->stream>>>(src, h, w, src_stride, dst, dst_stride, delta, T); }
    VKX_LAUNCH_CHECK();
    return VKX_OK;
}

VKX_EXPORT int vkx_color_shift_rgb_dev(vkx_ctx *ctx, const uint8_t *src, int h, int w, ptrdiff_t src_stride, int delta,
                                       uint8_t *dst, ptrdiff_t dst_stride)
{
    return launch_hsv<0>(ctx, src, h, w, src_stride, delta, dst, dst_stride);
}

VKX_EXPORT int vkx_cvt_rgb_hsv_u8_dev(vkx_ctx *ctx, const uint8_t *src, int h, int w, ptrdiff_t src_stride, int to_hsv,
                                      uint8_t *dst, ptrdiff_t dst_stride)
{
    return to_hsv ? launch_hsv<1>(ctx, src, h, w, src_stride, 0, dst, dst_stride)
                  : launch_hsv<2>(ctx, src, h, w, src_stride, 0, dst, dst_stride);
}

template <int MODE>
static int launch_cvt(vkx_ctx *ctx, const uint8_t *src, int h, int w, ptrdiff_t src_stride, int delta, float w0, float w1,
                      uint8_t *dst, ptrdiff_t dst_stride)
{
    int rc = check_plane(ctx, src, dst, h, w);
    if (rc) return rc;
    if (h == 0 || w == 0) return VKX_OK;
    if constexpr (MODE == 0 || MODE == 1 || MODE == 2 || MODE == 5) {
        if (dense16(src, src_stride, dst, dst_stride, (size_t)w * 3, h, 4)) {
            const size_t npix = (size_t)h * w;
            { VKX_TIMED(ctx, "k_cvt"); k_cvt4<MODE><<<vkx_blocks((npix + 3) / 4, 256), 256, 0, ctx->stream>>>(src, dst, npix, delta, w0, w1); }
            VKX_LAUNCH_CHECK();
            return VKX_OK;
        }
    }
    dim3 block(64, 4), grid(vkx_blocks(w, 64), vkx_blocks(h, 4));
    { VKX_TIMED(ctx, "k_cvt"); k_cvt<MODE><<<grid, block, 0, ctx->stream>>>(src, h, w, src_stride, dst, dst_stride, delta, w0, w1); }
    VKX_LAUNCH_CHECK();
    return VKX_OK;
}

VKX_EXPORT int vkx_cvt_color_u8_dev(vkx_ctx *ctx, const uint8_t *src, int h, int w, ptrdiff_t src_stride, int code,
                                    uint8_t *dst, ptrdiff_t dst_stride)
{
    switch (code) {
    case VKX_CVT_RGB2HLS_FULL: return launch_cvt<1>(ctx, src, h, w, src_stride, 0, 0.f, 0.f, dst, dst_stride);
    case VKX_CVT_HLS2RGB_FULL: return launch_cvt<2>(ctx, src, h, w, src_stride, 0, 0.f, 0.f, dst, dst_stride);
    case VKX_CVT_RGB2GRAY:
        VKX_REQUIRE(src != dst, "RGB2GRAY cannot run in place");
        return launch_cvt<3>(ctx, src, h, w, src_stride, 0, 0.f, 0.f, dst, dst_stride);
    case VKX_CVT_GRAY2RGB:
        VKX_REQUIRE(src != dst, "GRAY2RGB cannot run in place");
        return launch_cvt<4>(ctx, src, h, w, src_stride, 0, 0.f, 0.f, dst, dst_stride);
    case VKX_CVT_RGBA2RGB: case VKX_CVT_RGB2RGBA: case VKX_CVT_GRAY2RGBA: case VKX_CVT_RGBA2GRAY: {
        int rc = check_plane(ctx, src, dst, h, w);
        if (rc) return rc;
        VKX_REQUIRE(src != dst, "the alpha conversions cannot run in place");
        if (h == 0 || w == 0) return VKX_OK;
        dim3 block(64, 4), grid(vkx_blocks(w, 64), vkx_blocks(h, 4));
        VKX_TIMED(ctx, "k_cvt_alpha");
        if (code == VKX_CVT_RGBA2RGB) k_cvt_alpha<6><<<grid, block, 0, ctx->stream>>>(src, h, w, src_stride, dst, dst_stride);
        else if (code == VKX_CVT_RGB2RGBA) k_cvt_alpha<7><<<grid, block, 0, ctx->stream>>>(src, h, w, src_stride, dst, dst_stride);
        else if (code == VKX_CVT_GRAY2RGBA) k_cvt_alpha<8><<<grid, block, 0, ctx->stream>>>(src, h, w, src_stride, dst, dst_stride);
        else k_cvt_alpha<9><<<grid, block, 0, ctx->stream>>>(src, h, w, src_stride, dst, dst_stride);
        VKX_LAUNCH_CHECK();
        return VKX_OK;
    }
    case VKX_CVT_RGB2HSV_FULL: return vkx_cvt_rgb_hsv_u8_dev(ctx, src, h, w, src_stride, 1, dst, dst_stride);
    case VKX_CVT_HSV2RGB_FULL: return vkx_cvt_rgb_hsv_u8_dev(ctx, src, h, w, src_stride, 0, dst, dst_stride);
    default:
        vkx_set_error("unknown colour conversion code %d", code);
        return VKX_ERR_INVALID;
    }
}

VKX_EXPORT int vkx_brightness_shift_rgb_dev(vkx_ctx *ctx, const uint8_t *src, int h, int w, ptrdiff_t src_stride, int delta,
                                            uint8_t *dst, ptrdiff_t dst_stride)
{
    return launch_cvt<0>(ctx, src, h, w, src_stride, delta, 0.f, 0.f, dst, dst_stride);
}

VKX_EXPORT int vkx_color_balance_rgb_dev(vkx_ctx *ctx, const uint8_t *src, int h, int w, ptrdiff_t src_stride, double ratio,
                                         uint8_t *dst, ptrdiff_t dst_stride)
{
    if (!(ratio >= 0.0 && ratio <= 1.0)) {
        vkx_set_error("ratio=%g is invalid.", ratio);
        return VKX_ERR_INVALID;
    }
    return launch_cvt<5>(ctx, src, h, w, src_stride, 0, (float)(1 - ratio), (float)ratio, dst, dst_stride);
}

VKX_EXPORT int vkx_blend_u8_dev(vkx_ctx *ctx, const uint8_t *a, ptrdiff_t a_stride, const uint8_t *b, ptrdiff_t b_stride, int h,
                                int w, int cn, double w0, double w1, unsigned channel_mask, uint8_t *dst, ptrdiff_t dst_stride)
{
    int rc = check_plane(ctx, a, dst, h, w);
    if (rc) return rc;
    VKX_REQUIRE(b != nullptr && cn >= 1 && cn <= 4, "bad argument");
    if (h == 0 || w == 0) return VKX_OK;
    dim3 block(64, 4), grid(vkx_blocks((size_t)w * cn, 64), vkx_blocks(h, 4));
    VKX_TIMED(ctx, "k_blend_u8");
    k_blend_u8<<<grid, block, 0, ctx->stream>>>(a, a_stride, b, b_stride, h, w * cn, cn, (float)w0, (float)w1, channel_mask, dst, dst_stride);
    VKX_LAUNCH_CHECK();
    return VKX_OK;
}

VKX_EXPORT int vkx_fog_f32_u8_dev(vkx_ctx *ctx, const uint8_t *src, int h, int w, int cn, ptrdiff_t src_stride, const float *weight,
                                  ptrdiff_t weight_stride_el, const float *fog /* host, cn values */, uint8_t *dst, ptrdiff_t dst_stride)
{
    int rc = check_plane(ctx, src, dst, h, w);
    if (rc) return rc;
    VKX_REQUIRE(weight && fog && cn >= 1 && cn <= 4, "bad argument");
    if (h == 0 || w == 0) return VKX_OK;
    float f[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < cn; c++) f[c] = fog[c];
    dim3 block(64, 4), grid(vkx_blocks(w, 64), vkx_blocks(h, 4));
    VKX_TIMED(ctx, "k_fog_f32");
    k_fog_f32<<<grid, block, 0, ctx->stream>>>(src, src_stride, weight, weight_stride_el, h, w, cn, f[0], f[1], f[2], f[3], dst, dst_stride);
    VKX_LAUNCH_CHECK();
    return VKX_OK;
}

VKX_EXPORT int vkx_mean_shift_u8_dev(vkx_ctx *ctx, const uint8_t *src, int h, int w, int cn, ptrdiff_t src_stride,
                                     int delta, int has_threshold, int threshold, int cycle, unsigned channel_mask,
                                     uint8_t *dst, ptrdiff_t dst_stride)
{
    int rc = check_plane(ctx, src, dst, h, w);
    if (rc) return rc;
    VKX_REQUIRE(cn >= 1 && cn <= 4, "1..4 channels");
    if (h == 0 || w == 0) return VKX_OK;
    if (dense16(src, src_stride, dst, dst_stride, (size_t)w * cn, h)) {
        const size_t n = (size_t)h * w * cn;
        { VKX_TIMED(ctx, "k_mean_shift"); k_mean_shift16<<<vkx_blocks((n + 15) / 16, 256), 256, 0, ctx->stream>>>(src, dst, n, cn, delta, has_threshold,
                                                                                                           threshold, cycle, channel_mask); }
        VKX_LAUNCH_CHECK();
        return VKX_OK;
    }
    dim3 block(64, 4), grid(vkx_blocks((size_t)w * cn, 64), vkx_blocks(h, 4));
    { VKX_TIMED(ctx, "k_mean_shift"); k_mean_shift<<<grid, block, 0, ctx->stream>>>(src, h, w, cn, src_stride, dst, dst_stride, delta, has_threshold,
                                                  threshold, cycle, channel_mask); }
    VKX_LAUNCH_CHECK();
    return VKX_OK;
}

VKX_EXPORT int vkx_pointwise_u8_dev(vkx_ctx *ctx, const uint8_t *src, int h, int w, int cn, ptrdiff_t src_stride, int op,
                                    int p0, int p1, unsigned channel_mask, uint8_t *dst, ptrdiff_t dst_stride)
{
    int rc = check_plane(ctx, src, dst, h, w);
    if (rc) return rc;
    VKX_REQUIRE(cn >= 1 && cn <= 4, "1..4 channels");
    VKX_REQUIRE(op >= VKX_POINT_COMPLEMENT && op <= VKX_POINT_PERMUTE, "unknown operator");
    if (op == VKX_POINT_COMPLEMENT) VKX_REQUIRE(p0 >= -1 && p0 <= 255, "threshold must be -1 or in [0, 255]");
    if (op == VKX_POINT_POSTERIZE) VKX_REQUIRE(p0 >= 0 && p0 < 8, "num_bits must be in [0, 7]");
    if (op == VKX_POINT_PERMUTE) {
        VKX_REQUIRE(src != dst, "channel permutation cannot run in place");
        for (int c = 0; c < cn; c++) VKX_REQUIRE(((p0 >> (2 * c)) & 3) < cn, "permutation index out of range");
    }
    if (h == 0 || w == 0) return VKX_OK;
    if (op != VKX_POINT_PERMUTE && dense16(src, src_stride, dst, dst_stride, (size_t)w * cn, h)) {
        const size_t n = (size_t)h * w * cn;
        { VKX_TIMED(ctx, "k_pointwise"); k_pointwise16<<<vkx_blocks((n + 15) / 16, 256), 256, 0, ctx->stream>>>(src, dst, n, cn, op, p0, p1, channel_mask); }
        VKX_LAUNCH_CHECK();
        return VKX_OK;
    }
    dim3 block(64, 4), grid(vkx_blocks((size_t)w * cn, 64), vkx_blocks(h, 4));
    { VKX_TIMED(ctx, "k_pointwise"); k_pointwise<<<grid, block, 0, ctx->stream>>>(src, h, w, cn, src_stride, dst, dst_stride, op, p0, p1, channel_mask); }
    VKX_LAUNCH_CHECK();
    return VKX_OK;
}

VKX_EXPORT int vkx_impulse_noise_u8_dev(vkx_ctx *ctx, const uint8_t *src, int h, int w, int cn, ptrdiff_t src_stride,
                                        const uint8_t *selector, ptrdiff_t selector_stride, uint8_t *dst,
                                        ptrdiff_t dst_stride)
{
    int rc = check_plane(ctx, src, dst, h, w);
    if (rc) return rc;
    VKX_REQUIRE(selector != nullptr, "NULL selector plane");
    VKX_REQUIRE(cn >= 1 && cn <= 4, "1..4 channels");
    if (h == 0 || w == 0) return VKX_OK;
    if (cn != 2 && dense16(src, src_stride, dst, dst_stride, (size_t)w * cn, h) && (h == 1 || selector_stride == w) &&
        ((uintptr_t)selector & 15) == 0) {
        const size_t npix = (size_t)h * w;
        const unsigned blocks = vkx_blocks((npix + 15) / 16, 256);
        VKX_TIMED(ctx, "k_impulse_noise");
        if (cn == 1) k_impulse_noise16<1><<<blocks, 256, 0, ctx->stream>>>(src, selector, dst, npix);
        else if (cn == 3) k_impulse_noise16<3><<<blocks, 256, 0, ctx->stream>>>(src, selector, dst, npix);
        else k_impulse_noise16<4><<<blocks, 256, 0, ctx->stream>>>(src, selector, dst, npix);
        VKX_LAUNCH_CHECK();
        return VKX_OK;
    }
    dim3 block(64, 4), grid(vkx_blocks((size_t)w * cn, 64), vkx_blocks(h, 4));
    { VKX_TIMED(ctx, "k_impulse_noise"); k_impulse_noise<<<grid, block, 0, ctx->stream>>>(src, h, w, cn, src_stride, selector, selector_stride, dst, dst_stride); }
    VKX_LAUNCH_CHECK();
    return VKX_OK;
}

VKX_EXPORT int vkx_speckle_noise_u8_dev(vkx_ctx *ctx, const uint8_t *src, int h, int w, int cn, ptrdiff_t src_stride,
                                        const double *noise, ptrdiff_t noise_stride_el, uint8_t *dst, ptrdiff_t dst_stride)
{
    int rc = check_plane(ctx, src, dst, h, w);
    if (rc) return rc;
    VKX_REQUIRE(noise != nullptr, "NULL noise plane");
    VKX_REQUIRE(cn >= 1 && cn <= 4, "1..4 channels");
    if (h == 0 || w == 0) return VKX_OK;
    dim3 block(64, 4), grid(vkx_blocks((size_t)w * cn, 64), vkx_blocks(h, 4));
    { VKX_TIMED(ctx, "k_speckle_noise"); k_speckle_noise<<<grid, block, 0, ctx->stream>>>(src, h, w * cn, src_stride, noise, noise_stride_el, dst, dst_stride); }
    VKX_LAUNCH_CHECK();
    return VKX_OK;
}

VKX_EXPORT int vkx_histogram_u8_dev(vkx_ctx *ctx, const uint8_t *src, int h, int w, int cn, ptrdiff_t src_stride,
                                    int32_t *hist)
{
    VKX_REQUIRE(ctx && src && hist, "NULL argument");
    VKX_REQUIRE(h >= 0 && w >= 0, "bad shape");
    VKX_REQUIRE(cn >= 1 && cn <= 4, "1..4 channels");
    VKX_HIP(hipMemsetAsync(hist, 0, sizeof(int32_t) * 256 * cn, ctx->stream));
    if (h == 0 || w == 0) return VKX_OK;
    if ((h == 1 || (size_t)src_stride == (size_t)w * cn) && ((uintptr_t)src & 15) == 0) {
        const size_t n = (size_t)h * w * cn;
        const unsigned blocks = std::min(vkx_blocks((n + 15) / 16, 256), 1024u);      // 4 per CU
        { VKX_TIMED(ctx, "k_histogram"); k_histogram16<<<blocks, 256, 0, ctx->stream>>>(src, n, cn, hist); }
        VKX_LAUNCH_CHECK();
        return VKX_OK;
    }
    dim3 grid(std::min(vkx_blocks((size_t)w * cn, 256), 16u), std::min((unsigned)h, 256u));
    { VKX_TIMED(ctx, "k_histogram"); k_histogram<<<grid, 256, 0, ctx->stream>>>(src, h, w, cn, src_stride, hist); }
    VKX_LAUNCH_CHECK();
    return VKX_OK;
}

VKX_EXPORT int vkx_apply_lut_u8_dev(vkx_ctx *ctx, const uint8_t *src, int h, int w, int cn, ptrdiff_t src_stride,
                                    const uint8_t *lut_host, unsigned channel_mask, uint8_t *dst, ptrdiff_t dst_stride)
{
    int rc = check_plane(ctx, src, dst, h, w);
    if (rc) return rc;
    VKX_REQUIRE(lut_host != nullptr, "NULL table");
    VKX_REQUIRE(cn >= 1 && cn <= 4, "1..4 channels");
    if (h == 0 || w == 0) return VKX_OK;
    // the table is the caller's memory: it travels through the page-locked descriptor ring, no stream synchronisation.  The dense
    // kernel stages it to LDS once per workgroup straight from the (mapped) ring: no copy, one dispatch per call instead of two
    // (PageResizingStep binarises seven planes per page through this entry)
    void *staged = nullptr;
    rc = vkx_desc_ring_take(ctx, (size_t)256 * cn, &staged);
    if (rc) return rc;
    memcpy(staged, lut_host, (size_t)256 * cn);
    const uint8_t *table = (const uint8_t *)vkx_ring_device_ptr(staged);
    const bool dense = dense16(src, src_stride, dst, dst_stride, (size_t)w * cn, h);
    if (!table || !dense) {          // the strided kernel reads the table per pixel: from device memory
        rc = vkx_scratch_reserve(ctx, &ctx->misc, 1024);
        if (rc) return rc;
        VKX_HIP(hipMemcpyAsync(ctx->misc.ptr, staged, (size_t)256 * cn, hipMemcpyHostToDevice, ctx->stream));
        table = (const uint8_t *)ctx->misc.ptr;
    }
    if (dense) {
        const size_t n = (size_t)h * w * cn;
        { VKX_TIMED(ctx, "k_apply_lut"); k_apply_lut16<<<vkx_blocks((n + 15) / 16, 256), 256, 0, ctx->stream>>>(src, dst, n, cn, table, channel_mask); }
        VKX_LAUNCH_CHECK();
        return VKX_OK;
    }
    dim3 block(64, 4), grid(vkx_blocks((size_t)w * cn, 64), vkx_blocks(h, 4));
    { VKX_TIMED(ctx, "k_apply_lut"); k_apply_lut<<<grid, block, 0, ctx->stream>>>(src, h, w, cn, src_stride, dst, dst_stride, table, channel_mask); }
    VKX_LAUNCH_CHECK();
    return VKX_OK;
}

VKX_EXPORT int vkx_apply_lut_u8_planes_dev(vkx_ctx *ctx, const vkx_lut_plane *planes, int n_planes)
{
    VKX_REQUIRE(ctx && planes, "NULL argument");
    VKX_REQUIRE(n_planes >= 1 && n_planes <= 8, "1 .. 8 planes per call");
    LutPlanes P;
    memset(&P, 0, sizeof(P));
    size_t n_max = 0;
    void *staged = nullptr;
    int rc = vkx_desc_ring_take(ctx, (size_t)256 * n_planes, &staged);
    if (rc) return rc;
    for (int i = 0; i < n_planes; i++) {
        const vkx_lut_plane &pl = planes[i];
        VKX_REQUIRE((pl.n_bytes == 0 || (pl.src && pl.dst)) && pl.lut_host, "NULL plane or table");
        VKX_REQUIRE((((uintptr_t)pl.src | (uintptr_t)pl.dst) & 15) == 0, "planes are 16-byte aligned, dense");
        P.src[i] = pl.src; P.dst[i] = pl.dst; P.n[i] = pl.n_bytes;
        n_max = std::max(n_max, pl.n_bytes);
        memcpy((uint8_t *)staged + 256 * i, pl.lut_host, 256);
    }
    if (n_max == 0) return VKX_OK;
    const uint8_t *tables = (const uint8_t *)vkx_ring_device_ptr(staged);
    if (!tables) {
        if ((rc = vkx_scratch_reserve(ctx, &ctx->misc, 2048))) return rc;
        VKX_HIP(hipMemcpyAsync(ctx->misc.ptr, staged, (size_t)256 * n_planes, hipMemcpyHostToDevice, ctx->stream));
        tables = (const uint8_t *)ctx->misc.ptr;
    }
    vkx_device_guard guard(ctx);
    dim3 grid(vkx_blocks((n_max + 15) / 16, 256), n_planes);
    { VKX_TIMED(ctx, "k_apply_lut"); k_apply_lut16_planes<<<grid, 256, 0, ctx->stream>>>(P, tables); }
    VKX_LAUNCH_CHECK();
    return VKX_OK;
}

VKX_EXPORT int vkx_gather_u8_dev(vkx_ctx *ctx, const uint8_t *src, int sh, int sw, int cn, ptrdiff_t src_stride,
                                 const int32_t *pos_y, const int32_t *pos_x, ptrdiff_t pos_stride_el, uint8_t *dst, int dh,
                                 int dw, ptrdiff_t dst_stride)
{
    VKX_REQUIRE(ctx && src && pos_y && pos_x && dst, "NULL argument");
    VKX_REQUIRE(sh >= 0 && sw >= 0 && dh >= 0 && dw >= 0, "bad shape");
    VKX_REQUIRE(cn == 1 || cn == 3 || cn == 4, "1, 3 or 4 channels");
    VKX_REQUIRE(src != dst, "gather cannot run in place");
    if (dh == 0 || dw == 0) return VKX_OK;
    int rc = vkx_scratch_reserve(ctx, &ctx->misc, 256);
    if (rc) return rc;
    int *bad = (int *)ctx->misc.ptr;
    VKX_HIP(hipMemsetAsync(bad, 0, sizeof(int), ctx->stream));
    dim3 block(64, 4), grid(vkx_blocks(dw, 64), vkx_blocks(dh, 4));
    {
        VKX_TIMED(ctx, "k_gather");
        switch (cn) {
        case 1: k_gather<1><<<grid, block, 0, ctx->stream>>>(src, sh, sw, src_stride, pos_y, pos_x, pos_stride_el, dst, dh, dw, dst_stride, bad); break;
        case 3: k_gather<3><<<grid, block, 0, ctx->stream>>>(src, sh, sw, src_stride, pos_y, pos_x, pos_stride_el, dst, dh, dw, dst_stride, bad); break;
        default: k_gather<4><<<grid, block, 0, ctx->stream>>>(src, sh, sw, src_stride, pos_y, pos_x, pos_stride_el, dst, dh, dw, dst_stride, bad); break;
        }
    }
    VKX_LAUNCH_CHECK();
    int flag = 0;
    VKX_HIP(hipMemcpyAsync(&flag, bad, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    VKX_HIP(hipStreamSynchronize(ctx->stream));
    if (flag) {
        vkx_set_error("gather index outside the %dx%d source", sh, sw);
        return VKX_ERR_INVALID;
    }
    return VKX_OK;
}

VKX_EXPORT int vkx_saturate_i64_u8_dev(vkx_ctx *ctx, const int64_t *src, size_t n, uint8_t *dst)
{
    VKX_REQUIRE(ctx && (n == 0 || (src && dst)), "NULL argument");
    if (n == 0) return VKX_OK;
    { VKX_TIMED(ctx, "k_saturate_i64"); k_saturate_i64<<<vkx_blocks(n, 256), 256, 0, ctx->stream>>>((const long long *)src, n, dst); }
    VKX_LAUNCH_CHECK();
    return VKX_OK;
}

VKX_EXPORT int vkx_add_noise_i16_dev(vkx_ctx *ctx, const uint8_t *src, int h, int w, int cn, ptrdiff_t src_stride,
                                     const int16_t *noise, ptrdiff_t noise_stride_el, uint8_t *dst, ptrdiff_t dst_stride)
{
    int rc = check_plane(ctx, src, dst, h, w);
    if (rc) return rc;
    VKX_REQUIRE(noise != nullptr, "NULL noise plane");
    VKX_REQUIRE(cn >= 1 && cn <= 4, "1..4 channels");
    if (h == 0 || w == 0) return VKX_OK;
    if (dense16(src, src_stride, dst, dst_stride, (size_t)w * cn, h) && (h == 1 || noise_stride_el == (ptrdiff_t)w * cn) &&
        ((uintptr_t)noise & 15) == 0) {
        const size_t n = (size_t)h * w * cn;
        { VKX_TIMED(ctx, "k_add_noise"); k_add_noise16<<<vkx_blocks((n + 15) / 16, 256), 256, 0, ctx->stream>>>(src, noise, dst, n); }
        VKX_LAUNCH_CHECK();
        return VKX_OK;
    }
    dim3 block(64, 4), grid(vkx_blocks((size_t)w * cn, 64), vkx_blocks(h, 4));
    { VKX_TIMED(ctx, "k_add_noise"); k_add_noise<<<grid, block, 0, ctx->stream>>>(src, h, w * cn, src_stride, noise, noise_stride_el, dst, dst_stride); }
    VKX_LAUNCH_CHECK();
    return VKX_OK;
}

VKX_EXPORT int vkx_line_streak_u8_dev(vkx_ctx *ctx, uint8_t *img, int h, int w, int cn, ptrdiff_t stride, int thickness,
                                      int gap, int dash_thickness, int dash_gap, const uint8_t color[4], double alpha,
                                      int enable_vert, int enable_hori)
{
    int rc = check_plane(ctx, img, img, h, w);
    if (rc) return rc;
    VKX_REQUIRE(color != nullptr, "NULL color");
    VKX_REQUIRE(thickness + gap > 0, "thickness + gap must be positive");
    if (alpha < 0.0 || alpha > 1.0) {
        vkx_set_error("alpha=%g is invalid.", alpha);
        return VKX_ERR_INVALID;
    }
    if (h == 0 || w == 0 || alpha == 0.0 || (!enable_vert && !enable_hori)) return VKX_OK;
    StreakParams P;
    P.thickness = thickness;
    P.step = thickness + gap;
    P.dash = dash_thickness > 0 && dash_gap > 0;
    P.dash_step = P.dash ? dash_thickness + dash_gap : 1;
    P.dash_gap = dash_gap;
    P.enable_vert = enable_vert;
    P.enable_hori = enable_hori;
    P.copy = alpha == 1.0;
    P.alpha = (float)alpha;
    for (int c = 0; c < 4; c++) P.color[c] = c < cn ? color[c] : 0;
    dim3 block(64, 4), grid(vkx_blocks(w, 64), vkx_blocks(h, 4 * kStreakRows));
    switch (cn) {
    case 1: { VKX_TIMED(ctx, "k_line_streak"); k_line_streak<1><<<grid, block, 0, ctx->stream>>>(img, h, w, stride, P); } break;
    case 3: { VKX_TIMED(ctx, "k_line_streak"); k_line_streak<3><<<grid, block, 0, ctx->stream>>>(img, h, w, stride, P); } break;
    case 4: { VKX_TIMED(ctx, "k_line_streak"); k_line_streak<4><<<grid, block, 0, ctx->stream>>>(img, h, w, stride, P); } break;
    default: vkx_set_error("unsupported channel count %d", cn); return VKX_ERR_UNSUPPORTED;
    }
    VKX_LAUNCH_CHECK();
    return VKX_OK;
}

// Device copy of the RGB->HSV division tables {int sdiv[256]; int hdiv[256];} for other translation units.
int vkx_hsv_tables(vkx_ctx *ctx, const void **out)
{
    const HsvTables *T = nullptr;
    int rc = hsv_tables(ctx, &T);
    *out = T;
    return rc;
}
