// fill_np_array (element/opt.py:118-209) as an ordered list of box-clipped layers blended in place on gfx950:
// the text-layer alpha composite of PageAssemblerStep.run (pipeline/text_detection/page_assembler.py:155-236).
// A single layer is one launch over its box (k_fill).  A layer LIST is binned on the host into 64 x 16 destination
// tiles (CSR, ascending layer index = the reference's order) and applied by ONE launch (k_composite): a workgroup
// owns a tile, every lane keeps its pixels in registers, walks the tile's layers in order and writes back once --
// a page of hundreds of text-line layers costs one destination read and one write instead of one launch and one
// read-modify-write per layer.  Lanes cover 64 consecutive pixels of a row so all plane traffic is coalesced;
// untouched pixels are neither read nor written.
#include "vkx_internal.h"

#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include <vector>

namespace {

template <typename T> struct LayerDev {
    int up, left, height, width;
    const uint8_t *mask;
    ptrdiff_t mask_stride;
    const float *alpha;
    ptrdiff_t alpha_stride;
    const T *value;
    ptrdiff_t value_stride; // elements of T per box row
    float alpha_scalar;
    int copy; // scalar alpha == 1.0
    int mode; // VKX_FILL_*: honoured in the copy branch only (element/opt.py:150-158)
    T value_const[4];
};

// uint8: trunc(fl32(1-a)*fl32(d) + fl32(a*fl32(v))), products rounded separately (-ffp-contract=off);
// float32: the same expression without the final conversion.
__device__ __forceinline__ uint8_t blend_px(float w0, float w1, uint8_t d, uint8_t v)
{
    const float t0 = w0 * (float)d, t1 = w1 * (float)v;
    return (uint8_t)(t0 + t1);
}
__device__ __forceinline__ float blend_px(float w0, float w1, float d, float v)
{
    const float t0 = w0 * d, t1 = w1 * v;
    return t0 + t1;
}

template <typename T, int CN>
__global__ void __launch_bounds__(256) k_fill(T *dst, ptrdiff_t dstride, LayerDev<T> L)
{
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x >= L.width || y >= L.height) return;
    const float a = L.alpha ? L.alpha[(ptrdiff_t)y * L.alpha_stride + x] : L.alpha_scalar;
    const bool sel = L.mask ? L.mask[(ptrdiff_t)y * L.mask_stride + x] > 0 : (L.alpha ? a > 0.0f : true);
    if (!sel) return;
    T *d = dst + (ptrdiff_t)(L.up + y) * dstride + (ptrdiff_t)(L.left + x) * CN;
    const T *v = L.value ? L.value + (ptrdiff_t)y * L.value_stride + (ptrdiff_t)x * CN : nullptr;
    if (L.copy) {
#pragma unroll
        for (int c = 0; c < CN; c++) {
            const T val = v ? v[c] : L.value_const[c];
            if (L.mode == VKX_FILL_PLAIN || (L.mode == VKX_FILL_KEEP_MAX ? d[c] < val : d[c] > val)) d[c] = val;
        }
    } else {
        const float w1 = a, w0 = 1.0f - w1;
#pragma unroll
        for (int c = 0; c < CN; c++) d[c] = blend_px(w0, w1, d[c], v ? v[c] : L.value_const[c]);
    }
}

constexpr int kTileW = 64, kTileH = 16;

// All layers overlapping one tile, in order; registers hold the lane's kTileH / 4 pixels.
template <typename T, int CN>
__global__ void __launch_bounds__(256) k_composite(T *dst, ptrdiff_t dstride, int h, int w,
                                                   const LayerDev<T> *__restrict__ layers,
                                                   const int *__restrict__ tile_ids, const int *__restrict__ tile_begin,
                                                   const int *__restrict__ tile_layers, int tiles_x,
                                                   T *const *__restrict__ pages = nullptr, int tiles_per_page = 0)
{
    int tile = tile_ids[blockIdx.x];
    if (pages) {                       // a batch of equally shaped destinations: the tile id carries the page
        const int page = tile / tiles_per_page;
        tile -= page * tiles_per_page;
        dst = pages[page];
    }
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int x = tx * kTileW + (threadIdx.x & 63);
    const int ybase = ty * kTileH + (threadIdx.x >> 6);
    constexpr int NR = kTileH / 4;
    T px[NR][CN];
    bool dirty[NR];
    const bool xin = x < w;
#pragma unroll
    for (int r = 0; r < NR; r++) {
        const int y = ybase + 4 * r;
        dirty[r] = false;
#pragma unroll
        for (int c = 0; c < CN; c++) px[r][c] = T(0);
        if (xin && y < h) {
            const T *d = dst + (ptrdiff_t)y * dstride + (ptrdiff_t)x * CN;
#pragma unroll
            for (int c = 0; c < CN; c++) px[r][c] = d[c];
        }
    }
    const int lb = tile_begin[blockIdx.x], le = tile_begin[blockIdx.x + 1];
    for (int li = lb; li < le; li++) {
        const LayerDev<T> &L = layers[tile_layers[li]];      // uniform: scalar loads
        const int bx = x - L.left;
        if (bx < 0 || bx >= L.width) continue;
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const int by = ybase + 4 * r - L.up;
            if (by < 0 || by >= L.height) continue;
            const float a = L.alpha ? L.alpha[(ptrdiff_t)by * L.alpha_stride + bx] : L.alpha_scalar;
            const bool sel = L.mask ? L.mask[(ptrdiff_t)by * L.mask_stride + bx] > 0 : (L.alpha ? a > 0.0f : true);
            if (!sel) continue;
            const T *v = L.value ? L.value + (ptrdiff_t)by * L.value_stride + (ptrdiff_t)bx * CN : nullptr;
            if (L.copy) {
#pragma unroll
                for (int c = 0; c < CN; c++) {
                    const T val = v ? v[c] : L.value_const[c];
                    if (L.mode == VKX_FILL_PLAIN || (L.mode == VKX_FILL_KEEP_MAX ? px[r][c] < val : px[r][c] > val))
                        px[r][c] = val;
                }
            } else {
                const float w1 = a, w0 = 1.0f - w1;
#pragma unroll
                for (int c = 0; c < CN; c++) px[r][c] = blend_px(w0, w1, px[r][c], v ? v[c] : L.value_const[c]);
            }
            dirty[r] = true;
        }
    }
#pragma unroll
    for (int r = 0; r < NR; r++) {
        if (!dirty[r]) continue;
        T *d = dst + (ptrdiff_t)(ybase + 4 * r) * dstride + (ptrdiff_t)x * CN;
#pragma unroll
        for (int c = 0; c < CN; c++) d[c] = px[r][c];
    }
}

// uint8 RGB destinations whose rows are whole 4-pixel groups (row pitch and width multiples of 4: every page the pipeline
// composes): a lane owns FOUR consecutive pixels of one row = 12 bytes = three dwords, so the destination moves as
// dwordx3 accesses, float alpha planes as dwordx4, byte masks as one dword and image layers as three.  The records of
// the tile's layers are copied to LDS by the whole workgroup in one round trip (instead of one dependent scalar load
// chain per layer), the planes of layer n + 1 are requested before layer n is blended, a wavefront whose 256 pixels a
// layer does not select skips its arithmetic, and the destination is not read where the tile's first layer is an opaque
// plain copy (the page background).  Arithmetic per pixel: exactly blend_px / the copy rules of k_composite.
constexpr int kRgbRecs = 32;     // layer records staged in LDS per pass
typedef uint32_t u32_a1 __attribute__((aligned(1)));
typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));

struct RgbFetch {
    float a[4];          // alpha of the four pixels (scalar alpha replicated)
    uint32_t sel;        // bit i: pixel i is selected by the layer
    uint32_t v[3];       // the twelve value bytes (image layers) or the constant colour replicated
};

__device__ __forceinline__ RgbFetch rgb_fetch(const LayerDev<uint8_t> &L, int x0, int y)
{
    RgbFetch f;
    f.sel = 0;
    f.a[0] = f.a[1] = f.a[2] = f.a[3] = L.alpha_scalar;
    const uint32_t c0 = L.value_const[0], c1 = L.value_const[1], c2 = L.value_const[2];
    f.v[0] = c0 | (c1 << 8) | (c2 << 16) | (c0 << 24);
    f.v[1] = c1 | (c2 << 8) | (c0 << 16) | (c1 << 24);
    f.v[2] = c2 | (c0 << 8) | (c1 << 16) | (c2 << 24);
    const int by = y - L.up, bx = x0 - L.left;
    if (by < 0 || by >= L.height || bx + 3 < 0 || bx >= L.width) return f;
    if (bx >= 0 && bx + 3 < L.width) {
        uint32_t inside = 0xf;
        if (L.alpha) {
            const f32x4_a4 a = *(const f32x4_a4 *)(L.alpha + (ptrdiff_t)by * L.alpha_stride + bx);
            f.a[0] = a.x; f.a[1] = a.y; f.a[2] = a.z; f.a[3] = a.w;
        }
        if (L.mask) {
            const uint32_t m = *(const u32_a1 *)(L.mask + (ptrdiff_t)by * L.mask_stride + bx);
            inside = ((m & 0xffu) ? 1u : 0u) | ((m & 0xff00u) ? 2u : 0u) | ((m & 0xff0000u) ? 4u : 0u) | ((m >> 24) ? 8u : 0u);
        } else if (L.alpha) {
            inside = (f.a[0] > 0.0f ? 1u : 0u) | (f.a[1] > 0.0f ? 2u : 0u) | (f.a[2] > 0.0f ? 4u : 0u) | (f.a[3] > 0.0f ? 8u : 0u);
        }
        f.sel = inside;
        if (L.value) {
            const u32_a1 *v = (const u32_a1 *)(L.value + (ptrdiff_t)by * L.value_stride + (ptrdiff_t)bx * 3);
            f.v[0] = v[0]; f.v[1] = v[1]; f.v[2] = v[2];
        }
    } else {
        // the layer's box edge cuts the group: pixel by pixel
        uint8_t vb[12];
        for (int i = 0; i < 12; i++) vb[i] = (uint8_t)(f.v[i >> 2] >> (8 * (i & 3)));
        for (int i = 0; i < 4; i++) {
            const int b = bx + i;
            if (b < 0 || b >= L.width) continue;
            const float a = L.alpha ? L.alpha[(ptrdiff_t)by * L.alpha_stride + b] : L.alpha_scalar;
            f.a[i] = a;
            const bool sel = L.mask ? L.mask[(ptrdiff_t)by * L.mask_stride + b] > 0 : (L.alpha ? a > 0.0f : true);
            if (sel) f.sel |= 1u << i;
            if (L.value)
                for (int c = 0; c < 3; c++) vb[3 * i + c] = L.value[(ptrdiff_t)by * L.value_stride + (ptrdiff_t)b * 3 + c];
        }
        for (int k = 0; k < 3; k++) f.v[k] = vb[4 * k] | (vb[4 * k + 1] << 8) | (vb[4 * k + 2] << 16) | ((uint32_t)vb[4 * k + 3] << 24);
    }
    return f;
}

// One layer applied to a lane's four pixels (three dwords): blend_px / the copy rules of k_composite on all twelve bytes, then the bytes
// of the pixels the layer selects replace the lane's.  `copy` and `mode` are wave-uniform.
__device__ __forceinline__ void rgb_apply(uint32_t (&px)[3], const RgbFetch &cur, int copy, int mode)
{
    const uint32_t s0 = 0u - (cur.sel & 1u), s1 = 0u - ((cur.sel >> 1) & 1u), s2 = 0u - ((cur.sel >> 2) & 1u), s3 = 0u - ((cur.sel >> 3) & 1u);
    const uint32_t m[3] = {(s0 & 0x00ffffffu) | (s1 & 0xff000000u), (s1 & 0x0000ffffu) | (s2 & 0xffff0000u), (s2 & 0x000000ffu) | (s3 & 0xffffff00u)};
    uint32_t nv[3];
    if (copy) {
        if (mode == VKX_FILL_PLAIN) {
            nv[0] = cur.v[0]; nv[1] = cur.v[1]; nv[2] = cur.v[2];
        } else {
#pragma unroll
            for (int q = 0; q < 3; q++) {
                uint32_t r = 0;
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const uint32_t d = (px[q] >> (8 * b)) & 0xffu, v = (cur.v[q] >> (8 * b)) & 0xffu;
                    r |= (mode == VKX_FILL_KEEP_MAX ? (d < v ? v : d) : (d > v ? v : d)) << (8 * b);
                }
                nv[q] = r;
            }
        }
    } else {
        float w1[4], w0[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            w1[i] = cur.a[i];
            w0[i] = 1.0f - w1[i];
        }
#pragma unroll
        for (int q = 0; q < 3; q++) {
            uint32_t r = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int i = (4 * q + b) / 3;                        // the pixel of byte 4 q + b
                const float d = (float)((px[q] >> (8 * b)) & 0xffu), v = (float)((cur.v[q] >> (8 * b)) & 0xffu);
                const float t0 = w0[i] * d, t1 = w1[i] * v;
                r |= ((uint32_t)(int)(t0 + t1) & 0xffu) << (8 * b);  // (uint8_t)(float): blend_px
            }
            nv[q] = r;
        }
    }
#pragma unroll
    for (int q = 0; q < 3; q++) px[q] = (px[q] & ~m[q]) | (nv[q] & m[q]);
}

// A workgroup owns a RUN of kRgbRun consecutive tile slots: the tables of the whole run arrive in three round trips (slots' ids and
// ranges; the layer indices; the layer records, staged in LDS by all lanes) instead of three per tile, and the (tile, layer) pairs of the
// run are walked as ONE list -- the planes of pair n + 1, which may belong to the next tile, are requested before pair n is blended, a
// tile's pixels are initialised at its first pair and stored at its last.  (Round 5: one tile per workgroup spent 60 % of its
// wavefront cycles waiting on four dependent round trips for 1 024 pixels.)
constexpr int kRgbRun = 8;
__global__ void __launch_bounds__(256) k_composite_rgb(uint8_t *dst, ptrdiff_t dstride, int h, int w,
                                                       const LayerDev<uint8_t> *__restrict__ layers,
                                                       const int *__restrict__ tile_ids, const int *__restrict__ tile_begin,
                                                       const int *__restrict__ tile_layers, int tiles_x,
                                                       uint8_t *const *__restrict__ pages, int tiles_per_page, int n_tiles, int run)
{
    __shared__ LayerDev<uint8_t> recs[kRgbRecs];
    __shared__ int s_begin[kRgbRun + 1], s_x[kRgbRun], s_y[kRgbRun];
    __shared__ uint8_t *s_dst[kRgbRun];
    const int t0 = blockIdx.x * run, nt = min(run, n_tiles - t0);
    if ((int)threadIdx.x <= nt) s_begin[threadIdx.x] = tile_begin[t0 + threadIdx.x];
    if ((int)threadIdx.x < nt) {
        int tile = tile_ids[t0 + threadIdx.x];
        uint8_t *d = dst;
        if (pages) {
            const int page = tile / tiles_per_page;
            tile -= page * tiles_per_page;
            d = pages[page];
        }
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
        s_x[threadIdx.x] = tx * kTileW;
        s_y[threadIdx.x] = ty * kTileH;
        s_dst[threadIdx.x] = d;
    }
    __syncthreads();
    const int lb = s_begin[0], le = s_begin[nt];
    const int lx = 4 * (threadIdx.x & 15), ly = threadIdx.x >> 4;
    uint32_t px[3] = {0, 0, 0};
    bool dirty = false;
    int k = 0;                                   // slot of the current pair's tile (wave-uniform, like every index below)
    for (int base = lb; base < le; base += kRgbRecs) {
        const int n = min(kRgbRecs, le - base);
        if (base != lb) __syncthreads();
        // the records of this pass: n * 20 dwords, one per thread and step
        constexpr int kDw = (int)(sizeof(LayerDev<uint8_t>) / 4);
        for (int i = threadIdx.x; i < n * kDw; i += 256) {
            const int r = i / kDw, d = i - r * kDw;
            ((uint32_t *)&recs[r])[d] = ((const uint32_t *)&layers[tile_layers[base + r]])[d];
        }
        __syncthreads();
        int x0 = s_x[k] + lx, y = s_y[k] + ly;
        bool in_page = x0 < w && y < h;         // w % 4 == 0: a group is inside or outside as a whole
        const LayerDev<uint8_t> *Lp = &recs[0];
        RgbFetch cur;
        cur.sel = 0;
        if (in_page) cur = rgb_fetch(*Lp, x0, y);
        for (int li = 0; li < n; li++) {
            const int gi = base + li;
            const LayerDev<uint8_t> &L = *Lp;
            const bool last_of_tile = gi + 1 == s_begin[k + 1];
            int xn = x0, yn = y;
            bool in_next = in_page;
            RgbFetch nxt;
            nxt.sel = 0;
            if (li + 1 < n) {
                if (last_of_tile) {
                    xn = s_x[k + 1] + lx;
                    yn = s_y[k + 1] + ly;
                    in_next = xn < w && yn < h;
                }
                if (in_next) nxt = rgb_fetch(recs[li + 1], xn, yn);
            } else if (last_of_tile && k + 1 < nt) {                 // the next pass starts with the next tile
                xn = s_x[k + 1] + lx;
                yn = s_y[k + 1] + ly;
                in_next = xn < w && yn < h;
            }
            uint32_t *drow = (uint32_t *)(s_dst[k] + (ptrdiff_t)y * dstride + (ptrdiff_t)x0 * 3);
            if (gi == s_begin[k]) {
                px[0] = px[1] = px[2] = 0;
                dirty = false;
                // the destination is needed unless the first layer writes all four pixels unconditionally
                const bool covers = L.copy && L.mode == VKX_FILL_PLAIN && !L.mask && y >= L.up && y < L.up + L.height &&
                                    x0 >= L.left && x0 + 3 < L.left + L.width;
                if (in_page && !covers) { px[0] = drow[0]; px[1] = drow[1]; px[2] = drow[2]; }
            }
            if (__ballot(cur.sel != 0)) {
                // the layer's kind is wave-uniform: one branch per layer, straight-line arithmetic for the four pixels, the pixels the layer
                // does not select put back by a byte mask
                rgb_apply(px, cur, __builtin_amdgcn_readfirstlane(L.copy), __builtin_amdgcn_readfirstlane(L.mode));
                dirty = dirty || cur.sel != 0;
            }
            if (last_of_tile) {
                if (in_page && dirty) { drow[0] = px[0]; drow[1] = px[1]; drow[2] = px[2]; }
                k++;
            }
            cur = nxt;
            Lp = &recs[li + 1 < n ? li + 1 : li];
            x0 = xn; y = yn; in_page = in_next;
        }
    }
}

// The 4-pixel-group kernel for uint8 RGB when every destination is dword aligned with whole groups per row.
template <typename T, int CN>
bool composite_rgb_groups(vkx_ctx *, T *, ptrdiff_t, int, int, unsigned char *, size_t, size_t, size_t, size_t, size_t, size_t, int,
                          T *const *, int, int)
{
    return false;
}
template <>
bool composite_rgb_groups<uint8_t, 3>(vkx_ctx *ctx, uint8_t *dst, ptrdiff_t dstride, int h, int w, unsigned char *base, size_t o0,
                                      size_t o1, size_t o2, size_t o3, size_t o4, size_t n_tiles, int tiles_x,
                                      uint8_t *const *pages, int n_pages, int tiles_pp)
{
    if ((w & 3) || (dstride & 3)) return false;
    if (pages) {
        for (int p = 0; p < n_pages; p++)
            if ((uintptr_t)pages[p] & 3) return false;
    } else if ((uintptr_t)dst & 3) {
        return false;
    }
    VKX_TIMED(ctx, "k_composite_rgb");
    // tile slots per workgroup: long runs where the launch has workgroups to spare, one tile each for a few pages (swept over 2 .. 64 pages of
    // 1 024 tiles with runs of 1 / 2 / 4 / 8: 8 pages 0.050 / 0.052 / 0.073 / 0.085 ms, 16 pages 0.086 / 0.084 / 0.090 / 0.124, 32 pages 0.160 / 0.154 / 0.153 / 0.161,
    // 64 pages 0.325 / - / - / 0.275)
    static const int run_env = getenv("VKX_RGB_RUN") ? atoi(getenv("VKX_RGB_RUN")) : 0;
    const int run = run_env > 0 ? std::min(run_env, kRgbRun) : (n_tiles >= 49152 ? kRgbRun : n_tiles >= 24576 ? 4 : n_tiles >= 12288 ? 2 : 1);
    k_composite_rgb<<<(unsigned)((n_tiles + run - 1) / run), 256, 0, ctx->stream>>>(
        dst, dstride, h, w, (const LayerDev<uint8_t> *)(base + o0), (const int *)(base + o1), (const int *)(base + o2),
        (const int *)(base + o3), tiles_x, pages ? (uint8_t *const *)(base + o4) : nullptr, tiles_pp, (int)n_tiles, run);
    return hipGetLastError() == hipSuccess;
}

// Host side of k_composite: bins `devl` (already validated, skippable layers removed) into tiles, stages the CSR and
// the layer records in ctx->misc, launches.
// `page_begin` (size n_pages + 1, or empty for a single destination): layers [page_begin[p], page_begin[p + 1]) belong to
// destination pages[p]; all destinations share h, w and the row pitch.
template <typename T, int CN>
int composite_launch(vkx_ctx *ctx, T *dst, int h, int w, ptrdiff_t dstride, const std::vector<LayerDev<T>> &devl,
                     T *const *pages = nullptr, const std::vector<int> &page_begin = std::vector<int>())
{
    const int tiles_x = (w + kTileW - 1) / kTileW, tiles_y = (h + kTileH - 1) / kTileH;
    const int tiles_pp = tiles_x * tiles_y, n_pages = pages ? (int)page_begin.size() - 1 : 1;
    if ((long long)tiles_pp * n_pages > 0x7fffffffLL) return VKX_ERR_UNSUPPORTED;
    std::vector<int> count((size_t)tiles_pp * n_pages, 0), layer_page(devl.size(), 0);
    for (size_t i = 0, p = 0; i < devl.size(); i++) {      // layers are grouped by page in ascending order
        while (pages && (int)i >= page_begin[p + 1]) p++;
        layer_page[i] = (int)p;
    }
    for (size_t i = 0; i < devl.size(); i++) {
        const LayerDev<T> &L = devl[i];
        const size_t base_t = (size_t)layer_page[i] * tiles_pp;
        for (int ty = L.up / kTileH; ty <= (L.up + L.height - 1) / kTileH; ty++)
            for (int tx = L.left / kTileW; tx <= (L.left + L.width - 1) / kTileW; tx++) count[base_t + (size_t)ty * tiles_x + tx]++;
    }
    std::vector<int> tile_ids, tile_begin, slot(count.size(), -1);
    int total = 0;
    for (size_t t = 0; t < count.size(); t++)
        if (count[t]) {
            slot[t] = (int)tile_ids.size();
            tile_ids.push_back((int)t);
            tile_begin.push_back(total);
            total += count[t];
        }
    tile_begin.push_back(total);
    std::vector<int> cursor(tile_begin.begin(), tile_begin.end() - 1), tile_layers((size_t)total);
    for (size_t i = 0; i < devl.size(); i++) {
        const LayerDev<T> &L = devl[i];
        const size_t base_t = (size_t)layer_page[i] * tiles_pp;
        for (int ty = L.up / kTileH; ty <= (L.up + L.height - 1) / kTileH; ty++)
            for (int tx = L.left / kTileW; tx <= (L.left + L.width - 1) / kTileW; tx++)
                tile_layers[(size_t)cursor[slot[base_t + (size_t)ty * tiles_x + tx]]++] = (int)i;   // ascending i per tile
    }
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o0 = 0, o1 = o0 + up(sizeof(LayerDev<T>) * devl.size()), o2 = o1 + up(sizeof(int) * tile_ids.size());
    const size_t o3 = o2 + up(sizeof(int) * tile_begin.size()), o4 = o3 + up(sizeof(int) * tile_layers.size());
    const size_t bytes = o4 + up(sizeof(T *) * (pages ? (size_t)n_pages : 0));
    // the tables travel through the ctx's page-locked descriptor ring: the launch returns without a stream synchronisation (a page
    // assembler issues one such call per page).  One destination (a page of the operator API): the kernel reads them in the ring, in
    // place -- a few records per tile, one dispatch less per page; a batch of pages (ChainBatch.set_layers: tens of thousands of
    // tiles) gets them as ONE copy into device memory
    void *ring = nullptr;
    int rc = vkx_desc_ring_take(ctx, bytes, &ring);
    if (rc) return rc;
    unsigned char *stage = (unsigned char *)ring;
    memcpy(stage + o0, devl.data(), sizeof(LayerDev<T>) * devl.size());
    memcpy(stage + o1, tile_ids.data(), sizeof(int) * tile_ids.size());
    memcpy(stage + o2, tile_begin.data(), sizeof(int) * tile_begin.size());
    memcpy(stage + o3, tile_layers.data(), sizeof(int) * tile_layers.size());
    if (pages) memcpy(stage + o4, pages, sizeof(T *) * (size_t)n_pages);
    unsigned char *base = pages ? nullptr : (unsigned char *)const_cast<void *>(vkx_ring_device_ptr(stage));
    if (!base) {
        if ((rc = vkx_scratch_reserve(ctx, &ctx->misc, bytes))) return rc;
        base = (unsigned char *)ctx->misc.ptr;
        VKX_HIP(hipMemcpyAsync(base, stage, bytes, hipMemcpyHostToDevice, ctx->stream));
    }
    if (composite_rgb_groups<T, CN>(ctx, dst, dstride, h, w, base, o0, o1, o2, o3, o4, tile_ids.size(), tiles_x, pages, n_pages, tiles_pp))
        return VKX_OK;
    { VKX_TIMED(ctx, "k_composite");
      k_composite<T, CN><<<(unsigned)tile_ids.size(), 256, 0, ctx->stream>>>(
          dst, dstride, h, w, (const LayerDev<T> *)(base + o0), (const int *)(base + o1), (const int *)(base + o2),
          (const int *)(base + o3), tiles_x, pages ? (T *const *)(base + o4) : nullptr, tiles_pp); }
    VKX_LAUNCH_CHECK();
    return VKX_OK;
}

template <typename LAYER>
int check_layers(const LAYER *layers, int n_layers, int h, int w)
{
    for (int i = 0; i < n_layers; i++) {
        const LAYER &l = layers[i];
        if (l.height < 0 || l.width < 0 || l.up < 0 || l.left < 0 || l.up + l.height > h || l.left + l.width > w) {
            vkx_set_error("layer %d: box (up=%d left=%d h=%d w=%d) outside the %dx%d destination", i, l.up, l.left,
                          l.height, l.width, h, w);
            return VKX_ERR_INVALID;
        }
        if (!l.alpha && (l.alpha_scalar < 0.0 || l.alpha_scalar > 1.0)) {
            vkx_set_error("alpha=%g is invalid.", l.alpha_scalar);
            return VKX_ERR_INVALID;
        }
        if (l.mode < VKX_FILL_PLAIN || l.mode > VKX_FILL_KEEP_MIN) {
            vkx_set_error("layer %d: unknown fill mode %d", i, l.mode);
            return VKX_ERR_INVALID;
        }
    }
    return VKX_OK;
}

} // namespace

template <typename T, typename LAYER>
bool to_layer_dev(const LAYER &l, LayerDev<T> *L)
{
    if (l.height == 0 || l.width == 0) return false;
    if (!l.alpha && l.alpha_scalar == 0.0) return false; // element/opt.py:143-144
    L->up = l.up; L->left = l.left; L->height = l.height; L->width = l.width;
    L->mask = l.mask; L->mask_stride = l.mask_stride;
    L->alpha = l.alpha; L->alpha_stride = l.alpha_stride_el;
    L->alpha_scalar = (float)l.alpha_scalar;
    L->copy = !l.alpha && l.alpha_scalar == 1.0;
    L->mode = l.mode;
    return true;
}

VKX_EXPORT int vkx_fill_u8_dev(vkx_ctx *ctx, uint8_t *dst, int h, int w, int cn, ptrdiff_t dst_stride,
                               const vkx_layer *layers, int n_layers)
{
    VKX_REQUIRE(ctx && dst, "NULL argument");
    VKX_REQUIRE(n_layers >= 0 && (n_layers == 0 || layers), "bad layer list");
    VKX_REQUIRE(cn == 1 || cn == 3 || cn == 4, "1, 3 or 4 channels");
    int rc = check_layers(layers, n_layers, h, w);
    if (rc) return rc;
    std::vector<LayerDev<uint8_t>> devl;
    devl.reserve((size_t)n_layers);
    for (int i = 0; i < n_layers; i++) {
        LayerDev<uint8_t> L;
        if (!to_layer_dev(layers[i], &L)) continue;
        L.value = layers[i].value; L.value_stride = layers[i].value_stride;
        for (int c = 0; c < 4; c++) L.value_const[c] = layers[i].value_const[c];
        devl.push_back(L);
    }
    if (devl.empty()) return VKX_OK;
    if (devl.size() == 1) {
        const LayerDev<uint8_t> &L = devl[0];
        dim3 block(64, 4), grid(vkx_blocks(L.width, 64), vkx_blocks(L.height, 4));
        switch (cn) {
        case 1: { VKX_TIMED(ctx, "k_fill"); k_fill<uint8_t, 1><<<grid, block, 0, ctx->stream>>>(dst, dst_stride, L); } break;
        case 3: { VKX_TIMED(ctx, "k_fill"); k_fill<uint8_t, 3><<<grid, block, 0, ctx->stream>>>(dst, dst_stride, L); } break;
        default: { VKX_TIMED(ctx, "k_fill"); k_fill<uint8_t, 4><<<grid, block, 0, ctx->stream>>>(dst, dst_stride, L); } break;
        }
        VKX_LAUNCH_CHECK();
        return VKX_OK;
    }
    switch (cn) {
    case 1: return composite_launch<uint8_t, 1>(ctx, dst, h, w, dst_stride, devl);
    case 3: return composite_launch<uint8_t, 3>(ctx, dst, h, w, dst_stride, devl);
    default: return composite_launch<uint8_t, 4>(ctx, dst, h, w, dst_stride, devl);
    }
}

// The layer lists of n_pages equally shaped uint8 destinations in ONE launch (a batch of pages assembled together: the
// per-page launch is latency bound -- 1024 workgroups for a 1024^2 page).  Page p takes layers[layer_begin[p] ..
// layer_begin[p + 1]) in order, exactly as vkx_fill_u8_dev(dsts[p], ...) would.
VKX_EXPORT int vkx_fill_u8_batch_dev(vkx_ctx *ctx, uint8_t *const *dsts_host, int n_pages, int h, int w, int cn,
                                     ptrdiff_t dst_stride, const vkx_layer *layers, const int32_t *layer_begin_host)
{
    VKX_REQUIRE(ctx && dsts_host && layer_begin_host, "NULL argument");
    VKX_REQUIRE(n_pages >= 1 && cn >= 1, "bad batch");
    VKX_REQUIRE(cn == 1 || cn == 3 || cn == 4, "1, 3 or 4 channels");
    VKX_REQUIRE(layer_begin_host[0] == 0, "layer_begin[0] must be 0");
    for (int p = 0; p < n_pages; p++) {
        VKX_REQUIRE(dsts_host[p] != nullptr, "NULL destination");
        VKX_REQUIRE(layer_begin_host[p + 1] >= layer_begin_host[p], "layer_begin must not decrease");
    }
    const int n_layers = layer_begin_host[n_pages];
    VKX_REQUIRE(n_layers == 0 || layers, "bad layer list");
    int rc = check_layers(layers, n_layers, h, w);
    if (rc) return rc;
    std::vector<LayerDev<uint8_t>> devl;
    std::vector<int> page_begin((size_t)n_pages + 1, 0);
    devl.reserve((size_t)n_layers);
    for (int p = 0; p < n_pages; p++) {
        for (int i = layer_begin_host[p]; i < layer_begin_host[p + 1]; i++) {
            LayerDev<uint8_t> L;
            if (!to_layer_dev(layers[i], &L)) continue;
            L.value = layers[i].value; L.value_stride = layers[i].value_stride;
            for (int c = 0; c < 4; c++) L.value_const[c] = layers[i].value_const[c];
            devl.push_back(L);
        }
        page_begin[(size_t)p + 1] = (int)devl.size();
    }
    if (devl.empty()) return VKX_OK;
    switch (cn) {
    case 1: return composite_launch<uint8_t, 1>(ctx, nullptr, h, w, dst_stride, devl, dsts_host, page_begin);
    case 3: return composite_launch<uint8_t, 3>(ctx, nullptr, h, w, dst_stride, devl, dsts_host, page_begin);
    default: return composite_launch<uint8_t, 4>(ctx, nullptr, h, w, dst_stride, devl, dsts_host, page_begin);
    }
}

VKX_EXPORT int vkx_fill_f32_dev(vkx_ctx *ctx, float *dst, int h, int w, ptrdiff_t dst_stride_el,
                                const vkx_layer_f32 *layers, int n_layers)
{
    VKX_REQUIRE(ctx && dst, "NULL argument");
    VKX_REQUIRE(n_layers >= 0 && (n_layers == 0 || layers), "bad layer list");
    int rc = check_layers(layers, n_layers, h, w);
    if (rc) return rc;
    std::vector<LayerDev<float>> devl;
    devl.reserve((size_t)n_layers);
    for (int i = 0; i < n_layers; i++) {
        LayerDev<float> L;
        if (!to_layer_dev(layers[i], &L)) continue;
        L.value = layers[i].value; L.value_stride = layers[i].value_stride_el;
        L.value_const[0] = layers[i].value_const;
        L.value_const[1] = L.value_const[2] = L.value_const[3] = 0.f;
        devl.push_back(L);
    }
    if (devl.empty()) return VKX_OK;
    if (devl.size() == 1) {
        const LayerDev<float> &L = devl[0];
        dim3 block(64, 4), grid(vkx_blocks(L.width, 64), vkx_blocks(L.height, 4));
        { VKX_TIMED(ctx, "k_fill"); k_fill<float, 1><<<grid, block, 0, ctx->stream>>>(dst, dst_stride_el, L); }
        VKX_LAUNCH_CHECK();
        return VKX_OK;
    }
    return composite_launch<float, 1>(ctx, dst, h, w, dst_stride_el, devl);
}
