// The fused geometric + photometric chain on gfx950: ONE launch for a ragged batch of independent RGB images.
//
//   k_chain_setup   one lane per grid cell of every image: inverse homography, cv.fillPoly edge table, and the
//                   binning of the cell into the destination tiles its bounding box (+ blur halo) touches.
//   k_chain_fused   one 512-lane workgroup per 64x64 destination tile:
//      A  the tile's candidate cells are pulled into LDS and rasterised into an LDS ownership tile with
//         ds_max ("the later cell in row-major order wins", grid_rendering/type.py:222-256);
//      C  every pixel of the tile + halo: inv_H * (x, y, 1) in double (the FMA chain of the reference's dgemm),
//         1/32-px quantisation, bilinear gather straight from HBM/L2 with two unaligned 8-byte loads;
//      D  separable 8.8 fixed-point Gaussian: horizontal pass LDS -> LDS (4 outputs per lane);
//      E  vertical pass, RGB -> HSV_FULL -> hue shift -> RGB, + int16 noise, clip, 12-byte stores.
//   The dense float map, the remapped image and the blurred image never exist in HBM: the kernel reads the
//   source image (and the noise plane, an API input) once and writes the result once.
//
// Arithmetic is identical to the single-purpose kernels in grid.hip / photo.hip, which stay the reference
// implementation inside the library and serve every shape this kernel does not take.
#include "vkx_internal.h"
#include "vkx_cell.h"

#include <float.h>
#include <stdlib.h>

namespace {

constexpr int T = 64;          // destination tile side
constexpr int RMAX = 3;        // blur radius limit of the fused path (ksize <= 7)
constexpr int EMAX = T + 2 * RMAX;
constexpr int NLDSCELL = 64;   // candidate cells whose records are cached in LDS at a time
constexpr int NTHREADS = 512;
constexpr int CGROUP = 3;      // pixels a lane maps + gathers together in phase C (memory-level parallelism)

// explicit global address space: pointers loaded from a descriptor in memory would otherwise be "flat"
#define VKX_GLOBAL __attribute__((address_space(1)))
typedef const uint8_t VKX_GLOBAL *gsrc_t;
typedef uint8_t VKX_GLOBAL *gdst_t;
typedef unsigned long long u64_u1 __attribute__((aligned(1)));
typedef uint32_t u32_u1 __attribute__((aligned(1)));

struct ItemDev {
    const uint8_t *src;
    uint8_t *dst;
    const int16_t *noise;
    const int32_t *sv, *dv;
    ptrdiff_t sstride, dstride, nstride;
    int sh, sw, dh, dw;
    int rows, cols;
    int tiles_x, tiles_y;
    int tile_base;     // index of this image's first tile in the batch-wide numbering
    int cell_base;     // index of this image's first cell in the batch-wide cell table
    int R;             // blur radius (0 = no blur)
    int hue_on, hue_delta;
    unsigned short kq[8];
};

struct TileBin {       // candidate cell rectangle of one tile: [rmin, rmax] x [cmin, cmax]
    int rmin, cmin;    // atomicMin, initialised to 0x7f7f7f7f
    int rmax1, cmax1;  // atomicMax of (index + 1), initialised to 0
};

__device__ __forceinline__ int find_item(const int *__restrict__ prefix, int n, int v)
{
    // largest i with prefix[i] <= v
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (prefix[mid] <= v) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__global__ void __launch_bounds__(256) k_chain_setup(const ItemDev *__restrict__ items, const int *__restrict__ cell_prefix,
                                                     int n_items, int total_cells, vkc::CellC *__restrict__ cells,
                                                     TileBin *__restrict__ bins)
{
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= total_cells) return;
    const int ii = find_item(cell_prefix, n_items, gid);
    const ItemDev &it = items[ii];
    const int cell = gid - it.cell_base;
    vkc::CellC rec;
    int xmin, xmax, ymin, ymax;
    vkc::build_cell(it.sv, it.dv, it.rows, it.cols, cell, rec, xmin, xmax, ymin, ymax);
    cells[gid] = rec;
    // bin into every tile whose halo-extended window meets the cell's bounding box
    const int r = cell / (it.cols - 1), c = cell - r * (it.cols - 1);
    int tx0 = (xmin - it.R) / T, tx1 = (xmax + it.R) / T, ty0 = (ymin - it.R) / T, ty1 = (ymax + it.R) / T;
    if (xmin - it.R < 0) tx0 = 0;
    if (ymin - it.R < 0) ty0 = 0;
    tx1 = min(tx1, it.tiles_x - 1);
    ty1 = min(ty1, it.tiles_y - 1);
    for (int ty = ty0; ty <= ty1; ty++)
        for (int tx = tx0; tx <= tx1; tx++) {
            TileBin *b = bins + it.tile_base + ty * it.tiles_x + tx;
            atomicMin(&b->rmin, r);
            atomicMin(&b->cmin, c);
            atomicMax(&b->rmax1, r + 1);
            atomicMax(&b->cmax1, c + 1);
        }
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

struct HsvLut {
    int sdiv[256];
    int hdiv[256];
};

__device__ __forceinline__ void hue_shift_px(const int *sdiv, const int *hdiv, int delta, int &r, int &g, int &b)
{
    // RGB -> HSV_FULL (integer LUT division)
    const int v = max(b, max(g, r)), vmin = min(b, min(g, r));
    const int diff = v - vmin;
    const int vr = v == r ? -1 : 0, vg = v == g ? -1 : 0;
    const int S = (diff * sdiv[v] + (1 << 11)) >> 12;
    int hh = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
    hh = (hh * hdiv[diff] + (1 << 11)) >> 12;
    hh += hh < 0 ? 256 : 0;
    int H = vkd::clamp_u8(hh);
    H = (H + delta) & 255;   // python modulo 256 of a sum that may be negative
    // HSV_FULL -> RGB (float32 scalar formula, no FMA)
    const float s = S * (1.0f / 255.0f);
    const float fv = v * (1.0f / 255.0f);
    float fb, fg, fr;
    if (s == 0) {
        fb = fg = fr = fv;
    } else {
        float h = (float)H * (6.0f / 256);
        int sector = (int)floorf(h);
        h -= sector;
        if ((unsigned)sector >= 6u) { sector = 0; h = 0.f; }
        const float t0 = fv;
        const float t1 = fv * (1.f - s);
        const float t2 = fv * (1.f - s * h);
        const float t3 = fv * (1.f - s * (1.f - h));
        switch (sector) {
        case 0: fb = t1; fg = t3; fr = t0; break;
        case 1: fb = t1; fg = t0; fr = t2; break;
        case 2: fb = t3; fg = t0; fr = t1; break;
        case 3: fb = t0; fg = t2; fr = t1; break;
        case 4: fb = t0; fg = t1; fr = t3; break;
        default: fb = t2; fg = t1; fr = t0; break;
        }
    }
    r = vkd::clamp_u8(vkd::cv_round(fr * 255.0f));
    g = vkd::clamp_u8(vkd::cv_round(fg * 255.0f));
    b = vkd::clamp_u8(vkd::cv_round(fb * 255.0f));
}

// Tile-local view the phases share.
struct TileGeom {
    int x0, y0, tw, th;       // the tile proper
    int ex0, ey0, ex1, ey1;   // tile + halo, clipped to the image
    int Ew, Eh;
};

__global__ void __launch_bounds__(NTHREADS) k_chain_fused(const ItemDev *__restrict__ items, const int *__restrict__ tile_prefix,
                                                          int n_items, int total_tiles,
                                                          const vkc::CellC *__restrict__ cells,
                                                          const TileBin *__restrict__ bins,
                                                          const HsvLut *__restrict__ lut, int phase_limit)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // LDS carve (all offsets multiples of 16)
    uint32_t *own = (uint32_t *)smem;                                   // [EMAX*EMAX] owner tag, then packed RGB
    uint2 *hb = (uint2 *)(smem + sizeof(uint32_t) * EMAX * EMAX);        // [EMAX*T] 3 x u16 horizontal sums
    vkc::CellC *lcell = (vkc::CellC *)((unsigned char *)hb + sizeof(uint2) * EMAX * T);  // [NLDSCELL]
    int *lsdiv = (int *)((unsigned char *)lcell + sizeof(vkc::CellC) * NLDSCELL);        // [256]
    int *lhdiv = lsdiv + 256;                                                            // [256]

    // XCD-aware tile order: consecutive workgroup ids land on different XCDs (id % 8); give every XCD a
    // contiguous run of tiles so neighbouring tiles (shared source rows, shared cells) meet in one L2.
    const int nwg = gridDim.x;
    const int per = (nwg + 7) >> 3;
    const int tile_id = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (tile_id >= total_tiles) return;
    const int tid = threadIdx.x;
    const int ii = find_item(tile_prefix, n_items, tile_id);
    const ItemDev &it = items[ii];
    const int tl = tile_id - it.tile_base;
    const int ty = tl / it.tiles_x, tx = tl - ty * it.tiles_x;
    const int R = it.R;
    const int dw = it.dw, dh = it.dh;
    TileGeom g;
    g.x0 = tx * T; g.y0 = ty * T;
    g.tw = min(T, dw - g.x0); g.th = min(T, dh - g.y0);
    g.ex0 = max(0, g.x0 - R); g.ey0 = max(0, g.y0 - R);
    g.ex1 = min(dw, g.x0 + g.tw + R); g.ey1 = min(dh, g.y0 + g.th + R);
    g.Ew = g.ex1 - g.ex0; g.Eh = g.ey1 - g.ey0;
    const int Ew = g.Ew, Eh = g.Eh, tw = g.tw, th = g.th;
    const int invEw = (1 << 20) / Ew + 1, invEh = (1 << 20) / Eh + 1, invTw = (1 << 20) / tw + 1;
    const bool quads = (tw & 3) == 0;   // the tile's rows split into 4-pixel groups (all but right-edge tiles)

    const gsrc_t src = (gsrc_t)it.src;
    const gdst_t dst = (gdst_t)it.dst;
    const int16_t VKX_GLOBAL *noise = (const int16_t VKX_GLOBAL *)it.noise;
    const ptrdiff_t sstride = it.sstride, dstride = it.dstride, nstride = it.nstride;
    const int sh = it.sh, sw = it.sw;

    // The noise of this lane's output pixels is needed last; ask for it first so HBM latency hides under the
    // whole kernel: two 4-pixel groups (2 x 12 int16 = 2 x 24 B) per lane.
    uint32_t nz[2][6];
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
        for (int k = 0; k < 6; k++) nz[u][k] = 0;
    if (noise && quads) {
        const int qpr = tw >> 2;
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int q = tid + u * NTHREADS;
            if (q < th * qpr) {
                const int cy = q / qpr, cq = q - cy * qpr;
                const int16_t VKX_GLOBAL *np_ = noise + (ptrdiff_t)(g.y0 + cy) * nstride + (ptrdiff_t)(g.x0 + cq * 4) * 3;
#pragma unroll
                for (int k = 0; k < 6; k++) nz[u][k] = *(const u32_u1 VKX_GLOBAL *)(np_ + 2 * k);
            }
        }
    }

    const TileBin bin = bins[tile_id];
    const int r0 = bin.rmin, c0 = bin.cmin;
    const int nr = max(0, bin.rmax1 - bin.rmin), ncol = max(0, bin.cmax1 - bin.cmin);
    const int nc = bin.rmax1 > 0 ? nr * ncol : 0;
    const int cw = it.cols - 1;   // cells per lattice row
    const vkc::CellC *gcell = cells + it.cell_base;

    // ---- A: clear the ownership tile, then rasterise the candidates chunk by chunk out of LDS
    for (int p = tid; p < Ew * Eh; p += NTHREADS) own[p] = 0;
    if (it.hue_on) {
        if (tid < 256) lsdiv[tid] = lut->sdiv[tid];
        else lhdiv[tid - 256] = lut->hdiv[tid - 256];
    }
    for (int base = 0; base < max(nc, 1); base += NLDSCELL) {
        const int cn_ = min(NLDSCELL, nc - base);
        if (base > 0) __syncthreads();            // the previous chunk is still being read
        {
            // 128-byte records, 8 lanes x 16 B per record
            const int rec = tid >> 3, part = tid & 7;
            if (rec < cn_) {
                const int k = base + rec;
                const int rr = k / ncol, cc = k - rr * ncol;
                const uint4 *s4 = (const uint4 *)(gcell + (r0 + rr) * cw + (c0 + cc));
                ((uint4 *)(lcell + rec))[part] = s4[part];
            }
        }
        __syncthreads();
        // A.1 interior: one (candidate, window row) pair per step; spans [ceil(xa), floor(xb)] of the x-sorted
        //     edge crossings (16.16 fixed point), clipped to the window
        for (int p = tid; p < cn_ * Eh; p += NTHREADS) {
            const int kk = (int)(((long long)p * invEh) >> 20), row = p - kk * Eh;
            const vkc::CellC &c = lcell[kk];
            const uint32_t tag = (uint32_t)(base + kk) + 1;
            const int y = g.ey0 + row;
            int xs[4], n = 0, xmin = INT_MAX, xmax = INT_MIN;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int a = (i + 3) & 3;
                const int ya = c.vy[a], yb = c.vy[i];
                xmin = min(xmin, (int)c.vx[i]);
                xmax = max(xmax, (int)c.vx[i]);
                const int e0 = min(ya, yb), e1 = max(ya, yb);
                if (e0 != e1 && e0 <= y && y < e1) xs[n++] = c.ex[i] + (y - e0) * c.edx[i];
            }
            if (n == 0) continue;
            for (int a = 1; a < n; a++) {
                const int v = xs[a];
                int b = a - 1;
                while (b >= 0 && xs[b] > v) { xs[b + 1] = xs[b]; b--; }
                xs[b + 1] = v;
            }
            const bool check = c.flags & 1;
            const double h6 = check ? c.H[6] : 0.0, h7 = check ? c.H[7] : 0.0;
            for (int a = 0; a + 1 < n; a += 2) {
                const int x1 = max(max((xs[a] + 65535) >> 16, xmin), g.ex0);
                const int x2 = min(min(xs[a + 1] >> 16, xmax), g.ex1 - 1);
                uint32_t *o = own + row * Ew - g.ex0;
                for (int x = x1; x <= x2; x++) {
                    if (check && fma(1.0, 1.0, fma(h7, (double)y, h6 * (double)x)) == 0) continue;
                    atomicMax(o + x, tag);
                }
            }
        }
        // A.2 outline: one (candidate, edge) pair per step; 8-connected Bresenham from the edge's left end,
        //     advanced incrementally (cv::LineIterator) over the part of the edge inside the window
        for (int p = tid; p < cn_ * 4; p += NTHREADS) {
            const int kk = p >> 2, i = p & 3;
            const vkc::CellC &c = lcell[kk];
            const uint32_t tag = (uint32_t)(base + kk) + 1;
            const int a = (i + 3) & 3;
            int lx = c.vx[a], ly = c.vy[a], rx = c.vx[i], ry = c.vy[i];
            if (rx < lx) { const int t1 = lx, t2 = ly; lx = rx; ly = ry; rx = t1; ry = t2; }
            const int dx = rx - lx, dy = ry - ly, ady = dy < 0 ? -dy : dy, sy = dy < 0 ? -1 : 1;
            const bool ymajor = ady > dx;
            const int dmaj = ymajor ? ady : dx, dmin = ymajor ? dx : ady;
            int k0, k1;   // range of major steps whose pixel can lie inside the window
            if (ymajor) {
                if (sy > 0) { k0 = max(0, g.ey0 - ly); k1 = min(ady, g.ey1 - 1 - ly); }
                else        { k0 = max(0, ly - (g.ey1 - 1)); k1 = min(ady, ly - g.ey0); }
            } else {
                k0 = max(0, g.ex0 - lx); k1 = min(dx, g.ex1 - 1 - lx);
            }
            if (k0 > k1) continue;
            int m = vkc::bres_minor(k0, dmaj, dmin);
            // LineIterator's error term after k0 steps: err = dmaj - 2 dmin (k0 + 1) + 2 dmaj m
            long long err = (long long)dmaj - 2LL * dmin * (k0 + 1) + 2LL * dmaj * m;
            const bool check = c.flags & 1;
            const double h6 = check ? c.H[6] : 0.0, h7 = check ? c.H[7] : 0.0;
            for (int s = k0; s <= k1; s++) {
                const int x = ymajor ? lx + m : lx + s;
                const int y = ymajor ? ly + sy * s : ly + sy * m;
                if (x >= g.ex0 && x < g.ex1 && y >= g.ey0 && y < g.ey1 &&
                    !(check && fma(1.0, 1.0, fma(h7, (double)y, h6 * (double)x)) == 0))
                    atomicMax(own + (y - g.ey0) * Ew + (x - g.ex0), tag);
                const bool step = err < 0;
                err += -2LL * dmin + (step ? 2LL * dmaj : 0LL);
                m += step ? 1 : 0;
            }
        }
    }
    __syncthreads();
    if (phase_limit == 1) return;

    // If the candidates did not fit one chunk, LDS now holds the LAST chunk; phase C wants chunk 0.
    if (nc > NLDSCELL) {
        const int rec = tid >> 3, part = tid & 7;
        if (rec < NLDSCELL) {
            const int rr = rec / ncol, cc = rec - rr * ncol;
            const uint4 *s4 = (const uint4 *)(gcell + (r0 + rr) * cw + (c0 + cc));
            ((uint4 *)(lcell + rec))[part] = s4[part];
        }
        __syncthreads();
    }

    // ---- C: source coordinates and bilinear gather for the tile + halo; packed RGB replaces the tag in LDS.
    //      CGROUP pixels per lane at a time so that their fp64 chains and their loads overlap.
    for (int p0 = tid; p0 < Ew * Eh; p0 += NTHREADS * CGROUP) {
        int X[CGROUP], Y[CGROUP];
#pragma unroll
        for (int u = 0; u < CGROUP; u++) {
            const int p = p0 + u * NTHREADS;
            X[u] = 0; Y[u] = 0;
            if (p < Ew * Eh) {
                const int ly = (int)(((long long)p * invEw) >> 20), lx = p - ly * Ew;
                const uint32_t o = own[p];
                if (o != 0) {
                    const int k = (int)o - 1;
                    double h[8];
                    if (k < NLDSCELL) {
#pragma unroll
                        for (int j = 0; j < 8; j++) h[j] = lcell[k].H[j];
                    } else {
                        const vkc::CellC VKX_GLOBAL *gc =
                            (const vkc::CellC VKX_GLOBAL *)(gcell + (r0 + k / ncol) * cw + (c0 + k % ncol));
#pragma unroll
                        for (int j = 0; j < 8; j++) h[j] = gc->H[j];
                    }
                    const double fx = (double)(g.ex0 + lx), fy = (double)(g.ey0 + ly);
                    const double nx = fma(h[2], 1.0, fma(h[1], fy, h[0] * fx));
                    const double ny = fma(h[5], 1.0, fma(h[4], fy, h[3] * fx));
                    const double de = fma(1.0, 1.0, fma(h[7], fy, h[6] * fx));
                    X[u] = vkd::cv_round((float)(nx / de) * 32.f);
                    Y[u] = vkd::cv_round((float)(ny / de) * 32.f);
                }
            }
        }
        unsigned long long ta[CGROUP], tb[CGROUP];
        bool fast[CGROUP];
#pragma unroll
        for (int u = 0; u < CGROUP; u++) {
            const int sx = vkd::sat_short(X[u] >> 5), sy = vkd::sat_short(Y[u] >> 5);
            fast[u] = sx >= 0 && sx + 2 < sw && sy >= 0 && sy + 1 < sh && (p0 + u * NTHREADS) < Ew * Eh;
            ta[u] = 0; tb[u] = 0;
            if (fast[u]) {
                // interior: the two 6-byte tap pairs come in as two unaligned 8-byte loads (3 sx + 8 <= 3 sw)
                const gsrc_t q0 = src + (ptrdiff_t)sy * sstride + (ptrdiff_t)sx * 3;
                ta[u] = *(const u64_u1 VKX_GLOBAL *)q0;
                tb[u] = *(const u64_u1 VKX_GLOBAL *)(q0 + sstride);
            }
        }
#pragma unroll
        for (int u = 0; u < CGROUP; u++) {
            const int p = p0 + u * NTHREADS;
            if (p >= Ew * Eh) continue;
            const int fx = X[u] & 31, fy = Y[u] & 31;
            const int w00 = (32 - fy) * (32 - fx), w01 = (32 - fy) * fx, w10 = fy * (32 - fx), w11 = fy * fx;
            uint32_t out = 0;
            if (fast[u]) {
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const int v0 = (int)((ta[u] >> (8 * k)) & 0xff), v1 = (int)((ta[u] >> (8 * (k + 3))) & 0xff);
                    const int v2 = (int)((tb[u] >> (8 * k)) & 0xff), v3 = (int)((tb[u] >> (8 * (k + 3))) & 0xff);
                    out |= (uint32_t)((v0 * w00 + v1 * w01 + v2 * w10 + v3 * w11 + 512) >> 10) << (8 * k);
                }
            } else {
                uint8_t px[3];
                vkd::sample_u8<3>(it.src, sh, sw, sstride, X[u], Y[u], px);
                out = (uint32_t)px[0] | ((uint32_t)px[1] << 8) | ((uint32_t)px[2] << 16);
            }
            own[p] = out;
        }
    }
    __syncthreads();
    if (phase_limit == 2) return;

    // ---- D: horizontal 8.8 pass over the rows of the window, for the tile's own columns
    if (R > 0) {
        const int K = 2 * R + 1;
        uint32_t kq[2 * RMAX + 1];
#pragma unroll
        for (int i = 0; i < 2 * RMAX + 1; i++) kq[i] = i < K ? it.kq[i] : 0;
        const bool inner_x = g.x0 - R >= 0 && g.x0 + tw + R <= dw;   // no reflection at the left / right border
        if (quads && inner_x) {
            // a lane produces 4 adjacent outputs from 4 + 2R adjacent inputs
            const int qpr = tw >> 2;
            for (int q = tid; q < Eh * qpr; q += NTHREADS) {
                const int ly = q / qpr, cq = q - ly * qpr;
                const uint32_t *in = own + ly * Ew + (g.x0 - g.ex0) + cq * 4 - R;
                uint32_t px[4 + 2 * RMAX];
#pragma unroll
                for (int i = 0; i < 4 + 2 * RMAX; i++) px[i] = i < 4 + 2 * R ? in[i] : 0;
#pragma unroll
                for (int o = 0; o < 4; o++) {
                    uint32_t a0 = 0, a1 = 0, a2 = 0;
#pragma unroll
                    for (int i = 0; i < 2 * RMAX + 1; i++) {
                        if (i < K) {
                            const uint32_t v = px[o + i];
                            a0 += kq[i] * (v & 0xff);
                            a1 += kq[i] * ((v >> 8) & 0xff);
                            a2 += kq[i] * ((v >> 16) & 0xff);
                        }
                    }
                    hb[ly * tw + cq * 4 + o] = make_uint2(a0 | (a1 << 16), a2);
                }
            }
        } else {
            for (int q = tid; q < Eh * tw; q += NTHREADS) {
                const int ly = (int)(((long long)q * invTw) >> 20), cx = q - ly * tw;
                const int gx = g.x0 + cx;
                uint32_t a0 = 0, a1 = 0, a2 = 0;
                for (int i = 0; i < K; i++) {
                    const int xx = reflect101(gx + i - R, dw) - g.ex0;
                    const uint32_t v = own[ly * Ew + xx];
                    a0 += kq[i] * (v & 0xff);
                    a1 += kq[i] * ((v >> 8) & 0xff);
                    a2 += kq[i] * ((v >> 16) & 0xff);
                }
                hb[q] = make_uint2(a0 | (a1 << 16), a2);
            }
        }
        __syncthreads();
    }
    if (phase_limit == 3) return;

    // ---- E: vertical pass, hue shift, noise, store
    const int K = 2 * R + 1;
    uint32_t kq[2 * RMAX + 1];
#pragma unroll
    for (int i = 0; i < 2 * RMAX + 1; i++) kq[i] = (R > 0 && i < K) ? it.kq[i] : 0;
    const bool hue_on = it.hue_on != 0;
    const int hue_delta = it.hue_delta;
    if (quads) {
        const int qpr = tw >> 2;
        const bool inner_y = g.y0 - R >= 0 && g.y0 + th + R <= dh;
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int q = tid + u * NTHREADS;
            if (q >= th * qpr) continue;
            const int cy = q / qpr, cq = q - cy * qpr;
            const int gy = g.y0 + cy, gx = g.x0 + cq * 4;
            int rgb[4][3];
            if (R > 0) {
                uint32_t acc[4][3];
#pragma unroll
                for (int o = 0; o < 4; o++) { acc[o][0] = 0; acc[o][1] = 0; acc[o][2] = 0; }
#pragma unroll
                for (int j = 0; j < 2 * RMAX + 1; j++) {
                    if (j < K) {
                        const int yy = (inner_y ? gy + j - R : reflect101(gy + j - R, dh)) - g.ey0;
                        const uint2 *row = hb + yy * tw + cq * 4;
#pragma unroll
                        for (int o = 0; o < 4; o++) {
                            const uint2 h = row[o];
                            acc[o][0] += kq[j] * (h.x & 0xffff);
                            acc[o][1] += kq[j] * (h.x >> 16);
                            acc[o][2] += kq[j] * (h.y & 0xffff);
                        }
                    }
                }
#pragma unroll
                for (int o = 0; o < 4; o++)
#pragma unroll
                    for (int c = 0; c < 3; c++) rgb[o][c] = (int)((acc[o][c] + 32768u) >> 16);
            } else {
#pragma unroll
                for (int o = 0; o < 4; o++) {
                    const uint32_t v = own[(gy - g.ey0) * Ew + (gx + o - g.ex0)];
                    rgb[o][0] = v & 0xff; rgb[o][1] = (v >> 8) & 0xff; rgb[o][2] = (v >> 16) & 0xff;
                }
            }
            if (hue_on) {
#pragma unroll
                for (int o = 0; o < 4; o++) hue_shift_px(lsdiv, lhdiv, hue_delta, rgb[o][0], rgb[o][1], rgb[o][2]);
            }
            if (noise) {
#pragma unroll
                for (int e = 0; e < 12; e++) {
                    const uint32_t w = nz[u][e >> 1];
                    const int16_t nv = (int16_t)((e & 1) ? (w >> 16) : (w & 0xffff));
                    int &ch = rgb[e / 3][e % 3];
                    ch = vkd::clamp_u8((int16_t)((int16_t)ch + nv));
                }
            }
            uint32_t w3[3];
            {
                uint8_t by[12];
#pragma unroll
                for (int e = 0; e < 12; e++) by[e] = (uint8_t)rgb[e / 3][e % 3];
#pragma unroll
                for (int k = 0; k < 3; k++)
                    w3[k] = (uint32_t)by[4 * k] | ((uint32_t)by[4 * k + 1] << 8) | ((uint32_t)by[4 * k + 2] << 16) |
                            ((uint32_t)by[4 * k + 3] << 24);
            }
            gdst_t d = dst + (ptrdiff_t)gy * dstride + (ptrdiff_t)gx * 3;
#pragma unroll
            for (int k = 0; k < 3; k++) *(u32_u1 VKX_GLOBAL *)(d + 4 * k) = w3[k];
        }
    } else {
        for (int q = tid; q < th * tw; q += NTHREADS) {
            const int cy = (int)(((long long)q * invTw) >> 20), cx = q - cy * tw;
            const int gx = g.x0 + cx, gy = g.y0 + cy;
            int r, gg, b;
            if (R > 0) {
                uint32_t a0 = 0, a1 = 0, a2 = 0;
                for (int j = 0; j < K; j++) {
                    const int yy = reflect101(gy + j - R, dh) - g.ey0;
                    const uint2 h = hb[yy * tw + cx];
                    a0 += kq[j] * (h.x & 0xffff);
                    a1 += kq[j] * (h.x >> 16);
                    a2 += kq[j] * (h.y & 0xffff);
                }
                r = (int)((a0 + 32768u) >> 16);
                gg = (int)((a1 + 32768u) >> 16);
                b = (int)((a2 + 32768u) >> 16);
            } else {
                const uint32_t v = own[(gy - g.ey0) * Ew + (gx - g.ex0)];
                r = v & 0xff; gg = (v >> 8) & 0xff; b = (v >> 16) & 0xff;
            }
            if (hue_on) hue_shift_px(lsdiv, lhdiv, hue_delta, r, gg, b);
            if (noise) {
                const int16_t VKX_GLOBAL *np_ = noise + (ptrdiff_t)gy * nstride + (ptrdiff_t)gx * 3;
                r = vkd::clamp_u8((int16_t)((int16_t)r + np_[0]));
                gg = vkd::clamp_u8((int16_t)((int16_t)gg + np_[1]));
                b = vkd::clamp_u8((int16_t)((int16_t)b + np_[2]));
            }
            gdst_t d = dst + (ptrdiff_t)gy * dstride + (ptrdiff_t)gx * 3;
            d[0] = (uint8_t)r; d[1] = (uint8_t)gg; d[2] = (uint8_t)b;
        }
    }
}

constexpr size_t kFusedLds = sizeof(uint32_t) * EMAX * EMAX + sizeof(uint2) * EMAX * T + sizeof(vkc::CellC) * NLDSCELL +
                             sizeof(int) * 512;

} // namespace

// Returns VKX_ERR_UNSUPPORTED (without setting an error) when the batch has a shape the fused path does not
// take; the caller then runs the per-stage kernels.
int vkx_chain_fused_try(vkx_ctx *ctx, const vkx_chain_item *items, int n_items)
{
    if (n_items <= 0) return VKX_OK;
    std::vector<ItemDev> dev(n_items);
    std::vector<int> prefix(2 * (size_t)n_items + 2);
    int *tile_prefix = prefix.data(), *cell_prefix = prefix.data() + n_items + 1;
    long long tiles = 0, ncells = 0;
    for (int i = 0; i < n_items; i++) {
        const vkx_chain_item &it = items[i];
        if (it.blur_ksize > 2 * RMAX + 1 || (it.blur_ksize > 1 && (it.blur_ksize & 1) == 0)) return VKX_ERR_UNSUPPORTED;
        if (it.sh > 32767 || it.sw > 32767 || it.dh > 32767 || it.dw > 32767) return VKX_ERR_UNSUPPORTED;
        if (it.rows < 2 || it.cols < 2) return VKX_ERR_UNSUPPORTED;
        if (it.blur_ksize > 1 && (it.dh == 1 || it.dw == 1)) return VKX_ERR_UNSUPPORTED; // kernel collapses per axis
        ItemDev &d = dev[i];
        d.src = it.src; d.dst = it.dst; d.noise = it.noise; d.sv = it.src_vertices; d.dv = it.dst_vertices;
        d.sstride = it.src_stride; d.dstride = it.dst_stride; d.nstride = it.noise_stride_el;
        d.sh = it.sh; d.sw = it.sw; d.dh = it.dh; d.dw = it.dw; d.rows = it.rows; d.cols = it.cols;
        d.tiles_x = (it.dw + T - 1) / T; d.tiles_y = (it.dh + T - 1) / T;
        d.tile_base = (int)tiles; d.cell_base = (int)ncells;
        d.R = it.blur_ksize > 1 ? it.blur_ksize / 2 : 0;
        d.hue_on = it.hue_enabled; d.hue_delta = it.hue_delta;
        for (int k = 0; k < 8; k++) d.kq[k] = 0;
        if (d.R > 0 && vkx_gaussian_kernel_q8_host(it.blur_ksize, it.blur_sigma, d.kq)) return VKX_ERR_UNSUPPORTED;
        tile_prefix[i] = (int)tiles; cell_prefix[i] = (int)ncells;
        tiles += (long long)d.tiles_x * d.tiles_y;
        ncells += (long long)(it.rows - 1) * (it.cols - 1);
        if (tiles > 0x3fffffff || ncells > 0x3fffffff) return VKX_ERR_UNSUPPORTED;
    }
    tile_prefix[n_items] = (int)tiles; cell_prefix[n_items] = (int)ncells;

    // device scratch: cell table, tile bins, item descriptors + prefix arrays, HSV tables
    int rc;
    if ((rc = vkx_scratch_reserve(ctx, &ctx->cells, sizeof(vkc::CellC) * (size_t)ncells))) return rc;
    if ((rc = vkx_scratch_reserve(ctx, &ctx->owner, sizeof(TileBin) * (size_t)tiles))) return rc;
    const size_t items_bytes = sizeof(ItemDev) * (size_t)n_items, prefix_bytes = sizeof(int) * prefix.size();
    const size_t items_off = 0, prefix_off = (items_bytes + 255) & ~(size_t)255;
    if ((rc = vkx_scratch_reserve(ctx, &ctx->misc, prefix_off + prefix_bytes))) return rc;
    const HsvLut *lut = nullptr;
    if ((rc = vkx_hsv_tables(ctx, (const void **)&lut))) return rc;
    unsigned char *misc = (unsigned char *)ctx->misc.ptr;
    // the host vectors die with this frame, so the upload is completed before returning from this block
    VKX_HIP(hipMemcpyAsync(misc + items_off, dev.data(), items_bytes, hipMemcpyHostToDevice, ctx->stream));
    VKX_HIP(hipMemcpyAsync(misc + prefix_off, prefix.data(), prefix_bytes, hipMemcpyHostToDevice, ctx->stream));
    VKX_HIP(hipStreamSynchronize(ctx->stream));
    const ItemDev *d_items = (const ItemDev *)(misc + items_off);
    const int *d_tile_prefix = (const int *)(misc + prefix_off), *d_cell_prefix = d_tile_prefix + n_items + 1;
    TileBin *bins = (TileBin *)ctx->owner.ptr;
    vkc::CellC *cells = (vkc::CellC *)ctx->cells.ptr;

    // bins: mins start at 0x7f7f7f7f, maxs at 0 -> one strided 2D memset per half
    VKX_HIP(hipMemset2DAsync(bins, sizeof(TileBin), 0x7f, 8, (size_t)tiles, ctx->stream));
    VKX_HIP(hipMemset2DAsync((unsigned char *)bins + 8, sizeof(TileBin), 0x00, 8, (size_t)tiles, ctx->stream));
    { VKX_TIMED(ctx, "k_chain_setup"); k_chain_setup<<<vkx_blocks((size_t)ncells, 256), 256, 0, ctx->stream>>>(d_items, d_cell_prefix, n_items, (int)ncells, cells, bins); }
    VKX_LAUNCH_CHECK();
    static bool attr_set = false;
    if (!attr_set) {
        VKX_HIP(hipFuncSetAttribute((const void *)k_chain_fused, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFusedLds));
        attr_set = true;
    }
    const int nwg = (int)(((tiles + 7) / 8) * 8);
    // profiling aid: VKX_FUSED_PHASES=1|2|3 stops the kernel after phase A | C | D
    static const int phase_limit = [] { const char *e = getenv("VKX_FUSED_PHASES"); return e ? atoi(e) : 0; }();
    { VKX_TIMED(ctx, "k_chain_fused"); k_chain_fused<<<nwg, NTHREADS, kFusedLds, ctx->stream>>>(d_items, d_tile_prefix, n_items, (int)tiles, cells, bins, lut, phase_limit); }
    VKX_LAUNCH_CHECK();
    return VKX_OK;
}
