// The fused geometric + photometric chain on gfx950: ONE launch for a ragged batch of independent RGB images.
//
//   k_chain_setup      one lane per grid cell of every image: closed-form inverse homography, cv.fillPoly edge table, and
//                      the binning of the cell into the destination tiles its bounding box (+ blur halo) touches.
//   k_chain_setup_svd  the few cells whose homography needs the Jacobi-SVD least squares (workspace in LDS).
//   k_chain_fused      one 512-lane workgroup (8 wavefronts) per destination tile.  The tile's WINDOW is 64 x 64
//                      pixels -- one wavefront spans a window row, lane = column -- and holds the tile proper
//                      (((64 - 2R) & ~3)^2 pixels, R = blur radius) plus its halo:
//      A  the tile's candidate cells are pulled into LDS and rasterised into an LDS ownership plane with
//         ds_max ("the later cell in row-major order wins", grid_rendering/type.py:222-256);
//      C  per window-row pair: inv_H * (x, y, 1) in double (the FMA chain of the reference's dgemm), both quotients
//         through one refined reciprocal whose float32 rounding is proven per wavefront (exact IEEE division
//         otherwise), 1/32-px quantisation, bilinear gather straight from HBM/L2 with two unaligned 8-byte loads per
//         pixel, v_dot4_u32_u8 horizontal pairs;
//      D  horizontal 8.8 fixed-point Gaussian pass with wavefront shuffles (the remapped pixel never leaves
//         its register), results to LDS in place of the ownership tags as (row | next row << 16) planes per channel;
//      E  vertical pass out of LDS as v_dot2_u32_u16 (two taps per instruction), RGB -> HSV_FULL -> hue shift -> RGB
//         (one byte permute per pixel selects the sector's channels), + int16 noise, clip, optional line_streak
//         blends; neighbouring lanes pack 4 pixels into 3 dwords with one shuffle and store.
//      Variants of the tile body (template KIND): interior windows (border logic compiled out), empty tiles (no cell
//      reaches the window: constant colour), and the element mode of k_tile_remap.
//   k_tile_remap       the same tile machinery for vkx_grid_remap: 1-4 elements of any supported type (uint8 x 1 / 3 / 4
//                      channels, float32) gathered through one lattice, no photometric stage.
//   The dense float map, the remapped image and the blurred image never exist in HBM: the kernel reads the
//   source image (and the noise plane, an API input) once and writes the result once.
//
// Arithmetic is identical to the single-purpose kernels in grid.hip / photo.hip, which stay the reference
// implementation inside the library and serve every shape these kernels do not take.
#include "vkx_internal.h"
#include "vkx_cell.h"
#include "vkx_color.h"

#include <float.h>
#include <memory>
#include <stdlib.h>
#include <string.h>

namespace {

constexpr int W = 64;          // window side = wavefront width
constexpr int RMAX = 3;        // blur radius limit of the fused path (ksize <= 7)
constexpr int NLDSCELL = 64;   // candidate cells whose records are cached in LDS at a time
#ifndef VKX_FUSED_NWAVES
#define VKX_FUSED_NWAVES 8
#endif
constexpr int NWAVES = VKX_FUSED_NWAVES;    // wavefronts per workgroup: 8 -> a 64 x 64 window, 16 -> 64 wide x 128 high
constexpr int NTHREADS = 64 * NWAVES;
constexpr int ROWS_PER_WAVE = 8;
constexpr int WH = NWAVES * ROWS_PER_WAVE;  // window height
// side of the tile proper: the window minus the blur halo, rounded down to whole 4-pixel (12-byte) store groups
__host__ __device__ constexpr int tile_side(int R) { return (W - 2 * R) & ~3; }
// height of the tile proper: the window minus the blur halo (rows need no store-group rounding)
__host__ __device__ constexpr int tile_height(int R) { return WH == W ? tile_side(R) : WH - 2 * R; }
#ifndef VKX_FUSED_CGROUP
#define VKX_FUSED_CGROUP 2
#endif
constexpr int CGROUP = VKX_FUSED_CGROUP;      // window rows a wavefront maps + gathers together (memory-level parallelism)
#ifndef VKX_FUSED_WAVES_PER_EU
#define VKX_FUSED_WAVES_PER_EU 8  // register budget: 64 VGPRs -> 8 waves/SIMD = 4 workgroups per CU (136 KB of LDS)
#endif

// explicit global address space: pointers loaded from a descriptor in memory would otherwise be "flat"
#define VKX_GLOBAL __attribute__((address_space(1)))
typedef const uint8_t VKX_GLOBAL *gsrc_t;
typedef uint8_t VKX_GLOBAL *gdst_t;
typedef unsigned long long u64_u1 __attribute__((aligned(1)));
typedef uint32_t u32_u1 __attribute__((aligned(1)));
typedef uint16_t u16_u1 __attribute__((aligned(1)));

struct ItemDev {
    const uint8_t *src;
    uint8_t *dst;
    const int16_t *noise;
    const int32_t *sv, *dv;
    ptrdiff_t sstride, dstride, nstride;
    int sh, sw, dh, dw;
    int rows, cols;
    int tiles_x, tiles_y;
    int cell_base;     // index of this image's first cell in the batch-wide cell table
    int R;             // blur radius (0 = no blur); the tile proper is (64 - 2R) pixels wide and high
    int hue_on, hue_delta;
    unsigned short kq[8];
    // line_streak stage (photometric/streak.py:56-99), after the noise
    int streak_on, streak_step, streak_thickness, streak_dash, streak_dash_step, streak_dash_gap;
    int streak_vert, streak_hori, streak_copy;
    int streak_color[3];
    float streak_alpha;
    // element mode (k_tile_remap): the same tile machinery gathers up to four elements of any supported type through
    // the shared lattice instead of running the RGB chain
    int n_elems, pad_;
    // noise as the generator's tile buffer (VKX_NP_NORMAL_TILES, nprand.hip): `noise` then points at the slots
    const uint2 *noise_table;     // [noise_tiles + 1] x (index of the tile's first sample, its first valid slot element)
    int noise_tiled, noise_tiles, noise_slot;
    float noise_tiles_per_sample;
    const uint2 *noise_rows;      // [dh][tiles_x] x (slot offset of the row's first sample in that tile column, samples before the next
                                  // generator tile begins | step of the offset beyond them << 16): k_chain_noise_rows
    struct Elem {
        const void *src;
        void *dst;
        ptrdiff_t sstride, dstride;   // bytes (uint8) / elements (float32)
        int cn, is_f32;
    } el[4];
};

// The element descriptors of k_tile_remap travel as a KERNEL ARGUMENT: read from the descriptor in global memory, every field was
// re-loaded after every store of the row loop (a store through el.dst may alias the descriptor as far as the compiler can tell:
// 120 scalar loads per wavefront, 66 % of the wavefront cycles waiting); kernel arguments are constant memory.
struct ElemPack {
    ItemDev::Elem el[4];
    int n;
};

struct TileBin {       // candidate cell rectangle of one tile: [rmin, rmax] x [cmin, cmax]
    int rmin, cmin;    // atomicMin, initialised to 0x7f7f7f7f
    int rmax1, cmax1;  // atomicMax of (index + 1), initialised to 0
};

// The raster half of a cell record as it sits in LDS (bytes 64..119 of vkc::CellC).
struct CellR {
    int ex[4], edx[4];
    short vx[4], vy[4];
    int flags, pad;
};
static_assert(sizeof(CellR) == 56, "CellR mirrors the tail of CellC");

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

constexpr int kLocalBins = 1024;   // tile bins a setup block aggregates in LDS before touching the global ones

__global__ void __launch_bounds__(256) k_chain_setup(const ItemDev *__restrict__ items, const int *__restrict__ cell_prefix,
                                                     int n_items, int total_cells, int slots,
                                                     vkc::CellC *__restrict__ cells, TileBin *__restrict__ bins,
                                                     int *__restrict__ deferred /* [0] = count, then cell ids */)
{
    // Device-scope atomics are the cost of this kernel (every cell updates the candidate rectangle of the 1 - 4 tiles it
    // touches, and neighbouring cells hit the same bins).  A block's 256 consecutive cells cover a few tile rows of one
    // image: their updates meet in LDS first and every touched bin is flushed once.
    __shared__ int lbins[kLocalBins * 4];
    __shared__ int s_item, s_tiles_x, s_ty_min, s_ty_max;
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const bool active = gid < total_cells;
    int ii = -1, r = 0, c = 0, tx0 = 0, tx1 = -1, ty0 = 0, ty1 = -1, tiles_x = 1;
    if (active) {
        ii = find_item(cell_prefix, n_items, gid);
        const ItemDev &it = items[ii];
        const int cell = gid - it.cell_base;
        vkc::CellC rec;
        int xmin, xmax, ymin, ymax;
        // closed-form homography here; cells whose quads have collinear vertices go to k_chain_setup_svd
        if (vkc::build_cell<vkc::kCellDirectOnly>(it.sv, it.dv, it.rows, it.cols, cell, rec, xmin, xmax, ymin, ymax)) cells[gid] = rec;
        else deferred[1 + atomicAdd(&deferred[0], 1)] = gid;
        // the cell belongs to every tile whose window [t*Tw - R, t*Tw + Tw + R) meets its bounding box
        r = cell / (it.cols - 1); c = cell - r * (it.cols - 1);
        const int Tw = tile_side(it.R), Th = tile_height(it.R);
        tx0 = (xmin - it.R) / Tw; tx1 = (xmax + it.R) / Tw; ty0 = (ymin - it.R) / Th; ty1 = (ymax + it.R) / Th;
        if (xmin - it.R < 0) tx0 = 0;
        if (ymin - it.R < 0) ty0 = 0;
        tiles_x = it.tiles_x;
        tx1 = min(tx1, tiles_x - 1);
        ty1 = min(ty1, it.tiles_y - 1);
    }
    if (threadIdx.x == 0) { s_item = ii; s_tiles_x = tiles_x; s_ty_min = INT_MAX; s_ty_max = -1; }
    __syncthreads();
    const bool local = active && ii == s_item && ty0 <= ty1 && tx0 <= tx1;   // cells of the block's first image
    if (local) { atomicMin(&s_ty_min, ty0); atomicMax(&s_ty_max, ty1); }
    __syncthreads();
    const int tymin = s_ty_min, nb = s_ty_max >= tymin ? (s_ty_max - tymin + 1) * s_tiles_x : 0;
    const bool use_lds = nb > 0 && nb <= kLocalBins;
    if (use_lds)
        for (int i = threadIdx.x; i < nb; i += 256) {
            lbins[4 * i] = 0x7f7f7f7f; lbins[4 * i + 1] = 0x7f7f7f7f; lbins[4 * i + 2] = 0; lbins[4 * i + 3] = 0;
        }
    __syncthreads();
    for (int ty = ty0; ty <= ty1; ty++)
        for (int tx = tx0; tx <= tx1; tx++) {
            if (local && use_lds) {
                int *b = lbins + 4 * ((ty - tymin) * tiles_x + tx);
                atomicMin(b, r); atomicMin(b + 1, c); atomicMax(b + 2, r + 1); atomicMax(b + 3, c + 1);
            } else {
                TileBin *b = bins + (size_t)ii * slots + ty * tiles_x + tx;
                atomicMin(&b->rmin, r);
                atomicMin(&b->cmin, c);
                atomicMax(&b->rmax1, r + 1);
                atomicMax(&b->cmax1, c + 1);
            }
        }
    __syncthreads();
    if (use_lds)
        for (int i = threadIdx.x; i < nb; i += 256) {
            if (lbins[4 * i + 2] == 0) continue;       // no cell of this block reached the tile
            TileBin *b = bins + (size_t)s_item * slots + tymin * s_tiles_x + i;
            atomicMin(&b->rmin, lbins[4 * i]);
            atomicMin(&b->cmin, lbins[4 * i + 1]);
            atomicMax(&b->rmax1, lbins[4 * i + 2]);
            atomicMax(&b->cmax1, lbins[4 * i + 3]);
        }
}

// The deferred cells (Jacobi SVD least squares, 1.2 KB of workspace per lane): normally none.
__global__ void __launch_bounds__(64) k_chain_setup_svd(const ItemDev *__restrict__ items, const int *__restrict__ cell_prefix,
                                                        int n_items, vkc::CellC *__restrict__ cells,
                                                        const int *__restrict__ deferred)
{
    // eight lanes per cell (vkc::homography_jacobi_group8): registers only, no workspace
    const int n = deferred[0];
    const int sub = threadIdx.x & 7;
    for (int j = blockIdx.x * 8 + (threadIdx.x >> 3); j < n; j += gridDim.x * 8) {
        const int gid = deferred[1 + j];
        const ItemDev &it = items[find_item(cell_prefix, n_items, gid)];
        vkc::CellC rec;
        int xmin, xmax, ymin, ymax;
        vkc::build_cell<vkc::kCellJacobiGroup8>(it.sv, it.dv, it.rows, it.cols, gid - it.cell_base, rec, xmin, xmax, ymin, ymax);
        if (sub == 0) cells[gid] = rec;
    }
}

// floor(n / d) for 0 <= n < 2^21, d >= 1 through the float32 reciprocal `rcp_d` = 1 / d (any 1-ulp reciprocal will do): the
// product (n + 0.5) * rcp_d is off by less than 0.5 / d, the distance of (n + 0.5) / d from the nearest integer, and one
// correction step makes the result independent of that bound.  Replaces the ~25-instruction integer division sequence at
// the head of every tile (tile row / column, candidate row / column).
__device__ __forceinline__ int div_small(int n, int d, float rcp_d)
{
    int q = (int)(((float)n + 0.5f) * rcp_d);
    const int r = n - q * d;
    q += r >= d ? 1 : 0;
    q -= r < 0 ? 1 : 0;
    return q;
}

// a * b + c on the low 24 bits of a and b (full rate)
__device__ __forceinline__ uint32_t mad_u24(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// Multiply-adds with a wave-uniform operand (a blur weight, a rounding constant) are written with the compiler's own
// intrinsics, never as inline assembly with an "s" operand: the compiler keeps such values in SGPRs, may spill them to VGPR
// lanes and reload them with v_readlane right before the use, and gfx950 needs wait states between a VALU write of an SGPR
// and a VALU read of it -- wait states the hazard recogniser inserts for its own instructions only, not for the text of an
// asm statement (seen as sporadic wrong horizontal sums in the 3-tap variant).
__device__ __forceinline__ uint32_t mad_u24_ks(uint32_t k_uniform, uint32_t b, uint32_t c)
{
    return __umul24(k_uniform, b) + c;
}
__device__ __forceinline__ uint32_t mad_u24_cs(uint32_t a, uint32_t b, uint32_t c_uniform)
{
    return __umul24(a, b) + c_uniform;
}

// c + a.lo * b.lo + a.hi * b.hi on unsigned 16-bit halves (v_dot2_u32_u16)
typedef unsigned short ushort2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t dot2_u16_ks(uint32_t k_uniform, uint32_t b, uint32_t c)
{
    return __builtin_amdgcn_udot2(__builtin_bit_cast(ushort2_t, k_uniform), __builtin_bit_cast(ushort2_t, b), c, false);
}

// a wave-uniform 64-bit value, moved into SGPRs
__device__ __forceinline__ uint64_t uniform64(uint64_t v)
{
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// nx / de and ny / de, both correctly rounded: the instruction sequence the compiler emits for an IEEE double division
// (v_div_scale, v_rcp + two Newton steps, v_div_fmas, v_div_fixup), with the refined reciprocal of the scaled denominator
// computed once when both quotients scale the denominator identically -- the normal case; any lane that disagrees sends
// the wavefront through the two independent divisions.  Bit for bit the same quotients either way.
__device__ __forceinline__ void div_pair(double nx, double ny, double de, double &qx, double &qy)
{
    bool fdx, fdy, fnx, fny;
    const double sdx = __builtin_amdgcn_div_scale(nx, de, false, &fdx);
    const double sdy = __builtin_amdgcn_div_scale(ny, de, false, &fdy);
    if (__builtin_amdgcn_ballot_w64(__double_as_longlong(sdx) != __double_as_longlong(sdy)) != 0) {
        qx = nx / de;
        qy = ny / de;
        return;
    }
    double r = __builtin_amdgcn_rcp(sdx);
    double e = fma(-sdx, r, 1.0);
    r = fma(r, e, r);
    e = fma(-sdx, r, 1.0);
    r = fma(r, e, r);
    const double snx = __builtin_amdgcn_div_scale(nx, de, true, &fnx);
    const double sny = __builtin_amdgcn_div_scale(ny, de, true, &fny);
    const double px = snx * r, py = sny * r;
    const double ex = fma(-sdx, px, snx), ey = fma(-sdx, py, sny);
    qx = __builtin_amdgcn_div_fixup(__builtin_amdgcn_div_fmas(ex, r, px, fnx), de, nx);
    qy = __builtin_amdgcn_div_fixup(__builtin_amdgcn_div_fmas(ey, r, py, fny), de, ny);
}

// The two quotients as float32 (all the map keeps of them, grid_rendering/type.py:209-261 stores float32 and cv.remap reads
// float32): nx * r and ny * r with r = 1 / de refined to full double precision are within a few ulp (2^-50 relative) of the
// exact quotients, so they round to the same float32 unless they lie that close to a rounding midpoint of float32 -- the 29
// significand bits a float32 drops within 2^-48 of 1000...0.  Returns false (wavefront-wide) when any active lane cannot
// prove it; the caller then takes the exactly rounded division.  |de| >= 1e-6 and finite here (cells whose denominator may
// vanish are flagged and never come this way), so no operand scaling is needed.
__device__ __forceinline__ bool div_pair_f32_fast(double nx, double ny, double de, float &fx, float &fy)
{
    double r = __builtin_amdgcn_rcp(de);
    double e = fma(-de, r, 1.0);
    r = fma(r, e, r);
    e = fma(-de, r, 1.0);
    r = fma(r, e, r);
    const double qx = nx * r, qy = ny * r;
    const uint32_t lx = (uint32_t)__double_as_longlong(qx) & 0x1fffffffu, ly = (uint32_t)__double_as_longlong(qy) & 0x1fffffffu;
    const bool danger = (lx - (0x10000000u - 16u)) < 32u || (ly - (0x10000000u - 16u)) < 32u;
    fx = (float)qx;
    fy = (float)qy;
    return __builtin_amdgcn_ballot_w64(danger) == 0;
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

// cv.remap's bilinear sample of an RGB pixel whose 2 x 2 taps are not all inside the source (the rim of the result and
// everything that maps outside): vkd::sample_u8<3> on the kernel's global-address-space source pointer, every tap its own
// predicated byte load.  Returns r | b << 16 and g.
__device__ __forceinline__ void rim_sample_rgb(gsrc_t src, int sh, int sw, ptrdiff_t sstride, int X, int Y, uint32_t &prb, uint32_t &pg)
{
    const int sx = vkd::sat_short(X >> 5), sy = vkd::sat_short(Y >> 5);
    const int fx = X & 31, fy = Y & 31;
    prb = 0; pg = 0;
    if (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0) return;
    const bool x0 = sx >= 0, x1 = sx + 1 < sw, y0 = sy >= 0, y1 = sy + 1 < sh;
    // (sy, sx) may be -1: offsets are formed in 64 bits from the row / column that exist
    const gsrc_t r0 = src + (ptrdiff_t)sy * sstride + (ptrdiff_t)sx * 3, r1 = r0 + sstride;
    const int w00 = (32 - fy) * (32 - fx), w01 = (32 - fy) * fx, w10 = fy * (32 - fx), w11 = fy * fx;
    uint32_t c[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int v0 = (x0 && y0) ? r0[k] : 0;
        const int v1 = (x1 && y0) ? r0[3 + k] : 0;
        const int v2 = (x0 && y1) ? r1[k] : 0;
        const int v3 = (x1 && y1) ? r1[3 + k] : 0;
        c[k] = (uint32_t)(v0 * w00 + v1 * w01 + v2 * w10 + v3 * w11 + 512) >> 10;
    }
    prb = c[0] | (c[2] << 16);
    pg = c[1];
}

struct HsvLut {
    int sdiv[256];
    int hdiv[256];
};

__device__ __forceinline__ void hue_shift_px(const int *sdiv, const int *hdiv, int delta, int &r, int &g, int &b)
{
    int H, S, V;
    vkd::rgb2hsv_full<true>(sdiv, hdiv, r, g, b, H, S, V);
    H = (H + delta) & 255;   // python modulo 256 of a sum that may be negative (H itself is only valid modulo 256 here)
    vkd::hsv2rgb_full(H, S, V, r, g, b);
}

// Dword pitch of the ownership plane.  Odd: the raster walks columns AND rows, an odd pitch keeps both off a single LDS
// bank.  97: once the tags of the window rows (2m, 2m + 1) are consumed, their 2 x 97 dwords take the horizontal sums of
// that row pair -- three 64-dword planes R, G, B whose dwords hold (row 2m | row 2m + 1 << 16), the operand layout of the
// vertical pass's v_dot2_u32_u16 -- written by the wavefront that read the tags.
constexpr int P_ = 97;
constexpr int kPairPitch = 2 * P_;  // dwords between the plane triples of consecutive row pairs
constexpr size_t kLdsOwn = sizeof(uint32_t) * WH * P_;
constexpr size_t kLdsHbB = 1024;                                 // phase A's work-list prefix sums
constexpr size_t kLdsCellR = sizeof(CellR) * NLDSCELL;
constexpr size_t kLdsCellH = sizeof(double) * 9 * NLDSCELL;     // 72-byte pitch keeps same-index reads of different
                                                                // cells on different LDS banks
constexpr size_t kLdsLut = sizeof(int) * 512;
constexpr size_t kLdsSel = sizeof(uint32_t) * 8;              // byte-permute selectors of the six hue sectors, then one flag word
constexpr size_t kLdsNoiseRows = sizeof(uint32_t) * 2 * WH;    // tiled noise: (slot offset, samples before the next tile | step into it << 16) per output row
constexpr size_t kFusedLds = kLdsOwn + kLdsHbB + kLdsCellR + kLdsCellH + kLdsLut + kLdsSel + 16 + kLdsNoiseRows;

typedef const double __attribute__((address_space(3))) *lds_cdouble_t;


// saturating float -> int32 convert of an already integral value (v_cvt_i32_f32: +-big -> INT_MAX / INT_MIN, NaN -> 0)
__device__ __forceinline__ int cvt_i32_sat(float v)
{
    int r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}

// INTERIOR: the tile's whole 64 x 64 window lies inside the image and its candidates fit one LDS chunk -- the common
// case (about 89 % of the tiles of a 2048^2 page).  Border handling (row / column validity, BORDER_REFLECT_101 lane tables,
// the global-memory fallback for candidates beyond the chunk) compiles away; the arithmetic is the same.
// STREAK: the line_streak stage is compiled in (its own kernel instance, so batches without it keep the lean one)
// RC: the blur radius as a compile-time constant (0..RMAX), or -1 = read it from the item.  With RC fixed the tap
// loops of phases D / E unroll to exactly K taps and the tile geometry folds into immediates.
template <int KIND, bool STREAK = false, int RC = -1>   // 0 generic, 1 interior, 2 empty, 3 element remap (any element types)
__device__ __forceinline__ void chain_tile(const ItemDev &it, const int tx, const int ty,
                                           const vkc::CellC *__restrict__ cells, const TileBin &bin,
                                           const HsvLut *__restrict__ lut, int phase_limit, const ElemPack *els = nullptr)
{
    constexpr bool INTERIOR = KIND == 1, EMPTY = KIND == 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // `own` holds owner tags during phases A / C, then the horizontal sums / packed pixels of phases D / E; row pitch P.
    uint32_t *own = (uint32_t *)smem;
    uint16_t *hbB = (uint16_t *)(smem + kLdsOwn);
    CellR *lcr = (CellR *)(smem + kLdsOwn + kLdsHbB);
    double *lch = (double *)(smem + kLdsOwn + kLdsHbB + kLdsCellR);
    int *lsdiv = (int *)(smem + kLdsOwn + kLdsHbB + kLdsCellR + kLdsCellH);
    int *lhdiv = lsdiv + 256;
    uint32_t *lsel = (uint32_t *)(lhdiv + 256);      // [8] hue sector selectors (vkd::kHsvSelectors)
    int *lflag = (int *)(lsel + 8);                  // != 0: a candidate cell's projective denominator may vanish
    uint32_t *lnrow = (uint32_t *)(smem + kLdsOwn + kLdsHbB + kLdsCellR + kLdsCellH + kLdsLut + kLdsSel + 16);   // [64 rows][2]

    // the wavefront index is uniform: read it into an SGPR so that every row index, row address and row predicate
    // derived from it is scalar arithmetic instead of per-lane (64-bit, quarter-rate) multiplies
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int R = RC >= 0 ? RC : it.R, K = 2 * R + 1, Tw = tile_side(R), Th = tile_height(R);
    const int dw = it.dw, dh = it.dh;
    const int x0 = tx * Tw, y0 = ty * Th;                         // the tile proper
    const int tw = INTERIOR ? Tw : min(Tw, dw - x0), th = INTERIOR ? Th : min(Th, dh - y0);
    const int wx0 = x0 - R, wy0 = y0 - R;                         // window origin (may be negative)
    const int cx0 = INTERIOR ? wx0 : max(wx0, 0), cx1 = INTERIOR ? wx0 + W : min(wx0 + W, dw);   // window clipped to the image
    const int cy0 = INTERIOR ? wy0 : max(wy0, 0), cy1 = INTERIOR ? wy0 + WH : min(wy0 + WH, dh);

    const gsrc_t src = (gsrc_t)it.src;
    const ptrdiff_t sstride = it.sstride;
    const int sh = it.sh, sw = it.sw;

    const int ocx = lane - R;                                     // output column inside the tile (phase E)
    const bool ocol = ocx >= 0 && ocx < tw;
    const int r0 = bin.rmin, c0 = bin.cmin;
    const int nr = max(0, bin.rmax1 - bin.rmin), ncol = max(0, bin.cmax1 - bin.cmin);
    const int nc = bin.rmax1 > 0 ? nr * ncol : 0;
    const int cw = it.cols - 1;   // cells per lattice row
    const vkc::CellC *gcell = cells + it.cell_base;

    // Candidate records [base, base + cn) go into the two LDS tables, 16 lanes x 8 B per 128-byte record and two records per
    // lane (NLDSCELL = 2 x 32).  Fetch and store are split so that the first chunk's loads are in flight, together with the
    // hue tables, while the ownership plane is being cleared: one memory round trip at the head of the tile instead of three.
    constexpr int kRecPerLane = NLDSCELL / (NTHREADS / 16);      // 4 (256 lanes: fused_nw4.hip), 2 (512 lanes) or 1 (1024 lanes)
    static_assert(NLDSCELL == kRecPerLane * (NTHREADS / 16), "whole records per lane and chunk");
    const float rcp_ncol = __builtin_amdgcn_rcpf((float)max(ncol, 1));
    auto chunk_fetch = [&](int base, int cn_, unsigned long long (&v)[kRecPerLane]) {
#pragma unroll
        for (int h = 0; h < kRecPerLane; h++) {
            const int rec = (tid >> 4) + h * (NTHREADS / 16);
            v[h] = 0;
            if (rec < cn_) {
                const int k = base + rec;
                const int rr = div_small(k, ncol, rcp_ncol), cc = k - rr * ncol;
                const unsigned long long VKX_GLOBAL *s8 =
                    (const unsigned long long VKX_GLOBAL *)(gcell + (r0 + rr) * cw + (c0 + cc));
                v[h] = s8[tid & 15];
            }
        }
    };
    auto chunk_store = [&](int cn_, const unsigned long long (&v)[kRecPerLane]) {
#pragma unroll
        for (int h = 0; h < kRecPerLane; h++) {
            const int rec = (tid >> 4) + h * (NTHREADS / 16), part = tid & 15;
            if (rec < cn_) {
                if (part < 8) ((unsigned long long *)(lch + rec * 9))[part] = v[h];
                else if (part < 15) ((unsigned long long *)(lcr + rec))[part - 8] = v[h];
            }
        }
    };
    auto load_chunk = [&](int base, int cn_) {
        unsigned long long v[kRecPerLane];
        chunk_fetch(base, cn_, v);
        chunk_store(cn_, v);
    };

    // Tiled noise (the numpy stream's own tile slots instead of a plane, nprand.hip): k_chain_noise_rows has looked up, for
    // every output row and tile column, where the row's first sample sits in the slots.  Lane = output row of the tile: one
    // 8-byte load per lane of wavefront 0 at the head of the tile, parked in LDS for phase E once the ownership plane is clear.
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    u32x2 nrec = {0u, 0u};
    const bool nrows_here = !(KIND == 3) && wave < WH / 64 && it.noise_rows != nullptr;
    const int nrow_y = wave * 64 + lane;          // (the first WH / 64 wavefronts: one output row per lane)
    if (nrows_here && nrow_y < th)
        nrec = ((const u32x2 VKX_GLOBAL *)it.noise_rows)[(size_t)(y0 + nrow_y) * (size_t)it.tiles_x + (size_t)tx];

    uint32_t kq[2 * RMAX + 1];
#pragma unroll
    for (int i = 0; i < 2 * RMAX + 1; i++) kq[i] = (R > 0 && i < K) ? it.kq[i < K - 1 - i ? i : K - 1 - i] : 0;   // symmetric (checked on the host)
    if constexpr (!EMPTY) {
        // ---- A: clear the ownership plane, then rasterise the candidates chunk by chunk out of LDS
        unsigned long long first[kRecPerLane];
        chunk_fetch(0, min(NLDSCELL, nc), first);
        int lutv = 0;
        uint32_t selv = 0;
        if (it.hue_on) {
            if (NTHREADS == 512 || tid < 512) lutv = tid < 256 ? lut->sdiv[tid] : lut->hdiv[tid - 256];
            if (tid < 8) selv = vkd::kHsvSelectors[tid];
        }
#pragma unroll
        for (int i = 0; i < (WH * P_ / 4 + NTHREADS - 1) / NTHREADS; i++)
            if (tid + i * NTHREADS < WH * P_ / 4) ((uint4 *)own)[tid + i * NTHREADS] = make_uint4(0, 0, 0, 0);
        if (it.hue_on) {
            if (NTHREADS == 512 || tid < 512) lsdiv[tid] = lutv;      // sdiv[256] and hdiv[256] are adjacent
            if (tid < 8) lsel[tid] = selv;
        }
        if (tid == 0) *lflag = 0;
        if (nrows_here) *(u32x2 *)(lnrow + 2 * nrow_y) = nrec;
        // (an interior tile has at most NLDSCELL candidates: one pass, no loop)
        for (int base = 0; base < (INTERIOR ? 1 : max(nc, 1)); base += NLDSCELL) {
            const int cn_ = INTERIOR ? nc : min(NLDSCELL, nc - base);
            if (base > 0) {
                __syncthreads();            // the previous chunk is still being read
                load_chunk(base, cn_);
            } else {
                chunk_store(cn_, first);
            }
            __syncthreads();
            if (phase_limit == 11) return;
            // Work list of the chunk: for every candidate the window rows its scanlines can touch (compacted with a
            // prefix sum so that no lane idles on rows outside the cell), then one item per (candidate, edge).
            int *lpref = (int *)hbB;                 // [NLDSCELL + 1] exclusive prefix of row counts (hbB is free in A)
            int *lylo = lpref + NLDSCELL + 1;        // [NLDSCELL] first window row of each candidate
            if (wave == 0) {
                int hk = 0, ylo = 0;
                bool flagged = false;
                if (lane < cn_) {
                    const CellR &c = lcr[lane];
                    flagged = c.flags & 1;
                    int vmin = INT_MAX, vmax = INT_MIN;
#pragma unroll
                    for (int i = 0; i < 4; i++) { vmin = min(vmin, (int)c.vy[i]); vmax = max(vmax, (int)c.vy[i]); }
                    ylo = max(vmin, cy0);
                    hk = max(0, min(vmax, cy1) - ylo);    // scanlines vmin <= y < vmax
                }
                int incl = hk;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const int t = __shfl_up(incl, d, 64);
                    if (lane >= d) incl += t;
                }
                lpref[lane + 1] = incl;
                if (lane == 0) lpref[0] = 0;
                lylo[lane] = ylo;
                if (__builtin_amdgcn_ballot_w64(flagged) != 0 && lane == 0) *lflag = 1;
            }
            __syncthreads();
            if (phase_limit == 12) return;
            const int nrows = lpref[NLDSCELL];
            // A.1 interior: spans [ceil(xa), floor(xb)] of the x-sorted edge crossings (16.16 fixed point) of one
            //     (candidate, scanline) item, clipped to the window
            for (int p = tid; p < nrows; p += NTHREADS) {
                int lo = 0, hi = cn_ - 1;            // largest kk with lpref[kk] <= p
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (lpref[mid] <= p) lo = mid; else hi = mid - 1;
                }
                const int kk = lo;
                const int y = lylo[kk] + (p - lpref[kk]);
                const int row = y - wy0;
                const CellR &c = lcr[kk];
                const uint32_t tag = (uint32_t)(base + kk) + 1;
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
                for (int a = 1; a < n; a++) {
                    const int v = xs[a];
                    int b = a - 1;
                    while (b >= 0 && xs[b] > v) { xs[b + 1] = xs[b]; b--; }
                    xs[b + 1] = v;
                }
                const bool check = c.flags & 1;
                const double h6 = check ? lch[kk * 9 + 6] : 0.0, h7 = check ? lch[kk * 9 + 7] : 0.0;
                uint32_t *o = own + row * P_ - wx0;       // o[x]
                for (int a = 0; a + 1 < n; a += 2) {
                    const int x1 = max(max((xs[a] + 65535) >> 16, xmin), cx0);
                    const int x2 = min(min(xs[a + 1] >> 16, xmax), cx1 - 1);
                    if (!check) {
                        // four pixels per trip at immediate offsets (per pixel the loop spent three vector and three scalar instructions
                        // beside its ds_max), the last one to three under their own predicates
                        uint32_t *q = o + x1;
                        int left = x2 - x1 + 1;
                        for (; left >= 4; left -= 4, q += 4) {
                            atomicMax(q, tag); atomicMax(q + 1, tag); atomicMax(q + 2, tag); atomicMax(q + 3, tag);
                        }
                        if (left >= 1) atomicMax(q, tag);
                        if (left >= 2) atomicMax(q + 1, tag);
                        if (left >= 3) atomicMax(q + 2, tag);
                    } else {
                        for (int x = x1; x <= x2; x++) {
                            if (fma(1.0, 1.0, fma(h7, (double)y, h6 * (double)x)) == 0) continue;
                            atomicMax(o + x, tag);
                        }
                    }
                }
            }
            if (phase_limit == 13) return;
            // A.2 outline: one (candidate, edge) pair per step; 8-connected Bresenham from the edge's left end,
            //     advanced incrementally (cv::LineIterator) over the part of the edge inside the window
            // (handed out from the LAST lane downwards: the span items above keep the first wavefronts busy, the outline
            //  items go to the ones they leave idle, and the two kinds of work overlap instead of following each other)
            for (int p = NTHREADS - 1 - tid; p < cn_ * 4; p += NTHREADS) {
                const int kk = p >> 2, i = p & 3;
                const CellR &c = lcr[kk];
                const uint32_t tag = (uint32_t)(base + kk) + 1;
                const int a = (i + 3) & 3;
                int lx = c.vx[a], ly = c.vy[a], rx = c.vx[i], ry = c.vy[i];
                if (rx < lx) { const int t1 = lx, t2 = ly; lx = rx; ly = ry; rx = t1; ry = t2; }
                const int dx = rx - lx, dy = ry - ly, ady = dy < 0 ? -dy : dy, sy = dy < 0 ? -1 : 1;
                const bool ymajor = ady > dx;
                const int dmaj = ymajor ? ady : dx, dmin = ymajor ? dx : ady;
                int k0, k1;   // range of major steps whose pixel can lie inside the window
                if (ymajor) {
                    if (sy > 0) { k0 = max(0, cy0 - ly); k1 = min(ady, cy1 - 1 - ly); }
                    else        { k0 = max(0, ly - (cy1 - 1)); k1 = min(ady, ly - cy0); }
                } else {
                    k0 = max(0, cx0 - lx); k1 = min(dx, cx1 - 1 - lx);
                }
                if (k0 > k1) continue;
                // 32-bit arithmetic throughout: coordinates are below 2^15, so 2 k dmin - dmaj fits an int32 and the error term
                // (bounded by 2 dmaj along the line) survives the wrap-around of its three large summands
                int m = 0;
                {
                    const int num = 2 * k0 * dmin - dmaj;
                    if (dmaj != 0 && num > 0) m = (int)(((uint32_t)num + 2u * (uint32_t)dmaj - 1u) / (2u * (uint32_t)dmaj));
                }
                // LineIterator's error term after k0 steps: err = dmaj - 2 dmin (k0 + 1) + 2 dmaj m
                int err = (int)((uint32_t)dmaj - 2u * (uint32_t)dmin * (uint32_t)(k0 + 1) + 2u * (uint32_t)dmaj * (uint32_t)m);
                const int up = 2 * dmaj - 2 * dmin, down = -2 * dmin;
                // the pixel of step k0 in window coordinates, and what a major / minor step adds to it
                int px = (ymajor ? lx + m : lx + k0) - wx0, py = (ymajor ? ly + sy * k0 : ly + sy * m) - wy0;
                const int ax = ymajor ? 0 : 1, ay = ymajor ? sy : 0, bx = ymajor ? 1 : 0, by = ymajor ? 0 : sy;
                const int lox = cx0 - wx0, nx_ = cx1 - cx0, loy = cy0 - wy0, ny_ = cy1 - cy0;
                const bool check = c.flags & 1;
                if (!check) {
                    for (int s = k0; s <= k1; s++) {
                        const bool inside = INTERIOR ? ((unsigned)px < (unsigned)W && (unsigned)py < (unsigned)WH)
                                                     : ((unsigned)(px - lox) < (unsigned)nx_ && (unsigned)(py - loy) < (unsigned)ny_);
                        if (inside) atomicMax(own + py * P_ + px, tag);
                        const bool step = err < 0;
                        err += step ? up : down;
                        px += ax + (step ? bx : 0);
                        py += ay + (step ? by : 0);
                    }
                } else {
                    const double h6 = lch[kk * 9 + 6], h7 = lch[kk * 9 + 7];
                    for (int s = k0; s <= k1; s++) {
                        const bool inside = (unsigned)(px - lox) < (unsigned)nx_ && (unsigned)(py - loy) < (unsigned)ny_;
                        if (inside && !(fma(1.0, 1.0, fma(h7, (double)(py + wy0), h6 * (double)(px + wx0))) == 0))
                            atomicMax(own + py * P_ + px, tag);
                        const bool step = err < 0;
                        err += step ? up : down;
                        px += ax + (step ? bx : 0);
                        py += ay + (step ? by : 0);
                    }
                }
            }
        }
        __syncthreads();
        if (phase_limit == 1) return;

        // If the candidates did not fit one chunk, LDS now holds the LAST chunk; phase C wants chunk 0.
        if (!INTERIOR && nc > NLDSCELL) {
            load_chunk(0, NLDSCELL);
            __syncthreads();
        }

        // ---- C + D: every wavefront owns 8 consecutive window rows; CGROUP rows at a time so that the fp64 chains
        //      and the gathers of several rows overlap.  Lane = window column.
        const int gx = wx0 + lane;
        const bool colok = INTERIOR || (gx >= cx0 && gx < cx1);
        int srcl[2 * RMAX + 1];   // lane (x 4) holding tap i of this lane's horizontal stencil (BORDER_REFLECT_101)
#pragma unroll
        for (int i = 0; i < 2 * RMAX + 1; i++) {
            int s = (R > 0 && i < K) ? (INTERIOR ? lane + i - R : reflect101(gx + i - R, dw) - wx0) : lane;
            srcl[i] = min(max(s, 0), W - 1) << 2;   // ds_bpermute takes the source lane as a byte address
        }
        const unsigned fast_xlim = (unsigned)max(sw - 2, 0), fast_ylim = (unsigned)max(sh - 1, 0);
        const double fxd = (double)gx;               // the lane's column: the same for every row of the tile
        // a candidate whose denominator may vanish sends the whole tile through the exactly rounded divisions (NaN / inf
        // quotients must come out as the reference's)
        const bool exact_div = __builtin_amdgcn_readfirstlane(*lflag) != 0;
        for (int g0 = 0; g0 < ROWS_PER_WAVE; g0 += CGROUP) {
            int X[CGROUP], Y[CGROUP];
            bool rowok[CGROUP];
#pragma unroll
            for (int u = 0; u < CGROUP; u++) {
                const int ly = wave * ROWS_PER_WAVE + g0 + u;
                const int gy = wy0 + ly;
                rowok[u] = INTERIOR || (gy >= cy0 && gy < cy1);
                X[u] = 0; Y[u] = 0;
                if (rowok[u] && colok) {
                    const uint32_t o = own[ly * P_ + lane];
                    if (o != 0) {
                        const int k = (int)o - 1;
                        double h[8];
                        if (INTERIOR || k < NLDSCELL) {
                            // one base register, the eight loads at immediate offsets
                            lds_cdouble_t hp = (lds_cdouble_t)(lch + __umul24(k, 9));
                            asm volatile("" : "+v"(hp));
#pragma unroll
                            for (int j = 0; j < 8; j++) h[j] = hp[j];
                        } else {
                            const vkc::CellC VKX_GLOBAL *gc =
                                (const vkc::CellC VKX_GLOBAL *)(gcell + (r0 + div_small(k, ncol, rcp_ncol)) * cw + (c0 + k - div_small(k, ncol, rcp_ncol) * ncol));
#pragma unroll
                            for (int j = 0; j < 8; j++) h[j] = gc->H[j];
                        }
                        const double fy = (double)gy;
                        const double nx = fma(h[2], 1.0, fma(h[1], fy, h[0] * fxd));
                        const double ny = fma(h[5], 1.0, fma(h[4], fy, h[3] * fxd));
                        const double de = fma(1.0, 1.0, fma(h[7], fy, h[6] * fxd));
                        float qxf, qyf;
                        if (exact_div || !div_pair_f32_fast(nx, ny, de, qxf, qyf)) {
                            double qx, qy;
                            div_pair(nx, ny, de, qx, qy);
                            X[u] = vkd::cv_round((float)qx * 32.f);
                            Y[u] = vkd::cv_round((float)qy * 32.f);
                        } else {
                            // finite quotients: beyond int32 the saturating convert lands outside the source on the
                            // same side as cvRound's INT_MIN does after the int16 saturation of cv.remap -- border
                            // pixels either way
                            X[u] = cvt_i32_sat(rintf(qxf * 32.f));
                            Y[u] = cvt_i32_sat(rintf(qyf * 32.f));
                        }
                    }
                }
            }
            if constexpr (KIND == 3) {
                // element mode: R = 0, the window is the tile; every element is sampled at the shared coordinate
#pragma unroll
                for (int u = 0; u < CGROUP; u++) {
                    const int gy = wy0 + wave * ROWS_PER_WAVE + g0 + u;
                    if (!(rowok[u] && colok)) continue;
                    // all four taps inside the source (every pixel but a rim of the result): unpredicated loads, two
                    // samples per load where they are adjacent in memory; the rim takes the generic samplers
                    const int sx = X[u] >> 5, sy = Y[u] >> 5, fx = X[u] & 31, fy = Y[u] & 31;
                    const bool inner = (unsigned)sx < fast_xlim && (unsigned)sy < fast_ylim;
                    const int w00 = (32 - fy) * (32 - fx), w01 = (32 - fy) * fx, w10 = fy * (32 - fx), w11 = fy * fx;
                    // (byte stores for RGB: packing four pixels into three dword stores -- the chain kernel's store -- measured
                    //  0.206 against 0.177 ms here: the shuffle and the group logic cost registers this variant does not have)
                    for (int e = 0; e < els->n; e++) {
                        const ItemDev::Elem el = els->el[e];
                        if (el.is_f32) {
                            float v;
                            if (inner) {
                                typedef float f32x2_u4 __attribute__((ext_vector_type(2), aligned(4)));
                                const float VKX_GLOBAL *q0 = (const float VKX_GLOBAL *)el.src + (ptrdiff_t)sy * el.sstride + sx;
                                const f32x2_u4 a = *(const f32x2_u4 VKX_GLOBAL *)q0, b = *(const f32x2_u4 VKX_GLOBAL *)(q0 + el.sstride);
                                const float ax = fx * (1.f / 32), ay = fy * (1.f / 32), bx = 1.f - ax, by = 1.f - ay;
                                const float p0 = a.x * (by * bx), p1 = a.y * (by * ax), p2 = b.x * (ay * bx), p3 = b.y * (ay * ax);
                                v = ((p0 + p1) + p2) + p3;
                            } else {
                                v = vkd::sample_f32((const float VKX_GLOBAL *)el.src, sh, sw, el.sstride, X[u], Y[u]);
                            }
                            ((float VKX_GLOBAL *)el.dst)[(ptrdiff_t)gy * el.dstride + gx] = v;
                        } else {
                            gdst_t d = (gdst_t)el.dst + (ptrdiff_t)gy * el.dstride + (ptrdiff_t)gx * el.cn;
                            const gsrc_t sp = (gsrc_t)el.src;
                            if (el.cn == 1) {
                                if (inner) {
                                    const gsrc_t q0 = sp + (ptrdiff_t)sy * el.sstride + sx;
                                    const uint32_t a = *(const u16_u1 VKX_GLOBAL *)q0, b = *(const u16_u1 VKX_GLOBAL *)(q0 + el.sstride);
                                    d[0] = (uint8_t)(((a & 0xff) * w00 + (a >> 8) * w01 + (b & 0xff) * w10 + (b >> 8) * w11 + 512) >> 10);
                                } else {
                                    { uint8_t p1[1]; vkd::sample_u8<1>(sp, sh, sw, el.sstride, X[u], Y[u], p1); d[0] = p1[0]; }
                                }
                            } else if (el.cn == 3) {
                                uint8_t p3[3];
                                if (inner) {
                                    const gsrc_t q0 = sp + (ptrdiff_t)sy * el.sstride + (ptrdiff_t)sx * 3;
                                    const unsigned long long a = *(const u64_u1 VKX_GLOBAL *)q0, b = *(const u64_u1 VKX_GLOBAL *)(q0 + el.sstride);
#pragma unroll
                                    for (int k = 0; k < 3; k++) {
                                        const int v0 = (int)((a >> (8 * k)) & 0xff), v1 = (int)((a >> (8 * (k + 3))) & 0xff);
                                        const int v2 = (int)((b >> (8 * k)) & 0xff), v3 = (int)((b >> (8 * (k + 3))) & 0xff);
                                        p3[k] = (uint8_t)((v0 * w00 + v1 * w01 + v2 * w10 + v3 * w11 + 512) >> 10);
                                    }
                                } else {
                                    vkd::sample_u8<3>(sp, sh, sw, el.sstride, X[u], Y[u], p3);
                                }
                                d[0] = p3[0]; d[1] = p3[1]; d[2] = p3[2];
                            } else {
                                uint8_t p4[4];
                                if (inner) {
                                    const gsrc_t q0 = sp + (ptrdiff_t)sy * el.sstride + (ptrdiff_t)sx * 4;
                                    const unsigned long long a = *(const u64_u1 VKX_GLOBAL *)q0, b = *(const u64_u1 VKX_GLOBAL *)(q0 + el.sstride);
#pragma unroll
                                    for (int k = 0; k < 4; k++) {
                                        const int v0 = (int)((a >> (8 * k)) & 0xff), v1 = (int)((a >> (8 * (k + 4))) & 0xff);
                                        const int v2 = (int)((b >> (8 * k)) & 0xff), v3 = (int)((b >> (8 * (k + 4))) & 0xff);
                                        p4[k] = (uint8_t)((v0 * w00 + v1 * w01 + v2 * w10 + v3 * w11 + 512) >> 10);
                                    }
                                } else {
                                    vkd::sample_u8<4>(sp, sh, sw, el.sstride, X[u], Y[u], p4);
                                }
                                *(uint32_t VKX_GLOBAL *)d = (uint32_t)p4[0] | ((uint32_t)p4[1] << 8) | ((uint32_t)p4[2] << 16) |
                                                 ((uint32_t)p4[3] << 24);
                            }
                        }
                    }
                }
                continue;
            }
            unsigned long long ta[CGROUP], tb[CGROUP];
            bool fast[CGROUP];
#pragma unroll
            for (int u = 0; u < CGROUP; u++) {
                // 0 <= sx, sx + 2 < sw, 0 <= sy, sy + 1 < sh as two unsigned compares; cv.remap's int16 saturation of the
                // integer coordinate cannot change the verdict (sw, sh <= 32767) and only matters on the slow path
                const int sx = X[u] >> 5, sy = Y[u] >> 5;
                fast[u] = rowok[u] && colok && (unsigned)sx < fast_xlim && (unsigned)sy < fast_ylim;
                ta[u] = 0; tb[u] = 0;
                if (fast[u]) {
                    // interior: the two 6-byte tap pairs come in as two unaligned 8-byte loads (3 sx + 8 <= 3 sw)
                    // sy < 2^15, sstride < 2^24 and sh * sstride < 2^32 (checked on the host): full-rate 24-bit multiplies
                    // and a 32-bit lane offset on the uniform base pointer
                    const uint32_t o0 = __umul24((uint32_t)sy, (uint32_t)sstride) + __umul24((uint32_t)sx, 3u);
                    // (the tap rows as ALIGNED dwords + a byte alignment in registers -- worth 17 % to remap.hip's warp kernel, whose
                    //  gather is bound by the texture addresser -- measured here: k_chain_fused 9.27 against 9.14 ms, C2 unchanged
                    //  (eight more VALU instructions per row in a VALU-bound kernel, 12 bytes of scratch); in the element mode
                    //  of k_tile_remap 0.174 against 0.177 ms on C5: inside the run-to-run spread.  Not adopted in either.)
                    ta[u] = *(const u64_u1 VKX_GLOBAL *)(src + (size_t)o0);
                    tb[u] = *(const u64_u1 VKX_GLOBAL *)(src + (size_t)(o0 + (uint32_t)sstride));
                }
            }
            uint32_t prb[CGROUP], pg[CGROUP];                // r | b << 16, g of the row's remapped pixel
#pragma unroll
            for (int u = 0; u < CGROUP; u++) {
                prb[u] = 0; pg[u] = 0;
                if (!rowok[u]) continue;                     // uniform over the wavefront
                if (colok) {
                    const int fx = X[u] & 31, fy = Y[u] & 31;
                    if (fast[u]) {
                        // (32 - fy) ((32 - fx) v00 + fx v01) + fy ((32 - fx) v10 + fx v11): the same integer as cv.remap's four
                        // weighted taps.  Horizontal pairs with v_dot4_u32_u8 -- bytes 0 and 3 of a dword are one channel of
                        // two neighbouring RGB pixels -- then the vertical pair with 24-bit multiply-adds.
                        const uint32_t wx = (uint32_t)(32 - fx) | ((uint32_t)fx << 24);
                        const uint32_t t0 = (uint32_t)ta[u], t1 = (uint32_t)(ta[u] >> 32), b0 = (uint32_t)tb[u], b1 = (uint32_t)(tb[u] >> 32);
                        const uint32_t ar = __builtin_amdgcn_udot4(t0, wx, 0u, false);
                        const uint32_t ag = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(t1, t0, 1), wx, 0u, false);
                        const uint32_t ab = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(t1, t0, 2), wx, 0u, false);
                        const uint32_t br = __builtin_amdgcn_udot4(b0, wx, 0u, false);
                        const uint32_t bg = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(b1, b0, 1), wx, 0u, false);
                        const uint32_t bb = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(b1, b0, 2), wx, 0u, false);
                        const uint32_t wy0 = (uint32_t)(32 - fy), wy1 = (uint32_t)fy;
                        const uint32_t r = mad_u24(br, wy1, mad_u24_cs(ar, wy0, 512u)) >> 10;
                        const uint32_t g = mad_u24(bg, wy1, mad_u24_cs(ag, wy0, 512u)) >> 10;
                        const uint32_t b = mad_u24(bb, wy1, mad_u24_cs(ab, wy0, 512u)) >> 10;
                        prb[u] = r | (b << 16);
                        pg[u] = g;
                    } else {
                        rim_sample_rgb(src, sh, sw, sstride, X[u], Y[u], prb[u], pg[u]);
                    }
                }
            }
            static_assert(CGROUP == 2, "phase D works on the window-row pair of one iteration");
            const int ly0 = wave * ROWS_PER_WAVE + g0;       // even: the pair (ly0, ly0 + 1)
            if (!(rowok[0] || rowok[1])) continue;           // uniform over the wavefront
            if (R > 0) {
                // D: horizontal u8 x 8.8 pass of both rows; tap i of lane l lives in lane srcl[i].  r and b travel as the two
                // 16-bit halves of one dword (255 * 256 < 2^16: the halves never carry into each other), the g of the two
                // rows share a dword the same way: three shuffles and three multiply-adds per tap and row pair
                const uint32_t gg = pg[0] | (pg[1] << 16);
                uint32_t a0 = 0, a1 = 0, ag = 0;
#pragma unroll
                for (int i = 0; i < 2 * RMAX + 1; i++) {
                    if (i < K) {
                        uint32_t v0 = prb[0], v1 = prb[1], vg = gg;
                        if (!(INTERIOR && RC >= 0 && i == RC)) {      // the centre tap of an interior window is the lane's own pixel
                            v0 = (uint32_t)__builtin_amdgcn_ds_bpermute(srcl[i], (int)prb[0]);
                            v1 = (uint32_t)__builtin_amdgcn_ds_bpermute(srcl[i], (int)prb[1]);
                            vg = (uint32_t)__builtin_amdgcn_ds_bpermute(srcl[i], (int)gg);
                        }
                        // spelled out: given the symmetric taps the compiler factors k (a + b) for the shuffled
                        // values, whose sum it cannot bound
                        a0 = mad_u24_ks(kq[i], v0, a0);
                        a1 = mad_u24_ks(kq[i], v1, a1);
                        ag = mad_u24_ks(kq[i], vg, ag);
                    }
                }
                // (row | next row << 16) per channel, over the consumed owner tags of the pair
                uint32_t *pl = own + ly0 * P_ + lane;
                pl[0] = __builtin_amdgcn_perm(a1, a0, 0x05040100u);      // R: low halves
                pl[64] = ag;                                             // G
                pl[128] = __builtin_amdgcn_perm(a1, a0, 0x07060302u);    // B: high halves
            } else {
#pragma unroll
                for (int u = 0; u < CGROUP; u++)
                    if (rowok[u]) own[(ly0 + u) * P_ + lane] = (prb[u] & 0xffu) | (pg[u] << 8) | (prb[u] & 0xff0000u);
            }
        }
        if constexpr (KIND == 3) return;   // element mode ends with the gathers
        __syncthreads();
        if (phase_limit == 2) return;

    }

    // ---- E: vertical pass, hue shift, noise, store.  Wavefront w takes output rows w, w + 8, ...
    // This wavefront's output rows are cy = wave + 8 i, column = lane - R.  All their noise (exactly 6 bytes per
    // pixel: a dword + a short) is requested up front, so eight rows of HBM latency overlap; asking earlier (before
    // phase A) would pin 16 VGPRs through the register-heavy phases and spill.
    // The descriptor fields only this phase needs are read here, through a pointer the optimiser cannot trace back:
    // hoisted to the top of the kernel they would sit in SGPRs through phases A - D and spill.
    // (a global-address-space pointer: no FLAT instruction anywhere in this kernel -- flat loads tick both memory counters
    //  and return out of order with respect to LDS operations)
    const ItemDev VKX_GLOBAL *ite_ = (const ItemDev VKX_GLOBAL *)&it;
    asm volatile("" : "+s"(ite_));
    const ItemDev VKX_GLOBAL &ite = *ite_;
    // (such loads come back in VGPRs; the row address arithmetic wants them scalar)
    const gdst_t dst = (gdst_t)uniform64((uint64_t)ite.dst);
    const int16_t VKX_GLOBAL *noise = (const int16_t VKX_GLOBAL *)uniform64((uint64_t)ite.noise);
    const ptrdiff_t dstride = (ptrdiff_t)uniform64((uint64_t)ite.dstride), nstride = (ptrdiff_t)uniform64((uint64_t)ite.nstride);
    uint32_t nzA[ROWS_PER_WAVE];
    uint32_t nzB[ROWS_PER_WAVE];
    // (as the other descriptor fields of this phase: read again here rather than kept in SGPRs through phases A - D)
    const bool tiled = noise && __builtin_amdgcn_readfirstlane(ite.noise_tiled) != 0;
    if constexpr (EMPTY) {                   // (no phase A: parked here)
        if (nrows_here) *(u32x2 *)(lnrow + 2 * nrow_y) = nrec;
        if (tiled) __syncthreads();
    }
    const uint32_t k3 = (uint32_t)(ocx * 3);
    u32x2 nrow = {0u, 0u};           // lane r < 8: the parked record of this wavefront's row r
    if (tiled) nrow = *(const u32x2 *)(lnrow + 2 * (wave + NWAVES * (lane & 7)));
#pragma unroll
    for (int i = 0; i < ROWS_PER_WAVE; i++) {
        nzA[i] = 0; nzB[i] = 0;
        const int cy = wave + NWAVES * i;
        if (noise && ocol && cy < th) {
            const int16_t VKX_GLOBAL *np_;
            if (tiled) {
                // (the row's base is a scalar: the loads take it as their SGPR base, the lane's 6 ocx bytes as the offset)
                const int16_t VKX_GLOBAL *rowp = noise + (size_t)(uint32_t)__builtin_amdgcn_readlane((int)nrow.x, i);
                const uint32_t sd = (uint32_t)__builtin_amdgcn_readlane((int)nrow.y, i);
                uint32_t off = k3;
                if ((sd & 0xffffu) < 3u * W) {       // the row runs into the next generator tile (6 % of the rows)
                    asm volatile("" ::: "memory");   // (a real branch: as a select the common rows would pay for the rare ones)
                    off += k3 >= (sd & 0xffffu) ? (uint32_t)((int)sd >> 16) : 0u;
                }
                np_ = rowp + off;
            } else {
                np_ = noise + (ptrdiff_t)(y0 + cy) * nstride + (ptrdiff_t)(x0 + ocx) * 3;
            }
            nzA[i] = *(const u32_u1 VKX_GLOBAL *)np_;
            nzB[i] = *(const u16_u1 VKX_GLOBAL *)(np_ + 2);
        }
    }
    // vertical taps two at a time: the tile row cy = wave + 8 i has the parity of the wavefront, so the pairing of the taps
    // kq[0 .. K) with the (even row | odd row << 16) dwords is fixed per wavefront
    uint32_t wpair[RMAX + 1];
#pragma unroll
    for (int q = 0; q <= RMAX; q++) {
        const uint32_t e0 = 2 * q < K ? kq[2 * q < 2 * RMAX + 1 ? 2 * q : 0] : 0u, e1 = 2 * q + 1 < K ? kq[2 * q + 1 < 2 * RMAX + 1 ? 2 * q + 1 : 0] : 0u;
        const uint32_t o0 = (q > 0 && 2 * q - 1 < K) ? kq[q > 0 ? 2 * q - 1 : 0] : 0u, o1 = 2 * q < K ? kq[2 * q < 2 * RMAX + 1 ? 2 * q : 0] : 0u;
        wpair[q] = (wave & 1) ? (o0 | (o1 << 16)) : (e0 | (e1 << 16));
    }
    const bool hue_on = __builtin_amdgcn_readfirstlane(ite.hue_on) != 0;
    const int hue_delta = __builtin_amdgcn_readfirstlane(ite.hue_delta);
    const int right4 = min(lane + 1, 63) << 2;       // ds_bpermute address of the right-hand neighbour
    const uint32_t dcol4 = (uint32_t)(x0 * 3 + (ocx >> 2) * 12 + (ocx & 3) * 4);   // this lane's dword of a 4-pixel group
    const int full4 = (tw >> 2) << 2;          // columns covered by whole 4-pixel (12-byte) groups
    const bool streak_on = STREAK && ite.streak_on != 0;
    const int sxm = streak_on ? (x0 + ocx) % ite.streak_step : 0;        // this lane's column phase in the stripe period
    const int sxd = streak_on ? (x0 + ocx) % ite.streak_dash_step : 0;
    // EMPTY: no lattice cell reaches the window, so every pixel of it maps to (0, 0) (the reference's unfilled map
    // entries) = the source's first pixel; the blur of a constant is that constant (kernel taps sum to 256), and the
    // hue shift is evaluated once per lane instead of once per pixel.
    uint32_t epx = 0;                                 // r | g << 8 | b << 16
    if constexpr (EMPTY) {
        int er = src[0], eg = src[1], eb = src[2];
        if (hue_on) hue_shift_px(lut->sdiv, lut->hdiv, hue_delta, er, eg, eb);
        epx = (uint32_t)er | ((uint32_t)eg << 8) | ((uint32_t)eb << 16);
    }
#pragma unroll
    for (int i = 0; i < ROWS_PER_WAVE; i++) {
        const int cy = wave + NWAVES * i;
        if (cy >= th) continue;                 // uniform over the wavefront
        const int gy = y0 + cy;
        uint32_t P = epx;                       // the pixel as r | g << 8 | b << 16
        if constexpr (!EMPTY) {
            if (R > 0) {
                int r, g, b;
                if constexpr (INTERIOR && RC > 0) {
                    // taps cy .. cy + 2 RC of the window = RC + 1 row pairs, two taps per v_dot2_u32_u16
                    const uint32_t *pl = own + (cy & ~1) * P_ + lane;
                    uint32_t ar = 32768u, ag = 32768u, ab = 32768u;
#pragma unroll
                    for (int q = 0; q <= RC; q++) {
                        ar = dot2_u16_ks(wpair[q], pl[q * kPairPitch], ar);
                        ag = dot2_u16_ks(wpair[q], pl[q * kPairPitch + 64], ag);
                        ab = dot2_u16_ks(wpair[q], pl[q * kPairPitch + 128], ab);
                    }
                    r = (int)(ar >> 16); g = (int)(ag >> 16); b = (int)(ab >> 16);
                } else {
                    uint32_t a0 = 0, a1 = 0, a2 = 0;
#pragma unroll
                    for (int j = 0; j < 2 * RMAX + 1; j++) {
                        if (j < K) {
                            const int yy = INTERIOR ? cy + j : reflect101(gy + j - R, dh) - wy0;
                            const uint16_t *h16 = (const uint16_t *)(own + (yy & ~1) * P_ + lane) + (yy & 1);
                            a0 += __umul24(kq[j], (uint32_t)h16[0]);
                            a1 += __umul24(kq[j], (uint32_t)h16[128]);
                            a2 += __umul24(kq[j], (uint32_t)h16[256]);
                        }
                    }
                    r = (int)((a0 + 32768u) >> 16); g = (int)((a1 + 32768u) >> 16); b = (int)((a2 + 32768u) >> 16);
                }
                if (phase_limit == 20) {      // debugging aid: the horizontal sums of the centre row, >> 8
                    const int yy = INTERIOR ? cy + R : reflect101(gy, dh) - wy0;
                    const uint16_t *h16 = (const uint16_t *)(own + (yy & ~1) * P_ + lane) + (yy & 1);
                    r = h16[0] >> 8; g = h16[128] >> 8; b = h16[256] >> 8;
                }
                if (hue_on) P = vkd::hue_shift_packed(lsdiv, lhdiv, lsel, hue_delta, r, g, b);
                else P = (uint32_t)r | ((uint32_t)g << 8) | ((uint32_t)b << 16);
            } else {
                P = own[(cy + R) * P_ + lane];
                if (hue_on) P = vkd::hue_shift_packed(lsdiv, lhdiv, lsel, hue_delta, (int)(P & 0xff), (int)((P >> 8) & 0xff),
                                                      (int)((P >> 16) & 0xff));
            }
        }
        if (noise || (STREAK && streak_on)) {
            int r = (int)(P & 0xff), g = (int)((P >> 8) & 0xff), b = (int)((P >> 16) & 0xff);
            if (noise) {
                r = vkd::clamp_u8(r + (int)(int16_t)(nzA[i] & 0xffff));
                g = vkd::clamp_u8(g + (int)(int16_t)(nzA[i] >> 16));
                b = vkd::clamp_u8(b + (int)(int16_t)(nzB[i] & 0xffff));
            }
            if (STREAK && streak_on) {
                // stripe masks: vertical x % (t + g) < t, horizontal y % (t + g) < t, dash gaps cut them; vertical stripes
                // are blended first, then horizontal ones (crossings twice); trunc(fl32(1 - a) * v + a * c)
                const int ym = gy % ite.streak_step, yd = gy % ite.streak_dash_step;
                bool mv = ite.streak_vert && sxm < ite.streak_thickness;
                bool mh = ite.streak_hori && ym < ite.streak_thickness;
                if (ite.streak_dash) {
                    if (yd < ite.streak_dash_gap) mv = false;
                    if (sxd < ite.streak_dash_gap) mh = false;
                }
                const float w1 = ite.streak_alpha, w0 = 1.0f - w1;
                int *ch[3] = {&r, &g, &b};
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    int v = *ch[c];
                    const int col = ite.streak_color[c];
                    if (mv) { const float t0 = w0 * (float)v, t1 = w1 * (float)col; v = ite.streak_copy ? col : (int)(uint8_t)(t0 + t1); }
                    if (mh) { const float t0 = w0 * (float)v, t1 = w1 * (float)col; v = ite.streak_copy ? col : (int)(uint8_t)(t0 + t1); }
                    *ch[c] = v;
                }
            }
            P = (uint32_t)r | ((uint32_t)g << 8) | ((uint32_t)b << 16);
        }
        // 4 adjacent pixels = 12 bytes = 3 dwords: lanes with (column & 3) = 0, 1, 2 each build one dword from
        // their own pixel and their right neighbour's
        const uint32_t Pn = (uint32_t)__builtin_amdgcn_ds_bpermute(right4, (int)P);
        gdst_t drow = dst + (ptrdiff_t)gy * dstride + (ptrdiff_t)x0 * 3;
        if (ocol) {
            const int m = ocx & 3;
            if (INTERIOR || ocx < full4) {     // an interior tile is whole 4-pixel groups
                if (m < 3) {
                    const uint32_t wv = (P >> (8 * m)) | (Pn << (24 - 8 * m));   // one store for all three lanes
                    // dh * dstride < 2^32 (checked on the host): scalar row offset + lane offset on the scalar base
                    const uint32_t off = (uint32_t)gy * (uint32_t)dstride + dcol4;
                    *(u32_u1 VKX_GLOBAL *)(dst + (size_t)off) = wv;
                }
            } else {
                gdst_t d = drow + ocx * 3;
                d[0] = (uint8_t)P; d[1] = (uint8_t)(P >> 8); d[2] = (uint8_t)(P >> 16);
            }
        }
    }
}

template <bool STREAK>
__global__ void __launch_bounds__(NTHREADS, VKX_FUSED_WAVES_PER_EU) k_chain_fused(const ItemDev *__restrict__ items,
                                                          const vkc::CellC *__restrict__ cells,
                                                          const TileBin *__restrict__ bins,
                                                          const HsvLut *__restrict__ lut, int phase_limit)
{
    // grid = (tile slots, images).  XCD-aware tile order: consecutive workgroup ids land on different XCDs
    // (id % 8); give every XCD a contiguous run of an image's tiles so neighbouring tiles (shared source rows,
    // shared cells) meet in one L2.
    // The run an XCD takes rotates with the image index: the cheap tiles (outside the distorted page, mostly the
    // first and last rows) would otherwise always land on the same two XCDs and leave them idle at the end.
    const int slots = gridDim.x;                 // multiple of 8, >= tiles of the largest image
    const ItemDev &it = items[blockIdx.y];
    const int ntiles = it.tiles_x * it.tiles_y;
    const int per = (ntiles + 7) >> 3;           // this image's run length
    const int run = (int)((blockIdx.x + blockIdx.y) & 7), pos = (int)(blockIdx.x >> 3);
    const int tl = run * per + pos;
    if (pos >= per || tl >= ntiles) return;
    if (phase_limit == 10) return;
    const int tile_id = (int)blockIdx.y * slots + tl;   // bins are laid out [image][slot]
    const TileBin bin = bins[tile_id];
    const int ty = __builtin_amdgcn_readfirstlane(div_small(tl, it.tiles_x, __builtin_amdgcn_rcpf((float)it.tiles_x)));
    const int tx = tl - ty * it.tiles_x;
    const int Tw = tile_side(it.R), Th = tile_height(it.R);
    const int wx0 = tx * Tw - it.R, wy0 = ty * Th - it.R;
    const int nc = bin.rmax1 > 0 ? max(0, bin.rmax1 - bin.rmin) * max(0, bin.cmax1 - bin.cmin) : 0;
    const bool interior = wx0 >= 0 && wy0 >= 0 && wx0 + W <= it.dw && wy0 + WH <= it.dh && nc <= NLDSCELL && phase_limit != 3;
#ifdef VKX_FUSED_CENSUS
    // tools/isa_census.py: only the hot variant (interior window, 5-tap blur) so that its ISA can be read in isolation
    (void)interior;
    chain_tile<1, STREAK, VKX_FUSED_CENSUS>(it, tx, ty, cells, bin, lut, phase_limit);
    return;
#endif
    if (nc == 0) chain_tile<2, STREAK>(it, tx, ty, cells, bin, lut, phase_limit);
    else if (interior) {
        // the common case gets the blur radius as a compile-time constant
        switch (it.R) {
        case 0: chain_tile<1, STREAK, 0>(it, tx, ty, cells, bin, lut, phase_limit); break;
        case 1: chain_tile<1, STREAK, 1>(it, tx, ty, cells, bin, lut, phase_limit); break;
        case 2: chain_tile<1, STREAK, 2>(it, tx, ty, cells, bin, lut, phase_limit); break;
        default: chain_tile<1, STREAK, 3>(it, tx, ty, cells, bin, lut, phase_limit); break;
        }
    } else {
        switch (it.R) {
        case 0: chain_tile<0, STREAK, 0>(it, tx, ty, cells, bin, lut, phase_limit); break;
        case 1: chain_tile<0, STREAK, 1>(it, tx, ty, cells, bin, lut, phase_limit); break;
        case 2: chain_tile<0, STREAK, 2>(it, tx, ty, cells, bin, lut, phase_limit); break;
        default: chain_tile<0, STREAK, 3>(it, tx, ty, cells, bin, lut, phase_limit); break;
        }
    }
}

// Element mode of the same tile machinery: Image / Mask / ScoreMap (uint8 x 1, 3, 4 channels, float32) of one call
// gathered through the shared lattice in one launch -- ownership in LDS, no dense map, no int32 ownership plane in HBM.
__global__ void __launch_bounds__(NTHREADS, 8) k_tile_remap(const ItemDev *__restrict__ items,
                                                           const vkc::CellC *__restrict__ cells,
                                                           const TileBin *__restrict__ bins, const ElemPack pack)
{
    const int slots = gridDim.x;
    const ItemDev &it = items[blockIdx.y];
    const int ntiles = it.tiles_x * it.tiles_y;
    const int per = (ntiles + 7) >> 3;
    const int run = (int)((blockIdx.x + blockIdx.y) & 7), pos = (int)(blockIdx.x >> 3);
    const int tl = run * per + pos;
    if (pos >= per || tl >= ntiles) return;
    const int tile_id = (int)blockIdx.y * slots + tl;
    const TileBin bin = bins[tile_id];
    const int ty = __builtin_amdgcn_readfirstlane(div_small(tl, it.tiles_x, __builtin_amdgcn_rcpf((float)it.tiles_x)));
    const int tx = tl - ty * it.tiles_x;
    chain_tile<3>(it, tx, ty, cells, bin, nullptr, 0, &pack);
}

// Tiled noise: where the first sample of every (output row, tile column) of every image sits in the generator's slots.
// Sample i of the dense [dh, dw, 3] plane lives in the slot of the generator tile t with table[t].first <= i < table[t + 1].first:
// the expected tile from the mean yield of a tile, four table entries around it in one round trip, a walk along the table should
// they not bracket it.  One lane per record; 8 bytes per 64-pixel row segment against the 360 bytes of noise it places.
__global__ void __launch_bounds__(256) k_chain_noise_rows(const ItemDev *__restrict__ items)
{
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    const ItemDev &it = items[blockIdx.y];         // grid = (records of the largest image / 256, images)
    if (!it.noise_rows) return;
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= (uint32_t)it.dh * (uint32_t)it.tiles_x) return;
    const uint32_t y = k / (uint32_t)it.tiles_x, tx = k - y * (uint32_t)it.tiles_x;
    const u32x2 VKX_GLOBAL *tab = (const u32x2 VKX_GLOBAL *)it.noise_table;
    const uint32_t T = (uint32_t)it.noise_tiles, slot = (uint32_t)it.noise_slot;
    const uint32_t i0 = (y * (uint32_t)it.dw + tx * (uint32_t)tile_side(it.R)) * 3u;
    const uint32_t g = min((uint32_t)((float)i0 * it.noise_tiles_per_sample), T - 1);
    const uint32_t gb = g > 0 ? g - 1 : 0;
    const u32x2 e0 = tab[min(gb, T)], e1 = tab[min(gb + 1, T)], e2 = tab[min(gb + 2, T)], e3 = tab[min(gb + 3, T)];
    uint32_t t;
    u32x2 a, b;
    if (i0 >= e2.x) { t = gb + 2; a = e2; b = e3; }
    else if (i0 >= e1.x) { t = gb + 1; a = e1; b = e2; }
    else { t = gb; a = e0; b = e1; }
    while (a.x > i0 && t > 0) { t--; b = a; a = tab[t]; }
    while (b.x <= i0 && t + 1 < T) { t++; a = b; b = tab[t + 1]; }
    // (the step is kSlot - the tile's samples +- the leading elements two tiles skip: well inside an int16)
    const uint32_t before = min(b.x - i0, 0xffffu), step = slot + b.y - a.y - (b.x - a.x);
    u32x2 rec;
    rec.x = t * slot + a.y + (i0 - a.x);
    rec.y = before | (step << 16);
    ((u32x2 VKX_GLOBAL *)it.noise_rows)[k] = rec;
}

// One dispatch instead of a descriptor copy and three memsets (every dispatch on a pipeline lane's stream costs the
// pipeline tens of microseconds while other lanes' plane transfers are in flight): the descriptors are read from the
// page-locked ring through its device mapping, the tile bins start at (min 0x7f7f7f7f, max + 1 = 0), the deferred list
// is empty.
__global__ void __launch_bounds__(256) k_chain_prologue(uint32_t *__restrict__ desc_dst, const uint32_t *__restrict__ desc_src,
                                                        unsigned n_words, TileBin *__restrict__ bins, unsigned nbins,
                                                        int *__restrict__ deferred)
{
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_words) desc_dst[i] = desc_src[i];
    if (i < nbins) bins[i] = TileBin{0x7f7f7f7f, 0x7f7f7f7f, 0, 0};
    if (i == 0) *deferred = 0;
}

} // namespace

// Shared host side of the two tile kernels.  A PLAN holds the device descriptors of a batch; its three steps may be queued on
// different streams by the caller (vkx_chain_rgb_batch_np_dev, chain.hip): `setup` (descriptor copy, cell records, tile bins:
// independent of the pixels and of the noise), `noise_rows` (per image range: after the generator's tile tables) and `tiles`
// (per image range: the pixel kernel).  Every step launches on ctx->stream as it is when the step is called.
struct vkx_chain_plan {
    std::vector<ItemDev> dev;
    std::vector<int> prefix;                  // first cell of every image in the batch-wide cell table
    std::vector<long long> row_prefix;        // first (row, tile column) record of every image with tiled noise
    long long ncells = 0;
    int max_tiles = 0, slots = 0;
    bool elements = false, streak = false;
    const ItemDev *d_items = nullptr;
    const int *d_cell_prefix = nullptr;
    TileBin *bins = nullptr;
    vkc::CellC *cells = nullptr;
    int *deferred = nullptr;
    const HsvLut *lut = nullptr;
};

void vkx_chain_plan_free(vkx_chain_plan *p) { delete p; }
int vkx_chain_plan_items(const vkx_chain_plan *p) { return (int)p->dev.size(); }

// scratch + descriptors + prologue / cell setup kernels.  The chain mode keeps its own scratch (chain_cells / chain_bins /
// chain_misc): its setup may run on a side stream while other entry points use the shared slots on the compute stream.
int vkx_chain_plan_setup(vkx_ctx *ctx, vkx_chain_plan *p)
{
    const int n_items = (int)p->dev.size();
    int rc;
    vkx_scratch *s_cells = p->elements ? &ctx->cells : &ctx->chain_cells;
    vkx_scratch *s_bins = p->elements ? &ctx->owner : &ctx->chain_bins;
    vkx_scratch *s_misc = p->elements ? &ctx->misc : &ctx->chain_misc;
    // device scratch: cell table, tile bins, item descriptors + prefix arrays, HSV tables
    const size_t cells_bytes = sizeof(vkc::CellC) * (size_t)p->ncells;   // then the deferred list: count + ids
    if ((rc = vkx_scratch_reserve(ctx, s_cells, cells_bytes + sizeof(int) * ((size_t)p->ncells + 1)))) return rc;
    p->slots = ((p->max_tiles + 7) / 8) * 8;              // tile slots per image in the launch grid
    const size_t nbins = (size_t)p->slots * n_items;
    if (n_items > 65535) return VKX_ERR_UNSUPPORTED;      // gridDim.y
    if ((rc = vkx_scratch_reserve(ctx, s_bins, sizeof(TileBin) * nbins))) return rc;
    const size_t items_bytes = sizeof(ItemDev) * (size_t)n_items, prefix_bytes = sizeof(int) * p->prefix.size();
    const size_t items_off = 0, prefix_off = (items_bytes + 255) & ~(size_t)255;
    const size_t misc_bytes = prefix_off + prefix_bytes;
    if ((rc = vkx_scratch_reserve(ctx, s_misc, misc_bytes))) return rc;
    if (!p->elements && (rc = vkx_hsv_tables(ctx, (const void **)&p->lut))) return rc;
    unsigned char *misc = (unsigned char *)s_misc->ptr;
    // the descriptors travel through the ctx's page-locked ring: the copy is queued and the launch returns without a
    // stream synchronisation (host-array pipelines keep several launches in flight)
    void *ring = nullptr;
    if ((rc = vkx_desc_ring_take(ctx, misc_bytes, &ring))) return rc;
    memcpy((unsigned char *)ring + items_off, p->dev.data(), items_bytes);
    memcpy((unsigned char *)ring + prefix_off, p->prefix.data(), prefix_bytes);
    p->d_items = (const ItemDev *)(misc + items_off);
    p->d_cell_prefix = (const int *)(misc + prefix_off);
    p->bins = (TileBin *)s_bins->ptr;
    p->cells = (vkc::CellC *)s_cells->ptr;
    p->deferred = (int *)((unsigned char *)s_cells->ptr + cells_bytes);
    {
        void *ring_dev = nullptr;
        VKX_HIP(hipHostGetDevicePointer(&ring_dev, ring, 0));
        const size_t n_words = (misc_bytes + 3) / 4;
        VKX_TIMED(ctx, "k_chain_prologue");
        k_chain_prologue<<<vkx_blocks(std::max(n_words, nbins), 256), 256, 0, ctx->stream>>>((uint32_t *)misc, (const uint32_t *)ring_dev,
                                                                                            (unsigned)n_words, p->bins, (unsigned)nbins, p->deferred);
        VKX_LAUNCH_CHECK();
    }
    { VKX_TIMED(ctx, "k_chain_setup"); k_chain_setup<<<vkx_blocks((size_t)p->ncells, 256), 256, 0, ctx->stream>>>(p->d_items, p->d_cell_prefix, n_items, (int)p->ncells, p->slots, p->cells, p->bins, p->deferred); }
    VKX_LAUNCH_CHECK();
    { VKX_TIMED(ctx, "k_chain_setup_svd"); k_chain_setup_svd<<<128, 64, 0, ctx->stream>>>(p->d_items, p->d_cell_prefix, n_items, p->cells, p->deferred); }
    VKX_LAUNCH_CHECK();
    return VKX_OK;
}

// the (row, tile column) records of the images [first, first + count) whose noise is a generator's tile buffer
int vkx_chain_plan_noise_rows(vkx_ctx *ctx, vkx_chain_plan *p, int first, int count)
{
    if (p->row_prefix.empty() || count <= 0) return VKX_OK;
    long long most = 0;                      // records of the largest image of the range
    for (int i = first; i < first + count; i++) most = std::max(most, p->row_prefix[i + 1] - p->row_prefix[i]);
    if (most <= 0) return VKX_OK;
    VKX_TIMED(ctx, "k_chain_noise_rows");
    k_chain_noise_rows<<<dim3(vkx_blocks((size_t)most, 256), count), 256, 0, ctx->stream>>>(p->d_items + first);
    VKX_LAUNCH_CHECK();
    return VKX_OK;
}

// the pixel kernel over the images [first, first + count)
int vkx_chain_plan_tiles(vkx_ctx *ctx, vkx_chain_plan *p, int first, int count)
{
    if (count <= 0) return VKX_OK;
    // profiling aids: VKX_FUSED_PHASES=1|2 stops the kernel after phase A | C+D, 10..13 inside phase A (tools/phases_a.sh),
    // 20 writes the horizontal sums of the centre row instead of the finished pixel
    static const int phase_limit = [] { const char *e = getenv("VKX_FUSED_PHASES"); return e ? atoi(e) : 0; }();
    const ItemDev *items = p->d_items + first;
    const TileBin *bins = p->bins + (size_t)first * p->slots;      // bins are laid out [image][slot]
    const dim3 grid(p->slots, count);
    if (p->elements) {
        VKX_TIMED(ctx, "k_tile_remap");
        ElemPack pack;
        memset(&pack, 0, sizeof(pack));
        pack.n = p->dev[first].n_elems;
        for (int e = 0; e < 4; e++) pack.el[e] = p->dev[first].el[e];
        k_tile_remap<<<grid, NTHREADS, kFusedLds, ctx->stream>>>(items, p->cells, bins, pack);
    } else {
        VKX_TIMED_MAJOR(ctx, "k_chain_fused");
        if (p->streak) k_chain_fused<true><<<grid, NTHREADS, kFusedLds, ctx->stream>>>(items, p->cells, bins, p->lut, phase_limit);
        else k_chain_fused<false><<<grid, NTHREADS, kFusedLds, ctx->stream>>>(items, p->cells, bins, p->lut, phase_limit);
    }
    VKX_LAUNCH_CHECK();
    return VKX_OK;
}

// Builds the plan of a chain batch.  Returns VKX_ERR_UNSUPPORTED (without setting an error) when the batch has a shape the
// fused path does not take; the caller then runs the per-stage kernels.
int vkx_chain_plan_build(vkx_ctx *ctx, const vkx_chain_item *items, int n_items, vkx_chain_plan **out)
{
    *out = nullptr;
    if (n_items <= 0) return VKX_OK;
    std::unique_ptr<vkx_chain_plan> plan(new vkx_chain_plan());
    std::vector<ItemDev> &dev = plan->dev;
    dev.resize(n_items);
    plan->prefix.resize((size_t)n_items + 1);
    int *cell_prefix = plan->prefix.data();
    std::vector<long long> &row_prefix = plan->row_prefix;
    row_prefix.assign((size_t)n_items + 1, 0);
    long long tiles = 0, ncells = 0;
    int max_tiles = 0;
    for (int i = 0; i < n_items; i++) {
        const vkx_chain_item &it = items[i];
        if (it.blur_ksize > 2 * RMAX + 1 || (it.blur_ksize > 1 && (it.blur_ksize & 1) == 0)) return VKX_ERR_UNSUPPORTED;
        if (it.sh > 32767 || it.sw > 32767 || it.dh > 32767 || it.dw > 32767) return VKX_ERR_UNSUPPORTED;
        if (it.rows < 2 || it.cols < 2) return VKX_ERR_UNSUPPORTED;
        if (it.blur_ksize > 1 && (it.dh == 1 || it.dw == 1)) return VKX_ERR_UNSUPPORTED; // kernel collapses per axis
        ItemDev &d = dev[i];
        memset(&d, 0, sizeof(d));
        d.src = it.src; d.dst = it.dst; d.noise = it.noise; d.sv = it.src_vertices; d.dv = it.dst_vertices;
        d.sstride = it.src_stride; d.dstride = it.dst_stride; d.nstride = it.noise_stride_el;
        d.noise_table = nullptr; d.noise_tiled = 0; d.noise_tiles = 0; d.noise_slot = 0; d.noise_tiles_per_sample = 0.f;
        d.noise_rows = nullptr;
        if (it.noise && it.noise_tiled) {
            const long long n = (long long)it.dh * it.dw * 3;
            if (n > 0x7fffffffLL) return VKX_ERR_UNSUPPORTED;
            const vkx_np_tiles_shape shape = vkx_np_tiles_shape_of(n);
            if (shape.n_tiles * shape.slot_elems > 0xffffffffLL) return VKX_ERR_UNSUPPORTED;      // 32-bit slot offsets
            d.noise = (const int16_t *)((const unsigned char *)it.noise + shape.slots_offset);
            d.noise_table = (const uint2 *)((const unsigned char *)it.noise + shape.table_offset);
            d.noise_tiled = 1; d.noise_tiles = (int)shape.n_tiles; d.noise_slot = shape.slot_elems;
            d.noise_tiles_per_sample = (float)(1.0 / shape.samples_per_tile);
        }
        d.sh = it.sh; d.sw = it.sw; d.dh = it.dh; d.dw = it.dw; d.rows = it.rows; d.cols = it.cols;
        d.R = it.blur_ksize > 1 ? it.blur_ksize / 2 : 0;
        const int Tw = tile_side(d.R), Th = tile_height(d.R);
        d.tiles_x = (it.dw + Tw - 1) / Tw; d.tiles_y = (it.dh + Th - 1) / Th;
        d.cell_base = (int)ncells;
        d.hue_on = it.hue_enabled; d.hue_delta = it.hue_delta;
        d.streak_on = it.streak_enabled && it.streak_alpha != 0.0 && (it.streak_enable_vert || it.streak_enable_hori);
        if (it.streak_enabled) {
            if (it.streak_thickness + it.streak_gap <= 0) return VKX_ERR_UNSUPPORTED;   // the staged path reports it
            d.streak_thickness = it.streak_thickness;
            d.streak_step = it.streak_thickness + it.streak_gap;
            d.streak_dash = it.streak_dash_thickness > 0 && it.streak_dash_gap > 0;
            d.streak_dash_step = d.streak_dash ? it.streak_dash_thickness + it.streak_dash_gap : 1;
            d.streak_dash_gap = it.streak_dash_gap;
            d.streak_vert = it.streak_enable_vert; d.streak_hori = it.streak_enable_hori;
            d.streak_copy = it.streak_alpha == 1.0;
            d.streak_alpha = (float)it.streak_alpha;
            for (int c = 0; c < 3; c++) d.streak_color[c] = it.streak_color[c];
        }
        for (int k = 0; k < 8; k++) d.kq[k] = 0;
        if (d.R > 0 && vkx_gaussian_kernel_q8_host(it.blur_ksize, it.blur_sigma, d.kq)) return VKX_ERR_UNSUPPORTED;
        for (int k = 0; k < d.R; k++)
            if (d.kq[k] != d.kq[2 * d.R - k]) return VKX_ERR_UNSUPPORTED;    // the kernel reads one half of the taps
        // 24-bit row multiply, 32-bit byte offsets inside the source plane
        if (it.src_stride <= 0 || it.src_stride >= (1 << 24) || (long long)it.sh * it.src_stride + 8 > 0xffffffffLL) return VKX_ERR_UNSUPPORTED;
        if (it.dst_stride <= 0 || (long long)it.dh * it.dst_stride > 0xffffffffLL) return VKX_ERR_UNSUPPORTED;
        cell_prefix[i] = (int)ncells;
        row_prefix[i + 1] = row_prefix[i] + (d.noise_tiled ? (long long)it.dh * d.tiles_x : 0);
        tiles += (long long)d.tiles_x * d.tiles_y;
        if (d.tiles_x * d.tiles_y > max_tiles) max_tiles = d.tiles_x * d.tiles_y;
        ncells += (long long)(it.rows - 1) * (it.cols - 1);
        if (tiles > 0x3fffffff || ncells > 0x3fffffff) return VKX_ERR_UNSUPPORTED;
        plan->streak = plan->streak || d.streak_on != 0;
    }
    cell_prefix[n_items] = (int)ncells;
    if (n_items > 65535) return VKX_ERR_UNSUPPORTED;
    if (row_prefix[n_items] > 0) {
        int rc = vkx_scratch_reserve(ctx, &ctx->noise_rows, (size_t)row_prefix[n_items] * sizeof(uint2));
        if (rc) return rc;
        for (int i = 0; i < n_items; i++)
            if (dev[i].noise_tiled) dev[i].noise_rows = (const uint2 *)ctx->noise_rows.ptr + row_prefix[i];
    } else {
        row_prefix.clear();
    }
    plan->ncells = ncells;
    plan->max_tiles = max_tiles;
    *out = plan.release();
    return VKX_OK;
}

// The cell setup of a chain batch depends on nothing but the lattices: it runs on the context's side stream, after the pixel
// kernel of the previous chain call (which reads the scratch it writes) and after the point of the compute stream the caller
// marked with vkx_chain_lattices_ready (by default: after everything queued on the compute stream before this call), so that it
// shares the device with whatever large kernel the compute stream is running (a composite, the generator's draw pass).
int vkx_chain_plan_setup_aside(vkx_ctx *ctx, vkx_chain_plan *p, hipEvent_t *done)
{
    // VKX_CHAIN_SETUP_ASIDE=0 / 1 forces one way.  By default the side stream is for BATCHES (>= kAsideMinItems images: their setup is 0.1 - 0.3 ms
    // that hides under the draw pass or the composite); a call of a few images -- a HostPipeline lane sends one per call -- keeps its setup on the
    // compute stream: the events that order two streams cost more than the setup of one image (tens of microseconds), and cross-stream events between a
    // lane's copies and kernels serialise the two copy directions on this stack (tools/pipe_probe2.py) -- with every lane's setup aside the pipeline's
    // chain legs ran at half their rate (device noise 8.5 -> 4.1 Gpx/s; found in the round-5 records, profiles/r5a .. r5j `dropin`).
    static const int aside_env = [] { const char *e = getenv("VKX_CHAIN_SETUP_ASIDE"); return e ? (e[0] == '0' ? 0 : 1) : -1; }();
    constexpr size_t kAsideMinItems = 16;
    const bool aside = aside_env >= 0 ? aside_env != 0 : p->dev.size() >= kAsideMinItems;
    *done = nullptr;
    if (!aside) {
        if (ctx->lattices_armed) {       // the lattices may be the product of another stream (vkx_camera_states_dev)
            vkx_device_guard guard(ctx);
            VKX_HIP(hipStreamWaitEvent(ctx->stream, ctx->lattices_ready, 0));
            ctx->lattices_armed = false;
        }
        return vkx_chain_plan_setup(ctx, p);
    }
    int rc;
    hipStream_t main_stream = ctx->stream;
    hipStream_t side = vkx_stream_by_id(ctx, VKX_STREAM_COPY_OUT, &rc);
    if (rc) return rc;
    vkx_device_guard guard(ctx);
    if (ctx->lattices_armed) {
        VKX_HIP(hipStreamWaitEvent(side, ctx->lattices_ready, 0));
        ctx->lattices_armed = false;      // one chain call per mark
    } else if ((rc = vkx_stream_order(ctx, side, main_stream))) return rc;
    if (ctx->chain_done) VKX_HIP(hipStreamWaitEvent(side, ctx->chain_done, 0));
    ctx->stream = side;
    rc = vkx_chain_plan_setup(ctx, p);
    ctx->stream = main_stream;
    if (rc) return rc;
    if (!ctx->chain_setup_done) VKX_HIP(hipEventCreateWithFlags(&ctx->chain_setup_done, hipEventDisableTiming));
    VKX_HIP(hipEventRecord(ctx->chain_setup_done, side));
    *done = ctx->chain_setup_done;       // the caller's pixel kernels wait for it (hipStreamWaitEvent) where they are queued
    return VKX_OK;
}

// the pixel kernel of a chain call has been queued on `stream`: the next call's setup waits for it
int vkx_chain_mark_done(vkx_ctx *ctx, hipStream_t stream)
{
    if (!ctx->chain_done) VKX_HIP(hipEventCreateWithFlags(&ctx->chain_done, hipEventDisableTiming));
    VKX_HIP(hipEventRecord(ctx->chain_done, stream));
    return VKX_OK;
}

int vkx_chain_fused_try(vkx_ctx *ctx, const vkx_chain_item *items, int n_items)
{
    vkx_chain_plan *raw = nullptr;
    int rc = vkx_chain_plan_build(ctx, items, n_items, &raw);
    if (rc || !raw) return rc;
    std::unique_ptr<vkx_chain_plan, void (*)(vkx_chain_plan *)> plan(raw, vkx_chain_plan_free);
    hipStream_t main_stream = ctx->stream;
    auto schedule = [&]() -> int {
        int rc;
        hipEvent_t setup_done = nullptr;
        if ((rc = vkx_chain_plan_setup_aside(ctx, raw, &setup_done))) return rc;
        if (setup_done) VKX_HIP(hipStreamWaitEvent(ctx->stream, setup_done, 0));      // (the device copy of the descriptors is the prologue's)
        if ((rc = vkx_chain_plan_noise_rows(ctx, raw, 0, n_items))) return rc;
        if ((rc = vkx_chain_plan_tiles(ctx, raw, 0, n_items))) return rc;
        vkx_device_guard guard(ctx);
        return vkx_chain_mark_done(ctx, ctx->stream);
    };
    rc = schedule();
    if (rc) vkx_ctx_join_streams(ctx, main_stream);     // a setup left on the side stream must not race the caller's next call
    return rc;
}

// vkx_grid_remap through the tile kernel; VKX_ERR_UNSUPPORTED (no error set) for shapes it does not take.
int vkx_tile_remap_try(vkx_ctx *ctx, const vkx_elem *elems, int n_elems, int sh, int sw, const int32_t *src_vertices,
                       const int32_t *dst_vertices, int rows, int cols, int dh, int dw)
{
    if (n_elems < 1 || n_elems > 4 || rows < 2 || cols < 2) return VKX_ERR_UNSUPPORTED;
    if (sh > 32767 || sw > 32767 || dh > 32767 || dw > 32767 || sh < 1 || sw < 1 || dh < 1 || dw < 1) return VKX_ERR_UNSUPPORTED;
    vkx_chain_plan plan;
    plan.dev.resize(1);
    plan.prefix.resize(2);
    std::vector<int> &prefix = plan.prefix;
    ItemDev &d = plan.dev[0];
    memset(&d, 0, sizeof(d));
    d.sv = src_vertices; d.dv = dst_vertices;
    d.sh = sh; d.sw = sw; d.dh = dh; d.dw = dw; d.rows = rows; d.cols = cols;
    d.R = 0;
    const int Tw = tile_side(0), Th = tile_height(0);
    d.tiles_x = (dw + Tw - 1) / Tw; d.tiles_y = (dh + Th - 1) / Th;
    d.cell_base = 0;
    d.n_elems = n_elems;
    for (int e = 0; e < n_elems; e++) {
        if (!elems[e].src || !elems[e].dst) return VKX_ERR_UNSUPPORTED;
        if (!elems[e].is_f32 && elems[e].cn != 1 && elems[e].cn != 3 && elems[e].cn != 4) return VKX_ERR_UNSUPPORTED;
        if (elems[e].is_f32 && elems[e].cn != 1) return VKX_ERR_UNSUPPORTED;
        d.el[e].src = elems[e].src; d.el[e].dst = elems[e].dst;
        d.el[e].sstride = elems[e].src_stride; d.el[e].dstride = elems[e].dst_stride;
        d.el[e].cn = elems[e].cn; d.el[e].is_f32 = elems[e].is_f32;
    }
    const long long ncells = (long long)(rows - 1) * (cols - 1);
    if ((long long)d.tiles_x * d.tiles_y > 0x3fffffff || ncells > 0x3fffffff) return VKX_ERR_UNSUPPORTED;
    prefix[0] = 0; prefix[1] = (int)ncells;
    plan.elements = true;
    plan.ncells = ncells;
    plan.max_tiles = d.tiles_x * d.tiles_y;
    int rc = vkx_chain_plan_setup(ctx, &plan);
    if (rc) return rc;
    return vkx_chain_plan_tiles(ctx, &plan, 0, 1);
}
