// Image-grid distortions on gfx950: per-cell inverse homography, exact cv.fillPoly cell ownership and the
// map / fused multi-element gather.  Restates ImageGrid.generate_remap_params
// (mechanism/distortion/geometric/grid_rendering/type.py:209-261) without its Python cell loop:
//
//   k_cell_setup   one lane per cell: closed-form dst->src homography (Jacobi-SVD least squares for cells
//                  with three collinear vertices), the 16.16 fixed-point edge table of cv.fillPoly.
//   k_cell_raster  one wavefront per cell: Bresenham outline + even-odd scanline spans, resolved with
//                  atomicMax(owner, cell+1) -- "the later cell in row-major order wins" (type.py:222-256).
//   k_owner_*      one lane per destination pixel: inv_H * (x, y, 1) with the FMA chain of the reference's
//                  dgemm, then either the float32 map or the bilinear gather of every element.
#include "vkx_internal.h"

#include <float.h>

namespace {

struct CellRec {
    double H[9];
    long long ex[4];   // 16.16 x of edge i at its upper end
    long long edx[4];  // 16.16 dx per scanline
    int ey0[4], ey1[4];
    int vx[4], vy[4];  // destination quad, clockwise from (row, col)
    int flags;         // bit 0: denominator may vanish inside the bounding box
    int pad;
};

// ---------------------------------------------------------------------------------------------------
// Homography (see oracle/vkx_oracle.c homography_direct / homography_jacobi for the specification).
// ---------------------------------------------------------------------------------------------------
__device__ void square_to_quad_scaled(const double q[8], double G[9])
{
    const double x0 = q[0], y0 = q[1], x1 = q[2], y1 = q[3], x2 = q[4], y2 = q[5], x3 = q[6], y3 = q[7];
    const double sx = x0 - x1 + x2 - x3, sy = y0 - y1 + y2 - y3;
    const double dx1 = x1 - x2, dy1 = y1 - y2, dx2 = x3 - x2, dy2 = y3 - y2;
    const double den = dx1 * dy2 - dx2 * dy1;
    const double g = sx * dy2 - dx2 * sy;
    const double h = dx1 * sy - sx * dy1;
    G[0] = den * (x1 - x0) + g * x1; G[1] = den * (x3 - x0) + h * x3; G[2] = den * x0;
    G[3] = den * (y1 - y0) + g * y1; G[4] = den * (y3 - y0) + h * y3; G[5] = den * y0;
    G[6] = g;                        G[7] = h;                        G[8] = den;
}

__device__ bool quad_in_general_position(const double q[8])
{
    for (int a = 0; a < 4; a++) {
        const int b = (a + 1) & 3, c = (a + 2) & 3;
        const double cr = (q[2 * b] - q[2 * a]) * (q[2 * c + 1] - q[2 * a + 1]) -
                          (q[2 * b + 1] - q[2 * a + 1]) * (q[2 * c] - q[2 * a]);
        if (cr == 0) return false;
    }
    return true;
}

__device__ bool homography_direct(const double qf[8], const double qt[8], double H[9])
{
    if (!quad_in_general_position(qf) || !quad_in_general_position(qt)) return false;
    double Gf[9], Gt[9], Af[9], Hp[9];
    square_to_quad_scaled(qf, Gf);
    square_to_quad_scaled(qt, Gt);
    Af[0] = Gf[4] * Gf[8] - Gf[5] * Gf[7];
    Af[1] = Gf[2] * Gf[7] - Gf[1] * Gf[8];
    Af[2] = Gf[1] * Gf[5] - Gf[2] * Gf[4];
    Af[3] = Gf[5] * Gf[6] - Gf[3] * Gf[8];
    Af[4] = Gf[0] * Gf[8] - Gf[2] * Gf[6];
    Af[5] = Gf[2] * Gf[3] - Gf[0] * Gf[5];
    Af[6] = Gf[3] * Gf[7] - Gf[4] * Gf[6];
    Af[7] = Gf[1] * Gf[6] - Gf[0] * Gf[7];
    Af[8] = Gf[0] * Gf[4] - Gf[1] * Gf[3];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++)
            Hp[r * 3 + c] = (Gt[r * 3] * Af[c] + Gt[r * 3 + 1] * Af[3 + c]) + Gt[r * 3 + 2] * Af[6 + c];
    if (Hp[8] == 0 || !isfinite(Hp[8])) return false;
    for (int i = 0; i < 8; i++) H[i] = Hp[i] / Hp[8];
    H[8] = 1.;
    return true;
}

__device__ double vk_hypot(double a, double b)
{
    a = fabs(a); b = fabs(b);
    if (a < b) { const double t = a; a = b; b = t; }
    if (a == 0) return 0;
    const double r = b / a;
    return a * sqrt(1 + r * r);
}

// One-sided Jacobi SVD of the 8x8 DLT system + back substitution with the 2*eps*sum(w) cut-off.
// Only reached for degenerate quads (a handful per image at most), so it is written for clarity.
__device__ void homography_jacobi(const float from[8], const float to[8], double H[9])
{
    double At[8][8], W[8], Vt[8][8], b[8], x[8];
    for (int i = 0; i < 4; i++) {
        const float fx = from[2 * i], fy = from[2 * i + 1], tx = to[2 * i], ty = to[2 * i + 1];
        // a[r][c] with r = i / i+4, stored transposed: At[c][r]
        for (int c = 0; c < 8; c++) { At[c][i] = 0; At[c][i + 4] = 0; }
        At[0][i] = fx; At[1][i] = fy; At[2][i] = 1;
        At[3][i + 4] = fx; At[4][i + 4] = fy; At[5][i + 4] = 1;
        At[6][i] = (double)(-fx * tx);
        At[7][i] = (double)(-fy * tx);
        At[6][i + 4] = (double)(-fx * ty);
        At[7][i + 4] = (double)(-fy * ty);
        b[i] = tx;
        b[i + 4] = ty;
    }
    const int m = 8, n = 8;
    const double eps = DBL_EPSILON * 10, minval = DBL_MIN;
    double c, s, sd;
    for (int i = 0; i < n; i++) {
        sd = 0;
        for (int k = 0; k < m; k++) { const double t = At[i][k]; sd += t * t; }
        W[i] = sd;
        for (int k = 0; k < n; k++) Vt[i][k] = 0;
        Vt[i][i] = 1;
    }
    for (int iter = 0; iter < 30; iter++) {
        bool changed = false;
        for (int i = 0; i < n - 1; i++)
            for (int j = i + 1; j < n; j++) {
                double a = W[i], p = 0, bb = W[j];
                for (int k = 0; k < m; k++) p += At[i][k] * At[j][k];
                if (fabs(p) <= eps * sqrt(a * bb)) continue;
                p *= 2;
                const double beta = a - bb, gamma = vk_hypot(p, beta);
                if (beta < 0) {
                    const double delta = (gamma - beta) * 0.5;
                    s = sqrt(delta / gamma);
                    c = p / (gamma * s * 2);
                } else {
                    c = sqrt((gamma + beta) / (gamma * 2));
                    s = p / (gamma * c * 2);
                }
                a = bb = 0;
                for (int k = 0; k < m; k++) {
                    const double t0 = c * At[i][k] + s * At[j][k];
                    const double t1 = -s * At[i][k] + c * At[j][k];
                    At[i][k] = t0; At[j][k] = t1;
                    a += t0 * t0; bb += t1 * t1;
                }
                W[i] = a; W[j] = bb;
                changed = true;
                for (int k = 0; k < n; k++) {
                    const double t0 = c * Vt[i][k] + s * Vt[j][k];
                    const double t1 = -s * Vt[i][k] + c * Vt[j][k];
                    Vt[i][k] = t0; Vt[j][k] = t1;
                }
            }
        if (!changed) break;
    }
    for (int i = 0; i < n; i++) {
        sd = 0;
        for (int k = 0; k < m; k++) { const double t = At[i][k]; sd += t * t; }
        W[i] = sqrt(sd);
    }
    for (int i = 0; i < n - 1; i++) {
        int j = i;
        for (int k = i + 1; k < n; k++)
            if (W[j] < W[k]) j = k;
        if (i != j) {
            double t = W[i]; W[i] = W[j]; W[j] = t;
            for (int k = 0; k < m; k++) { t = At[i][k]; At[i][k] = At[j][k]; At[j][k] = t; }
            for (int k = 0; k < n; k++) { t = Vt[i][k]; Vt[i][k] = Vt[j][k]; Vt[j][k] = t; }
        }
    }
    for (int i = 0; i < n; i++) {
        sd = W[i];
        s = sd > minval ? 1 / sd : 0.;
        for (int k = 0; k < m; k++) At[i][k] *= s;
    }
    double threshold = 0;
    for (int i = 0; i < 8; i++) { x[i] = 0; threshold += W[i]; }
    threshold *= DBL_EPSILON * 2;
    for (int i = 0; i < 8; i++) {
        double wi = W[i];
        if (fabs(wi) <= threshold) continue;
        wi = 1 / wi;
        double acc = 0;
        for (int j = 0; j < 8; j++) acc += At[i][j] * b[j];
        acc *= wi;
        for (int j = 0; j < 8; j++) x[j] = x[j] + acc * Vt[i][j];
    }
    for (int i = 0; i < 8; i++) H[i] = x[i];
    H[8] = 1.;
}

__global__ void __launch_bounds__(64) k_cell_setup(const int32_t *__restrict__ src_v, const int32_t *__restrict__ dst_v,
                                                   int rows, int cols, CellRec *__restrict__ cells)
{
    const int ncell = (rows - 1) * (cols - 1);
    const int cell = blockIdx.x * 64 + threadIdx.x;
    if (cell >= ncell) return;
    const int r = cell / (cols - 1), c = cell - r * (cols - 1);
    const int idx[4] = {r * cols + c, r * cols + c + 1, (r + 1) * cols + c + 1, (r + 1) * cols + c};
    float from[8], to[8];
    double qf[8], qt[8];
    CellRec rec;
    int xmin = INT_MAX, xmax = INT_MIN, ymin = INT_MAX, ymax = INT_MIN;
    for (int k = 0; k < 4; k++) {
        const int dx = dst_v[2 * idx[k]], dy = dst_v[2 * idx[k] + 1];
        rec.vx[k] = dx; rec.vy[k] = dy;
        from[2 * k] = (float)dx; from[2 * k + 1] = (float)dy;
        to[2 * k] = (float)src_v[2 * idx[k]]; to[2 * k + 1] = (float)src_v[2 * idx[k] + 1];
        qf[2 * k] = from[2 * k]; qf[2 * k + 1] = from[2 * k + 1];
        qt[2 * k] = to[2 * k]; qt[2 * k + 1] = to[2 * k + 1];
        xmin = min(xmin, dx); xmax = max(xmax, dx);
        ymin = min(ymin, dy); ymax = max(ymax, dy);
    }
    if (!homography_direct(qf, qt, rec.H)) homography_jacobi(from, to, rec.H);
    // edge i runs from vertex (i+3)&3 to vertex i, exactly as CollectPolyEdges walks the contour
    for (int i = 0; i < 4; i++) {
        const int a = (i + 3) & 3;
        const long long xa = (long long)rec.vx[a] << 16, xb = (long long)rec.vx[i] << 16;
        const int ya = rec.vy[a], yb = rec.vy[i];
        if (ya == yb) { rec.ey0[i] = 0; rec.ey1[i] = 0; rec.ex[i] = 0; rec.edx[i] = 0; continue; }
        rec.edx[i] = (xb - xa) / (long long)(yb - ya);
        if (ya < yb) { rec.ey0[i] = ya; rec.ey1[i] = yb; rec.ex[i] = xa; }
        else         { rec.ey0[i] = yb; rec.ey1[i] = ya; rec.ex[i] = xb; }
    }
    // Can the projective denominator vanish on the cell's bounding box?  It is affine in (x, y), so its
    // extrema sit at the corners.
    double dmin = DBL_MAX, dmax = -DBL_MAX;
    const int cx[2] = {xmin, xmax}, cy[2] = {ymin, ymax};
    for (int a = 0; a < 2; a++)
        for (int b = 0; b < 2; b++) {
            const double d = rec.H[6] * cx[a] + rec.H[7] * cy[b] + rec.H[8];
            dmin = fmin(dmin, d); dmax = fmax(dmax, d);
        }
    rec.flags = (dmin > 1e-6 || dmax < -1e-6) && isfinite(dmin) && isfinite(dmax) ? 0 : 1;
    rec.pad = 0;
    cells[cell] = rec;
}

__device__ __forceinline__ int bres_minor(int k, int dmaj, int dmin)
{
    if (dmaj == 0) return 0;
    const long long num = 2LL * k * dmin - dmaj;
    if (num <= 0) return 0;
    return (int)((num + 2LL * dmaj - 1) / (2LL * dmaj));
}

__device__ __forceinline__ void claim(int32_t *owner, int dh, int dw, int x, int y, const CellRec &c, int tag)
{
    if ((unsigned)x >= (unsigned)dw || (unsigned)y >= (unsigned)dh) return;
    if (c.flags & 1) {
        const double de = fma(c.H[8], 1.0, fma(c.H[7], (double)y, c.H[6] * (double)x));
        if (de == 0) return;
    }
    atomicMax(&owner[(size_t)y * dw + x], tag);
}

__global__ void __launch_bounds__(64) k_cell_raster(const CellRec *__restrict__ cells, int ncell, int32_t *owner,
                                                    int dh, int dw)
{
    const int cell = blockIdx.x;
    if (cell >= ncell) return;
    const CellRec &c = cells[cell];
    const int lane = threadIdx.x, tag = cell + 1;
    int xmin = INT_MAX, xmax = INT_MIN;
    for (int k = 0; k < 4; k++) { xmin = min(xmin, c.vx[k]); xmax = max(xmax, c.vx[k]); }
    // 1. outline: 8-connected Bresenham of every edge, walked from its left end (LineIterator, left_to_right)
    for (int i = 0; i < 4; i++) {
        const int a = (i + 3) & 3;
        int lx = c.vx[a], ly = c.vy[a], rx = c.vx[i], ry = c.vy[i];
        if (rx < lx) { const int tx = lx, ty = ly; lx = rx; ly = ry; rx = tx; ry = ty; }
        const int dx = rx - lx, dy = ry - ly, ady = dy < 0 ? -dy : dy, sy = dy < 0 ? -1 : 1;
        if (ady > dx) {
            for (int k = lane; k <= ady; k += 64) claim(owner, dh, dw, lx + bres_minor(k, ady, dx), ly + sy * k, c, tag);
        } else {
            for (int k = lane; k <= dx; k += 64) claim(owner, dh, dw, lx + k, ly + sy * bres_minor(k, dx, ady), c, tag);
        }
    }
    // 2. interior: scanlines y0 <= y < y1 of the non-horizontal edges, x-sorted crossings paired even-odd,
    //    span [ceil(xa), floor(xb)] in 16.16 fixed point (FillEdgeCollection)
    int ylo = INT_MAX, yhi = INT_MIN;
    for (int i = 0; i < 4; i++)
        if (c.ey0[i] != c.ey1[i]) { ylo = min(ylo, c.ey0[i]); yhi = max(yhi, c.ey1[i]); }
    for (int y = ylo + lane; y < yhi; y += 64) {
        long long xs[4];
        int n = 0;
        for (int i = 0; i < 4; i++)
            if (c.ey0[i] != c.ey1[i] && c.ey0[i] <= y && y < c.ey1[i])
                xs[n++] = c.ex[i] + (long long)(y - c.ey0[i]) * c.edx[i];
        for (int a = 1; a < n; a++) {
            const long long v = xs[a];
            int b = a - 1;
            while (b >= 0 && xs[b] > v) { xs[b + 1] = xs[b]; b--; }
            xs[b + 1] = v;
        }
        for (int a = 0; a + 1 < n; a += 2) {
            int x1 = (int)((xs[a] + 65535) >> 16), x2 = (int)(xs[a + 1] >> 16);
            x1 = max(x1, xmin);
            x2 = min(x2, xmax);
            for (int x = x1; x <= x2; x++) claim(owner, dh, dw, x, y, c, tag);
        }
    }
}

// inv_H * (x, y, 1): the k = 0,1,2 FMA accumulation of the reference's dgemm, then two IEEE divisions
// and the float64 -> float32 store of map_x / map_y (type.py:226-256).
__device__ __forceinline__ void map_at(const CellRec *__restrict__ cells, int o, int x, int y, float &mx, float &my)
{
    if (o == 0) { mx = 0.f; my = 0.f; return; }
    const double *H = cells[o - 1].H;
    const double fx = (double)x, fy = (double)y;
    const double nx = fma(H[2], 1.0, fma(H[1], fy, H[0] * fx));
    const double ny = fma(H[5], 1.0, fma(H[4], fy, H[3] * fx));
    const double de = fma(H[8], 1.0, fma(H[7], fy, H[6] * fx));
    mx = (float)(nx / de);
    my = (float)(ny / de);
}

__global__ void __launch_bounds__(256) k_owner_to_map(const CellRec *__restrict__ cells, const int32_t *__restrict__ owner,
                                                      int dh, int dw, float *__restrict__ map_x, float *__restrict__ map_y,
                                                      ptrdiff_t mstride)
{
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x >= dw || y >= dh) return;
    float mx, my;
    map_at(cells, owner[(size_t)y * dw + x], x, y, mx, my);
    map_x[(ptrdiff_t)y * mstride + x] = mx;
    map_y[(ptrdiff_t)y * mstride + x] = my;
}

constexpr int kMaxElems = 4;
struct ElemPack {
    vkx_elem e[kMaxElems];
    int n;
};

__global__ void __launch_bounds__(256) k_owner_remap(const CellRec *__restrict__ cells, const int32_t *__restrict__ owner,
                                                     int dh, int dw, int sh, int sw, ElemPack pack)
{
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x >= dw || y >= dh) return;
    float mx, my;
    map_at(cells, owner[(size_t)y * dw + x], x, y, mx, my);
    const int X = vkd::cv_round(mx * 32.f), Y = vkd::cv_round(my * 32.f);
    for (int i = 0; i < pack.n; i++) {
        const vkx_elem &e = pack.e[i];
        if (e.is_f32) {
            ((float *)e.dst)[(ptrdiff_t)y * e.dst_stride + x] =
                vkd::sample_f32((const float *)e.src, sh, sw, e.src_stride, X, Y);
        } else {
            uint8_t *d = (uint8_t *)e.dst + (ptrdiff_t)y * e.dst_stride + (ptrdiff_t)x * e.cn;
            uint8_t px[4];
            if (e.cn == 3) vkd::sample_u8<3>((const uint8_t *)e.src, sh, sw, e.src_stride, X, Y, px);
            else if (e.cn == 1) vkd::sample_u8<1>((const uint8_t *)e.src, sh, sw, e.src_stride, X, Y, px);
            else vkd::sample_u8<4>((const uint8_t *)e.src, sh, sw, e.src_stride, X, Y, px);
            for (int k = 0; k < e.cn; k++) d[k] = px[k];
        }
    }
}

// Builds the cell table and the ownership plane on the ctx stream.
int build_owner(vkx_ctx *ctx, const int32_t *src_v, const int32_t *dst_v, int rows, int cols, int dh, int dw,
                int32_t *owner)
{
    VKX_REQUIRE(ctx && src_v && dst_v, "NULL argument");
    VKX_REQUIRE(rows >= 2 && cols >= 2, "grid needs at least 2x2 vertices");
    VKX_REQUIRE(dh > 0 && dw > 0, "bad destination shape");
    const int ncell = (rows - 1) * (cols - 1);
    int rc = vkx_scratch_reserve(ctx, &ctx->cells, sizeof(CellRec) * (size_t)ncell);
    if (rc) return rc;
    VKX_HIP(hipMemsetAsync(owner, 0, sizeof(int32_t) * (size_t)dh * dw, ctx->stream));
    CellRec *cells = (CellRec *)ctx->cells.ptr;
    { VKX_TIMED(ctx, "k_cell_setup"); k_cell_setup<<<vkx_blocks(ncell, 64), 64, 0, ctx->stream>>>(src_v, dst_v, rows, cols, cells); }
    VKX_LAUNCH_CHECK();
    { VKX_TIMED(ctx, "k_cell_raster"); k_cell_raster<<<ncell, 64, 0, ctx->stream>>>(cells, ncell, owner, dh, dw); }
    VKX_LAUNCH_CHECK();
    return VKX_OK;
}

} // namespace

VKX_EXPORT int vkx_grid_to_map_dev(vkx_ctx *ctx, const int32_t *src_vertices, const int32_t *dst_vertices, int rows,
                                   int cols, int dh, int dw, float *map_x, float *map_y, ptrdiff_t map_stride_el,
                                   int32_t *owner)
{
    VKX_REQUIRE(map_x && map_y, "NULL map");
    int32_t *own = owner;
    if (!own) {
        int rc = vkx_scratch_reserve(ctx, &ctx->owner, sizeof(int32_t) * (size_t)dh * dw);
        if (rc) return rc;
        own = (int32_t *)ctx->owner.ptr;
    }
    int rc = build_owner(ctx, src_vertices, dst_vertices, rows, cols, dh, dw, own);
    if (rc) return rc;
    dim3 block(64, 4), grid(vkx_blocks(dw, 64), vkx_blocks(dh, 4));
    { VKX_TIMED(ctx, "k_owner_to_map"); k_owner_to_map<<<grid, block, 0, ctx->stream>>>((const CellRec *)ctx->cells.ptr, own, dh, dw, map_x, map_y,
                                                    map_stride_el); }
    VKX_LAUNCH_CHECK();
    return VKX_OK;
}

VKX_EXPORT int vkx_grid_remap_dev(vkx_ctx *ctx, const vkx_elem *elems, int n_elems, int sh, int sw,
                                  const int32_t *src_vertices, const int32_t *dst_vertices, int rows, int cols, int dh,
                                  int dw)
{
    VKX_REQUIRE(ctx && elems, "NULL argument");
    VKX_REQUIRE(n_elems >= 1 && n_elems <= kMaxElems, "1..4 elements per call");
    VKX_REQUIRE(sh > 0 && sw > 0 && sh <= 32767 && sw <= 32767, "bad source shape");
    ElemPack pack;
    pack.n = n_elems;
    for (int i = 0; i < n_elems; i++) {
        pack.e[i] = elems[i];
        VKX_REQUIRE(elems[i].src && elems[i].dst, "NULL element plane");
        if (elems[i].is_f32) VKX_REQUIRE(elems[i].cn == 1, "float32 elements are single channel");
        else VKX_REQUIRE(elems[i].cn == 1 || elems[i].cn == 3 || elems[i].cn == 4, "uint8 elements need 1, 3 or 4 channels");
    }
    int rc = vkx_scratch_reserve(ctx, &ctx->owner, sizeof(int32_t) * (size_t)dh * dw);
    if (rc) return rc;
    int32_t *own = (int32_t *)ctx->owner.ptr;
    rc = build_owner(ctx, src_vertices, dst_vertices, rows, cols, dh, dw, own);
    if (rc) return rc;
    dim3 block(64, 4), grid(vkx_blocks(dw, 64), vkx_blocks(dh, 4));
    { VKX_TIMED(ctx, "k_owner_remap"); k_owner_remap<<<grid, block, 0, ctx->stream>>>((const CellRec *)ctx->cells.ptr, own, dh, dw, sh, sw, pack); }
    VKX_LAUNCH_CHECK();
    return VKX_OK;
}
