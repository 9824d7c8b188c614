// Image-grid distortions on gfx950, single-image entry points: per-cell inverse homography, exact cv.fillPoly
// cell ownership and the map / multi-element gather.  Restates ImageGrid.generate_remap_params
// (mechanism/distortion/geometric/grid_rendering/type.py:209-261) without its Python cell loop:
//
//   k_cell_setup   one lane per cell: closed-form dst->src homography (Jacobi-SVD least squares for cells
//                  with three collinear vertices), the 16.16 fixed-point edge table of cv.fillPoly.
//   k_cell_raster  one wavefront per cell: Bresenham outline + even-odd scanline spans, resolved with
//                  atomicMax(owner, cell+1) -- "the later cell in row-major order wins" (type.py:222-256).
//   k_owner_*      one lane per destination pixel: inv_H * (x, y, 1) with the FMA chain of the reference's
//                  dgemm, then either the float32 map or the bilinear gather of every element.
//
// The batched RGB chain has its own tile-fused kernel (fused.hip); these kernels serve the generic API
// (any element mix, owner / map outputs) and are the in-library reference the fused kernel is tested against.
#include "vkx_internal.h"
#include "vkx_cell.h"

#include <stdlib.h>
#include <string.h>

namespace {

using vkc::CellC;

__global__ void __launch_bounds__(64) k_cell_setup(const int32_t *__restrict__ src_v, const int32_t *__restrict__ dst_v,
                                                   int rows, int cols, CellC *__restrict__ cells)
{
    const int ncell = (rows - 1) * (cols - 1);
    const int cell = blockIdx.x * 64 + threadIdx.x;
    if (cell >= ncell) return;
    CellC rec;
    int xmin, xmax, ymin, ymax;
    vkc::build_cell(src_v, dst_v, rows, cols, cell, rec, xmin, xmax, ymin, ymax);
    cells[cell] = rec;
}

__device__ __forceinline__ void claim(int32_t *owner, int dh, int dw, int x, int y, const CellC &c, int tag)
{
    if ((unsigned)x >= (unsigned)dw || (unsigned)y >= (unsigned)dh) return;
    if (c.flags & 1) {
        const double de = fma(1.0, 1.0, fma(c.H[7], (double)y, c.H[6] * (double)x));
        if (de == 0) return;
    }
    atomicMax(&owner[(size_t)y * dw + x], tag);
}

__global__ void __launch_bounds__(64) k_cell_raster(const CellC *__restrict__ cells, int ncell, int32_t *owner, int dh,
                                                    int dw)
{
    const int cell = blockIdx.x;
    if (cell >= ncell) return;
    const CellC &c = cells[cell];
    const int lane = threadIdx.x, tag = cell + 1;
    int xmin = INT_MAX, xmax = INT_MIN, ylo = INT_MAX, yhi = INT_MIN;
    for (int k = 0; k < 4; k++) {
        xmin = min(xmin, (int)c.vx[k]); xmax = max(xmax, (int)c.vx[k]);
        ylo = min(ylo, (int)c.vy[k]); yhi = max(yhi, (int)c.vy[k]);
    }
    // 1. outline: 8-connected Bresenham of every edge, walked from its left end (LineIterator, left_to_right)
    for (int i = 0; i < 4; i++) {
        const int a = (i + 3) & 3;
        int lx = c.vx[a], ly = c.vy[a], rx = c.vx[i], ry = c.vy[i];
        if (rx < lx) { const int tx = lx, ty = ly; lx = rx; ly = ry; rx = tx; ry = ty; }
        const int dx = rx - lx, dy = ry - ly, ady = dy < 0 ? -dy : dy, sy = dy < 0 ? -1 : 1;
        if (ady > dx) {
            for (int k = lane; k <= ady; k += 64) claim(owner, dh, dw, lx + vkc::bres_minor(k, ady, dx), ly + sy * k, c, tag);
        } else {
            for (int k = lane; k <= dx; k += 64) claim(owner, dh, dw, lx + k, ly + sy * vkc::bres_minor(k, dx, ady), c, tag);
        }
    }
    // 2. interior: scanlines y0 <= y < y1 of the non-horizontal edges, x-sorted crossings paired even-odd,
    //    span [ceil(xa), floor(xb)] in 16.16 fixed point (FillEdgeCollection)
    for (int y = ylo + lane; y < yhi; y += 64) {
        int xs[4], n = 0;
        for (int i = 0; i < 4; i++) {
            const int a = (i + 3) & 3;
            const int ya = c.vy[a], yb = c.vy[i];
            const int e0 = min(ya, yb), e1 = max(ya, yb);
            if (e0 != e1 && e0 <= y && y < e1) xs[n++] = c.ex[i] + (y - e0) * c.edx[i];
        }
        for (int a = 1; a < n; a++) {
            const int v = xs[a];
            int b = a - 1;
            while (b >= 0 && xs[b] > v) { xs[b + 1] = xs[b]; b--; }
            xs[b + 1] = v;
        }
        for (int a = 0; a + 1 < n; a += 2) {
            const int x1 = max((xs[a] + 65535) >> 16, xmin), x2 = min(xs[a + 1] >> 16, xmax);
            for (int x = x1; x <= x2; x++) claim(owner, dh, dw, x, y, c, tag);
        }
    }
}

// inv_H * (x, y, 1): the k = 0,1,2 FMA accumulation of the reference's dgemm, then two IEEE divisions
// and the float64 -> float32 store of map_x / map_y (type.py:226-256).
__device__ __forceinline__ void map_at(const CellC *__restrict__ cells, int o, int x, int y, float &mx, float &my)
{
    if (o == 0) { mx = 0.f; my = 0.f; return; }
    const double *H = cells[o - 1].H;
    const double fx = (double)x, fy = (double)y;
    const double nx = fma(H[2], 1.0, fma(H[1], fy, H[0] * fx));
    const double ny = fma(H[5], 1.0, fma(H[4], fy, H[3] * fx));
    const double de = fma(1.0, 1.0, fma(H[7], fy, H[6] * fx));
    mx = (float)(nx / de);
    my = (float)(ny / de);
}

__global__ void __launch_bounds__(256) k_owner_to_map(const CellC *__restrict__ cells, const int32_t *__restrict__ owner,
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

__global__ void __launch_bounds__(256) k_owner_remap(const CellC *__restrict__ cells, const int32_t *__restrict__ owner,
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
    VKX_REQUIRE(dh > 0 && dw > 0 && dh <= 32767 && dw <= 32767, "destination shape must be within 1..32767");
    const int ncell = (rows - 1) * (cols - 1);
    int rc = vkx_scratch_reserve(ctx, &ctx->cells, sizeof(CellC) * (size_t)ncell);
    if (rc) return rc;
    VKX_HIP(hipMemsetAsync(owner, 0, sizeof(int32_t) * (size_t)dh * dw, ctx->stream));
    CellC *cells = (CellC *)ctx->cells.ptr;
    { VKX_TIMED(ctx, "k_cell_setup"); k_cell_setup<<<vkx_blocks(ncell, 64), 64, 0, ctx->stream>>>(src_v, dst_v, rows, cols, cells); }
    VKX_LAUNCH_CHECK();
    { VKX_TIMED(ctx, "k_cell_raster"); k_cell_raster<<<ncell, 64, 0, ctx->stream>>>(cells, ncell, owner, dh, dw); }
    VKX_LAUNCH_CHECK();
    return VKX_OK;
}


// func_point of the grid-based distortions (grid_rendering/interface.py:194-216), one lane per point: the cell is
// (y // grid_size, x // grid_size) of the ROUNDED point, its forward homography src quad -> dst quad is solved like
// get_trans_mat (grid_rendering/type.py:166-180), and the SMOOTH point goes through it in double.  The three
// accumulations follow the reference's np.matmul(trans_mat, (x, y, 1.0)) as OpenBLAS' dgemv evaluates a length-3 row:
// fma(h2, 1.0, fma(h0, x, h1 * y)).
__global__ void __launch_bounds__(64) k_project_points(const int32_t *__restrict__ src_v, const int32_t *__restrict__ dst_v,
                                                       int rows, int cols, int grid_size, const int32_t *__restrict__ pts_i,
                                                       const double *__restrict__ pts_s, int n, double *__restrict__ out,
                                                       int *__restrict__ bad)
{
    // eight lanes per point: a cell whose quads are not in general position is solved by the eight-lane Jacobi solver
    // (vkc::homography_jacobi_group8: registers only, no scratch segment); the closed form is computed by all eight alike
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    const bool writer = (threadIdx.x & 7) == 0;
    if (i >= n) return;
    const int px = pts_i[2 * i], py = pts_i[2 * i + 1];
    // python floor division
    int r = py / grid_size, c = px / grid_size;
    if (py % grid_size != 0 && (py < 0) != (grid_size < 0)) r--;
    if (px % grid_size != 0 && (px < 0) != (grid_size < 0)) c--;
    if (r < 0 || c < 0 || r >= rows - 1 || c >= cols - 1) {      // the reference indexes out of its cell table here
        if (writer) {
            *(volatile int *)bad = i + 1;      // any offending index serves (plain store: the flag may live in mapped host memory)
            out[2 * i] = 0.0; out[2 * i + 1] = 0.0;
        }
        return;
    }
    const int idx[4] = {r * cols + c, r * cols + c + 1, (r + 1) * cols + c + 1, (r + 1) * cols + c};
    float from[8], to[8];
    double qf[8], qt[8], H[9];
    for (int k = 0; k < 4; k++) {
        from[2 * k] = (float)src_v[2 * idx[k]]; from[2 * k + 1] = (float)src_v[2 * idx[k] + 1];
        to[2 * k] = (float)dst_v[2 * idx[k]]; to[2 * k + 1] = (float)dst_v[2 * idx[k] + 1];
        qf[2 * k] = from[2 * k]; qf[2 * k + 1] = from[2 * k + 1];
        qt[2 * k] = to[2 * k]; qt[2 * k + 1] = to[2 * k + 1];
    }
    if (!vkc::homography_direct(qf, qt, H)) vkc::homography_jacobi_group8(from, to, H);
    if (!writer) return;
    const double sx = pts_s[2 * i], sy = pts_s[2 * i + 1];
    const double tx = fma(H[2], 1.0, fma(H[0], sx, H[1] * sy));
    const double ty = fma(H[5], 1.0, fma(H[3], sx, H[4] * sy));
    const double t = fma(H[8], 1.0, fma(H[6], sx, H[7] * sy));
    out[2 * i] = tx / t;
    out[2 * i + 1] = ty / t;
}

} // namespace

VKX_EXPORT int vkx_grid_project_points(vkx_ctx *ctx, const int32_t *src_vertices_host, const int32_t *dst_vertices_host,
                                       int rows, int cols, int grid_size, const int32_t *pts_xy_host,
                                       const double *pts_smooth_xy_host, int n, double *out_xy_host)
{
    VKX_REQUIRE(ctx && src_vertices_host && dst_vertices_host, "NULL argument");
    VKX_REQUIRE(rows >= 2 && cols >= 2 && grid_size > 0 && n >= 0, "bad lattice");
    if (n == 0) return VKX_OK;
    VKX_REQUIRE(pts_xy_host && pts_smooth_xy_host && out_xy_host, "NULL argument");
    const size_t vbytes = sizeof(int32_t) * 2 * (size_t)rows * cols;
    const size_t ibytes = sizeof(int32_t) * 2 * (size_t)n, dbytes = sizeof(double) * 2 * (size_t)n;
    const size_t off_dv = (vbytes + 255) & ~(size_t)255, off_pi = off_dv * 2, off_ps = off_pi + ((ibytes + 255) & ~(size_t)255);
    const size_t off_out = off_ps + ((dbytes + 255) & ~(size_t)255), off_bad = off_out + ((dbytes + 255) & ~(size_t)255);
    // Everything the kernel reads, and what it writes, in ONE block of the page-locked (mapped) ring: the lattices and the points are
    // read once, the results are read by the host right after -- the kernel works on the host block in place and the call is one
    // dispatch (it was four copies in, a memset, the kernel and two copies out: 5 - 7 dispatches of a page's 98).  The flag is a plain
    // store (any offending index serves; atomics on host memory need PCIe atomics).
    void *ring = nullptr;
    int rc = off_bad + 256 <= ((size_t)8 << 20) ? vkx_desc_ring_take(ctx, off_bad + 256, &ring) : VKX_ERR_UNSUPPORTED;
    const uint8_t *mapped = rc == VKX_OK ? (const uint8_t *)vkx_ring_device_ptr(ring) : nullptr;
    if (mapped) {
        uint8_t *host = (uint8_t *)ring;
        memcpy(host, src_vertices_host, vbytes);
        memcpy(host + off_dv, dst_vertices_host, vbytes);
        memcpy(host + off_pi, pts_xy_host, ibytes);
        memcpy(host + off_ps, pts_smooth_xy_host, dbytes);
        *(volatile int *)(host + off_bad) = 0;
        vkx_device_guard guard(ctx);
        { VKX_TIMED(ctx, "k_project_points"); k_project_points<<<vkx_blocks((size_t)n * 8, 64), 64, 0, ctx->stream>>>((const int32_t *)mapped, (const int32_t *)(mapped + off_dv), rows, cols, grid_size, (const int32_t *)(mapped + off_pi), (const double *)(mapped + off_ps), n, (double *)(mapped + off_out), (int *)(mapped + off_bad)); }
        VKX_LAUNCH_CHECK();
        VKX_HIP(hipStreamSynchronize(ctx->stream));
        const int bad = *(volatile int *)(host + off_bad);
        if (bad) {
            vkx_set_error("point %d lies outside the lattice cells", bad - 1);
            return VKX_ERR_OUT_OF_LATTICE;
        }
        memcpy(out_xy_host, host + off_out, dbytes);
        return VKX_OK;
    }
    rc = vkx_scratch_reserve(ctx, &ctx->stage[0], off_bad + 256);
    if (rc) return rc;
    uint8_t *base = (uint8_t *)ctx->stage[0].ptr;
    VKX_HIP(hipMemcpyAsync(base, src_vertices_host, vbytes, hipMemcpyHostToDevice, ctx->stream));
    VKX_HIP(hipMemcpyAsync(base + off_dv, dst_vertices_host, vbytes, hipMemcpyHostToDevice, ctx->stream));
    VKX_HIP(hipMemcpyAsync(base + off_pi, pts_xy_host, ibytes, hipMemcpyHostToDevice, ctx->stream));
    VKX_HIP(hipMemcpyAsync(base + off_ps, pts_smooth_xy_host, dbytes, hipMemcpyHostToDevice, ctx->stream));
    VKX_HIP(hipMemsetAsync(base + off_bad, 0, sizeof(int), ctx->stream));
    { VKX_TIMED(ctx, "k_project_points"); k_project_points<<<vkx_blocks((size_t)n * 8, 64), 64, 0, ctx->stream>>>((const int32_t *)base, (const int32_t *)(base + off_dv), rows, cols, grid_size, (const int32_t *)(base + off_pi), (const double *)(base + off_ps), n, (double *)(base + off_out), (int *)(base + off_bad)); }
    VKX_LAUNCH_CHECK();
    int bad = 0;
    VKX_HIP(hipMemcpyAsync(out_xy_host, base + off_out, dbytes, hipMemcpyDeviceToHost, ctx->stream));
    VKX_HIP(hipMemcpyAsync(&bad, base + off_bad, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    VKX_HIP(hipStreamSynchronize(ctx->stream));
    if (bad) {
        vkx_set_error("point %d lies outside the lattice cells", bad - 1);
        return VKX_ERR_OUT_OF_LATTICE;
    }
    return VKX_OK;
}

VKX_EXPORT int vkx_grid_to_map_dev(vkx_ctx *ctx, const int32_t *src_vertices, const int32_t *dst_vertices, int rows,
                                   int cols, int dh, int dw, float *map_x, float *map_y, ptrdiff_t map_stride_el,
                                   int32_t *owner)
{
    VKX_REQUIRE(ctx && map_x && map_y, "NULL argument");
    VKX_REQUIRE(dh > 0 && dw > 0, "bad destination shape");
    int32_t *own = owner;
    if (!own) {
        int rc = vkx_scratch_reserve(ctx, &ctx->owner, sizeof(int32_t) * (size_t)dh * dw);
        if (rc) return rc;
        own = (int32_t *)ctx->owner.ptr;
    }
    int rc = build_owner(ctx, src_vertices, dst_vertices, rows, cols, dh, dw, own);
    if (rc) return rc;
    dim3 block(64, 4), grid(vkx_blocks(dw, 64), vkx_blocks(dh, 4));
    { VKX_TIMED(ctx, "k_owner_to_map"); k_owner_to_map<<<grid, block, 0, ctx->stream>>>((const CellC *)ctx->cells.ptr, own, dh, dw, map_x, map_y, map_stride_el); }
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
    VKX_REQUIRE(dh > 0 && dw > 0, "bad destination shape");
    ElemPack pack;
    pack.n = n_elems;
    for (int i = 0; i < n_elems; i++) {
        pack.e[i] = elems[i];
        VKX_REQUIRE(elems[i].src && elems[i].dst, "NULL element plane");
        if (elems[i].is_f32) VKX_REQUIRE(elems[i].cn == 1, "float32 elements are single channel");
        else VKX_REQUIRE(elems[i].cn == 1 || elems[i].cn == 3 || elems[i].cn == 4, "uint8 elements need 1, 3 or 4 channels");
    }
    // tile kernel first (ownership in LDS, one launch for all elements); the global-ownership-plane kernels below take
    // whatever it declines (VKX_GRID_GLOBAL=1 forces them: the two paths must agree bit for bit)
    static const bool force_global = getenv("VKX_GRID_GLOBAL") != nullptr;
    int rc = force_global ? VKX_ERR_UNSUPPORTED
                          : vkx_tile_remap_try(ctx, elems, n_elems, sh, sw, src_vertices, dst_vertices, rows, cols, dh, dw);
    if (rc != VKX_ERR_UNSUPPORTED) return rc;
    rc = vkx_scratch_reserve(ctx, &ctx->owner, sizeof(int32_t) * (size_t)dh * dw);
    if (rc) return rc;
    int32_t *own = (int32_t *)ctx->owner.ptr;
    rc = build_owner(ctx, src_vertices, dst_vertices, rows, cols, dh, dw, own);
    if (rc) return rc;
    dim3 block(64, 4), grid(vkx_blocks(dw, 64), vkx_blocks(dh, 4));
    { VKX_TIMED(ctx, "k_owner_remap"); k_owner_remap<<<grid, block, 0, ctx->stream>>>((const CellC *)ctx->cells.ptr, own, dh, dw, sh, sw, pack); }
    VKX_LAUNCH_CHECK();
    return VKX_OK;
}
