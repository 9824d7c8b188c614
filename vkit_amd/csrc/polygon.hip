// cv.fillPoly(mask, [pts], 1) for one polygon of any vertex count on gfx950
// (reference: PolygonInternals.np_mask, element/polygon.py:70-77; used by Polygon.fill_* and by the active mask
// of image-grid distortions, grid_rendering/interface.py:177-192).
//
//   k_poly_outline  one lane per (edge, major step): pixel of the 8-connected Bresenham line walked from the
//                   edge's left end (cv::LineIterator), closed form of the error recurrence.
//   k_poly_spans    one workgroup per scanline: lanes collect the 16.16 fixed-point crossings of the half-open
//                   edges (y0 <= y < y1) into LDS, sort them, and fill the even-odd spans
//                   [ceil(xa), floor(xb)] cooperatively.
// Vertices are host data (a few to a few thousand points); the edge table is built on the host and staged.
#include "vkx_internal.h"
#include <string.h>
#include "vkx_cell.h"

#include <algorithm>

namespace {

struct PolyEdge {
    int xa, ya, xb, yb;        // contour order
    int lx, ly, dmaj, dmin;    // Bresenham from the left end
    int sy, ymajor;
    int step_base;             // prefix of (dmaj + 1) over the edges
    int y0, y1;                // scanline range of the edge (y0 == y1: horizontal, not in the edge table)
    int poly, pad;             // batched paint: 1-based paint order of the polygon this edge belongs to
    long long x0_fix, dx_fix;  // 16.16 x at y0, dx per scanline
};

__global__ void __launch_bounds__(256) k_poly_outline(const PolyEdge *__restrict__ edges, int nedges, int total_steps,
                                                      uint8_t *__restrict__ mask, int h, int w, ptrdiff_t stride)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= total_steps) return;
    int lo = 0, hi = nedges - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (edges[mid].step_base <= t) lo = mid; else hi = mid - 1;
    }
    const PolyEdge e = edges[lo];
    const int k = t - e.step_base;
    const int m = vkc::bres_minor(k, e.dmaj, e.dmin);
    const int x = e.ymajor ? e.lx + m : e.lx + k;
    const int y = e.ymajor ? e.ly + e.sy * k : e.ly + e.sy * m;
    if ((unsigned)x < (unsigned)w && (unsigned)y < (unsigned)h) mask[(ptrdiff_t)y * stride + x] = 1;
}

constexpr int kMaxCross = 512;

__global__ void __launch_bounds__(256) k_poly_spans(const PolyEdge *__restrict__ edges, int nedges, int ymin,
                                                    uint8_t *__restrict__ mask, int h, int w, ptrdiff_t stride,
                                                    int *__restrict__ overflow)
{
    __shared__ long long xs[kMaxCross];
    __shared__ int count;
    const int y = ymin + blockIdx.x;
    if (threadIdx.x == 0) count = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < nedges; i += 256) {
        const PolyEdge &e = edges[i];
        if (e.y0 != e.y1 && e.y0 <= y && y < e.y1) {
            const int slot = atomicAdd(&count, 1);
            if (slot < kMaxCross) xs[slot] = e.x0_fix + (long long)(y - e.y0) * e.dx_fix;
        }
    }
    __syncthreads();
    int n = count;
    if (n > kMaxCross) {
        if (threadIdx.x == 0) atomicExch(overflow, 1);
        n = kMaxCross;
    }
    if (threadIdx.x == 0) {
        for (int a = 1; a < n; a++) {
            const long long v = xs[a];
            int b = a - 1;
            while (b >= 0 && xs[b] > v) { xs[b + 1] = xs[b]; b--; }
            xs[b + 1] = v;
        }
    }
    __syncthreads();
    if ((unsigned)y >= (unsigned)h) return;
    uint8_t *row = mask + (ptrdiff_t)y * stride;
    for (int a = 0; a + 1 < n; a += 2) {
        long long x1 = (xs[a] + 65535) >> 16, x2 = xs[a + 1] >> 16;
        if (x1 < 0) x1 = 0;
        if (x2 >= w) x2 = w - 1;
        for (long long x = x1 + threadIdx.x; x <= x2; x += 256) row[x] = 1;
    }
}

// Edge table of one closed contour (cv::fillPoly -> CollectPolyEdges + the LINE_8 outline).
void build_edges(const int32_t *pts, int npts, int poly, PolyEdge *edges, long long *steps, int *ymin, int *ymax)
{
    for (int i = 0; i < npts; i++) {
        const int a = (i + npts - 1) % npts;
        PolyEdge &e = edges[i];
        e.xa = pts[2 * a]; e.ya = pts[2 * a + 1];
        e.xb = pts[2 * i]; e.yb = pts[2 * i + 1];
        int lx = e.xa, ly = e.ya, rx = e.xb, ry = e.yb;
        if (rx < lx) { std::swap(lx, rx); std::swap(ly, ry); }
        const int dx = rx - lx, dy = ry - ly, ady = dy < 0 ? -dy : dy;
        e.lx = lx; e.ly = ly; e.sy = dy < 0 ? -1 : 1;
        e.ymajor = ady > dx;
        e.dmaj = e.ymajor ? ady : dx;
        e.dmin = e.ymajor ? dx : ady;
        e.step_base = (int)*steps;
        *steps += e.dmaj + 1;
        e.y0 = std::min(e.ya, e.yb); e.y1 = std::max(e.ya, e.yb);
        e.poly = poly; e.pad = 0;
        if (e.ya != e.yb) {
            const long long xa = (long long)e.xa << 16, xb = (long long)e.xb << 16;
            e.dx_fix = (xb - xa) / (long long)(e.yb - e.ya);
            e.x0_fix = e.ya < e.yb ? xa : xb;
            *ymin = std::min(*ymin, e.y0); *ymax = std::max(*ymax, e.y1);
        } else {
            e.dx_fix = 0; e.x0_fix = 0;
        }
    }
}

// ---- batched ordered paint: many polygons, later ones win ------------------------------------------------------
struct PaintItem { // one (polygon, scanline) pair
    int edge_begin, edge_end; // the polygon's edges
    int y, order;             // scanline, 1-based paint order
};

__global__ void __launch_bounds__(256) k_paint_outline(const PolyEdge *__restrict__ edges, int nedges, int total_steps,
                                                       int *__restrict__ owner, int h, int w)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= total_steps) return;
    int lo = 0, hi = nedges - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (edges[mid].step_base <= t) lo = mid; else hi = mid - 1;
    }
    const PolyEdge e = edges[lo];
    const int k = t - e.step_base;
    const int m = vkc::bres_minor(k, e.dmaj, e.dmin);
    const int x = e.ymajor ? e.lx + m : e.lx + k;
    const int y = e.ymajor ? e.ly + e.sy * k : e.ly + e.sy * m;
    // (several label plane sets painted by one call live in bands of h rows of one raster: e.pad is the band, the polygon is clipped to it)
    if ((unsigned)x < (unsigned)w && (unsigned)(y - e.pad * h) < (unsigned)h) atomicMax(&owner[(size_t)y * w + x], e.poly);
}

constexpr int kPaintCross = 64; // crossings of one polygon on one scanline handled by the batched path

// One wave per (polygon, scanline): lanes test the polygon's edges, crossings are ranked by counting (no sort
// loop: rank = number of crossings that precede it, ties by edge index), spans are painted lane-parallel.
__global__ void __launch_bounds__(256) k_paint_spans(const PolyEdge *__restrict__ edges,
                                                     const PaintItem *__restrict__ items, int n_items,
                                                     int *__restrict__ owner, int h, int w, int *__restrict__ overflow)
{
    __shared__ long long xs_all[4][kPaintCross];
    __shared__ long long sorted_all[4][kPaintCross];
    __shared__ int count_all[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int it = blockIdx.x * 4 + wave;
    long long *xs = xs_all[wave], *sorted = sorted_all[wave];
    if (lane == 0) count_all[wave] = 0;
    __syncthreads();
    const bool live = it < n_items;
    PaintItem item = {0, 0, 0, 0};
    if (live) item = items[it];
    for (int i = item.edge_begin + lane; i < item.edge_end; i += 64) {
        const PolyEdge &e = edges[i];
        if (e.y0 != e.y1 && e.y0 <= item.y && item.y < e.y1) {
            const int slot = atomicAdd(&count_all[wave], 1);
            // tie-break on the edge index keeps the ranking a permutation
            if (slot < kPaintCross) xs[slot] = e.x0_fix + (long long)(item.y - e.y0) * e.dx_fix;
        }
    }
    __syncthreads();
    int n = count_all[wave];
    if (n > kPaintCross) {
        if (lane == 0) atomicExch(overflow, 1);
        n = 0;
    }
    if (lane < n) {
        const long long v = xs[lane];
        int rank = 0;
        for (int j = 0; j < n; j++) {
            const long long u = xs[j];
            rank += (u < v) || (u == v && j < lane);
        }
        sorted[rank] = v;
    }
    __syncthreads();
    if (!live || (unsigned)item.y >= (unsigned)h) return;
    int *row = owner + (size_t)item.y * w;
    for (int a = 0; a + 1 < n; a += 2) {
        long long x1 = (sorted[a] + 65535) >> 16, x2 = sorted[a + 1] >> 16;
        if (x1 < 0) x1 = 0;
        if (x2 >= w) x2 = w - 1;
        for (long long x = x1 + lane; x <= x2; x += 64) atomicMax(&row[x], item.order);
    }
}

// FRESH: the output planes are uninitialised memory -- every pixel is written (0 outside every polygon), so the caller needs no
// memset of its own.  Either way the owner word is cleared as it is read: the raster is all zero again when the kernel is done and
// the next call needs no memset either (a page paints four label plane sets: 3 - 4 fill dispatches per call were 14 of its 98).
// blockIdx.z = the band of the raster = the plane set (vkx_paint_poly_sets_fresh_dev: the sets of a page in one call).
constexpr int kPaintSets = 8;
struct PaintOut {
    uint8_t *mask[kPaintSets];
    float *score[kPaintSets];
    ptrdiff_t mask_stride[kPaintSets], score_stride[kPaintSets];
    int value_off[kPaintSets];      // the set's first value in `values`
};

template <bool FRESH>
__global__ void __launch_bounds__(256) k_paint_resolve(int *__restrict__ owner_all, const float *__restrict__ values, PaintOut out, int h, int w)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int set = blockIdx.z;
    if (x >= w || y >= h) return;
    int *owner = owner_all + (size_t)set * h * w;
    uint8_t *mask = out.mask[set];
    float *score = out.score[set];
    const int o = owner[(size_t)y * w + x];
    if (o <= 0) {
        if (o < 0) owner[(size_t)y * w + x] = 0;
        if (FRESH) {
            if (mask) mask[(ptrdiff_t)y * out.mask_stride[set] + x] = 0;
            if (score) score[(ptrdiff_t)y * out.score_stride[set] + x] = 0.0f;
        }
        return;
    }
    owner[(size_t)y * w + x] = 0;
    if (mask) mask[(ptrdiff_t)y * out.mask_stride[set] + x] = 1;
    if (score) score[(ptrdiff_t)y * out.score_stride[set] + x] = values[out.value_off[set] + o - 1];
}

} // namespace

// The ordered paint of up to kPaintSets label plane sets of one shape in ONE call: set k's polygons are shifted into band k (rows
// [k h, (k + 1) h)) of one ownership raster and clipped to it, so one table copy and three kernels serve all of them (a page paints four
// sets -- text-line mask + heights, char mask, seal char mask, char heights: 16 dispatches as four calls, 4 as one).
static int paint_sets_dev(vkx_ctx *ctx, const vkx_paint_set *sets, int n_sets, int h, int w, bool fresh)
{
    VKX_REQUIRE(ctx && sets, "NULL argument");
    VKX_REQUIRE(n_sets >= 1 && n_sets <= kPaintSets, "1 .. 8 plane sets per call");
    VKX_REQUIRE(h > 0 && w > 0 && (long long)h * n_sets < 0x3fffffff, "bad shape");
    long long total_pts = 0, total_polys = 0;
    for (int k = 0; k < n_sets; k++) {
        const vkx_paint_set &S = sets[k];
        VKX_REQUIRE(S.n_polys >= 0 && (S.n_polys == 0 || (S.pts_host && S.poly_offsets_host)), "bad polygon list");
        VKX_REQUIRE(S.mask || S.score, "no output plane");
        VKX_REQUIRE(!S.score || S.values_host || S.n_polys == 0, "score output needs per-polygon values");
        if (S.n_polys) {
            VKX_REQUIRE(S.poly_offsets_host[S.n_polys] >= 0, "bad polygon offsets");
            total_pts += S.poly_offsets_host[S.n_polys];
            total_polys += S.n_polys;
        }
    }
    VKX_REQUIRE(total_pts < 0x3fffffff, "too many vertices");
    if (total_polys == 0 && !fresh) return VKX_OK;
    std::vector<PolyEdge> edges((size_t)total_pts);
    std::vector<PaintItem> items;
    std::vector<float> values((size_t)total_polys);
    std::vector<int32_t> shifted;
    PaintOut out;
    memset(&out, 0, sizeof(out));
    long long steps = 0;
    bool may_overflow = false, any_values = false;
    size_t e_base = 0, v_base = 0;
    for (int k = 0; k < n_sets; k++) {
        const vkx_paint_set &S = sets[k];
        out.mask[k] = S.mask; out.mask_stride[k] = S.mask_stride;
        out.score[k] = S.score; out.score_stride[k] = S.score_stride_el;
        out.value_off[k] = (int)v_base;
        if (S.n_polys == 0) continue;
        const int set_pts = S.poly_offsets_host[S.n_polys];
        const int y_off = k * h;
        const int32_t *pts = S.pts_host;
        if (y_off) {            // the band: the set's vertices moved down by k h rows
            shifted.resize((size_t)set_pts * 2);
            for (int i = 0; i < set_pts; i++) { shifted[2 * (size_t)i] = S.pts_host[2 * (size_t)i]; shifted[2 * (size_t)i + 1] = S.pts_host[2 * (size_t)i + 1] + y_off; }
            pts = shifted.data();
        }
        for (int p = 0; p < S.n_polys; p++) {
            const int b = S.poly_offsets_host[p], e = S.poly_offsets_host[p + 1];
            VKX_REQUIRE(b <= e && e <= set_pts, "bad polygon offsets");
            may_overflow = may_overflow || e - b > kPaintCross;
            if (b == e) continue;
            int ymin = INT_MAX, ymax = INT_MIN;
            build_edges(pts + 2 * (size_t)b, e - b, p + 1, edges.data() + e_base + b, &steps, &ymin, &ymax);
            VKX_REQUIRE(steps < 0x7fffffff, "polygon outlines too long");
            for (size_t i = e_base + b; i < e_base + e; i++) edges[i].pad = k;
            for (int y = std::max(ymin, y_off); y < std::min(ymax, y_off + h); y++)
                items.push_back(PaintItem{(int)e_base + b, (int)e_base + e, y, p + 1});
        }
        if (S.values_host) { memcpy(values.data() + v_base, S.values_host, sizeof(float) * (size_t)S.n_polys); any_values = true; }
        e_base += (size_t)set_pts;
        v_base += (size_t)S.n_polys;
    }
    const size_t ebytes = (sizeof(PolyEdge) * edges.size() + 255) & ~(size_t)255;
    const size_t ibytes = (sizeof(PaintItem) * items.size() + 255) & ~(size_t)255;
    const size_t vbytes = (sizeof(float) * values.size() + 255) & ~(size_t)255;
    int rc = vkx_scratch_reserve(ctx, &ctx->misc, 256 + ebytes + ibytes + vbytes);
    if (rc) return rc;
    // the ownership raster is the paint's own block: zero at rest (k_paint_resolve clears what it reads), memset only when it grows
    const size_t owner_bytes = (size_t)h * w * 4 * (size_t)n_sets;
    if (owner_bytes > ctx->paint_owner.cap) ctx->paint_owner_zeroed = 0;
    rc = vkx_scratch_reserve(ctx, &ctx->paint_owner, owner_bytes);
    if (rc) return rc;
    unsigned char *misc = (unsigned char *)ctx->misc.ptr;
    int *overflow = (int *)misc;
    PolyEdge *d_edges = (PolyEdge *)(misc + 256);
    PaintItem *d_items = (PaintItem *)(misc + 256 + ebytes);
    float *d_values = (float *)(misc + 256 + ebytes + ibytes);
    int *owner = (int *)ctx->paint_owner.ptr;
    // The edge / item / value tables travel through the context's page-locked ring as ONE asynchronous copy: the call returns
    // with its kernels queued (a page paints four label planes: two stream synchronisations and a flag read-back per call were
    // 0.45 ms of a 2.2 ms page).  A polygon of at most kPaintCross vertices cannot cross a scanline more often than the span
    // kernel holds, so only calls with larger polygons read the overflow flag back (and synchronise for it).
    const size_t table_bytes = ebytes + ibytes + vbytes;
    vkx_device_guard guard(ctx);
    if (table_bytes) {
        void *ring = nullptr;
        if ((rc = vkx_desc_ring_take(ctx, table_bytes, &ring))) return rc;
        if (!edges.empty()) memcpy(ring, edges.data(), sizeof(PolyEdge) * edges.size());
        if (!items.empty()) memcpy((unsigned char *)ring + ebytes, items.data(), sizeof(PaintItem) * items.size());
        if (!values.empty()) memcpy((unsigned char *)ring + ebytes + ibytes, values.data(), sizeof(float) * values.size());
        VKX_HIP(hipMemcpyAsync(d_edges, ring, table_bytes, hipMemcpyHostToDevice, ctx->stream));
    }
    if (may_overflow) VKX_HIP(hipMemsetAsync(overflow, 0, sizeof(int), ctx->stream));      // (only such calls can set it, and only they read it)
    if (ctx->paint_owner_zeroed < owner_bytes) VKX_HIP(hipMemsetAsync(owner, 0, ctx->paint_owner.cap, ctx->stream));
    ctx->paint_owner_zeroed = 0;          // dirty until the resolve kernel of THIS call has been queued (an error exit in between re-zeroes next time)
    if (steps > 0) {
        { VKX_TIMED(ctx, "k_paint_outline"); k_paint_outline<<<vkx_blocks((size_t)steps, 256), 256, 0, ctx->stream>>>(d_edges, (int)total_pts, (int)steps, owner, h, w); }
        VKX_LAUNCH_CHECK();
    }
    if (!items.empty()) {
        { VKX_TIMED(ctx, "k_paint_spans"); k_paint_spans<<<vkx_blocks(items.size(), 4), 256, 0, ctx->stream>>>(d_edges, d_items, (int)items.size(), owner, h * n_sets, w, overflow); }
        VKX_LAUNCH_CHECK();
    }
    {
        dim3 grid(vkx_blocks(w, 64), vkx_blocks(h, 4), n_sets);
        VKX_TIMED(ctx, "k_paint_resolve");
        if (fresh) k_paint_resolve<true><<<grid, 256, 0, ctx->stream>>>(owner, any_values ? d_values : nullptr, out, h, w);
        else k_paint_resolve<false><<<grid, 256, 0, ctx->stream>>>(owner, any_values ? d_values : nullptr, out, h, w);
        VKX_LAUNCH_CHECK();
        ctx->paint_owner_zeroed = ctx->paint_owner.cap;
    }
    if (!may_overflow) return VKX_OK;
    int flag = 0;
    VKX_HIP(hipMemcpyAsync(&flag, overflow, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    VKX_HIP(hipStreamSynchronize(ctx->stream));
    if (flag) {
        vkx_set_error("a polygon has more than %d edge crossings on one scanline", kPaintCross);
        return VKX_ERR_UNSUPPORTED;
    }
    return VKX_OK;
}

static int paint_polys_dev(vkx_ctx *ctx, const int32_t *pts_host, const int32_t *poly_offsets_host, int n_polys,
                           const float *values_host, uint8_t *mask, ptrdiff_t mask_stride, float *score,
                           ptrdiff_t score_stride_el, int h, int w, bool fresh)
{
    VKX_REQUIRE(ctx && (n_polys == 0 || (pts_host && poly_offsets_host)), "NULL argument");
    VKX_REQUIRE(n_polys >= 0 && h > 0 && w > 0, "bad shape");
    vkx_paint_set S;
    S.pts_host = pts_host; S.poly_offsets_host = poly_offsets_host; S.n_polys = n_polys; S.values_host = values_host;
    S.mask = mask; S.mask_stride = mask_stride; S.score = score; S.score_stride_el = score_stride_el;
    return paint_sets_dev(ctx, &S, 1, h, w, fresh);
}

VKX_EXPORT int vkx_paint_poly_sets_fresh_dev(vkx_ctx *ctx, const vkx_paint_set *sets, int n_sets, int h, int w)
{
    return paint_sets_dev(ctx, sets, n_sets, h, w, true);
}

VKX_EXPORT int vkx_paint_polys_dev(vkx_ctx *ctx, const int32_t *pts_host, const int32_t *poly_offsets_host, int n_polys,
                                   const float *values_host, uint8_t *mask, ptrdiff_t mask_stride, float *score,
                                   ptrdiff_t score_stride_el, int h, int w)
{
    return paint_polys_dev(ctx, pts_host, poly_offsets_host, n_polys, values_host, mask, mask_stride, score, score_stride_el, h, w, false);
}

VKX_EXPORT int vkx_paint_polys_fresh_dev(vkx_ctx *ctx, const int32_t *pts_host, const int32_t *poly_offsets_host, int n_polys,
                                         const float *values_host, uint8_t *mask, ptrdiff_t mask_stride, float *score,
                                         ptrdiff_t score_stride_el, int h, int w)
{
    return paint_polys_dev(ctx, pts_host, poly_offsets_host, n_polys, values_host, mask, mask_stride, score, score_stride_el, h, w, true);
}

// Host planes: mask / score are updated in place (read, painted, written back).
VKX_EXPORT int vkx_paint_polys(vkx_ctx *ctx, const int32_t *pts_host, const int32_t *poly_offsets_host, int n_polys,
                               const float *values_host, uint8_t *mask, ptrdiff_t mask_stride, float *score,
                               ptrdiff_t score_stride_el, int h, int w)
{
    VKX_REQUIRE(ctx && (mask || score), "NULL argument");
    VKX_REQUIRE(h > 0 && w > 0, "bad shape");
    const size_t mbytes = mask ? (((size_t)h * w + 255) & ~(size_t)255) : 0;
    const size_t sbytes = score ? (size_t)h * w * 4 : 0;
    int rc = vkx_scratch_reserve(ctx, &ctx->stage[1], mbytes + sbytes);
    if (rc) return rc;
    uint8_t *d_mask = mask ? (uint8_t *)ctx->stage[1].ptr : nullptr;
    float *d_score = score ? (float *)((unsigned char *)ctx->stage[1].ptr + mbytes) : nullptr;
    if (mask)
        VKX_HIP(vkx_copy_plane(d_mask, (size_t)w, mask, (size_t)mask_stride, (size_t)w, (size_t)h, hipMemcpyHostToDevice, ctx->stream));
    if (score)
        VKX_HIP(vkx_copy_plane(d_score, (size_t)w * 4, score, (size_t)score_stride_el * 4, (size_t)w * 4, (size_t)h, hipMemcpyHostToDevice, ctx->stream));
    rc = vkx_paint_polys_dev(ctx, pts_host, poly_offsets_host, n_polys, values_host, d_mask, w, d_score, w, h, w);
    if (rc) return rc;
    if (mask)
        VKX_HIP(vkx_copy_plane(mask, (size_t)mask_stride, d_mask, (size_t)w, (size_t)w, (size_t)h, hipMemcpyDeviceToHost, ctx->stream));
    if (score)
        VKX_HIP(vkx_copy_plane(score, (size_t)score_stride_el * 4, d_score, (size_t)w * 4, (size_t)w * 4, (size_t)h, hipMemcpyDeviceToHost, ctx->stream));
    VKX_HIP(hipStreamSynchronize(ctx->stream));
    return VKX_OK;
}

VKX_EXPORT int vkx_fill_poly_mask_u8_dev(vkx_ctx *ctx, const int32_t *pts_host, int npts, uint8_t *mask, int h, int w,
                                         ptrdiff_t stride)
{
    VKX_REQUIRE(ctx && pts_host && mask, "NULL argument");
    VKX_REQUIRE(npts > 0 && h > 0 && w > 0, "bad shape");
    std::vector<PolyEdge> edges((size_t)npts);
    long long steps = 0;
    int ymin = INT_MAX, ymax = INT_MIN;
    for (int i = 0; i < npts; i++) {
        const int xb = pts_host[2 * i], yb = pts_host[2 * i + 1];
        VKX_REQUIRE(xb >= 0 && xb < w && yb >= 0 && yb < h, "polygon vertex outside the mask");
    }
    build_edges(pts_host, npts, 0, edges.data(), &steps, &ymin, &ymax);
    VKX_REQUIRE(steps < 0x7fffffff, "polygon outline too long");
    const size_t ebytes = sizeof(PolyEdge) * edges.size();
    int rc = vkx_scratch_reserve(ctx, &ctx->misc, ebytes + 256);
    if (rc) return rc;
    unsigned char *misc = (unsigned char *)ctx->misc.ptr;
    int *overflow = (int *)misc;
    PolyEdge *d_edges = (PolyEdge *)(misc + 256);
    VKX_HIP(hipMemsetAsync(overflow, 0, sizeof(int), ctx->stream));
    VKX_HIP(hipMemcpyAsync(d_edges, edges.data(), ebytes, hipMemcpyHostToDevice, ctx->stream));
    VKX_HIP(hipStreamSynchronize(ctx->stream)); // `edges` lives on this frame
    { VKX_TIMED(ctx, "k_poly_outline"); k_poly_outline<<<vkx_blocks((size_t)steps, 256), 256, 0, ctx->stream>>>(d_edges, npts, (int)steps, mask, h, w, stride); }
    VKX_LAUNCH_CHECK();
    if (ymin < ymax) {
        { VKX_TIMED(ctx, "k_poly_spans"); k_poly_spans<<<ymax - ymin, 256, 0, ctx->stream>>>(d_edges, npts, ymin, mask, h, w, stride, overflow); }
        VKX_LAUNCH_CHECK();
        int flag = 0;
        VKX_HIP(hipMemcpyAsync(&flag, overflow, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        VKX_HIP(hipStreamSynchronize(ctx->stream));
        if (flag) {
            vkx_set_error("polygon has more than %d edge crossings on one scanline", kMaxCross);
            return VKX_ERR_UNSUPPORTED;
        }
    }
    return VKX_OK;
}

// Host-memory variant: zero-initialised mask of shape [h, w] with the polygon set to 1.
VKX_EXPORT int vkx_fill_poly_mask_u8(vkx_ctx *ctx, const int32_t *pts_host, int npts, uint8_t *mask, int h, int w,
                                     ptrdiff_t stride)
{
    VKX_REQUIRE(ctx && pts_host && mask, "NULL argument");
    VKX_REQUIRE(npts > 0 && h > 0 && w > 0, "bad shape");
    const size_t bytes = (size_t)h * w;
    int rc = vkx_scratch_reserve(ctx, &ctx->stage[1], bytes);
    if (rc) return rc;
    uint8_t *d = (uint8_t *)ctx->stage[1].ptr;
    VKX_HIP(hipMemsetAsync(d, 0, bytes, ctx->stream));
    rc = vkx_fill_poly_mask_u8_dev(ctx, pts_host, npts, d, h, w, w);
    if (rc) return rc;
    VKX_HIP(vkx_copy_plane(mask, (size_t)stride, d, (size_t)w, (size_t)w, (size_t)h, hipMemcpyDeviceToHost, ctx->stream));
    VKX_HIP(hipStreamSynchronize(ctx->stream));
    return VKX_OK;
}
