// similarity_mls lattice construction on the device: SimilarityMlsPointProjector.project_point
// (mechanism/distortion/geometric/mls.py:38-135) for every vertex of the source lattice in one launch, one lane per
// vertex.  The reference evaluates Schaefer's similarity MLS (section 2.2) with float32 numpy arrays; everything
// downstream only sees the ROUNDED vertex, so the float32 roundings -- including the accumulation orders of numpy's
// reductions and of the OpenBLAS kernels behind np.matmul -- are part of the result:
//   np.sum over a contiguous axis       numpy's pairwise sum: n < 8 sequential; otherwise 8 running sums over whole
//                                       blocks of 8, ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the tail sequentially
//   np.sum(.., axis=0) of an (N, 2)     row after row, sequential
//   (N,) @ (N, 2)                       sgemv on a 2 x N matrix.  5 <= N <= 48 (OpenBLAS's SkylakeX small-matrix kernel on the
//                                       AVX-512 host the goldens come from): acc = fma(w_i, p_i, acc) in handle order;
//                                       other N (the generic two-row tail of sgemv_n_4.c): handles four at a time as
//                                       acc += fma(w0, p0, w1 p1); acc += fma(w2, p2, w3 p3), the rest one fma each
//   (N, 2) @ (2, 2), (N,1,2) @ (N,2,2)  fma(a1, b1, a0 b0)
// (established against numpy 2.2.6 + OpenBLAS 0.3.29 with the generator of tests/golden/make_golden.py; the lattices of
// tests/golden/mls_states.npz pin it).  Compiled with -ffp-contract=off: fused operations appear only as fmaf().
#include "vkx_internal.h"

#include <string.h>
#include <algorithm>
#include <vector>

namespace {

constexpr int kPairwiseBlock = 128;  // numpy's pairwise sum halves its range recursively beyond this many elements
constexpr int kLdsHandles = 2048;    // handle tables up to this size are staged in LDS (16 bytes per handle)

struct Handles {
    const float *p, *q;        // [n, 2] (x, y): integer handle positions as float32 (PointTuple.to_smooth_np_array)
    const double *ps, *qs;     // [n, 2] smooth positions: a vertex exactly on a source handle maps to its target
    int n;
};

// w_i = 1 / |p_i - v|^2 in float32
__device__ __forceinline__ float weight(const float *p, int i, float vx, float vy, bool &zero)
{
    const float dx = p[2 * i] - vx, dy = p[2 * i + 1] - vy;
    const float d2 = dx * dx + dy * dy;
    if (d2 == 0.f) zero = true;
    return 1.f / d2;
}

template <typename F>
__device__ __forceinline__ float pairwise_block(int lo, int n, F term)     // n <= kPairwiseBlock
{
    if (n < 8) {
        float res = term(lo);
        for (int i = 1; i < n; i++) res = res + term(lo + i);
        return res;
    }
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; j++) r[j] = term(lo + j);
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
#pragma unroll
        for (int j = 0; j < 8; j++) r[j] = r[j] + term(lo + i + j);
    }
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res = res + term(lo + i);
    return res;
}

template <typename F>
__device__ __forceinline__ float pairwise_sum(int n, F term)
{
    if (n <= kPairwiseBlock) return pairwise_block(0, n, term);
    // sum(a, n) = sum(a, n2) + sum(a + n2, n - n2) with n2 = n / 2 rounded down to a multiple of 8, as an explicit stack
    // (depth 12 reaches 128 * 2^11 handles)
    int lo_s[12], n_s[12], stage[12];
    float left[12], res = 0.f;
    int sp = 1;
    lo_s[0] = 0; n_s[0] = n; stage[0] = 0;
    while (sp > 0) {
        const int k = sp - 1;
        int n2 = n_s[k] / 2;
        n2 -= n2 % 8;
        if (stage[k] == 0) {
            if (n_s[k] <= kPairwiseBlock || sp == 12) {
                res = pairwise_block(lo_s[k], n_s[k], term);
                sp--;
            } else {
                stage[k] = 1;
                lo_s[sp] = lo_s[k]; n_s[sp] = n2; stage[sp] = 0;
                sp++;
            }
        } else if (stage[k] == 1) {
            left[k] = res;
            stage[k] = 2;
            lo_s[sp] = lo_s[k] + n2; n_s[sp] = n_s[k] - n2; stage[sp] = 0;
            sp++;
        } else {
            res = left[k] + res;
            sp--;
        }
    }
    return res;
}

// One vertex through SimilarityMlsPointProjector.project_point; false where the reference's np.errstate(divide='raise') fires.
// sp / sq: the float32 handle tables (LDS or global); hd.ps / hd.qs: the smooth positions (exact handle hits).
__device__ __forceinline__ bool mls_vertex(const Handles &hd, const float *sp, const float *sq, double vxd, double vyd, double &ox, double &oy)
{
    const int n = hd.n;
    // identity on exact handle hits (mls.py:57-61); a later duplicate handle overrides an earlier one like the dict
    int hit = -1;
    for (int i = 0; i < n; i++)
        if (hd.ps[2 * i] == vxd && hd.ps[2 * i + 1] == vyd) hit = i;
    if (hit >= 0) {
        ox = hd.qs[2 * hit];
        oy = hd.qs[2 * hit + 1];
        return true;
    }
    const float vx = (float)vxd, vy = (float)vyd;   // python float operands enter float32 arithmetic as float32
    bool zero = false;
    const float sw = pairwise_sum(n, [&](int i) { return weight(sp, i, vx, vy, zero); });
    if (zero) {                                     // np.errstate(divide='raise')
        ox = 0.0; oy = 0.0;
        return false;
    }
    // weighted centroids p*, q*
    float psx, psy, qsx, qsy;
    auto wn = [&](int i) { return weight(sp, i, vx, vy, zero) / sw; };
    psx = psy = qsx = qsy = 0.f;
    {
        int i = 0;
        if (n < 5 || n > 48) {
            for (; i + 4 <= n; i += 4) {
#pragma unroll
                for (int h = 0; h < 4; h += 2) {
                    const float wa = wn(i + h), wb = wn(i + h + 1);
                    const float *pa = sp + 2 * (i + h), *qa = sq + 2 * (i + h);
                    psx = psx + fmaf(wa, pa[0], wb * pa[2]); psy = psy + fmaf(wa, pa[1], wb * pa[3]);
                    qsx = qsx + fmaf(wa, qa[0], wb * qa[2]); qsy = qsy + fmaf(wa, qa[1], wb * qa[3]);
                }
            }
        }
        for (; i < n; i++) {
            const float w = wn(i);
            psx = fmaf(w, sp[2 * i], psx); psy = fmaf(w, sp[2 * i + 1], psy);
            qsx = fmaf(w, sq[2 * i], qsx); qsy = fmaf(w, sq[2 * i + 1], qsy);
        }
    }
    // anchor = [[ax, ay], [ay, -ax]] with (ax, ay) = v - p*
    const float ax = vx - psx, ay = vy - psy;
    // mu_s = sum_i w_i |p_hat_i|^2
    const float mu = pairwise_sum(n, [&](int i) {
        const float hx = sp[2 * i] - psx, hy = sp[2 * i + 1] - psy;
        return weight(sp, i, vx, vy, zero) * (hx * hx + hy * hy);
    });
    // sum_i q_hat_i A_i, A_i = w_i [p_hat_i ; -p_hat_i^perp] anchor
    float sx = 0.f, sy = 0.f;
    for (int i = 0; i < n; i++) {
        const float w = weight(sp, i, vx, vy, zero);
        const float hx = sp[2 * i] - psx, hy = sp[2 * i + 1] - psy;
        const float gx = sq[2 * i] - qsx, gy = sq[2 * i + 1] - qsy;
        const float t0 = fmaf(hy, ay, hx * ax), t1 = fmaf(hy, -ax, hx * ay);     // p_hat . anchor
        const float b0 = fmaf(-hx, ay, hy * ax), b1 = fmaf(-hx, -ax, hy * ay);   // (-p_hat^perp) . anchor
        const float a00 = w * t0, a01 = w * t1, a10 = w * b0, a11 = w * b1;
        const float e0 = fmaf(gy, a10, gx * a00), e1 = fmaf(gy, a11, gx * a01);
        if (i == 0) { sx = e0; sy = e1; }
        else { sx = sx + e0; sy = sy + e1; }
    }
    ox = (double)(sx / mu + qsx);
    oy = (double)(sy / mu + qsy);
    return true;
}


__device__ __forceinline__ void stage_handles(const Handles &hd, float *staged, const float *&sp, const float *&sq)
{
    sp = hd.p; sq = hd.q;
    if (hd.n <= kLdsHandles) {
        for (int i = threadIdx.x; i < 2 * hd.n; i += blockDim.x) { staged[i] = hd.p[i]; staged[2 * hd.n + i] = hd.q[i]; }
        __syncthreads();
        sp = staged;
        sq = staged + 2 * hd.n;
    }
}

__global__ void __launch_bounds__(64) k_mls_project(Handles hd, const double *__restrict__ vertices, int n_vertices,
                                                    double *__restrict__ out, int *__restrict__ bad)
{
    extern __shared__ float staged[];              // [2 n] source handles, [2 n] targets when n <= kLdsHandles
    const float *sp, *sq;
    stage_handles(hd, staged, sp, sq);
    const int v = blockIdx.x * 64 + threadIdx.x;
    if (v >= n_vertices) return;
    double ox, oy;
    if (!mls_vertex(hd, sp, sq, vertices[2 * v], vertices[2 * v + 1], ox, oy)) atomicMax(bad, v + 1);
    out[2 * v] = ox;
    out[2 * v + 1] = oy;
}

// ---- a batch of similarity_mls STATES (SimilarityMlsState.__init__, geometric/mls.py:140-157 in the reference): the source lattice of
// every state projected through its own handle set (grid = vertex blocks x states), then the tail of
// create_dst_image_grid_and_shift_amounts_and_resize_ratios (grid_rendering/grid_creator.py:44-115, resize_as_src = False) per state:
// shift by the minimum of the ROUNDED positions, round half to even, extent = result shape.
struct MlsStateDev {
    Handles hd;
    int32_t *sv, *dv;          // [rows, cols, 2] (x, y)
    double *smooth;            // [rows * cols, 2] scratch: the projected positions
    int height, width, grid_size, rows, cols, pad;
};

__device__ __forceinline__ void lattice_xy(const MlsStateDev &st, int i, int &x, int &y)
{
    const int r = i / st.cols, c = i - r * st.cols;
    x = c + 1 < st.cols ? c * st.grid_size : st.width - 1;
    y = r + 1 < st.rows ? r * st.grid_size : st.height - 1;
}

__global__ void __launch_bounds__(64) k_mls_states_project(const MlsStateDev *__restrict__ states, unsigned *__restrict__ flags)
{
    extern __shared__ float staged[];
    const MlsStateDev &st = states[blockIdx.y];
    const int N = st.rows * st.cols;
    if ((int)blockIdx.x * 64 >= N) return;         // (uniform: the grid is sized for the largest lattice of the batch)
    const float *sp, *sq;
    stage_handles(st.hd, staged, sp, sq);
    const int v = blockIdx.x * 64 + threadIdx.x;
    if (v >= N) return;
    int xi, yi;
    lattice_xy(st, v, xi, yi);
    double ox, oy;
    if (!mls_vertex(st.hd, sp, sq, (double)xi, (double)yi, ox, oy)) atomicOr(&flags[blockIdx.y], VKX_GRID_STATE_DIVIDE);
    st.smooth[2 * v] = ox;
    st.smooth[2 * v + 1] = oy;
}

__global__ void __launch_bounds__(256) k_mls_states_tail(const MlsStateDev *__restrict__ states, const unsigned *__restrict__ flags,
                                                         vkx_grid_state *__restrict__ out)
{
    __shared__ double red[4][4];
    __shared__ unsigned bad;
    const MlsStateDev &st = states[blockIdx.x];
    const int N = st.rows * st.cols, tid = threadIdx.x;
    if (tid == 0) bad = flags[blockIdx.x];
    __syncthreads();
    double mnx = INFINITY, mny = INFINITY;
    unsigned f = 0;
    for (int i = tid; i < N; i += 256) {
        const double px = st.smooth[2 * i], py = st.smooth[2 * i + 1];
        if (isnan(px) || isnan(py)) f |= VKX_GRID_STATE_NAN;
        else if (isinf(px) || isinf(py)) f |= VKX_GRID_STATE_INF;
        mnx = fmin(mnx, rint(px));
        mny = fmin(mny, rint(py));
    }
    for (int d = 32; d >= 1; d >>= 1) { mnx = fmin(mnx, __shfl_xor(mnx, d)); mny = fmin(mny, __shfl_xor(mny, d)); }
    if (f) atomicOr(&bad, f);
    if ((tid & 63) == 0) { red[tid >> 6][0] = mnx; red[tid >> 6][1] = mny; }
    __syncthreads();
    mnx = fmin(fmin(red[0][0], red[1][0]), fmin(red[2][0], red[3][0]));
    mny = fmin(fmin(red[0][1], red[1][1]), fmin(red[2][1], red[3][1]));
    const bool ok = bad == 0;
    __syncthreads();
    const double sx = ok ? -mnx : 0., sy = ok ? -mny : 0.;
    double mxx = -INFINITY, mxy = -INFINITY;
    for (int i = tid; i < N; i += 256) {
        int xi, yi;
        lattice_xy(st, i, xi, yi);
        st.sv[2 * i] = xi; st.sv[2 * i + 1] = yi;
        const double qx = rint(st.smooth[2 * i] + sx), qy = rint(st.smooth[2 * i + 1] + sy);
        mxx = fmax(mxx, qx); mxy = fmax(mxy, qy);
        if (ok) { st.dv[2 * i] = (int)qx; st.dv[2 * i + 1] = (int)qy; }
    }
    for (int d = 32; d >= 1; d >>= 1) { mxx = fmax(mxx, __shfl_xor(mxx, d)); mxy = fmax(mxy, __shfl_xor(mxy, d)); }
    if ((tid & 63) == 0) { red[tid >> 6][2] = mxx; red[tid >> 6][3] = mxy; }
    __syncthreads();
    if (tid == 0) {
        mxx = fmax(fmax(red[0][2], red[1][2]), fmax(red[2][2], red[3][2]));
        mxy = fmax(fmax(red[0][3], red[1][3]), fmax(red[2][3], red[3][3]));
        vkx_grid_state g;
        g.rows = st.rows; g.cols = st.cols;
        const bool fits = ok && mxx < 2147483647. && mxy < 2147483647. && fabs(mnx) < 2147483647. && fabs(mny) < 2147483647.;
        g.dh = fits ? (int)mxy + 1 : 0; g.dw = fits ? (int)mxx + 1 : 0;
        g.shift_y = fits ? (int)mny : 0; g.shift_x = fits ? (int)mnx : 0;
        g.flags = bad | (ok && !fits ? VKX_GRID_STATE_RANGE : 0u);
        g.reserved = 0;
        out[blockIdx.x] = g;
    }
}

} // namespace

VKX_EXPORT int vkx_mls_project_dev(vkx_ctx *ctx, const float *src_handles, const float *dst_handles,
                                   const double *src_handles_smooth, const double *dst_handles_smooth, int n_handles,
                                   const double *vertices_xy, int n_vertices, double *out_xy, int32_t *status)
{
    VKX_REQUIRE(ctx && src_handles && dst_handles && src_handles_smooth && dst_handles_smooth, "NULL argument");
    VKX_REQUIRE(n_handles >= 1 && n_vertices >= 0, "bad sizes");
    if (n_vertices == 0) return VKX_OK;
    VKX_REQUIRE(vertices_xy && out_xy && status, "NULL argument");
    vkx_device_guard guard(ctx);
    Handles hd{src_handles, dst_handles, src_handles_smooth, dst_handles_smooth, n_handles};
    { VKX_TIMED(ctx, "k_mls_project"); k_mls_project<<<vkx_blocks((size_t)n_vertices, 64), 64, n_handles <= kLdsHandles ? sizeof(float) * 4 * (size_t)n_handles : 0, ctx->stream>>>(hd, vertices_xy, n_vertices, out_xy, status); }
    VKX_LAUNCH_CHECK();
    return VKX_OK;
}

VKX_EXPORT int vkx_mls_project(vkx_ctx *ctx, const float *src_handles, const float *dst_handles,
                               const double *src_handles_smooth, const double *dst_handles_smooth, int n_handles,
                               const double *vertices_xy, int n_vertices, double *out_xy)
{
    VKX_REQUIRE(ctx && src_handles && dst_handles && src_handles_smooth && dst_handles_smooth, "NULL argument");
    VKX_REQUIRE(n_handles >= 1 && n_vertices >= 0, "bad sizes");
    if (n_vertices == 0) return VKX_OK;
    VKX_REQUIRE(vertices_xy && out_xy, "NULL argument");
    vkx_device_guard guard(ctx);
    const size_t hf = (sizeof(float) * 2 * (size_t)n_handles + 255) & ~(size_t)255;
    const size_t hdb = (sizeof(double) * 2 * (size_t)n_handles + 255) & ~(size_t)255;
    const size_t vb = (sizeof(double) * 2 * (size_t)n_vertices + 255) & ~(size_t)255;
    const size_t off_q = hf, off_ps = 2 * hf, off_qs = off_ps + hdb, off_v = off_qs + hdb, off_out = off_v + vb,
                 off_bad = off_out + vb;
    int rc = vkx_scratch_reserve(ctx, &ctx->stage[0], off_bad + 256);
    if (rc) return rc;
    uint8_t *base = (uint8_t *)ctx->stage[0].ptr;
    VKX_HIP(hipMemcpyAsync(base, src_handles, sizeof(float) * 2 * n_handles, hipMemcpyHostToDevice, ctx->stream));
    VKX_HIP(hipMemcpyAsync(base + off_q, dst_handles, sizeof(float) * 2 * n_handles, hipMemcpyHostToDevice, ctx->stream));
    VKX_HIP(hipMemcpyAsync(base + off_ps, src_handles_smooth, sizeof(double) * 2 * n_handles, hipMemcpyHostToDevice, ctx->stream));
    VKX_HIP(hipMemcpyAsync(base + off_qs, dst_handles_smooth, sizeof(double) * 2 * n_handles, hipMemcpyHostToDevice, ctx->stream));
    VKX_HIP(hipMemcpyAsync(base + off_v, vertices_xy, sizeof(double) * 2 * (size_t)n_vertices, hipMemcpyHostToDevice, ctx->stream));
    VKX_HIP(hipMemsetAsync(base + off_bad, 0, sizeof(int), ctx->stream));
    rc = vkx_mls_project_dev(ctx, (const float *)base, (const float *)(base + off_q), (const double *)(base + off_ps),
                             (const double *)(base + off_qs), n_handles, (const double *)(base + off_v), n_vertices,
                             (double *)(base + off_out), (int32_t *)(base + off_bad));
    if (rc) return rc;
    int bad = 0;
    VKX_HIP(hipMemcpyAsync(out_xy, base + off_out, sizeof(double) * 2 * (size_t)n_vertices, hipMemcpyDeviceToHost, ctx->stream));
    VKX_HIP(hipMemcpyAsync(&bad, base + off_bad, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    VKX_HIP(hipStreamSynchronize(ctx->stream));
    if (bad) {
        vkx_set_error("vkx_mls_project: vertex %d coincides with an integer handle position (divide by zero)", bad - 1);
        return VKX_ERR_DIVIDE;
    }
    return VKX_OK;
}

static int mls_axis_ticks(int length, int grid_size) { const int n = (length + grid_size - 1) / grid_size; return ((n - 1) * grid_size != length - 1) ? n + 1 : n; }

// The lattices of n similarity_mls states (resize_as_src = False, the operator's default): handle tables through the page-locked
// ring, one projection launch over (vertex blocks, states), one tail launch, on the stream `stream` (see vkx_camera_states_dev).
VKX_EXPORT int vkx_mls_states_dev(vkx_ctx *ctx, const vkx_mls_config *configs, int n, int32_t *const *src_vertices,
                                  int32_t *const *dst_vertices, vkx_grid_state *states_host, int stream)
{
    VKX_REQUIRE(ctx && configs && src_vertices && dst_vertices && states_host, "NULL argument");
    VKX_REQUIRE(n >= 1 && n <= 65535, "1 .. 65535 states per call");
    int rc;
    hipStream_t st = vkx_stream_by_id(ctx, stream, &rc);
    if (rc) return rc;
    std::vector<MlsStateDev> host((size_t)n);
    size_t table_bytes = 0, smooth_doubles = 0;
    int max_vertices = 0, max_handles = 0;
    for (int i = 0; i < n; i++) {
        const vkx_mls_config &c = configs[i];
        VKX_REQUIRE(c.height >= 1 && c.width >= 1 && c.grid_size >= 1 && c.n_handles >= 1, "bad shape / grid size / handle count");
        VKX_REQUIRE(c.src_handles && c.dst_handles && c.src_handles_smooth && c.dst_handles_smooth, "NULL handle table");
        VKX_REQUIRE(src_vertices[i] && dst_vertices[i], "NULL lattice");
        MlsStateDev &s = host[i];
        s.rows = mls_axis_ticks(c.height, c.grid_size); s.cols = mls_axis_ticks(c.width, c.grid_size);
        VKX_REQUIRE((long long)s.rows * s.cols <= (1 << 24), "lattice beyond 2^24 vertices");
        s.height = c.height; s.width = c.width; s.grid_size = c.grid_size; s.pad = 0;
        s.sv = src_vertices[i]; s.dv = dst_vertices[i];
        s.hd.n = c.n_handles;
        table_bytes += ((size_t)c.n_handles * 48 + 255) & ~(size_t)255;      // 2 x float32 [n, 2] + 2 x float64 [n, 2]
        smooth_doubles += (size_t)s.rows * s.cols * 2;
        max_vertices = std::max(max_vertices, s.rows * s.cols);
        max_handles = std::max(max_handles, c.n_handles);
    }
    const size_t desc_bytes = ((size_t)n * sizeof(MlsStateDev) + 255) & ~(size_t)255, out_bytes = ((size_t)n * sizeof(vkx_grid_state) + 255) & ~(size_t)255;
    const size_t flag_bytes = ((size_t)n * 4 + 255) & ~(size_t)255;
    const size_t upload = desc_bytes + table_bytes;
    if ((rc = vkx_scratch_reserve(ctx, &ctx->mls_work, upload + out_bytes + flag_bytes + smooth_doubles * sizeof(double)))) return rc;
    unsigned char *base = (unsigned char *)ctx->mls_work.ptr;
    void *ring = nullptr;
    if ((rc = vkx_desc_ring_take(ctx, upload, &ring))) return rc;
    unsigned char *r = (unsigned char *)ring;
    size_t off = desc_bytes, soff = 0;
    double *smooth_base = (double *)(base + upload + out_bytes + flag_bytes);
    for (int i = 0; i < n; i++) {
        const vkx_mls_config &c = configs[i];
        MlsStateDev &s = host[i];
        const size_t nf = (size_t)c.n_handles * 2;
        memcpy(r + off, c.src_handles, nf * 4);               s.hd.p = (const float *)(base + off);
        memcpy(r + off + nf * 4, c.dst_handles, nf * 4);      s.hd.q = (const float *)(base + off + nf * 4);
        memcpy(r + off + nf * 8, c.src_handles_smooth, nf * 8);   s.hd.ps = (const double *)(base + off + nf * 8);
        memcpy(r + off + nf * 16, c.dst_handles_smooth, nf * 8);  s.hd.qs = (const double *)(base + off + nf * 16);
        off += ((size_t)c.n_handles * 48 + 255) & ~(size_t)255;
        s.smooth = smooth_base + soff;
        soff += (size_t)s.rows * s.cols * 2;
    }
    memcpy(r, host.data(), (size_t)n * sizeof(MlsStateDev));
    vkx_device_guard guard(ctx);
    hipStream_t main_stream = ctx->stream;
    unsigned *flags = (unsigned *)(base + upload + out_bytes);
    vkx_grid_state *out = (vkx_grid_state *)(base + upload);
    ctx->stream = st;
    hipError_t e = hipMemcpyAsync(base, ring, upload, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemsetAsync(flags, 0, flag_bytes, st);
    if (e == hipSuccess) {
        VKX_TIMED(ctx, "k_mls_states_project");
        const size_t lds = max_handles <= kLdsHandles ? sizeof(float) * 4 * (size_t)max_handles : 0;
        k_mls_states_project<<<dim3(vkx_blocks((size_t)max_vertices, 64), n), 64, lds, st>>>((const MlsStateDev *)base, flags);
        e = hipGetLastError();
    }
    if (e == hipSuccess) {
        VKX_TIMED(ctx, "k_mls_states_tail");
        k_mls_states_tail<<<n, 256, 0, st>>>((const MlsStateDev *)base, flags, out);
        e = hipGetLastError();
    }
    rc = VKX_OK;
    if (e != hipSuccess) { vkx_set_error("%s: %s", __func__, hipGetErrorString(e)); rc = VKX_ERR_HIP; }
    if (!rc) rc = vkx_small_to_host(ctx, states_host, out, (size_t)n * sizeof(vkx_grid_state));
    ctx->stream = main_stream;
    if (rc) return rc;
    if (!ctx->lattices_ready) VKX_HIP(hipEventCreateWithFlags(&ctx->lattices_ready, hipEventDisableTiming));
    VKX_HIP(hipEventRecord(ctx->lattices_ready, st));
    ctx->lattices_armed = true;
    return VKX_OK;
}
