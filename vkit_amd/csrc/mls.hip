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

__global__ void __launch_bounds__(64) k_mls_project(Handles hd, const double *__restrict__ vertices, int n_vertices,
                                                    double *__restrict__ out, int *__restrict__ bad)
{
    extern __shared__ float staged[];              // [2 n] source handles, [2 n] targets when n <= kLdsHandles
    const int n = hd.n;
    const float *sp = hd.p, *sq = hd.q;
    if (n <= kLdsHandles) {
        for (int i = threadIdx.x; i < 2 * n; i += 64) { staged[i] = hd.p[i]; staged[2 * n + i] = hd.q[i]; }
        __syncthreads();
        sp = staged;
        sq = staged + 2 * n;
    }
    const int v = blockIdx.x * 64 + threadIdx.x;
    if (v >= n_vertices) return;
    const double vxd = vertices[2 * v], vyd = vertices[2 * v + 1];
    // identity on exact handle hits (mls.py:57-61); a later duplicate handle overrides an earlier one like the dict
    int hit = -1;
    for (int i = 0; i < n; i++)
        if (hd.ps[2 * i] == vxd && hd.ps[2 * i + 1] == vyd) hit = i;
    if (hit >= 0) {
        out[2 * v] = hd.qs[2 * hit];
        out[2 * v + 1] = hd.qs[2 * hit + 1];
        return;
    }
    const float vx = (float)vxd, vy = (float)vyd;   // python float operands enter float32 arithmetic as float32
    bool zero = false;
    const float sw = pairwise_sum(n, [&](int i) { return weight(sp, i, vx, vy, zero); });
    if (zero) {                                     // np.errstate(divide='raise')
        atomicMax(bad, v + 1);
        out[2 * v] = 0.0; out[2 * v + 1] = 0.0;
        return;
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
    out[2 * v] = (double)(sx / mu + qsx);
    out[2 * v + 1] = (double)(sy / mu + qsy);
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
