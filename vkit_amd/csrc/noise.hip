// Throughput-mode noise: the int16 plane of gaussion_noise (photometric/noise.py:44-54) drawn on the device.
//
// The reference draws np.round(rng.normal(0, std, shape)) from the caller's numpy Generator: 12.6 M float64 normals per
// 2048^2 RGB page on one host core (16 Mpx/s measured), and the parity path has to ship the resulting
// int16 plane over the link (6 bytes per result pixel).  This entry point produces a plane with the SAME DISTRIBUTION --
// integer k with probability Phi((k + 1/2) / std) - Phi((k - 1/2) / std), the law of round(N(0, std)) -- but not the same
// values: it cannot, the values are a function of numpy's bit stream.  It is a separately labelled mode; everything that
// must match the reference pixel for pixel keeps taking the caller's plane.
//
// Definition (restated in oracle/vkx_oracle.c, compared bit for bit):
//   * the plane is the flat sequence of its n = h * w * cn samples in C order; sample s takes the (s & 3)-th of the four
//     16-bit uniforms of ONE Philox2x32-10 block (Salmon et al., SC'11; multiplier 0xD256D193, Weyl constant
//     0x9E3779B9) with counter = (s >> 2, seed >> 32) and key = (uint32) seed: words (r0, r1) -> r0 & 0xffff, r0 >> 16,
//     r1 & 0xffff, r1 >> 16;
//   * a 16-bit uniform u selects table[u], the smallest k with Phi((k + 1/2) / std) > (u + 1/2) / 65536 (inverse CDF at
//     the midpoint of the u-th of 65536 equal slices, built on the host in double precision, clamped to +-32767).
// The quantisation of the probabilities to multiples of 2^-16 cuts the tails beyond ~4.2 std.
#include "vkx_internal.h"

#include <math.h>
#include <algorithm>

namespace {

__device__ __forceinline__ void philox2x32_10(uint32_t c0, uint32_t c1, uint32_t key, uint32_t &o0, uint32_t &o1)
{
#pragma unroll
    for (int r = 0; r < 10; r++) {
        // both halves of the product from one v_mad_u64_u32 (full rate; v_mul_hi_u32 + v_mul_lo_u32 are quarter rate)
        const uint64_t prod = (uint64_t)0xD256D193u * c0;
        c0 = (uint32_t)(prod >> 32) ^ key ^ c1;
        c1 = (uint32_t)prod;
        key += 0x9E3779B9u;
    }
    o0 = c0; o1 = c1;
}

// One Philox block = four consecutive samples = 8 bytes of the plane per lane and iteration: the stores of a wavefront
// are one contiguous 512-byte run.  The inverse-CDF table sits in LDS as int8 when every entry fits (std up to ~30, the
// policy's whole range; 64 KB, two workgroups per CU); larger deviations read the int16 table through L2.
struct NoisePlane {            // device form of vkx_noise_plane
    int16_t *dst;
    long long n;               // samples
    long long stride_el;
    int w_el, pad;
    uint32_t key, stream;
};

// All planes of a call in one launch: the 64 KB table is staged once per workgroup, not once per plane and workgroup (a
// 2048^2 x 3 plane is six iterations per thread).  `planes` == nullptr: the single plane `one` of the kernel arguments.
template <bool LDS_TABLE>
__global__ void __launch_bounds__(1024) k_noise_normal_i16(const NoisePlane *__restrict__ planes, NoisePlane one, int n_planes,
                                                           const int16_t *__restrict__ table, const int8_t *__restrict__ table8)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int8_t *lt = (int8_t *)smem;
    if (LDS_TABLE) {
        const uint4 *g = (const uint4 *)table8;
        uint4 *l = (uint4 *)smem;
        for (int i = threadIdx.x; i < 65536 / 16; i += 1024) l[i] = g[i];
        __syncthreads();
    }
    for (int p = 0; p < n_planes; p++) {
        const NoisePlane pl = planes ? planes[p] : one;
        int16_t *__restrict__ dst = pl.dst;
        const long long n = pl.n, stride_el = pl.stride_el;
        const int w_el = pl.w_el;
        const long long groups = (n + 3) >> 2;
        // the workgroups take turns at the short last stride of a plane
        const unsigned first = (blockIdx.x + (unsigned)p * 37u) % gridDim.x;
        for (long long q = (long long)first * 1024 + threadIdx.x; q < groups; q += (long long)gridDim.x * 1024) {
            uint32_t r0, r1;
            philox2x32_10((uint32_t)q, pl.stream, pl.key, r0, r1);
            const uint32_t u0 = r0 & 0xffffu, u1 = r0 >> 16, u2 = r1 & 0xffffu, u3 = r1 >> 16;
            int k0, k1, k2, k3;
            if (LDS_TABLE) { k0 = lt[u0]; k1 = lt[u1]; k2 = lt[u2]; k3 = lt[u3]; }
            else { k0 = table[u0]; k1 = table[u1]; k2 = table[u2]; k3 = table[u3]; }
            const long long s = q << 2;
            if (stride_el == w_el && s + 3 < n) {
                // contiguous plane: the four samples are 8 adjacent bytes
                uint2 v;
                v.x = (uint32_t)(k0 & 0xffff) | ((uint32_t)k1 << 16);
                v.y = (uint32_t)(k2 & 0xffff) | ((uint32_t)k3 << 16);
                *(uint2 *)(dst + s) = v;
            } else {
                const int ks[4] = {k0, k1, k2, k3};
                for (int j = 0; j < 4; j++) {
                    const long long t = s + j;
                    if (t >= n) break;
                    const long long row = t / w_el;
                    dst[row * stride_el + (t - row * w_el)] = (int16_t)ks[j];
                }
            }
        }
    }
}

} // namespace

// table[u] for u in [0, 65536): see the header.  Host, double precision; `std` > 0.
static void build_normal_table(double std, int16_t *table)
{
    // Phi((k + 1/2) / std) for growing k, walked together with u (both monotone)
    const double inv = 1.0 / (std * 1.4142135623730951);
    int k = -32767;
    // start near the first k whose cumulative probability reaches the first slice
    {
        const double z = -4.6 * std;       // Phi(-4.6) < 2^-18
        if (z > -32766.0) k = (int)floor(z) - 1;
        if (k < -32767) k = -32767;
    }
    double cdf = 0.5 * erfc(-((double)k + 0.5) * inv);
    for (int u = 0; u < 65536; u++) {
        const double p = ((double)u + 0.5) / 65536.0;
        while (cdf <= p && k < 32767) {
            k++;
            cdf = 0.5 * erfc(-((double)k + 0.5) * inv);
        }
        table[u] = (int16_t)k;
    }
}

VKX_EXPORT int vkx_noise_normal_table(double std, int16_t *table_host)
{
    VKX_REQUIRE(table_host != nullptr, "NULL table");
    VKX_REQUIRE(std > 0.0 && std < 8000.0, "std out of range");
    build_normal_table(std, table_host);
    return VKX_OK;
}

VKX_EXPORT int vkx_noise_normal_i16_batch_dev(vkx_ctx *ctx, const vkx_noise_plane *planes, int n_planes, double std)
{
    VKX_REQUIRE(ctx && planes, "NULL argument");
    VKX_REQUIRE(n_planes >= 1 && n_planes <= 65536, "1 .. 65536 planes per call");
    VKX_REQUIRE(std > 0.0 && std < 8000.0, "std out of range");
    long long most = 0;
    int live = 0;
    for (int i = 0; i < n_planes; i++) {
        const vkx_noise_plane &p = planes[i];
        VKX_REQUIRE(p.h >= 0 && p.w >= 0 && p.cn >= 1 && p.cn <= 4, "bad shape");
        VKX_REQUIRE((long long)p.h * p.w * p.cn <= 0x3ffffffffLL, "more than 2^34 samples");
        if (p.h == 0 || p.w == 0) continue;
        VKX_REQUIRE(p.dst != nullptr, "NULL plane");
        VKX_REQUIRE(p.stride_el >= (ptrdiff_t)p.w * p.cn, "row stride shorter than a row");
        // a dense plane leaves as 8-byte stores (four samples of one Philox block)
        VKX_REQUIRE(p.stride_el != (ptrdiff_t)p.w * p.cn || ((uintptr_t)p.dst & 7) == 0, "a dense plane must be 8-byte aligned");
        most = std::max(most, (long long)p.h * p.w * p.cn);
        live++;
    }
    if (!live) return VKX_OK;
    // the tables of the last std stay on the device (a chain uses one std for a whole batch): int16 [65536], then the
    // same entries as int8 when they all fit
    constexpr size_t kT16 = 65536 * sizeof(int16_t), kT8 = 65536;
    int rc;
    if (!ctx->noise_table.ptr || ctx->noise_table_std != std) {
        if ((rc = vkx_scratch_reserve(ctx, &ctx->noise_table, kT16 + kT8))) return rc;
        void *ring = nullptr;
        if ((rc = vkx_desc_ring_take(ctx, kT16 + kT8, &ring))) return rc;
        int16_t *t16 = (int16_t *)ring;
        int8_t *t8 = (int8_t *)ring + kT16;
        build_normal_table(std, t16);
        ctx->noise_table_fits8 = t16[0] >= -127 && t16[65535] <= 127;
        for (int u = 0; u < 65536; u++) t8[u] = (int8_t)(ctx->noise_table_fits8 ? t16[u] : 0);
        vkx_device_guard guard(ctx);
        VKX_HIP(hipMemcpyAsync(ctx->noise_table.ptr, ring, kT16 + kT8, hipMemcpyHostToDevice, ctx->stream));
        ctx->noise_table_std = std;
    }
    const int16_t *t16 = (const int16_t *)ctx->noise_table.ptr;
    const int8_t *t8 = (const int8_t *)ctx->noise_table.ptr + kT16;
    auto device_form = [](const vkx_noise_plane &p) {
        NoisePlane d;
        d.dst = p.dst; d.n = (long long)p.h * p.w * p.cn; d.stride_el = (long long)p.stride_el; d.w_el = p.w * p.cn; d.pad = 0;
        d.key = (uint32_t)p.seed; d.stream = (uint32_t)(p.seed >> 32);
        return d;
    };
    NoisePlane one = {};
    const NoisePlane *d_planes = nullptr;
    int n_dev = 0;
    if (live == 1) {
        for (int i = 0; i < n_planes; i++)
            if (planes[i].h && planes[i].w) one = device_form(planes[i]);
        n_dev = 1;
    } else {
        if ((rc = vkx_scratch_reserve(ctx, &ctx->misc, sizeof(NoisePlane) * (size_t)live))) return rc;
        void *ring = nullptr;
        if ((rc = vkx_desc_ring_take(ctx, sizeof(NoisePlane) * (size_t)live, &ring))) return rc;
        NoisePlane *h = (NoisePlane *)ring;
        for (int i = 0; i < n_planes; i++)
            if (planes[i].h && planes[i].w) h[n_dev++] = device_form(planes[i]);
        vkx_device_guard guard(ctx);
        if ((rc = vkx_small_to_device(ctx, ctx->misc.ptr, ring, sizeof(NoisePlane) * (size_t)live))) return rc;
        d_planes = (const NoisePlane *)ctx->misc.ptr;
    }
    vkx_device_guard guard(ctx);
    const unsigned blocks = (unsigned)std::min<long long>(((most + 3) / 4 + 1023) / 1024, 512);    // 2 resident workgroups per CU
    VKX_TIMED(ctx, "k_noise_normal_i16");
    if (ctx->noise_table_fits8)
        k_noise_normal_i16<true><<<blocks, 1024, kT8, ctx->stream>>>(d_planes, one, n_dev, t16, t8);
    else
        k_noise_normal_i16<false><<<blocks, 1024, 0, ctx->stream>>>(d_planes, one, n_dev, t16, t8);
    VKX_LAUNCH_CHECK();
    return VKX_OK;
}

VKX_EXPORT int vkx_noise_normal_i16_dev(vkx_ctx *ctx, int16_t *dst, ptrdiff_t stride_el, int h, int w, int cn, double std,
                                        uint64_t seed)
{
    VKX_REQUIRE(ctx && dst, "NULL argument");
    vkx_noise_plane p = {};
    p.dst = dst; p.stride_el = stride_el; p.h = h; p.w = w; p.cn = cn; p.seed = seed;
    return vkx_noise_normal_i16_batch_dev(ctx, &p, 1, std);
}

VKX_EXPORT int vkx_noise_normal_i16(vkx_ctx *ctx, int16_t *dst, ptrdiff_t stride_el, int h, int w, int cn, double std,
                                    uint64_t seed)
{
    VKX_REQUIRE(ctx && dst, "NULL argument");
    VKX_REQUIRE(h >= 0 && w >= 0 && cn >= 1 && cn <= 4 && stride_el >= (ptrdiff_t)w * cn, "bad shape");
    if (h == 0 || w == 0) return VKX_OK;
    const size_t bytes = sizeof(int16_t) * (size_t)h * w * cn;
    int rc = vkx_scratch_reserve(ctx, &ctx->stage[1], bytes);
    if (rc) return rc;
    rc = vkx_noise_normal_i16_dev(ctx, (int16_t *)ctx->stage[1].ptr, (ptrdiff_t)w * cn, h, w, cn, std, seed);
    if (rc) return rc;
    vkx_device_guard guard(ctx);
    if (stride_el == (ptrdiff_t)w * cn)
        VKX_HIP(hipMemcpyAsync(dst, ctx->stage[1].ptr, bytes, hipMemcpyDeviceToHost, ctx->stream));
    else
        VKX_HIP(vkx_copy_plane(dst, (size_t)stride_el * 2, ctx->stage[1].ptr, (size_t)w * cn * 2, (size_t)w * cn * 2, (size_t)h,
                                 hipMemcpyDeviceToHost, ctx->stream));
    VKX_HIP(hipStreamSynchronize(ctx->stream));
    return VKX_OK;
}
