// numpy's float32 sums of a uint8 image, as np.mean forms them in std_shift (photometric/color.py:165-210:
// ``np.mean(mat)`` of the float32 copy of a single plane, ``np.mean(mat.reshape(-1, k), axis=0)`` of a k-channel one) -- the value
// depends on the ORDER of the float32 additions, which numpy (2.2, the version the goldens were made with; checked against numpy
// itself in tests/test_reduce_semantics.py and tests/test_gpu_reduce.py) fixes as follows:
//   * one plane (H x W), and the channels picked with ``mat[:, :, channels]`` (an advanced index: numpy lays the copy out channel
//     first, ``reshape(-1, k)`` is a view of it, so the reduced axis is the contiguous one): per channel the reduction runs over the
//     contiguous elements in buffer-sized pieces of 8 192; every piece is summed pairwise (exact here: 8 192 x 255 < 2^24, so ANY
//     order gives the integer), the pieces are accumulated one after the other in float32;
//   * all C >= 2 channels of an interleaved image, axis 0 of (N, C): the inner loop runs over the C outputs, the reduced axis is
//     the outer one -- a plain sequential float32 accumulation ``s[c] += v[i][c]`` over the N pixels.
// A sequential float32 sum of non-negative integers is exact below 2^24; above, in the binade [2^(23+e), 2^(24+e)), the sum is a
// multiple of q = 2^e and every addition rounds v to a multiple of q -- to the nearest, a tie (v mod q == q / 2) to the EVEN
// multiple, which depends on the parity of s / q.  That parity is the only state the rounding has: for a run of elements and a fixed
// binade the map (parity in) -> (units added, parity out) composes, so
//   k_sum_summaries  one lane per run of kRun pixels and selected channel: the exact sum (e = 0) and, for e = 1 .. 6 and both
//                    parities, (units, parity out);
//   k_sum_walk       one wavefront per channel walks the runs in order: the binade of the running sum picks the entry; a run
//                    during which the sum would leave its binade (at most six per channel) is added element by element in float32.
// Host: the mean is the float32 quotient sum / N, as np.mean divides.
#include "vkx_internal.h"

#include <vector>

namespace {

constexpr int kRun = 256;             // pixels per run of the sequential form
constexpr int kPiece = 8192;          // numpy's reduction buffer (elements) of the single-plane form
constexpr int kRegimes = 7;           // e = 0 .. 6: sums below 2^30 (255 x 2^22 pixels)

struct RunSummary {                   // of one run and channel
    uint32_t exact;                   // e = 0: the integer sum
    uint32_t step[kRegimes - 1][2];   // e = 1 .. 6, parity in 0 / 1: units of 2^e added << 1 | parity out
};

struct SumArgs {
    const uint8_t *src;
    long long n_px;                   // h * w
    int w, cn;
    ptrdiff_t stride;
    int sel[4], n_sel;
};

__device__ __forceinline__ const uint8_t *px_ptr(const SumArgs &a, long long p)
{
    if (a.stride == (ptrdiff_t)a.w * a.cn) return a.src + p * a.cn;       // dense rows
    const uint32_t row = (uint32_t)p / (uint32_t)a.w;                     // p < 2^22
    return a.src + (ptrdiff_t)row * a.stride + (ptrdiff_t)((uint32_t)p - row * (uint32_t)a.w) * a.cn;
}

__global__ void __launch_bounds__(64) k_sum_summaries(SumArgs a, long long n_runs, RunSummary *__restrict__ out)
{
    const long long run = (long long)blockIdx.x * 64 + threadIdx.x;
    if (run >= n_runs) return;
    const long long p0 = run * kRun, p1 = min(p0 + kRun, a.n_px);
    for (int k = 0; k < a.n_sel; k++) {
        const int c = a.sel[k];
        uint32_t exact = 0;
        uint32_t units[kRegimes - 1][2], par[kRegimes - 1][2];
#pragma unroll
        for (int e = 0; e < kRegimes - 1; e++) { units[e][0] = units[e][1] = 0; par[e][0] = 0; par[e][1] = 1; }
        for (long long p = p0; p < p1; p++) {
            const uint32_t v = px_ptr(a, p)[c];
            exact += v;
#pragma unroll
            for (int e = 1; e < kRegimes; e++) {
                const uint32_t q = 1u << e, half = q >> 1;
                const uint32_t up = v >> e, b = v & (q - 1);
                const uint32_t round_up = b > half ? 1u : 0u, tie = b == half ? 1u : 0u;
#pragma unroll
                for (int s = 0; s < 2; s++) {
                    // a tie goes to the even multiple: up one unit exactly when (s / q + v / q) is odd
                    const uint32_t t = up + round_up + (tie & ((par[e - 1][s] + up) & 1u));
                    units[e - 1][s] += t;
                    par[e - 1][s] = (par[e - 1][s] + t) & 1u;
                }
            }
        }
        RunSummary r;
        r.exact = exact;
#pragma unroll
        for (int e = 0; e < kRegimes - 1; e++) { r.step[e][0] = units[e][0] << 1 | par[e][0]; r.step[e][1] = units[e][1] << 1 | par[e][1]; }
        out[run * a.n_sel + k] = r;
    }
}

// bits of a positive integer
__device__ __forceinline__ int bit_length(unsigned long long v) { return 64 - __builtin_clzll(v | 1ull); }

__global__ void __launch_bounds__(64) k_sum_walk(SumArgs a, long long n_runs, const RunSummary *__restrict__ runs, float *__restrict__ out)
{
    __shared__ RunSummary batch[64];
    const int k = blockIdx.x, c = a.sel[k], lane = threadIdx.x;
    unsigned long long s = 0;          // the float32 running sum: an integer float32 represents exactly
    for (long long r0 = 0; r0 < n_runs; r0 += 64) {
        if (r0 + lane < n_runs) batch[lane] = runs[(r0 + lane) * a.n_sel + k];
        __syncthreads();
        if (lane == 0) {
            const int m = (int)min((long long)64, n_runs - r0);
            for (int i = 0; i < m; i++) {
                const int bl = bit_length(s), e = bl > 24 ? bl - 24 : 0;
                unsigned long long next;
                bool ok;
                if (e == 0) {
                    next = s + batch[i].exact;
                    ok = next < (1ull << 24);
                } else if (e < kRegimes) {
                    const uint32_t st = batch[i].step[e - 1][(s >> e) & 1ull];
                    next = s + ((unsigned long long)(st >> 1) << e);
                    ok = next < (1ull << bl);                    // the sum never left [2^(bl-1), 2^bl): every addition rounded on that grid
                } else {
                    next = 0; ok = false;
                }
                if (!ok) {                                       // the binade changes inside this run: float32 additions, one by one
                    float f = (float)s;
                    const long long p0 = (r0 + i) * kRun, p1 = min(p0 + kRun, a.n_px);
                    for (long long p = p0; p < p1; p++) f += (float)px_ptr(a, p)[c];
                    next = (unsigned long long)f;
                }
                s = next;
            }
        }
        __syncthreads();
    }
    if (lane == 0) out[k] = (float)s;
}

// the single-plane form: exact integer sums of the pieces of kPiece contiguous elements of channel sel[0] ...
__global__ void __launch_bounds__(256) k_sum_pieces(SumArgs a, uint32_t *__restrict__ pieces_all, int n_pieces)
{
    __shared__ uint32_t part[4];
    const long long p0 = (long long)blockIdx.x * kPiece, p1 = min(p0 + kPiece, a.n_px);
    const int c = a.sel[blockIdx.y];
    uint32_t *pieces = pieces_all + (size_t)blockIdx.y * n_pieces;
    uint32_t acc = 0;
    for (long long p = p0 + threadIdx.x; p < p1; p += 256) acc += px_ptr(a, p)[c];
    for (int d = 32; d > 0; d >>= 1) acc += __shfl_down(acc, d, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) pieces[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// ... accumulated one after the other in float32
__global__ void k_sum_pieces_walk(const uint32_t *__restrict__ pieces, int n_pieces, float *__restrict__ out)
{
    const uint32_t *mine = pieces + (size_t)blockIdx.x * n_pieces;
    float f = 0.f;
    for (int i = 0; i < n_pieces; i++) f += (float)mine[i];
    out[blockIdx.x] = f;
}

} // namespace

// sums_host [n_sel] = the float32 sums numpy forms for np.mean (see the head of this file); src is a DEVICE plane.  Synchronous: the
// caller goes on with the values on the host (std_shift builds its 256-entry tables from them).
VKX_EXPORT int vkx_sum_f32_u8_dev(vkx_ctx *ctx, const uint8_t *src, int h, int w, int cn, ptrdiff_t src_stride, const int32_t *channels_host,
                                  int n_sel, int sequential, float *sums_host)
{
    VKX_REQUIRE(ctx && src && channels_host && sums_host, "NULL argument");
    VKX_REQUIRE(h >= 1 && w >= 1 && cn >= 1 && cn <= 4 && n_sel >= 1 && n_sel <= 4, "bad shape");
    VKX_REQUIRE((long long)h * w <= (1ll << 22), "at most 2^22 pixels (sums below 2^30)");
    SumArgs a;
    a.src = src; a.n_px = (long long)h * w; a.w = w; a.cn = cn; a.stride = src_stride; a.n_sel = n_sel;
    for (int k = 0; k < 4; k++) {
        a.sel[k] = k < n_sel ? channels_host[k] : 0;
        VKX_REQUIRE(a.sel[k] >= 0 && a.sel[k] < cn, "channel outside the image");
    }
    vkx_device_guard guard(ctx);
    int rc;
    if (!sequential) {
        const int n_pieces = (int)((a.n_px + kPiece - 1) / kPiece);
        if ((rc = vkx_scratch_reserve(ctx, &ctx->misc, sizeof(uint32_t) * (size_t)n_pieces * n_sel + 256))) return rc;
        float *d_out = (float *)ctx->misc.ptr;
        uint32_t *pieces = (uint32_t *)((unsigned char *)ctx->misc.ptr + 256);
        { VKX_TIMED(ctx, "k_sum_pieces"); k_sum_pieces<<<dim3(n_pieces, n_sel), 256, 0, ctx->stream>>>(a, pieces, n_pieces); }
        VKX_LAUNCH_CHECK();
        k_sum_pieces_walk<<<n_sel, 1, 0, ctx->stream>>>(pieces, n_pieces, d_out);
        VKX_LAUNCH_CHECK();
        VKX_HIP(hipMemcpyAsync(sums_host, d_out, sizeof(float) * n_sel, hipMemcpyDeviceToHost, ctx->stream));
    } else {
        const long long n_runs = (a.n_px + kRun - 1) / kRun;
        if ((rc = vkx_scratch_reserve(ctx, &ctx->misc, sizeof(RunSummary) * (size_t)n_runs * n_sel + 256))) return rc;
        float *d_out = (float *)ctx->misc.ptr;
        RunSummary *runs = (RunSummary *)((unsigned char *)ctx->misc.ptr + 256);
        { VKX_TIMED(ctx, "k_sum_summaries"); k_sum_summaries<<<vkx_blocks((size_t)n_runs, 64), 64, 0, ctx->stream>>>(a, n_runs, runs); }
        VKX_LAUNCH_CHECK();
        { VKX_TIMED(ctx, "k_sum_walk"); k_sum_walk<<<n_sel, 64, 0, ctx->stream>>>(a, n_runs, runs, d_out); }
        VKX_LAUNCH_CHECK();
        VKX_HIP(hipMemcpyAsync(sums_host, d_out, sizeof(float) * n_sel, hipMemcpyDeviceToHost, ctx->stream));
    }
    VKX_HIP(hipStreamSynchronize(ctx->stream));
    return VKX_OK;
}

// host plane: staged through the context
VKX_EXPORT int vkx_sum_f32_u8(vkx_ctx *ctx, const uint8_t *src, int h, int w, int cn, ptrdiff_t src_stride, const int32_t *channels_host,
                              int n_sel, int sequential, float *sums_host)
{
    VKX_REQUIRE(ctx && src && channels_host && sums_host, "NULL argument");
    VKX_REQUIRE(h >= 1 && w >= 1 && cn >= 1 && cn <= 4, "bad shape");
    const size_t row = (size_t)w * cn, bytes = row * h;
    int rc = vkx_scratch_reserve(ctx, &ctx->stage[0], bytes);
    if (rc) return rc;
    {
        vkx_device_guard guard(ctx);
        VKX_HIP(vkx_copy_plane(ctx->stage[0].ptr, row, src, (size_t)src_stride, row, (size_t)h, hipMemcpyHostToDevice, ctx->stream));
    }
    return vkx_sum_f32_u8_dev(ctx, (const uint8_t *)ctx->stage[0].ptr, h, w, cn, (ptrdiff_t)row, channels_host, n_sel, sequential, sums_host);
}
