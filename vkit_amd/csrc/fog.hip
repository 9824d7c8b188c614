// The fog density field of vkit's `fog` (reference photometric/effect.py:89-216: generate_diamond_square_mask + the stretch to
// [ratio_min, ratio_max]) on the device, with the caller's PCG64 stream and numpy's roundings:
//   * a (2^k + 1)^2 float32 lattice; four corner draws (the host makes them: four scalar rng.uniform calls);
//   * per level l = 0 .. k - 1 (step = 2^(k - l), n = 2^l + 1 corners per side, nw = roughness ** l as the host computed it):
//       centres   (n - 1)^2 draws:  around = f32(f32(c + down) + f32(c + right))  -- the corner twice, its lower and its right
//                 neighbour: the reference's sum, not the four corners of the square --;
//                 value64 = f64(f32(f32(f32(1 - nw) * around) / 4)) + nw * u      (python scalars are weak against a float32 array);
//                 the field keeps f32(value64), the next two families read value64 itself (numpy's `centres` array is float64);
//       h-edges   n (n - 1) draws:  around64 = f64(f32(c + right)) + (centre above + centre below, rows wrap; the last corner row reuses
//                 the first row's sum: np.vstack([.., centres_vert[0]]));  value = ((1 - nw) * around64) / 4 + nw * u;
//       v-edges   (n - 1) n draws:  around64 = f64(f32(c + down)) + (centre left + centre right, columns wrap; the last corner column
//                 takes centres_hori[0][i] -- the first ROW, reshaped to a column: np.hstack([.., centres_hori[0].reshape(-1, 1)]));
//     draws in C order per family, families in that order: position of every draw known in closed form -> vkx_pcg64_doubles_dev;
//   * the stretch on the cropped field: x - min, / max of that, * f32(ratio_max - ratio_min), + f32(ratio_min), float32 each.
// tests/test_gpu_fog.py compares field and mask bit for bit with the host restatement (vkit_amd/.../effect.py), which the golden
// vectors of the reference pin.
#include "vkx_internal.h"

#include <vector>

namespace {

__global__ void __launch_bounds__(256) k_fog_centres(float *__restrict__ field, int size, int step, int n, double nw, const double *__restrict__ u,
                                                     double *__restrict__ centres)
{
    const int m = n - 1;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= m * m) return;
    const int i = idx / m, j = idx - i * m, half = step >> 1;
    const float c = field[(size_t)(i * step) * size + j * step];
    const float down = field[(size_t)((i + 1) * step) * size + j * step], right = field[(size_t)(i * step) * size + (j + 1) * step];
    const float around = (c + down) + (c + right);
    const float w1 = (float)(1.0 - nw);
    const float t = (w1 * around) / 4.0f;
    const double v = (double)t + nw * u[idx];
    centres[idx] = v;
    field[(size_t)(i * step + half) * size + j * step + half] = (float)v;
}

__global__ void __launch_bounds__(256) k_fog_edges(float *__restrict__ field, int size, int step, int n, double nw, const double *__restrict__ uh,
                                                   const double *__restrict__ uv, const double *__restrict__ centres)
{
    const int m = n - 1, half = step >> 1;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int per = n * m;
    if (idx >= 2 * per) return;
    const double w1 = 1.0 - nw;
    if (idx < per) {                      // midpoints of the horizontal edges: corner row i, between columns j and j + 1
        const int i = idx / m, j = idx - i * m;
        const float c = field[(size_t)(i * step) * size + j * step], right = field[(size_t)(i * step) * size + (j + 1) * step];
        const int ic = i == m ? 0 : i;    // the appended row repeats row 0
        const int above = ic == 0 ? m - 1 : ic - 1;
        const double cv = centres[ic * m + j] + centres[above * m + j];
        const double around = (double)(c + right) + cv;
        field[(size_t)(i * step) * size + j * step + half] = (float)((w1 * around) / 4 + nw * uh[idx]);
    } else {                              // midpoints of the vertical edges: between corner rows i and i + 1, column j
        const int e = idx - per;
        const int i = e / n, j = e - i * n;
        const float c = field[(size_t)(i * step) * size + j * step], down = field[(size_t)((i + 1) * step) * size + j * step];
        double ch;
        if (j < m) {
            const int left = j == 0 ? m - 1 : j - 1;
            ch = centres[i * m + j] + centres[i * m + left];
        } else {                          // the appended column: centres_hori[0][i], the FIRST row's entry i
            const int left = i == 0 ? m - 1 : i - 1;
            ch = centres[i] + centres[left];
        }
        const double around = (double)(c + down) + ch;
        field[(size_t)(i * step + half) * size + j * step] = (float)((w1 * around) / 4 + nw * uv[e]);
    }
}

// floats >= 0 order like their bit patterns
__global__ void __launch_bounds__(256) k_fog_min(const float *__restrict__ field, int size, int up, int left, int h, int w, unsigned *__restrict__ red)
{
    __shared__ unsigned s[4];
    unsigned mn = 0x7f800000u;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < (long long)h * w; i += (long long)gridDim.x * 256) {
        const int y = (int)(i / w), x = (int)(i - (long long)y * w);
        mn = min(mn, __float_as_uint(field[(size_t)(up + y) * size + left + x]));
    }
    for (int o = 32; o; o >>= 1) mn = min(mn, (unsigned)__shfl_xor((int)mn, o));
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = mn;
    __syncthreads();
    if (threadIdx.x == 0) atomicMin(&red[0], min(min(s[0], s[1]), min(s[2], s[3])));
}

__global__ void __launch_bounds__(256) k_fog_shift_max(const float *__restrict__ field, int size, int up, int left, int h, int w, float *__restrict__ mask,
                                                       unsigned *__restrict__ red)
{
    __shared__ unsigned s[4];
    const float mn = __uint_as_float(red[0]);
    unsigned mx = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < (long long)h * w; i += (long long)gridDim.x * 256) {
        const int y = (int)(i / w), x = (int)(i - (long long)y * w);
        const float v = field[(size_t)(up + y) * size + left + x] - mn;
        mask[i] = v;
        mx = max(mx, __float_as_uint(v));
    }
    for (int o = 32; o; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(&red[1], max(max(s[0], s[1]), max(s[2], s[3])));
}

__global__ void __launch_bounds__(256) k_fog_scale(float *__restrict__ mask, long long n, const unsigned *__restrict__ red, float span, float lo)
{
    const float mx = __uint_as_float(red[1]);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float v = mask[i] / mx;
        v = v * span;
        mask[i] = v + lo;
    }
}

}   // namespace

VKX_EXPORT int vkx_fog_field_f32_dev(vkx_ctx *ctx, const uint64_t *state, const uint64_t *inc, int levels, const double *noise_weight_host,
                                     const float *corners_host, float *field, long long *consumed_host)
{
    VKX_REQUIRE(ctx && state && inc && noise_weight_host && corners_host && field && consumed_host, "NULL argument");
    VKX_REQUIRE(levels >= 1 && levels <= 14, "1 .. 14 levels (lattices of 3 .. 16385 points a side)");
    vkx_device_guard guard(ctx);
    const int size = (1 << levels) + 1;
    // draws of level l: (n - 1)^2 + 2 n (n - 1) = (n - 1)(3 n - 1), n = 2^l + 1
    long long total = 0;
    std::vector<long long> first((size_t)levels);
    for (int l = 0; l < levels; l++) {
        const long long n = (1ll << l) + 1;
        first[(size_t)l] = total;
        total += (n - 1) * (3 * n - 1);
    }
    const long long m_last = 1ll << (levels - 1);
    int rc;
    if ((rc = vkx_scratch_reserve(ctx, &ctx->fog_work, sizeof(double) * (size_t)(total + m_last * m_last) + 256))) return rc;
    double *u = (double *)ctx->fog_work.ptr, *centres = u + total;
    if ((rc = vkx_pcg64_doubles_dev(ctx, state, inc, total, u))) return rc;
    // the corners: field[0, 0], field[0, -1], field[-1, -1], field[-1, 0] in the reference's order
    const size_t at[4] = {0, (size_t)size - 1, (size_t)size * size - 1, (size_t)(size - 1) * size};
    for (int k = 0; k < 4; k++) VKX_HIP(hipMemcpyAsync(field + at[k], corners_host + k, sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    for (int l = 0; l < levels; l++) {
        const int n = (1 << l) + 1, m = n - 1, step = (size - 1) >> l;
        const double nw = noise_weight_host[l];
        const double *uc = u + first[(size_t)l], *uh = uc + (long long)m * m, *uv = uh + (long long)n * m;
        { VKX_TIMED(ctx, "k_fog_centres"); k_fog_centres<<<vkx_blocks((size_t)m * m, 256), 256, 0, ctx->stream>>>(field, size, step, n, nw, uc, centres); }
        { VKX_TIMED(ctx, "k_fog_edges"); k_fog_edges<<<vkx_blocks((size_t)2 * n * m, 256), 256, 0, ctx->stream>>>(field, size, step, n, nw, uh, uv, centres); }
    }
    VKX_LAUNCH_CHECK();
    *consumed_host = total;
    return VKX_OK;
}

VKX_EXPORT int vkx_fog_stretch_f32_dev(vkx_ctx *ctx, const float *field, int size, int up, int left, int h, int w, double span, double lo, float *mask)
{
    VKX_REQUIRE(ctx && field && mask, "NULL argument");
    VKX_REQUIRE(size >= 1 && h >= 1 && w >= 1 && up >= 0 && left >= 0 && up + h <= size && left + w <= size, "crop outside the field");
    vkx_device_guard guard(ctx);
    int rc;
    if ((rc = vkx_scratch_reserve(ctx, &ctx->misc, 256))) return rc;
    unsigned *red = (unsigned *)ctx->misc.ptr;
    const unsigned init[2] = {0x7f800000u, 0u};
    VKX_HIP(hipMemcpyAsync(red, init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
    const int grid = (int)std::min<size_t>(1024, vkx_blocks((size_t)h * w, 256));
    { VKX_TIMED(ctx, "k_fog_min"); k_fog_min<<<grid, 256, 0, ctx->stream>>>(field, size, up, left, h, w, red); }
    { VKX_TIMED(ctx, "k_fog_shift_max"); k_fog_shift_max<<<grid, 256, 0, ctx->stream>>>(field, size, up, left, h, w, mask, red); }
    { VKX_TIMED(ctx, "k_fog_scale"); k_fog_scale<<<grid, 256, 0, ctx->stream>>>(mask, (long long)h * w, red, (float)span, (float)lo); }
    VKX_LAUNCH_CHECK();
    return VKX_OK;
}

// ---- glass_blur's shuffle planes (reference photometric/blur.py:204-250; mirror: glass_shuffle_planes) on the device -------------
// pos_y / pos_x int32 [h, w]: the source position every pixel currently shows.  One ROUND: a lattice of centres (rows r0 + i pitch,
// columns c0 + j pitch, pitch = 2 delta + 1); centre k = (i, j) in C order takes `to` = clip(its CURRENT source position + jump_k) and
// the two entries are exchanged the way numpy's tuple assignment does it:
//     pos[centres], pos[to] = pos[to], pos[centres]
// -- both right sides are read first; then every centre takes what its `to` showed; then every `to` takes what its centre showed, IN
// ORDER of k, so that of several centres with one `to` the last wins and a `to` that is itself a centre is overwritten.  The jumps are
// the caller's rng.integers draws (host: a few thousand values per round); the planes never leave the device.
namespace {

__global__ void __launch_bounds__(256) k_glass_init(int *__restrict__ pos_y, int *__restrict__ pos_x, int h, int w)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)h * w) return;
    const int y = (int)(i / w);
    pos_y[i] = y;
    pos_x[i] = (int)(i - (long long)y * w);
}

__global__ void __launch_bounds__(256) k_glass_read(const int *__restrict__ pos_y, const int *__restrict__ pos_x, int h, int w, int r0, int c0, int pitch,
                                                    int n_rows, int n_cols, const int *__restrict__ jump_y, const int *__restrict__ jump_x,
                                                    int *__restrict__ to, uint32_t *__restrict__ from_to, uint32_t *__restrict__ from_centre)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n_rows * n_cols) return;
    const int i = k / n_cols, j = k - i * n_cols;
    const long long centre = (long long)(r0 + i * pitch) * w + c0 + j * pitch;
    const int cy = pos_y[centre], cx = pos_x[centre];
    const int ty = min(max(cy + jump_y[k], 0), h - 1), tx = min(max(cx + jump_x[k], 0), w - 1);
    const int t = ty * w + tx;
    to[k] = t;
    from_to[k] = ((uint32_t)pos_y[t] << 16) | (uint32_t)pos_x[t];
    from_centre[k] = ((uint32_t)cy << 16) | (uint32_t)cx;
}

__global__ void __launch_bounds__(256) k_glass_write(int *__restrict__ pos_y, int *__restrict__ pos_x, int w, int r0, int c0, int pitch, int n_rows,
                                                     int n_cols, const int *__restrict__ to, const uint32_t *__restrict__ from_to,
                                                     const uint32_t *__restrict__ from_centre, unsigned long long *__restrict__ win)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n_rows * n_cols) return;
    const int i = k / n_cols, j = k - i * n_cols;
    const long long centre = (long long)(r0 + i * pitch) * w + c0 + j * pitch;
    pos_y[centre] = (int)(from_to[k] >> 16);
    pos_x[centre] = (int)(from_to[k] & 0xffffu);
    atomicMax(&win[to[k]], ((unsigned long long)(k + 1) << 32) | from_centre[k]);       // the last centre in C order wins its `to`
}

__global__ void __launch_bounds__(256) k_glass_resolve(int *__restrict__ pos_y, int *__restrict__ pos_x, int n, const int *__restrict__ to,
                                                       const unsigned long long *__restrict__ win)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const unsigned long long v = win[to[k]];          // centres that share a `to` write the same winner
    pos_y[to[k]] = (int)((uint32_t)v >> 16);
    pos_x[to[k]] = (int)((uint32_t)v & 0xffffu);
}

__global__ void __launch_bounds__(256) k_glass_clear(int n, const int *__restrict__ to, unsigned long long *__restrict__ win)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k < n) win[to[k]] = 0ull;
}

}   // namespace

VKX_EXPORT int vkx_glass_init_dev(vkx_ctx *ctx, int32_t *pos_y, int32_t *pos_x, int h, int w)
{
    VKX_REQUIRE(ctx && pos_y && pos_x && h >= 1 && w >= 1 && h <= 32767 && w <= 32767, "bad argument");
    vkx_device_guard guard(ctx);
    int rc;
    if ((rc = vkx_scratch_reserve(ctx, &ctx->glass_win, sizeof(unsigned long long) * (size_t)h * w))) return rc;
    VKX_HIP(hipMemsetAsync(ctx->glass_win.ptr, 0, sizeof(unsigned long long) * (size_t)h * w, ctx->stream));
    { VKX_TIMED(ctx, "k_glass_init"); k_glass_init<<<vkx_blocks((size_t)h * w, 256), 256, 0, ctx->stream>>>(pos_y, pos_x, h, w); }
    VKX_LAUNCH_CHECK();
    return VKX_OK;
}

VKX_EXPORT int vkx_glass_round_dev(vkx_ctx *ctx, int32_t *pos_y, int32_t *pos_x, int h, int w, int r0, int c0, int pitch, int n_rows, int n_cols,
                                   const int32_t *jump_y_host, const int32_t *jump_x_host)
{
    VKX_REQUIRE(ctx && pos_y && pos_x && h >= 1 && w >= 1 && pitch >= 1 && r0 >= 0 && c0 >= 0 && n_rows >= 0 && n_cols >= 0, "bad argument");
    const long long n = (long long)n_rows * n_cols;
    if (n == 0) return VKX_OK;
    VKX_REQUIRE(jump_y_host && jump_x_host, "NULL jumps");
    VKX_REQUIRE(r0 + (long long)(n_rows - 1) * pitch < h && c0 + (long long)(n_cols - 1) * pitch < w && n < (1ll << 31), "lattice outside the plane");
    VKX_REQUIRE(ctx->glass_win.ptr && ctx->glass_win.cap >= sizeof(unsigned long long) * (size_t)h * w, "vkx_glass_init_dev first");
    vkx_device_guard guard(ctx);
    int rc;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t part = up(sizeof(int) * (size_t)n);
    if ((rc = vkx_scratch_reserve(ctx, &ctx->fog_work, 5 * part))) return rc;
    unsigned char *base = (unsigned char *)ctx->fog_work.ptr;
    int *d_jy = (int *)base, *d_jx = (int *)(base + part), *d_to = (int *)(base + 2 * part);
    uint32_t *d_ft = (uint32_t *)(base + 3 * part), *d_fc = (uint32_t *)(base + 4 * part);
    VKX_HIP(hipMemcpyAsync(d_jy, jump_y_host, sizeof(int) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    VKX_HIP(hipMemcpyAsync(d_jx, jump_x_host, sizeof(int) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    unsigned long long *win = (unsigned long long *)ctx->glass_win.ptr;
    const unsigned grid = (unsigned)vkx_blocks((size_t)n, 256);
    VKX_TIMED(ctx, "k_glass_round");
    k_glass_read<<<grid, 256, 0, ctx->stream>>>(pos_y, pos_x, h, w, r0, c0, pitch, n_rows, n_cols, d_jy, d_jx, d_to, d_ft, d_fc);
    k_glass_write<<<grid, 256, 0, ctx->stream>>>(pos_y, pos_x, w, r0, c0, pitch, n_rows, n_cols, d_to, d_ft, d_fc, win);
    k_glass_resolve<<<grid, 256, 0, ctx->stream>>>(pos_y, pos_x, (int)n, d_to, win);
    k_glass_clear<<<grid, 256, 0, ctx->stream>>>((int)n, d_to, win);
    VKX_LAUNCH_CHECK();
    VKX_HIP(hipStreamSynchronize(ctx->stream));       // the host jump arrays are the caller's
    return VKX_OK;
}
