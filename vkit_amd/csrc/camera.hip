// Camera-model states on the device: the vertex lattices of camera_plane_only / camera_cubic_curve for a BATCH of configs,
// built from the configs themselves (mechanism/distortion/geometric/camera.py:58-265, 324-423; grid_rendering/grid_creator.py:
// 44-115; element/point.py:31-47 in the reference), so that a caller that holds configs -- not states -- never runs the
// reference's per-state Python (0.28 ms per 2048^2 state: 72 ms per 256-image batch against a 16.6 ms GPU step).
//
// Two halves, both restating the reference's numpy / OpenCV arithmetic operation for operation:
//   vkx_camera_model_host   the dozen SCALARS of a state on the host, in C with the libm Python's math module calls
//                           (CameraModel.__init__, the 2-D -> 3-D strategy's constructor, cv.Rodrigues): float32 where numpy
//                           computes in float32, with the accumulation orders of numpy / OpenBLAS as they are on the host the
//                           goldens were generated on (numpy 2.2.6 + OpenBLAS 0.3.29, AVX-512):
//                             np.linalg.norm of a float32 3-vector   sdot: float32 products accumulated in double, sqrtf
//                             (3,3)^T @ (3,1)  float32               fma(a2, x2, fma(a1, x1, a0 x0))        (sgemv_n scalar tail)
//                             (3,3) @ (3,1)    float32               rows 0, 1: (a0 x0 + a1 x1) + a2 x2 without fusing;
//                                                                    row 2: fma(a2, x2, fma(a0, x0, a1 x1))  (sgemv_t: vector body + tail)
//                             (2,2) @ (2,N)    float32               fma(a1, y, a0 x)                        (sgemm)
//   k_camera_states         one workgroup per state, the per-VERTEX work: lift to 3-D (cubic: float32 projection on the curve
//                           direction, float64 Horner polynomial, minus its mean -- numpy's pairwise sum in pieces of 8 192
//                           elements, as np.mean adds), pinhole projection in float64 with OpenCV's operation order, the shift by
//                           the minimum of the ROUNDED positions (grid_creator.py), rounding half to even, result shape.
// tests/golden/camera_states.npz (generated from the host path in the build container) pins both halves; tests/
// test_camera_states.py compares the host scalars with the Python operators live.
// Compiled with -ffp-contract=off: fused operations appear only as fma() / fmaf().
#include "vkx_internal.h"

#include <math.h>
#include <float.h>
#include <string.h>
#include <vector>

namespace {

constexpr int kPiece = 8192;   // numpy's reduction buffer: np.add.reduce adds a long contiguous axis in pieces of this many elements
constexpr int kBlock = 128;    // numpy's pairwise sum halves its range recursively beyond this many elements
constexpr int kLeaves = 128;   // a piece has at most 8192 / 65 leaves

struct StateDev {
    vkx_camera_model m;
    int32_t *sv, *dv;          // [rows, cols, 2] (x, y)
    double *z;                 // [rows * cols] scratch (cubic curve)
    int height, width, grid_size, kind;
};

// numpy's pairwise sum over n <= 128 contiguous doubles
__device__ __forceinline__ double pairwise_block(const double *__restrict__ a, int n)
{
    if (n < 8) {
        double res = 0.;
        for (int i = 0; i < n; i++) res = res + a[i];
        return res;
    }
    double r[8];
#pragma unroll
    for (int j = 0; j < 8; j++) r[j] = a[j];
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
#pragma unroll
        for (int j = 0; j < 8; j++) r[j] = r[j] + a[i + j];
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res = res + a[i];
    return res;
}

// sum(a, n) = sum(a, n2) + sum(a + n2, n - n2), n2 = n / 2 rounded down to a multiple of 8, down to blocks of <= 128: the
// traversal as an explicit stack; `leaf(lo, n)` returns the value of a block (pass 1 records the blocks and returns 0, pass 2
// hands back the sums the lanes formed in between)
struct WalkStack {          // in LDS: indexed private arrays would live in scratch memory (a round trip to HBM per access)
    int lo_s[8], n_s[8], stage[8];
    double left[8];
};

template <typename F>
__device__ __forceinline__ double pairwise_walk(int n, WalkStack &ws, F leaf)
{
    int *lo_s = ws.lo_s, *n_s = ws.n_s, *stage = ws.stage;
    double *left = ws.left, res = 0.;
    int sp = 1;
    lo_s[0] = 0; n_s[0] = n; stage[0] = 0;
    while (sp > 0) {
        const int k = sp - 1;
        int n2 = n_s[k] / 2;
        n2 -= n2 % 8;
        if (stage[k] == 0) {
            if (n_s[k] <= kBlock) {
                res = leaf(lo_s[k], n_s[k]);
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

// cv.projectPoints with zero distortion on one point (geometric/camera.py project_points here; calib3d's cvProjectPoints2)
__device__ __forceinline__ void project(const vkx_camera_model &m, double X, double Y, double Z, double &px, double &py)
{
    const double x = ((m.R[0] * X + m.R[1] * Y) + m.R[2] * Z) + m.t[0];
    const double y = ((m.R[3] * X + m.R[4] * Y) + m.R[5] * Z) + m.t[1];
    const double z = ((m.R[6] * X + m.R[7] * Y) + m.R[8] * Z) + m.t[2];
    const double iz = z != 0. ? 1.0 / z : 1.0;
    px = (x * iz) * m.fx + m.cx;
    py = (y * iz) * m.fy + m.cy;
    if (m.points_f32) { px = (double)(float)px; py = (double)(float)py; }      // the result takes the dtype of the 3-D points
}

__device__ __forceinline__ double wave_min(double v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fmin(v, __shfl_xor(v, d));
    return v;
}
__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fmax(v, __shfl_xor(v, d));
    return v;
}

__global__ void __launch_bounds__(256) k_camera_states(const StateDev *__restrict__ states, vkx_grid_state *__restrict__ out)
{
    __shared__ int leaf_lo[kLeaves], leaf_n[kLeaves];
    __shared__ double leaf_sum[kLeaves];
    __shared__ int n_leaves;
    __shared__ WalkStack walk;
    __shared__ double s_mean;
    __shared__ double red[4][4];
    __shared__ unsigned bad;
    const StateDev &st = states[blockIdx.x];
    const vkx_camera_model &m = st.m;
    const int rows = m.rows, cols = m.cols, N = rows * cols;
    const int tid = threadIdx.x;
    if (tid == 0) { bad = 0; s_mean = 0.; }
    __syncthreads();
    auto vertex_xy = [&](int i, int &x, int &y) {
        const int r = i / cols, c = i - r * cols;
        x = c + 1 < cols ? c * st.grid_size : st.width - 1;
        y = r + 1 < rows ? r * st.grid_size : st.height - 1;
    };
    const bool cubic = st.kind == VKX_CAMERA_CUBIC_CURVE;
    if (cubic) {
        // depth along the curve direction (CameraCubicCurvePoint2dTo3dStrategy.lift): float32 projection and ratio, float64 Horner
        for (int i = tid; i < N; i += 256) {
            int xi, yi;
            vertex_xy(i, xi, yi);
            const float x = (float)xi, y = (float)yi;
            const float along = fmaf(m.a1, y, m.a0 * x);
            const float ratio = __fdiv_rn(along - m.along_min, m.along_range);
            const double r = (double)ratio;
            double v = 0.0 + m.poly[0];
            v = v * r + m.poly[1];
            v = v * r + m.poly[2];
            v = v * r + m.poly[3];
            st.z[i] = (v * (double)m.along_range) * m.curve_scale;
        }
        __syncthreads();
        // pos_zs.mean(): pieces of 8 192 elements, each numpy's pairwise sum, accumulated one after the other; / N
        double total = 0.;
        for (int lo = 0; lo < N; lo += kPiece) {
            const int n = min(kPiece, N - lo);
            if (tid == 0) {
                int count = 0;
                pairwise_walk(n, walk, [&](int l, int k) { leaf_lo[count] = l; leaf_n[count] = k; count++; return 0.; });
                n_leaves = count;
            }
            __syncthreads();
            if (tid < n_leaves) leaf_sum[tid] = pairwise_block(st.z + lo + leaf_lo[tid], leaf_n[tid]);
            __syncthreads();
            if (tid == 0) {
                int next = 0;
                total = total + pairwise_walk(n, walk, [&](int, int) { return leaf_sum[next++]; });
            }
            __syncthreads();
        }
        if (tid == 0) s_mean = total / (double)N;
        __syncthreads();
    }
    const double mean = s_mean;
    auto smooth_of = [&](int i, double &px, double &py) {
        int xi, yi;
        vertex_xy(i, xi, yi);
        const double Z = cubic ? st.z[i] - mean : 0.0;
        project(m, (double)xi, (double)yi, Z, px, py);
    };
    // the minimum of the ROUNDED positions is the shift (grid_creator.py: shift_amount = int(rounded.min()))
    double mnx = INFINITY, mny = INFINITY;
    unsigned flags = 0;
    for (int i = tid; i < N; i += 256) {
        double px, py;
        smooth_of(i, px, py);
        if (isnan(px) || isnan(py)) flags |= VKX_GRID_STATE_NAN;
        else if (isinf(px) || isinf(py)) flags |= VKX_GRID_STATE_INF;
        mnx = fmin(mnx, rint(px));
        mny = fmin(mny, rint(py));
    }
    mnx = wave_min(mnx); mny = wave_min(mny);
    if (flags) atomicOr(&bad, flags);
    if ((tid & 63) == 0) { red[tid >> 6][0] = mnx; red[tid >> 6][1] = mny; }
    __syncthreads();
    mnx = fmin(fmin(red[0][0], red[1][0]), fmin(red[2][0], red[3][0]));
    mny = fmin(fmin(red[0][1], red[1][1]), fmin(red[2][1], red[3][1]));
    const bool finite = bad == 0;
    __syncthreads();
    // smooth + (-shift), rounded half to even; the lattice's extent is the result shape (ImageGrid)
    const double sx = finite ? -mnx : 0., sy = finite ? -mny : 0.;
    double mxx = -INFINITY, mxy = -INFINITY;
    for (int i = tid; i < N; i += 256) {
        int xi, yi;
        vertex_xy(i, xi, yi);
        st.sv[2 * i] = xi; st.sv[2 * i + 1] = yi;
        double px, py;
        smooth_of(i, px, py);
        const double qx = rint(px + sx), qy = rint(py + sy);
        mxx = fmax(mxx, qx); mxy = fmax(mxy, qy);
        if (finite) { st.dv[2 * i] = (int)qx; st.dv[2 * i + 1] = (int)qy; }
    }
    mxx = wave_max(mxx); mxy = wave_max(mxy);
    if ((tid & 63) == 0) { red[tid >> 6][2] = mxx; red[tid >> 6][3] = mxy; }
    __syncthreads();
    if (tid == 0) {
        mxx = fmax(fmax(red[0][2], red[1][2]), fmax(red[2][2], red[3][2]));
        mxy = fmax(fmax(red[0][3], red[1][3]), fmax(red[2][3], red[3][3]));
        vkx_grid_state g;
        g.rows = rows; g.cols = cols;
        // a lattice that does not fit the int32 vertices (or the chain's 32 767-pixel planes) is the caller's to refuse
        const bool fits = finite && mxx < 2147483647. && mxy < 2147483647. && fabs(mnx) < 2147483647. && fabs(mny) < 2147483647.;
        g.dh = fits ? (int)mxy + 1 : 0; g.dw = fits ? (int)mxx + 1 : 0;
        g.shift_y = fits ? (int)mny : 0; g.shift_x = fits ? (int)mnx : 0;
        g.flags = bad | (finite && !fits ? VKX_GRID_STATE_RANGE : 0u);
        g.reserved = 0;
        out[blockIdx.x] = g;
    }
}

// Python's float % for a positive divisor
double py_mod(double a, double b)
{
    double r = fmod(a, b);
    if (r != 0. && ((r < 0.) != (b < 0.))) r += b;
    return r;
}

double clip(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

int axis_ticks(int length, int grid_size) { const int n = (length + grid_size - 1) / grid_size; return ((n - 1) * grid_size != length - 1) ? n + 1 : n; }

} // namespace

VKX_EXPORT int vkx_camera_model_host(const vkx_camera_config *cfg, vkx_camera_model *out)
{
    VKX_REQUIRE(cfg && out, "NULL argument");
    VKX_REQUIRE(cfg->kind == VKX_CAMERA_PLANE_ONLY || cfg->kind == VKX_CAMERA_CUBIC_CURVE, "kind: VKX_CAMERA_PLANE_ONLY or VKX_CAMERA_CUBIC_CURVE");
    VKX_REQUIRE(cfg->height >= 1 && cfg->width >= 1 && cfg->grid_size >= 1, "bad shape / grid size");
    const int h = cfg->height, w = cfg->width;
    // complete_camera_model_config (camera.py:218-241): [height // 2, width // 2] is consumed as (x, y) -- sic, like the reference
    double pp_in[3] = {cfg->principal_point[0], cfg->principal_point[1], cfg->principal_point_len >= 3 ? cfg->principal_point[2] : 0.};
    double focal = cfg->focal_length, distance = cfg->camera_distance;
    const bool complete = cfg->principal_point_len >= 2 && focal != 0. && distance != 0.;
    if (!complete) {
        if (cfg->principal_point_len < 2) { pp_in[0] = h / 2; pp_in[1] = w / 2; pp_in[2] = 0.; }
        if (focal == 0. || distance == 0.) { focal = h > w ? h : w; distance = focal; }
    }
    // prep_rotation_unit_vec: float32, normalised by np.linalg.norm (sdot accumulates the float32 products in double)
    float u[3];
    for (int k = 0; k < 3; k++) u[k] = (float)cfg->rotation_unit_vec[k];
    const float sq = (float)(((double)(u[0] * u[0]) + (double)(u[1] * u[1])) + (double)(u[2] * u[2]));
    const float length = sqrtf(sq);
    if (length != 1.0f)
        for (int k = 0; k < 3; k++) u[k] = u[k] / length;
    const double theta = clip(cfg->rotation_theta, -89., 89.) / 180 * M_PI;
    float rvec[3];
    for (int k = 0; k < 3; k++) rvec[k] = u[k] * (float)theta;        // float32 array * Python float
    // cv.Rodrigues in float64 (camera.py rodrigues here)
    double R[9];
    {
        double rx = rvec[0], ry = rvec[1], rz = rvec[2];
        const double th = sqrt((rx * rx + ry * ry) + rz * rz);
        if (th < DBL_EPSILON) {
            for (int k = 0; k < 9; k++) R[k] = (k % 4 == 0) ? 1. : 0.;
        } else {
            const double c = cos(th), s = sin(th), c1 = 1.0 - c, ith = 1.0 / th;
            rx = rx * ith; ry = ry * ith; rz = rz * ith;
            const double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
            const double r_x[9] = {0., -rz, ry, rz, 0., -rx, -ry, rx, 0.};
            for (int k = 0; k < 9; k++) R[k] = (c * ((k % 4 == 0) ? 1. : 0.) + c1 * rrt[k]) + s * r_x[k];
        }
    }
    // generate_translation_vec in float32: t = R (R^T [0, 0, d]^T - pp)
    float R32[9], pp[3], origin[3], shifted[3], t32[3];
    for (int k = 0; k < 9; k++) R32[k] = (float)R[k];
    for (int k = 0; k < 3; k++) pp[k] = (float)pp_in[k];
    const float d32 = (float)distance;
    for (int i = 0; i < 3; i++) origin[i] = fmaf(R32[6 + i], d32, fmaf(R32[3 + i], 0.f, R32[i] * 0.f));
    for (int i = 0; i < 3; i++) shifted[i] = origin[i] - pp[i];
    for (int i = 0; i < 2; i++) t32[i] = (R32[3 * i] * shifted[0] + R32[3 * i + 1] * shifted[1]) + R32[3 * i + 2] * shifted[2];
    t32[2] = fmaf(R32[8], shifted[2], fmaf(R32[6], shifted[0], R32[7] * shifted[1]));
    for (int k = 0; k < 9; k++) out->R[k] = R[k];
    for (int k = 0; k < 3; k++) out->t[k] = (double)t32[k];
    out->fx = out->fy = (double)(float)focal;
    out->cx = out->cy = 0.;
    out->rows = axis_ticks(h, cfg->grid_size);
    out->cols = axis_ticks(w, cfg->grid_size);
    out->reserved = 0;
    out->a0 = out->a1 = out->along_min = out->along_range = 0.f;
    out->poly[0] = out->poly[1] = out->poly[2] = out->poly[3] = 0.;
    out->curve_scale = 0.;
    out->points_f32 = cfg->kind == VKX_CAMERA_PLANE_ONLY;     // float32 points in, float32 projections out
    if (cfg->kind == VKX_CAMERA_CUBIC_CURVE) {
        const double alpha = tan(clip(cfg->curve_alpha, -80., 80.) / 180 * M_PI);
        const double beta = tan(clip(cfg->curve_beta, -80., 80.) / 180 * M_PI);
        const double direction = py_mod(cfg->curve_direction, 180.) / 180 * M_PI;
        out->a0 = (float)cos(direction);
        out->a1 = (float)sin(direction);
        const float cx[4] = {0.f, (float)(w - 1), (float)(w - 1), 0.f}, cy[4] = {0.f, 0.f, (float)(h - 1), (float)(h - 1)};
        float lo = INFINITY, hi = -INFINITY;
        for (int j = 0; j < 4; j++) {
            const float along = fmaf(out->a1, cy[j], out->a0 * cx[j]);
            lo = fminf(lo, along); hi = fmaxf(hi, along);
        }
        out->along_min = lo;
        out->along_range = hi - lo;
        out->poly[0] = alpha + beta;
        out->poly[1] = -2 * alpha - beta;
        out->poly[2] = alpha;
        out->poly[3] = 0.;
        out->curve_scale = cfg->curve_scale;
    }
    return VKX_OK;
}

// The lattices of n states: scalars on the host, vertices on the device, one workgroup per state, on the stream `stream`
// (VKX_STREAM_COMPUTE, or one of the side streams so that the states of the next batch are built while the compute stream still
// runs the current one).  states_host (page-locked for an asynchronous copy) is valid after that stream has been synchronised.
// Records the context's lattices-ready point (vkx_chain_lattices_ready) on that stream.
VKX_EXPORT int vkx_camera_states_dev(vkx_ctx *ctx, const vkx_camera_config *configs, int n, int32_t *const *src_vertices,
                                     int32_t *const *dst_vertices, vkx_grid_state *states_host, int stream)
{
    VKX_REQUIRE(ctx && configs && src_vertices && dst_vertices && states_host, "NULL argument");
    VKX_REQUIRE(n >= 1 && n <= 65535, "1 .. 65535 states per call");
    int rc;
    hipStream_t st = vkx_stream_by_id(ctx, stream, &rc);
    if (rc) return rc;
    std::vector<StateDev> host((size_t)n);
    size_t z_total = 0;
    for (int i = 0; i < n; i++) {
        StateDev &s = host[i];
        if ((rc = vkx_camera_model_host(&configs[i], &s.m))) return rc;
        VKX_REQUIRE(src_vertices[i] && dst_vertices[i], "NULL lattice");
        VKX_REQUIRE((long long)s.m.rows * s.m.cols <= (1 << 24), "lattice beyond 2^24 vertices");
        s.sv = src_vertices[i]; s.dv = dst_vertices[i];
        s.height = configs[i].height; s.width = configs[i].width; s.grid_size = configs[i].grid_size; s.kind = configs[i].kind;
        s.z = (double *)(z_total * sizeof(double));     // offset for now
        if (s.kind == VKX_CAMERA_CUBIC_CURVE) z_total += (size_t)s.m.rows * s.m.cols;
    }
    const size_t desc_bytes = ((size_t)n * sizeof(StateDev) + 255) & ~(size_t)255, out_bytes = ((size_t)n * sizeof(vkx_grid_state) + 255) & ~(size_t)255;
    // (grown with every stream of the context drained: vkx_scratch_reserve)
    if ((rc = vkx_scratch_reserve(ctx, &ctx->camera_work, desc_bytes + out_bytes + z_total * sizeof(double)))) return rc;
    unsigned char *base = (unsigned char *)ctx->camera_work.ptr;
    for (int i = 0; i < n; i++) host[i].z = (double *)(base + desc_bytes + out_bytes) + (size_t)(uintptr_t)host[i].z / sizeof(double);
    void *ring = nullptr;
    if ((rc = vkx_desc_ring_take(ctx, (size_t)n * sizeof(StateDev), &ring))) return rc;
    memcpy(ring, host.data(), (size_t)n * sizeof(StateDev));
    vkx_device_guard guard(ctx);
    hipStream_t main_stream = ctx->stream;
    // the previous call's states may still be read from this scratch by ITS kernel only (same stream order when the caller keeps to
    // one stream; a caller that alternates streams synchronises in between, as ChainBatch does)
    ctx->stream = st;
    rc = vkx_small_to_device(ctx, base, ring, (size_t)n * sizeof(StateDev));
    if (!rc) {
        VKX_TIMED(ctx, "k_camera_states");
        k_camera_states<<<n, 256, 0, st>>>((const StateDev *)base, (vkx_grid_state *)(base + desc_bytes));
    }
    if (!rc && hipGetLastError() != hipSuccess) { vkx_set_error("kernel launch failed in %s", __func__); rc = VKX_ERR_HIP; }
    if (!rc) rc = vkx_small_to_host(ctx, states_host, base + desc_bytes, (size_t)n * sizeof(vkx_grid_state));
    ctx->stream = main_stream;
    if (rc) return rc;
    if (!ctx->lattices_ready) VKX_HIP(hipEventCreateWithFlags(&ctx->lattices_ready, hipEventDisableTiming));
    VKX_HIP(hipEventRecord(ctx->lattices_ready, st));
    ctx->lattices_armed = true;
    return VKX_OK;
}
