// Per-cell device code shared by grid.hip and fused.hip: the dst->src homography of one lattice cell and the
// cv.fillPoly edge table of its destination quad.  Specification and reference citations:
// oracle/vkx_oracle.c (homography_direct / homography_jacobi, vko_fill_poly_closed_form).
#pragma once
#include "vkx_internal.h"

#include <float.h>

namespace vkc {

// den * (unit square -> quad) as an exact-integer matrix (Heckbert's square-to-quad construction).
__device__ inline void square_to_quad_scaled(const double q[8], double G[9])
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

__device__ inline bool quad_in_general_position(const double q[8])
{
    for (int a = 0; a < 4; a++) {
        const int b = (a + 1) & 3, c = (a + 2) & 3;
        const double cr = (q[2 * b] - q[2 * a]) * (q[2 * c + 1] - q[2 * a + 1]) -
                          (q[2 * b + 1] - q[2 * a + 1]) * (q[2 * c] - q[2 * a]);
        if (cr == 0) return false;
    }
    return true;
}

// Closed-form quad -> quad homography (no three collinear vertices on either side).
__device__ inline bool homography_direct(const double qf[8], const double qt[8], double H[9])
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

__device__ inline double vk_hypot(double a, double b)
{
    a = fabs(a); b = fabs(b);
    if (a < b) { const double t = a; a = b; b = t; }
    if (a == 0) return 0;
    const double r = b / a;
    return a * sqrt(1 + r * r);
}

// Minimum-norm least squares of the 8x8 DLT system: one-sided Jacobi SVD + back substitution with the
// 2*eps*sum(w) cut-off.  Only reached for degenerate quads (a handful per image), written for clarity.
// `ws`: kJacobiWs doubles of private workspace (thread-private scratch by default; the batched setup kernel
// hands in LDS so that its launch needs no scratch segment at all).
constexpr int kJacobiWs = 8 * 8 * 2 + 8 * 3;
__device__ __forceinline__ void homography_jacobi_ws(const float from[8], const float to[8], double H[9], double *ws)
{
    double (*At)[8] = (double (*)[8])ws;
    double (*Vt)[8] = (double (*)[8])(ws + 64);
    double *W = ws + 128, *b = ws + 136, *x = ws + 144;
    for (int i = 0; i < 4; i++) {
        const float fx = from[2 * i], fy = from[2 * i + 1], tx = to[2 * i], ty = to[2 * i + 1];
        for (int c = 0; c < 8; c++) { At[c][i] = 0; At[c][i + 4] = 0; }
        At[0][i] = fx; At[1][i] = fy; At[2][i] = 1;
        At[3][i + 4] = fx; At[4][i + 4] = fy; At[5][i + 4] = 1;
        At[6][i] = (double)(-fx * tx);      // float products, as cv::Point2f arithmetic forms them
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

// The same solver run by EIGHT lanes per system (lane k of the group holds column k of A^T and of V^T in registers; the
// singular values W are replicated): rotations update their sixteen elements in parallel, and every sum the scalar code
// accumulates over k is accumulated in the same order from the same 0.0 -- the eight addends are fetched from the group's
// lanes and added one after the other -- so the result is the scalar solver's bit for bit.  The pair loop is unrolled: no
// indexed register, no LDS workspace (the one-lane form keeps 152 doubles in LDS and pays its latency on every access).
// All eight lanes of a group must be active; every lane returns the whole H.
__device__ __forceinline__ double group_sum8(double t)
{
    double s = 0.0;                          // the scalar loop's order: 0 + t[0] + t[1] + ... + t[7]
#pragma unroll
    for (int k = 0; k < 8; k++) s += __shfl(t, k, 8);
    return s;
}
__device__ inline void homography_jacobi_group8(const float from[8], const float to[8], double H[9])
{
    const int k = (int)(threadIdx.x & 7);          // this lane's column (equation)
    const int pt = k & 3;
    const float fx = from[2 * pt], fy = from[2 * pt + 1], tx = to[2 * pt], ty = to[2 * pt + 1];
    const bool low = k < 4;
    double At[8], Vt[8], W[8];
    At[0] = low ? fx : 0; At[1] = low ? fy : 0; At[2] = low ? 1 : 0;
    At[3] = low ? 0 : fx; At[4] = low ? 0 : fy; At[5] = low ? 0 : 1;
    At[6] = low ? (double)(-fx * tx) : (double)(-fx * ty);      // float products, as cv::Point2f arithmetic forms them
    At[7] = low ? (double)(-fy * tx) : (double)(-fy * ty);
    const double b = low ? tx : ty;
    const double eps = DBL_EPSILON * 10, minval = DBL_MIN;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        W[i] = group_sum8(At[i] * At[i]);
        Vt[i] = i == k ? 1 : 0;
    }
    bool active = true;
    for (int iter = 0; iter < 30; iter++) {
        if (active) {
            bool changed = false;
#pragma unroll
            for (int i = 0; i < 7; i++)
#pragma unroll
                for (int j = i + 1; j < 8; j++) {
                    double a = W[i], bb = W[j];
                    double p = group_sum8(At[i] * At[j]);
                    if (!(fabs(p) <= eps * sqrt(a * bb))) {
                        double c, sn;
                        p *= 2;
                        const double beta = a - bb, gamma = vk_hypot(p, beta);
                        if (beta < 0) {
                            const double delta = (gamma - beta) * 0.5;
                            sn = sqrt(delta / gamma);
                            c = p / (gamma * sn * 2);
                        } else {
                            c = sqrt((gamma + beta) / (gamma * 2));
                            sn = p / (gamma * c * 2);
                        }
                        const double t0 = c * At[i] + sn * At[j];
                        const double t1 = -sn * At[i] + c * At[j];
                        At[i] = t0; At[j] = t1;
                        W[i] = group_sum8(t0 * t0);
                        W[j] = group_sum8(t1 * t1);
                        changed = true;
                        const double v0 = c * Vt[i] + sn * Vt[j];
                        const double v1 = -sn * Vt[i] + c * Vt[j];
                        Vt[i] = v0; Vt[j] = v1;
                    }
                }
            if (!changed) active = false;
        }
        if (!__any(active)) break;
    }
#pragma unroll
    for (int i = 0; i < 8; i++) W[i] = sqrt(group_sum8(At[i] * At[i]));
    // selection sort, descending (strict <, first maximum wins): the row swaps as predicated exchanges -- the rows are
    // registers, so the value of W[j] travels beside the runtime index j
#pragma unroll
    for (int i = 0; i < 7; i++) {
        int j = i;
        double wj = W[i];
#pragma unroll
        for (int q = i + 1; q < 8; q++)
            if (wj < W[q]) { j = q; wj = W[q]; }
#pragma unroll
        for (int q = i + 1; q < 8; q++)
            if (j == q) {
                double t = W[i]; W[i] = W[q]; W[q] = t;
                t = At[i]; At[i] = At[q]; At[q] = t;
                t = Vt[i]; Vt[i] = Vt[q]; Vt[q] = t;
            }
    }
    double threshold = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const double sd = W[i];
        const double s = sd > minval ? 1 / sd : 0.;
        At[i] *= s;
        threshold += W[i];
    }
    threshold *= DBL_EPSILON * 2;
    double x = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        double wi = W[i];
        if (fabs(wi) <= threshold) continue;
        wi = 1 / wi;
        double acc = group_sum8(At[i] * b);
        acc *= wi;
        x = x + acc * Vt[i];
    }
#pragma unroll
    for (int i = 0; i < 8; i++) H[i] = __shfl(x, i, 8);
    H[8] = 1.;
}

__device__ __noinline__ void homography_jacobi(const float from[8], const float to[8], double H[9])
{
    double ws[kJacobiWs];
    homography_jacobi_ws(from, to, H, ws);
}

// Number of minor-axis steps an 8-connected Bresenham line (cv::LineIterator, walked from its left end) has
// taken after k major steps: ceil((2 k dmin - dmaj) / (2 dmaj)), never negative.
__device__ __forceinline__ int bres_minor(int k, int dmaj, int dmin)
{
    if (dmaj == 0) return 0;
    const long long num = 2LL * k * dmin - dmaj;
    if (num <= 0) return 0;
    return (int)((num + 2LL * dmaj - 1) / (2LL * dmaj));
}

// Compact per-cell record (128 B): everything a tile needs to rasterise the cell and to map its pixels.
// Coordinates are limited to [0, 32767] (the same limit cv.remap imposes through its int16 coordinates).
struct CellC {
    double H[8];       // row-major inverse homography, H[8] == 1 implied
    int ex[4];         // 16.16 x of edge i (vertex (i+3)&3 -> vertex i) at its upper end
    int edx[4];        // 16.16 dx per scanline (C division of (dX << 16) by dY); 0 for horizontal edges
    short vx[4], vy[4];
    int flags;         // bit 0: the projective denominator may vanish inside the bounding box
    int pad[3];
};
static_assert(sizeof(CellC) == 128, "CellC layout");

// Builds the record of cell `cell` (row-major index over (rows-1) x (cols-1)) and returns its bounding box.
// MODE kCellDirectOnly: the closed-form solver only; returns false (record incomplete) when the cell's quads are
// not in general position, so that a caller can leave the register-hungry SVD to a second, rarely busy kernel.
enum { kCellDirectOnly = 0, kCellJacobiPrivate = 1, kCellJacobiWorkspace = 2, kCellJacobiGroup8 = 3 };
template <int MODE = kCellJacobiPrivate>
__device__ inline bool build_cell(const int32_t *__restrict__ src_v, const int32_t *__restrict__ dst_v, int rows, int cols,
                                  int cell, CellC &rec, int &xmin, int &xmax, int &ymin, int &ymax,
                                  double *jacobi_ws = nullptr)
{
    (void)rows;
    const int r = cell / (cols - 1), c = cell - r * (cols - 1);
    const int idx[4] = {r * cols + c, r * cols + c + 1, (r + 1) * cols + c + 1, (r + 1) * cols + c};
    float from[8], to[8];
    double qf[8], qt[8], H[9];
    int vx[4], vy[4];
    xmin = INT_MAX; xmax = INT_MIN; ymin = INT_MAX; ymax = INT_MIN;
    for (int k = 0; k < 4; k++) {
        vx[k] = dst_v[2 * idx[k]]; vy[k] = dst_v[2 * idx[k] + 1];
        from[2 * k] = (float)vx[k]; from[2 * k + 1] = (float)vy[k];
        to[2 * k] = (float)src_v[2 * idx[k]]; to[2 * k + 1] = (float)src_v[2 * idx[k] + 1];
        qf[2 * k] = from[2 * k]; qf[2 * k + 1] = from[2 * k + 1];
        qt[2 * k] = to[2 * k]; qt[2 * k + 1] = to[2 * k + 1];
        xmin = min(xmin, vx[k]); xmax = max(xmax, vx[k]);
        ymin = min(ymin, vy[k]); ymax = max(ymax, vy[k]);
        rec.vx[k] = (short)vx[k]; rec.vy[k] = (short)vy[k];
    }
    if (!homography_direct(qf, qt, H)) {
        if (MODE == kCellDirectOnly) return false;
        if (MODE == kCellJacobiGroup8) homography_jacobi_group8(from, to, H);      // all 8 lanes of the group are here
        else if (MODE == kCellJacobiWorkspace) homography_jacobi_ws(from, to, H, jacobi_ws);
        else homography_jacobi(from, to, H);
    }
    for (int i = 0; i < 8; i++) rec.H[i] = H[i];
    for (int i = 0; i < 4; i++) {
        const int a = (i + 3) & 3;
        const long long xa = (long long)vx[a] << 16, xb = (long long)vx[i] << 16;
        const int ya = vy[a], yb = vy[i];
        if (ya == yb) { rec.ex[i] = 0; rec.edx[i] = 0; continue; }
        rec.edx[i] = (int)((xb - xa) / (long long)(yb - ya));
        rec.ex[i] = (int)(ya < yb ? xa : xb);
    }
    // The projective denominator is affine in (x, y): its extrema over the bounding box sit at the corners.
    double dmin = DBL_MAX, dmax = -DBL_MAX;
    const int cx[2] = {xmin, xmax}, cy[2] = {ymin, ymax};
    for (int a = 0; a < 2; a++)
        for (int b = 0; b < 2; b++) {
            const double d = H[6] * cx[a] + H[7] * cy[b] + 1.0;
            dmin = fmin(dmin, d); dmax = fmax(dmax, d);
        }
    rec.flags = (dmin > 1e-6 || dmax < -1e-6) && isfinite(dmin) && isfinite(dmax) ? 0 : 1;
    rec.pad[0] = rec.pad[1] = rec.pad[2] = 0;
    return true;
}

} // namespace vkc
