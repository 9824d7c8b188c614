/*
 * vkx_oracle.c -- CPU restatement of the arithmetic on vkit's distortion hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product (vkit_amd + libvkx.so) never
 * links, imports or executes anything in oracle/.
 *
 * Parity status: *unpinned at the cv2 boundary*.  The reference (/root/reference, pure
 * Python) delegates the arithmetic below to OpenCV (opencv-python-headless >=4.5.1.48,
 * setup.cfg:20-22), which is neither vendored in the reference nor installed in this
 * image.  Functions marked [cv2] restate the published OpenCV 4.5.x algorithm; functions
 * marked [numpy] restate numpy arithmetic that the reference runs itself and ARE pinned
 * against golden vectors produced by importing the reference (tests/golden/).
 *
 * Every function cites the reference call site it stands in for (paths relative to
 * /root/reference/vkit).
 *
 * Plain scalar C99, one thread, no SIMD, no FMA contraction (build with
 * -ffp-contract=off); explicit fma() only where the reference's BLAS uses it.
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <limits.h>
#include <float.h>
#include <stdio.h>

#define VKO_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------
 * Rounding helpers: OpenCV cvRound on x86 = cvtss2si / cvtsd2si (round-half-even, the
 * "integer indefinite" 0x80000000 for NaN / out-of-range).
 * ---------------------------------------------------------------------------------- */
static inline int cv_round_d(double v)
{
    if (!(v >= -2147483648.5 && v < 2147483647.5)) return INT_MIN;
    return (int)nearbyint(v);
}
static inline int cv_round_f(float v) { return cv_round_d((double)v); }
static inline int sat_short(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }
static inline uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

/* ------------------------------------------------------------------------------------
 * [cv2] Bilinear interpolation tables of cv::remap (imgwarp.cpp initInterTab2D):
 * INTER_BITS=5, INTER_TAB_SIZE=32, INTER_REMAP_COEF_BITS=15.  The int16 table is the
 * float table * 32768 with saturate_cast<short>, followed by the "sum must be 32768"
 * fix-up; the only entry it changes is (fy,fx)=(0,0): {32767,0,0,1}.
 * ---------------------------------------------------------------------------------- */
static short g_tab_i[32 * 32][4];
static float g_tab_f[32 * 32][4];
static int g_tab_ready = 0;

static void init_tabs(void)
{
    if (g_tab_ready) return;
    float t1[32][2];
    const float scale = 1.f / 32;
    for (int i = 0; i < 32; i++) {
        float x = i * scale;
        t1[i][0] = 1.f - x;
        t1[i][1] = x;
    }
    for (int i = 0; i < 32; i++)
        for (int j = 0; j < 32; j++) {
            short *it = g_tab_i[i * 32 + j];
            float *ft = g_tab_f[i * 32 + j];
            int isum = 0;
            for (int k1 = 0; k1 < 2; k1++)
                for (int k2 = 0; k2 < 2; k2++) {
                    float v = t1[i][k1] * t1[j][k2];
                    ft[k1 * 2 + k2] = v;
                    it[k1 * 2 + k2] = (short)sat_short(cv_round_f(v * 32768));
                    isum += it[k1 * 2 + k2];
                }
            if (isum != 32768) {
                /* OpenCV scans a 2x2 window anchored at (ksize/2,ksize/2)=(1,1); for the
                 * bilinear 2x2 kernel only element [1][1] lies inside the entry, the other
                 * three probes read not-yet-written (zero) slots of the next entry. */
                int diff = isum - 32768;
                int probe[4] = { it[3], 0, 0, 0 };
                int mk = 0, Mk = 0;
                for (int k = 0; k < 4; k++) {
                    if (probe[k] < probe[mk]) mk = k;
                    else if (probe[k] > probe[Mk]) Mk = k;
                }
                /* mk == Mk == 0 -> element [1][1] */
                (void)mk; (void)Mk;
                it[3] = (short)(it[3] - diff);
            }
        }
    g_tab_ready = 1;
}

/* One destination pixel of remapBilinear, BORDER_CONSTANT(0).  X,Y: source coordinate in
 * 1/32 px fixed point (what remap/warpAffine/warpPerspective hand to the interpolator). */
static inline void px_u8(const uint8_t *src, int sh, int sw, int cn, ptrdiff_t sstep,
                         int X, int Y, uint8_t *d)
{
    int sx = sat_short(X >> 5), sy = sat_short(Y >> 5);
    const short *w = g_tab_i[(Y & 31) * 32 + (X & 31)];
    if (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0) {
        for (int k = 0; k < cn; k++) d[k] = 0;
        return;
    }
    int x0 = sx >= 0 && sx < sw, x1 = sx + 1 >= 0 && sx + 1 < sw;
    int y0 = sy >= 0 && sy < sh, y1 = sy + 1 >= 0 && sy + 1 < sh;
    for (int k = 0; k < cn; k++) {
        int v0 = (x0 && y0) ? src[(ptrdiff_t)sy * sstep + sx * cn + k] : 0;
        int v1 = (x1 && y0) ? src[(ptrdiff_t)sy * sstep + (sx + 1) * cn + k] : 0;
        int v2 = (x0 && y1) ? src[(ptrdiff_t)(sy + 1) * sstep + sx * cn + k] : 0;
        int v3 = (x1 && y1) ? src[(ptrdiff_t)(sy + 1) * sstep + (sx + 1) * cn + k] : 0;
        d[k] = sat_u8((v0 * w[0] + v1 * w[1] + v2 * w[2] + v3 * w[3] + (1 << 14)) >> 15);
    }
}

static inline float px_f32(const float *src, int sh, int sw, ptrdiff_t sstep_el, int X, int Y)
{
    int sx = sat_short(X >> 5), sy = sat_short(Y >> 5);
    const float *w = g_tab_f[(Y & 31) * 32 + (X & 31)];
    if (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0) return 0.f;
    int x0 = sx >= 0 && sx < sw, x1 = sx + 1 >= 0 && sx + 1 < sw;
    int y0 = sy >= 0 && sy < sh, y1 = sy + 1 >= 0 && sy + 1 < sh;
    float v0 = (x0 && y0) ? src[(ptrdiff_t)sy * sstep_el + sx] : 0.f;
    float v1 = (x1 && y0) ? src[(ptrdiff_t)sy * sstep_el + sx + 1] : 0.f;
    float v2 = (x0 && y1) ? src[(ptrdiff_t)(sy + 1) * sstep_el + sx] : 0.f;
    float v3 = (x1 && y1) ? src[(ptrdiff_t)(sy + 1) * sstep_el + sx + 1] : 0.f;
    float p0 = v0 * w[0], p1 = v1 * w[1], p2 = v2 * w[2], p3 = v3 * w[3];
    return ((p0 + p1) + p2) + p3;
}

/* ------------------------------------------------------------------------------------
 * [cv2] cv.remap(src, map_x, map_y, INTER_LINEAR) -- grid_rendering/grid_blender.py:60,70,80
 * ---------------------------------------------------------------------------------- */
VKO_API int vko_remap_u8(const uint8_t *src, int sh, int sw, int cn, ptrdiff_t sstep,
                         const float *mapx, const float *mapy, ptrdiff_t mstep_el,
                         uint8_t *dst, int dh, int dw, ptrdiff_t dstep)
{
    init_tabs();
    for (int y = 0; y < dh; y++)
        for (int x = 0; x < dw; x++) {
            int X = cv_round_f(mapx[(ptrdiff_t)y * mstep_el + x] * 32);
            int Y = cv_round_f(mapy[(ptrdiff_t)y * mstep_el + x] * 32);
            px_u8(src, sh, sw, cn, sstep, X, Y, dst + (ptrdiff_t)y * dstep + x * cn);
        }
    return 0;
}

VKO_API int vko_remap_f32(const float *src, int sh, int sw, ptrdiff_t sstep_el,
                          const float *mapx, const float *mapy, ptrdiff_t mstep_el,
                          float *dst, int dh, int dw, ptrdiff_t dstep_el)
{
    init_tabs();
    for (int y = 0; y < dh; y++)
        for (int x = 0; x < dw; x++) {
            int X = cv_round_f(mapx[(ptrdiff_t)y * mstep_el + x] * 32);
            int Y = cv_round_f(mapy[(ptrdiff_t)y * mstep_el + x] * 32);
            dst[(ptrdiff_t)y * dstep_el + x] = px_f32(src, sh, sw, sstep_el, X, Y);
        }
    return 0;
}

/* ------------------------------------------------------------------------------------
 * [cv2] cv.warpAffine(mat, trans_mat(2x3 forward), dsize) -- geometric/affine.py:38-43
 * imgwarp.cpp: M -> double, inverted in double, AB_BITS=10 fixed-point row/col deltas.
 * Writes the 1/32-px fixed point coordinates of every destination pixel.
 * ---------------------------------------------------------------------------------- */
static void affine_invert(const double Mf[6], double M[6])
{
    memcpy(M, Mf, sizeof(double) * 6);
    double D = M[0] * M[4] - M[1] * M[3];
    D = D != 0 ? 1. / D : 0;
    double A11 = M[4] * D, A22 = M[0] * D;
    M[0] = A11; M[1] *= -D;
    M[3] *= -D; M[4] = A22;
    double b1 = -M[0] * M[2] - M[1] * M[5];
    double b2 = -M[3] * M[2] - M[4] * M[5];
    M[2] = b1; M[5] = b2;
}

VKO_API int vko_warp_affine_coords(const double Mfwd[6], int dh, int dw, int *X, int *Y)
{
    double M[6];
    affine_invert(Mfwd, M);
    const int AB_SCALE = 1 << 10, round_delta = 16;
    int *adelta = (int *)malloc(sizeof(int) * 2 * (size_t)dw), *bdelta = adelta + dw;
    if (!adelta) return -1;
    for (int x = 0; x < dw; x++) {
        adelta[x] = cv_round_d(M[0] * x * AB_SCALE);
        bdelta[x] = cv_round_d(M[3] * x * AB_SCALE);
    }
    for (int y = 0; y < dh; y++) {
        int X0 = cv_round_d((M[1] * y + M[2]) * AB_SCALE) + round_delta;
        int Y0 = cv_round_d((M[4] * y + M[5]) * AB_SCALE) + round_delta;
        for (int x = 0; x < dw; x++) {
            X[(size_t)y * dw + x] = (X0 + adelta[x]) >> 5;
            Y[(size_t)y * dw + x] = (Y0 + bdelta[x]) >> 5;
        }
    }
    free(adelta);
    return 0;
}

/* [cv2] cv.warpPerspective(mat, trans_mat(3x3 forward), dsize) -- geometric/affine.py:43.
 * cv::invert 3x3 (cofactors * 1/det), then per pixel in double with the 32x32 block
 * decomposition of WarpPerspectiveInvoker (X0 is evaluated at the block's first column). */
static int invert3(const double *S, double *t)
{
#define Sd(r, c) S[(r) * 3 + (c)]
    double d = Sd(0, 0) * (Sd(1, 1) * Sd(2, 2) - Sd(1, 2) * Sd(2, 1)) -
               Sd(0, 1) * (Sd(1, 0) * Sd(2, 2) - Sd(1, 2) * Sd(2, 0)) +
               Sd(0, 2) * (Sd(1, 0) * Sd(2, 1) - Sd(1, 1) * Sd(2, 0));
    if (d == 0.) return 0;
    d = 1. / d;
    t[0] = (Sd(1, 1) * Sd(2, 2) - Sd(1, 2) * Sd(2, 1)) * d;
    t[1] = (Sd(0, 2) * Sd(2, 1) - Sd(0, 1) * Sd(2, 2)) * d;
    t[2] = (Sd(0, 1) * Sd(1, 2) - Sd(0, 2) * Sd(1, 1)) * d;
    t[3] = (Sd(1, 2) * Sd(2, 0) - Sd(1, 0) * Sd(2, 2)) * d;
    t[4] = (Sd(0, 0) * Sd(2, 2) - Sd(0, 2) * Sd(2, 0)) * d;
    t[5] = (Sd(0, 2) * Sd(1, 0) - Sd(0, 0) * Sd(1, 2)) * d;
    t[6] = (Sd(1, 0) * Sd(2, 1) - Sd(1, 1) * Sd(2, 0)) * d;
    t[7] = (Sd(0, 1) * Sd(2, 0) - Sd(0, 0) * Sd(2, 1)) * d;
    t[8] = (Sd(0, 0) * Sd(1, 1) - Sd(0, 1) * Sd(1, 0)) * d;
#undef Sd
    return 1;
}

VKO_API int vko_warp_perspective_coords(const double Mfwd[9], int dh, int dw, int *X, int *Y)
{
    double M[9];
    if (!invert3(Mfwd, M)) memset(M, 0, sizeof M); /* cv::invert leaves zeros on failure */
    const int BLOCK_SZ = 32;
    int bh0 = BLOCK_SZ / 2 < dh ? BLOCK_SZ / 2 : dh;
    int bw0 = BLOCK_SZ * BLOCK_SZ / bh0 < dw ? BLOCK_SZ * BLOCK_SZ / bh0 : dw;
    for (int y = 0; y < dh; y++)
        for (int xb = 0; xb < dw; xb += bw0) {
            int bw = bw0 < dw - xb ? bw0 : dw - xb;
            double X0 = M[0] * xb + M[1] * y + M[2];
            double Y0 = M[3] * xb + M[4] * y + M[5];
            double W0 = M[6] * xb + M[7] * y + M[8];
            for (int x1 = 0; x1 < bw; x1++) {
                double W = W0 + M[6] * x1;
                W = W ? 32 / W : 0;
                double fX = fmax((double)INT_MIN, fmin((double)INT_MAX, (X0 + M[0] * x1) * W));
                double fY = fmax((double)INT_MIN, fmin((double)INT_MAX, (Y0 + M[3] * x1) * W));
                X[(size_t)y * dw + xb + x1] = cv_round_d(fX);
                Y[(size_t)y * dw + xb + x1] = cv_round_d(fY);
            }
        }
    return 0;
}

/* Interpolate with precomputed fixed-point coordinates (tail of warpAffine/Perspective). */
VKO_API int vko_sample_fixed_u8(const uint8_t *src, int sh, int sw, int cn, ptrdiff_t sstep,
                                const int *X, const int *Y, uint8_t *dst, int dh, int dw,
                                ptrdiff_t dstep)
{
    init_tabs();
    for (int y = 0; y < dh; y++)
        for (int x = 0; x < dw; x++)
            px_u8(src, sh, sw, cn, sstep, X[(size_t)y * dw + x], Y[(size_t)y * dw + x],
                  dst + (ptrdiff_t)y * dstep + x * cn);
    return 0;
}

VKO_API int vko_sample_fixed_f32(const float *src, int sh, int sw, ptrdiff_t sstep_el,
                                 const int *X, const int *Y, float *dst, int dh, int dw,
                                 ptrdiff_t dstep_el)
{
    init_tabs();
    for (int y = 0; y < dh; y++)
        for (int x = 0; x < dw; x++)
            dst[(ptrdiff_t)y * dstep_el + x] =
                px_f32(src, sh, sw, sstep_el, X[(size_t)y * dw + x], Y[(size_t)y * dw + x]);
    return 0;
}

/* ------------------------------------------------------------------------------------
 * [cv2] cv.getPerspectiveTransform(from(4x2 f32), to(4x2 f32), DECOMP_SVD)
 *   grid_rendering/type.py:172-176,189-193; geometric/affine.py:326-330,386-390.
 *
 * Two solvers:
 *   solver 1 (JACOBI): restatement of OpenCV's built-in path: 8x8 DLT system in double
 *     (the -x*X products are formed in FLOAT, as Point2f arithmetic does), one-sided
 *     Jacobi SVD (lapack.cpp JacobiSVDImpl_) and SVBkSb back substitution with the
 *     2*DBL_EPSILON*sum(w) cut-off -> minimum-norm least squares for singular systems.
 *   solver 0 (HYBRID, the parity definition used by the product): a closed-form
 *     quad->quad homography built from exact integer sub-determinants when no three
 *     vertices of either quad are collinear, solver 1 otherwise.
 * cv2 wheels are built against LAPACK (dgesdd), so even solver 1 is not bit-identical to a
 * given cv2 binary; all accurate solvers agree to ~1e-9 px on the maps they induce.
 * ---------------------------------------------------------------------------------- */
static double vk_hypot(double a, double b)
{
    a = fabs(a); b = fabs(b);
    if (a < b) { double t = a; a = b; b = t; }
    if (a == 0) return 0;
    double r = b / a;
    return a * sqrt(1 + r * r);
}

static void jacobi_svd8(double At[8][8], double W[8], double Vt[8][8])
{
    const int m = 8, n = 8;
    const double eps = DBL_EPSILON * 10, minval = DBL_MIN;
    int i, j, k, iter, max_iter = 30 > m ? 30 : m;
    double c, s, sd;
    for (i = 0; i < n; i++) {
        for (k = 0, sd = 0; k < m; k++) { double t = At[i][k]; sd += t * t; }
        W[i] = sd;
        for (k = 0; k < n; k++) Vt[i][k] = 0;
        Vt[i][i] = 1;
    }
    for (iter = 0; iter < max_iter; iter++) {
        int changed = 0;
        for (i = 0; i < n - 1; i++)
            for (j = i + 1; j < n; j++) {
                double *Ai = At[i], *Aj = At[j];
                double a = W[i], p = 0, b = W[j];
                for (k = 0; k < m; k++) p += Ai[k] * Aj[k];
                if (fabs(p) <= eps * sqrt(a * b)) continue;
                p *= 2;
                /* OpenCV calls libm hypot(p, beta); evaluated here with the scaled form below (a few
                 * ulp apart at most) so that the CPU and GPU statements agree bit for bit. */
                double beta = a - b, gamma = vk_hypot(p, beta);
                if (beta < 0) {
                    double delta = (gamma - beta) * 0.5;
                    s = sqrt(delta / gamma);
                    c = p / (gamma * s * 2);
                } else {
                    c = sqrt((gamma + beta) / (gamma * 2));
                    s = p / (gamma * c * 2);
                }
                a = b = 0;
                for (k = 0; k < m; k++) {
                    double t0 = c * Ai[k] + s * Aj[k];
                    double t1 = -s * Ai[k] + c * Aj[k];
                    Ai[k] = t0; Aj[k] = t1;
                    a += t0 * t0; b += t1 * t1;
                }
                W[i] = a; W[j] = b;
                changed = 1;
                double *Vi = Vt[i], *Vj = Vt[j];
                for (k = 0; k < n; k++) {
                    double t0 = c * Vi[k] + s * Vj[k];
                    double t1 = -s * Vi[k] + c * Vj[k];
                    Vi[k] = t0; Vj[k] = t1;
                }
            }
        if (!changed) break;
    }
    for (i = 0; i < n; i++) {
        for (k = 0, sd = 0; k < m; k++) { double t = At[i][k]; sd += t * t; }
        W[i] = sqrt(sd);
    }
    for (i = 0; i < n - 1; i++) {
        j = i;
        for (k = i + 1; k < n; k++)
            if (W[j] < W[k]) j = k;
        if (i != j) {
            double t = W[i]; W[i] = W[j]; W[j] = t;
            for (k = 0; k < m; k++) { t = At[i][k]; At[i][k] = At[j][k]; At[j][k] = t; }
            for (k = 0; k < n; k++) { t = Vt[i][k]; Vt[i][k] = Vt[j][k]; Vt[j][k] = t; }
        }
    }
    for (i = 0; i < n; i++) {
        sd = W[i];
        s = sd > minval ? 1 / sd : 0.;
        for (k = 0; k < m; k++) At[i][k] *= s;
    }
}

static void homography_jacobi(const float from[8], const float to[8], double H[9])
{
    double a[8][8], b[8], At[8][8], W[8], Vt[8][8], x[8];
    for (int i = 0; i < 4; i++) {
        float fx = from[2 * i], fy = from[2 * i + 1], tx = to[2 * i], ty = to[2 * i + 1];
        a[i][0] = a[i + 4][3] = fx;
        a[i][1] = a[i + 4][4] = fy;
        a[i][2] = a[i + 4][5] = 1;
        a[i][3] = a[i][4] = a[i][5] = a[i + 4][0] = a[i + 4][1] = a[i + 4][2] = 0;
        a[i][6] = (double)(-fx * tx);     /* float product, as Point2f arithmetic */
        a[i][7] = (double)(-fy * tx);
        a[i + 4][6] = (double)(-fx * ty);
        a[i + 4][7] = (double)(-fy * ty);
        b[i] = tx;
        b[i + 4] = ty;
    }
    for (int r = 0; r < 8; r++)
        for (int c = 0; c < 8; c++) At[c][r] = a[r][c];
    jacobi_svd8(At, W, Vt);
    /* SVBkSb, nb == 1: x = sum_i (u_i . b / w_i) v_i over w_i above the cut-off */
    double threshold = 0;
    for (int i = 0; i < 8; i++) { x[i] = 0; threshold += W[i]; }
    threshold *= DBL_EPSILON * 2;
    for (int i = 0; i < 8; i++) {
        double wi = W[i];
        if (fabs(wi) <= threshold) continue;
        wi = 1 / wi;
        double s = 0;
        for (int j = 0; j < 8; j++) s += At[i][j] * b[j];
        s *= wi;
        for (int j = 0; j < 8; j++) x[j] = x[j] + s * Vt[i][j];
    }
    for (int i = 0; i < 8; i++) H[i] = x[i];
    H[8] = 1.;
}

/* den * (unit square -> quad) as an exact-integer matrix (Heckbert's square-to-quad). */
static void square_to_quad_scaled(const double q[8], double G[9])
{
    double x0 = q[0], y0 = q[1], x1 = q[2], y1 = q[3], x2 = q[4], y2 = q[5], x3 = q[6], y3 = q[7];
    double sx = x0 - x1 + x2 - x3, sy = y0 - y1 + y2 - y3;
    double dx1 = x1 - x2, dy1 = y1 - y2, dx2 = x3 - x2, dy2 = y3 - y2;
    double den = dx1 * dy2 - dx2 * dy1;
    double g = sx * dy2 - dx2 * sy;
    double h = dx1 * sy - sx * dy1;
    G[0] = den * (x1 - x0) + g * x1; G[1] = den * (x3 - x0) + h * x3; G[2] = den * x0;
    G[3] = den * (y1 - y0) + g * y1; G[4] = den * (y3 - y0) + h * y3; G[5] = den * y0;
    G[6] = g;                        G[7] = h;                        G[8] = den;
}

static int quad_in_general_position(const double q[8])
{
    for (int a = 0; a < 4; a++) {
        int b = (a + 1) & 3, c = (a + 2) & 3;
        double cr = (q[2 * b] - q[2 * a]) * (q[2 * c + 1] - q[2 * a + 1]) -
                    (q[2 * b + 1] - q[2 * a + 1]) * (q[2 * c] - q[2 * a]);
        if (cr == 0) return 0;
    }
    return 1;
}

static int homography_direct(const float from[8], const float to[8], double H[9])
{
    double qf[8], qt[8], Gf[9], Gt[9], Af[9], Hp[9];
    for (int i = 0; i < 8; i++) { qf[i] = from[i]; qt[i] = to[i]; }
    if (!quad_in_general_position(qf) || !quad_in_general_position(qt)) return 0;
    square_to_quad_scaled(qf, Gf);
    square_to_quad_scaled(qt, Gt);
    /* adjugate of Gf (exact in double for pixel-sized integer vertices) */
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
    if (Hp[8] == 0 || !isfinite(Hp[8])) return 0;
    for (int i = 0; i < 8; i++) H[i] = Hp[i] / Hp[8];
    H[8] = 1.;
    return 1;
}

VKO_API int vko_get_perspective_transform(const float from[8], const float to[8], int solver,
                                          double H[9])
{
    if (solver == 0 && homography_direct(from, to, H)) return 0;
    homography_jacobi(from, to, H);
    return solver == 0 ? 1 : 0; /* 1 = hybrid fell back to the SVD path */
}

/* ------------------------------------------------------------------------------------
 * [cv2] cv.fillPoly(mask, [pts], 1) with LINE_8, shift 0 -- element/polygon.py:70-77.
 * drawing.cpp of OpenCV 4.5.1 (the reference's minimum pinned version):
 *   CollectPolyEdges: every edge is first drawn with the 8-connected Bresenham Line()
 *   (LineIterator, left_to_right), then non-horizontal edges enter the edge table with
 *   16.16 fixed-point x and dx = (dX<<16)/dY (C division);
 *   FillEdgeCollection: scanlines y0 <= y < y1, x-sorted active edges paired even-odd,
 *   span [ (xa + 65535) >> 16 , xb >> 16 ].
 * ---------------------------------------------------------------------------------- */
static void draw_line8(uint8_t *img, int h, int w, int x1, int y1, int x2, int y2)
{
    /* All callers pass in-image end points (polygon relative to its own bounding box). */
    (void)h;
    int dx = x2 - x1, dy = y2 - y1;
    int s = dx < 0 ? -1 : 0;
    /* left_to_right */
    dx = (dx ^ s) - s;
    dy = (dy ^ s) - s;
    x1 ^= (x1 ^ x2) & s;
    y1 ^= (y1 ^ y2) & s;
    ptrdiff_t istep = w, bt_pix = 1;
    uint8_t *ptr = img + (ptrdiff_t)y1 * w + x1;
    s = dy < 0 ? -1 : 0;
    dy = (dy ^ s) - s;
    istep = (istep ^ s) - s;
    s = dy > dx ? -1 : 0;
    /* conditional swaps */
    dx ^= dy & s; dy ^= dx & s; dx ^= dy & s;
    { ptrdiff_t t; if (s) { t = bt_pix; bt_pix = istep; istep = t; } }
    int err = dx - (dy + dy);
    int plusDelta = dx + dx, minusDelta = -(dy + dy);
    ptrdiff_t plusStep = istep, minusStep = bt_pix;
    int count = dx + 1;
    for (int i = 0; i < count; i++) {
        *ptr = 1;
        int mask = err < 0 ? -1 : 0;
        err += minusDelta + (plusDelta & mask);
        ptr += minusStep + (mask ? plusStep : 0);
    }
}

typedef struct PolyEdge {
    int y0, y1;
    int64_t x, dx;
    struct PolyEdge *next;
} PolyEdge;

static int cmp_edges(const void *pa, const void *pb)
{
    const PolyEdge *a = (const PolyEdge *)pa, *b = (const PolyEdge *)pb;
    if (a->y0 != b->y0) return a->y0 < b->y0 ? -1 : 1;
    if (a->x != b->x) return a->x < b->x ? -1 : 1;
    if (a->dx != b->dx) return a->dx < b->dx ? -1 : 1;
    return 0;
}

VKO_API int vko_fill_poly(uint8_t *img, int h, int w, const int32_t *pts, int npts)
{
    if (npts <= 0) return 0;
    PolyEdge *edges = (PolyEdge *)calloc((size_t)npts + 2, sizeof(PolyEdge));
    if (!edges) return -1;
    int total = 0;
    int64_t p0x = (int64_t)pts[2 * (npts - 1)] << 16, p0y = pts[2 * (npts - 1) + 1];
    for (int i = 0; i < npts; i++) {
        int64_t p1x = (int64_t)pts[2 * i] << 16, p1y = pts[2 * i + 1];
        int t0x = (int)((p0x + (1 << 15)) >> 16), t1x = (int)((p1x + (1 << 15)) >> 16);
        draw_line8(img, h, w, t0x, (int)p0y, t1x, (int)p1y);
        if (p0y != p1y) {
            PolyEdge e;
            if (p0y < p1y) { e.y0 = (int)p0y; e.y1 = (int)p1y; e.x = p0x; }
            else           { e.y0 = (int)p1y; e.y1 = (int)p0y; e.x = p1x; }
            e.dx = (p1x - p0x) / (p1y - p0y);
            e.next = 0;
            edges[total++] = e;
        }
        p0x = p1x; p0y = p1y;
    }
    if (total < 2) { free(edges); return 0; }
    int y_max = INT_MIN;
    for (int i = 0; i < total; i++) if (edges[i].y1 > y_max) y_max = edges[i].y1;
    qsort(edges, (size_t)total, sizeof(PolyEdge), cmp_edges);
    PolyEdge tmp; memset(&tmp, 0, sizeof tmp);
    edges[total].y0 = INT_MAX; /* sentinel */
    int i = 0;
    tmp.next = 0;
    PolyEdge *e = &edges[i];
    if (y_max > h) y_max = h;
    for (int y = e->y0; y < y_max; y++) {
        PolyEdge *last, *prelast, *keep_prelast;
        int draw = 0;
        int clipline = y < 0;
        prelast = &tmp;
        last = tmp.next;
        while (last || e->y0 == y) {
            if (last && last->y1 == y) {
                prelast->next = last->next;
                last = last->next;
                continue;
            }
            keep_prelast = prelast;
            if (last && (e->y0 > y || last->x < e->x)) {
                prelast = last;
                last = last->next;
            } else if (i < total) {
                prelast->next = e;
                e->next = last;
                prelast = e;
                e = &edges[++i];
            } else
                break;
            if (draw) {
                if (!clipline) {
                    uint8_t *timg = img + (ptrdiff_t)y * w;
                    int x1, x2;
                    if (keep_prelast->x > prelast->x) {
                        x1 = (int)((prelast->x + 65535) >> 16);
                        x2 = (int)(keep_prelast->x >> 16);
                    } else {
                        x1 = (int)((keep_prelast->x + 65535) >> 16);
                        x2 = (int)(prelast->x >> 16);
                    }
                    if (x1 < w && x2 >= 0) {
                        if (x1 < 0) x1 = 0;
                        if (x2 >= w) x2 = w - 1;
                        for (int x = x1; x <= x2; x++) timg[x] = 1;
                    }
                }
                keep_prelast->x += keep_prelast->dx;
                prelast->x += prelast->dx;
            }
            draw ^= 1;
        }
        /* bubble sort of the active list by the advanced x */
        keep_prelast = 0;
        do {
            prelast = &tmp;
            last = tmp.next;
            PolyEdge *last_exchange = 0;
            while (last != keep_prelast && last && last->next != 0) {
                PolyEdge *te = last->next;
                if (last->x > te->x) {
                    prelast->next = te;
                    last->next = te->next;
                    te->next = last;
                    prelast = te;
                    last_exchange = prelast;
                } else {
                    prelast = last;
                    last = te;
                }
            }
            if (last_exchange == 0) break;
            keep_prelast = last_exchange;
        } while (keep_prelast != tmp.next && keep_prelast != &tmp);
    }
    free(edges);
    return 0;
}

/* Independent closed-form statement of the same fill (used to cross-check the literal
 * scan converter above and as the specification the HIP rasteriser is written against):
 * pixel set = Bresenham pixels of the edges  U  even-odd spans of the half-open edges. */
static int bres_minor(int k, int dmaj, int dmin)
{
    /* number of minor steps taken after k major steps: ceil((2*k*dmin - dmaj)/(2*dmaj)), >= 0 */
    if (dmaj == 0) return 0;
    long num = 2L * k * dmin - dmaj;
    if (num <= 0) return 0;
    return (int)((num + 2L * dmaj - 1) / (2L * dmaj));
}

VKO_API int vko_fill_poly_closed_form(uint8_t *img, int h, int w, const int32_t *pts, int npts)
{
    (void)h;
    if (npts <= 0) return 0;
    int y_lo = INT_MAX, y_hi = INT_MIN;
    for (int i = 0; i < npts; i++) {
        int a = (i + npts - 1) % npts;
        int xa = pts[2 * a], ya = pts[2 * a + 1], xb = pts[2 * i], yb = pts[2 * i + 1];
        /* Bresenham from the left end */
        int lx = xa, ly = ya, rx = xb, ry = yb;
        if (xb < xa) { lx = xb; ly = yb; rx = xa; ry = ya; }
        int dx = rx - lx, dy = ry - ly, ady = dy < 0 ? -dy : dy, sy = dy < 0 ? -1 : 1;
        if (ady > dx) {
            for (int k = 0; k <= ady; k++) img[(ptrdiff_t)(ly + sy * k) * w + lx + bres_minor(k, ady, dx)] = 1;
        } else {
            for (int k = 0; k <= dx; k++) img[(ptrdiff_t)(ly + sy * bres_minor(k, dx, ady)) * w + lx + k] = 1;
        }
        if (ya != yb) {
            if ((ya < yb ? ya : yb) < y_lo) y_lo = ya < yb ? ya : yb;
            if ((ya > yb ? ya : yb) > y_hi) y_hi = ya > yb ? ya : yb;
        }
    }
    int64_t *xs = (int64_t *)malloc(sizeof(int64_t) * (size_t)npts);
    if (!xs) return -1;
    for (int y = y_lo; y < y_hi; y++) {
        int n = 0;
        for (int i = 0; i < npts; i++) {
            int a = (i + npts - 1) % npts;
            int64_t xa = (int64_t)pts[2 * a] << 16, xb = (int64_t)pts[2 * i] << 16;
            int ya = pts[2 * a + 1], yb = pts[2 * i + 1];
            if (ya == yb) continue;
            int64_t dxe = (xb - xa) / (yb - ya);
            int y0 = ya < yb ? ya : yb, y1 = ya < yb ? yb : ya;
            int64_t x0 = ya < yb ? xa : xb;
            if (y0 <= y && y < y1) xs[n++] = x0 + (int64_t)(y - y0) * dxe;
        }
        for (int a = 1; a < n; a++) { /* insertion sort */
            int64_t v = xs[a]; int b = a - 1;
            while (b >= 0 && xs[b] > v) { xs[b + 1] = xs[b]; b--; }
            xs[b + 1] = v;
        }
        for (int a = 0; a + 1 < n; a += 2) {
            int x1 = (int)((xs[a] + 65535) >> 16), x2 = (int)(xs[a + 1] >> 16);
            if (x1 < 0) x1 = 0;
            if (x2 >= w) x2 = w - 1;
            for (int x = x1; x <= x2; x++) img[(ptrdiff_t)y * w + x] = 1;
        }
    }
    free(xs);
    return 0;
}

/* ------------------------------------------------------------------------------------
 * ImageGrid.generate_remap_params -- grid_rendering/type.py:209-261 (with :182-207 and
 * element/polygon.py:70-133).  Vertices are the ROUNDED integer grid points, (rows x cols
 * x 2) int32 as (x, y).  Cells in row-major order; later cells overwrite earlier ones;
 * pixels whose denominator is exactly 0 are skipped; untouched pixels stay (0, 0).
 * The per-pixel product np.matmul(inv_H (3x3 f64), [x; y; 1]) is a BLAS dgemm whose
 * micro-kernel accumulates k = 0,1,2 with FMA: acc = fma(m2, 1, fma(m1, y, m0 * x))
 * (verified against numpy 2.2.6 + OpenBLAS 0.3.29 for every N >= 2).
 * ---------------------------------------------------------------------------------- */
VKO_API int vko_grid_to_map(const int32_t *src_v, const int32_t *dst_v, int rows, int cols,
                            int dh, int dw, int solver, float *map_x, float *map_y,
                            int32_t *owner /* optional, cell index + 1 */)
{
    memset(map_x, 0, sizeof(float) * (size_t)dh * dw);
    memset(map_y, 0, sizeof(float) * (size_t)dh * dw);
    if (owner) memset(owner, 0, sizeof(int32_t) * (size_t)dh * dw);
    for (int r = 0; r + 1 < rows; r++)
        for (int c = 0; c + 1 < cols; c++) {
            int idx[4] = { r * cols + c, r * cols + c + 1, (r + 1) * cols + c + 1, (r + 1) * cols + c };
            float from[8], to[8];
            int32_t q[8];
            int xmin = INT_MAX, xmax = INT_MIN, ymin = INT_MAX, ymax = INT_MIN;
            for (int k = 0; k < 4; k++) {
                q[2 * k] = dst_v[2 * idx[k]]; q[2 * k + 1] = dst_v[2 * idx[k] + 1];
                from[2 * k] = (float)q[2 * k]; from[2 * k + 1] = (float)q[2 * k + 1];
                to[2 * k] = (float)src_v[2 * idx[k]]; to[2 * k + 1] = (float)src_v[2 * idx[k] + 1];
                if (q[2 * k] < xmin) xmin = q[2 * k];
                if (q[2 * k] > xmax) xmax = q[2 * k];
                if (q[2 * k + 1] < ymin) ymin = q[2 * k + 1];
                if (q[2 * k + 1] > ymax) ymax = q[2 * k + 1];
            }
            double H[9];
            vko_get_perspective_transform(from, to, solver, H);
            int bw = xmax - xmin + 1, bh = ymax - ymin + 1, np_ = 4;
            int32_t rel[8];
            for (int k = 0; k < 4; k++) { rel[2 * k] = q[2 * k] - xmin; rel[2 * k + 1] = q[2 * k + 1] - ymin; }
            /* PointList.from_np_array drops a closing duplicate (element/point.py:163-166) */
            if (rel[0] == rel[6] && rel[1] == rel[7]) np_ = 3;
            uint8_t *mask = (uint8_t *)calloc((size_t)bw * bh, 1);
            if (!mask) return -1;
            vko_fill_poly(mask, bh, bw, rel, np_);
            for (int yy = 0; yy < bh; yy++)
                for (int xx = 0; xx < bw; xx++) {
                    if (!mask[(size_t)yy * bw + xx]) continue;
                    int X = xx + xmin, Y = yy + ymin;
                    if (X < 0 || X >= dw || Y < 0 || Y >= dh) continue; /* cannot happen for a valid grid */
                    double fx = (double)X, fy = (double)Y;
                    double nx = fma(H[2], 1.0, fma(H[1], fy, H[0] * fx));
                    double ny = fma(H[5], 1.0, fma(H[4], fy, H[3] * fx));
                    double de = fma(H[8], 1.0, fma(H[7], fy, H[6] * fx));
                    if (de == 0) continue;
                    map_x[(size_t)Y * dw + X] = (float)(nx / de);
                    map_y[(size_t)Y * dw + X] = (float)(ny / de);
                    if (owner) owner[(size_t)Y * dw + X] = r * (cols - 1) + c + 1;
                }
            free(mask);
        }
    return 0;
}

/* ------------------------------------------------------------------------------------
 * [cv2] cv.GaussianBlur(u8, (k,k), sigma) -- photometric/blur.py:54-69.
 * OpenCV >= 4.2 bit-exact path (smooth.dispatch.cpp / smooth.simd.hpp): kernel in double
 * (getGaussianKernelBitExact), quantised to unsigned 8.8 fixed point with error diffusion
 * and centre = 256 - 2*sum(others) (getGaussianKernelFixedPoint_ED), horizontal pass
 * u8*8.8 -> 8.8, vertical pass 8.8*8.8 -> 16.16, (v + 32768) >> 16.  BORDER_REFLECT_101.
 * ---------------------------------------------------------------------------------- */
VKO_API int vko_gaussian_kernel_q8(int n, double sigma, uint16_t *kq)
{
    if (n <= 0 || (n & 1) == 0 || n > 255) return -1;
    if (sigma <= 0) return -2; /* not reachable from the path (sigma sampled in [0.5, 1]) */
    double scale2X = -0.125 / (sigma * sigma);
    int n2 = (n - 1) / 2;
    double values[128], kd[255];
    double sum = 0;
    for (int i = 0, x = 1 - n; i < n2; i++, x += 2) {
        double t = exp((double)(x * x) * scale2X);
        values[i] = t;
        sum += t;
    }
    sum *= 2;
    sum += 1;
    double mul1 = 1. / sum;
    for (int i = 0; i < n2; i++) { double t = values[i] * mul1; kd[i] = t; kd[n - 1 - i] = t; }
    kd[n2] = 1. * mul1;
    /* error-diffused 8.8 quantisation */
    double err = 0;
    int64_t isum = 0;
    for (int i = 0; i < n2; i++) {
        double adj = kd[i] * 256. + err;
        int64_t v0 = cv_round_d(adj);
        err = adj - (double)v0;
        kq[i] = (uint16_t)v0;
        kq[n - 1 - i] = (uint16_t)v0;
        isum += v0;
    }
    isum *= 2;
    kq[n2] = (uint16_t)(256 - isum);
    return 0;
}

static inline int reflect101(int p, int len)
{
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do {
        if (p < 0) p = -p;            /* -p - 1 + delta, delta = 1 */
        else p = 2 * (len - 1) - p;   /* len - 1 - (p - len) - delta */
    } while ((unsigned)p >= (unsigned)len);
    return p;
}

VKO_API int vko_gaussian_blur_u8(const uint8_t *src, int h, int w, int cn, ptrdiff_t sstep,
                                 int ksize, double sigma, uint8_t *dst, ptrdiff_t dstep)
{
    uint16_t kx[255], ky[255];
    int kw = ksize, kh = ksize;
    if (h == 1) kh = 1; /* GaussianBlur collapses the kernel on 1-pixel-high/wide images */
    if (w == 1) kw = 1;
    if (kw == 1 && kh == 1) {
        for (int y = 0; y < h; y++) memcpy(dst + (ptrdiff_t)y * dstep, src + (ptrdiff_t)y * sstep, (size_t)w * cn);
        return 0;
    }
    if (kw > 1) { if (vko_gaussian_kernel_q8(kw, sigma, kx)) return -1; } else kx[0] = 256;
    if (kh > 1) { if (vko_gaussian_kernel_q8(kh, sigma, ky)) return -1; } else ky[0] = 256;
    uint16_t *tmp = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)h * w * cn);
    if (!tmp) return -1;
    int rx = kw / 2, ry = kh / 2;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            for (int c = 0; c < cn; c++) {
                uint32_t acc = 0;
                for (int i = 0; i < kw; i++) {
                    int xx = reflect101(x + i - rx, w);
                    acc += (uint32_t)kx[i] * src[(ptrdiff_t)y * sstep + xx * cn + c];
                }
                tmp[((size_t)y * w + x) * cn + c] = (uint16_t)(acc > 65535 ? 65535 : acc);
            }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            for (int c = 0; c < cn; c++) {
                uint32_t acc = 0;
                for (int j = 0; j < kh; j++) {
                    int yy = reflect101(y + j - ry, h);
                    acc += (uint32_t)ky[j] * tmp[((size_t)yy * w + x) * cn + c];
                }
                dst[(ptrdiff_t)y * dstep + x * cn + c] = sat_u8((int)((acc + 32768u) >> 16));
            }
    free(tmp);
    return 0;
}

/* ------------------------------------------------------------------------------------
 * [cv2] cv.cvtColor RGB2HSV_FULL / HSV2RGB_FULL on uint8 -- element/image.py:188-202,
 * 794-808, reached from photometric/color.py:93-116 (color_shift).
 * RGB->HSV: integer LUT division (color_hsv.simd.hpp RGB2HSV_b), exactly reproducible.
 * HSV->RGB: float formula of HSV2RGB_native (scalar path, no FMA).  cv2's SIMD lanes use an
 * algebraically equal but differently associated form (v - v*s ...), so a given cv2 binary
 * may differ from this definition by 1 LSB on rare tie pixels.
 * ---------------------------------------------------------------------------------- */
static int g_sdiv[256], g_hdiv256[256], g_hsv_ready = 0;
static void init_hsv(void)
{
    if (g_hsv_ready) return;
    g_sdiv[0] = g_hdiv256[0] = 0;
    for (int i = 1; i < 256; i++) {
        g_sdiv[i] = cv_round_d((255 << 12) / (1. * i));
        g_hdiv256[i] = cv_round_d((256 << 12) / (6. * i));
    }
    g_hsv_ready = 1;
}

static inline void rgb2hsv_px(const uint8_t *s, uint8_t *d)
{
    int r = s[0], g = s[1], b = s[2];
    int v = b, vmin = b;
    if (g > v) v = g;
    if (r > v) v = r;
    if (g < vmin) vmin = g;
    if (r < vmin) vmin = r;
    int diff = v - vmin;
    int vr = v == r ? -1 : 0, vg = v == g ? -1 : 0;
    int sat = (diff * g_sdiv[v] + (1 << 11)) >> 12;
    int hh = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
    hh = (hh * g_hdiv256[diff] + (1 << 11)) >> 12;
    hh += hh < 0 ? 256 : 0;
    d[0] = sat_u8(hh);
    d[1] = (uint8_t)sat;
    d[2] = (uint8_t)v;
}

static inline void hsv2rgb_px(const uint8_t *s, uint8_t *d)
{
    static const int sector_data[6][3] = { {1,3,0}, {1,0,2}, {3,0,1}, {0,2,1}, {0,1,3}, {2,1,0} };
    const float hscale = 6.0f / 256;
    float h = (float)s[0];
    float sat = s[1] * (1.0f / 255.0f);
    float v = s[2] * (1.0f / 255.0f);
    float b, g, r;
    if (sat == 0) b = g = r = v;
    else {
        float tab[4];
        h *= hscale;
        h = fmodf(h, 6.f);
        int sector = (int)floorf(h);
        h -= sector;
        if ((unsigned)sector >= 6u) { sector = 0; h = 0.f; }
        tab[0] = v;
        tab[1] = v * (1.f - sat);
        tab[2] = v * (1.f - sat * h);
        tab[3] = v * (1.f - sat * (1.f - h));
        b = tab[sector_data[sector][0]];
        g = tab[sector_data[sector][1]];
        r = tab[sector_data[sector][2]];
    }
    d[0] = sat_u8(cv_round_f(r * 255.0f));
    d[1] = sat_u8(cv_round_f(g * 255.0f));
    d[2] = sat_u8(cv_round_f(b * 255.0f));
}

VKO_API int vko_rgb2hsv_full(const uint8_t *src, size_t npx, uint8_t *dst)
{
    init_hsv();
    for (size_t i = 0; i < npx; i++) rgb2hsv_px(src + 3 * i, dst + 3 * i);
    return 0;
}

VKO_API int vko_hsv2rgb_full(const uint8_t *src, size_t npx, uint8_t *dst)
{
    for (size_t i = 0; i < npx; i++) hsv2rgb_px(src + 3 * i, dst + 3 * i);
    return 0;
}

/* color_shift on an RGB image: RGB->HSV_FULL, H = (H + delta) mod 256 (python modulo),
 * HSV_FULL->RGB -- photometric/color.py:93-116, :32-55, photometric/opt.py:45-57. */
VKO_API int vko_color_shift_rgb(const uint8_t *src, size_t npx, int delta, uint8_t *dst)
{
    init_hsv();
    for (size_t i = 0; i < npx; i++) {
        uint8_t hsv[3];
        rgb2hsv_px(src + 3 * i, hsv);
        int hh = ((int)hsv[0] + delta) % 256;
        if (hh < 0) hh += 256;
        hsv[0] = (uint8_t)hh;
        hsv2rgb_px(hsv, dst + 3 * i);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------
 * [cv2] cv.cvtColor RGB2HLS_FULL / HLS2RGB_FULL / RGB2GRAY on uint8 -- element/image.py:188-202,771-814
 * (HSL images are HLS with the last two channels swapped by the reference, :183-186,207-210).
 * imgproc/color_hsv.simd.hpp RGB2HLS_b / HLS2RGB_b: 8-bit data go through the float formula (RGB2HLS_f /
 * HLS2RGB_f, scalar path restated here; products and sums round separately), hrange = 256 for *_FULL;
 * color_rgb.simd.hpp RGB2Gray<uchar>: 15-bit fixed point, (R*9798 + G*19235 + B*3735 + 2^14) >> 15.
 * ---------------------------------------------------------------------------------- */
static void rgb2hls_px(const uint8_t *p, uint8_t *q)
{
    float r = p[0] * (1.f / 255.f), g = p[1] * (1.f / 255.f), b = p[2] * (1.f / 255.f);
    float h = 0.f, s = 0.f, l;
    float vmin, vmax, diff;
    vmax = vmin = r;
    if (vmax < g) vmax = g;
    if (vmax < b) vmax = b;
    if (vmin > g) vmin = g;
    if (vmin > b) vmin = b;
    diff = vmax - vmin;
    l = (vmax + vmin) * 0.5f;
    if (diff > FLT_EPSILON) {
        s = l < 0.5f ? diff / (vmax + vmin) : diff / (2 - vmax - vmin);
        diff = 60.f / diff;
        if (vmax == r) h = (g - b) * diff;
        else if (vmax == g) h = (b - r) * diff + 120.f;
        else h = (r - g) * diff + 240.f;
        if (h < 0.f) h += 360.f;
    }
    const float hscale = 256.f / 360.f;
    q[0] = sat_u8(cv_round_f(h * hscale));
    q[1] = sat_u8(cv_round_f(l * 255.f));
    q[2] = sat_u8(cv_round_f(s * 255.f));
}

static void hls2rgb_px(const uint8_t *p, uint8_t *q)
{
    float h = (float)p[0], l = p[1] * (1.f / 255.f), s = p[2] * (1.f / 255.f);
    float b = l, g = l, r = l;
    if (s != 0) {
        static const int sector_data[][3] = { { 1, 3, 0 }, { 1, 0, 2 }, { 3, 0, 1 }, { 0, 2, 1 }, { 0, 1, 3 }, { 2, 1, 0 } };
        float tab[4];
        const float hscale = 6.f / 256.f;
        float p2 = l <= 0.5f ? l * (1 + s) : l + s - l * s;
        float p1 = 2 * l - p2;
        h *= hscale;
        if (h < 0) do h += 6; while (h < 0);
        else if (h >= 6) do h -= 6; while (h >= 6);
        int sector = (int)floorf(h);
        h -= sector;
        tab[0] = p2;
        tab[1] = p1;
        tab[2] = p1 + (p2 - p1) * (1 - h);
        tab[3] = p1 + (p2 - p1) * h;
        b = tab[sector_data[sector][0]];
        g = tab[sector_data[sector][1]];
        r = tab[sector_data[sector][2]];
    }
    q[0] = sat_u8(cv_round_f(r * 255.f));
    q[1] = sat_u8(cv_round_f(g * 255.f));
    q[2] = sat_u8(cv_round_f(b * 255.f));
}

VKO_API int vko_rgb2hls_full(const uint8_t *src, size_t npx, uint8_t *dst)
{
    for (size_t i = 0; i < npx; i++) rgb2hls_px(src + 3 * i, dst + 3 * i);
    return 0;
}

VKO_API int vko_hls2rgb_full(const uint8_t *src, size_t npx, uint8_t *dst)
{
    for (size_t i = 0; i < npx; i++) hls2rgb_px(src + 3 * i, dst + 3 * i);
    return 0;
}

static uint8_t rgb2gray_px(const uint8_t *p)
{
    return (uint8_t)((p[0] * 9798 + p[1] * 19235 + p[2] * 3735 + (1 << 14)) >> 15);
}

VKO_API int vko_rgb2gray(const uint8_t *src, size_t npx, uint8_t *dst)
{
    for (size_t i = 0; i < npx; i++) dst[i] = rgb2gray_px(src + 3 * i);
    return 0;
}

/* brightness_shift on an RGB image with the default HSL intermediate -- photometric/color.py:125-160:
 * RGB -> HLS_FULL, L = clip(L + delta) (the reference's channel 2 of its H,S,L order), HLS_FULL -> RGB. */
VKO_API int vko_brightness_shift_rgb(const uint8_t *src, size_t npx, int delta, uint8_t *dst)
{
    for (size_t i = 0; i < npx; i++) {
        uint8_t hls[3];
        rgb2hls_px(src + 3 * i, hls);
        if (delta != 0) {
            int l = (int)hls[1] + delta;
            hls[1] = (uint8_t)(l < 0 ? 0 : (l > 255 ? 255 : l));
        }
        hls2rgb_px(hls, dst + 3 * i);
    }
    return 0;
}

/* color_balance on an RGB image -- photometric/color.py:364-397: gray = RGB2GRAY replicated to 3 channels,
 * float32 blend fl32(1 - ratio) * gray + fl32(ratio) * px (products and sum round separately), np.clip, astype. */
VKO_API int vko_color_balance_rgb(const uint8_t *src, size_t npx, double ratio, uint8_t *dst)
{
    const float w0 = (float)(1 - ratio), w1 = (float)ratio;
    for (size_t i = 0; i < npx; i++) {
        const float gray = (float)rgb2gray_px(src + 3 * i);
        for (int c = 0; c < 3; c++) {
            float t0 = w0 * gray;
            float t1 = w1 * (float)src[3 * i + c];
            float v = t0 + t1;
            v = v < 0.f ? 0.f : (v > 255.f ? 255.f : v);
            dst[3 * i + c] = (uint8_t)v;
        }
    }
    return 0;
}

/* [numpy] mean_shift core: int16(px) + delta, optional threshold gate, CLIP or CYCLE
 * -- photometric/color.py:32-55, photometric/opt.py:41-57.  channels bit mask (0 = all). */
VKO_API int vko_mean_shift_u8(const uint8_t *src, size_t npx, int cn, int delta, int has_thr,
                              int thr, int cycle, unsigned chmask, uint8_t *dst)
{
    for (size_t i = 0; i < npx; i++)
        for (int c = 0; c < cn; c++) {
            int v = src[i * cn + c];
            if (chmask == 0 || (chmask >> c) & 1) {
                if (delta != 0) {
                    int apply = 1;
                    if (has_thr) apply = delta > 0 ? (v <= thr) : (thr <= v);
                    if (apply) v += delta;
                    if (cycle) { v %= 256; if (v < 0) v += 256; }
                    else v = v < 0 ? 0 : (v > 255 ? 255 : v);
                }
            }
            dst[i * cn + c] = (uint8_t)v;
        }
    return 0;
}

/* [numpy] gaussion_noise tail: clip(int16(px) + int16 noise, 0, 255) -- photometric/noise.py:51-53 */
VKO_API int vko_add_noise_i16(const uint8_t *src, const int16_t *noise, size_t n, uint8_t *dst)
{
    for (size_t i = 0; i < n; i++) {
        int v = (int16_t)((int16_t)src[i] + noise[i]);
        dst[i] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
    return 0;
}

/* ------------------------------------------------------------------------------------
 * [numpy] fill_np_array -- element/opt.py:118-209 via Box.fill_np_array element/box.py:311-340.
 * One layer onto a uint8 HxWxC destination, in place.
 *   box          : up, left, bh, bw (inside dst)
 *   mask         : optional uint8 [bh, bw], selected where > 0
 *   alpha_plane  : optional float32 [bh, bw]; if mask == NULL selects alpha > 0
 *   alpha_scalar : used when alpha_plane == NULL (already a Python float)
 *   value_plane  : optional uint8 [bh, bw, cn]; else value_const[cn]
 * Blend: out = (uint8) trunc( fl32(fl32(1 - a) * fl32(dst)) + fl32(a * fl32(val)) ).
 * ---------------------------------------------------------------------------------- */
VKO_API int vko_fill_u8(uint8_t *dst, int h, int w, int cn, ptrdiff_t dstep, int up, int left,
                        int bh, int bw, const uint8_t *mask, ptrdiff_t mask_step,
                        const float *alpha_plane, ptrdiff_t alpha_step_el, double alpha_scalar,
                        const uint8_t *value_plane, ptrdiff_t value_step,
                        const uint8_t *value_const)
{
    if (up < 0 || left < 0 || up + bh > h || left + bw > w) return -1;
    if (!alpha_plane) {
        if (alpha_scalar < 0.0 || alpha_scalar > 1.0) return -2;
        if (alpha_scalar == 0.0) return 0;
    }
    float a_s = (float)alpha_scalar;
    for (int y = 0; y < bh; y++)
        for (int x = 0; x < bw; x++) {
            float a = alpha_plane ? alpha_plane[(ptrdiff_t)y * alpha_step_el + x] : a_s;
            int sel = mask ? mask[(ptrdiff_t)y * mask_step + x] > 0 : (alpha_plane ? a > 0.0f : 1);
            if (!sel) continue;
            uint8_t *d = dst + (ptrdiff_t)(up + y) * dstep + (left + x) * cn;
            const uint8_t *v = value_plane ? value_plane + (ptrdiff_t)y * value_step + x * cn : value_const;
            if (!alpha_plane && alpha_scalar == 1.0) {
                for (int c = 0; c < cn; c++) d[c] = v[c];
            } else {
                float w1 = a, w0 = 1.0f - w1;
                for (int c = 0; c < cn; c++) {
                    float t0 = w0 * (float)d[c];
                    float t1 = w1 * (float)v[c];
                    float s = t0 + t1;
                    d[c] = (uint8_t)s;
                }
            }
        }
    return 0;
}

/* [numpy] the same function with keep_max_value / keep_min_value (element/opt.py:150-158): they act only in
 * the scalar alpha == 1 branch; mode 0 plain, 1 keep max (write where dst < value), 2 keep min. */
VKO_API int vko_fill_u8_mode(uint8_t *dst, int h, int w, int cn, ptrdiff_t dstep, int up, int left,
                             int bh, int bw, const uint8_t *mask, ptrdiff_t mask_step,
                             const float *alpha_plane, ptrdiff_t alpha_step_el, double alpha_scalar,
                             const uint8_t *value_plane, ptrdiff_t value_step,
                             const uint8_t *value_const, int mode)
{
    if (mode == 0 || alpha_plane || alpha_scalar != 1.0)
        return vko_fill_u8(dst, h, w, cn, dstep, up, left, bh, bw, mask, mask_step, alpha_plane,
                           alpha_step_el, alpha_scalar, value_plane, value_step, value_const);
    if (up < 0 || left < 0 || up + bh > h || left + bw > w) return -1;
    for (int y = 0; y < bh; y++)
        for (int x = 0; x < bw; x++) {
            if (mask && !(mask[(ptrdiff_t)y * mask_step + x] > 0)) continue;
            uint8_t *d = dst + (ptrdiff_t)(up + y) * dstep + (left + x) * cn;
            const uint8_t *v = value_plane ? value_plane + (ptrdiff_t)y * value_step + x * cn : value_const;
            for (int c = 0; c < cn; c++)
                if (mode == 1 ? d[c] < v[c] : d[c] > v[c]) d[c] = v[c];
        }
    return 0;
}

/* float32 destinations (ScoreMap): same branches, blend kept in float32 (astype(float32) is the identity). */
VKO_API int vko_fill_f32(float *dst, int h, int w, ptrdiff_t dstep_el, int up, int left, int bh, int bw,
                         const uint8_t *mask, ptrdiff_t mask_step, const float *alpha_plane,
                         ptrdiff_t alpha_step_el, double alpha_scalar, const float *value_plane,
                         ptrdiff_t value_step_el, float value_const, int mode)
{
    if (up < 0 || left < 0 || up + bh > h || left + bw > w) return -1;
    if (!alpha_plane) {
        if (alpha_scalar < 0.0 || alpha_scalar > 1.0) return -2;
        if (alpha_scalar == 0.0) return 0;
    }
    float a_s = (float)alpha_scalar;
    int copy = !alpha_plane && alpha_scalar == 1.0;
    for (int y = 0; y < bh; y++)
        for (int x = 0; x < bw; x++) {
            float a = alpha_plane ? alpha_plane[(ptrdiff_t)y * alpha_step_el + x] : a_s;
            int sel = mask ? mask[(ptrdiff_t)y * mask_step + x] > 0 : (alpha_plane ? a > 0.0f : 1);
            if (!sel) continue;
            float *d = dst + (ptrdiff_t)(up + y) * dstep_el + (left + x);
            float v = value_plane ? value_plane[(ptrdiff_t)y * value_step_el + x] : value_const;
            if (copy) {
                if (mode == 0 || (mode == 1 ? *d < v : *d > v)) *d = v;
            } else {
                float w1 = a, w0 = 1.0f - w1;
                float t0 = w0 * *d;
                float t1 = w1 * v;
                *d = t0 + t1;
            }
        }
    return 0;
}

/* [numpy] complement / posterization / channel_permutation -- photometric/color.py:299-357, 423-432.
 * op 0 complement (p0 threshold or -1, p1 lte), 1 posterize (p0 bits), 2 permute (p0: 2 bits per channel). */
VKO_API int vko_pointwise_u8(const uint8_t *src, size_t npix, int cn, int op, int p0, int p1,
                             unsigned chmask, uint8_t *dst)
{
    for (size_t i = 0; i < npix; i++)
        for (int c = 0; c < cn; c++) {
            int v = src[i * cn + c];
            int on = chmask == 0 || ((chmask >> c) & 1u);
            if (op == 0) {
                if (on && (p0 < 0 || (p1 ? v <= p0 : p0 <= v))) v = 255 - v;
            } else if (op == 1) {
                if (on) v &= (0xFF >> p0) << p0;
            } else if (op == 2) {
                v = src[i * cn + ((p0 >> (2 * c)) & 3)];
            } else {
                return -1;
            }
            dst[i * cn + c] = (uint8_t)v;
        }
    return 0;
}

/* [numpy] impulse_noise -- photometric/noise.py:125-150 (selector per pixel: 1 salt, 2 pepper). */
VKO_API int vko_impulse_noise_u8(const uint8_t *src, size_t npix, int cn, const uint8_t *sel, uint8_t *dst)
{
    for (size_t i = 0; i < npix; i++)
        for (int c = 0; c < cn; c++)
            dst[i * cn + c] = sel[i] == 1 ? 255 : (sel[i] == 2 ? 0 : src[i * cn + c]);
    return 0;
}

/* [numpy] speckle_noise -- photometric/noise.py:172-183: float32 mat + mat * float64 noise in float64,
 * np.clip, astype(uint8). */
VKO_API int vko_speckle_noise_u8(const uint8_t *src, size_t n, const double *noise, uint8_t *dst)
{
    for (size_t i = 0; i < n; i++) {
        double m = (double)(float)src[i];
        double t = m * noise[i];
        double v = m + t;
        v = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v);
        dst[i] = (uint8_t)v;
    }
    return 0;
}

/* [numpy] line_streak masks + two sequential blends -- photometric/streak.py:24-41,56-99 */
VKO_API int vko_line_streak_u8(uint8_t *img, int h, int w, int cn, ptrdiff_t step, int thickness,
                               int gap, int dash_thickness, int dash_gap, const uint8_t *color,
                               double alpha, int enable_vert, int enable_hori)
{
    int st = thickness + gap, dst_ = dash_thickness + dash_gap;
    int dash = dash_thickness > 0 && dash_gap > 0;
    uint8_t *mask = (uint8_t *)malloc((size_t)h * w);
    if (!mask) return -1;
    for (int pass = 0; pass < 2; pass++) {
        if (pass == 0 && !enable_vert) continue;
        if (pass == 1 && !enable_hori) continue;
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                int m;
                if (pass == 0) { m = (x % st) < thickness; if (dash && (y % dst_) < dash_gap) m = 0; }
                else           { m = (y % st) < thickness; if (dash && (x % dst_) < dash_gap) m = 0; }
                mask[(size_t)y * w + x] = (uint8_t)m;
            }
        int rc = vko_fill_u8(img, h, w, cn, step, 0, 0, h, w, mask, w, 0, 0, alpha, 0, 0, color);
        if (rc) { free(mask); return rc; }
    }
    free(mask);
    return 0;
}

/* ------------------------------------------------------------------------------------
 * [cv2] cv.resize(src, (dw, dh), interpolation=cv.INTER_CUBIC) -- element/image.py:836-852 (bottom layer
 * of fill_page_inactive_region, pipeline/text_detection/page_distortion.py:146-161), element/mask.py:454-479,
 * element/score_map.py:616-640.  imgproc/resize.cpp, generic path: resizeGeneric_ with HResizeCubic /
 * VResizeCubic (scalar code; cv2's SIMD vertical pass for uint8 evaluates the same sum in float32 and can
 * differ by 1 LSB -- parity unpinned at this boundary like every other cv2 call).
 *   scale = 1 / (dsize / ssize) in double; per destination index d: f = (float)((d + 0.5) * scale - 0.5),
 *   s = floor(f), f -= s; taps s-1 .. s+2 with border replication; Keys cubic, A = -0.75, in float32.
 *   uint8: coefficients rounded to 11-bit fixed point (cvRound), horizontal pass in int32, vertical pass
 *   saturate_u8((sum + (1 << 21)) >> 22).  float32: both passes in float32, left to right, no FMA.
 * ---------------------------------------------------------------------------------- */
static void vko_cubic_coeffs(float x, float c[4])
{
    const float A = -0.75f;
    c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
    c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
    c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
    c[3] = 1.f - c[0] - c[1] - c[2];
}

static int vko_clip_index(int x, int n) { return x < 0 ? 0 : (x >= n ? n - 1 : x); }

static void vko_resize_axis(int ssize, int dsize, int *ofs, float *coef /* [dsize][4] */)
{
    double inv_scale = (double)dsize / ssize;
    double scale = 1. / inv_scale;
    for (int d = 0; d < dsize; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s0 = (int)floorf(f);
        f -= s0;
        ofs[d] = s0;
        vko_cubic_coeffs(f, coef + 4 * d);
    }
}

VKO_API int vko_resize_cubic_u8(const uint8_t *src, int sh, int sw, int cn, ptrdiff_t sstep,
                                uint8_t *dst, int dh, int dw, ptrdiff_t dstep)
{
    if (sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0 || cn <= 0) return -1;
    int *xofs = (int *)malloc(sizeof(int) * dw), *yofs = (int *)malloc(sizeof(int) * dh);
    float *xc = (float *)malloc(sizeof(float) * 4 * dw), *yc = (float *)malloc(sizeof(float) * 4 * dh);
    short *xa = (short *)malloc(sizeof(short) * 4 * dw), *yb = (short *)malloc(sizeof(short) * 4 * dh);
    int32_t *rows = (int32_t *)malloc(sizeof(int32_t) * 4 * (size_t)dw * cn);
    if (!xofs || !yofs || !xc || !yc || !xa || !yb || !rows) return -2;
    vko_resize_axis(sw, dw, xofs, xc);
    vko_resize_axis(sh, dh, yofs, yc);
    for (int i = 0; i < 4 * dw; i++) xa[i] = (short)sat_short(cv_round_f(xc[i] * 2048.f));
    for (int i = 0; i < 4 * dh; i++) yb[i] = (short)sat_short(cv_round_f(yc[i] * 2048.f));
    for (int dy = 0; dy < dh; dy++) {
        for (int k = 0; k < 4; k++) {
            const uint8_t *S = src + (ptrdiff_t)vko_clip_index(yofs[dy] - 1 + k, sh) * sstep;
            int32_t *D = rows + (size_t)k * dw * cn;
            for (int dx = 0; dx < dw; dx++)
                for (int c = 0; c < cn; c++) {
                    uint32_t v = 0;
                    for (int j = 0; j < 4; j++) {
                        int sx = vko_clip_index(xofs[dx] - 1 + j, sw);
                        v += (uint32_t)((int32_t)S[sx * cn + c] * (int32_t)xa[4 * dx + j]);
                    }
                    D[dx * cn + c] = (int32_t)v;
                }
        }
        uint8_t *out = dst + (ptrdiff_t)dy * dstep;
        for (int x = 0; x < dw * cn; x++) {
            uint32_t v = 0;
            for (int k = 0; k < 4; k++)
                v += (uint32_t)rows[(size_t)k * dw * cn + x] * (uint32_t)(int32_t)yb[4 * dy + k];
            int32_t r = ((int32_t)(v + (1u << 21))) >> 22;
            out[x] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
        }
    }
    free(xofs); free(yofs); free(xc); free(yc); free(xa); free(yb); free(rows);
    return 0;
}

VKO_API int vko_resize_cubic_f32(const float *src, int sh, int sw, ptrdiff_t sstep_el, float *dst, int dh,
                                 int dw, ptrdiff_t dstep_el)
{
    if (sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0) return -1;
    int *xofs = (int *)malloc(sizeof(int) * dw), *yofs = (int *)malloc(sizeof(int) * dh);
    float *xc = (float *)malloc(sizeof(float) * 4 * dw), *yc = (float *)malloc(sizeof(float) * 4 * dh);
    float *rows = (float *)malloc(sizeof(float) * 4 * (size_t)dw);
    if (!xofs || !yofs || !xc || !yc || !rows) return -2;
    vko_resize_axis(sw, dw, xofs, xc);
    vko_resize_axis(sh, dh, yofs, yc);
    for (int dy = 0; dy < dh; dy++) {
        for (int k = 0; k < 4; k++) {
            const float *S = src + (ptrdiff_t)vko_clip_index(yofs[dy] - 1 + k, sh) * sstep_el;
            float *D = rows + (size_t)k * dw;
            for (int dx = 0; dx < dw; dx++) {
                const float *a = xc + 4 * dx;
                float t0 = S[vko_clip_index(xofs[dx] - 1, sw)] * a[0];
                float t1 = S[vko_clip_index(xofs[dx], sw)] * a[1];
                float t2 = S[vko_clip_index(xofs[dx] + 1, sw)] * a[2];
                float t3 = S[vko_clip_index(xofs[dx] + 2, sw)] * a[3];
                float v = t0 + t1;
                v = v + t2;
                v = v + t3;
                D[dx] = v;
            }
        }
        float *out = dst + (ptrdiff_t)dy * dstep_el;
        const float *b = yc + 4 * dy;
        for (int x = 0; x < dw; x++) {
            float t0 = rows[x] * b[0];
            float t1 = rows[(size_t)dw + x] * b[1];
            float t2 = rows[(size_t)2 * dw + x] * b[2];
            float t3 = rows[(size_t)3 * dw + x] * b[3];
            float v = t0 + t1;
            v = v + t2;
            v = v + t3;
            out[x] = v;
        }
    }
    free(xofs); free(yofs); free(xc); free(yc); free(rows);
    return 0;
}

/* [cv2] cv.resize(..., INTER_LINEAR) and INTER_NEAREST on uint8 -- photometric/effect.py:61-79 (pixelation:
 * shrink bilinearly, grow back with nearest neighbour).  imgproc/resize.cpp:
 *   LINEAR  f = (float)((d + 0.5) * scale - 0.5), s = floor(f), f -= s; s < 0 -> (s, f) = (0, 0); s >= size - 1 ->
 *           (size - 1, 0) horizontally (rows are clipped instead); coefficients (1 - f, f) as cvRound(c * 2048) shorts;
 *           horizontal pass in int32; vertical pass uchar((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2).
 *           An exact 2 x 2 shrink is routed to INTER_AREA: (a + b + c + d + 2) >> 2.
 *   NEAREST source index = min(floor(d * scale), size - 1), scale = 1 / (dsize / ssize) in double. */
static void vko_linear_axis(int ssize, int dsize, int *ofs, short *coef, int horizontal)
{
    double inv_scale = (double)dsize / ssize, scale = 1. / inv_scale;
    for (int d = 0; d < dsize; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s0 = (int)floorf(f);
        f -= s0;
        if (horizontal) {
            if (s0 < 0) { f = 0; s0 = 0; }
            if (s0 >= ssize - 1) { f = 0; s0 = ssize - 1; }
        }
        ofs[d] = s0;
        coef[2 * d] = (short)sat_short(cv_round_f((1.f - f) * 2048.f));
        coef[2 * d + 1] = (short)sat_short(cv_round_f(f * 2048.f));
    }
}

VKO_API int vko_resize_linear_u8(const uint8_t *src, int sh, int sw, int cn, ptrdiff_t sstep, uint8_t *dst,
                                 int dh, int dw, ptrdiff_t dstep)
{
    if (sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0 || cn <= 0) return -1;
    if (sw == 2 * dw && sh == 2 * dh) {   /* is_area_fast with iscale 2: INTER_AREA's 2 x 2 box */
        for (int y = 0; y < dh; y++)
            for (int x = 0; x < dw; x++)
                for (int c = 0; c < cn; c++) {
                    const uint8_t *p = src + (ptrdiff_t)(2 * y) * sstep + (2 * x) * cn + c;
                    dst[(ptrdiff_t)y * dstep + x * cn + c] = (uint8_t)((p[0] + p[cn] + p[sstep] + p[sstep + cn] + 2) >> 2);
                }
        return 0;
    }
    int *xofs = (int *)malloc(sizeof(int) * dw), *yofs = (int *)malloc(sizeof(int) * dh);
    short *xa = (short *)malloc(sizeof(short) * 2 * dw), *yb = (short *)malloc(sizeof(short) * 2 * dh);
    if (!xofs || !yofs || !xa || !yb) return -2;
    vko_linear_axis(sw, dw, xofs, xa, 1);
    vko_linear_axis(sh, dh, yofs, yb, 0);
    for (int dy = 0; dy < dh; dy++) {
        const uint8_t *S0 = src + (ptrdiff_t)vko_clip_index(yofs[dy], sh) * sstep;
        const uint8_t *S1 = src + (ptrdiff_t)vko_clip_index(yofs[dy] + 1, sh) * sstep;
        int b0 = yb[2 * dy], b1 = yb[2 * dy + 1];
        for (int dx = 0; dx < dw; dx++) {
            int sx0 = xofs[dx], sx1 = vko_clip_index(sx0 + 1, sw);
            int a0 = xa[2 * dx], a1 = xa[2 * dx + 1];
            for (int c = 0; c < cn; c++) {
                int h0 = S0[sx0 * cn + c] * a0 + S0[sx1 * cn + c] * a1;
                int h1 = S1[sx0 * cn + c] * a0 + S1[sx1 * cn + c] * a1;
                dst[(ptrdiff_t)dy * dstep + dx * cn + c] =
                    (uint8_t)((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2);
            }
        }
    }
    free(xofs); free(yofs); free(xa); free(yb);
    return 0;
}

VKO_API int vko_resize_nearest_u8(const uint8_t *src, int sh, int sw, int cn, ptrdiff_t sstep, uint8_t *dst,
                                  int dh, int dw, ptrdiff_t dstep)
{
    if (sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0 || cn <= 0) return -1;
    double ifx = 1. / ((double)dw / sw), ify = 1. / ((double)dh / sh);
    for (int y = 0; y < dh; y++) {
        int sy = (int)floor(y * ify);
        if (sy > sh - 1) sy = sh - 1;
        for (int x = 0; x < dw; x++) {
            int sx = (int)floor(x * ifx);
            if (sx > sw - 1) sx = sw - 1;
            for (int c = 0; c < cn; c++) dst[(ptrdiff_t)y * dstep + x * cn + c] = src[(ptrdiff_t)sy * sstep + sx * cn + c];
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------
 * [cv2] the other interpolations PageResizingStep samples (pipeline/text_detection/page_resizing.py:110-181 through
 * utility/opt.py:125-148: INTER_NEAREST_EXACT, INTER_LINEAR_EXACT, INTER_CUBIC, INTER_LANCZOS4, and INTER_AREA when
 * shrinking), applied to the page Image (uint8 x 3), its Masks (uint8) and ScoreMaps (float32).  imgproc/resize.cpp.
 *
 * LANCZOS4: the generic path of INTER_CUBIC with 8 taps s-3 .. s+4 and interpolateLanczos4's coefficients
 *   (sin / cos of -(x + 3 - i) pi / 4 by angle addition from one sin / cos pair, divided by y^2, normalised in float32;
 *   x < FLT_EPSILON -> the unit tap 3); uint8 through 11-bit coefficients, int32 sums and (sum + 2^21) >> 22;
 *   float32 sums left to right.
 * NEAREST_EXACT: resizeNN_bitexact -- 16.16 steps ifx = ((ssize << 16) + dsize / 2) / dsize, start ifx / 2 - ssize % 2,
 *   index min((ifx * d + ifx0) >> 16, ssize - 1).
 * LINEAR_EXACT on uint8: resize_bitExact<uchar, interpolationLinear> -- coordinates in double ("softdouble"),
 *   8.8 fixed-point weights (cvRound(frac * 256), 256 - that), destination pixels left / right of the first / last
 *   source sample copy it, horizontal pass ufixedpoint16, vertical pass 16.16 with (sum + 2^15) >> 16, rows outside
 *   round one horizontal row with (v + 128) >> 8; an exact 2 x 2 shrink is the INTER_AREA box.  On float32 the
 *   bit-exact table has no entry and cv.resize falls back to INTER_LINEAR: float32 weights (1 - f, f), s < 0 ->
 *   (0, f = 0), s >= size - 1 -> (size - 1, f = 0) on both axes, S0 * a0 + S1 * a1 per pass.
 * AREA (shrinking): integer scale factors -> ResizeAreaFast: box sum (int32 / float32) times 1.f / area, cvRound for
 *   uint8; the 2 x 2 box follows its vector body: (sum + 2) >> 2 on uint8, ((a + b) + (c + d)) * 0.25f on float32 (the
 *   scalar tail of a cv2 build rounds the last few columns differently); otherwise ResizeArea with computeResizeAreaTab's fractional cell weights,
 *   float32 accumulation in table order.
 * ---------------------------------------------------------------------------------- */
#define VKO_PI 3.1415926535897932384626433832795   /* CV_PI */
static void vko_lanczos4_coeffs(float x, float c[8])
{
    static const double s45 = 0.70710678118654752440084436210485;
    static const double cs[8][2] = {{1, 0}, {-s45, -s45}, {0, 1}, {s45, -s45}, {-1, 0}, {s45, s45}, {0, -1}, {-s45, s45}};
    if (x < FLT_EPSILON) {
        for (int i = 0; i < 8; i++) c[i] = 0;
        c[3] = 1;
        return;
    }
    float sum = 0;
    double y0 = -(x + 3) * VKO_PI * 0.25, s0 = sin(y0), c0 = cos(y0);
    for (int i = 0; i < 8; i++) {
        double y = -(x + 3 - i) * VKO_PI * 0.25;
        c[i] = (float)((cs[i][0] * s0 + cs[i][1] * c0) / (y * y));
        sum += c[i];
    }
    sum = 1.f / sum;
    for (int i = 0; i < 8; i++) c[i] *= sum;
}

static void vko_lanczos_axis(int ssize, int dsize, int *ofs, float *coef /* [dsize][8] */)
{
    double inv_scale = (double)dsize / ssize;
    double scale = 1. / inv_scale;
    for (int d = 0; d < dsize; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s0 = (int)floorf(f);
        f -= s0;
        ofs[d] = s0;
        vko_lanczos4_coeffs(f, coef + 8 * d);
    }
}

VKO_API int vko_resize_lanczos4_u8(const uint8_t *src, int sh, int sw, int cn, ptrdiff_t sstep,
                                   uint8_t *dst, int dh, int dw, ptrdiff_t dstep)
{
    if (sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0 || cn <= 0) return -1;
    int *xofs = (int *)malloc(sizeof(int) * dw), *yofs = (int *)malloc(sizeof(int) * dh);
    float *xc = (float *)malloc(sizeof(float) * 8 * dw), *yc = (float *)malloc(sizeof(float) * 8 * dh);
    short *xa = (short *)malloc(sizeof(short) * 8 * dw), *yb = (short *)malloc(sizeof(short) * 8 * dh);
    int32_t *rows = (int32_t *)malloc(sizeof(int32_t) * 8 * (size_t)dw * cn);
    if (!xofs || !yofs || !xc || !yc || !xa || !yb || !rows) return -2;
    vko_lanczos_axis(sw, dw, xofs, xc);
    vko_lanczos_axis(sh, dh, yofs, yc);
    for (int i = 0; i < 8 * dw; i++) xa[i] = (short)sat_short(cv_round_f(xc[i] * 2048.f));
    for (int i = 0; i < 8 * dh; i++) yb[i] = (short)sat_short(cv_round_f(yc[i] * 2048.f));
    for (int dy = 0; dy < dh; dy++) {
        for (int k = 0; k < 8; k++) {
            const uint8_t *S = src + (ptrdiff_t)vko_clip_index(yofs[dy] - 3 + k, sh) * sstep;
            int32_t *D = rows + (size_t)k * dw * cn;
            for (int dx = 0; dx < dw; dx++)
                for (int c = 0; c < cn; c++) {
                    uint32_t v = 0;
                    for (int j = 0; j < 8; j++) {
                        int sx = vko_clip_index(xofs[dx] - 3 + j, sw);
                        v += (uint32_t)((int32_t)S[sx * cn + c] * (int32_t)xa[8 * dx + j]);
                    }
                    D[dx * cn + c] = (int32_t)v;
                }
        }
        uint8_t *out = dst + (ptrdiff_t)dy * dstep;
        for (int x = 0; x < dw * cn; x++) {
            uint32_t v = 0;
            for (int k = 0; k < 8; k++)
                v += (uint32_t)rows[(size_t)k * dw * cn + x] * (uint32_t)(int32_t)yb[8 * dy + k];
            int32_t r = ((int32_t)(v + (1u << 21))) >> 22;
            out[x] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
        }
    }
    free(xofs); free(yofs); free(xc); free(yc); free(xa); free(yb); free(rows);
    return 0;
}

VKO_API int vko_resize_lanczos4_f32(const float *src, int sh, int sw, ptrdiff_t sstep_el, float *dst, int dh,
                                    int dw, ptrdiff_t dstep_el)
{
    if (sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0) return -1;
    int *xofs = (int *)malloc(sizeof(int) * dw), *yofs = (int *)malloc(sizeof(int) * dh);
    float *xc = (float *)malloc(sizeof(float) * 8 * dw), *yc = (float *)malloc(sizeof(float) * 8 * dh);
    float *rows = (float *)malloc(sizeof(float) * 8 * (size_t)dw);
    if (!xofs || !yofs || !xc || !yc || !rows) return -2;
    vko_lanczos_axis(sw, dw, xofs, xc);
    vko_lanczos_axis(sh, dh, yofs, yc);
    for (int dy = 0; dy < dh; dy++) {
        for (int k = 0; k < 8; k++) {
            const float *S = src + (ptrdiff_t)vko_clip_index(yofs[dy] - 3 + k, sh) * sstep_el;
            for (int dx = 0; dx < dw; dx++) {
                const float *a = xc + 8 * dx;
                float v = S[vko_clip_index(xofs[dx] - 3, sw)] * a[0];
                for (int j = 1; j < 8; j++) {
                    float t = S[vko_clip_index(xofs[dx] - 3 + j, sw)] * a[j];
                    v = v + t;
                }
                rows[(size_t)k * dw + dx] = v;
            }
        }
        const float *b = yc + 8 * dy;
        for (int x = 0; x < dw; x++) {
            float v = rows[x] * b[0];
            for (int k = 1; k < 8; k++) {
                float t = rows[(size_t)k * dw + x] * b[k];
                v = v + t;
            }
            dst[(ptrdiff_t)dy * dstep_el + x] = v;
        }
    }
    free(xofs); free(yofs); free(xc); free(yc); free(rows);
    return 0;
}

/* element size in bytes: 1 / 3 / 4 (a float32 plane is 4-byte elements) */
VKO_API int vko_resize_nearest_exact_u8(const uint8_t *src, int sh, int sw, int cn, ptrdiff_t sstep, uint8_t *dst,
                                        int dh, int dw, ptrdiff_t dstep)
{
    if (sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0 || cn <= 0) return -1;
    int ifx = (int)((((int64_t)sw << 16) + dw / 2) / dw), ifx0 = ifx / 2 - sw % 2;
    int ify = (int)((((int64_t)sh << 16) + dh / 2) / dh), ify0 = ify / 2 - sh % 2;
    for (int y = 0; y < dh; y++) {
        int sy = (int)(((int64_t)ify * y + ify0) >> 16);
        if (sy > sh - 1) sy = sh - 1;
        for (int x = 0; x < dw; x++) {
            int sx = (int)(((int64_t)ifx * x + ifx0) >> 16);
            if (sx > sw - 1) sx = sw - 1;
            for (int c = 0; c < cn; c++) dst[(ptrdiff_t)y * dstep + x * cn + c] = src[(ptrdiff_t)sy * sstep + sx * cn + c];
        }
    }
    return 0;
}

/* interpolationLinear: offsets, 8.8 weight of the right / lower sample, and the [min, max) range of destination
 * indices that interpolate */
static void vko_linear_exact_axis(int ssize, int dsize, int *ofs, int *w1, int *dmin, int *dmax)
{
    double inv_scale = (double)dsize / ssize, scale = 1.0 / inv_scale;
    int mn = 0, mx = dsize;
    for (int d = 0; d < dsize; d++) {
        double fval = scale * ((double)d + 0.5) - 0.5;
        int ival = (int)floor(fval);
        w1[d] = 0;
        if (ival >= 0 && ssize > 1) {
            if (ival < ssize - 1) {
                w1[d] = (int)nearbyint((fval - (double)ival) * 256.0);
            } else {
                ival = ssize - 1;
                if (d < mx) mx = d;
            }
        } else {
            if (d + 1 > mn) mn = d + 1;
            ival = 0;
        }
        ofs[d] = ival;
    }
    if (mx < mn) mx = mn;
    *dmin = mn; *dmax = mx;
}

VKO_API int vko_resize_linear_exact_u8(const uint8_t *src, int sh, int sw, int cn, ptrdiff_t sstep, uint8_t *dst,
                                       int dh, int dw, ptrdiff_t dstep)
{
    if (sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0 || cn <= 0) return -1;
    if (sw == 2 * dw && sh == 2 * dh && cn != 2)
        return vko_resize_linear_u8(src, sh, sw, cn, sstep, dst, dh, dw, dstep);   /* the INTER_AREA 2 x 2 box */
    int *xofs = (int *)malloc(sizeof(int) * dw), *yofs = (int *)malloc(sizeof(int) * dh);
    int *xw = (int *)malloc(sizeof(int) * dw), *yw = (int *)malloc(sizeof(int) * dh);
    uint32_t *h0 = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)dw * cn), *h1 = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)dw * cn);
    if (!xofs || !yofs || !xw || !yw || !h0 || !h1) return -2;
    int xmin, xmax, ymin, ymax;
    vko_linear_exact_axis(sw, dw, xofs, xw, &xmin, &xmax);
    vko_linear_exact_axis(sh, dh, yofs, yw, &ymin, &ymax);
    const int xlast = xofs[dw - 1], ylast = yofs[dh - 1];
    for (int dy = 0; dy < dh; dy++) {
        int r0, r1, two = 0;
        if (dy < ymin) r0 = r1 = 0;
        else if (dy >= ymax) r0 = r1 = ylast;
        else { r0 = yofs[dy]; r1 = r0 + 1; two = 1; }
        for (int pass = 0; pass < 1 + two; pass++) {
            const uint8_t *S = src + (ptrdiff_t)(pass ? r1 : r0) * sstep;
            uint32_t *H = pass ? h1 : h0;
            for (int dx = 0; dx < dw; dx++)
                for (int c = 0; c < cn; c++) {
                    uint32_t v;
                    if (dx < xmin) v = (uint32_t)S[c] << 8;
                    else if (dx >= xmax) v = (uint32_t)S[xlast * cn + c] << 8;
                    else v = (uint32_t)(256 - xw[dx]) * S[xofs[dx] * cn + c] + (uint32_t)xw[dx] * S[(xofs[dx] + 1) * cn + c];
                    H[dx * cn + c] = v;     /* ufixedpoint16: never above 255 << 8 */
                }
        }
        uint8_t *out = dst + (ptrdiff_t)dy * dstep;
        for (int x = 0; x < dw * cn; x++) {
            uint32_t r;
            if (!two) r = (h0[x] + 128u) >> 8;
            else r = (h0[x] * (uint32_t)(256 - yw[dy]) + h1[x] * (uint32_t)yw[dy] + (1u << 15)) >> 16;
            out[x] = (uint8_t)(r > 255 ? 255 : r);
        }
    }
    free(xofs); free(yofs); free(xw); free(yw); free(h0); free(h1);
    return 0;
}

VKO_API int vko_resize_linear_f32(const float *src, int sh, int sw, ptrdiff_t sstep_el, float *dst, int dh, int dw,
                                  ptrdiff_t dstep_el)
{
    if (sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0) return -1;
    if (sw == 2 * dw && sh == 2 * dh) {   /* is_area_fast with iscale 2: ResizeAreaFast on float32 */
        for (int y = 0; y < dh; y++)
            for (int x = 0; x < dw; x++) {
                const float *p = src + (ptrdiff_t)(2 * y) * sstep_el + 2 * x;
                float top = p[0] + p[1], bottom = p[sstep_el] + p[sstep_el + 1];   /* ResizeAreaFastVec_SIMD_32f's pairing */
                float sum = top + bottom;
                dst[(ptrdiff_t)y * dstep_el + x] = sum * 0.25f;
            }
        return 0;
    }
    const double sx_ = 1. / ((double)dw / sw), sy_ = 1. / ((double)dh / sh);
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * sy_ - 0.5);
        int y0 = (int)floorf(fy);
        fy -= y0;
        if (y0 < 0) { y0 = 0; fy = 0; }
        if (y0 >= sh - 1) { y0 = sh - 1; fy = 0; }
        const float *S0 = src + (ptrdiff_t)y0 * sstep_el, *S1 = src + (ptrdiff_t)vko_clip_index(y0 + 1, sh) * sstep_el;
        const float b0 = 1.f - fy, b1 = fy;
        for (int dx = 0; dx < dw; dx++) {
            float fx = (float)((dx + 0.5) * sx_ - 0.5);
            int x0 = (int)floorf(fx);
            fx -= x0;
            if (x0 < 0) { x0 = 0; fx = 0; }
            if (x0 >= sw - 1) { x0 = sw - 1; fx = 0; }
            const int x1 = vko_clip_index(x0 + 1, sw);
            const float a0 = 1.f - fx, a1 = fx;
            float p0 = S0[x0] * a0, p1 = S0[x1] * a1, q0 = S1[x0] * a0, q1 = S1[x1] * a1;
            float h0 = p0 + p1, h1 = q0 + q1;
            float t0 = h0 * b0, t1 = h1 * b1;
            dst[(ptrdiff_t)dy * dstep_el + dx] = t0 + t1;
        }
    }
    return 0;
}

/* computeResizeAreaTab: (destination index, source index, weight) triples of one axis, in order */
typedef struct { int di, si; float alpha; } vko_dec_alpha;

static int vko_area_tab(int ssize, int dsize, double scale, vko_dec_alpha *tab)
{
    int k = 0;
    for (int dx = 0; dx < dsize; dx++) {
        double fsx1 = dx * scale, fsx2 = fsx1 + scale;
        double cell = scale < ssize - fsx1 ? scale : ssize - fsx1;
        int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
        if (sx2 > ssize - 1) sx2 = ssize - 1;
        if (sx1 > sx2) sx1 = sx2;
        if (sx1 - fsx1 > 1e-3) {
            tab[k].di = dx; tab[k].si = sx1 - 1;
            tab[k++].alpha = (float)((sx1 - fsx1) / cell);
        }
        for (int sx = sx1; sx < sx2; sx++) {
            tab[k].di = dx; tab[k].si = sx;
            tab[k++].alpha = (float)(1.0 / cell);
        }
        if (fsx2 - sx2 > 1e-3) {
            double w = fsx2 - sx2;
            if (w > 1.) w = 1.;
            if (w > cell) w = cell;
            tab[k].di = dx; tab[k].si = sx2;
            tab[k++].alpha = (float)(w / cell);
        }
    }
    return k;
}

/* INTER_AREA for a shrink on both axes (dw <= sw, dh <= sh); is_f32 selects float32 planes (cn = 1) */
VKO_API int vko_resize_area(const void *src_, int sh, int sw, int cn, ptrdiff_t sstep_el, void *dst_, int dh, int dw,
                            ptrdiff_t dstep_el, int is_f32)
{
    if (sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0 || cn <= 0 || dw > sw || dh > sh) return -1;
    const uint8_t *s8 = (const uint8_t *)src_;
    const float *sf = (const float *)src_;
    uint8_t *d8 = (uint8_t *)dst_;
    float *df = (float *)dst_;
    const double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);
    const int iscale_x = (int)(scale_x > 2147483647. ? 2147483647. : nearbyint(scale_x));
    const int iscale_y = (int)(scale_y > 2147483647. ? 2147483647. : nearbyint(scale_y));
    const int fast = fabs(scale_x - iscale_x) < DBL_EPSILON && fabs(scale_y - iscale_y) < DBL_EPSILON;
    if (fast) {
        const int area = iscale_x * iscale_y;
        const float scale = 1.f / area;
        for (int y = 0; y < dh; y++)
            for (int x = 0; x < dw; x++)
                for (int c = 0; c < cn; c++) {
                    if (is_f32 && iscale_x == 2 && iscale_y == 2) {
                        /* the vector body of the 2 x 2 case pairs the rows: (a + b) + (c + d) */
                        const float *S = sf + (ptrdiff_t)(y * 2) * sstep_el + (ptrdiff_t)(x * 2) * cn + c;
                        float top = S[0] + S[cn], bottom = S[sstep_el] + S[sstep_el + cn];
                        float sum = top + bottom;
                        df[(ptrdiff_t)y * dstep_el + x * cn + c] = sum * 0.25f;
                    } else if (is_f32) {
                        float sum = 0;
                        int k = 0;
                        const float *S = sf + (ptrdiff_t)(y * iscale_y) * sstep_el + (ptrdiff_t)(x * iscale_x) * cn + c;
                        /* ofs[k] runs row-major over the box; summed in groups of four, then one by one */
                        for (; k <= area - 4; k += 4) {
                            float a0 = S[(k / iscale_x) * sstep_el + (k % iscale_x) * cn];
                            float a1 = S[((k + 1) / iscale_x) * sstep_el + ((k + 1) % iscale_x) * cn];
                            float a2 = S[((k + 2) / iscale_x) * sstep_el + ((k + 2) % iscale_x) * cn];
                            float a3 = S[((k + 3) / iscale_x) * sstep_el + ((k + 3) % iscale_x) * cn];
                            float g = a0 + a1;
                            g = g + a2;
                            g = g + a3;
                            sum = sum + g;
                        }
                        for (; k < area; k++) sum = sum + S[(k / iscale_x) * sstep_el + (k % iscale_x) * cn];
                        df[(ptrdiff_t)y * dstep_el + x * cn + c] = sum * scale;
                    } else {
                        int sum = 0;
                        const uint8_t *S = s8 + (ptrdiff_t)(y * iscale_y) * sstep_el + (ptrdiff_t)(x * iscale_x) * cn + c;
                        for (int k = 0; k < area; k++) sum += S[(k / iscale_x) * sstep_el + (k % iscale_x) * cn];
                        int r = (iscale_x == 2 && iscale_y == 2) ? (sum + 2) >> 2 : cv_round_f((float)sum * scale);
                        d8[(ptrdiff_t)y * dstep_el + x * cn + c] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
                    }
                }
        return 0;
    }
    vko_dec_alpha *xtab = (vko_dec_alpha *)malloc(sizeof(vko_dec_alpha) * ((size_t)sw + 2 * (size_t)dw + 2));
    vko_dec_alpha *ytab = (vko_dec_alpha *)malloc(sizeof(vko_dec_alpha) * ((size_t)sh + 2 * (size_t)dh + 2));
    float *buf = (float *)malloc(sizeof(float) * (size_t)dw * cn), *sum = (float *)malloc(sizeof(float) * (size_t)dw * cn);
    if (!xtab || !ytab || !buf || !sum) return -2;
    const int xn = vko_area_tab(sw, dw, scale_x, xtab), yn = vko_area_tab(sh, dh, scale_y, ytab);
    int prev_dy = ytab[0].di;
    for (int i = 0; i < dw * cn; i++) sum[i] = 0;
    for (int j = 0; j < yn; j++) {
        const float beta = ytab[j].alpha;
        const int dy = ytab[j].di, sy = ytab[j].si;
        for (int i = 0; i < dw * cn; i++) buf[i] = 0;
        for (int k = 0; k < xn; k++) {
            const float alpha = xtab[k].alpha;
            for (int c = 0; c < cn; c++) {
                float v = is_f32 ? sf[(ptrdiff_t)sy * sstep_el + xtab[k].si * cn + c]
                                 : (float)s8[(ptrdiff_t)sy * sstep_el + xtab[k].si * cn + c];
                float t = v * alpha;
                buf[xtab[k].di * cn + c] = buf[xtab[k].di * cn + c] + t;
            }
        }
        if (dy != prev_dy) {
            for (int i = 0; i < dw * cn; i++) {
                if (is_f32) df[(ptrdiff_t)prev_dy * dstep_el + i] = sum[i];
                else { int r = cv_round_f(sum[i]); d8[(ptrdiff_t)prev_dy * dstep_el + i] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r)); }
                sum[i] = beta * buf[i];
            }
            prev_dy = dy;
        } else {
            for (int i = 0; i < dw * cn; i++) { float t = beta * buf[i]; sum[i] = sum[i] + t; }
        }
    }
    for (int i = 0; i < dw * cn; i++) {
        if (is_f32) df[(ptrdiff_t)prev_dy * dstep_el + i] = sum[i];
        else { int r = cv_round_f(sum[i]); d8[(ptrdiff_t)prev_dy * dstep_el + i] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r)); }
    }
    free(xtab); free(ytab); free(buf); free(sum);
    return 0;
}

/* ------------------------------------------------------------------------------------
 * [cv2] defocus_blur / motion_blur -- photometric/blur.py:85-192: a small float32 kernel (a disc, or a rotated line,
 * normalised) is smoothed with cv.GaussianBlur and applied with cv.filter2D(image, -1, kernel).
 *   cv.GaussianBlur on float32: getGaussianKernel(ksize, sigma, CV_32F) = the bit-exact double kernel cast to float32,
 *     then sepFilter2D: rows first, then columns, symmetric form s = x0 k0 + (x-1 + x+1) k1 + (x-2 + x+2) k2 ...,
 *     float32, BORDER_REFLECT_101.
 *   cv.filter2D, uint8 -> uint8, float32 kernel below the DFT threshold: correlation anchored at the kernel centre,
 *     BORDER_REFLECT_101, the non-zero taps in row-major order, s += k * float(px) in float32, cvRound + saturate.
 * A cv2 build may fuse the multiply-adds of its vector body or hand the filter to IPP; the scalar C++ code is what is
 * restated here (parity unpinned, like every cv2 call).
 * ---------------------------------------------------------------------------------- */
VKO_API int vko_gaussian_kernel_f32(int n, double sigma, float *kf)
{
    if (n <= 0 || (n & 1) == 0 || n > 255 || sigma <= 0) return -1;
    double scale2X = -0.125 / (sigma * sigma);
    int n2 = (n - 1) / 2;
    double values[128];
    double sum = 0;
    for (int i = 0, x = 1 - n; i < n2; i++, x += 2) {
        double t = exp((double)(x * x) * scale2X);
        values[i] = t;
        sum += t;
    }
    sum *= 2;
    sum += 1;
    double mul1 = 1. / sum;
    for (int i = 0; i < n2; i++) { double t = values[i] * mul1; kf[i] = (float)t; kf[n - 1 - i] = (float)t; }
    kf[n2] = (float)(1. * mul1);
    return 0;
}

VKO_API int vko_gaussian_blur_f32(const float *src, int h, int w, ptrdiff_t sstep_el, int ksize, double sigma, float *dst,
                                  ptrdiff_t dstep_el)
{
    float k[255];
    if (h <= 0 || w <= 0 || vko_gaussian_kernel_f32(ksize, sigma, k)) return -1;
    const int r = ksize / 2;
    float *tmp = (float *)malloc(sizeof(float) * (size_t)h * w);
    if (!tmp) return -2;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const float *S = src + (ptrdiff_t)y * sstep_el;
            float s0 = S[x] * k[r];
            for (int j = 1; j <= r; j++) {
                float pair = S[reflect101(x - j, w)] + S[reflect101(x + j, w)];
                float t = pair * k[r + j];
                s0 = s0 + t;
            }
            tmp[(size_t)y * w + x] = s0;
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float s0 = k[r] * tmp[(size_t)y * w + x];
            for (int j = 1; j <= r; j++) {
                float pair = tmp[(size_t)reflect101(y + j, h) * w + x] + tmp[(size_t)reflect101(y - j, h) * w + x];
                float t = k[r + j] * pair;
                s0 = s0 + t;
            }
            dst[(ptrdiff_t)y * dstep_el + x] = s0;
        }
    free(tmp);
    return 0;
}

VKO_API int vko_filter2d_u8(const uint8_t *src, int h, int w, int cn, ptrdiff_t sstep, const float *kernel, int kh, int kw,
                            uint8_t *dst, ptrdiff_t dstep)
{
    if (h <= 0 || w <= 0 || cn <= 0 || kh <= 0 || kw <= 0) return -1;
    const int ay = kh / 2, ax = kw / 2;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            for (int c = 0; c < cn; c++) {
                float s0 = 0.f;
                for (int ky = 0; ky < kh; ky++)
                    for (int kx = 0; kx < kw; kx++) {
                        const float f = kernel[ky * kw + kx];
                        if (f == 0) continue;
                        const int sy = reflect101(y + ky - ay, h), sx = reflect101(x + kx - ax, w);
                        const float t = f * (float)src[(ptrdiff_t)sy * sstep + sx * cn + c];
                        s0 = s0 + t;
                    }
                int r = cv_round_f(s0);
                dst[(ptrdiff_t)y * dstep + x * cn + c] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
            }
    return 0;
}

/* ------------------------------------------------------------------------------------
 * [cv2] cv.Rodrigues(rvec) (double internals) and cv.projectPoints with zero distortion
 * -- geometric/camera.py:96, :189-195.  calib3d/calibration.cpp cvRodrigues2 /
 * cvProjectPoints2Internal.
 * ---------------------------------------------------------------------------------- */
VKO_API int vko_rodrigues(const double r_in[3], double R[9])
{
    double rx = r_in[0], ry = r_in[1], rz = r_in[2];
    double theta = sqrt(rx * rx + ry * ry + rz * rz);
    if (theta < DBL_EPSILON) {
        memset(R, 0, sizeof(double) * 9);
        R[0] = R[4] = R[8] = 1;
        return 0;
    }
    double c = cos(theta), s = sin(theta), c1 = 1. - c, itheta = theta ? 1. / theta : 0.;
    rx *= itheta; ry *= itheta; rz *= itheta;
    double rrt[9] = { rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz };
    double r_x[9] = { 0, -rz, ry, rz, 0, -rx, -ry, rx, 0 };
    double I[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    for (int k = 0; k < 9; k++) R[k] = c * I[k] + c1 * rrt[k] + s * r_x[k];
    return 0;
}

VKO_API int vko_project_points(const double *pts3, size_t n, const double rvec[3],
                               const double tvec[3], double fx, double fy, double cx, double cy,
                               double *out2)
{
    double R[9];
    vko_rodrigues(rvec, R);
    for (size_t i = 0; i < n; i++) {
        double X = pts3[3 * i], Y = pts3[3 * i + 1], Z = pts3[3 * i + 2];
        double x = R[0] * X + R[1] * Y + R[2] * Z + tvec[0];
        double y = R[3] * X + R[4] * Y + R[5] * Z + tvec[1];
        double z = R[6] * X + R[7] * Y + R[8] * Z + tvec[2];
        z = z ? 1. / z : 1;
        x *= z; y *= z;
        out2[2 * i] = x * fx + cx;
        out2[2 * i + 1] = y * fy + cy;
    }
    return 0;
}

VKO_API int vko_version(void) { return 1; }

/* -----------------------------------------------------------------------------------------------------------------
 * SimilarityMlsPointProjector.project_point -- vkit/mechanism/distortion/geometric/mls.py:38-135, all vertices.
 * float32 numpy arithmetic in the reference's order of operations.  The accumulation orders are those numpy 2.2.6 +
 * OpenBLAS 0.3.29 use for these shapes in the container the goldens were generated in (pinned by
 * tests/golden/mls_states.npz / mls_lattices.npz, which come from the imported reference):
 *   np.sum over a contiguous axis: numpy's pairwise sum (n < 8 sequential; else 8 running sums over whole blocks of 8,
 *   ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the tail); np.sum(axis=0) of an (N, 2): row after row;
 *   (N,) @ (N, 2): acc = fma(w_i, p_i, acc), for exactly four handles fma(w0,p0,w1 p1) + fma(w2,p2,w3 p3);
 *   (N, 2) @ (2, 2) and (N,1,2) @ (N,2,2): fma(a1, b1, a0 b0).
 * Returns 0, or 1 + the index of a vertex where np.errstate(divide='raise') fires (mls.py:72-74).
 * ----------------------------------------------------------------------------------------------------------------- */
static float vko_mls_weight(const float *p, int i, float vx, float vy)
{
    const float dx = p[2 * i] - vx, dy = p[2 * i + 1] - vy;
    const float d2 = dx * dx + dy * dy;
    return 1.f / d2;
}

static float vko_pairwise_sum_f32(const float *a, int n)
{
    /* numpy's pairwise_sum (loops_utils.h.src): < 8 sequential, up to the 128-element block eight running sums, beyond it
     * the halves (the first a multiple of 8) summed recursively */
    if (n < 8) {
        float res = a[0];
        for (int i = 1; i < n; i++) res = res + a[i];
        return res;
    }
    if (n > 128) {
        int n2 = n / 2;
        n2 -= n2 % 8;
        return vko_pairwise_sum_f32(a, n2) + vko_pairwise_sum_f32(a + n2, n - n2);
    }
    float r[8];
    for (int j = 0; j < 8; j++) r[j] = a[j];
    int i = 8;
    for (; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; j++) r[j] = r[j] + a[i + j];
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res = res + a[i];
    return res;
}

VKO_API int vko_mls_project(const float *p, const float *q, const double *ps, const double *qs, int n,
                            const double *vertices, int n_vertices, double *out)
{
    if (n < 1) return -1;
    float *w = (float *)malloc(sizeof(float) * 2 * (size_t)n), *t = w + n;
    if (!w) return -1;
    for (int v = 0; v < n_vertices; v++) {
        const double vxd = vertices[2 * v], vyd = vertices[2 * v + 1];
        int hit = -1;
        for (int i = 0; i < n; i++)
            if (ps[2 * i] == vxd && ps[2 * i + 1] == vyd) hit = i;      /* mls.py:57-61, dict: last duplicate wins */
        if (hit >= 0) { out[2 * v] = qs[2 * hit]; out[2 * v + 1] = qs[2 * hit + 1]; continue; }
        const float vx = (float)vxd, vy = (float)vyd;
        for (int i = 0; i < n; i++) {
            const float dx = p[2 * i] - vx, dy = p[2 * i + 1] - vy;
            if (dx * dx + dy * dy == 0.f) { free(w); return v + 1; }
            w[i] = vko_mls_weight(p, i, vx, vy);                        /* mls.py:64-74 */
        }
        const float sw = vko_pairwise_sum_f32(w, n);
        float psx, psy, qsx, qsy;
        /* mls.py:77-78  w_norm (N,) @ p (N, 2): cblas_sgemv on a 2 x N column-major matrix.  OpenBLAS 0.3.29 as numpy 2.2.6
         * ships it, on an AVX-512 host (the machine the goldens were generated on): 5 <= N <= 48 goes to the SkylakeX
         * small-matrix kernel, one fused multiply-add per handle in order; every other N to the generic two-row tail of
         * sgemv_n_4.c, which takes the handles four at a time as
         *   temp += fma(w0, p0, w1 p1);  temp += fma(w2, p2, w3 p3);
         * and the rest one fused multiply-add each.  (A host whose OpenBLAS picks other kernels -- no AVX-512 -- rounds the
         * reference's own lattice differently for 5 <= N <= 48: the reference is machine dependent here.) */
        psx = psy = qsx = qsy = 0.f;
        {
            int i = 0;
            if (n < 5 || n > 48) {
                for (; i + 4 <= n; i += 4) {
                    for (int h = 0; h < 4; h += 2) {
                        const float wa = w[i + h] / sw, wb = w[i + h + 1] / sw;
                        const float *pa = p + 2 * (i + h), *pb = pa + 2, *qa = q + 2 * (i + h), *qb = qa + 2;
                        psx = psx + fmaf(wa, pa[0], wb * pb[0]); psy = psy + fmaf(wa, pa[1], wb * pb[1]);
                        qsx = qsx + fmaf(wa, qa[0], wb * qb[0]); qsy = qsy + fmaf(wa, qa[1], wb * qb[1]);
                    }
                }
            }
            for (; i < n; i++) {
                const float wn = w[i] / sw;
                psx = fmaf(wn, p[2 * i], psx); psy = fmaf(wn, p[2 * i + 1], psy);
                qsx = fmaf(wn, q[2 * i], qsx); qsy = fmaf(wn, q[2 * i + 1], qsy);
            }
        }
        const float ax = vx - psx, ay = vy - psy;                       /* mls.py:89-106 */
        float sx = 0.f, sy = 0.f;
        for (int i = 0; i < n; i++) {
            const float hx = p[2 * i] - psx, hy = p[2 * i + 1] - psy;   /* mls.py:81-86 */
            const float gx = q[2 * i] - qsx, gy = q[2 * i + 1] - qsy;
            t[i] = w[i] * (hx * hx + hy * hy);                          /* mls.py:130 */
            const float t0 = fmaf(hy, ay, hx * ax), t1 = fmaf(hy, -ax, hx * ay);      /* mls.py:108 */
            const float b0 = fmaf(-hx, ay, hy * ax), b1 = fmaf(-hx, -ax, hy * ay);    /* mls.py:109 */
            const float a00 = w[i] * t0, a01 = w[i] * t1, a10 = w[i] * b0, a11 = w[i] * b1;   /* mls.py:111-114 */
            const float e0 = fmaf(gy, a10, gx * a00), e1 = fmaf(gy, a11, gx * a01);           /* mls.py:118-129 */
            if (i == 0) { sx = e0; sy = e1; } else { sx = sx + e0; sy = sy + e1; }
        }
        const float mu = vko_pairwise_sum_f32(t, n);
        out[2 * v] = (double)(sx / mu + qsx);                           /* mls.py:131 */
        out[2 * v + 1] = (double)(sy / mu + qsy);
    }
    free(w);
    return 0;
}

/* -----------------------------------------------------------------------------------------------------------------
 * Throughput-mode noise plane (include/vkx.h: vkx_noise_normal_i16) -- NOT a restatement of the reference: the reference
 * draws np.round(rng.normal(0, std, shape)) from numpy's stream (photometric/noise.py:44-54); this mode only shares the
 * distribution.  What is pinned here is the library's own definition: Philox2x32-10 (Salmon, Moraes, Dror, Shaw,
 * "Parallel random numbers: as easy as 1, 2, 3", SC'11; Random123 known answers in tests/test_oracle_golden.py) per
 * pixel, four 16-bit uniforms, inverse-CDF table of round(N(0, std)) at the slice midpoints.
 * ----------------------------------------------------------------------------------------------------------------- */
VKO_API void vko_philox2x32_10(uint32_t c0, uint32_t c1, uint32_t key, uint32_t out[2])
{
    for (int r = 0; r < 10; r++) {
        const uint64_t prod = (uint64_t)0xD256D193u * (uint64_t)c0;
        const uint32_t hi = (uint32_t)(prod >> 32), lo = (uint32_t)prod;
        c0 = hi ^ key ^ c1;
        c1 = lo;
        key += 0x9E3779B9u;
    }
    out[0] = c0; out[1] = c1;
}

VKO_API void vko_noise_normal_table(double std, int16_t *table)
{
    const double inv = 1.0 / (std * 1.4142135623730951);
    int k = -32767;
    const double z = -4.6 * std;
    if (z > -32766.0) k = (int)floor(z) - 1;
    if (k < -32767) k = -32767;
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

VKO_API int vko_noise_normal_i16(int16_t *dst, int h, int w, int cn, double std, uint64_t seed)
{
    if (cn < 1 || cn > 4 || !(std > 0)) return -1;
    int16_t *table = (int16_t *)malloc(65536 * sizeof(int16_t));
    if (!table) return -1;
    vko_noise_normal_table(std, table);
    /* sample s of the flat plane: the (s & 3)-th 16-bit uniform of Philox block s >> 2 */
    const long long n = (long long)h * w * cn;
    for (long long q = 0; q * 4 < n; q++) {
        uint32_t r[2];
        vko_philox2x32_10((uint32_t)q, (uint32_t)(seed >> 32), (uint32_t)seed, r);
        const uint32_t u[4] = {r[0] & 0xffffu, r[0] >> 16, r[1] & 0xffffu, r[1] >> 16};
        for (int j = 0; j < 4 && q * 4 + j < n; j++) dst[q * 4 + j] = table[u[j]];
    }
    free(table);
    return 0;
}

/* -----------------------------------------------------------------------------------------------------------------
 * [cv2] cv.ellipse(mask, center, axes, angle=0, startAngle=0, endAngle=360, color=1, thickness >= 1) as
 * ellipse_streak_image calls it (photometric/streak.py:306-326): LINE_8, shift 0, one call per concentric box.
 * Restated from OpenCV 4.5.x modules/imgproc/src/drawing.cpp (parity unpinned like every [cv2] member: no cv2 in this
 * image):  ellipse -> EllipseEx (arc step from the larger axis) -> ellipse2Poly (float SinTable of 7-decimal literals,
 * double vertices in 16.16) -> rounded, consecutive duplicates dropped -> PolyLine(open) -> ThickLine per segment:
 * thickness <= 1: Line2 (16.16 DDA after clipLine); otherwise FillConvexPoly of the offset quad (its edges through
 * Line2, then the two-edge scan conversion) and a filled Circle at the segment ends.
 * ----------------------------------------------------------------------------------------------------------------- */
#define VKO_XY_SHIFT 16
#define VKO_XY_ONE (1 << VKO_XY_SHIFT)

static float vko_sin_tab[451];
static int vko_sin_tab_ready = 0;

static void vko_init_sin_tab(void)
{
    /* SinTable[]: sin of whole degrees 0..450 written as 7-decimal float literals */
    if (vko_sin_tab_ready) return;
    for (int d = 0; d <= 450; d++) {
        char buf[32];
        snprintf(buf, sizeof buf, "%.7f", sin((double)d * 3.14159265358979323846 / 180.0));
        vko_sin_tab[d] = strtof(buf, 0);
    }
    vko_sin_tab_ready = 1;
}

typedef struct { int64_t x, y; } vko_pt2l;

static inline void vko_put(uint8_t *img, int h, int w, int64_t x, int64_t y)
{
    if (0 <= x && x < w && 0 <= y && y < h) img[(ptrdiff_t)y * w + x] = 1;
}

static inline void vko_hline(uint8_t *img, int w, int y, int x1, int x2)
{
    for (int x = x1; x <= x2; x++) img[(ptrdiff_t)y * w + x] = 1;
}

/* clipLine(Size2l, Point2l&, Point2l&) */
static int vko_clip_line(int64_t width, int64_t height, vko_pt2l *p1, vko_pt2l *p2)
{
    const int64_t right = width - 1, bottom = height - 1;
    if (width <= 0 || height <= 0) return 0;
    int64_t x1 = p1->x, y1 = p1->y, x2 = p2->x, y2 = p2->y;
    int c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8;
    int c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8;
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
        int64_t a;
        if (c1 & 12) {
            a = c1 < 8 ? 0 : bottom;
            x1 += (int64_t)((double)(a - y1) * (double)(x2 - x1) / (double)(y2 - y1));
            y1 = a;
            c1 = (x1 < 0) + (x1 > right) * 2;
        }
        if (c2 & 12) {
            a = c2 < 8 ? 0 : bottom;
            x2 += (int64_t)((double)(a - y2) * (double)(x2 - x1) / (double)(y2 - y1));
            y2 = a;
            c2 = (x2 < 0) + (x2 > right) * 2;
        }
        if ((c1 & c2) == 0 && (c1 | c2) != 0) {
            if (c1) {
                a = c1 == 1 ? 0 : right;
                y1 += (int64_t)((double)(a - x1) * (double)(y2 - y1) / (double)(x2 - x1));
                x1 = a;
                c1 = 0;
            }
            if (c2) {
                a = c2 == 1 ? 0 : right;
                y2 += (int64_t)((double)(a - x2) * (double)(y2 - y1) / (double)(x2 - x1));
                x2 = a;
                c2 = 0;
            }
        }
    }
    p1->x = x1; p1->y = y1; p2->x = x2; p2->y = y2;
    return (c1 | c2) == 0;
}

/* Line2: end points in 16.16 */
static void vko_line2(uint8_t *img, int h, int w, vko_pt2l pt1, vko_pt2l pt2)
{
    if (!vko_clip_line((int64_t)w << VKO_XY_SHIFT, (int64_t)h << VKO_XY_SHIFT, &pt1, &pt2)) return;
    int64_t dx = pt2.x - pt1.x, dy = pt2.y - pt1.y;
    const int64_t j = dx < 0 ? -1 : 0, i = dy < 0 ? -1 : 0;
    const int64_t ax = (dx ^ j) - j, ay = (dy ^ i) - i;
    int64_t x_step, y_step;
    int ecount;
    if (ax > ay) {
        dy = (dy ^ j) - j;
        if (j) { vko_pt2l t = pt1; pt1 = pt2; pt2 = t; }
        x_step = VKO_XY_ONE;
        y_step = (dy * VKO_XY_ONE) / (ax | 1);        /* (dy << XY_SHIFT) / (ax | 1), C division */
        ecount = (int)((pt2.x - pt1.x) >> VKO_XY_SHIFT);
    } else {
        dx = (dx ^ i) - i;
        if (i) { vko_pt2l t = pt1; pt1 = pt2; pt2 = t; }
        x_step = (dx * VKO_XY_ONE) / (ay | 1);
        y_step = VKO_XY_ONE;
        ecount = (int)((pt2.y - pt1.y) >> VKO_XY_SHIFT);
    }
    pt1.x += VKO_XY_ONE >> 1;
    pt1.y += VKO_XY_ONE >> 1;
    vko_put(img, h, w, (pt2.x + (VKO_XY_ONE >> 1)) >> VKO_XY_SHIFT, (pt2.y + (VKO_XY_ONE >> 1)) >> VKO_XY_SHIFT);
    if (ax > ay) {
        pt1.x >>= VKO_XY_SHIFT;
        while (ecount >= 0) {
            vko_put(img, h, w, pt1.x, pt1.y >> VKO_XY_SHIFT);
            pt1.x++;
            pt1.y += y_step;
            ecount--;
        }
    } else {
        pt1.y >>= VKO_XY_SHIFT;
        while (ecount >= 0) {
            vko_put(img, h, w, pt1.x >> VKO_XY_SHIFT, pt1.y);
            pt1.x += x_step;
            pt1.y++;
            ecount--;
        }
    }
    (void)x_step;
}

/* FillConvexPoly(img, v, npts, color, LINE_8, shift = XY_SHIFT) */
static void vko_fill_convex_poly16(uint8_t *img, int h, int w, const vko_pt2l *v, int npts)
{
    struct { int idx, di; int64_t x, dx; int ye; } edge[2];
    const int shift = VKO_XY_SHIFT, delta = 1 << shift >> 1;
    const int delta1 = VKO_XY_ONE >> 1, delta2 = VKO_XY_ONE >> 1;
    int imin = 0, edges = npts;
    int64_t xmin = v[0].x, xmax = v[0].x, ymin = v[0].y, ymax = v[0].y;
    vko_pt2l p0 = v[npts - 1];
    for (int i = 0; i < npts; i++) {
        const vko_pt2l p = v[i];
        if (p.y < ymin) { ymin = p.y; imin = i; }
        if (p.y > ymax) ymax = p.y;
        if (p.x > xmax) xmax = p.x;
        if (p.x < xmin) xmin = p.x;
        vko_line2(img, h, w, p0, p);
        p0 = p;
    }
    xmin = (xmin + delta) >> shift;
    xmax = (xmax + delta) >> shift;
    ymin = (ymin + delta) >> shift;
    ymax = (ymax + delta) >> shift;
    if (npts < 3 || (int)xmax < 0 || (int)ymax < 0 || (int)xmin >= w || (int)ymin >= h) return;
    if (ymax > h - 1) ymax = h - 1;
    int y = (int)ymin;
    edge[0].idx = edge[1].idx = imin;
    edge[0].ye = edge[1].ye = y;
    edge[0].di = 1;
    edge[1].di = npts - 1;
    edge[0].x = edge[1].x = -VKO_XY_ONE;
    edge[0].dx = edge[1].dx = 0;
    do {
        for (int i = 0; i < 2; i++) {
            if (y >= edge[i].ye) {
                int idx0 = edge[i].idx;
                const int di = edge[i].di;
                int idx = idx0 + di;
                if (idx >= npts) idx -= npts;
                for (; edges-- > 0;) {
                    const int ty = (int)((v[idx].y + delta) >> shift);
                    if (ty > y) {
                        const int64_t xs = v[idx0].x, xe = v[idx].x;
                        edge[i].ye = ty;
                        edge[i].dx = ((xe - xs) * 2 + (ty - y)) / (2 * (ty - y));
                        edge[i].x = xs;
                        edge[i].idx = idx;
                        break;
                    }
                    idx0 = idx;
                    idx += di;
                    if (idx >= npts) idx -= npts;
                }
            }
        }
        if (edges < 0) break;
        if (y >= 0) {
            int left = 0, right = 1;
            if (edge[0].x > edge[1].x) { left = 1; right = 0; }
            int xx1 = (int)((edge[left].x + delta1) >> VKO_XY_SHIFT);
            int xx2 = (int)((edge[right].x + delta2) >> VKO_XY_SHIFT);
            if (xx2 >= 0 && xx1 < w) {
                if (xx1 < 0) xx1 = 0;
                if (xx2 >= w) xx2 = w - 1;
                vko_hline(img, w, y, xx1, xx2);
            }
        }
        edge[0].x += edge[0].dx;
        edge[1].x += edge[1].dx;
    } while (++y <= (int)ymax);
}

/* Circle(img, center, radius, color, fill = 1) */
static void vko_circle_fill(uint8_t *img, int h, int w, int cx, int cy, int radius)
{
    int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
    while (dx >= dy) {
        const int y11 = cy - dy, y12 = cy + dy, y21 = cy - dx, y22 = cy + dx;
        int x11 = cx - dx, x12 = cx + dx, x21 = cx - dy, x22 = cx + dy;
        /* the "inside" fast path writes the same pixels as the clipped one */
        if (x11 < w && x12 >= 0 && y21 < h && y22 >= 0) {
            if (x11 < 0) x11 = 0;
            if (x12 > w - 1) x12 = w - 1;
            if ((unsigned)y11 < (unsigned)h) vko_hline(img, w, y11, x11, x12);
            if ((unsigned)y12 < (unsigned)h) vko_hline(img, w, y12, x11, x12);
            if (x21 < w && x22 >= 0) {
                if (x21 < 0) x21 = 0;
                if (x22 > w - 1) x22 = w - 1;
                if ((unsigned)y21 < (unsigned)h) vko_hline(img, w, y21, x21, x22);
                if ((unsigned)y22 < (unsigned)h) vko_hline(img, w, y22, x21, x22);
            }
        }
        dy++;
        err += plus;
        plus += 2;
        const int mask = (err <= 0) - 1;
        err -= minus & mask;
        dx += mask;
        minus -= mask & 2;
    }
}

/* ThickLine(img, p0, p1, color, thickness, LINE_8, flags, shift = XY_SHIFT) */
static void vko_thick_line(uint8_t *img, int h, int w, vko_pt2l p0, vko_pt2l p1, int thickness, int flags)
{
    if (thickness <= 1) {
        vko_line2(img, h, w, p0, p1);
        return;
    }
    const double inv_xy_one = 1.0 / VKO_XY_ONE;
    vko_pt2l pt[4], dp = {0, 0};
    const double dx = (double)(p0.x - p1.x) * inv_xy_one, dy = (double)(p1.y - p0.y) * inv_xy_one;
    double r = dx * dx + dy * dy;
    const int odd = thickness & 1;
    thickness <<= VKO_XY_SHIFT - 1;
    if (fabs(r) > DBL_EPSILON) {
        r = ((double)thickness + (double)(odd * VKO_XY_ONE) * 0.5) / sqrt(r);
        dp.x = cv_round_d(dy * r);
        dp.y = cv_round_d(dx * r);
        pt[0].x = p0.x + dp.x; pt[0].y = p0.y + dp.y;
        pt[1].x = p0.x - dp.x; pt[1].y = p0.y - dp.y;
        pt[2].x = p1.x - dp.x; pt[2].y = p1.y - dp.y;
        pt[3].x = p1.x + dp.x; pt[3].y = p1.y + dp.y;
        vko_fill_convex_poly16(img, h, w, pt, 4);
    }
    for (int i = 0; i < 2; i++) {
        if (flags & (i + 1)) {
            const int cx = (int)((p0.x + (VKO_XY_ONE >> 1)) >> VKO_XY_SHIFT);
            const int cy = (int)((p0.y + (VKO_XY_ONE >> 1)) >> VKO_XY_SHIFT);
            vko_circle_fill(img, h, w, cx, cy, (thickness + (VKO_XY_ONE >> 1)) >> VKO_XY_SHIFT);
        }
        p0 = p1;
    }
}

/* The 16.16 outline vertices of one ellipse (EllipseEx's `v`): returns their number (<= 74) */
VKO_API int vko_ellipse_vertices(int cx, int cy, int ax, int ay, int64_t *xy /* [74][2] */)
{
    vko_init_sin_tab();
    const int64_t cxs = (int64_t)cx * VKO_XY_ONE, cys = (int64_t)cy * VKO_XY_ONE;
    int64_t aw = (int64_t)ax * VKO_XY_ONE, ah = (int64_t)ay * VKO_XY_ONE;
    if (aw < 0) aw = -aw;
    if (ah < 0) ah = -ah;
    int delta = (int)(((aw > ah ? aw : ah) + (VKO_XY_ONE >> 1)) >> VKO_XY_SHIFT);
    delta = delta < 3 ? 90 : delta < 10 ? 30 : delta < 15 ? 18 : 5;
    const float alpha = vko_sin_tab[450], beta = vko_sin_tab[0];     /* sincos(angle = 0) */
    int n = 0;
    int64_t prev_x = -1, prev_y = -1;      /* Point2l(0xFFFFFFFFFFFFFFFF, 0xFFFFFFFFFFFFFFFF) */
    int npoly = 0;
    double first_x = 0, first_y = 0;
    for (int i = 0; i < 360 + delta; i += delta) {
        int angle = i > 360 ? 360 : i;
        const double x = (double)aw * vko_sin_tab[450 - angle], y = (double)ah * vko_sin_tab[angle];
        const double px = (double)cxs + x * alpha - y * beta, py = (double)cys + x * beta + y * alpha;
        if (npoly == 0) { first_x = px; first_y = py; }
        npoly++;
        int64_t qx = (int64_t)cv_round_d(px / VKO_XY_ONE) * VKO_XY_ONE;
        int64_t qy = (int64_t)cv_round_d(py / VKO_XY_ONE) * VKO_XY_ONE;
        qx += cv_round_d(px - (double)qx);
        qy += cv_round_d(py - (double)qy);
        if (qx != prev_x || qy != prev_y) {
            xy[2 * n] = qx; xy[2 * n + 1] = qy;
            n++;
            prev_x = qx; prev_y = qy;
        }
    }
    (void)first_x; (void)first_y;
    if (n == 1) {          /* a zero-size polygon: two copies of the centre */
        xy[0] = xy[2] = cxs; xy[1] = xy[3] = cys;
        n = 2;
    }
    return n;
}

VKO_API int vko_ellipse_outline(uint8_t *mask, int h, int w, int cx, int cy, int ax, int ay, int thickness)
{
    if (thickness < 1 || thickness > 32767) return -1;       /* cv.ellipse: 0 < thickness <= MAX_THICKNESS here */
    int64_t xy[74 * 2];
    const int n = vko_ellipse_vertices(cx, cy, ax, ay, xy);
    int flags = 3;                                           /* PolyLine(is_closed = false): 2 + !is_closed */
    for (int i = 1; i < n; i++) {
        const vko_pt2l p0 = {xy[2 * i - 2], xy[2 * i - 1]}, p1 = {xy[2 * i], xy[2 * i + 1]};
        vko_thick_line(mask, h, w, p0, p1, thickness, flags);
        flags = 2;
    }
    return 0;
}
