"""Host-side ``cv.getPerspectiveTransform(src, dst, cv.DECOMP_SVD)`` for the few places that need a single
homography on the host: the skew states (reference affine.py:326-330, 386-390) and the point projection of
image-grid distortions (grid_rendering/type.py:166-180).  Per-cell homographies of the dense remap are solved
on the GPU (``k_cell_setup`` in csrc/grid.hip) with the same definition:

* quads in general position: closed-form quad->quad map assembled from exact integer sub-determinants
  (plain double arithmetic, no fused multiply-add, fixed evaluation order);
* otherwise (three collinear vertices): minimum-norm least squares of the 8x8 DLT system through a one-sided
  Jacobi SVD with OpenCV's 2*eps*sum(w) singular-value cut-off.
"""
import math

import numpy as np

_EPS = 2.220446049250313e-16
_DBL_MIN = 2.2250738585072014e-308


def _square_to_quad_scaled(q):
    (x0, y0), (x1, y1), (x2, y2), (x3, y3) = q
    sx = x0 - x1 + x2 - x3
    sy = y0 - y1 + y2 - y3
    dx1, dy1, dx2, dy2 = x1 - x2, y1 - y2, x3 - x2, y3 - y2
    den = dx1 * dy2 - dx2 * dy1
    g = sx * dy2 - dx2 * sy
    h = dx1 * sy - sx * dy1
    return [
        den * (x1 - x0) + g * x1, den * (x3 - x0) + h * x3, den * x0,
        den * (y1 - y0) + g * y1, den * (y3 - y0) + h * y3, den * y0,
        g, h, den,
    ]


def _in_general_position(q):
    for a in range(4):
        b, c = (a + 1) & 3, (a + 2) & 3
        cross = (q[b][0] - q[a][0]) * (q[c][1] - q[a][1]) - (q[b][1] - q[a][1]) * (q[c][0] - q[a][0])
        if cross == 0:
            return False
    return True


def _direct(qf, qt):
    if not (_in_general_position(qf) and _in_general_position(qt)):
        return None
    G = _square_to_quad_scaled(qf)
    T = _square_to_quad_scaled(qt)
    A = [
        G[4] * G[8] - G[5] * G[7], G[2] * G[7] - G[1] * G[8], G[1] * G[5] - G[2] * G[4],
        G[5] * G[6] - G[3] * G[8], G[0] * G[8] - G[2] * G[6], G[2] * G[3] - G[0] * G[5],
        G[3] * G[7] - G[4] * G[6], G[1] * G[6] - G[0] * G[7], G[0] * G[4] - G[1] * G[3],
    ]
    Hp = [(T[r * 3] * A[c] + T[r * 3 + 1] * A[3 + c]) + T[r * 3 + 2] * A[6 + c] for r in range(3) for c in range(3)]
    if Hp[8] == 0 or not math.isfinite(Hp[8]):
        return None
    return [Hp[i] / Hp[8] for i in range(8)] + [1.0]


def _hypot(a, b):
    a, b = abs(a), abs(b)
    if a < b:
        a, b = b, a
    if a == 0:
        return 0.0
    r = b / a
    return a * math.sqrt(1 + r * r)


def _jacobi(pf, pt):
    n = 8
    At = [[0.0] * n for _ in range(n)]
    rhs = [0.0] * n
    for i in range(4):
        fx, fy, tx, ty = (np.float32(v) for v in (pf[i][0], pf[i][1], pt[i][0], pt[i][1]))
        At[0][i], At[1][i], At[2][i] = float(fx), float(fy), 1.0
        At[3][i + 4], At[4][i + 4], At[5][i + 4] = float(fx), float(fy), 1.0
        At[6][i], At[7][i] = float(-fx * tx), float(-fy * tx)          # float32 products (Point2f arithmetic)
        At[6][i + 4], At[7][i + 4] = float(-fx * ty), float(-fy * ty)
        rhs[i], rhs[i + 4] = float(tx), float(ty)
    Vt = [[1.0 if r == c else 0.0 for c in range(n)] for r in range(n)]
    W = [0.0] * n
    for i in range(n):
        sd = 0.0
        for k in range(n):
            sd += At[i][k] * At[i][k]
        W[i] = sd
    eps = _EPS * 10
    for _ in range(30):
        changed = False
        for i in range(n - 1):
            for j in range(i + 1, n):
                Ai, Aj = At[i], At[j]
                a, b, p = W[i], W[j], 0.0
                for k in range(n):
                    p += Ai[k] * Aj[k]
                if abs(p) <= eps * math.sqrt(a * b):
                    continue
                p *= 2
                beta = a - b
                gamma = _hypot(p, beta)
                if beta < 0:
                    s = math.sqrt((gamma - beta) * 0.5 / gamma)
                    c = p / (gamma * s * 2)
                else:
                    c = math.sqrt((gamma + beta) / (gamma * 2))
                    s = p / (gamma * c * 2)
                a = b = 0.0
                for k in range(n):
                    t0 = c * Ai[k] + s * Aj[k]
                    t1 = -s * Ai[k] + c * Aj[k]
                    Ai[k], Aj[k] = t0, t1
                    a += t0 * t0
                    b += t1 * t1
                W[i], W[j] = a, b
                changed = True
                Vi, Vj = Vt[i], Vt[j]
                for k in range(n):
                    t0 = c * Vi[k] + s * Vj[k]
                    t1 = -s * Vi[k] + c * Vj[k]
                    Vi[k], Vj[k] = t0, t1
        if not changed:
            break
    for i in range(n):
        sd = 0.0
        for k in range(n):
            sd += At[i][k] * At[i][k]
        W[i] = math.sqrt(sd)
    for i in range(n - 1):
        j = i
        for k in range(i + 1, n):
            if W[j] < W[k]:
                j = k
        if i != j:
            W[i], W[j] = W[j], W[i]
            At[i], At[j] = At[j], At[i]
            Vt[i], Vt[j] = Vt[j], Vt[i]
    for i in range(n):
        s = 1 / W[i] if W[i] > _DBL_MIN else 0.0
        At[i] = [v * s for v in At[i]]
    threshold = 0.0
    for w in W:
        threshold += w
    threshold *= _EPS * 2
    x = [0.0] * n
    for i in range(n):
        if abs(W[i]) <= threshold:
            continue
        acc = 0.0
        for j in range(n):
            acc += At[i][j] * rhs[j]
        acc *= 1 / W[i]
        for j in range(n):
            x[j] = x[j] + acc * Vt[i][j]
    return x + [1.0]


def get_perspective_transform(pts_from: np.ndarray, pts_to: np.ndarray) -> np.ndarray:
    """3x3 float64 homography taking the four ``pts_from`` (x, y) onto ``pts_to``."""
    pf = [(float(np.float32(p[0])), float(np.float32(p[1]))) for p in np.asarray(pts_from).reshape(4, 2)]
    pt = [(float(np.float32(p[0])), float(np.float32(p[1]))) for p in np.asarray(pts_to).reshape(4, 2)]
    H = _direct(pf, pt)
    if H is None:
        H = _jacobi(pf, pt)
    return np.asarray(H, dtype=np.float64).reshape(3, 3)
