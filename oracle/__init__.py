"""CPU oracle for the distortion hot path -- TEST INFRASTRUCTURE ONLY.

numpy-facing wrappers over ``oracle/vkx_oracle.c`` (built by ``oracle/Makefile`` into
``oracle/_build/libvkx_oracle.so``).  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this package; the product package
``vkit_amd`` never does.

Parity status: **unpinned at the cv2 boundary** (OpenCV is neither vendored in the reference
nor installed here); the numpy-only members are pinned by ``tests/golden``.  See the header of
``vkx_oracle.c`` for the per-function provenance.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, '_build', 'libvkx_oracle.so')

SOLVER_HYBRID = 0  # closed-form homography, Jacobi-SVD only for degenerate quads (parity definition)
SOLVER_JACOBI = 1  # restatement of OpenCV's built-in DECOMP_SVD path for every cell


def build(force=False):
    srcs = [os.path.join(_HERE, 'vkx_oracle.c'), os.path.join(_HERE, 'np_random.c'), os.path.join(_HERE, 'vkx_cpu.c')]
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(s) for s in srcs):
        subprocess.run(['make', '-C', _HERE], check=True, capture_output=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def _p(arr):
    return arr.ctypes.data_as(ctypes.c_void_p)


_ss = ctypes.c_ssize_t


def _as3(img):
    if img.ndim == 2:
        return img.reshape(img.shape[0], img.shape[1], 1), True
    return img, False


def remap(src, map_x, map_y):
    """cv.remap(src, map_x, map_y, INTER_LINEAR), BORDER_CONSTANT 0; uint8 HxW[xC] or float32 HxW."""
    map_x = np.ascontiguousarray(map_x, dtype=np.float32)
    map_y = np.ascontiguousarray(map_y, dtype=np.float32)
    dh, dw = map_x.shape
    src = np.ascontiguousarray(src)
    if src.dtype == np.uint8:
        s3, squeeze = _as3(src)
        sh, sw, cn = s3.shape
        dst = np.zeros((dh, dw, cn), np.uint8)
        rc = lib().vko_remap_u8(_p(s3), sh, sw, cn, _ss(sw * cn), _p(map_x), _p(map_y), _ss(dw),
                                _p(dst), dh, dw, _ss(dw * cn))
        assert rc == 0
        return dst[:, :, 0] if squeeze else dst
    assert src.dtype == np.float32 and src.ndim == 2
    sh, sw = src.shape
    dst = np.zeros((dh, dw), np.float32)
    rc = lib().vko_remap_f32(_p(src), sh, sw, _ss(sw), _p(map_x), _p(map_y), _ss(dw), _p(dst), dh, dw,
                             _ss(dw))
    assert rc == 0
    return dst


def _sample_fixed(src, X, Y):
    dh, dw = X.shape
    src = np.ascontiguousarray(src)
    if src.dtype == np.uint8:
        s3, squeeze = _as3(src)
        sh, sw, cn = s3.shape
        dst = np.zeros((dh, dw, cn), np.uint8)
        rc = lib().vko_sample_fixed_u8(_p(s3), sh, sw, cn, _ss(sw * cn), _p(X), _p(Y), _p(dst), dh, dw,
                                       _ss(dw * cn))
        assert rc == 0
        return dst[:, :, 0] if squeeze else dst
    assert src.dtype == np.float32 and src.ndim == 2
    sh, sw = src.shape
    dst = np.zeros((dh, dw), np.float32)
    rc = lib().vko_sample_fixed_f32(_p(src), sh, sw, _ss(sw), _p(X), _p(Y), _p(dst), dh, dw, _ss(dw))
    assert rc == 0
    return dst


def warp_affine_coords(M, dsize):
    dw, dh = int(dsize[0]), int(dsize[1])
    M = np.ascontiguousarray(np.asarray(M, dtype=np.float64).reshape(6))
    X = np.zeros((dh, dw), np.int32)
    Y = np.zeros((dh, dw), np.int32)
    assert lib().vko_warp_affine_coords(_p(M), dh, dw, _p(X), _p(Y)) == 0
    return X, Y


def warp_perspective_coords(M, dsize):
    dw, dh = int(dsize[0]), int(dsize[1])
    M = np.ascontiguousarray(np.asarray(M, dtype=np.float64).reshape(9))
    X = np.zeros((dh, dw), np.int32)
    Y = np.zeros((dh, dw), np.int32)
    assert lib().vko_warp_perspective_coords(_p(M), dh, dw, _p(X), _p(Y)) == 0
    return X, Y


def warp_affine(src, M, dsize):
    """cv.warpAffine(src, M, dsize): M is the forward 2x3 matrix, dsize = (width, height)."""
    X, Y = warp_affine_coords(M, dsize)
    return _sample_fixed(src, X, Y)


def warp_perspective(src, M, dsize):
    """cv.warpPerspective(src, M, dsize): M is the forward 3x3 matrix, dsize = (width, height)."""
    X, Y = warp_perspective_coords(M, dsize)
    return _sample_fixed(src, X, Y)


def get_perspective_transform(pts_from, pts_to, solver=SOLVER_HYBRID):
    a = np.ascontiguousarray(np.asarray(pts_from, dtype=np.float32).reshape(8))
    b = np.ascontiguousarray(np.asarray(pts_to, dtype=np.float32).reshape(8))
    H = np.zeros(9, np.float64)
    lib().vko_get_perspective_transform(_p(a), _p(b), int(solver), _p(H))
    return H.reshape(3, 3)


def fill_poly(shape, pts, closed_form=False):
    """cv.fillPoly(zeros(shape, uint8), [pts], 1); pts int32 (N, 2) as (x, y), inside the array."""
    h, w = int(shape[0]), int(shape[1])
    pts = np.ascontiguousarray(np.asarray(pts, dtype=np.int32).reshape(-1, 2))
    img = np.zeros((h, w), np.uint8)
    fn = lib().vko_fill_poly_closed_form if closed_form else lib().vko_fill_poly
    assert fn(_p(img), h, w, _p(pts), int(pts.shape[0])) == 0
    return img


def grid_to_map(src_vertices, dst_vertices, dst_shape, solver=SOLVER_HYBRID, want_owner=False):
    """ImageGrid.generate_remap_params: vertices int32 (rows, cols, 2) as (x, y)."""
    sv = np.ascontiguousarray(src_vertices, dtype=np.int32)
    dv = np.ascontiguousarray(dst_vertices, dtype=np.int32)
    rows, cols = sv.shape[:2]
    dh, dw = int(dst_shape[0]), int(dst_shape[1])
    mx = np.zeros((dh, dw), np.float32)
    my = np.zeros((dh, dw), np.float32)
    owner = np.zeros((dh, dw), np.int32) if want_owner else None
    rc = lib().vko_grid_to_map(_p(sv), _p(dv), rows, cols, dh, dw, int(solver), _p(mx), _p(my),
                               _p(owner) if want_owner else None)
    assert rc == 0
    return (mx, my, owner) if want_owner else (mx, my)


def grid_project_points(src_vertices, dst_vertices, grid_size, points_xy, smooth_xy):
    """FuncImageGridBased.func_point (grid_rendering/interface.py:194-216) point by point, as the reference loops
    (distortion/interface.py:638-661): cell of the rounded point, get_trans_mat (grid_rendering/type.py:166-180),
    np.matmul(trans_mat, (smooth_x, smooth_y, 1.0)) and the two float64 divisions."""
    sv = np.asarray(src_vertices)
    dv = np.asarray(dst_vertices)
    out = np.empty((len(points_xy), 2), np.float64)
    cache = {}
    for i, ((x, y), (fx, fy)) in enumerate(zip(points_xy, smooth_xy)):
        r, c = int(y) // int(grid_size), int(x) // int(grid_size)
        if not (0 <= r < sv.shape[0] - 1 and 0 <= c < sv.shape[1] - 1):
            raise IndexError('list index out of range')
        if (r, c) not in cache:
            quad = lambda v: np.asarray([v[r, c], v[r, c + 1], v[r + 1, c + 1], v[r + 1, c]], np.float32)
            cache[(r, c)] = get_perspective_transform(quad(sv), quad(dv))
        tx, ty, t = np.matmul(cache[(r, c)], (float(fx), float(fy), 1.0))
        out[i] = (float(tx / t), float(ty / t))
    return out


def gaussian_kernel_q8(ksize, sigma):
    k = np.zeros(ksize, np.uint16)
    assert lib().vko_gaussian_kernel_q8(int(ksize), ctypes.c_double(sigma), _p(k)) == 0
    return k


def gaussian_blur(img, ksize, sigma):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    s3, squeeze = _as3(img)
    h, w, cn = s3.shape
    dst = np.zeros_like(s3)
    rc = lib().vko_gaussian_blur_u8(_p(s3), h, w, cn, _ss(w * cn), int(ksize), ctypes.c_double(sigma),
                                    _p(dst), _ss(w * cn))
    assert rc == 0
    return dst[:, :, 0] if squeeze else dst


def rgb2hsv_full(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    dst = np.zeros_like(img)
    lib().vko_rgb2hsv_full(_p(img), ctypes.c_size_t(img.size // 3), _p(dst))
    return dst


def hsv2rgb_full(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    dst = np.zeros_like(img)
    lib().vko_hsv2rgb_full(_p(img), ctypes.c_size_t(img.size // 3), _p(dst))
    return dst


def color_shift_rgb(img, delta):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    dst = np.zeros_like(img)
    lib().vko_color_shift_rgb(_p(img), ctypes.c_size_t(img.size // 3), int(delta), _p(dst))
    return dst


def mean_shift(img, delta, threshold=None, channels=None, cycle=False):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    cn = 1 if img.ndim == 2 else img.shape[2]
    chmask = 0
    if channels:
        for c in channels:
            chmask |= 1 << c
    dst = np.zeros_like(img)
    lib().vko_mean_shift_u8(_p(img), ctypes.c_size_t(img.size // cn), cn, int(delta),
                            int(threshold is not None), int(threshold or 0), int(cycle), chmask, _p(dst))
    return dst


def add_noise_i16(img, noise):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    noise = np.ascontiguousarray(noise, dtype=np.int16)
    assert img.shape == noise.shape
    dst = np.zeros_like(img)
    lib().vko_add_noise_i16(_p(img), _p(noise), ctypes.c_size_t(img.size), _p(dst))
    return dst


FILL_PLAIN, FILL_KEEP_MAX, FILL_KEEP_MIN = 0, 1, 2


def fill(dst, box, value, mask=None, alpha=1.0, mode=FILL_PLAIN):
    """fill_np_array on uint8 (HxW[xC]) or float32 (HxW) destinations; ``dst`` is modified in place.

    box = (up, left, height, width); value: tuple / scalar (constant) or array [h, w(, c)] of dst's dtype;
    mask: uint8 [h, w] or None; alpha: python float or float32 array [h, w]; mode: keep_max / keep_min.
    """
    assert dst.flags.c_contiguous and dst.flags.writeable
    up, left, bh, bw = (int(v) for v in box)
    mask_p, mask_step = None, 0
    if mask is not None:
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        assert mask.shape == (bh, bw)
        mask_p, mask_step = _p(mask), bw
    alpha_p, alpha_step, alpha_s = None, 0, 1.0
    if isinstance(alpha, np.ndarray):
        alpha = np.ascontiguousarray(alpha, dtype=np.float32)
        assert alpha.shape == (bh, bw)
        alpha_p, alpha_step = _p(alpha), bw
    else:
        alpha_s = float(alpha)
    if dst.dtype == np.float32:
        assert dst.ndim == 2
        h, w = dst.shape
        vplane_p, vstep, vconst = None, 0, 0.0
        if isinstance(value, np.ndarray):
            value = np.ascontiguousarray(value.astype(np.float32))
            assert value.shape == (bh, bw)
            vplane_p, vstep = _p(value), bw
        else:
            vconst = float(np.float32(value))
        rc = lib().vko_fill_f32(_p(dst), h, w, _ss(w), up, left, bh, bw, mask_p, _ss(mask_step), alpha_p,
                                _ss(alpha_step), ctypes.c_double(alpha_s), vplane_p, _ss(vstep),
                                ctypes.c_float(vconst), int(mode))
    else:
        assert dst.dtype == np.uint8
        d3, _ = _as3(dst)
        h, w, cn = d3.shape
        vplane_p, vstep, vconst_p = None, 0, None
        if isinstance(value, np.ndarray):
            value = np.ascontiguousarray(value.astype(np.uint8))
            v3, _ = _as3(value)
            assert v3.shape == (bh, bw, cn)
            vplane_p, vstep = _p(v3), bw * cn
        else:
            vconst = np.full(cn, value, dtype=np.uint8) if not isinstance(value, tuple) else np.asarray(
                value, dtype=np.uint8)
            assert vconst.shape == (cn,)
            vconst_p = _p(vconst)
        rc = lib().vko_fill_u8_mode(_p(d3), h, w, cn, _ss(w * cn), up, left, bh, bw, mask_p, _ss(mask_step),
                                    alpha_p, _ss(alpha_step), ctypes.c_double(alpha_s), vplane_p, _ss(vstep),
                                    vconst_p, int(mode))
    if rc == -2:
        raise RuntimeError(f'alpha={alpha_s} is invalid.')
    assert rc == 0, rc
    return dst


def _px3(fn, img, *args, out_channels=3):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    assert img.ndim == 3 and img.shape[2] == 3
    out = np.empty(img.shape if out_channels == 3 else img.shape[:2], np.uint8)
    rc = fn(_p(img), ctypes.c_size_t(img.shape[0] * img.shape[1]), *args, _p(out))
    assert rc == 0
    return out


def rgb2hls_full(img):
    return _px3(lib().vko_rgb2hls_full, img)


def hls2rgb_full(img):
    return _px3(lib().vko_hls2rgb_full, img)


def rgb2gray(img):
    return _px3(lib().vko_rgb2gray, img, out_channels=1)


def brightness_shift_rgb(img, delta):
    return _px3(lib().vko_brightness_shift_rgb, img, int(delta))


def color_balance_rgb(img, ratio):
    return _px3(lib().vko_color_balance_rgb, img, ctypes.c_double(ratio))


def _chmask(channels):
    mask = 0
    for c in channels or ():
        mask |= 1 << int(c)
    return mask


def complement(img, threshold=None, enable_threshold_lte=False, channels=None):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    cn = 1 if img.ndim == 2 else img.shape[2]
    out = np.empty_like(img)
    rc = lib().vko_pointwise_u8(_p(img), ctypes.c_size_t(img.size // cn), cn, 0,
                                -1 if threshold is None else int(threshold), int(bool(enable_threshold_lte)),
                                _chmask(channels), _p(out))
    assert rc == 0
    return out


def posterization(img, num_bits, channels=None):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    cn = 1 if img.ndim == 2 else img.shape[2]
    out = np.empty_like(img)
    rc = lib().vko_pointwise_u8(_p(img), ctypes.c_size_t(img.size // cn), cn, 1, int(num_bits), 0,
                                _chmask(channels), _p(out))
    assert rc == 0
    return out


def permute_channels(img, indices):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    cn = img.shape[2]
    packed = 0
    for c, idx in enumerate(indices):
        packed |= (int(idx) & 3) << (2 * c)
    out = np.empty_like(img)
    rc = lib().vko_pointwise_u8(_p(img), ctypes.c_size_t(img.size // cn), cn, 2, packed, 0, 0, _p(out))
    assert rc == 0
    return out


def impulse_noise(img, selector):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    cn = 1 if img.ndim == 2 else img.shape[2]
    selector = np.ascontiguousarray(selector, dtype=np.uint8)
    assert selector.shape == img.shape[:2]
    out = np.empty_like(img)
    rc = lib().vko_impulse_noise_u8(_p(img), ctypes.c_size_t(img.size // cn), cn, _p(selector), _p(out))
    assert rc == 0
    return out


def speckle_noise(img, noise):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    noise = np.ascontiguousarray(noise, dtype=np.float64)
    assert noise.shape == img.shape
    out = np.empty_like(img)
    rc = lib().vko_speckle_noise_u8(_p(img), ctypes.c_size_t(img.size), _p(noise), _p(out))
    assert rc == 0
    return out


def _resize_u8(fn, src, dsize_hw):
    dh, dw = int(dsize_hw[0]), int(dsize_hw[1])
    src = np.ascontiguousarray(src, dtype=np.uint8)
    s3, squeeze = _as3(src)
    sh, sw, cn = s3.shape
    dst = np.empty((dh, dw, cn), np.uint8)
    rc = fn(_p(s3), sh, sw, cn, _ss(sw * cn), _p(dst), dh, dw, _ss(dw * cn))
    assert rc == 0, rc
    return dst[:, :, 0] if squeeze else dst


def resize_linear(src, dsize_hw):
    """cv.resize(src, (dw, dh), interpolation=cv.INTER_LINEAR) on uint8."""
    return _resize_u8(lib().vko_resize_linear_u8, src, dsize_hw)


def resize_nearest(src, dsize_hw):
    """cv.resize(src, (dw, dh), interpolation=cv.INTER_NEAREST) on uint8."""
    return _resize_u8(lib().vko_resize_nearest_u8, src, dsize_hw)


def pixelation(img, ratio):
    """pixelation_image -- photometric/effect.py:61-79."""
    h, w = img.shape[:2]
    small = resize_linear(img, (round(h * ratio), round(w * ratio)))
    return resize_nearest(small, (h, w))


def zoom_in_blur(img, ratio, step, alpha):
    """zoom_in_blur_image -- photometric/blur.py:264-316, numpy statements of the reference around resize_cubic."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape[:2]
    acc = img.astype(np.uint16)
    count = 1
    for factor in np.arange(1 + step, 1 + ratio + step, step):
        rh, rw = round(h * factor), round(w * factor)
        big = resize_cubic(img, (rh, rw))
        up, left = (rh - h) // 2, (rw - w) // 2
        acc += big[up:up + h, left:left + w]
        count += 1
    out = (1 - alpha) * img + alpha * np.round(acc / count)
    return np.clip(out, 0, 255).astype(np.uint8)


def _select(img, channels):
    return img[:, :, list(channels)] if channels else img


def boundary_equalization(img, channels=None):
    """[numpy] boundary_equalization_image -- photometric/color.py:214-252, photometric/opt.py:49-57: per channel
    (v - min) * (255 / (max - min)) in float32, np.round, clip, uint8.  Plain numpy, the reference's statements."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    mat = _select(img, channels).astype(np.float32)
    if mat.ndim == 2:
        delta = mat.max() - mat.min()
        if delta == 0.0:
            return img
        mat -= mat.min()
        mat *= np.float32(255.0) / delta
    else:
        flat = mat.reshape(-1, mat.shape[-1])
        val_min, val_max = flat.min(axis=0), flat.max(axis=0)
        delta = val_max - val_min
        mask = delta > 0
        if not mask.any():
            return img
        mat[:, :, mask] -= val_min[mask]
        mat[:, :, mask] *= 255.0 / delta[mask]
    mat = np.clip(np.round(mat), 0, 255).astype(np.uint8)
    if channels:
        out = img.copy()
        out[:, :, list(channels)] = mat
        return out
    return mat


def equalize_hist_plane(plane):
    """[cv2] cv.equalizeHist -- imgproc/histogram.cpp: lut[i] = saturate_u8(cvRound(cumsum * scale)), scale =
    255 / (total - hist[first non-zero bin]) in float32; a single-valued plane comes back unchanged."""
    plane = np.ascontiguousarray(plane, dtype=np.uint8)
    hist = np.bincount(plane.ravel(), minlength=256).astype(np.int64)
    first = int(np.nonzero(hist)[0][0]) if plane.size else 0
    total = plane.size
    if plane.size == 0 or hist[first] == total:
        return plane.copy()
    scale = np.float32(255.0) / np.float32(total - hist[first])
    lut = np.zeros(256, np.uint8)
    acc = 0
    for i in range(first + 1, 256):
        acc += int(hist[i])
        lut[i] = np.uint8(min(max(int(np.rint(np.float32(acc) * scale)), 0), 255))
    return lut[plane]


def histogram_equalization(img, channels=None):
    """histogram_equalization_image -- photometric/color.py:255-285: cv.equalizeHist on every selected channel."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    if img.ndim == 2:
        return equalize_hist_plane(img)
    out = img.copy()
    for c in (channels if channels else range(img.shape[2])):
        out[:, :, c] = equalize_hist_plane(img[:, :, c])
    return out


def resize_cubic(src, dsize_hw):
    """cv.resize(src, (dw, dh), interpolation=cv.INTER_CUBIC) for uint8 HxW[xC] or float32 HxW."""
    dh, dw = int(dsize_hw[0]), int(dsize_hw[1])
    src = np.ascontiguousarray(src)
    if src.dtype == np.float32:
        assert src.ndim == 2
        sh, sw = src.shape
        dst = np.empty((dh, dw), np.float32)
        rc = lib().vko_resize_cubic_f32(_p(src), sh, sw, _ss(sw), _p(dst), dh, dw, _ss(dw))
    else:
        assert src.dtype == np.uint8
        s3, squeeze = _as3(src)
        sh, sw, cn = s3.shape
        dst = np.empty((dh, dw, cn), np.uint8)
        rc = lib().vko_resize_cubic_u8(_p(s3), sh, sw, cn, _ss(sw * cn), _p(dst), dh, dw, _ss(dw * cn))
        if squeeze:
            dst = dst[:, :, 0]
    assert rc == 0, rc
    return dst


INTER_NEAREST, INTER_LINEAR, INTER_CUBIC, INTER_AREA, INTER_LANCZOS4, INTER_LINEAR_EXACT, INTER_NEAREST_EXACT = range(7)


def resize(src, dsize_hw, interpolation):
    """cv.resize(src, (dw, dh), interpolation=<cv2 code>) for uint8 HxW[xC] or float32 HxW: the interpolations
    PageResizingStep samples (utility/opt.py:125-148) plus the plain NEAREST / LINEAR."""
    dh, dw = int(dsize_hw[0]), int(dsize_hw[1])
    src = np.ascontiguousarray(src)
    L = lib()
    if interpolation == INTER_CUBIC:
        return resize_cubic(src, dsize_hw)
    if src.dtype == np.float32:
        assert src.ndim == 2
        sh, sw = src.shape
        dst = np.empty((dh, dw), np.float32)
        if interpolation in (INTER_NEAREST, INTER_NEAREST_EXACT):
            fn = L.vko_resize_nearest_u8 if interpolation == INTER_NEAREST else L.vko_resize_nearest_exact_u8
            rc = fn(_p(src), sh, sw, 4, _ss(sw * 4), _p(dst), dh, dw, _ss(dw * 4))
        elif interpolation in (INTER_LINEAR, INTER_LINEAR_EXACT):     # no bit-exact float32 path: falls back to LINEAR
            rc = L.vko_resize_linear_f32(_p(src), sh, sw, _ss(sw), _p(dst), dh, dw, _ss(dw))
        elif interpolation == INTER_LANCZOS4:
            rc = L.vko_resize_lanczos4_f32(_p(src), sh, sw, _ss(sw), _p(dst), dh, dw, _ss(dw))
        elif interpolation == INTER_AREA:
            rc = L.vko_resize_area(_p(src), sh, sw, 1, _ss(sw), _p(dst), dh, dw, _ss(dw), 1)
        else:
            raise ValueError(interpolation)
        assert rc == 0, rc
        return dst
    assert src.dtype == np.uint8
    if interpolation == INTER_AREA:
        s3, squeeze = _as3(src)
        sh, sw, cn = s3.shape
        dst = np.empty((dh, dw, cn), np.uint8)
        rc = L.vko_resize_area(_p(s3), sh, sw, cn, _ss(sw * cn), _p(dst), dh, dw, _ss(dw * cn), 0)
        assert rc == 0, rc
        return dst[:, :, 0] if squeeze else dst
    fn = {INTER_NEAREST: L.vko_resize_nearest_u8, INTER_LINEAR: L.vko_resize_linear_u8,
          INTER_LANCZOS4: L.vko_resize_lanczos4_u8, INTER_LINEAR_EXACT: L.vko_resize_linear_exact_u8,
          INTER_NEAREST_EXACT: L.vko_resize_nearest_exact_u8}[interpolation]
    return _resize_u8(fn, src, dsize_hw)


def gaussian_blur_f32(mat, ksize, sigma):
    """cv.GaussianBlur(float32 HxW, (ksize, ksize), sigma) -- the kernel smoothing of blur.py:41-46."""
    mat = np.ascontiguousarray(mat, dtype=np.float32)
    h, w = mat.shape
    dst = np.empty_like(mat)
    assert lib().vko_gaussian_blur_f32(_p(mat), h, w, _ss(w), int(ksize), ctypes.c_double(sigma), _p(dst), _ss(w)) == 0
    return dst


def filter2d(img, kernel):
    """cv.filter2D(uint8 image, -1, float32 kernel)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    kernel = np.ascontiguousarray(kernel, dtype=np.float32)
    s3, squeeze = _as3(img)
    h, w, cn = s3.shape
    dst = np.empty_like(s3)
    rc = lib().vko_filter2d_u8(_p(s3), h, w, cn, _ss(w * cn), _p(kernel), kernel.shape[0], kernel.shape[1], _p(dst), _ss(w * cn))
    assert rc == 0, rc
    return dst[:, :, 0] if squeeze else dst


def _estimate_gaussian_kernel_size(sigma):
    kernel_size = max(3, round(3 * sigma) + 1)
    return kernel_size + 1 if kernel_size % 2 == 0 else kernel_size


def defocus_kernel(radius, anti_aliasing_sigma=0.5):
    """The disc kernel of defocus_blur_image (blur.py:85-112)."""
    aa = _estimate_gaussian_kernel_size(anti_aliasing_sigma)
    kernel_size = 2 * radius + 1 + aa // 2 * 2
    begin = -(kernel_size // 2)
    coords = np.arange(begin, begin + kernel_size)
    x, y = np.meshgrid(coords, coords)
    kernel = ((x ** 2 + y ** 2) <= radius ** 2).astype(np.float32)
    kernel /= kernel.sum()
    return gaussian_blur_f32(kernel, aa, anti_aliasing_sigma)


def rotation_matrix_2d(center, angle, scale):
    """cv.getRotationMatrix2D (imgproc/imgwarp.cpp): angle in degrees, counter-clockwise, float64."""
    angle = angle * np.pi / 180
    alpha, beta = np.cos(angle) * scale, np.sin(angle) * scale
    return np.asarray([[alpha, beta, (1 - alpha) * center[0] - beta * center[1]],
                       [-beta, alpha, beta * center[0] + (1 - alpha) * center[1]]], np.float64)


def motion_kernel(radius, angle, anti_aliasing_sigma=0.5):
    """The rotated line kernel of motion_blur_image (blur.py:135-176)."""
    aa = _estimate_gaussian_kernel_size(anti_aliasing_sigma)
    padding = aa // 2 * 2
    kernel_size = 2 * radius + 1
    half = padding // 2
    center, left = radius + half, half
    right = left + kernel_size - 1
    kernel_size += padding
    kernel = np.zeros((kernel_size, kernel_size), np.float32)
    kernel[center, left:right + 1] = 1.0
    trans_mat = rotation_matrix_2d((center, center), 360 - (angle % 360), 1.0)
    kernel = warp_affine(kernel, trans_mat, kernel.shape)
    kernel /= kernel.sum()
    return gaussian_blur_f32(kernel, aa, anti_aliasing_sigma)


def defocus_blur(img, radius, anti_aliasing_sigma=0.5):
    return filter2d(img, defocus_kernel(radius, anti_aliasing_sigma))


def motion_blur(img, radius, angle, anti_aliasing_sigma=0.5):
    return filter2d(img, motion_kernel(radius, angle, anti_aliasing_sigma))


def line_streak(img, thickness=1, gap=4, dash_thickness=0, dash_gap=0, color=(0, 0, 0), alpha=1.0,
                enable_vert=True, enable_hori=True):
    out = np.array(img, dtype=np.uint8, order='C')
    o3, _ = _as3(out)
    h, w, cn = o3.shape
    col = np.asarray(color, dtype=np.uint8)
    assert col.shape == (cn,)
    rc = lib().vko_line_streak_u8(_p(o3), h, w, cn, _ss(w * cn), int(thickness), int(gap), int(dash_thickness),
                                  int(dash_gap), _p(col), ctypes.c_double(alpha), int(enable_vert),
                                  int(enable_hori))
    assert rc == 0
    return out


def rodrigues(rvec):
    r = np.ascontiguousarray(np.asarray(rvec, dtype=np.float64).reshape(3))
    R = np.zeros(9, np.float64)
    lib().vko_rodrigues(_p(r), _p(R))
    return R.reshape(3, 3)


def project_points(pts3, rvec, tvec, fx, fy, cx=0.0, cy=0.0):
    p = np.ascontiguousarray(np.asarray(pts3, dtype=np.float64).reshape(-1, 3))
    r = np.ascontiguousarray(np.asarray(rvec, dtype=np.float64).reshape(3))
    t = np.ascontiguousarray(np.asarray(tvec, dtype=np.float64).reshape(3))
    out = np.zeros((p.shape[0], 2), np.float64)
    lib().vko_project_points(_p(p), ctypes.c_size_t(p.shape[0]), _p(r), _p(t), ctypes.c_double(fx),
                             ctypes.c_double(fy), ctypes.c_double(cx), ctypes.c_double(cy), _p(out))
    return out


def _centered_boxes(height, width, aspect_ratio, short_side_min, short_side_step):
    """generate_centered_boxes -- photometric/streak.py:109-145; boxes as (up, down, left, right)."""
    cy, cx = height // 2, width // 2
    boxes = []
    idx = 0
    while True:
        short_side = short_side_min + idx * short_side_step
        if aspect_ratio >= 1:
            h_min = short_side
            w_min = round(h_min * aspect_ratio)
        elif 0 < aspect_ratio < 1:
            w_min = short_side
            h_min = round(w_min / aspect_ratio)
        else:
            raise NotImplementedError()
        up = cy - h_min // 2
        down = up + h_min - 1
        left = cx - w_min // 2
        right = left + w_min - 1
        if (0 <= up and down < height) or (0 <= left and right < width):
            boxes.append((up, down, left, right))
            idx += 1
        else:
            break
    return boxes


def rectangle_streak(img, thickness=1, aspect_ratio=None, dash_thickness=0, dash_gap=0, short_side_min=10,
                     short_side_step=10, color=(0, 0, 0), alpha=1.0):
    """rectangle_streak_image -- photometric/streak.py:160-272 (numpy-only in the reference)."""
    out = np.array(img, dtype=np.uint8, order='C')
    H, W = out.shape[:2]
    if aspect_ratio is None:
        aspect_ratio = W / H
    vert = np.zeros((H, W), np.uint8)
    hori = np.zeros((H, W), np.uint8)
    for (up, down, left, right) in _centered_boxes(H, W, aspect_ratio, short_side_min, short_side_step):
        in_up, in_down = down - thickness + 1, up + thickness - 1
        in_left, in_right = right - thickness + 1, left + thickness - 1
        b_up, b_down = max(0, up), min(H - 1, down)
        if 0 <= in_right < W and b_up <= b_down:
            vert[b_up:b_down + 1, max(0, left):in_right + 1] = 1
        if 0 <= in_left < W and b_up <= b_down:
            vert[b_up:b_down + 1, in_left:min(W - 1, right) + 1] = 1
        b_left, b_right = max(0, in_right + 1), min(W - 1, in_left - 1)
        if 0 <= in_down < H and b_left <= b_right:
            hori[max(0, up):in_down + 1, b_left:b_right + 1] = 1
        if 0 <= in_up < H and b_left <= b_right:
            hori[in_up:min(H - 1, down) + 1, b_left:b_right + 1] = 1
    if dash_thickness > 0 and dash_gap > 0:
        step = dash_thickness + dash_gap
        for off in range(dash_gap):
            vert[off::step] = 0
            hori[:, off::step] = 0
    fill(out, (0, 0, H, W), tuple(color), mask=vert, alpha=float(alpha))
    fill(out, (0, 0, H, W), tuple(color), mask=hori, alpha=float(alpha))
    return out


def mls_project(src_handles_xy, dst_handles_xy, src_handles_smooth_xy, dst_handles_smooth_xy, vertices_xy):
    """SimilarityMlsPointProjector.project_point for every vertex -- geometric/mls.py:38-135."""
    p = np.ascontiguousarray(src_handles_xy, dtype=np.float32).reshape(-1, 2)
    q = np.ascontiguousarray(dst_handles_xy, dtype=np.float32).reshape(-1, 2)
    ps = np.ascontiguousarray(src_handles_smooth_xy, dtype=np.float64).reshape(-1, 2)
    qs = np.ascontiguousarray(dst_handles_smooth_xy, dtype=np.float64).reshape(-1, 2)
    v = np.ascontiguousarray(vertices_xy, dtype=np.float64).reshape(-1, 2)
    out = np.zeros_like(v)
    rc = lib().vko_mls_project(_p(p), _p(q), _p(ps), _p(qs), p.shape[0], _p(v), v.shape[0], _p(out))
    if rc > 0:
        raise FloatingPointError(f'vertex {rc - 1}: divide by zero')
    assert rc == 0
    return out


def philox2x32_10(c0, c1, key):
    out = (ctypes.c_uint32 * 2)()
    lib().vko_philox2x32_10(ctypes.c_uint32(c0), ctypes.c_uint32(c1), ctypes.c_uint32(key), out)
    return int(out[0]), int(out[1])


def noise_normal_table(std):
    table = np.zeros(65536, np.int16)
    lib().vko_noise_normal_table(ctypes.c_double(std), _p(table))
    return table


def noise_normal_i16(shape, std, seed):
    """The library's throughput-mode noise plane (its own definition, not the reference's values)."""
    h, w = int(shape[0]), int(shape[1])
    cn = int(shape[2]) if len(shape) > 2 else 1
    out = np.zeros(tuple(shape), np.int16)
    rc = lib().vko_noise_normal_i16(_p(out), h, w, cn, ctypes.c_double(std), ctypes.c_uint64(int(seed) & 0xffffffffffffffff))
    assert rc == 0
    return out


def ellipse_streak(img, thickness=1, aspect_ratio=None, short_side_min=10, short_side_step=10, color=(0, 0, 0), alpha=1.0):
    """ellipse_streak_image -- photometric/streak.py:296-330: [cv2] cv.ellipse per concentric box, then Mask.fill_image."""
    out = np.array(img, dtype=np.uint8, order='C')
    H, W = out.shape[:2]
    if aspect_ratio is None:
        aspect_ratio = W / H
    mask = np.zeros((H, W), np.uint8)
    for (up, down, left, right) in _centered_boxes(H, W, aspect_ratio, short_side_min, short_side_step):
        ellipse_outline(mask, (W // 2, H // 2), ((right + 1 - left) // 2, (down + 1 - up) // 2), thickness)
    fill(out, (0, 0, H, W), tuple(color), mask=mask, alpha=float(alpha))
    return out


def ellipse_outline(mask, center, axes, thickness):
    """[cv2] cv.ellipse(mask, center=(x, y), axes=(a, b), angle=0, startAngle=0, endAngle=360, color=1, thickness) in place
    (photometric/streak.py:312-324)."""
    assert mask.dtype == np.uint8 and mask.ndim == 2 and mask.flags.c_contiguous and mask.flags.writeable
    rc = lib().vko_ellipse_outline(_p(mask), mask.shape[0], mask.shape[1], int(center[0]), int(center[1]), int(axes[0]),
                                   int(axes[1]), int(thickness))
    assert rc == 0
    return mask


def ellipse_vertices(center, axes):
    """The rounded 16.16 outline vertices cv.ellipse hands to PolyLine for one axis-aligned full ellipse."""
    xy = np.zeros((74, 2), np.int64)
    n = lib().vko_ellipse_vertices(int(center[0]), int(center[1]), int(axes[0]), int(axes[1]), _p(xy))
    return xy[:n]


# ---------------------------------------------------------------------------------------------
# numpy Generator streams (oracle/np_random.c): PCG64 + ziggurat normal + uniform doubles + choice
# ---------------------------------------------------------------------------------------------
def np_state_words(rng):
    """{state lo, state hi, inc lo, inc hi} of a numpy Generator over PCG64."""
    st = rng.bit_generator.state
    assert st['bit_generator'] == 'PCG64'
    s, inc = st['state']['state'], st['state']['inc']
    m = (1 << 64) - 1
    return np.array([s & m, s >> 64, inc & m, inc >> 64], np.uint64)


def np_normal(words, n, loc, scale):
    """rng.normal(loc, scale, n) from the state words; returns (samples, words after, raw draws used)."""
    words = words.copy()
    out = np.empty(n, np.float64)
    lib().vko_np_normal.restype = ctypes.c_uint64
    used = lib().vko_np_normal(_p(words), ctypes.c_int64(n), ctypes.c_double(loc), ctypes.c_double(scale), _p(out))
    return out, words, used


def np_normal_i16(words, n, std):
    words = words.copy()
    out = np.empty(n, np.int16)
    lib().vko_np_normal_i16.restype = ctypes.c_uint64
    used = lib().vko_np_normal_i16(_p(words), ctypes.c_int64(n), ctypes.c_double(std), _p(out))
    return out, words, used


def np_random(words, n):
    words = words.copy()
    out = np.empty(n, np.float64)
    lib().vko_np_random.restype = ctypes.c_uint64
    used = lib().vko_np_random(_p(words), ctypes.c_int64(n), _p(out))
    return out, words, used


def np_choice_cdf(words, n, cdf):
    words = words.copy()
    cdf = np.ascontiguousarray(cdf, np.float64)
    out = np.empty(n, np.uint8)
    lib().vko_np_choice_cdf.restype = ctypes.c_uint64
    used = lib().vko_np_choice_cdf(_p(words), ctypes.c_int64(n), _p(cdf), len(cdf), _p(out))
    return out, words, used


def np_advance(words, delta):
    words = words.copy()
    lib().vko_np_advance(_p(words), ctypes.c_uint64(delta & ((1 << 64) - 1)), ctypes.c_uint64(delta >> 64))
    return words
