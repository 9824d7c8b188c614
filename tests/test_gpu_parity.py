"""Parity of the HIP path (through the C ABI of libvkx.so) against the CPU oracle.  Needs an MI355X.

Bar: bit-exact for uint8 / index planes; float32 (ScoreMap) planes are compared bit-for-bit as well
(the documented tolerance against cv2 itself is 2 ulp)."""
import os

import numpy as np
import pytest
from numpy.random import default_rng

import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def N():
    from vkit_amd import _native
    _native.default_ctx()
    return _native


@pytest.fixture(scope='module')
def grids(golden_dir):
    """name -> (src_vertices, dst_vertices, dst_shape, src_shape)."""
    out = {}
    M = np.load(os.path.join(golden_dir, 'mls_states.npz'))
    for key in ('64x64_s0_l5', '96x80_s1_l8', '130x257_s2_l10', '512x512_s0_l5', '300x200_s3_l1'):
        dv = M[key + '_dst_grid']
        sv = M[key + '_src_grid']
        h, w = (int(v) for v in key.split('_')[0].split('x'))
        out['mls_' + key] = (sv, dv, (int(dv[..., 1].max()) + 1, int(dv[..., 0].max()) + 1), (h, w))
    S = np.load(os.path.join(golden_dir, 'structure_oracle_patched.npz'))
    for key in [k[:-len('_dst_grid')] for k in S.files if k.startswith('cam_') and k.endswith('_dst_grid')]:
        dv = S[key + '_dst_grid']
        h, w = (int(v) for v in key.split('_')[2].split('x'))
        gs = max(15, int(0.01 * max(h, w)))
        ys = list(range(0, h, gs)) + ([h - 1] if (h - 1) % gs else [])
        xs = list(range(0, w, gs)) + ([w - 1] if (w - 1) % gs else [])
        sv = np.array([[(x, y) for x in xs] for y in ys], np.int32)
        assert sv.shape == dv.shape, key
        out[key] = (sv, dv, tuple(int(v) for v in S[key + '_result_shape']), (h, w))
    return out


def synthetic_grid(h, w, gs, amp, seed=0):
    ys = list(range(0, h, gs)) + ([h - 1] if (h - 1) % gs else [])
    xs = list(range(0, w, gs)) + ([w - 1] if (w - 1) % gs else [])
    sv = np.array([[(x, y) for x in xs] for y in ys], np.int32)
    rng = default_rng(seed)
    fy, fx = sv[..., 1] / h, sv[..., 0] / w
    px, py = rng.uniform(0, 6.28, 2)
    dx = amp * np.sin(2.1 * np.pi * fy + px) * np.cos(1.3 * np.pi * fx) + 0.04 * amp * sv[..., 1] / gs
    dy = amp * np.cos(1.7 * np.pi * fx + py) * np.sin(2.3 * np.pi * fy)
    dv = np.stack([np.rint(sv[..., 0] * 1.03 + dx), np.rint(sv[..., 1] * 1.05 + dy)], -1).astype(np.int32)
    dv[..., 0] -= dv[..., 0].min()
    dv[..., 1] -= dv[..., 1].min()
    return sv, dv, (int(dv[..., 1].max()) + 1, int(dv[..., 0].max()) + 1)


# ------------------------------------------------------------------------------------------ remap / warps
def test_remap_u8_and_f32_random_maps(N):
    rng = default_rng(0)
    for (sh, sw, dh, dw) in [(37, 53, 41, 67), (1, 1, 5, 5), (2, 129, 64, 64), (200, 3, 33, 300)]:
        mx = rng.uniform(-3, sw + 3, (dh, dw)).astype(np.float32)
        my = rng.uniform(-3, sh + 3, (dh, dw)).astype(np.float32)
        mx[0, 0], my[0, 0] = 1e30, -1e30
        mx[-1, -1] = np.nan
        mx[0, -1] = np.inf
        mx[1 % dh, 0] = sw - 1
        my[1 % dh, 0] = sh - 1
        for cn in (1, 3, 4):
            src = rng.integers(0, 256, (sh, sw, cn) if cn > 1 else (sh, sw), dtype=np.uint8)
            assert (N.remap(src, mx, my) == O.remap(src, mx, my)).all(), (sh, sw, cn)
        srcf = rng.random((sh, sw), dtype=np.float32)
        a, b = N.remap(srcf, mx, my), O.remap(srcf, mx, my)
        assert (a.view(np.uint32) == b.view(np.uint32)).all()


def test_remap_mask_is_bilinear_threshold(N):
    rng = default_rng(1)
    m = (rng.random((50, 60)) < 0.5).astype(np.uint8)
    mx = rng.uniform(0, 59, (70, 80)).astype(np.float32)
    my = rng.uniform(0, 49, (70, 80)).astype(np.float32)
    got = N.remap(m, mx, my)
    assert (got == O.remap(m, mx, my)).all()
    assert set(np.unique(got)) <= {0, 1}


def test_warp_affine(N):
    rng = default_rng(2)
    src = rng.integers(0, 256, (61, 47, 3), dtype=np.uint8)
    srcf = rng.random((61, 47), dtype=np.float32)
    for M, dsize in [
        (np.array([[0.8660253882408142, -0.5, 31.0], [0.5, 0.8660253882408142, 0.0]]), (73, 77)),
        (np.array([[1, -0.5773502588272095, 36.0], [0, 1, 0]]), (84, 61)),
        (np.array([[0, -1, 60], [1, 0, 0]], float), (61, 47)),
        (np.array([[1e-9, 0, 0], [0, 1e-9, 0]]), (8, 8)),
        (np.array([[0, 0, 0], [0, 0, 0]], float), (8, 8)),
    ]:
        assert (N.warp_affine(src, M, dsize) == O.warp_affine(src, M, dsize)).all()
        assert (N.warp_affine(src[..., 0], M, dsize) == O.warp_affine(src[..., 0], M, dsize)).all()
        a, b = N.warp_affine(srcf, M, dsize), O.warp_affine(srcf, M, dsize)
        assert (a.view(np.uint32) == b.view(np.uint32)).all()


def test_warp_perspective(N):
    rng = default_rng(3)
    src = rng.integers(0, 256, (90, 130, 3), dtype=np.uint8)
    srcf = rng.random((90, 130), dtype=np.float32)
    pts_src = np.array([[0, 0], [129, 0], [129, 89], [0, 89]], np.float32)
    for shrink in (0, 7, 31):
        pts_dst = np.array([[0, shrink // 2], [129, 0], [129, 89], [0, 89 - (shrink - shrink // 2)]], np.float32)
        M = O.get_perspective_transform(pts_src, pts_dst, O.SOLVER_JACOBI)
        for dsize in ((130, 90), (200, 33)):
            assert (N.warp_perspective(src, M, dsize) == O.warp_perspective(src, M, dsize)).all()
            a, b = N.warp_perspective(srcf, M, dsize), O.warp_perspective(srcf, M, dsize)
            assert (a.view(np.uint32) == b.view(np.uint32)).all()
    singular = np.zeros((3, 3))
    assert (N.warp_perspective(src, singular, (16, 16)) == O.warp_perspective(src, singular, (16, 16))).all()


# ------------------------------------------------------------------------------------------ grid distortions
def test_grid_to_map_on_reference_grids(N, grids):
    for name, (sv, dv, dshape, _sshape) in grids.items():
        mx, my, ow = N.grid_to_map(sv, dv, dshape, want_owner=True)
        ex, ey, eo = O.grid_to_map(sv, dv, dshape, want_owner=True)
        assert (ow == eo).all(), name
        assert (mx.view(np.uint32) == ex.view(np.uint32)).all(), name
        assert (my.view(np.uint32) == ey.view(np.uint32)).all(), name


def test_grid_to_map_degenerate_and_folded_cells(N):
    # collapsed last column, a bow-tie cell and a zero-area cell: the Jacobi-SVD / den == 0 paths
    sv = np.array([[(x, y) for x in (0, 15, 30, 31)] for y in (0, 15, 30, 45)], np.int32)
    dv = sv.copy()
    dv[:, 3, 0] = dv[:, 2, 0]               # last column collapses onto its neighbour
    dv[1, 1] = (22, 20)
    dv[2, 1] = (9, 24)                      # fold
    dv[3, 0] = dv[3, 1]                     # degenerate bottom-left cell
    dshape = (int(dv[..., 1].max()) + 1, int(dv[..., 0].max()) + 1)
    mx, my, ow = N.grid_to_map(sv, dv, dshape, want_owner=True)
    ex, ey, eo = O.grid_to_map(sv, dv, dshape, want_owner=True)
    assert (ow == eo).all()
    ok = np.isfinite(ex) & np.isfinite(ey)
    assert (mx[ok].view(np.uint32) == ex[ok].view(np.uint32)).all()
    assert (my[ok].view(np.uint32) == ey[ok].view(np.uint32)).all()
    assert (np.isfinite(mx) == np.isfinite(ex)).all()


def test_grid_remap_shared_map_three_elements(N, grids):
    rng = default_rng(4)
    for name in ('mls_96x80_s1_l8', 'mls_130x257_s2_l10', 'cam_cubic_200x312_s3_l9', 'cam_fold_128x160_s2_l6'):
        sv, dv, dshape, (h, w) = grids[name]
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        mask = (rng.random((h, w)) < 0.5).astype(np.uint8)
        score = rng.random((h, w), dtype=np.float32)
        got = N.grid_remap([img, mask, score], sv, dv, dshape)
        mx, my = O.grid_to_map(sv, dv, dshape)
        assert (got[0] == O.remap(img, mx, my)).all(), name
        assert (got[1] == O.remap(mask, mx, my)).all(), name
        assert (got[2].view(np.uint32) == O.remap(score, mx, my).view(np.uint32)).all(), name


def test_grid_remap_identity_is_identity_full_size(N):
    h = w = 2048
    ys = list(range(0, h, 20)) + [h - 1]
    xs = list(range(0, w, 20)) + [w - 1]
    v = np.array([[(x, y) for x in xs] for y in ys], np.int32)
    img = default_rng(1000).integers(0, 256, (h, w, 3), dtype=np.uint8)
    out = N.grid_remap([img], v, v, (h, w))[0]
    assert (out == img).all()


def test_grid_remap_full_size_against_oracle(N):
    h = w = 2048
    sv, dv, dshape = synthetic_grid(h, w, 20, 14.0, seed=7)
    img = default_rng(1001).integers(0, 256, (h, w, 3), dtype=np.uint8)
    got = N.grid_remap([img], sv, dv, dshape)[0]
    mx, my, eo = O.grid_to_map(sv, dv, dshape, want_owner=True)
    want = O.remap(img, mx, my)
    assert got.shape == want.shape
    assert (got == want).all()
    gx, gy, go = N.grid_to_map(sv, dv, dshape, want_owner=True)
    assert (go == eo).all() and (gx.view(np.uint32) == mx.view(np.uint32)).all()


def test_grid_4096_shared_state_three_outputs(N):
    h = w = 4096
    sv, dv, dshape = synthetic_grid(h, w, 40, 30.0, seed=9)
    rng = default_rng(5)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    mask = (rng.random((h, w)) < 0.5).astype(np.uint8)
    score = rng.random((h, w), dtype=np.float32)
    got = N.grid_remap([img, mask, score], sv, dv, dshape)
    mx, my = O.grid_to_map(sv, dv, dshape)
    assert (got[0] == O.remap(img, mx, my)).all()
    assert (got[1] == O.remap(mask, mx, my)).all()
    assert (got[2].view(np.uint32) == O.remap(score, mx, my).view(np.uint32)).all()


# ------------------------------------------------------------------------------------------ photometric
def test_gaussian_blur(N):
    rng = default_rng(6)
    # (RGB planes take the packed tile kernel: widths around its 62 / 60 / 58-column tiles and 4-pixel store groups)
    for shape in [(40, 50, 3), (33, 65), (1, 17, 3), (19, 1), (2, 2, 3), (3, 200, 4), (5, 5), (70, 61, 3), (35, 62, 3), (64, 123, 3),
                  (97, 258, 3), (31, 3, 3)]:
        img = rng.integers(0, 256, shape, dtype=np.uint8)
        img[-1, -1] = 255
        for ksize, sigma in [(3, 0.5), (3, 0.7), (5, 0.9), (5, 1.0), (7, 2.0)]:
            assert (N.gaussian_blur(img, ksize, sigma) == O.gaussian_blur(img, ksize, sigma)).all(), (shape, ksize)


def test_color_shift_and_cvt(N):
    rng = default_rng(7)
    img = rng.integers(0, 256, (70, 90, 3), dtype=np.uint8)
    # every (r, g, b) corner case: grays, primaries, v == r == g ties
    special = np.array([[[0, 0, 0], [255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [7, 7, 3], [3, 7, 7],
                         [7, 3, 7], [1, 0, 0], [254, 255, 255]]], np.uint8)
    for im in (img, special):
        assert (N.cvt_rgb_hsv(im, True) == O.rgb2hsv_full(im)).all()
        assert (N.cvt_rgb_hsv(im, False) == O.hsv2rgb_full(im)).all()
        for delta in (0, 37, -37, 127, -127, 255, 256, -300):
            assert (N.color_shift_rgb(im, delta) == O.color_shift_rgb(im, delta)).all(), delta


def test_color_conversions_exhaustive_hsv_cube_slice(N):
    # all 2^24 HSV triples -> RGB (the float path): 4096 x 4096 image
    h = np.arange(256, dtype=np.uint8)
    cube = np.stack(np.meshgrid(h, h, h, indexing='ij'), -1).reshape(4096, 4096, 3)
    assert (N.cvt_rgb_hsv(cube, False) == O.hsv2rgb_full(cube)).all()
    assert (N.cvt_rgb_hsv(cube, True) == O.rgb2hsv_full(cube)).all()


def test_mean_shift_noise_streak(N, golden_dir):
    G = np.load(os.path.join(golden_dir, 'numpy_path.npz'))
    src = G['photo_src']
    assert (N.mean_shift(src, 100) == G['mean_shift_100']).all()
    assert (N.mean_shift(src, -60, channels=[1]) == G['mean_shift_m60_c1']).all()
    assert (N.mean_shift(src, 128, threshold=127, cycle=True) == G['mean_shift_128_cycle_thr']).all()
    assert (N.mean_shift(src, -128, threshold=128, cycle=True) == G['mean_shift_m128_cycle_thr']).all()
    assert (N.mean_shift(src, 40, threshold=200) == G['mean_shift_40_thr200']).all()
    assert (N.mean_shift(src, 37, channels=[0], cycle=True) == G['color_shift_hsv_37']).all()
    assert (N.add_noise_i16(src, G['noise_std10_seed1_plane']) == G['noise_std10_seed1']).all()
    big = default_rng(8).integers(-400, 400, src.shape).astype(np.int16)
    assert (N.add_noise_i16(src, big) == O.add_noise_i16(src, big)).all()
    assert (N.line_streak(src, 1, 4, 0, 0, (0, 0, 0), 0.3, True, True) == G['line_streak_a03']).all()
    assert (N.line_streak(src, 2, 5, 3, 2, (9, 8, 7), 0.5, True, True) == G['line_streak_dash']).all()
    assert (N.line_streak(src, 1, 3, 0, 0, (1, 2, 3), 1.0, True, False) == G['line_streak_vert_a1']).all()
    gray = src[..., 0].copy()
    assert (N.line_streak(gray, 2, 3, 0, 0, (9,), 0.25, True, True) ==
            O.line_streak(gray, 2, 3, color=(9,), alpha=0.25)).all()


def test_fill_layers_against_reference_goldens(N, golden_dir):
    G = np.load(os.path.join(golden_dir, 'numpy_path.npz'))
    page0, m, val = G['fill_sm_in'], G['fill_mask_mask'], G['fill_mask_value']

    def run(layers):
        page = page0.copy()
        N.fill(page, layers)
        return page

    assert (run([N.make_layer(tuple(G['fill_sm_box']), 3, (10, 20, 30), alpha=G['fill_sm_alpha'])]) == G['fill_sm_out']).all()
    assert (run([N.make_layer((0, 0, 40, 56), 3, val, mask=m)]) == G['fill_mask_out']).all()
    for a in (0.3, 0.5, 0.999, 1.0, 0.0):
        assert (run([N.make_layer((0, 0, 40, 56), 3, (7, 99, 250), mask=m, alpha=a)]) == G[f'fill_mask_const_a{a}']).all()
        assert (run([N.make_layer((0, 0, 40, 56), 3, val, mask=m, alpha=a)]) == G[f'fill_mask_img_a{a}']).all()
    for a in (0.25, 1.0):
        assert (run([N.make_layer((3, 8, 10, 17), 3, G['fill_box_value'], alpha=a)]) == G[f'fill_box_img_a{a}']).all()
    assert (run([N.make_layer((3, 8, 10, 17), 3, (200, 10, 10), mask=G['fill_box_mask'], alpha=0.7)])
            == G['fill_box_mask_out']).all()
    assert (run([N.make_layer((3, 8, 10, 17), 3, (200, 10, 10), alpha=G['fill_box_alpha'])]) == G['fill_box_alpha_out']).all()
    with pytest.raises(N.VkxError):
        run([N.make_layer((0, 0, 4, 4), 3, (1, 2, 3), alpha=1.5)])
    with pytest.raises(N.VkxError):
        run([N.make_layer((38, 0, 4, 4), 3, (1, 2, 3))])


def test_fill_page_of_text_layers(N):
    """C4-shaped composite: 1024^2 page, 64 text-line layers 32x512, ~30 % non-zero alpha, glyph colour."""
    rng = default_rng(9)
    page = np.full((1024, 1024, 3), int(rng.integers(127, 256)), np.uint8)
    want = page.copy()
    layers = []
    for _ in range(64):
        up, left = int(rng.integers(0, 1024 - 32)), int(rng.integers(0, 1024 - 512))
        alpha = rng.random((32, 512), dtype=np.float32)
        alpha[rng.random((32, 512)) > 0.3] = 0
        layers.append(N.make_layer((up, left, 32, 512), 3, (10, 20, 30), alpha=alpha))
        O.fill(want, (up, left, 32, 512), (10, 20, 30), alpha=alpha)
    N.fill(page, layers)
    assert (page == want).all()


def test_chain_c3_one_full_size_image(N):
    """camera-like grid remap + gaussian_blur + color_shift + gaussion_noise on one 2048^2 image."""
    h = w = 2048
    sv, dv, dshape = synthetic_grid(h, w, 20, 18.0, seed=3)
    img = default_rng(1002).integers(0, 256, (h, w, 3), dtype=np.uint8)
    got = N.grid_remap([img], sv, dv, dshape)[0]
    got = N.gaussian_blur(got, 5, 1.0)
    got = N.color_shift_rgb(got, 37)
    noise = np.round(default_rng(5002).normal(0, 10.0, got.shape)).astype(np.int16)
    got = N.add_noise_i16(got, noise)
    mx, my = O.grid_to_map(sv, dv, dshape)
    want = O.add_noise_i16(O.color_shift_rgb(O.gaussian_blur(O.remap(img, mx, my), 5, 1.0), 37), noise)
    assert (got == want).all()


# ------------------------------------------------------------------------------------------ batched chain
def _chain_case(N, grids, names, seed, blur_sigmas, hue_deltas, with_noise, streaks=(None,)):
    from vkit_amd.batch import ChainBatch

    class _State:  # the two attributes ChainBatch reads from an image-grid state
        def __init__(self, sv, dv, dshape):
            from types import SimpleNamespace
            self.src_image_grid = SimpleNamespace(vertices=sv)
            self.dst_image_grid = SimpleNamespace(vertices=dv)
            self.result_shape = dshape

    rng = default_rng(seed)
    batch = ChainBatch()
    expect = []
    for i, name in enumerate(names):
        sv, dv, dshape, (h, w) = grids[name]
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        sigma = blur_sigmas[i % len(blur_sigmas)]
        delta = hue_deltas[i % len(hue_deltas)]
        noise = rng.integers(-60, 60, tuple(dshape) + (3,)).astype(np.int16) if with_noise[i % len(with_noise)] else None
        streak = streaks[i % len(streaks)]
        batch.add(img, _State(sv, dv, dshape), blur_sigma=sigma, hue_delta=delta, noise=noise, streak=streak)
        mx, my = O.grid_to_map(sv, dv, dshape)
        want = O.remap(img, mx, my)
        if sigma is not None:
            from vkit_amd.mechanism.distortion.photometric.blur import _estimate_gaussian_kernel_size
            want = O.gaussian_blur(want, _estimate_gaussian_kernel_size(sigma), sigma)
        if delta is not None:
            want = O.color_shift_rgb(want, delta)
        if noise is not None:
            want = O.add_noise_i16(want, noise)
        if streak is not None:
            want = O.line_streak(want, streak.thickness, streak.gap, streak.dash_thickness, streak.dash_gap, streak.color,
                                 streak.alpha, streak.enable_vert, streak.enable_hori)
        expect.append(want)
    batch.run()
    for i, want in enumerate(expect):
        got = batch.result(i)
        assert got.shape == want.shape, names[i]
        bad = int((got != want).sum())
        assert bad == 0, (names[i], bad, np.argwhere((got != want).any(axis=2))[:5].tolist())
    batch.close()


def test_chain_batch_ragged_against_oracle(N, grids):
    names = sorted(grids)
    _chain_case(N, grids, names, 11, [1.0, 0.6, None, 2.0], [37, None, -120], [True, False, True])
    _chain_case(N, grids, names[::-1], 12, [None], [None], [False])       # remap only
    _chain_case(N, grids, names, 13, [0.9], [255], [True])


def test_chain_batch_fuzz(N):
    """Random shapes / lattices / parameters in ragged batches: tile borders, partial tiles, windows wider than the
    image, empty tiles (lattices that leave part of the result uncovered), folds, every blur radius of the fused path."""
    rng = default_rng(2024)
    sigmas = [None, 0.5, 0.7, 1.0, 1.3, 2.0]
    for round_ in range(6):
        grids, names = {}, []
        for i in range(10):
            h, w = int(rng.integers(2, 330)), int(rng.integers(2, 330))
            gs = int(rng.integers(5, 45))
            sv, dv, dshape = synthetic_grid(h, w, gs, float(rng.uniform(0, 14)), seed=int(rng.integers(1 << 30)))
            kind = int(rng.integers(0, 4))
            if kind == 0:      # leave a margin: the result is larger than the warped page (uncovered tiles)
                dv = dv + np.asarray([int(rng.integers(0, 150)), int(rng.integers(0, 150))], np.int32)
                dshape = (int(dv[..., 1].max()) + 1 + int(rng.integers(0, 140)),
                          int(dv[..., 0].max()) + 1 + int(rng.integers(0, 140)))
            elif kind == 1 and dv.shape[0] > 2 and dv.shape[1] > 2:   # a fold: one interior vertex jumps
                r, c = int(rng.integers(1, dv.shape[0] - 1)), int(rng.integers(1, dv.shape[1] - 1))
                dv[r, c] = dv[r - 1, c - 1]
            name = f'f{round_}_{i}'
            grids[name] = (sv, dv, dshape, (h, w))
            names.append(name)
        from vkit_amd.mechanism.distortion import LineStreakConfig
        streaks = []
        for k in range(10):
            if rng.random() < 0.5:
                streaks.append(None)
                continue
            dash = rng.random() < 0.5
            streaks.append(LineStreakConfig(
                thickness=int(rng.integers(1, 4)), gap=int(rng.integers(1, 30)),
                dash_thickness=int(rng.integers(1, 20)) if dash else 0, dash_gap=int(rng.integers(1, 9)) if dash else 0,
                color=tuple(int(v) for v in rng.integers(0, 256, 3)),
                alpha=float(rng.choice([1.0, 0.5, 0.25, float(rng.random())])),
                enable_vert=bool(rng.random() < 0.8), enable_hori=bool(rng.random() < 0.8)))
        _chain_case(N, grids, names, 1000 + round_,
                    [sigmas[int(k)] for k in rng.integers(0, len(sigmas), 10)],
                    [None if k % 3 == 0 else int(k) - 128 for k in rng.integers(0, 256, 10)],
                    [bool(k) for k in rng.integers(0, 2, 10)], streaks)


def test_chain_batch_degenerate_grid(N):
    sv = np.array([[(x, y) for x in (0, 15, 30, 31)] for y in (0, 15, 30, 45)], np.int32)
    dv = sv.copy()
    dv[:, 3, 0] = dv[:, 2, 0]
    dv[1, 1] = (22, 20)
    dv[2, 1] = (9, 24)
    dv[3, 0] = dv[3, 1]
    dshape = (int(dv[..., 1].max()) + 1, int(dv[..., 0].max()) + 1)
    grids = {'deg': (sv, dv, dshape, (46, 32))}
    # non-finite map entries (den ~ 0 in degenerate cells) are compared through the pixels they produce
    _chain_case(N, grids, ['deg'], 14, [1.0], [99], [True])


def test_chain_batch_rim_stress(N):
    """Border tiles whose pixels map outside the source (the per-tap rim sampler next to the paired fast loads), every
    blur radius, the same batch issued repeatedly: a timing-dependent defect (a missed wait state, a stale register) shows
    up as a run-to-run difference or a deviation from the oracle.  Round 2 found one this way in the 3-tap variant."""
    rng = default_rng(77)
    grids, names = {}, []
    for i, (h, w, gs, amp) in enumerate([(200, 260, 15, 9.0), (130, 500, 15, 12.0), (333, 190, 20, 6.0)]):
        sv, dv, dshape = synthetic_grid(h, w, gs, amp, seed=100 + i)
        sv = sv.copy()
        # pull the source lattice outwards: its rim samples the outside of the source image (cv.remap's zero border)
        sv[0, :, 1] -= 3; sv[-1, :, 1] += 3; sv[:, 0, 0] -= 3; sv[:, -1, 0] += 3
        for k, sigma in enumerate((0.7, 1.0, 2.0)):
            name = f'rim{i}_{k}'
            grids[name] = (sv, dv, dshape, (h, w))
            names.append(name)
    for round_ in range(8):
        _chain_case(N, grids, names, 3000 + round_, [0.7, 1.0, 2.0], [None, 37, -5], [round_ % 2 == 0])


def test_chain_batch_full_size(N):
    sv, dv, dshape = synthetic_grid(2048, 2048, 20, 18.0, seed=3)
    grids = {'big': (sv, dv, dshape, (2048, 2048))}
    _chain_case(N, grids, ['big'], 15, [1.0], [37], [True])
    sv, dv, dshape = synthetic_grid(1000, 1500, 15, 9.0, seed=4)
    grids = {'odd': (sv, dv, dshape, (1000, 1500))}
    _chain_case(N, grids, ['odd'], 16, [0.7], [-5], [False])


def test_grid_remap_global_plane_path_subprocess(N):
    """vkx_grid_remap takes the tile kernel by default; VKX_GRID_GLOBAL=1 forces the global-ownership-plane kernels.
    Both must give the oracle's pixels for uint8 x 1 / 3 / 4 channels and float32 through one shared lattice."""
    import subprocess
    import sys
    code = (
        "import os, sys; sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'));"
        "import numpy as np; import oracle as O; import test_gpu_parity as T; from vkit_amd import _native as N;"
        "rng = np.random.default_rng(9);"
        "sv, dv, ds = T.synthetic_grid(333, 410, 17, 8.0, seed=6);"
        "mats = [rng.integers(0, 256, (333, 410, 3), dtype=np.uint8), rng.integers(0, 2, (333, 410), dtype=np.uint8),"
        "        rng.random((333, 410), dtype=np.float32), rng.integers(0, 256, (333, 410, 4), dtype=np.uint8)];"
        "outs = N.grid_remap(mats, sv, dv, ds); mx, my = O.grid_to_map(sv, dv, ds);"
        "assert all((o == O.remap(m, mx, my)).all() for o, m in zip(outs, mats)); print('ok')")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for env in (dict(os.environ), dict(os.environ, VKX_GRID_GLOBAL='1')):
        out = subprocess.run([sys.executable, '-c', code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and 'ok' in out.stdout, out.stdout + out.stderr


def test_chain_batch_staged_path_subprocess(N):
    """The per-stage fallback of vkx_chain_rgb_batch_dev (VKX_CHAIN_STAGED=1) gives the same pixels."""
    import subprocess
    import sys
    code = (
        "import os, sys; sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'));"
        "import numpy as np; import test_gpu_parity as T; from vkit_amd import _native as N;"
        "g = {}; sv, dv, ds = T.synthetic_grid(300, 420, 15, 7.0, seed=5); g['a'] = (sv, dv, ds, (300, 420));"
        "from vkit_amd.mechanism.distortion import LineStreakConfig as L;"
        "T._chain_case(N, g, ['a'], 17, [1.0], [37], [True], [L(thickness=2, gap=9, alpha=0.4, color=(9, 200, 30))]);"
        "T._chain_case(N, g, ['a'], 18, [None], [None], [True]);"
        # a numpy stream kept as the generator's tile buffer: the staged path expands it to its plane
        "import oracle as O; from vkit_amd.batch import ChainBatch; from types import SimpleNamespace as NS;"
        "img = np.random.default_rng(3).integers(0, 256, (300, 420, 3), dtype=np.uint8);"
        "st = NS(result_shape=ds, src_image_grid=NS(vertices=sv), dst_image_grid=NS(vertices=dv));"
        "b = ChainBatch(); b.add(img, st, blur_sigma=1.0, hue_delta=37, noise_std=9.0, noise_rng=np.random.default_rng(4)); b.run();"
        "assert b._items[0].noise_tiled == 1;"
        "mx, my = O.grid_to_map(sv, dv, ds); pl = np.round(np.random.default_rng(4).normal(0, 9.0, tuple(ds) + (3,))).astype(np.int16);"
        "assert (b.result(0) == O.add_noise_i16(O.color_shift_rgb(O.gaussian_blur(O.remap(img, mx, my), 5, 1.0), 37), pl)).all();"
        "print('ok')")
    env = dict(os.environ, VKX_CHAIN_STAGED='1')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, '-c', code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and 'ok' in out.stdout, out.stdout + out.stderr


def test_grid_project_points_matches_oracle(N):
    """vkx_grid_project_points against the point-by-point restatement: float64 results within 4 ulp (the reference's
    np.matmul is a BLAS dgemv whose accumulation order is build dependent), rounded pixel positions identical."""
    rng = default_rng(77)
    for h, w, gs, amp in [(300, 420, 20, 6.0), (97, 64, 15, 3.0), (512, 512, 50, 12.0)]:
        sv, dv, ds = synthetic_grid(h, w, gs, amp, seed=int(rng.integers(1 << 30)))
        n = 500
        xs = rng.integers(0, (sv.shape[1] - 1) * gs, n)
        ys = rng.integers(0, (sv.shape[0] - 1) * gs, n)
        pts = np.stack([xs, ys], axis=1).astype(np.int32)
        smooth = pts + rng.uniform(-0.5, 0.5, pts.shape)
        smooth[:50] = pts[:50]                     # exact lattice-aligned points too
        got = N.project_points(sv, dv, gs, pts, smooth)
        want = O.grid_project_points(sv, dv, gs, pts, smooth)
        assert got.shape == want.shape == (n, 2)
        ulp = np.spacing(np.abs(want))
        assert (np.abs(got - want) <= 4 * ulp).all()
        off_tie = np.abs(want - np.floor(want) - 0.5) > 1e-9
        assert (np.rint(got)[off_tie] == np.rint(want)[off_tie]).all()
    # a point past the last cell is an IndexError, like the reference's cell table
    with pytest.raises(IndexError):
        N.project_points(sv, dv, gs, [[(sv.shape[1] - 1) * gs, 3]], [[float((sv.shape[1] - 1) * gs), 3.0]])
    assert N.project_points(sv, dv, gs, np.zeros((0, 2), np.int32), np.zeros((0, 2))).shape == (0, 2)


def test_grid_distortion_points_and_polygons_batch(N):
    """Distortion.distort(points=, polygons=) of a grid-based op runs the batch hooks and agrees with func_point."""
    import vkit_amd.mechanism.distortion as D
    from vkit_amd.element import Point, PointList, Polygon
    from vkit_amd.mechanism.distortion.geometric.grid_rendering.interface import FuncImageGridBased
    cfg = D.CameraCubicCurveConfig(curve_alpha=40.0, curve_beta=-25.0, curve_direction=30.0, curve_scale=1.0,
                                   camera_model_config=D.CameraModelConfig(rotation_unit_vec=[1.0, 0.5, 0.1], rotation_theta=20.0),
                                   grid_size=16)
    shape = (200, 260)
    pts = PointList(Point.create(y=y, x=x) for y, x in [(0, 0), (10.25, 33.5), (199, 259), (100, 100), (57.75, 201.125)])
    polys = [Polygon.create(points=[Point.create(y=5, x=5), Point.create(y=5, x=90), Point.create(y=60, x=90)]),
             Polygon.create(points=[Point.create(y=150, x=20), Point.create(y=190, x=200), Point.create(y=120, x=250),
                                    Point.create(y=100, x=30)])]
    res = D.camera_cubic_curve.distort(cfg, shape, points=pts, polygons=polys, get_state=True)
    st = res.state
    for p, q in zip(pts, res.points):
        ref = FuncImageGridBased.func_point(cfg, st, shape, p, None)
        assert (q.y, q.x) == (ref.y, ref.x)
        assert abs(q.smooth_y - ref.smooth_y) < 1e-9 and abs(q.smooth_x - ref.smooth_x) < 1e-9
    assert [len(p.points) for p in res.polygons] == [3, 4]
    for poly, out in zip(polys, res.polygons):
        for p, q in zip(poly.points, out.points):
            ref = FuncImageGridBased.func_point(cfg, st, shape, p, None)
            assert (q.y, q.x) == (ref.y, ref.x)


def test_fused_chain_hue_shift_whole_colour_cube(N):
    """Every RGB triple through the fused chain kernel (identity lattice, hue shift only): its HSV round trip carries
    the hue modulo 256 and takes floor / fraction of the sector in integers -- same bytes as the staged cvtColor pair."""
    from types import SimpleNamespace
    from vkit_amd.batch import ChainBatch
    v = np.arange(256, dtype=np.uint8)
    cube = np.ascontiguousarray(np.stack(np.meshgrid(v, v, v, indexing='ij'), -1).reshape(4096, 4096, 3))
    coords = list(range(0, 4096, 64)) + [4095]
    lattice = np.array([[(x, y) for x in coords] for y in coords], np.int32)
    state = SimpleNamespace(result_shape=(4096, 4096), src_image_grid=SimpleNamespace(vertices=lattice),
                            dst_image_grid=SimpleNamespace(vertices=lattice))
    for delta in (37, -100, 255):
        batch = ChainBatch()
        batch.add(cube, state, blur_sigma=None, hue_delta=delta, noise=None)
        batch.run()
        got = batch.result(0)
        batch.close()
        assert (got == O.color_shift_rgb(cube, delta)).all(), delta


def test_remap_multi_shared_map(N):
    """vkx_remap_multi: Image + Mask + ScoreMap + a 4-channel plane through one explicit dense map, incl. out-of-range
    and non-finite map entries -- equal to cv.remap element by element."""
    rng = default_rng(17)
    h, w, dh, dw = 93, 121, 80, 150
    mats = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8), (rng.random((h, w)) < 0.5).astype(np.uint8),
            rng.random((h, w), dtype=np.float32), rng.integers(0, 256, (h, w, 4), dtype=np.uint8)]
    mx = rng.uniform(-5, w + 5, (dh, dw)).astype(np.float32)
    my = rng.uniform(-5, h + 5, (dh, dw)).astype(np.float32)
    mx[3, 4] = np.nan
    my[5, 6] = np.inf
    for got, m in zip(N.remap_multi(mats, mx, my), mats):
        want = O.remap(m, mx, my)
        assert got.dtype == want.dtype and (got == want).all()
    # a lattice's own map reproduces vkx_grid_remap
    sv, dv, ds = synthetic_grid(h, w, 15, 4.0, seed=3)
    gx, gy = N.grid_to_map(sv, dv, ds)
    for a, b in zip(N.remap_multi(mats, gx, gy), N.grid_remap(mats, sv, dv, ds)):
        assert (a == b).all()
