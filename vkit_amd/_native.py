"""ctypes binding of libvkx.so (include/vkx.h) and numpy-facing call helpers.

The HIP library is the only implementation of the pixel work: importing this module fails loudly when
``vkit_amd/libvkx.so`` has not been built (``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C vkit_amd/csrc``), and every call raises :class:`VkxError` when no MI355X is visible.  There is
no CPU fallback.
"""
import ctypes
import math
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# VKX_LIB: an alternative build of the same library (A/B experiments); the default is the in-tree libvkx.so
LIB_PATH = os.environ.get('VKX_LIB') or os.path.join(_HERE, 'libvkx.so')

c_int = ctypes.c_int
c_void_p = ctypes.c_void_p
c_ssize = ctypes.c_ssize_t
c_size = ctypes.c_size_t
c_double = ctypes.c_double
c_uint = ctypes.c_uint


class VkxError(RuntimeError):
    pass


# error codes of include/vkx.h
ERR_INVALID, ERR_HIP, ERR_NOMEM, ERR_UNSUPPORTED, ERR_OUT_OF_LATTICE, ERR_DIVIDE = -1, -2, -3, -4, -5, -6


class VkxElem(ctypes.Structure):
    _fields_ = [
        ('src', c_void_p),
        ('dst', c_void_p),
        ('src_stride', c_ssize),
        ('dst_stride', c_ssize),
        ('cn', ctypes.c_int32),
        ('is_f32', ctypes.c_int32),
    ]


class VkxChainItem(ctypes.Structure):
    _fields_ = [
        ('src', c_void_p),
        ('dst', c_void_p),
        ('src_stride', c_ssize),
        ('dst_stride', c_ssize),
        ('sh', ctypes.c_int32),
        ('sw', ctypes.c_int32),
        ('dh', ctypes.c_int32),
        ('dw', ctypes.c_int32),
        ('src_vertices', c_void_p),
        ('dst_vertices', c_void_p),
        ('rows', ctypes.c_int32),
        ('cols', ctypes.c_int32),
        ('noise', c_void_p),
        ('noise_stride_el', c_ssize),
        ('blur_sigma', c_double),
        ('blur_ksize', ctypes.c_int32),
        ('hue_delta', ctypes.c_int32),
        ('hue_enabled', ctypes.c_int32),
        ('streak_enabled', ctypes.c_int32),
        ('streak_thickness', ctypes.c_int32),
        ('streak_gap', ctypes.c_int32),
        ('streak_dash_thickness', ctypes.c_int32),
        ('streak_dash_gap', ctypes.c_int32),
        ('streak_enable_vert', ctypes.c_int32),
        ('streak_enable_hori', ctypes.c_int32),
        ('streak_color', ctypes.c_uint8 * 4),
        ('noise_tiled', ctypes.c_int32),
        ('streak_alpha', c_double),
    ]


class VkxLutPlane(ctypes.Structure):
    _fields_ = [('src', c_void_p), ('dst', c_void_p), ('n_bytes', c_size), ('lut_host', c_void_p)]


class VkxPaintSet(ctypes.Structure):
    _fields_ = [
        ('pts_host', c_void_p),
        ('poly_offsets_host', c_void_p),
        ('n_polys', ctypes.c_int32),
        ('values_host', c_void_p),
        ('mask', c_void_p),
        ('mask_stride', c_ssize),
        ('score', c_void_p),
        ('score_stride_el', c_ssize),
    ]


class VkxLayer(ctypes.Structure):
    _fields_ = [
        ('up', ctypes.c_int32),
        ('left', ctypes.c_int32),
        ('height', ctypes.c_int32),
        ('width', ctypes.c_int32),
        ('mask', c_void_p),
        ('mask_stride', c_ssize),
        ('alpha', c_void_p),
        ('alpha_stride_el', c_ssize),
        ('alpha_scalar', c_double),
        ('value', c_void_p),
        ('value_stride', c_ssize),
        ('value_const', ctypes.c_uint8 * 4),
        ('mode', ctypes.c_int32),
    ]


class VkxLayerF32(ctypes.Structure):
    _fields_ = [
        ('up', ctypes.c_int32),
        ('left', ctypes.c_int32),
        ('height', ctypes.c_int32),
        ('width', ctypes.c_int32),
        ('mask', c_void_p),
        ('mask_stride', c_ssize),
        ('alpha', c_void_p),
        ('alpha_stride_el', c_ssize),
        ('alpha_scalar', c_double),
        ('value', c_void_p),
        ('value_stride_el', c_ssize),
        ('value_const', ctypes.c_float),
        ('mode', ctypes.c_int32),
    ]


class VkxNpJob(ctypes.Structure):
    _fields_ = [
        ('state', ctypes.c_uint64 * 2),
        ('inc', ctypes.c_uint64 * 2),
        ('n', ctypes.c_int64),
        ('kind', ctypes.c_int32),
        ('cn', ctypes.c_int32),
        ('scale', c_double),
        ('cdf', c_double * 3),
        ('src', c_void_p),
        ('dst', c_void_p),
    ]


class VkxCameraConfig(ctypes.Structure):
    _fields_ = [
        ('kind', ctypes.c_int32), ('height', ctypes.c_int32), ('width', ctypes.c_int32), ('grid_size', ctypes.c_int32),
        ('rotation_unit_vec', c_double * 3),
        ('rotation_theta', c_double),
        ('focal_length', c_double), ('camera_distance', c_double),
        ('principal_point', c_double * 3),
        ('principal_point_len', ctypes.c_int32), ('reserved', ctypes.c_int32),
        ('curve_alpha', c_double), ('curve_beta', c_double), ('curve_direction', c_double), ('curve_scale', c_double),
    ]


class VkxCameraModel(ctypes.Structure):
    _fields_ = [
        ('R', c_double * 9), ('t', c_double * 3),
        ('fx', c_double), ('fy', c_double), ('cx', c_double), ('cy', c_double),
        ('a0', ctypes.c_float), ('a1', ctypes.c_float), ('along_min', ctypes.c_float), ('along_range', ctypes.c_float),
        ('poly', c_double * 4), ('curve_scale', c_double),
        ('rows', ctypes.c_int32), ('cols', ctypes.c_int32), ('points_f32', ctypes.c_int32), ('reserved', ctypes.c_int32),
    ]


class VkxMlsConfig(ctypes.Structure):
    _fields_ = [
        ('height', ctypes.c_int32), ('width', ctypes.c_int32), ('grid_size', ctypes.c_int32), ('n_handles', ctypes.c_int32),
        ('src_handles', c_void_p), ('dst_handles', c_void_p),
        ('src_handles_smooth', c_void_p), ('dst_handles_smooth', c_void_p),
    ]


class VkxGridState(ctypes.Structure):
    _fields_ = [
        ('rows', ctypes.c_int32), ('cols', ctypes.c_int32), ('dh', ctypes.c_int32), ('dw', ctypes.c_int32),
        ('shift_y', ctypes.c_int32), ('shift_x', ctypes.c_int32), ('flags', ctypes.c_uint32), ('reserved', ctypes.c_uint32),
    ]


CAMERA_PLANE_ONLY, CAMERA_CUBIC_CURVE = 0, 1
GRID_STATE_NAN, GRID_STATE_INF, GRID_STATE_RANGE, GRID_STATE_DIVIDE = 1, 2, 4, 8


class VkxNoisePlane(ctypes.Structure):
    _fields_ = [
        ('dst', ctypes.c_void_p),
        ('stride_el', ctypes.c_ssize_t),
        ('h', ctypes.c_int), ('w', ctypes.c_int), ('cn', ctypes.c_int), ('reserved', ctypes.c_int),
        ('seed', ctypes.c_uint64),
    ]


class VkxNpResult(ctypes.Structure):
    _fields_ = [
        ('draws', ctypes.c_uint64),
        ('samples', ctypes.c_uint64),
        ('flags', ctypes.c_uint32),
        ('reserved', ctypes.c_uint32),
    ]


NP_NORMAL_I16, NP_NORMAL_ADD_U8, NP_SPECKLE_U8, NP_CHOICE3_U8, NP_IMPULSE_U8, NP_NORMAL_TILES = 0, 1, 2, 3, 4, 5
NP_AMBIGUOUS, NP_SHORT = 1, 2

FILL_PLAIN, FILL_KEEP_MAX, FILL_KEEP_MIN = 0, 1, 2
STREAM_COMPUTE, STREAM_COPY_IN, STREAM_COPY_OUT = 0, 1, 2


# name -> argtypes; every function returns int unless noted.
_PLANE_U8 = [c_void_p, c_int, c_int, c_int, c_ssize]  # ptr, h, w, cn, stride
_SIGNATURES = {
    'vkx_device_count': [ctypes.POINTER(c_int)],
    'vkx_device_pci_bus_id': [c_int, ctypes.c_char_p, c_int],
    'vkx_ctx_create': [c_int, ctypes.POINTER(c_void_p)],
    'vkx_ctx_destroy': [c_void_p],
    'vkx_ctx_sync': [c_void_p],
    'vkx_ctx_set_stream': [c_void_p, c_void_p],
    'vkx_malloc': [c_void_p, c_size, ctypes.POINTER(c_void_p)],
    'vkx_free': [c_void_p, c_void_p],
    'vkx_upload': [c_void_p, c_void_p, c_void_p, c_size],
    'vkx_download': [c_void_p, c_void_p, c_void_p, c_size],
    'vkx_memset': [c_void_p, c_void_p, c_int, c_size],
    'vkx_chain_rgb_batch_dev': [c_void_p, ctypes.POINTER(VkxChainItem), c_int],
    'vkx_chain_rgb_batch_np_dev': [c_void_p, ctypes.POINTER(VkxChainItem), c_int, ctypes.POINTER(VkxNpJob), c_int,
                                   ctypes.POINTER(VkxNpResult)],
    'vkx_chain_lattices_ready': [c_void_p],
    'vkx_camera_model_host': [ctypes.POINTER(VkxCameraConfig), ctypes.POINTER(VkxCameraModel)],
    'vkx_camera_states_dev': [c_void_p, ctypes.POINTER(VkxCameraConfig), c_int, c_void_p, c_void_p, c_void_p, c_int],
    'vkx_mls_states_dev': [c_void_p, ctypes.POINTER(VkxMlsConfig), c_int, c_void_p, c_void_p, c_void_p, c_int],
    'vkx_noise_normal_table': [c_double, c_void_p],
    'vkx_noise_normal_i16_dev': [c_void_p, c_void_p, c_ssize, c_int, c_int, c_int, c_double, ctypes.c_uint64],
    'vkx_noise_normal_i16': [c_void_p, c_void_p, c_ssize, c_int, c_int, c_int, c_double, ctypes.c_uint64],
    'vkx_noise_normal_i16_batch_dev': [c_void_p, ctypes.POINTER(VkxNoisePlane), c_int, c_double],
    'vkx_np_draw_batch_dev': [c_void_p, ctypes.POINTER(VkxNpJob), c_int, ctypes.POINTER(VkxNpResult)],
    'vkx_np_tiles_layout': [ctypes.c_int64] + [ctypes.POINTER(ctypes.c_int64)] * 5,
    'vkx_np_tiles_expand_dev': [c_void_p, c_void_p, ctypes.c_int64, c_void_p],
    'vkx_np_draw': [c_void_p, ctypes.POINTER(VkxNpJob), ctypes.POINTER(VkxNpResult)],
    'vkx_np_poisson_loggam_table': [c_void_p, c_int],
    'vkx_glass_init_dev': [c_void_p, c_void_p, c_void_p, c_int, c_int],
    'vkx_glass_round_dev': [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    'vkx_fog_field_f32_dev': [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p],
    'vkx_fog_stretch_f32_dev': [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, ctypes.c_double, ctypes.c_double, c_void_p],
    'vkx_host_alloc': [c_void_p, c_size, ctypes.POINTER(c_void_p)],
    'vkx_host_free': [c_void_p, c_void_p],
    'vkx_upload_async': [c_void_p, c_void_p, c_void_p, c_size],
    'vkx_download_async': [c_void_p, c_void_p, c_void_p, c_size],
    'vkx_memcpy_async': [c_void_p, c_int, c_void_p, c_void_p, c_size, c_int],
    'vkx_ctx_order': [c_void_p, c_int, c_int],
    'vkx_ctx_sync_stream': [c_void_p, c_int],
    'vkx_event_record': [c_void_p, c_int, ctypes.POINTER(c_void_p)],
    'vkx_event_wait': [c_void_p, c_void_p],
    'vkx_ctx_set_timing': [c_void_p, c_int],
    'vkx_ctx_collect_timings': [c_void_p, ctypes.POINTER(c_int)],
    'vkx_ctx_get_timing': [c_void_p, c_int, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(c_double),
                           ctypes.POINTER(ctypes.c_longlong)],
    'vkx_ctx_reset_timings': [c_void_p],
}
for _sfx in ('', '_dev'):
    _SIGNATURES['vkx_saturate_i64_u8' + _sfx] = [c_void_p, c_void_p, c_size, c_void_p]
    _SIGNATURES['vkx_remap_u8' + _sfx] = [c_void_p] + _PLANE_U8 + [c_void_p, c_void_p, c_ssize, c_void_p, c_int, c_int, c_ssize]
    _SIGNATURES['vkx_remap_f32' + _sfx] = [c_void_p, c_void_p, c_int, c_int, c_ssize, c_void_p, c_void_p, c_ssize,
                                           c_void_p, c_int, c_int, c_ssize]
    for _w in ('vkx_warp_affine', 'vkx_warp_perspective'):
        _SIGNATURES[_w + '_u8' + _sfx] = [c_void_p] + _PLANE_U8 + [c_void_p, c_void_p, c_int, c_int, c_ssize]
        _SIGNATURES[_w + '_f32' + _sfx] = [c_void_p, c_void_p, c_int, c_int, c_ssize, c_void_p, c_void_p, c_int, c_int,
                                           c_ssize]
    _SIGNATURES['vkx_grid_to_map' + _sfx] = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                             c_void_p, c_ssize, c_void_p]
    _SIGNATURES['vkx_grid_remap' + _sfx] = [c_void_p, ctypes.POINTER(VkxElem), c_int, c_int, c_int, c_void_p, c_void_p,
                                            c_int, c_int, c_int, c_int]
    _SIGNATURES['vkx_remap_multi' + _sfx] = [c_void_p, ctypes.POINTER(VkxElem), c_int, c_int, c_int, c_void_p, c_void_p, c_ssize,
                                             c_int, c_int]
    _SIGNATURES['vkx_gaussian_blur_u8' + _sfx] = [c_void_p] + _PLANE_U8 + [c_int, c_double, c_void_p, c_ssize]
    _SIGNATURES['vkx_grid_project_points'] = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int,
                                              c_void_p]
    _SIGNATURES['vkx_mls_project'] = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p]
    _SIGNATURES['vkx_mls_project_dev'] = _SIGNATURES['vkx_mls_project'] + [c_void_p]
    _SIGNATURES['vkx_color_shift_rgb' + _sfx] = [c_void_p, c_void_p, c_int, c_int, c_ssize, c_int, c_void_p, c_ssize]
    _SIGNATURES['vkx_cvt_rgb_hsv_u8' + _sfx] = [c_void_p, c_void_p, c_int, c_int, c_ssize, c_int, c_void_p, c_ssize]
    _SIGNATURES['vkx_mean_shift_u8' + _sfx] = [c_void_p] + _PLANE_U8 + [c_int, c_int, c_int, c_int, c_uint, c_void_p,
                                                                        c_ssize]
    _SIGNATURES['vkx_add_noise_i16' + _sfx] = [c_void_p] + _PLANE_U8 + [c_void_p, c_ssize, c_void_p, c_ssize]
    _SIGNATURES['vkx_cvt_color_u8' + _sfx] = [c_void_p, c_void_p, c_int, c_int, c_ssize, c_int, c_void_p, c_ssize]
    _SIGNATURES['vkx_blend_u8' + _sfx] = [c_void_p, c_void_p, c_ssize, c_void_p, c_ssize, c_int, c_int, c_int, c_double, c_double, c_uint,
                                         c_void_p, c_ssize]
    _SIGNATURES['vkx_fog_f32_u8' + _sfx] = [c_void_p] + _PLANE_U8 + [c_void_p, c_ssize, c_void_p, c_void_p, c_ssize]
    _SIGNATURES['vkx_brightness_shift_rgb' + _sfx] = [c_void_p, c_void_p, c_int, c_int, c_ssize, c_int, c_void_p, c_ssize]
    _SIGNATURES['vkx_color_balance_rgb' + _sfx] = [c_void_p, c_void_p, c_int, c_int, c_ssize, c_double, c_void_p, c_ssize]
    _SIGNATURES['vkx_histogram_u8' + _sfx] = [c_void_p] + _PLANE_U8 + [c_void_p]
    _SIGNATURES['vkx_sum_f32_u8' + _sfx] = [c_void_p] + _PLANE_U8 + [c_void_p, c_int, c_int, c_void_p]
    _SIGNATURES['vkx_np_poisson_u8' + _sfx] = [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_longlong, c_void_p, c_void_p, c_void_p]
    _SIGNATURES['vkx_apply_lut_u8' + _sfx] = [c_void_p] + _PLANE_U8 + [c_void_p, c_uint, c_void_p, c_ssize]
    _SIGNATURES['vkx_gather_u8' + _sfx] = [c_void_p] + _PLANE_U8 + [c_void_p, c_void_p, c_ssize, c_void_p, c_int, c_int, c_ssize]
    _SIGNATURES['vkx_pointwise_u8' + _sfx] = [c_void_p] + _PLANE_U8 + [c_int, c_int, c_int, c_uint, c_void_p, c_ssize]
    _SIGNATURES['vkx_impulse_noise_u8' + _sfx] = [c_void_p] + _PLANE_U8 + [c_void_p, c_ssize, c_void_p, c_ssize]
    _SIGNATURES['vkx_speckle_noise_u8' + _sfx] = [c_void_p] + _PLANE_U8 + [c_void_p, c_ssize, c_void_p, c_ssize]
    _SIGNATURES['vkx_line_streak_u8' + _sfx] = [c_void_p] + _PLANE_U8 + [c_int, c_int, c_int, c_int, c_void_p, c_double,
                                                                         c_int, c_int]
    _SIGNATURES['vkx_fill_u8' + _sfx] = [c_void_p] + _PLANE_U8 + [ctypes.POINTER(VkxLayer), c_int]
    _SIGNATURES['vkx_fill_f32' + _sfx] = [c_void_p, c_void_p, c_int, c_int, c_ssize, ctypes.POINTER(VkxLayerF32), c_int]
    _SIGNATURES['vkx_resize_u8' + _sfx] = [c_void_p] + _PLANE_U8 + [c_void_p, c_int, c_int, c_ssize, c_int]
    _SIGNATURES['vkx_zoom_in_blur_u8' + _sfx] = [c_void_p] + _PLANE_U8 + [c_void_p, c_int, c_double, c_void_p, c_ssize]
    _SIGNATURES['vkx_resize_cubic_u8' + _sfx] = [c_void_p] + _PLANE_U8 + [c_void_p, c_int, c_int, c_ssize]
    _SIGNATURES['vkx_resize_cubic_f32' + _sfx] = [c_void_p, c_void_p, c_int, c_int, c_ssize, c_void_p, c_int, c_int, c_ssize]
    _SIGNATURES['vkx_filter2d_u8' + _sfx] = [c_void_p] + _PLANE_U8 + [c_void_p, c_int, c_int, c_void_p, c_ssize]
    _SIGNATURES['vkx_resize_f32' + _sfx] = [c_void_p, c_void_p, c_int, c_int, c_ssize, c_void_p, c_int, c_int, c_ssize, c_int]
    _SIGNATURES['vkx_paint_polys' + _sfx] = [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_ssize, c_void_p,
                                             c_ssize, c_int, c_int]
    _SIGNATURES['vkx_fill_poly_mask_u8' + _sfx] = [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_ssize]
    _SIGNATURES['vkx_ellipse_mask_u8' + _sfx] = [c_void_p, c_void_p, c_ssize, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int]
    _SIGNATURES['vkx_ellipse_streak_u8' + _sfx] = [c_void_p] + _PLANE_U8 + [c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_double]

_SIGNATURES['vkx_paint_polys_fresh_dev'] = _SIGNATURES['vkx_paint_polys_dev']
_SIGNATURES['vkx_apply_lut_u8_planes_dev'] = [c_void_p, ctypes.POINTER(VkxLutPlane), c_int]
_SIGNATURES['vkx_paint_poly_sets_fresh_dev'] = [c_void_p, ctypes.POINTER(VkxPaintSet), c_int, c_int, c_int]
_SIGNATURES['vkx_fill_u8_dev_host_layers'] = [c_void_p] + _PLANE_U8 + [ctypes.POINTER(VkxLayer), c_int]
_SIGNATURES['vkx_fill_u8_batch_dev'] = [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_ssize, ctypes.POINTER(VkxLayer), c_void_p]
EXPORTED_SYMBOLS = sorted(list(_SIGNATURES) + ['vkx_version', 'vkx_last_error', 'vkx_ctx_stream'])

_lib = None
_lib_lock = threading.Lock()
HIP_RUNTIME = 'system'


def _preload_shared_hip_runtime():
    """One HIP / HSA runtime per process.

    PyTorch-ROCm wheels bundle their own ``libamdhip64.so`` (SONAME ``libamdhip64.so.7``, same as /opt/rocm's).
    Two HSA runtimes in one process cannot both open the device: whichever initialises second reports "no
    ROCm-capable device".  So when torch is installed, its copy is loaded first (RTLD_GLOBAL); ``libvkx.so``'s
    ``NEEDED libamdhip64.so.7`` then binds to it by SONAME, and a later ``import torch`` reuses the same object.
    ``VKX_HIP_RUNTIME=system`` opts out (processes that never import torch).
    """
    global HIP_RUNTIME
    if os.environ.get('VKX_HIP_RUNTIME', '') == 'system':
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec('torch')
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], 'lib', 'libamdhip64.so')
    if os.path.exists(path):
        ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
        HIP_RUNTIME = path


def lib():
    """The loaded shared library (argtypes installed).  Raises ImportError if it was never built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lib_lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise ImportError(
                    f'{LIB_PATH} is missing: build the HIP extension first '
                    '(python -c "import __graft_entry__ as g; g.build()").  vkit_amd has no CPU fallback.')
            _preload_shared_hip_runtime()
            handle = ctypes.CDLL(LIB_PATH)
            for name, args in _SIGNATURES.items():
                fn = getattr(handle, name)
                fn.argtypes = args
                fn.restype = c_int
            handle.vkx_version.restype = c_int
            handle.vkx_last_error.restype = ctypes.c_char_p
            handle.vkx_ctx_stream.restype = c_void_p
            handle.vkx_ctx_stream.argtypes = [c_void_p]
            _lib = handle
    return _lib


def last_error():
    return lib().vkx_last_error().decode('utf-8', 'replace')


def check(rc):
    if rc != 0:
        raise VkxError(f'libvkx error {rc}: {last_error()}')


def device_count():
    n = c_int(0)
    check(lib().vkx_device_count(ctypes.byref(n)))
    return n.value


class Context:
    """One HIP stream + device scratch on one GPU.  Not thread safe; use one per thread."""

    def __init__(self, device=0):
        self._h = c_void_p()
        self.device = int(device)
        check(lib().vkx_ctx_create(self.device, ctypes.byref(self._h)))

    @property
    def handle(self):
        if not self._h:
            raise VkxError('context already destroyed')
        return self._h

    def sync(self):
        check(lib().vkx_ctx_sync(self.handle))

    def set_stream(self, stream_ptr):
        check(lib().vkx_ctx_set_stream(self.handle, c_void_p(stream_ptr) if stream_ptr else None))

    def set_timing(self, enabled):
        check(lib().vkx_ctx_set_timing(self.handle, 2 if enabled == 2 else int(bool(enabled))))

    def reset_timings(self):
        check(lib().vkx_ctx_reset_timings(self.handle))

    def timings(self):
        """{kernel name: (total milliseconds, launches)} measured with HIP events on the launch stream."""
        n = c_int(0)
        check(lib().vkx_ctx_collect_timings(self.handle, ctypes.byref(n)))
        out = {}
        for i in range(n.value):
            name, ms, cnt = ctypes.c_char_p(), c_double(), ctypes.c_longlong()
            check(lib().vkx_ctx_get_timing(self.handle, i, ctypes.byref(name), ctypes.byref(ms), ctypes.byref(cnt)))
            out[name.value.decode()] = (ms.value, cnt.value)
        return out

    def malloc(self, nbytes):
        ptr = c_void_p()
        check(lib().vkx_malloc(self.handle, int(nbytes), ctypes.byref(ptr)))
        return ptr.value

    def free(self, ptr):
        check(lib().vkx_free(self.handle, c_void_p(ptr)))

    def upload(self, dptr, array):
        array = np.ascontiguousarray(array)
        check(lib().vkx_upload(self.handle, c_void_p(dptr), _ptr(array), array.nbytes))

    def download(self, dptr, array):
        assert array.flags.c_contiguous and array.flags.writeable
        check(lib().vkx_download(self.handle, _ptr(array), c_void_p(dptr), array.nbytes))
        return array

    # ---- page-locked host memory, copy streams, ordering (include/vkx.h) -------------------------------------------
    def host_alloc(self, nbytes):
        ptr = c_void_p()
        check(lib().vkx_host_alloc(self.handle, int(nbytes), ctypes.byref(ptr)))
        return ptr.value

    def host_free(self, ptr):
        check(lib().vkx_host_free(self.handle, c_void_p(ptr)))

    def pinned_empty(self, shape, dtype=np.uint8):
        """A numpy array in page-locked host memory owned by this context's pool (recycled when the array dies)."""
        return self.pinned_pool.empty(shape, dtype)

    @property
    def pinned_pool(self):
        pool = getattr(self, '_pinned_pool', None)
        if pool is None:
            pool = self._pinned_pool = PinnedPool(self)
        return pool

    def upload_async(self, dptr, array):
        assert array.flags.c_contiguous
        check(lib().vkx_upload_async(self.handle, c_void_p(dptr), _ptr(array), array.nbytes))

    def download_async(self, dptr, array):
        assert array.flags.c_contiguous and array.flags.writeable
        check(lib().vkx_download_async(self.handle, _ptr(array), c_void_p(dptr), array.nbytes))

    def copy_in(self, dptr, array, stream=STREAM_COMPUTE):
        """host array -> device, queued on ``stream`` (default: in order with the kernels)"""
        assert array.flags.c_contiguous
        check(lib().vkx_memcpy_async(self.handle, int(stream), c_void_p(dptr), _ptr(array), array.nbytes, 1))

    def copy_out(self, dptr, array, stream=STREAM_COMPUTE):
        assert array.flags.c_contiguous and array.flags.writeable
        check(lib().vkx_memcpy_async(self.handle, int(stream), _ptr(array), c_void_p(dptr), array.nbytes, 0))

    def order(self, later_stream, earlier_stream):
        check(lib().vkx_ctx_order(self.handle, int(later_stream), int(earlier_stream)))

    def sync_stream(self, stream):
        check(lib().vkx_ctx_sync_stream(self.handle, int(stream)))

    def event_record(self, stream):
        ev = c_void_p()
        check(lib().vkx_event_record(self.handle, int(stream), ctypes.byref(ev)))
        return ev.value

    def event_wait(self, event):
        check(lib().vkx_event_wait(self.handle, c_void_p(event)))

    # ---- device-resident arrays (DevArray): pooled device memory behind the elements' lazy ``.mat`` ------------------
    @property
    def device_pool(self):
        pool = getattr(self, '_device_pool', None)
        if pool is None:
            pool = self._device_pool = DevicePool(self)
        return pool

    def dev_empty(self, shape, dtype=np.uint8):
        """An uninitialised C-contiguous array in device memory."""
        return self.device_pool.empty(shape, dtype)

    def to_device(self, array):
        """``array`` (numpy or DevArray) as a DevArray of this context; a host array is uploaded (synchronous copy)."""
        if isinstance(array, DevArray):
            return array
        array = np.ascontiguousarray(array)
        out = self.device_pool.empty(array.shape, array.dtype)
        if array.nbytes:
            self.upload(out.ptr, array)
        return out

    def close(self):
        if self._h:
            pool = getattr(self, '_pinned_pool', None)
            if pool is not None:
                pool.close()
            dpool = getattr(self, '_device_pool', None)
            if dpool is not None:
                dpool.close()
            lib().vkx_ctx_destroy(self._h)
            self._h = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DevArray:
    """A C-contiguous array in device memory: ``shape`` / ``dtype`` / ``ndim`` / ``nbytes`` like numpy, ``ptr`` the device
    address, ``host()`` the (cached, page-locked) numpy copy.  Every kernel of a context runs on its one stream, so arrays
    chain from call to call without synchronisation; the block returns to the context's pool when the object dies --
    whatever is still queued on the stream has been queued before any later user of the block."""

    __slots__ = ('ctx', 'ptr', 'shape', 'dtype', '_host', '_size', '__weakref__')

    def __init__(self, ctx, ptr, shape, dtype, size):
        self.ctx, self.ptr, self.shape, self.dtype, self._size = ctx, ptr, tuple(int(v) for v in shape), np.dtype(dtype), size
        self._host = None

    @property
    def ndim(self):
        return len(self.shape)

    @property
    def size(self):
        return math.prod(self.shape)

    @property
    def nbytes(self):
        return self.size * self.dtype.itemsize

    def host(self):
        """The numpy copy (downloaded once; read-only views of it are what the elements' ``.mat`` hand out)."""
        if self._host is None:
            out = self.ctx.pinned_empty(self.shape, self.dtype)
            if self.nbytes:
                self.ctx.download(self.ptr, out)
            self._host = out
        return self._host

    def invalidate_host(self):
        """Call after a kernel has written the array in place."""
        self._host = None

    def __repr__(self):
        return f'DevArray(shape={self.shape}, dtype={self.dtype}, device={self.ctx.device})'


class DevicePool:
    """Size-classed free lists of device blocks (hipMalloc / hipFree cost ~100 us and hipFree synchronises the device)."""

    GRANULE = 1 << 16

    def __init__(self, ctx, cap_bytes=4 << 30):
        import weakref
        self._weakref = weakref
        self._ctx = weakref.ref(ctx)
        self._free = {}
        self._idle = 0
        self.cap_bytes = cap_bytes
        self._lock = threading.Lock()

    def empty(self, shape, dtype=np.uint8):
        ctx = self._ctx()
        dtype = np.dtype(dtype)
        nbytes = math.prod(int(v) for v in shape) * dtype.itemsize
        size = max(self.GRANULE, (nbytes + self.GRANULE - 1) // self.GRANULE * self.GRANULE)
        with self._lock:
            blocks = self._free.get(size)
            ptr = blocks.pop() if blocks else None
            if ptr is not None:
                self._idle -= size
        if ptr is None:
            ptr = ctx.malloc(size)
        arr = DevArray(ctx, ptr, shape, dtype, size)
        self._weakref.finalize(arr, self._release, ptr, size)
        return arr

    def _release(self, ptr, size):
        ctx = self._ctx()
        if ctx is None or not ctx._h:
            return                      # the context is gone and took its allocations with the process
        with self._lock:
            if self._idle + size <= self.cap_bytes:
                self._free.setdefault(size, []).append(ptr)
                self._idle += size
                return
        try:
            ctx.free(ptr)
        except Exception:
            pass

    def close(self):
        ctx = self._ctx()
        with self._lock:
            blocks, self._free, self._idle = self._free, {}, 0
        if ctx is not None and ctx._h:
            for ptrs in blocks.values():
                for ptr in ptrs:
                    try:
                        ctx.free(ptr)
                    except Exception:
                        pass


# Resident mode: inside ``with resident():`` the wrappers below leave their results on the device (DevArray) even for
# host inputs; outside, a result is resident exactly when an input was.
_RESIDENT = threading.local()


class resident:
    def __init__(self, enabled=True):
        self.enabled = enabled

    def __enter__(self):
        self.prev = getattr(_RESIDENT, 'on', False)
        _RESIDENT.on = self.enabled
        return self

    def __exit__(self, *exc):
        _RESIDENT.on = self.prev
        return False


def resident_mode():
    return getattr(_RESIDENT, 'on', False)


_CT_NP = {ctypes.c_void_p: np.uint64, ctypes.c_int32: np.int32, ctypes.c_uint32: np.uint32, ctypes.c_int64: np.int64,
          ctypes.c_uint64: np.uint64, ctypes.c_double: np.float64, ctypes.c_float: np.float32, ctypes.c_uint8: np.uint8,
          ctypes.c_ssize_t: np.int64, ctypes.c_size_t: np.uint64, ctypes.c_int: np.int32, ctypes.c_uint: np.uint32}


def struct_view(array):
    """A numpy structured view of a ctypes array of Structures (pointers as uint64): whole-batch field updates without a Python
    loop over the records.  The view aliases the ctypes memory."""
    cls = array._type_
    fields = []
    for name, ct in cls._fields_:
        if hasattr(ct, '_length_') and hasattr(ct, '_type_') and not issubclass(ct, ctypes.Structure):
            fields.append((name, _CT_NP[ct._type_], (ct._length_,)))
        else:
            fields.append((name, _CT_NP[ct]))
    offsets = [getattr(cls, name).offset for name, _ in cls._fields_]
    dt = np.dtype({'names': [f[0] for f in fields], 'formats': [f[1] if len(f) == 2 else (f[1], f[2]) for f in fields],
                   'offsets': offsets, 'itemsize': ctypes.sizeof(cls)})
    return np.frombuffer(array, dtype=dt)


def camera_config(config, shape) -> VkxCameraConfig:
    """The C record of a ``CameraPlaneOnlyConfig`` / ``CameraCubicCurveConfig`` for an image of ``shape`` (vkx_camera_states_dev)."""
    rec = VkxCameraConfig()
    cubic = hasattr(config, 'curve_scale')
    rec.kind = CAMERA_CUBIC_CURVE if cubic else CAMERA_PLANE_ONLY
    rec.height, rec.width, rec.grid_size = int(shape[0]), int(shape[1]), int(config.grid_size)
    cm = config.camera_model_config
    for k in range(3):
        rec.rotation_unit_vec[k] = float(cm.rotation_unit_vec[k])
    rec.rotation_theta = float(cm.rotation_theta)
    rec.focal_length = float(cm.focal_length or 0.0)
    rec.camera_distance = float(cm.camera_distance or 0.0)
    pp = list(cm.principal_point) if cm.principal_point else []
    if len(pp) not in (0, 2, 3):
        raise ValueError('principal_point: two or three coordinates')
    rec.principal_point_len = len(pp)
    for k, v in enumerate(pp):
        rec.principal_point[k] = float(v)
    if cubic:
        rec.curve_alpha, rec.curve_beta = float(config.curve_alpha), float(config.curve_beta)
        rec.curve_direction, rec.curve_scale = float(config.curve_direction), float(config.curve_scale)
    return rec


def mls_config(config, shape):
    """(VkxMlsConfig, keepalive arrays) of a ``SimilarityMlsConfig`` for an image of ``shape`` (vkx_mls_states_dev): the float32
    integer handle positions the reference feeds its arithmetic (``PointTuple.to_smooth_np_array``) and the float64 smooth ones."""
    if getattr(config, 'resize_as_src', False):
        raise ValueError('the batched similarity_mls states take resize_as_src=False (the default)')
    src, dst = config.src_handle_points, config.dst_handle_points
    if len(src) != len(dst) or len(src) < 1:
        raise ValueError('handle lists differ in length')
    p = np.ascontiguousarray(src.to_smooth_np_array(), dtype=np.float32).reshape(-1, 2)
    q = np.ascontiguousarray(dst.to_smooth_np_array(), dtype=np.float32).reshape(-1, 2)
    ps = np.asarray([(h.smooth_x, h.smooth_y) for h in src], dtype=np.float64).reshape(-1, 2)
    qs = np.asarray([(h.smooth_x, h.smooth_y) for h in dst], dtype=np.float64).reshape(-1, 2)
    rec = VkxMlsConfig()
    rec.height, rec.width, rec.grid_size, rec.n_handles = int(shape[0]), int(shape[1]), int(config.grid_size), p.shape[0]
    rec.src_handles, rec.dst_handles = p.ctypes.data, q.ctypes.data
    rec.src_handles_smooth, rec.dst_handles_smooth = ps.ctypes.data, qs.ctypes.data
    return rec, (p, q, ps, qs)


def lattice_shape(height, width, grid_size):
    """(rows, cols) of the source lattice: a vertex every ``grid_size`` pixels plus the last row / column (grid_creator.py:22-41)."""
    def ticks(length):
        n = (length + grid_size - 1) // grid_size
        return n + 1 if (n - 1) * grid_size != length - 1 else n
    return ticks(int(height)), ticks(int(width))


def camera_model_host(config, shape) -> VkxCameraModel:
    """The host scalars of a camera state as the library computes them (parity tests)."""
    rec, out = camera_config(config, shape), VkxCameraModel()
    check(lib().vkx_camera_model_host(ctypes.byref(rec), ctypes.byref(out)))
    return out


def device_pci_bus_id(device: int) -> str:
    buf = ctypes.create_string_buffer(32)
    check(lib().vkx_device_pci_bus_id(int(device), buf, 32))
    return buf.value.decode()


def device_copy(array, ctx=None):
    """A fresh DevArray of ``ctx`` holding a copy of ``array`` (numpy: uploaded; DevArray: copied on the device, asynchronously on
    the compute stream)."""
    ctx = ctx or (array.ctx if isinstance(array, DevArray) else default_ctx())
    if not isinstance(array, DevArray):
        return ctx.to_device(np.ascontiguousarray(array))
    out = ctx.dev_empty(array.shape, array.dtype)
    if array.nbytes:
        check(lib().vkx_memcpy_async(ctx.handle, STREAM_COMPUTE, c_void_p(out.ptr), c_void_p(array.ptr), int(array.nbytes), 2))
    return out


def host_array(a):
    """numpy view of ``a`` (DevArray -> its host copy)."""
    return a.host() if isinstance(a, DevArray) else a


class PinnedPool:
    """Recycled page-locked host buffers behind numpy arrays.

    Results of the host-facing calls land here instead of in fresh ``np.empty`` arrays: a device -> host copy into a
    freshly allocated pageable array runs at the speed of its first-touch page faults (8 GB/s measured, profiles/
    r1e_pcie_inclusive.json), into page-locked memory at the speed of the link (54 GB/s).  A block returns to the pool
    when the last numpy view of it is garbage-collected; the pool keeps at most ``cap_bytes`` of idle blocks and hands
    out plain ``np.empty`` arrays once ``max_bytes`` are out.  ``VKX_PINNED_RESULTS=0`` switches the pool off."""

    GRANULE = 1 << 16

    def __init__(self, ctx, cap_bytes=2 << 30, max_bytes=6 << 30):
        import weakref
        self._weakref = weakref
        self._ctx = weakref.ref(ctx)
        self._free = {}            # rounded size -> [pointers]
        self._idle = 0
        self._out = 0
        self.cap_bytes, self.max_bytes = cap_bytes, max_bytes
        self._lock = threading.Lock()
        self.enabled = os.environ.get('VKX_PINNED_RESULTS', '1') != '0'

    def empty(self, shape, dtype=np.uint8):
        dtype = np.dtype(dtype)
        nbytes = math.prod(int(v) for v in shape) * dtype.itemsize
        ctx = self._ctx()
        if not self.enabled or nbytes < self.GRANULE or ctx is None or not ctx._h:
            return np.empty(shape, dtype)
        size = (nbytes + self.GRANULE - 1) // self.GRANULE * self.GRANULE
        with self._lock:
            if self._out + size > self.max_bytes:
                return np.empty(shape, dtype)
            blocks = self._free.get(size)
            ptr = blocks.pop() if blocks else None
            if ptr is not None:
                self._idle -= size
            self._out += size
        if ptr is None:
            try:
                ptr = ctx.host_alloc(size)
            except VkxError:
                # page-locked memory is a limited resource (memlock ulimit, many worker processes): a pageable array is
                # slower to fill, never wrong
                with self._lock:
                    self._out -= size
                return np.empty(shape, dtype)
        buf = (ctypes.c_char * size).from_address(ptr)
        self._weakref.finalize(buf, self._release, ptr, size)
        return np.frombuffer(buf, dtype=dtype, count=nbytes // dtype.itemsize).reshape(shape)

    def _release(self, ptr, size):
        ctx = self._ctx()
        with self._lock:
            self._out -= size
            if ctx is not None and ctx._h and self._idle + size <= self.cap_bytes:
                self._free.setdefault(size, []).append(ptr)
                self._idle += size
                return
        _free_pinned(ctx, ptr)

    def close(self):
        ctx = self._ctx()
        with self._lock:
            blocks, self._free, self._idle = self._free, {}, 0
        if ctx is not None and ctx._h:
            for ptrs in blocks.values():
                for ptr in ptrs:
                    try:
                        ctx.host_free(ptr)
                    except Exception:
                        pass


_orphan_blocks = []
_orphan_lock = threading.Lock()


def _free_pinned(ctx, ptr):
    """hipHostFree through ``ctx``; a block whose context is gone waits for the next live one (vkx_host_free needs a
    context for the device guard), so nothing stays page-locked for the life of the process."""
    with _orphan_lock:
        todo = _orphan_blocks + [ptr]
        del _orphan_blocks[:]
    if ctx is None or not ctx._h:
        live = [c for c in _default_ctx.values() if c._h]
        ctx = live[0] if live else None
    if ctx is None:
        with _orphan_lock:
            _orphan_blocks.extend(todo)
        return
    for p in todo:
        try:
            ctx.host_free(p)
        except Exception:
            pass


_default_ctx = {}
_ctx_lock = threading.Lock()


def default_device():
    for key in ('VKX_DEVICE', 'LOCAL_RANK'):
        if os.environ.get(key, '') != '':
            n = device_count()
            return int(os.environ[key]) % max(n, 1)
    return 0


def default_ctx():
    """Process-wide context (per thread) on ``VKX_DEVICE`` / ``LOCAL_RANK`` / GPU 0."""
    key = (os.getpid(), threading.get_ident())
    ctx = _default_ctx.get(key)
    if ctx is None:
        with _ctx_lock:
            ctx = _default_ctx.get(key)
            if ctx is None:
                ctx = Context(default_device())
                _default_ctx[key] = ctx
    return ctx


# --------------------------------------------------------------------------------------------------------------
# numpy helpers over the host-pointer entry points
# --------------------------------------------------------------------------------------------------------------
def _ptr(a):
    return c_void_p(a.ctypes.data)


def _u8_plane(img):
    """C-contiguous uint8 view + (h, w, cn, stride)."""
    img = np.ascontiguousarray(img)
    if img.dtype != np.uint8:
        raise TypeError(f'expected uint8, got {img.dtype}')
    if img.ndim == 2:
        h, w = img.shape
        cn = 1
    elif img.ndim == 3:
        h, w, cn = img.shape
    else:
        raise ValueError(f'expected HxW or HxWxC, got shape {img.shape}')
    return img, h, w, cn, w * cn


def _out_like(img, dh, dw, ctx=None):
    shape = (dh, dw) if img.ndim == 2 else (dh, dw, img.shape[2])
    return (ctx or default_ctx()).pinned_empty(shape, img.dtype)


class _Call:
    """One wrapper call: decides host or device path from its inputs, hands out pointers, allocates outputs of the same
    kind.  ``fn(name)`` is the C entry point of that path (``name`` or ``name + '_dev'``: the two share a signature)."""

    def __init__(self, ctx, *arrays):
        # the call runs on the context of its device inputs (their producer's stream: ordered behind it without a
        # synchronisation, and temporaries return to the pool whose stream used them).  Inputs of several contexts -- an
        # element produced on another thread -- are made safe the blunt way: their streams are drained first.
        owners = []
        for a in arrays:
            if isinstance(a, DevArray) and a.ctx is not None and all(a.ctx is not o for o in owners):
                owners.append(a.ctx)
        if ctx is None and owners:
            ctx = owners[0]
        self.ctx = ctx or default_ctx()
        for o in owners:
            if o is not self.ctx:
                o.sync()
        self.dev = resident_mode() or bool(owners)
        self.keep = []

    def fn(self, name):
        return getattr(lib(), name + '_dev' if self.dev else name)

    def src(self, a, dtype=None):
        """Pointer of an input array on this call's side (uploads / downloads as needed)."""
        if self.dev:
            if not isinstance(a, DevArray):
                a = np.ascontiguousarray(a, dtype=dtype) if dtype is not None else np.ascontiguousarray(a)
                a = self.ctx.to_device(a)
            self.keep.append(a)
            return c_void_p(a.ptr)
        a = host_array(a)
        a = np.ascontiguousarray(a, dtype=dtype) if dtype is not None else np.ascontiguousarray(a)
        self.keep.append(a)
        return _ptr(a)

    def out(self, shape, dtype=np.uint8):
        arr = self.ctx.dev_empty(shape, dtype) if self.dev else self.ctx.pinned_empty(shape, dtype)
        return arr, (c_void_p(arr.ptr) if self.dev else _ptr(arr))


def _shape_u8(img):
    """(h, w, cn, row stride in bytes) of a uint8 HxW[xC] numpy array or DevArray."""
    if np.dtype(img.dtype) != np.uint8:
        raise TypeError(f'expected uint8, got {img.dtype}')
    if img.ndim == 2:
        return img.shape[0], img.shape[1], 1, img.shape[1]
    if img.ndim == 3:
        return img.shape[0], img.shape[1], img.shape[2], img.shape[1] * img.shape[2]
    raise ValueError(f'expected HxW or HxWxC, got shape {img.shape}')


def remap(src, map_x, map_y, ctx=None):
    ctx = ctx or default_ctx()
    map_x = np.ascontiguousarray(map_x, dtype=np.float32)
    map_y = np.ascontiguousarray(map_y, dtype=np.float32)
    if map_x.shape != map_y.shape or map_x.ndim != 2:
        raise ValueError('map_x / map_y must be 2-D and of equal shape')
    dh, dw = map_x.shape
    if src.dtype == np.float32:
        src = np.ascontiguousarray(src)
        if src.ndim != 2:
            raise ValueError('float32 sources are single channel')
        sh, sw = src.shape
        dst = ctx.pinned_empty((dh, dw), np.float32)
        check(lib().vkx_remap_f32(ctx.handle, _ptr(src), sh, sw, sw, _ptr(map_x), _ptr(map_y), dw, _ptr(dst), dh, dw, dw))
        return dst
    src, sh, sw, cn, sstride = _u8_plane(src)
    dst = _out_like(src, dh, dw, ctx)
    check(lib().vkx_remap_u8(ctx.handle, _ptr(src), sh, sw, cn, sstride, _ptr(map_x), _ptr(map_y), dw, _ptr(dst), dh, dw,
                             dw * cn))
    return dst


def _warp(kind, src, mat, dsize, ctx):
    dw, dh = int(dsize[0]), int(dsize[1])
    n = 6 if kind == 'affine' else 9
    M = np.ascontiguousarray(np.asarray(mat, dtype=np.float64).reshape(n))
    call = _Call(ctx, src)
    if np.dtype(src.dtype) == np.float32:
        if src.ndim != 2:
            raise ValueError('float32 sources are single channel')
        sh, sw = src.shape
        dst, dptr = call.out((dh, dw), np.float32)
        check(call.fn(f'vkx_warp_{kind}_f32')(call.ctx.handle, call.src(src), sh, sw, sw, _ptr(M), dptr, dh, dw, dw))
        return dst
    sh, sw, cn, sstride = _shape_u8(src)
    dst, dptr = call.out((dh, dw) if src.ndim == 2 else (dh, dw, cn), np.uint8)
    check(call.fn(f'vkx_warp_{kind}_u8')(call.ctx.handle, call.src(src), sh, sw, cn, sstride, _ptr(M), dptr, dh, dw, dw * cn))
    return dst


def warp_affine(src, mat, dsize, ctx=None):
    return _warp('affine', src, mat, dsize, ctx)


def warp_perspective(src, mat, dsize, ctx=None):
    return _warp('perspective', src, mat, dsize, ctx)


def _vertices(v):
    v = np.ascontiguousarray(v, dtype=np.int32)
    if v.ndim != 3 or v.shape[2] != 2:
        raise ValueError('vertices must be int32 [rows, cols, 2] as (x, y)')
    return v


def grid_to_map(src_vertices, dst_vertices, dst_shape, want_owner=False, ctx=None):
    ctx = ctx or default_ctx()
    sv, dv = _vertices(src_vertices), _vertices(dst_vertices)
    if sv.shape != dv.shape:
        raise ValueError('source / destination grids differ in shape')
    rows, cols = sv.shape[:2]
    dh, dw = int(dst_shape[0]), int(dst_shape[1])
    mx = ctx.pinned_empty((dh, dw), np.float32)
    my = ctx.pinned_empty((dh, dw), np.float32)
    owner = ctx.pinned_empty((dh, dw), np.int32) if want_owner else None
    check(lib().vkx_grid_to_map(ctx.handle, _ptr(sv), _ptr(dv), rows, cols, dh, dw, _ptr(mx), _ptr(my), dw,
                                _ptr(owner) if want_owner else None))
    return (mx, my, owner) if want_owner else (mx, my)


def grid_remap(mats, src_vertices, dst_vertices, dst_shape, ctx=None):
    """Gathers every array of ``mats`` (uint8 HxW[xC] / float32 HxW, all of one source shape; numpy or DevArray) through one
    grid.  Results are of the inputs' kind (DevArray when any input is one, or in resident mode)."""
    sv, dv = _vertices(src_vertices), _vertices(dst_vertices)
    if sv.shape != dv.shape:
        raise ValueError('source / destination grids differ in shape')
    rows, cols = sv.shape[:2]
    dh, dw = int(dst_shape[0]), int(dst_shape[1])
    if any(np.dtype(m.dtype) == np.float32 and m.ndim == 3 for m in mats):
        # float32 colour images (the *_GCN modes): cv.remap treats the channels alike -- every channel as a float32 plane of the
        # same call, the results stacked on the host
        flat, counts = [], []
        for m in mats:
            if np.dtype(m.dtype) == np.float32 and m.ndim == 3:
                hm = host_array(m)
                flat += [np.ascontiguousarray(hm[:, :, c]) for c in range(hm.shape[2])]
                counts.append(hm.shape[2])
            else:
                flat.append(m)
                counts.append(0)
        res = iter(grid_remap(flat, src_vertices, dst_vertices, dst_shape, ctx))
        return [np.stack([host_array(next(res)) for _ in range(n)], axis=-1) if n else next(res) for n in counts]
    call = _Call(ctx, *mats)
    sv_p, dv_p = call.src(sv), call.src(dv)         # the device entry point reads the lattices on the device
    outs = []
    for i in range(0, len(mats), 4):
        chunk = list(mats[i:i + 4])
        sh, sw = chunk[0].shape[:2]
        arr = (VkxElem * len(chunk))()
        for j, m in enumerate(chunk):
            if tuple(m.shape[:2]) != (sh, sw):
                raise ValueError('all elements of one call must share the source shape')
            if np.dtype(m.dtype) == np.float32:
                if m.ndim != 2:
                    raise ValueError('float32 elements are single channel')
                out, optr = call.out((dh, dw), np.float32)
                arr[j] = VkxElem(call.src(m).value, optr.value, sw, dw, 1, 1)
            elif np.dtype(m.dtype) == np.uint8:
                cn = 1 if m.ndim == 2 else m.shape[2]
                out, optr = call.out((dh, dw) if m.ndim == 2 else (dh, dw, cn), np.uint8)
                arr[j] = VkxElem(call.src(m).value, optr.value, sw * cn, dw * cn, cn, 0)
            else:
                raise TypeError(f'unsupported dtype {m.dtype}')
            outs.append(out)
        check(call.fn('vkx_grid_remap')(call.ctx.handle, arr, len(chunk), sh, sw, sv_p, dv_p, rows, cols, dh, dw))
    return outs


def remap_multi(mats, map_x, map_y, ctx=None):
    """cv.remap of every array of ``mats`` (uint8 HxW[xC] / float32 HxW, one source shape) through one dense map."""
    ctx = ctx or default_ctx()
    map_x = np.ascontiguousarray(map_x, dtype=np.float32)
    map_y = np.ascontiguousarray(map_y, dtype=np.float32)
    if map_x.shape != map_y.shape or map_x.ndim != 2:
        raise ValueError('map_x / map_y must be 2-D and of one shape')
    dh, dw = map_x.shape
    outs = []
    for i in range(0, len(mats), 8):
        chunk = [np.ascontiguousarray(m) for m in mats[i:i + 8]]
        sh, sw = chunk[0].shape[:2]
        arr = (VkxElem * len(chunk))()
        chunk_out = []
        for j, m in enumerate(chunk):
            if m.shape[:2] != (sh, sw):
                raise ValueError('all elements of one call must share the source shape')
            if m.dtype == np.float32 and m.ndim == 2:
                out = ctx.pinned_empty((dh, dw), np.float32)
                arr[j] = VkxElem(m.ctypes.data, out.ctypes.data, sw, dw, 1, 1)
            elif m.dtype == np.uint8:
                cn = 1 if m.ndim == 2 else m.shape[2]
                out = _out_like(m, dh, dw, ctx)
                arr[j] = VkxElem(m.ctypes.data, out.ctypes.data, sw * cn, dw * cn, cn, 0)
            else:
                raise TypeError(f'unsupported element {m.dtype} {m.shape}')
            chunk_out.append(out)
        check(lib().vkx_remap_multi(ctx.handle, arr, len(chunk), sh, sw, _ptr(map_x), _ptr(map_y), dw, dh, dw))
        outs.extend(chunk_out)
    return outs


def project_points(src_vertices, dst_vertices, grid_size, points_xy, smooth_xy, ctx=None):
    """FuncImageGridBased.func_point for a batch: ``points_xy`` int [n, 2] rounded (x, y), ``smooth_xy`` float64 [n, 2];
    returns float64 [n, 2] (x', y').  IndexError for a point outside the lattice cells, like the reference."""
    ctx = ctx or default_ctx()
    sv, dv = _vertices(src_vertices), _vertices(dst_vertices)
    rows, cols = sv.shape[:2]
    pi = np.ascontiguousarray(points_xy, dtype=np.int32).reshape(-1, 2)
    ps = np.ascontiguousarray(smooth_xy, dtype=np.float64).reshape(-1, 2)
    if pi.shape != ps.shape:
        raise ValueError('points_xy and smooth_xy must have the same length')
    out = ctx.pinned_empty(ps.shape, ps.dtype)
    rc = lib().vkx_grid_project_points(ctx.handle, _ptr(sv), _ptr(dv), rows, cols, int(grid_size), _ptr(pi), _ptr(ps),
                                       pi.shape[0], _ptr(out))
    if rc == ERR_OUT_OF_LATTICE:
        raise IndexError(last_error())
    check(rc)
    return out


def mls_project(src_handles_xy, dst_handles_xy, src_handles_smooth_xy, dst_handles_smooth_xy, vertices_xy, ctx=None):
    """SimilarityMlsPointProjector.project_point (geometric/mls.py:38-135) for every vertex in one launch.

    ``src_handles_xy`` / ``dst_handles_xy``: float32 [n, 2], the integer handle positions the reference feeds its float32
    arithmetic (``PointTuple.to_smooth_np_array``); ``*_smooth_xy``: float64 [n, 2], the handles' smooth positions (a
    vertex exactly on a source handle maps to its target); ``vertices_xy``: float64 [m, 2].  Returns float64 [m, 2].
    FloatingPointError where the reference's ``np.errstate(divide='raise')`` fires."""
    ctx = ctx or default_ctx()
    p = np.ascontiguousarray(src_handles_xy, dtype=np.float32).reshape(-1, 2)
    q = np.ascontiguousarray(dst_handles_xy, dtype=np.float32).reshape(-1, 2)
    ps = np.ascontiguousarray(src_handles_smooth_xy, dtype=np.float64).reshape(-1, 2)
    qs = np.ascontiguousarray(dst_handles_smooth_xy, dtype=np.float64).reshape(-1, 2)
    if not (p.shape == q.shape == ps.shape == qs.shape):
        raise ValueError('handle arrays differ in length')
    v = np.ascontiguousarray(vertices_xy, dtype=np.float64).reshape(-1, 2)
    out = ctx.pinned_empty(v.shape, v.dtype)
    rc = lib().vkx_mls_project(ctx.handle, _ptr(p), _ptr(q), _ptr(ps), _ptr(qs), p.shape[0], _ptr(v), v.shape[0], _ptr(out))
    if rc == ERR_DIVIDE:
        raise FloatingPointError(last_error())
    check(rc)
    return out


def noise_normal_i16(shape, std, seed, ctx=None):
    """Throughput-mode noise plane: int16 ``shape`` = (h, w[, cn]) with the distribution of ``np.round(rng.normal(0, std))``,
    drawn on the device from Philox2x32-10 keyed by ``seed`` (include/vkx.h: vkx_noise_normal_i16).  NOT the reference's
    values -- those come from the caller's numpy stream (``gaussion_noise_plane``)."""
    ctx = ctx or default_ctx()
    h, w = int(shape[0]), int(shape[1])
    cn = int(shape[2]) if len(shape) > 2 else 1
    out = ctx.pinned_empty(tuple(shape), np.int16)
    check(lib().vkx_noise_normal_i16(ctx.handle, _ptr(out), w * cn, h, w, cn, float(std), int(seed) & 0xffffffffffffffff))
    return out


def noise_normal_table(std):
    table = np.empty(65536, np.int16)
    check(lib().vkx_noise_normal_table(float(std), _ptr(table)))
    return table


def gaussian_blur(img, ksize, sigma, ctx=None):
    call = _Call(ctx, img)
    h, w, cn, stride = _shape_u8(img)
    dst, dptr = call.out(img.shape, np.uint8)
    check(call.fn('vkx_gaussian_blur_u8')(call.ctx.handle, call.src(img), h, w, cn, stride, int(ksize), float(sigma), dptr, stride))
    return dst


def color_shift_rgb(img, delta, ctx=None):
    call = _Call(ctx, img)
    h, w, cn, stride = _shape_u8(img)
    if cn != 3:
        raise ValueError('color_shift_rgb needs an HxWx3 image')
    dst, dptr = call.out(img.shape, np.uint8)
    check(call.fn('vkx_color_shift_rgb')(call.ctx.handle, call.src(img), h, w, stride, int(delta), dptr, stride))
    return dst


def cvt_rgb_hsv(img, to_hsv, ctx=None):
    call = _Call(ctx, img)
    h, w, cn, stride = _shape_u8(img)
    if cn != 3:
        raise ValueError('RGB <-> HSV needs an HxWx3 image')
    dst, dptr = call.out(img.shape, np.uint8)
    check(call.fn('vkx_cvt_rgb_hsv_u8')(call.ctx.handle, call.src(img), h, w, stride, int(bool(to_hsv)), dptr, stride))
    return dst


def mean_shift(img, delta, threshold=None, channels=None, cycle=False, ctx=None):
    call = _Call(ctx, img)
    h, w, cn, stride = _shape_u8(img)
    dst, dptr = call.out(img.shape, np.uint8)
    check(call.fn('vkx_mean_shift_u8')(call.ctx.handle, call.src(img), h, w, cn, stride, int(delta), int(threshold is not None),
                                       int(threshold or 0), int(bool(cycle)), _channel_mask(channels), dptr, stride))
    return dst


POINT_COMPLEMENT, POINT_POSTERIZE, POINT_PERMUTE = 0, 1, 2
(CVT_RGB2HSV_FULL, CVT_HSV2RGB_FULL, CVT_RGB2HLS_FULL, CVT_HLS2RGB_FULL, CVT_RGB2GRAY, CVT_GRAY2RGB, CVT_RGBA2RGB, CVT_RGB2RGBA,
 CVT_GRAY2RGBA, CVT_RGBA2GRAY) = range(10)
_CVT_CHANNELS = {CVT_RGB2GRAY: (3, 1), CVT_GRAY2RGB: (1, 3), CVT_RGBA2RGB: (4, 3), CVT_RGB2RGBA: (3, 4), CVT_GRAY2RGBA: (1, 4),
                 CVT_RGBA2GRAY: (4, 1)}


def cvt_color(img, code, ctx=None):
    """cv.cvtColor for the codes of include/vkx.h (VKX_CVT_*)."""
    call = _Call(ctx, img)
    h, w, cn, stride = _shape_u8(img)
    want_cn, out_cn = _CVT_CHANNELS.get(code, (3, 3))
    if cn != want_cn:
        raise ValueError(f'conversion code {code} takes {want_cn}-channel input')
    dst, dptr = call.out((h, w) if out_cn == 1 else (h, w, out_cn), np.uint8)
    check(call.fn('vkx_cvt_color_u8')(call.ctx.handle, call.src(img), h, w, stride, int(code), dptr, w * out_cn))
    return dst


def blend_u8(a, b, w0, w1, channels=None, ctx=None):
    """uint8(clip(w0 * a + w1 * b, 0, 255)) on ``channels`` (None = all), ``b`` elsewhere (include/vkx.h vkx_blend_u8)."""
    call = _Call(ctx, a, b)
    h, w, cn, stride = _shape_u8(a)
    if tuple(b.shape) != tuple(a.shape) or np.dtype(b.dtype) != np.uint8:
        raise ValueError('the two planes must agree in shape and dtype')
    dst, dptr = call.out(a.shape, np.uint8)
    check(call.fn('vkx_blend_u8')(call.ctx.handle, call.src(a), stride, call.src(b), stride, h, w, cn, float(w0), float(w1),
                                  _channel_mask(channels), dptr, stride))
    return dst


def fog_f32(img, weight, fog_values, ctx=None):
    """uint8(clip((1 - weight) * img + weight * fog_values[c], 0, 255)): float32 weight plane (H, W), float32 fog per channel."""
    call = _Call(ctx, img, weight)
    h, w, cn, stride = _shape_u8(img)
    if tuple(weight.shape) != (h, w):
        raise ValueError('weight plane must be (H, W)')
    fog = np.ascontiguousarray(np.asarray(fog_values, dtype=np.float32).reshape(-1))
    if fog.shape[0] != cn:
        raise ValueError('one fog value per channel')
    dst, dptr = call.out(img.shape, np.uint8)
    check(call.fn('vkx_fog_f32_u8')(call.ctx.handle, call.src(img), h, w, cn, stride, call.src(weight, np.float32), w, _ptr(fog), dptr,
                                    stride))
    return dst


def brightness_shift_rgb(img, delta, ctx=None):
    call = _Call(ctx, img)
    h, w, cn, stride = _shape_u8(img)
    if cn != 3:
        raise ValueError('expected an RGB image')
    dst, dptr = call.out(img.shape, np.uint8)
    check(call.fn('vkx_brightness_shift_rgb')(call.ctx.handle, call.src(img), h, w, stride, int(delta), dptr, stride))
    return dst


def color_balance_rgb(img, ratio, ctx=None):
    call = _Call(ctx, img)
    h, w, cn, stride = _shape_u8(img)
    if cn != 3:
        raise ValueError('expected an RGB image')
    dst, dptr = call.out(img.shape, np.uint8)
    check(call.fn('vkx_color_balance_rgb')(call.ctx.handle, call.src(img), h, w, stride, float(ratio), dptr, stride))
    return dst


def _channel_mask(channels):
    chmask = 0
    for c in channels or ():
        chmask |= 1 << int(c)
    return chmask


def pointwise(img, op, p0=0, p1=0, channels=None, ctx=None):
    """complement / posterization / channel permutation (include/vkx.h VKX_POINT_*)."""
    call = _Call(ctx, img)
    h, w, cn, stride = _shape_u8(img)
    dst, dptr = call.out(img.shape, np.uint8)
    check(call.fn('vkx_pointwise_u8')(call.ctx.handle, call.src(img), h, w, cn, stride, int(op), int(p0), int(p1),
                                      _channel_mask(channels), dptr, stride))
    return dst


def permute_channels(img, indices, ctx=None):
    """img[:, :, indices]."""
    packed = 0
    for c, idx in enumerate(indices):
        packed |= (int(idx) & 3) << (2 * c)
    if len(indices) != (1 if img.ndim == 2 else img.shape[2]):
        raise ValueError('one index per channel')
    return pointwise(img, POINT_PERMUTE, packed, ctx=ctx)


def histogram(img, ctx=None):
    """Per-channel histogram, int32 [cn, 256] (a host array: the tables built from it are host arithmetic)."""
    call = _Call(ctx, img)
    h, w, cn, stride = _shape_u8(img)
    hist, hptr = call.out((cn, 256), np.int32)
    if not call.dev:
        hist[...] = 0
    check(call.fn('vkx_histogram_u8')(call.ctx.handle, call.src(img), h, w, cn, stride, hptr))
    return np.array(host_array(hist))


_reduce_order_ok = None


def numpy_reduce_order_ok() -> bool:
    """csrc/reduce.hip adds in the order THIS numpy adds when it reduces a float32 array -- pieces of its 8 192-element reduction
    buffer for a contiguous axis, one element after the other along the outer axis of an interleaved image -- which is an
    implementation detail of numpy (2.2.x here), not a contract: ``np.setbufsize`` or another release may change it, and beyond 2^24
    the float32 sum then differs in its last bits.  One probe per process whose sums leave the exact range compares numpy with the
    two orders restated in plain numpy; the device mean is used only while they agree and the buffer size is the default."""
    global _reduce_order_ok
    if np.getbufsize() != 8192:
        return False
    if _reduce_order_ok is None:
        rng = np.random.default_rng(20240917)
        px = (rng.integers(0, 128, (200_003, 3)) * 2 + 1).astype(np.uint8)       # odd values: ties in every binade; sums ~ 2.5e7 > 2^24
        f = px.astype(np.float32)

        def pieces(col):
            total = np.float32(0)
            flat = col.astype(np.int64)
            for lo in range(0, flat.size, 8192):
                total = np.float32(total + np.float32(int(flat[lo:lo + 8192].sum())))
            return total

        ok = True
        sequential = np.mean(f.reshape(-1, 3), axis=0)
        for c in range(3):
            ok = ok and np.float32(np.cumsum(f[:, c], dtype=np.float32)[-1] / np.float32(px.shape[0])) == sequential[c]
        plane = px[:182_000, 0].reshape(2000, 91)
        ok = ok and np.float32(pieces(plane.reshape(-1)) / plane.size) == np.mean(plane.astype(np.float32))
        img = px[:199_800].reshape(600, 333, 3)
        picked = img[:, :, [2, 0]].astype(np.float32)
        want = np.mean(picked.reshape(-1, 2), axis=0)
        for k, c in enumerate((2, 0)):
            ok = ok and np.float32(pieces(img[:, :, c].reshape(-1)) / (600 * 333)) == want[k]
        _reduce_order_ok = bool(ok)
    return _reduce_order_ok


def mean_f32_u8(img, channels=None, ctx=None):
    """``np.mean`` of the float32 copy of a uint8 image, the way ``std_shift`` takes it (photometric/color.py:165-210) and with
    numpy's own roundings: ``np.mean(mat)`` for an H x W image (a numpy float32 scalar), ``np.mean(mat.reshape(-1, k), axis=0)`` for
    the ``k`` selected channels of an H x W x C one (float32 [k]); ``channels`` = None: all.  None when the image is outside the
    device path's limits (more than 2^22 pixels) or the installed numpy does not add in the order the kernel restates
    (``numpy_reduce_order_ok``): the caller takes numpy."""
    h, w, cn, stride = _shape_u8(img)
    if h * w == 0 or h * w > (1 << 22) or not numpy_reduce_order_ok():
        return None
    sel = list(range(cn)) if channels is None else [int(c) for c in channels]
    if not 1 <= len(sel) <= 4:
        return None
    call = _Call(ctx, img)
    idx = np.asarray(sel, dtype=np.int32)
    sums = np.zeros(len(sel), np.float32)
    # all channels of an interleaved image reduce along the outer axis (sequentially); a plane and picked channels piecewise
    sequential = int(channels is None and img.ndim == 3 and cn >= 2)
    check(call.fn('vkx_sum_f32_u8')(call.ctx.handle, call.src(img), h, w, cn, stride, _ptr(idx), len(sel), sequential, _ptr(sums)))
    mean = sums / (h * w)        # float32 / int, as np.mean divides (um.true_divide(ret, rcount)): float32
    return np.float32(mean[0]) if img.ndim == 2 else mean


def filter2d(img, kernel, ctx=None):
    """cv.filter2D(img, -1, kernel) for uint8 images and float32 kernels of up to 15 x 15 taps."""
    call = _Call(ctx, img)
    h, w, cn, stride = _shape_u8(img)
    kernel = np.ascontiguousarray(kernel, dtype=np.float32)
    if kernel.ndim != 2:
        raise ValueError('kernel must be 2-D')
    dst, dptr = call.out(img.shape, np.uint8)
    check(call.fn('vkx_filter2d_u8')(call.ctx.handle, call.src(img), h, w, cn, stride, _ptr(kernel), kernel.shape[0], kernel.shape[1],
                                     dptr, stride))
    return dst


def apply_lut(img, lut, channels=None, ctx=None):
    """dst[..., c] = lut[c][img[..., c]] on the selected channels; lut uint8 [cn, 256]."""
    call = _Call(ctx, img)
    h, w, cn, stride = _shape_u8(img)
    lut = np.ascontiguousarray(lut, dtype=np.uint8)
    if lut.shape != (cn, 256):
        raise ValueError(f'table must be uint8 [{cn}, 256]')
    dst, dptr = call.out(img.shape, np.uint8)
    check(call.fn('vkx_apply_lut_u8')(call.ctx.handle, call.src(img), h, w, cn, stride, _ptr(lut), _channel_mask(channels), dptr, stride))
    return dst


def apply_lut_planes(planes, luts):
    """``lut_k[plane_k]`` for up to eight dense single-channel uint8 DevArrays of one context, each with its own table uint8 [256], in ONE
    launch (vkx_apply_lut_u8_planes_dev) -> new DevArrays."""
    n = len(planes)
    if n == 0:
        return []
    ctx = planes[0].ctx
    arr = (VkxLutPlane * n)()
    outs, keep = [], []
    for k, (plane, lut) in enumerate(zip(planes, luts)):
        if not isinstance(plane, DevArray) or np.dtype(plane.dtype) != np.uint8 or plane.ctx is not ctx:
            raise ValueError('apply_lut_planes takes uint8 DevArrays of one context')
        table = np.ascontiguousarray(lut, dtype=np.uint8).reshape(-1)
        if table.shape != (256,):
            raise ValueError('one table uint8 [256] per plane')
        out = ctx.dev_empty(plane.shape, np.uint8)
        arr[k].src, arr[k].dst, arr[k].n_bytes, arr[k].lut_host = plane.ptr, out.ptr, plane.nbytes, table.ctypes.data
        outs.append(out)
        keep.append(table)
    check(lib().vkx_apply_lut_u8_planes_dev(ctx.handle, arr, n))
    return outs


def gather(img, pos_y, pos_x, ctx=None):
    """img[pos_y, pos_x] for two integer index planes of one shape (host arrays, or int32 DevArrays as glass_shuffle_planes_dev leaves them)."""
    call = _Call(ctx, img, pos_y, pos_x)
    h, w, cn, stride = _shape_u8(img)
    if not isinstance(pos_y, DevArray):
        pos_y = np.ascontiguousarray(pos_y, dtype=np.int32)
    if not isinstance(pos_x, DevArray):
        pos_x = np.ascontiguousarray(pos_x, dtype=np.int32)
    if len(pos_y.shape) != 2 or tuple(pos_y.shape) != tuple(pos_x.shape) or np.dtype(pos_y.dtype) != np.int32 or np.dtype(pos_x.dtype) != np.int32:
        raise ValueError('index planes must be 2-D int32 and of one shape')
    dh, dw = pos_y.shape
    dst, dptr = call.out((dh, dw) + tuple(img.shape[2:]), np.uint8)
    check(call.fn('vkx_gather_u8')(call.ctx.handle, call.src(img), h, w, cn, stride, call.src(pos_y), call.src(pos_x), dw, dptr, dh, dw,
                                   dw * cn))
    return dst


def glass_shuffle_planes_dev(shape, delta, loop, rng, ctx=None):
    """The (row, column) source planes of glass_blur's shuffle (reference photometric/blur.py:204-250) as int32 DevArrays: the lattice
    phases and jumps are ``rng``'s own draws, call for call the reference's; the swaps -- numpy's tuple assignment with its order of
    duplicates -- run on the device (``vkx_glass_round_dev``), the planes never exist on the host."""
    ctx = ctx or default_ctx()
    height, width = int(shape[0]), int(shape[1])
    pos_y, pos_x = ctx.dev_empty((height, width), np.int32), ctx.dev_empty((height, width), np.int32)
    check(lib().vkx_glass_init_dev(ctx.handle, c_void_p(pos_y.ptr), c_void_p(pos_x.ptr), height, width))
    pitch = 2 * delta + 1
    for _ in range(loop):
        r0 = int(rng.integers(0, pitch))
        n_rows = len(range(r0, height - delta, pitch))
        c0 = int(rng.integers(0, pitch))
        n_cols = len(range(c0, width - delta, pitch))
        grid = (n_rows, n_cols)
        jump_y = np.ascontiguousarray(rng.integers(-delta, delta + 1, grid), dtype=np.int32)
        jump_x = np.ascontiguousarray(rng.integers(-delta, delta + 1, grid), dtype=np.int32)
        check(lib().vkx_glass_round_dev(ctx.handle, c_void_p(pos_y.ptr), c_void_p(pos_x.ptr), height, width, r0, c0, pitch, n_rows, n_cols,
                                        _ptr(jump_y), _ptr(jump_x)))
    return pos_y, pos_x


def saturate_i64(samples, ctx=None):
    """np.clip(samples, 0, 255).astype(np.uint8) for an int64 array."""
    call = _Call(ctx, samples)
    dst, dptr = call.out(np.shape(samples), np.uint8)
    check(call.fn('vkx_saturate_i64_u8')(call.ctx.handle, call.src(samples, np.int64), int(np.prod(np.shape(samples))), dptr))
    return dst


def impulse_noise(img, selector, ctx=None):
    call = _Call(ctx, img)
    h, w, cn, stride = _shape_u8(img)
    if tuple(np.shape(selector)) != (h, w):
        raise ValueError('selector plane must be (H, W)')
    dst, dptr = call.out(img.shape, np.uint8)
    check(call.fn('vkx_impulse_noise_u8')(call.ctx.handle, call.src(img), h, w, cn, stride, call.src(selector, np.uint8), w, dptr, stride))
    return dst


def speckle_noise(img, noise, ctx=None):
    call = _Call(ctx, img)
    h, w, cn, stride = _shape_u8(img)
    if tuple(np.shape(noise)) != tuple(img.shape):
        raise ValueError('noise plane must have the image shape')
    dst, dptr = call.out(img.shape, np.uint8)
    check(call.fn('vkx_speckle_noise_u8')(call.ctx.handle, call.src(img), h, w, cn, stride, call.src(noise, np.float64), w * cn, dptr,
                                          stride))
    return dst


def add_noise_i16(img, noise, ctx=None):
    call = _Call(ctx, img)
    h, w, cn, stride = _shape_u8(img)
    if tuple(np.shape(noise)) != tuple(img.shape):
        raise ValueError('noise plane must have the image shape')
    dst, dptr = call.out(img.shape, np.uint8)
    check(call.fn('vkx_add_noise_i16')(call.ctx.handle, call.src(img), h, w, cn, stride, call.src(noise, np.int16), w * cn, dptr, stride))
    return dst


# ---- the caller's numpy Generator stream drawn on the device (include/vkx.h: vkx_np_*) -------------------------------
_PCG_MULT = 0x2360ED051FC65DA44385DF649FCCF645
_M128 = (1 << 128) - 1
_M64 = (1 << 64) - 1


def np_stream(rng):
    """(state, inc) of a numpy Generator over PCG64, or None for any other bit generator (the caller then draws on the
    host).  ``VKX_HOST_RNG=1`` forces the host path."""
    if os.environ.get('VKX_HOST_RNG', '') == '1':
        return None
    bit_generator = getattr(rng, 'bit_generator', None)
    if type(bit_generator).__name__ != 'PCG64':
        return None
    st = bit_generator.state['state']
    return int(st['state']), int(st['inc'])


def pcg64_jump(state, inc, delta):
    """PCG64 state after ``delta`` raw draws (pcg_advance_lcg_128)."""
    acc_mult, acc_plus, cur_mult, cur_plus = 1, 0, _PCG_MULT, inc
    while delta > 0:
        if delta & 1:
            acc_mult = (acc_mult * cur_mult) & _M128
            acc_plus = (acc_plus * cur_mult + cur_plus) & _M128
        cur_plus = ((cur_mult + 1) * cur_plus) & _M128
        cur_mult = (cur_mult * cur_mult) & _M128
        delta >>= 1
    return (acc_mult * state + acc_plus) & _M128


def np_consume(rng, draws):
    """Moves the caller's generator past ``draws`` raw 64-bit draws, as if numpy had made them: the LCG state jumps, a
    buffered 32-bit half (``has_uint32``) stays where it is (``bit_generator.advance`` would drop it)."""
    st = rng.bit_generator.state
    st['state']['state'] = pcg64_jump(int(st['state']['state']), int(st['state']['inc']), int(draws))
    rng.bit_generator.state = st


class NpResults:
    """``VkxNpResult[n]`` in page-locked memory: the device -> host copy of vkx_np_draw_batch_dev stays asynchronous (into a
    pageable array the runtime stages it and blocks the calling thread until the stream has drained)."""

    def __init__(self, ctx, n):
        self._ctx, self.n = ctx, int(n)
        self._ptr = ctx.host_alloc(max(1, self.n) * ctypes.sizeof(VkxNpResult))
        self.array = (VkxNpResult * max(1, self.n)).from_address(self._ptr)

    def __getitem__(self, i):
        return self.array[i]

    def close(self):
        if self._ptr:
            _free_pinned(self._ctx, self._ptr)
            self._ptr = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def np_job(kind, stream, n, scale=0.0, cdf=(2.0, 2.0, 2.0), cn=1, src=None, dst=None):
    job = VkxNpJob()
    state, inc = stream
    job.state[0], job.state[1] = state & _M64, state >> 64
    job.inc[0], job.inc[1] = inc & _M64, inc >> 64
    job.n, job.kind, job.cn, job.scale = int(n), kind, cn, float(scale)
    for k in range(3):
        job.cdf[k] = float(cdf[k])
    job.src, job.dst = src, dst
    return job


def np_draw(kind, rng, out_shape, out_dtype, src=None, scale=0.0, cdf=(2.0, 2.0, 2.0), cn=1, ctx=None):
    """One stream job (``vkx_np_draw`` on host arrays, ``vkx_np_draw_batch_dev`` when ``src`` is a DevArray or in resident mode).
    Returns the result (numpy array or DevArray) and advances ``rng`` when the device result is known to be numpy's; None
    (``rng`` untouched) when the generator is not PCG64, the stream is outside the device path's limits, or a decision fell
    inside the libm ambiguity margin -- the caller then draws on the host."""
    stream = np_stream(rng)
    size = int(np.prod(out_shape, dtype=np.int64))
    if stream is None or size == 0:
        return None
    call = _Call(ctx, *([src] if src is not None else []))
    n = size // cn if kind == NP_IMPULSE_U8 else size
    dst, dptr = call.out(out_shape, out_dtype)
    job = np_job(kind, stream, n, scale, cdf, cn, call.src(src) if src is not None else None, dptr)
    res = VkxNpResult()
    if call.dev:
        rc = lib().vkx_np_draw_batch_dev(call.ctx.handle, ctypes.byref(job), 1, ctypes.byref(res))
        if rc == 0:
            call.ctx.sync()             # the verdict on the stream (flags) decides whether the result stands
    else:
        rc = lib().vkx_np_draw(call.ctx.handle, ctypes.byref(job), ctypes.byref(res))
    if rc == ERR_INVALID:   # a stream the device path does not take (scale / length limits): host
        return None
    check(rc)
    if res.flags:
        return None
    np_consume(rng, res.draws)
    return dst


def np_tiles_layout(n):
    """(tiles, slot elements, table offset, slots offset, bytes) of the tile buffer of a ``NP_NORMAL_TILES`` job of ``n`` samples."""
    vals = [ctypes.c_int64() for _ in range(5)]
    check(lib().vkx_np_tiles_layout(int(n), *[ctypes.byref(v) for v in vals]))
    return tuple(int(v.value) for v in vals)


def np_tiles_plane(buffer, n):
    """numpy restatement of ``vkx_np_tiles_expand_dev``: the int16 plane a finished tile buffer (host copy, uint8) stands for."""
    tiles, slot, table_off, slots_off, _bytes = np_tiles_layout(n)
    table = buffer[table_off:table_off + 8 * (tiles + 1)].view(np.uint32).reshape(tiles + 1, 2)
    slots = buffer[slots_off:slots_off + tiles * slot * 2].view(np.int16).reshape(tiles, slot)
    out = np.empty(n, np.int16)
    for t in range(tiles):
        first, skip = int(table[t, 0]), int(table[t, 1])
        if first >= n:
            break
        count = min(int(table[t + 1, 0]), n) - first
        out[first:first + count] = slots[t, skip:skip + count]
    return out


def np_gaussion_noise(img, std, rng, ctx=None):
    """``clip(int16(img) + np.round(rng.normal(0, std, img.shape)).astype(int16), 0, 255)`` with the samples drawn on the
    device from ``rng``'s stream (photometric/noise.py:44-54); None when the host has to draw."""
    return np_draw(NP_NORMAL_ADD_U8, rng, img.shape, np.uint8, img, scale=std, ctx=ctx)


def np_normal_i16(shape, std, rng, ctx=None):
    """``np.round(rng.normal(0, std, shape)).astype(np.int16)`` drawn on the device; None when the host has to draw."""
    return np_draw(NP_NORMAL_I16, rng, shape, np.int16, scale=std, ctx=ctx)


def np_speckle_noise(img, std, rng, ctx=None):
    """``uint8(clip(img + img * rng.normal(0, std, img.shape), 0, 255))`` (photometric/noise.py:172-183)."""
    return np_draw(NP_SPECKLE_U8, rng, img.shape, np.uint8, img, scale=std, ctx=ctx)


def np_poisson_u8(img, rng, ctx=None):
    """``clip(rng.poisson(img.astype(float32)), 0, 255).astype(uint8)`` (photometric/noise.py:81-90) with every sample drawn on the
    device from ``rng``'s stream, value for value numpy's (``vkx_np_poisson_u8``), and ``rng`` moved past the draws numpy would have
    made.  None -- ``rng`` untouched -- when the generator is not PCG64 or the device declines (``np_poisson_last_flags()`` says why,
    per calling thread): the caller then calls ``rng.poisson`` itself."""
    _poisson_tls.flags = 0
    stream = np_stream(rng)
    if stream is None or img.dtype != np.uint8 or img.size == 0 or img.size > (1 << 30):
        return None
    call = _Call(ctx, img)
    dst, dptr = call.out(img.shape, np.uint8)
    state, inc = stream
    st = (ctypes.c_uint64 * 2)(state & _M64, state >> 64)
    ic = (ctypes.c_uint64 * 2)(inc & _M64, inc >> 64)
    consumed, flags = ctypes.c_longlong(0), ctypes.c_uint(0)
    check(call.fn('vkx_np_poisson_u8')(call.ctx.handle, st, ic, call.src(img), int(img.size), dptr, ctypes.byref(consumed), ctypes.byref(flags)))
    _poisson_tls.flags = int(flags.value)
    if flags.value:
        return None
    np_consume(rng, consumed.value)
    return dst


_poisson_tls = threading.local()


def np_poisson_last_flags() -> int:
    """VKX_NP_POISSON_* flags of the calling thread's last ``np_poisson_u8`` call (0: the device result stood, or the call never
    reached the device).  Thread-local: pipelines that draw from several threads do not see each other's refusals."""
    return getattr(_poisson_tls, 'flags', 0)


# fog lattices up to (2^12 + 1)^2 on the device: the context keeps 3 size^2 doubles of level scratch (1.6 GB at level 12) for its
# lifetime; a page beyond 4096 px per side (level 13: 6.4 GB, level 14: 26 GB) takes the numpy path instead of pinning that
FOG_DEVICE_MAX_LEVELS = 12


def np_fog_mask(shape, roughness, ratio_min, ratio_max, rng, ctx=None):
    """The fog density plane of ``fog`` (reference photometric/effect.py:89-216): ``generate_diamond_square_mask(shape, roughness, rng)``
    stretched to ``[ratio_min, ratio_max]``, float32 ``(H, W)`` as a DevArray, with ``rng`` left where the reference leaves it.  The four
    corner draws and the two crop offsets are ``rng``'s own calls on the host; the ~``size^2`` uniform draws of the lattice levels are
    taken from its PCG64 stream on the device (``vkx_fog_field_f32_dev``).  None -- ``rng`` untouched -- for another bit generator or a
    lattice without levels: the caller builds the field with numpy."""
    height, width = int(shape[0]), int(shape[1])
    size = int(2**np.ceil(np.log2(max(height, width))) + 1)
    levels = int(round(np.log2(size - 1))) if size > 2 else 0
    if np_stream(rng) is None or levels < 1 or levels > FOG_DEVICE_MAX_LEVELS or (1 << levels) + 1 != size:
        return None
    ctx = ctx or default_ctx()
    corners = np.array([rng.uniform(0.0, 1.0) for _ in range(4)]).astype(np.float32)      # field[0, 0], [0, -1], [-1, -1], [-1, 0]
    state, inc = np_stream(rng)
    st = (ctypes.c_uint64 * 2)(state & _M64, state >> 64)
    ic = (ctypes.c_uint64 * 2)(inc & _M64, inc >> 64)
    noise_weight = np.array([roughness**level for level in range(levels)], dtype=np.float64)
    field = ctx.dev_empty((size, size), np.float32)
    consumed = ctypes.c_longlong(0)
    check(lib().vkx_fog_field_f32_dev(ctx.handle, st, ic, levels, _ptr(noise_weight), _ptr(corners), c_void_p(field.ptr), ctypes.byref(consumed)))
    np_consume(rng, consumed.value)
    up = int(rng.integers(0, size - height + 1))
    left = int(rng.integers(0, size - width + 1))
    mask = ctx.dev_empty((height, width), np.float32)
    check(lib().vkx_fog_stretch_f32_dev(ctx.handle, c_void_p(field.ptr), size, up, left, height, width, float(ratio_max - ratio_min), float(ratio_min),
                                        c_void_p(mask.ptr)))
    return mask


def _choice_cdf(p):
    # Generator.choice: cdf = p.cumsum(); cdf /= cdf[-1]
    cdf = np.cumsum(np.asarray(p, dtype=np.float64))
    cdf /= cdf[-1]
    return cdf


def np_impulse_noise(img, prob_salt, prob_pepper, rng, ctx=None):
    """``rng.choice((0, 1, 2), size=(H, W), p=[keep, salt, pepper])`` drawn on the device and applied
    (photometric/noise.py:125-150)."""
    h, w, cn, _stride = _shape_u8(img)
    p = np.array([1 - prob_salt - prob_pepper, prob_salt, prob_pepper], dtype=np.float64)
    if not (p >= 0).all() or not np.isfinite(p).all():
        return None      # let numpy raise its own error on the host path
    return np_draw(NP_IMPULSE_U8, rng, img.shape, np.uint8, img, cdf=_choice_cdf(p), cn=cn, ctx=ctx)


def line_streak(img, thickness, gap, dash_thickness, dash_gap, color, alpha, enable_vert, enable_hori, ctx=None):
    """Returns a new array (the kernel works in place on a copy)."""
    call = _Call(ctx, img)
    h, w, cn, stride = _shape_u8(img)
    col = np.zeros(4, np.uint8)
    col[:cn] = np.asarray(color, dtype=np.uint8).reshape(-1)[:cn]
    if call.dev:
        out = call.ctx.dev_empty(img.shape, np.uint8)
        check(lib().vkx_memcpy_async(call.ctx.handle, STREAM_COMPUTE, c_void_p(out.ptr), call.src(img), out.nbytes, 2))
        optr = c_void_p(out.ptr)
    else:
        out = np.array(host_array(img), dtype=np.uint8, order='C')
        optr = _ptr(out)
    check(call.fn('vkx_line_streak_u8')(call.ctx.handle, optr, h, w, cn, stride, int(thickness), int(gap), int(dash_thickness),
                                        int(dash_gap), _ptr(col), float(alpha), int(bool(enable_vert)), int(bool(enable_hori))))
    return out


def ellipse_mask(mask, center, axes, thickness, ctx=None):
    """Draws cv.ellipse(mask, center=(x, y), axes=(a, b), 0, 0, 360, 1, thickness) for every row of ``axes`` onto the
    writable uint8 plane ``mask`` (in place)."""
    ctx = ctx or default_ctx()
    if mask.dtype != np.uint8 or mask.ndim != 2 or not mask.flags.c_contiguous or not mask.flags.writeable:
        raise ValueError('mask must be a writable C-contiguous uint8 plane')
    axes = np.ascontiguousarray(np.asarray(axes, dtype=np.int32).reshape(-1, 2))
    h, w = mask.shape
    check(lib().vkx_ellipse_mask_u8(ctx.handle, _ptr(mask), w, h, w, int(center[0]), int(center[1]), _ptr(axes),
                                    int(axes.shape[0]), int(thickness)))
    return mask


def ellipse_streak(img, center, axes, thickness, color, alpha, ctx=None):
    """ellipse_streak_image's raster and blend; returns a new array."""
    call = _Call(ctx, img)
    h, w, cn, stride = _shape_u8(img)
    axes = np.ascontiguousarray(np.asarray(axes, dtype=np.int32).reshape(-1, 2))
    col = np.zeros(4, np.uint8)
    col[:cn] = np.asarray(color, dtype=np.uint8).reshape(-1)[:cn]
    if call.dev:
        out = call.ctx.dev_empty(img.shape, np.uint8)
        check(lib().vkx_memcpy_async(call.ctx.handle, STREAM_COMPUTE, c_void_p(out.ptr), call.src(img), out.nbytes, 2))
        optr = c_void_p(out.ptr)
    else:
        out = call.ctx.pinned_empty(img.shape, np.uint8)
        np.copyto(out, host_array(img))
        optr = _ptr(out)
    check(call.fn('vkx_ellipse_streak_u8')(call.ctx.handle, optr, h, w, cn, stride, int(center[0]), int(center[1]), _ptr(axes),
                                           int(axes.shape[0]), int(thickness), _ptr(col), float(alpha)))
    return out


def fill_poly_mask(shape, pts, ctx=None):
    """cv.fillPoly(zeros(shape, uint8), [pts], 1); pts int (N, 2) as (x, y), all inside the array.  Always a host array (the
    host entry point clears the raster; its callers -- ``Polygon.np_mask`` -- go on with numpy)."""
    ctx = ctx or default_ctx()
    h, w = int(shape[0]), int(shape[1])
    pts = np.ascontiguousarray(np.asarray(pts, dtype=np.int32).reshape(-1, 2))
    mask = ctx.pinned_empty((h, w), np.uint8)
    check(lib().vkx_fill_poly_mask_u8(ctx.handle, _ptr(pts), int(pts.shape[0]), _ptr(mask), h, w, w))
    return mask


INTER_NEAREST, INTER_LINEAR, INTER_CUBIC = 0, 1, 2


INTER_NEAREST, INTER_LINEAR, INTER_CUBIC, INTER_AREA, INTER_LANCZOS4, INTER_LINEAR_EXACT, INTER_NEAREST_EXACT = range(7)


def resize(src, dsize_hw, interpolation, ctx=None):
    """cv.resize(src, (dw, dh), interpolation=<cv2 code 0..6>) for uint8 HxW[xC] or float32 HxW arrays."""
    call = _Call(ctx, src)
    dh, dw = int(dsize_hw[0]), int(dsize_hw[1])
    if np.dtype(src.dtype) == np.float32:
        if src.ndim != 2:
            raise ValueError('float32 planes are HxW')
        sh, sw = src.shape
        dst, dptr = call.out((dh, dw), np.float32)
        check(call.fn('vkx_resize_f32')(call.ctx.handle, call.src(src), sh, sw, sw, dptr, dh, dw, dw, int(interpolation)))
        return dst
    sh, sw, cn, stride = _shape_u8(src)
    dst, dptr = call.out((dh, dw) + tuple(src.shape[2:]), np.uint8)
    check(call.fn('vkx_resize_u8')(call.ctx.handle, call.src(src), sh, sw, cn, stride, dptr, dh, dw, dw * cn, int(interpolation)))
    return dst


def zoom_in_blur(img, sizes_hw, alpha, ctx=None):
    """include/vkx.h vkx_zoom_in_blur_u8: sizes_hw = [(height, width), ...] of the enlarged copies."""
    call = _Call(ctx, img)
    h, w, cn, stride = _shape_u8(img)
    sizes = np.ascontiguousarray(np.asarray(sizes_hw, dtype=np.int32).reshape(-1, 2))
    dst, dptr = call.out(img.shape, np.uint8)
    check(call.fn('vkx_zoom_in_blur_u8')(call.ctx.handle, call.src(img), h, w, cn, stride, _ptr(sizes), int(sizes.shape[0]), float(alpha),
                                         dptr, stride))
    return dst


def resize_cubic(src, dsize_hw, ctx=None):
    """cv.resize(src, (dw, dh), interpolation=cv.INTER_CUBIC) for uint8 HxW[xC] or float32 HxW arrays."""
    call = _Call(ctx, src)
    dh, dw = int(dsize_hw[0]), int(dsize_hw[1])
    if np.dtype(src.dtype) == np.float32:
        if src.ndim != 2:
            raise ValueError('float32 resize takes a 2-D array')
        sh, sw = src.shape
        dst, dptr = call.out((dh, dw), np.float32)
        check(call.fn('vkx_resize_cubic_f32')(call.ctx.handle, call.src(src), sh, sw, sw, dptr, dh, dw, dw))
        return dst
    sh, sw, cn, stride = _shape_u8(src)
    dst, dptr = call.out((dh, dw) + tuple(src.shape[2:]), np.uint8)
    check(call.fn('vkx_resize_cubic_u8')(call.ctx.handle, call.src(src), sh, sw, cn, stride, dptr, dh, dw, dw * cn))
    return dst


def _paint(flat, offsets, n, values, mask, score, ctx, fresh=False):
    plane = mask if mask is not None else score
    if plane is None:
        raise ValueError('mask or score is required')
    h, w = plane.shape
    dev = isinstance(plane, DevArray)
    if dev:
        # in place: on the stream that produced the planes (a second plane of another context is drained first)
        owner = plane.ctx
        for arr in (mask, score):
            if isinstance(arr, DevArray) and arr.ctx is not owner:
                arr.ctx.sync()
        if ctx is not owner:
            if ctx is not None and ctx is not default_ctx():
                ctx.sync()
            ctx = owner
    ctx = ctx or default_ctx()
    for arr, dt in ((mask, np.uint8), (score, np.float32)):
        if arr is None:
            continue
        if isinstance(arr, DevArray) != dev:
            raise ValueError('mask and score planes must both be numpy arrays or both DevArrays')
        if np.dtype(arr.dtype) != dt or tuple(arr.shape) != (h, w) or (not dev and (not arr.flags.c_contiguous or not arr.flags.writeable)):
            raise ValueError(f'planes must be writable C-contiguous {(h, w)} arrays (uint8 mask, float32 score)')
    vals = None
    if score is not None:
        vals = np.ascontiguousarray(np.asarray(values, dtype=np.float32))
        if vals.shape != (n,):
            raise ValueError('one value per polygon is required with a score plane')
    pm = (c_void_p(mask.ptr) if dev else _ptr(mask)) if mask is not None else None
    ps = (c_void_p(score.ptr) if dev else _ptr(score)) if score is not None else None
    if fresh and not dev:
        raise ValueError('fresh planes are device planes (host planes are painted in place)')
    fn = (lib().vkx_paint_polys_fresh_dev if fresh else lib().vkx_paint_polys_dev) if dev else lib().vkx_paint_polys
    check(fn(ctx.handle, _ptr(flat), _ptr(offsets), n, _ptr(vals) if vals is not None else None, pm, w, ps, w, h, w))
    for arr in (mask, score):
        if dev and arr is not None:
            arr.invalidate_host()


def paint_polys(polygons, values=None, mask=None, score=None, ctx=None, fresh=False):
    """Ordered paint of ``polygons`` (sequence of int (N_i, 2) arrays of (x, y) in plane coordinates) into the
    writable uint8 ``mask`` and / or float32 ``score`` planes (numpy arrays or DevArrays), in place: later polygons win on
    overlaps.  ``fresh=True`` (DevArrays only): the planes are uninitialised and every pixel of them is written (0 outside every
    polygon)."""
    pts = [np.asarray(p, dtype=np.int32).reshape(-1, 2) for p in polygons]
    offsets = np.zeros(len(pts) + 1, np.int32)
    if pts:
        offsets[1:] = np.cumsum([len(p) for p in pts])
    flat = np.ascontiguousarray(np.concatenate(pts, axis=0)) if pts else np.zeros((0, 2), np.int32)
    _paint(flat, offsets, len(pts), values, mask, score, ctx, fresh)


def paint_poly_sets_fresh(sets, shape, ctx=None):
    """The ordered paint of several label plane sets of one ``shape`` in ONE call (vkx_paint_poly_sets_fresh_dev: the four sets of a page).
    ``sets``: sequence of (points_xy int (N, 2), offsets int (P + 1), values float (P) or None, mask DevArray or None, score DevArray or
    None); the planes are uninitialised DevArrays of ``shape`` on one context and every pixel of them is written."""
    h, w = shape
    n = len(sets)
    if n == 0:
        return
    arr = (VkxPaintSet * n)()
    keep = []
    for k, (points_xy, offsets, values, mask, score) in enumerate(sets):
        flat = np.ascontiguousarray(points_xy, dtype=np.int32).reshape(-1, 2)
        offsets = np.ascontiguousarray(offsets, dtype=np.int32)
        n_polys = offsets.shape[0] - 1
        rec = arr[k]
        rec.pts_host, rec.poly_offsets_host, rec.n_polys = flat.ctypes.data, offsets.ctypes.data, n_polys
        keep += [flat, offsets]
        if score is not None:
            vals = np.ascontiguousarray(np.asarray(values if values is not None else (), dtype=np.float32))
            if vals.shape != (n_polys,):
                raise ValueError('one value per polygon is required with a score plane')
            rec.values_host = vals.ctypes.data
            keep.append(vals)
        for plane, dt, field, stride in ((mask, np.uint8, 'mask', 'mask_stride'), (score, np.float32, 'score', 'score_stride_el')):
            if plane is None:
                continue
            if not isinstance(plane, DevArray) or np.dtype(plane.dtype) != dt or tuple(plane.shape) != (h, w):
                raise ValueError(f'planes must be {(h, w)} DevArrays (uint8 mask, float32 score)')
            if ctx is None:
                ctx = plane.ctx
            elif plane.ctx is not ctx:
                raise ValueError('the planes of one call live on one context')
            setattr(rec, field, plane.ptr)
            setattr(rec, stride, w)
        if mask is None and score is None:
            raise ValueError('mask or score is required')
    check(lib().vkx_paint_poly_sets_fresh_dev(ctx.handle, arr, n, h, w))
    for _p, _o, _v, mask, score in sets:
        for plane in (mask, score):
            if plane is not None:
                plane.invalidate_host()


def paint_polys_flat(points_xy, offsets, values=None, mask=None, score=None, ctx=None, fresh=False):
    """``paint_polys`` for polygons that already are one int (N, 2) vertex array + (P + 1) offsets (element/soup.py)."""
    flat = np.ascontiguousarray(points_xy, dtype=np.int32).reshape(-1, 2)
    offsets = np.ascontiguousarray(offsets, dtype=np.int32)
    _paint(flat, offsets, offsets.shape[0] - 1, values, mask, score, ctx, fresh)


def dev_zeros(shape, dtype=np.uint8, ctx=None):
    """A zeroed DevArray."""
    ctx = ctx or default_ctx()
    arr = ctx.dev_empty(shape, dtype)
    if arr.nbytes:
        check(lib().vkx_memset(ctx.handle, c_void_p(arr.ptr), 0, arr.nbytes))
    return arr


_F32 = np.dtype(np.float32)
_U8 = np.dtype(np.uint8)
LAYER_MASK_ON_DEVICE, LAYER_ALPHA_ON_DEVICE, LAYER_VALUE_ON_DEVICE = 0x100, 0x200, 0x400      # include/vkx.h VKX_LAYER_*_ON_DEVICE


def _plane(plane, dtype):
    """(array to keep alive, its address): host planes C-contiguous in ``dtype``, DevArrays as they are."""
    if isinstance(plane, DevArray):
        return plane, plane.ptr
    if plane.dtype != dtype or not plane.flags.c_contiguous:
        plane = np.ascontiguousarray(plane, dtype=dtype)
    return plane, plane.__array_interface__['data'][0]


def make_layer(box, cn, value, mask=None, alpha=1.0, mode=FILL_PLAIN, dtype=np.uint8):
    """One composite layer for a uint8 (cn channels) or float32 (cn == 1) destination.
    box = (up, left, height, width).  Returns (VkxLayer | VkxLayerF32, keepalive list)."""
    up, left, bh, bw = box
    dtype = np.dtype(dtype)
    is_f32 = dtype == _F32
    if is_f32 and cn != 1:
        raise ValueError('float32 destinations are single channel')
    keep = []
    layer = VkxLayerF32() if is_f32 else VkxLayer()
    layer.up, layer.left, layer.height, layer.width, layer.mode = int(up), int(left), int(bh), int(bw), int(mode)
    if mask is not None:
        mask, address = _plane(mask, _U8)
        if tuple(mask.shape) != (bh, bw) or mask.dtype != _U8:
            raise ValueError(f'mask shape {mask.shape} != box shape {(bh, bw)}')
        keep.append(mask)
        layer.mask, layer.mask_stride = address, bw
    if isinstance(alpha, (np.ndarray, DevArray)):
        alpha, address = _plane(alpha, _F32)
        if tuple(alpha.shape) != (bh, bw) or alpha.dtype != _F32:
            raise ValueError(f'alpha shape {alpha.shape} != box shape {(bh, bw)}')
        keep.append(alpha)
        layer.alpha, layer.alpha_stride_el = address, bw
        layer.alpha_scalar = 1.0
    else:
        layer.alpha_scalar = float(alpha)
    if isinstance(value, (np.ndarray, DevArray)):
        value, address = _plane(value, dtype)
        want = (bh, bw) if cn == 1 and value.ndim == 2 else (bh, bw, cn)
        if tuple(value.shape) != want or value.dtype != dtype:
            raise RuntimeError('value is np.ndarray but shape is not matched.')
        keep.append(value)
        if is_f32:
            layer.value, layer.value_stride_el = address, bw
        else:
            layer.value, layer.value_stride = address, bw * cn
    elif is_f32:
        layer.value_const = float(np.float32(value))
    else:
        if isinstance(value, tuple):
            if len(value) != cn:
                raise RuntimeError('value is tuple but len(value) != num_channels.')
            vals = value
        else:
            vals = (value,) * cn
        const = layer.value_const
        for c in range(cn):
            v = vals[c]
            const[c] = int(v) if 0 <= v <= 255 else int(np.uint8(v))     # out of range: numpy's own complaint
    return layer, keep


def fill(dst, layers, ctx=None):
    """Applies ``layers`` (list of (layer, keepalive) from make_layer with dst's dtype) to ``dst`` in place: a writable uint8
    or float32 numpy array, or a DevArray (host planes of the layers are then uploaded for the call, DevArray planes are used
    where they are; a host destination takes host planes only)."""
    dev = isinstance(dst, DevArray)
    ctx = ctx or (dst.ctx if dev else None) or default_ctx()     # in place: on the stream that produced the destination
    if np.dtype(dst.dtype) not in (np.uint8, np.float32) or (not dev and (not dst.flags.c_contiguous or not dst.flags.writeable)):
        raise ValueError('dst must be a writable C-contiguous uint8 or float32 array')
    h, w = dst.shape[:2]
    cn = 1 if dst.ndim == 2 else dst.shape[2]
    is_f32 = np.dtype(dst.dtype) == np.float32
    cls = VkxLayerF32 if is_f32 else VkxLayer
    arr = (cls * max(len(layers), 1))()
    keep = []
    if dev and not is_f32:
        # host planes are staged by the library (read in place from its page-locked ring: no copy, no synchronisation), planes that
        # already are DevArrays are used where they are (their bit in ``mode`` says so); the page stays on the device
        for i, (layer, planes) in enumerate(layers):
            if not isinstance(layer, cls):
                raise TypeError('layer built for another destination dtype')
            arr[i] = layer
            for plane in planes:
                if isinstance(plane, DevArray):
                    keep.append(plane)
                    for field, bit in (('mask', LAYER_MASK_ON_DEVICE), ('alpha', LAYER_ALPHA_ON_DEVICE), ('value', LAYER_VALUE_ON_DEVICE)):
                        if getattr(layer, field) == plane.ptr:
                            arr[i].mode |= bit
        check(lib().vkx_fill_u8_dev_host_layers(ctx.handle, c_void_p(dst.ptr), h, w, cn, w * cn, arr, len(layers)))
        dst.invalidate_host()
        return dst
    for i, (layer, planes) in enumerate(layers):
        if not isinstance(layer, cls):
            raise TypeError('layer built for another destination dtype')
        arr[i] = layer
        if not dev and any(isinstance(plane, DevArray) for plane in planes):
            raise ValueError('a host destination cannot take device planes')
        if dev:
            # the record holds host addresses of the arrays in `planes` (or DevArrays' own addresses): device copies
            for field in ('mask', 'alpha', 'value'):
                addr = getattr(layer, field)
                if not addr:
                    continue
                for plane in planes:
                    if isinstance(plane, DevArray):
                        if plane.ptr == addr:
                            keep.append(plane)
                            break
                    elif plane.ctypes.data == addr:
                        d = ctx.to_device(plane)
                        keep.append(d)
                        setattr(arr[i], field, d.ptr)
                        break
                else:
                    raise ValueError('layer plane without its keepalive array: pass the pair make_layer returned')
    if is_f32 and cn != 1:
        raise ValueError('float32 destinations are single channel')
    dptr = c_void_p(dst.ptr) if dev else _ptr(dst)
    if is_f32:
        check((lib().vkx_fill_f32_dev if dev else lib().vkx_fill_f32)(ctx.handle, dptr, h, w, w, arr, len(layers)))
    else:
        check((lib().vkx_fill_u8_dev if dev else lib().vkx_fill_u8)(ctx.handle, dptr, h, w, cn, w * cn, arr, len(layers)))
    if dev:
        dst.invalidate_host()
    return dst
