"""The C-ABI library loads on a CPU-only box and exports every symbol include/vkx.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'vkx.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(vkx_[a-z0-9_]+)\s*\(', text)))


def test_header_declares_the_expected_surface():
    names = declared_symbols()
    for must in ('vkx_ctx_create', 'vkx_remap_u8', 'vkx_remap_u8_dev', 'vkx_grid_remap_dev', 'vkx_fill_u8',
                 'vkx_gaussian_blur_u8_dev', 'vkx_color_shift_rgb_dev', 'vkx_add_noise_i16_dev', 'vkx_warp_affine_u8'):
        assert must in names


def test_library_exports_every_declared_symbol():
    from vkit_amd import _native
    if not os.path.exists(_native.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    handle = ctypes.CDLL(_native.LIB_PATH)
    missing = [n for n in declared_symbols() if not hasattr(handle, n)]
    assert not missing, missing
    # and the Python binding covers the same surface
    assert sorted(_native.EXPORTED_SYMBOLS) == declared_symbols()
    assert _native.lib().vkx_version() >= 1


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    from vkit_amd import _native
    with pytest.raises(_native.VkxError):
        _native.Context(0)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'vkit_amd')
    offenders = []
    for dirpath, _dirs, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                text = open(os.path.join(dirpath, f), errors='replace').read()
                if re.search(r'^\s*(import|from)\s+oracle\b', text, flags=re.M) or 'vkx_oracle' in text.replace(
                        'oracle/vkx_oracle.c', ''):
                    offenders.append(os.path.join(dirpath, f))
    assert not offenders, offenders
