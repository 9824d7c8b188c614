"""vkit_amd: MI355X-native (gfx950) implementation of vkit's per-pixel distortion hot path.

Python host code mirrors the reference's ``Distortion`` / ``DistortionPolicy`` operator API and calls the
hand-written HIP kernels of ``libvkx.so`` through the ctypes C ABI declared in ``include/vkx.h``.
"""
__version__ = '0.1.0'
