"""``gaussion_noise`` (sic, reference: photometric/noise.py:25-61), ``impulse_noise`` (:100-157) and
``speckle_noise`` (:160-190) and ``poisson_noise`` (:64-98): ``rng.poisson`` of the image's own values on the caller's stream
plus a saturating narrow.

The samples are the caller-visible numpy Generator stream -- ``np.round(rng.normal(0, std, shape))`` in C order, one
draw per channel value -- so that a stored ``config.rng_state`` reproduces the same pixels.  For the default bit
generator (PCG64) the stream itself is drawn on the device (``vkx_np_draw``: jump-ahead + ziggurat, value for value
numpy's) and the caller's generator is moved past the draws it would have made; any other bit generator, or a draw the
device declares ambiguous in the last bits of ``exp`` / ``log1p``, takes the host draw below and only the pixel
arithmetic runs on the GPU."""
from typing import Any, Mapping, Optional

import attrs
import numpy as np
from numpy.random import Generator as RandomGenerator

from vkit_amd import _native
from vkit_amd.element import Image
from ..interface import Distortion, DistortionConfig, DistortionNopState


@attrs.define
class GaussionNoiseConfig(DistortionConfig):
    std: float

    _rng_state: Optional[Mapping[str, Any]] = None

    @property
    def supports_rng_state(self) -> bool:
        return True

    @property
    def rng_state(self) -> Optional[Mapping[str, Any]]:
        return self._rng_state

    @rng_state.setter
    def rng_state(self, val: Mapping[str, Any]):
        self._rng_state = val


def gaussion_noise_plane(std: float, shape, rng: RandomGenerator) -> np.ndarray:
    """int16 round-half-even of N(0, std) draws, the reference's sample order."""
    return np.round(rng.normal(0, std, shape)).astype(np.int16)


def gaussion_noise_image(config: GaussionNoiseConfig, state, image: Image, rng: Optional[RandomGenerator]):
    assert rng
    mat = _native.np_gaussion_noise(image.arr, config.std, rng)
    if mat is None:
        noise = gaussion_noise_plane(config.std, image.arr.shape, rng)
        mat = _native.add_noise_i16(image.arr, noise)
    # the mode is re-inferred from the array, like the reference
    return Image(mat=mat)


gaussion_noise = Distortion(
    config_cls=GaussionNoiseConfig,
    state_cls=DistortionNopState[GaussionNoiseConfig],
    func_image=gaussion_noise_image,
)


@attrs.define
class ImpulseNoiseConfig(DistortionConfig):
    prob_salt: float
    prob_pepper: float

    _rng_state: Optional[Mapping[str, Any]] = None

    @property
    def supports_rng_state(self) -> bool:
        return True

    @property
    def rng_state(self) -> Optional[Mapping[str, Any]]:
        return self._rng_state

    @rng_state.setter
    def rng_state(self, val: Mapping[str, Any]):
        self._rng_state = val


def impulse_noise_image(config: ImpulseNoiseConfig, state, image: Image, rng: Optional[RandomGenerator]):
    """Salt (255) / pepper (0) on whole pixels; the per-pixel selector is ``rng.choice((0, 1, 2), size=(H, W), p=...)``
    on the caller-visible stream, the writes run on the GPU."""
    assert rng
    prob_presv = 1 - config.prob_salt - config.prob_pepper
    assert prob_presv >= 0.0
    mat = _native.np_impulse_noise(image.arr, config.prob_salt, config.prob_pepper, rng)
    if mat is None:
        selector = rng.choice((0, 1, 2), size=image.shape, p=[prob_presv, config.prob_salt, config.prob_pepper])
        mat = _native.impulse_noise(image.arr, selector.astype(np.uint8))
    return Image(mat=mat)


impulse_noise = Distortion(
    config_cls=ImpulseNoiseConfig,
    state_cls=DistortionNopState[ImpulseNoiseConfig],
    func_image=impulse_noise_image,
)


@attrs.define
class SpeckleNoiseConfig(DistortionConfig):
    std: float

    _rng_state: Optional[Mapping[str, Any]] = None

    @property
    def supports_rng_state(self) -> bool:
        return True

    @property
    def rng_state(self) -> Optional[Mapping[str, Any]]:
        return self._rng_state

    @rng_state.setter
    def rng_state(self, val: Mapping[str, Any]):
        self._rng_state = val


def speckle_noise_image(config: SpeckleNoiseConfig, state, image: Image, rng: Optional[RandomGenerator]):
    """``clip(px + px * N(0, std))`` with float64 samples, one per channel value in C order."""
    assert rng
    mat = _native.np_speckle_noise(image.arr, config.std, rng)
    if mat is None:
        noise = rng.normal(0, config.std, image.arr.shape)
        mat = _native.speckle_noise(image.arr, noise)
    return Image(mat=mat)


speckle_noise = Distortion(
    config_cls=SpeckleNoiseConfig,
    state_cls=DistortionNopState[SpeckleNoiseConfig],
    func_image=speckle_noise_image,
)


@attrs.define
class PoissonNoiseConfig(DistortionConfig):
    _rng_state: Optional[Mapping[str, Any]] = None

    @property
    def supports_rng_state(self) -> bool:
        return True

    @property
    def rng_state(self) -> Optional[Mapping[str, Any]]:
        return self._rng_state

    @rng_state.setter
    def rng_state(self, val: Mapping[str, Any]):
        self._rng_state = val


def poisson_noise_image(config: PoissonNoiseConfig, state, image: Image, rng: Optional[RandomGenerator]):
    """Every value is replaced by a Poisson draw with that value as its mean (float32 rates, C order)."""
    assert rng
    # the caller's PCG64 stream drawn on the device, value for value numpy's although every element takes a data-dependent number of
    # draws (vkx_np_poisson_u8); None: another bit generator, or one of the device path's rare refusals -- numpy draws, as before
    mat = _native.np_poisson_u8(image.arr, rng) if image.arr.dtype == np.uint8 else None
    if mat is None:
        samples = rng.poisson(image.mat.astype(np.float32))
        mat = _native.saturate_i64(samples)
    return Image(mat=mat)


poisson_noise = Distortion(
    config_cls=PoissonNoiseConfig,
    state_cls=DistortionNopState[PoissonNoiseConfig],
    func_image=poisson_noise_image,
)

