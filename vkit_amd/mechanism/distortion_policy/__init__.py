"""``vkit_amd.mechanism.distortion_policy``: level-driven policies over the distortions of the accelerated path."""
from .type import DistortionConfigGenerator, DistortionPolicy, DistortionPolicyFactory
from .random_distortion import (RandomDistortion, RandomDistortionDebug, RandomDistortionFactory,  # noqa: F401
                                RandomDistortionFactoryConfig, UNSUPPORTED_POLICY_NAMES, random_distortion_factory)
