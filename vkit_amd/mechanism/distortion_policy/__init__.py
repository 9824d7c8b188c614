from .type import DistortionConfigGenerator, DistortionPolicy, DistortionPolicyFactory
from .random_distortion import (
    random_distortion_factory,
    RandomDistortion,
    RandomDistortionDebug,
    RandomDistortionFactoryConfig,
    RandomDistortionFactory,
    UNSUPPORTED_POLICY_NAMES,
)
