"""``DistortionPolicy``: a distortion + a level-driven config generator (reference: distortion_policy/type.py)."""
from typing import Any, Generic, Mapping, Optional, Tuple, Type, TypeVar, Union

from numpy.random import Generator as RandomGenerator

from vkit_amd.element import Shapable
from vkit_amd.utility import PathType, dyn_structure, get_generic_classes
from ..distortion.interface import Distortion, DistortionConfig, DistortionState

_T_GENERATOR_CONFIG = TypeVar('_T_GENERATOR_CONFIG')
_T_CONFIG = TypeVar('_T_CONFIG', bound=DistortionConfig)
_T_STATE = TypeVar('_T_STATE', bound=DistortionState)


class DistortionConfigGenerator(Generic[_T_GENERATOR_CONFIG, _T_CONFIG]):
    """Callable ``(shape, rng) -> config`` bound to a generator config and a level in 1..10."""

    @classmethod
    def get_generator_config_cls(cls) -> Type[_T_GENERATOR_CONFIG]:
        return get_generic_classes(cls)[0]  # type: ignore

    @classmethod
    def get_config_cls(cls) -> Type[_T_CONFIG]:
        return get_generic_classes(cls)[1]  # type: ignore

    def __init__(self, config: _T_GENERATOR_CONFIG, level: int) -> None:
        assert 1 <= level <= 10
        self.config = config
        self.level = level

    def __call__(self, shape: Tuple[int, int], rng: RandomGenerator) -> _T_CONFIG:
        raise NotImplementedError()


class DistortionPolicy(Generic[_T_GENERATOR_CONFIG, _T_CONFIG, _T_STATE]):

    def __init__(self, distortion: Distortion[_T_CONFIG, _T_STATE], config_for_config_generator: _T_GENERATOR_CONFIG,
                 config_generator_cls: Type[DistortionConfigGenerator[_T_GENERATOR_CONFIG, _T_CONFIG]]):
        self.distortion = distortion
        self.config_for_config_generator = config_for_config_generator
        self.config_generator_cls = config_generator_cls

    def distort(self, level: int, shapable_or_shape: Optional[Union[Shapable, Tuple[int, int]]] = None, *,
                rng: Optional[RandomGenerator] = None, enable_debug: bool = False, **elements):
        """Samples a config for ``level`` and applies the distortion to the given elements -- any of ``image``,
        ``mask``, ``score_map``, ``point``, ``points``, ``corner_points``, ``polygon``, ``polygons``, passed through
        to ``Distortion.distort`` (reference type.py:84-118); with ``enable_debug`` the result carries config and state."""
        return self.distortion.distort(
            self.config_generator_cls(self.config_for_config_generator, level), shapable_or_shape,
            rng=rng, get_config=enable_debug, get_state=enable_debug, **elements)

    @property
    def name(self):
        return self.config_generator_cls.get_config_cls().get_name()

    def __repr__(self):
        return f'DistortionPolicy({self.name})'


class DistortionPolicyFactory(Generic[_T_GENERATOR_CONFIG, _T_CONFIG, _T_STATE]):

    def __init__(self, distortion: Distortion[_T_CONFIG, _T_STATE],
                 config_generator_cls: Type[DistortionConfigGenerator[_T_GENERATOR_CONFIG, _T_CONFIG]]):
        self.distortion = distortion
        self.config_generator_cls = config_generator_cls

    def create(self, config: Optional[Union[Mapping[str, Any], PathType, _T_GENERATOR_CONFIG]] = None):
        config = dyn_structure(config, self.config_generator_cls.get_generator_config_cls(),
                               support_path_type=True, support_none_type=True)
        return DistortionPolicy(self.distortion, config, self.config_generator_cls)

    @property
    def name(self):
        return self.config_generator_cls.get_config_cls().get_name()
