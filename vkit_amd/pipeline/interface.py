"""Step protocol (reference: vkit/pipeline/interface.py:64-120), reduced to what the two steps need:
``Step(config).run(input, rng) -> output`` and a factory that structures a dict config."""
from typing import Any, Generic, Mapping, Optional, Type, TypeVar, Union

from numpy.random import Generator as RandomGenerator

from vkit_amd.utility import dyn_structure, get_generic_classes

_T_CONFIG = TypeVar('_T_CONFIG')
_T_INPUT = TypeVar('_T_INPUT')
_T_OUTPUT = TypeVar('_T_OUTPUT')


class PipelineStep(Generic[_T_CONFIG, _T_INPUT, _T_OUTPUT]):

    @classmethod
    def get_config_cls(cls) -> Type[_T_CONFIG]:
        return get_generic_classes(cls)[0]

    @classmethod
    def get_input_cls(cls) -> Type[_T_INPUT]:
        return get_generic_classes(cls)[1]

    @classmethod
    def get_output_cls(cls) -> Type[_T_OUTPUT]:
        return get_generic_classes(cls)[2]

    def __init__(self, config: _T_CONFIG):
        self.config = config

    def run(self, input: _T_INPUT, rng: RandomGenerator) -> _T_OUTPUT:
        raise NotImplementedError()


class PipelineStepFactory(Generic[_T_CONFIG, _T_INPUT, _T_OUTPUT]):

    def __init__(self, pipeline_step_cls: Type[PipelineStep]):
        self.pipeline_step_cls = pipeline_step_cls

    @property
    def name(self):
        return self.pipeline_step_cls.__name__

    def create(self, config: Optional[Union[Mapping[str, Any], Any]] = None):
        config_cls = self.pipeline_step_cls.get_config_cls()
        config = dyn_structure(config, config_cls, support_none_type=True)
        return self.pipeline_step_cls(config)
