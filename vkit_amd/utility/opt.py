"""Small host-side utilities the operator / policy layer relies on (reference: vkit/utility/opt.py).

``dyn_structure`` reproduces the reference's config loading contract (instance | mapping | JSON path | None
-> attrs instance, unknown keys rejected) without cattrs, which is not available in this environment.
"""
import enum
import json
import os
import re
import typing
from collections import abc
from os import PathLike
from typing import Any, Optional, Sequence, Type, TypeVar, Union

import attrs
from numpy.random import Generator as RandomGenerator

PathType = Union[str, PathLike]
_T = TypeVar('_T')


def _is_path(obj: Any):
    return isinstance(obj, (str, PathLike))


def _structure_value(value: Any, tp: Any):
    """Best-effort structuring of ``value`` into annotation ``tp`` (attrs classes, enums, Optional, sequences)."""
    if value is None or tp is Any or tp is None:
        return value
    origin = typing.get_origin(tp)
    if origin is Union:
        args = [a for a in typing.get_args(tp) if a is not type(None)]
        if len(args) == 1:
            return _structure_value(value, args[0])
        return value
    if origin in (list, tuple, abc.Sequence, typing.Sequence) and isinstance(value, (list, tuple)):
        args = typing.get_args(tp)
        if origin is tuple and args and args[-1] is not Ellipsis:
            return tuple(_structure_value(v, a) for v, a in zip(value, args))
        inner = args[0] if args else Any
        seq = [_structure_value(v, inner) for v in value]
        return tuple(seq) if origin is tuple else seq
    if isinstance(tp, type):
        if attrs.has(tp) and isinstance(value, abc.Mapping):
            return _structure_attrs(value, tp)
        if issubclass(tp, enum.Enum) and not isinstance(value, tp):
            return tp(value)
    return value


def _structure_attrs(mapping: abc.Mapping, cls: Type[_T]) -> _T:
    try:
        hints = typing.get_type_hints(cls)
    except Exception:
        hints = {}
    fields = {f.name.lstrip('_'): f for f in attrs.fields(cls) if f.init}
    extra = set(mapping) - set(fields)
    if extra:
        raise TypeError(f'{cls.__name__}: unexpected keys {sorted(extra)}')
    kwargs = {}
    for key, value in mapping.items():
        kwargs[key] = _structure_value(value, hints.get(fields[key].name, Any))
    return cls(**kwargs)


def dyn_structure(dyn_object: Any, target_cls: Type[_T], support_path_type: bool = False,
                  force_path_type: bool = False, support_none_type: bool = False) -> _T:
    if support_none_type and dyn_object is None:
        return target_cls()
    if support_path_type or force_path_type:
        is_path = _is_path(dyn_object)
        if force_path_type:
            assert is_path
        if is_path:
            with open(os.path.expandvars(os.path.expanduser(os.fspath(dyn_object)))) as fin:
                dyn_object = json.load(fin)
    try:
        if isinstance(dyn_object, target_cls):
            return dyn_object
    except TypeError:
        pass
    if isinstance(dyn_object, abc.Mapping):
        return _structure_attrs(dyn_object, target_cls)
    if isinstance(dyn_object, abc.Sequence) and not isinstance(dyn_object, (str, bytes)):
        return _structure_value(list(dyn_object), target_cls)
    raise NotImplementedError(f'cannot structure {type(dyn_object).__name__} into {target_cls}')


def rng_choice(rng: RandomGenerator, items: Sequence[_T], probs: Optional[Sequence[float]] = None) -> _T:
    return items[rng.choice(len(items), p=probs)]


def rng_choice_with_size(rng: RandomGenerator, items: Sequence[_T], size: int,
                         probs: Optional[Sequence[float]] = None, replace: bool = True) -> Sequence[_T]:
    indices = rng.choice(len(items), p=probs, size=size, replace=replace)
    return [items[idx] for idx in indices]


def rng_shuffle(rng: RandomGenerator, items: Sequence[_T]) -> Sequence[_T]:
    indices = list(range(len(items)))
    rng.shuffle(indices)
    return tuple(items[idx] for idx in indices)


# cv2 interpolation codes PageResizingStep draws from (reference utility/opt.py:125-148): the bit-exact variants of
# NEAREST / LINEAR, CUBIC and LANCZOS4; INTER_AREA joins the list only for a shrink
_CV_INTER_FLAGS = (6, 5, 2, 4)       # NEAREST_EXACT, LINEAR_EXACT, CUBIC, LANCZOS4
_CV_INTER_AREA = 3


def sample_cv_resize_interpolation(rng: RandomGenerator, include_cv_inter_area: bool = False) -> int:
    flags = _CV_INTER_FLAGS + ((_CV_INTER_AREA,) if include_cv_inter_area else ())
    return rng_choice(rng, flags)


def normalize_to_probs(weights: Sequence[float]):
    total = sum(weights)
    return [weight / total for weight in weights]


def get_config_class_snake_case_name(class_name: str):
    name = re.sub(r'(?<!^)(?=[A-Z])', '_', class_name).lower()
    return name[:-len('_config')] if name.endswith('_config') else name


def get_generic_classes(cls: Type[Any]):
    return typing.get_args(cls.__orig_bases__[0])  # type: ignore
