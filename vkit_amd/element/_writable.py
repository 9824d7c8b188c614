"""Write-control shared by Image / Mask / ScoreMap: arrays are read-only outside ``writable_context``."""
from contextlib import ContextDecorator

import numpy as np


class WritableContext(ContextDecorator):

    def __init__(self, element, on_exit=None):
        super().__init__()
        self.element = element
        self.on_exit = on_exit

    def __enter__(self):
        mat = self.element.mat       # a device-resident element moves to the host: the caller is about to write numpy
        if self.element._mat is not mat:
            object.__setattr__(self.element, '_mat', np.array(mat))
            mat = self.element._mat
        try:
            mat.flags.writeable = True
        except ValueError:
            # a view of a read-only base: copy on write
            object.__setattr__(self.element, '_mat', np.array(mat))

    def __exit__(self, *exc):
        self.element._mat.flags.writeable = False
        if self.on_exit:
            self.on_exit()


class LazyMat:
    """``mat`` of Image / Mask / ScoreMap: the attrs field ``_mat`` (constructor argument ``mat``) holds a numpy array or a
    ``vkit_amd._native.DevArray``; ``.mat`` is always numpy (a device array is downloaded on first touch and cached
    read-only), ``.arr`` is whatever is held -- what the operators hand to the native wrappers, so that a chain of operators
    never leaves the device."""

    __slots__ = ()

    @property
    def mat(self) -> np.ndarray:
        held = self._mat
        if isinstance(held, np.ndarray):
            return held
        host = held.host()
        host.flags.writeable = False
        return host

    @property
    def arr(self):
        return self._mat

    @property
    def on_device(self) -> bool:
        return not isinstance(self._mat, np.ndarray)
