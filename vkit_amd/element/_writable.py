"""Write-control shared by Image / Mask / ScoreMap: arrays are read-only outside ``writable_context``."""
from contextlib import ContextDecorator

import numpy as np


class WritableContext(ContextDecorator):

    def __init__(self, element, on_exit=None):
        super().__init__()
        self.element = element
        self.on_exit = on_exit

    def __enter__(self):
        mat = self.element.mat
        try:
            mat.flags.writeable = True
        except ValueError:
            # a view of a read-only base: copy on write
            object.__setattr__(self.element, 'mat', np.array(mat))

    def __exit__(self, *exc):
        self.element.mat.flags.writeable = False
        if self.on_exit:
            self.on_exit()
