"""Float32 score / probability plane (reference: vkit/element/score_map.py).

On the hot path a ScoreMap is (a) an element that grid / affine distortions remap with float32 bilinear
weights and (b) the per-pixel alpha of a text-line layer in the page composite.
"""
from typing import Optional, Tuple

import attrs
import numpy as np

from vkit_amd import _native
from ._writable import LazyMat, WritableContext
from .opt import _deferred_stack, generate_shape_and_resized_shape
from .type import ElementSetOperationMode, Shapable


@attrs.define(frozen=True, eq=False)
class ScoreMap(LazyMat, Shapable):
    _mat: np.ndarray = attrs.field(alias='mat')
    box: Optional['Box'] = None
    is_prob: bool = True

    def __attrs_post_init__(self):
        if self._mat.ndim != 2:
            raise RuntimeError('ndim should == 2.')
        if self.box and self.shape != self.box.shape:
            raise RuntimeError('self.shape != box.shape.')
        if self._mat.dtype != np.float32:
            raise RuntimeError('mat.dtype != np.float32')
        if isinstance(self._mat, np.ndarray):
            self._mat.flags.writeable = False
        if self.is_prob and self._mat.size:
            if self.mat.min() < 0.0 or self.mat.max() > 1.0:
                raise RuntimeError('score not in range [0.0, 1.0]')

    @classmethod
    def from_shape(cls, shape: Tuple[int, int], value: float = 0.0, is_prob: bool = True):
        height, width = shape
        if is_prob:
            assert 0.0 <= value <= 1.0
        return cls(mat=np.full((height, width), fill_value=value, dtype=np.float32), is_prob=is_prob)

    @classmethod
    def from_shapable(cls, shapable: Shapable, value: float = 0.0, is_prob: bool = True):
        return cls.from_shape(shapable.shape, value=value, is_prob=is_prob)

    @property
    def height(self):
        return self._mat.shape[0]

    @property
    def width(self):
        return self._mat.shape[1]

    @property
    def equivalent_box(self):
        return self.box or Box.from_shapable(self)

    @property
    def writable_context(self):
        return WritableContext(self)

    def copy(self):
        return attrs.evolve(self, mat=self.mat.copy())

    def assign_mat(self, mat: np.ndarray):
        with self.writable_context:
            object.__setattr__(self, '_mat', mat)

    def to_shifted_score_map(self, offset_y: int = 0, offset_x: int = 0):
        assert self.box
        return attrs.evolve(self, box=self.box.to_shifted_box(offset_y=offset_y, offset_x=offset_x))

    def to_resized_score_map(self, resized_height: Optional[int] = None, resized_width: Optional[int] = None,
                             cv_resize_interpolation: int = 2):
        """cv.resize (any cv2 code 0..6), clipped back to [0, 1] for probability maps (reference score_map.py:616-637)."""
        assert not self.box
        if cv_resize_interpolation not in range(7):
            raise ValueError(f'unknown cv2 interpolation code {cv_resize_interpolation}')
        _, _, resized_height, resized_width = generate_shape_and_resized_shape(
            shapable_or_shape=self.shape, resized_height=resized_height, resized_width=resized_width)
        mat = _native.resize(self.arr, (resized_height, resized_width), cv_resize_interpolation)
        if self.is_prob:
            mat = np.clip(_native.host_array(mat), 0.0, 1.0)
        return attrs.evolve(self, mat=mat)

    def to_cropped_score_map(self, up=None, down=None, left=None, right=None):
        assert not self.box
        up = up or 0
        down = down or self.height - 1
        left = left or 0
        right = right or self.width - 1
        return attrs.evolve(self, mat=self.mat[up:down + 1, left:right + 1])

    def to_box_attached(self, box: 'Box'):
        assert self.height == box.height and self.width == box.width
        return attrs.evolve(self, box=box)

    def to_box_detached(self):
        assert self.box
        return attrs.evolve(self, box=None)

    def to_mask(self, threshold: float = 0.0):
        return Mask(mat=(self.mat > threshold).astype(np.uint8), box=self.box)

    # self is both the selection (alpha > 0) and the weight
    def fill_np_array(self, mat: np.ndarray, value, keep_max_value: bool = False, keep_min_value: bool = False):
        self.equivalent_box.fill_np_array(mat, value, alpha=self, keep_max_value=keep_max_value,
                                          keep_min_value=keep_min_value)

    def fill_image(self, image: 'Image', value):
        # The text lines of a page: a box-attached probability map and a constant colour recorded by an open deferred
        # composite on a uint8 image.  Same layer as the generic path below records (box geometry, the map as alpha plane,
        # no selection plane), without its five calls per layer.
        box, target, alpha = self.box, image._mat, self._mat
        if (box is not None and image.box is None and type(value) is tuple and self.is_prob and isinstance(alpha, np.ndarray)
                and target.dtype == np.uint8 and target.ndim == 3 and len(value) == target.shape[2]):
            stack = _deferred_stack()
            if (stack and stack[-1].base is target and 0 <= box.up <= box.down < target.shape[0]
                    and 0 <= box.left <= box.right < target.shape[1]):
                stack[-1].layers.append(_native.make_layer((box.up, box.left, alpha.shape[0], alpha.shape[1]), len(value), value,
                                                           alpha=alpha))
                return
        self.equivalent_box.fill_image(image, value, alpha=self)


def generate_fill_by_score_maps_mask(shape, score_maps, mode: ElementSetOperationMode):
    if mode == ElementSetOperationMode.UNION:
        return None
    raise NotImplementedError('non-UNION score-map set operations are outside the accelerated path')


from .box import Box  # noqa: E402
from .mask import Mask  # noqa: E402
from .image import Image  # noqa: E402
