"""Shape protocol and set-operation modes of the element layer (reference: vkit/element/type.py:17-44)."""
from enum import Enum, unique
from typing import Tuple


class Shapable:
    """Mixin for everything that occupies ``height`` x ``width`` pixels; subclasses provide the two extents."""

    height: int
    width: int

    @property
    def shape(self) -> Tuple[int, int]:
        return (self.height, self.width)

    @property
    def area(self) -> int:
        h, w = self.shape
        return h * w


@unique
class ElementSetOperationMode(Enum):
    """How overlapping elements combine: covered by any / by exactly one / by several of them."""
    UNION = 'union'
    DISTINCT = 'distinct'
    INTERSECT = 'intersect'
