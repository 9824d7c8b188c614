"""Shape protocol and set-operation modes of the element layer (reference: vkit/element/type.py:17-44)."""
from enum import Enum, unique
from typing import Tuple


class Shapable:
    """Anything with a (height, width)."""

    @property
    def height(self) -> int:
        raise NotImplementedError()

    @property
    def width(self) -> int:
        raise NotImplementedError()

    @property
    def area(self) -> int:
        return self.height * self.width

    @property
    def shape(self) -> Tuple[int, int]:
        return self.height, self.width


@unique
class ElementSetOperationMode(Enum):
    UNION = 'union'          # covered by one or more elements
    DISTINCT = 'distinct'    # covered by exactly one element
    INTERSECT = 'intersect'  # covered by more than one element
