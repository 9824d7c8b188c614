from .type import Shapable, ElementSetOperationMode
from .point import Point, PointList, PointTuple
from .box import Box
from .polygon import Polygon
from .soup import PointArray, PolygonSoup
from .mask import Mask
from .score_map import ScoreMap
from .image import Image, ImageMode, ImageSetItemConfig
