"""Vertex projectors map source lattice vertices to their distorted positions (reference:
grid_rendering/point_projector.py)."""
from typing import Iterable, Union

from vkit_amd.element import Point, PointList, PointTuple


class PointProjector:

    def project_point(self, src_point: Point) -> Point:
        raise NotImplementedError()

    def project_points(self, src_points: Union[PointList, PointTuple, Iterable[Point]]):
        return PointList(self.project_point(p) for p in src_points).to_point_tuple()
