"""Config generators of the camera family (reference: distortion_policy/geometric/camera.py)."""
from typing import Tuple

import attrs
import numpy as np
from numpy.random import Generator as RandomGenerator

from vkit_amd.mechanism import distortion
from ..opt import generate_grid_size, sample_float, sample_int
from ..type import DistortionConfigGenerator, DistortionPolicyFactory


def sample_camera_model_config(level: int, level_1_max: int, rotation_theta_max: int, vec_z_max: float,
                               rng: RandomGenerator):
    rotation_theta = sample_int(level, 1, rotation_theta_max, 0.5, rng)
    theta_xy = rng.uniform(0, 2 * np.pi)
    vec_x, vec_y, vec_z = np.cos(theta_xy), np.sin(theta_xy), 0.0
    if level > level_1_max:
        # tilt the axis out of the page plane (normalised later); vec_z == 1 would be a plain rotation
        vec_z = rng.uniform(0, vec_z_max)
        vec_x = (1 - vec_z) * vec_x
        vec_y = (1 - vec_z) * vec_y
    return distortion.CameraModelConfig(rotation_unit_vec=[vec_x, vec_y, vec_z], rotation_theta=rotation_theta)


def _camera_and_grid(generator, shape, rng):
    cfg = generator.config
    camera_model_config = sample_camera_model_config(generator.level, cfg.level_1_max, cfg.rotation_theta_max,
                                                     cfg.vec_z_max, rng)
    return camera_model_config, generate_grid_size(cfg.grid_size_min, cfg.grid_size_ratio, shape)


@attrs.define
class CameraPlaneOnlyConfigGeneratorConfig:
    level_1_max: int = 5
    rotation_theta_max: int = 17
    vec_z_max: float = 0.5
    grid_size_min: int = 15
    grid_size_ratio: float = 0.01


class CameraPlaneOnlyConfigGenerator(
        DistortionConfigGenerator[CameraPlaneOnlyConfigGeneratorConfig, distortion.CameraPlaneOnlyConfig]):

    def __call__(self, shape: Tuple[int, int], rng: RandomGenerator):
        camera_model_config, grid_size = _camera_and_grid(self, shape, rng)
        return distortion.CameraPlaneOnlyConfig(camera_model_config=camera_model_config, grid_size=grid_size)


camera_plane_only_policy_factory = DistortionPolicyFactory(distortion.camera_plane_only, CameraPlaneOnlyConfigGenerator)


@attrs.define
class CameraCubicCurveConfigGeneratorConfig:
    curve_slope_range_min: float = 10.0
    curve_slope_range_max: float = 90.0
    curve_slope_max: float = 45
    level_1_max: int = 5
    rotation_theta_max: int = 17
    vec_z_max: float = 0.5
    grid_size_min: int = 15
    grid_size_ratio: float = 0.01


class CameraCubicCurveConfigGenerator(
        DistortionConfigGenerator[CameraCubicCurveConfigGeneratorConfig, distortion.CameraCubicCurveConfig]):

    def __call__(self, shape: Tuple[int, int], rng: RandomGenerator):
        cfg = self.config
        slope_range = sample_float(self.level, cfg.curve_slope_range_min, cfg.curve_slope_range_max, None, rng)
        alpha_ratio = rng.uniform()
        curve_alpha = slope_range * alpha_ratio
        curve_beta = slope_range - curve_alpha
        curve_alpha = min(cfg.curve_slope_max, curve_alpha)
        curve_beta = min(cfg.curve_slope_max, curve_beta)
        if rng.random() < 0.5:
            curve_alpha *= -1
        if rng.random() < 0.5:
            curve_beta *= -1
        curve_direction = rng.uniform(0, 180)
        camera_model_config, grid_size = _camera_and_grid(self, shape, rng)
        return distortion.CameraCubicCurveConfig(
            curve_alpha=curve_alpha,
            curve_beta=curve_beta,
            curve_direction=curve_direction,
            curve_scale=1.0,
            camera_model_config=camera_model_config,
            grid_size=grid_size,
        )


camera_cubic_curve_policy_factory = DistortionPolicyFactory(distortion.camera_cubic_curve,
                                                            CameraCubicCurveConfigGenerator)


def _sample_line_perturbation(shape, rng: RandomGenerator):
    """A random anchor point, a direction in [0, 180) and a +-z push of a quarter of the long side."""
    height, width = shape
    point = (rng.integers(0, width), rng.integers(0, height))
    direction = rng.uniform(0, 180)
    perturb_z = max(shape) / 4
    if rng.random() < 0.5:
        perturb_z *= -1.0
    return point, direction, (0.0, 0.0, perturb_z)


@attrs.define
class CameraPlaneLineFoldConfigGeneratorConfig:
    fold_alpha_min: float = 0.1
    fold_alpha_max: float = 1.25
    level_1_max: int = 5
    rotation_theta_max: int = 17
    vec_z_max: float = 0.5
    grid_size_min: int = 15
    grid_size_ratio: float = 0.01


class CameraPlaneLineFoldConfigGenerator(
        DistortionConfigGenerator[CameraPlaneLineFoldConfigGeneratorConfig, distortion.CameraPlaneLineFoldConfig]):

    def __call__(self, shape: Tuple[int, int], rng: RandomGenerator):
        point, direction, perturb_vec = _sample_line_perturbation(shape, rng)
        fold_alpha = sample_float(self.level, self.config.fold_alpha_min, self.config.fold_alpha_max, None, rng,
                                  inverse_level=True)
        camera_model_config, grid_size = _camera_and_grid(self, shape, rng)
        return distortion.CameraPlaneLineFoldConfig(
            fold_point=point,
            fold_direction=direction,
            fold_perturb_vec=perturb_vec,
            fold_alpha=fold_alpha,
            camera_model_config=camera_model_config,
            grid_size=grid_size,
        )


camera_plane_line_fold_policy_factory = DistortionPolicyFactory(distortion.camera_plane_line_fold,
                                                                CameraPlaneLineFoldConfigGenerator)


@attrs.define
class CameraPlaneLineCurveConfigGeneratorConfig:
    curve_alpha_min: float = 1.0
    curve_alpha_max: float = 2.0
    level_1_max: int = 5
    rotation_theta_max: int = 17
    vec_z_max: float = 0.5
    grid_size_min: int = 15
    grid_size_ratio: float = 0.01


class CameraPlaneLineCurveConfigGenerator(
        DistortionConfigGenerator[CameraPlaneLineCurveConfigGeneratorConfig, distortion.CameraPlaneLineCurveConfig]):

    def __call__(self, shape: Tuple[int, int], rng: RandomGenerator):
        point, direction, perturb_vec = _sample_line_perturbation(shape, rng)
        curve_alpha = sample_float(self.level, self.config.curve_alpha_min, self.config.curve_alpha_max, None, rng,
                                   inverse_level=True)
        camera_model_config, grid_size = _camera_and_grid(self, shape, rng)
        return distortion.CameraPlaneLineCurveConfig(
            curve_point=point,
            curve_direction=direction,
            curve_perturb_vec=perturb_vec,
            curve_alpha=curve_alpha,
            camera_model_config=camera_model_config,
            grid_size=grid_size,
        )


camera_plane_line_curve_policy_factory = DistortionPolicyFactory(distortion.camera_plane_line_curve,
                                                                 CameraPlaneLineCurveConfigGenerator)
