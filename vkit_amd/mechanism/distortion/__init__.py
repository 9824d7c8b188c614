"""``vkit_amd.mechanism.distortion``: the distortions of the accelerated path under the names the reference
exports (vkit/mechanism/distortion/__init__.py:17-108)."""
from .interface import (
    Distortion, DistortionConfig, DistortionInternals, DistortionNopState, DistortionResult, DistortionState,
)

# photometric
from .photometric.opt import OutOfBoundBehavior
from .photometric.color import (
    MeanShiftConfig, mean_shift, ColorShiftConfig, color_shift, ComplementConfig, complement, PosterizationConfig,
    posterization, ChannelPermutationConfig, channel_permutation, BrightnessShiftConfig, brightness_shift,
    StdShiftConfig, std_shift, ColorBalanceConfig, color_balance, BoundaryEqualizationConfig,
    boundary_equalization, HistogramEqualizationConfig, histogram_equalization,
)
from .photometric.blur import (
    GaussianBlurConfig, gaussian_blur, DefocusBlurConfig, defocus_blur, MotionBlurConfig, motion_blur,
    GlassBlurConfig, glass_blur, ZoomInBlurConfig, zoom_in_blur,
)
from .photometric.noise import (
    GaussionNoiseConfig, gaussion_noise, ImpulseNoiseConfig, impulse_noise, SpeckleNoiseConfig, speckle_noise,
    PoissonNoiseConfig, poisson_noise,
)
from .photometric.effect import FogConfig, fog, JpegQualityConfig, jpeg_quality, PixelationConfig, pixelation
from .photometric.streak import (
    EllipseStreakConfig, ellipse_streak, LineStreakConfig, line_streak, RectangleStreakConfig, rectangle_streak,
)

# geometric
from .geometric.affine import (
    ShearHoriConfig, shear_hori, ShearVertConfig, shear_vert, RotateConfig, rotate, SkewHoriConfig, skew_hori,
    SkewVertConfig, skew_vert,
)
from .geometric.mls import SimilarityMlsConfig, similarity_mls
from .geometric.camera import (
    CameraModelConfig, CameraPlaneOnlyConfig, camera_plane_only, CameraCubicCurveConfig, camera_cubic_curve,
    CameraPlaneLineFoldConfig, camera_plane_line_fold, CameraPlaneLineCurveConfig, camera_plane_line_curve,
)
