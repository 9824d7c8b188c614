from .page_assembler import *  # noqa: F401,F403
from .page_assembler import page_assembler_step_factory  # noqa: F401
from .page_distortion import (  # noqa: F401
    ElementFlattener,
    PageDistortionStep,
    PageDistortionStepConfig,
    PageDistortionStepInput,
    PageDistortionStepOutput,
    page_distortion_step_factory,
)
from .page_resizing import (  # noqa: F401
    PageResizingStep,
    PageResizingStepConfig,
    PageResizingStepInput,
    PageResizingStepOutput,
    page_resizing_step_factory,
)
