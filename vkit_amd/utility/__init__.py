from .opt import (
    PathType,
    dyn_structure,
    get_config_class_snake_case_name,
    get_generic_classes,
    normalize_to_probs,
    rng_choice,
    rng_choice_with_size,
    rng_shuffle,
    sample_cv_resize_interpolation,
)
