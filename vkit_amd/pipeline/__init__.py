"""The two pipeline steps that sit on the accelerated path (reference: vkit/pipeline/text_detection/
page_assembler.py and page_distortion.py).  Everything upstream of them -- layout, fonts, text rendering,
barcodes, seal impressions -- is outside the path; their outputs enter here as plain element containers."""
from .interface import PipelineStep, PipelineStepFactory  # noqa: F401
