"""MI355X-native UNet frame-interpolation forward (drop-in for the reference's model/unet.py
hot path).  See DESIGN.md / INTEGRATION.md.  The HIP extension (libfiunet_hip.so) is loaded
lazily on the first forward; nothing here falls back to CPU."""
from .unet import FrameInterpolationUNet, GraphedForward, UNet, count_parameters  # noqa: F401
from .inference import (  # noqa: F401
    FrameInterpolator, generate_multiple_intermediate_frames, interpolate_frames,
    interpolate_sequence, interpolate_sequence_host, load_model, postprocess_image, preprocess_image,
    sequence_pair_fn,
)
from .serving import InterpolationService  # noqa: F401
from . import evaluation, metrics, optical_flow, serving, synthetic, tiling, transport, video  # noqa: F401

__all__ = [
    "FrameInterpolationUNet", "GraphedForward", "UNet", "count_parameters", "FrameInterpolator",
    "generate_multiple_intermediate_frames", "interpolate_frames", "interpolate_sequence",
    "interpolate_sequence_host", "sequence_pair_fn", "transport", "optical_flow",
    "load_model", "postprocess_image", "preprocess_image", "evaluation", "metrics", "tiling", "video",
    "InterpolationService", "serving", "synthetic",
]
