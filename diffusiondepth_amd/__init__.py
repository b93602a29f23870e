"""diffusiondepth_amd -- MI355X-native DDIM denoise hot path of DiffusionDepth.

Host side (this package, Python because the reference is Python) mirrors the reference's
head / pipeline / scheduler / depth-transform interfaces; all arithmetic on the hot path runs in
``libddepth_hip.so`` (hand-written HIP for gfx950, C ABI in ``include/ddepth.h``).  There is no CPU
fallback: importing works anywhere, running needs the built library and a GPU.
"""
from .scheduler import DDIMScheduler
from .backend import HipDenoiser, precision_id, library_path, load_library
from .modules import ScheduledCNNRefine, CNNDDIMPipiline, DeepDepthTransformWithUpsampling
from .head import (DDIMDepthEstimate_Res, DDIMDepthEstimate_Swin_ADD, DDIMDepthEstimate_Swin_ADDHAHI, DDIMDepthEstimate_MPVIT_ADDHAHI,
                   DDIMDepthEstimate_ResVis, DDIMDepthEstimate_Swin_ADDHAHIVis)
from .necks import HAHIHeteroNeck
from .nlspn import NLSPN
from .model import Diffusion_DCbase_Model

__all__ = [
    "DDIMScheduler", "HipDenoiser", "precision_id", "library_path", "load_library",
    "ScheduledCNNRefine", "CNNDDIMPipiline", "DeepDepthTransformWithUpsampling",
    "DDIMDepthEstimate_Res", "DDIMDepthEstimate_Swin_ADD", "DDIMDepthEstimate_Swin_ADDHAHI", "DDIMDepthEstimate_MPVIT_ADDHAHI",
    "DDIMDepthEstimate_ResVis", "DDIMDepthEstimate_Swin_ADDHAHIVis", "HAHIHeteroNeck", "NLSPN", "Diffusion_DCbase_Model",
]
__version__ = "0.1.0"
