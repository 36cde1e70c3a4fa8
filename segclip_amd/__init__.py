"""segclip_amd - MI355X (gfx950) native implementation of the SegCLIP image-text contrastive
forward/backward hot path behind the reference's module API (modules.modeling.SegCLIP,
modules.module_clip.CLIP).  See DESIGN.md / INTEGRATION.md."""
from . import config  # noqa: F401
from .config import set_compute_dtype, set_cross_mode, noise_injection  # noqa: F401

__version__ = "0.1.0"
