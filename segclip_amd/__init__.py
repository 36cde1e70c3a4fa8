"""segclip_amd - MI355X (gfx950) native implementation of the SegCLIP image-text contrastive
forward/backward hot path behind the reference's module API (modules.modeling.SegCLIP,
modules.module_clip.CLIP).  See DESIGN.md / INTEGRATION.md."""
import os as _os

# The HIP runtime multiplexes streams onto 4 hardware queues by default and assigns them dynamically; with an RCCL
# communicator alive, the text-tower side stream ends up on the main stream's queue and the two towers run strictly one
# after the other (measured: 53.3 instead of 50.2 ms per step, tools/debug/hwq_env_probe.py).  Eight queues keep them
# apart.  The variable is read at the first HIP call, so setting it here (before any device work) is early enough; an
# explicit user setting wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# Kernel arguments in device memory: the command processor reads a dispatch's argument block from HBM instead of host memory
# over the fabric - the dependent chains of a step (~650 launches) start each kernel sooner.  Measured on MI355X, same box,
# unset / 0 / 1: see DESIGN 4.5 (38.2 vs 39.0 ms per step).  Also read at the first HIP call; an explicit user setting wins.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

from . import config  # noqa: F401,E402
from .config import set_compute_dtype, set_cross_mode, noise_injection  # noqa: F401,E402

__version__ = "0.1.0"
