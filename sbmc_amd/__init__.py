"""sbmc_amd -- MI355X-native hot path of Sample-Based Monte Carlo denoising.

Hand-written gfx950 HIP kernels behind a C ABI (include/sbmc_hip.h) for the
per-sample kernel-splatting operators, exposed through the reference's own
``sbmc.functions`` / ``sbmc.modules`` / ``sbmc.models`` API.
"""
from . import miopen_db as _miopen_db

_miopen_db.install()        # before the first convolution: MIOpen reads its user find-db location lazily

from .models import Multisteps, KPCN  # noqa: F401,E402
from . import functions, modules, models, losses, binio, denoise, interfaces  # noqa: F401,E402

__version__ = "0.1.0"
