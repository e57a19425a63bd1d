"""tfra_amd — MI355X-native backend for TFRA's dynamic-embedding hot path.

    import tfra_amd.dynamic_embedding as de     # mirrors `tfra.dynamic_embedding`

Everything runs on hand-written HIP kernels (gfx950) behind the C ABI in
include/tfra_mi355x.h; PyTorch is used only for device memory, streams and
torch.distributed (RCCL).  There is no CPU fallback.
"""
from . import _capi  # noqa: F401

__version__ = "0.1.0"
