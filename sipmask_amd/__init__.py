"""sipmask_amd -- MI355X (gfx950) native SipMask hot path.

Python host on PyTorch-ROCm (device memory, streams, torch.distributed) over a C-ABI HIP
library (include/sipmask_hip.h -> sipmask_amd/libsipmask_hip.so).  The package mirrors the
reference's plugin seam (HEADS/DETECTORS registries, SipMaskHead kwargs, state_dict keys,
mmdet.ops entry points) for this one path only; see DESIGN.md.
"""
__version__ = "0.1.0"
