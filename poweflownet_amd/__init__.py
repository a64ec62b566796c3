"""poweflownet_amd -- MI355X (gfx950) implementation of PowerFlowNet's message-passing hot path
(`MaskEmbdMultiMPN`, reference networks/MPN.py) behind the reference's own nn.Module / PyG-Data surface.

Layout: `csrc/` HIP kernels + C ABI (include/pfn_hip.h), `networks/MPN.py` the host-side mirror of the
reference classes, `data.py` / `synth.py` the mini PyG data surface and synthetic grids, `dp.py` the
one-collective data-parallel step, `utils/` the train/eval loop counterparts."""
from .data import Batch, Data, DataLoader  # noqa: F401

__all__ = ["Batch", "Data", "DataLoader"]
