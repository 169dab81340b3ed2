"""fluid_hip — Python host binding of the MI355X stable-fluids hot path (libfluid_hip.so).

`FluidSim` mirrors the reference's simulation globals (config / initFramebuffers / splat /
multipleSplats / step / update / framebufferToTexture); `StripeSim` (fluid_hip.stripes) runs one
row stripe per GPU with ghost-row exchange over torch.distributed (RCCL).
"""
from ._abi import FluidError, build, device_count, lib  # noqa: F401
from .sim import Canvas, DEFAULT_CONFIG, FluidSim, HSVtoRGB, getResolution, mulberry32  # noqa: F401
