"""dust_amd - MI355X-native (HIP, gfx950) backend for the SVGD-MPC inner loop of lubaroli/dust.

`dust_amd.backend.Context` is the object form of the C ABI (include/dust_amd.h); `dust_amd.controllers`,
`dust_amd.inference`, `dust_amd.kernels`, `dust_amd.models` mirror the reference's class names on top of it.
Nothing here falls back to the CPU: the HIP library must be built (`__graft_entry__.build()`) and a GPU present.
"""
from . import _lib  # noqa: F401
from .backend import Context, MpfContext, make_config  # noqa: F401

__all__ = ["Context", "MpfContext", "make_config"]
