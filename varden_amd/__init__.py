"""varden_amd -- MI355X-native implementation of VARDEN's per-timestep hot path.

The compute path is hand-written HIP for gfx950 behind the C-ABI of include/varden_amd.h
(varden_amd/csrc/libvarden_amd.so).  This package is the Python host-side mirror of the reference's
interface for that path (used by the tests and bench.py); varden_amd/fortran/ holds the Fortran
ISO_C_BINDING mirror.  There is no CPU fallback: without the built library and a GPU every entry
point raises.
"""
from . import capi  # noqa: F401
from .capi import VardenError, default_params  # noqa: F401
