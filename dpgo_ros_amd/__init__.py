"""dpgo_ros_amd -- MI355X-native RBCD hot path behind the DPGO::PGOAgent surface.

Only what the path needs: `csrc/` (HIP kernels + the C-ABI, built into libdpgo_hip.so) and
`capi` (ctypes mirror of the reference interface).  No CPU fallback exists.
"""
from . import capi  # noqa: F401
from .capi import Agent, Team, default_params  # noqa: F401
