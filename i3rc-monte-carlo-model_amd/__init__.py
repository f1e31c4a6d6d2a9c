"""MI355X-native photon-tracing integrator for the I3RC community Monte Carlo model (hot path only).

csrc/      hand-written HIP kernels for gfx950 + the C ABI (include/i3rc_hip.h)
fortran/   Fortran-95 module shell keeping the reference's API over the C ABI
host.py    Python mirror of the same interface (test / bench harness)
"""
from . import binding, build, host, multigpu, phasefunctions  # noqa: F401
from .binding import I3RCError  # noqa: F401
from .host import (Domain, Integrator, PhotonStream, RandomNumberSequence, SurfaceDescription,  # noqa: F401
                   new_Domain, new_Integrator, new_PhaseFunctionTable, new_PhotonStream,
                   new_RandomNumberSequence, new_SurfaceDescription, read_Domain)
from .phasefunctions import PhaseFunction, PhaseFunctionTable, henyey_greenstein  # noqa: F401
