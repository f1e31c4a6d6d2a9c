"""ctypes binding of the C ABI declared in include/i3rc_hip.h.

There is no CPU fallback: if the HIP library is missing or no GPU is present, calls fail loudly."""
import ctypes as C
import os

import numpy as np

from . import build as _build

fp = C.POINTER(C.c_float)
dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int32)
lp = C.POINTER(C.c_int64)
up = C.POINTER(C.c_uint32)

MAX_COMPONENTS = 255
MAX_DIRECTIONS = 255
NUM_COUNTERS = 16
COUNTER_NAMES = ["photons", "dropped", "cellSteps", "scatterings", "surfaceHits", "exitsTop", "roulette",
                 "shadowSteps", "tracerCalls", "rngDraws", "raysSkipped"]

# every symbol include/i3rc_hip.h declares (checked by tests/test_host_cpu.py::test_cabi_library_exports_every_declared_symbol)
SYMBOLS = [
    "i3rc_hip_create", "i3rc_hip_destroy", "i3rc_hip_last_error", "i3rc_hip_set_inverse_table",
    "i3rc_hip_set_forward_tables", "i3rc_hip_set_params", "i3rc_hip_set_surface", "i3rc_hip_set_directions",
    "i3rc_hip_get_tally_layout", "i3rc_hip_bind_tally_buffer", "i3rc_hip_set_stream", "i3rc_hip_use_own_stream", "i3rc_hip_zero_tallies",
    "i3rc_hip_launch_batch", "i3rc_hip_run_batches", "i3rc_hip_run_batches_moments", "i3rc_hip_get_moments_layout", "i3rc_hip_compute_batch", "i3rc_hip_expect_batches", "i3rc_hip_run_replay", "i3rc_hip_trace_rays", "i3rc_hip_synchronize",
    "i3rc_hip_fetch_tallies", "i3rc_hip_normalise", "i3rc_hip_last_kernel_ms", "i3rc_hip_kernel_ms_history", "i3rc_hip_set_tuning", "i3rc_hip_force_general_kernel", "i3rc_hip_select_kernel", "i3rc_hip_set_light_threshold", "i3rc_hip_set_launch_limit",
    "i3rc_hip_set_batch_fusion", "i3rc_hip_select_grid_place", "i3rc_hip_has_column_records", "i3rc_hip_column_records", "i3rc_hip_column_records_base", "i3rc_hip_lds_plan", "i3rc_hip_set_lds_tallies", "i3rc_hip_last_kernel_name", "i3rc_hip_timed_launch_count", "i3rc_hip_philox_blocks", "i3rc_hip_arith_check", "i3rc_hip_find_index", "i3rc_hip_surface_reflectance", "i3rc_hip_device_count", "i3rc_hip_version",
]


class Params(C.Structure):
    _fields_ = [
        ("surfaceAlbedo", C.c_float), ("useSurfaceBDRF", C.c_int32), ("useRayTracing", C.c_int32),
        ("useRussianRoulette", C.c_int32), ("useHybridPhaseFunsForIntenCalcs", C.c_int32),
        ("numOrdersOrigPhaseFunIntenCalcs", C.c_int32), ("useRussianRouletteForIntensity", C.c_int32),
        ("zetaMin", C.c_float), ("limitIntensityContributions", C.c_int32), ("maxIntensityContribution", C.c_float),
    ]


class Source(C.Structure):
    _fields_ = [("kind", C.c_int32), ("solarMu", C.c_float), ("solarAzimuth", C.c_float),
                ("x", fp), ("y", fp), ("z", fp), ("mu", fp), ("phi", fp)]


class TallyLayout(C.Structure):
    _fields_ = [("fluxUp", C.c_int64), ("fluxDown", C.c_int64), ("fluxAbsorbed", C.c_int64),
                ("volumeAbsorption", C.c_int64), ("intensityByComponent", C.c_int64), ("intensityExcess", C.c_int64),
                ("counters", C.c_int64), ("total", C.c_int64)]


class MomentsLayout(C.Structure):
    _fields_ = [(k, C.c_int64) for k in ("fluxUp", "fluxDown", "fluxAbsorbed", "volumeAbsorption", "intensity", "absorbedProfile",
                                         "meanFluxUp", "meanFluxDown", "meanFluxAbsorbed", "meanIntensity", "total")]


class I3RCError(RuntimeError):
    pass


_lib = None


def library_path():
    return _build.LIB


def load():
    """Load csrc/libi3rc_hip.so; raise (never fall back) if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise I3RCError(f"{path} not built: run __graft_entry__.build() (hipcc --offload-arch=gfx950); "
                        "the integrator has no CPU fallback")
    L = C.CDLL(path)
    H = C.c_void_p
    L.i3rc_hip_create.argtypes = [C.POINTER(H), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fp, fp, fp, fp, fp, fp, ip]
    L.i3rc_hip_destroy.argtypes = [H]
    L.i3rc_hip_last_error.argtypes = [H]
    L.i3rc_hip_last_error.restype = C.c_char_p
    L.i3rc_hip_set_inverse_table.argtypes = [H, C.c_int, C.c_int, C.c_int, fp]
    L.i3rc_hip_set_forward_tables.argtypes = [H, C.c_int, C.c_int, C.c_int, fp, fp]
    L.i3rc_hip_set_params.argtypes = [H, C.POINTER(Params)]
    L.i3rc_hip_set_surface.argtypes = [H, C.c_int, C.c_int, fp, fp, fp]
    L.i3rc_hip_set_directions.argtypes = [H, C.c_int, fp]
    L.i3rc_hip_get_tally_layout.argtypes = [H, C.POINTER(TallyLayout)]
    L.i3rc_hip_bind_tally_buffer.argtypes = [H, C.c_void_p, C.c_size_t]
    L.i3rc_hip_set_stream.argtypes = [H, C.c_void_p]
    L.i3rc_hip_use_own_stream.argtypes = [H]
    L.i3rc_hip_zero_tallies.argtypes = [H]
    L.i3rc_hip_launch_batch.argtypes = [H, C.c_uint32, C.c_uint32, C.c_int64, C.c_int64, C.POINTER(Source)]
    if hasattr(L, "i3rc_hip_run_batches"):   # (absent from older builds loaded for A/B timing)
        L.i3rc_hip_run_batches.argtypes = [H, C.c_uint32, C.c_uint32, C.c_int, C.c_int64, C.POINTER(Source), C.c_int, dp]
    if hasattr(L, "i3rc_hip_run_batches_moments"):
        L.i3rc_hip_run_batches_moments.argtypes = [H, C.c_uint32, C.c_uint32, C.c_int, C.c_int64, C.POINTER(Source), dp, dp, dp]
        L.i3rc_hip_get_moments_layout.argtypes = [H, C.POINTER(MomentsLayout)]
    if hasattr(L, "i3rc_hip_expect_batches"):
        L.i3rc_hip_expect_batches.argtypes = [H, C.c_uint32, C.c_uint32, C.c_int, C.c_int64, C.POINTER(Source), C.POINTER(C.c_int)]
    if hasattr(L, "i3rc_hip_compute_batch"):
        L.i3rc_hip_compute_batch.argtypes = [H, C.c_uint32, C.c_uint32, C.c_int64, C.POINTER(Source), C.c_int, dp]
    L.i3rc_hip_run_replay.argtypes = [H, C.c_int64, C.POINTER(Source), fp, C.c_int64, lp, ip, ip, fp, ip, ip]
    L.i3rc_hip_trace_rays.argtypes = [H, C.c_int64, fp, fp, ip, fp, fp, ip]
    L.i3rc_hip_synchronize.argtypes = [H]
    L.i3rc_hip_fetch_tallies.argtypes = [H, dp]
    L.i3rc_hip_normalise.argtypes = [H, dp, fp, fp, fp, fp, fp, fp]
    L.i3rc_hip_last_kernel_ms.argtypes = [H, fp]
    L.i3rc_hip_kernel_ms_history.argtypes = [H, C.c_int, fp]
    if hasattr(L, "i3rc_hip_last_kernel_name"):   # (absent from older builds of the library loaded for A/B timing: tools/lib_compare.py)
        L.i3rc_hip_last_kernel_name.argtypes = [H]
        L.i3rc_hip_timed_launch_count.argtypes = [H]
        L.i3rc_hip_timed_launch_count.restype = C.c_int64
        L.i3rc_hip_last_kernel_name.restype = C.c_char_p
    L.i3rc_hip_set_tuning.argtypes = [H, C.c_int, C.c_int]
    L.i3rc_hip_force_general_kernel.argtypes = [H, C.c_int]
    L.i3rc_hip_select_kernel.argtypes = [H, C.c_int]
    L.i3rc_hip_set_light_threshold.argtypes = [H, C.c_int]
    L.i3rc_hip_set_launch_limit.argtypes = [H, C.c_int64]
    if hasattr(L, "i3rc_hip_set_batch_fusion"):
        L.i3rc_hip_set_batch_fusion.argtypes = [H, C.c_int]
    if hasattr(L, "i3rc_hip_select_grid_place"):
        L.i3rc_hip_select_grid_place.argtypes = [H, C.c_int]
        L.i3rc_hip_has_column_records.argtypes = [H]
        L.i3rc_hip_column_records.argtypes = [C.c_int, C.c_int, C.c_int, fp, up]
    if hasattr(L, "i3rc_hip_column_records_base"):
        L.i3rc_hip_column_records_base.argtypes = [C.c_int, C.c_int, C.c_int, fp, up, fp]
    if hasattr(L, "i3rc_hip_lds_plan"):
        L.i3rc_hip_lds_plan.argtypes = [ip, ip]
        L.i3rc_hip_set_lds_tallies.argtypes = [H, C.c_int]
    L.i3rc_hip_philox_blocks.argtypes = [H, C.c_uint32, C.c_uint32, C.c_int64, C.c_int64, C.c_int, up, fp]
    L.i3rc_hip_arith_check.argtypes = [H, C.c_int64, fp, fp, lp, lp]
    if hasattr(L, "i3rc_hip_find_index"):
        L.i3rc_hip_find_index.argtypes = [H, C.c_int, fp, C.c_int64, fp, ip, ip]
        L.i3rc_hip_surface_reflectance.argtypes = [H, C.c_int64, fp, fp, fp]
    L.i3rc_hip_device_count.restype = C.c_int
    L.i3rc_hip_version.restype = C.c_char_p
    _lib = L
    return L


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def pf(a):
    return a.ctypes.data_as(fp)
