"""Build the gfx950 shared library (in-tree, so it travels to the GPU box with the snapshot)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libi3rc_hip.so")
SOURCES = ["i3rc_hip.hip", "kernels.hpp", "tracer.hpp", "philox.hpp"]
HEADER = os.path.join(os.path.dirname(HERE), "include", "i3rc_hip.h")

# -ffp-contract=off: float32 results must follow the reference's operator order (no FMA contraction);
# see DESIGN.md "float32 semantics".
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off"]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


COMM_LIB = os.path.join(CSRC, "libi3rc_comm.so")
COMM_HEADER = os.path.join(os.path.dirname(HERE), "include", "i3rc_comm.h")


def build_comm(force=False):
    """Process layer (RCCL all-reduce / shared-memory test backend): csrc/libi3rc_comm.so, host code only."""
    src = os.path.join(CSRC, "i3rc_comm.cpp")
    if not force and os.path.exists(COMM_LIB) and os.path.getmtime(COMM_LIB) > max(os.path.getmtime(src), os.path.getmtime(COMM_HEADER)):
        return COMM_LIB
    subprocess.check_call([hipcc(), "-O2", "-std=c++17", "-fPIC", "-shared", "-o", COMM_LIB, src,
                           "-L/opt/rocm/lib", "-lrccl", "-lamdhip64", "-lrt", "-lpthread"], cwd=CSRC)
    return COMM_LIB


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [HEADER]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 ... -> csrc/libi3rc_hip.so (cross-compiles without a GPU)."""
    build_comm(force)
    if not force and not needs_build():
        return LIB
    cmd = [hipcc()] + HIPCC_FLAGS + ["-o", LIB, os.path.join(CSRC, "i3rc_hip.hip")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


def variant_lib(name):
    return os.path.join(CSRC, f"libi3rc_hip_var_{name}.so")


def build_variant(name, flags, force=False):
    """A side build of the library with extra -D flags (measurement knobs, tools/variant_bench.py; the nested-order build of
    tests/test_gpu_features.py): csrc/libi3rc_hip_var_<name>.so, rebuilt when a source is newer."""
    lib = variant_lib(name)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [HEADER]
    if not force and os.path.exists(lib) and all(os.path.getmtime(d) <= os.path.getmtime(lib) for d in deps):
        return lib
    subprocess.check_call([hipcc()] + HIPCC_FLAGS + list(flags) + ["-o", lib, os.path.join(CSRC, "i3rc_hip.hip")], cwd=CSRC)
    return lib


# the measurement build in which the general kernels keep the reference's nested local estimate (kernels.hpp, I3RC_NESTED_BUILD)
NESTED_FLAGS = ["-DI3RC_NESTED_BUILD"]


if __name__ == "__main__":
    print(build(force=True, verbose=True))
