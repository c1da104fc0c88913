"""Builds libtp3d_hip.so in-tree with hipcc for gfx950 (no cmake, no torch C++ ABI).

The library is a plain C-ABI shared object (include/tp3d_hip.h), so it does not depend on the
PyTorch build it is used with; at run time it resolves the HIP runtime PyTorch has already loaded.
"""
import glob
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libtp3d_hip.so")
ARCH = "gfx950"

# -ffp-contract=off: squared distances must not be fused into v_fma (bit-exact indices vs the oracle).
HIPCC_FLAGS = [
    "--offload-arch=" + ARCH,
    "-O3",
    "-ffp-contract=off",
    "-fno-honor-nans",  # fminf/fmaxf -> bare v_min/v_max (no canonicalising v_max x,x); inputs are finite clouds
    "-fPIC",
    "-shared",
    "-fvisibility=hidden",
    "-std=c++17",
]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libtp3d_hip.so cannot be built")
    return exe


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False):
    """Compile every HIP source into torch_points3d_amd/libtp3d_hip.so. Returns the path."""
    if not force and not is_stale():
        return LIB_PATH
    cmd = [_hipcc()] + HIPCC_FLAGS + ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-o", LIB_PATH] + sources()
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
