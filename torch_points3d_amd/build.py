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
    "-fvisibility=hidden",
    "-std=c++17",
]
OBJ_DIR = os.path.join(CSRC, "_obj")


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
    if _linked_sources() != [os.path.basename(x) for x in sources()]:
        return True  # a source file was added or removed since the last link
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


MANIFEST = os.path.join(OBJ_DIR, "linked.txt")


def _linked_sources():
    try:
        with open(MANIFEST) as f:
            return f.read().split()
    except OSError:
        return None


def _headers():
    return glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))


def build_library(force=False, verbose=False):
    """Compile every HIP source into torch_points3d_amd/libtp3d_hip.so. Returns the path.

    One object file per source (csrc/_obj/, rebuilt only when the source or a header is newer), compiled in
    parallel, then one link: a one-file change costs one hipcc run instead of the whole library."""
    if not force and not is_stale():
        return LIB_PATH
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = _hipcc()
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    hdr_time = max([os.path.getmtime(h) for h in _headers()] + [os.path.getmtime(os.path.abspath(__file__))])
    jobs, objs = [], []
    for src in sources():
        obj = os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_time):
            jobs.append([hipcc] + HIPCC_FLAGS + inc + ["-c", src, "-o", obj])
    workers = max(1, min(len(jobs), (os.cpu_count() or 2)))
    running = []
    failed = None
    for cmd in jobs:
        if verbose:
            print(" ".join(cmd))
        while len(running) >= workers:
            failed = _reap(running) or failed
        running.append((cmd, subprocess.Popen(cmd)))
    while running:
        failed = _reap(running) or failed
    if failed:
        raise subprocess.CalledProcessError(1, failed)
    link = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB_PATH] + objs
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)
    with open(MANIFEST, "w") as f:
        f.write("\n".join(os.path.basename(x) for x in sources()) + "\n")
    return LIB_PATH


def _reap(running):
    """Wait for the oldest compile job; returns its command line if it failed."""
    cmd, proc = running.pop(0)
    return cmd if proc.wait() != 0 else None


CPU_SRC = os.path.join(PKG_DIR, "csrc_cpu", "points_cpu.c")
CPU_LIB_PATH = os.path.join(PKG_DIR, "libtp3d_cpu.so")


def build_cpu_library(force=False):
    """gcc-compile the host-side searches (include/tp3d_cpu.h) into torch_points3d_amd/libtp3d_cpu.so.  No HIP, no
    OpenMP: the library must be loadable and usable inside forked DataLoader workers."""
    deps = [CPU_SRC, os.path.join(ROOT, "include", "tp3d_cpu.h")]
    if not force and os.path.exists(CPU_LIB_PATH) and all(os.path.getmtime(d) <= os.path.getmtime(CPU_LIB_PATH) for d in deps):
        return CPU_LIB_PATH
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        raise RuntimeError("no C compiler found: libtp3d_cpu.so cannot be built")
    subprocess.check_call([cc, "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-Wall", "-I" + os.path.join(ROOT, "include"),
                           CPU_SRC, "-o", CPU_LIB_PATH, "-lpthread", "-lm"])
    return CPU_LIB_PATH


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
    print(build_cpu_library(force=True))
