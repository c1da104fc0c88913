"""Runs one pytest selection in a fresh process whose EVERY device allocation ends at the end of its own mapping
(tests/guard/guard_alloc.cpp) with kernel launches serialised, so that an access past the end of any tensor faults at the
launch that makes it and the Python traceback of the abort names the call.  Usage (on the GPU box):

    python tests/guard/run_guarded.py tests/test_gpu_parity.py -k test_model_gradients_elementwise
    python tests/guard/run_guarded.py --script bench.py --no-graph --steps 2 --warmup 1 --no-cpu-baseline
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def build():
    so = os.path.join("/tmp", "tp3d_guard_alloc.so")
    src = os.path.join(HERE, "guard_alloc.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-fPIC", "-shared", src, "-o", so], check=True)
    return so


CHILD = r"""
import sys, faulthandler, runpy
faulthandler.enable()
import torch
alloc = torch.cuda.memory.CUDAPluggableAllocator(sys.argv[1], "guard_malloc", "guard_free")
torch.cuda.memory.change_current_allocator(alloc)
if len(sys.argv) > 2 and sys.argv[2] == "--script":  # any script instead of a pytest selection
    script = sys.argv[3]
    sys.argv = [script] + sys.argv[4:]
    runpy.run_path(script, run_name="__main__")
    sys.exit(0)
import pytest
sys.exit(pytest.main(["-q", "-p", "no:cacheprovider"] + sys.argv[2:]))
"""


def main():
    so = build()
    env = dict(os.environ)
    env["AMD_SERIALIZE_KERNEL"] = "3"   # wait before and after every kernel: the fault belongs to the launch in flight
    env["HIP_LAUNCH_BLOCKING"] = "1"
    env["PYTHONPATH"] = ROOT + os.pathsep + os.path.join(ROOT, "tests") + os.pathsep + env.get("PYTHONPATH", "")
    return subprocess.call([sys.executable, "-c", CHILD, so] + sys.argv[1:], env=env, cwd=ROOT)


if __name__ == "__main__":
    sys.exit(main())
