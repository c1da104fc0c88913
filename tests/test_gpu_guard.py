"""Out-of-bounds audit: the GPU parity tests again in a process whose every device allocation ends at the end of its own
mapping (tests/guard/guard_alloc.cpp as PyTorch's allocator) and whose kernel launches are serialised.  An access past the
end of any tensor -- by a kernel of this library or of a vendor library -- then faults at the launch that makes it instead
of once in a dozen runs.  This is how the rounds-1/2 abort was located (MIOpen's implicit-GEMM backward-data kernel in the
reference-graph variants; see tests/conftest.py) and how ATen's indexing_backward_kernel_small_stride was found."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.gpu
def test_kernel_parity_tests_under_the_guard_allocator():
    # the kernel-level and fused-path tests (the graph-capturing ones cannot run on a pluggable allocator)
    sel = ["tests/test_gpu_parity.py", "tests/test_gpu_fused.py", "tests/test_gpu_kpconv.py", "-m", "gpu", "-k",
           "not trajectory and not graph and not c3_full_batch"]
    env = dict(os.environ)
    env.setdefault("MIOPEN_DEBUG_CONV_IMPLICIT_GEMM", "0")
    out = subprocess.run([sys.executable, os.path.join(HERE, "guard", "run_guarded.py")] + sel, env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    tail = out.stdout[-3000:]
    assert "Memory access fault" not in out.stdout, tail
    assert out.returncode == 0, tail
