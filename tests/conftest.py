import os
import sys

# The reference-graph variants of the parity tests (fused=False: the reference's own Conv2d / BatchNorm2d graph around
# the HIP spatial kernels) run their 1x1 convolutions on MIOpen, whose implicit-GEMM backward-data kernel
# (igemm_bwd_gtcx35_nhwc_fp32_*) reads past the end of small tensors: harmless inside PyTorch's caching allocator unless
# the tensor sits at the end of a mapped segment, then "Memory access fault by GPU" and an abort -- the intermittent
# abort of rounds 1 and 2, located with tests/guard/run_guarded.py.  That solver family is switched off for the test
# process (must happen before MIOpen is first used); the product path (fused=True) never calls MIOpen.
os.environ.setdefault("MIOPEN_DEBUG_CONV_IMPLICIT_GEMM", "0")

import numpy as np  # noqa: E402
import pytest  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): oracle/tpk_ref.py over oracle/libtpk_ref_cpu.so."""
    from oracle import tpk_ref
    tpk_ref.build()
    return tpk_ref


@pytest.fixture(scope="session")
def hip():
    """The product kernels; importing never falls back to anything else."""
    from torch_points3d_amd import torchpoints
    return torchpoints


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    out = {}
    for k in z.files:
        v = z[k]
        if v.dtype == np.int16:
            v = v.astype(np.int64)
        out[k] = torch.from_numpy(v) if v.dtype != np.float64 else v
    return out


@pytest.fixture(autouse=True)
def _guard_bands(request):
    """TP3D_TEST_CANARY=1: run every GPU test with guard bands around all device buffers, checked after each C-ABI
    call (tests/canary.py) -- the diagnostic mode for out-of-bounds writes."""
    if os.environ.get("TP3D_TEST_CANARY") and "gpu" in request.keywords and torch.cuda.is_available():
        from canary import Canary
        with Canary():
            yield
    else:
        yield
