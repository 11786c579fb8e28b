import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# torch before the solver library: both then share torch's bundled ROCm runtime (one HIP + one HSA runtime in the process).  The binding
# (capi._bind_torch_runtime) makes the other order safe as well; tests/test_abi_cpu.py checks both orders in fresh processes.
import torch  # noqa: E402,F401
import _bootstrap  # noqa: E402

_bootstrap.load()
from chrono_wisc_amd import capi  # noqa: E402

ORACLE_LIB = os.path.join(ROOT, "oracle", "libsph_oracle.so")

# The product's default arithmetic is SPH_ARITH_FAST (hardware reciprocals, FMA, packed FP32).  The parity tests written against
# the oracle's exact bits -- equal variable step sizes, bit-identical slabs, golden vectors -- pin SPH_ARITH_EXACT for every
# context they create (sph_create reads CHSPH_ARITH); tests/test_fast_gpu.py asks for the fast arithmetic explicitly and holds
# one evaluation from an identical state to 1e-5 of each field's inf-norm and 10 steps to 1e-4 (include/chsph_mi.h).
os.environ.setdefault("CHSPH_ARITH", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _build_oracle():
    import subprocess
    if not os.path.exists(ORACLE_LIB):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def oracle():
    """CPU restatement of the reference algorithm (test infrastructure only)."""
    _build_oracle()
    return capi.CApi(ORACLE_LIB, "ref_")


@pytest.fixture(scope="session")
def gpu():
    """The product library.  Fails loudly (no skip, no fallback) when the extension is missing."""
    return capi.CApi()
