"""pytest configuration of the test-suite.

Markers
  gpu   needs a real MI355X (run by the driver with ``-m gpu`` on the GPU box);
        everything else runs on the CPU-only build container.

Fixtures
  cpu_api   the repository's ``mpc_interface`` API with ``tools.extend_matrices``
            temporarily served by the oracle, so that *host logic* (problem
            builders, index maps, plan compiler) can be exercised without a GPU.
            This patch exists only inside tests; the product has no CPU path.
  gpu_api   the unpatched API (every matrix comes from the HIP kernels).
"""
import os
import sys

import pytest

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for path in (os.path.join(_ROOT, "mpc-interface_amd"), _ROOT, os.path.dirname(__file__)):
    if path not in sys.path:
        sys.path.insert(0, path)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a HIP device (MI355X)")
    # compiled code objects of this test session go to a scratch directory of their own (the
    # library's default is ~/.cache/mpcasm), unless the caller chose one
    if "MPCASM_CACHE_DIR" not in os.environ:
        import tempfile

        os.environ["MPCASM_CACHE_DIR"] = tempfile.mkdtemp(prefix="mpcasm-test-cache-")


@pytest.fixture
def cpu_api(monkeypatch):
    from mpcasm import problems
    from oracle import qp_oracle
    import mpc_interface.tools as tools

    monkeypatch.setattr(tools, "extend_matrices", qp_oracle.extend_matrices)
    return problems.load_api("mpc_interface")


@pytest.fixture
def gpu_api():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from mpcasm import problems

    return problems.load_api("mpc_interface")
