"""pytest configuration: the `gpu` marker and import paths.

- product modules (drop-in names) live in optical-flow-fpga_amd/python
- the CPU oracle (test infrastructure) lives in oracle/
"""
import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
PRODUCT = ROOT / "optical-flow-fpga_amd" / "python"
ORACLE = ROOT / "oracle"
GOLDEN = ROOT / "tests" / "golden"

for p in (str(PRODUCT), str(ORACLE), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

os.environ.setdefault("OFLK_QUIET", "1")

# torch first (bench.py does the same): liboflk then binds to the HIP runtime torch has already
# loaded; the other order leaves two runtimes in the process and torch may then see no GPU
try:
    import torch  # noqa: F401
except Exception:  # the CPU suite does not need it
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    """a plain `pytest tests` on a box without a usable GPU skips the gpu-marked tests instead of failing in them"""
    gpu_items = [it for it in items if it.get_closest_marker("gpu")]
    if not gpu_items:
        return
    try:
        import _oflk

        have = _oflk.device_count() >= 1
        why = "no usable HIP device (oflk_device_count() < 1)"
    except Exception as e:   # library not built
        have, why = False, f"liboflk not loadable: {e}"
    if not have:
        # An explicit `-m gpu` run (the GPU box's) or a machine that shows a GPU driver must FAIL here, not skip: a broken
        # build or runtime would otherwise read as "all skipped, rc 0".  Only a plain run on a box without a GPU skips.
        asked = "gpu" in (config.getoption("-m") or "") and "not gpu" not in (config.getoption("-m") or "")
        if asked or os.path.exists("/dev/kfd"):
            raise pytest.UsageError(f"GPU tests selected but the HIP path is unusable: {why}")
        skip = pytest.mark.skip(reason=why)
        for it in gpu_items:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    import oflk_oracle

    oflk_oracle.build()
    return oflk_oracle


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
