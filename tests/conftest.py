"""pytest configuration: `gpu` marker + shared helpers.

`-m "not gpu"`: oracle vs golden vectors, host logic, C-ABI symbol checks (no GPU needed).
`-m gpu`:       parity tests proper -- they call the HIP kernels through the C-ABI on cuda:0.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def _gpu_count():
    # torch.cuda.device_count() does not initialise the GPU on this image (torch.cuda.is_available() does): a process that has
    # initialised it may not exec another program on the MI355X pool, and the fork server below must start before that
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:  # pragma: no cover
        return 0


# Tests that must START PROGRAMS on the GPU box (tests/test_bench_launch_gpu.py: `python bench.py --gpus 2`) do it through a fork
# server that is started here, while this process is still GPU-free: its children never inherit a GPU-initialised state.
FORKSERVER = None
import multiprocessing as _mp                              # noqa: E402
if _mp.parent_process() is None and _gpu_count() > 0:      # the pytest process itself, never a child that imports this module
    from multiprocessing import forkserver as _fs
    FORKSERVER = _mp.get_context("forkserver")
    _fs.ensure_running()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests are skipped (not failed) when no device is present, e.g. a bare `pytest tests/`.
    if _gpu_count() > 0:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load
