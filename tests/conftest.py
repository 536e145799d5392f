import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _usable_cores() -> int:
    """CPU share of this process: the cgroup quota when there is one (a GPU box shows 256 cores and grants 16), else the
    affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return n


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The oracle is many small torch ops in Python loops: with torch's default of one thread per VISIBLE core (256 on a
    # GPU box that grants 16) every op pays for a 256-thread fork/join and the suite runs an order of magnitude slower.
    import torch
    torch.set_num_threads(max(1, min(8, _usable_cores())))


@pytest.fixture(scope="session")
def lib():
    from synference_amd import _lib
    return _lib.load()
