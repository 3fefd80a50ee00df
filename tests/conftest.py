import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A per-test time limit (pytest-timeout, when the image has it): a test that hangs - once in the round a runtime call
    on a pool box never returned - must end the run with a failure and a stack dump after minutes,
    not hold the GPU box until the caller's limit.  `thread` method: it also ends a test stuck inside a native call."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(600, method="thread"))


@pytest.fixture(scope="session", autouse=True)
def _host_thread_pools():
    """numpy's BLAS and torch's CPU pools size themselves by the LOGICAL CPU count - 256 on the GPU boxes, behind a 16-CPU
    cgroup quota, where surplus spinning threads cost up to 50 x (oracle/oracle.c): cap every pool at what the process may
    really use, as the oracle does for its own OpenMP regions."""
    try:
        from oracle.oracle import usable_cpus
        n = usable_cpus()
    except Exception:
        n = None
    ctl = None
    if n:
        try:
            import threadpoolctl
            ctl = threadpoolctl.threadpool_limits(limits=n)
        except Exception:
            ctl = None
        try:
            import torch
            torch.set_num_threads(n)
        except Exception:
            pass
    yield
    if ctl is not None:
        ctl.restore_original_limits()


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc
