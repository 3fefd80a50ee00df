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


# ---- gradient tolerance of the fp32 path against the fp64 oracle: ONE constant, set from measurements ------------------
# |g - g_ref|_max <= GRAD_TOL * max(|g_ref|_max, 1e-3) per tensor.  Round 6 measured every such comparison of the GPU suite
# (profiles/r6_grad_tolerance_measured.txt: 1548 comparisons in test_gpu_configs.py, largest 3.7e-5, the N = 1024 XCD-pair
# chain at T = 1000) - rounds 1-5 had asserted 2e-3, "a guess" (VERDICT round 5).  1e-4 is the north star's bar for logits
# and loss, 2.7 x the largest gradient error seen.
GRAD_TOL = 1e-4
_GRAD_ERRORS = []


def check_grad(got, ref, tag, key, tol=GRAD_TOL):
    """Asserts one gradient tensor against its oracle value and records the measured error for the session report."""
    import numpy as np
    den = max(float(np.abs(ref).max()), 1e-3)
    err = float(np.abs(got - ref).max())
    _GRAD_ERRORS.append((err / den, str(tag), str(key), tol))
    assert err < tol * den, (tag, key, err, tol * den)


def pytest_sessionfinish(session, exitstatus):
    """gpurun_out/r6/grad_tolerance_measured.txt: the largest measured error per tolerance class and the top 25."""
    if not _GRAD_ERRORS:
        return
    out = os.path.join(ROOT, "gpurun_out", "r6")
    try:
        os.makedirs(out, exist_ok=True)
        by_tol = {}
        for e, tag, k, tol in _GRAD_ERRORS:
            if e > by_tol.get(tol, (-1.0,))[0]:
                by_tol[tol] = (e, tag, k)
        with open(os.path.join(out, "grad_tolerance_measured.txt"), "w") as f:
            f.write("# tolerance class -> largest measured |g - g_ref|_max / max(|g_ref|_max, 1e-3), case, tensor (%d comparisons)\n"
                    % len(_GRAD_ERRORS))
            for tol, (e, tag, k) in sorted(by_tol.items()):
                f.write("%g\t%.3e\t%s\t%s\n" % (tol, e, tag, k))
            for e, tag, k, tol in sorted(_GRAD_ERRORS, reverse=True)[:25]:
                f.write("top\t%.3e\t%s\t%s\t(tol %g)\n" % (e, tag, k, tol))
    except OSError:
        pass
