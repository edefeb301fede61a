import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def usable_cores(cap=16):
    """CPU threads this process may actually keep busy: the affinity mask AND the cgroup quota.  The GPU boxes show 256 logical CPUs to
    a container whose quota is 16: torch then starts 128 intra-op threads and every CPU-oracle call of the parity tests crawls (measured
    round 4, tools/diag_suite_time.py: the 70-frame medium-config oracle stream 13.7 s with 128 threads, 1.1 s with 16)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p_ = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p_))
        except (OSError, ValueError):
            pass
    return max(1, min(n, cap))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the CPU oracles of the parity tests (and the subprocesses the tests start) run on the cores this container really has
    n = usable_cores()
    os.environ.setdefault("OMP_NUM_THREADS", str(n))
    os.environ.setdefault("MKL_NUM_THREADS", str(n))
    import torch
    torch.set_num_threads(n)
    try:
        torch.set_num_interop_threads(max(1, min(n, 4)))
    except RuntimeError:
        pass                          # (already started: harmless)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _release_gpu_objects_between_tests(request):
    """hipGraphs and their memory pools of a finished test are destroyed HERE, at a quiescent point, rather than whenever
    the garbage collector happens to run inside a later test's capture or replay."""
    yield
    if request.node.get_closest_marker("gpu") is None:
        return
    import gc
    import torch
    if torch.cuda.is_available():
        torch.cuda.synchronize()
        gc.collect()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
