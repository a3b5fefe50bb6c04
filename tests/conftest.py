import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the suite IS the A/B harness: the engine honours its switches (MCGRA_NO_LOWRANK, MCGRA_KEEP_GSYM, ...) only beside MCGRA_AB=1
# (attack.hip: ab_env); set before the fork server of the multi-process tests starts, so their ranks inherit it
os.environ["MCGRA_AB"] = "1"
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def _start_clean_process_factory():
    """tests/test_gpu_multiproc.py runs several ranks of one attack as separate processes on the box's GPU.  A process
    that has initialised the GPU must not fork + exec others (the GPU box refuses that), so the factory of those ranks --
    a multiprocessing fork server, a process that never touches the GPU and forks its children WITHOUT exec -- is started
    here, before anything in this pytest process initialises the device (torch.cuda.device_count() does not)."""
    try:
        import torch
        if torch.cuda.device_count() == 0:
            return
    except Exception:
        return
    import multiprocessing as mp
    from multiprocessing import forkserver
    mp.set_forkserver_preload([])
    forkserver.ensure_running()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    _start_clean_process_factory()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
