import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    path = os.path.join(GOLDEN, name + ".npz")
    if not os.path.exists(path):
        pytest.skip(f"golden fixture {name} missing")
    return np.load(path, allow_pickle=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(scope="session")
def unet_sd():
    from surfd_amd.synth import synth_unet_state_dict
    return synth_unet_state_dict()


@pytest.fixture(scope="session")
def decoder_sd32():
    from surfd_amd.synth import synth_decoder_state_dict
    return synth_decoder_state_dict()


def spawn_bounded(fn, args, nprocs, timeout_s=900):
    """torch.multiprocessing.spawn with a deadline: a collective that never completes (a rank died, RCCL without peer access on
    a new box) must fail the test, not hang the suite.  Only the processes started here are killed."""
    import time
    import torch.multiprocessing as mp
    ctx = mp.start_processes(fn, args=args, nprocs=nprocs, join=False, start_method="spawn")
    deadline = time.time() + timeout_s
    while not ctx.join(timeout=5):
        if time.time() > deadline:
            for proc in ctx.processes:
                if proc.is_alive():
                    proc.kill()
            pytest.fail(f"{fn.__name__}: {nprocs} ranks did not finish within {timeout_s} s")
