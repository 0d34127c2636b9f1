import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU oracle is ~150 k small ATen calls per plan: on the GPU boxes' 256-logical-core hosts torch's default thread count makes every one of
    # them slower (bench.py's cpu_baseline probe: 16 threads is the fastest form).  Cap it for the whole session (MPDX_TEST_THREADS overrides).
    import torch
    cap = int(os.environ.get("MPDX_TEST_THREADS", "16"))
    if cap > 0 and torch.get_num_threads() > cap:
        torch.set_num_threads(cap)
    if cap > 0:   # child processes of the tests (the fuzz tools, the variant / two-rank scripts) inherit the same cap
        os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(cap, os.cpu_count() or cap))))


@pytest.fixture(scope="session")
def golden_dir():
    return ROOT / "tests" / "golden"
