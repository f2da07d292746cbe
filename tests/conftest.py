import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The oracles (tests' checker) run on the host with torch.  On the GPU box torch defaults to 128 threads of 256 host cpus and the f32 / bf16-emulating
    # oracle steps of tests/test_c2_parity_gpu.py take 269 s; with 32 threads (what bench.py's cpu_baseline uses) the same two tests take 86 s.
    try:
        import torch
        torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    except Exception:
        pass


@pytest.fixture(scope="session")
def lib():
    """Builds (if needed) and loads liburse_hip.so."""
    import __graft_entry__ as g
    g.build()
    from urgent2026_challenge_track1_amd import _lib
    return _lib.load()
