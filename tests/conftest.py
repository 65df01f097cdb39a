import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


@pytest.fixture(scope="session", autouse=True)
def oracle_thread_count():
    """The oracle's OpenMP loops default to every hardware thread; on the GPU box (2 x 128 threads) that is ~40x SLOWER than 32
    threads for these per-point loops (measured, bench.py), and the NDT parity tests run hundreds of oracle passes."""
    try:
        import oracle as O
        O.set_num_threads(min(32, os.cpu_count() or 1))
    except Exception:  # noqa: BLE001  (oracle not built yet: the tests that need it will say so)
        pass


@pytest.fixture(scope="session")
def small_pair():
    """~4k-point VLP-16 pair (voxel 0.4 m) that the oracle aligns in milliseconds."""
    from hdl_graph_slam_amd import synth
    tgt, src, T = synth.make_pair("VLP-16", 1, downsample=0.4)
    return tgt, src, T


@pytest.fixture(scope="session")
def medium_pair():
    """VLP-16 pair with the hdl prefilter (voxel 0.1 m) — BASELINE config 1 (~15-20k points)."""
    from hdl_graph_slam_amd import synth
    tgt, src, T = synth.make_pair("VLP-16", 2, downsample=0.1)
    return tgt, src, T
