"""pytest plugin: runs the `-m gpu` tests against the host emulation of the kernels (tests/emul/simt.py) instead of an MI355X.

  PYTHONPATH=tests/emul/plugins python -m pytest tests -m gpu -p simt_everywhere -k "not full_size and not rccl and not hdl32_raw and not dense and not adapter_matches"

(13 minutes for the 79 tests that fit; the full-size cases and the two tests that start other processes against the real
library are left out.)  HGS_SIMT_LIB=<path> substitutes another build of the emulated library, e.g. one compiled with
-fsanitize=address,undefined (scripts/emulated_sanitizers.sh)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


def pytest_configure(config):
    from emul import simt
    from hdl_graph_slam_amd import _lib as L
    L.LIB_PATH, L._lib = os.environ.get("HGS_SIMT_LIB") or simt.build(), None
