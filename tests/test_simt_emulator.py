"""Self-test of the SIMT emulation shim itself (tests/emul/simt_selftest.cpp): the cross-lane operations among the live lanes of
a wave, returned lanes, block barriers, atomics, grids — the semantics the emulated product tests rely on."""
import os
import subprocess

import pytest

simt = pytest.importorskip("emul.simt", reason="needs tests/emul")
HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emul")


def test_emulator_semantics(tmp_path):
    cxx = simt._clang()
    if cxx is None:
        pytest.skip("clang++ not available")
    exe = str(tmp_path / "simt_selftest")
    subprocess.run([cxx, "-x", "c++", "-std=c++17", "-O1", "-Wno-unknown-attributes", "-I", os.path.join(HERE, "simt"),
                    os.path.join(HERE, "simt_selftest.cpp"), os.path.join(HERE, "simt_runtime.cpp"), "-o", exe], check=True)
    for order in ("", "reverse", "shuffle:3"):
        out = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, HGS_SIMT_ORDER=order))
        assert out.returncode == 0 and out.stdout.strip() == "ok", (order, out.stdout, out.stderr)
