"""Every `file:line` citation of the reference in the interface header, the adapters, the host mirrors and the documents points
at an existing file of /root/reference with at least that many lines.  Skipped where the reference is absent (the GPU box)."""
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
CITE = re.compile(r"(?<![\w/.])((?:[\w.-]+/)*[\w.-]+\.(?:cpp|hpp|launch|xml|txt))`?:(\d+)(?:[-–](\d+))?")


def _sources():
    pats = ["include/*.h", "adapters/*.hpp", "hdl_graph_slam_amd/*.py", "hdl_graph_slam_amd/csrc/*", "oracle/*.hpp", "oracle/*.cpp", "DESIGN.md",
            "INTEGRATION.md", "bench.py", "tests/cpp/*.cpp"]
    return sorted(f for p in pats for f in glob.glob(os.path.join(ROOT, p)))


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is not present here")
def test_reference_citations_resolve():
    index = {}
    for d, _, files in os.walk(REF):
        for f in files:
            index.setdefault(f, []).append(os.path.join(d, f))
    own = {os.path.basename(f) for f in glob.glob(os.path.join(ROOT, "**", "*"), recursive=True)}
    checked, bad = 0, []
    for src in _sources():
        text = open(src, errors="replace").read()
        for m in CITE.finditer(text):
            path, lo, hi = m.group(1), int(m.group(2)), int(m.group(3) or m.group(2))
            base = os.path.basename(path)
            cands = [c for c in index.get(base, []) if c.endswith("/" + path) or "/" not in path]
            if not cands:
                if base in own or base not in index:
                    continue            # a citation of this repository's own files, or of a third-party file named in prose
                bad.append((os.path.relpath(src, ROOT), m.group(0), "no such path in the reference"))
                continue
            n_lines = max(sum(1 for _ in open(c, errors="replace")) for c in cands)
            checked += 1
            if hi < lo or hi > n_lines:
                bad.append((os.path.relpath(src, ROOT), m.group(0), f"file has {n_lines} lines"))
    assert checked > 100, checked
    assert not bad, bad[:20]
