#!/usr/bin/env python3
"""Static picture of the gfx950 code of selected kernels: instruction mix, registers, scratch, LDS.

    python scripts/isa_stats.py k_ndt_pass k_gicp_linearize            # compiles hgs_kernels.hip with --save-temps into /tmp/hgs_isa
    python scripts/isa_stats.py --asm /tmp/hgs_isa/x.s k_knn_cov

Counts are of the kernel's TEXT (every instruction once, whatever the loop structure): what a change did to the register budget, the
scratch frame and the kind of instructions issued — not how often they run (that is rocprofv3 --pmc's SQ_INSTS_*).
"""
from __future__ import annotations

import argparse
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def compile_asm(extra):
    out = "/tmp/hgs_isa"
    os.makedirs(out, exist_ok=True)
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function", "--save-temps", *extra, "-c",
           os.path.join(ROOT, "hdl_graph_slam_amd", "csrc", "hgs_kernels.hip"), "-o", "k.o"]
    subprocess.run(cmd, cwd=out, check=True, stderr=subprocess.DEVNULL)
    return os.path.join(out, "hgs_kernels-hip-amdgcn-amd-amdhsa-gfx950.s")


def kernels(asm):
    text = open(asm).read()
    for m in re.finditer(r"\n(_Z\w+):[^\n]*\n(.*?)\n\t\.amdhsa_kernel \1\n(.*?)\.end_amdhsa_kernel", text, re.S):
        yield m.group(1), m.group(2), m.group(3)


def demangle(name):
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
    except OSError:
        return name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("patterns", nargs="+")
    ap.add_argument("--asm")
    ap.add_argument("-D", action="append", default=[])
    a = ap.parse_args()
    asm = a.asm or compile_asm(["-D" + d for d in a.D])
    for name, body, meta in kernels(asm):
        pretty = demangle(name)
        if not any(p in pretty for p in a.patterns):
            continue
        ins = [l.split()[0] for l in body.split("\n") if l.startswith("\t") and not l.lstrip().startswith((".", ";"))]
        c = collections.Counter(ins)
        grp = collections.Counter()
        for k, v in c.items():
            if k.startswith("v_mfma"):
                grp["mfma"] += v
            elif k.startswith("v_pk_"):
                grp["valu_pk"] += v
            elif k.startswith("v_") and "f64" in k:
                grp["valu_f64"] += v
            elif k.startswith("v_"):
                grp["valu_other"] += v
            elif k.startswith("s_"):
                grp["salu"] += v
            elif k.startswith("ds_"):
                grp["lds"] += v
            elif k.startswith(("global_", "flat_", "buffer_")):
                grp["vmem"] += v
            elif k.startswith("scratch_"):
                grp["scratch"] += v
        def field(key):
            m = re.search(r"\." + key + r"\s+(\d+)", meta)
            return int(m.group(1)) if m else None
        print(f"{pretty}")
        print(f"   instructions {len(ins)}: " + ", ".join(f"{k} {v}" for k, v in sorted(grp.items())))
        print(f"   next_free_vgpr {field('amdhsa_next_free_vgpr')}  accum_offset {field('amdhsa_accum_offset')}  next_free_sgpr {field('amdhsa_next_free_sgpr')}  "
              f"scratch {field('amdhsa_private_segment_fixed_size')} B  LDS {field('amdhsa_group_segment_fixed_size')} B")
        top = ", ".join(f"{k} {v}" for k, v in c.most_common(12))
        print(f"   top: {top}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
