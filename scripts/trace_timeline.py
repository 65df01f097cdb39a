#!/usr/bin/env python3
"""Timeline of ONE registration out of a rocprofv3 --kernel-trace CSV: every kernel with its duration and the idle gap in front of it.

    python scripts/trace_timeline.py <..._kernel_trace.csv> [--anchor k_gicp_init|k_ndt_init] [--which -2]

An `align` starts at its init kernel (k_gicp_init / k_ndt_init); the script prints the `which`-th one (default: the second to last, a
warm steady-state call) from the first kernel after the previous registration's last one up to the next anchor, then the totals:
kernel time, gap time, and both per kernel name.  This is the evidence for the single-registration (latency) path: what a launch
boundary costs next to the work."""
import argparse
import collections
import csv
import re


def short(name):
    m = re.search(r"radix_sort_onesweep_(\w+)", name)
    if m:
        return f"rocprim::onesweep_{m.group(1)}"
    name = re.sub(r"\(.*", "", name).replace("void ", "").replace("hgs::", "")
    return name[:44]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--anchor", default="")
    ap.add_argument("--which", type=int, default=-2)
    ap.add_argument("--max-rows", type=int, default=80)
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.csv)))
    ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in rows), key=lambda t: t[0])
    anchor = a.anchor or ("k_gicp_init" if any(k[2].startswith("k_gicp_init") for k in ks) else "k_ndt_init")
    idx = [i for i, k in enumerate(ks) if k[2].startswith(anchor)]
    if len(idx) < 3:
        raise SystemExit(f"fewer than three {anchor} kernels in the trace")
    i0 = idx[a.which]
    nxt = [i for i in idx if i > i0]
    i1 = nxt[0] if nxt else len(ks)
    # the registration's front (index build, covariances) precedes its init kernel: walk back to the previous registration's last kernel,
    # recognised by a gap of more than 200 us (host work between two align calls)
    j = i0
    while j > 0 and ks[j][0] - ks[j - 1][1] < 200_000 and (j - 1) not in idx:
        j -= 1
    # ... and forward from the init kernel to the end of this registration (the next front starts after a long gap)
    e = i0
    while e + 1 < i1 and ks[e + 1][0] - ks[e][1] < 200_000:
        e += 1
    seg = ks[j:e + 1]
    t0 = seg[0][0]
    print(f"registration {a.which} of {len(idx)}: {len(seg)} kernels, {(seg[-1][1] - t0) / 1e3:.1f} us from the first kernel's start to the last one's end\n")
    print("| t us | kernel | duration us | gap in front us |\n|---:|---|---:|---:|")
    per = collections.OrderedDict()
    gap_total = 0
    for n, (s, t, name) in enumerate(seg):
        gap = 0 if n == 0 else max(0, s - max(x[1] for x in seg[:n]))
        gap_total += gap
        p = per.setdefault(name, [0, 0, 0])
        p[0] += 1
        p[1] += t - s
        p[2] += gap
        if n < a.max_rows:
            print(f"| {(s - t0) / 1e3:.1f} | {name} | {(t - s) / 1e3:.2f} | {gap / 1e3:.2f} |")
    if len(seg) > a.max_rows:
        print(f"| ... | ({len(seg) - a.max_rows} more) | | |")
    busy = sum(t - s for s, t, _ in seg)
    print(f"\nkernel time {busy / 1e3:.1f} us, idle between kernels {gap_total / 1e3:.1f} us\n")
    print("| kernel | launches | total us | avg us | idle in front, total us | avg us |\n|---|---:|---:|---:|---:|---:|")
    for name, (c, d, g) in sorted(per.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
        print(f"| {name} | {c} | {d / 1e3:.1f} | {d / c / 1e3:.2f} | {g / 1e3:.1f} | {g / c / 1e3:.2f} |")


if __name__ == "__main__":
    main()
