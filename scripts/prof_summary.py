#!/usr/bin/env python3
"""Compact summary of a rocprofv3 `--kernel-trace --stats` kernel_stats.csv (rocPRIM template names are shortened)."""
import csv
import re
import sys


def short(name: str) -> str:
    m = re.search(r"radix_sort_onesweep_(\w+)", name)
    if m:
        return f"rocprim::radix_sort_onesweep_{m.group(1)}"
    name = re.sub(r"\(.*", "", name)
    return name.replace("void ", "")[:60]


def main(path):
    rows = list(csv.DictReader(open(path)))
    agg = {}
    for r in rows:
        k = short(r["Name"])
        a = agg.setdefault(k, [0, 0.0, 1e30, 0.0])
        a[0] += int(r["Calls"])
        a[1] += float(r["TotalDurationNs"])
        a[2] = min(a[2], float(r["MinNs"]))
        a[3] = max(a[3], float(r["MaxNs"]))
    total = sum(a[1] for a in agg.values())
    print(f"| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---:|---:|---:|---:|---:|---:|")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"| {k} | {a[0]} | {a[1] / 1e6:.3f} | {a[1] / a[0] / 1e3:.2f} | {a[2] / 1e3:.2f} | {a[3] / 1e3:.2f} | {100 * a[1] / total:.2f} |")
    print(f"\ntotal kernel time {total / 1e6:.3f} ms")


if __name__ == "__main__":
    main(sys.argv[1])
