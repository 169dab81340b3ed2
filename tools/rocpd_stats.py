#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.x, rocpd sqlite output) kernel trace into the per-kernel stats table
`rocprofv3 --kernel-trace --stats` describes: calls, total / average / min / max duration, share.
Usage: tools/rocpd_stats.py <results.db> [> profiles/rNN_kernel_stats.md]"""
import re
import sqlite3
import sys


def short(name):
    name = name.replace("fluid::(anonymous namespace)::", "")
    name = re.sub(r"^void\s+", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:70]


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, start, end from kernels").fetchall()
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(short(name), [0, 0, 1 << 62, 0])
        d = e - s
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values()) or 1
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("| %s | %d | %.3f | %.2f | %.2f | %.2f | %.1f |" % (k, a[0], a[1] / 1e6, a[1] / a[0] / 1e3, a[2] / 1e3, a[3] / 1e3, 100.0 * a[1] / total))


if __name__ == "__main__":
    main(sys.argv[1])
