#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd .db (kernel trace) as a per-kernel stats table (markdown/CSV-ish).
usage: tools/rocpd_summary.py <results.db> [out.md]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, start, end from kernels").fetchall()
    agg = {}
    for name, s, e in rows:
        short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        a = agg.setdefault(short, [0, 0, 10**18, 0])
        d = e - s
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values()) or 1
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"| `{name[:110]}` | {a[0]} | {a[1] / 1e6:.3f} | {a[1] / a[0] / 1e3:.2f} | {a[2] / 1e3:.2f} | {a[3] / 1e3:.2f} | {100 * a[1] / total:.2f} |")
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()
