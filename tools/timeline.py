#!/usr/bin/env python3
"""Kernel timeline of the LAST `--last N` kernels (or of one step delimited by a marker kernel) from a rocprofv3 rocpd .db:
start offset, duration, gap to the previous kernel's end, stream/queue if present.  usage: tools/timeline.py <db> [--last N]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    last = int(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else 60
    cur = db.cursor()
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    extra = [c for c in ("queue_id", "stream_id", "stream", "queue") if c in cols]
    q = f"select {name_col}, start, end{''.join(', ' + c for c in extra)} from kernels order by start"
    rows = cur.execute(q).fetchall()
    rows = rows[-last:]
    t0 = rows[0][1]
    prev_end = t0
    print(f"# {len(rows)} kernels; columns: start_us dur_us gap_us {' '.join(extra)} name")
    for r in rows:
        name, s, e = r[0], r[1], r[2]
        short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {(s - prev_end) / 1e3:8.1f} {' '.join(str(x) for x in r[3:])} {short}")
        prev_end = max(prev_end, e)
    print(f"# span {(max(r[2] for r in rows) - t0) / 1e3:.1f} us")


if __name__ == "__main__":
    main()
