#!/usr/bin/env python3
"""Per-kernel launch count / average / total from a rocprofv3 --kernel-trace database (rocpd .db)."""
import collections
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
d = collections.defaultdict(list)
for name, s, e in c.execute("select name, start, end from kernels"):
    d[name[:90]].append(e - s)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k:90s} n={len(v):5d} avg={sum(v) / len(v) / 1000:9.1f} us total={sum(v) / 1e6:9.2f} ms")
if len(sys.argv) > 3 and sys.argv[2] == "--seq":       # durations of one kernel in launch order, REPS per line
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
    ds = [(e - s) / 1000 for name, s, e in c.execute("select name, start, end from kernels order by start") if sys.argv[3] in name]
    for i in range(0, len(ds), reps):
        print(" ".join(f"{x:8.1f}" for x in ds[i:i + reps]))
