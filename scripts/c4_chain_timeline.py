#!/usr/bin/env python3
"""Start/end of every solve-chain launch of the last C4 solve in a rocprofv3 --kernel-trace database."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table' or type='view'")]
kd = [t for t in tabs if "rocpd_kernel_dispatch" in t][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = [r for r in db.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x, d.stream_id from {kd} d join {ks} s on d.kernel_id=s.id order by d.start")
        if "lk_solve" in r[0]]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows = rows[-n:]
t0 = rows[0][1]
for name, s, e, g, st in rows:
    print(f"  {name[34:62]:28s} grid {g // 64:5d}  stream {st}  {(s - t0) / 1e3:8.1f} .. {(e - t0) / 1e3:8.1f} us  ({(e - s) / 1e3:7.1f})")
print(f"  chain {(max(r[2] for r in rows) - t0) / 1e3:.1f} us")
