#!/usr/bin/env python
"""Per-kernel call count and average duration from a rocprofv3 --kernel-trace results database (rocpd sqlite)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2:]
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
q = f"select s.kernel_name, count(*), avg(d.end-d.start), sum(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 4 desc"
for n, c, a, t in cur.execute(q):
    if not pat or any(k in n for k in pat):
        print(f"{n[:72]:72s} {c:6d} calls {a / 1e3:9.2f} us avg {t / 1e6:9.2f} ms total")
