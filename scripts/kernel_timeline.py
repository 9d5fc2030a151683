import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table' or type='view'")]
kd = [t for t in tabs if "rocpd_kernel_dispatch" in t][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = list(db.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
n = int(sys.argv[2])
rows = rows[-n:]
t0 = rows[0][1]
for name, s, e, g in rows:
    print(f"  {name[:70]:70s} grid {g:9d} {(s-t0)/1e3:9.1f} .. {(e-t0)/1e3:9.1f} us ({(e-s)/1e3:8.1f})")
