"""per-kernel totals out of a rocprofv3 results .db (rocpd): python tools/probes/rocpd_stats.py file.db [--seq]"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); c = db.cursor()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = list(c.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x, d.workgroup_size_x from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
short = lambda n: re.sub(r'\(.*', '', re.sub(r'^void ', '', n))[:70]
if '--seq' in sys.argv:
    for n, s, e, g, w in rows: print("%10.3f us  %-70s grid %d x %d" % ((e - s) / 1e3, short(n), g // max(w, 1), w))
else:
    agg = {}
    for n, s, e, g, w in rows:
        a = agg.setdefault(short(n), [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
    for n, (k, t) in sorted(agg.items(), key=lambda x: -x[1][1]): print("%6d calls %12.1f us total %10.2f us avg  %s" % (k, t, t / k, n))
