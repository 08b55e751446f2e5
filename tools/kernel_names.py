"""Distinct kernel names (untruncated) of a rocprofv3 --kernel-trace database, with launch counts: python tools/kernel_names.py <db>"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
for name, n in con.execute("select s.kernel_name, count(*) from %s d join %s s on d.kernel_id = s.id group by s.kernel_name order by 2 desc" % (disp, sym)):
    print("%6d  %s" % (n, name))
