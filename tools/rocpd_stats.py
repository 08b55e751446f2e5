"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) as a per-kernel
table: calls, total / average / min / max duration.  Used to produce the
summaries committed under profiles/."""
import sqlite3
import sys


def last_scene(db, marker="fps_kernelILi10", top=40):
    """Per-kernel table of the LAST scene only (from the last dispatch whose name
    contains `marker` to the end) = steady state, free of warm-up / MIOpen find."""
    con = sqlite3.connect(db)
    cur = con.cursor()
    rows = cur.execute("select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d join "
                       "rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
    starts = [r[1] for r in rows if marker in r[0]]
    t0 = starts[-1]
    last = [r for r in rows if r[1] >= t0]
    agg = {}
    for n, s_, e in last:
        a = agg.setdefault(n, [0, 0, 10 ** 18, 0])
        a[0] += 1
        a[1] += e - s_
        a[2] = min(a[2], e - s_)
        a[3] = max(a[3], e - s_)
    tot = sum(v[1] for v in agg.values())
    print("# last scene: %d dispatches, %.3f ms busy, %.3f ms first-to-last"
          % (len(last), tot / 1e6, (max(r[2] for r in last) - t0) / 1e6))
    print("%-90s %7s %12s %12s %12s %12s %6s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct"))
    for n, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        nm = n if len(n) <= 90 else n[:87] + "..."
        print("%-90s %7d %12.3f %12.2f %12.2f %12.2f %6.2f" % (nm, v[0], v[1] / 1e6, v[1] / v[0] / 1e3,
                                                                 v[2] / 1e3, v[3] / 1e3, 100.0 * v[1] / tot))


def main(db, top=40):
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
    sym_cols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    name_col = "kernel_name" if "kernel_name" in sym_cols else "display_name"
    q = ("select s.%s, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), "
         "max(d.end - d.start) from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
         "on d.kernel_id = s.id group by s.%s order by 3 desc" % (name_col, name_col))
    rows = cur.execute(q).fetchall()
    total = sum(r[2] for r in rows)
    span = cur.execute("select min(start), max(end) from rocpd_kernel_dispatch").fetchone()
    print("# kernels: %d dispatches, %.3f ms busy, %.3f ms first-to-last"
          % (sum(r[1] for r in rows), total / 1e6, (span[1] - span[0]) / 1e6))
    print("%-90s %7s %12s %12s %12s %12s %6s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct"))
    for name, n, tot, avg, mn, mx in rows[:top]:
        nm = name if len(name) <= 90 else name[:87] + "..."
        print("%-90s %7d %12.3f %12.2f %12.2f %12.2f %6.2f" % (nm, n, tot / 1e6, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "--last-scene":
        last_scene(sys.argv[1])
    else:
        main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
