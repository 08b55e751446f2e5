"""Which kernels share the chip with each decoder launch?  (the 128^3 configuration's slow round 1
and its 2-5 ms tail launches.)  Reads a rocprofv3 --kernel-trace database (rocpd sqlite) of `python bench.py --config
mise128 ...`:
    python tools/decoder_rounds.py <db> [marker=occ_decode8_kernel]
Decoder launches of the steady-state window are classified by their grid (workgroups) and duration into the MISE rounds'
shapes; for every class: launches, mean / max duration, the mean time other kernels ran inside the launch's [start, end],
and which ones (top 4 by overlapped time).  Then the 12 slowest SMALL launches (grid < 256 workgroups) one by one with
everything that overlapped them."""
import sqlite3
import sys


def load(db):
    con = sqlite3.connect(db)
    tables = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
    disp = next(t for t in tables if t.startswith("rocpd_kernel_dispatch"))
    sym = next(t for t in tables if t.startswith("rocpd_info_kernel_symbol"))
    cols = [r[1] for r in con.execute("pragma table_info(%s)" % disp)]
    gx = "d.grid_size_x" if "grid_size_x" in cols else ("d.grid_x" if "grid_x" in cols else "0")
    wx = "d.workgroup_size_x" if "workgroup_size_x" in cols else ("d.workgroup_x" if "workgroup_x" in cols else "1")
    q = "d.queue_id" if "queue_id" in cols else "0"
    rows = con.execute("select s.kernel_name, d.start, d.end, %s, %s, %s from %s d join %s s on d.kernel_id = s.id "
                       "order by d.start" % (gx, wx, q, disp, sym)).fetchall()
    return [(n, int(s), int(e), int(g or 0), int(w or 1), qq) for n, s, e, g, w, qq in rows], cols


def short(n):
    return n.replace("_ZN12_GLOBAL__N_1", "").split("(")[0][:48]


def main():
    db = sys.argv[1]
    marker = sys.argv[2] if len(sys.argv) > 2 else "occ_decode8_kernel"
    rows, cols = load(db)
    dec = [r for r in rows if marker in r[0]]
    big = [r for r in dec if r[2] - r[1] > 2_000_000]
    if len(big) < 8:
        raise SystemExit("only %d long marker launches" % len(big))
    t0, t1 = big[len(big) // 4][1], big[(len(big) * 9) // 10][2]
    win = [r for r in rows if r[2] > t0 and r[1] < t1]
    others = [r for r in win if marker not in r[0]]
    print("# window %.1f ms, %d dispatches, %d decoder launches; dispatch columns: %s"
          % ((t1 - t0) / 1e6, len(win), sum(1 for r in dec if t0 <= r[1] < t1), ",".join(cols)))

    def overlaps(r):
        acc = {}
        for n, s, e, *_ in others:
            if e <= r[1]:
                continue
            if s >= r[2]:
                break
            acc[n] = acc.get(n, 0) + min(e, r[2]) - max(s, r[1])
        return acc

    classes = {}
    detail = []
    for r in dec:
        if not (t0 <= r[1] < t1):
            continue
        wgs = r[3] // max(r[4], 1) if r[3] >= r[4] else r[3]
        dur = (r[2] - r[1]) / 1e6
        key = ("full grid (%d wg)" % wgs if wgs >= 200 else "small grid (< 200 wg)",
               "> 20 ms" if dur > 20 else "5-20 ms" if dur > 5 else "1-5 ms" if dur > 1 else "< 1 ms")
        ov = overlaps(r)
        c = classes.setdefault(key, [0, 0.0, 0.0, 0.0, {}])
        c[0] += 1
        c[1] += dur
        c[2] = max(c[2], dur)
        c[3] += sum(ov.values()) / 1e6
        for n, v in ov.items():
            c[4][n] = c[4].get(n, 0) + v
        if wgs < 200:
            detail.append((dur, wgs, r, ov))
    print("%-24s %-8s %8s %9s %9s %12s  top overlapping kernels (ms per launch)" % ("grid", "duration", "launches", "mean ms", "max ms", "other-ms/l"))
    for key in sorted(classes):
        n, tot, mx, ovt, names = classes[key]
        top = sorted(names.items(), key=lambda kv: -kv[1])[:4]
        print("%-24s %-8s %8d %9.3f %9.3f %12.3f  %s" % (key[0], key[1], n, tot / n, mx, ovt / n,
              "; ".join("%s %.2f" % (short(k), v / 1e6 / n) for k, v in top)))
    print("# slowest small-grid launches: duration, workgroups, then every kernel that overlapped it (its overlap / its own duration)")
    for dur, wgs, r, ov in sorted(detail, key=lambda d: -d[0])[:12]:
        print("  %.3f ms, %d wg, queue %s:" % (dur, wgs, r[5]))
        for n, s, e, g, w, qq in others:
            if e <= r[1] or s >= r[2]:
                continue
            print("      %-48s overlap %7.3f ms of its %8.3f ms  (queue %s)" % (short(n), (min(e, r[2]) - max(s, r[1])) / 1e6, (e - s) / 1e6, qq))
    small = sorted(d[0] for d in detail)
    if small:
        print("# small-grid launches: n %d, median %.3f ms, p90 %.3f ms, max %.3f ms" % (len(small), small[len(small) // 2], small[(len(small) * 9) // 10], small[-1]))


if __name__ == "__main__":
    main()
