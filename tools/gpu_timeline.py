"""Where does the scene period go?  Timeline analysis of a rocprofv3 --kernel-trace database (rocpd sqlite) of
`python bench.py` with several scenes in flight:
    python tools/gpu_timeline.py <db> [marker=occ_decode8_kernel]
Window = from the 25 % to the 90 % quantile of the marker kernel's launches (steady state of the timed region).
Reports, as a share of the window: time with at least one kernel running, time with the marker kernel running, time with
only other kernels running, idle time; the per-kernel exclusive time (time during which the kernel was the ONLY one
running), and the largest idle gaps with the kernels on either side."""
import sqlite3
import sys


def load(db):
    con = sqlite3.connect(db)
    tables = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
    disp = next(t for t in tables if t.startswith("rocpd_kernel_dispatch"))
    sym = next(t for t in tables if t.startswith("rocpd_info_kernel_symbol"))
    rows = con.execute("select s.kernel_name, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start"
                       % (disp, sym)).fetchall()
    return [(n, int(s), int(e)) for n, s, e in rows]


def short(n):
    n = n.replace("_ZN12_GLOBAL__N_1", "")
    return n[:60]


def main():
    db = sys.argv[1]
    marker = sys.argv[2] if len(sys.argv) > 2 else "occ_decode8_kernel"
    rows = load(db)
    ms = [r for r in rows if marker in r[0] and r[2] - r[1] > 2_000_000]        # real launches (> 2 ms)
    if len(ms) < 8:
        raise SystemExit("only %d marker launches" % len(ms))
    t0, t1 = ms[len(ms) // 4][1], ms[(len(ms) * 9) // 10][2]
    win = [(n, max(s, t0), min(e, t1)) for n, s, e in rows if e > t0 and s < t1]
    n_marker = sum(1 for r in ms if t0 <= r[1] < t1)
    # sweep
    ev = []
    for i, (n, s, e) in enumerate(win):
        ev.append((s, 1, i))
        ev.append((e, -1, i))
    ev.sort()
    active = set()
    last = t0
    busy = marker_busy = other_only = idle = 0
    excl = {}
    present = {}          # kernel -> time it was running while the marker kernel was NOT (1/n share when n kernels run)
    with_marker = {}      # kernel -> time it was running together with the marker kernel
    gaps = []
    prev_end_name = None
    for t, d, i in ev:
        dt = t - last
        if dt > 0:
            if not active:
                idle += dt
                gaps.append((dt, last, prev_end_name))
            else:
                busy += dt
                names = {win[j][0] for j in active}
                if any(marker in n for n in names):
                    marker_busy += dt
                    for n in names:
                        if marker not in n:
                            with_marker[n] = with_marker.get(n, 0) + dt
                else:
                    other_only += dt
                    for n in names:
                        present[n] = present.get(n, 0) + dt / len(names)
                if len(active) == 1:
                    n = win[next(iter(active))][0]
                    excl[n] = excl.get(n, 0) + dt
        if d == 1:
            if not active and gaps and gaps[-1][1] == last and len(gaps[-1]) == 3:
                gaps[-1] = gaps[-1] + (win[i][0],)
            active.add(i)
        else:
            active.discard(i)
            prev_end_name = win[i][0]
        last = t
    W = float(t1 - t0)
    print("# window %.1f ms, %d marker launches (%.2f ms per launch period), %d dispatches" % (W / 1e6, n_marker, W / 1e6 / max(1, n_marker), len(win)))
    print("busy (>= 1 kernel)      %6.2f %%" % (100 * busy / W))
    print("  marker kernel running %6.2f %%" % (100 * marker_busy / W))
    print("  only other kernels    %6.2f %%" % (100 * other_only / W))
    print("idle                    %6.2f %%" % (100 * idle / W))
    print("# exclusive time (the kernel was the only one running), share of the window")
    for n, v in sorted(excl.items(), key=lambda kv: -kv[1])[:14]:
        print("  %6.2f %%  %s" % (100 * v / W, short(n)))
    print("# time WITHOUT the marker kernel, attributed to the kernels running then (1/n each when n run together)")
    for n, v in sorted(present.items(), key=lambda kv: -kv[1])[:16]:
        print("  %6.2f %%  %s" % (100 * v / W, short(n)))
    print("# kernels running TOGETHER with the marker kernel (share of the window)")
    for n, v in sorted(with_marker.items(), key=lambda kv: -kv[1])[:10]:
        print("  %6.2f %%  %s" % (100 * v / W, short(n)))
    print("# largest idle gaps: ms, after -> before")
    for g in sorted(gaps, key=lambda g: -g[0])[:12]:
        print("  %7.3f ms  %s -> %s" % (g[0] / 1e6, short(g[2] or "?"), short(g[3]) if len(g) > 3 else "?"))
    tot_gaps = sorted((g[0] for g in gaps), reverse=True)
    print("# %d gaps; the 10 largest hold %.1f %% of the idle time" % (len(gaps), 100 * sum(tot_gaps[:10]) / max(1, idle)))


if __name__ == "__main__":
    main()
