"""Ordered kernel sequence of the LAST scene of a rocprofv3 --kernel-trace database (one scene in flight):
which launches make up a scene, in order, with their GPU time -- the work list for folding glue launches.
    python tools/launch_sequence.py <db> [marker]"""
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"^_ZN\d+_GLOBAL__N_1\d+", "", n)
    n = re.sub(r"^_ZN2at6native\d*", "at::", n)
    return n[:78]


def main(db, marker="fps_kernelILi10"):
    con = sqlite3.connect(db)
    rows = con.execute("select s.kernel_name, d.start, d.end, d.grid_size_x, d.workgroup_size_x from rocpd_kernel_dispatch d "
                       "join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
    starts = [i for i, r in enumerate(rows) if marker in r[0]]
    # the last COMPLETE scene: between the last two markers
    a, b = (starts[-2], starts[-1]) if len(starts) >= 2 else (starts[-1], len(rows))
    seq = rows[a:b]
    t0 = seq[0][1]
    print("# %d launches, %.3f ms busy, %.3f ms first-to-last" % (
        len(seq), sum(r[2] - r[1] for r in seq) / 1e6, (max(r[2] for r in seq) - t0) / 1e6))
    for i, (n, s, e, g, w) in enumerate(seq):
        print("%4d %9.3f ms  +%8.2f us  grid %8d/%4d  %s" % (i, (s - t0) / 1e6, (e - s) / 1e3, g, w, short(n)))


if __name__ == "__main__":
    main(*sys.argv[1:])
