"""CPU: tools/gpu_timeline.py on a synthetic rocpd database (the shares it reports are what DESIGN.md section 5 quotes)."""
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_timeline_shares_on_a_synthetic_trace(tmp_path):
    db = str(tmp_path / "t.db")
    con = sqlite3.connect(db)
    con.execute("create table rocpd_info_kernel_symbol_x (id integer, kernel_name text)")
    con.execute("create table rocpd_kernel_dispatch_x (kernel_id integer, start integer, end integer)")
    con.executemany("insert into rocpd_info_kernel_symbol_x values (?, ?)", [(1, "occ_decode8_kernel"), (2, "gemm"), (3, "copy")])
    ms = 1_000_000
    rows = []
    # period of 10 ms: decoder 0-6, gemm 6-8 (alone), copy 5-7 (1 ms beside the decoder, 1 ms beside the gemm), idle 8-10
    for i in range(40):
        t = i * 10 * ms
        rows += [(1, t, t + 6 * ms), (2, t + 6 * ms, t + 8 * ms), (3, t + 5 * ms, t + 7 * ms)]
    con.executemany("insert into rocpd_kernel_dispatch_x values (?, ?, ?)", rows)
    con.commit()
    con.close()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_timeline.py"), db],
                         capture_output=True, text=True, check=True).stdout
    vals = {}
    for line in out.splitlines():
        for key in ("busy (>= 1 kernel)", "marker kernel running", "only other kernels", "idle"):
            if line.strip().startswith(key):
                vals[key] = float(line.split()[-2])
    # the window starts at a decoder start and ends at a decoder end: shares within ~1 % of 80 / 60 / 20 / 20
    assert abs(vals["busy (>= 1 kernel)"] - 80) < 1.5, out
    assert abs(vals["marker kernel running"] - 60) < 1.5, out
    assert abs(vals["only other kernels"] - 20) < 1.5, out
    assert abs(vals["idle"] - 20) < 1.5, out
    # attribution of the time without the decoder: gemm 1 ms alone + half of 1 ms shared = 15 %, copy 5 %
    att = [l for l in out.splitlines() if l.strip().endswith("gemm") or l.strip().endswith("copy")]
    assert any(abs(float(l.split()[0]) - 15) < 1 for l in att if l.strip().endswith("gemm")), out
    assert any(abs(float(l.split()[0]) - 5) < 1 for l in att if l.strip().endswith("copy")), out
