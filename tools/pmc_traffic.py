"""HBM traffic of the decoder launches INSIDE the benchmark from two rocprofv3 PMC passes
(FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950, MI355X_MICROARCH.md "rocprofv3 PMC slots"):

  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d DIR/fetch -- python bench.py ...
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d DIR/write -- python bench.py ...
  python tools/pmc_traffic.py DIR N_QUERY_POINTS_OF_THOSE_LAUNCHES > profiles/rNN_decoder_traffic.txt

Corrections as the guide prescribes: FETCH_SIZE is in KB and reports half the bytes of wide (16 B per
lane) streaming reads on gfx950 -> x2 (an upper bound here: not every read of this kernel is a wide
stream); WRITE_SIZE in KB, uncalibrated.  Also writes profiles/decoder_traffic.json, which bench.py
reads for `roofline.traffic` (the counters cannot be read from inside the benchmarked process)."""
import csv
import glob
import json
import os
import sys


def collect(d, counter, kernel_substr):
    tot, n = 0.0, 0
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") == counter and kernel_substr in row.get("Kernel_Name", ""):
                    tot += float(row["Counter_Value"])
                    n += 1
    return tot, n


def collect_rows(d, counter, kernel_substr):
    """per-dispatch values in dispatch order"""
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") == counter and kernel_substr in row.get("Kernel_Name", ""):
                    rows.append((int(row.get("Dispatch_Id", len(rows))), float(row["Counter_Value"])))
    return [v for _, v in sorted(rows)]


def group_main():
    """python tools/pmc_traffic.py --group DIR TAG: tools/group_only.py under the two PMC passes (5 launches at B = 32,
    then 5 at B = 8) -> profiles/group_traffic.json {"32": bytes per launch, "8": ...}"""
    d, tag = sys.argv[2], sys.argv[3]
    f = collect_rows(os.path.join(d, "gfetch"), "FETCH_SIZE", "group_lds_kernel")
    w = collect_rows(os.path.join(d, "gwrite"), "WRITE_SIZE", "group_lds_kernel")
    assert len(f) == 10 and len(w) == 10, (len(f), len(w))
    out = {}
    for B, sl in ((32, slice(1, 5)), (8, slice(6, 10))):          # first launch of each size dropped (cold)
        fb = sum(f[sl]) / 4 * 1024 * 2                             # KB, gfx950 x2 correction for 16-B/lane streams
        wb = sum(w[sl]) / 4 * 1024
        alg = B * ((3 + 128) * 1024 * 32 * 4 + 1024 * 32 * 4 + 128 * 2048 * 4 + 2048 * 12 + 1024 * 12)
        out[str(B)] = {"bytes_per_launch": fb + wb, "fetch_bytes": fb, "write_bytes": wb, "algorithmic_bytes": alg}
        print("group_lds_kernel B=%2d: FETCH_SIZE %.1f MB (x2 corrected) + WRITE_SIZE %.1f MB = %.1f MB per launch; "
              "algorithmic %.1f MB (ratio %.2f)" % (B, fb / 1e6, wb / 1e6, (fb + wb) / 1e6, alg / 1e6, (fb + wb) / alg))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out["source"] = ("profiles/%s_group_pmc.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over "
                     "tools/group_only.py; FETCH x2 gfx950 correction)" % tag)
    with open(os.path.join(root, "profiles", "group_traffic.json"), "w") as fh:
        json.dump(out, fh)


def main():
    if sys.argv[1] == "--group":
        return group_main()
    first = None
    if "--first" in sys.argv:                 # only the first K decoder launches in dispatch order: the SCENES' launches
        i = sys.argv.index("--first")         # (bench.py's self-check and "alone" launches follow the timed region)
        first = int(sys.argv[i + 1])
        del sys.argv[i:i + 2]
    d, n_points = sys.argv[1], float(sys.argv[2])
    tag = sys.argv[3] if len(sys.argv) > 3 else "r02"
    # the scenes' launches: the main kernel (occ_decode8_kernel) and, for rounds of <= 384 tiles, the tail kernel
    # (occ_decode_tail_kernel) -- both match; the four-wave kernel (occ_decode_kernel) is not launched by bench.py
    kern = "occ_decode"
    if first:
        f_rows = collect_rows(os.path.join(d, "fetch"), "FETCH_SIZE", kern)[:first]
        w_rows = collect_rows(os.path.join(d, "write"), "WRITE_SIZE", kern)[:first]
        fetch_kb, nf, write_kb, nw = sum(f_rows), len(f_rows), sum(w_rows), len(w_rows)
    else:
        fetch_kb, nf = collect(os.path.join(d, "fetch"), "FETCH_SIZE", kern)
        write_kb, nw = collect(os.path.join(d, "write"), "WRITE_SIZE", kern)
    fetch_b = fetch_kb * 1024 * 2          # gfx950: wide streaming reads are tallied at half their size
    write_b = write_kb * 1024
    bpq = (fetch_b + write_b) / n_points
    print("decoder launches: %d (fetch pass) / %d (write pass); %.0f query points in total" % (nf, nw, n_points))
    print("FETCH_SIZE sum %.0f KB -> %.1f MB after the gfx950 x2 correction for 16-B/lane streaming reads" % (fetch_kb, fetch_b / 1e6))
    print("WRITE_SIZE sum %.0f KB =  %.1f MB  (logits: %.1f MB + tile padding)" % (write_kb, write_b / 1e6, n_points * 4 / 1e6))
    print("total %.1f MB / %.2f M points = %.1f B per query point (algorithmic 16 B: 12 in, 4 out; the shared "
          "round-0 lattice makes the real input ~3 B)" % ((fetch_b + write_b) / 1e6, n_points / 1e6, bpq))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "profiles", "decoder_traffic.json"), "w") as fh:
        json.dump({"bytes_per_query": round(bpq, 2),
                   "source": "profiles/%s_decoder_traffic.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes "
                             "over the decoder launches of bench.py; FETCH x2 gfx950 correction)" % tag}, fh)


if __name__ == "__main__":
    main()
