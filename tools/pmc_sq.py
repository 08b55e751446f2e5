"""Mean per launch of every counter collected for one kernel over several `rocprofv3 --pmc ... --output-format csv`
passes:  python tools/pmc_sq.py DIR [kernel substring = occ_decode8_kernel] [--json MODE SOURCE]
(DIR holds one sub-directory per pass).  --json f16x3|f16x1 "<source file>": also record the matrix-pipe-busy share in
profiles/decoder_mfma_busy.json, which bench.py quotes as `roofline.mfma_busy` (counters cannot be read in-process)."""
import json
import csv
import glob
import os
import sys


def main():
    argv = list(sys.argv)
    js = None
    if "--json" in argv:
        i = argv.index("--json")
        js = (argv[i + 1], argv[i + 2])
        del argv[i:i + 3]
    d = argv[1]
    sub = argv[2] if len(argv) > 2 else "occ_decode8_kernel"
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if sub in row.get("Kernel_Name", ""):
                    a = acc.setdefault(row["Counter_Name"], [0.0, set()])
                    a[0] += float(row["Counter_Value"])
                    a[1].add((f, row.get("Dispatch_Id")))
    for k in sorted(acc):
        tot, disp = acc[k]
        print("%-28s %.4e per launch (%d launches)" % (k, tot / max(len(disp), 1), len(disp)))
    g = lambda k: acc[k][0] / max(len(acc[k][1]), 1) if k in acc else None
    if g("SQ_WAVE_CYCLES") and g("SQ_VALU_MFMA_BUSY_CYCLES"):
        # SQ_WAVE_CYCLES counts per wave; two waves share a SIMD, MFMA_BUSY counts per SIMD x4 (round 3's convention)
        busy = g("SQ_VALU_MFMA_BUSY_CYCLES") / (4 * g("SQ_WAVE_CYCLES") / 2)
        print("matrix pipe busy: %.1f %% of SIMD time" % (100 * busy))
        if js:
            path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "decoder_mfma_busy.json")
            try:
                doc = json.load(open(path))
            except (OSError, ValueError):
                doc = {}
            doc[js[0]] = {"mfma_busy": round(busy, 4), "SQ_VALU_MFMA_BUSY_CYCLES": g("SQ_VALU_MFMA_BUSY_CYCLES"),
                          "SQ_WAVE_CYCLES": g("SQ_WAVE_CYCLES"), "kernel": sub}
            doc["source"] = js[1]
            json.dump(doc, open(path, "w"), indent=1)
    for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_LDS"):
        if g(k) and g("SQ_WAVE_CYCLES"):
            print("%s / SQ_WAVE_CYCLES = %.1f %%" % (k, 100 * g(k) / g("SQ_WAVE_CYCLES")))


if __name__ == "__main__":
    main()
