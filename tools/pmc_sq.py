"""Mean per launch of every counter collected for one kernel over several `rocprofv3 --pmc ... --output-format csv`
passes:  python tools/pmc_sq.py DIR [kernel substring = occ_decode8_kernel]  (DIR holds one sub-directory per pass)."""
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    sub = sys.argv[2] if len(sys.argv) > 2 else "occ_decode8_kernel"
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
        print("matrix pipe busy: %.1f %% of SIMD time" % (100 * g("SQ_VALU_MFMA_BUSY_CYCLES") / (4 * g("SQ_WAVE_CYCLES") / 2)))
    for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_LDS"):
        if g(k) and g("SQ_WAVE_CYCLES"):
            print("%s / SQ_WAVE_CYCLES = %.1f %%" % (k, 100 * g(k) / g("SQ_WAVE_CYCLES")))


if __name__ == "__main__":
    main()
