"""MFMA utilisation per kernel group from a rocprofv3 PMC pass of bench.py.

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d out/mfma -- python3 bench.py ...
    python pmc_mfma.py out/mfma profiles/r01_c_mfma_util.json

SQ_VALU_MFMA_BUSY_CYCLES counts cycles in which a SIMD's matrix pipe is busy, summed over the chip's 1024
SIMDs (MI355X_MICROARCH.md, cycle-constants table).  Utilisation = busy cycles / (kernel duration x clock x
1024 SIMDs), with the 2.38 GHz that a round-1 microbenchmark (s_memtime against s_memrealtime) measured inside these kernels (s_memtime / s_memrealtime);
GRBM_GUI_ACTIVE / 8 / duration is printed beside it as the counter-side estimate of the clock."""
import collections
import csv
import glob
import json
import sys

from pmc_traffic import GROUPS, match

CLOCK_HZ, SIMDS = 2.38e9, 1024


def main():
    f = (glob.glob(sys.argv[1] + "/*/*counter_collection.csv") + glob.glob(sys.argv[1] + "/*counter_collection.csv"))[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        for g, pred in GROUPS:
            if match(pred, r):
                a = agg[g]
                a[r["Counter_Name"]] += float(r["Counter_Value"])
                if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
                    a["ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                    a["launches"] += 1
                break
    out = {}
    for g, a in sorted(agg.items()):
        if not a["launches"] or not a["ns"]:
            continue
        out[g] = {"launches": int(a["launches"]), "avg_us": round(a["ns"] / a["launches"] / 1e3, 2),
                  "mfma_busy_frac": round(a["SQ_VALU_MFMA_BUSY_CYCLES"] / (a["ns"] * 1e-9 * CLOCK_HZ * SIMDS), 4),
                  "clock_ghz_from_grbm": round(a["GRBM_GUI_ACTIVE"] / 8 / a["ns"], 3) if a["GRBM_GUI_ACTIVE"] else None}
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
