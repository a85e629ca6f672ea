"""Per-kernel-group averages of every counter of a rocprofv3 PMC pass of bench.py (raw values per launch).

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES \\
              SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CU_CYCLES --output-format csv -d out/sq -- python3 bench.py ...
    python pmc_sq.py out/sq profiles/r01_c_sq_counters.json

SQ_WAIT_ANY = wave parked at s_waitcnt / barrier; SQ_WAIT_INST_ANY = wave ready but its pipe is taken (for an
MFMA-bound kernel: waiting for the matrix pipe); SQ_ACTIVE_INST_ANY = issuing.  The three are disjoint shares
of SQ_WAVE_CYCLES (MI355X_MICROARCH.md, PMC slots); the json adds them as fractions."""
import collections
import csv
import glob
import json
import sys

from pmc_traffic import GROUPS, match


def main():
    f = (glob.glob(sys.argv[1] + "/*/*counter_collection.csv") + glob.glob(sys.argv[1] + "/*counter_collection.csv"))[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(collections.Counter)
    for r in csv.DictReader(open(f)):
        for g, pred in GROUPS:
            if match(pred, r):
                agg[g][r["Counter_Name"]] += float(r["Counter_Value"])
                n[g][r["Counter_Name"]] += 1
                break
    out = {}
    for g in sorted(agg):
        a = {k: agg[g][k] / n[g][k] for k in agg[g]}
        o = {k: round(v) for k, v in a.items()}
        wc = a.get("SQ_WAVE_CYCLES")
        if wc:
            for k, name in (("SQ_WAIT_ANY", "parked_frac"), ("SQ_WAIT_INST_ANY", "issue_stall_frac"), ("SQ_ACTIVE_INST_ANY", "issuing_frac")):
                if k in a:
                    o[name] = round(a[k] / wc, 3)
        if a.get("SQ_LDS_IDX_ACTIVE"):
            o["lds_conflict_frac"] = round(a.get("SQ_LDS_BANK_CONFLICT", 0.0) / a["SQ_LDS_IDX_ACTIVE"], 4)
        out[g] = o
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
