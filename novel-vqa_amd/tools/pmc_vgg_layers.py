"""Per-LAUNCH table of the extractor's kernels (one row per convolution layer of one forward) from rocprofv3 passes of
tools/bench_vgg.py:   pmc_vgg_layers.py <trace dir> <mfma dir> <sq dir> <fetch dir> <write dir> out.json
Rows are matched across passes by dispatch order within the LAST forward of each pass (every pass runs the same program)."""
import csv
import glob
import json
import sys

CLK = 2.4e9  # MI355X_MICROARCH.md: engine clock used for the busy fractions


def rows(d, suffix):
    f = (glob.glob(d + "/*/*" + suffix) + glob.glob(d + "/*" + suffix))[0]
    return list(csv.DictReader(open(f)))


def short(n):
    n = n.split("(")[0]
    for a, b in (("void nvqa::", ""), ("(anonymous namespace)::", ""), ("nvqa::", "")):
        n = n.replace(a, b)
    return n[:90]


def last_forward(seq, key):
    """seq: dispatch-ordered records; the last forward starts at the last k_nchw_to_nhwc4"""
    start = max(i for i, r in enumerate(seq) if "k_nchw_to_nhwc4" in key(r))
    return seq[start:]


def main():
    tr = sorted(rows(sys.argv[1], "kernel_trace.csv"), key=lambda r: int(r["Start_Timestamp"]))
    tr = last_forward(tr, lambda r: r["Kernel_Name"])
    out = [{"kernel": short(r["Kernel_Name"]), "grid": int(r["Grid_Size"]) if "Grid_Size" in r else None,
            "us": round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1)} for r in tr]
    for d in sys.argv[2:6]:
        by = {}
        for r in rows(d, "counter_collection.csv"):
            by.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"]})[r["Counter_Name"]] = float(r["Counter_Value"])
        seq = last_forward([by[k] for k in sorted(by)], lambda r: r["name"])
        if len(seq) != len(out):
            print("pass", d, "has", len(seq), "launches, trace has", len(out), file=sys.stderr)
            continue
        for o, c in zip(out, seq):
            o.update({k: v for k, v in c.items() if k != "name"})
    for o in out:
        wc = o.get("SQ_WAVE_CYCLES")
        if wc:
            o["parked"] = round(o.get("SQ_WAIT_ANY", 0) / wc, 3)
            o["issue_stall"] = round(o.get("SQ_WAIT_INST_ANY", 0) / wc, 3)
            o["issuing"] = round(o.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3)
        if o.get("SQ_LDS_IDX_ACTIVE"):
            o["lds_conflict"] = round(o.get("SQ_LDS_BANK_CONFLICT", 0) / o["SQ_LDS_IDX_ACTIVE"], 4)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in o:
            o["mfma_busy"] = round(o["SQ_VALU_MFMA_BUSY_CYCLES"] / (o["us"] * 1e-6 * CLK * 1024), 4)
        if "FETCH_SIZE" in o or "WRITE_SIZE" in o:
            o["hbm_bytes"] = round((2 * o.get("FETCH_SIZE", 0) + o.get("WRITE_SIZE", 0)) * 1024)
        for k in list(o):
            if k.startswith("SQ_") or k.startswith("GRBM") or k in ("FETCH_SIZE", "WRITE_SIZE"):
                del o[k]
    json.dump(out, open(sys.argv[6], "w"), indent=1)
    for o in out:
        print(o)


if __name__ == "__main__":
    main()
