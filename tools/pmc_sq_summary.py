"""Per-kernel sums of every counter in a rocprofv3 --pmc pass (any SQ_* / TCC_* set), as JSON.

    python tools/pmc_sq_summary.py <dir with *counter_collection.csv> <out.json> ["note"]

Per kernel: number of dispatches, the sum of each counter over them, and each counter divided by SQ_WAVE_CYCLES when that
counter is in the pass (share of the waves' resident cycles)."""
import collections
import csv
import glob
import json
import re
import sys


def main():
    d, out = sys.argv[1:3]
    note = sys.argv[3] if len(sys.argv) > 3 else ""
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
            name = re.sub(r"^void ", "", name).split("(")[0]
            acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[name].add(r["Dispatch_Id"])
    ks = {}
    for k, v in sorted(acc.items(), key=lambda kv: -max(kv[1].values())):
        e = {"dispatches": len(disp[k]), "sum": {c: x for c, x in sorted(v.items())}}
        wc = v.get("SQ_WAVE_CYCLES")
        if wc:
            e["per_wave_cycle"] = {c: round(x / wc, 4) for c, x in sorted(v.items()) if c != "SQ_WAVE_CYCLES"}
        ks[k] = e
    json.dump({"note": note, "kernels": ks}, open(out, "w"), indent=1)
    for k, e in list(ks.items())[:16]:
        print(f"{k[:56]:56s} n={e['dispatches']:5d}", e.get("per_wave_cycle", ""))


if __name__ == "__main__":
    main()
