"""Per-kernel HBM traffic from two rocprofv3 counter passes (FETCH_SIZE and WRITE_SIZE cannot share a pass: TCC slots).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir_f> -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-phase-timing
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d <dir_w> -- python3 bench.py ...   (same)
    python tools/pmc_summary.py <dir_f> <dir_w> profiles/rNN/pmc_traffic.json

FETCH_SIZE / WRITE_SIZE are in KiB per dispatch.  gfx950 tallies the 128-byte requests of wide coalesced reads at 64 bytes,
so FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM / rocprofv3 section) before it is compared with byte counts."""
import collections
import csv
import glob
import json
import re
import sys


def per_kernel(d, counter):
    acc, n = collections.defaultdict(float), collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
            name = re.sub(r"^void ", "", name).split("(")[0]
            acc[name] += float(r["Counter_Value"])
            n[name] += 1
    return {k: (acc[k] / n[k], n[k]) for k in acc}


def main():
    df, dw, out = sys.argv[1:4]
    note = sys.argv[4] if len(sys.argv) > 4 else None
    f, w = per_kernel(df, "FETCH_SIZE"), per_kernel(dw, "WRITE_SIZE")
    ks = {}
    for k in sorted(set(f) | set(w)):
        e = {}
        if k in f:
            e["FETCH_SIZE_KiB_avg"], e["FETCH_SIZE_n"] = f[k]
        if k in w:
            e["WRITE_SIZE_KiB_avg"], e["WRITE_SIZE_n"] = w[k]
        if k in f and k in w:
            e["read_bytes_corrected"] = f[k][0] * 1024 * 2
            e["write_bytes"] = w[k][0] * 1024
            e["hbm_bytes_per_launch"] = e["read_bytes_corrected"] + e["write_bytes"]
            e["hbm_bytes_all_launches"] = e["read_bytes_corrected"] * f[k][1] + e["write_bytes"] * w[k][1]
        ks[k] = e
    json.dump({"note": note or "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python3 bench.py --steps 20 "
                               "--warmup 5 --no-cpu-baseline --no-phase-timing`; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 "
                               "reports 1/2 of wide coalesced reads)", "kernels": ks}, open(out, "w"), indent=1)
    for k, e in ks.items():
        if "hbm_bytes_per_launch" in e:
            print(f"{k[:60]:60s} {e['hbm_bytes_per_launch'] / 1e6:9.2f} MB/launch")


if __name__ == "__main__":
    main()
