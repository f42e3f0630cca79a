"""average each PMC counter over the dispatches of one kernel (substring match)

    python scripts/pmc_summary.py <root> <kernel substring>                       table on stdout
    python scripts/pmc_summary.py <root> --traffic-json out.json key=substr[,batch,pixels] ...
        HBM traffic record bench.py reads (profiles/rNN_traffic.json): per key the mean FETCH_SIZE / WRITE_SIZE (KiB)
        of the matching dispatches, bytes = 2 * FETCH_SIZE (gfx950: FETCH_SIZE reads half of a wide coalesced
        streaming read, MI355X_MICROARCH.md HBM section) + WRITE_SIZE, the launch shape the record is valid for, and
        a hash of the kernel's source files (bench.py drops the record as stale when they change)
"""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCES = ["s2anet_amd/csrc/dcn_ops.hip", "s2anet_amd/csrc/common.hpp"]      # k_dcn_patch and k_conv_f16 live here


def collect(root, pat):
    agg = collections.defaultdict(list)
    for f in sorted(glob.glob(root + "/p*/*/*_counter_collection.csv") + glob.glob(root + "/p*/*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def sha16(paths):
    h = hashlib.sha256()
    for p in paths:
        with open(os.path.join(ROOT, p), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def main():
    root = sys.argv[1]
    if sys.argv[2] != "--traffic-json":
        for k, v in collect(root, sys.argv[2]).items():
            print(f"{k:32s} n={len(v):4d} mean={sum(v)/len(v):16.1f} min={min(v):16.1f} max={max(v):16.1f}")
        return
    out, rec = sys.argv[3], {"kernels": {}}
    for spec in sys.argv[4:]:
        key, rest = spec.split("=", 1)
        parts = rest.split(",")
        agg = collect(root, parts[0])
        if not agg.get("FETCH_SIZE") or not agg.get("WRITE_SIZE"):
            print("no FETCH_SIZE / WRITE_SIZE dispatches for", parts[0], file=sys.stderr)
            continue
        fk = sum(agg["FETCH_SIZE"]) / len(agg["FETCH_SIZE"])
        wk = sum(agg["WRITE_SIZE"]) / len(agg["WRITE_SIZE"])
        rec["kernels"][key] = {"kernel": parts[0], "fetch_kib": round(fk, 1), "write_kib": round(wk, 1),
                               "bytes": round((2 * fk + wk) * 1024), "dispatches": len(agg["FETCH_SIZE"]),
                               "batch": int(parts[1]), "pixels": int(parts[2]),
                               "sources": SOURCES, "source_sha16": sha16(SOURCES)}
    rec["command"] = "rocprofv3 --kernel-trace --pmc <FETCH_SIZE | WRITE_SIZE> -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline"
    rec["git_head"] = os.environ.get("S2A_GIT_HEAD")       # filled in when the record is copied into profiles/ (no .git on the GPU box)
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
