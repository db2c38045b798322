"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (gpurun_out/pmc_*/p_counter_collection.csv) into
profiles/r01_pmc_*_per_kernel.csv and print per-kernel HBM traffic (FETCH_SIZE doubled on gfx950, see
MI355X_MICROARCH.md, HBM section)."""
import collections
import csv

out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = list(csv.DictReader(open(f"gpurun_out/pmc_{c}/p_counter_collection.csv")))
    agg = collections.defaultdict(list)
    for r in rows:
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    with open(f"profiles/r01_pmc_{c}_per_kernel.csv", "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "dispatches", "mean_KB", "min_KB", "max_KB"])
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([k, len(v), round(sum(v) / len(v), 1), min(v), max(v)])
            out.setdefault(k, {})[c] = sum(v) / len(v)
for k, v in out.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v and v["FETCH_SIZE"] > 500:
        print(k[:70].ljust(70), "fetch", int(v["FETCH_SIZE"] * 1024 * 2), "write", int(v["WRITE_SIZE"] * 1024))
