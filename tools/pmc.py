import csv, sys, collections
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = list(csv.DictReader(open(f"gpurun_out/pmc_{c}/p_counter_collection.csv")))
    agg = collections.defaultdict(list)
    for r in rows:
        agg[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
    print("==", c, "(KB per dispatch, mean)")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:10]:
        print("  %-62s n=%4d mean=%10.1f KB  -> %7.2f MB" % (k, len(v), sum(v) / len(v), sum(v) / len(v) / 1024))
