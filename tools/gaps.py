"""Inter-kernel gaps from a rocprofv3 --kernel-trace CSV (gpurun_out/<dir>/p_kernel_trace.csv): idle time between
consecutive dispatches, grouped by (previous kernel -> next kernel)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
lo, hi = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5, float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
rows = rows[int(len(rows) * lo):int(len(rows) * hi)]
gaps = collections.defaultdict(list)
busy = 0
for a, b in zip(rows, rows[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    gaps[(a["Kernel_Name"][:28], b["Kernel_Name"][:28])].append(g)
    busy += int(a["End_Timestamp"]) - int(a["Start_Timestamp"])
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print("span %.1f us, busy %.1f us (%.1f%%)" % (span / 1e3, busy / 1e3, 100.0 * busy / span))
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print("%-30s -> %-30s n=%4d mean gap %6.2f us" % (k[0], k[1], len(v), sum(v) / len(v) / 1e3))
