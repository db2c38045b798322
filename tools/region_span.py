"""GPU-side span of the timed regions of `bench.py --steps 20 --warmup 5 --child-trace` from a rocprofv3 kernel trace: first kernel start ->
last kernel end per region (regions = runs of kernels separated by > 60 us of idle), the idle inside a region, and the kernels around each gap
> 4 us.  usage: python tools/region_span.py <kernel_trace.csv>"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]))
rows.sort()
regions, cur = [], [rows[0]]
for a in rows[1:]:
    if a[0] - max(x[1] for x in cur[-8:]) > 60000:
        regions.append(cur); cur = []
    cur.append(a)
regions.append(cur)
for i, reg in enumerate(regions):
    if len(reg) < 40:
        continue
    span = (max(x[1] for x in reg) - reg[0][0]) / 1e3
    busy = sum(x[1] - x[0] for x in reg) / 1e3
    gaps = []
    end = reg[0][1]
    for k in range(1, len(reg)):
        g = (reg[k][0] - end) / 1e3
        if g > 4.0:
            gaps.append((round(g, 1), reg[k - 1][2], reg[k][2]))
        end = max(end, reg[k][1])
    print(f"region {i}: {len(reg)} kernels, span {span:.1f} us, kernel time {busy:.1f} us, idle {span - busy:.1f} us ({(span - busy) / len(reg):.2f} per kernel); gaps > 4 us: {gaps[:6]}")
