cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS --kernel-trace -d /tmp/pl -o pl --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 120 --warmup 60 --repeats 1 --child-trace --no-cpu-baseline --no-traffic --no-extras > /tmp/pl.log 2>&1)
tail -2 /tmp/pl.log | cut -c1-200
f=$(find /tmp/pl -name "*counter_collection.csv" | head -1); echo $f
python - <<PY
import csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open("$f")):
    k = r["Kernel_Name"][:40]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    if "frozen" in k or "tail" in k or "l1_gemm" in k or "dw_adam" in k:
        n = max(cnt[(k, c)] for c in d)
        print(k, {c: round(v / n) for c, v in d.items()}, "launches", n)
PY
