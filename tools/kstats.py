import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0
for r in rows[:18]:
    print(r["Name"][:84].ljust(84), r["Calls"].rjust(6), "%8.1f us" % (float(r["AverageNs"]) / 1000), r["Percentage"])
