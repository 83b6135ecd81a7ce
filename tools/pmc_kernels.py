"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel name, mean of each counter per launch."""
import csv, sys, collections
path = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(path)):
    acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    print(k[:60], {c: round(sum(v) / len(v), 1) for c, v in sorted(cs.items())}, "launches", len(next(iter(cs.values()))))
