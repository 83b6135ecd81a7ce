#!/usr/bin/env python3
"""Per-kernel means of every rocprofv3 counter_collection.csv under a directory (kernel names cut at the LAST top-level
'(' so that template arguments and '(anonymous namespace)::' survive)."""
import collections, csv, glob, os, sys

def kname(full):
    depth = 0
    for i in range(len(full) - 1, -1, -1):          # strip the trailing argument list only
        ch = full[i]
        if ch == ')': depth += 1
        elif ch == '(':
            depth -= 1
            if depth == 0: return full[:i].strip()
    return full.strip()

def main():
    root = sys.argv[1]
    only = sys.argv[2:] or None
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[kname(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    names = sorted({c for cs in acc.values() for c in cs})
    print("kernel," + ",".join(names) + ",launches")
    for k in sorted(acc):
        if only and not any(o in k for o in only): continue
        n = max(len(v) for v in acc[k].values())
        print(k + "," + ",".join(f"{sum(acc[k][c]) / len(acc[k][c]):.1f}" if c in acc[k] else "" for c in names) + f",{n}")

if __name__ == "__main__":
    main()
