#!/bin/bash
# Runs ON THE GPU BOX: HBM bytes of the traversal kernel (FETCH_SIZE, WRITE_SIZE in separate passes, raw counter values).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-pmc_hbm}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $O/$C -o run -- python3 $R/bench.py --no-cpu-baseline --no-ray --steps 10 --warmup 3 > $O/$C.json 2> $O/$C.err
done
python3 - <<PY
import csv, collections
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open("$O/%s/run_counter_collection.csv" % C)):
        k = r["Kernel_Name"]; k = k[:k.find("(")] if "(" in k else k
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    for k, (v, c) in sorted(acc.items()):
        if k.startswith("cd::") or "cd::" in k: print(f"{C} {k[:50]:50s} per launch {v / c:14.0f}")
PY
