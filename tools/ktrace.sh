#!/bin/bash
# Runs ON THE GPU BOX: per-kernel average durations of the bench command (collision path only).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-ktrace}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o run -- python3 $R/bench.py --no-cpu-baseline --no-ray --steps 20 --warmup 5 > $O/bench.json 2> $O/err.log
rm -f $O/run_kernel_trace.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/run_kernel_stats.csv")))
tot=0
for r in rows:
    n=r['Name']; n=n[:n.find('(')] if '(' in n else n
    print(f"{n[:60]:60s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:8.1f}")
PY
