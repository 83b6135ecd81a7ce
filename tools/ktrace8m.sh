#!/bin/bash
# Runs ON THE GPU BOX: per-kernel average durations of 20 fused steps on the 8 M-triangle soup.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-ktrace8m}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o run -- python3 $R/tools/trace8m.py > $O/out.log 2> $O/err.log
rm -f $O/run_kernel_trace.csv
grep "8 M soup" $O/out.log
python3 - <<PY
import csv
for r in csv.DictReader(open("$O/run_kernel_stats.csv")):
    n=r['Name']; n=n[:n.find('(')] if '(' in n else n
    print(f"{n[:60]:60s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:8.1f}")
PY
