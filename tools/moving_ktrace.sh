#!/bin/bash
# Runs ON THE GPU BOX: per-kernel averages of the step on a mesh that MOVES (tools/hint_moving.py: upload + step per frame, sheet B one quad a frame)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/kt_moving; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o run -- python3 $R/tools/hint_moving.py 40 1.0 > $O/out.log 2> $O/err.log
rm -f $O/run_kernel_trace.csv
cat $O/out.log
python3 - <<PY
import csv
tot=0
for r in csv.DictReader(open("$O/run_kernel_stats.csv")):
    n=r['Name']; n=n[:n.find('(')] if '(' in n else n
    if int(r['Calls'])>=100: print(f"   {n[:60]:60s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:8.1f}")
PY
