#!/bin/bash
# Runs ON THE GPU BOX: tools/rt_time.py plainly and under rocprofv3 --kernel-trace --stats (per-kernel averages of the ray tracer's launches).
# usage: rt_ktrace.sh TAG
R=$GRAFT_REPO_ROOT; TAG=$1
O=$R/gpurun_out/rt_kt_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/rt_time.py 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o run -- python3 $R/tools/rt_time.py > $O/out.log 2> $O/err.log
rm -f $O/run_kernel_trace.csv
python3 - <<PY
import csv
for r in csv.DictReader(open("$O/run_kernel_stats.csv")):
    print(f"   {r['Name'][:90]:90s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:8.2f}")
PY
