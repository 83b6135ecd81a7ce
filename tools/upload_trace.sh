#!/bin/bash
# Runs ON THE GPU BOX: kernel trace of tools/upload_time.py (what cd_update_vertices launches: the cell table's kernels).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/upload_trace; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o run -- python3 $R/tools/upload_time.py > $O/out.log 2> $O/err.log
rm -f $O/run_kernel_trace.csv
cat $O/out.log | tail -5
python3 - <<PY
import csv
for r in csv.DictReader(open("$O/run_kernel_stats.csv")):
    n=r['Name'].replace('(anonymous namespace)::',''); n=n[:n.find('(')] if '(' in n else n
    print(f"   {n[:60]:60s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms {float(r['TotalDurationNs'])/1e6:8.2f}")
PY
