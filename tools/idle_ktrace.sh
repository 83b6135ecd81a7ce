#!/bin/bash
# Runs ON THE GPU BOX: per-kernel averages of the static 1 M step back to back, with an idle host between steps, and with an upload of the SAME vertices between steps
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for ARGS in "0" "15" "15 1" "0 1"; do
  O=$R/gpurun_out/kt_idle_$(echo $ARGS | tr ' ' '_'); mkdir -p $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -o run -- python3 $R/tools/idle_steps.py $ARGS > $O/out.log 2> $O/err.log
  rm -f $O/run_kernel_trace.csv
  echo "== $(tail -1 $O/out.log)"
  python3 - <<PY
import csv
for r in csv.DictReader(open("$O/run_kernel_stats.csv")):
    n=r['Name']; n=n[:n.find('(')] if '(' in n else n
    if int(r['Calls'])>=100 and 'cd::' in n: print(f"   {n[:50]:50s} {float(r['AverageNs'])/1e3:8.1f}")
PY
done
