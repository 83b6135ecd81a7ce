#!/bin/bash
# Runs ON THE GPU BOX: per-kernel averages of cd_build_tree (tools/trace_build.py) for builds made by tools/ab_build.sh.   usage: ab_build_trace.sh "NAME ..." "MESH ..." [filter]
R=$GRAFT_REPO_ROOT; export STEPS=${STEPS:-100}
cd /tmp && export TMPDIR=/tmp
for M in $2; do for N in $1; do
  O=$R/gpurun_out/bt_${N}_$M; mkdir -p $O
  MI355CD_LIB=$R/gpu-computing-course_amd/ab/libmi355cd_$N.so rocprofv3 --kernel-trace --stats --output-format csv -d $O -o run -- python3 $R/tools/trace_build.py $M > $O/out.log 2> $O/err.log
  rm -f $O/run_kernel_trace.csv
  echo "#### build $N mesh $M"
  python3 - <<PY | grep -E "${3:-.}"
import csv
for r in csv.DictReader(open("$O/run_kernel_stats.csv")):
    n=r['Name']; n=n[:n.find('(')] if '(' in n else n
    if int(r['Calls']) >= $STEPS // 2: print(f"   {n[:70]:70s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:8.1f}")
PY
done; done
