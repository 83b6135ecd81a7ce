#!/bin/bash
# Runs ON THE GPU BOX: per-kernel average durations of 200 fused steps (tools/trace_steps.py; env STEPS overrides the 200) for each option set given.
# usage: ktrace_opts.sh TAG MESH "OPTS1" "OPTS2" ...      e.g.  ktrace_opts.sh split cloth1M "7=0" "7=1 8=64"
R=$GRAFT_REPO_ROOT; export STEPS=${STEPS:-200}; TAG=$1; MESH=$2; shift 2
cd /tmp && export TMPDIR=/tmp
i=0
for OPTS in "$@"; do
  O=$R/gpurun_out/kt_${TAG}_$i; mkdir -p $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -o run -- python3 $R/tools/trace_steps.py $MESH $OPTS > $O/out.log 2> $O/err.log
  python3 - <<PY
# kernels launched more than once per step (k_os_pass: the pass without and the pass with look-back): the average of each launch of the step
import csv, collections
seq = collections.defaultdict(list)
for r in csv.DictReader(open("$O/run_kernel_trace.csv")):
    n = r['Kernel_Name']; n = n[:n.find('(')] if '(' in n else n
    seq[n].append((int(r['Start_Timestamp']), int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
for n, v in seq.items():
    if len(v) >= 2 * $STEPS and len(v) % $STEPS == 0:
        m = len(v) // $STEPS; v.sort()
        print(f"   {n[:60]:60s} per launch of the step: " + " ".join(f"{sum(d for _, d in v[i::m]) / len(v[i::m]) / 1e3:.1f}" for i in range(m)) + " us")
PY
  rm -f $O/run_kernel_trace.csv
  echo "== $MESH $OPTS : $(cat $O/out.log | tail -1)"
  python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/run_kernel_stats.csv")))
tot=0
for r in rows:
    n=r['Name']; n=n[:n.find('(')] if '(' in n else n
    if int(r['Calls']) >= $STEPS // 2:
        tot+=float(r['AverageNs'])*int(r['Calls'])/($STEPS * 1e3)
        print(f"   {n[:70]:70s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:8.1f}")
print(f"   sum per step {tot:.1f} us")
PY
  i=$((i+1))
done
