#!/bin/bash
# Runs ON THE GPU BOX: per-kernel average durations of the fused step on the 1 M soup (bench.py --soup's workload), 60 steps.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ktrace_soup; mkdir -p $O
cat > /tmp/soup_run.py <<PY
import sys
sys.path[:0] = ["$R/gpu-computing-course_amd/pyhost"]
import numpy as np, mi355_synth as synth, mi355cd
v, t = synth.soup(1_000_000, 0.01, 1234)
with mi355cd.CollisionDetector(v, t) as cd, mi355cd.HostPairs(1 << 22) as hp:
    cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0); cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 0)
    for _ in range(60): cd.self_collide_into(hp.array)
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o run -- python3 /tmp/soup_run.py > $O/out.log 2> $O/err.log
rm -f $O/run_kernel_trace.csv
cp $O/run_kernel_stats.csv $R/gpurun_out/soup_1M_kernel_stats.csv
python3 - <<PY
import csv
for r in csv.DictReader(open("$O/run_kernel_stats.csv")):
    n=r['Name']; n=n[:n.find('(')] if '(' in n else n
    print(f"{n[:50]:50s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:8.1f}")
PY
