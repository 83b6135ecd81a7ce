#!/bin/bash
# Runs ON THE GPU BOX.  EXPERIMENT: what k_local_sort's time is made of -- builds that return early (tools/ab_build.sh lsK -DLS_ABLATE=K).  Those builds leave the keys
# UNSORTED: only cd_morton_sort runs here, never a tree build or a traversal.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/lsabl; mkdir -p $O
cat > /tmp/lsabl_run.py <<PY
import sys
sys.path[:0] = ["$R/gpu-computing-course_amd/pyhost"]
import mi355_synth as synth, mi355cd
v, t = synth.cloth_pair(500)
with mi355cd.CollisionDetector(v, t) as cd:
    cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0); cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 0)
    for _ in range(40): cd.morton_sort()
PY
cd /tmp && export TMPDIR=/tmp
for b in "$@"; do
  if [ $b = default ]; then unset MI355CD_LIB; else export MI355CD_LIB=$R/gpu-computing-course_amd/ab/libmi355cd_$b.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -o $b -- python3 /tmp/lsabl_run.py > $O/$b.out 2> $O/$b.err || exit 1
  python3 - <<PY
import csv
for r in csv.DictReader(open("$O/${b}_kernel_stats.csv")):
    n = r['Name']
    if 'k_local_sort' in n or 'k_os_pass' in n or 'k_morton' in n: print("$b", n[:n.find('(')][:40], "calls", r['Calls'], "avg_us %.1f" % (float(r['AverageNs'])/1e3))
PY
  rm -f $O/${b}_kernel_trace.csv
done
