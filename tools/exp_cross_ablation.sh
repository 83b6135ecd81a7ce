#!/bin/bash
# Runs ON THE GPU BOX.  EXPERIMENT: what k_cross_fused's time is made of -- builds with parts of the kernel switched off (tools/ab_build.sh xsearch -DCROSS_ABLATE_PIECES: the
# searches alone; xnone -DCROSS_ABLATE_ALL: launch + upper levels into LDS).  Those builds write WRONG records: only cd_build_tree runs here, never a traversal.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/xabl; mkdir -p $O
cat > /tmp/xabl_run.py <<PY
import sys
sys.path[:0] = ["$R/gpu-computing-course_amd/pyhost"]
import mi355_synth as synth, mi355cd
v, t = synth.cloth_pair(500)
with mi355cd.CollisionDetector(v, t) as cd:
    cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0); cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 0)
    for _ in range(40): cd.build_tree()
PY
cd /tmp && export TMPDIR=/tmp
for b in "$@"; do
  if [ $b = default ]; then unset MI355CD_LIB; else export MI355CD_LIB=$R/gpu-computing-course_amd/ab/libmi355cd_$b.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -o $b -- python3 /tmp/xabl_run.py > $O/$b.out 2> $O/$b.err || exit 1
  python3 - <<PY
import csv
for r in csv.DictReader(open("$O/${b}_kernel_stats.csv")):
    n = r['Name']
    if 'k_cross_fused' in n or 'k_build_block' in n: print("$b", n[:n.find('(')], "avg_us %.1f" % (float(r['AverageNs'])/1e3))
PY
  rm -f $O/${b}_kernel_trace.csv
done
