#!/bin/bash
# Runs ON THE GPU BOX: kernel timeline of the last fused step of tools/step_timeline.py (start, end, gap to the previous kernel).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-step_tl}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O -o run -- python3 $R/tools/step_timeline.py ${2:-2} > $O/out.log 2> $O/err.log
python3 $R/tools/print_timeline.py $O/run_kernel_trace.csv k_morton 12
rm -f $O/run_kernel_trace.csv
