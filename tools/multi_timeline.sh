#!/bin/bash
# Runs ON THE GPU BOX: kernel timeline of the last multi step of the one-rank rehearsal at config 4's scale.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-multi_tl}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O -o run -- python3 $R/tools/multi_selfpeer.py 500 slice > $O/out.log 2> $O/err.log
python3 $R/tools/print_timeline.py $O/run_kernel_trace.csv k_morton 26
rm -f $O/run_kernel_trace.csv
