#!/bin/bash
# Runs ON THE GPU BOX: the two SQ passes (instruction mix, issue / wait cycles) for the bench workload, all kernels.
#   tools/pmc_quick.sh <outdir-under-gpurun_out>
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-pmc_quick}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-ray --steps 10 --warmup 3"
run() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $O/$name -o run -- $B > $O/$name.json 2> $O/$name.err || echo "pass $name failed"; echo "pass $name done"; }
run sqA SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
run sqB SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES SQ_WAIT_INST_LDS
python3 $R/tools/pmc_summary.py $O > $O/summary.csv
