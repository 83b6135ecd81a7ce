#!/bin/bash
# Runs ON THE GPU BOX: issue / latency counters of the ray-tracer kernels (BASELINE config 5 frame).
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-pmc_rt}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/tools/rt_time.py"
run() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $O/$name -o run -- $B > $O/$name.json 2> $O/$name.err || echo "pass $name failed"; echo "pass $name done"; }
run sqA SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
run sqB SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_INSTS_SMEM
run lat VmemLatency
run lat2 SmemLatency
run grbm GRBM_GUI_ACTIVE
run wr WRITE_SIZE
run fe FETCH_SIZE
