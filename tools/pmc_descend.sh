#!/bin/bash
# Runs ON THE GPU BOX: issue / latency / texture-path counters of the traversal kernels for the bench workload.
#   tools/pmc_descend.sh <outdir-under-gpurun_out>
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-pmc_descend}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-ray --steps 10 --warmup 3"
run() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $O/$name -o run -- $B > $O/$name.json 2> $O/$name.err || echo "pass $name failed"; echo "pass $name done"; }
run sqA SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
run sqB SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES SQ_WAIT_INST_LDS
run lat VmemLatency
run lat2 SmemLatency
run lat3 LdsLatency
run ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
run tcp2 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum
run grbm GRBM_GUI_ACTIVE GRBM_TA_BUSY
ls $O
