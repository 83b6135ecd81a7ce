#!/bin/bash
# Runs ON THE GPU BOX: HBM-side counters (separate --pmc passes, MI355X_MICROARCH.md) + the SQ issue / wait split of every kernel of the
# fused step on one of tools/trace_steps.py's meshes; per-kernel means -> gpurun_out/pmc_hbm_TAG/summary.csv
# usage: pmc_hbm.sh TAG MESH [OPTS]
R=$GRAFT_REPO_ROOT; TAG=$1; MESH=$2; OPTS=$3
export STEPS=${STEPS:-20}
O=$R/gpurun_out/pmc_hbm_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $O/$name -o run -- python3 $R/tools/trace_steps.py $MESH $OPTS > $O/$name.log 2> $O/$name.err || echo "pass $name failed"; echo "pass $name done"; }
run fetch FETCH_SIZE
run rdreq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
run write WRITE_SIZE
run l2 TCC_HIT_sum TCC_MISS_sum
run sqA SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run sqB SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS
python3 $R/tools/pmc_summary.py $O > $O/summary.csv
find $O -name "*counter_collection.csv" -delete
find $O -name "*.db" -delete
cat $O/summary.csv
