#!/bin/bash
# Runs ON THE GPU BOX: scalar-side counters of the bench command + a plain bench line.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r02_base}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
echo "bench done"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $O/pmc_sq2 -o run -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/pmc_sq2.json 2> $O/pmc_sq2.err
echo "SQ2 pass done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $O/pmc_sq3 -o run -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/pmc_sq3.json 2> $O/pmc_sq3.err
echo "SQ3 pass done"
ls $O
