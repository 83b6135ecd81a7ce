#!/bin/bash
# Runs ON THE GPU BOX: SQ issue / wait counters of the kernels of 200 fused steps (tools/trace_steps.py) with the options given.
# usage: pmc_opts.sh TAG MESH "OPTS" [kernel name filter ...]
R=$GRAFT_REPO_ROOT; TAG=$1; MESH=$2; OPTS=$3; shift 3
O=$R/gpurun_out/pmc_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $O/$name -o run -- python3 $R/tools/trace_steps.py $MESH $OPTS > $O/$name.log 2> $O/$name.err || echo "pass $name failed"; }
run sqA SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
run sqB SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES SQ_WAIT_INST_LDS
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS
python3 $R/tools/pmc_summary.py $O "$@" > $O/summary.csv
python3 - <<PY
import csv
rows = list(csv.reader(open("$O/summary.csv")))
hdr = rows[0]
for r in rows[1:]:
    extra = len(r) - len(hdr)                      # (a kernel name with commas in its template arguments)
    name = ",".join(r[:1 + extra]); vals = r[1 + extra:]
    print(name[:60]); print("   " + "  ".join(f"{k}={float(v)/1e6:.2f}M" for k, v in zip(hdr[1:], vals) if k != "launches" and v))
PY
find $O -name "*counter_collection.csv" -delete
