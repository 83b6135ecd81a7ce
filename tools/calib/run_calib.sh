#!/bin/bash
# Runs ON THE GPU BOX: builds tools/calib/fetch_calib.hip and collects FETCH_SIZE and the size-resolved TCC_EA0_RDREQ counters for it
# (separate --pmc passes, no trace options), then prints bytes per kernel against the known useful bytes.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-calib}; mkdir -p $O
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o $O/fetch_calib $R/tools/calib/fetch_calib.hip
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o run -- $O/fetch_calib > $O/fetch.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $O/rdreq -o run -- $O/fetch_calib > $O/rdreq.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_32B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/dram -o run -- $O/fetch_calib > $O/dram.log 2>&1 || echo "dram pass failed"
python3 $R/tools/calib/summarise_calib.py $O
rm -f $O/fetch_calib
