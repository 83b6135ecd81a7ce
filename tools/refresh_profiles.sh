#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): rocprofv3 summaries of the default bench command + HBM counters.
# Output lands in gpurun_out/prof_final/ ; tools/summarise_profiles.py turns it into profiles/<round>/.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_final; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_plain.json 2> $O/bench_plain.err
echo "plain bench done"
# (--no-extras: the default command also measures two secondary workloads with the same kernels; the trace holds the headline workload's launches only)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o run -- python3 $R/bench.py --no-extras > $O/bench_under_rocprof.json 2> $O/trace.err
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o run -- python3 $R/bench.py --no-cpu-baseline --no-ray --no-extras --no-parity --steps 20 --warmup 5 > $O/pmc_fetch.json 2> $O/pmc_fetch.err
echo "FETCH_SIZE pass done"
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $O/pmc_rdreq -o run -- python3 $R/bench.py --no-cpu-baseline --no-ray --no-extras --no-parity --steps 20 --warmup 5 > $O/pmc_rdreq.json 2> $O/pmc_rdreq.err
echo "size-resolved read request pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o run -- python3 $R/bench.py --no-cpu-baseline --no-ray --no-extras --no-parity --steps 20 --warmup 5 > $O/pmc_write.json 2> $O/pmc_write.err
echo "WRITE_SIZE pass done"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_l2 -o run -- python3 $R/bench.py --no-cpu-baseline --no-ray --no-extras --no-parity --steps 20 --warmup 5 > $O/pmc_l2.json 2> $O/pmc_l2.err
echo "L2 hit / miss pass done"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pmc_sq -o run -- python3 $R/bench.py --no-cpu-baseline --no-ray --no-extras --no-parity --steps 20 --warmup 5 > $O/pmc_sq.json 2> $O/pmc_sq.err
echo "SQ pass done"
rocprofv3 --pmc SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_SALU --output-format csv -d $O/pmc_sq2 -o run -- python3 $R/bench.py --no-cpu-baseline --no-ray --no-extras --no-parity --steps 20 --warmup 5 > $O/pmc_sq2.json 2> $O/pmc_sq2.err
echo "SQ pass 2 (scalar side, issue / wait split) done"
rm -f $O/trace/run_kernel_trace.csv        # tens of MB; the stats file is the summary
# ---- round 5: the step at 8 M triangles (BASELINE config 4's shards merged into one mesh): kernel trace + the HBM and SQ counter passes
STEPS=40 bash $R/tools/ktrace_opts.sh prof8M cfg4_8M "" > $O/8M_ktrace.log 2>&1
cp $R/gpurun_out/kt_prof8M_0/run_kernel_stats.csv $O/8M_kernel_stats.csv
echo "8 M kernel trace done"
STEPS=20 bash $R/tools/pmc_hbm.sh prof8M cfg4_8M > $O/8M_pmc.log 2>&1
cp $R/gpurun_out/pmc_hbm_prof8M/summary.csv $O/8M_pmc_per_kernel.csv
echo "8 M counter passes done"
# ---- the ray tracer's kernels (tools/rt_time.py: single frames, frames back to back, the animation loop): kernel trace + HBM counters
rocprofv3 --kernel-trace --stats --output-format csv -d $O/rt_trace -o run -- python3 $R/tools/rt_time.py > $O/rt_time_under_rocprof.log 2> $O/rt_trace.err
rm -f $O/rt_trace/run_kernel_trace.csv
for P in "fetch FETCH_SIZE" "rdreq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "write WRITE_SIZE" "l2 TCC_HIT_sum TCC_MISS_sum" "sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  set -- $P; name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $O/rt_pmc_$name -o run -- python3 $R/tools/rt_time.py > $O/rt_pmc_$name.log 2> $O/rt_pmc_$name.err || echo "rt pass $name failed"
done
python3 $R/tools/rt_time.py > $O/rt_time_plain.log 2>/dev/null
echo "ray tracer passes done"
ls -la $O $O/trace
