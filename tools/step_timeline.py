#!/usr/bin/env python3
"""30 fused steps of the bench workload with the bench's stamp setting; run under rocprofv3 --kernel-trace and print the last step's
kernel timeline with tools/print_timeline.py (gaps between dependent kernels are what this is for).  GPU only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gpu-computing-course_amd", "pyhost")]
import mi355_synth as synth, mi355cd
v, t = synth.cloth_pair(500)
with mi355cd.CollisionDetector(v, t) as cd:
    cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0)
    cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, int(sys.argv[1]) if len(sys.argv) > 1 else 2)
    for _ in range(30): cd.self_collide(cap=1 << 22, copy=False)
