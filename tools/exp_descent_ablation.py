"""TIMING EXPERIMENT (results are wrong by construction; needs a build with -DCD_ABLATE: tools/ab_build.sh abl -DCD_ABLATE, run with MI355CD_LIB=.../ab/libmi355cd_abl.so): k_descend_half with parts switched off (debug key 103, bits 8..) --
what the kernel's time is made of.  1 no descent (phase 2), 2 no shared chain (1b), 4 no in-wave hops (1a), 8 no candidate hand-over, 16 no record loads for the query box."""
import os, sys, statistics
sys.path[:0] = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpu-computing-course_amd", "pyhost")]
import numpy as np, mi355_synth as synth, mi355cd
v, t = synth.cloth_pair(500)
buf = np.empty((1 << 22, 2), dtype=np.uint32)
with mi355cd.CollisionDetector(v, t) as cd:
    cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0); cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 0)
    for rep in range(2):
        for mask in (0, 1, 2, 4, 8, 3, 7, 15, 31, 63, 32, 0):
            cd.set_option(103, mask << 8)
            xs = []
            for _ in range(30):
                cd.self_collide_into(buf); xs.append(cd.fast_stats.ms_descend_clock * 1e3)
            print(f"ablate {mask:2d}: descent (device clock) median {statistics.median(xs[5:]):6.1f} us  visits {cd.fast_stats.node_visits}  steps/wave {cd.fast_stats.wave_steps / 15625:.1f}  candidates {cd.fast_stats.candidates}")
