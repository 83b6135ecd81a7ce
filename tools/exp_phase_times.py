import os, sys
sys.path[:0] = ["/root/repo/gpu-computing-course_amd/pyhost"]
import numpy as np, mi355_synth as synth, mi355cd
for name, (v, t) in (("cloth1M", synth.cloth_pair(500)), ("soup1M", synth.soup(1_000_000, 0.01, 1234))):
    buf = np.empty((1 << 22, 2), dtype=np.uint32)
    with mi355cd.CollisionDetector(v, t) as cd:
        cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0); cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 0)
        for _ in range(5): cd.self_collide_into(buf)
        cd.set_option(103, 1)
        cd.self_collide_into(buf)
        c = cd.debug_counters().astype(np.float64)
        nw = (t.shape[0] + 63) // 64
        # s_memtime ticks at the shader clock? print raw per wave
        print(name, "per wave: p1b steps %.1f  p1a steps %.1f  hops_in/q %.2f hops_out/q %.2f vis/q %.2f" % (c[0]/nw, c[5]/nw, c[1]/t.shape[0], c[2]/t.shape[0], c[3]/t.shape[0]))
        tot = c[6:11].sum()
        print("   s_memtime ticks per wave: phase0 %.0f  1a %.0f  1b %.0f  phase2 %.0f  epilogue %.0f  total %.0f ; descent clock %.1f us" % (*(c[6:11]/nw), tot/nw, cd.fast_stats.ms_descend_clock*1e3))
