#!/usr/bin/env python3
"""20 fused steps of the 8 M-triangle soup (run under rocprofv3 --kernel-trace --stats: per-kernel table at 8 M).  GPU only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gpu-computing-course_amd", "pyhost")]
import mi355_synth as synth, mi355cd
v, t = synth.soup(8_000_000, 0.005, 1234)
with mi355cd.CollisionDetector(v, t) as cd:
    cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0)
    for _ in range(20): pairs, n, rc = cd.self_collide(cap=1 << 23, copy=False)
    st = cd.stats()
    print(f"8 M soup: device {st.ms_pipeline:.3f} ms per step, pairs {n}, tested {st.pairs_tested}")
