#!/usr/bin/env python3
"""Diagnostic counters of the half traversal (cd_set_option 103) on the two 1 M bench workloads.  GPU only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gpu-computing-course_amd", "pyhost")]
import mi355_synth as synth, mi355cd

for name, (verts, vidx) in (("cloth1M", synth.cloth_pair(500)), ("soup1M", synth.soup(1_000_000, 0.01, 1234))):
    with mi355cd.CollisionDetector(verts, vidx) as cd:
        cd.self_collide()
        cd.set_option(103, 1)
        cd.find_collisions(cap=1 << 22)
        st = cd.stats(); d = cd.debug_counters().tolist()
        waves = len(vidx) / 64
        print(f"{name}: wave_steps/wave={st.wave_steps / waves:.2f}  phase 1a steps/wave={d[5] / waves:.2f}  chain steps/wave={d[0] / waves:.2f}  "
              f"hops/query in-wave={d[1] / len(vidx):.2f} chain={d[2] / len(vidx):.2f}  descent visits/query={d[3] / len(vidx):.3f}  "
              f"descend={st.ms_descend * 1e3:.1f} us")
        print("   mean cycles per wave (s_memtime): query %.0f  phase1a %.0f  phase1b %.0f  phase2 %.0f  tail %.0f" % tuple(x / waves for x in d[6:11]))
