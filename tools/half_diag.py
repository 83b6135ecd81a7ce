"""Diagnostics of the half traversal (the descent kernel's DIAG instance, CD_DBG_DIAG): steps and s_memtime ticks per phase, lane use,
candidates, for the bench workloads.  usage: half_diag.py [MESH ...]   GPU only."""
import os, sys
sys.path[:0] = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpu-computing-course_amd", "pyhost")]
import numpy as np, mi355_synth as synth, mi355cd
meshes = {"cloth1M": lambda: synth.cloth_pair(500), "cloth1Md": lambda: synth.cloth_pair(500, round_f32=False), "soup1M": lambda: synth.soup(1_000_000, 0.01, 1234)}
for name in (sys.argv[1:] or list(meshes)):
    v, t = meshes[name]()
    buf = np.empty((1 << 22, 2), dtype=np.uint32)
    with mi355cd.CollisionDetector(v, t) as cd:
        cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0); cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 0)
        for _ in range(5): cd.self_collide_into(buf)
        st = cd.stats()
        print(f"{name}: pairs {st.n_pairs} tested {st.pairs_tested} node visits {st.node_visits} wave-steps {st.wave_steps} candidates to k_exact {st.candidates} descent clock {st.ms_descend_clock*1e3:.1f} us")
        cd.debug_set(mi355cd.CD_DBG_DIAG, 1)
        cd.self_collide_into(buf)
        c = cd.debug_counters()
        nw = (t.shape[0] + 63) // 64
        tot = sum(c[6:11])
        print(f"   per wave: phase-1a steps {c[5]/nw:.1f} 1b steps {c[0]/nw:.1f} | lane hops in wave {c[1]/t.shape[0]:.2f}/q above {c[2]/t.shape[0]:.2f}/q phase-2 visits {c[3]/t.shape[0]:.2f}/q"
              f" | ticks: phase 0 {100*c[6]/tot:.0f} % 1a {100*c[7]/tot:.0f} % 1b {100*c[8]/tot:.0f} % phase 2 {100*c[9]/tot:.0f} % hand-over {100*c[10]/tot:.0f} %  (sum {tot/nw:.0f} ticks per wave)")
