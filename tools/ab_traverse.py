#!/usr/bin/env python3
"""A/B of traversal settings in ONE process (interleaved rounds, median and min reported):
variant 0 vs variant 1 at several queries-per-wave, on the bench workloads.  GPU only."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gpu-computing-course_amd", "pyhost")]
import mi355_synth as synth, mi355cd

def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    for name, (verts, vidx) in (("cloth1M", synth.cloth_pair(500)), ("soup1M", synth.soup(1_000_000, 0.01, 1234))):
        with mi355cd.CollisionDetector(verts, vidx) as cd:
            cd.self_collide()
            settings = [(4, 64), (3, 64)]
            times = {s: [] for s in settings}; desc = {}; stats = {}
            for r in range(rounds):
                for s in settings:
                    cd.set_option(mi355cd.CD_OPT_TRAVERSAL, s[0]); cd.set_option(mi355cd.CD_OPT_QUERIES_PER_WAVE, s[1])
                    cd.find_collisions(cap=1 << 22)
                    st_ = cd.stats(); times[s].append(st_.ms_traverse); desc.setdefault(s, []).append(st_.ms_descend); stats[s] = st_
            for s in settings:
                st = stats[s]
                print(f"{name} variant={s[0]}: pairs={st.n_pairs} tested={st.pairs_tested} visits={st.node_visits} wave_steps={st.wave_steps} "
                      f"candidates={st.candidates} lane_util={st.node_visits / max(1, 64 * st.wave_steps):.3f} steps/wave={st.wave_steps / (len(vidx) / 64):.1f}")
                print(f"  variant={s[0]} qpw={s[1]:5d}  traverse median={statistics.median(times[s])*1e3:8.1f} us  min={min(times[s])*1e3:8.1f} us"
                      f"   descend median={statistics.median(desc[s])*1e3:8.1f} us  min={min(desc[s])*1e3:8.1f} us")

if __name__ == "__main__":
    main()
