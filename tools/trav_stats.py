import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpu-computing-course_amd", "pyhost"))
import mi355cd, mi355_synth as synth
v, t = synth.cloth_pair(500)
with mi355cd.CollisionDetector(v, t) as cd:
    for _ in range(3): cd.self_collide()
    s = cd.stats()
    print({k: getattr(s, k) for k, _ in s._fields_})
    print("lanes/step", s.node_visits / max(1, s.wave_steps) )
