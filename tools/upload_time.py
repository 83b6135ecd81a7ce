"""cd_update_vertices timed on the bench meshes with fp32-valued and with full-double vertices: what the cell table of the vertices
(cd_bvh.h) costs per upload.  GPU only."""
import os, sys, time
sys.path[:0] = [os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpu-computing-course_amd", "pyhost")]
import numpy as np, mi355_synth as synth, mi355cd
for name, (v, t) in (("cloth1M float", synth.cloth_pair(500)), ("cloth1M double", synth.cloth_pair(500, round_f32=False)), ("soup1M float", synth.soup(1_000_000, 0.01, 1234)),
                     ("soup1M double", (synth.soup(1_000_000, 0.01, 1234)[0] * (1 + 1e-9), synth.soup(1_000_000, 0.01, 1234)[1]))):
    with mi355cd.CollisionDetector(v, t) as cd:
        ts = []
        for _ in range(8):
            t0 = time.perf_counter(); cd.update_vertices(v); ts.append((time.perf_counter() - t0) * 1e3)
        print(f"{name}: V = {v.shape[0]}  cd_update_vertices median {sorted(ts)[len(ts)//2]:.3f} ms  min {min(ts):.3f} ms")
