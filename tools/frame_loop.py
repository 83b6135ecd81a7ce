"""A per-frame loop as a simulation would run it -- cd_update_vertices (12 MB of new positions) + cd_self_collide every frame -- on the 1 M cloth with
fp32-valued and with full-double vertices, with and without the cell table (CD_OPT_CELL_TABLE): wall time per frame and its parts.  GPU only."""
import os, sys, time, statistics
sys.path[:0] = [os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpu-computing-course_amd", "pyhost")]
import numpy as np, mi355_synth as synth, mi355cd
for name, rf in (("float", True), ("double", False)):
    v, t = synth.cloth_pair(500, round_f32=rf)
    frames = [v, v + (1e-9 if not rf else 0.0)]                 # two position sets, alternating (both of the same kind)
    if rf: frames[1] = (v * 1.0).copy()
    for table in (1, 0):
        if rf and table == 0: continue
        with mi355cd.CollisionDetector(v, t) as cd, mi355cd.HostPairs(1 << 22) as hp:
            cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0); cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 0); cd.set_option(mi355cd.CD_OPT_CELL_TABLE, table)
            for _ in range(10): cd.update_vertices(frames[0]); cd.self_collide_into(hp.array)
            tu, ts = [], []
            for f in range(60):
                t0 = time.perf_counter(); cd.update_vertices(frames[f & 1]); t1 = time.perf_counter(); n, rc = cd.self_collide_into(hp.array); t2 = time.perf_counter()
                tu.append((t1 - t0) * 1e3); ts.append((t2 - t1) * 1e3)
            print(f"cloth1M {name:6s} cell table {table}: per frame upload {statistics.median(tu):.3f} ms + step {statistics.median(ts):.3f} ms = {statistics.median(tu)+statistics.median(ts):.3f} ms   pairs {n}")
