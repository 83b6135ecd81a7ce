#!/usr/bin/env python3
"""EXPERIMENT (round 4): the half traversal as ONE kernel against chain kernel + item kernel (CD_OPT_SPLIT_DESCENT), and the item
kernel's chunk size (CD_OPT_ITEM_CHUNK), on the bench workloads: wall time per step (blocks of STEPS steps alternating between the
settings) and the descent's own device clock.  usage: exp_split.py [ROUNDS STEPS]   GPU only."""
import os, sys, statistics, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gpu-computing-course_amd", "pyhost")]
import numpy as np
import mi355_synth as synth, mi355cd
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
settings = [("fused", 0, 128), ("split", 1, 128)]
for name, (verts, vidx) in (("cloth1M", synth.cloth_pair(500)), ("soup1M", synth.soup(1_000_000, 0.01, 1234)), ("soup100k", synth.soup(100_000, 0.02, 1234))):
    buf = np.empty((1 << 22, 2), dtype=np.uint32)
    with mi355cd.CollisionDetector(verts, vidx) as cd:
        cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0); cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 0)
        for _ in range(30): cd.self_collide_into(buf)
        wall = {s[0]: [] for s in settings}; clk = {s[0]: [] for s in settings}; res = {}
        for r in range(rounds):
            for lab, split, chunk in (settings if r % 2 == 0 else settings[::-1]):
                cd.set_option(mi355cd.CD_OPT_SPLIT_DESCENT, split); cd.set_option(mi355cd.CD_OPT_ITEM_CHUNK, chunk)
                for _ in range(5): cd.self_collide_into(buf)
                t0 = time.perf_counter()
                for _ in range(steps): n, rc = cd.self_collide_into(buf)
                wall[lab].append((time.perf_counter() - t0) * 1e6 / steps)
                clk[lab].append(cd.fast_stats.ms_descend_clock * 1e3)
                res[lab] = (n, cd.fast_stats.pairs_tested, rc)
        print(name)
        for lab, _, _ in settings:
            print(f"   {lab:10s}: wall per step median {statistics.median(wall[lab]):7.2f} us  min {min(wall[lab]):7.2f} us | descent, device clock median {statistics.median(clk[lab]):6.1f} us | pairs {res[lab][0]} tested {res[lab][1]} rc {res[lab][2]}", flush=True)
        assert len({v for v in res.values()}) == 1, res
