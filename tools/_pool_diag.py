import os, sys
sys.path[:0] = ["/root/repo/gpu-computing-course_amd/pyhost"]
os.environ["CD_POOL_DIAG"] = "1"
import numpy as np, mi355_synth as synth, mi355cd
verts, vidx = synth.cloth_pair(500)
buf = np.empty((1 << 22, 2), dtype=np.uint32)
with mi355cd.CollisionDetector(verts, vidx) as cd:
    cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0); cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 0)
    for cons in (512, 0, 2048):
        cd.debug_set(mi355cd.CD_DBG_POOL_CONSUMERS, cons)
        for _ in range(20): cd.self_collide_into(buf)
        for rep in range(2):
            cd.self_collide_into(buf)
            d = cd.debug_counters()
            khz = float(d[10]); us = lambda x: x / khz * 1e3
            print(f"consumers {cons}: producers end {us(d[0]):.1f}  first consumer start {us(d[1]):.1f}  last consumer start {us(d[6]):.1f}  first batch {us(d[5]):.1f}  first final {us(d[7]):.1f} last final {us(d[8]):.1f}  last consumer end {us(d[2]):.1f} us | batches {d[3]}  mean batch {us(d[4])/max(1,d[3]):.2f} us  most batches of one consumer {d[9]}")
