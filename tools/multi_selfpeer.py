#!/usr/bin/env python3
"""cd_multi_step on a one-rank communicator in self-peer mode (every phase runs): per-phase times.  GPU only."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gpu-computing-course_amd", "pyhost")]
import mi355_synth as synth, mi355cd
quads = int(sys.argv[1]) if len(sys.argv) > 1 else 500
SERIAL = 8 if "serial" in sys.argv[2:] else 0                                                     # CD_MULTI_CROSS_SERIAL
SLICE = SERIAL | mi355cd.CD_MULTI_SELF_SLICE if "slice" in sys.argv[2:] else SERIAL      # exchange a tenth of the triangles (config 4's overlap)
if "priority" in sys.argv[2:]:
    SLICE |= mi355cd.CD_MULTI_PRIORITY_STREAM
v, t, ids, vb = synth.cloth_shard(0, quads)
with mi355cd.CollisionDetector(v, t, ids) as cd:
    cd.set_morton_frame(mi355cd.CD_FRAME_AUTO)
    cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0)
    with mi355cd.MultiStep(cd, mi355cd.multi_unique_id(), 0, 1, flags=mi355cd.CD_MULTI_SELF_PEER | mi355cd.CD_MULTI_TIMING | SLICE) as ms:
        for it in range(6):
            t0 = time.perf_counter()
            pairs, n, rc, mi = ms.step(1 << 22)
            dt = time.perf_counter() - t0
            print(f"step {it}: wall {dt*1e3:.3f} ms rc {rc} syncs {mi.host_syncs} attempts {mi.attempts} sent {mi.sent_queries} local {mi.local_pairs} cross {mi.cross_pairs} | tree {mi.ms_tree*1e3:.0f} ag {mi.ms_allgather*1e3:.0f} "
                  f"pack {mi.ms_pack*1e3:.0f} counts {mi.ms_counts*1e3:.0f} xch {mi.ms_exchange*1e3:.0f} local {mi.ms_local*1e3:.0f} cross {mi.ms_cross*1e3:.0f} us")
        ms.set_flags(mi355cd.CD_MULTI_SELF_PEER | SLICE)
        K = 30
        t0 = time.perf_counter()
        for _ in range(K):
            ms.step(1 << 22)
        print(f"untimed (no phase events): {(time.perf_counter() - t0) / K * 1e3:.3f} ms per step")
