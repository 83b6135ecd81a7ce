#!/usr/bin/env python3
"""What bounds the descent once the order hint is in place: the wave times the descent leaves for the hint (classes of 1.28 us, cd_debug_hint), per XCD --
load (sum / 1024 wave slots), longest wave, list scheduling in the order the kernel actually took them, against the kernel's own clock.  1 M cloth and soup.  GPU only."""
import os, sys, heapq
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gpu-computing-course_amd", "pyhost")]
import numpy as np
import mi355_synth as synth, mi355cd
def makespan(dur, slots):
    h = [0.0] * slots; heapq.heapify(h); end = 0.0
    for d in dur:
        t = heapq.heappop(h) + d; end = max(end, t); heapq.heappush(h, t)
    return end
buf = np.empty((1 << 22, 2), dtype=np.uint32)
for name, (verts, vidx) in (("cloth1M", synth.cloth_pair(500)), ("soup1M", synth.soup(1_000_000, 0.01, 1234))):
    with mi355cd.CollisionDetector(verts, vidx) as cd:
        cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0); cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 0)
        def group_times():
            """the classes the last traversal's waves left, by group of 64 sorted positions (every triangle of a group holds its wave's class)"""
            _, _, tri = cd.debug_hint(with_order=False, with_tri=True)
            _, perm = cd.export_keys()
            return tri[perm[::64]].astype(np.float64)
        for hint in (0, 1):
            cd.set_option(mi355cd.CD_OPT_ORDER_HINT, 1)
            for _ in range(10): cd.self_collide_into(buf)
            taken = None
            if not hint:
                cd.set_option(mi355cd.CD_OPT_ORDER_HINT, 0); cd.build_tree()                    # a tree without an order ...
                cd.set_option(mi355cd.CD_OPT_ORDER_HINT, 1); cd.find_collisions(cap=1 << 22)    # ... traversed in the plain order, leaving its times
                clock = cd.stats().ms_descend_clock * 1e3
            else:
                cd.self_collide_into(buf)
                _, taken = cd.debug_hint()                                                      # the order this step took
                clock = cd.fast_stats.ms_descend_clock * 1e3
            dur = (group_times() + 0.5) * 1.28
            line = f"{name} hint {'on ' if hint else 'off'}: kernel's clock {clock:5.1f} us | waves: mean {dur.mean():.1f} p50 {np.median(dur):.1f} p90 {np.percentile(dur, 90):.1f} p99 {np.percentile(dur, 99):.1f} max {dur.max():.1f} us (top class = 40+) | load sum / 8192 = {dur.sum() / 8192:.1f} us"
            if taken is not None:
                per = [makespan(dur[taken[x::8]], 1024) for x in range(8)]
                line += f" | list scheduling of these times in the order taken, per XCD on 1024 slots: {min(per):.1f} .. {max(per):.1f} us; longest-first on the SAME times: {max(makespan(np.sort(dur[taken[x::8]])[::-1], 1024) for x in range(8)):.1f}"
            print(line, flush=True)
            if taken is not None:
                # what if the K longest groups of an XCD ran as TWO waves of 32 queries each (the other 32 lanes take shared subtrees)?  A half keeps a wave's fixed part
                # (phase 0, the shared chain, hand-over: ~8 us at full occupancy) and half of the rest.
                for fixed in (6.0, 10.0):
                    out = []
                    for K in (0, 8, 32, 64, 128, 256):
                        per = []
                        for x in range(8):
                            d = np.sort(dur[taken[x::8]])[::-1]
                            halves = np.minimum(d[:K], fixed) + np.maximum(d[:K] - fixed, 0) / 2
                            per.append(makespan(np.sort(np.concatenate([halves, halves, d[K:]]))[::-1], 1024))
                        out.append(f"K={K}: {max(per):.1f}")
                    print(f"    split model (fixed part {fixed} us), longest-first list scheduling, worst XCD: " + "  ".join(out), flush=True)
