"""EXPERIMENT (needs a -DCD_ABLATE build: tools/ab_build.sh abl -DCD_ABLATE, run with MI355CD_LIB=.../ab/libmi355cd_abl.so): start / end tick and step counts of every
wave of k_descend_half (1 M cloth and soup) -> how the kernel's time divides into the bulk and the tail (the last waves running on a chip that is emptying), and what
another dispatch ORDER of the same waves could give (list scheduling of the measured durations on the chip's 8192 wave slots)."""
import ctypes as C, heapq, os, sys
sys.path[:0] = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpu-computing-course_amd", "pyhost")]
import numpy as np, mi355_synth as synth, mi355cd

def list_schedule(dur, slots=8192):
    h = [0.0] * slots
    for d in dur:
        t = heapq.heappop(h); heapq.heappush(h, t + d)
    return max(h)

for name, (v, t) in (("cloth1M", synth.cloth_pair(500)), ("soup1M", synth.soup(1_000_000, 0.01, 1234))):
    buf = np.empty((1 << 22, 2), dtype=np.uint32)
    with mi355cd.CollisionDetector(v, t) as cd:
        cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0); cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 0)
        for _ in range(5): cd.self_collide_into(buf)
        cd.set_option(103, 128 << 8)
        cd.self_collide_into(buf)
        nw = (t.shape[0] + 63) // 64
        out = np.zeros((nw, 2), dtype=np.uint32)
        lib = mi355cd.load_library()
        lib.cd_debug_wave_times.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        rc = lib.cd_debug_wave_times(cd._ctx, out.ctypes.data_as(C.c_void_p), nw)
        assert rc == 0
        s16 = (out[:, 0] & 0xffff).astype(np.int64); e16 = (out[:, 0] >> 16).astype(np.int64)
        ref = s16[0]
        st = ((s16 - ref + 32768) % 65536 - 32768); en = st + ((e16 - s16) % 65536)
        st = (st - st.min()) / 100.0; en = (en - (en - (en)).min()) / 100.0
        en = en - 0.0
        # (re-base both on the earliest start)
        base = ((s16 - ref + 32768) % 65536 - 32768).min() / 100.0
        en = en - base
        dur = en - st
        steps = (out[:, 1] & 0xff).astype(np.int64); p1 = ((out[:, 1] >> 8) & 0xff).astype(np.int64); p1a = ((out[:, 1] >> 16) & 0xff).astype(np.int64); vis = (out[:, 1] >> 24).astype(np.int64) * 4
        end_sorted = np.sort(en); total = en.max()
        print(f"{name}: kernel (first start -> last end) {total:.1f} us  device clock stat {cd.fast_stats.ms_descend_clock*1e3:.1f} us | wave duration mean {dur.mean():.1f} median {np.median(dur):.1f} p90 {np.percentile(dur,90):.1f} p99 {np.percentile(dur,99):.1f} max {dur.max():.1f} us")
        print(f"   waves finished by: 50 % at {end_sorted[nw//2]:.1f} us, 90 % at {end_sorted[int(nw*0.9)]:.1f}, 99 % at {end_sorted[int(nw*0.99)]:.1f}, all at {total:.1f} us; last wave started at {st.max():.1f} us")
        print("   in flight:", "  ".join(f"{x} us: {int(((st <= x) & (en > x)).sum())}" for x in (10, 20, 30, 40, 45, 50)))
        order = np.argsort(st, kind="stable")
        d_in_order = dur[order]
        rng = np.random.default_rng(1)
        print(f"   list scheduling of the measured durations on 8192 slots: dispatch order {list_schedule(d_in_order):.1f} us, random order {list_schedule(rng.permutation(dur)):.1f}, longest first {list_schedule(np.sort(dur)[::-1]):.1f}, "
              f"shortest first {list_schedule(np.sort(dur)):.1f}, sum / 8192 = {dur.sum()/8192:.1f} us")
        c = lambda a: np.corrcoef(a, dur)[0, 1]
        print(f"   duration vs: steps r={c(steps):.2f} (mean {steps.mean():.1f}, max {steps.max()}), phase-1 steps r={c(p1):.2f} (mean {p1.mean():.1f}), phase-1a r={c(p1a):.2f}, phase-2 steps r={c(steps-p1):.2f} (mean {(steps-p1).mean():.1f}), visits r={c(vis):.2f}, start time r={c(st):.2f}")
        first = st < 5.0
        print(f"   first round (start < 5 us): {first.sum()} waves, duration mean {dur[first].mean():.1f}; later: {(~first).sum()} waves, mean {dur[~first].mean():.1f}")
        # a predictor known before the launch?  the neighbours' durations (spatial correlation)
        g = np.arange(nw)
        print(f"   duration of group g vs group g+1: r={np.corrcoef(dur[:-1], dur[1:])[0,1]:.2f}; phase-2 steps g vs g+1: r={np.corrcoef((steps-p1)[:-1], (steps-p1)[1:])[0,1]:.2f}")
        np.savez_compressed(f"gpurun_out/wave_times_{name}.npz", st=st, en=en, steps=steps, p1=p1, p1a=p1a, vis=vis)
