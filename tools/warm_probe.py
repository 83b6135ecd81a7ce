"""Why the driver's 25-step command reads ~3 % above the 200-step default: per-step wall times of the headline step on a fresh context, after idle gaps, after 1500 steps, and on a
second context on a busy chip.  GPU only.  Result: profiles/r06_experiments/first_steps.log."""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [os.path.join(ROOT, "gpu-computing-course_amd", "pyhost")]
import numpy as np, mi355cd, mi355_synth as synth
v, t = synth.cloth_pair(500)
buf = np.empty((1 << 22, 2), dtype=np.uint32)
def run(cd, k):
    out = []
    for _ in range(k):
        t0 = time.perf_counter(); cd.self_collide_into(buf); out.append((time.perf_counter() - t0) * 1e3)
    return out
with mi355cd.CollisionDetector(v, t) as cd:
    cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0); cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 0)
    a = run(cd, 30); print("fresh context, 30 steps:      ", " ".join(f"{x:.4f}" for x in a[:10]), "... mean of last 10", f"{np.mean(a[-10:]):.4f}")
    time.sleep(3.0)
    b = run(cd, 30); print("after 3 s idle, 30 steps:      ", " ".join(f"{x:.4f}" for x in b[:10]), "... mean of last 10", f"{np.mean(b[-10:]):.4f}")
    run(cd, 1500)
    c = run(cd, 30); print("after 1500 steps, 30 steps:    ", " ".join(f"{x:.4f}" for x in c[:10]), "... mean of last 10", f"{np.mean(c[-10:]):.4f}")
    time.sleep(0.3)
    d = run(cd, 30); print("after 0.3 s idle, 30 steps:    ", " ".join(f"{x:.4f}" for x in d[:10]), "... mean of last 10", f"{np.mean(d[-10:]):.4f}")
# a second context right after (new allocations, new graph), chip warm
with mi355cd.CollisionDetector(v, t) as cd:
    cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0); cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 0)
    e = run(cd, 30); print("second context, chip warm:     ", " ".join(f"{x:.4f}" for x in e[:10]), "... mean of last 10", f"{np.mean(e[-10:]):.4f}")
    desc = []
    for _ in range(200): cd.self_collide_into(buf); desc.append(cd.fast_stats.ms_descend_clock * 1e3)
    print("descent clock, steps 31..230 of that context: first 5", [round(x, 1) for x in desc[:5]], "last 5", [round(x, 1) for x in desc[-5:]])
