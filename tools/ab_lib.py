"""A/B of builds of libmi355cd.so on one box (tools/ab_build.sh NAME FLAGS -> gpu-computing-course_amd/ab/libmi355cd_NAME.so):
usage: ab_lib.py NAME1 NAME2 ...  ('default' = the shipped build).  Every build runs in its own process (MI355CD_LIB picks it),
ROUNDS times round-robin; per build the medians over all rounds of: descent (device clock), device pipeline, wall per step."""
import os, subprocess, sys, statistics, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "gpu-computing-course_amd")
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import time
    sys.path.insert(0, os.environ.get("MI355CD_PYHOST", os.path.join(PKG, "pyhost")))       # an older build's bindings (ab/pyhost_NAME/) when its ABI is older
    import numpy as np, mi355cd, mi355_synth as synth
    out = {}
    buf = np.empty((1 << 22, 2), dtype=np.uint32)
    for name, (v, t) in (("cloth1M", synth.cloth_pair(500)), ("soup1M", synth.soup(1_000_000, 0.01, 1234))):
        with mi355cd.CollisionDetector(v, t) as cd:
            cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0); cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 0)
            for _ in range(20): cd.self_collide_into(buf)
            d = []
            t0 = time.perf_counter()
            for _ in range(150): n, rc = cd.self_collide_into(buf); d.append(cd.fast_stats.ms_descend_clock)
            w = (time.perf_counter() - t0) / 150
            cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 15)
            p = []
            for _ in range(30): cd.self_collide_into(buf); p.append(cd.fast_stats.ms_pipeline)
            out[name] = {"descend_us": statistics.median(d) * 1e3, "wall_us": w * 1e6, "pipeline_us": statistics.median(p) * 1e3, "pairs": int(n), "tested": int(cd.fast_stats.pairs_tested)}
    print(json.dumps(out))
    sys.exit(0)
names = sys.argv[1:]
rounds = 3
res = {nm: [] for nm in names}
for r in range(rounds):
    for nm in names:
        env = dict(os.environ)
        if nm != "default":
            env["MI355CD_LIB"] = os.path.join(PKG, "ab", f"libmi355cd_{nm}.so")
            if os.path.isdir(os.path.join(PKG, "ab", f"pyhost_{nm}")):
                env["MI355CD_PYHOST"] = os.path.join(PKG, "ab", f"pyhost_{nm}")
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True, timeout=600)
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        if not line:
            print(nm, "FAILED", p.stderr[-500:]); continue
        res[nm].append(json.loads(line[-1]))
for nm in names:
    for wl in ("cloth1M", "soup1M"):
        xs = [r[wl] for r in res[nm]]
        if xs:
            print(f"{nm:16s} {wl}: descend {statistics.median(x['descend_us'] for x in xs):6.1f} us  pipeline {statistics.median(x['pipeline_us'] for x in xs):6.1f} us  wall {statistics.median(x['wall_us'] for x in xs):6.1f} us  pairs {xs[0]['pairs']} tested {xs[0]['tested']}")
