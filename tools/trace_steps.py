"""Runs N fused steps on a bench workload with options from the command line -- to be run under rocprofv3 --kernel-trace --stats.
usage: trace_steps.py MESH [KEY=VALUE ...]    MESH: cloth1M | soup1M | soup100k | cloth1Md | cloth4M | cfg4_8M | soup8M | cfg4_8M_r5frame | <any>_auto
(cfg4_8M: CD_FRAME_AUTO computed in the first step and kept, as bench.py's config4_merged_8M; cfg4_8M_r5frame: round 5's per-axis frame with the fixed interleave;
 a mesh name ending in _auto runs that mesh in the kept AUTO frame instead of the reference's)
env STEPS (default 200)"""
import os, sys
sys.path[:0] = [os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpu-computing-course_amd", "pyhost")]
import numpy as np, mi355_synth as synth, mi355cd
mesh = sys.argv[1]
frame = None
auto = mesh.endswith("_auto")
if auto: mesh = mesh[:-5]
if mesh.startswith("cloth1M_shift"):                     # the 1 M cloth with sheet B slid along x by that many quads (tools/hint_moving.py's frames, at rest)
    v, t = synth.cloth_pair(500); ids = None
    h = v.shape[0] // 2; v[h:, 0] = np.float32(v[h:, 0] + np.float32(float(mesh[len("cloth1M_shift"):]) * 2.9 / 500))
elif mesh.startswith("clothq"):                          # cloth_pair(Q): clothq158 = 100 k triangles, clothq350 = 490 k ...
    v, t = synth.cloth_pair(int(mesh[6:])); ids = None
elif mesh in ("cfg4_8M", "cfg4_8M_r5frame", "cfg4_2M"):
    v, t, ids, off, span = synth.config4_merged(8, 250 if mesh == "cfg4_2M" else 500)
    if mesh == "cfg4_8M_r5frame": frame = (off, span)
    else: auto = True
else:
  ids = None
  v, t = {"cloth4M": lambda: synth.cloth_pair(1000), "soup8M": lambda: synth.soup(8_000_000, 0.005, 1234), "cloth1M": lambda: synth.cloth_pair(500), "soup1M": lambda: synth.soup(1_000_000, 0.01, 1234), "soup100k": lambda: synth.soup(100_000, 0.02, 1234),
        "cloth1Md": lambda: synth.cloth_pair(500, round_f32=False)}[mesh]()
buf = np.empty((1 << 22, 2), dtype=np.uint32)
STEPS = int(os.environ.get("STEPS", "200"))
with mi355cd.CollisionDetector(v, t, ids) as cd:
    if frame: cd.set_morton_frame(mi355cd.CD_FRAME_CUSTOM, frame[0], frame[1])
    cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0); cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 0)
    if auto:
        cd.set_morton_frame(mi355cd.CD_FRAME_AUTO); cd.self_collide_into(buf)
        print("kept AUTO frame:", cd.keep_auto_frame())
    for kv in sys.argv[2:]:
        k, val = kv.split("=")
        if k.startswith("d"): cd.debug_set(int(k[1:]), int(val))          # dKEY=VALUE: cd_debug_option
        else: cd.set_option(int(k), int(val))
    for _ in range(STEPS): n, rc = cd.self_collide_into(buf)
    st = cd.stats()
    print(mesh, sys.argv[2:], "pairs", n, "tested", cd.fast_stats.pairs_tested, "rc", rc, "sort_passes", st.sort_passes, "visits/query", st.node_visits / float(t.shape[0]),
          "lanes busy", st.node_visits / float(max(st.wave_steps, 1)))
