"""Runs N x cd_build_tree (Morton keys, sort, fused build: no traversal) on one of tools/trace_steps.py's meshes -- to be run under rocprofv3 --kernel-trace --stats, with
ablation builds of libmi355cd.so too (tools/ab_build.sh; the records of such a build may be garbage: nothing walks them here).  usage: trace_build.py MESH   env STEPS (default 100), MI355CD_LIB"""
import os, sys
sys.path[:0] = [os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpu-computing-course_amd", "pyhost")]
import mi355_synth as synth, mi355cd
mesh = sys.argv[1]
ids = None
if mesh == "cfg4_8M": v, t, ids, _, _ = synth.config4_merged(8, 500)
else: v, t = {"cloth4M": lambda: synth.cloth_pair(1000), "cloth1M": lambda: synth.cloth_pair(500), "soup1M": lambda: synth.soup(1_000_000, 0.01, 1234)}[mesh]()
with mi355cd.CollisionDetector(v, t, ids) as cd:
    cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0); cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 0)
    if ids is not None:
        cd.set_morton_frame(mi355cd.CD_FRAME_AUTO); cd.build_tree(); cd.keep_auto_frame()
    for _ in range(int(os.environ.get("STEPS", "100"))): cd.build_tree()
print(mesh, "done")
