#!/usr/bin/env python3
"""bench.py -- headline benchmark of the collision hot path on MI355X.

Metric (BASELINE.json): triangle-pairs tested / second (+ total collision time) on 1 M-triangle
self-collision.  A *step* is one full pass of the hot path over one batch of synthetic input with the
input already resident in HBM: Morton keys -> radix sort -> LBVH hierarchy -> AABB refit -> BVH traversal
with the exact triangle test (cd_self_collide through the C ABI).  A *pair tested* is a (query, leaf)
pair whose AABBs strictly overlap and therefore reaches the neighbour filter / SAT.

N = 1 : BASELINE config 3, the 1 M-triangle synthetic cloth-vs-cloth.
N > 1 : BASELINE config 4, one 1 M-triangle cloth object per rank, neighbours overlapping by 10 % along
        x (weak scaling); per step each rank builds its tree, all-gathers root AABBs over RCCL, exchanges the
        overlapping leaves with its neighbours (grouped send / recv beside the local traversal) and traverses the
        received queries.  The step is cd_multi_step of libmi355cd.so: the C++ side issues the RCCL calls;
        torch.distributed only bootstraps (unique-id broadcast) and reduces the timings.  UNMEASURED ON >1 GPU so
        far: the builder's box has one GPU (a 1-rank communicator in self-peer mode runs every phase there).

Prints ONE JSON line on rank 0.  Launch for N > 1:
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
or plainly `python3 bench.py --gpus N ...`: without WORLD_SIZE / RANK in the environment the script starts that command itself, as
a child process and before anything touches the GPU, relays rank 0's line and returns the child's exit code.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(ROOT, "gpu-computing-course_amd", "pyhost")]

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
TRAVERSAL_BYTES_PER_TRI = 40.0   # box 24 + links 8 + leaf payload ~8 amortised, read once
BUILD_BYTES_PER_TRI = 100.0      # SURVEY.md 8d rows S3 + S4: hierarchy 24 + refit 76 -- what the fused block-build kernel does in one pass
TOTAL_BYTES_PER_TRI = 460.0      # whole path, compact layouts


def cpu_baseline(verts, vidx, reps=3):
    """The CPU oracle (a port of the reference's sequential cpu.cuh path) timed on this box's host
    cores: `reps` full passes over the SAME workload, single thread.  Checker code, used here only as
    the reported baseline."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    oracle.self_collide(verts[:3000], vidx[:1000], want_pairs=False)          # page in
    best = None
    t_all = 0.0
    for _ in range(reps):
        t0 = time.perf_counter()
        _, st, tm = oracle.self_collide(verts, vidx, want_pairs=False, threads=1)
        dt = time.perf_counter() - t0
        t_all += dt
        if best is None or dt < best[0]:
            best = (dt, st, tm)
    dt, st, tm = best
    out = {"value": st.pairs_tested / dt, "unit": "pairs_tested/s", "cores": 1, "kind": "port",
           "sample": f"{reps} full passes of the same workload (best of {reps}; {t_all:.1f} s of CPU work), single thread, gcc -O2 -ffp-contract=off",
           "total_collision_ms": dt * 1e3, "pairs_tested": int(st.pairs_tested), "n_pairs": int(st.n_pairs),
           "stage_ms": {"morton": tm.ms_morton, "sort": tm.ms_sort, "hierarchy": tm.ms_hierarchy, "refit": tm.ms_refit, "traverse": tm.ms_traverse}}
    # all-core variant: every stage but the sort on OpenMP threads -- one iteration per CUDA thread of the
    # reference's own kernels (oracle/cd_oracle.c build_parallel); the sort stays sequential like the reference's host sort
    try:
        nproc = len(os.sched_getaffinity(0))
    except AttributeError:
        nproc = os.cpu_count() or 1
    # a container's CPU share is a cgroup quota, which the affinity mask does not show: more threads than the quota only
    # fight each other (256 threads on a 16-CPU share ran 3x SLOWER than one core), so the thread count is swept and the
    # best one reported, with the mask's size beside it
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max") and txt[0] != "max":
                quota = max(1, int(round(int(txt[0]) / int(txt[1]))))
            elif path.endswith("cfs_quota_us") and int(txt[0]) > 0:
                quota = max(1, int(round(int(txt[0]) / int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()))))
            if quota:
                break
        except Exception:
            pass
    out["nproc"] = nproc
    out["cgroup_cpu_quota"] = quota
    if nproc > 1:
        tries = sorted({t for t in (8, 16, 32, 64, quota or 0, nproc) if 1 < t <= nproc and t <= 128})
        best2 = None
        for th in tries:
            oracle.self_collide(verts[:30000], vidx[:10000], want_pairs=False, threads=th)        # (re)size the thread pool
            for _ in range(2):
                t0 = time.perf_counter()
                _, st2, tm2 = oracle.self_collide(verts, vidx, want_pairs=False, threads=th)
                dt2 = time.perf_counter() - t0
                if best2 is None or dt2 < best2[0]:
                    best2 = (dt2, st2, tm2, th)
        dt2, st2, tm2, th = best2
        out["omp"] = {"value": st2.pairs_tested / dt2, "cores": th, "total_collision_ms": dt2 * 1e3,
                      "stage_ms": {"morton": tm2.ms_morton, "sort (sequential)": tm2.ms_sort, "hierarchy": tm2.ms_hierarchy, "refit": tm2.ms_refit,
                                   "traverse": tm2.ms_traverse},
                      "sample": f"best thread count of {tries} (2 full passes each), sort on one thread; affinity mask {nproc} CPUs, cgroup quota {quota}"}
    return out


def parity_check(pairs, tested, verts, vidx, ids=None, off=None, span=None, own_id_range=None):
    """Checker leg (after the timed region, never inside it): the pair SET and the pairs-tested count of the LAST timed step
    against the CPU oracle's on the same input (collision.cuh:19-88, tri_contact.cuh:80-87).  own_id_range (multi-GPU): the
    oracle ran on this rank's mesh merged with its lower neighbour's; the rank owns the pairs whose LARGER id is its own."""
    import hashlib
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    kw = {} if off is None else {"off": off, "span": span}
    want, st, _ = oracle.self_collide(verts, vidx, ids, want_pairs=True, threads=1, **kw)
    if own_id_range is not None:
        lo, hi = own_id_range
        want = want[(want[:, 1] >= lo) & (want[:, 1] < hi)]
    got_set, want_set = oracle.pair_set(pairs), oracle.pair_set(want)
    out = {"pairs_equal": bool(np.array_equal(got_set, want_set)), "n_pairs": int(got_set.size), "oracle_n_pairs": int(want_set.size),
           "pair_set_sha256": hashlib.sha256(got_set.tobytes()).hexdigest()[:16], "oracle_pair_set_sha256": hashlib.sha256(want_set.tobytes()).hexdigest()[:16]}
    if own_id_range is None:
        out["pairs_tested_equal"] = bool(int(tested) == int(st.pairs_tested))
        out["pairs_tested"] = int(tested); out["oracle_pairs_tested"] = int(st.pairs_tested)
    out["ok"] = out["pairs_equal"] and out.get("pairs_tested_equal", True)
    return out


def reference_compiled_check(name, pairs, tested):
    """Checker leg: the pair set and the pairs-tested count against the END RESULT the REFERENCE's own predicates give for this mesh
    (tests/golden/contact_ref.npz: tri_contact.cuh / box.cuh / triangle.cuh compiled unmodified in the build container)."""
    import hashlib
    path = os.path.join(ROOT, "tests", "golden", "contact_ref.npz")
    if not os.path.exists(path):
        return None
    ref = np.load(path)
    if name + "_pairs_sha" not in ref.files:
        return None
    p = np.asarray(pairs, dtype=np.uint64).reshape(-1, 2)
    keys = np.sort((p[:, 0] << np.uint64(32)) | p[:, 1])
    return bool(hashlib.sha256(keys.tobytes()).hexdigest() == str(ref[name + "_pairs_sha"]) and int(tested) == int(ref[name + "_tested"]))


def secondary_measurement(torch, which, steps=60, warmup=10):
    """Secondary workloads beside the headline, same call and options, the last step's pair set and pairs_tested against the oracle and
    against the reference-compiled end result.  N = 1 only.
      soup_1M         SURVEY.md 8d, inputs item 3 ("also report the 1 M soup"): the config-2 generator at 1 000 000 triangles, e = 0.01 -- own
                      vertices per triangle (no shared edges: the neighbour filter drops nothing, every overlapping leaf pair goes through the SAT);
      cloth_1M_double the headline's surfaces with vertices NOT rounded to float32 (vec3f.cuh:14-23 stores FP64; only the loader rounds,
                      load_obj.h:38): no leaf box is exact in fp32, so every candidate takes k_exact's FP64-box path and the boxes[] fetches."""
    import mi355_synth as synth
    import mi355cd
    if which == "soup_1M":
        verts, vidx = synth.soup(1_000_000, 0.01, 1234)
        label, fixture = "triangle soup, 1 000 000 triangles of edge 0.01 in the reference's box, own vertices per triangle (config-2 generator at 1 M)", "soup1M"
    else:
        verts, vidx = synth.cloth_pair(500, round_f32=False)
        label, fixture = "cloth-vs-cloth of the headline with full-double vertices (not rounded to float32): the FP64-box path of the exact kernel", "cloth1M_double"
    with mi355cd.CollisionDetector(verts, vidx) as cd, mi355cd.HostPairs(1 << 22) as hp:
        cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0); cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 0)
        for _ in range(warmup):
            cd.self_collide_into(hp.array)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tested = 0
        for _ in range(steps):
            n, rc = cd.self_collide_into(hp.array)
            tested += cd.fast_stats.pairs_tested
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if rc != 0:
            raise RuntimeError(which + ": pair capacity too small")
        last = np.array(hp.array[:n], copy=True); last_tested = cd.fast_stats.pairs_tested
        clock = cd.fast_stats.ms_descend_clock
    pc = parity_check(last, last_tested, verts, vidx)
    return {"workload": label, "ms_per_step": dt * 1e3 / steps, "pairs_tested_per_s": tested / dt, "pairs_tested_per_step": int(last_tested), "colliding_pairs": int(n),
            "steps": steps, "descend_device_clock_ms": clock, "parity_checked": bool(pc["ok"]), "reference_compiled_end_result": reference_compiled_check(fixture, last, last_tested)}


def moving_mesh_measurement(torch, quads, frames=40, shift_quads=1.0):
    """The per-frame loop INTEGRATION.md describes, on a mesh that MOVES: sheet B of the headline's cloth pair slides `shift_quads` quads along x per frame
    (cd_update_vertices with float-valued vertices, then cd_self_collide).  Only the STEP is timed (the clock starts after the upload has returned: a synchronous
    copy), frame by frame, with the order hint on and off in alternating runs -- the hint carries the previous frame's wave times over BY TRIANGLE, so this is what
    it is worth when the triangles sort into other groups every frame (the static timed region is the best a hint can be).  The last frame's pair set and
    pairs_tested are checked against the CPU oracle."""
    import statistics
    import mi355_synth as synth
    import mi355cd
    verts, vidx = synth.cloth_pair(quads)
    quad = 2.88 / quads
    h = verts.shape[0] // 2
    res = {1: [], 0: []}
    with mi355cd.CollisionDetector(verts, vidx) as cd, mi355cd.HostPairs(1 << 22) as hp:
        cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0); cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 0)
        for hint in (1, 0, 1, 0):
            cd.set_option(mi355cd.CD_OPT_ORDER_HINT, hint)
            wall, clk = [], []
            for f in range(frames):
                v = verts.copy()
                # (there and back again, 20 quads out: the sheets stay inside the reference's Morton frame, morton.h:43-58 -- 26 quads out sheet B's centroids leave it, the
                #  keys pass 2^60 and the sort takes its next form: rounds 4's tools/hint_moving.py and this leg's first version measured THAT in their last 14 frames)
                pos = 20.0 - abs(20.0 - (f % 40))
                v[h:, 0] = np.float32(v[h:, 0] + np.float32(pos * shift_quads * quad))        # (float-valued like the loader's output: no cell table)
                cd.update_vertices(v)
                t0 = time.perf_counter()
                n, rc = cd.self_collide_into(hp.array)
                dt = time.perf_counter() - t0
                if rc != 0:
                    raise RuntimeError("moving mesh: pair capacity too small")
                if f >= 5:
                    wall.append(dt * 1e3); clk.append(cd.fast_stats.ms_descend_clock)
            res[hint].append((statistics.mean(wall), statistics.mean(clk)))
            last = np.array(hp.array[:n], copy=True); last_tested = cd.fast_stats.pairs_tested; last_v = v
        cd.set_option(mi355cd.CD_OPT_ORDER_HINT, 1)
    pc = parity_check(last, last_tested, last_v, vidx)
    on = [statistics.mean(x[i] for x in res[1]) for i in (0, 1)]; off = [statistics.mean(x[i] for x in res[0]) for i in (0, 1)]
    return {"workload": f"the headline's cloth pair, sheet B sliding {shift_quads:g} quad(s) along x per frame, 20 quads out and back (inside the reference's Morton frame): cd_update_vertices + cd_self_collide per frame, {frames} frames a run, 2 runs each way",
            "timed": "the step only (wall clock around cd_self_collide, polled completion; the upload before it is a synchronous copy and stays outside), mean over the frames after the 5th",
            "ms_per_step": on[0], "ms_per_step_without_hint": off[0], "descend_device_clock_ms": on[1], "descend_device_clock_ms_without_hint": off[1],
            "parity_checked": bool(pc["ok"]), "colliding_pairs_last_frame": int(pc["n_pairs"])}


def from_obj_measurement(torch, quads):
    """End to end FROM THE FILE, once: the counterpart of the reference's "Total Time" (main.cu:55,170-171: one event pair around loadObj ... the read-back, 187.2 ms on its
    data against 71 ms of kernels).  The headline's mesh is written as an OBJ in the reference's dialect (`v x y z`, `f a/ta b/tb c/tc`, load_obj.h:48-56,64-80; %.9g so that
    the parsed floats are the generator's), then cd_load_obj (threaded parser) -> cd_create (upload) -> the FIRST cd_self_collide of the context (allocations, first-touch,
    no order hint yet), each timed on the wall clock; the pairs against the oracle on the PARSED mesh."""
    import ctypes as C
    import tempfile
    import mi355_synth as synth
    import mi355cd
    verts, vidx = synth.cloth_pair(quads)
    fd, path = tempfile.mkstemp(suffix=".obj", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        with os.fdopen(fd, "w") as f:
            f.write("# bench.py from_obj: BASELINE config 3\n")
            f.write("".join(["v %.9g %.9g %.9g\n" % (a, b, c) for a, b, c in verts.tolist()]))
            f.write("".join(["f %d/%d %d/%d %d/%d\n" % (a + 1, a + 1, b + 1, b + 1, c + 1, c + 1) for a, b, c in vidx.tolist()]))
        size = os.path.getsize(path)
        lib = mi355cd.load_library()
        threads = min(16, len(os.sched_getaffinity(0)))
        pv, pf = C.POINTER(C.c_double)(), C.POINTER(C.c_uint32)()
        nv, nt = C.c_uint32(0), C.c_uint32(0)
        open(path, "rb").read()                                               # (page cache warm: the parse is timed, not the disk)
        t0 = time.perf_counter()
        rc = lib.cd_load_obj(path.encode(), C.byref(pv), C.byref(nv), C.byref(pf), C.byref(nt), threads)
        t1 = time.perf_counter()
        if rc != 0:
            raise RuntimeError("cd_load_obj failed: %d" % rc)
        try:
            pverts = np.ctypeslib.as_array(pv, shape=(nv.value, 3)); pvidx = np.ctypeslib.as_array(pf, shape=(nt.value, 3))
            same = bool(np.array_equal(pverts, verts) and np.array_equal(pvidx, vidx))
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            cd = mi355cd.CollisionDetector(pverts, pvidx)                     # cd_create: allocations + upload (main.cu:78-88)
            t3 = time.perf_counter()
            with cd:
                pairs, n, rc2 = cd.self_collide(cap=1 << 22)                 # the context's FIRST step, pairs into an ordinary host buffer (main.cu:91-151)
                t4 = time.perf_counter()
                tested = cd.stats().pairs_tested
                pc = parity_check(pairs[:n], tested, np.array(pverts), np.array(pvidx))
        finally:
            lib.cd_free_obj(pv, pf)
    finally:
        os.unlink(path)
    return {"what": "OBJ file -> cd_load_obj -> cd_create -> first cd_self_collide, each once, wall clock (the reference's Total Time, main.cu:55,170-171: 187.2 ms on its data)",
            "obj_bytes": int(size), "threads": int(threads), "parse_ms": (t1 - t0) * 1e3, "create_ms": (t3 - t2) * 1e3, "first_step_ms": (t4 - t3) * 1e3,
            "total_ms": (t1 - t0 + t4 - t2) * 1e3, "parsed_mesh_equals_generated": same, "colliding_pairs": int(n), "parity_checked": bool(pc["ok"] and rc2 == 0)}


def size_measurement(torch, which, steps=40, warmup=8):
    """The step ABOVE 1 M triangles on one GPU, same call and options as the headline, untimed-region extras like soup_1M:
      cloth_4M          cloth-vs-cloth with 1000 x 1000 quads per sheet = 4 000 000 triangles (the headline's surfaces, twice as fine);
      config4_merged_8M BASELINE config 4's eight shards of 1 M triangles as ONE mesh of 8 000 000 (mi355_synth.config4_merged) -- the single-GPU
                        point a scaling curve over 8 GPUs is read against.
    ms_per_step (wall, polled completion), then untimed profiling steps: every kernel stamp on (build_block / descend / exact / device pipeline time) and
    per-stage events (morton / sort / build / traverse); whole_path = 460 B x N / device time against the 8 TB/s roof.  The last timed step's pair set and
    pairs_tested are checked against the CPU oracle on the same mesh."""
    import mi355_synth as synth
    import mi355cd
    ids = frame = None
    if which == "cloth_4M":
        verts, vidx = synth.cloth_pair(1000)
        label = "cloth-vs-cloth, 2 sheets x 1000x1000 quads = 4 000 000 triangles, self-collision"
    else:
        verts, vidx, ids, off5, span5 = synth.config4_merged(8, 500); frame = "auto"
        label = "BASELINE config 4's eight 1 M-triangle cloth objects (10 % x-overlap between neighbours) merged into ONE mesh of 8 000 000 triangles, self-collision on one GPU"
    nt = int(vidx.shape[0])
    cap = 1 << 22
    frame_info = {"mode": "CD_FRAME_REFERENCE (morton.h:43-58: the mesh lies in the reference's frame)"}
    with mi355cd.CollisionDetector(verts, vidx, ids) as cd, mi355cd.HostPairs(cap) as hp:
        cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0); cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 0)
        if frame is not None:
            # The reference's constants do not fit this mesh: the library derives the frame (CD_FRAME_AUTO, the adaptive frame of cd_math.h) in a first step and the
            # bench KEEPS what it computed (cd_get_morton_frame -> cd_set_morton_frame_layout), as the reference keeps its constants -- nothing is restated here.
            cd.set_morton_frame(mi355cd.CD_FRAME_AUTO)
            cd.self_collide_into(hp.array)
            foff, fspan, flay = cd.keep_auto_frame()
            frame_info = {"mode": "CD_FRAME_AUTO computed in a first step, then kept (cd_get_morton_frame -> cd_set_morton_frame_layout)", "offset": [float(x) for x in foff], "span": [float(x) for x in fspan],
                          "layout_word": hex(flay), "layout": {"axes_by_weight": [int(flay & 3), int((flay >> 2) & 3), int((flay >> 4) & 3)], "leading_bits_of_A": int((flay >> 8) & 255),
                                                                "pairs_AB": int((flay >> 16) & 255), "triples_ABC": int((flay >> 24) & 255)}}
        for _ in range(warmup):
            n, rc = cd.self_collide_into(hp.array)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tested = 0
        for _ in range(steps):
            n, rc = cd.self_collide_into(hp.array)
            tested += cd.fast_stats.pairs_tested
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if rc != 0:
            raise RuntimeError(which + ": pair capacity too small")
        last = np.array(hp.array[:n], copy=True); last_tested = cd.fast_stats.pairs_tested
        prof = 10
        kern = {"build_block": 0.0, "descend": 0.0, "exact": 0.0}; pipeline = 0.0
        cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 15)
        for _ in range(prof):
            cd.self_collide_into(hp.array); st = cd.stats()
            kern["build_block"] += st.ms_build_block / prof; kern["descend"] += st.ms_descend / prof; kern["exact"] += st.ms_exact / prof; pipeline += st.ms_pipeline / prof
        stage = {"morton": 0.0, "sort": 0.0, "build_fused(hierarchy+refit+records)": 0.0, "traverse": 0.0}
        cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 1)
        for _ in range(prof):
            cd.self_collide_into(hp.array); st = cd.stats()
            stage["morton"] += st.ms_morton / prof; stage["sort"] += st.ms_sort / prof
            stage["build_fused(hierarchy+refit+records)"] += (st.ms_hierarchy + st.ms_refit) / prof; stage["traverse"] += st.ms_traverse / prof
        sort_passes = int(st.sort_passes)
        lane_use = {"node_visits_per_step": int(st.node_visits), "wave_steps_per_step": int(st.wave_steps), "lanes_busy_of_64": st.node_visits / float(max(st.wave_steps, 1)),
                    "node_visits_per_query": st.node_visits / float(nt)}
        frame_ab = None
        if frame is not None:
            # the same mesh in round 5's frame (per-axis normalisation with morton.h:70-89's fixed interleave: cells of 400 : 1 on this mesh) -- what the adaptive frame is worth
            cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0); cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 0)
            cd.set_morton_frame(mi355cd.CD_FRAME_CUSTOM, off5, span5)
            for _ in range(4):
                cd.self_collide_into(hp.array)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(12):
                cd.self_collide_into(hp.array)
            torch.cuda.synchronize(); dt5 = (time.perf_counter() - t0) / 12
            cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 15)
            d5 = 0.0
            for _ in range(4):
                cd.self_collide_into(hp.array); st5 = cd.stats(); d5 += st5.ms_descend / 4
            frame_ab = {"per_axis_fixed_interleave(round 5)": {"ms_per_step": dt5 * 1e3, "descend_ms": d5, "node_visits_per_query": st5.node_visits / float(nt), "sort_passes": int(st5.sort_passes)}}
    if ids is None:
        pc = parity_check(last, last_tested, verts, vidx)
    else:
        pc = parity_check(last, last_tested, verts, vidx, ids, off=off5, span=span5)
    ach = TOTAL_BYTES_PER_TRI * nt / (pipeline * 1e-3) / 1e9
    return {"workload": label, "triangles": nt, "morton_frame": frame_info, "descent_lane_use": lane_use, **({"frame_ab": frame_ab} if frame_ab else {}), "ms_per_step": dt * 1e3 / steps, "pairs_tested_per_s": tested / dt, "pairs_tested_per_step": int(last_tested), "colliding_pairs": int(n),
            "steps": steps, "total_collision_ms_device": pipeline, "kernel_ms": kern, "stage_ms": stage, "sort_passes": sort_passes,
            "kernel_ms_note": f"from {prof} extra untimed steps with all kernel stamps on (HIP events on the kernels' own dispatch packets); stage_ms from {prof} more with per-stage events (their sum exceeds the device time by the event gaps)",
            "whole_path": {"bytes": TOTAL_BYTES_PER_TRI * nt, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS},
            "parity_checked": bool(pc["ok"]), "parity": pc}


def ray_tracer_measurement(dim=4096, n_spheres=4096, frames=5, rank=0, world=1, dist=None, torch=None, device=None, multi=False):
    """Secondary path (BASELINE config 5): 4096^2 image, 4096 spheres, frame kept on the device (the reference
    copies every frame to the host for glDrawPixels, anime_ray.cu:128-131; that PCIe copy is not kernel time).
    world > 1: REPLICAS ONLY -- spheres replicated, rank r renders image rows [r, r+1) * dim / world (rt_render_rows), no
    collective on the data path; frames/s = frames all ranks finished / max-over-ranks wall time.  The row slabs going to the
    host (the reference's per-frame D2H) and their assembly into one frame (an all-gather over RCCL) are timed separately.
    Every rank checks 64 rows of its slab against the oracle (the whole frame takes the CPU minutes)."""
    import statistics
    import mi355_synth as synth
    import mi355rt
    spheres, shifts = synth.sphere_scene(n_spheres, dim, seed=7)
    rows = (rank * dim // world, (rank + 1) * dim // world)
    out = {"workload": f"{dim}x{dim} RGBA8 frame, {n_spheres} spheres (BASELINE config 5)" + (f", image rows sharded over {world} replicas (spheres replicated, no collective)" if multi else "")}
    with mi355rt.RayTracer(spheres, dim) as rt:
        for name, mode in (("binned", mi355rt.RT_MODE_BINNED), ("brute", mi355rt.RT_MODE_BRUTE)):
            if multi and name == "brute":
                continue
            rt.set_mode(mode)
            rt.render(shifts, rows=rows, download=False)
            ms = []
            for _ in range(frames):
                rt.render(shifts, rows=rows, download=False)
                ms.append(rt.stats().ms_render)
            st = rt.stats()
            m = statistics.median(ms)
            out[name] = {"ms_per_frame": m, "sphere_tests_per_frame": int(st.sphere_tests), "sphere_tests_per_s": st.sphere_tests / (m * 1e-3)}
            if mode == mi355rt.RT_MODE_BINNED and not multi:
                # 16 frames queued back to back (time stamps on the first and the last kernel only): what a loop that does not come back
                # to the host per frame sees -- a single frame's two stamps cost it ~5 us of idle GPU between its two kernels
                rb = []
                for _ in range(3):
                    rt.render_repeat(shifts, 16, download=False)
                    rb.append(rt.stats().ms_render)
                out[name]["ms_per_frame_back_to_back"] = statistics.median(rb)
                # the animation loop (anime_ray.cu:115-131: the spheres move, the frame is rendered, per frame) as rt_anim_loop runs it: ONE launch per frame, frame f
                # rendering while the spheres move on to frame f + 1 and are binned for it in the same launch -- k_prepare is off the frame's critical path
                rt.anim_init()
                la = []
                for _ in range(3):
                    rt.anim_loop(32, 2, 35, 1, 18, download=False)
                    la.append(rt.stats().ms_render)
                out[name]["ms_per_frame_animation_loop"] = statistics.median(la)
                last = rt.anim_loop(1, 2, 35, 1, 18, download=True)
                ash, _, _ = rt.anim_state()
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                import oracle as _orc
                yl = dim // 2
                out[name]["animation_loop_parity_checked"] = bool(np.array_equal(last[yl:yl + 64], _orc.rt_render(spheres, ash, dim, rows=(yl, yl + 64))[yl:yl + 64]))
                out[name]["animation_loop_note"] = ("97 frames of the curve-move animation through rt_anim_loop; rows [%d, %d) of the last one pixel-equal to the CPU oracle rendering the "
                                                    "state read back from the device" % (yl, yl + 64))
        rt.set_mode(mi355rt.RT_MODE_BINNED)
        if multi:
            # whole-job frame rate: K frames per rank (its rows), barrier + synchronize both sides, max over ranks
            kf = 50
            dist.barrier(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(kf):
                rt.render(shifts, rows=rows, download=False)
            torch.cuda.synchronize(); dist.barrier()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            out["frames_per_s_whole_job"] = kf / float(t.item())
            out["ms_per_frame_wall_max_over_ranks"] = float(t.item()) * 1e3 / kf
            t0 = time.perf_counter()
            slab = rt.render(shifts, rows=rows, download=True)           # + D2H of the rank's slab (anime_ray.cu:128-131)
            t_d2h = time.perf_counter() - t0
            # the slabs assembled into one frame in rank-order: an all-gather over RCCL (the library keeps its frame buffer to itself, so the
            # slab goes back to the device first, untimed; a display process would rather read the ranks' host slabs where they are)
            dslab = torch.from_numpy(slab).to(device)
            frame = torch.empty((world,) + tuple(dslab.shape), dtype=torch.uint8, device=device)
            torch.cuda.synchronize(); dist.barrier()
            t0 = time.perf_counter()
            dist.all_gather_into_tensor(frame, dslab)
            torch.cuda.synchronize()
            t_gather = time.perf_counter() - t0
            out["row_slab_render_plus_d2h_ms"] = t_d2h * 1e3
            out["row_slabs_all_gather_rccl_ms"] = t_gather * 1e3
        else:
            slab = rt.render(shifts, rows=rows, download=True)
        # checker: 64 rows of this rank's slab against the oracle (sphere.cuh:34-44, anime_ray.cu:61-87)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle
        y0 = rows[0] + ((rows[1] - rows[0]) // 2 // 64) * 64
        want = oracle.rt_render(spheres, shifts, dim, rows=(y0, y0 + 64))
        ok = bool(np.array_equal(slab[y0 - rows[0]: y0 - rows[0] + 64], want[y0:y0 + 64] if want.shape[0] == dim else want))
        if multi:
            f = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
            dist.all_reduce(f, op=dist.ReduceOp.MIN)
            ok = bool(int(f.item()))
        out["parity_checked"] = ok
        out["parity_note"] = f"rows [{y0}, {y0 + 64}) of every rank's slab pixel-equal to the CPU oracle" + (" (MIN over ranks)" if multi else "") + "; whole frames in tests/test_rt_gpu.py"
    frame_bytes = (rows[1] - rows[0]) * dim * 4 + n_spheres * 32
    a = frame_bytes / (out["binned"]["ms_per_frame"] * 1e-3) / 1e9
    out["roofline"] = {"bound": "hbm", "kernel": "k_render<binned>", "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a / HBM_PEAK_GBS,
                       "algorithmic_bytes_per_launch": frame_bytes, "note": "per GPU (its rows)"}
    return out


def self_launch(n):
    """`python3 bench.py --gpus N` without a launcher around it: start `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>` as a child process, pass its output through (rank 0 prints
    the ONE JSON line) and hand back its return code.  Nothing in this process has imported torch or touched the GPU."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")                 # dmabuf IPC: what RCCL needs on this host driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: --gpus %d without WORLD_SIZE: launching %s" % (n, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--quads", type=int, default=500, help="quads per sheet edge; 500 -> 1 000 000 triangles per rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle check of the last timed step's pair set")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary workloads measured after the headline (the 1 M soup of SURVEY.md 8d; the cloth with full-double "
                                                             "vertices): a kernel trace of the command then holds the headline workload's launches only")
    ap.add_argument("--no-ray", action="store_true", help="skip the secondary ray-tracer measurement (BASELINE config 5)")
    ap.add_argument("--traversal", type=int, default=None, help="CD_OPT_TRAVERSAL override (0 lane-private FP64, 1 wave-queued)")
    ap.add_argument("--qpw", type=int, default=None, help="CD_OPT_QUERIES_PER_WAVE override")
    ap.add_argument("--no-order-hint", action="store_true", help="CD_OPT_ORDER_HINT 0 for the whole run: the timed region steps in the plain order (what a mesh that moves between "
                                                                  "steps gets); by default the line carries that measurement beside the headline (order_hint.ms_per_step_without)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        # `python3 bench.py --gpus N` started plainly: this process becomes the LAUNCHER -- before torch is imported, so before anything here
        # touches the GPU -- and runs the one-rank-per-GPU job as a CHILD (never an exec), relays rank 0's line and returns the child's code
        sys.exit(self_launch(args.gpus))

    import torch
    import torch.distributed as dist
    import mi355_synth as synth
    import mi355cd
    import mi355_multi as multi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("MI355_DIST_BACKEND", "nccl") != "nccl":
        local_rank = 0                                                # rehearsal: every rank on cuda:0
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (or start `python3 bench.py --gpus N` plainly: it launches them itself)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # MI355_DIST_BACKEND=gloo is a single-GPU REHEARSAL of the multi-process flow (ranks share cuda:0, payloads
    # staged through the host); the measured configuration is always nccl (= RCCL over xGMI), one rank per GPU.
    backend = os.environ.get("MI355_DIST_BACKEND", "nccl")
    # MI355_BENCH_SELF_PEER=1 (rehearsal on ONE GPU, launched with --nproc-per-node 1): the multi-GPU code path of this
    # file and cd_multi_step run with a one-rank communicator that exchanges with itself -- never a measured configuration
    self_peer = os.environ.get("MI355_BENCH_SELF_PEER", "0") == "1" and world == 1
    # MI355_BENCH_MULTI_PATH=1 with --gpus 1: the N = 1 point of a scaling curve measured on the SAME code path as N > 1 -- cd_multi_step with a one-rank RCCL
    # communicator and no peer: box, both all-gathers, the pack launch, the two host synchronisations, no exchange.  Its `value` is the plain line's minus what that
    # machinery costs a step; `path` in the line says which one ran.
    multi_n1 = os.environ.get("MI355_BENCH_MULTI_PATH", "0") == "1" and world == 1 and not self_peer
    multi_path = world > 1 or self_peer or multi_n1
    if multi_path:
        if world == 1 and "MASTER_ADDR" not in os.environ:           # a one-rank job started plainly: the rendezvous is this process
            import socket
            with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
                sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)        # nccl == RCCL on ROCm
        else:
            dist.init_process_group(backend)

    # ---- workload (synthetic, seeded), resident in HBM before the timed region
    if not multi_path:
        verts, vidx = synth.cloth_pair(args.quads)
        ids = None
        frame = mi355cd.CD_FRAME_REFERENCE                            # geometry lies in the reference's Morton frame
        workload = f"cloth-vs-cloth, 2 sheets x {args.quads}x{args.quads} quads = {vidx.shape[0]} triangles, self-collision (BASELINE config 3)"
    else:
        verts, vidx, ids, vbase = synth.cloth_shard(rank, args.quads)
        frame = mi355cd.CD_FRAME_AUTO
        workload = (("ONE-GPU REHEARSAL (self peer) of " if self_peer else "") + f"{world} x cloth-vs-cloth objects of {vidx.shape[0]} triangles, 10% x-overlap between neighbours, "
                    f"sharded by object (BASELINE config 4 shape)")
    nt = vidx.shape[0]
    # neighborCount compares vertex INDICES: a per-rank base makes them global for the cross-rank pass
    engine = multi.HipEngine(verts, vidx, ids, device, frame, vertex_id_base=vbase if multi_path else 0)
    if multi_path:
        # The reference normalises Morton keys with constants of its data set (morton.h:43-58), fixed once.  A shard of config 4 lies outside that frame for rank > 0:
        # the library derives the rank's frame itself (CD_FRAME_AUTO: the adaptive frame of cd_math.h) in one build at set-up, and the bench KEEPS it -- fixed like the
        # reference's constants, not recomputed from the centroids in every step (the AUTO pass over the triangles is ~11 us of a 0.31 ms step).
        engine.cd.build_tree()
        engine.cd.keep_auto_frame()
    if args.traversal is not None:
        engine.cd.set_option(mi355cd.CD_OPT_TRAVERSAL, args.traversal)
    if args.no_order_hint:
        engine.cd.set_option(mi355cd.CD_OPT_ORDER_HINT, 0)
    if args.qpw is not None:
        engine.cd.set_option(mi355cd.CD_OPT_QUERIES_PER_WAVE, args.qpw)
    cap = 1 << 22
    host_pairs = mi355cd.HostPairs(cap)                               # cd_alloc_host_pairs: pinned, the GPU writes the pairs straight into it
    pair_buf = host_pairs.array

    comm_device = None if backend == "nccl" else "cpu"
    ms = None
    multi_fallback = None
    if multi_path and backend == "nccl":
        # the product path: cd_multi_step (C++ over RCCL).  Rank 0 makes the ncclUniqueId, torch.distributed hands it round.
        uid = torch.zeros(128, dtype=torch.uint8, device=device)
        if rank == 0:
            uid = torch.frombuffer(bytearray(mi355cd.multi_unique_id()), dtype=torch.uint8).to(device)
        dist.broadcast(uid, src=0)
        # If the library cannot set its step up on some rank (librccl.so not loadable, ncclCommInitRank refused, an allocation ...) EVERY rank drops to the same step orchestrated
        # from Python over torch.distributed's RCCL communicator (pyhost/mi355_multi.collide_step: the same kernels behind the same C ABI, exchange by all_to_all_single) -- the
        # line then says so (`multi_fallback`) instead of the run dying without a line.  MI355_BENCH_NO_CD_MULTI=1 takes that path on purpose (tests).
        rc_create = 0
        if os.environ.get("MI355_BENCH_NO_CD_MULTI", "0") == "1":
            rc_create = 1
        else:
            try:
                ms = mi355cd.MultiStep(engine.cd, bytes(uid.cpu().numpy().tobytes()), rank, world, query_cap_per_peer=nt // 8 + 1024,
                                       flags=mi355cd.CD_MULTI_SELF_PEER if self_peer else 0)
            except mi355cd.CdError as e:
                rc_create = int(e.rc) or 1
        worst = torch.tensor([abs(rc_create)], dtype=torch.int64, device=device)
        dist.all_reduce(worst, op=dist.ReduceOp.MAX)
        if int(worst.item()) != 0:
            if ms is not None:
                ms.close()
                ms = None
            multi_fallback = (f"cd_multi_create did not succeed on every rank (this rank: {rc_create}, worst: -{int(worst.item())}): the step is orchestrated from Python over "
                              "torch.distributed (RCCL) -- mi355_multi.collide_step: same kernels, exchange by all_to_all_single, payloads on the device")
    last_info = {}

    def step():
        if ms is not None:
            pairs, n, rc, mi = ms.step(cap)
            if rc != 0:
                raise RuntimeError(f"pair capacity {cap} too small for {n} pairs")
            last_info["mi"] = mi
            return pairs, int(mi.pairs_tested), {"local_pairs": int(mi.local_pairs), "cross_pairs": int(mi.cross_pairs), "sent_queries": int(mi.sent_queries),
                                                 "recv_queries": int(mi.recv_queries), "n_peers": int(mi.n_peers), "host_syncs": int(mi.host_syncs),
                                                 "attempts": int(mi.attempts)}
        if world == 1 and not multi_path:
            # one GPU: the C ABI's call and nothing else -- pairs stay in the buffer the library wrote them into
            n, rc = engine.cd.self_collide_into(pair_buf)
            if rc != 0:
                raise RuntimeError(f"pair capacity {cap} too small for {n} pairs")
            return pair_buf[:n], None, {}
        return multi.collide_step(engine, dist, rank, world, cap, comm_device)

    # The warm-up steps are the TIMED step: same options (round 6: until now they ran with the library's default per-stage events and kernel stamps -- barrier packets and idle
    # gaps the timed steps do not have -- and the first two timed steps of every run came out 5 % long, whatever W was: 2 % of the driver's 20-step line; per_step_wall_ms shows it)
    engine.cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0)
    if not multi_path:
        engine.cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 0)
    for _ in range(args.warmup):
        step()
    # The DOMINANT kernel (the descent) is timed (1) by ITSELF in every step of the timed region -- first wave start -> last wave end on the
    # device's constant wall clock (cd_stats.ms_descend_clock), free -- and (2) with HIP events riding on its dispatch packet (cd_stats.ms_descend)
    # in the profiling steps right behind the timed region, with the other kernels': a stamped kernel costs its step ~7 us of idle GPU around it,
    # so the K timed steps carry no stamp at all (round 5; the two clocks and rocprofv3's average agree to a microsecond).  The device time of the
    # whole pipeline and the per-stage breakdown come from those extra, untimed steps too (see the end)
    engine.cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0)
    if not multi_path:
        engine.cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 0)
    # ---- timed region: exactly K steps, barrier + synchronize on both sides, max over ranks
    stage = {"morton": 0.0, "sort": 0.0, "build_fused(hierarchy+refit+records)": 0.0, "traverse": 0.0}
    kern = {"descend": 0.0, "descend_device_clock": 0.0, "exact": 0.0, "build_block": 0.0}   # the two kernels inside "traverse" + the fused hierarchy / refit kernel
    tested_total = 0
    pairs_found = 0
    pipeline_ms = 0.0
    info = {}
    if multi_path:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step_desc = []                                                    # the descent kernel's own device-clock duration, step by step
    step_wall = []                                                    # per-step wall clock (a step ends when the host has its pairs: the stamps cost two clock reads a step)
    t_prev = t0
    for i_step in range(args.steps):
        pairs, tested, info = step()
        t_now = time.perf_counter(); step_wall.append((t_now - t_prev) * 1e3); t_prev = t_now
        st = engine.cd.fast_stats if not multi_path else engine.cd.stats()   # (refreshed by the step's own call)
        if tested is None:
            tested = st.pairs_tested
        if not multi_path:
            kern["descend_device_clock"] += st.ms_descend_clock
            step_desc.append(st.ms_descend_clock)
        tested_total += tested
        pairs_found = pairs.shape[0]
    torch.cuda.synchronize()
    if multi_path:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    last_pairs = np.array(pairs, copy=True)                           # the LAST TIMED step's output (the buffer is reused below): what parity_check sees
    last_tested = tested
    if multi_path:
        rdev = device if backend == "nccl" else torch.device("cpu")
        t = torch.tensor([elapsed], dtype=torch.float64, device=rdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([tested_total, pairs_found], dtype=torch.int64, device=rdev)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        tested_total, pairs_found = int(c[0].item()), int(c[1].item())

    if rank == 0:
        k = args.steps
        ms_per_step = elapsed * 1e3 / k
        line = {
            "metric": "triangle-pairs tested/sec (total collision time in ms_per_step), 1M-tri self-collision",
            "value": tested_total / elapsed, "unit": "pairs_tested/s", "n_gpus": world, "steps": k, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload, "triangles_per_gpu": int(nt), "pairs_tested_per_step": tested_total // k,
                       "colliding_pairs": int(pairs_found), "sharding": "by object" if multi_path else "none",
                       "step": ("cd_multi_step" if ms is not None else "cd_self_collide into a pinned pair buffer of cd_alloc_host_pairs, library defaults but CD_OPT_STAGE_TIMING 0 / "
                                "CD_OPT_KERNEL_STAMPS 0 (the step's end is then read off the report's sequence word: CD_OPT_POLL, default on)")},
            # `value` counts what the reference counts: (query, leaf) pairs whose AABBs strictly overlap, every unordered leaf pair TWICE
            # (collision.cuh:31-44 lets each leaf query the whole tree).  The default half traversal decides each unordered pair ONCE and
            # credits 2 (box.cuh:40-43 and neighborCount are symmetric; equal to the oracle's counter in every test): the device executes
            # half as many exact box decisions as `value` says
            "path": ("cd_multi_step (C++ over RCCL)" + (", one-rank communicator exchanging with itself: REHEARSAL" if self_peer else (", one-rank communicator, no peer" if multi_n1 else ""))) if ms is not None
                    else ("cd_self_collide" if not multi_path else ("Python orchestration of the multi-GPU step over torch.distributed (RCCL), device payloads: FALLBACK" if multi_fallback
                                                                    else "Python rehearsal of the multi-GPU step over " + backend)),
            "pairs_tested_counting": "reference-equivalent: the half traversal decides each unordered leaf pair once and credits the 2 ordered tests the reference makes",
            "box_decisions_executed_per_step": (tested_total // k) // (1 if args.traversal in (0, 1) else 2),
        }
        # where the K steps' time went, step by step (the driver's 20-step runs sit ~2 % above this file's 200-step default)
        sw = sorted(step_wall)
        line["per_step_wall_ms"] = {"min": sw[0], "median": sw[len(sw) // 2], "max": sw[-1], "first_8": step_wall[:8], "last_4": step_wall[-4:],
                                    "mean_first_quarter": sum(step_wall[:max(1, k // 4)]) / max(1, k // 4), "mean_last_quarter": sum(step_wall[-max(1, k // 4):]) / max(1, k // 4),
                                    "descend_device_clock_first_8": step_desc[:8], "descend_device_clock_last_4": step_desc[-4:],
                                    "note": "a step ends when the host has its pairs; the descent's duration is the kernel's own (device wall clock): where the first steps of a run are long, the kernel is"}
        if not multi_path:
            prof_steps = min(k, 20)
            engine.cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 15)   # untimed: the same step with every kernel stamp + pipeline start / end
            for _ in range(prof_steps):
                step()
                st = engine.cd.stats()
                kern["exact"] += st.ms_exact * k / prof_steps; kern["build_block"] += st.ms_build_block * k / prof_steps; kern["descend"] += st.ms_descend * k / prof_steps
                pipeline_ms += st.ms_pipeline * k / prof_steps
            engine.cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 1)  # untimed: the same step with events around every stage
            for _ in range(prof_steps):
                step()
                st = engine.cd.stats()
                stage["morton"] += st.ms_morton; stage["sort"] += st.ms_sort
                stage["build_fused(hierarchy+refit+records)"] += st.ms_hierarchy + st.ms_refit      # one pass: k_build_block + k_cross_fused (ms_hierarchy is 0 on the fused path)
                stage["traverse"] += st.ms_traverse
            for s in stage:
                stage[s] /= prof_steps
            dev_total = pipeline_ms / k
            line["total_collision_ms_device"] = dev_total          # untimed steps with all kernel stamps: pipeline start -> end of the traversal kernels
            line["stage_ms"] = stage                               # untimed profiling steps (stage events add idle gaps)
            line["stage_ms_note"] = f"from {prof_steps} extra untimed steps with per-stage events; their sum exceeds total_collision_ms_device by the event gaps"
            line["traversal_pairs_tested_per_s"] = (tested_total / k) / (stage["traverse"] * 1e-3)
            # roofline of the DOMINANT kernel -- whichever single launch of the step is longest, decided from the live
            # per-kernel times (HIP events riding on the kernels' own dispatch packets, on the library's stream):
            #   k_build_block:            hierarchy + refit + records of the 512-leaf blocks in one pass (cd_build.h), ALGORITHMIC
            #                             bytes 100 B/triangle (SURVEY.md 8d rows S3 + S4);
            #   k_descend_half:           the fp32 descent, 40 B/triangle (row S5: the tree read once).
            # The other one is reported beside it.  `traffic` = HBM bytes per launch from the rocprofv3 --pmc passes
            # (profiles/traffic.json, produced by tools/refresh_profiles.sh + tools/summarise_profiles.py).
            for k_ in kern:
                kern[k_] /= k
            traffic_of = {}
            l2_of = {}
            whole_traffic = whole_traffic_raw = None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath):
                tj = json.load(open(tpath))
                if tj.get("triangles") == nt:
                    whole_traffic = tj.get("whole_path_hbm_bytes_per_step"); whole_traffic_raw = tj.get("whole_path_hbm_bytes_per_step_raw")
                    for name, v in tj.get("whole_path_per_kernel", {}).items():
                        traffic_of[name] = v
                    l2_of["k_descend_half"] = tj.get("l2_hit_rate")

            def roof(symbol, label, ms, bytes_per_tri):
                ach = bytes_per_tri * nt / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
                tr = next((v for kk, v in traffic_of.items() if symbol in kk), None) or {}
                return {"bound": "hbm", "kernel": label, "kernel_symbol": symbol, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": ach / HBM_PEAK_GBS, "traffic": tr.get("hbm_bytes_per_step"), "traffic_raw": tr.get("hbm_bytes_per_step_raw"),
                        "traffic_source": "profiles/traffic.json (rocprofv3 --pmc passes of an earlier run of this command; NOT measured in this run): "
                                          "traffic = reads by request size (32 n32 + 64 n64 + 128 n128, TCC_EA0_RDREQ_*) + WRITE_SIZE; traffic_raw = FETCH_SIZE as reported "
                                          "(every request tallied at 64 B) + WRITE_SIZE; calibration: profiles/r03_experiments/fetch_size_calibration.csv",
                        "l2_hit_rate": l2_of.get(symbol),
                        "algorithmic_bytes_per_launch": bytes_per_tri * nt, "avg_launch_ms": ms}

            cands = [roof("k_descend_half", "k_descend_half (fp32 BVH descent, half traversal, groups of 64 leaves in the order hint's order; its candidates go to k_exact)", kern["descend"], TRAVERSAL_BYTES_PER_TRI),
                     roof("k_build_block", "k_build_block (Karras hierarchy + AABB refit + traversal records of the 512-leaf blocks, fused)",
                          kern["build_block"], BUILD_BYTES_PER_TRI)]
            # (with the order hint the two are within a microsecond of each other and would take turns from run to run: the descent -- the kernel every earlier round's
            #  line priced -- stays the line's `roofline` unless the other launch is more than 5 % longer; both are in the line either way)
            if cands[1]["avg_launch_ms"] > 1.05 * cands[0]["avg_launch_ms"]:
                cands.reverse()
            line["kernel_ms"] = kern
            line["kernel_ms_note"] = (f"descend_device_clock: live, EVERY step of the timed region, the kernel's own first-wave-start -> last-wave-end on the device wall clock (free); "
                                      f"descend, exact, build_block and total_collision_ms_device: HIP events riding on the kernels' dispatch packets (on the library's stream) in {prof_steps} steps "
                                      f"of the same loop right BEHIND the timed region -- a stamped kernel costs its step ~7 us of idle GPU, so none of the K timed steps carries one "
                                      f"(rounds 2-4 stamped every 8th timed step: ~1 us of the headline)")
            # what the descent's steps are filled with (VERDICT r02 asked for it beside the wait share in profiles/rNN/pmc_sq_per_kernel.csv): a wave-step is one
            # pass of a wave through a hop or a descent step, a node visit one lane's box test(s) in it
            st = engine.cd.stats()
            if st.wave_steps:
                line["descent_lane_use"] = {"node_visits_per_step": int(st.node_visits), "wave_steps_per_step": int(st.wave_steps),
                                            "lanes_busy_of_64": st.node_visits / float(st.wave_steps), "node_visits_per_query": st.node_visits / float(nt)}
            line["roofline"] = dict(cands[0])
            line["roofline"]["other_kernels"] = cands[1:]
            line["roofline"]["whole_path"] = {"bytes": TOTAL_BYTES_PER_TRI * nt, "achieved": TOTAL_BYTES_PER_TRI * nt / (dev_total * 1e-3) / 1e9,
                                              "frac": TOTAL_BYTES_PER_TRI * nt / (dev_total * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": whole_traffic, "traffic_raw": whole_traffic_raw}
            # The ORDER HINT (CD_OPT_ORDER_HINT, default on: the descent takes its groups of 64 leaves longest-first, by the previous step's wave times, which are
            # remembered per triangle; include/mi355cd.h).  A bench steps ONE mesh K times, so the hint it measures with is as good as a hint gets (-7 us); a mesh that
            # moves between steps keeps about two thirds of that (tools/hint_moving.py: sheets a quad apart per frame, -4 us).  What the step costs WITHOUT it is
            # measured too, right here, the same way:
            # (with --no-extras this leg is skipped too: the committed kernel trace is of `bench.py --no-extras` and holds the headline's launches only)
            engine.cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0); engine.cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 0)
            line["order_hint"] = {"default": "on", "what": "scheduling only: every group of 64 leaves is traversed in every step, in the order of the previous step's wave times (longest first, per "
                                                           "XCD); the times are remembered per triangle, so the hint follows a mesh that moves; pairs and counters do not depend on it (tests/test_cd_gpu.py)",
                                  "note": "the timed region steps one mesh K times: its hint is the best a hint can be (-7 us; a mesh moving a quad per frame: -4 us, tools/hint_moving.py); "
                                          "what the same step costs without it is measured beside it"}
            if args.no_order_hint:
                line["order_hint"]["default"] = "on; OFF in this run (--no-order-hint)"
            elif not args.no_extras:
                engine.cd.set_option(mi355cd.CD_OPT_ORDER_HINT, 0)
                for _ in range(10):
                    step()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                off_steps, off_clock = min(k, 100), 0.0
                for _ in range(off_steps):
                    step()
                    off_clock += engine.cd.fast_stats.ms_descend_clock
                torch.cuda.synchronize()
                off_ms = (time.perf_counter() - t1) * 1e3 / off_steps
                engine.cd.set_option(mi355cd.CD_OPT_ORDER_HINT, 1)
                line["order_hint"].update({"ms_per_step_without": off_ms, "pairs_tested_per_s_without": (tested_total / k) / (off_ms * 1e-3),
                                           "descend_device_clock_ms_without": off_clock / off_steps, "steps_without": off_steps})
            if not args.no_extras:
                # The headline runs in CD_FRAME_REFERENCE: the constants of morton.h:43-58, keys bit-identical to morton3D (the mesh is the reference's data set's shape).
                # What the same step costs in the frame the LIBRARY derives for the mesh (CD_FRAME_AUTO, the adaptive frame of cd_math.h, computed in one step and kept):
                engine.cd.set_morton_frame(mi355cd.CD_FRAME_AUTO)
                step()
                aoff, aspan, alay = engine.cd.keep_auto_frame()
                for _ in range(10):
                    step()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                a_steps, a_clock = min(k, 100), 0.0
                for _ in range(a_steps):
                    apairs, _, _ = step()
                    a_clock += engine.cd.fast_stats.ms_descend_clock
                torch.cuda.synchronize()
                a_ms = (time.perf_counter() - t1) * 1e3 / a_steps
                ast = engine.cd.stats()
                line["adaptive_frame"] = {"what": "the headline's mesh and step in CD_FRAME_AUTO's frame (computed once by the library, then kept) instead of the reference's constants",
                                          "ms_per_step": a_ms, "descend_device_clock_ms": a_clock / a_steps, "steps": a_steps, "sort_passes": int(ast.sort_passes),
                                          "node_visits_per_query": ast.node_visits / float(nt), "layout_word": hex(alay),
                                          "layout": {"axes_by_weight": [int(alay & 3), int((alay >> 2) & 3), int((alay >> 4) & 3)], "leading_bits_of_A": int((alay >> 8) & 255),
                                                     "pairs_AB": int((alay >> 16) & 255), "triples_ABC": int((alay >> 24) & 255)},
                                          "same_pairs_as_the_headline": bool(np.array_equal(np.sort(np.asarray(apairs, dtype=np.uint64)[:, 0] << np.uint64(32) | np.asarray(apairs, dtype=np.uint64)[:, 1]),
                                                                                            np.sort(last_pairs.astype(np.uint64)[:, 0] << np.uint64(32) | last_pairs.astype(np.uint64)[:, 1]))),
                                          "pairs_tested_equal": bool(int(ast.pairs_tested) == int(last_tested))}
                engine.cd.set_morton_frame(mi355cd.CD_FRAME_REFERENCE)
            if not args.no_parity:
                # checker leg: the LAST TIMED step's pair set + pairs-tested count against the oracle (one more CPU pass, with pairs)
                pc = parity_check(last_pairs, last_tested, verts, vidx)
                pc["reference_compiled_end_result"] = reference_compiled_check("cloth1M", last_pairs, last_tested) if (args.quads == 500 and args.traversal is None) else None
                line["parity_checked"] = pc["ok"] and pc["reference_compiled_end_result"] is not False
                line["parity"] = pc
            if not args.no_cpu_baseline:
                line["cpu_baseline"] = cpu_baseline(verts, vidx)
                line["speedup_vs_cpu_1core"] = line["value"] / line["cpu_baseline"]["value"]
                if "omp" in line["cpu_baseline"]:
                    line["speedup_vs_cpu_allcores"] = line["value"] / line["cpu_baseline"]["omp"]["value"]
                    line["speedup_note"] = (f"reported baselines, not targets: 1 core of the box, and the best OpenMP thread count "
                                            f"({line['cpu_baseline']['omp']['cores']}) this process' CPU share supports")
        else:
            line["config"]["last_step_rank0"] = info
            # ---- the N > 1 line carries what the N = 1 line carries (north_star wants 1 / 2 / 4 / 8 GPUs "as absolute numbers and as fraction of
            # the HBM roofline, next to the CPU path"): `roofline` = the WHOLE PATH of the whole job, SURVEY 8(d)'s 460 B a triangle x the triangles of all ranks over the
            # step's wall time, against N x 8 TB/s; inside it rank 0's local pipeline kernel by kernel (the kernels cd_multi_step enqueues for the rank's own shard are
            # cd_self_collide's: stamped here in a few untimed cd_self_collide steps on rank 0's shard -- no collective involved, the other ranks are in their checker leg).
            total_tris = int(nt) * world
            ach = TOTAL_BYTES_PER_TRI * total_tris / (ms_per_step * 1e-3) / 1e9
            engine.cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0); engine.cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 15)
            lk = {"build_block": 0.0, "descend": 0.0, "exact": 0.0}; lpipe = 0.0; lsteps = 10
            for _ in range(3):
                engine.cd.self_collide_into(pair_buf)
            for _ in range(lsteps):
                engine.cd.self_collide_into(pair_buf); st = engine.cd.stats()
                lk["build_block"] += st.ms_build_block / lsteps; lk["descend"] += st.ms_descend / lsteps; lk["exact"] += st.ms_exact / lsteps; lpipe += st.ms_pipeline / lsteps
            dom = max((("k_descend_half", lk["descend"], TRAVERSAL_BYTES_PER_TRI), ("k_build_block", lk["build_block"], BUILD_BYTES_PER_TRI)), key=lambda x: x[1])
            dom_ach = dom[2] * nt / (dom[1] * 1e-3) / 1e9 if dom[1] > 0 else 0.0
            line["kernel_ms"] = lk
            line["kernel_ms_note"] = (f"rank 0's LOCAL pipeline: HIP events on the kernels' dispatch packets in {lsteps} untimed cd_self_collide steps on rank 0's shard behind the timed region "
                                      "(the same kernels cd_multi_step enqueues for the rank's own triangles; the cross pass and the exchange are in phase_ms)")
            line["roofline"] = {"bound": "hbm", "kernel": "whole path, whole job: every rank's Morton keys + sort + hierarchy + refit + traversal of its shard and of the queries it received",
                                "achieved": ach, "peak": HBM_PEAK_GBS * world, "unit": "GB/s", "frac": ach / (HBM_PEAK_GBS * world), "traffic": None,
                                "algorithmic_bytes_per_step": TOTAL_BYTES_PER_TRI * total_tris, "triangles_all_ranks": total_tris, "avg_step_ms": ms_per_step,
                                "what": f"SURVEY 8(d): 460 B a triangle x {total_tris} triangles / ms_per_step (wall, max over ranks) against {world} x {HBM_PEAK_GBS / 1000:.0f} TB/s",
                                "rank0_local_pipeline": {"total_collision_ms_device": lpipe, "frac_of_one_gpu": (TOTAL_BYTES_PER_TRI * nt / (lpipe * 1e-3) / 1e9 / HBM_PEAK_GBS) if lpipe > 0 else None,
                                                         "dominant_kernel": {"kernel_symbol": dom[0], "avg_launch_ms": dom[1], "algorithmic_bytes_per_launch": dom[2] * nt,
                                                                             "achieved": dom_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom_ach / HBM_PEAK_GBS}}}
            engine.cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 15)
            if not args.no_cpu_baseline:
                # the CPU path beside it: the oracle on ONE shard (rank 0's), one core and the best OpenMP thread count -- a RATE (pairs tested / s), the same for one
                # shard or N of them on the same cores; the job's N shards one after the other take N x its total_collision_ms
                cb = cpu_baseline(verts, vidx)
                cb["sample"] = f"ONE shard of the job (rank 0's {int(nt)} triangles; the job has {world}): " + cb["sample"]
                cb["job_total_collision_ms_on_these_cores"] = cb["total_collision_ms"] * world
                line["cpu_baseline"] = cb
                line["speedup_vs_cpu_1core"] = line["value"] / cb["value"]
                if "omp" in cb:
                    line["speedup_vs_cpu_allcores"] = line["value"] / cb["omp"]["value"]
    if multi_path and not args.no_parity:
        # checker leg on EVERY rank: the oracle on this rank's mesh merged with its lower neighbour's (the only rank whose triangles have
        # smaller ids and overlap this one's); the rank owns exactly the oracle pairs whose larger id is its own (tri_contact.cuh:81)
        if self_peer:
            pc = parity_check(last_pairs[: len(last_pairs) // 2], None, verts, vidx, ids, own_id_range=(0, 1 << 32))
        else:
            mv, mt, mi_ = [verts], [vidx], [ids]
            if rank > 0:
                lv, lt, li, _ = synth.cloth_shard(rank - 1, args.quads)
                mv, mt, mi_ = [lv, verts], [lt, vidx + np.uint32(lv.shape[0])], [li, ids]
            allv = np.concatenate(mv); allt = np.concatenate(mt).astype(np.uint32); alli = np.concatenate(mi_).astype(np.uint32)
            cen = (allv[allt[:, 0]] + allv[allt[:, 1]] + allv[allt[:, 2]]) / 3
            off = cen.min(0); span = (cen.max(0) - off) * (1.0 + 1.0 / 1048576.0)
            pc = parity_check(last_pairs, None, allv, allt, alli, off=off, span=span, own_id_range=(int(ids[0]), int(ids[-1]) + 1))
        f = torch.tensor([1 if pc["ok"] else 0], dtype=torch.int32, device=device if backend == "nccl" else torch.device("cpu"))
        dist.all_reduce(f, op=dist.ReduceOp.MIN)
        if rank == 0:
            line["parity_checked"] = bool(int(f.item()))
            line["parity"] = dict(pc, note="rank 0's own check shown; parity_checked = MIN over ranks: every rank's pair list of the last timed step == "
                                           "the oracle's pairs (on the rank's mesh merged with its lower neighbour's) whose larger id the rank owns")
    if not multi_path and not args.no_extras and rank == 0:
        line["soup_1M"] = secondary_measurement(torch, "soup_1M")
        line["cloth_1M_double"] = secondary_measurement(torch, "cloth_1M_double")
        line["cloth_4M"] = size_measurement(torch, "cloth_4M")
        line["config4_merged_8M"] = size_measurement(torch, "config4_merged_8M")
        line["moving_mesh"] = moving_mesh_measurement(torch, args.quads)
        # (the headline steps ONE mesh at rest K times; what the same call takes on a mesh that moves is said where the step is described)
        line["config"]["step"] += (" -- on a mesh AT REST; the same step on the mesh with sheet B sliding a quad per frame (cd_update_vertices between steps, outside the timer): "
                                   "%.4f ms with the order hint, %.4f without (moving_mesh: other geometry every frame -- longer runs for the window sort, other candidates -- not colder caches)"
                                   % (line["moving_mesh"]["ms_per_step"], line["moving_mesh"]["ms_per_step_without_hint"]))
        line["from_obj"] = from_obj_measurement(torch, args.quads)
    if not args.no_ray and (backend == "nccl" or not multi_path):
        rtm = ray_tracer_measurement(rank=rank, world=world if multi_path else 1, dist=dist if multi_path else None, torch=torch, device=device, multi=multi_path)
        if rank == 0:
            line["ray_tracer"] = rtm
    if multi_path:
        # which transport ran, what the communicator saw, and where the time goes (max over ranks of each phase, from a
        # few extra untimed steps with HIP events at the phase boundaries)
        phase = None
        if ms is not None:
            ms.set_flags(mi355cd.CD_MULTI_TIMING | (mi355cd.CD_MULTI_SELF_PEER if self_peer else 0))
            acc = np.zeros(7)
            reps = 5
            for _ in range(reps):
                step()
                mi = last_info["mi"]
                acc += np.array([mi.ms_tree, mi.ms_allgather, mi.ms_pack, mi.ms_counts, mi.ms_exchange, mi.ms_local, mi.ms_cross])
            t = torch.tensor(acc / reps, dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            names = ["tree", "allgather", "pack", "counts", "exchange", "local", "cross"]
            phase = {k: float(v) for k, v in zip(names, t.tolist())}
            observed = int(last_info["mi"].world)
        else:
            observed = dist.get_world_size()
        if rank == 0:
            line["backend"] = ("rccl (C++: cd_multi_step of libmi355cd.so issues ncclAllGather / ncclSend / ncclRecv)" if ms is not None
                               else ("rccl through torch.distributed (Python orchestration, payloads on the device): FALLBACK" if multi_fallback
                                     else f"{backend} (REHEARSAL: Python orchestration, payloads staged through the host)"))
            if multi_fallback:
                line["multi_fallback"] = multi_fallback
            line["world_size_observed"] = observed
            line["phase_ms"] = phase
            line["phase_ms_note"] = ("max over ranks, from 5 extra untimed steps with events at the phase boundaries.  The phases are NOT additive: allgather / "
                                     "pack / counts / exchange / cross run on a second stream beside tree / local (allgather counts from the step's start, "
                                     "cross from the end of the exchange to the end of the pass over the received queries)")
    if rank == 0:
        print(json.dumps(line))
    if ms is not None:
        ms.close()
    engine.close()
    pair_buf = None
    host_pairs.close()
    if multi_path:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
