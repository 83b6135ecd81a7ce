"""Stress of the polled completion (CD_OPT_POLL): is every pair in host memory when the sequence word is?  Debug key 110 makes the library fill the
pair area with 0xff before each step and scan it the moment it sees the word (key 111: steps with a pair still 0xff, key 112: steps that fell back
to the stream synchronise).  Meshes from ~1 k to > 32 768 pairs (the most the report kernel posts), ordinary and pinned buffers, STEPS steps each;
every step's pair set is also compared with the first (synchronised) step's.  With LOAD = 1 a second thread keeps the host link busy in both
directions (64 MB torch copies on another stream) while the steps run.  usage: poll_stress.py [STEPS [LOAD]]   GPU only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gpu-computing-course_amd", "pyhost"), os.path.join(ROOT, "tests")]
import numpy as np, mi355cd, mi355_synth as synth, oracle
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
load = len(sys.argv) > 2 and sys.argv[2] == "1"
stop = False
if load:
    import threading, torch
    def hammer():
        h = torch.empty(64 << 20, dtype=torch.uint8).pin_memory(); h2 = torch.empty(64 << 20, dtype=torch.uint8).pin_memory()
        d = torch.empty(64 << 20, dtype=torch.uint8, device="cuda"); d2 = torch.zeros(64 << 20, dtype=torch.uint8, device="cuda")
        st = torch.cuda.Stream()
        k = 0
        with torch.cuda.stream(st):
            while not stop:
                d.copy_(h, non_blocking=True); h2.copy_(d2, non_blocking=True); st.synchronize(); k += 1
        print(f"(background copies: {k} x 2 x 64 MB)")
    th = threading.Thread(target=hammer); th.start()
cases = [("cloth120", synth.cloth_pair(120)), ("cloth122", synth.cloth_pair(122)), ("soup20k", synth.soup(20_000, 0.08, 5)), ("soup60k", synth.soup(60_000, 0.08, 21)),
         ("cloth500", synth.cloth_pair(500)), ("soup300k", synth.soup(300_000, 0.03, 9))]
bad = 0
for name, (verts, vidx) in cases:
    with mi355cd.CollisionDetector(verts, vidx) as cd, mi355cd.HostPairs(1 << 20) as hp:
        plain = np.empty((1 << 20, 2), dtype=np.uint32)
        cd.set_option(mi355cd.CD_OPT_STAGE_TIMING, 0); cd.set_option(mi355cd.CD_OPT_KERNEL_STAMPS, 0); cd.set_option(mi355cd.CD_OPT_POLL, 0)
        n0, rc = cd.self_collide_into(plain)
        want = oracle.pair_set(plain[:n0].copy())
        cd.set_option(mi355cd.CD_OPT_POLL, 1); cd.debug_set(mi355cd.CD_DBG_POLL_SCAN, 1)
        mism = 0
        for it in range(steps):
            buf = plain if it % 2 == 0 else hp.array
            n, rc = cd.self_collide_into(buf)
            if rc != 0 or n != n0 or not np.array_equal(oracle.pair_set(buf[:n]), want): mism += 1
        stale = cd.debug_get(mi355cd.CD_DBG_GET_POLL_STALE); fb = cd.debug_get(mi355cd.CD_DBG_GET_POLL_FALLBACKS)
        why = cd.debug_get(mi355cd.CD_DBG_GET_POLL_FB_WHY); busy, late_word, lost = why & 0xffff, (why >> 16) & 0xffff, (why >> 32) & 0xffff
        print(f"{name}: {n0} pairs, {steps} polled steps: pair-set mismatches {mism}, steps with a pair not yet in host memory {stale}, fallbacks to the stream {fb} "
              f"(stream still busy at the 20 ms time-out {busy}, word late in flight {late_word}, word LOST {lost}), longest ordinary polled wait {cd.debug_get(mi355cd.CD_DBG_GET_POLL_MAX_WAIT_US)} us", flush=True)
        bad += mism + stale + lost
stop = True
if load: th.join()
print("BAD" if bad else "clean")
sys.exit(1 if bad else 0)
