"""cd_multi_step with world > 1 on ONE GPU: every rank is a host thread of this process with its own cd_ctx, and
libmi355cd.so talks to tests/loopback_rccl (an in-process stand-in for librccl.so, selected through MI355CD_RCCL_LIBRARY)
instead of RCCL, which refuses two ranks on one device.  Everything else -- the pack launch for all peers, the count
matrix, the collective capacity growth, slab and receive offsets, both traversal passes -- is the product code path.
Run as a child process by tests/test_cd_gpu.py (the library choice is per process).

usage: multi_loopback_driver.py QUADS QCAP STEPS X0,X1,... [INJECT_RANK:INJECT_STEP[:alloc]]    (rank r's object sits at x = Xr * 2.88)
QUADS may be a comma list, one value per rank (shards of unequal size: the default capacity nt / 8 + 1024 then differs per rank
and must be agreed at creation).  INJECT: that rank sets CD_MULTI_INJECT_FAILURE before that step -- EVERY rank must return an
error from that step (the rank itself CD_ERR_INJECTED, the others CD_ERR_PEER), none may block, and the following steps must be
right again.  With ":alloc" the rank sets CD_MULTI_INJECT_ALLOC_FAILURE instead: its next allocation of the send / receive slabs
fails -- give a QCAP small enough that the step has to grow them -- the rank returns the allocation's error, the others CD_ERR_PEER.
INJECT "R:create": rank R's cd_multi_create meets an allocation failure (the flag at creation) -- it still joins creation's agreement on the capacities, with 0:
every rank's creation returns (R the allocation's error, the others CD_ERR_PEER), nobody waits in the all-gather.  INJECT "R:S:orphan": before step S rank R's
context is destroyed under its cd_multi; its cd_multi_step joins the step's collectives with an empty box and CD_ERR_ORDER in its status word: R returns
CD_ERR_ORDER, the others CD_ERR_PEER, nobody blocks (give STEPS = S + 1: the orphan stays one).
Checks, and exits non-zero on failure: union of all ranks' pairs == oracle on the merged mesh, no duplicates, summed
pairs_tested == the single tree's, sent/received totals consistent across ranks, peers as the root boxes say."""
import json
import os
import sys
import threading

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
os.environ["MI355CD_RCCL_LIBRARY"] = os.path.join(HERE, "loopback_rccl", "librccl_loopback.so")
import torch  # noqa: F401,E402  (its HIP runtime first, see conftest.py)
sys.path[:0] = [os.path.join(ROOT, "gpu-computing-course_amd", "pyhost"), HERE]
import numpy as np  # noqa: E402
import mi355cd  # noqa: E402
import mi355_multi as multi  # noqa: E402
import mi355_synth as synth  # noqa: E402
import oracle  # noqa: E402


def main():
    qcap, steps = int(sys.argv[2]), int(sys.argv[3])
    xs = [float(x) for x in sys.argv[4].split(",")]
    W = len(xs)
    quads_of = [int(q) for q in sys.argv[1].split(",")]
    quads_of = quads_of * W if len(quads_of) == 1 else quads_of
    inj = sys.argv[5].split(":") if len(sys.argv) > 5 else None
    inject_create = int(inj[0]) if inj and inj[1] == "create" else None
    inject = (int(inj[0]), int(inj[1])) if inj and inject_create is None else None
    inject_alloc = bool(inj) and len(inj) > 2 and inj[2] == "alloc"
    orphan = bool(inj) and len(inj) > 2 and inj[2] == "orphan"
    width = 2.88
    shards, vbase, tbase = [], 0, 0
    for r in range(W):
        v, t = synth.cloth_pair(quads_of[r], x_offset=xs[r] * width)
        ids = (np.arange(t.shape[0], dtype=np.uint64) + tbase).astype(np.uint32)
        if r % 2:                                   # odd ranks hold the larger IDs first: the ID rule must not depend on rank order
            ids = ids[::-1].copy()
        shards.append((v, t, ids, vbase))
        vbase += v.shape[0]; tbase += t.shape[0]
    cds = []
    for v, t, ids, vb in shards:
        cd = mi355cd.CollisionDetector(v, t, ids)
        cd.set_morton_frame(mi355cd.CD_FRAME_AUTO)
        cd.set_vertex_id_base(vb)
        cds.append(cd)
    uid = mi355cd.multi_unique_id()                 # also loads the loopback library once, before the threads start
    assert uid[8:16] == b"loopback", "libmi355cd.so did not load the loopback library"
    results, errors = [None] * W, [None] * W

    create_rcs = [0] * W

    def rank_main(r):
        try:
            try:
                ms_ = mi355cd.MultiStep(cds[r], uid, r, W, query_cap_per_peer=qcap,
                                        flags=mi355cd.CD_MULTI_TIMING | (mi355cd.CD_MULTI_INJECT_ALLOC_FAILURE if inject_create == r else 0))
            except mi355cd.CdError as e:
                create_rcs[r] = e.rc
                results[r] = []
                return
            with ms_ as ms:
                out = []
                for it in range(steps):
                    if inject and inject == (r, it) and orphan:
                        cds[r].close()                      # cd_destroy under the cd_multi: the step below must still join its peers' collectives
                    elif inject and inject == (r, it):
                        ms.set_flags(mi355cd.CD_MULTI_TIMING | (mi355cd.CD_MULTI_INJECT_ALLOC_FAILURE if inject_alloc else mi355cd.CD_MULTI_INJECT_FAILURE))
                    try:
                        pairs, n, rc, info = ms.step(cap=1 << 21)
                        out.append((pairs.copy(), n, rc, {k: getattr(info, k) for k, _ in info._fields_}))
                    except mi355cd.CdError as e:
                        out.append((np.zeros((0, 2), dtype=np.uint32), 0, e.rc, {}))
                results[r] = out
        except BaseException as e:                  # noqa: BLE001
            errors[r] = repr(e)

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(W)]
    for t in th:
        t.start()
    for t in th:
        t.join(240)
    hung = [r for r, t in enumerate(th) if t.is_alive()]
    if hung or any(errors):
        print(json.dumps({"ok": False, "hung": hung, "errors": errors}))
        os._exit(2)

    if inject_create is not None:
        want = [-2 if r == inject_create else mi355cd.CD_ERR_PEER for r in range(W)]          # -2: -hipErrorOutOfMemory
        ok = create_rcs == want
        print(json.dumps({"ok": ok, "create_rcs": create_rcs, "want": want}))
        for cd in cds:
            cd.close()
        sys.exit(0 if ok else 1)
    if orphan:
        it = inject[1]
        rcs = [results[r][it][2] for r in range(W)]
        want = [mi355cd.CD_ERR_ORDER if r == inject[0] else mi355cd.CD_ERR_PEER for r in range(W)]
        before_ok = all(results[r][k][2] == 0 for r in range(W) for k in range(it))
        ok = rcs == want and before_ok
        print(json.dumps({"ok": ok, "orphan_step_rcs": rcs, "want": want, "steps_before_ok": before_ok}))
        for r, cd in enumerate(cds):
            if r != inject[0]:
                cd.close()
        sys.exit(0 if ok else 1)
    roots = [cd.root_box() for cd in cds]
    want_peers = [sum(1 for s in range(W) if s != r and multi.boxes_overlap(roots[r], roots[s])) for r in range(W)]
    verts = np.concatenate([s[0] for s in shards]); vidx = np.concatenate([s[1] + np.uint32(s[3]) for s in shards])
    ids = np.concatenate([s[2] for s in shards])
    cen = (verts[vidx[:, 0]] + verts[vidx[:, 1]] + verts[vidx[:, 2]]) / 3
    ref = oracle.pipeline(verts, vidx, ids, off=cen.min(0), span=(cen.max(0) - cen.min(0)) * (1 + 2.0 ** -20))
    want = oracle.pair_set(ref["pairs"])
    summary = {"ok": True, "world": W, "want_pairs": int(len(want)), "want_peers": want_peers, "steps": []}
    for it in range(steps):
        if inject and it == inject[1]:
            rcs = [results[r][it][2] for r in range(W)]
            want_rcs = [(-2 if inject_alloc else mi355cd.CD_ERR_INJECTED) if r == inject[0] else mi355cd.CD_ERR_PEER for r in range(W)]     # -2: -hipErrorOutOfMemory
            summary["steps"].append({"injected": True, "rcs": rcs, "checks": {"all_ranks_failed_together": rcs == want_rcs}})
            summary["ok"] = summary["ok"] and rcs == want_rcs
            continue
        got = np.concatenate([results[r][it][0] for r in range(W)], axis=0)
        infos = [results[r][it][3] for r in range(W)]
        gs = oracle.pair_set(got)
        checks = {
            "rc_ok": all(results[r][it][2] == 0 for r in range(W)),
            "no_duplicates": len(gs) == len(np.unique(gs)),
            "pair_set_equals_oracle": bool(np.array_equal(gs, want)),
            "pairs_tested_equals_single_tree": sum(i["pairs_tested"] for i in infos) == ref["stats"].pairs_tested,
            "sent_equals_received": sum(i["sent_queries"] for i in infos) == sum(i["recv_queries"] for i in infos),
            "peers_as_root_boxes_say": [i["n_peers"] for i in infos] == want_peers,
            "world_rank": all(i["world"] == W and i["rank"] == r for r, i in enumerate(infos)),
            "same_attempts_everywhere": len({i["attempts"] for i in infos}) == 1,
            "same_capacity_everywhere": len({i["query_cap"] for i in infos}) == 1,
            "n_counts": all(results[r][it][1] == infos[r]["local_pairs"] + infos[r]["cross_pairs"] for r in range(W)),
        }
        summary["steps"].append({"checks": checks, "attempts": infos[0]["attempts"], "query_cap": infos[0]["query_cap"],
                                 "host_syncs": [i["host_syncs"] for i in infos], "sent": [i["sent_queries"] for i in infos],
                                 "recv": [i["recv_queries"] for i in infos], "local": [i["local_pairs"] for i in infos],
                                 "cross": [i["cross_pairs"] for i in infos], "got_pairs": int(len(gs))})
        summary["ok"] = summary["ok"] and all(checks.values())
    for cd in cds:
        cd.close()
    print(json.dumps(summary))
    sys.exit(0 if summary["ok"] else 1)


if __name__ == "__main__":
    main()
