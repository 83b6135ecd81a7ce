"""Multi-GPU collision detection: one process per GPU, triangles sharded by object.

The reference is single-GPU; this orchestration is the new work the north star defines
(SURVEY.md 8e).  Per step, on every rank r:

  1. local self-collision of the rank's own object(s)                      (no communication)
  2. all-gather of the per-rank root AABB (6 doubles)                      (RCCL over xGMI; 48 B/rank)
  3. for every peer s whose root AABB strictly overlaps r's (box.cuh:40-43): compact the local leaves
     that overlap root(s) into cd_query records and exchange them         (all-to-all, variable sizes)
  4. traverse the received queries against the local tree; a cross pair {a, b} with a.ID < b.ID is
     reported exactly once, by the rank that owns b (tri_contact.cuh:81 ID rule).

Triangle IDs and vertex indices must be GLOBAL so the ID rule and neighborCount (triangle.cuh:18-30)
stay meaningful across ranks.

The engine (what actually computes) is passed in: the product engine is `HipEngine` over
libmi355cd.so; tests inject a CPU stand-in to exercise this orchestration under gloo.  There is no
fallback here: HipEngine raises without a GPU.
"""
from __future__ import annotations

import numpy as np

QUERY_BYTES = 88


def boxes_overlap(a, b) -> bool:
    """checkBoxOverlap, box.cuh:40-43 (strict, product form) on {x1,x2,y1,y2,z1,z2}."""
    return bool((a[0] - b[1]) * (b[0] - a[1]) > 0 and (a[2] - b[3]) * (b[2] - a[3]) > 0 and (a[4] - b[5]) * (b[4] - a[5]) > 0)


class HipEngine:
    """Rank-local compute on one MI355X through the C ABI (mi355cd)."""

    def __init__(self, verts, vidx, ids, device, frame=None, vertex_id_base=0):
        import torch
        import mi355cd
        self.torch = torch
        self.device = device
        torch.cuda.set_device(device)                 # the library allocates on the current HIP device
        self.cd = mi355cd.CollisionDetector(verts, vidx, ids)
        self.cd.set_morton_frame(mi355cd.CD_FRAME_AUTO if frame is None else frame)
        self.cd.set_vertex_id_base(int(vertex_id_base))
        self.nt = vidx.shape[0]

    def self_collide(self, cap):
        pairs, n, rc = self.cd.self_collide(cap)
        st = self.cd.stats()
        return pairs, n, st.pairs_tested

    def build_tree(self):
        """Morton keys, sort, hierarchy, refit -- everything the exchange needs -- without the traversal."""
        self.cd.build_tree()

    def find_collisions(self, cap):
        pairs, n, rc = self.cd.find_collisions(cap)
        return pairs, n, self.cd.stats().pairs_tested

    def root_box(self):
        return self.cd.root_box()

    def pack_queries(self, box):
        """-> uint8 device tensor holding the cd_query records of local leaves overlapping `box`."""
        torch = self.torch
        cap = max(1024, self.nt // 8)
        while True:
            buf = torch.empty(cap * QUERY_BYTES, dtype=torch.uint8, device=self.device)
            n, rc = self.cd.pack_queries_into(box, buf.data_ptr(), cap)
            if n <= cap:
                return buf[: n * QUERY_BYTES]
            cap = int(n)

    def empty_queries(self, nbytes=0):
        return self.torch.empty(nbytes, dtype=self.torch.uint8, device=self.device)

    def find_collisions_queries(self, qbuf, cap):
        nq = qbuf.numel() // QUERY_BYTES
        if nq == 0:
            return np.zeros((0, 2), dtype=np.uint32), 0, 0
        self.torch.cuda.synchronize()                 # the collective that filled qbuf ran on torch's stream
        pairs, n, rc = self.cd.find_collisions_queries(qbuf.data_ptr(), nq, cap)
        return pairs, n, self.cd.stats().pairs_tested

    def close(self):
        self.cd.close()


def collide_step(engine, dist, rank, world, cap=1 << 22, comm_device=None):
    """One multi-GPU step, orchestrated from Python over torch.distributed.  Returns (pairs ndarray[k,2] found by THIS
    rank, pairs_tested by this rank, info dict).  `dist` is torch.distributed.

    This is the REHEARSAL / test orchestration (gloo on CPU with a stand-in engine, or gloo with ranks sharing one GPU):
    the measured multi-GPU path is cd_multi_step in libmi355cd.so, which issues the RCCL calls itself (bench.py uses it
    whenever the backend is nccl).  comm_device: where the collectives' tensors live; a gloo rehearsal passes "cpu".

    Failure is COLLECTIVE: a rank whose engine raises (capacity overflow, CdError ...) keeps taking part in every
    collective of the step with empty payloads, and an all-reduce(MAX) of an error flag at the end makes every rank
    raise together -- no rank is left blocked in a collective its peer has abandoned."""
    import torch
    info = {"local_pairs": 0, "cross_pairs": 0, "sent_queries": 0, "recv_queries": 0, "peers": []}
    err = []

    def guard(fn, default):
        if err:
            return default
        try:
            return fn()
        except Exception as e:                              # noqa: BLE001 -- reported collectively below
            err.append(e)
            return default

    empty_pairs = (np.zeros((0, 2), dtype=np.uint32), 0, 0)
    if world == 1:
        local_pairs, n_local, tested = engine.self_collide(cap)
        if n_local > cap:
            raise RuntimeError(f"pair capacity {cap} too small for {n_local} local pairs")
        info["local_pairs"] = int(n_local)
        return local_pairs, int(tested), info

    # 1a. the local tree (no traversal yet: the exchange below overlaps with it)
    guard(engine.build_tree, None)

    # 2. all-gather of root AABBs (a failed rank contributes an empty box: it strictly overlaps nothing)
    edev = engine.empty_queries().device
    dev = edev if comm_device is None else torch.device(comm_device)
    mine = torch.from_numpy(np.ascontiguousarray(guard(engine.root_box, np.zeros(6)))).to(dev)
    roots = torch.empty(world * 6, dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(roots, mine)
    roots = roots.cpu().numpy().reshape(world, 6)
    peers = [s for s in range(world) if s != rank and boxes_overlap(roots[rank], roots[s])]
    info["peers"] = peers

    # 3. query exchange: counts first, then the records (variable-size all-to-all), started asynchronously
    send = [guard(lambda s=s: engine.pack_queries(roots[s]), engine.empty_queries()) if s in peers else engine.empty_queries() for s in range(world)]
    if err:
        send = [engine.empty_queries() for _ in range(world)]
    send_counts = torch.tensor([t.numel() for t in send], dtype=torch.int64, device=dev)
    recv_counts = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_to_all_single(recv_counts, send_counts)
    recv_counts = recv_counts.cpu().tolist()
    sendbuf = (torch.cat(send) if send else engine.empty_queries()).to(dev)
    recvbuf = torch.empty(int(sum(recv_counts)), dtype=torch.uint8, device=dev)
    if sendbuf.is_cuda:
        torch.cuda.synchronize()                     # pack_queries wrote sendbuf on the library's stream (already synchronised); be explicit
    work = dist.all_to_all_single(recvbuf, sendbuf, output_split_sizes=recv_counts, input_split_sizes=[t.numel() for t in send],
                                  async_op=True)
    info["sent_queries"] = int(sendbuf.numel() // QUERY_BYTES)
    info["recv_queries"] = int(recvbuf.numel() // QUERY_BYTES)

    # 1b. local traversal while the records travel
    local_pairs, n_local, tested = guard(lambda: engine.find_collisions(cap), empty_pairs)
    if n_local > cap and not err:
        err.append(RuntimeError(f"pair capacity {cap} too small for {n_local} local pairs"))
    info["local_pairs"] = int(n_local)
    work.wait()
    recvbuf = recvbuf.to(edev)

    # 4. received queries against the local tree
    cross_pairs, n_cross, tested_cross = guard(lambda: engine.find_collisions_queries(recvbuf, cap), empty_pairs)
    if n_cross > cap and not err:
        err.append(RuntimeError(f"pair capacity {cap} too small for {n_cross} cross pairs"))
    info["cross_pairs"] = int(n_cross)

    # 5. every rank learns whether ANY rank failed, and all raise together
    flag = torch.tensor([1 if err else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    if int(flag.item()):
        raise RuntimeError(f"collide_step failed on at least one rank (this rank: {err[0]!r})" if err else
                           "collide_step failed on another rank") from (err[0] if err else None)
    pairs = np.concatenate([local_pairs, cross_pairs], axis=0) if n_cross else local_pairs
    return pairs, int(tested) + int(tested_cross), info
