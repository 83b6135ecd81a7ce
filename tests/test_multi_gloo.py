"""N > 1 path on CPU: the multi-GPU orchestration (mi355_multi.collide_step) under torch.distributed
gloo, world_size 2 and 3, with an ORACLE-backed stand-in engine injected by this test (the product
engine is HipEngine over libmi355cd.so; the orchestration code under test is the shipped one).
Property: the union of the per-rank pair lists equals the single-process oracle result on the merged mesh,
every cross pair is reported exactly once, and the exchange only happens between overlapping ranks."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import mi355_multi as multi
import mi355_synth as synth
import mi355cd
import oracle

QUADS = 24


class OracleEngine:
    """Test double with HipEngine's interface, computing with the CPU oracle on CPU tensors."""

    def __init__(self, verts, vidx, ids, vertex_id_base):
        self.verts, self.vidx, self.ids, self.vbase = verts, vidx, ids, np.uint32(vertex_id_base)
        cen = (verts[vidx[:, 0]] + verts[vidx[:, 1]] + verts[vidx[:, 2]]) / 3
        self.off = cen.min(0); self.span = (cen.max(0) - self.off) * (1.0 + 1.0 / 1048576.0)
        self.tree = None

    def self_collide(self, cap):
        self.tree = oracle.pipeline(self.verts, self.vidx, self.ids, off=self.off, span=self.span)
        return self.tree["pairs"], self.tree["stats"].n_pairs, self.tree["stats"].pairs_tested

    def build_tree(self):
        self.tree = oracle.pipeline(self.verts, self.vidx, self.ids, off=self.off, span=self.span)

    def find_collisions(self, cap):
        return self.tree["pairs"], self.tree["stats"].n_pairs, self.tree["stats"].pairs_tested

    def root_box(self):
        return self.tree["boxes"][0].copy()

    def pack_queries(self, box):
        n = self.vidx.shape[0]
        t = self.tree
        leaf_boxes = t["boxes"][n - 1:]
        keep = [j for j in range(n) if multi.boxes_overlap(leaf_boxes[j], box)]
        q = np.zeros(len(keep), dtype=mi355cd.QUERY_DTYPE)
        tri = t["perm"][keep]
        q["v"] = self.verts[self.vidx[tri]].reshape(-1, 9)
        q["id"] = self.ids[tri]
        q["vidx"] = self.vidx[tri] + self.vbase
        return torch.from_numpy(q.view(np.uint8).copy())

    def empty_queries(self, nbytes=0):
        return torch.empty(nbytes, dtype=torch.uint8)

    def find_collisions_queries(self, qbuf, cap):
        q = qbuf.numpy().view(mi355cd.QUERY_DTYPE)
        if q.shape[0] == 0:
            return np.zeros((0, 2), dtype=np.uint32), 0, 0
        t = self.tree
        pairs, st = oracle.find_collisions_queries(q, self.verts, self.vidx, t["perm"], t["left"], t["right"], t["boxes"], self.ids,
                                                   vbase=self.vbase)
        return pairs, st.n_pairs, st.pairs_tested


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, overlap, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    verts, vidx, ids, vbase = synth.cloth_shard(rank, QUADS, overlap=overlap)
    eng = OracleEngine(verts, vidx, ids, vbase)
    pairs, tested, info = multi.collide_step(eng, dist, rank, world, cap=1 << 20)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), pairs=pairs, tested=tested, peers=np.array(info["peers"], dtype=np.int64),
             sent=info["sent_queries"], recv=info["recv_queries"], cross=info["cross_pairs"], local=info["local_pairs"])
    dist.barrier()
    dist.destroy_process_group()


def _merged(world, overlap):
    vs, ts, ids = [], [], []
    for r in range(world):
        v, t, i, vb = synth.cloth_shard(r, QUADS, overlap=overlap)
        vs.append(v); ts.append(t + np.uint32(vb)); ids.append(i)
    return np.concatenate(vs), np.concatenate(ts), np.concatenate(ids)


@pytest.mark.parametrize("world,overlap", [(2, 0.10), (3, 0.10), (2, -0.05)])
def test_sharded_step_equals_single_process(tmp_path, world, overlap):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, overlap, str(tmp_path)), nprocs=world, join=True)
    res = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    got = np.concatenate([r["pairs"] for r in res], axis=0)
    verts, vidx, ids = _merged(world, overlap)
    cen = (verts[vidx[:, 0]] + verts[vidx[:, 1]] + verts[vidx[:, 2]]) / 3
    ref = oracle.pipeline(verts, vidx, ids, off=cen.min(0), span=(cen.max(0) - cen.min(0)) * (1 + 2.0 ** -20))
    gs = oracle.pair_set(got)
    assert len(gs) == len(np.unique(gs)), "a cross pair was reported twice"
    assert np.array_equal(gs, oracle.pair_set(ref["pairs"]))
    if overlap > 0:
        assert sum(int(r["cross"]) for r in res) > 0
        assert all(len(r["peers"]) >= 1 for r in res)
        if world == 3:
            assert res[0]["peers"].tolist() == [1] and res[1]["peers"].tolist() == [0, 2] and res[2]["peers"].tolist() == [1]
    else:       # objects apart: the step degenerates to the 48-byte all-gather, nothing exchanged
        assert all(len(r["peers"]) == 0 and int(r["sent"]) == 0 and int(r["recv"]) == 0 for r in res)
    assert sum(int(r["sent"]) for r in res) == sum(int(r["recv"]) for r in res)


class FailingEngine(OracleEngine):
    """Raises where a real engine would on a capacity overflow / CdError -- on one rank only."""

    def __init__(self, *a, fail_in):
        super().__init__(*a)
        self.fail_in = fail_in

    def pack_queries(self, box):
        if self.fail_in == "pack":
            raise RuntimeError("injected pack failure")
        return super().pack_queries(box)

    def find_collisions(self, cap):
        if self.fail_in == "local":
            raise RuntimeError("injected traversal failure")
        return super().find_collisions(cap)


def _failing_worker(rank, world, port, fail_in, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    verts, vidx, ids, vbase = synth.cloth_shard(rank, QUADS, overlap=0.10)
    eng = FailingEngine(verts, vidx, ids, vbase, fail_in=fail_in if rank == 1 else None)
    try:
        multi.collide_step(eng, dist, rank, world, cap=1 << 20)
        msg = "no error"
    except RuntimeError as e:
        msg = str(e)
    open(os.path.join(out_dir, f"rank{rank}.txt"), "w").write(msg)
    dist.barrier()                                   # reachable only if NO rank is stuck in a collective of the step
    dist.destroy_process_group()


@pytest.mark.timeout(120)
@pytest.mark.parametrize("fail_in", ["pack", "local"])
def test_failure_on_one_rank_is_raised_on_every_rank(tmp_path, fail_in):
    """A rank-local failure between collectives must not leave the peers blocked in the next collective: the failing
    rank keeps taking part with empty payloads, and every rank raises after the error flag's all-reduce."""
    port = _free_port()
    mp.spawn(_failing_worker, args=(2, port, fail_in, str(tmp_path)), nprocs=2, join=True)
    m0 = (tmp_path / "rank0.txt").read_text(); m1 = (tmp_path / "rank1.txt").read_text()
    assert "failed on another rank" in m0
    assert "injected" in m1 and "this rank" in m1


def test_boxes_overlap_is_the_reference_predicate():
    a = np.array([0, 1, 0, 1, 0, 1.0])
    assert multi.boxes_overlap(a, np.array([0.5, 2, 0.5, 2, 0.5, 2.0]))
    assert not multi.boxes_overlap(a, np.array([1.0, 2, 0, 1, 0, 1.0]))        # touching faces: strict
    assert not multi.boxes_overlap(a, np.array([2.0, 3, 0, 1, 0, 1.0]))
    for _ in range(200):
        b = np.sort(np.random.rand(3, 2), axis=1).ravel(); c = np.sort(np.random.rand(3, 2), axis=1).ravel()
        assert multi.boxes_overlap(b, c) == bool(oracle.lib().orc_box_overlap(oracle._p(b), oracle._p(c)))
