#!/usr/bin/env python3
"""BASELINE config 4, one neighbour pair on one GPU: rank 0's leaves that overlap rank 1's root box, packed by k_pack_queries
(the kernel the multi-GPU step launches once for all peers), and rank 1's pass over them.  Run under rocprofv3 --kernel-trace
--stats for the kernel times; prints the counts and wall times of the blocking calls.  GPU only."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gpu-computing-course_amd", "pyhost")]
import torch
import mi355_synth as synth, mi355cd, mi355_multi as multi
dev = torch.device("cuda", 0)
eng = [multi.HipEngine(*synth.cloth_shard(r, 500)[:3], dev, vertex_id_base=synth.cloth_shard(r, 500)[3]) for r in (0, 1)]
for e in eng:
    e.build_tree()
roots = [e.root_box() for e in eng]
buf = torch.empty(300_000 * 88, dtype=torch.uint8, device=dev)
K = 50
n, rc = eng[0].cd.pack_queries_into(roots[1], buf.data_ptr(), 300_000)
t0 = time.perf_counter()
for _ in range(K): n, rc = eng[0].cd.pack_queries_into(roots[1], buf.data_ptr(), 300_000)
tp = (time.perf_counter() - t0) / K
pairs, npairs, rc2 = eng[1].cd.find_collisions_queries(buf.data_ptr(), n, 1 << 20)
t0 = time.perf_counter()
for _ in range(K): pairs, npairs, rc2 = eng[1].cd.find_collisions_queries(buf.data_ptr(), n, 1 << 20)
tx = (time.perf_counter() - t0) / K
print(f"queries packed for the neighbour: {n} of {eng[0].nt} leaves; cd_pack_queries {tp*1e6:.0f} us per blocking call; "
      f"cd_find_collisions_queries {tx*1e6:.0f} us per blocking call, {npairs} cross pairs, {eng[1].cd.stats().pairs_tested} pairs tested")
